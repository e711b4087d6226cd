"""Shared fixtures for the GPU parity tests: synthetic models on disk -> oracle model + product model."""
import dataclasses
import os

import numpy as np

from conftest import pk
from parakeet_cpp_amd import synth

_CACHE = {}


def make_pair(tmpdir, cfg, seed=42, with_vocab=False):
    """Returns (weights dict, oracle Model, product capi.Model on GPU 0)."""
    import oracle
    from parakeet_cpp_amd import capi
    key = (cfg.name, cfg.num_layers, cfg.hidden_size, seed, cfg.head, cfg.num_lstm_layers)
    if key in _CACHE:
        return _CACHE[key]
    W = synth.synth_weights(cfg, seed=seed)
    wp = os.path.join(str(tmpdir), f"{cfg.name}_{cfg.num_layers}_{seed}.safetensors")
    synth.save_weights(wp, W)
    vp = None
    if with_vocab:
        vp = os.path.join(str(tmpdir), f"{cfg.name}_vocab.txt")
        synth.save_vocab(vp, synth.synth_vocab(cfg.vocab_size - 1))
    gm = capi.Model(wp, cfg, vocab_path=vp, device=0)
    om = oracle.Model(cfg, W)
    _CACHE[key] = (W, om, gm)
    return _CACHE[key]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bits_equal(got, want, what):
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    if not np.array_equal(bits(got), bits(want)):
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        bad = np.argwhere(bits(got) != bits(want))
        raise AssertionError(f"{what}: {len(bad)} of {got.size} elements differ; max abs diff {diff.max():.3e}; "
                             f"first at {tuple(bad[0])}: got {got[tuple(bad[0])]!r} want {want[tuple(bad[0])]!r}")


def tiny(**kw):
    return pk.make_tiny_config(**kw)


def one_layer_110m(n=1):
    return dataclasses.replace(pk.make_110m_config(), num_layers=n, name=f"110m-{n}L")
