"""GPU parity at BASELINE configs[2] shapes: tdt-600m (128 mel bins, d = 1024, 8 heads of 128, ffn 4096, vocab 8193,
2 LSTM layers), 30 s clips (480000 samples -> 3001 mel frames -> T = 376).  fp32 (>= the bf16 BASELINE names).
 * oracle-checked, bit for bit: a 2-layer cut of the model on 30 s clips (the oracle is a scalar CPU program: 24 layers
   x 30 s would take minutes) -- mel, subsampling, both blocks, TDT ids / frames;
 * the full 24-layer model on a batch of 32 x 30 s through the resident pipeline: size-independent properties
   (run-to-run determinism, batch invariance, monotone in-range timestamps)."""
import dataclasses

import numpy as np
import pytest

import gpu_common as G
from conftest import pk
from parakeet_cpp_amd import synth
from test_gpu_e2e import run_batch, tok

pytestmark = pytest.mark.gpu
N30 = 480000


@pytest.fixture(scope="module")
def cut_pair(tmp_path_factory):
    cfg = dataclasses.replace(pk.make_tdt_600m_config(), num_layers=2, name="tdt-600m-2L")
    return G.make_pair(tmp_path_factory.mktemp("b2"), cfg, seed=7)


def test_600m_two_layer_cut_vs_oracle(cut_pair, orc):
    W, om, gm = cut_pair
    pcm = synth.synth_pcm(2, N30, seed=99)
    feats = gm.mel(pcm)
    ofeats = np.stack([orc.mel(p, n_mels=128) for p in pcm])
    assert feats.shape == (2, 3001, 128)
    G.assert_bits_equal(feats, ofeats, "mel features (128 bins, 30 s)")
    G.assert_bits_equal(gm.subsample(feats), om.subsampling(ofeats), "subsampling (128 bins -> 4096 -> 1024)")
    enc = gm.encode(feats)
    oenc = om.encoder(ofeats)
    assert enc.shape == (2, 376, 1024)
    G.assert_bits_equal(enc, oenc, "2-layer encoder, d=1024, hd=128, T=376")
    g, o = gm.tdt_decode(enc), om.tdt_greedy(oenc)
    assert np.array_equal(g["lens"], o["lens"])
    for b in range(2):
        n = o["lens"][b]
        assert np.array_equal(g["ids"][b, :n], o["ids"][b, :n]), "TDT ids (vocab 8193, 2 LSTM layers)"
        assert np.array_equal(g["start"][b, :n], o["start"][b, :n]) and np.array_equal(g["end"][b, :n], o["end"][b, :n])
    assert o["lens"].sum() > 10, "degenerate decode"


def test_600m_full_model_batch32_properties(tmp_path_factory):
    from parakeet_cpp_amd import capi
    cfg = pk.make_tdt_600m_config()
    wp = str(tmp_path_factory.mktemp("b24") / "tdt600m.safetensors")
    synth.save_weights(wp, synth.synth_weights(cfg, seed=42))
    gm = capi.Model(wp, cfg, device=0)
    pcm = synth.synth_pcm(32, N30, seed=4321)
    r1 = run_batch(gm, pcm, 1)
    r2 = run_batch(gm, pcm, 1)
    assert np.array_equal(r1["lens"], r2["lens"]) and np.array_equal(r1["ids"], r2["ids"]), "run-to-run determinism"
    assert (r1["lens"] > 0).all()
    alone = run_batch(gm, pcm[9:10], 1)
    pair = run_batch(gm, pcm[[9, 30]], 1)
    assert tok(alone, 0) == tok(r1, 9) == tok(pair, 0), "batch invariance"
    assert tok(pair, 1) == tok(r1, 30)
    for b in range(32):
        n = r1["lens"][b]
        s, e = r1["start"][b, :n], r1["end"][b, :n]
        assert (np.diff(s) >= 0).all() and (e >= s).all() and (e < 376).all() and (s >= 0).all()
        assert (r1["ids"][b, :n] < 8192).all() and (r1["ids"][b, :n] >= 0).all()
    gm.close()
