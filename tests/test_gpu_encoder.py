"""GPU parity: FastConformer encoder (pk_subsample / pk_encode; reference src/encoder.cpp:208-271) against the
oracle, bit-for-bit, stage by stage (so a mismatch names the kernel) and end to end."""
import numpy as np
import pytest

import gpu_common as G
from parakeet_cpp_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_pair(tmp_path_factory):
    return G.make_pair(tmp_path_factory.mktemp("tiny"), G.tiny())


@pytest.fixture(scope="module")
def wide_pair(tmp_path_factory):
    return G.make_pair(tmp_path_factory.mktemp("wide"), G.one_layer_110m(2), seed=3)


@pytest.mark.parametrize("B,Tm", [(1, 101), (3, 203), (2, 57), (1, 9)])
def test_subsampling_bits(tiny_pair, B, Tm):
    W, om, gm = tiny_pair
    feats = np.random.default_rng(Tm).standard_normal((B, Tm, om.cfg.mel_bins)).astype(np.float32)
    G.assert_bits_equal(gm.subsample(feats), om.subsampling(feats), "subsampling")


@pytest.mark.parametrize("stage", [1, 2, 3, 4])
def test_block_stages_bits_tiny(tiny_pair, orc, stage):
    W, om, gm = tiny_pair
    feats = np.random.default_rng(5).standard_normal((2, 301, 80)).astype(np.float32)
    x = om.subsampling(feats)
    want = om.conformer_block(0, x, stop_after=stage)
    G.assert_bits_equal(gm.encode(feats, stop_layer=0, stop_stage=stage), want, f"block 0 stage {stage}")


def test_encoder_bits_tiny_ragged_lengths(tiny_pair):
    W, om, gm = tiny_pair
    for B, n in [(1, 16000), (4, 23456), (2, 4000)]:
        pcm = synth.synth_pcm(B, n, seed=n)
        feats = gm.mel(pcm)
        G.assert_bits_equal(gm.encode(feats), om.encoder(feats), f"encoder B={B} n={n}")


@pytest.mark.parametrize("stage", [1, 2, 3, 4])
def test_block_stages_bits_full_width(wide_pair, stage):
    """d=512, 8 heads, ffn 2048, T=126 (the cfg-A shapes), two layers."""
    W, om, gm = wide_pair
    pcm = synth.synth_pcm(2, 160000, seed=77)
    feats = gm.mel(pcm)
    x = om.subsampling(feats)
    want = om.conformer_block(0, x, stop_after=stage)
    G.assert_bits_equal(gm.encode(feats, stop_layer=0, stop_stage=stage), want, f"full-width block 0 stage {stage}")


def test_encoder_bits_full_width(wide_pair):
    W, om, gm = wide_pair
    pcm = synth.synth_pcm(3, 160000, seed=78)
    feats = gm.mel(pcm)
    G.assert_bits_equal(gm.subsample(feats), om.subsampling(feats), "full-width subsampling")
    G.assert_bits_equal(gm.encode(feats), om.encoder(feats), "full-width 2-layer encoder")


@pytest.mark.parametrize("heads,Tm", [(2, 1233), (2, 2401), (1, 1233), (1, 505), (4, 1233), (4, 2401)])
def test_attention_long_sequences_and_head_dims(tmp_path_factory, heads, Tm):
    """The attention kernel streams K / P / V through a fixed chunk buffer: T = 155 and 301 exercise 2-3 chunks per
    phase (hd = 64 chunk 128, hd = 128 chunk 64), heads = 1 / 4 the hd = 128 / 32 instantiations, odd T the zero pad."""
    cfg = G.tiny(num_heads=heads, num_layers=1, name=f"tiny-h{heads}")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp(f"h{heads}"), cfg, seed=11)
    feats = np.random.default_rng(Tm + heads).standard_normal((2, Tm, 80)).astype(np.float32)
    x = om.subsampling(feats)
    want = om.conformer_block(0, x, stop_after=2)
    G.assert_bits_equal(gm.encode(feats, stop_layer=0, stop_stage=2), want, f"attention heads={heads} Tm={Tm}")
