"""Sortformer speaker diarization (reference src/sortformer.cpp:41-121, include/parakeet/sortformer.hpp:28-129).
CPU: the oracle's restatement -- probs_to_segments against hand-worked cases of the reference loop, the forward pass against an
independent torch restatement (fp32 round-off).  GPU: the product (pk_sortformer_*) against the oracle, bit for bit, on a small
model and on the real 117M shapes; segments identical."""
import dataclasses
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import torch_ref as TR
from conftest import pk
from parakeet_cpp_amd import synth
from test_transformer import torch_transformer


def tiny_sf(**kw):
    nest = pk.make_nest_encoder_config(name="nest-tiny", subsampling_channels=32, hidden_size=128, num_layers=2, num_heads=2,
                                       ffn_intermediate=256, mel_bins=128)
    sf = pk.SortformerConfig(nest_encoder=nest, transformer_hidden=96, transformer_layers=2, transformer_heads=4, transformer_ffn=192)
    return dataclasses.replace(sf, **kw)


def feats_like(B, Tm, F_, seed):
    """Un-normalised log-mel-like features (Sortformer runs with AudioConfig::normalize = false, src/main.cpp:516)."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((B, Tm, F_)) * 2.0 - 6.0).astype(np.float32)


# ---- probs_to_segments (src/sortformer.cpp:71-113) -----------------------------------------------------------
def test_probs_to_segments_hand_worked(orc):
    T, S = 10, 3
    p = np.zeros((T, S), np.float32)
    p[2:5, 0] = 0.9                      # speaker 0: frames 2..4            -> [0.16, 0.32]
    p[0:2, 1] = 0.7; p[6:10, 1] = 0.8    # speaker 1: 0..1 and 6..9 (open)   -> [0.00, 0.08], [0.48, 0.72]
    p[4, 2] = 0.5                        # == threshold: NOT active (strict '>', :84)
    p[7, 2] = 0.5000001                  # a one-frame segment               -> [0.56, 0.56]
    segs = orc.probs_to_segments(p, 0.5)
    f = lambda k: float(np.float32(k) * np.float32(0.08))
    assert segs == [(1, f(0), f(1)), (0, f(2), f(4)), (1, f(6), f(9)), (2, f(7), f(7))]      # sorted by start (:106-110)
    assert orc.probs_to_segments(np.zeros((5, 2), np.float32), 0.5) == []
    assert orc.probs_to_segments(np.ones((5, 1), np.float32), 0.5) == [(0, 0.0, f(4))]


def test_oracle_sortformer_matches_torch(orc):
    sf = tiny_sf()
    W = synth.synth_sortformer_weights(sf, seed=5)
    feats = feats_like(2, 97, 128, 1)
    got = orc.Model(sf.nest_encoder, W).sortformer_forward(feats, sf)
    # independent restatement: torch conv / linear / layer_norm / softmax
    We = {k.replace("nest_encoder_.", "encoder_."): v for k, v in W.items()}
    x = TR.subsampling(We, feats) * np.float32(np.sqrt(np.float32(sf.nest_encoder.hidden_size)))     # xscaling (streaming_encoder.cpp:402-406)
    pe = TR.pos_emb(x.shape[1], x.shape[2])
    for l in range(sf.nest_encoder.num_layers):
        x = TR.conformer_block(We, l, x, pe, sf.nest_encoder.num_heads)
    t = torch.from_numpy
    x = F.linear(t(np.ascontiguousarray(x)), t(W["projection_.weight"]), t(W["projection_.bias"])).numpy()
    x = torch_transformer(W, "transformer_.", x, sf.transformer_layers, sf.transformer_heads, sf.pre_ln, sf.has_final_norm)
    h = F.relu(F.linear(F.relu(t(x)), t(W["first_hidden_.weight"]), t(W["first_hidden_.bias"])))
    want = torch.sigmoid(F.linear(h, t(W["output_proj_.weight"]), t(W["output_proj_.bias"]))).numpy()
    assert got.shape == want.shape == (2, 13, 4)
    assert np.abs(got - want).max() < 2e-4, np.abs(got - want).max()
    assert 0.02 < got.min() < 0.5 < got.max() < 0.98, "degenerate test: activities should straddle the threshold"


# ---- GPU -------------------------------------------------------------------------------------------------------
def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Tm", [(1, 97), (3, 401)])
def test_gpu_sortformer_bit_identical_small(orc, tmp_path, B, Tm):
    from parakeet_cpp_amd import capi
    sf = tiny_sf()
    W = synth.synth_sortformer_weights(sf, seed=5)
    wp = str(tmp_path / "sf.safetensors")
    synth.save_weights(wp, W)
    feats = feats_like(B, Tm, 128, 2)
    want = orc.Model(sf.nest_encoder, W).sortformer_forward(feats, sf)
    g = capi.Sortformer(wp, sf)
    got = g.forward(feats)
    assert np.array_equal(_bits(got), _bits(want)), np.abs(got - want).max()
    for b in range(B):
        assert capi.sortformer_segments(got[b], 0.5) == orc.probs_to_segments(want[b], 0.5)
    assert sum(len(orc.probs_to_segments(want[b], 0.5)) for b in range(B)) > 0
    g.close()


@pytest.mark.gpu
def test_gpu_sortformer_117m_from_pcm(orc, tmp_path):
    """make_sortformer_117m_config shapes (17-layer NEST with 128 mels + 18-layer post-LN transformer of 192 / 8 heads of 24), from PCM:
    preprocess_audio(normalize = false) -> forward -> segments, as run_sortformer (src/main.cpp:497-538) does."""
    from parakeet_cpp_amd import capi
    sf = pk.make_sortformer_117m_config()
    W = synth.synth_sortformer_weights(sf, seed=11)
    wp = str(tmp_path / "sf117.safetensors")
    synth.save_weights(wp, W)
    pcm = synth.synth_pcm(2, 64000, seed=4)                                  # 2 x 4 s
    feats = np.stack([orc.mel(p, n_mels=128, normalize=False) for p in pcm])
    want = orc.Model(sf.nest_encoder, W).sortformer_forward(feats, sf)
    g = capi.Sortformer(wp, sf)
    got = g.forward_pcm(pcm)
    assert got.shape == want.shape == (2, 51, 4)
    assert np.array_equal(_bits(got), _bits(want)), np.abs(got - want).max()
    for b in range(2):
        assert capi.sortformer_segments(got[b], sf.activity_threshold) == orc.probs_to_segments(want[b], sf.activity_threshold)
    g.close()


def test_oracle_sortformer_chunks_consistent_with_stream_encoder(orc):
    """diarize_chunk = forward_chunk (already pinned against the torch streaming restatement, tests/test_stream_oracle.py) + the head
    of the offline path on the chunk's frames: chunked probabilities must equal head(orc.Stream.encode(chunk)) computed separately."""
    sf = tiny_sf()
    W = synth.synth_sortformer_weights(sf, seed=5)
    om = orc.Model(sf.nest_encoder, W)
    feats = feats_like(1, 100, 128, 3)[0]
    st = orc.Stream(om, sf.att_context_left, sf.att_context_right)
    got, n_frames = [], 0
    for a, b in [(0, 16), (16, 40), (40, 49), (49, 100)]:
        p = st.sortformer_chunk(feats[a:b], sf)
        got.append(p)
        n_frames += p.shape[0]
    assert n_frames == (100 // 8) and sum(p.shape[0] > 0 for p in got) >= 3
    assert all(np.isfinite(p).all() and ((p > 0) & (p < 1)).all() for p in got if p.size)


@pytest.mark.gpu
def test_gpu_sortformer_diarize_chunk_bit_identical(orc, tmp_path):
    """Streaming diarization: pk_sortformer_diarize_chunk on ragged feature chunks == the oracle's diarize_chunk, bit for bit,
    including chunks that only fill the subsampling buffer (0 frames out); reset starts an identical second session."""
    from parakeet_cpp_amd import capi
    sf = tiny_sf()
    W = synth.synth_sortformer_weights(sf, seed=5)
    wp = str(tmp_path / "sf.safetensors")
    synth.save_weights(wp, W)
    g = capi.Sortformer(wp, sf)
    feats = feats_like(1, 200, 128, 7)[0]
    cuts = [(0, 16), (16, 19), (19, 64), (64, 72), (72, 136), (136, 200)]
    for session in range(2):
        st = orc.Stream(orc.Model(sf.nest_encoder, W), sf.att_context_left, sf.att_context_right)
        total = 0
        for a, b in cuts:
            want = st.sortformer_chunk(feats[a:b], sf)
            got = g.diarize_chunk(feats[a:b])
            assert got.shape == want.shape, (session, a, b)
            assert np.array_equal(_bits(got), _bits(want)), (session, a, b)
            if got.shape[0]:
                assert capi.sortformer_segments(got, 0.5) == orc.probs_to_segments(want, 0.5)
            total += got.shape[0]
        assert total == 200 // 8
        g.reset_stream()
    g.close()
