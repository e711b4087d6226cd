"""Streaming path, CPU: the oracle's restatement (oracle/pk_oracle.c, section "Streaming path") against the independent
torch restatement in tests/torch_ref_stream.py, chunk by chunk, with carried state on both sides.  fp32 round-off
tolerances; the state-carrying logic (overlap samples, leftover mel frames, K/V + conv caches, the un-shifted position
scores, the context mask) has to agree exactly in shape and within 2e-4 in value over a dozen chunks."""
import numpy as np
import pytest

from conftest import pk
from parakeet_cpp_amd import synth
from torch_ref_stream import TorchStream


@pytest.mark.parametrize("left,right,chunk", [(10, 1, 2560), (70, 0, 2560), (6, 0, 4000), (70, 13, 1999)])
def test_stream_oracle_matches_torch(orc, left, right, chunk):
    cfg = pk.make_tiny_config(num_layers=2)
    W = synth.synth_weights(cfg, seed=5)
    om = orc.Model(cfg, W)
    st = orc.Stream(om, left, right)
    ts = TorchStream(cfg, W, orc.mel_filterbank(n_mels=cfg.mel_bins), left, right)
    pcm = synth.synth_pcm(1, chunk * 14, seed=left + chunk)[0]
    n_enc = 0
    for i in range(14):
        seg = pcm[i * chunk:(i + 1) * chunk]
        m, tm = st.mel(seg), ts.mel(seg)
        assert (tm is None) == (m.shape[0] == 0)
        if tm is None:
            continue
        assert m.shape == tuple(tm.shape)
        assert np.abs(m - tm.numpy()).max() < 2e-3            # log-mel, unnormalised (log amplifies round-off in empty bins)
        e, te = st.encode(m), ts.encode(torch_from(m))
        assert (te is None) == (e.shape[0] == 0)
        if te is None:
            continue
        assert e.shape == tuple(te.shape)
        assert np.abs(e - te.numpy()).max() < 2e-4, f"chunk {i}"
        n_enc += e.shape[0]
    assert n_enc >= 10


def torch_from(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def test_stream_decode_carries_state(orc):
    """rnnt_streaming_decode_chunk (src/eou.cpp:17-98): feeding an encoder sequence in chunks with carried LSTM state and last token
    gives the same ids as one call when no duration skips past a chunk boundary; frames are absolute."""
    cfg = pk.make_tiny_config(num_layers=2)
    W = synth.synth_weights(cfg, seed=5)
    om = orc.Model(cfg, W)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 40, cfg.hidden_size)).astype(np.float32)
    x = (x - x.mean(-1, keepdims=True)) / x.std(-1, keepdims=True)
    whole = orc.Stream(om, 70, 0).decode(x[0])
    st = orc.Stream(om, 70, 0)
    ids, starts = [], []
    for t in range(40):                                            # one frame per chunk: no skip can be lost ... if all durations <= 1
        r = st.decode(x[0, t:t + 1])
        ids += r["ids"].tolist(); starts += r["start"].tolist()
    assert len(whole["ids"]) > 0
    if (np.diff(np.r_[whole["start"], 40]) <= 1).all() and (whole["end"] - whole["start"] <= 0).all():
        assert ids == whole["ids"].tolist() and starts == whole["start"].tolist()
    assert starts == sorted(starts) and all(0 <= s < 40 for s in starts)


def test_stream_oracle_bf16_mode_rounds_every_product(orc):
    """The tolerance-class mode of the streaming path (orc_config.gemm_bf16; the specification of kernels/gemm_smallm_bf16.hip): every Linear /
    1x1-conv product of a chunk -- subsampling, ffn, q / k / v / out, pointwise convs -- takes operands rounded to bf16; checked against an
    independent restatement: the torch stream with every weight matrix pre-rounded to bf16 is NOT the same thing (activations are rounded too),
    so the check here is structural: the mode deviates from fp32 at bf16-epsilon class (not less: some product left unrounded would show a
    much smaller gap; not more), state carried over 12 chunks, and the decode's per-decision records equal its tokens."""
    import dataclasses
    cfg = dataclasses.replace(pk.make_110m_config(), num_layers=2, name="110m-2L-bf16-stream")
    W = synth.synth_weights(cfg, seed=42)
    f32 = orc.Stream(orc.Model(cfg, W), 70, 1)
    b16 = orc.Stream(orc.Model(dataclasses.replace(cfg, gemm_bf16=True), W), 70, 1)
    pcm = synth.synth_pcm(1, 2560 * 12, seed=77)[0]
    gaps, n_tok = [], 0
    for i in range(12):
        seg = pcm[i * 2560:(i + 1) * 2560]
        m1, m2 = f32.mel(seg), b16.mel(seg)
        assert np.array_equal(m1.view(np.uint32), m2.view(np.uint32)), "the log-mel front end has no product in it: identical in both modes"
        if m1.shape[0] == 0:
            continue
        e1, e2 = f32.encode(m1), b16.encode(m2)
        assert e1.shape == e2.shape
        if e1.shape[0] == 0:
            continue
        gaps.append(float(np.abs(e1 - e2).mean() / np.abs(e1).max()))
        r = b16.decode(e2, margins=True)
        lab = [int(k) for k in r["step_label"] if k != cfg.blank_id]
        assert lab == r["ids"].tolist(), "the per-decision labels (blank dropped) are the chunk's tokens"
        assert len(r["step_margin"]) == len(r["step_label"]) and (r["step_margin"] >= 0).all()
        n_tok += len(r["ids"])
        f32.decode(e1)
    assert len(gaps) >= 8 and n_tok > 0
    assert 2e-4 < max(gaps) < 1e-2, f"bf16-vs-fp32 gap of the oracle's stream: {max(gaps):.2e} of max|x| (bf16 epsilon class expected)"


@pytest.mark.parametrize("left,right,chunk", [(10, 1, 2560), (70, 0, 4000)])
def test_stream_oracle_bf16_mode_matches_torch_bf16(orc, left, right, chunk):
    """The oracle's tolerance-class streaming mode against an INDEPENDENT statement of the same specification: the torch restatement with both
    operands of every Linear / 1x1-conv product rounded to bf16 (torch's RNE) and fp32 accumulation.  The two differ in accumulation order only,
    so they agree far inside the mode's own distance from fp32 -- but not to fp32 round-off: a last-bit difference in an activation can round to
    the neighbouring bf16 value in one of them.  (What this pins: WHICH products are rounded, and that nothing else is.)"""
    import dataclasses
    cfg = dataclasses.replace(pk.make_tiny_config(num_layers=2), gemm_bf16=True)
    W = synth.synth_weights(cfg, seed=5)
    st = orc.Stream(orc.Model(cfg, W), left, right)
    f32 = orc.Stream(orc.Model(dataclasses.replace(cfg, gemm_bf16=False), W), left, right)
    ts = TorchStream(cfg, W, orc.mel_filterbank(n_mels=cfg.mel_bins), left, right, bf16=True)
    pcm = synth.synth_pcm(1, chunk * 14, seed=left + chunk)[0]
    dev, gap, n = [], [], 0
    for i in range(14):
        seg = pcm[i * chunk:(i + 1) * chunk]
        m, mf, tm = st.mel(seg), f32.mel(seg), ts.mel(seg)
        if tm is None:
            continue
        e, ef, te = st.encode(m), f32.encode(mf), ts.encode(torch_from(m))
        assert (te is None) == (e.shape[0] == 0)
        if te is None:
            continue
        mx = np.abs(ef).max()
        dev.append(np.abs(e - te.numpy()).mean() / mx)
        gap.append(np.abs(e - ef).mean() / mx)
        assert np.abs(e - te.numpy()).max() <= 2e-2 * mx, f"chunk {i}"
        n += e.shape[0]
    assert n >= 10
    assert max(dev) < 0.5 * min(gap), f"oracle bf16 vs torch bf16 {max(dev):.2e} of max|x|; the mode's distance from fp32 {min(gap):.2e}"
