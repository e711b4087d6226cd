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
