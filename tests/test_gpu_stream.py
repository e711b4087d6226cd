"""GPU parity of the streaming path (pk_stream_*; reference NemotronTranscriber / StreamingTranscriber::transcribe_chunk:
src/audio.cpp:195-259, src/streaming_encoder.cpp:430-472, src/eou.cpp:17-98) against the oracle's Stream, bit for bit, stage by
stage and chunk by chunk with carried state: several lock-step streams on the GPU vs one oracle Stream per stream."""
import dataclasses

import numpy as np
import pytest

import gpu_common as G
from conftest import pk
from parakeet_cpp_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_pair(om, gm, orc, S, left, right, chunk, n_chunks, seed):
    gs = capi.Stream(gm, S, left, right)
    os_ = [orc.Stream(om, left, right) for _ in range(S)]
    pcm = synth.synth_pcm(S, chunk * n_chunks, seed=seed)
    n_tok, n_enc = 0, 0
    for i in range(n_chunks):
        seg = pcm[:, i * chunk:(i + 1) * chunk]
        gmel = gs.mel(seg)
        omel = [o.mel(seg[s]) for s, o in enumerate(os_)]
        assert gmel.shape[1] == omel[0].shape[0]
        if gmel.shape[1] == 0:
            continue
        G.assert_bits_equal(gmel, np.stack(omel), f"stream log-mel, chunk {i}")
        genc = gs.encode(gmel)
        oenc = [o.encode(omel[s]) for s, o in enumerate(os_)]
        assert genc.shape[1] == oenc[0].shape[0]
        if genc.shape[1] == 0:
            continue
        G.assert_bits_equal(genc, np.stack(oenc), f"stream encoder, chunk {i}")
        n_enc += genc.shape[1]
        g = gs.decode(genc)
        for s, o in enumerate(os_):
            r = o.decode(oenc[s])
            n = len(r["ids"])
            assert g["lens"][s] == n, f"chunk {i} stream {s}"
            assert np.array_equal(g["ids"][s, :n], r["ids"]) and np.array_equal(g["start"][s, :n], r["start"])
            assert np.array_equal(g["end"][s, :n], r["end"])
            G.assert_bits_equal(g["conf"][s, :n], r["conf"], "stream confidence")
            n_tok += n
    gs.close()
    return n_enc, n_tok


@pytest.mark.parametrize("left,right,chunk", [(10, 1, 2560), (70, 0, 2560), (6, 0, 4000)])
def test_stream_stages_tiny(tmp_path_factory, orc, left, right, chunk):
    cfg = G.tiny(num_layers=2, name="tiny-stream")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("ts"), cfg, seed=5)
    n_enc, n_tok = run_pair(om, gm, orc, 3, left, right, chunk, 14, seed=left + chunk)
    assert n_enc >= 10


def test_stream_push_equals_stages_and_emits(tmp_path_factory, orc):
    """pk_stream_push (device-resident mel -> encoder -> decode) == the oracle's push; 2-layer cut of the 110M architecture so that
    the decoder actually emits tokens; 4 streams with different audio."""
    cfg = dataclasses.replace(pk.make_110m_config(), num_layers=2, name="110m-2L-stream")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("s110"), cfg, seed=42)
    S, chunk, n_chunks = 4, 2560, 24
    gs = capi.Stream(gm, S, 70, 1)
    os_ = [orc.Stream(om, 70, 1) for _ in range(S)]
    pcm = synth.synth_pcm(S, chunk * n_chunks, seed=77)
    total = 0
    for i in range(n_chunks):
        seg = pcm[:, i * chunk:(i + 1) * chunk]
        g = gs.push(seg)
        for s, o in enumerate(os_):
            r = o.push(seg[s])
            n = 0 if r is None else len(r["ids"])
            assert g["lens"][s] == n, f"chunk {i} stream {s}"
            if n:
                assert np.array_equal(g["ids"][s, :n], r["ids"]) and np.array_equal(g["start"][s, :n], r["start"])
            total += n
    assert total > 0, "degenerate test: nothing decoded"
    # reset() starts the sessions over: same audio -> same tokens
    gs.reset()
    again = sum(int(gs.push(pcm[:, i * chunk:(i + 1) * chunk])["lens"].sum()) for i in range(n_chunks))
    assert again == total
    gs.close()
