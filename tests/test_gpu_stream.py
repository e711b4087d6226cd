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


# ---- full depth: BASELINE configs[4] at its own shapes ------------------------------------------------------------------------------------
import os  # noqa: E402

from conftest import ROOT  # noqa: E402

STREAM_GOLD = os.path.join(ROOT, "tests", "golden", "nemotron600m_stream_depth24_seed42.npz")


def _bits_sum_xor(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()
    if u.size == 0:
        return np.zeros(2, np.uint64)
    return np.array([int(u.astype(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF), int(np.bitwise_xor.reduce(u))], np.uint64)


def test_full_depth_nemotron_600m_16_streams(tmp_path):
    """nemotron-600m as shipped (24 layers, d 1024, vocab 8193, 2 LSTM layers; att_context 70 / 1, 160 ms chunks): 16 lock-step streams on the
    GPU, chunk by chunk against the 24-layer CPU oracle's committed outputs (tools/make_golden_stream_600m.py) -- log-mel and encoder output
    bits of every chunk through the staged calls, token ids / frames / confidence bits of every chunk through pk_stream_push -- and against
    the token ids of the reference's own streaming classes (/root/reference/src/streaming_encoder.cpp:162-272,430-472, src/nemotron.cpp:24-52,
    src/eou.cpp:17-98) recorded in the same fixture."""
    if not os.path.exists(STREAM_GOLD):
        pytest.skip("tests/golden/nemotron600m_stream_depth24_seed42.npz is missing (tools/make_golden_stream_600m.py)")
    g = np.load(STREAM_GOLD, allow_pickle=False)
    cfg = pk.make_nemotron_600m_config()
    W = synth.synth_weights(cfg, seed=int(g["weights_seed"]))
    wp = str(tmp_path / "nemotron600m.safetensors")
    synth.save_weights(wp, W)
    del W
    S0, n_chunks, chunk = int(g["n_streams"]), int(g["n_chunks"]), int(g["chunk"])
    pcm0 = synth.synth_pcm(S0, chunk * n_chunks, seed=int(g["pcm_seed"]))
    assert np.array_equal(np.asarray(pcm0, np.float64).sum(axis=1), g["pcm_digest"]), "the regenerated audio differs from the fixture's"
    S = 16
    pcm = np.ascontiguousarray(pcm0[np.arange(S) % S0])            # stream s carries the audio of fixture stream s % 2
    gm = capi.Model(wp, cfg, device=0)
    left, right, mt = int(g["att_left"]), int(g["att_right"]), int(g["max_tok"])
    # pass 1: the product call, device-resident stages
    gs = capi.Stream(gm, S, left, right)
    total = 0
    for i in range(n_chunks):
        r = gs.push(pcm[:, i * chunk:(i + 1) * chunk], max_tokens=mt)
        for s in range(S):
            n = int(g["n_tok"][i, s % S0])
            assert r["lens"][s] == n, f"chunk {i} stream {s}: {r['lens'][s]} tokens, the oracle emitted {n}"
            assert np.array_equal(r["ids"][s, :n], g["ids"][i, s % S0, :n]), f"chunk {i} stream {s}: token ids"
            assert np.array_equal(r["start"][s, :n], g["start"][i, s % S0, :n]) and np.array_equal(r["end"][s, :n], g["end"][i, s % S0, :n]), f"chunk {i} stream {s}: frames"
            assert np.array_equal(r["conf"][s, :n].view(np.uint32), g["conf_bits"][i, s % S0, :n]), f"chunk {i} stream {s}: confidence bits"
            total += n
    assert total >= 8 * int(g["n_tok"].sum()) > 0, "degenerate test: nothing decoded"
    gs.close()
    # pass 2: stage by stage (a fresh session set): the bits of every chunk's log-mel and encoder output
    gs = capi.Stream(gm, S, left, right)
    for i in range(n_chunks):
        m = gs.mel(pcm[:, i * chunk:(i + 1) * chunk])
        assert m.shape[1] == int(g["mel_n"][i, 0])
        for s in range(S):
            assert np.array_equal(_bits_sum_xor(m[s]), g["mel_bits"][i, s % S0]), f"chunk {i} stream {s}: log-mel bits"
        if m.shape[1] == 0:
            continue
        e = gs.encode(m)
        assert e.shape[1] == int(g["enc_n"][i, 0])
        if e.shape[1] == 0:
            continue
        for s in range(S):
            assert np.array_equal(e[s, 0].view(np.uint32), g["enc_row0"][i, s % S0].view(np.uint32)), f"chunk {i} stream {s}: first encoder row"
            assert np.array_equal(_bits_sum_xor(e[s]), g["enc_bits"][i, s % S0]), f"chunk {i} stream {s}: 24-layer encoder output bits"
        d = gs.decode(e, max_tokens=mt)
        for s in range(S):
            n = int(g["n_tok"][i, s % S0])
            assert d["lens"][s] == n and np.array_equal(d["ids"][s, :n], g["ids"][i, s % S0, :n])
    gs.close()
    gm.close()
    if "ref_equal_oracle" in g.files:
        assert bool(g["ref_equal_oracle"]), "fixture: the oracle's tokens differed from the reference code's on stream 0"
