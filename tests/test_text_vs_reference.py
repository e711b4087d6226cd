"""Host-side text post-processing of the product (pk_detokenize / pk_tokenize / pk_group_timestamps, csrc/text.cpp)
against the REAL reference objects: oracle/_ref/libpk_ref_text.so is built from /root/reference/src/vocab.cpp and
src/timestamp.cpp where they lie (the only reference units that compile without axiom).  Where the reference
library is absent (GPU box: /root/reference does not exist, but the prebuilt _ref travels) the reference's own
gtest expectations (tests/test_all.cpp:45-129, 434-477, 1217-1276) still run."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, pk
from parakeet_cpp_amd import capi, synth

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libpk_ref_text.so")
MARK = "▁"


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    d = tmp_path_factory.mktemp("txt")
    cfg = pk.make_tiny_config(vocab_size=513, blank_id=512, ctc_vocab_size=513)
    wp, vp = str(d / "w.safetensors"), str(d / "vocab.txt")
    synth.save_weights(wp, synth.synth_weights(cfg))
    pieces = synth.synth_vocab(512)
    synth.save_vocab(vp, pieces)
    return capi.Model(wp, cfg, vocab_path=vp), pieces, vp


def detok(m, ids):
    ids = np.asarray(ids, np.int32)
    L = capi.lib()
    n = L.pk_detokenize(m._h, ids.ctypes.data_as(capi.i32p), len(ids), None, 0)
    buf = C.create_string_buffer(n + 1)
    L.pk_detokenize(m._h, ids.ctypes.data_as(capi.i32p), len(ids), buf, n + 1)
    return buf.value.decode()


def tokenize(m, text):
    L = capi.lib()
    ids = np.zeros(4096, np.int32)
    n = L.pk_tokenize(m._h, text.encode(), ids.ctypes.data_as(capi.i32p), 4096)
    return ids[:n].tolist()


def group(m, toks, sentences=False):
    L = capi.lib()
    n = len(toks)
    ids = np.array([t[0] for t in toks], np.int32); st = np.array([t[1] for t in toks], np.int32)
    en = np.array([t[2] for t in toks], np.int32); cf = np.array([t[3] if len(t) > 3 else 1.0 for t in toks], np.float32)
    words = C.create_string_buffer(1 << 16)
    ws, we, wc = (np.zeros(1024, np.float32) for _ in range(3))
    k = L.pk_group_timestamps(m._h, ids.ctypes.data_as(capi.i32p), st.ctypes.data_as(capi.i32p), en.ctypes.data_as(capi.i32p),
                              cf.ctypes.data_as(capi.f32p), n, int(sentences), words, 1 << 16, ws.ctypes.data_as(capi.f32p),
                              we.ctypes.data_as(capi.f32p), wc.ctypes.data_as(capi.f32p), 1024)
    w = words.value.decode().split("\n") if k else []
    return [(w[i], float(ws[i]), float(we[i]), float(wc[i])) for i in range(k)]


def test_reference_gtest_expectations(model):
    m, pieces, _ = model
    assert capi.lib().pk_vocab_size(m._h) == 513                 # Tokenizer.VocabSize: pieces + blank (test_all.cpp:434-445)
    assert detok(m, [9999]) == "[9999]"                          # :459-465
    assert detok(m, []) == ""                                    # :467-470
    text = "an ka to"
    ids = tokenize(m, text)
    assert ids and all(0 <= i < 512 for i in ids)
    assert tokenize(m, "") == []                                 # :1234-1240
    t = detok(m, ids)
    assert t.replace(" ", "") != ""
    # frame_to_seconds KATs :45-50 through the word grouping
    g = group(m, [(pieces.index(next(p for p in pieces if p.startswith(MARK))), 125, 125)])
    assert abs(g[0][1] - 10.0) < 1e-6


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="reference text library not built (needs /root/reference once)")
def test_against_the_real_reference_objects(model):
    m, pieces, vp = model
    R = C.CDLL(REF_SO)
    R.ref_tok_load.restype = C.c_void_p
    R.ref_tok_load.argtypes = [C.c_char_p]
    R.ref_tok_decode.argtypes = [C.c_void_p, capi.i32p, C.c_int, C.c_char_p, C.c_int]
    R.ref_tok_encode.argtypes = [C.c_void_p, C.c_char_p, capi.i32p, C.c_int]
    R.ref_group_timestamps.argtypes = [C.c_void_p, capi.i32p, capi.i32p, capi.i32p, capi.f32p, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                       capi.f32p, capi.f32p, capi.f32p, C.c_int]
    R.ref_tok_vocab_size.argtypes = [C.c_void_p]
    t = R.ref_tok_load(vp.encode())
    assert t and R.ref_tok_vocab_size(t) == capi.lib().pk_vocab_size(m._h)
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(0, 40))
        ids = rng.integers(-2, 520, n).astype(np.int32)         # includes out-of-range ids -> "[id]"
        buf = C.create_string_buffer(1 << 14)
        R.ref_tok_decode(t, ids.ctypes.data_as(capi.i32p), n, buf, 1 << 14)
        assert detok(m, ids) == buf.value.decode(), ids
    for trial in range(100):
        words = ["".join(p.replace(MARK, "") for p in rng.choice(pieces, int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 8)))]
        text = " ".join(words) + ("?" if trial % 7 == 0 else "")
        out = np.zeros(4096, np.int32)
        k = R.ref_tok_encode(t, text.encode(), out.ctypes.data_as(capi.i32p), 4096)
        assert tokenize(m, text) == out[:k].tolist(), text
    for trial in range(100):
        n = int(rng.integers(0, 30))
        ids = rng.integers(-1, 515, n).astype(np.int32)
        st = np.sort(rng.integers(0, 126, n)).astype(np.int32)
        en = (st + rng.integers(0, 4, n)).astype(np.int32)
        cf = rng.uniform(0.1, 1.0, n).astype(np.float32)
        for sent in (0, 1):
            words = C.create_string_buffer(1 << 16)
            ws, we, wc = (np.zeros(1024, np.float32) for _ in range(3))
            k = R.ref_group_timestamps(t, ids.ctypes.data_as(capi.i32p), st.ctypes.data_as(capi.i32p), en.ctypes.data_as(capi.i32p),
                                       cf.ctypes.data_as(capi.f32p), n, sent, words, 1 << 16, ws.ctypes.data_as(capi.f32p),
                                       we.ctypes.data_as(capi.f32p), wc.ctypes.data_as(capi.f32p), 1024)
            want = [(w, float(ws[i]), float(we[i]), float(wc[i])) for i, w in enumerate(words.value.decode().split("\n") if k else [])]
            got = group(m, list(zip(ids.tolist(), st.tolist(), en.tolist(), cf.tolist())), bool(sent))
            assert got == want


def test_group_timestamps_reference_cases(model):
    """tests/test_all.cpp:63-129 restated on a purpose-made vocab."""
    import tempfile
    cfg = pk.make_tiny_config(vocab_size=9, blank_id=8, ctc_vocab_size=9)
    with tempfile.TemporaryDirectory() as d:
        wp, vp = os.path.join(d, "w.safetensors"), os.path.join(d, "v.txt")
        synth.save_weights(wp, synth.synth_weights(cfg))
        pieces = [MARK + "the", MARK + "quick", MARK + "fox", MARK + "run", "ning", MARK + "Hello", MARK + "world.", MARK + "How"]
        synth.save_vocab(vp, pieces, with_scores=False)
        m = capi.Model(wp, cfg, vocab_path=vp)
        assert group(m, []) == []
        w = group(m, [(0, 0, 2), (1, 5, 8), (2, 12, 15)])
        assert [x[0] for x in w] == ["the", "quick", "fox"]
        w = group(m, [(3, 0, 3), (4, 4, 6)])
        assert len(w) == 1 and w[0][0] == "running" and abs(w[0][1] - 0.0) < 1e-7 and abs(w[0][2] - 0.48) < 1e-6
        s = group(m, [(5, 0, 2), (6, 3, 5), (7, 8, 10), (0, 11, 13)], sentences=True)
        assert [x[0] for x in s] == ["Hello world.", "How the"]
        w = group(m, [(999, 0, 1), (0, 2, 4)])
        assert len(w) == 1 and w[0][0] == "the"
        assert detok(m, [5, 6]) == "Hello world."


def test_wav_reader(tmp_path):
    L = capi.lib()
    pcm = synth.synth_pcm(1, 8000, seed=3)[0]
    p = str(tmp_path / "a.wav")
    synth.write_wav_pcm16(p, pcm)
    out = capi.f32p()
    n, sr = C.c_int64(), C.c_int()
    capi.check(L.pk_read_wav(p.encode(), C.byref(out), C.byref(n), C.byref(sr)))
    got = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.pk_free(out)
    assert sr.value == 16000 and n.value == 8000
    want = (np.clip(pcm, -1, 1) * 32767.0).astype("<i2").astype(np.float32) / 32768.0     # int16 -> f32 is /32768 (test_all.cpp AudioIO)
    assert np.array_equal(got, want)
    assert L.pk_read_wav(str(tmp_path / "nope.wav").encode(), C.byref(out), C.byref(n), C.byref(sr)) == -2
    (tmp_path / "junk.wav").write_bytes(b"not a wav file at all")
    assert L.pk_read_wav(str(tmp_path / "junk.wav").encode(), C.byref(out), C.byref(n), C.byref(sr)) == -2
