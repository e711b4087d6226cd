"""GPU: the one-call boundary pk_transcribe_pcm (Transcriber::transcribe, reference transcribe.hpp:99-179) on RAGGED input -- clips of
different lengths in one call (grouped by length on the GPU, results in the caller's order), the shortest clips the front end
accepts (2 mel frames -> 1 encoder frame), duplicates -- against the oracle per clip; text, token timestamps, word grouping; and
the error behaviour at the boundary (too-short clip, missing vocabulary for boost phrases, bad decoder)."""
import numpy as np
import pytest

import gpu_common as G
from parakeet_cpp_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair(tmp_path_factory):
    import dataclasses
    from conftest import pk
    cfg = dataclasses.replace(pk.make_110m_config(), num_layers=2, name="110m-2L-api")     # the tiny model decodes real mel input to blanks only
    return G.make_pair(tmp_path_factory.mktemp("api"), cfg, with_vocab=True)


def oracle_clip(orc, om, pcm, blank):
    enc = om.encoder(orc.mel(pcm)[None])
    return om.tdt_greedy(enc), orc.ctc_greedy(om.ctc_logprobs(enc), blank)


def test_ragged_clips_in_one_call(pair, orc):
    W, om, gm = pair
    lens = [16000, 24000, 16000, 4000, 24000, 257, 1600, 16000, 48000]
    clips = [synth.synth_pcm(1, n, seed=100 + i)[0] for i, n in enumerate(lens)]
    clips[7] = clips[0].copy()                                              # a duplicate clip decodes identically
    pieces = synth.synth_vocab(om.cfg.vocab_size - 1)
    for dec in ("tdt", "ctc"):
        res = gm.transcribe_pcm(clips, dec, timestamps=True)
        assert len(res) == len(clips)
        for i, pcm in enumerate(clips):
            t, c = oracle_clip(orc, om, pcm, om.cfg.ctc_vocab_size - 1)
            o = t if dec == "tdt" else c
            n = o["lens"][0]
            assert res[i]["token_ids"] == o["ids"][0, :n].tolist(), (dec, i, lens[i])
            assert res[i]["start"] == o["start"][0, :n].tolist() and res[i]["end"] == o["end"][0, :n].tolist()
            assert np.array_equal(np.float32(res[i]["conf"]), o["conf"][0, :n])
            text = "".join(pieces[k] for k in res[i]["token_ids"]).replace("▁", " ")
            assert res[i]["text"] == (text[1:] if text.startswith(" ") else text)
            starts = [w[1] for w in res[i]["words"]]
            assert starts == sorted(starts)
        assert res[7]["token_ids"] == res[0]["token_ids"]
        assert sum(len(r["token_ids"]) for r in res) > 0
        assert len(res[5]["token_ids"]) <= om.cfg.max_symbols_per_step       # 257 samples: ONE encoder frame


def test_boundary_errors(pair):
    W, om, gm = pair
    ok = synth.synth_pcm(1, 16000, seed=1)[0]
    with pytest.raises(RuntimeError, match="more than 256 samples"):
        gm.transcribe_pcm([ok, np.zeros(200, np.float32)], "tdt")
    L = capi.lib()
    import ctypes as C
    opt = capi.PkOptions()
    opt.decoder = 7
    off = np.array([0, 16000], np.int64)
    res = C.POINTER(capi.PkResult)()
    st = L.pk_transcribe_pcm(gm._h, ok.ctypes.data_as(capi.f32p), off.ctypes.data_as(capi.i64p), 1, C.byref(opt), C.byref(res))
    assert st != 0
    assert L.pk_transcribe_pcm(None, ok.ctypes.data_as(capi.f32p), off.ctypes.data_as(capi.i64p), 1, None, C.byref(res)) != 0


def test_model_without_vocabulary(tmp_path, orc):
    """vocab_path = NULL: token ids still come back, text stays empty, boost PHRASES (which need Tokenizer::encode) are refused while
    boost TOKENS work."""
    cfg = G.tiny(name="tiny-novocab")
    W = synth.synth_weights(cfg, seed=42)
    wp = str(tmp_path / "m.safetensors")
    synth.save_weights(wp, W)
    gm = capi.Model(wp, cfg, device=0)
    om = orc.Model(cfg, W)
    pcm = synth.synth_pcm(1, 16000, seed=5)[0]
    r = gm.transcribe_pcm([pcm], "tdt")[0]
    t, _ = oracle_clip(orc, om, pcm, cfg.ctc_vocab_size - 1)
    assert r["token_ids"] == t["ids"][0, :t["lens"][0]].tolist() and r["text"] == ""
    with pytest.raises(RuntimeError, match="vocabulary"):
        gm.transcribe_pcm([pcm], "tdt", boost_phrases=["hello"])
    with pytest.raises(RuntimeError, match="vocabulary"):
        gm.set_boost_phrases(["hello"])
    gm.set_boost_tokens([[1, 2]], 2.0)
    assert gm.boost_trie_size() == 3
    gm.set_boost_tokens([])
    gm.close()


def test_long_utterance_uses_the_global_scratch_attention(pair, orc):
    """Beyond ~85 s a [32][T] score block no longer fits LDS: the attention kernel then keeps its score blocks in global scratch
    (same code, same bits).  A 100 s clip (T = 1251 frames) must decode exactly like the oracle; an 80 s clip (LDS path) too."""
    W, om, gm = pair
    for seconds in (80, 100):
        pcm = synth.synth_pcm(1, 16000 * seconds, seed=seconds)[0]
        enc = om.encoder(orc.mel(pcm)[None])
        assert enc.shape[1] == (1001, 1251)[seconds == 100]
        feats = gm.mel(pcm[None])
        G.assert_bits_equal(gm.encode(feats), enc, f"{seconds} s encoder output")
        t, c = om.tdt_greedy(enc), orc.ctc_greedy(om.ctc_logprobs(enc), om.cfg.ctc_vocab_size - 1)
        r = gm.transcribe_pcm([pcm], "tdt")[0]
        assert r["token_ids"] == t["ids"][0, :t["lens"][0]].tolist()
        r = gm.transcribe_pcm([pcm], "ctc")[0]
        assert r["token_ids"] == c["ids"][0, :c["lens"][0]].tolist() and len(r["token_ids"]) > 0


def test_strict_weights_at_upload(tmp_path):
    """to_gpu() checks every tensor it uploads: a missing tensor, a wrong shape and a wrong dtype are named in the error (the reference
    loads with strict = false and would run on uninitialised weights, transcribe.hpp:63)."""
    cfg = G.tiny(name="tiny-strict")
    W = synth.synth_weights(cfg, seed=42)
    name = "encoder_.layers_.1.conv_.pointwise_conv2_.weight"
    for what, mutate in (("missing", lambda d: d.pop(name)),
                         ("shape", lambda d: d.__setitem__(name, np.zeros((3, 5), np.float32))),
                         ("dtype", lambda d: d.__setitem__(name, d[name].astype(np.float16)))):
        d = dict(W)
        mutate(d)
        p = str(tmp_path / f"{what}.safetensors")
        synth.save_weights(p, d)
        with pytest.raises(RuntimeError, match="pointwise_conv2_"):
            capi.Model(p, cfg, device=0)
    # a TRANSPOSED matrix has the right element count and the wrong extents: refused by name (fc1 [ffn][d] vs fc2 [d][ffn])
    fc1 = "encoder_.layers_.0.ffn1_.fc1_.weight"
    d = dict(W)
    d[fc1] = np.ascontiguousarray(W[fc1].T)
    p = str(tmp_path / "transposed.safetensors")
    synth.save_weights(p, d)
    with pytest.raises(RuntimeError, match="fc1_.weight"):
        capi.Model(p, cfg, device=0)


def test_failed_to_gpu_is_not_sticky(tmp_path):
    """A failed upload must not leave the model marked resident: the second to_gpu() reports the same error again (it used to return
    PK_OK on a half-initialised model) and compute entry points keep answering PK_ERR_NO_DEVICE."""
    cfg = G.tiny(name="tiny-sticky")
    W = synth.synth_weights(cfg, seed=42)
    W.pop("encoder_.layers_.1.ffn2_.fc2_.bias")
    p = str(tmp_path / "broken.safetensors")
    synth.save_weights(p, W)
    m = capi.Model(p, cfg)                                # host-side load succeeds: tensors are checked when they are uploaded
    for _ in range(2):
        with pytest.raises(RuntimeError, match="fc2_.bias"):
            m.to_gpu(0)
    with pytest.raises(RuntimeError, match="not on a GPU"):
        m.mel(synth.synth_pcm(1, 16000, seed=1))
    m.close()
