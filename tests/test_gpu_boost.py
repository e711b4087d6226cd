"""GPU parity: phrase boosting (reference src/phrase_boost.cpp; TranscribeOptions.boost_phrases transcribe.hpp:41-42, :110-139).
The device-side ContextTrie (CSR) + boosted CTC / TDT greedy decoders against the oracle's restatement: token ids, frames and
lengths identical, confidences bit-identical; the reference's own KAT (empty trie == unboosted, tests/test_all.cpp:1369-1440)
through the C ABI; the one-call API with boost phrase STRINGS (Tokenizer::encode -> trie) and the C++ facade."""
import dataclasses

import numpy as np
import pytest

import gpu_common as G
from conftest import pk
from parakeet_cpp_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_pair(tmp_path_factory):
    return G.make_pair(tmp_path_factory.mktemp("tinyb"), G.tiny())


def enc_like(B, T, d, seed):
    x = np.random.default_rng(seed).standard_normal((B, T, d)).astype(np.float32)
    return (x - x.mean(-1, keepdims=True)) / x.std(-1, keepdims=True)


def same(g, o, B, what, keys=("ids", "start", "end")):
    assert np.array_equal(g["lens"], o["lens"]), (what, g["lens"], o["lens"])
    for b in range(B):
        n = o["lens"][b]
        for k in keys:
            assert np.array_equal(g[k][b, :n], o[k][b, :n]), (what, k, b)
        G.assert_bits_equal(g["conf"][b, :n], o["conf"][b, :n], what + " confidence")


def phrases_from(u, rng, V, blank, n_random=6):
    """Phrases that WILL be walked (prefixes / infixes of the unboosted output with a changed tail) plus random ones."""
    ph = []
    for b in range(len(u["lens"])):
        toks = u["ids"][b, :u["lens"][b]].tolist()
        if len(toks) >= 4:
            ph.append(toks[:2] + [int(rng.integers(0, V - 1))])
            ph.append(toks[1:3] + [int(rng.integers(0, V - 1)), int(rng.integers(0, V - 1))])
    for _ in range(n_random):
        ph.append([int(t) for t in rng.integers(0, V - 1, size=rng.integers(1, 7)) if t != blank] or [1])
    ph.append(ph[0][:1] + [5, 6, 7])
    return ph


@pytest.mark.parametrize("boost", [5.0, 0.75])
def test_ctc_boosted_matches_oracle(tiny_pair, orc, boost):
    W, om, gm = tiny_pair
    V, blank = om.cfg.ctc_vocab_size, om.cfg.ctc_vocab_size - 1
    enc = enc_like(5, 126, om.cfg.hidden_size, 1)
    lp = om.ctc_logprobs(enc)
    u = orc.ctc_greedy(lp, blank)
    ph = phrases_from(u, np.random.default_rng(3), V, blank)
    try:
        gm.set_boost_tokens(ph, boost)
        assert gm.boost_trie_size() == orc.Trie(ph).size()
        g = gm.ctc_decode(enc, return_logp=True)
    finally:
        gm.set_boost_tokens([])
    G.assert_bits_equal(g["logp"], lp, "ctc log-probs")
    o = orc.ctc_greedy_boosted(lp, blank, orc.Trie(ph), boost)
    same(g, o, 5, "boosted ctc")
    changed = sum(o["ids"][b, :o["lens"][b]].tolist() != u["ids"][b, :u["lens"][b]].tolist() for b in range(5))
    assert changed > 0, "degenerate test: the boost changed nothing"
    assert gm.boost_trie_size() == 0
    same(gm.ctc_decode(enc), u, 5, "ctc after clearing the boost")


def test_ctc_reference_kats_through_the_abi(tiny_pair, orc):
    """BoostedCTCDecode.EmptyTrieMatchesUnboosted / TimestampsEmptyTrieMatchesUnboosted: a root-only trie decodes as unboosted."""
    W, om, gm = tiny_pair
    enc = enc_like(3, 60, om.cfg.hidden_size, 2)
    u = gm.ctc_decode(enc)
    try:
        gm.set_boost_tokens([[]], 5.0)                       # boosting ON with a trie that holds only the root
        assert gm.boost_trie_size() == 1
        b = gm.ctc_decode(enc)
    finally:
        gm.set_boost_tokens([])
    same(b, u, 3, "root-only trie")
    assert u["lens"].sum() > 0


@pytest.mark.parametrize("boost", [5.0, 1.5])
def test_tdt_boosted_matches_oracle(tiny_pair, orc, boost):
    W, om, gm = tiny_pair
    V, blank = om.cfg.vocab_size, om.cfg.blank_id
    enc = enc_like(6, 48, om.cfg.hidden_size, 4)
    u = om.tdt_greedy(enc)
    ph = phrases_from(u, np.random.default_rng(5), V, blank)
    try:
        gm.set_boost_tokens(ph, boost)
        g = gm.tdt_decode(enc)
        with pytest.raises(RuntimeError):                    # 64+ token phrases are refused, the previous trie stays
            gm.set_boost_tokens([list(range(1, 66))], boost)
    finally:
        gm.set_boost_tokens([])
    o = om.tdt_greedy_boosted(enc, orc.Trie(ph), boost)
    assert not o["overflow"]
    same(g, o, 6, "boosted tdt")
    assert np.array_equal(g["steps"], o["steps"])
    changed = sum(o["ids"][b, :o["lens"][b]].tolist() != u["ids"][b, :u["lens"][b]].tolist() for b in range(6))
    assert changed > 0, "degenerate test: the boost changed nothing"
    same(gm.tdt_decode(enc), u, 6, "tdt after clearing the boost")


def test_tdt_root_only_trie_and_zero_boost(tiny_pair, orc):
    W, om, gm = tiny_pair
    enc = enc_like(4, 40, om.cfg.hidden_size, 6)
    u = gm.tdt_decode(enc)
    try:
        gm.set_boost_tokens([[]], 5.0)
        same(gm.tdt_decode(enc), u, 4, "root-only trie")
        gm.set_boost_tokens([u["ids"][0, :3].tolist(), [1, 2, 3]], 0.0)
        same(gm.tdt_decode(enc), u, 4, "zero boost")
    finally:
        gm.set_boost_tokens([])


def test_rnnt_refuses_boost(tmp_path_factory):
    cfg = G.tiny(head="rnnt", durations=[], joint_prefix="joint_.", ctc_vocab_size=0, name="tinyrnnt")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("rnntb"), cfg)
    enc = enc_like(1, 20, cfg.hidden_size, 7)
    try:
        gm.set_boost_tokens([[1, 2]], 5.0)
        with pytest.raises(RuntimeError, match="phrase boosting"):
            gm.tdt_decode(enc)
    finally:
        gm.set_boost_tokens([])
    gm.tdt_decode(enc)


def test_transcribe_with_boost_phrases_110m(tmp_path, orc):
    """Transcriber::transcribe(samples, opts) with opts.boost_phrases (transcribe.hpp:91-180) on 110m-shaped weights: phrase
    strings -> Tokenizer::encode -> trie -> boosted TDT / CTC == the oracle on the same pipeline; the per-call phrases do not
    stick to the model; the C++ facade (--boost) returns the same ids."""
    cfg = dataclasses.replace(pk.make_110m_config(), num_layers=2, name="110m-2L-boost")
    W = synth.synth_weights(cfg, seed=42)
    wp, vp, ap = str(tmp_path / "model.safetensors"), str(tmp_path / "vocab.txt"), str(tmp_path / "clip.wav")
    synth.save_weights(wp, W)
    pieces = synth.synth_vocab(1024)
    synth.save_vocab(vp, pieces)
    from parakeet_cpp_amd import capi
    gm = capi.Model(wp, cfg, vocab_path=vp, device=0)
    om = orc.Model(cfg, W)
    pcm = synth.synth_pcm(2, 48000, seed=21)
    pcm = (np.clip(pcm, -1, 1) * 32767.0).astype("<i2").astype(np.float32) / 32768.0
    enc = om.encoder(np.stack([orc.mel(p) for p in pcm]))
    u = om.tdt_greedy(enc)
    # phrases as TEXT: detokenised pieces of the unboosted output with another piece appended
    def text_of(ids):
        t = "".join(pieces[i] for i in ids).replace("▁", " ")
        return t[1:] if t.startswith(" ") else t
    toks = u["ids"][0, :u["lens"][0]].tolist()
    phrases = [text_of(toks[:2] + [77]), text_of([300, 301, 302]), text_of(toks[2:4] + [500, 501])]
    ph_ids = [gm.tokenize(p) for p in phrases]
    assert all(len(p) > 0 for p in ph_ids)
    trie = orc.Trie(ph_ids)
    want = {"tdt": om.tdt_greedy_boosted(enc, trie, 4.0), "ctc": orc.ctc_greedy_boosted(om.ctc_logprobs(enc), 1024, trie, 4.0)}
    base = gm.transcribe_pcm(list(pcm), "tdt")
    for dec in ("tdt", "ctc"):
        r = gm.transcribe_pcm(list(pcm), dec, timestamps=True, boost_phrases=phrases, boost_score=4.0)
        for b in range(2):
            n = want[dec]["lens"][b]
            assert r[b]["token_ids"] == want[dec]["ids"][b, :n].tolist(), dec
            assert r[b]["start"] == want[dec]["start"][b, :n].tolist() and r[b]["end"] == want[dec]["end"][b, :n].tolist()
            assert np.array_equal(np.float32(r[b]["conf"]), want[dec]["conf"][b, :n])
            assert r[b]["text"] == text_of(r[b]["token_ids"])
    assert gm.boost_trie_size() == 0                                          # per-call phrases do not stick
    again = gm.transcribe_pcm(list(pcm), "tdt")
    assert [x["token_ids"] for x in again] == [x["token_ids"] for x in base]
    assert [x["token_ids"] for x in base] == [u["ids"][b, :u["lens"][b]].tolist() for b in range(2)]
    assert any(want["tdt"]["ids"][b, :want["tdt"]["lens"][b]].tolist() != base[b]["token_ids"] for b in range(2)), "boost changed nothing"
    # TranscriberBoostEmptyMatchesUnboosted (tests/test_all.cpp:1445-1457)
    assert [x["token_ids"] for x in gm.transcribe_pcm(list(pcm), "tdt", boost_phrases=[])] == [x["token_ids"] for x in base]
    gm.close()


def test_batch_pipeline_uses_model_level_boost(tmp_path, orc):
    """pk_batch_run with pk_set_boost_tokens on the resident pipeline (the bench path) == oracle, BASELINE shape scaled to 8 x 10 s."""
    cfg = dataclasses.replace(pk.make_110m_config(), num_layers=1, name="110m-1L-boostbatch")
    W, om, gm = G.make_pair(tmp_path, cfg)
    from parakeet_cpp_amd import capi
    pcm = synth.synth_pcm(8, 160000, seed=3)
    enc = om.encoder(np.stack([orc.mel(p) for p in pcm]))
    u = om.tdt_greedy(enc)
    ph = phrases_from(u, np.random.default_rng(8), cfg.vocab_size, cfg.blank_id)
    o = om.tdt_greedy_boosted(enc, orc.Trie(ph), 3.0)
    try:
        gm.set_boost_tokens(ph, 3.0)
        bt = capi.Batch(gm, 8, 160000)
        bt.upload(pcm)
        bt.run("tdt")
        g = bt.results()
        bt.run("tdt"); bt.run("tdt")                         # the pipelined path (decode(k) under encoder(k+1)) gives the same
        g2 = bt.results()
        bt.close()
    finally:
        gm.set_boost_tokens([])
    same(g, o, 8, "batch boosted tdt")
    same(g2, o, 8, "pipelined batch boosted tdt")
    assert any(o["ids"][b, :o["lens"][b]].tolist() != u["ids"][b, :u["lens"][b]].tolist() for b in range(8))
