"""Phrase boosting, CPU side: the oracle's ContextTrie and boosted greedy decoders against the reference's own unit tests
(tests/test_all.cpp:1280-1440: ContextTrie.*, BoostedCTCDecode.*), against an independent set/dict restatement on random
inputs, and the boosted TDT loop through properties (empty trie / zero boost == unboosted, a boost that flips exactly the
first decision, an overwhelming boost)."""
import numpy as np
import pytest

from boost_ref import PyTrie, ctc_boosted


# ---- ContextTrie (tests/test_all.cpp:1280-1353) -------------------------------------------------------------
def test_trie_empty(orc):
    t = orc.Trie()
    assert t.size() == 1                                   # just the root
    assert t.boosted_tokens({0}) == set()


def test_trie_insert_and_size(orc):
    t = orc.Trie([[10, 20, 30]])
    assert t.size() == 4
    t.insert([])                                           # empty phrases are ignored (:12-13)
    assert t.size() == 4
    t.insert([10, 20, 30])                                 # re-inserting adds nothing
    assert t.size() == 4


def test_trie_boosted_tokens_and_advance(orc):
    t = orc.Trie([[10, 20, 30], [10, 25]])
    assert t.boosted_tokens({0}) == {10}
    nxt = t.advance({0}, 10)
    assert 0 in nxt and len(nxt) == 2
    assert t.boosted_tokens(nxt) == {10, 20, 25}
    assert t.advance({0}, 999) == {0}                      # AdvanceNonMatchingToken: only the root survives
    assert t.boosted_tokens({-5, 12345}) == set()          # out-of-range states are skipped (:43-44)


def test_trie_multiple_phrases(orc):
    t = orc.Trie([[10, 20], [10, 30], [40, 50]])
    assert t.boosted_tokens({0}) == {10, 40}
    assert t.boosted_tokens(t.advance({0}, 10)) == {20, 30, 10, 40}


# ---- boosted CTC (tests/test_all.cpp:1369-1440) -------------------------------------------------------------
def _pattern_lp(pattern, vocab=1025):
    lp = np.full((1, len(pattern), vocab), -10.0, np.float32)
    for t, p in enumerate(pattern):
        lp[0, t, p] = 0.0
    return lp


def test_ctc_empty_trie_matches_unboosted(orc):
    lp = _pattern_lp([5, 5, 1024, 8, 8, 8])
    u, b = orc.ctc_greedy(lp, 1024), orc.ctc_greedy_boosted(lp, 1024, orc.Trie(), 5.0)
    for k in ("ids", "lens", "start", "end", "conf"):
        assert np.array_equal(u[k], b[k]), k
    assert b["ids"][0, :2].tolist() == [5, 8]


def test_ctc_boost_flips_decision(orc):
    lp = np.full((1, 3, 1025), -10.0, np.float32)
    lp[0, 0, 42], lp[0, 0, 43], lp[0, 0, 1024] = -0.1, -0.2, -5.0
    lp[0, 1, 1024] = lp[0, 2, 1024] = 0.0
    u = orc.ctc_greedy(lp, 1024)
    assert u["lens"][0] == 1 and u["ids"][0, 0] == 42
    b = orc.ctc_greedy_boosted(lp, 1024, orc.Trie([[43]]), 5.0)
    assert b["lens"][0] == 1 and b["ids"][0, 0] == 43
    assert b["conf"][0, 0] == orc.math_v("exp", np.float32([-0.2]))[0]     # confidence = exp(UNBOOSTED log-prob) (:151-152)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ctc_boosted_vs_independent_restatement(orc, seed):
    rng = np.random.default_rng(seed)
    B, T, V, blank = 3, 60, 48, 47
    logits = rng.standard_normal((B, T, V)).astype(np.float32) * 2.0
    logits[..., blank] += 1.5                                               # a realistic share of blanks
    lp = (logits - np.log(np.exp(logits).sum(-1, keepdims=True))).astype(np.float32)
    phrases = [rng.integers(0, V - 1, size=rng.integers(1, 6)).tolist() for _ in range(12)]
    phrases += [phrases[0][:1] + [3, 4], [blank, 2]]                        # shared prefixes; a phrase through the blank id
    boost = [5.0, 1.25, 0.0][seed]
    got = orc.ctc_greedy_boosted(lp, blank, orc.Trie(phrases), boost)
    pt = PyTrie(phrases)
    assert orc.Trie(phrases).size() == pt.size()
    n_changed = 0
    for b in range(B):
        ids, st, en, lps = ctc_boosted(lp[b], pt, boost, blank)
        n = got["lens"][b]
        assert n == len(ids)
        assert got["ids"][b, :n].tolist() == ids and got["start"][b, :n].tolist() == st and got["end"][b, :n].tolist() == en
        assert np.array_equal(got["conf"][b, :n], orc.math_v("exp", np.float32(lps)))
        u = orc.ctc_greedy(lp[b:b + 1], blank)
        n_changed += u["ids"][0, :u["lens"][0]].tolist() != ids
    assert (n_changed > 0) == (boost > 0.0), "boost 0 changes nothing; a positive boost must change something here"


# ---- boosted TDT (src/phrase_boost.cpp:177-350) through properties ------------------------------------------
def _enc(B, T, d, seed):
    x = np.random.default_rng(seed).standard_normal((B, T, d)).astype(np.float32)
    return (x - x.mean(-1, keepdims=True)) / x.std(-1, keepdims=True)


def _same(a, b):
    assert np.array_equal(a["lens"], b["lens"])
    for k in ("ids", "start", "end", "conf", "steps"):
        assert np.array_equal(a[k], b[k]), k


def test_tdt_empty_trie_and_zero_boost_match_unboosted(orc, tiny_oracle):
    enc = _enc(3, 40, tiny_oracle.cfg.hidden_size, 5)
    u = tiny_oracle.tdt_greedy(enc)
    assert u["lens"].sum() > 0
    _same(u, tiny_oracle.tdt_greedy_boosted(enc, orc.Trie(), 5.0))
    toks = u["ids"][0, :u["lens"][0]].tolist()
    _same(u, tiny_oracle.tdt_greedy_boosted(enc, orc.Trie([toks[:3], [1, 2, 3]]), 0.0))


def test_tdt_boost_flips_exactly_the_first_decision(orc, tiny_oracle):
    enc = _enc(1, 30, tiny_oracle.cfg.hidden_size, 7)
    u = tiny_oracle.tdt_greedy(enc, first_logp=True)
    lp = u["first_logp"][0]
    order = np.argsort(-lp, kind="stable")
    top, second = int(order[0]), int(order[1])
    gap = float(lp[top] - lp[second])
    assert gap > 0
    b = tiny_oracle.tdt_greedy_boosted(enc, orc.Trie([[second]]), gap * 1.5 + 1e-3)
    first = b["ids"][0, 0] if top != tiny_oracle.cfg.blank_id else None
    if second != tiny_oracle.cfg.blank_id:
        assert b["lens"][0] > 0 and b["ids"][0, 0] == second and b["start"][0, 0] == 0
        assert b["conf"][0, 0] == orc.math_v("exp", np.float32([lp[second]]))[0]      # raw, unboosted probability (:313-315)
    else:
        assert first is None or first != top
    nb = tiny_oracle.tdt_greedy_boosted(enc, orc.Trie([[second]]), gap * 0.5)           # half the gap: the first decision stands
    if top != tiny_oracle.cfg.blank_id:
        assert nb["ids"][0, 0] == top


def test_tdt_overwhelming_boost_emits_only_the_phrase_token(orc, tiny_oracle):
    enc = _enc(2, 25, tiny_oracle.cfg.hidden_size, 9)
    b = tiny_oracle.tdt_greedy_boosted(enc, orc.Trie([[7]]), 1e6)
    assert b["lens"].min() > 0
    for i in range(2):
        assert set(b["ids"][i, :b["lens"][i]].tolist()) == {7}
