"""GPU parity: CTC head + greedy (pk_ctc_decode; src/ctc.cpp:12-127) and the on-device TDT / RNNT greedy loop
(pk_tdt_decode; src/tdt.cpp:36-201, src/rnnt.cpp:56-177).  Token ids, frames and lengths must be identical;
log-probs and confidences bit-identical."""
import dataclasses

import numpy as np
import pytest

import gpu_common as G
from conftest import pk

pytestmark = pytest.mark.gpu

V, BLANK = 65, 64


@pytest.fixture(scope="module")
def tiny_pair(tmp_path_factory):
    return G.make_pair(tmp_path_factory.mktemp("tiny"), G.tiny())


def enc_like(B, T, d, seed):
    x = np.random.default_rng(seed).standard_normal((B, T, d)).astype(np.float32)
    return (x - x.mean(-1, keepdims=True)) / x.std(-1, keepdims=True)


def test_ctc_bits_and_ids(tiny_pair, orc):
    W, om, gm = tiny_pair
    enc = enc_like(5, 126, om.cfg.hidden_size, 1)
    g = gm.ctc_decode(enc, return_logp=True)
    lp = om.ctc_logprobs(enc)
    G.assert_bits_equal(g["logp"], lp, "ctc log-probs")
    o = orc.ctc_greedy(lp, BLANK)
    assert np.array_equal(g["lens"], o["lens"])
    for b in range(5):
        n = o["lens"][b]
        assert np.array_equal(g["ids"][b, :n], o["ids"][b, :n])
        assert np.array_equal(g["start"][b, :n], o["start"][b, :n])
        assert np.array_equal(g["end"][b, :n], o["end"][b, :n])
        G.assert_bits_equal(g["conf"][b, :n], o["conf"][b, :n], "ctc confidence")
    assert o["lens"].sum() > 0, "degenerate test: nothing decoded"


def check_tdt(gm, om, enc):
    g = gm.tdt_decode(enc)
    o = om.tdt_greedy(enc, max_steps=0)
    assert not o["overflow"]
    assert np.array_equal(g["lens"], o["lens"]), (g["lens"], o["lens"])
    assert np.array_equal(g["steps"], o["steps"])
    for b in range(enc.shape[0]):
        n = o["lens"][b]
        assert np.array_equal(g["ids"][b, :n], o["ids"][b, :n]), b
        assert np.array_equal(g["start"][b, :n], o["start"][b, :n])
        assert np.array_equal(g["end"][b, :n], o["end"][b, :n])
        G.assert_bits_equal(g["conf"][b, :n], o["conf"][b, :n], "tdt confidence")
    return o


def test_tdt_ids_frames_conf(tiny_pair):
    W, om, gm = tiny_pair
    o = check_tdt(gm, om, enc_like(7, 126, om.cfg.hidden_size, 2))
    assert o["lens"].sum() > 5, "degenerate test: nothing decoded"
    check_tdt(gm, om, enc_like(1, 13, om.cfg.hidden_size, 3))       # tiny T, single utterance
    check_tdt(gm, om, enc_like(64, 40, om.cfg.hidden_size, 4))      # full batch width


def test_tdt_two_lstm_layers(tmp_path_factory):
    cfg = G.tiny(num_lstm_layers=2, name="tiny2l")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("t2"), cfg, seed=11)
    o = check_tdt(gm, om, enc_like(4, 60, cfg.hidden_size, 5))
    assert o["lens"].sum() > 0


def test_rnnt_head(tmp_path_factory):
    cfg = G.tiny(head="rnnt", durations=[], joint_prefix="joint_.", ctc_vocab_size=0, name="tinyrnnt")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("tr"), cfg, seed=12)
    enc = enc_like(4, 50, cfg.hidden_size, 6)
    g = gm.tdt_decode(enc)
    o = om.rnnt_greedy(enc)
    assert np.array_equal(g["lens"], o["lens"])
    for b in range(4):
        n = o["lens"][b]
        assert np.array_equal(g["ids"][b, :n], o["ids"][b, :n])
        assert np.array_equal(g["start"][b, :n], o["start"][b, :n])
    assert o["lens"].sum() > 0


@pytest.mark.parametrize("B", [1, 5, 64, 70])
def test_persistent_loop_equals_per_phase_loop(tiny_pair, B):
    """The single-launch decode loop (kernels/decode_persist.hip: grid barriers, system-scope exchange) and the hipGraph replay against the
    per-phase launches of the same device code (pk_model_set_decode_loop) and against the oracle: every output word identical."""
    W, om, gm = tiny_pair
    enc = enc_like(B, 57, om.cfg.hidden_size, 100 + B)
    gm.set_decode_loop("phases")
    a = gm.tdt_decode(enc)
    try:
        for mode in ("persistent", "graph"):
            gm.set_decode_loop(mode)
            b = gm.tdt_decode(enc)
            for k in ("lens", "steps", "ids", "start", "end"):
                assert np.array_equal(a[k], b[k]), (mode, k)
            G.assert_bits_equal(a["conf"], b["conf"], "confidence")
    finally:
        gm.set_decode_loop("phases")
    o = check_tdt(gm, om, enc)
    assert o["lens"].sum() > 0


@pytest.mark.parametrize("B", [300, 1100, 2048, 2100])
def test_prediction_net_caching_large_lockstep_batches(tiny_pair, B):
    """Prediction-net caching (kernels.hpp TdtState::need): after a blank the LSTM cells and pred_proj of the next step are skipped and the launches
    run over the compacted list of utterances that emitted a token.  The list is built per workgroup from 1 .. 8 flags per thread (B <= 2048);
    larger lock-step batches (2100) run every row.  Utterances of the same batch repeat with a period of 37 so the oracle decodes 37, and every
    copy must carry the same words -- whatever tile row the compaction gave it."""
    W, om, gm = tiny_pair
    base = enc_like(37, 41, om.cfg.hidden_size, 7)
    enc = np.ascontiguousarray(base[np.arange(B) % 37])
    g = gm.tdt_decode(enc)
    o = om.tdt_greedy(base, max_steps=0)
    assert not o["overflow"] and o["lens"].sum() > 0
    idx = np.arange(B) % 37
    assert np.array_equal(g["lens"], o["lens"][idx])
    assert np.array_equal(g["steps"], o["steps"][idx])
    for b in range(B):
        n = o["lens"][idx[b]]
        for k in ("ids", "start", "end"):
            assert np.array_equal(g[k][b, :n], o[k][idx[b], :n]), (b, k)
        G.assert_bits_equal(g["conf"][b, :n], o["conf"][idx[b], :n], "tdt confidence")


def test_persistent_loop_110m_heads_and_two_layers(tmp_path_factory):
    for cfg in (dataclasses.replace(pk.make_110m_config(), num_layers=1, name="110m-1L-persist"),
                G.tiny(name="tiny-2lstm-persist", num_lstm_layers=2)):
        W, om, gm = G.make_pair(tmp_path_factory.mktemp("persist"), cfg, seed=9)
        enc = enc_like(16 if cfg.hidden_size == 512 else 7, 126, cfg.hidden_size, 3)
        gm.set_decode_loop("phases")
        a = gm.tdt_decode(enc)
        gm.set_decode_loop("persistent")
        try:
            b = gm.tdt_decode(enc)
        finally:
            gm.set_decode_loop("phases")
        for k in ("lens", "steps", "ids", "start", "end"):
            assert np.array_equal(a[k], b[k]), (cfg.name, k)
        G.assert_bits_equal(a["conf"], b["conf"], "confidence")
        check_tdt(gm, om, enc)


def _pair_with(tmpdir, cfg, seed, edit):
    """make_pair with the synthetic weights edited before either model is built."""
    import os
    import oracle
    from parakeet_cpp_amd import capi, synth
    W = synth.synth_weights(cfg, seed=seed)
    edit(W)
    wp = os.path.join(str(tmpdir), f"{cfg.name}_{seed}.safetensors")
    synth.save_weights(wp, W)
    return oracle.Model(cfg, W), capi.Model(wp, cfg, device=0)


def test_single_utterance_frame_window(tmp_path_factory):
    """One utterance per call decodes through the frame window (TdtState::F: the joint of the next frames in the heads product's spare rows, tdt_decide walking
    through the blanks inside it).  Every output word -- ids, frames, confidences, the count of joint evaluations -- against the oracle's one-decision-at-a-time loop:
    ordinary weights, weights pushed towards blanks of duration 1 (long walks, windows that run out), blanks of duration 4 (the window ends at the first blank),
    sequences shorter than the window, the RNNT head (no durations), two LSTM layers, and the 110m decoder shapes (K = 640 kernels, vocabulary 1025)."""
    tmp = tmp_path_factory.mktemp("win")
    pre = "tdt_joint_."

    def towards(blank_bias, dur, dur_bias):
        def edit(W):
            b = W[pre + "label_proj_.bias"]; b[-1] += blank_bias
            if dur is not None:
                W[pre + "duration_proj_.bias"][dur] += dur_bias
        return edit

    cases = [("plain", G.tiny(name="tiny-win-plain"), lambda W: None),
             ("blank-dur1", G.tiny(name="tiny-win-b1"), towards(1.5, 1, 3.0)),
             ("blank-dur4", G.tiny(name="tiny-win-b4"), towards(1.5, 4, 3.0)),
             ("all-blank", G.tiny(name="tiny-win-all"), towards(30.0, 1, 3.0)),
             ("two-lstm", G.tiny(name="tiny-win-2l", num_lstm_layers=2), towards(1.0, 1, 1.0))]
    n_blank_steps = 0
    for tag, cfg, edit in cases:
        om, gm = _pair_with(tmp, cfg, 21, edit)
        for B, T, seed in ((1, 126, 1), (1, 57, 2), (1, 5, 3), (1, 1, 4), (2, 40, 5), (3, 33, 6)):
            o = check_tdt(gm, om, enc_like(B, T, cfg.hidden_size, seed))
            n_blank_steps += int(o["steps"].sum() - o["lens"].sum())
    assert n_blank_steps > 200, "degenerate test: the walk was hardly taken"
    cfg = G.tiny(head="rnnt", durations=[], joint_prefix="joint_.", ctc_vocab_size=0, name="tiny-win-rnnt")
    om, gm = _pair_with(tmp, cfg, 22, lambda W: W["joint_.out_proj_.bias"].__setitem__(-1, W["joint_.out_proj_.bias"][-1] + 1.0))
    for T, seed in ((50, 7), (3, 8)):
        enc = enc_like(1, T, cfg.hidden_size, seed)
        g, o = gm.tdt_decode(enc), om.rnnt_greedy(enc)
        assert np.array_equal(g["lens"], o["lens"])
        n = o["lens"][0]
        assert np.array_equal(g["ids"][0, :n], o["ids"][0, :n]) and np.array_equal(g["start"][0, :n], o["start"][0, :n])
    cfg = dataclasses.replace(pk.make_110m_config(), num_layers=1, name="110m-1L-win")
    om, gm = _pair_with(tmp, cfg, 23, towards(2.0, 2, 2.0))
    for T, seed in ((126, 9), (40, 10)):
        o = check_tdt(gm, om, enc_like(1, T, cfg.hidden_size, seed))
        assert o["steps"].sum() > o["lens"].sum()


def test_teacher_forced_scores_bit_identical(tiny_pair):
    """pk_tdt_score == orc_tdt_score row for row: along the oracle's own greedy path (then the walk reproduces the greedy decode), and along
    an ARBITRARY path of random labels / durations (states no greedy decode visits) -- /root/reference/src/tdt.cpp:15-24,62-106."""
    W, om, gm = tiny_pair
    enc = enc_like(1, 60, om.cfg.hidden_size, 11)[0]
    o = om.tdt_score(enc)
    g = gm.tdt_score(enc, o["labels"], o["dur_idx"])
    assert g["n"] == o["n"] > 20
    G.assert_bits_equal(g["label_lp"], o["label_lp"], "label log-prob rows along the greedy path")
    G.assert_bits_equal(g["dur_lp"], o["dur_lp"], "duration log-prob rows along the greedy path")
    assert (o["labels"] != BLANK).sum() > 3, "degenerate test: nothing decoded"
    rng = np.random.default_rng(0)
    n = 150
    lab = np.where(rng.random(n) < 0.5, BLANK, rng.integers(0, BLANK, n)).astype(np.int32)
    dur = rng.integers(0, len(om.cfg.durations), n).astype(np.int32)
    o2 = om.tdt_score(enc, lab, dur)
    g2 = gm.tdt_score(enc, lab, dur)
    assert g2["n"] == o2["n"] and 10 < o2["n"] <= n
    G.assert_bits_equal(g2["label_lp"], o2["label_lp"], "label log-prob rows along a random path")
    G.assert_bits_equal(g2["dur_lp"], o2["dur_lp"], "duration log-prob rows along a random path")
