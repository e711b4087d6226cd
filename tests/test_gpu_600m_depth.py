"""FULL-DEPTH parity of BASELINE configs[2] (tdt-600m, 24 layers, 30 s clips) against tests/golden/tdt600m_depth24_seed42.npz -- the CPU
oracle's outputs (and the reference's own code's token ids) for the first clips of bench.py's rank-0 batch, generated once in the
authoring container by tools/make_golden_600m.py (the 24-layer oracle pass is too long for the GPU box's test run).

 * fp32 (the bit contract): mel features, subsampling output and the encoder stream after EVERY one of the 24 blocks carry the oracle's
   checksums (sum and xor of all uint32 bit patterns), TDT ids / frames / confidences are identical, the per-clip minimum top-1/top-2 margin
   is bit-equal, and the ids equal those of the reference's own tdt_greedy_decode.
 * bf16 mode (the tolerance contract): two correct bf16 implementations differ wherever a 1-ulp fp32 difference flips a bf16 rounding
   (2^-8 relative), and that compounds over 24 layers.  Stated and asserted here:
     - drift curve: per layer, on the sampled rows of clip 0, max|gpu - oracle_bf16| <= DRIFT_MAX * max|x| and the mean <= DRIFT_MEAN * max|x|
       (the curve is printed; it must also stay below the bf16-vs-fp32 gap of the oracle itself -- the GPU is closer to the bf16 oracle than
       bf16 is to fp32);
     - tokens: the synthetic random-weight model decides with margins down to 5e-5 (a trained model's are ~1), so "x % of the tokens agree"
       is not a property of the implementation.  The property that IS: walking the oracle's decisions in order, the GPU's tokens may leave the
       oracle's only at a decision whose top-1/top-2 margin is below MARGIN_TOL (the logit error of the mode); every token before the first
       such near-tie must be identical.  The minimum margin of the GPU's own decode is reported next to it.
"""
import dataclasses
import os

import numpy as np
import pytest

from conftest import ROOT, pk
from tolerance import first_divergence
from parakeet_cpp_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "tdt600m_depth24_seed42.npz")
DRIFT_MAX, DRIFT_MEAN = 2e-2, 3e-3          # of max|x| of the layer (observed at layer 24: 8e-3 / 1.4e-3, profiles/r03_600m_depth24_parity.txt)
MARGIN_TOL = 2e-2                           # label log-prob error class of the bf16 mode at depth 24 (observed first divergences at margins 3.6e-3 .. 7.6e-3)


def bits_sum_xor(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()
    return np.array([int(u.astype(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF), int(np.bitwise_xor.reduce(u))], np.uint64)


@pytest.fixture(scope="module")
def gold():
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/tdt600m_depth24_seed42.npz is missing (tools/make_golden_600m.py)")
    return np.load(GOLD, allow_pickle=False)


@pytest.fixture(scope="module")
def setup(gold, tmp_path_factory):
    cfg = pk.make_tdt_600m_config()
    W = synth.synth_weights(cfg, seed=int(gold["weights_seed"]))
    names = [str(n) for n in gold["weight_digest_names"]]
    dig = np.array([float(np.asarray(W[n], np.float64).sum()) for n in names])
    assert np.array_equal(dig, gold["weight_digest"]), "the regenerated synthetic weights differ from the fixture's (numpy version drift?)"
    n_clips, n = int(gold["n_clips"]), int(gold["n_samples"])
    pcm = synth.synth_pcm(int(gold["pcm_batch"]), n, seed=int(gold["pcm_seed"]))[:n_clips]
    assert np.array_equal(np.asarray(pcm, np.float64).sum(axis=1), gold["pcm_digest"]), "the regenerated clips differ from the fixture's"
    wp = str(tmp_path_factory.mktemp("d24") / "tdt600m.safetensors")
    synth.save_weights(wp, W)
    return cfg, wp, pcm


def tokens(r, b):
    return r["ids"][b, : r["lens"][b]].tolist()


def test_fp32_every_layer_bit_identical_at_depth_24(gold, setup):
    from parakeet_cpp_amd import capi
    cfg, wp, pcm = setup
    gm = capi.Model(wp, cfg, device=0)
    feats = gm.mel(pcm)
    assert np.array_equal(bits_sum_xor(feats), gold["fp32_feats_bits"]), "mel features (128 bins, 30 s)"
    assert np.array_equal(bits_sum_xor(gm.subsample(feats)), gold["fp32_sub_bits"]), "subsampling output"
    first_bad = None
    for l in range(cfg.num_layers):
        x = gm.encode(feats, stop_layer=l + 1, stop_stage=0)
        if not np.array_equal(bits_sum_xor(x), gold["fp32_layer_bits"][l]) and first_bad is None:
            first_bad = l
        if l == 0:
            assert np.array_equal(x[0, :: int(gold["row_step"])].view(np.uint32), gold["fp32_rows"][0].view(np.uint32)), "layer 0 rows"
    assert first_bad is None, f"encoder stream differs from the oracle from layer {first_bad} on"
    enc = gm.encode(feats)
    assert np.array_equal(bits_sum_xor(enc), gold["fp32_enc_bits"]), "24-layer encoder output"
    g = gm.tdt_decode(enc)
    assert np.array_equal(g["lens"], gold["fp32_lens"]) and np.array_equal(g["steps"], gold["fp32_steps"])
    for b in range(len(pcm)):
        n = int(gold["fp32_lens"][b])
        for k in ("ids", "start", "end"):
            assert np.array_equal(g[k][b, :n], gold["fp32_" + k][b, :n]), (k, b)
        assert np.array_equal(g["conf"][b, :n].view(np.uint32), gold["fp32_conf_bits"][b, :n]), ("confidence bits", b)
        if "ref_ids" in gold.files and bool(gold["ref_ids_equal_oracle"][b]):
            assert tokens(g, b) == gold["ref_ids"][b, : int(gold["ref_lens"][b])].tolist(), "ids of the reference's own tdt_greedy_decode"
    assert np.array_equal(g["min_margin"].view(np.uint32), gold["fp32_min_margin"].view(np.uint32)), "min top-1/top-2 margin"
    print(f"fp32 depth 24: {len(pcm)} x 30 s clips, tokens {g['lens'].tolist()}, min margins {g['min_margin'].tolist()}")
    if "ref_ids" in gold.files:
        assert gold["ref_ids_equal_oracle"].all(), "fixture: the oracle's ids differed from the reference code's"
    gm.close()


def test_bf16_drift_curve_and_token_contract_at_depth_24(gold, setup):
    from parakeet_cpp_amd import capi
    cfg, wp, pcm = setup
    cfg16 = dataclasses.replace(cfg, gemm_bf16=True, name="tdt-600m-bf16-d24")
    gm = capi.Model(wp, cfg16, device=0)
    feats = gm.mel(pcm)
    step = int(gold["row_step"])
    curve = []
    for l in range(cfg.num_layers):
        x = gm.encode(feats[:1], stop_layer=l + 1, stop_stage=0)[0, ::step]
        d = np.abs(x - gold["bf16_rows"][l])
        mx = float(gold["bf16_layer_absmax"][l])
        gap = np.abs(gold["bf16_rows"][l] - gold["fp32_rows"][l])
        curve.append((l, d.max() / mx, d.mean() / mx, gap.max() / mx, gap.mean() / mx))
    print("layer: gpu-vs-oracle(bf16) max, mean | oracle bf16-vs-fp32 max, mean   (fractions of max|x| of the layer)")
    for l, a, b, c, e in curve:
        print(f"  {l:2d}: {a:.2e} {b:.2e} | {c:.2e} {e:.2e}")
    for l, a, b, c, e in curve:
        assert a <= DRIFT_MAX and b <= DRIFT_MEAN, f"layer {l}: drift max {a:.3e} mean {b:.3e} of max|x|"
    assert curve[-1][2] < curve[-1][4], "after 24 layers the GPU must be closer to the bf16 oracle than the bf16 oracle is to fp32"
    enc = gm.encode(feats)
    d_all = np.abs(enc[:, ::step] - gold["bf16_enc_rows_all"])
    print(f"final encoder rows, all clips: max {d_all.max():.3e} mean {d_all.mean():.3e}")
    g = gm.tdt_decode(enc)
    report = []
    for b in range(len(pcm)):
        got = tokens(g, b)
        at, mg = first_divergence(got, gold["bf16_step_label"][b], gold["bf16_step_margin"][b], cfg.blank_id,
                                  got_frames=(g["start"][b], g["end"][b]), oracle_frames=(gold["bf16_start"][b], gold["bf16_end"][b]))
        want_n = int(gold["bf16_lens"][b])
        report.append((b, len(got), want_n, at, mg, float(g["min_margin"][b])))
        if at is not None:
            assert mg <= MARGIN_TOL, (f"clip {b}: the GPU's tokens leave the bf16 oracle's at token {at}, but the closest decision there has margin "
                                      f"{mg:.3e} > {MARGIN_TOL}: not a near-tie")
    print("clip: gpu tokens, oracle tokens, first differing token (None = identical), oracle margin there, gpu min margin")
    for r in report:
        print("  ", r)
    assert all(r[1] > 0 for r in report)
    gm.close()


# ---- teacher-forced joint scores: the logits-level half of the configs[2] statement ------------------------------------------------------
SCORE = os.path.join(ROOT, "tests", "golden", "tdt600m_depth24_score_seed42.npz")
LOGP_TOL, LOGP_MEAN = 3e-2, 8e-3   # bf16 mode at depth 24: max / mean |log-prob(gpu) - log-prob(bf16 oracle)| along the oracle's path, encoder drift included
                                   # (observed on the three clips: max 2.06e-2 .. 2.48e-2, mean 4.6e-3 .. 5.0e-3; profiles/r04_600m_teacher_forced.txt)


@pytest.fixture(scope="module")
def score():
    if not os.path.exists(SCORE):
        pytest.skip("tests/golden/tdt600m_depth24_score_seed42.npz is missing (tools/make_golden_600m_score.py)")
    return np.load(SCORE, allow_pickle=False)


def test_fp32_teacher_forced_rows_bit_identical(gold, score, setup):
    """pk_tdt_score walks the fp32 oracle's greedy path (every step's label AND duration given): the label log-prob ROW of every step carries
    the oracle's bit checksums, its top-8 and the duration log-probs are bit-equal -- /root/reference/src/tdt.cpp:15-24,62-106."""
    from parakeet_cpp_amd import capi
    cfg, wp, pcm = setup
    assert np.array_equal(score["pcm_digest"], gold["pcm_digest"])
    gm = capi.Model(wp, cfg, device=0)
    enc = gm.encode(gm.mel(pcm))
    for b in range(len(pcm)):
        n = int(score["fp32_n"][b])
        r = gm.tdt_score(enc[b], score["fp32_labels"][b, :n], score["fp32_dur_idx"][b, :n])
        assert r["n"] == n, f"clip {b}: {r['n']} steps walked, the oracle's path has {n}"
        u = r["label_lp"].view(np.uint32)
        assert np.array_equal(np.bitwise_xor.reduce(u, axis=1), score["fp32_row_xor"][b, :n]), f"clip {b}: label log-prob rows (xor of bits)"
        assert np.array_equal(u.astype(np.uint64).sum(axis=1), score["fp32_row_sum"][b, :n]), f"clip {b}: label log-prob rows (sum of bits)"
        top = np.take_along_axis(r["label_lp"], score["fp32_top_ids"][b, :n].astype(np.int64), axis=1)
        assert np.array_equal(top.view(np.uint32), score["fp32_top_lp"][b, :n].view(np.uint32)), f"clip {b}: top-8 label log-probs"
        assert np.array_equal(r["dur_lp"].view(np.uint32), score["fp32_dur_lp"][b, :n].view(np.uint32)), f"clip {b}: duration log-probs"
        assert np.array_equal(r["label_lp"].argmax(axis=1), score["fp32_labels"][b, :n]), f"clip {b}: the GPU's own argmax along the path"
    gm.close()


def test_bf16_teacher_forced_logits_within_bound(score, setup):
    """The tolerance statement of the bf16 mode at the logits: the GPU walks the bf16 ORACLE's decision path of every clip to the end (so a
    near-tie does not end the comparison), and at EVERY step |delta log-prob| <= LOGP_TOL on the oracle's top-8 labels and on all duration
    log-probs; every decision whose margin exceeds 2 x LOGP_TOL (both sides can move by the bound) is the GPU's own argmax too."""
    from parakeet_cpp_amd import capi
    cfg, wp, pcm = setup
    gm = capi.Model(wp, dataclasses.replace(cfg, gemm_bf16=True, name="tdt-600m-bf16-score"), device=0)
    enc = gm.encode(gm.mel(pcm))
    worst, n_dec, n_clear, n_agree_all, rep = 0.0, 0, 0, 0, []
    for b in range(len(pcm)):
        n = int(score["bf16_n"][b])
        lab, dur = score["bf16_labels"][b, :n], score["bf16_dur_idx"][b, :n]
        r = gm.tdt_score(enc[b], lab, dur)
        assert r["n"] == n, f"clip {b}: {r['n']} steps walked, the oracle's path has {n}"
        top = np.take_along_axis(r["label_lp"], score["bf16_top_ids"][b, :n].astype(np.int64), axis=1)
        d_lab = np.abs(top - score["bf16_top_lp"][b, :n])
        d_dur = np.abs(r["dur_lp"] - score["bf16_dur_lp"][b, :n])
        mg = score["bf16_margin"][b, :n]
        g_lab, g_dur = r["label_lp"].argmax(axis=1), r["dur_lp"].argmax(axis=1)
        clear_l, clear_d = mg[:, 0] > 2 * LOGP_TOL, mg[:, 1] > 2 * LOGP_TOL
        assert np.array_equal(g_lab[clear_l], lab[clear_l]), f"clip {b}: a label decision with margin > {2 * LOGP_TOL} differs"
        assert np.array_equal(g_dur[clear_d], dur[clear_d]), f"clip {b}: a duration decision with margin > {2 * LOGP_TOL} differs"
        worst = max(worst, float(d_lab.max()), float(d_dur.max()))
        n_dec += 2 * n; n_clear += int(clear_l.sum() + clear_d.sum()); n_agree_all += int((g_lab == lab).sum() + (g_dur == dur).sum())
        rep.append((b, n, float(d_lab.max()), float(d_lab.mean()), float(d_dur.max()), float(d_dur.mean()), int((g_lab != lab).sum()), int((g_dur != dur).sum())))
    print("clip: steps, label max / mean |dlogp| (top-8), duration max / mean |dlogp|, label / duration argmax flips along the path")
    for x in rep:
        print("  ", x)
    print(f"bf16 teacher-forced: max |dlogp| {worst:.3e} (bound {LOGP_TOL}); {n_clear} of {n_dec} decisions have margin > {2 * LOGP_TOL} (all agree); "
          f"{n_agree_all} of {n_dec} agree in all")
    gm.close()
    assert worst <= LOGP_TOL, f"max |delta log-prob| {worst:.3e} > {LOGP_TOL}"
    assert max(max(x[3], x[5]) for x in rep) <= LOGP_MEAN, f"mean |delta log-prob| > {LOGP_MEAN}"


FP32_RATIO = 1.25    # the GPU's bf16 mode may be at most this much farther from the fp32 reference arithmetic than the mode's own specification is


def test_bf16_logits_vs_fp32_reference_path(score, setup):
    """The bf16 mode against the REFERENCE's arithmetic (fp32; /root/reference/src/tdt.cpp:15-24,62-106), not only against its own
    specification: the bf16 GPU model walks the FP32 oracle's decision path of every clip (score["fp32_labels"] / ["fp32_dur_idx"]) and its
    log-probs are compared with the fp32 oracle's on that oracle's top-8 labels and on all durations.  The yardstick is the same walk by the
    bf16-mode ORACLE (fixture keys bf16_on_fp32_*, tools/make_golden_600m_score.py): that distance is what the mode costs by definition; the
    GPU's may exceed it by at most FP32_RATIO (max and mean).  Both pairs of numbers are printed (and carried in bench.py's also[2])."""
    from parakeet_cpp_amd import capi
    if "bf16_on_fp32_top_lp" not in score.files:
        pytest.skip("the score fixture predates the bf16-on-fp32-path keys (tools/make_golden_600m_score.py --augment)")
    cfg, wp, pcm = setup
    gm = capi.Model(wp, dataclasses.replace(cfg, gemm_bf16=True, name="tdt-600m-bf16-vs-fp32"), device=0)
    enc = gm.encode(gm.mel(pcm))
    g_lab, g_dur, o_lab, o_dur, flips, n_dec = [], [], [], [], 0, 0
    for b in range(len(pcm)):
        n = int(score["fp32_n"][b])
        lab, dur = score["fp32_labels"][b, :n], score["fp32_dur_idx"][b, :n]
        r = gm.tdt_score(enc[b], lab, dur)
        assert r["n"] == n, f"clip {b}: {r['n']} steps walked, the fp32 oracle's path has {n}"
        top = np.take_along_axis(r["label_lp"], score["fp32_top_ids"][b, :n].astype(np.int64), axis=1)
        g_lab.append(np.abs(top - score["fp32_top_lp"][b, :n])); g_dur.append(np.abs(r["dur_lp"] - score["fp32_dur_lp"][b, :n]))
        o_lab.append(np.abs(score["bf16_on_fp32_top_lp"][b, :n] - score["fp32_top_lp"][b, :n]))
        o_dur.append(np.abs(score["bf16_on_fp32_dur_lp"][b, :n] - score["fp32_dur_lp"][b, :n]))
        mg = score["fp32_margin"][b, :n]
        clear_l, clear_d = mg[:, 0] > 2 * LOGP_TOL, mg[:, 1] > 2 * LOGP_TOL
        gl, gd = r["label_lp"].argmax(axis=1), r["dur_lp"].argmax(axis=1)
        assert np.array_equal(gl[clear_l], lab[clear_l]) and np.array_equal(gd[clear_d], dur[clear_d]), f"clip {b}: a clear fp32 decision differs in the bf16 mode"
        flips += int((gl != lab).sum() + (gd != dur).sum()); n_dec += 2 * n
    gm.close()
    cat = lambda xs: np.concatenate([x.ravel() for x in xs])
    g_all, o_all = np.concatenate([cat(g_lab), cat(g_dur)]), np.concatenate([cat(o_lab), cat(o_dur)])
    print(f"bf16 GPU vs the fp32 oracle along the fp32 path: max |dlogp| {g_all.max():.3e} mean {g_all.mean():.3e} "
          f"(labels {cat(g_lab).max():.3e} / {cat(g_lab).mean():.3e}, durations {cat(g_dur).max():.3e} / {cat(g_dur).mean():.3e}); "
          f"{flips} of {n_dec} argmax decisions differ from fp32's (all within the margin bound)")
    print(f"bf16 ORACLE vs the fp32 oracle along the same path: max {o_all.max():.3e} mean {o_all.mean():.3e} "
          f"(labels {cat(o_lab).max():.3e} / {cat(o_lab).mean():.3e}, durations {cat(o_dur).max():.3e} / {cat(o_dur).mean():.3e})")
    assert g_all.mean() <= FP32_RATIO * o_all.mean(), f"mean |dlogp| vs fp32: GPU {g_all.mean():.3e} > {FP32_RATIO} x the mode's own {o_all.mean():.3e}"
    assert g_all.max() <= FP32_RATIO * o_all.max(), f"max |dlogp| vs fp32: GPU {g_all.max():.3e} > {FP32_RATIO} x the mode's own {o_all.max():.3e}"
