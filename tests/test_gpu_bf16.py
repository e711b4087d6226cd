"""GPU parity of the bf16 mode (pk_config.gemm_bf16 = 1: bf16 operands / fp32 accumulate on v_mfma_f32_32x32x16_bf16 for the
encoder-side Linear / 1x1-conv products -- the precision BASELINE configs[2] names for tdt-600m).  The fp32 path is compared
bit for bit; this one cannot be (the MFMA does not sum a 16-wide product group as a sequential chain), so:
 * GEMM: against a float64 product of the SAME bf16-rounded operands: |err| <= 2e-6 * sum|a*b| + tiny (fp32 accumulation class);
 * model: against the oracle in its gemm_bf16 mode (operands rounded identically, k-ordered fp32 accumulation):
   a 1-ulp fp32 difference in an activation can flip its bf16 rounding (2^-8 relative) in one implementation and not the
   other, so two correct bf16 implementations agree only to bf16-epsilon class: encoder output within 2e-2 * max|x| (and a
   mean deviation below 2e-3 * max|x|, i.e. well under the bf16-vs-fp32 gap itself), token agreement (1 - edit distance / length) >= 95 %
   asserted (observed: identical on the tiny model, one differing token in 66 on the 600M cut) so that a legitimate near-tie flip does not fail the suite.
   Since round 2 the mode also stores GEMM-only activations as bf16 (same rounded operand values) and evaluates SiLU / sigmoid / the attention
   softmax on the hardware exp2 / rcp (1 ulp fp32, far below bf16 epsilon): same tolerances."""
import dataclasses

import numpy as np
import pytest

import gpu_common as G
from conftest import pk
from parakeet_cpp_amd import synth

pytestmark = pytest.mark.gpu


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("M,N,K,epi", [(64, 64, 64, "none"), (300, 1025, 512, "none"), (8064, 512, 2048, "resid"), (1000, 2048, 512, "silu"),
                                       (777, 256, 256, "relu"), (500, 512, 1024, "glu"), (8064, 4096, 1024, "none")])
def test_bf16_gemm_against_float64(M, N, K, epi):
    from parakeet_cpp_amd import capi
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    rows = 2 * N if epi == "glu" else N
    W = (rng.standard_normal((rows, K)) / np.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(rows)).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) if epi == "resid" else None
    got = capi.diag_gemm(A, W, bias, epi=epi, resid=resid, alpha=0.5, bf16=True)
    Aq, Wq = bf16_round(A).astype(np.float64), bf16_round(W).astype(np.float64)
    acc = Aq @ Wq.T + bias
    mag = np.abs(Aq) @ np.abs(Wq.T) + np.abs(bias)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    if epi == "relu":
        want = np.maximum(acc, 0)
    elif epi == "silu":
        want = acc * sig(acc)
    elif epi == "resid":
        want = resid + 0.5 * acc
    elif epi == "glu":
        want, mag = acc[:, :N] * sig(acc[:, N:]), mag[:, :N] + mag[:, N:]
    else:
        want = acc
    err = np.abs(got - want)
    assert np.all(err <= 2e-6 * mag + 1e-6), f"max err {err.max():.3e} (bound {float((2e-6 * mag + 1e-6).max()):.3e})"


@pytest.mark.parametrize("M,N,K,epi", [(2048, 512, 128, "none"), (4096, 4096, 1024, "silu"), (12032, 1024, 4096, "resid"), (12032, 3072, 1024, "none"),
                                       (2500, 1024, 1024, "glu"), (3000, 1028, 512, "relu"), (8064, 2048, 512, "silu"), (2304, 1024, 256, "resid"),
                                       # more than one round of tiles: the PERSISTENT form (one workgroup per CU walks its tiles, next tile's first K tile
                                       # requested under the epilogue) -- both tile heights, every epilogue, ragged edges, odd and even K-tile counts
                                       (12032, 4096, 1024, "silu"), (12032, 1024, 1024, "glu"), (9000, 2052, 1024, "resid"), (16384, 2048, 1088, "relu"),
                                       (12032, 3072, 1024, "resid")])
def test_bf16_glds_gemm_against_float64(M, N, K, epi):
    """The direct-to-LDS kernel (gemm_bf16_glds.hpp: activations already bf16 in HBM, XOR-swizzled LDS image, 256-row tiles): every tile
    variant the dispatch can pick (wide / narrow outputs, GLU, ragged M and N edges) against a float64 product of the same rounded operands."""
    from parakeet_cpp_amd import capi
    rng = np.random.default_rng(7 * M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    rows = 2 * N if epi == "glu" else N
    W = (rng.standard_normal((rows, K)) / np.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(rows)).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) if epi == "resid" else None
    got = capi.diag_gemm(A, W, bias, epi=epi, resid=resid, alpha=0.5, bf16=True, a16=True)
    Aq, Wq = bf16_round(A).astype(np.float64), bf16_round(W).astype(np.float64)
    acc = Aq @ Wq.T + bias
    mag = np.abs(Aq) @ np.abs(Wq.T) + np.abs(bias)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    if epi == "relu":
        want = np.maximum(acc, 0)
    elif epi == "silu":
        want = acc * sig(acc)
    elif epi == "resid":
        want = resid + 0.5 * acc
    elif epi == "glu":
        want, mag = acc[:, :N] * sig(acc[:, N:]), mag[:, :N] + mag[:, N:]
    else:
        want = acc
    err = np.abs(got - want)
    bound = 2e-6 * mag + 1e-6
    if epi == "resid":
        # (covers the EXPERIMENTAL form that accumulates the product ONTO the residual, GemmArgs::resid_init: accumulators starting from resid / alpha +
        #  bias round a sum of that size once per MFMA step -- worst case half an ulp of |resid| / alpha per step, times alpha at the end)
        bound = bound + 6e-8 * (K / 16) * np.abs(resid)
    assert np.all(err <= bound), f"max err {err.max():.3e} (bound there {float(bound.flat[err.argmax()]):.3e}) at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("M,N,K,epi,a16", [(32, 4096, 1024, "silu", True), (32, 1024, 4096, "resid", True), (32, 3072, 1024, "none", True),
                                           (32, 1024, 1024, "resid", False), (32, 1024, 1024, "glu", True), (16, 640, 1024, "none", False),
                                           (33, 512, 2048, "resid", True), (100, 1024, 4352, "none", False), (7, 2048, 512, "silu", True),
                                           (32, 1030, 256, "relu", False), (64, 512, 512, "glu", False), (2, 4096, 1024, "silu", False),
                                           (128, 1024, 4096, "resid", True), (20, 512, 256, "glu", True),
                                           (32, 2048, 1024, "silu", True), (20, 3200, 256, "none", False), (32, 1024, 1024, "glu", False)])
def test_bf16_smallm_gemm_against_float64(M, N, K, epi, a16):
    """The weight-stream kernel for a handful of rows (kernels/gemm_smallm_bf16.hip: the streaming chunks of the tolerance-class mode; M <= 128,
    K % 256 == 0): one and two row tiles per wave, several row groups, K slices of 256 / 512 k over 1..8 waves (and more slices than waves:
    K = 4352, the subsampling projection), 32 / 16 / 8 rows per workgroup, fp32 and bf16 activations, every epilogue, ragged N -- against float64
    of the same rounded operands."""
    from parakeet_cpp_amd import capi
    rng = np.random.default_rng(11 * M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    rows = 2 * N if epi == "glu" else N
    W = (rng.standard_normal((rows, K)) / np.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(rows)).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) if epi == "resid" else None
    got = capi.diag_gemm(A, W, bias, epi=epi, resid=resid, alpha=0.5, bf16=True, a16=a16)
    Aq, Wq = bf16_round(A).astype(np.float64), bf16_round(W).astype(np.float64)
    acc = Aq @ Wq.T + bias
    mag = np.abs(Aq) @ np.abs(Wq.T) + np.abs(bias)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    if epi == "relu":
        want = np.maximum(acc, 0)
    elif epi == "silu":
        want = acc * sig(acc)
    elif epi == "resid":
        want = resid + 0.5 * acc
    elif epi == "glu":
        want, mag = acc[:, :N] * sig(acc[:, N:]), mag[:, :N] + mag[:, N:]
    else:
        want = acc
    err = np.abs(got - want)
    assert np.all(err <= 2e-6 * mag + 1e-6), f"max err {err.max():.3e} (bound {float((2e-6 * mag + 1e-6).max()):.3e}) at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("M,N,K,epi", [(32, 4096, 1024, "silu"), (32, 3072, 1024, "none"), (32, 1024, 1024, "glu"), (32, 2048, 512, "silu"),
                                       (6, 1536, 512, "none"), (16, 512, 512, "glu"), (100, 640, 1024, "resid"), (16, 1030, 256, "relu"),
                                       (40, 256, 2048, "none")])
def test_bf16_smallm_gemm_with_folded_layernorm(M, N, K, epi):
    """kernels/gemm_smallm_bf16.hip with the LayerNorm of the input rows folded in (GemmArgs::ln_g: the streaming chunks of the tolerance-class
    mode): statistics from the K slices of the waves (two LDS exchanges), rows normalised and rounded in registers.  Against float64 of the
    operands the specification rounds: bf16(LayerNorm_fp32(x)) and bf16(W).  The kernel's fp32 LayerNorm differs from numpy's in its last bit
    for a few elements, and such an element may round to the neighbouring bf16 value (2^-8 relative): the bound allows a 2^-8 error on a few
    operands per row on top of the accumulation class -- err <= 2e-6 * sum|a w| + 2^-8 * 4 * max|a w|."""
    from parakeet_cpp_amd import capi
    rng = np.random.default_rng(13 * M + N + K)
    A = (rng.standard_normal((M, K)) * rng.uniform(0.5, 3.0, (M, 1)) + rng.uniform(-2, 2, (M, 1))).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, K).astype(np.float32)
    beta = (0.2 * rng.standard_normal(K)).astype(np.float32)
    rows = 2 * N if epi == "glu" else N
    W = (rng.standard_normal((rows, K)) / np.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(rows)).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) if epi == "resid" else None
    got = capi.diag_ln_gemm_bf16(A, gamma, beta, W, bias, epi=epi, resid=resid, alpha=0.5, eps=1e-5)
    # (as a streaming session runs it: weights in operand tiles, fp32 rows by LDS-DMA; the natural-layout path gives the same bits)
    capi.diag_smallm_bf16_tiles(0)
    try:
        plain = capi.diag_ln_gemm_bf16(A, gamma, beta, W, bias, epi=epi, resid=resid, alpha=0.5, eps=1e-5)
    finally:
        capi.diag_smallm_bf16_tiles(1)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32)), f"{int((got != plain).sum())} outputs differ between the weight layouts"
    A64 = A.astype(np.float64)
    mean = A64.mean(axis=1, keepdims=True)
    var = ((A64 - mean) ** 2).mean(axis=1, keepdims=True)
    ln = ((A64 - mean) / np.sqrt(var + 1e-5) * gamma + beta).astype(np.float32)
    Aq, Wq = bf16_round(ln).astype(np.float64), bf16_round(W).astype(np.float64)
    acc = Aq @ Wq.T + bias
    mag = np.abs(Aq) @ np.abs(Wq.T) + np.abs(bias)
    flip = 4.0 * 2.0 ** -8 * (np.abs(Aq).max(axis=1, keepdims=True) * np.abs(Wq).max(axis=1)[None, :])
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    if epi == "relu":
        want = np.maximum(acc, 0)
    elif epi == "silu":
        want = acc * sig(acc)
    elif epi == "resid":
        want = resid + 0.5 * acc
    elif epi == "glu":
        want, mag, flip = acc[:, :N] * sig(acc[:, N:]), mag[:, :N] + mag[:, N:], flip[:, :N] + flip[:, N:]
    else:
        want = acc
    err = np.abs(got - want)
    bound = 2e-6 * mag + flip + 1e-6
    assert np.all(err <= bound), f"max err {err.max():.3e} (bound there {float(bound.flat[err.argmax()]):.3e}) at {np.unravel_index(err.argmax(), err.shape)}"
    # and the plain accumulation class for the bulk: at most a handful of elements per row may carry a flipped operand
    loose = (err > 2e-6 * mag + 1e-6).mean()
    print(f"M {M} N {N} K {K} {epi}: max err {err.max():.2e}; elements beyond the pure accumulation bound: {100 * loose:.2f} %")


def close(got, want, what):
    d, mx = np.abs(got - want), np.abs(want).max()
    print(f"{what}: GPU bf16 vs oracle bf16: max {d.max():.2e} mean {d.mean():.2e} (max|x| {mx:.2f})")
    assert d.max() <= 2e-2 * mx and d.mean() <= 2e-3 * mx, what


def agreement(a, b):
    """1 - edit distance / length: a single flipped / inserted token must not count as a shifted tail."""
    n, m = len(a), len(b)
    prev = list(range(m + 1))
    for i in range(1, n + 1):
        cur = [i] + [0] * m
        for j in range(1, m + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return 1.0 - prev[m] / max(n, m, 1)


def test_bf16_tiny_model_vs_bf16_oracle(tmp_path_factory, orc):
    cfg = G.tiny(subsampling_channels=64, gemm_bf16=True, name="tiny-bf16")      # every GEMM K a multiple of 64
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("tb"), cfg, seed=5)
    pcm = synth.synth_pcm(3, 32000, seed=21)
    feats = gm.mel(pcm)
    enc, oenc = gm.encode(feats), om.encoder(feats)
    close(enc, oenc, "tiny")
    fp32 = orc.Model(dataclasses.replace(cfg, gemm_bf16=False), W).encoder(feats)
    gap = np.abs(oenc - fp32)
    print(f"bf16-vs-fp32 oracle gap: max {gap.max():.2e} mean {gap.mean():.2e}")
    assert gap.max() > 1e-3, "the bf16 oracle mode must actually differ from fp32"
    assert np.abs(enc - oenc).mean() < gap.mean(), "GPU bf16 should be closer to the bf16 oracle than bf16 is to fp32"
    g, o = gm.tdt_decode(enc), om.tdt_greedy(oenc)
    c, oc = gm.ctc_decode(enc), orc.ctc_greedy(om.ctc_logprobs(oenc), cfg.blank_id)
    for b in range(3):
        assert agreement(g["ids"][b, : g["lens"][b]].tolist(), o["ids"][b, : o["lens"][b]].tolist()) >= 0.95
        assert agreement(c["ids"][b, : c["lens"][b]].tolist(), oc["ids"][b, : oc["lens"][b]].tolist()) >= 0.95


def test_bf16_600m_two_layer_cut_vs_bf16_oracle(tmp_path_factory, orc):
    cfg = dataclasses.replace(pk.make_tdt_600m_config(), num_layers=2, gemm_bf16=True, name="tdt-600m-2L-bf16")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("b2b"), cfg, seed=7)
    pcm = synth.synth_pcm(1, 480000, seed=98)
    feats = gm.mel(pcm)
    enc, oenc = gm.encode(feats), om.encoder(feats)
    assert enc.shape == (1, 376, 1024)
    close(enc, oenc, "600m 2-layer cut")
    g, o = gm.tdt_decode(enc), om.tdt_greedy(oenc)
    assert agreement(g["ids"][0, : g["lens"][0]].tolist(), o["ids"][0, : o["lens"][0]].tolist()) >= 0.95
    assert o["lens"][0] > 5


def test_bf16_short_clip_runs_the_smallm_products(tmp_path_factory, orc):
    """One 10 s clip in the bf16 mode is 126 encoder rows: since round 4 its Linear products (K % 256 == 0, M <= 128) run on the small-M bf16 kernel
    (kernels/gemm_smallm_bf16.hip: bf16 activation rows in, bf16 qkv rows out for the bf16 attention) instead of the 64 x 64 tile kernel.  Same
    specification, same bounds as the 30 s test above."""
    cfg = dataclasses.replace(pk.make_tdt_600m_config(), num_layers=2, gemm_bf16=True, name="tdt-600m-2L-bf16")
    W, om, gm = G.make_pair(tmp_path_factory.mktemp("b2s"), cfg, seed=7)
    pcm = synth.synth_pcm(1, 160000, seed=99)
    feats = gm.mel(pcm)
    enc, oenc = gm.encode(feats), om.encoder(feats)
    assert enc.shape == (1, 126, 1024)
    close(enc, oenc, "600m 2-layer cut, 10 s clip")
    g, o = gm.tdt_decode(enc), om.tdt_greedy(oenc)
    assert agreement(g["ids"][0, : g["lens"][0]].tolist(), o["ids"][0, : o["lens"][0]].tolist()) >= 0.95


@pytest.mark.parametrize("S,c,d,ln,has_cache", [(16, 2, 1024, True, 1), (16, 2, 1024, True, 0), (16, 1, 1024, True, 1), (8, 4, 1024, True, 1), (5, 2, 512, True, 1),
                                                (3, 1, 256, False, 1), (7, 4, 512, False, 0), (64, 2, 1024, True, 1), (1, 2, 1024, True, 1)])
def test_bf16_glu_epilogue_with_depthwise_conv_tail_bit_identical(S, c, d, ln, has_cache):
    """The streaming conv module of the tolerance-class mode: the causal depthwise conv (kernel 9) + BatchNorm + SiLU in the GLU epilogue of pw1
    (kernels.hpp DwTail: one launch less per block) against the separate launch (stream_dwconv_kernel) on the same GLU product -- bit for bit,
    activations and the next chunk's cache; and the conv itself against float64 of the same GLU values (reference
    src/streaming_encoder.cpp:41-78: cat(cache, x), depthwise conv without padding, the last 8 rows are the new cache)."""
    from parakeet_cpp_amd import capi
    rng = np.random.default_rng(1000 * S + 10 * c + d)
    M = S * c
    A = (rng.standard_normal((M, d)) * rng.uniform(0.5, 2.0, (M, 1))).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, d).astype(np.float32) if ln else None
    beta = (0.2 * rng.standard_normal(d)).astype(np.float32) if ln else None
    W = (rng.standard_normal((2 * d, d)) / np.sqrt(d)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(2 * d)).astype(np.float32)
    cache = rng.standard_normal((S, 8, d)).astype(np.float32)
    dw_w = (rng.standard_normal((9, d)) / 3).astype(np.float32)
    dw_b, mu = (0.1 * rng.standard_normal(d)).astype(np.float32), (0.1 * rng.standard_normal(d)).astype(np.float32)
    rstd, bg, bb = rng.uniform(0.5, 2.0, d).astype(np.float32), rng.uniform(0.5, 1.5, d).astype(np.float32), (0.1 * rng.standard_normal(d)).astype(np.float32)
    args = (A, W, bias, cache, has_cache, dw_w, dw_b, mu, rstd, bg, bb, c)
    out_f, cache_f = capi.diag_glu_dwconv_bf16(*args, fused=1, gamma=gamma, beta=beta)
    out_s, cache_s = capi.diag_glu_dwconv_bf16(*args, fused=0, gamma=gamma, beta=beta)
    capi.diag_smallm_bf16_tiles(0)                                         # natural weight layout, per-lane row loads: the same bits
    try:
        out_n, cache_n = capi.diag_glu_dwconv_bf16(*args, fused=1, gamma=gamma, beta=beta)
    finally:
        capi.diag_smallm_bf16_tiles(1)
    assert np.array_equal(out_f.view(np.uint32), out_n.view(np.uint32)) and np.array_equal(cache_f.view(np.uint32), cache_n.view(np.uint32))
    assert np.array_equal(out_f.view(np.uint32), out_s.view(np.uint32)), f"{int((out_f != out_s).sum())} activations differ"
    assert np.array_equal(cache_f.view(np.uint32), cache_s.view(np.uint32)), f"{int((cache_f != cache_s).sum())} cache words differ"
    # the new cache's tail IS the chunk's GLU rows (c <= 8): recover them and restate the conv in float64
    glu = cache_f[:, 8 - c:, :]                                           # [S][c][d]
    old = cache if has_cache else np.zeros_like(cache)
    cat = np.concatenate([old, glu], axis=1).astype(np.float64)           # [S][8 + c][d]
    assert np.array_equal(cache_f, cat[:, c:, :].astype(np.float32))
    y = np.stack([(cat[:, t:t + 9, :] * dw_w[None].astype(np.float64)).sum(axis=1) for t in range(c)], axis=1) + dw_b
    y = (y - mu) * rstd * bg + bb
    want = (y / (1.0 + np.exp(-y))).reshape(M, d)
    assert np.abs(out_f - want).max() <= 1e-5 * (1.0 + np.abs(want).max())


@pytest.mark.parametrize("M,d,f", [(32, 1024, 4096), (16, 1024, 4096), (8, 512, 2048), (64, 1024, 4096), (128, 512, 1024), (24, 256, 1024)])
def test_bf16_smallm_ffn_activation_operand_tiles_bit_identical(M, d, f):
    """fc1 -> fc2 of a streaming chunk (tolerance-class mode): the bf16 activations between the two small-M products in the kernel's 8-row operand
    tiles (GemmArgs::out_t8 / a_t8: the consumer's loads read whole lines) against the same two launches with plain rows -- the same values in the
    same lanes, bit for bit; and the module against float64 of the operands the specification rounds."""
    from parakeet_cpp_amd import capi
    rng = np.random.default_rng(M + d + f)
    x = (rng.standard_normal((M, d)) * rng.uniform(0.5, 2.0, (M, 1))).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, d).astype(np.float32), (0.2 * rng.standard_normal(d)).astype(np.float32)
    W1, b1 = (rng.standard_normal((f, d)) / np.sqrt(d)).astype(np.float32), (0.1 * rng.standard_normal(f)).astype(np.float32)
    W2, b2 = (rng.standard_normal((d, f)) / np.sqrt(f)).astype(np.float32), (0.1 * rng.standard_normal(d)).astype(np.float32)
    tiled = capi.diag_ffn_bf16_smallm(x, gamma, beta, W1, b1, W2, b2, act_tiles=1)
    rows = capi.diag_ffn_bf16_smallm(x, gamma, beta, W1, b1, W2, b2, act_tiles=0)
    capi.diag_smallm_bf16_tiles(0)
    try:
        plain = capi.diag_ffn_bf16_smallm(x, gamma, beta, W1, b1, W2, b2, act_tiles=0)
    finally:
        capi.diag_smallm_bf16_tiles(1)
    assert np.array_equal(tiled.view(np.uint32), plain.view(np.uint32)), "operand-tiled weights / LDS-DMA rows vs the natural layouts"
    assert np.array_equal(tiled.view(np.uint32), rows.view(np.uint32)), f"{int((tiled != rows).sum())} outputs differ"
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=1, keepdims=True)
    ln = ((x64 - mean) / np.sqrt(((x64 - mean) ** 2).mean(axis=1, keepdims=True) + 1e-5) * gamma + beta).astype(np.float32)
    h = bf16_round(ln).astype(np.float64) @ bf16_round(W1).astype(np.float64).T + b1
    h = bf16_round((h / (1.0 + np.exp(-h))).astype(np.float32)).astype(np.float64)
    want = x64 + 0.5 * (h @ bf16_round(W2).astype(np.float64).T + b2)
    assert np.abs(tiled - want).max() <= 2e-2 * (1.0 + np.abs(want).max()), f"max err {np.abs(tiled - want).max():.3e}"


@pytest.mark.parametrize("M,N,K", [(32, 4096, 1024), (16, 4096, 1024), (32, 2048, 512), (6, 1536, 512), (128, 1024, 1024), (8, 256, 2048)])
def test_bf16_smallm_gemm_with_two_folded_layernorms(M, N, K):
    """A block's final_norm_ folded -- with the next block's first norm -- into that block's fc1 (GemmArgs::pre_g; streaming, tolerance-class mode:
    one launch less per block).  The kernel normalises the rows twice (four LDS exchanges) and its first column tile writes LN(x; pre) -- the residual
    stream from there on.  pre_out against float64 (fp32 LayerNorm class), the product against float64 of the operands the specification rounds, with
    the flipped-rounding allowance of the single-norm test; operand-tiled and natural weight layouts bit-identical."""
    from parakeet_cpp_amd import capi
    rng = np.random.default_rng(7 * M + N + K)
    A = (rng.standard_normal((M, K)) * rng.uniform(0.5, 3.0, (M, 1)) + rng.uniform(-2, 2, (M, 1))).astype(np.float32)
    pg, pb = rng.uniform(0.5, 1.5, K).astype(np.float32), (0.2 * rng.standard_normal(K)).astype(np.float32)
    g, b = rng.uniform(0.5, 1.5, K).astype(np.float32), (0.2 * rng.standard_normal(K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    got, pre = capi.diag_ln2_gemm_bf16(A, pg, pb, g, b, W, bias)
    capi.diag_smallm_bf16_tiles(0)
    try:
        got0, pre0 = capi.diag_ln2_gemm_bf16(A, pg, pb, g, b, W, bias)
    finally:
        capi.diag_smallm_bf16_tiles(1)
    assert np.array_equal(got.view(np.uint32), got0.view(np.uint32)) and np.array_equal(pre.view(np.uint32), pre0.view(np.uint32))

    def ln(x, gam, bet):
        m = x.mean(axis=1, keepdims=True)
        return (x - m) / np.sqrt(((x - m) ** 2).mean(axis=1, keepdims=True) + 1e-5) * gam + bet
    y = ln(A.astype(np.float64), pg, pb)
    assert np.abs(pre - y).max() <= 2e-6 * (1.0 + np.abs(y).max()), f"pre_out max err {np.abs(pre - y).max():.3e}"
    z = ln(pre.astype(np.float64), g, b).astype(np.float32)              # the second norm sees the fp32 rows the first one produced
    zq, Wq = bf16_round(z).astype(np.float64), bf16_round(W).astype(np.float64)
    acc = zq @ Wq.T + bias
    mag = np.abs(zq) @ np.abs(Wq.T) + np.abs(bias)
    flip = 4.0 * 2.0 ** -8 * (np.abs(zq).max(axis=1, keepdims=True) * np.abs(Wq).max(axis=1)[None, :])
    want = acc / (1.0 + np.exp(-acc))
    err = np.abs(got - want)
    bound = 2e-6 * mag + flip + 1e-6 + 2e-3 * np.abs(want)               # (+ the hardware exp2 / rcp SiLU of the mode: 1-ulp class, generous)
    assert np.all(err <= bound), f"max err {err.max():.3e} (bound there {float(bound.flat[err.argmax()]):.3e})"
