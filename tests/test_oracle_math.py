"""Accuracy of the oracle's deterministic fp32 math against float64 libm (pins the written
specification both the oracle and the HIP kernels implement), and exactness of the vectorised
GEMM micro-kernel against the scalar k-ordered fma chain."""
import numpy as np


def ulp_err(y, ref64):
    u = np.abs(np.spacing(ref64.astype(np.float32))).astype(np.float64)
    return float(np.max(np.abs(y.astype(np.float64) - ref64) / u))


def test_exp_log_tanh_sigmoid_accuracy(orc):
    rng = np.random.default_rng(0)
    x = np.linspace(-87, 88, 400001).astype(np.float32)
    assert ulp_err(orc.math_v("exp", x), np.exp(x.astype(np.float64))) < 1.5
    xl = np.exp(rng.uniform(-80, 80, 400000)).astype(np.float32)
    assert ulp_err(orc.math_v("log", xl), np.log(xl.astype(np.float64))) < 1.5
    xt = rng.uniform(-10, 10, 400000).astype(np.float32)
    assert ulp_err(orc.math_v("tanh", xt), np.tanh(xt.astype(np.float64))) < 2.5
    xs = rng.uniform(-20, 20, 400000).astype(np.float32)
    assert ulp_err(orc.math_v("sigmoid", xs), 1 / (1 + np.exp(-xs.astype(np.float64)))) < 3.5


def test_math_edge_cases(orc):
    x = np.array([0.0, -0.0, 88.8, -88.0, -104.0, np.inf, -np.inf], np.float32)
    y = orc.math_v("exp", x)
    assert y[0] == 1 and y[1] == 1 and np.isinf(y[2]) and y[3] == 0 and y[4] == 0 and np.isinf(y[5]) and y[6] == 0
    assert np.isnan(orc.math_v("exp", np.array([np.nan], np.float32))[0])
    l = orc.math_v("log", np.array([1.0, 0.0, -1.0, 1e-40, np.inf], np.float32))
    assert l[0] == 0 and np.isneginf(l[1]) and np.isnan(l[2]) and abs(l[3] - np.log(1e-40)) < 1e-4 and np.isposinf(l[4])
    t = orc.math_v("tanh", np.array([0.0, 20.0, -20.0, 1e-6], np.float32))
    assert t[0] == 0 and t[1] == 1 and t[2] == -1 and t[3] == np.float32(1e-6)


def test_sum64_is_the_documented_order(orc):
    x = np.random.default_rng(1).standard_normal(1001).astype(np.float32)
    p = np.zeros(64, np.float32)
    for i, v in enumerate(x):
        p[i & 63] = np.float32(p[i & 63] + v)
    off = 32
    while off >= 1:
        p = np.array([np.float32(p[l] + p[l ^ off]) for l in range(64)], np.float32)
        off >>= 1
    assert orc.sum64(x) == p[0]


def test_vector_gemm_equals_scalar_fma_chain(orc):
    rng = np.random.default_rng(3)
    for (M, N, K) in [(37, 53, 96), (1, 200, 64), (126, 251, 64), (7, 1025, 128)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = rng.standard_normal((N, K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        assert np.array_equal(orc.linear(A, W, b), orc.linear(A, W, b, scalar=True))
    # and the chain really is k-ordered fmaf from zero: check one element by hand in float64-exact fma
    a, w = A[0].astype(np.float64), W[0].astype(np.float64)
    acc = np.float32(0)
    for k in range(a.size):
        acc = np.float32(a[k] * w[k] + np.float64(acc))   # products of two f32 are exact in f64; one rounding
    assert orc.linear(A[:1], W[:1])[0, 0] == acc


def test_exp_nonpos_is_exp(tmp_path):
    """pk_devmath.h: dexpf_nonpos (softmax / log-softmax arguments) returns dexpf's bits for x <= 0 -- host restatement of both, strided
    sweep over all non-positive floats (tools/verify_exp_nonpos.c; stride 1 = exhaustive, ~80 s)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "verify_exp_nonpos")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "verify_exp_nonpos.c"), "-lm"])
    out = subprocess.run([exe, "997"], capture_output=True, text=True)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, out.stdout
