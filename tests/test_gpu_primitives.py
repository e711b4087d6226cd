"""GPU parity, layer 0: the numerical building blocks behind the C ABI, bit-for-bit against the oracle.
These pin the assumptions the whole numerics contract rests on: the device polynomial math equals the
oracle's, IEEE sqrt/div are correctly rounded on gfx950, the wavefront butterfly equals sum64, and the
fp32 MFMA GEMM is a natural-k fma chain."""
import numpy as np
import pytest

from conftest import pk  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from parakeet_cpp_amd import capi
    assert capi.device_count() >= 1, "no HIP device: the product has no CPU path"
    return capi


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("fn,lo,hi", [("exp", -100, 90), ("tanh", -12, 12), ("sigmoid", -30, 30), ("silu", -30, 30)])
def test_device_math_is_bit_identical(capi, orc, fn, lo, hi):
    x = np.random.default_rng(0).uniform(lo, hi, 1 << 20).astype(np.float32)
    x[:8] = [0.0, -0.0, 1e-30, -1e-30, 88.72, -87.33, 0.55, -0.55]
    assert np.array_equal(bits(capi.diag_math(fn, x)), bits(orc.math_v(fn, x)))


@pytest.mark.parametrize("fn", ["sigmoid", "silu"])
def test_epilogue_activation_forms_are_the_specification(capi, orc, fn):
    """The GEMM epilogues evaluate SiLU / sigmoid with a shorter instruction sequence where the argument is of ordinary size
    (pk_devmath.h: dsilu4 / dsigmoid4).  (a) sampled against the ORACLE: ordinary arguments, the guard's edges, and uniformly random bit
    patterns (waves that mix in-range and out-of-range lanes); (b) all 2^32 bit patterns on the device: wherever the short sequence claims
    validity it equals the specification path, and the guarded four-at-a-time form equals it everywhere."""
    rng = np.random.default_rng(5)
    x = rng.uniform(-30, 30, 1 << 20).astype(np.float32)
    assert np.array_equal(bits(capi.diag_math(fn + "4", x)), bits(orc.math_v(fn, x)))
    e = np.array([0.0, -0.0, 1e-30, -1e-30, 1e-40, -1e-40, 79.99999, -79.99999, 80.0, -80.0, 87.4, -87.4, 88.8, -88.8, 1e30, -1e30,
                  np.inf, -np.inf, 1.1754944e-38, -1.1754944e-38], np.float32)
    x = np.concatenate([np.repeat(e, 64), rng.uniform(-90, 90, 1 << 16).astype(np.float32)])
    assert np.array_equal(bits(capi.diag_math(fn + "4", x)), bits(orc.math_v(fn, x)))
    u = rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32).view(np.float32)
    u = u[~np.isnan(u)]
    assert np.array_equal(bits(capi.diag_math(fn + "4", u)), bits(orc.math_v(fn, u)))
    checked, mism, first = capi.diag_math_exhaustive(fn)
    assert mism == 0, f"short {fn} sequence differs from the specification at bit pattern {first:#010x}"
    assert checked == 2 * 0x42A00000 - (1 if fn == "silu" else 0)          # every |x| < 80 (SiLU: except -0)
    checked, mism, first = capi.diag_math_exhaustive(fn, guarded=True)
    assert (checked, mism) == (1 << 32, 0), f"guarded {fn} differs at {first:#010x}"


def test_device_log_sqrt_rcp_bit_identical(capi, orc):
    rng = np.random.default_rng(1)
    x = np.exp(rng.uniform(-87, 87, 1 << 20)).astype(np.float32)
    x[:6] = [1.0, 2.0, 5.96046448e-8, 1e-40, 1.4142135, 0.70710677]
    for fn in ("log", "sqrt", "rcp"):
        assert np.array_equal(bits(capi.diag_math(fn, x)), bits(orc.math_v(fn, x))), fn


def test_wave_butterfly_is_sum64(capi, orc):
    rng = np.random.default_rng(2)
    for n in (1, 5, 64, 126, 512, 1001, 1025):
        x = rng.standard_normal((7, n)).astype(np.float32)
        got = capi.diag_sum64(x)
        want = np.array([orc.sum64(r) for r in x], np.float32)
        assert np.array_equal(bits(got), bits(want)), n


@pytest.mark.parametrize("d", [128, 512, 1024])
def test_layernorm_bit_identical(capi, orc, d):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((130, d)) * 3 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    assert np.array_equal(bits(capi.diag_layernorm(x, g, b)), bits(orc.layer_norm(x, g, b)))


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (126, 512, 512), (300, 1025, 512), (252, 640, 2048), (1000, 256, 256),
                                   (8000, 1536, 512), (8064, 2048, 512), (8064, 512, 2048)])   # the big-tile kernels at the headline's own shapes
def test_mfma_gemm_is_natural_k_fma_chain(capi, orc, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    assert np.array_equal(bits(capi.diag_gemm(A, W, b)), bits(orc.linear(A, W, b)))
    assert np.array_equal(bits(capi.diag_gemm(A, W, None)), bits(orc.linear(A, W, None)))


@pytest.mark.parametrize("M,N,K,epi", [(8064, 2048, 512, "silu"), (8064, 1536, 512, "none"), (8064, 512, 512, "glu"), (1600, 1024, 512, "relu"),
                                       (12032, 4096, 1024, "silu"), (2001, 1100, 96, "none")])
def test_layernorm_from_row_statistics_in_the_tile_gemm_bit_identical(capi, orc, M, N, K, epi):
    """Round 6: on large fp32 batches the LayerNorm in front of a wide product is a statistics pass and the tile kernel normalises while it stages A
    (gemm_pipe.hpp: LNA).  The headline's own shapes (fc1, qkv, pw1 + GLU of tdt-ctc-110m at 64 x 10 s; fc1 of tdt-600m; ragged sizes): the product
    equals the ORACLE's layer_norm + linear bit for bit, and the un-folded pair of launches."""
    rng = np.random.default_rng(M + N + K)
    X = (rng.standard_normal((M, K)) * 2 + 0.3).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    be = (0.1 * rng.standard_normal(K)).astype(np.float32)
    rows = N * 2 if epi == "glu" else N
    W = (rng.standard_normal((rows, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(rows).astype(np.float32)
    y = orc.linear(orc.layer_norm(X, g, be), W, b)
    if epi == "silu":
        want = orc.math_v("silu", y)
    elif epi == "relu":
        want = np.where(y > 0, y, np.float32(0))
    elif epi == "glu":
        want = y[:, :N] * orc.math_v("sigmoid", y[:, N:])
    else:
        want = y
    folded, _ = capi.diag_ln_gemm(X, g, be, W, b, epi, fold=True)
    plain, _ = capi.diag_ln_gemm(X, g, be, W, b, epi, fold=False)
    assert np.array_equal(bits(folded), bits(want)), "statistics pass + normalise-on-stage differs from the oracle"
    assert np.array_equal(bits(plain), bits(want))


def test_final_norm_then_statistics_of_the_next_norm_bit_identical(capi, orc):
    """A block's final_norm_ written out with the statistics of ITS rows for the next block's ffn1 norm (launch_layernorm_then_stats), that norm applied
    by fc1's tile kernel: the residual stream and the product equal layer_norm(layer_norm(x)) -> linear of the oracle bit for bit."""
    rng = np.random.default_rng(77)
    M, N, K = 8064, 2048, 512
    A = (rng.standard_normal((M, K)) * 3 - 0.2).astype(np.float32)
    pg, pb = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32), (0.1 * rng.standard_normal(K)).astype(np.float32)
    g, be = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32), (0.1 * rng.standard_normal(K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    x1 = orc.layer_norm(A, pg, pb)
    want = orc.math_v("silu", orc.linear(orc.layer_norm(x1, g, be), W, b))
    for fold in (True, False):
        out, y1 = capi.diag_ln_gemm(A, g, be, W, b, "silu", fold=fold, pre_gamma=pg, pre_beta=pb)
        assert np.array_equal(bits(y1), bits(x1)), f"fold={fold}: residual stream after final_norm_"
        assert np.array_equal(bits(out), bits(want)), f"fold={fold}: fc1 on the doubly normalised rows"


def test_gemm_asymmetric_identity_detects_transposes(capi):
    # A = I, asymmetric W: out must be exactly W^T (catches row/col swaps in the MFMA C layout)
    K = 64
    A = np.eye(K, dtype=np.float32)
    W = (np.arange(96 * K, dtype=np.float32).reshape(96, K) * 0.25)
    assert np.array_equal(capi.diag_gemm(A, W), W.T)


def test_gemm_epilogues_bit_identical(capi, orc):
    rng = np.random.default_rng(7)
    M, N, K = 260, 512, 256
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y = orc.linear(A, W, b)
    assert np.array_equal(bits(capi.diag_gemm(A, W, b, "relu")), bits(np.where(y > 0, y, np.float32(0))))
    assert np.array_equal(bits(capi.diag_gemm(A, W, b, "silu")), bits(orc.math_v("silu", y)))
    r = rng.standard_normal((M, N)).astype(np.float32)
    assert np.array_equal(bits(capi.diag_gemm(A, W, b, "resid", r, 0.5)), bits(r + y * np.float32(0.5)))
    assert np.array_equal(bits(capi.diag_gemm(A, W, b, "resid", r, 1.0)), bits(r + y))
    # GLU: W has 2N' rows (value rows then gate rows)
    Np = N // 2
    want = y[:, :Np] * orc.math_v("sigmoid", y[:, Np:])
    assert np.array_equal(bits(capi.diag_gemm(A, W, b, "glu")), bits(want))
