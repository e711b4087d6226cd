/*
 * oracle/pk_oracle_math.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Deterministic fp32 transcendental functions + the canonical 64-way summation
 * used by the CPU oracle.  The reference (Frikallo/parakeet.cpp) delegates
 * exp/log/tanh/sigmoid and every reduction to the un-vendored `axiom` library,
 * whose implementations and summation orders are not recoverable
 * (SURVEY.md section 8c).  The oracle therefore FIXES one documented evaluation
 * for each (DESIGN.md "Numerics contract"); the HIP kernels implement the same
 * written specification independently (parakeet.cpp_amd/csrc/pk_devmath.h), so
 * stage outputs can be compared bit-for-bit.  Accuracy of these functions against
 * libm is pinned by tests/test_oracle_math.py (<= 2 ulp exp/log, <= 3 ulp tanh).
 *
 * Only IEEE-754 binary32 add/mul/fma/div/sqrt and integer ops are used, and the
 * file must be compiled with -ffp-contract=off so that no implicit contraction
 * happens: every fused operation is an explicit fmaf().
 *
 * Coefficients: tools/fit_math.py.
 */
#ifndef PK_ORACLE_MATH_H
#define PK_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float orc_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t orc_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp(x): n = rne(x*log2e); r = x - n*ln2 (two-part); e^r = 1 + (r + r^2*E(r)); scale by 2^n in two steps. */
static inline float orc_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -87.33654022216797f) return 0.0f;
    const float t = fmaf(x, 1.44269502162933349609375f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
    float e = 0x1.6d4332p-10f;
    e = fmaf(e, r, 0x1.120b74p-7f);
    e = fmaf(e, r, 0x1.5554e8p-5f);
    e = fmaf(e, r, 0x1.5554dcp-3f);
    e = fmaf(e, r, 0.5f);
    const float q = fmaf(r * r, e, r);
    const float p = q + 1.0f;
    const int ni = (int)n;
    const int n1 = ni >> 1;
    const int n2 = ni - n1;
    const float s1 = orc_u2f((uint32_t)(n1 + 127) << 23);
    const float s2 = orc_u2f((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}

/* log(x): x = m*2^e, m in [sqrt(.5), sqrt(2)); f = m-1; log(1+f) = f - f^2/2 + f^3*L(f); + e*ln2 (two-part). */
static inline float orc_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t ix = orc_f2u(x);
    if (ix < 0x00800000u) { x = x * 8388608.0f; e = -23; ix = orc_f2u(x); }
    e += (int)(ix >> 23) - 127;
    float m = orc_u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421353816986083984375f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float l = 0x1.24df7ap-4f;
    l = fmaf(l, f, -0x1.da0762p-4f);
    l = fmaf(l, f, 0x1.ddaecep-4f);
    l = fmaf(l, f, -0x1.fc5924p-4f);
    l = fmaf(l, f, 0x1.23d638p-3f);
    l = fmaf(l, f, -0x1.555eep-3f);
    l = fmaf(l, f, 0x1.999d54p-3f);
    l = fmaf(l, f, -0x1.fffff2p-3f);
    l = fmaf(l, f, 0x1.555554p-2f);
    const float fe = (float)e;
    float y = (f * z) * l;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* tanh(x): |x|<0.55: x + x^3*T(x^2); |x|>9: +-1; else 1 - 2/(exp(2|x|)+1). */
static inline float orc_tanhf(float x) {
    const float ax = fabsf(x);
    if (ax < 0.55f) {
        const float z = x * x;
        float t = -0x1.b18f62p-8f;
        t = fmaf(t, z, 0x1.5d2fdp-6f);
        t = fmaf(t, z, -0x1.b9a194p-5f);
        t = fmaf(t, z, 0x1.110ffp-3f);
        t = fmaf(t, z, -0x1.555554p-2f);
        return fmaf(x * z, t, x);
    }
    float r;
    if (ax > 9.0f) {
        r = 1.0f;
    } else {
        const float t = orc_expf(2.0f * ax);
        r = 1.0f - 2.0f / (t + 1.0f);
    }
    return copysignf(r, x);
}

static inline float orc_sigmoidf(float x) { return 1.0f / (1.0f + orc_expf(-x)); }
static inline float orc_siluf(float x) { return x / (1.0f + orc_expf(-x)); }

/*
 * Canonical 64-way sum ("sum64"): partial[l] = x[l] + x[l+64] + ... (increasing
 * index, left to right), then a butterfly p[l] = p[l] + p[l^off] for
 * off = 32,16,8,4,2,1.  IEEE addition is commutative, so every slot holds the
 * same value after each stage; slot 0 is returned.  This is exactly what one
 * 64-lane wavefront computes with a strided accumulate + xor-shuffle reduce.
 */
static inline float orc_sum64(const float *x, int64_t n, int64_t stride) {
    float p[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0f;
    for (int64_t i = 0; i < n; ++i) p[i & 63] = p[i & 63] + x[i * stride];
    for (int off = 32; off >= 1; off >>= 1) {
        float q[64];
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        for (int l = 0; l < 64; ++l) p[l] = q[l];
    }
    return p[0];
}

#endif /* PK_ORACLE_MATH_H */
