// parakeet.cpp_amd/csrc/pk_devmath.h -- deterministic fp32 math for the HIP kernels.
//
// DESIGN.md "Numerics contract": exp / log / tanh are fixed polynomial evaluations built from
// IEEE add / mul / fma only, so a kernel's output does not depend on the ROCm device-libs
// version and can be compared bit-for-bit with the CPU oracle, which implements the same
// written specification independently.  Coefficients come from tools/fit_math.py.
// All translation units are compiled with -ffp-contract=off: every fusion below is explicit.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

namespace pk {

// e^x.  n = rne(x*log2e) by the 1.5*2^23 trick; r = x - n*ln2 in two parts; e^r = 1 + (r + r^2 E(r));
// result scaled by 2^n in two exact steps (n/2 floor, remainder).  > 88.7228 -> +inf, < -87.3365 -> 0.
__device__ __forceinline__ float dexpf(float x) {
    // Branch-free: the range cases are SELECTS on the finished arithmetic (same values as the early returns of the specification; a lane
    // outside the range computes garbage that is discarded).  With early returns the compiler emitted three nested exec-mask branches per
    // element, which also kept it from interleaving the independent chains of an epilogue's four outputs.
    const float t = __builtin_fmaf(x, 1.44269502162933349609375f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
    float e = 0x1.6d4332p-10f;
    e = __builtin_fmaf(e, r, 0x1.120b74p-7f);
    e = __builtin_fmaf(e, r, 0x1.5554e8p-5f);
    e = __builtin_fmaf(e, r, 0x1.5554dcp-3f);
    e = __builtin_fmaf(e, r, 0.5f);
    const float q = __builtin_fmaf(r * r, e, r);
    const float p = q + 1.0f;
    const int ni = (int)__builtin_amdgcn_fmed3f(n, -160.0f, 160.0f);   // in range: n itself; outside: any finite value (a float -> int
                                                                       // conversion out of range would be undefined behaviour)
    // p 2^n with ONE rounding: v_ldexp_f32.  The specification's two exact-then-rounded multiplications (2^(n>>1), then the rest) give
    // the same float for every x of the arithmetic range (all 2 237 668 969 of them compared on the host, tools/verify_exp_nonpos.c).
    float y = __builtin_ldexpf(p, ni);
    y = x < -87.33654022216797f ? 0.0f : y;
    y = x > 88.72283935546875f ? __builtin_huge_valf() : y;
    y = x != x ? x : y;
    return y;
}

// e^x for x <= 0 (softmax / log-softmax arguments: value minus the row maximum).  The SAME value as dexpf(x), bit for bit, for every
// x <= 0 including -0, denormals and -inf (tools/verify_exp_nonpos.c checks all 2 139 095 042 of them): the overflow branch is dead, NaN
// propagates through the arithmetic, and because n >= -126 the scale 2^n is a normal number, so one multiplication rounds exactly
// like dexpf's two exact-then-rounded steps.  9 VALU operations fewer per element.
__device__ __forceinline__ float dexpf_nonpos(float x) {
    const float t = __builtin_fmaf(x, 1.44269502162933349609375f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
    float e = 0x1.6d4332p-10f;
    e = __builtin_fmaf(e, r, 0x1.120b74p-7f);
    e = __builtin_fmaf(e, r, 0x1.5554e8p-5f);
    e = __builtin_fmaf(e, r, 0x1.5554dcp-3f);
    e = __builtin_fmaf(e, r, 0.5f);
    const float q = __builtin_fmaf(r * r, e, r);
    const float p = q + 1.0f;
    const float y = __builtin_ldexpf(p, (int)__builtin_amdgcn_fmed3f(n, -160.0f, 160.0f));
    return x < -87.33654022216797f ? 0.0f : y;                      // (a select, not a branch: see dexpf)
}

// ln x.  x = m 2^e with m in [sqrt(1/2), sqrt(2)); f = m - 1; ln(1+f) = f - f^2/2 + f^3 L(f); + e ln2 in two parts.
__device__ __forceinline__ float dlogf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_huge_valf();
    if (x == __builtin_huge_valf()) return x;
    int e = 0;
    unsigned ix = __float_as_uint(x);
    if (ix < 0x00800000u) {
        x = x * 8388608.0f;
        e = -23;
        ix = __float_as_uint(x);
    }
    e += (int)(ix >> 23) - 127;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421353816986083984375f) {
        m = m * 0.5f;
        e += 1;
    }
    const float f = m - 1.0f;
    const float z = f * f;
    float l = 0x1.24df7ap-4f;
    l = __builtin_fmaf(l, f, -0x1.da0762p-4f);
    l = __builtin_fmaf(l, f, 0x1.ddaecep-4f);
    l = __builtin_fmaf(l, f, -0x1.fc5924p-4f);
    l = __builtin_fmaf(l, f, 0x1.23d638p-3f);
    l = __builtin_fmaf(l, f, -0x1.555eep-3f);
    l = __builtin_fmaf(l, f, 0x1.999d54p-3f);
    l = __builtin_fmaf(l, f, -0x1.fffff2p-3f);
    l = __builtin_fmaf(l, f, 0x1.555554p-2f);
    const float fe = (float)e;
    float y = (f * z) * l;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// tanh x.  |x| < 0.55: x + x^3 T(x^2); |x| > 9: +-1; otherwise 1 - 2/(e^{2|x|} + 1) with the sign restored.
__device__ __forceinline__ float dtanhf(float x) {
    const float ax = __builtin_fabsf(x);
    if (ax < 0.55f) {
        const float z = x * x;
        float t = -0x1.b18f62p-8f;
        t = __builtin_fmaf(t, z, 0x1.5d2fdp-6f);
        t = __builtin_fmaf(t, z, -0x1.b9a194p-5f);
        t = __builtin_fmaf(t, z, 0x1.110ffp-3f);
        t = __builtin_fmaf(t, z, -0x1.555554p-2f);
        return __builtin_fmaf(x * z, t, x);
    }
    float r;
    if (ax > 9.0f) {
        r = 1.0f;
    } else {
        const float t = dexpf(2.0f * ax);
        r = 1.0f - 2.0f / (t + 1.0f);
    }
    return __builtin_copysignf(r, x);
}

__device__ __forceinline__ float dsigmoidf(float x) { return 1.0f / (1.0f + dexpf(-x)); }
// bf16 mode only (GemmArgs::fast_act): hardware v_exp_f32 / v_rcp_f32, ~1 ulp each -- NOT part of the bit-exact contract
__device__ __forceinline__ float fast_sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269502162933349609375f)); }
__device__ __forceinline__ float fast_siluf(float x) { return x * fast_sigmoidf(x); }
__device__ __forceinline__ float dsiluf(float x) { return x / (1.0f + dexpf(-x)); }

// ---- the same SiLU / sigmoid VALUES in fewer operations, for the GEMM epilogues (fc1 applies SiLU to 16.5 M results per launch) -------------
// For an argument of ordinary size the specification's range selects are dead, 2^n can be applied as an add to the exponent field, and the
// IEEE division needs neither v_div_scale nor v_div_fixup: reciprocal estimate, one Newton step, quotient, one exact-remainder correction.
// That this 6-operation quotient is the CORRECTLY ROUNDED one (= the specification's `/`) is not a theorem for arbitrary operands; it is
// checked for every operand pair these functions can produce: pk_diag_math_exhaustive runs all 2^32 bit patterns of x through
// "mid_ok(x) => mid(x) == spec(x)" on the device (tests/test_gpu_primitives.py; profiles/r04_silu_exhaustive.txt), and the specification
// path is what the oracle parity tests pin.  Lanes outside the checked range (|x| >= 80, NaN, -0) send their whole wave down the
// specification path -- results are identical by construction, only the instruction count differs (22 instead of 34 per element).
__device__ __forceinline__ float dexpf_mid(float a) {                   // e^a for |a| < 80, bit-equal to dexpf(a)
    const float t = __builtin_fmaf(a, 1.44269502162933349609375f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693145751953125f, a);
    r = __builtin_fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
    float e = 0x1.6d4332p-10f;
    e = __builtin_fmaf(e, r, 0x1.120b74p-7f);
    e = __builtin_fmaf(e, r, 0x1.5554e8p-5f);
    e = __builtin_fmaf(e, r, 0x1.5554dcp-3f);
    e = __builtin_fmaf(e, r, 0.5f);
    const float q = __builtin_fmaf(r * r, e, r);
    const float p = q + 1.0f;
    // t = 1.5 * 2^23 + n exactly, so its low mantissa bits hold n in two's complement: p * 2^n = bits(p) + (n << 23), one v_lshl_add_u32
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t) << 23));
}
__device__ __forceinline__ bool dsilu_mid_ok(float x) { return __builtin_fabsf(x) < 80.0f && __float_as_uint(x) != 0x80000000u; }
__device__ __forceinline__ bool dsigmoid_mid_ok(float x) { return __builtin_fabsf(x) < 80.0f; }
__device__ __forceinline__ float dsiluf_mid(float x) {                  // == dsiluf(x) where dsilu_mid_ok(x)
    const float d = 1.0f + dexpf_mid(-x);
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, x), r, q);
}
__device__ __forceinline__ float dsigmoidf_mid(float x) {               // == dsigmoidf(x) where dsigmoid_mid_ok(x)
    const float d = 1.0f + dexpf_mid(-x);
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
// four results at a time with ONE wave-uniform range check (the epilogues finish 4 consecutive output columns per thread)
__device__ __forceinline__ void dsilu4(float (&v)[4]) {
    const bool ok = dsilu_mid_ok(v[0]) && dsilu_mid_ok(v[1]) && dsilu_mid_ok(v[2]) && dsilu_mid_ok(v[3]);
#ifdef PK_AB_ACT_SPEC                                             // A/B builds only (tools/experiments/act_short_ab.sh): always the long sequences
    if (false) {
#else
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dsiluf_mid(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dsiluf(v[e]);
    }
}
__device__ __forceinline__ void dsigmoid4(float (&v)[4]) {
    const bool ok = dsigmoid_mid_ok(v[0]) && dsigmoid_mid_ok(v[1]) && dsigmoid_mid_ok(v[2]) && dsigmoid_mid_ok(v[3]);
#ifdef PK_AB_ACT_SPEC
    if (false) {
#else
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dsigmoidf_mid(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dsigmoidf(v[e]);
    }
}

// 16 bytes to LDS as a ds_write2_b64 pair.  On gfx950 a 16-byte ds_write_b128 next to fragment reads is several times slower than the
// same bytes written as two 8-byte halves (fp32 GEMM: 98 vs 120-130 TF; bf16 GEMM: 320 vs 580 TF; round 2).  The compiler does not know
// the inline store: lds_store_fence() before the barrier that publishes it.
__device__ __forceinline__ void lds_store16(void *lds_ptr, const float4 &v) {
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    const unsigned addr = (unsigned)(size_t)lds_ptr;                 // low 32 bits of a flat LDS address = the LDS offset
    asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" ::"v"(addr), "v"(v2f_{v.x, v.y}), "v"(v2f_{v.z, v.w}) : "memory");
}
__device__ __forceinline__ void lds_store16_b32(void *lds_ptr, const float4 &v) {      // the same 16 bytes as four 4-byte stores
    const unsigned addr = (unsigned)(size_t)lds_ptr;
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:8\n\tds_write_b32 %0, %4 offset:12"
                 ::"v"(addr), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
}
__device__ __forceinline__ void lds_store_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Butterfly stage of the canonical 64-lane sum ("sum64"): p += p(lane ^ off), off = 32,16,...,1.
// Every lane ends with the same value (IEEE add commutes).
// The value of lane (l ^ OFF), without the LDS crossbar: __shfl_xor is a ds_bpermute_b32 (address VALU + LDS round trip, ~100 clocks of dependent latency);
// the same exchange is two quad permutes (1, 2), two bank-masked row shifts (4), a row rotation (8) -- DPP modifiers on a v_mov -- and the gfx950 row swaps
// v_permlane16_swap / v_permlane32_swap (16, 32).  The values that meet in an add / max are the same, so every reduction built on it keeps its bits
// (tools/ubench/wave_xor_probe.cpp pins the exchange against __shfl_xor and times both).
template <int OFF>
__device__ __forceinline__ int wave_xor_i(int x) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "one bit of the lane index");
    int r;
    if constexpr (OFF == 1) {
        r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);          // quad_perm:[1,0,3,2]
    } else if constexpr (OFF == 2) {
        r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);          // quad_perm:[2,3,0,1]
    } else if constexpr (OFF == 4) {
        r = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);         // row_shl:4 into banks 0 and 2 (lanes 0-3, 8-11 of a row take lane + 4)
        r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);         // row_shr:4 into banks 1 and 3 (lanes 4-7, 12-15 take lane - 4)
    } else if constexpr (OFF == 8) {
        r = __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);         // row_ror:8
    } else if constexpr (OFF == 16) {
        const auto s = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);   // [0]: rows 0, 0', 2, 2' ; [1]: rows 1, 1', 3, 3' of the input
        r = (int)((threadIdx.x & 16) ? s[0] : s[1]);
    } else {
        const auto s = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);   // [0]: low half twice ; [1]: high half twice
        r = (int)((threadIdx.x & 32) ? s[0] : s[1]);
    }
    return r;
}
template <int OFF>
__device__ __forceinline__ float wave_xor(float v) { return __int_as_float(wave_xor_i<OFF>(__float_as_int(v))); }
// `body(off_tag)` for the six offsets 32, 16, 8, 4, 2, 1 in the canonical butterfly order; decltype(off_tag)::value is the offset
template <typename F>
__device__ __forceinline__ void wave_butterfly(F &&body) {
    body(std::integral_constant<int, 32>{});
    body(std::integral_constant<int, 16>{});
    body(std::integral_constant<int, 8>{});
    body(std::integral_constant<int, 4>{});
    body(std::integral_constant<int, 2>{});
    body(std::integral_constant<int, 1>{});
}
__device__ __forceinline__ float wave_sum64(float p) {
    p = p + wave_xor<32>(p);
    p = p + wave_xor<16>(p);
    p = p + wave_xor<8>(p);
    p = p + wave_xor<4>(p);
    p = p + wave_xor<2>(p);
    p = p + wave_xor<1>(p);
    return p;
}
// the last three steps of the trees: what wave_sum64 / wave_max64 return in lanes 0 .. 7 when lanes 8 .. 63 hold the identity (+0 for a sum of non-negative terms,
// -inf for a maximum) -- the first three steps then add exact zeros / compare against -inf, so leaving them out keeps the bits
__device__ __forceinline__ float wave_sum_low8(float p) {
    p = p + wave_xor<4>(p);
    p = p + wave_xor<2>(p);
    p = p + wave_xor<1>(p);
    return p;
}
__device__ __forceinline__ float wave_max_low8(float p) {
    p = fmaxf(p, wave_xor<4>(p));
    p = fmaxf(p, wave_xor<2>(p));
    p = fmaxf(p, wave_xor<1>(p));
    return p;
}
__device__ __forceinline__ float wave_max64(float p) {
    p = fmaxf(p, wave_xor<32>(p));
    p = fmaxf(p, wave_xor<16>(p));
    p = fmaxf(p, wave_xor<8>(p));
    p = fmaxf(p, wave_xor<4>(p));
    p = fmaxf(p, wave_xor<2>(p));
    p = fmaxf(p, wave_xor<1>(p));
    return p;
}

}  // namespace pk
