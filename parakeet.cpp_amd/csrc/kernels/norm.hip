// parakeet.cpp_amd/csrc/kernels/norm.hip -- LayerNorm and the small diagnostic kernels.
//
// LayerNorm (axiom nn::LayerNorm, eps 1e-5; reference call sites src/encoder.cpp:40,60,182,202):
// one wavefront per row, the row lives in registers, mean and variance are canonical sum64
// reductions (strided per-lane accumulate + xor butterfly), y = fma((x-mean)*rstd, gamma, beta).
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

// RPW rows per wavefront.  Round 4 measured 1 / 2 / 4 on the batch shapes (tools/experiments/ln_rpw_ab.sh, profiles/r04_ln_rpw_ab.txt): sharing one
// load of gamma / beta between the rows of a wave does NOT pay -- tdt-ctc-110m 0.87 / 0.92 / 1.02 ms of LayerNorm per step, tdt-600m bf16 2.23 /
// -- / 2.39: the kernel wants the most independent waves it can get (it runs at ~4 TB/s of combined read + write).  One row per wave it stays.
#ifndef PK_LN_RPW
#define PK_LN_RPW 1
#endif
template <int PER_LANE, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, int64_t rows, int d,
                                                        const float *__restrict__ g, const float *__restrict__ b,
                                                        float eps, float *__restrict__ y, int y_bf16) {
    __builtin_amdgcn_s_setprio(3);                                  // (see gemm_pipe.hpp)
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    float v[RPW][PER_LANE], gv[PER_LANE], bv[PER_LANE];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;  // (rows past the end re-read the last one; never stored)
        const float *xr = x + row * d;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int i = lane + 64 * j;
            v[r][j] = i < d ? xr[i] : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {                            // gamma / beta in the same round trip (not one after the reductions)
        const int i = lane + 64 * j;
        gv[j] = i < d ? g[i] : 0.0f;
        bv[j] = i < d ? b[i] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
        float p = 0.0f;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j)
            if (lane + 64 * j < d) p = p + v[r][j];
        const float mean = wave_sum64(p) / (float)d;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int i = lane + 64 * j;
            if (i < d) {
                const float c = v[r][j] - mean;
                q = q + c * c;
            }
        }
        const float var = wave_sum64(q) / (float)d;
        const float rstd = 1.0f / __builtin_sqrtf(var + eps);
        float *yr = y + row * d;
        __bf16 *yh = reinterpret_cast<__bf16 *>(y) + row * d;      // bf16 mode: the consuming GEMM's operand, rounded here instead of there
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int i = lane + 64 * j;
            if (i < d) {
                const float o = __builtin_fmaf((v[r][j] - mean) * rstd, gv[j], bv[j]);
                if (y_bf16 == 1) yh[i] = (__bf16)o;
                else if (y_bf16 == 2) yr[(i & ~15) | ((i & 3) << 2) | ((i >> 2) & 3)] = o;   // sigma K layout (kernels.hpp: GemmArgs::a_sigma)
                else yr[i] = o;
            }
        }
    }
}

// Two LayerNorms back to back on the same row: y1 = LN(x; g1, b1) (a block's final_norm_, src/encoder.cpp:202) and
// y2 = LN(y1; g2, b2) (the next block's ffn1_ norm, :40) -- the row stays in registers, one read of x instead of two, one launch
// instead of two.  Every value goes through exactly the operations of two layernorm_kernel passes (same sum64 butterflies).
template <int PER_LANE>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float *__restrict__ x, int64_t rows, int d, const float *__restrict__ g1,
                                                         const float *__restrict__ b1, const float *__restrict__ g2,
                                                         const float *__restrict__ b2, float eps, float *__restrict__ y1,
                                                         float *__restrict__ y2, int y2_bf16) {
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * d;
    float v[PER_LANE], gv[2][PER_LANE], bv[2][PER_LANE];
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
        const int i = lane + 64 * j;
        v[j] = i < d ? xr[i] : 0.0f;
        gv[0][j] = i < d ? g1[i] : 0.0f; bv[0][j] = i < d ? b1[i] : 0.0f;
        gv[1][j] = i < d ? g2[i] : 0.0f; bv[1][j] = i < d ? b2[i] : 0.0f;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float *yr = (pass ? y2 : y1) + row * d;
        float p = 0.0f;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j)
            if (lane + 64 * j < d) p = p + v[j];
        const float mean = wave_sum64(p) / (float)d;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j)
            if (lane + 64 * j < d) {
                const float c = v[j] - mean;
                q = q + c * c;
            }
        const float var = wave_sum64(q) / (float)d;
        const float rstd = 1.0f / __builtin_sqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int i = lane + 64 * j;
            if (i < d) {
                v[j] = __builtin_fmaf((v[j] - mean) * rstd, gv[pass][j], bv[pass][j]);
                if (pass && y2_bf16 == 1) (reinterpret_cast<__bf16 *>(y2) + row * d)[i] = (__bf16)v[j];
                else if (pass && y2_bf16 == 2) yr[(i & ~15) | ((i & 3) << 2) | ((i >> 2) & 3)] = v[j];
                else yr[i] = v[j];
            }
        }
    }
}
// Statistics of the rows of x -- or, THEN = true, y1 = LN(x; g1, b1) written out and the statistics of y1's rows: the first half of layernorm_kernel /
// the first one and a half passes of layernorm2_kernel, operation for operation (the same strided partial sums, the same wave_sum64 butterflies, the
// same divisions and the same 1 / sqrt) -- stats[row] = {mean, rstd}.  The consumer (gemm_pipe.hpp, LNA) applies fma((x - mean) * rstd, gamma, beta).
template <int PER_LANE, bool THEN>
__global__ __launch_bounds__(256) void layernorm_stats_kernel(const float *__restrict__ x, int64_t rows, int d, const float *__restrict__ g1,
                                                              const float *__restrict__ b1, float eps, float *__restrict__ y1, float *__restrict__ stats) {
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * d;
    float v[PER_LANE];
    [[maybe_unused]] float gv[THEN ? PER_LANE : 1], bv[THEN ? PER_LANE : 1];
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
        const int i = lane + 64 * j;
        v[j] = i < d ? xr[i] : 0.0f;
        if constexpr (THEN) { gv[j] = i < d ? g1[i] : 0.0f; bv[j] = i < d ? b1[i] : 0.0f; }
    }
#pragma unroll
    for (int pass = 0; pass < (THEN ? 2 : 1); ++pass) {
        float p = 0.0f;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j)
            if (lane + 64 * j < d) p = p + v[j];
        const float mean = wave_sum64(p) / (float)d;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j)
            if (lane + 64 * j < d) {
                const float c = v[j] - mean;
                q = q + c * c;
            }
        const float var = wave_sum64(q) / (float)d;
        const float rstd = 1.0f / __builtin_sqrtf(var + eps);
        if (pass == (THEN ? 1 : 0)) {
            if (lane == 0) *reinterpret_cast<float2 *>(stats + 2 * row) = make_float2(mean, rstd);
        } else {
            float *yr = y1 + row * d;
#pragma unroll
            for (int j = 0; j < PER_LANE; ++j) {
                const int i = lane + 64 * j;
                if (i < d) {
                    v[j] = __builtin_fmaf((v[j] - mean) * rstd, gv[j], bv[j]);
                    yr[i] = v[j];
                }
            }
        }
    }
}
void launch_layernorm_stats(const float *x, int64_t rows, int d, float eps, float *stats, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (d <= 128) hipLaunchKernelGGL((layernorm_stats_kernel<2, false>), grid, dim3(256), 0, s, x, rows, d, nullptr, nullptr, eps, nullptr, stats);
    else if (d <= 512) hipLaunchKernelGGL((layernorm_stats_kernel<8, false>), grid, dim3(256), 0, s, x, rows, d, nullptr, nullptr, eps, nullptr, stats);
    else hipLaunchKernelGGL((layernorm_stats_kernel<16, false>), grid, dim3(256), 0, s, x, rows, d, nullptr, nullptr, eps, nullptr, stats);
}
void launch_layernorm_then_stats(const float *x, int64_t rows, int d, const float *g1, const float *b1, float eps, float *y1, float *stats, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (d <= 128) hipLaunchKernelGGL((layernorm_stats_kernel<2, true>), grid, dim3(256), 0, s, x, rows, d, g1, b1, eps, y1, stats);
    else if (d <= 512) hipLaunchKernelGGL((layernorm_stats_kernel<8, true>), grid, dim3(256), 0, s, x, rows, d, g1, b1, eps, y1, stats);
    else hipLaunchKernelGGL((layernorm_stats_kernel<16, true>), grid, dim3(256), 0, s, x, rows, d, g1, b1, eps, y1, stats);
}

void launch_layernorm2(const float *x, int64_t rows, int d, const float *g1, const float *b1, const float *g2, const float *b2, float eps,
                       float *y1, float *y2, hipStream_t s, int y2_bf16) {
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (d <= 128) hipLaunchKernelGGL(layernorm2_kernel<2>, grid, dim3(256), 0, s, x, rows, d, g1, b1, g2, b2, eps, y1, y2, y2_bf16);
    else if (d <= 512) hipLaunchKernelGGL(layernorm2_kernel<8>, grid, dim3(256), 0, s, x, rows, d, g1, b1, g2, b2, eps, y1, y2, y2_bf16);
    else hipLaunchKernelGGL(layernorm2_kernel<16>, grid, dim3(256), 0, s, x, rows, d, g1, b1, g2, b2, eps, y1, y2, y2_bf16);
}

void launch_layernorm(const float *x, int64_t rows, int d, const float *g, const float *b, float eps, float *y, hipStream_t s, int y_bf16) {
    // few rows (streaming chunks, single clips): one row per wave -- the launch is latency-bound and wants every wave it can get;
    // batches: PK_LN_RPW rows per wave share one load of gamma / beta
    if (rows < 4096) {
        const dim3 grid((unsigned)((rows + 3) / 4));
        if (d <= 128) hipLaunchKernelGGL((layernorm_kernel<2, 1>), grid, dim3(256), 0, s, x, rows, d, g, b, eps, y, y_bf16);
        else if (d <= 512) hipLaunchKernelGGL((layernorm_kernel<8, 1>), grid, dim3(256), 0, s, x, rows, d, g, b, eps, y, y_bf16);
        else hipLaunchKernelGGL((layernorm_kernel<16, 1>), grid, dim3(256), 0, s, x, rows, d, g, b, eps, y, y_bf16);
        return;
    }
    constexpr int RPW = PK_LN_RPW;
    const dim3 grid((unsigned)((rows + 4 * RPW - 1) / (4 * RPW)));
    if (d <= 128) hipLaunchKernelGGL((layernorm_kernel<2, RPW>), grid, dim3(256), 0, s, x, rows, d, g, b, eps, y, y_bf16);
    else if (d <= 512) hipLaunchKernelGGL((layernorm_kernel<8, RPW>), grid, dim3(256), 0, s, x, rows, d, g, b, eps, y, y_bf16);
    else hipLaunchKernelGGL((layernorm_kernel<16, RPW>), grid, dim3(256), 0, s, x, rows, d, g, b, eps, y, y_bf16);
}

__global__ __launch_bounds__(64) void sum64_rows_kernel(const float *__restrict__ x, int n, float *__restrict__ out) {
    const float *row = x + (int64_t)blockIdx.x * n;
    float p = 0.0f;
    for (int i = threadIdx.x; i < n; i += 64) p = p + row[i];
    p = wave_sum64(p);
    if (threadIdx.x == 0) out[blockIdx.x] = p;
}
void launch_sum64_rows(const float *x, int rows, int n, float *out, hipStream_t s) {
    hipLaunchKernelGGL(sum64_rows_kernel, dim3(rows), dim3(64), 0, s, x, n, out);
}

__global__ void math_kernel(int fn, const float *__restrict__ in, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        float y;
        switch (fn) {
        case 0: y = dexpf(x); break;
        case 1: y = dlogf(x); break;
        case 2: y = dtanhf(x); break;
        case 3: y = dsigmoidf(x); break;
        case 4: y = dsiluf(x); break;
        case 5: y = __builtin_sqrtf(x); break;
        case 6: y = 1.0f / x; break;
        case 7: y = x > 0.0f ? x : 0.0f; break;
        case 8: { float v[4] = {x, x, x, x}; dsigmoid4(v); y = v[0]; break; }   // the GEMM epilogues' guarded four-at-a-time forms
        case 9: { float v[4] = {x, x, x, x}; dsilu4(v); y = v[0]; break; }
        default: y = x;
        }
        out[i] = y;
    }
}
void launch_math(int fn, const float *in, float *out, int64_t n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(math_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, fn, in, out, n);
}

// All 2^32 bit patterns of x.  fn 3 / 4: wherever the short sigmoid / SiLU sequences of pk_devmath.h claim validity (d*_mid_ok), their value
// must equal the specification's bit for bit.  fn 13 / 14: the guarded four-at-a-time forms the GEMM epilogues call, on EVERY pattern
// (x with three neighbours: -x, x with the exponent's low bit flipped, the next pattern), against the specification.
// out[0] = patterns checked, out[1] = mismatches, out[2] = lowest mismatching pattern (or 2^32).
__global__ void math_exhaustive_kernel(int fn, unsigned long long *out) {
    unsigned long long checked = 0, bad = 0, first = 1ull << 32;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const unsigned u = (unsigned)i;
        const float x = __uint_as_float(u);
        bool mism = false;
        if (fn == 3 || fn == 4) {
            const bool ok = fn == 3 ? dsigmoid_mid_ok(x) : dsilu_mid_ok(x);
            if (!ok) continue;
            const float a = fn == 3 ? dsigmoidf_mid(x) : dsiluf_mid(x), b = fn == 3 ? dsigmoidf(x) : dsiluf(x);
            mism = __float_as_uint(a) != __float_as_uint(b);
        } else {
            float v[4] = {x, -x, __uint_as_float(u ^ 0x00800000u), __uint_as_float(u + 1u)}, w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = fn == 13 ? dsigmoidf(v[e]) : dsiluf(v[e]);
            if (fn == 13) dsigmoid4(v); else dsilu4(v);
#pragma unroll
            for (int e = 0; e < 4; ++e) mism = mism || (__float_as_uint(v[e]) != __float_as_uint(w[e]) && !(v[e] != v[e] && w[e] != w[e]));
        }
        ++checked;
        if (mism) { ++bad; first = i < first ? i : first; }
    }
    atomicAdd(out, checked);
    atomicAdd(out + 1, bad);
    atomicMin(out + 2, first);
}
void launch_math_exhaustive(int fn, unsigned long long *out3, hipStream_t s) {
    hipLaunchKernelGGL(math_exhaustive_kernel, dim3(8192), dim3(256), 0, s, fn, out3);
}

__global__ void scale_kernel(float *__restrict__ x, int64_t n, float a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = x[i] * a;
}
void launch_scale(float *x, int64_t n, float a, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, x, n, a);
}

}  // namespace pk
