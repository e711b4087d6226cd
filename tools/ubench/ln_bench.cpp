// tools/ubench/ln_bench.cpp -- how far is the LayerNorm launch from what the memory system gives?  (round 6)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/ln_bench.cpp -o tools/ubench/ln_bench
// v0: the product kernel's access pattern (one wave per row, 4-byte loads at lane + 64 j: the canonical sum64 lane assignment), fp32 or bf16 rows out.
// v1: 16-byte loads (lane holds 4 consecutive elements), free reduction order (what the tolerance-class mode may use), 16-byte / 8-byte stores.
// v2: v1 with TWO rows per wave in flight.
// v3: read-only (statistics), v4: copy kernel of the same bytes (the floor of a read + write launch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ inline float wsum(float p) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);
    return p;
}
template <int PER, bool B16>
__global__ __launch_bounds__(256) void ln_v0(const float *x, long rows, int d, const float *g, const float *b, float *y) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * d;
    float v[PER], gv[PER], bv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) { v[j] = xr[lane + 64 * j]; gv[j] = g[lane + 64 * j]; bv[j] = b[lane + 64 * j]; }
    float p = 0; 
#pragma unroll
    for (int j = 0; j < PER; ++j) p += v[j];
    const float mean = wsum(p) / d;
    float q = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { const float c = v[j] - mean; q += c * c; }
    const float rstd = 1.0f / sqrtf(wsum(q) / d + 1e-5f);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const float o = fmaf((v[j] - mean) * rstd, gv[j], bv[j]);
        if (B16) reinterpret_cast<__bf16 *>(y)[row * d + lane + 64 * j] = (__bf16)o; else y[row * d + lane + 64 * j] = o;
    }
}
template <int PER4, bool B16, int RPW>
__global__ __launch_bounds__(256) void ln_v1(const float *x, long rows, int d, const float *g, const float *b, float *y) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    float4 v[RPW][PER4], gv[PER4], bv[PER4];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
        for (int j = 0; j < PER4; ++j) v[r][j] = *reinterpret_cast<const float4 *>(x + row * d + 4 * lane + 256 * j);
    }
#pragma unroll
    for (int j = 0; j < PER4; ++j) { gv[j] = *reinterpret_cast<const float4 *>(g + 4 * lane + 256 * j); bv[j] = *reinterpret_cast<const float4 *>(b + 4 * lane + 256 * j); }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r;
        if (row >= rows) break;
        float p = 0;
#pragma unroll
        for (int j = 0; j < PER4; ++j) p += (v[r][j].x + v[r][j].y) + (v[r][j].z + v[r][j].w);
        const float mean = wsum(p) / d;
        float q = 0;
#pragma unroll
        for (int j = 0; j < PER4; ++j) { const float a = v[r][j].x - mean, bb = v[r][j].y - mean, c = v[r][j].z - mean, e = v[r][j].w - mean; q += (a * a + bb * bb) + (c * c + e * e); }
        const float rstd = 1.0f / sqrtf(wsum(q) / d + 1e-5f);
#pragma unroll
        for (int j = 0; j < PER4; ++j) {
            const float o0 = fmaf((v[r][j].x - mean) * rstd, gv[j].x, bv[j].x), o1 = fmaf((v[r][j].y - mean) * rstd, gv[j].y, bv[j].y);
            const float o2 = fmaf((v[r][j].z - mean) * rstd, gv[j].z, bv[j].z), o3 = fmaf((v[r][j].w - mean) * rstd, gv[j].w, bv[j].w);
            if (B16) { const bf16x4 o = {(__bf16)o0, (__bf16)o1, (__bf16)o2, (__bf16)o3}; *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(y) + row * d + 4 * lane + 256 * j) = o; }
            else *reinterpret_cast<float4 *>(y + row * d + 4 * lane + 256 * j) = make_float4(o0, o1, o2, o3);
        }
    }
}
template <int PER4>
__global__ __launch_bounds__(256) void ln_stats(const float *x, long rows, int d, float *st) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[PER4];
#pragma unroll
    for (int j = 0; j < PER4; ++j) v[j] = *reinterpret_cast<const float4 *>(x + row * d + 4 * lane + 256 * j);
    float p = 0;
#pragma unroll
    for (int j = 0; j < PER4; ++j) p += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wsum(p) / d;
    if (lane == 0) st[row] = mean;
}
template <bool B16>
__global__ __launch_bounds__(256) void copy_k(const float4 *x, long n4, float *y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = x[i];
        if (B16) { const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w}; reinterpret_cast<bf16x4 *>(y)[i] = o; }
        else reinterpret_cast<float4 *>(y)[i] = v;
    }
}
int main() {
    const long R = 12032, D = 1024;
    float *x[8], *y[8], *g, *b, *st;
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&x[i], R * D * 4)); CK(hipMalloc(&y[i], R * D * 4)); CK(hipMemset(x[i], 0, R * D * 4)); }
    CK(hipMalloc(&g, D * 4)); CK(hipMalloc(&b, D * 4)); CK(hipMalloc(&st, R * 8)); CK(hipMemset(g, 0, D * 4)); CK(hipMemset(b, 0, D * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, long rows, int d, double bytes, auto launch) {
        for (int i = 0; i < 8; ++i) launch(i);
        CK(hipEventRecord(e0, 0));
        const int reps = 64;
        for (int i = 0; i < reps; ++i) launch(i % 8);                 // (rotating through 8 x 49 MB buffers: nothing stays in the memory-side cache)
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s rows %5ld d %4d : %7.2f us per launch  %5.2f TB/s\n", name, rows, d, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) * 1e-12);
    };
    for (int cfg = 0; cfg < 2; ++cfg) {
        const long rows = cfg ? 8064 : 12032; const int d = cfg ? 512 : 1024;
        const unsigned gr = (unsigned)((rows + 3) / 4);
        const double rw32 = (double)rows * d * 8, rw16 = (double)rows * d * 6, ro = (double)rows * d * 4;
        if (d == 1024) {
            time("v0 4-byte loads, fp32 out", rows, d, rw32, [&](int i) { hipLaunchKernelGGL((ln_v0<16, false>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v0 4-byte loads, bf16 out", rows, d, rw16, [&](int i) { hipLaunchKernelGGL((ln_v0<16, true>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v1 16-byte loads, fp32 out", rows, d, rw32, [&](int i) { hipLaunchKernelGGL((ln_v1<4, false, 1>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v1 16-byte loads, bf16 out", rows, d, rw16, [&](int i) { hipLaunchKernelGGL((ln_v1<4, true, 1>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v2 16-byte loads, 2 rows per wave, bf16 out", rows, d, rw16, [&](int i) { hipLaunchKernelGGL((ln_v1<4, true, 2>), dim3((gr + 1) / 2), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v3 statistics only (read)", rows, d, ro, [&](int i) { hipLaunchKernelGGL((ln_stats<4>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, st); });
        } else {
            time("v0 4-byte loads, fp32 out", rows, d, rw32, [&](int i) { hipLaunchKernelGGL((ln_v0<8, false>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v1 16-byte loads, fp32 out", rows, d, rw32, [&](int i) { hipLaunchKernelGGL((ln_v1<2, false, 1>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v2 16-byte loads, 2 rows per wave, fp32 out", rows, d, rw32, [&](int i) { hipLaunchKernelGGL((ln_v1<2, false, 2>), dim3((gr + 1) / 2), dim3(256), 0, 0, x[i], rows, d, g, b, y[i]); });
            time("v3 statistics only (read)", rows, d, ro, [&](int i) { hipLaunchKernelGGL((ln_stats<2>), dim3(gr), dim3(256), 0, 0, x[i], rows, d, st); });
        }
        time("copy, 2048 blocks, fp32 out", rows, d, rw32, [&](int i) { hipLaunchKernelGGL((copy_k<false>), dim3(2048), dim3(256), 0, 0, (const float4 *)x[i], rows * d / 4, y[i]); });
        time("copy, 2048 blocks, bf16 out", rows, d, rw16, [&](int i) { hipLaunchKernelGGL((copy_k<true>), dim3(2048), dim3(256), 0, 0, (const float4 *)x[i], rows * d / 4, y[i]); });
    }
    return 0;
}
