// tools/ubench/lds_bw.cpp -- LDS store / load throughput per CU on gfx950, with the fp32 GEMM's staging address patterns.
// Every wave issues ITER x 8 LDS instructions of one kind back to back (inline asm, one s_waitcnt at the end); 512 workgroups of 512
// threads = 2 per CU, 16 waves per CU (the GEMM's occupancy).  Prints bytes / clock / CU at the measured shader clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
enum { W_B32 = 0, W_B64 = 1, W_2B64 = 2, W_B128 = 3, R_B128 = 4, R_B64 = 5, W_2B64_LIN = 6, W_B128_LIN = 7 };

template <int KIND>
__global__ __launch_bounds__(512) void lds_kernel(float *out, int iters, long long *clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    // GEMM staging pattern: chunk c = tid -> row (c/8 with the 8-row interleave), float column 2*(c%8) of a 36-float pitch
    const int rr = tid / 8, row = (rr & ~7) | ((rr & 1) << 2) | ((rr >> 1) & 3);
    unsigned addr;
    if (KIND == W_2B64_LIN || KIND == W_B128_LIN) addr = tid * 16;                           // linear: 16 bytes per lane
    else if (KIND == W_B128 || KIND == R_B128) addr = (row * 36 + 4 * (tid % 8)) * 4;       // 16-byte aligned column
    else addr = (row * 36 + 2 * (tid % 8)) * 4;
    float v0 = tid * 1.0f, v1 = tid * 2.0f, v2 = tid * 3.0f, v3 = tid * 4.0f;
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == W_B32) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v0) : "memory");
            if (KIND == W_B64) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v2f{v0, v1}) : "memory");
            if (KIND == W_2B64 || KIND == W_2B64_LIN) {
                v2f d0 = {v0, v1}, d1 = {v2, v3};
                if (KIND == W_2B64) asm volatile("ds_write2_b64 %0, %1, %2 offset1:8" ::"v"(addr), "v"(d0), "v"(d1) : "memory");
                else asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" ::"v"(addr), "v"(d0), "v"(d1) : "memory");
            }
            if (KIND == W_B128 || KIND == W_B128_LIN) {
                v4f q = {v0, v1, v2, v3};
                asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(q) : "memory");
            }
            if (KIND == R_B128) {
                v4f q;
                asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                r0 += q.x;
            }
            if (KIND == R_B64) {
                v2f q;
                asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                r1 += q.x;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    __syncthreads();
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    if (r0 + r1 + r2 + r3 == 12345.0f) out[tid] = smem[tid];
}

template <int KIND>
static void run(const char *name, int bytes_per_lane, float *dout, long long *dclk, hipStream_t s) {
    const int iters = 2000;
    const size_t lds = 64 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(512), dim3(512), lds, s, dout, 10, dclk);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(512), dim3(512), lds, s, dout, iters, dclk);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long clk;
    CK(hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost));
    // per CU: 16 waves x iters x 8 instructions x 64 lanes x bytes
    const double bytes_cu = 16.0 * iters * 8 * 64 * bytes_per_lane;
    printf("%-34s %8.3f ms   block-0 clocks %10lld   %6.1f B/clk/CU (clock64 of block 0)   %6.1f B/clk/CU at 2.4 GHz wall\n", name, ms, clk,
           bytes_cu / (double)clk, bytes_cu / (ms * 1e-3 * 2.4e9));
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float *dout;
    long long *dclk;
    CK(hipMalloc(&dout, 4096));
    CK(hipMalloc(&dclk, 64));
    run<W_B32>("ds_write_b32  (pitch-36 pattern)", 4, dout, dclk, s);
    run<W_B64>("ds_write_b64  (pitch-36 pattern)", 8, dout, dclk, s);
    run<W_2B64>("ds_write2_b64 (GEMM staging)", 16, dout, dclk, s);
    run<W_2B64_LIN>("ds_write2_b64 (linear 16 B / lane)", 16, dout, dclk, s);
    run<W_B128>("ds_write_b128 (pitch-36 pattern)", 16, dout, dclk, s);
    run<W_B128_LIN>("ds_write_b128 (linear 16 B / lane)", 16, dout, dclk, s);
    run<R_B128>("ds_read_b128  (pitch-36 pattern)", 16, dout, dclk, s);
    run<R_B64>("ds_read_b64   (pitch-36 pattern)", 8, dout, dclk, s);
    return 0;
}
