// tools/ubench/gemm_sweep.cpp -- times every tile variant of the fp32-MFMA GEMMs on the encoder's real shapes and checks
// each against the first-generation kernel bit for bit.  Build: make -C tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#define GP_CLOCKPROBE 1
#include "gemm_pipe_exp.hpp"      // instrumented copies; they shadow the product headers through their include guards
#include "gemm_bf16_exp.hpp"
#include "../../parakeet.cpp_amd/csrc/kernels/gemm.hip"
#include "gemm_dma.hpp"
#include "gemm_pd.hpp"
#include "gemm_bf16p.hpp"
#include "../../parakeet.cpp_amd/csrc/kernels/gemm_smallm.hip"   // first-generation kernel + launch_gemm
#include "../../parakeet.cpp_amd/csrc/kernels/gemm_smallm_bf16.hip"   // (launch_gemm_bf16 routes small M there)

using namespace pk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void mfma_clock_kernel(long long *out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)s; }
}

__global__ void stamp_kernel(long long *out) { out[0] = wall_clock64(); }

struct Shape { const char *name; int M, N, K, epi; };
struct Variant { const char *name; std::function<void(const GemmArgs &, int, hipStream_t)> run; };

template <int WGM, int WGN, int TM, int TN, int BK>
static void run_pipe(const GemmArgs &a, int epi, hipStream_t s) {
    switch (epi) {
    case EPI_NONE: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_NONE>(a, s); break;
    case EPI_RELU: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_RELU>(a, s); break;
    case EPI_SILU: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_SILU>(a, s); break;
    case EPI_RESID: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_RESID>(a, s); break;
    case EPI_GLU: if constexpr (TN % 2 == 0) launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_GLU>(a, s); break;
    }
}
template <int WGM, int WGN, int TM, int TN, int BK>
static void run_sb(const GemmArgs &a, int epi, hipStream_t s) {       // single LDS staging buffer
    switch (epi) {
    case EPI_NONE: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_NONE, 1>(a, s); break;
    case EPI_RELU: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_RELU, 1>(a, s); break;
    case EPI_SILU: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_SILU, 1>(a, s); break;
    case EPI_RESID: launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_RESID, 1>(a, s); break;
    case EPI_GLU: if constexpr (TN % 2 == 0) launch_gemm_pipe<WGM, WGN, TM, TN, BK, EPI_GLU, 1>(a, s); break;
    }
}
template <int WGM, int WGN, int TM, int TN>
static void run_dma(const GemmArgs &a, int epi, hipStream_t s) {      // direct-to-LDS staging
    switch (epi) {
    case EPI_NONE: launch_gemm_dma<WGM, WGN, TM, TN, EPI_NONE>(a, s); break;
    case EPI_RELU: launch_gemm_dma<WGM, WGN, TM, TN, EPI_RELU>(a, s); break;
    case EPI_SILU: launch_gemm_dma<WGM, WGN, TM, TN, EPI_SILU>(a, s); break;
    case EPI_RESID: launch_gemm_dma<WGM, WGN, TM, TN, EPI_RESID>(a, s); break;
    case EPI_GLU: if constexpr (TN % 2 == 0) launch_gemm_dma<WGM, WGN, TM, TN, EPI_GLU>(a, s); break;
    }
}
static void run_pd(const GemmArgs &a, int epi, hipStream_t s) {       // persistent workgroups, direct epilogue (gemm_pd.hpp); falls back to sb
    bool ok = false;
    switch (epi) {
    case EPI_NONE: ok = launch_gemm_pd<EPI_NONE>(a, s); break;
    case EPI_RELU: ok = launch_gemm_pd<EPI_RELU>(a, s); break;
    case EPI_SILU: ok = launch_gemm_pd<EPI_SILU>(a, s); break;
    case EPI_RESID: ok = launch_gemm_pd<EPI_RESID>(a, s); break;
    default: break;
    }
    if (!ok) run_sb<4, 2, 1, 2, 32>(a, epi, s);
}
template <int BM, int BN>
static void run_old(const GemmArgs &a, int epi, hipStream_t s) {
    switch (epi) {
    case EPI_NONE: launch_one<BM, BN, EPI_NONE>(a, s); break;
    case EPI_RELU: launch_one<BM, BN, EPI_RELU>(a, s); break;
    case EPI_SILU: launch_one<BM, BN, EPI_SILU>(a, s); break;
    case EPI_RESID: launch_one<BM, BN, EPI_RESID>(a, s); break;
    case EPI_GLU: if constexpr (BN == 128) launch_one<BM, BN, EPI_GLU>(a, s); break;
    }
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    {   // sustained MFMA peak on this box
        float *d;
        CK(hipMalloc(&d, 4));
        const int iters = 4000;
        hipLaunchKernelGGL(mfma_peak_kernel, dim3(256 * 2), dim3(256), 0, s, d, 100);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(mfma_peak_kernel, dim3(256 * 2), dim3(256), 0, s, d, iters);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 512.0 * 4 * (double)iters * 32 * 2.0 * 32 * 32 * 2;
        printf("mfma_peak: %.1f TF (%.3f ms)\n", fl / ms * 1e-9, ms);
        long long *dc, hc[3];
        CK(hipMalloc(&dc, 24));
        hipLaunchKernelGGL(mfma_clock_kernel, dim3(256 * 2), dim3(256), 0, s, dc, iters);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost));
        int wrate = 0;
        (void)hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
        printf("clock probe: clock64 delta %lld, wall_clock64 delta %lld (wall rate %d kHz) -> clock64 rate %.1f MHz; MFMA cycles/instr %.2f\n", hc[0], hc[1], wrate,
               (double)hc[0] / ((double)hc[1] / (wrate * 1e3)) * 1e-6, (double)hc[0] / (iters * 32.0) / 2.0);
    }
    const std::vector<Shape> shapes = {
        {"fc1_silu   8064x2048x512 ", 8064, 2048, 512, EPI_SILU},
        {"fc2_resid  8064x512x2048 ", 8064, 512, 2048, EPI_RESID},
        {"qkv        8064x1536x512 ", 8064, 1536, 512, EPI_NONE},
        {"out_resid  8064x512x512  ", 8064, 512, 512, EPI_RESID},
        {"pw1_glu    8064x512x512  ", 8064, 512, 512, EPI_GLU},
        {"sub_pw     321280x256x256", 321280, 256, 256, EPI_RELU},
        {"sub_proj   8064x512x2560 ", 8064, 512, 2560, EPI_NONE},
        {"ctc_head   8064x1025x512 ", 8064, 1025, 512, EPI_NONE},
        {"B fc1      12032x4096x1024", 12032, 4096, 1024, EPI_SILU},
        {"B fc2      12032x1024x4096", 12032, 1024, 4096, EPI_RESID},
        {"B out      12032x1024x1024", 12032, 1024, 1024, EPI_RESID},
    };
    const std::vector<Variant> variants = {
        {"old 128x128", run_old<128, 128>},
        {"old 128x64", run_old<128, 64>},
        {"old 64x64", run_old<64, 64>},
        {"pipe 64x64   w32x32 bk32", run_pipe<2, 2, 1, 1, 32>},
        {"pipe 64x64   w32x32 bk64", run_pipe<2, 2, 1, 1, 64>},
        {"pipe 128x64  w64x32 bk32", run_pipe<2, 2, 2, 1, 32>},
        {"pipe 64x128  w32x64 bk32", run_pipe<2, 2, 1, 2, 32>},
        {"pipe 128x128 w64x64 bk32", run_pipe<2, 2, 2, 2, 32>},
        {"pipe 128x128 w64x32 bk32 512t", run_pipe<2, 4, 2, 1, 32>},
        {"pipe 128x128 w32x64 bk32 512t", run_pipe<4, 2, 1, 2, 32>},
        {"pipe 128x128 w32x64 bk64 512t", run_pipe<4, 2, 1, 2, 64>},
        {"pipe 128x128 w64x32 bk64 512t", run_pipe<2, 4, 2, 1, 64>},
        {"pipe 128x128 w32x64 bk64 512t", run_pipe<4, 2, 1, 2, 64>},
        {"pipe 128x128 w64x32 bk64 512t", run_pipe<2, 4, 2, 1, 64>},
        {"dma  128x128 w64x64 256t 512t-class", run_dma<2, 2, 2, 2>},
        {"dma  128x128 w32x64 512t", run_dma<4, 2, 1, 2>},
        {"dma  128x128 w64x32 512t", run_dma<2, 4, 2, 1>},
        {"dma  64x128  w32x64 256t 512t-class", run_dma<2, 2, 1, 2>},
        {"dma  128x64  w64x32 256t 512t-class", run_dma<2, 2, 2, 1>},
        {"dma  64x128  w32x32 512t", run_dma<2, 4, 1, 1>},
        {"dma  64x64   w32x32 256t 512t-class", run_dma<2, 2, 1, 1>},
        {"sb   128x128 w64x64 bk32 256t 512t-class", run_sb<2, 2, 2, 2, 32>},
        {"sb   64x128  w32x64 bk32 256t 512t-class", run_sb<2, 2, 1, 2, 32>},
        {"sb   128x128 w32x64 bk32 512t", run_sb<4, 2, 1, 2, 32>},
        {"pd   128x128 w32x64 bk32 512t persistent", run_pd},
        {"sb   128x128 w64x32 bk32 512t", run_sb<2, 4, 2, 1, 32>},
        {"sb   256x128 w64x64 bk32 512t", run_sb<4, 2, 2, 2, 32>},
        {"sb   128x256 w64x64 bk32 512t", run_sb<2, 4, 2, 2, 32>},
        {"sb   64x128  w32x32 bk32 512t", run_sb<2, 4, 1, 1, 32>},
        {"sb   128x64  w32x32 bk32 512t", run_sb<4, 2, 1, 1, 32>},
        {"pipe 128x64  w32x32 bk32 512t", run_pipe<4, 2, 1, 1, 32>},
        {"pipe 64x128  w32x32 bk32 512t", run_pipe<2, 4, 1, 1, 32>},
        // round 6 (round-5 verdict item 4): the long-K / narrow products on single-buffered BK 64 tiles that run TWO (or three) workgroups per CU
        {"sb   128x128 w32x64 bk64 512t (production fc2)", run_sb<4, 2, 1, 2, 64>},
        {"sb   64x128  w32x64 bk64 256t", run_sb<2, 2, 1, 2, 64>},
        {"sb   128x64  w64x32 bk64 256t", run_sb<2, 2, 2, 1, 64>},
        {"sb   64x128  w32x32 bk64 512t", run_sb<2, 4, 1, 1, 64>},
        {"sb   128x64  w32x32 bk64 512t", run_sb<4, 2, 1, 1, 64>},
        {"sb   64x64   w32x32 bk64 256t", run_sb<2, 2, 1, 1, 64>},
    };
    size_t maxA = 0, maxW = 0, maxO = 0;
    for (auto &sh : shapes) {
        maxA = std::max(maxA, (size_t)sh.M * sh.K);
        maxW = std::max(maxW, (size_t)sh.N * sh.K * 2);
        maxO = std::max(maxO, (size_t)sh.M * sh.N);
    }
    float *dA, *dW, *dB, *dR, *dO, *dRef;
    CK(hipMalloc(&dA, maxA * 4)); CK(hipMalloc(&dW, maxW * 4)); CK(hipMalloc(&dB, 16384 * 4));
    CK(hipMalloc(&dR, maxO * 4)); CK(hipMalloc(&dO, maxO * 4)); CK(hipMalloc(&dRef, maxO * 4));
    {
        std::vector<float> h(std::max(std::max(maxA, maxW), maxO));
        unsigned x = 12345u;
        auto fill = [&](float *d, size_t n, float sc) {
            for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (sc / (1 << 23)); }
            CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        };
        fill(dA, maxA, 1.0f); fill(dW, maxW, 0.05f); fill(dB, 16384, 0.1f); fill(dR, maxO, 1.0f);
    }
    {   // clocks ramp for tens of milliseconds after idle: 0.3 s of GEMMs before anything is timed
        GemmArgs g{dA, 512, dW, 512, dB, dO, 2048, dR, 2048, 0.5f, 8064, 2048, 512};
        for (int i = 0; i < 1500; ++i) run_pipe<4, 2, 1, 2, 32>(g, EPI_NONE, s);
        CK(hipStreamSynchronize(s));
    }
    if (argc > 2 && strcmp(argv[2], "trace") == 0) {   // per-workgroup timeline of one launch -> gpurun_out/gemm_trace_<name>.bin
        struct KV { const char *name; std::function<void(const GemmArgs &, int, hipStream_t)> run; int nblk; };
        const std::vector<KV> kv = {{"p128x128_512t", run_pipe<2, 4, 2, 1, 32>, 63 * 16}, {"p64x64", run_pipe<2, 2, 1, 1, 32>, 126 * 32},
                                    {"p128x128_w64x64", run_pipe<2, 2, 2, 2, 32>, 63 * 16}, {"p128x128_w32x64_512t", run_pipe<4, 2, 1, 2, 32>, 63 * 16},
                                    {"sb128x128_w32x64_512t", run_sb<4, 2, 1, 2, 32>, 63 * 16}, {"sb256x256_512t", run_sb<2, 4, 4, 2, 32>, 32 * 8}};
        long long *dtrace;
        CK(hipMalloc(&dtrace, 8192 * 8 * 8));
        for (auto &v : kv) {
            GemmArgs g{dA, 512, dW, 512, dB, dO, 2048, dR, 2048, 0.5f, 8064, 2048, 512};
            long long *null = nullptr;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(gp_trace), &null, 8));
            for (int i = 0; i < 3; ++i) v.run(g, EPI_SILU, s);
            CK(hipStreamSynchronize(s));
            CK(hipMemset(dtrace, 0, 8192 * 8 * 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(gp_trace), &dtrace, 8));
            v.run(g, EPI_SILU, s);                                               // (warm: the symbol copy above idled the queue)
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, dtrace + 8192 * 8 - 8);
            v.run(g, EPI_SILU, s);
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, dtrace + 8192 * 8 - 7);
            CK(hipStreamSynchronize(s));
            std::vector<long long> h((size_t)v.nblk * 8 + 2);
            CK(hipMemcpy(h.data(), dtrace, (size_t)v.nblk * 8 * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h.data() + (size_t)v.nblk * 8, dtrace + 8192 * 8 - 8, 16, hipMemcpyDeviceToHost));
            std::string fn = std::string("gpurun_out/gemm_trace_") + v.name + ".bin";
            FILE *f = fopen(fn.c_str(), "wb");
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
            printf("wrote %s (%d blocks)\n", fn.c_str(), v.nblk);
        }
        return 0;
    }
    if (argc > 2 && strcmp(argv[2], "cold") == 0) {   // weights hot (same W every launch) vs cold (a different 4 MB W of a 512 MB pool per launch)
        float *pool;
        const size_t wfl = (size_t)2048 * 512, NPOOL = 128;
        CK(hipMalloc(&pool, wfl * 4 * NPOOL));
        for (size_t i = 0; i < NPOOL; ++i) CK(hipMemcpy(pool + i * wfl, dW, wfl * 4, hipMemcpyDeviceToDevice));
        struct SH { const char *name; int N, K, epi; std::function<void(const GemmArgs &, int, hipStream_t)> run; };
        const std::vector<SH> shs = {{"fc1 sb 128x128 w32x64", 2048, 512, EPI_SILU, run_sb<4, 2, 1, 2, 32>}, {"fc1 pipe 128x128 w32x64", 2048, 512, EPI_SILU, run_pipe<4, 2, 1, 2, 32>},
                                     {"fc2 sb 128x128 w32x64 bk64", 512, 2048, EPI_RESID, run_sb<4, 2, 1, 2, 64>}, {"fc2 pipe 128x128 w64x32 bk64", 512, 2048, EPI_RESID, run_pipe<2, 4, 2, 1, 64>},
                                     {"qkv sb 128x128 w32x64", 1536, 512, EPI_NONE, run_sb<4, 2, 1, 2, 32>}};
        for (auto &sh : shs)
            for (int cold = 0; cold < 3; ++cold) {      // 0 hot, 1 cold W, 2 cold W + the A operand overwritten (device memset = L2/MALL churn) before every launch
                float tot = 0;
                for (int i = 0; i < reps + 2; ++i) {
                    GemmArgs g{dA, sh.K, cold ? pool + (size_t)(i % NPOOL) * wfl : dW, sh.K, dB, dO, sh.N, dR, sh.N, 0.5f, 8064, sh.N, sh.K};
                    if (cold == 2) CK(hipMemsetAsync(dRef, 0, (size_t)64 << 20, s));
                    CK(hipEventRecord(e0, s));
                    sh.run(g, sh.epi, s);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (i >= 2) tot += ms;
                }
                printf("%-32s %s : %.1f us\n", sh.name, cold == 0 ? "hot W        " : cold == 1 ? "cold W       " : "cold W + churn", tot / reps * 1e3);
            }
        return 0;
    }
    if (argc > 2 && strcmp(argv[2], "trace2") == 0) {   // per-wave shader-clock stamps around the K-loop barrier -> gpurun_out/gemm_tr2_<name>.bin
        struct KV { const char *name; std::function<void(const GemmArgs &, int, hipStream_t)> run; int nblk; };
        const std::vector<KV> kv = {{"w32x64_512t", run_pipe<4, 2, 1, 2, 32>, 63 * 16}, {"w64x64_256t", run_pipe<2, 2, 2, 2, 32>, 63 * 16}};
        long long *dtr, *dtrace;
        const size_t n2 = (size_t)1008 * 16 * (GP_TR2_IT + 1) * 2;
        CK(hipMalloc(&dtr, n2 * 8));
        CK(hipMalloc(&dtrace, 8192 * 8 * 8));
        for (auto &v : kv) {
            GemmArgs g{dA, 2048, dW, 2048, dB, dO, 2048, dR, 2048, 0.5f, 8064, 2048, 2048};
            for (int i = 0; i < 3; ++i) v.run(g, EPI_NONE, s);
            CK(hipStreamSynchronize(s));
            CK(hipMemset(dtr, 0, n2 * 8));
            CK(hipMemset(dtrace, 0, 8192 * 8 * 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(gp_tr2), &dtr, 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(gp_trace), &dtrace, 8));
            v.run(g, EPI_NONE, s);
            CK(hipStreamSynchronize(s));
            long long *null = nullptr;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(gp_tr2), &null, 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(gp_trace), &null, 8));
            std::vector<long long> h(n2), h1((size_t)v.nblk * 8);
            CK(hipMemcpy(h.data(), dtr, n2 * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h1.data(), dtrace, h1.size() * 8, hipMemcpyDeviceToHost));
            std::string fn = std::string("gpurun_out/gemm_tr2_") + v.name + ".bin";
            FILE *f = fopen(fn.c_str(), "wb");
            fwrite(h.data(), 8, h.size(), f);
            fwrite(h1.data(), 8, h1.size(), f);
            fclose(f);
            printf("wrote %s\n", fn.c_str());
        }
        return 0;
    }
    if (argc > 2 && strcmp(argv[2], "bf16") == 0) {   // bf16-operand kernels on the tdt-600m shapes (timing only: W bits are reinterpreted)
        struct KV { const char *name; std::function<void(const GemmArgs &, hipStream_t)> silu, resid; };
        const std::vector<KV> kv = {
            {"bf16 128x128 8w 32x64", launch_gemm_bf16_t<4, 2, 1, 2, EPI_SILU>, launch_gemm_bf16_t<4, 2, 1, 2, EPI_RESID>},
            {"bf16 128x128 8w 64x32", launch_gemm_bf16_t<2, 4, 2, 1, EPI_SILU>, launch_gemm_bf16_t<2, 4, 2, 1, EPI_RESID>},
            {"bf16 128x128 4w 64x64", launch_gemm_bf16_t<2, 2, 2, 2, EPI_SILU>, launch_gemm_bf16_t<2, 2, 2, 2, EPI_RESID>},
            {"bf16 128x64  4w 64x32", launch_gemm_bf16_t<2, 2, 2, 1, EPI_SILU>, launch_gemm_bf16_t<2, 2, 2, 1, EPI_RESID>},
            {"bf16 64x128  4w 32x64", launch_gemm_bf16_t<2, 2, 1, 2, EPI_SILU>, launch_gemm_bf16_t<2, 2, 1, 2, EPI_RESID>},
            {"bf16 256x256 8w 128x64", launch_gemm_bf16_t<2, 4, 4, 2, EPI_SILU>, launch_gemm_bf16_t<2, 4, 4, 2, EPI_RESID>},
            {"bf16 256x256 8w 64x128", launch_gemm_bf16_t<4, 2, 2, 4, EPI_SILU>, launch_gemm_bf16_t<4, 2, 2, 4, EPI_RESID>},
            {"bf16 128x256 4w 64x128", launch_gemm_bf16_t<2, 2, 2, 4, EPI_SILU>, launch_gemm_bf16_t<2, 2, 2, 4, EPI_RESID>},
            {"bf16 256x128 4w 128x64", launch_gemm_bf16_t<2, 2, 4, 2, EPI_SILU>, launch_gemm_bf16_t<2, 2, 4, 2, EPI_RESID>},
            {"bf16 128x256 8w 64x64", launch_gemm_bf16_t<2, 4, 2, 2, EPI_SILU>, launch_gemm_bf16_t<2, 4, 2, 2, EPI_RESID>},
            {"bf16 256x128 8w 64x64", launch_gemm_bf16_t<4, 2, 2, 2, EPI_SILU>, launch_gemm_bf16_t<4, 2, 2, 2, EPI_RESID>},
        };
        // bf16 copy of A for the variants whose A operand is already bf16 in HBM (GemmArgs::a_bf16)
        __bf16 *dA16;
        {
            const size_t n = (size_t)12032 * 4096;
            std::vector<float> hf(n);
            CK(hipMemcpy(hf.data(), dA, n * 4, hipMemcpyDeviceToHost));
            std::vector<unsigned short> hb(n);
            for (size_t i = 0; i < n; ++i) { unsigned u; memcpy(&u, &hf[i], 4); hb[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
            CK(hipMalloc(&dA16, n * 2));
            CK(hipMemcpy(dA16, hb.data(), n * 2, hipMemcpyHostToDevice));
        }
        const std::vector<KV> kv16 = {
            {"bf16A 128x128 8w 32x64", launch_gemm_bf16_t<4, 2, 1, 2, EPI_SILU, true>, launch_gemm_bf16_t<4, 2, 1, 2, EPI_RESID, true>},
            {"bf16A 256x128 8w 64x64", launch_gemm_bf16_t<4, 2, 2, 2, EPI_SILU, true>, launch_gemm_bf16_t<4, 2, 2, 2, EPI_RESID, true>},
            {"bf16A 128x256 8w 64x64", launch_gemm_bf16_t<2, 4, 2, 2, EPI_SILU, true>, launch_gemm_bf16_t<2, 4, 2, 2, EPI_RESID, true>},
            {"bf16A 256x256 8w 64x128", launch_gemm_bf16_t<4, 2, 2, 4, EPI_SILU, true>, launch_gemm_bf16_t<4, 2, 2, 4, EPI_RESID, true>},
            {"bf16A 256x256 8w 128x64", launch_gemm_bf16_t<2, 4, 4, 2, EPI_SILU, true>, launch_gemm_bf16_t<2, 4, 4, 2, EPI_RESID, true>},
            {"bf16A 128x128 4w 64x64", launch_gemm_bf16_t<2, 2, 2, 2, EPI_SILU, true>, launch_gemm_bf16_t<2, 2, 2, 2, EPI_RESID, true>},
            {"bf16P 128x128 8w 32x64 pf3", launch_gemm_bf16p_t<4, 2, 1, 2, EPI_SILU, 3>, launch_gemm_bf16p_t<4, 2, 1, 2, EPI_RESID, 3>},
            {"bf16P 256x128 8w 64x64 pf2", launch_gemm_bf16p_t<4, 2, 2, 2, EPI_SILU, 2>, launch_gemm_bf16p_t<4, 2, 2, 2, EPI_RESID, 2>},
            {"bf16P 256x128 8w 64x64 pf3", launch_gemm_bf16p_t<4, 2, 2, 2, EPI_SILU, 3>, launch_gemm_bf16p_t<4, 2, 2, 2, EPI_RESID, 3>},
            {"bf16P 256x128 8w 64x64 pf4", launch_gemm_bf16p_t<4, 2, 2, 2, EPI_SILU, 4>, launch_gemm_bf16p_t<4, 2, 2, 2, EPI_RESID, 4>},
            {"bf16P 128x256 8w 64x64 pf3", launch_gemm_bf16p_t<2, 4, 2, 2, EPI_SILU, 3>, launch_gemm_bf16p_t<2, 4, 2, 2, EPI_RESID, 3>},
            {"bf16P 256x256 8w 64x128 pf2", launch_gemm_bf16p_t<4, 2, 2, 4, EPI_SILU, 2>, launch_gemm_bf16p_t<4, 2, 2, 4, EPI_RESID, 2>},
            {"bf16P 256x256 8w 64x128 pf3", launch_gemm_bf16p_t<4, 2, 2, 4, EPI_SILU, 3>, launch_gemm_bf16p_t<4, 2, 2, 4, EPI_RESID, 3>},
        };
        struct SH { const char *name; int M, N, K; bool resid; };
        const std::vector<SH> shs = {{"B fc1 12032x4096x1024 silu", 12032, 4096, 1024, false}, {"B fc2 12032x1024x4096 resid", 12032, 1024, 4096, true},
                                     {"B qkv 12032x3072x1024", 12032, 3072, 1024, false}, {"B out 12032x1024x1024 resid", 12032, 1024, 1024, true},
                                     {"A fc1 8064x2048x512 silu", 8064, 2048, 512, false}};
        for (auto &sh : shs) {
            printf("== %s  %.1f GFLOP\n", sh.name, 2.0 * sh.M * sh.N * sh.K * 1e-9);
            for (auto &v : kv) {
                GemmArgs g{dA, sh.K, dW, sh.K, dB, dO, sh.N, dR, sh.N, 0.5f, sh.M, sh.N, sh.K};
                auto &run = sh.resid ? v.resid : v.silu;
                for (int i = 0; i < 3; ++i) run(g, s);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) run(g, s);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                printf("   %-28s %8.1f us  %7.1f TF\n", v.name, ms * 1e3, 2.0 * sh.M * sh.N * sh.K / ms * 1e-9);
            }
            for (auto &v : kv16) {
                GemmArgs g{reinterpret_cast<const float *>(dA16), sh.K, dW, sh.K, dB, dO, sh.N, dR, sh.N, 0.5f, sh.M, sh.N, sh.K};
                g.a_bf16 = 1;
                auto &run = sh.resid ? v.resid : v.silu;
                for (int i = 0; i < 3; ++i) run(g, s);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) run(g, s);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                printf("   %-28s %8.1f us  %7.1f TF\n", v.name, ms * 1e3, 2.0 * sh.M * sh.N * sh.K / ms * 1e-9);
            }
        }
        return 0;
    }
    if (argc > 2 && strcmp(argv[2], "ml") == 0) {   // main-loop experiments (GP_EXP builds): production tiles, no epilogue math, two K
        struct KV { const char *name; std::function<void(const GemmArgs &, int, hipStream_t)> run; };
        const std::vector<KV> kv = {{"pipe 128x128 w32x64 512t", run_pipe<4, 2, 1, 2, 32>}, {"pipe 128x128 w64x64 256t", run_pipe<2, 2, 2, 2, 32>},
                                    {"pipe 64x64 w32x32 256t", run_pipe<2, 2, 1, 1, 32>}, {"sb 128x128 w64x64 256t", run_sb<2, 2, 2, 2, 32>},
                                    {"sb 64x128 w32x64 256t", run_sb<2, 2, 1, 2, 32>}, {"sb 128x128 w32x64 512t", run_sb<4, 2, 1, 2, 32>},
                                    {"pipe 128x256 w64x128 256t", run_pipe<2, 2, 2, 4, 32>}, {"pipe 256x128 w128x64 256t", run_pipe<2, 2, 4, 2, 32>},
                                    {"pipe 128x256 w64x64 512t", run_pipe<2, 4, 2, 2, 32>}, {"pipe 256x256 w128x64 512t", run_pipe<2, 4, 4, 2, 32>},
                                    {"sb 256x256 w128x64 512t", run_sb<2, 4, 4, 2, 32>}, {"sb 128x256 w64x128 256t", run_sb<2, 2, 2, 4, 32>},
                                    {"dma 128x128 w64x64 256t", run_dma<2, 2, 2, 2>}, {"dma 128x128 w32x64 512t", run_dma<4, 2, 1, 2>},
                                    {"dma 64x128 w32x64 256t", run_dma<2, 2, 1, 2>}};
        const int mlN = argc > 4 ? atoi(argv[4]) : 2048, Klo = argc > 5 ? atoi(argv[5]) : 512, Khi = argc > 6 ? atoi(argv[6]) : 2048;
        std::vector<KV> kv2 = kv;
        if (argc > 4) {
            kv2 = {{"pd 128x128 w32x64 512t persistent", run_pd}, {"pipe 128x128 w64x32 512t bk64", run_pipe<2, 4, 2, 1, 64>}, {"sb 128x128 w64x32 512t bk64", run_sb<2, 4, 2, 1, 64>},
                   {"pipe 128x128 w32x64 512t", run_pipe<4, 2, 1, 2, 32>}, {"sb 128x128 w32x64 512t", run_sb<4, 2, 1, 2, 32>},
                   {"sb 128x128 w32x64 512t bk64", run_sb<4, 2, 1, 2, 64>},
                   {"pipe 128x128 w64x64 256t", run_pipe<2, 2, 2, 2, 32>}, {"sb 128x128 w64x64 256t", run_sb<2, 2, 2, 2, 32>},
                   {"sb 128x128 w64x64 256t bk64", run_sb<2, 2, 2, 2, 64>},
                   {"pipe 64x128 w32x32 512t", run_pipe<2, 4, 1, 1, 32>}, {"sb 64x128 w32x64 256t", run_sb<2, 2, 1, 2, 32>},
                   {"pipe 128x64 w64x32 256t", run_pipe<2, 2, 2, 1, 32>}, {"sb 128x64 w64x32 256t", run_sb<2, 2, 2, 1, 32>},
                   {"sb 256x256 w128x64 512t", run_sb<2, 4, 4, 2, 32>}, {"sb 256x128 w128x32 512t", run_sb<2, 4, 4, 1, 32>},
                   {"sb 256x128 w64x64 512t", run_sb<4, 2, 2, 2, 32>}, {"sb 128x256 w64x64 512t", run_sb<2, 4, 2, 2, 32>}};
        }
        for (int epi : {(int)EPI_NONE, (int)EPI_SILU, (int)EPI_RESID})
        for (auto &v : kv2) {
            printf("ml GP_EXP=%d N=%d epi=%d %-30s:", GP_EXP, mlN, epi, v.name);
            float t[2];
            int i2 = 0;
            for (int K : {Klo, Khi}) {
                GemmArgs g{dA, K, dW, K, dB, dO, mlN, dR, mlN, 0.5f, 8064, mlN, K};
                for (int i = 0; i < 2; ++i) v.run(g, epi, s);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) v.run(g, epi, s);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t[i2++] = ms / reps * 1e3f;
                printf("  K=%d %.1fus (%.1f TF)", K, ms / reps * 1e3, 2.0 * 8064 * mlN * K / (ms / reps) * 1e-9);
            }
            const double slope = (t[1] - t[0]) / (double)(Khi - Klo);                      // us per k
            printf("  | main loop %.1f TF, fixed %.1f us\n", 2.0 * 8064 * mlN / slope * 1e-6, t[0] - slope * Klo);
            if (argc > 3 && argv[3][0] == 'x' && epi != EPI_NONE) continue;
        }
        return 0;
    }
    if (argc == 3) {   // K scan: time = fixed + per-K cost
        struct KV { const char *name; std::function<void(const GemmArgs &, int, hipStream_t)> run; };
        const std::vector<KV> kv = {{"old 128x128", run_old<128, 128>}, {"pipe 64x64 bk32", run_pipe<2, 2, 1, 1, 32>},
                                    {"pipe 128x128 w64x32 512t", run_pipe<2, 4, 2, 1, 32>}, {"pipe 128x128 w64x64", run_pipe<2, 2, 2, 2, 32>},};
        for (int epi : {(int)EPI_NONE, (int)EPI_SILU, (int)EPI_RESID})
            for (auto &v : kv) {
                printf("kscan epi=%d %-28s:", epi, v.name);
                for (int K : {64, 128, 256, 512, 1024, 2048}) {
                    GemmArgs g{dA, K, dW, K, dB, dO, 2048, dR, 2048, 0.5f, 8064, 2048, K};
                    for (int i = 0; i < 2; ++i) v.run(g, epi, s);
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < reps; ++i) v.run(g, epi, s);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    long long hc[2];
                    CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(gp_clk), 16));
                    printf("  K=%d %.1fus", K, ms / reps * 1e3);
                    if (K == 2048 && strncmp(v.name, "pipe", 4) == 0) printf(" [blk0 %.0f MHz]", (double)hc[0] / ((double)hc[1] / 100.0));
                }
                printf("\n");
            }
        return 0;
    }
    const bool quick = argc > 3;                                    // gemm_sweep <reps> - quick : the two FFN shapes, 512-thread variants
    std::vector<unsigned> href, hout;
    for (auto &sh : shapes) {
        if (quick && strncmp(sh.name, "B ", 2) == 0) continue;
        GemmArgs g{dA, sh.K, dW, sh.K, dB, dRef, sh.N, dR, sh.N, 0.5f, sh.M, sh.N, sh.K};
        const double flops = 2.0 * sh.M * (double)sh.N * sh.K * (sh.epi == EPI_GLU ? 2 : 1);
        const size_t no = (size_t)sh.M * sh.N;
        // reference = first-generation 64x64 kernel (always available, no GLU) or 128x128 for GLU
        CK(hipMemsetAsync(dRef, 0, no * 4, s));
        if (sh.epi == EPI_GLU) run_old<128, 128>(g, sh.epi, s); else run_old<64, 64>(g, sh.epi, s);
        CK(hipStreamSynchronize(s));
        href.resize(no);
        CK(hipMemcpy(href.data(), dRef, no * 4, hipMemcpyDeviceToHost));
        printf("== %s  %.2f GFLOP\n", sh.name, flops * 1e-9);
        for (auto &v : variants) {
            const bool is_old = strncmp(v.name, "old", 3) == 0;
            if (quick && strstr(v.name, "512t") == nullptr) continue;
            if (sh.epi == EPI_GLU) {
                if (is_old && strstr(v.name, "128x128") == nullptr) continue;
                if (strstr(v.name, "w64x32") || strstr(v.name, "w32x32")) continue;   // TN odd
            }
            g.out = dO;
            CK(hipMemsetAsync(dO, 0xff, no * 4, s));
            v.run(g, sh.epi, s);
            if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { printf("   %-40s LAUNCH FAILED\n", v.name); continue; }
            hout.resize(no);
            CK(hipMemcpy(hout.data(), dO, no * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < no; ++i) bad += hout[i] != href[i];
            for (int i = 0; i < 2; ++i) v.run(g, sh.epi, s);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) v.run(g, sh.epi, s);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            printf("   %-40s %8.1f us  %6.1f TF  %s\n", v.name, ms * 1e3, flops / ms * 1e-9, bad ? "MISMATCH" : "bit-equal");
            if (bad) printf("      (%zu of %zu elements differ)\n", bad, no);
        }
        fflush(stdout);
    }
    return 0;
}
