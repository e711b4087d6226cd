// tools/ubench/attn_bf16_bench.cpp -- times the bf16 relative-position attention kernel (kernels/attention_bf16.hip) on configs[2]'s shape
// (tdt-600m, 32 x 30 s: T = 376, 8 heads of 128) and on the 110M shape, and prints where a wavefront's clocks go (AB_TRACE phase sums).
// Random inputs; numerics are checked by tests/test_gpu_bf16.py, not here.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define AB_TRACE 1
#include "../../parakeet.cpp_amd/csrc/kernels/attention_bf16.hip"

using namespace pk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

static void run(const char *name, int B, int T, int d, int H, int reps, hipStream_t s) {
    const int P = 2 * T - 1, hd = d / H;
    const size_t nq = (size_t)B * T * 3 * d, np = (size_t)P * d;
    std::vector<unsigned short> h(nq > np ? nq : np);
    std::vector<float> hf((size_t)H * P > (size_t)d ? (size_t)H * P : d);
    unsigned x = 777u;
    auto rnd = [&](float sc) { x = x * 1664525u + 1013904223u; return ((int)(x >> 8) - (1 << 23)) * (sc / (1 << 23)); };
    unsigned short *qkv, *pos, *ctx;
    float *cvec, *bu;
    CK(hipMalloc(&qkv, nq * 2)); CK(hipMalloc(&pos, np * 2)); CK(hipMalloc(&ctx, (size_t)B * T * d * 2));
    CK(hipMalloc(&cvec, (size_t)H * P * 4)); CK(hipMalloc(&bu, d * 4));
    for (size_t i = 0; i < nq; ++i) h[i] = f2bf(rnd(1.0f));
    CK(hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < np; ++i) h[i] = f2bf(rnd(1.0f));
    CK(hipMemcpy(pos, h.data(), np * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)H * P; ++i) hf[i] = rnd(0.5f);
    CK(hipMemcpy(cvec, hf.data(), (size_t)H * P * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < d; ++i) hf[i] = rnd(0.1f);
    CK(hipMemcpy(bu, hf.data(), d * 4, hipMemcpyHostToDevice));
    const int n_qb = (T + AB_QB - 1) / AB_QB, n_wg = ((B * H + 7) / 8) * 8 * n_qb;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_relpos_attention_bf16(qkv, B, T, d, H, pos, cvec, bu, ctx, s, 0, SeqRag{});
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_relpos_attention_bf16(qkv, B, T, d, H, pos, cvec, bu, ctx, s, 0, SeqRag{});
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)B * H * (2.0 * T * T * hd * 2 + 2.0 * T * T * hd);
    printf("%-24s B=%d T=%d d=%d H=%d : %8.1f us  %6.1f TF  (%d workgroups of %d query rows, LDS %zu B)\n", name, B, T, d, H, ms / reps * 1e3,
           fl / (ms / reps) * 1e-9, n_wg, AB_QB, relpos_attention_bf16_lds_bytes(T, hd));
    long long *dtr;
    CK(hipMalloc(&dtr, (size_t)n_wg * 4 * 8 * 8));
    CK(hipMemset(dtr, 0, (size_t)n_wg * 4 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ab_trace), &dtr, 8));
    launch_relpos_attention_bf16(qkv, B, T, d, H, pos, cvec, bu, ctx, s, 0, SeqRag{});
    CK(hipStreamSynchronize(s));
    long long *null = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ab_trace), &null, 8));
    std::vector<long long> tr((size_t)n_wg * 4 * 8);
    CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
    const char *ph[7] = {"prologue (Q, K0, P0, V0)", "QK^T + QP^T MFMAs", "skew strip + mask + max", "exp2 + sums + rescale", "PV MFMAs", "V staging store", "barrier"};
    double sum[7] = {0}, tot = 0;
    size_t n = 0;
    for (size_t w = 0; w < (size_t)n_wg * 4; ++w) {
        const long long *t = &tr[w * 8];
        if (!t[1]) continue;                                          // inactive wave / padding slot
        for (int i = 0; i < 7; ++i) { sum[i] += (double)t[i]; tot += (double)t[i]; }
        ++n;
    }
    const int nkt = (T + 31) / 32;
    printf("   per-wave shader clocks (mean over %zu active waves, %d key tiles each): total %.0f = %.0f per key tile\n", n, nkt, tot / n, tot / n / nkt);
    for (int i = 0; i < 7; ++i) printf("      %-28s %9.0f  %5.1f %%   %7.0f per tile\n", ph[i], sum[i] / n, 100.0 * sum[i] / tot, sum[i] / n / nkt);
    CK(hipFree(qkv)); CK(hipFree(pos)); CK(hipFree(ctx)); CK(hipFree(cvec)); CK(hipFree(bu)); CK(hipFree(dtr));
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    run("tdt-600m 32 x 30 s", 32, 376, 1024, 8, reps, s);
    run("tdt-600m 64 x 10 s", 64, 126, 1024, 8, reps, s);
    run("110m-width 64 x 10 s", 64, 126, 512, 8, reps, s);
    return 0;
}
