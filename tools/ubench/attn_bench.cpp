// tools/ubench/attn_bench.cpp -- times the relative-position attention kernel (kernels/attention.hip, compiled here with -DATT_TRACE) on the two
// benchmark shapes and prints the per-phase shader-clock breakdown of its wavefronts.  Random inputs; results are not checked here
// (tests/test_gpu_encoder.py does that bit for bit).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define ATT_TRACE 1
#include "../../parakeet.cpp_amd/csrc/kernels/attention.hip"

using namespace pk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void run(const char *name, int B, int T, int d, int H, int reps, hipStream_t s) {
    const int P = 2 * T - 1;
    const size_t nq = (size_t)B * T * 3 * d, np = (size_t)P * d;
    std::vector<float> h(nq > np ? nq : np);
    unsigned x = 777u;
    auto fill = [&](float *dptr, size_t n, float sc) {
        for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (sc / (1 << 23)); }
        CK(hipMemcpy(dptr, h.data(), n * 4, hipMemcpyHostToDevice));
    };
    float *qkv, *pos, *bu, *bv, *ctx;
    CK(hipMalloc(&qkv, nq * 4)); CK(hipMalloc(&pos, np * 4)); CK(hipMalloc(&bu, d * 4)); CK(hipMalloc(&bv, d * 4)); CK(hipMalloc(&ctx, (size_t)B * T * d * 4));
    fill(qkv, nq, 1.0f); fill(pos, np, 1.0f); fill(bu, d, 0.1f); fill(bv, d, 0.1f);
    const int hd = d / H, n_rb = (T + 31) / 32, n_wg = ((B * H + 7) / 8) * 8 * n_rb;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_relpos_attention(qkv, B, T, d, H, pos, bu, bv, ctx, s, 0.0f, nullptr);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_relpos_attention(qkv, B, T, d, H, pos, bu, bv, ctx, s, 0.0f, nullptr);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)B * H * (2.0 * T * T * hd * 2 + 2.0 * T * T * hd);
    printf("%-28s B=%d T=%d d=%d H=%d : %8.1f us  %6.1f TF  (%d workgroups, LDS %zu B)\n", name, B, T, d, H, ms / reps * 1e3, fl / (ms / reps) * 1e-9, n_wg,
           relpos_attention_lds_bytes(T, hd));
    long long *dtr;
    CK(hipMalloc(&dtr, (size_t)n_wg * 4 * 8 * 8));
    CK(hipMemset(dtr, 0, (size_t)n_wg * 4 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(att_trace), &dtr, 8));
    launch_relpos_attention(qkv, B, T, d, H, pos, bu, bv, ctx, s, 0.0f, nullptr);
    CK(hipStreamSynchronize(s));
    long long *null = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(att_trace), &null, 8));
    std::vector<long long> tr((size_t)n_wg * 4 * 8);
    CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
    const char *ph[7] = {"Q load + bias", "QK^T -> S", "(q+v), P request, barrier", "QP^T shifted rmw", "barrier", "softmax", "V commit + AV + store"};
    double sum[7] = {0}, tot = 0;
    size_t n = 0;
    for (size_t w = 0; w < (size_t)n_wg * 4; ++w) {
        const long long *t = &tr[w * 8];
        if (!t[0] || !t[7]) continue;
        for (int i = 0; i < 7; ++i) sum[i] += (double)(t[i + 1] - t[i]);
        tot += (double)(t[7] - t[0]);
        ++n;
    }
    printf("   per-wave shader clocks (mean over %zu waves): total %.0f\n", n, tot / n);
    for (int i = 0; i < 7; ++i) printf("      %-22s %8.0f  %5.1f %%\n", ph[i], sum[i] / n, 100.0 * sum[i] / tot);
    CK(hipFree(qkv)); CK(hipFree(pos)); CK(hipFree(bu)); CK(hipFree(bv)); CK(hipFree(ctx)); CK(hipFree(dtr));
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    run("tdt-ctc-110m 64 x 10 s", 64, 126, 512, 8, reps, s);
    run("tdt-600m 32 x 30 s", 32, 376, 1024, 8, reps, s);
    return 0;
}
