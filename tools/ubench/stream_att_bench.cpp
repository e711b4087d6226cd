// tools/ubench/stream_att_bench.cpp -- the cached attention of a streaming chunk (kernels/stream.hip, compiled here with -DSA_TRACE) at the
// shape of BASELINE configs[4] (16 streams x 8 heads of 128, a 70-row cache + 2 new frames): time per launch with the caches hot in L2 and
// after a 1 GB sweep between launches (what a chunk's weight stream leaves of them), and the per-phase shader-clock breakdown of its waves.
// Random inputs; results are not checked here (tests/test_gpu_stream.py does that bit for bit).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define SA_TRACE 1
#include "../../parakeet.cpp_amd/csrc/kernels/stream.hip"

using namespace pk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main() {
    const int S = 16, H = 8, d = 1024, c = 2, left = 70, nc = 70, Tp = left + c, P = 2 * Tp - 1, hd = d / H;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<float> h;
    unsigned x = 4242u;
    auto mk = [&](size_t n, float sc) {
        h.resize(n);
        for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (sc / (1 << 23)); }
        float *p;
        CK(hipMalloc(&p, n * 4));
        CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
        return p;
    };
    float *qkv = mk((size_t)S * c * 3 * d, 1.0f), *kc = mk((size_t)S * left * d, 1.0f), *vc = mk((size_t)S * left * d, 1.0f);
    float *ko = mk((size_t)S * left * d, 1.0f), *vo = mk((size_t)S * left * d, 1.0f);
    float *pos = mk((size_t)P * d, 1.0f), *bu = mk(d, 0.1f), *bv = mk(d, 0.1f), *ctx = mk((size_t)S * c * d, 0.0f);
    char *sweep;
    const size_t sweep_bytes = (size_t)1 << 30;
    CK(hipMalloc(&sweep, sweep_bytes));
    auto launch = [&]() { launch_stream_attention(qkv, kc, vc, left, S, c, nc, d, H, pos, P, bu, bv, left, 1, ctx, s, ko, vo, left, 0); };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    const int reps = 50;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream attention S=%d H=%d hd=%d kv=%d, back to back (caches hot): %.2f us per launch\n", S, H, hd, nc + c, ms / reps * 1e3);
    float cold = 0.0f;
    for (int i = 0; i < 10; ++i) {
        CK(hipMemsetAsync(sweep, i, sweep_bytes, s));
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        cold += ms;
    }
    printf("after a 1 GB sweep (caches cold): %.2f us per launch (event pair around ONE launch: includes ~2-3 us of event overhead)\n", cold / 10 * 1e3);
    const int n_wg = S * H * (c + 1);
    long long *dtr;
    CK(hipMalloc(&dtr, (size_t)n_wg * 2 * 8 * 8));
    for (int pass = 0; pass < 2; ++pass) {
        CK(hipMemset(dtr, 0, (size_t)n_wg * 2 * 8 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sa_trace), &dtr, 8));
        if (pass == 1) CK(hipMemsetAsync(sweep, 7, sweep_bytes, s));
        launch();
        CK(hipStreamSynchronize(s));
        long long *null = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sa_trace), &null, 8));
        std::vector<long long> tr((size_t)n_wg * 2 * 8);
        CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
        const char *ph[5] = {"q + biases -> LDS, barrier", "score chains (key / position rows)", "softmax (3 barriers)", "value rows + chains", "store"};
        double sum[5] = {0}, tot = 0;
        size_t n = 0;
        for (size_t w = 0; w < (size_t)S * H * c * 2; ++w) {          // attention workgroups only (blockIdx.y < c), their first two waves
            const long long *t = &tr[w * 8];
            if (!t[0] || !t[5]) continue;
            for (int i = 0; i < 5; ++i) sum[i] += (double)(t[i + 1] - t[i]);
            tot += (double)(t[5] - t[0]);
            ++n;
        }
        printf("%s: %zu waves, mean wave lifetime %.0f shader clocks\n", pass ? "cold" : "hot", n, tot / n);
        for (int i = 0; i < 5; ++i) printf("   %-40s %9.0f  (%4.1f %%)\n", ph[i], sum[i] / n, 100.0 * sum[i] / tot);
    }
    return 0;
}
