// tools/ubench/smallm_bf16_trace.cpp -- WHERE the ~10 us of a streaming chunk's small-M bf16 product go (round 5): the nemotron-600m shapes (M = 32:
// 16 streams x 2 frames) through kernels/gemm_smallm_bf16.hip with the phase stamps of -DSB_TRACE, rotating through a pool of weight matrices larger
// than the Infinity Cache (so that every launch streams its weights from HBM as a chunk's 200 products do).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSB_TRACE -I parakeet.cpp_amd/csrc tools/ubench/smallm_bf16_trace.cpp -o tools/ubench/smallm_bf16_trace
// Per product: event-timed microseconds per launch (cold weights), then per phase the mean / p10 / p90 over all waves of one traced launch, in shader
// clocks, and the span from the first wave's entry to the last wave's exit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels/gemm_smallm_bf16.hip"

using namespace pk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Prod { const char *name; int N, K, epi, a16, ln; };

int main() {
    const int M = 32, POOL = 40;                                            // 40 x 8.4 MB = 336 MB of weights per product shape
    const Prod prods[] = {{"fc1  N 4096 K 1024 LN silu (fp32 rows in)", 4096, 1024, EPI_SILU, 0, 1}, {"fc2  N 1024 K 4096 resid (bf16 rows in)", 1024, 4096, EPI_RESID, 1, 0},
                          {"qkv  N 3072 K 1024 LN none", 3072, 1024, EPI_NONE, 0, 1}, {"out  N 1024 K 1024 resid (fp32 rows in)", 1024, 1024, EPI_RESID, 0, 0},
                          {"pw1  N 1024 K 1024 LN glu", 1024, 1024, EPI_GLU, 0, 1}};
    const size_t wmax = (size_t)4096 * 1024 * 2;                            // elements of the largest weight matrix (GLU: 2048 x 1024)
    std::vector<unsigned short> hw(wmax);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto &v : hw) v = rnd();
    std::vector<float> hx((size_t)M * 4096);
    for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; }
    unsigned short *dW; float *dX, *dB, *dO, *dR, *dG;
    CK(hipMalloc(&dW, wmax * 2 * POOL)); CK(hipMalloc(&dX, hx.size() * 4)); CK(hipMalloc(&dB, 8192 * 4)); CK(hipMalloc(&dO, (size_t)M * 8192 * 4));
    CK(hipMalloc(&dR, (size_t)M * 4096 * 4)); CK(hipMalloc(&dG, 4096 * 4 * 2));
    for (int p = 0; p < POOL; ++p) CK(hipMemcpy(dW + (size_t)p * wmax, hw.data(), wmax * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dB, 0, 8192 * 4)); CK(hipMemset(dR, 0, (size_t)M * 4096 * 4));
    { std::vector<float> g1(8192, 1.0f); CK(hipMemcpy(dG, g1.data(), 8192 * 4, hipMemcpyHostToDevice)); }
    unsigned long long *dT;
    const size_t words = (size_t)1024 * kSbMaxWaves * 8;
    CK(hipMalloc(&dT, words * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Prod &c : prods) {
        GemmArgs g{dX, c.K, reinterpret_cast<const float *>(dW), c.K, dB, dO, c.N, dR, c.N, 0.5f, M, c.N, c.K};
        g.a_bf16 = c.a16; g.fast_act = 1;
        if (c.ln) { g.ln_g = dG; g.ln_b = dG + 4096; g.ln_eps = 1e-5f; }
        unsigned long long *null_t = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sb_trace), &null_t, sizeof(null_t)));
        auto launch = [&](int p) { GemmArgs q = g; q.W = reinterpret_cast<const float *>(dW + (size_t)p * wmax); launch_gemm_smallm_bf16(q, c.epi, 0); };
        for (int r = 0; r < POOL; ++r) launch(r);
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 2 * POOL; ++r) launch(r % POOL);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemset(dT, 0, words * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sb_trace), &dT, sizeof(dT)));
        launch(7);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(words);
        CK(hipMemcpy(t.data(), dT, words * 8, hipMemcpyDeviceToHost));
        const double wbytes = (double)c.N * c.K * 2 * (c.epi == EPI_GLU ? 2 : 1);
        printf("%s: %.2f us per launch back to back with cold weights (%.2f TB/s of weights)\n", c.name, ms * 1e3 / (2 * POOL), wbytes / (ms * 1e-3 / (2 * POOL)) * 1e-12);
        const char *names[6] = {"entry -> all loads issued", "-> rows arrived (first use)", "-> LayerNorm / conversion done", "-> weights arrived, MFMAs issued",
                                "-> partial sums exchanged", "-> epilogue stores issued"};
        std::vector<double> d[6], whole;
        for (size_t w = 0; w < words / 8; ++w) {
            const unsigned long long *r = &t[w * 8];
            if (!r[0]) continue;
            unsigned long long prev = r[0];
            for (int i = 0; i < 6; ++i) {
                if (!r[i + 1]) continue;
                d[i].push_back((double)(r[i + 1] - prev));
                prev = r[i + 1];
            }
            whole.push_back((double)(prev - r[0]));
        }
        for (int i = 0; i < 6; ++i) {
            if (d[i].empty()) continue;
            std::sort(d[i].begin(), d[i].end());
            double sum = 0;
            for (double x : d[i]) sum += x;
            printf("    %-36s mean %7.0f clk   p10 %7.0f  p90 %7.0f  (n = %zu waves)\n", names[i], sum / d[i].size(), d[i][d[i].size() / 10], d[i][d[i].size() * 9 / 10], d[i].size());
        }
        std::sort(whole.begin(), whole.end());
        double sw = 0;
        for (double x : whole) sw += x;
        printf("    %-36s mean %7.0f clk   p10 %7.0f  p90 %7.0f  max %7.0f\n", "wave entry -> its last stamp", sw / whole.size(), whole[whole.size() / 10], whole[whole.size() * 9 / 10], whole.back());
    }
    return 0;
}
