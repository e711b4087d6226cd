// tools/ubench/gemm_bf16_trace.cpp -- WHERE the time of the persistent bf16 GEMM goes (round 5): fc1 of tdt-600m (12032 x 4096 x 1024, SiLU, bf16 out)
// with the phase stamps of kernels/gemm_bf16_glds.hpp (-DGL_TRACE): per workgroup and output tile the shader clock at tile start, at the first
// readable fragments, after every K tile's barrier, at the end of the K loop and after the epilogue's stores were issued.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGL_TRACE -I parakeet.cpp_amd/csrc tools/ubench/gemm_bf16_trace.cpp -o /tmp/gemm_bf16_trace && /tmp/gemm_bf16_trace
// Prints, in shader clocks and microseconds at the measured clock: the mean / p10 / p90 over workgroups of every phase, tile by tile, and the
// K-tile durations by position (first tiles of an output tile vs the rest).  Operands random bf16 in [-1, 1) (zero operands clock higher).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels/gemm.hip"
#include "kernels/gemm_smallm.hip"
#include "kernels/gemm_smallm_bf16.hip"

using namespace pk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 1024, epi = argc > 3 ? atoi(argv[3]) : EPI_SILU;
    const int blocked = argc > 4 ? atoi(argv[4]) : 0;                       // GemmArgs::out_blocked: the 32 x 16 block layout of the fc1 -> fc2 hand-off
    const int M = 12032;
    std::vector<unsigned short> h((size_t)M * K > (size_t)N * K ? (size_t)M * K : (size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto &v : h) v = rnd();
    void *dA, *dW; float *dB, *dO;
    CK(hipMalloc(&dA, (size_t)M * K * 2)); CK(hipMalloc(&dW, (size_t)N * K * 2)); CK(hipMalloc(&dB, N * 4)); CK(hipMalloc(&dO, (size_t)M * N * 4));
    CK(hipMemcpy(dA, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dB, 0, N * 4));
    const size_t words = (size_t)256 * GL_TRACE_TILES * GL_TRACE_SLOTS;
    unsigned long long *dT;
    CK(hipMalloc(&dT, words * 8));
    GemmArgs g{reinterpret_cast<const float *>(dA), K, reinterpret_cast<const float *>(dW), K, dB, dO, N, nullptr, 0, 1.0f, M, N, K};
    g.a_bf16 = 1; g.out_bf16 = 1; g.fast_act = 1; g.out_blocked = blocked;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long *null_t = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(gl_trace), &null_t, sizeof(null_t)));
    for (int r = 0; r < 5; ++r) launch_gemm_bf16(g, epi, 0);            // warm, untraced
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 20; ++r) launch_gemm_bf16(g, epi, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_launch = ms * 1e3 / 20;
    CK(hipMemset(dT, 0, words * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(gl_trace), &dT, sizeof(dT)));
    launch_gemm_bf16(g, epi, 0);                                           // ONE traced launch (stamps cost ~1 %)
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> t(words);
    CK(hipMemcpy(t.data(), dT, words * 8, hipMemcpyDeviceToHost));
    const int nk = K / 64;
    auto stat = [](std::vector<double> v, const char *name, double mhz) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        double sum = 0;
        for (double x : v) sum += x;
        printf("  %-44s mean %8.0f clk = %6.2f us   p10 %8.0f  p90 %8.0f   (n = %zu)\n", name, sum / v.size(), sum / v.size() / mhz, v[v.size() / 10], v[v.size() * 9 / 10], v.size());
    };
    // clock: whole-kernel span in shader clocks (first stamp of any workgroup .. last stamp) against the event-timed launch
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto x : t) if (x) { tmin = std::min(tmin, x); tmax = std::max(tmax, x); }
    const double mhz = (double)(tmax - tmin) / us_launch;                  // stamp units per microsecond (approximate: the traced launch vs the timed average)
    printf("%d x %d x %d, epi %d, out_blocked %d: %.1f us per launch (20 untraced); traced span %.0f ticks -> %.0f ticks per us\n", M, N, K, epi, blocked, us_launch, (double)(tmax - tmin), mhz);
    for (int tile = 0; tile < GL_TRACE_TILES; ++tile) {
        std::vector<double> pro, loop, epi_d, total, start_off;
        std::vector<std::vector<double>> kt(nk);
        for (int wg = 0; wg < 256; ++wg) {
            const unsigned long long *r = &t[((size_t)wg * GL_TRACE_TILES + tile) * GL_TRACE_SLOTS];
            if (!r[0] || !r[3 + nk]) continue;
            start_off.push_back((double)(r[0] - tmin));
            pro.push_back((double)(r[1] - r[0]));
            for (int k = 0; k < nk; ++k) kt[k].push_back((double)(r[2 + k] - (k ? r[1 + k] : r[1])));
            loop.push_back((double)(r[2 + nk] - r[1]));
            epi_d.push_back((double)(r[3 + nk] - r[2 + nk]));
            total.push_back((double)(r[3 + nk] - r[0]));
        }
        if (total.empty()) continue;
        printf("output tile %d of a workgroup:\n", tile);
        stat(start_off, "start after the first stamp of the launch", mhz);
        stat(pro, "tile start -> first fragments readable", mhz);
        for (int k = 0; k < nk; ++k) { char nm[64]; snprintf(nm, sizeof nm, "K tile %2d (barrier to barrier)", k); stat(kt[k], nm, mhz); }
        stat(loop, "K loop (first fragments -> last MFMA issued)", mhz);
        stat(epi_d, "epilogue (VALU + stores issued)", mhz);
        stat(total, "whole tile", mhz);
    }
    return 0;
}
