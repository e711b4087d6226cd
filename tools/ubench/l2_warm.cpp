// tools/ubench/l2_warm.cpp -- does a weight tile REQUESTED BY AN EARLIER LAUNCH arrive faster in the launch that consumes it?
// (round 6, verdict item 1: "take the weight fetch off the dependent chain" of a streaming chunk.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/l2_warm.cpp -o tools/ubench/l2_warm
// Consumer = the access pattern of the small-M bf16 kernel: one workgroup of 8 waves per tile, every load requested up front
// (1 KB of consecutive addresses per instruction), tile b read by workgroup b (observed placement: XCD b % 8).
// Warmer = a launch BEFORE it on the same stream (or on a second stream, --concurrent) that touches ONE dword per 128-byte line
// (64 lines = 8 KB per wave instruction): the lines land in the L2 of the XCD the toucher runs on, and in the memory-side cache.
//   match    : the toucher of tile b runs on XCD b % 8 (reads HW_REG_XCC_ID and takes the tiles of its own XCD)
//   mismatch : it takes the tiles of XCD (x + 3) % 8  -> memory-side cache only
// Reported: the consumer's own span (max end - min start over workgroups, 100 MHz wall clock) and the event time of the pair.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// tile_kb per workgroup (8 waves x NL KB), NL loads of 16 B per lane
template <int NL>
__global__ __launch_bounds__(512) void consume(const char *base, unsigned long long *stamps, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0 = wall_clock64();
    const char *p = base + (size_t)blockIdx.x * (8 * NL * 1024) + ((size_t)wave * NL) * 1024 + lane * 16;
    uint4 v[NL];
#pragma unroll
    for (int s = 0; s < NL; ++s) v[s] = *reinterpret_cast<const uint4 *>(p + 1024 * s);
    unsigned x = 0;
#pragma unroll
    for (int s = 0; s < NL; ++s) x ^= v[s].x ^ v[s].y ^ v[s].z ^ v[s].w;
    if (x == 0x12345u) out[blockIdx.x * 512 + threadIdx.x] = x;
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

// touches every 128-byte line of the tiles that belong to XCD (own + shift) % 8; `per_xcd` warmer workgroups share an XCD's tiles
__global__ __launch_bounds__(256) void warm(const char *base, int tile_bytes, int ntiles, int shift, unsigned *out, unsigned long long *stamps) {
    unsigned long long t0 = wall_clock64();
    const int x = ((int)xcc_id() + shift) & 7;
    const int per_xcd = gridDim.x / 8, sub = blockIdx.x / 8;              // (placement b % 8 is observed, the XCC id is read: the pairing is by id)
    const int lines_per_tile = tile_bytes / 128;
    const int tiles_x = (ntiles - x + 7) / 8;                             // tiles x, x + 8, ...
    const long total = (long)tiles_x * lines_per_tile;
    unsigned acc = 0;
    for (long i = (long)sub * 256 + threadIdx.x; i < total; i += (long)per_xcd * 256) {
        const long t = i / lines_per_tile, l = i % lines_per_tile;
        const char *p = base + ((size_t)(x + 8 * t)) * tile_bytes + (size_t)l * 128;
        acc ^= *reinterpret_cast<const unsigned *>(p);                       // (plain: the line stays in this XCD's L2; a non-temporal touch left match = mismatch)
    }
    if (acc == 0x12345u) out[blockIdx.x * 256 + threadIdx.x] = acc;
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && stamps) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

__global__ void spin(unsigned long long ticks) {                         // a "previous product": keeps the stream busy for ~ticks x 10 ns
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char **argv) {
    const size_t pool = (size_t)1536 << 20;
    char *d; unsigned *o; unsigned long long *st, *wst;
    CK(hipMalloc(&d, pool)); CK(hipMalloc(&o, 4096 * 512 * 4)); CK(hipMalloc(&st, 4096 * 16)); CK(hipMalloc(&wst, 4096 * 16));
    CK(hipMemset(d, 1, pool));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    hipEvent_t e0, e1, ew;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ew, hipEventDisableTiming));
    std::vector<unsigned long long> h(4096 * 2);
    printf("%-10s %5s %6s %6s | %9s %9s %9s | %9s\n", "mode", "nwg", "KB/wg", "MB", "span med", "span p90", "span min", "pair us");
    // mode 0 cold, 1 prefetched by a launch on the same stream (matching XCD), 2 the same, mismatched XCD, 3 warm = re-read of the same bytes,
    // 4 prefetched by a launch on a SECOND stream while a spin kernel holds the first (the warmer of a dependent chain), 5 the same mismatched
    auto run = [&](int mode, int nwg, int nl, int warm_wgs) {
        const int tile_bytes = 8 * nl * 1024;
        const size_t launch_bytes = (size_t)tile_bytes * nwg;
        const int nrot = (int)(pool / launch_bytes);
        const int reps = 40;
        std::vector<double> spans;
        float pair_ms = 0;
        for (int i = 0; i < reps + 5; ++i) {
            const char *b = d + (size_t)(mode == 3 ? 0 : (i % nrot)) * launch_bytes;
            CK(hipEventRecord(e0, s0));
            if (mode == 1 || mode == 2) hipLaunchKernelGGL(warm, dim3(warm_wgs), dim3(256), 0, s0, b, tile_bytes, nwg, mode == 2 ? 3 : 0, o, wst);
            if (mode == 4 || mode == 5) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, 600ull);                 // 6 us of "previous product" on the chain's stream
                hipLaunchKernelGGL(warm, dim3(warm_wgs), dim3(256), 0, s1, b, tile_bytes, nwg, mode == 5 ? 3 : 0, o, wst);
            }
            if (nl == 4) hipLaunchKernelGGL((consume<4>), dim3(nwg), dim3(512), 0, s0, b, st, o);
            else if (nl == 16) hipLaunchKernelGGL((consume<16>), dim3(nwg), dim3(512), 0, s0, b, st, o);
            else hipLaunchKernelGGL((consume<32>), dim3(nwg), dim3(512), 0, s0, b, st, o);
            CK(hipEventRecord(e1, s0));
            CK(hipDeviceSynchronize());
            if (i < 5) continue;
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); pair_ms += ms;
            CK(hipMemcpy(h.data(), st, (size_t)nwg * 16, hipMemcpyDeviceToHost));
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < nwg; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
            spans.push_back((double)(hi - lo) * 0.01);
        }
        std::sort(spans.begin(), spans.end());
        static const char *names[] = {"cold", "pre-match", "pre-mism", "warm", "side-match", "side-mism"};
        printf("%-10s %5d %6d %6.1f | %9.2f %9.2f %9.2f | %9.2f   (warmer wgs %d)\n", names[mode], nwg, 8 * nl, launch_bytes / 1048576.0, spans[spans.size() / 2],
               spans[spans.size() * 9 / 10], spans[0], pair_ms * 1e3 / reps, warm_wgs);
    };
    for (int nl : {4, 16, 32})
        for (int nwg : {64, 256}) {
            for (int mode = 0; mode < 6; ++mode) run(mode, nwg, nl, 64);
            printf("\n");
        }
    // how fast does the toucher itself run (its own span), by workgroup count: 8 MB and 32 MB
    printf("toucher alone (cold bytes), span us:\n");
    for (int wg : {8, 16, 32, 64, 128, 256})
        for (int nl : {16, 32}) {
            const int nwg = 256, tile_bytes = 8 * nl * 1024;
            const size_t launch_bytes = (size_t)tile_bytes * nwg;
            const int nrot = (int)(pool / launch_bytes);
            std::vector<double> spans;
            for (int i = 0; i < 25; ++i) {
                hipLaunchKernelGGL(warm, dim3(wg), dim3(256), 0, s0, d + (size_t)(i % nrot) * launch_bytes, tile_bytes, nwg, 0, o, wst);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), wst, (size_t)wg * 16, hipMemcpyDeviceToHost));
                unsigned long long lo = ~0ull, hi = 0;
                for (int w = 0; w < wg; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
                if (i >= 5) spans.push_back((double)(hi - lo) * 0.01);
            }
            std::sort(spans.begin(), spans.end());
            printf("  warmer wgs %4d  %5.1f MB : %7.2f us  = %6.2f TB/s\n", wg, launch_bytes / 1048576.0, spans[spans.size() / 2], launch_bytes / (spans[spans.size() / 2] * 1e-6) * 1e-12);
        }
    return 0;
}
