// tools/ubench/cu_ingest.cpp -- how fast does ONE compute unit pull bytes through its vector L1, as a function of the ACCESS PATTERN of the loads?
// (round 5: the small-M bf16 products of a streaming chunk cost ~2 us + 53 ns per KB their busiest CU pulls, whatever the kernel does with
// the bytes -- kernels/gemm_smallm_bf16.hip.  Is that the fabric, or the pattern -- 16 rows x 64 B per load instruction?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/cu_ingest.cpp -o tools/ubench/cu_ingest
// One workgroup of 8 waves per CU (grid = 256 x `over`), every wave issues NL 16-byte loads per lane UP FRONT, xors them, stores one word.
//   pattern rows:   lane (r = lane & 15, q = lane >> 4) reads row r, bytes 64 s + 16 q: 16 segments of 64 B per instruction, `rs` bytes apart
//   pattern lines:  lane (r = lane & 7,  q = lane >> 3) reads row q, bytes 128 s + 16 r: 8 whole 128-B lines per instruction, `rs` bytes apart
//   pattern contig: lane reads bytes 1024 s + 16 lane: 1 KB contiguous per instruction
// `share` workgroups (on the same XCD) read the same bytes (the row tiles of one column tile); cold = rotating through a 640 MB pool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PAT, int NL>
__global__ __launch_bounds__(512) void ingest(const uint4 *base, size_t tile_bytes, int ntiles, int rs, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x % ntiles;
    const char *p = reinterpret_cast<const char *>(base) + (size_t)tile * tile_bytes;
    uint4 v[NL];
    if constexpr (PAT == 0) {
        const char *q = p + (size_t)(lane & 15) * rs + (lane >> 4) * 16 + (size_t)wave * NL * 64;
#pragma unroll
        for (int s = 0; s < NL; ++s) v[s] = *reinterpret_cast<const uint4 *>(q + 64 * s);
    } else if constexpr (PAT == 1) {
        const char *q = p + (size_t)(lane >> 3) * rs + (lane & 7) * 16 + (size_t)wave * NL * 128;
#pragma unroll
        for (int s = 0; s < NL; ++s) v[s] = *reinterpret_cast<const uint4 *>(q + 128 * s);
    } else {
        const char *q = p + ((size_t)wave * NL) * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < NL; ++s) v[s] = *reinterpret_cast<const uint4 *>(q + 1024 * s);
    }
    unsigned x = 0;
#pragma unroll
    for (int s = 0; s < NL; ++s) x ^= v[s].x ^ v[s].y ^ v[s].z ^ v[s].w;
    if (x == 0x12345u) out[blockIdx.x * 512 + threadIdx.x] = x;
}

int main() {
    const size_t pool = (size_t)640 << 20;
    char *d; unsigned *o;
    CK(hipMalloc(&d, pool)); CK(hipMalloc(&o, 4096 * 512 * 4));
    CK(hipMemset(d, 1, pool));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-8s %3s %6s %5s %5s %6s | %8s %10s %9s %10s\n", "pattern", "NL", "rs", "share", "over", "cold", "us", "KB per CU", "B/clk/CU", "TB/s chip");
    auto run = [&](int pat, int nl, int rs, int share, int over, bool cold) {
        const int nwg = 256 * over, ntiles = nwg / share;
        // a tile = the bytes one workgroup reads: 8 waves x NL KB; rows pattern: 16 rows x rs (rs >= 8 NL 64), lines: 8 rows x rs (rs >= 8 NL 128)
        const size_t tile_bytes = pat == 0 ? (size_t)16 * rs : pat == 1 ? (size_t)8 * rs : (size_t)8 * nl * 1024;
        const size_t launch_bytes = tile_bytes * ntiles;
        const int nrot = cold ? (int)(pool / launch_bytes) : 1;
        if (nrot < 1) return;
        auto go = [&](int i) {
            const uint4 *b = reinterpret_cast<const uint4 *>(d + (size_t)(i % nrot) * launch_bytes);
#define GO(P, N) hipLaunchKernelGGL((ingest<P, N>), dim3(nwg), dim3(512), 0, 0, b, tile_bytes, ntiles, rs, o)
            if (pat == 0) { if (nl == 16) GO(0, 16); else GO(0, 32); }
            else if (pat == 1) { if (nl == 16) GO(1, 16); else GO(1, 32); }
            else { if (nl == 16) GO(2, 16); else GO(2, 32); }
        };
        const int reps = 60;
        for (int i = 0; i < 10; ++i) go(i);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) go(10 + i);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, kb = 8.0 * nl * over;            // KB one CU pulls (over workgroups of 8 NL KB)
        printf("%-8s %3d %6d %5d %5d %6s | %8.2f %10.0f %9.2f %10.2f\n", pat == 0 ? "rows" : pat == 1 ? "lines" : "contig", nl, rs, share, over, cold ? "cold" : "warm",
               us, kb, kb * 1024 / (us * 2400.0), (double)launch_bytes / (us * 1e-6) * 1e-12);
    };
    for (int cold = 1; cold >= 0; --cold)
        for (int share : {1, 4})
            for (int nl : {16, 32}) {
                run(0, nl, 8 * nl * 64, share, 1, cold);        // rows, row stride = exactly the tile's K extent (fc2: 8 KB rows when NL = 16)
                run(0, nl, 8 * nl * 64 * 4, share, 1, cold);    // rows, strided 4x further apart
                run(1, nl, 8 * nl * 128, share, 1, cold);
                run(2, nl, 0, share, 1, cold);
            }
    for (int over : {2, 4}) { run(0, 16, 8192, 1, over, true); run(2, 16, 0, 1, over, true); }
    // launch floor: the same kernel reading one line
    return 0;
}
