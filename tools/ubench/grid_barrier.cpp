// tools/ubench/grid_barrier.cpp -- cost of a grid barrier among G persistent workgroups on MI355X, for the single-launch decode loop
// (parakeet.cpp_amd/csrc/kernels/decode_persist.hip).  Variants:
//   0  flat: ONE arrival counter, system-scope relaxed atomics (what round 2 / 3 shipped: ~20 us at 160 workgroups)
//   1  XCD-hierarchical: per-XCC arrival counter -> the XCC's last arriver adds to a top counter and waits for all XCCs -> bumps the XCC's
//      generation word, which the other workgroups of the XCC poll.  Per-XCC words are touched with agent-scope atomics (RMW at the memory
//      side, but 8 independent words instead of one), the top counter sees 8 arrivals instead of G.
//   2  the same with the per-XCC words at WORKGROUP scope (sc0: resolved in the XCC's own L2 -- every CU of an XCC shares that L2; polls are
//      atomic RMWs so the per-CU vector L1 is never consulted)
// Data exchanged across a barrier goes through system-scope stores / loads (as in the decode kernel), so no cache-wide fence is part of the
// barrier.  Every barrier is VERIFIED: before barrier k workgroup w stores (k << 12 | w) to slot w, after it reads slot (w + 1) % G.
// usage: grid_barrier [G=160] [iters=2000] [work_ns=0]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Bar {
    unsigned *flat;        // [1]
    unsigned *xcnt;        // [8 * 32] per-XCC arrival counters, 128 bytes apart
    unsigned *xgen;        // [8 * 32] per-XCC generation words
    unsigned *top;         // [1]
    unsigned *census;      // [8 * 32] workgroups per XCC (counted at kernel start)
    unsigned *slots;       // [G] exchanged data
    unsigned *errors;      // [1]
    long long *clk;        // [2] wall clock of workgroup 0 at start / end
};

// every spin is bounded: 0.5 s of the 100 MHz wall clock, then an abort flag every workgroup honours (a broken protocol must not hang the GPU)
#define SPIN_WHILE(cond)                                                                                               \
    do {                                                                                                               \
        const long long t0_ = wall_clock64();                                                                          \
        while ((cond) && !dead) {                                                                                      \
            __builtin_amdgcn_s_sleep(1);                                                                               \
            if (__hip_atomic_load(b.errors + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) dead = true;             \
            else if (wall_clock64() - t0_ > 50000000LL) { __hip_atomic_store(b.errors + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); dead = true; } \
        }                                                                                                              \
    } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]

template <int VAR>
__global__ __launch_bounds__(256) void bar_kernel(Bar b, int iters, int work_ns) {
    const int wg = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    __shared__ unsigned s_n, s_x, s_dead;
    bool dead = false;
    // ---- census: how many workgroups sit on each XCC (placement is not assumed), then one flat barrier ----
    if (tid == 0) {
        const unsigned x = xcc_id();
        s_x = x;
        __hip_atomic_fetch_add(b.census + 32 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(b.flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        SPIN_WHILE(__hip_atomic_load(b.flat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)G);
        s_n = __hip_atomic_load(b.census + 32 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    const unsigned x = s_x, n_x = s_n;
    unsigned n_xcc = 0;
    for (int i = 0; i < 8; ++i) n_xcc += __hip_atomic_load(b.census + 32 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 1u : 0u;
    unsigned phase = 0;
    unsigned bad = 0;
    if (wg == 0 && tid == 0) b.clk[0] = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (tid == 0) __hip_atomic_store(b.slots + (it & 1) * G + wg, ((unsigned)it << 12) | (unsigned)wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // double-buffered: rewritten two barriers later
        if (work_ns > 0) {                                                   // arrival skew: a little "phase work" of varying length
            const long long t0 = wall_clock64(), ticks = (long long)work_ns * (1 + (wg * 7 + it) % 3) / 30;   // 100 MHz clock: 10 ns per tick
            while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
        }
        // ---- barrier ----
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): the system-scope store above is performed
        __syncthreads();
        ++phase;
        if (tid == 0) {
            if (VAR == 0) {
                __hip_atomic_fetch_add(b.flat + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                SPIN_WHILE(__hip_atomic_load(b.flat + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < phase * (unsigned)G);
            } else {
                constexpr int SC = VAR == 2 ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
                const unsigned old = __hip_atomic_fetch_add(b.xcnt + 32 * x, 1u, __ATOMIC_RELAXED, SC);
                if (old + 1 == phase * n_x) {                                // the XCC's last arriver: up to the top, wait for every XCC, release its XCC
                    __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    SPIN_WHILE(__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < phase * n_xcc);
                    __hip_atomic_fetch_add(b.xgen + 32 * x, 1u, __ATOMIC_RELAXED, SC);
                } else {
                    SPIN_WHILE(__hip_atomic_fetch_add(b.xgen + 32 * x, 0u, __ATOMIC_RELAXED, SC) < phase);
                }
            }
            s_dead = dead ? 1u : 0u;
        }
        __syncthreads();
        if (s_dead) break;
        // ---- verify: the neighbour's store of THIS iteration is visible ----
        if (tid == 0) {
            const unsigned nb = (unsigned)((wg + 1) % G);
            const unsigned v = __hip_atomic_load(b.slots + (it & 1) * G + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v != (((unsigned)it << 12) | nb)) ++bad;
        }
        __syncthreads();
    }
    if (wg == 0 && tid == 0) b.clk[1] = wall_clock64();
    if (tid == 0 && bad) atomicAdd(b.errors, bad);
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 160, iters = argc > 2 ? atoi(argv[2]) : 2000, work_ns = argc > 3 ? atoi(argv[3]) : 0;
    unsigned *buf;
    const size_t words = 4096 + 2 * G;
    CK(hipMalloc(&buf, words * 4 + 64));
    long long *clk;
    CK(hipMalloc(&clk, 16));
    printf("grid barrier among %d workgroups of 256 threads, %d barriers, phase work ~%d ns (x1..3)\n", G, iters, work_ns);
    for (int var = 0; var < 3; ++var) {
        CK(hipMemset(buf, 0, words * 4));
        Bar b{buf, buf + 256, buf + 512, buf + 768, buf + 1024, buf + 4096, buf + 800, clk};
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (var == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(G), dim3(256), 0, 0, b, iters, work_ns);
        if (var == 1) hipLaunchKernelGGL(bar_kernel<1>, dim3(G), dim3(256), 0, 0, b, iters, work_ns);
        if (var == 2) hipLaunchKernelGGL(bar_kernel<2>, dim3(G), dim3(256), 0, 0, b, iters, work_ns);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err = 0, census[8], dead = 0;
        long long c[2];
        CK(hipMemcpy(&err, buf + 800, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&dead, buf + 801, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) CK(hipMemcpy(&census[i], buf + 1024 + 32 * i, 4, hipMemcpyDeviceToHost));
        const double us = (double)(c[1] - c[0]) * 0.01 / iters;
        const char *name[3] = {"flat system-scope counter", "XCD-hierarchical, agent-scope per-XCC words", "XCD-hierarchical, workgroup-scope (XCC-local L2) words"};
        printf("  %-58s %7.2f us per barrier (kernel %.2f ms)  verify errors %u  census", name[var], us, ms, err);
        for (int i = 0; i < 8; ++i) printf(" %u", census[i]);
        printf("%s\n", dead ? "   ** TIMED OUT: the barrier did not complete **" : "");
    }
    return 0;
}
