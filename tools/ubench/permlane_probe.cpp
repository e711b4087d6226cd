// tools/ubench/permlane_probe.cpp -- pins the semantics of gfx950's v_permlane16_swap_b32 / v_permlane32_swap_b32 as
// kernels/gemm_smallm.hip uses them (a 4 x 4 transpose between the four 16-lane rows of a wave and four registers).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/permlane_probe.cpp -o /tmp/permlane_probe && /tmp/permlane_probe
// Every lane tags its four registers with (row << 4 | component); after the four swaps component e of row q must carry the tag
// (e << 4 | q).  Prints the table and exits non-zero on a mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned *out) {
    const unsigned lane = threadIdx.x, row = lane >> 4;
    unsigned c0 = row << 4 | 0, c1 = row << 4 | 1, c2 = row << 4 | 2, c3 = row << 4 | 3;
    auto s01 = __builtin_amdgcn_permlane16_swap(c0, c1, false, false);
    auto s23 = __builtin_amdgcn_permlane16_swap(c2, c3, false, false);
    auto t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
    auto t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
    out[4 * lane + 0] = t02[0];
    out[4 * lane + 1] = t13[0];
    out[4 * lane + 2] = t02[1];
    out[4 * lane + 3] = t13[1];
}

int main() {
    unsigned *d = nullptr, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int q = 0; q < 4; ++q) {
        printf("row %d:", q);
        for (int e = 0; e < 4; ++e) {
            const unsigned t = h[4 * (16 * q + 5) + e];
            printf("  comp %d <- (row %u, comp %u)", e, t >> 4, t & 15);
            for (int l = 0; l < 16; ++l) bad += h[4 * (16 * q + l) + e] != (unsigned)(e << 4 | q);
        }
        printf("\n");
    }
    printf(bad ? "MISMATCH (%d)\n" : "transpose OK\n", bad);
    return bad ? 1 : 0;
}
