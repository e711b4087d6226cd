// tools/ubench/addtid_probe.cpp -- semantics of ds_write_addtid_b32 on gfx950 (LDS address = M0[15:0] + offset + 4 * lane, no address VGPR):
// every lane of two wavefronts stores its tag through the instruction, the LDS image is copied out and checked.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/addtid_probe.cpp -o /tmp/addtid_probe && /tmp/addtid_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned *out) {
    __shared__ unsigned lds[1024];
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    for (unsigned i = tid; i < 1024; i += blockDim.x) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + wave * 512u * 4u);   // LDS byte offset of this wave's 512-word region (an SGPR)
    const unsigned v0 = 0x100u + tid, v1 = 0x200u + tid;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 :: "v"(v0), "v"(v1), "s"(base) : "memory", "m0");
    __syncthreads();
    for (unsigned i = tid; i < 1024; i += blockDim.x) out[i] = lds[i];
}

int main() {
    unsigned *d = nullptr, h[1024];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int w = 0; w < 2; ++w)
        for (int l = 0; l < 64; ++l) {
            bad += h[w * 512 + l] != 0x100u + 64 * w + l;
            bad += h[w * 512 + 256 + l] != 0x200u + 64 * w + l;
        }
    printf("wave 0: word 0 = %#x, word 63 = %#x, word 64 = %#x, word 256 = %#x ; wave 1: word 512 = %#x\n", h[0], h[63], h[64], h[256], h[512]);
    printf(bad ? "MISMATCH (%d)\n" : "ds_write_addtid_b32: address = M0 + offset + 4 * lane  OK\n", bad);
    return bad ? 1 : 0;
}
