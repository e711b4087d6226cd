// tools/ubench/tr16_probe.cpp -- what ds_read_b64_tr_b16 (gfx950 LDS transpose read) returns for per-lane addresses.
// Hypothesis H1 (used by kernels/attention_bf16.hip for the V^T operand): within each group of 16 lanes, lane i supplies the address of 4
// contiguous 16-bit elements = row (i >> 2), columns 4 (i & 3) .. +3 of a [4][16] block; lane i receives column i of that block (rows 0..3).
// Prints PASS / the observed mapping.  build: hipcc --offload-arch=gfx950 -O2 tr16_probe.cpp -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int VP = 136;   // row pitch (elements) of the row-major tile
__global__ void probe(const unsigned short *in, unsigned short *out_lin, unsigned short *out_h1) {
    __shared__ unsigned short sm[32 * VP];
    for (int i = threadIdx.x; i < 32 * VP; i += 64) sm[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(sm + 4 * l));
    for (int j = 0; j < 4; ++j) out_lin[l * 4 + j] = (unsigned short)a[j];
    // H1 addressing of a V tile: group q = l >> 4 -> dv block 16 (q & 1), key quad 4 (q >> 1); K0 = 8, D0 = 32
    const int i = l & 15, q = l >> 4;
    const int K0 = 8 + 4 * (q >> 1), D0 = 32 + 16 * (q & 1);
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(sm + (K0 + (i >> 2)) * VP + D0 + 4 * (i & 3)));
    for (int j = 0; j < 4; ++j) out_h1[l * 4 + j] = (unsigned short)b[j];
}
int main() {
    std::vector<unsigned short> h(32 * VP);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)i;
    unsigned short *din, *d1, *d2;
    hipMalloc(&din, h.size() * 2); hipMalloc(&d1, 512); hipMalloc(&d2, 512);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, d1, d2);
    unsigned short o1[256], o2[256];
    hipMemcpy(o1, d1, 512, hipMemcpyDeviceToHost); hipMemcpy(o2, d2, 512, hipMemcpyDeviceToHost);
    int bad_lin = 0, bad_h1 = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            if (o1[l * 4 + j] != (l & 15) + j * 16 + (l >> 4) * 64) ++bad_lin;            // the guide's lane-linear formula
            const int i = l & 15, q = l >> 4, K0 = 8 + 4 * (q >> 1), D0 = 32 + 16 * (q & 1);
            if (o2[l * 4 + j] != (K0 + j) * VP + D0 + i) ++bad_h1;
        }
    printf("lane-linear formula: %s (%d bad)   H1 per-lane addressing: %s (%d bad)\n", bad_lin ? "FAIL" : "PASS", bad_lin, bad_h1 ? "FAIL" : "PASS", bad_h1);
    if (bad_lin || bad_h1) {
        for (int l = 0; l < 64; ++l) printf("lane %2d: lin %4d %4d %4d %4d | h1 %4d %4d %4d %4d\n", l, o1[l*4], o1[l*4+1], o1[l*4+2], o1[l*4+3], o2[l*4], o2[l*4+1], o2[l*4+2], o2[l*4+3]);
    }
    return (bad_lin || bad_h1) ? 1 : 0;
}
