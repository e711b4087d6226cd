// tools/ubench/write_bw.cpp -- HBM write / read rates of plain streaming kernels (calibrates the GEMM epilogue's write burst).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void wr_linear(float4 *p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// tile pattern of the GEMM epilogue: block -> 128 rows x 128 cols (512 B segments, row stride ld floats)
__global__ void wr_tiles(float *p, int ld, int tiles_n) {
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    for (int c = threadIdx.x; c < 128 * 32; c += blockDim.x) {
        const int r = c >> 5, c4 = c & 31;
        *reinterpret_cast<float4 *>(p + (size_t)(tm * 128 + r) * ld + tn * 128 + c4 * 4) = make_float4(1.f, 2.f, 3.f, (float)c);
    }
}
// bf16 GEMM epilogue patterns for a 256 x 256 output tile, 512 threads (8 waves as 4 x 2, wave tile 64 x 128), ld in bf16 elements:
//   ROWS: through-LDS form -- thread c of a pass writes 8 bytes (4 bf16) of row c / 64, chunk c % 64: a wave covers one 512-byte row segment
//   ACC:  straight from transposed 32 x 32 accumulators -- lane (n, g) holds 16 values of row n: four 8-byte stores per tile at columns 8 q + 4 g
__global__ void wr_bf16_rows(unsigned short *p, int ld, int tiles_n) {
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    for (int c = threadIdx.x; c < 256 * 64; c += blockDim.x) {
        const int r = c >> 6, ch = c & 63;
        *reinterpret_cast<uint2 *>(p + (size_t)(tm * 256 + r) * ld + tn * 256 + ch * 4) = make_uint2(0x3f803f80u, (unsigned)c);
    }
}
__global__ void wr_bf16_acc(unsigned short *p, int ld, int tiles_n) {
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, g = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                        // 4 x 2 waves, 64 x 128 each
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 4; ++q) {
                const int row = tm * 256 + wm * 64 + i * 32 + n, col = tn * 256 + wn * 128 + j * 32 + 8 * q + 4 * g;
                *reinterpret_cast<uint2 *>(p + (size_t)row * ld + col) = make_uint2(0x3f803f80u, (unsigned)(q + lane));
            }
}
__global__ void rd_linear(const float4 *p, size_t n4, float *out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.w; }
    if (s == 1234.5f) out[0] = s;
}
int main() {
    const size_t bytes = 8064ull * 2048 * 4;
    float *d, *o;
    CK(hipMalloc(&d, bytes * 4)); CK(hipMalloc(&o, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto f, const char *name, double b) {
        for (int i = 0; i < 3; ++i) f();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, b / ms * 1e-9);
    };
    time([&] { hipLaunchKernelGGL(wr_linear, dim3(2048), dim3(256), 0, 0, (float4 *)d, bytes / 16); }, "write 66 MB linear float4 (2048 blocks)", bytes);
    time([&] { hipLaunchKernelGGL(wr_linear, dim3(512), dim3(512), 0, 0, (float4 *)d, bytes / 16); }, "write 66 MB linear float4 (512x512)", bytes);
    time([&] { hipLaunchKernelGGL(wr_tiles, dim3(63 * 16), dim3(512), 0, 0, d, 2048, 16); }, "write 66 MB as 128x128 tiles (GEMM pattern)", bytes);
    time([&] { hipLaunchKernelGGL(wr_linear, dim3(2048), dim3(256), 0, 0, (float4 *)d, bytes / 4); }, "write 264 MB linear float4", bytes * 4);
    time([&] { hipLaunchKernelGGL(rd_linear, dim3(2048), dim3(256), 0, 0, (const float4 *)d, bytes / 4, o); }, "read 264 MB linear float4", bytes * 4);
    time([&] { hipMemsetAsync(d, 0, bytes, 0); }, "hipMemsetAsync 66 MB", bytes);
    const double b16 = 12032.0 * 4096 * 2;                           // fc1 of tdt-600m, bf16 out: 47 x 16 tiles of 256 x 256
    time([&] { hipLaunchKernelGGL(wr_bf16_rows, dim3(47 * 16), dim3(512), 0, 0, (unsigned short *)d, 4096, 16); }, "bf16 98.6 MB, 256x256 tiles, row segments (LDS form)", b16);
    time([&] { hipLaunchKernelGGL(wr_bf16_acc, dim3(47 * 16), dim3(512), 0, 0, (unsigned short *)d, 4096, 16); }, "bf16 98.6 MB, 256x256 tiles, accumulator layout", b16);
    return 0;
}
