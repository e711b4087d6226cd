// parakeet.cpp_amd/csrc/kernels/kernels.hpp -- host-side launchers of the gfx950 kernels.
// Every launcher enqueues on `s` and returns; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pk {

// ---- mel front end (src/audio.cpp:100-158) ----------------------------------------------------
struct MelTables {
    const float *window;   // [512] symmetric Hann(400) zero-padded to n_fft (centred: offset 56)
    const float *tw_re;    // [256] cos(2 pi k / 512)
    const float *tw_im;    // [256] -sin(2 pi k / 512)
    const float *fb;       // [257][n_mels] Slaney filterbank (src/audio.cpp:40-94)
    const int *f_lo;       // [n_mels] first / last non-zero fft bin of each filter
    const int *f_hi;
    int n_mels;
    int power_via_abs;     // switch A2
};
void launch_mel_logmel(const float *pcm, int B, int64_t n_samples, int n_frames, const MelTables &t, float *logmel, hipStream_t s);
void launch_mel_normalize(const float *logmel, int B, int n_mels, int n_frames, int normalize, float *feats, hipStream_t s);

// ---- fp32 MFMA GEMM: out = epi(A[M][K] * W[N][K]^T + bias), natural-k fma chains ----------------
enum GemmEpi { EPI_NONE = 0, EPI_RELU = 1, EPI_SILU = 2, EPI_RESID = 3, EPI_GLU = 4 };
struct GemmArgs {
    const float *A; int64_t lda;
    const float *W; int64_t ldw;
    const float *bias;          // [N] (GLU: [2N]) or nullptr
    float *out; int64_t ldo;
    const float *resid; int64_t ldr; float alpha;   // EPI_RESID: out = resid + alpha * (acc + bias)
    int M, N, K;                // N = output width (GLU: W has 2N rows, a-part rows [0,N), gate rows [N,2N))
};
void launch_gemm(const GemmArgs &a, int epi, hipStream_t s);
double gemm_flops(const GemmArgs &a, int epi);

// ---- LayerNorm, canonical reductions, math diagnostics ------------------------------------------
void launch_layernorm(const float *x, int64_t rows, int d, const float *g, const float *b, float eps, float *y, hipStream_t s);
void launch_sum64_rows(const float *x, int rows, int n, float *out, hipStream_t s);
void launch_math(int fn, const float *in, float *out, int64_t n, hipStream_t s);

}  // namespace pk
