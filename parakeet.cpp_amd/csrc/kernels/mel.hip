// parakeet.cpp_amd/csrc/kernels/mel.hip -- mel-spectrogram front end for gfx950.
//
// Replaces preprocess_audio (reference src/audio.cpp:100-158): preemphasis -> centred, reflect-padded
// STFT (n_fft 512, hop 160, symmetric Hann 400) -> |X|^2 -> Slaney mel filterbank -> log(x + 2^-24)
// -> per-bin mean / unbiased-variance normalisation -> [B][n_frames][n_mels].
//
// HBM-bound stage (0.96 MB algorithmic traffic per 10 s clip).  Kernel 1: one wavefront per frame,
// four frames per workgroup; the 512-point FFT, the power spectrum and the filterbank dot products
// all stay in LDS / registers, only log-mel [B][n_mels][n_frames] is written.  Kernel 2: one
// wavefront per (clip, mel bin): two passes over 1001 contiguous frames (L2-resident), canonical
// sum64 reductions, normalised + transposed store.
//
// FFT-512 specification: radix-2 DIT, bit-reversed load order, butterfly
//   t = w * b with tr = fma(-wi, bi, wr*br), ti = fma(wi, br, wr*bi);  b' = a - t;  a' = a + t.
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

static constexpr int kNfft = 512;
static constexpr int kHop = 160;
static constexpr int kFramesPerBlock = 4;   // wavefronts per workgroup = frames in flight
// Offline kernel: frames each wavefront walks one after the other (frames per workgroup = 4 x this).  Measured (round 3, bench.py on MI355X):
// 1 -> 0.203 ms per 64-clip batch, 2 -> 0.234, 4 -> 0.237: the kernel is LDS-latency bound and wants the parallelism, so the store tile is
// [m][4 frames] (16-byte runs) and not the 64-byte runs 16 frames per workgroup would give.
static constexpr int kOfflineIters = 1;

// STREAM = false: preprocess_audio's framing (pre-emphasis, center=true, reflect padding), output [B][n_mels][n_frames].
// STREAM = true: StreamingAudioPreprocessor::process_chunk's framing (src/audio.cpp:222-241): the buffer is ALREADY
// pre-emphasised, center=false, frame t = samples [160 t, 160 t + 400) Hann-windowed and zero-padded on the right to the
// 512-point FFT; output [B][n_frames][n_mels] (the layout of its result, no normalisation follows).
// Every wavefront owns one frame and touches only its own LDS arrays, so the stages are ordered by wave-local fences (LDS
// operations of one wave complete in order once the counter is drained) instead of workgroup barriers: four frames of a
// workgroup no longer wait for each other ten times per FFT.
#define PK_FP(i) ((i) + ((i) >> 5))
#define PK_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <bool STREAM>
__global__ __launch_bounds__(256) void mel_logmel_kernel(const float *__restrict__ pcm, int64_t n_samples, int n_frames,
                                                         MelTables tb, float *__restrict__ logmel, MelRag rg) {
    // FFT arrays with one pad word per 32 (element i at i + (i >> 5)): the butterfly strides 2^lh of the in-place radix-2 stages would
    // otherwise put two to eight lanes on one LDS bank
    __shared__ float s_re[kFramesPerBlock][kNfft + kNfft / 32];
    __shared__ float s_im[kFramesPerBlock][kNfft + kNfft / 32];
    __shared__ float s_pw[kFramesPerBlock][260];
    // twiddles and the packed filterbank bands: read ~100 times per lane and frame -- from LDS instead of dependent L1 / L2 round trips.
    // The twiddles are laid out PER STAGE (stage lh uses w^(j << (8 - lh)), j < 2^lh, stored at 2^lh - 1 + j): the lanes of a butterfly
    // stage then read consecutive words (or broadcast) instead of a 2^(8-lh)-word stride that put 8 lanes on one bank in stages 3-6
    // (round 1: 8.5e7 LDS bank-conflict cycles per dispatch).
    __shared__ float s_twr[kNfft], s_twi[kNfft];
    __shared__ float s_fb[kMelMaxTaps];
    // Offline: the log-mel values of the workgroup's frames are collected here and leave as contiguous runs of [m][frames]; written straight
    // from the frame's wavefront they were 4-byte stores 4 KB apart (round 2 PMC: 56 MB of write traffic for a 20 MB tensor).
    constexpr int ITERS = STREAM ? 1 : kOfflineIters;
    constexpr int FPB = kFramesPerBlock * ITERS;
    __shared__ float s_out[STREAM ? 1 : 128][STREAM ? 1 : FPB + 1];
    const int b = blockIdx.y;
    const float *x = pcm + (int64_t)b * n_samples;
    float *lm_clip = logmel + (int64_t)b * tb.n_mels * n_frames;         // (offline layout: this clip's [n_mels][n_frames] block)
    if constexpr (!STREAM) {
        if (rg.pcm_off) {                                           // ragged batch: this clip's own extent (kernels.hpp: MelRag)
            const int64_t o = rg.pcm_off[b];
            x = pcm + o;
            n_samples = rg.pcm_off[b + 1] - o;
            n_frames = rg.Tm[b];
            lm_clip = logmel + (int64_t)tb.n_mels * rg.Tm_off[b];
            if ((int)blockIdx.x * FPB >= n_frames) return;          // the grid covers the longest clip (whole workgroup, before any barrier)
        }
    }
    for (int i = threadIdx.x; i < kNfft - 1; i += 256) {
        const int lh = 31 - __builtin_clz(i + 1), j = i + 1 - (1 << lh);   // i = 2^lh - 1 + j
        s_twr[i] = tb.tw_re[j << (8 - lh)];
        s_twi[i] = tb.tw_im[j << (8 - lh)];
    }
    for (int i = threadIdx.x; i < tb.fb_nnz; i += 256) s_fb[i] = tb.fbc[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *re = s_re[wave], *im = s_im[wave], *pw = s_pw[wave];
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
    const int t = blockIdx.x * FPB + it * kFramesPerBlock + wave;
    const bool live = t < n_frames;
    if (it) PK_WAVE_SYNC();                                         // the previous frame's reads of pw / re / im are done

    if (live) {
#pragma unroll
        for (int i = 0; i < kNfft / 64; ++i) {
            const int n = lane + 64 * i;
            const int r = (int)(__brev((unsigned)n) >> 23);       // 9-bit reversal
            if constexpr (STREAM) {
                re[PK_FP(r)] = n < 400 ? x[(int64_t)t * kHop + n] * tb.window_left[n] : 0.0f;
            } else {
                int64_t idx = (int64_t)t * kHop + n - kNfft / 2;  // center=true
                if (idx < 0) idx = -idx;                          // pad_mode="reflect"
                if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
                float v;
                if (idx == 0) {
                    v = x[0];                                     // preemphasis, src/audio.cpp:104-114
                } else {
                    const float p = 0.97f * x[idx - 1];
                    v = x[idx] - p;
                }
                re[PK_FP(r)] = v * tb.window[n];
            }
            im[PK_FP(r)] = 0.0f;
        }
    }
    PK_WAVE_SYNC();
#pragma unroll 1
    for (int lh = 0; lh < 9; ++lh) {
        const int h = 1 << lh;
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = lane + 64 * i;
                const int j = q & (h - 1);
                const int a = ((q >> lh) << (lh + 1)) + j;
                const int bb = a + h;
                const float cr = s_twr[h - 1 + j], ci = s_twi[h - 1 + j];
                const int pa = PK_FP(a), pb = PK_FP(bb);
                const float br = re[pb], bi = im[pb];
                const float tr = __builtin_fmaf(-ci, bi, cr * br);
                const float ti = __builtin_fmaf(ci, br, cr * bi);
                const float ar = re[pa], ai = im[pa];
                re[pb] = ar - tr;
                im[pb] = ai - ti;
                re[pa] = ar + tr;
                im[pa] = ai + ti;
            }
        }
        PK_WAVE_SYNC();
    }
    if (live) {
        for (int f = lane; f <= kNfft / 2; f += 64) {
            const float s = __builtin_fmaf(re[PK_FP(f)], re[PK_FP(f)], im[PK_FP(f)] * im[PK_FP(f)]);
            if (tb.power_via_abs) {                               // abs() then square, src/audio.cpp:123-124
                const float mag = __builtin_sqrtf(s);
                pw[f] = mag * mag;
            } else {
                pw[f] = s;
            }
        }
    }
    PK_WAVE_SYNC();
    if (live) {
        for (int m = lane; m < tb.n_mels; m += 64) {
            float acc = 0.0f;
            const int lo = tb.f_lo[m], hi = tb.f_hi[m];           // zero weights contribute fma(0, p, acc) = acc exactly
            const float *wm = s_fb + tb.fb_off[m] - lo;
            for (int f = lo; f <= hi; ++f) acc = __builtin_fmaf(wm[f], pw[f], acc);
            const float lm = dlogf(acc + 5.96046448e-8f);
            if constexpr (STREAM) logmel[((int64_t)b * n_frames + t) * tb.n_mels + m] = lm;
            else s_out[m][it * kFramesPerBlock + wave] = lm;
        }
    }
    }
    if constexpr (!STREAM) {
        __syncthreads();
        const int t0 = blockIdx.x * FPB;
        for (int idx = threadIdx.x; idx < tb.n_mels * FPB; idx += 256) {
            const int m = idx / FPB, tt = idx % FPB;
            if (t0 + tt < n_frames) lm_clip[(int64_t)m * n_frames + t0 + tt] = s_out[m][tt];
        }
    }
}

#undef PK_WAVE_SYNC
#undef PK_FP

// Per-bin mean / unbiased variance normalisation + transpose to [B][n_frames][n_mels] (src/audio.cpp:140-156).  One wavefront per
// (clip, mel bin) for the two canonical sum64 reductions over the frames; the 16 bins of a workgroup then go through a
// [64 frames][16 bins] LDS tile so that the transposed store writes 64-byte runs (the first version stored 4 bytes per 320-byte
// stride: 230 MB of write traffic for a 20 MB tensor, profiles/r01_pmc_hbm.json).
__global__ __launch_bounds__(1024) void mel_normalize_kernel(const float *__restrict__ logmel, int n_mels, int n_frames,
                                                             int normalize, float *__restrict__ feats, MelRag rg) {
    __shared__ float tile[64][17];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 16 + wave, b = blockIdx.y;
    const bool live = m < n_mels;
    int64_t frame0 = (int64_t)b * n_frames;                        // first frame of this clip in the packed frame axis
    if (rg.pcm_off) { frame0 = rg.Tm_off[b]; n_frames = rg.Tm[b]; }   // ragged batch: the statistics run over THIS clip's frames
    const float *row = logmel + frame0 * n_mels + (int64_t)(live ? m : 0) * n_frames;
    float mean = 0.0f, den = 1.0f;
    if (normalize && live) {
        float p = 0.0f;
        for (int t = lane; t < n_frames; t += 64) p = p + row[t];
        mean = wave_sum64(p) / (float)n_frames;                    // src/audio.cpp:142
        float q = 0.0f;
        for (int t = lane; t < n_frames; t += 64) {
            const float c = row[t] - mean;
            q = q + c * c;
        }
        const float var = wave_sum64(q) / (float)(n_frames - 1);  // unbiased, :146-148
        den = __builtin_sqrtf(var) + 1e-5f;                       // :149
    }
    const int of = threadIdx.x >> 4, ob = threadIdx.x & 15;        // store role: frame of the tile, bin of the group
    float *out = feats + frame0 * n_mels + blockIdx.x * 16 + ob;
    for (int t0 = 0; t0 < n_frames; t0 += 64) {
        const int t = t0 + lane;
        float v = 0.0f;
        if (live && t < n_frames) v = normalize ? (row[t] - mean) / den : row[t];
        tile[lane][wave] = v;
        __syncthreads();
        if (t0 + of < n_frames && blockIdx.x * 16 + ob < n_mels) out[(int64_t)(t0 + of) * n_mels] = tile[of][ob];
        __syncthreads();
    }
}

void launch_mel_logmel(const float *pcm, int B, int64_t n_samples, int n_frames, const MelTables &t, float *logmel, hipStream_t s, const MelRag &rag) {
    constexpr int fpb = kFramesPerBlock * kOfflineIters;
    if (rag.pcm_off) n_frames = rag.max_frames;
    dim3 grid((n_frames + fpb - 1) / fpb, B);
    hipLaunchKernelGGL(mel_logmel_kernel<false>, grid, dim3(256), 0, s, pcm, n_samples, n_frames, t, logmel, rag);
}
void launch_mel_stream(const float *pre, int B, int64_t n_samples, int n_frames, const MelTables &t, float *logmel_tf, hipStream_t s) {
    dim3 grid((n_frames + kFramesPerBlock - 1) / kFramesPerBlock, B);
    hipLaunchKernelGGL(mel_logmel_kernel<true>, grid, dim3(256), 0, s, pre, n_samples, n_frames, t, logmel_tf, MelRag());
}
void launch_mel_normalize(const float *logmel, int B, int n_mels, int n_frames, int normalize, float *feats, hipStream_t s, const MelRag &rag) {
    hipLaunchKernelGGL(mel_normalize_kernel, dim3((n_mels + 15) / 16, B), dim3(1024), 0, s, logmel, n_mels, n_frames, normalize, feats, rag);
}

}  // namespace pk
