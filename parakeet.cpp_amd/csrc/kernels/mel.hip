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
// Wavefronts (= frames) per workgroup.  Round 3 had 4 (one frame per wave; walking 2 / 4 frames per wave for longer store runs was slower:
// 0.203 -> 0.234 / 0.237 ms per 64-clip batch -- the kernel is LDS-latency bound and wants the parallelism).  Round 4: 16 wavefronts per
// workgroup, still one frame each: the [m][16 frames] store tile leaves as 64-byte runs.
static constexpr int kStreamWaves = 4;
#ifndef PK_MEL_WAVES
#define PK_MEL_WAVES 16
#endif
static constexpr int kOfflineWaves = PK_MEL_WAVES;
// One frame's FFT array: 512 elements in one of three padded layouts, one per radix-8 pass (see the kernel): the largest needs 575 words.
static constexpr int kMelFA = 576;
template <int NW> static constexpr size_t mel_lds_bytes(bool offline) {
    (void)offline;
    return (size_t)(NW * 2 * kMelFA + 2 * kNfft + kMelMaxTaps) * sizeof(float);
}

// STREAM = false: preprocess_audio's framing (pre-emphasis, center=true, reflect padding), output [B][n_mels][n_frames].
// STREAM = true: StreamingAudioPreprocessor::process_chunk's framing (src/audio.cpp:222-241): the buffer is ALREADY
// pre-emphasised, center=false, frame t = samples [160 t, 160 t + 400) Hann-windowed and zero-padded on the right to the
// 512-point FFT; output [B][n_frames][n_mels] (the layout of its result, no normalisation follows).
// Every wavefront owns one frame and touches only its own LDS arrays, so the stages are ordered by wave-local fences (LDS
// operations of one wave complete in order once the counter is drained) instead of workgroup barriers: four frames of a
// workgroup no longer wait for each other ten times per FFT.
// Element i of a frame's FFT array lives at a padded position that depends on which pass reads it next: every pass reads (and writes) a
// lane's eight elements with eight instructions, and across the 32 lanes LDS serves per clock those addresses must fall on 32 different banks.
//   L1: i + (i >> 5)        bit-reversed load -> pass A (a lane's elements 8 g + k: bank 8 (g & 3) + (g >> 2) + k)
//   L2: i + (i >> 3)        pass A -> pass B (written at 9 g + k, read at 64 G + r + 8 k -> bank r + 8 G + 9 k)
//   L3: i + 8 (i >> 6)      pass B -> pass C -> spectrum (written at r + 8 G + 8 k + ..., read at g + 72 k)
#define PK_L1(i) ((i) + ((i) >> 5))
#define PK_L2(i) ((i) + ((i) >> 3))
#define PK_L3(i) ((i) + 8 * ((i) >> 6))
#define PK_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <bool STREAM, int NW /* wavefronts = frames per workgroup */>
__global__ __launch_bounds__(64 * NW, STREAM ? 1 : 8 /* offline: two 16-wave workgroups per CU = 8 waves per SIMD, 64 VGPRs */) void mel_logmel_kernel(const float *__restrict__ pcm, int64_t n_samples, int n_frames,
                                                             MelTables tb, float *__restrict__ logmel, MelRag rg) {
    // FFT arrays: padded layouts PK_L1 / PK_L2 / PK_L3 (above), kMelFA words per frame
    // LDS (dynamic: NW = 16 needs 102 KB): per-frame FFT arrays and power spectrum, the tables, the output tile
    constexpr int FA = kMelFA;                                      // padded FFT array of one frame
    constexpr int ITERS = 1;
    constexpr int FPB = NW * ITERS;
    extern __shared__ __attribute__((aligned(16))) float mel_sm[];
    // (the power spectrum of a frame -- 257 values -- lives in the unused upper part of its own imaginary array, words 296 .. 552: the spectrum
    //  is read from words 0 .. 288 only; the frame's log-mel values go into the head of its real array, which is dead by then: 4.2 KB of LDS per
    //  frame instead of 5.3 + an output tile, so that 16-frame workgroups fit twice on a CU)
    float *const s_re_all = mel_sm, *const s_im_all = s_re_all + NW * FA;
    // twiddles and the packed filterbank bands: read ~100 times per lane and frame -- from LDS instead of dependent L1 / L2 round trips.
    // The twiddles are laid out PER STAGE (stage lh uses w^(j << (8 - lh)), j < 2^lh, stored at 2^lh - 1 + j): the lanes of a butterfly
    // stage then read consecutive words (or broadcast) instead of a 2^(8-lh)-word stride that put 8 lanes on one bank in stages 3-6
    // (round 1: 8.5e7 LDS bank-conflict cycles per dispatch).
    float *const s_twr = s_im_all + NW * FA, *const s_twi = s_twr + kNfft, *const s_fb = s_twi + kNfft;
    // Offline: the log-mel values of the workgroup's NW frames leave as contiguous runs of [m][NW frames]: 64-byte, sector-aligned runs at
    // NW = 16 with the padded frame pitch (round 4; the 16-byte runs of 4-frame workgroups were 56 MB of write traffic for the 20 MB tensor)
    const int b = blockIdx.y;
    const float *x = pcm + (int64_t)b * n_samples;
    float *lm_clip = logmel + (int64_t)b * tb.n_mels * mel_logmel_pitch(n_frames);         // (offline layout: this clip's [n_mels][pitch] block)
    if constexpr (!STREAM) {
        if (rg.pcm_off) {                                           // ragged batch: this clip's own extent (kernels.hpp: MelRag)
            const int64_t o = rg.pcm_off[b];
            x = pcm + o;
            n_samples = rg.pcm_off[b + 1] - o;
            n_frames = rg.Tm[b];
            lm_clip = logmel + (int64_t)tb.n_mels * rg.Tm_pad_off[b];
            if ((int)blockIdx.x * FPB >= n_frames) return;          // the grid covers the longest clip (whole workgroup, before any barrier)
        }
    }
    for (int i = threadIdx.x; i < kNfft - 1; i += 64 * NW) {
        const int lh = 31 - __builtin_clz(i + 1), j = i + 1 - (1 << lh);   // i = 2^lh - 1 + j
        s_twr[i] = tb.tw_re[j << (8 - lh)];
        s_twi[i] = tb.tw_im[j << (8 - lh)];
    }
    for (int i = threadIdx.x; i < tb.fb_nnz; i += 64 * NW) s_fb[i] = tb.fbc[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *re = s_re_all + wave * FA, *im = s_im_all + wave * FA, *pw = im + 296;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
    const int t = blockIdx.x * FPB + it * NW + wave;
    const bool live = t < n_frames;
    
    if (live) {
        // all of a lane's samples, their predecessors and window taps are REQUESTED first (24 independent loads), then combined: with the
        // pre-emphasis written as a branch per element the ISA had 16 dependent memory round trips per frame (load, vmcnt(0), load, vmcnt(0), ...)
        float xc[kNfft / 64], xp[kNfft / 64], wn[kNfft / 64];
        bool first[kNfft / 64];
#pragma unroll
        for (int i = 0; i < kNfft / 64; ++i) {
            const int n = lane + 64 * i;
            if constexpr (STREAM) {
                xc[i] = n < 400 ? x[(int64_t)t * kHop + n] : 0.0f;
                wn[i] = n < 400 ? tb.window_left[n] : 0.0f;
                xp[i] = 0.0f;
                first[i] = false;
            } else {
                int64_t idx = (int64_t)t * kHop + n - kNfft / 2;  // center=true
                if (idx < 0) idx = -idx;                          // pad_mode="reflect"
                if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
                first[i] = idx == 0;
                xc[i] = x[idx];
                xp[i] = x[first[i] ? 0 : idx - 1];
                wn[i] = tb.window[n];
            }
        }
#pragma unroll
        for (int i = 0; i < kNfft / 64; ++i) {
            const int n = lane + 64 * i;
            const int r = (int)(__brev((unsigned)n) >> 23);       // 9-bit reversal
            if constexpr (STREAM) {
                re[PK_L1(r)] = n < 400 ? xc[i] * wn[i] : 0.0f;
            } else {
                const float p = 0.97f * xp[i];                    // preemphasis, src/audio.cpp:104-114: y[0] = x[0], y[n] = x[n] - 0.97 x[n-1]
                const float v = first[i] ? xc[i] : xc[i] - p;
                re[PK_L1(r)] = v * wn[i];
            }
        }
    }
    PK_WAVE_SYNC();
    // The nine radix-2 stages, three at a time: a lane takes the eight elements that differ in index bits L, L + 1, L + 2 into registers, runs the
    // 4 + 4 + 4 butterflies of stages L, L + 1, L + 2 on them (the same operations on the same operands as stage-by-stage: bit-identical) and
    // stores them once -- a third of the LDS traffic of one round trip per stage, which is what bounds this kernel (105 KB of LDS traffic per
    // frame against 3 KB of HBM; round 4: with every load batched and no dependent round trips left the time did not move).  Twiddles of stage
    // lh: w^(j << (8 - lh)) at table slot 2^lh - 1 + j, j = (lower element's index) mod 2^lh: one per lane for the first stage of a pass, two for
    // the second, four for the third.
    auto bfly = [](float &ar, float &ai, float &br, float &bi, float cr, float ci) {
        const float tr = __builtin_fmaf(-ci, bi, cr * br);
        const float ti = __builtin_fmaf(ci, br, cr * bi);
        const float nr = ar - tr, ni = ai - ti;
        ar = ar + tr;
        ai = ai + ti;
        br = nr;
        bi = ni;
    };
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        constexpr int kL[3] = {0, 3, 6};
        const int L = kL[pass], h0 = 1 << L;
        if (live) {
            const int low = lane & (h0 - 1), a0 = ((lane >> L) << (L + 3)) | low;
            float xr[8], xi[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = a0 + k * h0;
                const int p = pass == 0 ? PK_L1(i) : pass == 1 ? PK_L2(i) : PK_L3(i);
                xr[k] = re[p];
                xi[k] = pass == 0 ? 0.0f : im[p];                   // the input is real: no zeros are parked in LDS for the first pass to read back
            }
            const float c0r = s_twr[h0 - 1 + low], c0i = s_twi[h0 - 1 + low];
            float c1r[2], c1i[2], c2r[4], c2i[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) { c1r[u] = s_twr[2 * h0 - 1 + low + u * h0]; c1i[u] = s_twi[2 * h0 - 1 + low + u * h0]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { c2r[u] = s_twr[4 * h0 - 1 + low + u * h0]; c2i[u] = s_twi[4 * h0 - 1 + low + u * h0]; }
#pragma unroll
            for (int k = 0; k < 8; k += 2) bfly(xr[k], xi[k], xr[k + 1], xi[k + 1], c0r, c0i);                       // stage L: pairs differ in bit L
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((k & 2) == 0) bfly(xr[k], xi[k], xr[k + 2], xi[k + 2], c1r[k & 1], c1i[k & 1]);                   // stage L + 1
#pragma unroll
            for (int k = 0; k < 4; ++k) bfly(xr[k], xi[k], xr[k + 4], xi[k + 4], c2r[k], c2i[k]);                     // stage L + 2
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = a0 + k * h0;
                const int p = pass == 0 ? PK_L2(i) : PK_L3(i);
                re[p] = xr[k];
                im[p] = xi[k];
            }
        }
        PK_WAVE_SYNC();
    }
    if (live) {
        for (int f = lane; f <= kNfft / 2; f += 64) {
            const float s = __builtin_fmaf(re[PK_L3(f)], re[PK_L3(f)], im[PK_L3(f)] * im[PK_L3(f)]);
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
            // the chain is sequential by contract (natural f order); its operands are not: four weight / power pairs per trip to LDS.  Past the
            // band's end both factors are taken as +0: fma(+0, +0, acc) = acc for the non-negative sums of this chain.
            for (int f = lo; f <= hi; f += 4) {
                float w4[4], p4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool in = f + k <= hi;
                    w4[k] = in ? wm[f + k] : 0.0f;
                    p4[k] = in ? pw[f + k] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(w4[k], p4[k], acc);
            }
            const float lm = dlogf(acc + 5.96046448e-8f);
            if constexpr (STREAM) logmel[((int64_t)b * n_frames + t) * tb.n_mels + m] = lm;
            else re[m] = lm;
        }
    }
    }
    if constexpr (!STREAM) {
        __syncthreads();
        const int t0 = blockIdx.x * FPB, pitch = mel_logmel_pitch(n_frames);
        for (int idx = threadIdx.x; idx < tb.n_mels * FPB; idx += 64 * NW) {
            const int m = idx / FPB, tt = idx % FPB;
            if (t0 + tt < n_frames) lm_clip[(int64_t)m * pitch + t0 + tt] = s_re_all[tt * FA + m];
        }
    }
}

#undef PK_WAVE_SYNC
#undef PK_L1
#undef PK_L2
#undef PK_L3

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
    int64_t frame0 = (int64_t)b * n_frames;                        // first frame of this clip in the packed frame axis of the features
    int64_t lm0 = (int64_t)b * mel_logmel_pitch(n_frames);         // ... and of the (row-padded) log-mel blocks
    if (rg.pcm_off) { frame0 = rg.Tm_off[b]; lm0 = rg.Tm_pad_off[b]; n_frames = rg.Tm[b]; }   // ragged batch: the statistics run over THIS clip's frames
    const float *row = logmel + lm0 * n_mels + (int64_t)(live ? m : 0) * mel_logmel_pitch(n_frames);
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
    constexpr int fpb = kOfflineWaves;
    if (rag.pcm_off) n_frames = rag.max_frames;
    dim3 grid((n_frames + fpb - 1) / fpb, B);
    constexpr size_t lds = mel_lds_bytes<kOfflineWaves>(true);
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(&mel_logmel_kernel<false, kOfflineWaves>), lds);
    hipLaunchKernelGGL((mel_logmel_kernel<false, kOfflineWaves>), grid, dim3(64 * kOfflineWaves), lds, s, pcm, n_samples, n_frames, t, logmel, rag);
}
void launch_mel_stream(const float *pre, int B, int64_t n_samples, int n_frames, const MelTables &t, float *logmel_tf, hipStream_t s) {
    dim3 grid((n_frames + kStreamWaves - 1) / kStreamWaves, B);
    constexpr size_t lds = mel_lds_bytes<kStreamWaves>(false);
    hipLaunchKernelGGL((mel_logmel_kernel<true, kStreamWaves>), grid, dim3(64 * kStreamWaves), lds, s, pre, n_samples, n_frames, t, logmel_tf, MelRag());
}
void launch_mel_normalize(const float *logmel, int B, int n_mels, int n_frames, int normalize, float *feats, hipStream_t s, const MelRag &rag) {
    hipLaunchKernelGGL(mel_normalize_kernel, dim3((n_mels + 15) / 16, B), dim3(1024), 0, s, logmel, n_mels, n_frames, normalize, feats, rag);
}

}  // namespace pk
