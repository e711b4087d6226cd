// parakeet.cpp_amd/csrc/kernels/subsample.hip -- the depthwise-separable conv subsampling stack
// (reference ConvSubsampling::forward, src/encoder.cpp:219-241).  Activations are channels-last
// ([B][H][W][C]) so the 1x1 convolutions are plain row-major GEMMs on the MFMA kernel and the
// 3x3 depthwise taps are coalesced along C.
//
//  sub_conv1_dw1:  Conv2d(1->C,3x3,s2,p1)+ReLU fused with the following depthwise 3x3 s2: the
//                  [B][C][501][40] conv1 output (1.3 GB at B=64) is never written to HBM; conv1 rows live
//                  in registers for the two depthwise output rows that use them.
//  sub_dw:         depthwise 3x3 s2 p1 on a channels-last tensor.
// Tap order (ky,kx) and "bias after the chain" follow the oracle exactly (bit-identical results).
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

// One thread = one channel of a strip of YS output rows x XC output columns.  The three conv1 rows a depthwise output row
// needs (y1 = 2*y2-1 .. 2*y2+1) are computed ONCE into registers (the last one is carried over to the next output row), so
// a depthwise output costs 2x2 conv1 evaluations (36 fma) + 9 instead of 81 + 9.  The strip's window of mel features
// ((4*YS+3) x (4*XC+3) values, zero-padded outside the image) is staged in LDS once; every thread of a strip reads the
// same LDS addresses (broadcast reads with compile-time offsets), so the inner loops have no bounds checks at all.
// Out-of-range taps multiply a zero instead of being skipped: fma(w, 0, acc) == acc for every acc reachable from +0, so
// the chain equals the oracle's skip-the-padding chain bit for bit.
template <int XC, int YS = 8 /* output rows per strip: 8 on batches (conv1 rows shared by consecutive outputs), 2 when one utterance is all there is */>
__global__ __launch_bounds__(256) void sub_conv1_dw1_kernel(const float *__restrict__ feats, int Tm, int F, int C,
                                                            const float *__restrict__ w1 /*[9][C]*/, const float *__restrict__ b1,
                                                            const float *__restrict__ wd /*[9][C]*/, const float *__restrict__ bd,
                                                            int H1, int W1, int H2, int W2, int n_xc, int n_ys, int64_t n_strips,
                                                            float *__restrict__ out, SubRag rg) {
    constexpr int NC = 2 * XC + 1;
    constexpr int WR = 4 * YS + 3, WC = 4 * XC + 3, PW = (WC + 3) & ~3, TILE = WR * PW;   // input window per strip
    extern __shared__ __attribute__((aligned(16))) float win[];       // [256/C][WR][PW]
    const int spb = 256 / C;                                           // strips per block
    // ---- stage the windows: window row wr <-> input row 4*y2_0 - 3 + wr, column wc <-> 4*x2_0 - 3 + wc ----------------
    for (int e = threadIdx.x; e < spb * TILE; e += 256) {
        const int sl = e / TILE, rem = e % TILE, wr = rem / PW, wc = rem % PW;
        const int64_t strip = (int64_t)blockIdx.x * spb + sl;
        float v = 0.0f;
        if (strip < n_strips && wc < WC) {
            const int xc = (int)(strip % n_xc);
            int y2s, tm_b;                                            // first output row of the strip, mel frames of its utterance
            int64_t frame0;                                           // first mel frame of the utterance in the packed frame axis
            if (rg.strips.u) {                                        // ragged batch: strips never straddle utterances (kernels.hpp: SubRag)
                const RagUnit un = rg.strips.u[strip / n_xc];
                y2s = un.r0; tm_b = rg.Tm[un.b]; frame0 = rg.Tm_off[un.b];
            } else {
                const int b = (int)(strip / ((int64_t)n_xc * n_ys));
                y2s = (int)((strip / n_xc) % n_ys) * YS; tm_b = Tm; frame0 = (int64_t)b * Tm;
            }
            const int iy = 4 * y2s - 3 + wr, ix = 4 * xc * XC - 3 + wc;
            if (iy >= 0 && iy < tm_b && ix >= 0 && ix < F) v = feats[(frame0 + iy) * F + ix];
        }
        win[e] = v;
    }
    __syncthreads();
    const int c = threadIdx.x % C, sl = threadIdx.x / C;
    const int64_t strip = (int64_t)blockIdx.x * spb + sl;
    if (strip >= n_strips) return;
    const int xc = (int)(strip % n_xc);
    int y2_0;
    int64_t orow0;                                                     // first output row of the utterance in the packed H2 axis
    if (rg.strips.u) {
        const RagUnit un = rg.strips.u[strip / n_xc];
        y2_0 = un.r0;
        const int tm_b = rg.Tm[un.b];
        H1 = (tm_b - 1) / 2 + 1;
        H2 = rg.H2[un.b];
        orow0 = rg.H2_off[un.b];
    } else {
        const int b = (int)(strip / ((int64_t)n_xc * n_ys));
        y2_0 = (int)((strip / n_xc) % n_ys) * YS;
        orow0 = (int64_t)b * H2;
    }
    const int x2_0 = xc * XC, x1_0 = 2 * x2_0 - 1;                     // r[j] holds conv1 column x1_0 + j
    const float *tile = win + sl * TILE;
    float k1[9], kd[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { k1[i] = w1[i * C + c]; kd[i] = wd[i * C + c]; }
    const float bias1 = b1[c], biasd = bd[c];
    float r0[NC], r1[NC], r2[NC];
    // conv1 + ReLU (src/encoder.cpp:223-224) of conv1 row y1 = 2*y2_0 - 1 + q: its input rows are window rows 2q .. 2q+2
    auto conv_row = [&](int q, float (&r)[NC]) {
        const int y1 = 2 * y2_0 - 1 + q;
        const float *rp = tile + 2 * q * PW;
        const bool row_ok = y1 >= 0 && y1 < H1;                        // else: zero padding of the depthwise conv
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int jy = 0; jy < 3; ++jy)
#pragma unroll
                for (int jx = 0; jx < 3; ++jx) acc = __builtin_fmaf(k1[jy * 3 + jx], rp[jy * PW + 2 * j + jx], acc);
            float v = acc + bias1;
            v = v > 0.0f ? v : 0.0f;                                   // ReLU :224
            const int x1 = x1_0 + j;
            r[j] = (row_ok && x1 >= 0 && x1 < W1) ? v : 0.0f;
        }
    };
    for (int yy = 0; yy < YS; ++yy) {
        const int y2 = y2_0 + yy;
        if (y2 >= H2) break;
        if (yy == 0) {
            conv_row(0, r0);
        } else {
#pragma unroll
            for (int j = 0; j < NC; ++j) r0[j] = r2[j];
        }
        conv_row(2 * yy + 1, r1);
        conv_row(2 * yy + 2, r2);
        float *orow = out + ((orow0 + y2) * W2 + x2_0) * C + c;
#pragma unroll
        for (int xl = 0; xl < XC; ++xl) {
            float acc2 = 0.0f;                                         // dw1 :226, taps in (ky,kx) order
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc2 = __builtin_fmaf(kd[kx], r0[2 * xl + kx], acc2);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc2 = __builtin_fmaf(kd[3 + kx], r1[2 * xl + kx], acc2);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc2 = __builtin_fmaf(kd[6 + kx], r2[2 * xl + kx], acc2);
            if (x2_0 + xl < W2) orow[(int64_t)xl * C] = acc2 + biasd;
        }
    }
}

// Round 6: the same computation with TWO adjacent channels per thread on packed fp32 instructions.  The one-channel kernel is VALU-bound (SQ counters: the vector
// ALUs ~90 % busy at four waves per SIMD) and every one of its fmas is a scalar-per-lane v_fma_f32 -- half the vector fp32 rate -- fed by one broadcast LDS read per
// window value.  Here a lane carries channels (c, c + 1): weights, accumulators and the three conv1 rows are 2-vectors, the window value is the SAME for both
// halves, so every tap is ONE v_pk_fma_f32 (weight pair, splat operand, accumulator pair) and one LDS read serves two channels; outputs leave as 8-byte stores.
// Each half is the identical IEEE fma chain in the identical tap order: bit-identical results (tests/test_gpu_encoder.py, test_gpu_ragged.py).  Chunks of XC = 4
// output columns (three rows of 9 two-vectors: 122 VGPRs, four waves per SIMD; chunks of 5 need 132 and measured 8 % slower).  80 mel bins: 0.250 -> 0.183 ms per
// 64 x 10 s batch, 128 bins: 0.561 -> 0.420 ms per 32 x 30 s batch (profiles/r06_sub_conv_packed_ab.txt).
template <int XC, int YS = 8>
__global__ __launch_bounds__(256) void sub_conv1_dw1_c2_kernel(const float *__restrict__ feats, int Tm, int F, int C,
                                                               const float *__restrict__ w1 /*[9][C]*/, const float *__restrict__ b1,
                                                               const float *__restrict__ wd /*[9][C]*/, const float *__restrict__ bd,
                                                               int H1, int W1, int H2, int W2, int n_xc, int n_ys, int64_t n_strips,
                                                               float *__restrict__ out, SubRag rg) {
    typedef float f2_ __attribute__((ext_vector_type(2)));
    constexpr int NC = 2 * XC + 1;
    constexpr int WR = 4 * YS + 3, WC = 4 * XC + 3, PW = (WC + 3) & ~3, TILE = WR * PW;   // input window per strip
    extern __shared__ __attribute__((aligned(16))) float win[];       // [spb][WR][PW]
    const int C2 = C >> 1, spb = 256 / C2;                             // threads per strip (two channels each), strips per block
    for (int e = threadIdx.x; e < spb * TILE; e += 256) {              // window row wr <-> input row 4*y2_0 - 3 + wr, column wc <-> 4*x2_0 - 3 + wc
        const int sl = e / TILE, rem = e % TILE, wr = rem / PW, wc = rem % PW;
        const int64_t strip = (int64_t)blockIdx.x * spb + sl;
        float v = 0.0f;
        if (strip < n_strips && wc < WC) {
            const int xc = (int)(strip % n_xc);
            int y2s, tm_b;
            int64_t frame0;
            if (rg.strips.u) {
                const RagUnit un = rg.strips.u[strip / n_xc];
                y2s = un.r0; tm_b = rg.Tm[un.b]; frame0 = rg.Tm_off[un.b];
            } else {
                const int b = (int)(strip / ((int64_t)n_xc * n_ys));
                y2s = (int)((strip / n_xc) % n_ys) * YS; tm_b = Tm; frame0 = (int64_t)b * Tm;
            }
            const int iy = 4 * y2s - 3 + wr, ix = 4 * xc * XC - 3 + wc;
            if (iy >= 0 && iy < tm_b && ix >= 0 && ix < F) v = feats[(frame0 + iy) * F + ix];
        }
        win[e] = v;
    }
    __syncthreads();
    const int c = 2 * (threadIdx.x % C2), sl = threadIdx.x / C2;
    const int64_t strip = (int64_t)blockIdx.x * spb + sl;
    if (strip >= n_strips) return;
    const int xc = (int)(strip % n_xc);
    int y2_0;
    int64_t orow0;
    if (rg.strips.u) {
        const RagUnit un = rg.strips.u[strip / n_xc];
        y2_0 = un.r0;
        const int tm_b = rg.Tm[un.b];
        H1 = (tm_b - 1) / 2 + 1;
        H2 = rg.H2[un.b];
        orow0 = rg.H2_off[un.b];
    } else {
        const int b = (int)(strip / ((int64_t)n_xc * n_ys));
        y2_0 = (int)((strip / n_xc) % n_ys) * YS;
        orow0 = (int64_t)b * H2;
    }
    const int x2_0 = xc * XC, x1_0 = 2 * x2_0 - 1;                     // r[j] holds conv1 column x1_0 + j
    const float *tile = win + sl * TILE;
    f2_ k1[9], kd[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float2 a = *reinterpret_cast<const float2 *>(w1 + i * C + c), d2 = *reinterpret_cast<const float2 *>(wd + i * C + c);
        k1[i] = f2_{a.x, a.y}; kd[i] = f2_{d2.x, d2.y};
    }
    const float2 b1v = *reinterpret_cast<const float2 *>(b1 + c), bdv = *reinterpret_cast<const float2 *>(bd + c);
    const f2_ bias1 = {b1v.x, b1v.y}, biasd = {bdv.x, bdv.y};
    f2_ r0[NC], r1[NC], r2[NC];
    auto conv_row = [&](int q, f2_ (&r)[NC]) {                       // conv1 + ReLU (src/encoder.cpp:223-224) of conv1 row y1 = 2*y2_0 - 1 + q
        const int y1 = 2 * y2_0 - 1 + q;
        const float *rp = tile + 2 * q * PW;
        const bool row_ok = y1 >= 0 && y1 < H1;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            f2_ acc = {0.0f, 0.0f};
#pragma unroll
            for (int jy = 0; jy < 3; ++jy)
#pragma unroll
                for (int jx = 0; jx < 3; ++jx) {
                    const float v = rp[jy * PW + 2 * j + jx];
                    acc = __builtin_elementwise_fma(k1[jy * 3 + jx], f2_{v, v}, acc);
                }
            f2_ v = acc + bias1;
            v.x = v.x > 0.0f ? v.x : 0.0f;                             // ReLU :224
            v.y = v.y > 0.0f ? v.y : 0.0f;
            const int x1 = x1_0 + j;
            r[j] = (row_ok && x1 >= 0 && x1 < W1) ? v : f2_{0.0f, 0.0f};
        }
    };
    for (int yy = 0; yy < YS; ++yy) {
        const int y2 = y2_0 + yy;
        if (y2 >= H2) break;
        if (yy == 0) {
            conv_row(0, r0);
        } else {
#pragma unroll
            for (int j = 0; j < NC; ++j) r0[j] = r2[j];
        }
        conv_row(2 * yy + 1, r1);
        conv_row(2 * yy + 2, r2);
        float *orow = out + ((orow0 + y2) * W2 + x2_0) * C + c;
#pragma unroll
        for (int xl = 0; xl < XC; ++xl) {
            f2_ acc2 = {0.0f, 0.0f};                                   // dw1 :226, taps in (ky,kx) order
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc2 = __builtin_elementwise_fma(kd[kx], r0[2 * xl + kx], acc2);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc2 = __builtin_elementwise_fma(kd[3 + kx], r1[2 * xl + kx], acc2);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc2 = __builtin_elementwise_fma(kd[6 + kx], r2[2 * xl + kx], acc2);
            const f2_ o = acc2 + biasd;
            if (x2_0 + xl < W2) *reinterpret_cast<float2 *>(orow + (int64_t)xl * C) = make_float2(o.x, o.y);
        }
    }
}

__global__ __launch_bounds__(256) void sub_dw_kernel(const float *__restrict__ in, int H, int W, int C,
                                                     const float *__restrict__ wd /*[9][C]*/, const float *__restrict__ bd,
                                                     int Ho, int Wo, int64_t n_pix, float *__restrict__ out, SubRag rg) {
    const int ppb = 256 / C;
    const int c = threadIdx.x % C;
    const int64_t pix = (int64_t)blockIdx.x * ppb + threadIdx.x / C;
    if (pix >= n_pix) return;
    const int xo = (int)(pix % Wo);
    int yo = (int)((pix / Wo) % Ho);
    const float *src;
    if (rg.strips.u) {                                          // ragged batch: output row pix / Wo of the packed axis = (utterance, local row)
        const RagUnit un = rg.strips.u[pix / Wo];
        yo = un.r0; H = rg.H2[un.b];
        src = in + (int64_t)rg.H2_off[un.b] * W * C;
    } else {
        src = in + (pix / ((int64_t)Wo * Ho)) * H * W * C;
    }
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * yo + ky - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * xo + kx - 1;
            if (ix < 0 || ix >= W) continue;
            acc = __builtin_fmaf(wd[(ky * 3 + kx) * C + c], src[((int64_t)iy * W + ix) * C + c], acc);
        }
    }
    out[pix * C + c] = acc + bd[c];
}

template <int XC, int YS = 8>
static void launch_c1d1(const float *feats, int B, int Tm, int F, int C, const float *w1, const float *b1, const float *wd,
                        const float *bd, int H1, int W1, int H2, int W2, float *out, hipStream_t s, const SubRag &rag) {
    constexpr int WR = 4 * YS + 3, PW = (4 * XC + 3 + 3) & ~3;
    const int spb = 256 / C, n_ys = (H2 + YS - 1) / YS, n_xc = (W2 + XC - 1) / XC;
    const int64_t n_strips = rag.strips.u ? (int64_t)rag.strips.count * n_xc : (int64_t)B * n_ys * n_xc;
    const size_t lds = (size_t)spb * WR * PW * sizeof(float);
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(&sub_conv1_dw1_kernel<XC, YS>), lds);
    hipLaunchKernelGGL((sub_conv1_dw1_kernel<XC, YS>), dim3((unsigned)((n_strips + spb - 1) / spb)), dim3(256), lds, s, feats, Tm, F, C, w1, b1,
                       wd, bd, H1, W1, H2, W2, n_xc, n_ys, n_strips, out, rag);
}
template <int XC, int YS = 8>
static void launch_c1d1_c2(const float *feats, int B, int Tm, int F, int C, const float *w1, const float *b1, const float *wd,
                           const float *bd, int H1, int W1, int H2, int W2, float *out, hipStream_t s, const SubRag &rag) {
    constexpr int WR = 4 * YS + 3, PW = (4 * XC + 3 + 3) & ~3;
    const int spb = 256 / (C / 2), n_ys = (H2 + YS - 1) / YS, n_xc = (W2 + XC - 1) / XC;
    const int64_t n_strips = rag.strips.u ? (int64_t)rag.strips.count * n_xc : (int64_t)B * n_ys * n_xc;
    const size_t lds = (size_t)spb * WR * PW * sizeof(float);
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(&sub_conv1_dw1_c2_kernel<XC, YS>), lds);
    hipLaunchKernelGGL((sub_conv1_dw1_c2_kernel<XC, YS>), dim3((unsigned)((n_strips + spb - 1) / spb)), dim3(256), lds, s, feats, Tm, F, C, w1, b1,
                       wd, bd, H1, W1, H2, W2, n_xc, n_ys, n_strips, out, rag);
}
// the two-channel packed kernel on batches (EXPERIMENTAL builds: PK_SUB_C2=0 keeps the one-channel kernel)
static bool sub_c2_on() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_SUB_C2"); return e ? atoi(e) != 0 : true; }();
    return on;
#else
    return true;
#endif
}
static constexpr int64_t kSmallStripRows = 1024;
int sub_conv1_dw1_strip_rows(int64_t total_h2_rows) { return total_h2_rows <= kSmallStripRows ? 2 : 8; }
void launch_sub_conv1_dw1(const float *feats, int B, int Tm, int F, int C, const float *w1, const float *b1, const float *wd,
                          const float *bd, float *out, hipStream_t s, const SubRag &rag) {
    const int H1 = (Tm - 1) / 2 + 1, W1 = (F - 1) / 2 + 1, H2 = (H1 - 1) / 2 + 1, W2 = (W1 - 1) / 2 + 1;
    // batches: 80 mel bins -> two 10-column chunks per output row; 128 -> four chunks of 8.  (Round 4: chunks of 20 / 16 hold three conv1 rows of
    // 41 / 33 values in registers, 169 / 144 VGPRs = three waves per SIMD; half the chunk = 108 VGPRs, four waves: 0.317 -> 0.243 ms at 64 x 10 s
    // for 5 % more conv1 columns; chunks of 5 measure the same.)  One or two utterances keep the wide chunk:
    // a strip of 8 output rows per thread shares its conv1 rows; one or two utterances give only a few dozen such strips -- strips of 2 rows
    // (1.5x the conv1 work, four times the workgroups, a quarter of the serial chain each: 77 -> ~25 us for one 10 s clip)
    // (ragged batch: the caller built rag.strips with rag.strip_rows = sub_conv1_dw1_strip_rows(total H2 rows) rows per unit)
    const bool small = rag.strips.u ? rag.strip_rows == 2 : (int64_t)B * H2 <= kSmallStripRows;
    // batches: two channels per thread on packed fp32 (sub_conv1_dw1_c2_kernel), chunks of 4 columns; C even with 256 % (C / 2) == 0 (8-byte aligned weight pairs)
    const bool c2 = !small && sub_c2_on() && C % 2 == 0 && C >= 64 && 256 % (C / 2) == 0;
    if (W2 <= 20 || W2 % 20 == 0) {
        if (small) launch_c1d1<20, 2>(feats, B, Tm, F, C, w1, b1, wd, bd, H1, W1, H2, W2, out, s, rag);
        else if (c2) launch_c1d1_c2<4>(feats, B, Tm, F, C, w1, b1, wd, bd, H1, W1, H2, W2, out, s, rag);
        else launch_c1d1<10>(feats, B, Tm, F, C, w1, b1, wd, bd, H1, W1, H2, W2, out, s, rag);
    } else {
        if (small) launch_c1d1<16, 2>(feats, B, Tm, F, C, w1, b1, wd, bd, H1, W1, H2, W2, out, s, rag);
        else if (c2) launch_c1d1_c2<4>(feats, B, Tm, F, C, w1, b1, wd, bd, H1, W1, H2, W2, out, s, rag);
        else launch_c1d1<8>(feats, B, Tm, F, C, w1, b1, wd, bd, H1, W1, H2, W2, out, s, rag);
    }
}
// Depthwise 3x3 stride-2 conv (dw2, src/encoder.cpp:230), channels-last.  One thread = 4 adjacent channels x XO adjacent output
// pixels of one output row: the 2*XO+1 input pixels of each of the three input rows are loaded once as float4 and shared by the XO
// outputs (the first version read every input pixel 2.25 times with 4-byte loads: 715 MB of HBM reads for a 329 MB tensor).  A
// wavefront covers the 256 channels of one item (1 KB contiguous per pixel).  Items are numbered (b, yo, x chunk) and dealt to the
// XCDs in contiguous ranges, so the rows two neighbouring output rows share are re-read from the same XCD's L2.
// Per output the taps run ky-major, kx-minor from +0 and out-of-image taps are skipped: the reference's (and the oracle's) chain.
template <int XO>
__global__ __launch_bounds__(256) void sub_dw4_kernel(const float *__restrict__ in, int H, int W, int C,
                                                      const float *__restrict__ wd /*[9][C]*/, const float *__restrict__ bd,
                                                      int Ho, int Wo, int n_xc, int64_t n_items, int n_blocks, float *__restrict__ out, SubRag rg) {
    const int c4n = C >> 2;                                     // float4 lanes per item
    const int ipb = 256 / c4n;                                  // items per block
    const int per_xcd = (n_blocks + 7) >> 3;
    const int lb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lb >= n_blocks) return;
    const int64_t item = (int64_t)lb * ipb + threadIdx.x / c4n;
    if (item >= n_items) return;
    const int c4 = threadIdx.x % c4n;
    const int xc = (int)(item % n_xc);
    int yo;
    int64_t in_row0, out_row;                                   // first input row of the utterance / this output row, packed axes
    if (rg.strips.u) {                                          // ragged batch (kernels.hpp: SubRag, one unit per output row)
        const RagUnit un = rg.strips.u[item / n_xc];
        yo = un.r0; H = rg.H2[un.b];
        in_row0 = rg.H2_off[un.b]; out_row = (int64_t)rg.T_off[un.b] + yo;
    } else {
        const int b = (int)(item / ((int64_t)n_xc * Ho));
        yo = (int)((item / n_xc) % Ho);
        in_row0 = (int64_t)b * H; out_row = (int64_t)b * Ho + yo;
    }
    const int x0 = xc * XO;
    const float4 *src = reinterpret_cast<const float4 *>(in + in_row0 * W * C) + c4;
    float4 wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = reinterpret_cast<const float4 *>(wd + (int64_t)k * C)[c4];
    const float4 bias = reinterpret_cast<const float4 *>(bd)[c4];
    float4 acc[XO];
#pragma unroll
    for (int xl = 0; xl < XO; ++xl) acc[xl] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * yo + ky - 1;
        if (iy < 0 || iy >= H) continue;                        // uniform over the wavefront
        float4 px[2 * XO + 1];
#pragma unroll
        for (int i = 0; i < 2 * XO + 1; ++i) {
            const int ix = 2 * x0 - 1 + i;
            px[i] = (ix >= 0 && ix < W) ? src[((int64_t)iy * W + ix) * c4n] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int xl = 0; xl < XO; ++xl)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * (x0 + xl) + kx - 1;
                if (ix < 0 || ix >= W) continue;                // skipped tap: the chain does not see it
                const float4 w = wt[ky * 3 + kx], v = px[2 * xl + kx];
                acc[xl].x = __builtin_fmaf(w.x, v.x, acc[xl].x);
                acc[xl].y = __builtin_fmaf(w.y, v.y, acc[xl].y);
                acc[xl].z = __builtin_fmaf(w.z, v.z, acc[xl].z);
                acc[xl].w = __builtin_fmaf(w.w, v.w, acc[xl].w);
            }
    }
    float4 *dst = reinterpret_cast<float4 *>(out + (out_row * Wo) * C) + c4;
#pragma unroll
    for (int xl = 0; xl < XO; ++xl)
        if (x0 + xl < Wo) dst[(int64_t)(x0 + xl) * c4n] = make_float4(acc[xl].x + bias.x, acc[xl].y + bias.y, acc[xl].z + bias.z, acc[xl].w + bias.w);
}

template <int XO>
static void launch_dw4(const float *in, int B, int H, int W, int C, const float *wd, const float *bd, int Ho, int Wo, float *out, hipStream_t s, const SubRag &rag) {
    const int n_xc = (Wo + XO - 1) / XO;
    const int64_t n_items = rag.strips.u ? (int64_t)rag.strips.count * n_xc : (int64_t)B * Ho * n_xc;
    const int ipb = 256 / (C / 4);
    const int n_blocks = (int)((n_items + ipb - 1) / ipb);
    hipLaunchKernelGGL(sub_dw4_kernel<XO>, dim3(((n_blocks + 7) / 8) * 8), dim3(256), 0, s, in, H, W, C, wd, bd, Ho, Wo, n_xc, n_items, n_blocks, out, rag);
}

void launch_sub_dw(const float *in, int B, int H, int W, int C, const float *wd, const float *bd, float *out, hipStream_t s, const SubRag &rag) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (C % 4 == 0 && 256 % (C / 4) == 0 && C >= 16) {
        if (Wo % 5 == 0) launch_dw4<5>(in, B, H, W, C, wd, bd, Ho, Wo, out, s, rag);
        else launch_dw4<4>(in, B, H, W, C, wd, bd, Ho, Wo, out, s, rag);
        return;
    }
    const int64_t n_pix = rag.strips.u ? (int64_t)rag.strips.count * Wo : (int64_t)B * Ho * Wo;
    const int ppb = 256 / C;
    hipLaunchKernelGGL(sub_dw_kernel, dim3((unsigned)((n_pix + ppb - 1) / ppb)), dim3(256), 0, s, in, H, W, C, wd, bd, Ho, Wo,
                       n_pix, out, rag);
}

}  // namespace pk
