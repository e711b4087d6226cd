// parakeet.cpp_amd/csrc/kernels/gemm.hip -- fp32 MFMA GEMM with fused epilogues (gfx950).
//
// out[M][N] = epi(A[M][K] * W[N][K]^T + bias).  This one kernel template carries ~93 % of the
// encoder's arithmetic: every nn::Linear / 1x1 Conv call site of the reference
// (src/encoder.cpp:41-45 FFN, :120-122 QKV, :148 pos_proj, :177 out_proj, :63/:70 pointwise convs,
// :227/:231 subsampling 1x1 convs, :240 proj_; src/ctc.cpp:19; src/tdt.cpp:17 enc_proj_).
//
// Design for CDNA4:
//  * v_mfma_f32_32x32x2_f32: exact fp32, bit-identical to a k-ordered fmaf chain (64 FLOP/clk/SIMD,
//    157 TF peak).  Lanes 0-31 feed k = 2s, lanes 32-63 feed k = 2s+1, so the accumulation order is
//    the natural k = 0,1,2,... order the CPU oracle uses: results are bit-identical to the oracle.
//  * 256-thread workgroup = 4 wavefronts in a 2x2 grid; each wave owns a (BM/2)x(BN/2) sub-tile held
//    as TMxTN 32x32 accumulators (16 VGPRs each).
//  * K is tiled by 32: A and W tiles are staged global -> registers (float4, 128-B rows, coalesced)
//    -> LDS with a 33-float row pitch, so the per-lane scalar ds_read_b32 fragment reads
//    (32 different rows, same k) hit 32 different banks.  fp32 MFMA is slow enough (64 cycles per
//    32x32x2) that 2 LDS reads per MFMA use <15 % of the LDS issue rate.
//  * Double-buffered LDS: the global loads of tile k+1 are issued before the MFMAs of tile k and
//    written to the other buffer after them -> one __syncthreads per K tile.
//  * XCD-aware block swizzle: consecutive tiles (sharing A rows / W rows) stay on one XCD's L2.
//  * Epilogues (bias, ReLU, SiLU, residual + alpha*y, GLU) are applied to the accumulator registers:
//    no separate elementwise passes over HBM.
#include "../pk_devmath.h"
#include "kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include "gemm_pipe.hpp"
#include "gemm_bf16.hpp"
#include "gemm_bf16_glds.hpp"
#ifdef PK_EXPERIMENTAL
#include "gemm_bf16_ring.hpp"
#endif

namespace pk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;
static constexpr int LDP = BK + 1;  // LDS row pitch in floats

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    constexpr int WM = BM / 2, WN = BN / 2;      // wave sub-tile
    constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 accumulators per wave
    constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;  // output columns per block
    static_assert(EPI != EPI_GLU || (TN % 2 == 0), "GLU needs an even number of column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware bijective remap (block b runs on XCD b % 8): XCD x gets a contiguous range of tiles.
    int bid = blockIdx.x;
    {
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * NOUT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // per-thread global source rows of the staging chunks
    const float *a_src[A_CH];
    const float *w_src[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + 256 * i, row = c >> 3, c4 = c & 7;
        int gr = m0 + row;
        gr = gr < g.M ? gr : g.M - 1;
        a_src[i] = g.A + (int64_t)gr * g.lda + c4 * 4;
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int c = tid + 256 * i, v = c >> 3, c4 = c & 7;
        int wr;
        if constexpr (EPI == EPI_GLU) {
            // virtual column v -> (wave column, tile, lane column); tiles [0,TN/2) are the value half,
            // tiles [TN/2,TN) the gate half of the SAME output columns, so one lane holds both.
            constexpr int HT = TN / 2;
            const int vw = v / WN, rem = v % WN, tn = rem >> 5, cc = rem & 31;
            int col = n0 + vw * (WN / 2) + (tn % HT) * 32 + cc;
            col = col < g.N ? col : g.N - 1;
            wr = (tn / HT) * g.N + col;
        } else {
            wr = n0 + v;
            wr = wr < g.N ? wr : g.N - 1;
        }
        w_src[i] = g.W + (int64_t)wr * g.ldw + c4 * 4;
    }

    float4 ra[A_CH], rw[W_CH];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const float4 *>(a_src[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i] = *reinterpret_cast<const float4 *>(w_src[i] + kt * BK);
    };
    auto lstore = [&](int buf) {
        float *As = smem + buf * (BM + BN) * LDP;
        float *Ws = As + BM * LDP;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int c = tid + 256 * i;
            float *d = As + (c >> 3) * LDP + (c & 7) * 4;
            d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int c = tid + 256 * i;
            float *d = Ws + (c >> 3) * LDP + (c & 7) * 4;
            d[0] = rw[i].x; d[1] = rw[i].y; d[2] = rw[i].z; d[3] = rw[i].w;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = g.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int frag = (lane & 31) * LDP + (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float *Ab = smem + cur * (BM + BN) * LDP + wm * WM * LDP + frag;
        const float *Wb = smem + cur * (BM + BN) * LDP + BM * LDP + wn * WN * LDP + frag;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ab[i * 32 * LDP + 2 * kk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Wb[j * 32 * LDP + 2 * kk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int lc = lane & 31, lr = 4 * (lane >> 5);
    constexpr int TNO = (EPI == EPI_GLU) ? TN / 2 : TN;
#pragma unroll
    for (int j = 0; j < TNO; ++j) {
        const int col = n0 + wn * (EPI == EPI_GLU ? WN / 2 : WN) + j * 32 + lc;
        if (col >= g.N) continue;
        const float bias = g.bias ? g.bias[col] : 0.0f;
        float bias_g = 0.0f;
        if constexpr (EPI == EPI_GLU) bias_g = g.bias ? g.bias[g.N + col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + lr;
                if (row >= g.M) continue;
                float v = acc[i][j][r];
                if (g.bias) v = v + bias;
                if constexpr (EPI == EPI_RELU) {
                    v = v > 0.0f ? v : 0.0f;
                } else if constexpr (EPI == EPI_SILU) {
                    v = dsiluf(v);
                } else if constexpr (EPI == EPI_RESID) {
                    const float y = v * g.alpha;
                    v = g.resid[(int64_t)row * g.ldr + col] + y;
                } else if constexpr (EPI == EPI_GLU) {
                    float gt = acc[i][j + TN / 2][r];
                    if (g.bias) gt = gt + bias_g;
                    v = v * dsigmoidf(gt);
                }
                if (g.remap_rows) g.out[(int64_t)(row / g.remap_rows) * g.remap_gs + (int64_t)(row % g.remap_rows) * g.remap_rs + (int64_t)col * g.remap_cs] = v;
                else g.out[(int64_t)row * g.ldo + (col < g.sigma_cols ? ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)) : col)] = v;
            }
        }
    }
}

template <int BM, int BN, int EPI>
static void launch_one(const GemmArgs &a, hipStream_t s) {
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * LDP * sizeof(float);
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(&gemm_nt_kernel<BM, BN, EPI>), lds);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, EPI>), dim3(n_tiles), dim3(256), lds, s, a, tiles_n, n_tiles);
}

// Tile choice, from tools/ubench/gemm_sweep on MI355X (profiles/r01_gemm_sweep.txt): the pipelined kernels win where
// the K loop is short (most of the encoder is K = 512); wide outputs like 128x128 tiles on 8 waves, long-K / narrow-N
// products 128x64, everything else 64x64 (more resident workgroups to overlap one tile's epilogue with another's MFMAs).
// Variant mask of the tile table.  Bits: 1 = long-K single-round products (fc2, sub_proj) on the single-buffered 128x128 / 8 waves of 32x64 /
// BK 64 tile; 2 = wide outputs (fc1, qkv, sub_pw) single-buffered; 4 = out_proj / pw2 on sb 64x128 / 4 waves; 8 = GLU on sb; 16 = wide outputs
// on sb BK 64; 64 = out_proj / pw2 on sb 128x128 / 8 waves; 256 = the long-K single-round products (fc2, sub_proj) on 8 waves of 64x32 / BK 32 / sb (round 6: fc2 -1.4 %,
// step -0.1 ms interleaved, profiles/r06_gemm_sweep_fc2_variants.txt).  331 = 75 + 256; 75 = what the engine measurements of round 2 picked
// (profiles/r02_gemm_variant_ab.txt: step 20.44 -> 19.73 ms with 11; bit 4 is level; bit 64 takes out_proj / pw2 from 0.81 to 0.77 ms per step).
// A production build has NO run-time switch: the mask is a constant.  Experiment builds (make EXPERIMENTAL=1 -> -DPK_EXPERIMENTAL) read
// PK_GEMM_VARIANT for the interleaved A/B runs of tools/experiments/gemm_variant_ab.sh.
static int gemm_variant_mask() {
#ifdef PK_EXPERIMENTAL
    static const int m = [] { const char *e = getenv("PK_GEMM_VARIANT"); return e ? atoi(e) : 331; }();
    return m;
#else
    return 331;
#endif
}

template <int EPI>
static void launch_epi(const GemmArgs &a, hipStream_t s) {
    if (a.K < 64) { launch_one<64, 64, EPI>(a, s); return; }           // pipelined kernels need >= 2 K tiles
    const int vm = gemm_variant_mask();
    // measured table: profiles/r01_gemm_sweep_v5.txt (128x128 tile on 8 waves of 32x64 wins for every wide output and for the
    // 321k-row subsampling products; long-K / narrow-N products like fc2 of the 110M model take 128x64)
    // long-K products whose 128x128 tiles fill the chip exactly once (fc2 / sub_proj of the 110M model: 252 tiles for 256 CUs): one
    // 8-wave workgroup per CU with BK = 64 -- half the barriers per k, nothing to share the CU with (-6 % vs two 128x64 workgroups)
    const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (a.M >= 1024 && a.N >= 256 && a.K >= 1024 && a.K % 64 == 0 && tiles128 <= 256) {
        // bit 256 (round 6): 8 waves of 64 x 32, BK 32, single-buffered -- 2-3 % ahead of the BK 64 tile in the sweep (profiles/r06_gemm_sweep_fc2_variants.txt: 137 vs 140.5 us)
        if (vm & 256) launch_gemm_pipe<2, 4, 2, 1, 32, EPI, 1>(a, s);
        else if (vm & 1) launch_gemm_pipe<4, 2, 1, 2, 64, EPI, 1>(a, s);
        else launch_gemm_pipe<2, 4, 2, 1, 64, EPI>(a, s);
        return;
    }
    // round 2: the SINGLE-buffered loop (template parameter NBUF = 1: half the LDS, two barriers per K tile) is ahead of the double-buffered
    // one on every large shape, in the micro-benchmark (tools/ubench/gemm_sweep ml: main loop 130-135 vs 118-125 TF) and, by less, in the
    // engine (fc2 -8 %, fc1 -3.6 %, qkv -5 %, GLU -3 %): gemm_variant_mask().
    if constexpr (EPI == EPI_NONE || EPI == EPI_RELU || EPI == EPI_SILU) {
        if (a.ln_stats) { launch_gemm_pipe<4, 2, 1, 2, 32, EPI, 1, true>(a, s); return; }    // (gemm_ln_stats_applies: the wide-output tile below, LayerNorm applied while staging A)
    }
    if (a.M >= 1024 && (a.N >= 1024 || (a.M >= 65536 && a.N >= 256))) {
        // bit 128: 192x128 tiles (8 waves of 96x32) where they turn a fractional second round of the 512 resident 128x128 workgroups into one
        // full round (attn_qkv of the 110M model at 64 x 10 s: 63 x 12 = 756 tiles = 1.48 rounds -> 42 x 12 = 504)
        // Measured (profiles/r04_gemm_tile192_ab.txt): NO gain, qkv 1.97 -> 1.99 ms per step -- workgroups are handed out as slots free up, so a
        // CU never idles for a 'round'; the bit stays off (results are bit-identical either way: same k order).
        if constexpr (EPI != EPI_GLU) {
            const int64_t tiles192 = (int64_t)((a.M + 191) / 192) * ((a.N + 127) / 128);
            if ((vm & 128) && tiles192 <= 512 && tiles128 > 512 && tiles128 < 900) { launch_gemm_pipe<2, 4, 3, 1, 32, EPI, 1>(a, s); return; }
        }
        if ((vm & 16) && a.K % 64 == 0 && a.K >= 128) launch_gemm_pipe<4, 2, 1, 2, 64, EPI, 1>(a, s);
        else if (vm & 2) launch_gemm_pipe<4, 2, 1, 2, 32, EPI, 1>(a, s);
        else launch_gemm_pipe<4, 2, 1, 2, 32, EPI>(a, s);
    }
    else if (a.M >= 1024 && a.N >= 256 && a.K >= 1024) launch_gemm_pipe<2, 2, 2, 1, 32, EPI>(a, s);
    else if (a.M >= 1024 && a.N >= 256) {
        if (vm & 64) launch_gemm_pipe<4, 2, 1, 2, 32, EPI, 1>(a, s);
        else if (vm & 4) launch_gemm_pipe<2, 2, 1, 2, 32, EPI, 1>(a, s);
        else launch_gemm_pipe<2, 4, 1, 1, 32, EPI>(a, s);      // 64x128 on 8 waves of 32x32: out_proj / pw2 (-7 %)
    }
    else launch_gemm_pipe<2, 2, 1, 1, 32, EPI>(a, s);
}

void launch_gemm_smallm(const GemmArgs &a, int epi, hipStream_t s);   // kernels/gemm_smallm.hip

// The products whose LayerNorm can ride on the A staging of the fp32 tile kernel (GemmArgs::ln_stats): exactly the shapes launch_epi / launch_gemm
// send to the single-buffered 128 x 128 / BK 32 tile -- wide outputs of large batches (fc1, qkv) and the GLU product.
bool gemm_ln_stats_applies(const GemmArgs &a, int epi) {
    if (!a.ln_g || !a.ln_b || !a.ln_stats || a.a_bf16 || a.out_bf16 || a.fast_act || a.a_sigma || a.W_sig) return false;
    if (a.M <= kSmallMRows || a.K < 64 || a.K % 32 != 0 || (a.lda & 3) != 0 || (a.ldw & 3) != 0) return false;
    if (epi == EPI_GLU) return true;
    if (epi != EPI_NONE && epi != EPI_RELU && epi != EPI_SILU) return false;
    const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (a.K >= 1024 && a.K % 64 == 0 && tiles128 <= 256 && a.N >= 256) return false;     // (the long-K single-round tile has no LNA instantiation)
    return a.M >= 1024 && a.N >= 1024;
}

void launch_gemm(const GemmArgs &a, int epi, hipStream_t s) {
    if (a.ln_stats && !gemm_ln_stats_applies(a, epi)) { fprintf(stderr, "parakeet_amd: internal error: GemmArgs::ln_stats on a product the tile kernel does not fold it into\n"); abort(); }
    // up to a few hundred rows (streaming chunks, ONE utterance of up to a minute -- the reference's own benchmark protocol is batch 1):
    // one wavefront per 16x16 tile ((M/16)(N/16) independent waves) instead of a few dozen fat workgroups with a long K loop each.
    // Measured with tools/bench_reference_protocol.py: 10 s clip (M = 126) 6.3 -> 2.9 ms, 30 s (M = 376) 6.7 -> 4.3 ms per encoder pass.
    if (a.M <= kSmallMRows && a.K % 64 == 0) { launch_gemm_smallm(a, epi, s); return; }
    switch (epi) {
    case EPI_NONE: launch_epi<EPI_NONE>(a, s); break;
    case EPI_RELU: launch_epi<EPI_RELU>(a, s); break;
    case EPI_SILU: launch_epi<EPI_SILU>(a, s); break;
    case EPI_RESID: launch_epi<EPI_RESID>(a, s); break;
    case EPI_GLU:
        if (a.K < 64) launch_one<128, 128, EPI_GLU>(a, s);
        else if (a.ln_stats) launch_gemm_pipe<4, 2, 1, 2, 32, EPI_GLU, 1, true>(a, s);
        else if (gemm_variant_mask() & 8) launch_gemm_pipe<4, 2, 1, 2, 32, EPI_GLU, 1>(a, s);
        else launch_gemm_pipe<4, 2, 1, 2, 32, EPI_GLU>(a, s);
        break;
    default: break;
    }
}

// bf16 operands / fp32 accumulate (a.W points to bf16 weights [N][K]); K % 64 == 0
// The direct-to-LDS bf16 kernel (gemm_bf16_glds.hpp; activations already bf16 in HBM), 256x256 macro tiles.  Measured inside the engine on
// tdt-600m (profiles/r03_bf16_tile_ab.txt, PK_BF16_TILE in EXPERIMENTAL builds): it is ahead of the register-staged 128x128 kernel where one
// operand is large -- fc1 (N = 4096: 6.70 -> 6.60 ms per step) and fc2 (K = 4096: 6.51 -> 6.11) -- and behind on qkv (2.50 -> 2.70), the GLU
// product (1.75 -> 2.03) and the N = 1024 / K = 1024 products; 256x128 and 128x128-on-4-waves variants lose everywhere.  The second pass
// (tools/ubench/gemm_bf16_k.cpp) found why: the 256-row tile count of those products falls between two rounds of the 256 CUs; with the tile
// height chosen per product (below) the kernel is ahead everywhere.  Modes (EXPERIMENTAL builds, PK_BF16_TILE): 0 = off, 1 = that rule,
// 2 = 256x128 everywhere, 3 = 128x128 on 4 waves of 64x64, 4 = 256x256 everywhere, 5 = 192x256 everywhere.
// Persistent form of the direct-to-LDS kernel (gemm_bf16_glds.hpp: PERSIST; one workgroup per CU walks its tiles, the next tile's first K tile is
// requested under the epilogue).  Built and measured in round 4 (profiles/r04_bf16_persist_ab.txt, interleaved A/B on one box, tdt-600m 32 x 30 s):
// bit-for-bit the same results, qkv 2.45 -> 2.41 ms and GLU 1.82 -> 1.77 ms per step, but fc1 6.75 -> 7.17 (140 -> 150 us) and the step 27.62 ->
// 27.94 ms: inside one launch the next tile cannot start before `s_waitcnt vmcnt(0)` has ALSO drained the epilogue's stores (gfx950 counts loads
// and stores in one counter, and they complete out of order relative to each other), which costs more than the cold prologue and the
// re-dispatch it removes; and the epilogue, confined to one 64 KB buffer, needs 8 row bands instead of 4.  OFF by default; EXPERIMENTAL builds:
// PK_BF16_PERSIST=1 selects it.
// Round 5: 2 = the persistent form with the DIRECT register epilogue (gemm_bf16_glds.hpp: operands swapped in the MFMA, no LDS / barrier / vector load
// in the epilogue, both first K tiles of the next output tile requested before it) -- the production setting for the products without a residual
// read whose tiles exceed one round of the CUs: bit-identical results, fc1 142 -> 134 us, qkv / GLU -4 %, the tdt-600m step 27.11 -> 26.73 ms
// (profiles/r05_bf16_direct_epilogue_ab.txt, r05_bf16_ring_ab.txt: three interleaved repetitions each).  3 = the direct epilogue on one tile per
// workgroup (-1.5 % on fc1 alone).  4 = the continuous-stream kernel of gemm_bf16_ring.hpp (ring of four 32-k slots, counted vmcnt): correct and
// NOT faster than 2 -- the K loop is not bound by the DMA's latency (r05_bf16_ring_ab.txt; SQ counters r05_pmc_sq_600m_bf16_p*.md: the matrix pipe
// is busy 31-33 % of the launch in every form).  EXPERIMENTAL builds: PK_BF16_PERSIST selects.
static int bf16_glds_persist() {
#ifdef PK_EXPERIMENTAL
    static const int m = [] { const char *e = getenv("PK_BF16_PERSIST"); return e ? atoi(e) : 2; }();
    return m;
#else
    return 2;
#endif
}
// Instantiation flags of the direct-to-LDS bf16 kernels (gemm_bf16_glds.hpp / gemm_bf16_ring.hpp), EXPERIMENTAL builds: PK_BF16_FLAGS bit 1 = STAGGER,
// bit 2 = ASMFRAG (hand-counted fragment reads), bit 4 = row-block walk of the persistent form (XCD x owns a block of tile rows:
// fetch bytes per fc1 launch 176 -> 150 MB, time +1.5 %: off -- profiles/r05_bf16_rowblock_ab.txt), bit 8 = residual products accumulate ONTO the residual (GemmArgs::resid_init:
// the accumulators start from resid / alpha + bias, read beside the first K tiles, instead of 49 MB of residual reads next to the 49 MB of stores at
// the tail).  Measured (profiles/r05_bf16_resid_init_ab.txt): fc2 123 -> 144 us, out_proj / pw2 24.5 -> 34.5 us -- in the MFMA's C layout the residual
// arrives as 96 four-byte loads per wave in front of the first MFMA, dearer than the coalesced float4 reads of the LDS epilogue: off.
// Production: 2 (ASMFRAG: bit-identical, -0.7 % per tdt-600m step, profiles/r05_bf16_asmfrag_ab.txt).  With PK_BF16_PERSIST=4 the low two bits select
// the ring kernel's form instead: 0 plain, 1 STAGGER, 2 PHASED, 3 PHASED + s_setprio -- every one of them measured level with the persistent
// form on fc1 (117 us) and behind it on the step (profiles/r05_bf16_ring_ab.txt, r05_bf16_phased_ab.txt): three different K-loop schedules, one
// time -- the loop is bound by what a CU can pull from L2 into LDS (64 KB per 64-k tile at ~14 B/clock against the 32 B/clock the MFMAs could use),
// not by how the pulls are scheduled.
static int bf16_glds_flags() {
#ifdef PK_EXPERIMENTAL
    static const int m = [] { const char *e = getenv("PK_BF16_FLAGS"); return e ? atoi(e) : 18; }();
    return m;
#else
    return 18;
#endif
}
static int bf16_glds_mode() {
#ifdef PK_EXPERIMENTAL
    static const int m = [] { const char *e = getenv("PK_BF16_TILE"); return e ? atoi(e) : 1; }();
    return m;
#else
    return 1;
#endif
}

// the one rule both launch_bf16_epi and gemm_bf16_blocked_handoff apply: the product runs on the tile-height-per-product direct-to-LDS kernels
static bool bf16_glds_rule(int M, int N, int K) {
    return bf16_glds_mode() == 1 && M >= 8192 && N >= 512 && K >= 128 && (int64_t)N * K >= (int64_t)1024 * 1024 && (N & 3) == 0;
}
bool gemm_bf16_blocked_handoff(int M, int N, int K, int epi, bool producer) {
    if (!bf16_glds_rule(M, N, K) || (N % 16) != 0 || (K % 64) != 0) return false;
    if (!producer) return bf16_glds_persist() != 4;                 // every gemm_bf16_glds_kernel form reads the blocked A (the ring kernel does not)
    const int p = bf16_glds_persist();
    return (p == 2 || p == 3) && epi != EPI_RESID && (int64_t)(M + 31) * N < ((int64_t)1 << 31);   // the register epilogue writes it (callers set fast_act: bf16 mode)
}
[[noreturn]] static void bf16_layout_bug(const char *what) {
    fprintf(stderr, "parakeet_amd: internal error: %s (GemmArgs::out_blocked / a_blocked on a kernel that does not implement it)\n", what);
    abort();
}

template <int EPI, bool A16>
static void launch_bf16_epi(const GemmArgs &a, hipStream_t s) {
    if constexpr (A16) {
        const int mode = bf16_glds_mode();
        if (mode && a.M >= 2048 && a.N >= 512 && a.K >= 128 && (a.lda % 8) == 0 && (a.ldw % 8) == 0 && a.remap_rows == 0 && (a.ldo & 3) == 0 && (a.N & 3) == 0) {
            if (mode != 1 && a.out_blocked) bf16_layout_bug("blocked output outside the production tile rule");
            if (mode == 3) { launch_gemm_bf16_glds<2, 2, 2, 2, EPI>(a, s); return; }
            if (mode == 2) { launch_gemm_bf16_glds<4, 2, 2, 2, EPI>(a, s); return; }
            if (mode == 4) { launch_gemm_bf16_glds<4, 2, 2, 4, EPI>(a, s); return; }
            if (mode == 5) { launch_gemm_bf16_glds<2, 4, 3, 2, EPI>(a, s); return; }
            // One tile per CU and round (128 KB of LDS): the kernel's time is rounds x (tile's K loop + ~12 us of prologue / epilogue), so the tile
            // HEIGHT is chosen per product for the fewest, fullest rounds of the 256 CUs (tools/ubench/gemm_bf16_k.cpp, profiles/r03_gemm_bf16_k.txt:
            // main loop 1.3 PF on either tile).  tdt-600m (M = 12032): fc1 (N 4096) 752 tiles of 256 rows = 2.94 rounds; fc2 / out / pw2 (N 1024) 252
            // tiles of 192 rows = ONE round (188 of 256 rows leave 68 CUs idle: 113 -> 99 us); qkv (N 3072) 756 of 192 = 2.95 rounds (564 of 256 = 2.2).
            if (mode == 1 && a.M >= 8192 && (int64_t)a.N * a.K >= (int64_t)1024 * 1024) {
                constexpr int NOUT = (EPI == EPI_GLU) ? 128 : 256;
                auto est = [&](int R) {
                    const int64_t tiles = (int64_t)((a.M + R - 1) / R) * ((a.N + NOUT - 1) / NOUT);
                    return (double)((tiles + 255) / 256) * ((double)R * a.K * 1.008e-4 + 12.0);
                };
                const bool tall = !(est(192) < est(256));
#ifdef PK_EXPERIMENTAL                                             // (measured level with / behind the persistent form: not in the production library)
                if constexpr (EPI != EPI_RESID) {
                    // the continuous-stream form (gemm_bf16_ring.hpp): more tiles than CUs, no residual read
                    constexpr int NO = (EPI == EPI_GLU) ? 128 : 256;
                    const int64_t tiles = (int64_t)((a.M + (tall ? 256 : 192) - 1) / (tall ? 256 : 192)) * ((a.N + NO - 1) / NO);
                    if (bf16_glds_persist() == 4 && tiles > 256 && gemm_bf16_ring_applies<EPI>(a)) {
                        if (tall) launch_gemm_bf16_ring<4, 2, 2, 4, EPI>(a, s, bf16_glds_flags() & 3);
                        else launch_gemm_bf16_ring<2, 4, 3, 2, EPI>(a, s, bf16_glds_flags() & 3);
                        return;
                    }
                }
#endif
#ifdef PK_EXPERIMENTAL
                if constexpr (EPI == EPI_RESID) {
                    if ((bf16_glds_flags() & 8) != 0 && a.alpha != 0.0f && a.remap_rows == 0) {                      // (bit 8: measured 16-42 % slower per product: off)
                        GemmArgs b = a;
                        b.resid_init = 1;
                        if (!tall) launch_gemm_bf16_glds<2, 4, 3, 2, EPI>(b, s, bf16_glds_persist(), (bf16_glds_flags() & 1) != 0, (bf16_glds_flags() & 2) != 0, (bf16_glds_flags() & 4) != 0);
                        else launch_gemm_bf16_glds<4, 2, 2, 4, EPI>(b, s, bf16_glds_persist(), (bf16_glds_flags() & 1) != 0, (bf16_glds_flags() & 2) != 0, (bf16_glds_flags() & 4) != 0);
                        return;
                    }
                }
#endif
                // bit 16 (round 6, production): residual products on the register epilogue with the accumulators started from the residual (gemm_bf16_glds.hpp)
                const bool rd = (bf16_glds_flags() & 16) != 0;
                if (!tall) launch_gemm_bf16_glds<2, 4, 3, 2, EPI>(a, s, bf16_glds_persist(), (bf16_glds_flags() & 1) != 0, (bf16_glds_flags() & 2) != 0, (bf16_glds_flags() & 4) != 0, rd);
                else launch_gemm_bf16_glds<4, 2, 2, 4, EPI>(a, s, bf16_glds_persist(), (bf16_glds_flags() & 1) != 0, (bf16_glds_flags() & 2) != 0, (bf16_glds_flags() & 4) != 0, rd);
                return;
            }
        }
    }
    if (a.out_blocked || a.a_blocked) bf16_layout_bug("blocked activation layout requested for the register-staged bf16 kernel");
    // round 2: the staging stores decide the rate of this kernel.  As 16-byte ds_write_b128 the 128x128 tile ran at 320-340 TF and 256x256
    // macro tiles were the way to 450 (profiles/r02_gemm_bf16_tiles.txt); the SAME 16 bytes written as a ds_write2_b64 pair
    // (gemm_bf16.hpp, lstore) take the 128x128 tile to 580 TF in the sweep and 510-600 TF in the engine (profiles/r02_gemm_bf16_ablation.txt)
    // -- the 16-byte LDS store is pathologically slow next to fragment reads on gfx950, as the fp32 kernel had already shown.
    if constexpr (EPI == EPI_GLU) {
        launch_gemm_bf16_t<4, 2, 1, 2, EPI_GLU, A16>(a, s);
    } else {
        if (a.M >= 1024 && (a.N >= 1024 || (a.M >= 65536 && a.N >= 256))) launch_gemm_bf16_t<4, 2, 1, 2, EPI, A16>(a, s);      // 128x128 on 8 waves of 32x64 (also the 1.4 M-row subsampling products)
        else if (a.M >= 1024 && a.N >= 256) launch_gemm_bf16_t<2, 2, 2, 1, EPI, A16>(a, s);
        else launch_gemm_bf16_t<2, 2, 1, 1, EPI, A16>(a, s);
    }
}
template <bool A16>
static void launch_gemm_bf16_a(const GemmArgs &a, int epi, hipStream_t s) {
    switch (epi) {
    case EPI_NONE: launch_bf16_epi<EPI_NONE, A16>(a, s); break;
    case EPI_RELU: launch_bf16_epi<EPI_RELU, A16>(a, s); break;
    case EPI_SILU: launch_bf16_epi<EPI_SILU, A16>(a, s); break;
    case EPI_RESID: launch_bf16_epi<EPI_RESID, A16>(a, s); break;
    case EPI_GLU: launch_bf16_epi<EPI_GLU, A16>(a, s); break;
    default: break;
    }
}
void launch_gemm_bf16(const GemmArgs &a, int epi, hipStream_t s) {
    // (a bf16 output goes through the wide epilogue only -- row-major, 4-column groups: Model::run_gemm checks)
    // a handful of rows (the streaming encoder's chunks): the weight-stream kernel of gemm_smallm_bf16.hip
    if (gemm_smallm_bf16_applies(a, epi)) { launch_gemm_smallm_bf16(a, epi, s); return; }
    if (a.a_bf16) launch_gemm_bf16_a<true>(a, epi, s);
    else launch_gemm_bf16_a<false>(a, epi, s);
}

double gemm_flops(const GemmArgs &a, int epi) {
    return 2.0 * (double)a.M * (double)a.N * (double)a.K * (epi == EPI_GLU ? 2.0 : 1.0);
}

}  // namespace pk
