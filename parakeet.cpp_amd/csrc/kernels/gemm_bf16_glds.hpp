// parakeet.cpp_amd/csrc/kernels/gemm_bf16_glds.hpp -- bf16 MFMA GEMM with DIRECT-TO-LDS staging (gfx950 global_load_lds_dwordx4) on large
// tiles, for the products of the tolerance-class mode whose activations already live in HBM as bf16 (GemmArgs::a_bf16).
//
// out[M][N] = epi(A16[M][K] * W16[N][K]^T + bias), fp32 accumulation on v_mfma_f32_32x32x16_bf16 -- the arithmetic of gemm_bf16.hpp.
// Round 2 located that kernel's loss (profiles/r02_gemm_bf16_ablation.txt): MFMAs + barriers alone run at 1.0-1.7 PF, the
// global -> VGPR -> ds_write staging takes half of the remaining time, and a 128x128 tile per 8 waves re-reads its operands from L2 twice
// as often as a 256x256 one.  This kernel removes the staging instructions altogether:
//  * every lane issues global_load_lds_dwordx4: 16 bytes (8 consecutive k of one tile row) go from global memory straight into LDS -- no
//    staging VGPRs, no ds_write, no conversion (both operands are bf16 in HBM).  One wave instruction fills one 1 KB block = 8 tile rows of
//    BK = 64 k.  The hardware puts lane q's 16 bytes at (block base + 16 q); WHICH 16-byte chunk of the row lands there is chosen through the
//    global address the lane reads: physical chunk p of row r holds logical chunk p ^ ((r >> 1) & 7).  With that XOR swizzle the
//    ds_read_b128 fragment reads (lane = row r of a 32-row operand tile, logical chunk 2 s + (lane >> 5)) are conflict-free for the
//    instruction's four 16-lane groups (MI355X_MICROARCH.md section LDS: groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- within a group
//    the rows of equal parity have distinct (r >> 1) & 7).  The LDS image is unpadded (the DMA destination is lane-linear).
//  * 256x256 macro tiles on 8 waves of 64x128 (wide outputs) or 256x128 / 128x128 on waves of 64x64: 6-8 fragment reads feed 8-4 MFMAs per
//    k-step instead of 3 feeding 2.
//  * two LDS buffers; the DMA of K tile kt+2 is issued right after the barrier that frees the buffer of tile kt and has a whole K tile of
//    MFMAs to land (s_waitcnt vmcnt(0) + the next barrier publish it) -- the schedule of tools/ubench/gemm_dma.hpp (round 2, fp32).
//  * the epilogue is gemm_pipe.hpp's (accumulators -> LDS -> 4 consecutive columns per thread), in 64-row bands, the residual rows of a band
//    requested at the start of its pass.
#ifndef PK_GEMM_BF16_GLDS_HPP
#define PK_GEMM_BF16_GLDS_HPP
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "../pk_devmath.h"
#include "kernels.hpp"
#include "gemm_pipe.hpp"
#include "gemm_bf16.hpp"

namespace pk {

// Phase stamps for tools/ubench/gemm_bf16_trace.cpp (-DGL_TRACE): the shader clock of lane 0 of wave 0 of every workgroup at the phase
// boundaries of its output tiles -- [workgroup][tile][GL_TRACE_SLOTS]: 0 = tile start, 1 = first fragments readable (prologue wait + barrier),
// 2 + kt = K tile kt's barrier passed, 2 + nk = K loop done, 3 + nk = epilogue issued.  Production builds: nothing.
#ifdef GL_TRACE
constexpr int GL_TRACE_SLOTS = 72, GL_TRACE_TILES = 4;
__device__ unsigned long long *gl_trace;
#define GL_STAMP(tile, i) do { if (gl_trace && threadIdx.x == 0 && (tile) < GL_TRACE_TILES && (i) < GL_TRACE_SLOTS) \
        gl_trace[((size_t)blockIdx.x * GL_TRACE_TILES + (tile)) * GL_TRACE_SLOTS + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GL_STAMP(tile, i) do { } while (0)
#endif

// DIRECT epilogue (round 5): the accumulators go from registers to global memory -- no LDS, no barrier, no load.  The K loop feeds the MFMA
// with the operands SWAPPED (acc = mfma(W fragment, A fragment)): the same products summed in the same k order, so every result is bit for
// bit the other form's, but the C/D layout now gives lane (lr = lane & 31, h = lane >> 5) output ROW lr of its 32-row tile and, per group
// q = reg >> 2, the FOUR CONSECUTIVE columns 8 q + 4 h .. + 3 of the 32-column tile: a float4, or 4 bf16 = 8 bytes; v_permlane32_swap between
// the groups q and q + 1 makes that 16 bytes per lane (lanes 0-31: columns 8 q .. 8 q + 7, lanes 32-63: 8 (q + 1) .. + 7 of the same row;
// cdna_hip_programming.md T21).  The bias of a column group is wave-uniform up to h: eight consecutive floats read through the scalar cache
// and selected by h, so the epilogue issues no vector load and its first instruction does not wait for the DMA of the next tile -- which the
// persistent loop has ALREADY requested into both staging buffers (nothing of the epilogue touches LDS).  bf16 or fp32 rows out; EPI_NONE,
// EPI_RELU, EPI_SILU, EPI_GLU (the residual epilogue would need vector loads: it keeps the LDS form).
// The run-time switches of the epilogue (hardware exp / rcp activations, bf16 rows out, blocked hand-off) are TEMPLATE parameters of the body
// and dispatched once per tile: inside the fully unrolled body they were a branch per four values, 64-bit address arithmetic per store and
// ~25 000 clocks of epilogue per 256 x 256 tile -- a quarter of the tile's time (tools/ubench/gemm_bf16_trace.cpp, profiles/r05_bf16_fc1_phase_trace.txt).
template <int WGM, int WGN, int TM, int TN, int EPI, bool FAST, bool OUT16, bool BLOCKED>
__device__ __forceinline__ void gl_epilogue_direct_body(const GemmArgs &g, bg_f32x16 (&acc)[TM][TN], int m0, int n0) {
    static_assert(EPI != EPI_RESID, "the residual epilogue keeps the LDS form");
    constexpr bool GLU = EPI == EPI_GLU;
    constexpr int WM = TM * 32, WN = TN * 32, TNO = GLU ? TN / 2 : TN, WNO = GLU ? WN / 2 : WN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WGN, wn = wave % WGN, lr = lane & 31, h = lane >> 5;
    const int cw = __builtin_amdgcn_readfirstlane(n0 + wn * WNO);   // first output column of this wave (wave-uniform)
    // the bias through the SCALAR cache: eight consecutive floats per load from a constant-address-space pointer at a wave-uniform index
    // (s_load_dwordx8); the per-half select picks among loaded SGPRs, so the compiler cannot fold it into a per-lane address
    typedef float f32x8_ __attribute__((ext_vector_type(8)));
    typedef const f32x8_ __attribute__((address_space(4))) *cf8p;
    const bool has_bias = g.bias != nullptr;
    auto sld8 = [&](int idx) {
        f32x8_ z = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        return has_bias ? *(cf8p)(const void *)(g.bias + idx) : z;
    };
    // per accumulator row block: this lane's output row, whether it exists, and the element offset of its first column of the wave
    bool rok[TM];
    int roff[TM];                                                   // (32-bit element offsets: the launcher checks the tensor stays below 2^31 elements)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WM + i * 32 + lr;
        rok[i] = row < g.M;
        if constexpr (BLOCKED) roff[i] = ((row >> 5) * ((int)g.ldo >> 4) + (cw >> 4)) * 512 + (row & 31) * 16 + 8 * h;
        else roff[i] = row * (int)g.ldo + cw + (OUT16 ? 8 : 4) * h;
    }
    __bf16 *out16 = reinterpret_cast<__bf16 *>(g.out);
#pragma unroll
    for (int j = 0; j < TNO; ++j) {
#pragma unroll
        for (int q = 0; q < 4; q += 2) {                            // column groups q, q + 1 of tile j: columns 8 q .. 8 q + 15
            const int cb = cw + j * 32 + 8 * q;                     // wave-uniform; the whole 16-column span is inside N or outside (N % 16 == 0: launcher)
            if (cb >= g.N) continue;
            float bs[2][4], bg[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x8_ b8 = sld8(cb + 8 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[u][e] = h ? b8[4 + e] : b8[e];
                if constexpr (GLU) {
                    const f32x8_ g8 = sld8(g.N + cb + 8 * u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bg[u][e] = h ? g8[4 + e] : g8[e];
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float v[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = acc[i][j][4 * (q + u) + e] + bs[u][e];
                    if constexpr (EPI == EPI_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] = v[u][e] > 0.0f ? v[u][e] : 0.0f;
                    } else if constexpr (EPI == EPI_SILU) {
                        if constexpr (FAST) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[u][e] = fast_siluf(v[u][e]);
                        } else {
                            dsilu4(v[u]);
                        }
                    } else if constexpr (GLU) {
                        float gt[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) gt[e] = acc[i][j + TN / 2][4 * (q + u) + e] + bg[u][e];
                        if constexpr (FAST) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) gt[e] = fast_sigmoidf(gt[e]);
                        } else {
                            dsigmoid4(gt);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] = v[u][e] * gt[e];
                    }
                }
                constexpr int cq = 0;
                (void)cq;
                if constexpr (OUT16) {
                    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
                    unsigned pk[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const bf16x2_ a = {(__bf16)v[u][0], (__bf16)v[u][1]}, b = {(__bf16)v[u][2], (__bf16)v[u][3]};
                        pk[u][0] = __builtin_bit_cast(unsigned, a);
                        pk[u][1] = __builtin_bit_cast(unsigned, b);
                    }
                    // upper half of group q <-> lower half of group q + 1: lanes 0-31 hold columns 8 q .. 8 q + 7, lanes 32-63 8 (q + 1) .. + 7
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    if (rok[i]) {
                        uint4 o;
                        o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                        // blocked hand-off (GemmArgs::out_blocked): block (row / 32, cb / 16) of 32 x 16 elements, row-major inside -- the
                        // wave's 64 stores of this instruction fill exactly one contiguous KB.  (compile-time offset from the row's base)
                        const int rel = BLOCKED ? (2 * j + q / 2) * 512 : j * 32 + 8 * q;
                        *reinterpret_cast<uint4 *>(out16 + roff[i] + rel) = o;
                    }
                } else if (rok[i]) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        *reinterpret_cast<float4 *>(g.out + roff[i] + j * 32 + 8 * (q + u)) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
                }
            }
        }
    }
}
// (the direct form exists for the hardware-exp activations only -- what the bf16 mode always asks for, GemmArgs::fast_act: the launcher sends a
// product with the polynomial activations to the LDS epilogue)
template <int WGM, int WGN, int TM, int TN, int EPI>
__device__ __forceinline__ void gl_epilogue_direct(const GemmArgs &g, bg_f32x16 (&acc)[TM][TN], int m0, int n0) {
    if (g.out_bf16) {
        if (g.out_blocked) gl_epilogue_direct_body<WGM, WGN, TM, TN, EPI, true, true, true>(g, acc, m0, n0);
        else gl_epilogue_direct_body<WGM, WGN, TM, TN, EPI, true, true, false>(g, acc, m0, n0);
    } else {
        gl_epilogue_direct_body<WGM, WGN, TM, TN, EPI, true, false, false>(g, acc, m0, n0);
    }
}

// Residual products with the register epilogue (round 6; round-5 verdict item 2: out_proj / pw2 42 us against a 27 us K loop, fc2 113 against 107).
// out = resid + alpha (A W^T + bias) is formed as alpha (resid / alpha + bias + A W^T): in the swapped-operand accumulator layout (above) a lane owns output row lr
// and, per register quad, FOUR CONSECUTIVE columns -- the residual arrives as 16-byte loads (24 per lane for a 192 x 256 tile on 8 waves), requested at the START
// of the tile next to the first two K tiles' DMA, and the accumulators are initialised with fma(resid, 1 / alpha, bias) (the bias through the scalar cache).  After
// the K loop the epilogue is 24 multiplications by alpha and 16-byte stores from registers: no LDS round trip, no barrier, no load -- the three-band LDS
// epilogue with its residual reads at the start of every band was a third of the narrow products' time.  Round 5 had tried the accumulate-onto-the-residual idea
// in the UN-swapped layout (GemmArgs::resid_init: the residual as 96 four-byte loads per wave in front of the first MFMA) and measured it 16-42 % slower; this is
// the same arithmetic with 16-byte accesses.  alpha = 0.5 / 1: the scaling by 1 / alpha is exact; the K partial sums are added onto a value of the residual's
// magnitude, i.e. rounded at ITS ulp (tolerance-class mode only: tests/test_gpu_bf16.py bounds it).  In place (out == resid): every lane reads and later writes its own elements.
template <int WGM, int WGN, int TM, int TN>
__device__ __forceinline__ void gl_resid_init(const GemmArgs &g, bg_f32x16 (&acc)[TM][TN], int m0, int n0) {
    constexpr int WM = TM * 32, WN = TN * 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WGN, wn = wave % WGN, lr = lane & 31, h = lane >> 5;
    const int cw = __builtin_amdgcn_readfirstlane(n0 + wn * WN);
    typedef float f32x8_ __attribute__((ext_vector_type(8)));
    typedef const f32x8_ __attribute__((address_space(4))) *cf8p;
    const bool has_bias = g.bias != nullptr;
    const float inv_alpha = 1.0f / g.alpha;
    int roff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int row = m0 + wm * WM + i * 32 + lr;
        row = row < g.M ? row : g.M - 1;                           // (rows past the end: a valid address, never stored)
        roff[i] = row * (int)g.ldr + cw + 4 * h;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = cw + j * 32 + 8 * q;                     // wave-uniform; N % 16 == 0 (launcher): the 8-column group is inside N or outside
            float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (cb < g.N && has_bias) {
                const f32x8_ b8 = *(cf8p)(const void *)(g.bias + cb);
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[e] = h ? b8[4 + e] : b8[e];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (cb < g.N) r = *reinterpret_cast<const float4 *>(g.resid + roff[i] + j * 32 + 8 * q);
                acc[i][j][4 * q] = __builtin_fmaf(r.x, inv_alpha, bs[0]); acc[i][j][4 * q + 1] = __builtin_fmaf(r.y, inv_alpha, bs[1]);
                acc[i][j][4 * q + 2] = __builtin_fmaf(r.z, inv_alpha, bs[2]); acc[i][j][4 * q + 3] = __builtin_fmaf(r.w, inv_alpha, bs[3]);
            }
        }
}
template <int WGM, int WGN, int TM, int TN>
__device__ __forceinline__ void gl_epilogue_resid_direct(const GemmArgs &g, bg_f32x16 (&acc)[TM][TN], int m0, int n0) {
    constexpr int WM = TM * 32, WN = TN * 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WGN, wn = wave % WGN, lr = lane & 31, h = lane >> 5;
    const int cw = __builtin_amdgcn_readfirstlane(n0 + wn * WN);
    const float alpha = g.alpha;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WM + i * 32 + lr;
        if (row >= g.M) continue;
        float *orow = g.out + row * (int)g.ldo + cw + 4 * h;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (cw + j * 32 + 8 * q >= g.N) continue;
                *reinterpret_cast<float4 *>(orow + j * 32 + 8 * q) =
                    make_float4(acc[i][j][4 * q] * alpha, acc[i][j][4 * q + 1] * alpha, acc[i][j][4 * q + 2] * alpha, acc[i][j][4 * q + 3] * alpha);
            }
    }
}

// PERSIST (round 4): ONE workgroup per CU walks its tiles inside the launch.  Measured in round 3 (tools/ubench/gemm_bf16_k.cpp): the K loop runs
// at 1.25-1.3 PF, the products of this model lose ~14 us per ROUND of tiles -- a cold two-tile DMA prologue on every CU at once, the epilogue,
// the re-dispatch -- on K loops of only 16 tiles.  Here the first K tile of tile i+1 is requested (DMA into the staging buffer the last K tile
// of tile i did not use) BEFORE the epilogue of tile i, which turns its accumulators row-major through the OTHER buffer only (64 KB: bands of
// 32 rows); the second K tile follows right after the epilogue, and the K loop of tile i+1 starts on data that has long landed.  XCD x owns
// the same contiguous range of tiles as in the one-tile-per-workgroup launch, dealt round-robin to its 32 workgroups.
// PERSIST + DIRECT (round 5): with the register epilogue nothing after the last K tile's barrier touches LDS, so BOTH first K tiles of the next
// output tile are requested before the epilogue starts; the wait at the top of the loop then covers DMA that landed microseconds ago and the
// epilogue's own stores (one counter for loads and stores on gfx950), instead of a cold two-tile prologue per round of tiles.
// ASMFRAG (round 5): the fragment reads are inline-asm ds_read_b128 with hand-counted s_waitcnt lgkmcnt(N).  Left to the compiler, two of the
// four MFMA groups of a K tile were preceded by `s_waitcnt lgkmcnt(0)` -- its wait for the fragments loaded one sub-step earlier also drained
// the six reads just issued for the NEXT sub-step (it does not count across the loop's back edge), i.e. a whole LDS round trip in front of the
// MFMAs it was meant to overlap (ISA of round 4's kernel; tools/ubench/gemm_bf16_trace.cpp: 4500 clocks per K tile against 2048 of MFMA work).
// Here the reads of sub-step s+1 stay in flight (lgkmcnt(TM + TN)) while the MFMAs of sub-step s issue; LDS reads return in order and no
// scalar load is outstanding inside the loop.  The destination registers are tied to the wait by empty "+v" statements (cdna_hip_programming.md
// 5.7 items 1 and 3), sched_barrier(0) keeps the MFMA builtins below them.
template <int WGM, int WGN, int TM, int TN, int EPI, bool PERSIST = false, bool DIRECT = false, bool STAGGER = false, bool ASMFRAG = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_glds_kernel(GemmArgs g, int tiles_n, int n_tiles, int rowblock = 0) {
    constexpr int BK = 64, NSUB = BK / 16;                          // bf16 elements per tile row; MFMA k-steps per K tile
    constexpr int NT = 64 * WGM * WGN, NW = WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int BUF = (BM + BN) * BK;                             // bf16 elements per staging buffer (unpadded: the swizzle spreads the banks)
    constexpr int NBLK = (BM + BN) / 8, NBPW = NBLK / NW;           // 1 KB blocks of 8 rows per K tile / per wave
    static_assert(NBLK % NW == 0, "blocks must split evenly over the waves");
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    static_assert(EPI != EPI_GLU || (TN % 2 == 0), "GLU needs an even number of column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char gl_smem_raw[];
    __bf16 *smem = reinterpret_cast<__bf16 *>(gl_smem_raw);
    float *smem_f = reinterpret_cast<float *>(gl_smem_raw);
    (void)NT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WGN, wn = wave % WGN;
    // STAGGER (round 5): waves w and w + NW / 2 share a SIMD; the second half requests its DMA pieces after the MFMA group that follows the
    // barrier instead of before it, so that one wave of a SIMD issues LDS-DMA (60-185 clocks per 1 KB piece) while the other feeds the matrix
    // pipe.  A template parameter: as a run-time flag the extra branches in the K loop cost more than the stagger gains
    // (profiles/r05_bf16_stagger_runtime_flags_ab.txt).
    const bool late = STAGGER && wv >= NW / 2;
    const int nk = g.K / BK;
    const __bf16 *A16 = reinterpret_cast<const __bf16 *>(g.A);
    const __bf16 *W16 = reinterpret_cast<const __bf16 *>(g.W);

    // tile `bid` (after the XCD remap) -> origin, grouped tile order (gemm_bf16.hpp): GROUPM tile rows down before the next tile column
    auto tile_origin = [&](int bid, int &m0, int &n0) {
        constexpr int GROUPM = (BM >= 256) ? 4 : 8;
        const int tiles_m = n_tiles / tiles_n, per_group = GROUPM * tiles_n;
        const int grp = bid / per_group, first_m = grp * GROUPM;
        const int gsz = (tiles_m - first_m) < GROUPM ? (tiles_m - first_m) : GROUPM;
        const int in = bid - grp * per_group;
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * NOUT;
    };
    // XCD-aware bijective remap (block b runs on XCD b % 8): XCD x gets a contiguous range of tiles [x_first, x_first + x_count)
    const int xq = n_tiles >> 3, xr = n_tiles & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int x_first = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, x_count = xq + (xcd < xr ? 1 : 0);
    const int per_xcd = PERSIST ? (int)(gridDim.x >> 3) : 1;        // workgroups per XCD walking that range (persistent: stride)
    int loc = idx;                                                  // index inside the XCD's range
    // rowblock (persistent walk, round 5): XCD x owns a BLOCK OF TILE ROWS (tiles_m split as evenly as it goes) and walks it row-fastest -- its
    // A rows (a few MB) stay in ITS L2 for the whole launch and only W streams through, instead of every XCD streaming (nearly) all of both.
    // Measured (profiles/r05_bf16_rowblock_ab.txt, fc1 of tdt-600m): L2 fetch bytes per launch 176 -> 150 MB, bit-identical, and 1.5 % SLOWER
    // (119.3 -> 121.2 us) -- the K loop is not waiting for the fabric behind L2.  Off in production (the launcher's rowblock_ok).
    const int tiles_m_all = n_tiles / tiles_n;
    const int rq = tiles_m_all >> 3, rr = tiles_m_all & 7;
    const int r_first = xcd < rr ? xcd * (rq + 1) : rr * (rq + 1) + (xcd - rr) * rq, r_cnt = rq + (xcd < rr ? 1 : 0);
    const bool rb = PERSIST && rowblock != 0;
    const int x_cnt = rb ? r_cnt * tiles_n : x_count;
    auto origin_of = [&](int l, int &m0, int &n0) {
        if (rb) { m0 = (r_first + l % r_cnt) * BM; n0 = (l / r_cnt) * NOUT; }
        else tile_origin(x_first + l, m0, n0);
    };
    if (loc >= x_cnt) return;                                       // (one-tile launches have exactly n_tiles workgroups)

    // DMA sources: wave w fills blocks w, w + NW, ...; lane q of block b supplies (row 8 b + q / 8, logical chunk (q % 8) ^ ((row >> 1) & 7)).
    // Rows 0 .. BM-1 of the stacked tile are A rows, BM .. BM+BN-1 are W rows.
    auto set_src = [&](int m0, int n0, const __bf16 *(&src)[NBPW]) {
#pragma unroll
        for (int i = 0; i < NBPW; ++i) {
            const int b = wv + NW * i;
            const int row = 8 * b + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            if (row < BM) {
                int gr = m0 + row;
                gr = gr < g.M ? gr : g.M - 1;
                // blocked hand-off (GemmArgs::a_blocked): chunk c of row gr, K tile 0 = block (gr / 32, c / 2), row gr % 32, half c % 2
                if (g.a_blocked) src[i] = A16 + ((int64_t)(gr >> 5) * (g.lda >> 4) + (c >> 1)) * 512 + (gr & 31) * 16 + 8 * (c & 1);
                else src[i] = A16 + (int64_t)gr * g.lda + 8 * c;
            } else {
                const int v = row - BM;
                int wr;
                if constexpr (EPI == EPI_GLU) {
                    constexpr int HT = TN / 2;     // tiles [0,HT) = value half, [HT,TN) = gate half of the SAME output columns
                    const int vw = v / WN, rem = v % WN, tn = rem >> 5, cc = rem & 31;
                    int col = n0 + vw * (WN / 2) + (tn % HT) * 32 + cc;
                    col = col < g.N ? col : g.N - 1;
                    wr = (tn / HT) * g.N + col;
                } else {
                    wr = n0 + v;
                    wr = wr < g.N ? wr : g.N - 1;
                }
                src[i] = W16 + (int64_t)wr * g.ldw + 8 * c;
            }
        }
    };
    // elements between consecutive K tiles of a source: BK in a row-major operand, 4 blocks of 512 in the blocked A (64 k = 4 column groups of 16)
    const int a_kstep = g.a_blocked ? 4 * 512 : BK;
    auto dma = [&](const __bf16 *const (&src)[NBPW], int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NBPW; ++i) {
            __bf16 *dst = smem + buf * BUF + (wv + NW * i) * 512;                    // 1 KB = 512 bf16 per block; wave-uniform
            const bool is_a = 8 * (wv + NW * i) < BM;                                // (wave-uniform: a block is 8 rows of A or of W; BM % 8 == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + kt * (is_a ? a_kstep : BK)),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    bg_f32x16 acc[TM][TN];
    // fragment addresses: operand-tile row r = tile base (a multiple of 32) + (lane & 31): element offset r * 64 + ((2 s + h) ^ x) * 8 with
    // x = (r >> 1) & 7 = ((lane & 31) >> 1) & 7 -- one swizzle value per lane for every tile
    const int h = lane >> 5, fx = ((lane & 31) >> 1) & 7;
    const int fa_base = (wm * WM + (lane & 31)) * BK, fb_base = (BM + wn * WN + (lane & 31)) * BK;
    bg_bf16x8 fa[2][TM], fb[2][TN];
    const unsigned lds0 = (unsigned)(size_t)smem;                   // low 32 bits of a flat LDS address = the LDS offset
    auto fragload = [&](int buf, int s, int slot) {
        if constexpr (ASMFRAG) {
            const unsigned e = (unsigned)(buf * BUF + (((2 * s + h) ^ fx) << 3));
            const unsigned aa = lds0 + 2u * (e + (unsigned)fa_base), ab = lds0 + 2u * (e + (unsigned)fb_base);
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[slot][i]) : "v"(aa), "n"(i * 32 * BK * 2));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[slot][j]) : "v"(ab), "n"(j * 32 * BK * 2));
        } else {
            const __bf16 *base = smem + buf * BUF + (((2 * s + h) ^ fx) << 3);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const bg_bf16x8 *>(base + fa_base + i * 32 * BK);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const bg_bf16x8 *>(base + fb_base + j * 32 * BK);
        }
    };
    // ASMFRAG: the fragments of `slot` are needed now; the TM + TN reads issued after them (frag_ready_one) / none (frag_ready_none) may stay in flight
    auto frag_pin = [&](int slot) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[slot][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[slot][j]));
    };
    auto frag_ready_one = [&](int slot) {
        if constexpr (ASMFRAG) {
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
            frag_pin(slot);
        }
    };
    auto frag_ready_none = [&](int slot) {
        if constexpr (ASMFRAG) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            frag_pin(slot);
        }
    };
    auto mma = [&](int slot) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (DIRECT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[slot][j], fa[slot][i], acc[i][j], 0, 0, 0);   // C^T: lane = output row
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
            }
    };
#define GL_SB() __builtin_amdgcn_sched_barrier(0)
#define GL_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)           /* vmcnt(0): this wave's LDS-DMA loads have landed */
#define GL_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)         /* lgkmcnt(0) only: this wave's LDS reads have returned */

    const __bf16 *src[NBPW];
    int m0, n0;
    origin_of(loc, m0, n0);
    set_src(m0, n0, src);
    int cur = 0;
    int tr_tile = 0;
    (void)tr_tile;
    dma(src, 0, 0);
    if (nk > 1) dma(src, 1, 1);
    for (;;) {
        GL_STAMP(tr_tile, 0);
        bool resid_in_acc = false;                                  // (EXPERIMENTAL builds only: measured slower, gemm.hip bf16_glds_flags bit 8)
#ifdef PK_EXPERIMENTAL
        if constexpr (EPI == EPI_RESID && !DIRECT && !PERSIST) resid_in_acc = g.resid_init != 0;
#endif
        if constexpr (DIRECT && EPI == EPI_RESID) {
            gl_resid_init<WGM, WGN, TM, TN>(g, acc, m0, n0);        // (requested behind the first two K tiles' DMA: lands under the prologue wait)
        } else if (resid_in_acc) {
            // out = resid + alpha (A W^T + bias) accumulated ONTO the residual (GemmArgs::resid_init, tolerance-class mode): the accumulators start
            // from resid / alpha + bias -- requested here, next to the first K tiles' DMA, instead of 49 MB of residual reads competing with the
            // 49 MB of output stores at the kernel's tail -- and the epilogue stores alpha * acc.  C layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            const float inv_alpha = 1.0f / g.alpha;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + j * 32 + (lane & 31);
                const bool col_ok = col < g.N;
                const float bs = (g.bias && col_ok) ? g.bias[col] : 0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        acc[i][j][r] = (col_ok && row < g.M) ? __builtin_fmaf(g.resid[(int64_t)row * g.ldr + col], inv_alpha, bs) : 0.0f;
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        }
        // K tile 0 of this tile is in flight (or long landed) in buffer `cur`, K tile 1 in the other: wait for BOTH before the first
        // fragment read (the persistent path issued tile 0 an epilogue ago -- what is waited for here is tile 1's request, one DMA latency,
        // once per tile instead of two on a cold chip)
        GL_WAIT_VM0();
        __syncthreads();
        GL_STAMP(tr_tile, 1);
        fragload(cur, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                fragload(cur, s + 1, (s + 1) & 1);
                frag_ready_one(s & 1);                        // (the reads just issued stay in flight)
                GL_SB(); mma(s & 1); GL_SB();
            }
            frag_ready_none((NSUB - 1) & 1);                  // the last fragments of this K tile are in registers: every read of `cur` has returned
            GL_WAIT_VM0();                 // K tile kt+1 (issued a whole tile ago) is in LDS
            __syncthreads();               // ... for every wave; and every wave holds its last fragments of tile kt: buffer `cur` is free
            GL_STAMP(tr_tile, 2 + kt);
            if (more1) fragload(cur ^ 1, 0, 0);
            if (more2 && !late) dma(src, kt + 2, cur);
            GL_SB(); mma((NSUB - 1) & 1); GL_SB();
            if constexpr (STAGGER) { if (more2 && late) dma(src, kt + 2, cur); }
            // (compiler-scheduled reads only; ASMFRAG counts by hand)
            // The fragments read after the barrier are long in their registers by now (eight DMA issues and eight MFMAs later); saying so HERE
            // keeps the compiler from draining LDS at the top of the loop instead -- where its wait (it cannot count across the back edge: lgkmcnt(0))
            // would also cover the six reads issued there for the NEXT sub-step and put a whole LDS round trip in front of the first MFMA group of
            // every K tile (found in the ISA, round 5; tools/ubench/gemm_bf16_trace.cpp: K tile 4500 -> see profiles/r05_bf16_asmfrag_ab.txt)
            if constexpr (!ASMFRAG) GL_WAIT_LGKM0();
            cur ^= 1;
        }
        // `cur` = the buffer the last K tile did NOT use (free since the barrier of the last iteration); the other one is free too once every
        // wave has passed that barrier -- which the epilogue's own first barrier guarantees again
        GL_STAMP(tr_tile, 2 + nk);
        static_assert(!(PERSIST && DIRECT && EPI == EPI_RESID), "the register residual epilogue runs one tile per workgroup");
        if constexpr (PERSIST && DIRECT) {
            const int em0 = m0, en0 = n0;
            loc += per_xcd;
            const bool more = loc < x_cnt;
            if (more) {                    // both staging buffers are free (every wave is past the last K tile's barrier, the epilogue uses none)
                origin_of(loc, m0, n0);
                set_src(m0, n0, src);
                dma(src, 0, cur);
                if (nk > 1) dma(src, 1, cur ^ 1);
            }
            gl_epilogue_direct<WGM, WGN, TM, TN, EPI>(g, acc, em0, en0);
            GL_STAMP(tr_tile, 3 + nk);
            ++tr_tile;
            if (!more) break;
        } else if constexpr (DIRECT && EPI == EPI_RESID) {
            gl_epilogue_resid_direct<WGM, WGN, TM, TN>(g, acc, m0, n0);
            break;
        } else if constexpr (DIRECT) {
            gl_epilogue_direct<WGM, WGN, TM, TN, EPI>(g, acc, m0, n0);
            break;
        } else if constexpr (PERSIST) {
            const int em0 = m0, en0 = n0;
            loc += per_xcd;
            const bool more = loc < x_cnt;
            if (more) {                    // next tile: its first K tile streams into `cur` UNDER this tile's epilogue
                origin_of(loc, m0, n0);
                set_src(m0, n0, src);
                dma(src, 0, cur);
            }
            gp_epilogue<WGM, WGN, TM, TN, EPI, BUF / 2, true>(g, acc, smem_f + (cur ^ 1) * (BUF / 2), em0, en0);   // one buffer: BUF bf16 = BUF / 2 floats
            if (!more) break;
            __syncthreads();               // every wave has read its last band out of the epilogue's buffer
            if (nk > 1) dma(src, 1, cur ^ 1);
        } else {
            if constexpr (EPI == EPI_RESID) {
                if (resid_in_acc) {                                 // bias and residual are in the accumulators: scale, then the plain LDS epilogue
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * g.alpha;
                    GemmArgs g2 = g;
                    g2.bias = nullptr;
                    gp_epilogue<WGM, WGN, TM, TN, EPI_NONE, BUF, true>(g2, acc, smem_f, m0, n0);
                    break;
                }
            }
            gp_epilogue<WGM, WGN, TM, TN, EPI, BUF, true>(g, acc, smem_f, m0, n0);      // 2 buffers x BUF bf16 = BUF floats
            break;
        }
    }
#undef GL_SB
#undef GL_WAIT_VM0
#undef GL_WAIT_LGKM0
}

// persist: 0 = one tile per workgroup, 1 = persistent with the LDS epilogue (round 4), 2 = persistent (more than 256 tiles) with the DIRECT
// register epilogue, 3 = the direct epilogue on one tile per workgroup
template <int WGM, int WGN, int TM, int TN, int EPI>
static void launch_gemm_bf16_glds(const GemmArgs &a, hipStream_t s, int persist = 0, bool stagger = false, bool asmfrag = false, bool rowblock_ok = false,
                                  bool resid_direct = false) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 64 * 2;
    // persistent walk by row blocks per XCD (kernel comment) when no XCD needs an extra round for it
    const int rb_tiles = ((tiles_m + 7) / 8) * tiles_n, even_tiles = (n_tiles + 7) / 8;
    const int rowblock = (rowblock_ok && tiles_m >= 8 && (rb_tiles + 31) / 32 <= (even_tiles + 31) / 32) ? 1 : 0;
    if constexpr (EPI != EPI_RESID) {
        const bool direct_ok = a.sigma_cols == 0 && a.remap_rows == 0 && (a.N % 16) == 0 && (a.ldo % 8) == 0 &&
                               (a.fast_act || (EPI != EPI_SILU && EPI != EPI_GLU)) && ((int64_t)(a.M + 31) * a.ldo < ((int64_t)1 << 31));
        if (persist >= 2 && direct_ok) {
            if (persist == 2 && n_tiles > 256) {
#ifdef PK_EXPERIMENTAL
                if (stagger) {
                    auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, true, true, true>;
                    static DynLdsSlots slots;
                    ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
                    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, rowblock);
                    return;
                }
#endif
                if (asmfrag) {
                    auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, true, true, false, true>;
                    static DynLdsSlots slots;
                    ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
                    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, rowblock);
                    return;
                }
                auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, true, true>;
                static DynLdsSlots slots;
                ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
                hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, rowblock);
                return;
            }
            auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, false, true>;
            static DynLdsSlots slots;
            ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
            hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, 0);
            return;
        }
    }
    if (a.out_blocked) { fprintf(stderr, "parakeet_amd: internal error: blocked output on the LDS epilogue\n"); abort(); }
    if constexpr (EPI == EPI_RESID) {
        // the register residual epilogue (gl_resid_init / gl_epilogue_resid_direct): row-major fp32 in and out, whole 16-column groups, 32-bit offsets
        const bool rd_ok = resid_direct && persist >= 2 && a.resid && a.alpha != 0.0f && !a.out_bf16 && !a.resid_init && a.sigma_cols == 0 && a.remap_rows == 0 &&
                           (a.N % 16) == 0 && (a.ldo % 4) == 0 && (a.ldr % 4) == 0 && ((int64_t)(a.M + 31) * a.ldo < ((int64_t)1 << 31)) &&
                           ((int64_t)(a.M + 31) * a.ldr < ((int64_t)1 << 31));
        if (rd_ok) {
            if (asmfrag) {
                auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, false, true, false, true>;
                static DynLdsSlots slots;
                ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
                hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, 0);
            } else {
                auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, false, true>;
                static DynLdsSlots slots;
                ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
                hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, 0);
            }
            return;
        }
    }
#ifdef PK_EXPERIMENTAL
    if (persist == 1 && n_tiles > 256) {                                // more than one round of the 256 CUs: one persistent workgroup per CU
        auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, true>;
        static DynLdsSlots slots;
        ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, rowblock);
        return;
    }
    if (stagger) {
        auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, false, false, true>;
        static DynLdsSlots slots;
        ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, 0);
        return;
    }
#endif
    if (asmfrag) {
        auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, false, false, false, true>;
        static DynLdsSlots slots;
        ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, 0);
        return;
    }
    auto kern = &gemm_bf16_glds_kernel<WGM, WGN, TM, TN, EPI, false>;
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles, 0);
}

}  // namespace pk
#endif  // PK_GEMM_BF16_GLDS_HPP
