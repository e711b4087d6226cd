// parakeet.cpp_amd/csrc/kernels/decode_gemv.hip -- the three per-step products of the TDT / RNNT loop
// (reference: LSTMCell::forward src/lstm.cpp:11-29, TDTJoint::forward src/tdt.cpp:15-24) for a lock-step batch
// of B <= a few hundred utterances:   out[B][N] = X[B][K] * W[N][K]^T,  K = 640.
//
// This is a latency problem (a 640-long dependent fp32 chain per output, ~130 dependent steps per clip), not a
// throughput one, so the tiling differs from the big GEMM:
//  * v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle dependent latency): one 16x16 (utterance x output) tile per
//    wavefront, a single accumulator chain of K/4 MFMAs in natural k order -> bit-identical to the oracle.
//  * Operands come straight from L2 with 16-byte loads: both X and W are stored in a "sigma" K layout -- inside every
//    block of 16 k the 4x4 index matrix is transposed -- so the float4 a lane loads holds exactly its k = 4s+kq
//    values for four consecutive MFMA steps.  W is permuted once at upload; X (h, z) is *written* permuted by the
//    fused epilogues of the previous product.
//  * A workgroup = one 16-output tile x up to four 16-utterance tiles: the W tile is fetched from L2 once and hits in
//    L1 for the other three waves; the per-XCD slice of the 10.8 MB of decode weights stays L2-resident across steps.
//  * Epilogues: LSTM cell (gates -> c', h'), joint activation relu(enc_proj[t_b] + .), or bias.
#include <cstdio>
#include <cstdlib>
#include "decode_dev.hpp"

namespace pk {

// MODE (chosen by the launcher from the arguments, so that the kernel's first basic block holds ALL its argument loads -- a branch on an argument in front of them made
// the compiler fetch the record in two or three dependent pieces, a scalar-cache round trip each, in a launch that is all latency): 0 = every row, 1 = the need flags
// as predicates (one row tile, decode_dev.hpp: PRED), 2 = the compacted list of flagged rows.  CHK: the grid is padded past the last column tile (tile count not a
// multiple of 8) and the padding workgroups leave at once; a launch whose tile count is a multiple of 8 (the cell: 160, the joint activation: 40) carries no such branch.
template <int EPI, int NCH, bool NTW, int WF = 1, int MODE = 0, bool CHK = false>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs a) {
    __shared__ float tile[4][16][17];
    // grid.x is the tile count rounded up to a multiple of 8 (launch_skinny_gemm): workgroup id % 8 = XCD, so XCD x owns the output tiles
    // nt % 8 == x of EVERY utterance group and re-reads only its eighth of W from its own L2 step after step
    if constexpr (CHK) {
        if ((EPI == SK_CELL ? 4 : 16) * (int)blockIdx.x >= (EPI == SK_CELL ? a.Hp : a.N)) return;
    }
    if constexpr (MODE == 1) {                                   // one row tile: flags as predicates, no list in front of the loads
        if (blockIdx.y == 0) skinny_tile<EPI, NCH, false, NTW, WF, true>(a, blockIdx.x, 0, tile);
    } else if constexpr (MODE == 2) {                            // prediction-net caching: only the utterances whose flag is set (TdtState::need)
        __shared__ int lst[kMaxListRows];
        __shared__ int wtot[4];
        const int cnt = dd_build_rowlist<false>(a.need, a.B, lst, wtot);
        if ((int)blockIdx.y * 64 >= cnt) return;
        skinny_tile<EPI, NCH, false, NTW, WF>(a, blockIdx.x, blockIdx.y, tile, lst, cnt);
    } else {
        skinny_tile<EPI, NCH, false, NTW, WF>(a, blockIdx.x, blockIdx.y, tile);
    }
}

template <int EPI, int NCH, bool NTW, int WF, int MODE>
static void launch_skinny_chk(const SkinnyArgs &a, dim3 grid, bool chk, hipStream_t s) {
    if (chk) hipLaunchKernelGGL((skinny_gemm_kernel<EPI, NCH, NTW, WF, MODE, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((skinny_gemm_kernel<EPI, NCH, NTW, WF, MODE, false>), grid, dim3(256), 0, s, a);
}
template <int EPI, int NCH, bool NTW, int WF>
static void launch_skinny_mode(const SkinnyArgs &a, dim3 grid, hipStream_t s) {
    const int n_tiles = EPI == SK_CELL ? a.Hp / 4 : (a.N + 15) / 16;
    const bool chk = n_tiles % 8 != 0;                           // (the heads: 65 tiles in a grid of 72 -- without the early leave the seven padding workgroups per row group ran whole chains beside the encoder: + 0.05 ms per headline step)
    const int mode = a.need ? (a.B <= 16 ? 1 : 2) : 0;
    if constexpr (WF > 1) {                                      // (the frame window comes with the need flags of a single utterance)
        if (mode != 1) { fprintf(stderr, "parakeet_amd: decode window without predicate rows -- engine bug\n"); abort(); }
        launch_skinny_chk<EPI, NCH, NTW, WF, 1>(a, grid, chk, s);
    } else {
        if (mode == 1) launch_skinny_chk<EPI, NCH, NTW, WF, 1>(a, grid, chk, s);
        else if (mode == 2) launch_skinny_chk<EPI, NCH, NTW, WF, 2>(a, grid, chk, s);
        else launch_skinny_chk<EPI, NCH, NTW, WF, 0>(a, grid, chk, s);
    }
}

template <int EPI>
static void launch_skinny_epi(const SkinnyArgs &a, dim3 grid, hipStream_t s) {
    const bool k640 = a.K == 640;
    if constexpr (EPI == SK_ACT) {
        if (a.F > 1) {                                               // frame window: its own instantiation (the extra enc_proj operands stay out of the batch kernel's registers)
            if (a.F > kDecWindowMax || a.nt_weights) { fprintf(stderr, "parakeet_amd: decode window %d > %d (or nt weights) -- engine bug\n", a.F, kDecWindowMax); abort(); }
            if (k640) launch_skinny_mode<EPI, 10, false, kDecWindowMax>(a, grid, s);
            else launch_skinny_mode<EPI, 0, false, kDecWindowMax>(a, grid, s);
            return;
        }
    }
    if (a.nt_weights) {
        if (k640) launch_skinny_mode<EPI, 10, true, 1>(a, grid, s);
        else launch_skinny_mode<EPI, 0, true, 1>(a, grid, s);
    } else {
        if (k640) launch_skinny_mode<EPI, 10, false, 1>(a, grid, s);
        else launch_skinny_mode<EPI, 0, false, 1>(a, grid, s);
    }
}

void launch_skinny_gemm(const SkinnyArgs &a, int epi, hipStream_t s) {
    const int n_tiles = epi == SK_CELL ? a.Hp / 4 : (a.N + 15) / 16;
    dim3 grid((n_tiles + 7) & ~7, (a.B + 63) / 64);              // padded: see skinny_gemm_kernel
    switch (epi) {
    case SK_BIAS: launch_skinny_epi<SK_BIAS>(a, grid, s); break;
    case SK_ACT: launch_skinny_epi<SK_ACT>(a, grid, s); break;
    case SK_CELL: launch_skinny_epi<SK_CELL>(a, grid, s); break;
    default: break;
    }
}

}  // namespace pk
