// parakeet.cpp_amd/csrc/kernels/attention_bf16.hip -- relative-position multi-head attention of the TOLERANCE-class mode (pk_config.gemm_bf16,
// the precision BASELINE configs[2] names), all three contractions on v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 softmax.
// Reference: ConformerAttention::rel_position_attention (src/encoder.cpp:135-171) with rel_shift (:85-109) in closed form:
//     S[i][j] = ( (q_i + u_h) . k_j  +  (q_i + v_h) . P_h[j - i + T - 1] ) / sqrt(hd) ,   ctx_i = softmax_j(S[i][:]) V
// The bit-exact fp32 kernel (attention.hip: 16x16x4 fp32 MFMA, [32][T] score block in LDS) is 34 % of the tdt-600m encoder in this mode; here:
//   * q, k, v arrive as bf16 (the qkv GEMM's epilogue rounds them), the projected position table as bf16 (pos_proj GEMM), ctx leaves as bf16
//     (it is only ever out_proj's operand).  (q + v_h) . P_p  is evaluated as  (q + u_h) . P_p + c_h[p]  with  c_h[p] = (v_h - u_h) . P_p  computed
//     once per (layer, head) in fp32 (pos_cvec_kernel): ONE biased copy of the query tile in registers instead of two.
//   * One workgroup = 128 query rows of one (utterance, head): 4 wavefronts x 32 rows, streaming over 32-key tiles with an online softmax,
//     so nothing of size T lives in LDS and the sequence length is unbounded.  All products are formed TRANSPOSED (keys x queries): in the
//     32x32 accumulator layout a lane then holds 16 keys of ONE query, so the softmax row statistics are in-lane reductions plus one exchange
//     with lane ^ 32, and the probability tile IS the B operand of  ctx^T = V^T P^T  after a cvt to bf16 -- no LDS round trip for P.
//   * rel_shift: the position scores of key tile t need band rows  p - base = (j - j0) - (i - i0) + 31  in [0, 62] = two 32-row blocks of
//     (q + u) P^T; consecutive key tiles share a block, so ONE new 32x32 block per tile goes into a wave-private LDS strip [64][34] and the
//     skewed read  strip[jj - n + 31][n]  is conflict-free (pitch 34: the lane stride is 33 words).
//   * P tiles are MFMA A operands straight from L2 (16 bytes per lane = the 8 k of one step), reloaded right after their last use so the
//     loads fly under the softmax / PV phases; K tiles are staged in LDS once per workgroup (natural rows, 16-byte fragment reads).  V needs the contraction index (key) along the lane's 8 operands: the V tile is staged in LDS as
//     [4 keys][16 dv] sub-blocks and read with ds_read_b64_tr_b16 (gfx950 transpose read; lane-linear = conflict-free, tools/ubench/tr16_probe.cpp),
//     double-buffered, one barrier per key tile.
// Numerics: tolerance class (compared with the oracle's gemm_bf16 mode, which rounds the same operands: oracle/pk_oracle.c attention()).
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float ab_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ab_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ab_bf16x4 __attribute__((ext_vector_type(4)));
typedef short ab_s16x4 __attribute__((ext_vector_type(4)));

static constexpr int AB_QB = 128;       // query rows per workgroup (4 waves x 32)
static constexpr int AB_SKP = 34;       // skew strip pitch (floats)

// c[l][h][p] = (v_h - u_h) . P[l][p][h*HD ..]   fp32, natural k order (the oracle evaluates the same chain)
__global__ void pos_cvec_kernel(const __bf16 *__restrict__ pos, const float *__restrict__ bias_u, const float *__restrict__ bias_v, int P, int d, int H,
                                int HD, float *__restrict__ cvec) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * P) return;
    const int h = idx / P, p = idx % P;
    const __bf16 *row = pos + (int64_t)p * d + h * HD;
    float acc = 0.0f;
    for (int k = 0; k < HD; ++k) acc = __builtin_fmaf(bias_v[h * HD + k] - bias_u[h * HD + k], (float)row[k], acc);
    cvec[idx] = acc;
}
void launch_pos_cvec(const void *pos_bf16, const float *bias_u, const float *bias_v, int P, int d, int n_heads, float *cvec, hipStream_t s) {
    const int n = n_heads * P;
    hipLaunchKernelGGL(pos_cvec_kernel, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<const __bf16 *>(pos_bf16), bias_u, bias_v, P, d, n_heads,
                       d / n_heads, cvec);
}

// Phase clocks for tools/ubench/attn_bf16_bench.cpp (-DAB_TRACE): per wave, shader clocks summed over the key tiles.  Production builds: nothing.
#ifdef AB_TRACE
__device__ long long *ab_trace;      // [workgroup][wave][8]
#define AB_T0() long long ab_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ab_t_ = clock64()
#define AB_STAMP(i) do { const long long n_ = clock64(); ab_acc_[i] += n_ - ab_t_; ab_t_ = n_; } while (0)
#define AB_FLUSH() do { if (ab_trace && lane == 0) for (int i_ = 0; i_ < 8; ++i_) ab_trace[((long long)blockIdx.x * 4 + wave) * 8 + i_] = ab_acc_[i_]; } while (0)
#else
#define AB_T0() do { } while (0)
#define AB_STAMP(i) do { } while (0)
#define AB_FLUSH() do { } while (0)
#endif

__device__ __forceinline__ int ab_rowidx(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }     // row of accumulator register r (32x32 C layout)

template <int HD, bool RAG = false /* ragged batch: its own instantiation (the uniform kernel sits at its 256-VGPR budget) */>
__global__ __launch_bounds__(256, 2) void relpos_attention_bf16_kernel(const __bf16 *__restrict__ qkv, int ldq, int d, int T,
                                                                       const __bf16 *__restrict__ pos /*[2T-1][d]*/, const float *__restrict__ cvec /*[H][2T-1]*/,
                                                                       const float *__restrict__ bias_u, float scale_log2e, __bf16 *__restrict__ ctx,
                                                                       int n_qb, int n_bh, int nkt, int pos_T, SeqRag rg) {
    constexpr int NK = HD / 16;          // MFMA k-steps of a contraction over the head dimension
    constexpr int NDT = HD / 32;         // 32-wide dv tiles of ctx^T
    constexpr int VCH = HD / 8;          // 16-byte chunks per V row
    constexpr int NV = 32 * VCH / 256;   // V chunks per thread per key tile
    static_assert(NV >= 1, "HD >= 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char ab_smem[];
    constexpr int KP = HD * 2 + 16;      // K tile row pitch in bytes (+16: the row-per-lane 16-byte fragment reads spread over the banks)
    __bf16 *Vimg = reinterpret_cast<__bf16 *>(ab_smem);                          // [2][32 * HD]   sub-blocked V tiles
    unsigned char *Kimg = ab_smem + 2 * 32 * HD * 2;                             // [2][32 keys][KP bytes]  K tiles, natural rows
    float *skew = reinterpret_cast<float *>(Kimg + 2 * 32 * KP);                 // [4][64 * AB_SKP]
    float *cb = skew + 4 * 64 * AB_SKP;                                          // [32 nkt + 128]  c band of this workgroup
    __builtin_amdgcn_s_setprio(3);

    const int H = d / HD;
    int h, qb;
    int64_t row0;                                                   // first row of this utterance in the (packed) row axis of qkv / ctx
    if constexpr (RAG) {   // ragged batch (kernels.hpp: SeqRag; attention.hip has the same mapping): this utterance's own length
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        h = (k / rg.units.count) * 8 + xcd;
        if (h >= H) return;
        const RagUnit un = rg.units.u[k % rg.units.count];
        qb = un.r0 / AB_QB;
        T = rg.T[un.b];
        row0 = rg.T_off[un.b];
        nkt = (T + 31) / 32;
    } else {   // the query blocks of one (utterance, head) take consecutive slots of ONE XCD (block id % 8 = XCD): K, V and the P band stay in its L2
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        qb = k % n_qb;
        const int bh = (k / n_qb) * 8 + xcd;
        if (bh >= n_bh) return;
        h = bh % H;
        row0 = (int64_t)(bh / H) * T;
    }
    // the table / c vector were built for pos_T >= T frames: row p of a T-frame table is row p + (pos_T - T) of it (engine.cpp: ensure_pos_tables)
    const int P = 2 * T - 1, pshift = pos_T - T, Ptab = 2 * pos_T - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, g = lane >> 5;
    const int i0w = qb * AB_QB + 32 * wave;                         // first query row of this wave
    const bool active = i0w < T;                                    // wave-uniform
    const __bf16 *qrow = qkv + row0 * ldq + h * HD;                 // q of (b, h); k at + d, v at + 2 d
    const __bf16 *prow = pos + (int64_t)pshift * d + h * HD;
    float *sk = skew + wave * 64 * AB_SKP;
    const int base0 = T - 32 - i0w;                                 // p of band row 0 of block 0 for this wave
    const int cb_len = 32 * nkt + 128;

    for (int i = tid; i < cb_len; i += 256) {                       // c band: p = (T - 128 - 128 qb) + i
        int p = T - AB_QB - qb * AB_QB + i;
        p = p < 0 ? 0 : (p > P - 1 ? P - 1 : p);
        cb[i] = cvec[(int64_t)h * Ptab + pshift + p];
    }
    // V tile staging (workgroup-cooperative): chunk c = tid + 256 i -> key c / VCH, 8 dv at 8 (c % VCH)
    float4 vreg[NV];
    auto v_load = [&](int t) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tid + 256 * i, key = c / VCH, ch = c % VCH;
            int kr = 32 * t + key;
            kr = kr < T ? kr : T - 1;
            vreg[i] = *reinterpret_cast<const float4 *>(qrow + (int64_t)kr * ldq + 2 * d + 8 * ch);
        }
    };
    auto v_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tid + 256 * i, key = c / VCH, ch = c % VCH;
            const int q4 = key >> 2, pq = q4 >> 1, gq = q4 & 1, dv16 = ch >> 1, dt = dv16 >> 1, dvh = dv16 & 1;
            const int blk = (pq * NDT + dt) * 4 + gq * 2 + dvh;
            lds_store16(Vimg + buf * 32 * HD + blk * 64 + (key & 3) * 16 + (ch & 1) * 8, vreg[i]);
        }
    };
    // K tile: staged like V (whole rows, coalesced: a wave instruction covers 4-8 full rows) and shared by the four waves.  Round 4: as an A operand
    // straight from L2 (row-per-lane 16-byte loads, 32 rows = 32 cache lines per instruction, every wave its own copy) the K and P loads kept the
    // CU's vector L1 busy for ~5.5 k clocks per key tile against 0.8 k of MFMA work (tools/ubench/attn_bf16_bench: profiles/r04_attn_bf16_phases.txt).
    float4 kst[NV];
    auto k_load = [&](int t) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tid + 256 * i, key = c / VCH, ch = c % VCH;
            int kr = 32 * t + key;
            kr = kr < T ? kr : T - 1;
            kst[i] = *reinterpret_cast<const float4 *>(qrow + (int64_t)kr * ldq + d + 8 * ch);
        }
    };
    auto k_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tid + 256 * i, key = c / VCH, ch = c % VCH;
            lds_store16(Kimg + buf * 32 * KP + key * KP + ch * 16, kst[i]);
        }
    };
    // P block: A operand straight from L2 (every wave needs a different block at a given step; an LDS ring of four blocks does not fit beside the
    // skew strips at two workgroups per CU): lane (row n, half g) takes the 8 k of step s at 16 s + 8 g
    ab_bf16x8 preg[NK], qreg[NK];
    auto p_load = [&](int m) {
        int pr = base0 + 32 * m + n;
        pr = pr < 0 ? 0 : (pr > P - 1 ? P - 1 : pr);
        const __bf16 *p = prow + (int64_t)pr * d + 8 * g;
#pragma unroll
        for (int s = 0; s < NK; ++s) preg[s] = *reinterpret_cast<const ab_bf16x8 *>(p + 16 * s);
    };
    // the 16 c values of block m this lane adds (band rows x = ab_rowidx(r, g): four runs of four floats) -- read as four 16-byte LDS loads BEFORE the
    // strip stores: interleaved one by one, every 4-byte read waited for its own LDS round trip in front of its store (the compiler cannot
    // move a read of cb across a store to sk), ~2 k clocks per key tile (profiles/r04_attn_phases.txt)
    auto c_load = [&](int m, float (&cv)[16]) {
        const float4 *c4 = reinterpret_cast<const float4 *>(cb + 96 - 32 * wave + 32 * m + 4 * g);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 c = c4[2 * k];
            cv[4 * k] = c.x; cv[4 * k + 1] = c.y; cv[4 * k + 2] = c.z; cv[4 * k + 3] = c.w;
        }
    };
    // position block m -> + c -> the wave's skew strip, half m & 1
    auto g_block = [&](int m) {
        ab_f32x16 ga;
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < NK; ++s) ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(preg[s], qreg[s], ga, 0, 0, 0);
        float cv[16];
        c_load(m, cv);
#pragma unroll
        for (int r = 0; r < 16; ++r) sk[(32 * (m & 1) + ab_rowidx(r, g)) * AB_SKP + n] = ga[r] + cv[r];
    };

    AB_T0();
    v_load(0);
    if (active) {
        int qr = i0w + n;
        qr = qr < T ? qr : T - 1;
        const __bf16 *qp = qrow + (int64_t)qr * ldq + 8 * g;
        const float *bu = bias_u + h * HD + 8 * g;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            const ab_bf16x8 raw = *reinterpret_cast<const ab_bf16x8 *>(qp + 16 * s);
#pragma unroll
            for (int e = 0; e < 8; ++e) qreg[s][e] = (__bf16)((float)raw[e] + bu[16 * s + e]);        // (q + u) as the reference forms it, rounded once
        }
        p_load(0);
    }
    k_load(0);
    v_store(0);
    k_store(0);
    lds_store_fence();
    __syncthreads();                                                // V tile 0 and the c band are in LDS
    if (active) {
        g_block(0);
        p_load(1);
    }

    ab_f32x16 O[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.0f;
    float m_run = -__builtin_huge_valf(), l_run = 0.0f;
    AB_STAMP(0);                                                    // prologue

    for (int t = 0; t < nkt; ++t) {
        const bool more = t + 1 < nkt;
        if (more) { v_load(t + 1); k_load(t + 1); }
        if (active) {
            // ---- content scores, transposed: S^T[key][query], and position block t + 1: two INDEPENDENT chains of NK MFMAs, issued alternately
            //      (one after the other, every MFMA waits for its predecessor's result: a wave issues in order) ----
            ab_f32x16 sa, ga;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sa[r] = 0.0f; ga[r] = 0.0f; }
            float cv[16];
            c_load(t + 1, cv);                                      // (under the MFMA chains)
            const unsigned char *kb = Kimg + (t & 1) * 32 * KP + n * KP + 16 * g;
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                const ab_bf16x8 ka = *reinterpret_cast<const ab_bf16x8 *>(kb + 32 * s);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qreg[s], sa, 0, 0, 0);
                ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(preg[s], qreg[s], ga, 0, 0, 0);
            }
            AB_STAMP(1);                                            // QK^T and position MFMAs (incl. the wait for this tile's K / P operands)
            if (more) p_load(t + 2);
            {   // block t + 1 (+ c) into the wave's skew strip, half (t + 1) & 1; then the skewed read of blocks t and t + 1
                const int m = t + 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) sk[(32 * (m & 1) + ab_rowidx(r, g)) * AB_SKP + n] = ga[r] + cv[r];
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float sv[16];
            float mloc = -__builtin_huge_valf();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = ab_rowidx(r, g);
                const int xr = jj - n + 31;                          // band row relative to block t: 0 .. 62
                const int phys = 32 * ((t + (xr >> 5)) & 1) + (xr & 31);
                const float v = (sa[r] + sk[phys * AB_SKP + n]) * scale_log2e;    // (content + position) * scale, in the exp2 domain
                sv[r] = v;
            }
            if (32 * (t + 1) > T) {                                  // (wave-uniform) only the last key tile has keys past the end to mask
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = (32 * t + ab_rowidx(r, g) < T) ? sv[r] : -__builtin_huge_valf();
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sv[r]);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                 // the strip half of block t is overwritten next iteration
            AB_STAMP(2);                                            // skew strip write / read, scale, mask, local max
            // ---- online softmax: lanes n and n + 32 hold the two halves of query n's keys ----
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sv[r] = __builtin_amdgcn_exp2f(sv[r] - m_new);
                psum = psum + sv[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            if (__ballot(alpha != 1.0f) != 0ull) {                  // (wave-uniform) the running maximum moved for some query: rescale ctx^T
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[dt][r] = O[dt][r] * alpha;
            }
            AB_STAMP(3);                                            // exp2, row sums, rescale
            // ---- ctx^T += V^T P^T : two k-steps of 16 keys; the probability registers are the B operand as they lie ----
            const __bf16 *vb = Vimg + (t & 1) * 32 * HD + lane * 4;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                ab_bf16x8 pB;
#pragma unroll
                for (int e = 0; e < 8; ++e) pB[e] = (__bf16)sv[8 * st + e];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const ab_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ab_s16x4 *)(vb + ((2 * st) * NDT + dt) * 256));
                    const ab_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ab_s16x4 *)(vb + ((2 * st + 1) * NDT + dt) * 256));
                    ab_s16x4 both[2] = {lo, hi};
                    ab_bf16x8 vA;
                    __builtin_memcpy(&vA, both, 16);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vA, pB, O[dt], 0, 0, 0);
                }
            }
        }
        AB_STAMP(4);                                                // PV MFMAs
        if (more) {
            v_store((t + 1) & 1);
            k_store((t + 1) & 1);
            lds_store_fence();
        }
        AB_STAMP(5);                                                // V staging store (incl. the wait for the V load)
        __syncthreads();                                            // V tile t + 1 visible; every wave is done with tile t's image
        AB_STAMP(6);                                                // barrier
    }
    AB_FLUSH();
    if (!active) return;
    l_run = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l_run);
    const int i = i0w + n;
    if (i < T) {
        __bf16 *orow = ctx + (row0 + i) * d + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                ab_bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)(O[dt][4 * rq + e] * inv);
                *reinterpret_cast<ab_bf16x4 *>(orow + 32 * dt + 8 * rq) = o;          // dv = 32 dt + 8 rq + 4 g + e
            }
    }
}

size_t relpos_attention_bf16_lds_bytes(int T, int hd) {
    if (hd != 64 && hd != 128) return 0;
    const int nkt = (T + 31) / 32;
    return (size_t)2 * 32 * hd * 2 + (size_t)2 * 32 * (hd * 2 + 16) + (size_t)4 * 64 * AB_SKP * 4 + (size_t)(32 * nkt + 128) * 4;
}

template <int HD>
static void launch_att_bf16(const void *qkv, int B, int T, int d, int n_heads, const void *pos, const float *cvec, const float *bias_u, void *ctx,
                            hipStream_t s, int pos_T, const SeqRag &rag) {
    if (rag.units.u) { T = rag.T_max; pos_T = rag.pos_T; }         // LDS (the c band) sized for the longest utterance
    if (pos_T <= 0) pos_T = T;
    const int n_qb = (T + AB_QB - 1) / AB_QB, n_bh = B * n_heads, nkt = (T + 31) / 32;
    const float scale_log2e = (1.0f / sqrtf((float)HD)) * 1.44269504088896340736f;
    const size_t lds = relpos_attention_bf16_lds_bytes(T, HD);
    if (rag.units.u) {
        auto kern = &relpos_attention_bf16_kernel<HD, true>;
        static DynLdsSlots slots_r;
        ensure_dyn_lds(slots_r, reinterpret_cast<const void *>(kern), lds);
        const dim3 grid((unsigned)(((n_heads + 7) / 8) * 8 * (int64_t)rag.units.count));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, static_cast<const __bf16 *>(qkv), 3 * d, d, T, static_cast<const __bf16 *>(pos), cvec, bias_u,
                           scale_log2e, static_cast<__bf16 *>(ctx), n_qb, n_bh, nkt, pos_T, rag);
        return;
    }
    auto kern = &relpos_attention_bf16_kernel<HD, false>;
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
    dim3 grid(((n_bh + 7) / 8) * 8 * n_qb);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, static_cast<const __bf16 *>(qkv), 3 * d, d, T, static_cast<const __bf16 *>(pos), cvec, bias_u,
                       scale_log2e, static_cast<__bf16 *>(ctx), n_qb, n_bh, nkt, pos_T, rag);
}
int relpos_attention_bf16_block_rows(int) { return AB_QB; }

void launch_relpos_attention_bf16(const void *qkv_bf16, int B, int T, int d, int n_heads, const void *pos_bf16, const float *cvec, const float *bias_u,
                                  void *ctx_bf16, hipStream_t s, int pos_T, const SeqRag &rag) {
    const int hd = d / n_heads;
    if (hd == 128) launch_att_bf16<128>(qkv_bf16, B, T, d, n_heads, pos_bf16, cvec, bias_u, ctx_bf16, s, pos_T, rag);
    else if (hd == 64) launch_att_bf16<64>(qkv_bf16, B, T, d, n_heads, pos_bf16, cvec, bias_u, ctx_bf16, s, pos_T, rag);
}

}  // namespace pk
