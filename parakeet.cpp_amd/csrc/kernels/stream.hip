// parakeet.cpp_amd/csrc/kernels/stream.hip -- the cached pieces of the streaming encoder (reference
// src/streaming_encoder.cpp: StreamingConformerAttention::forward_cached :162-272, CausalConformerConvModule::forward_cached
// :41-78).  A chunk is 1-3 encoder frames per stream and a few dozen cached keys, so these are latency kernels: one
// wavefront per (stream, head, query row), exact k-ordered fp32 fma chains on the VALU (bit-identical to the oracle), the
// canonical max / sum64 butterflies for the softmax.
#include "../pk_devmath.h"
#include "../common.hpp"
#include "kernels.hpp"

namespace pk {

// Phase stamps for tools/ubench/stream_att_bench.cpp (-DSA_TRACE): shader clock of lane 0 of every wave at the phase boundaries.  Production: nothing.
#ifdef SA_TRACE
__device__ long long *sa_trace;     // [workgroup][wave][8]
#define SA_STAMP(i) do { if (sa_trace && (threadIdx.x & 63) == 0 && threadIdx.x < 128) sa_trace[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 + (threadIdx.x >> 6)) * 8 + (i)] = clock64(); } while (0)
#else
#define SA_STAMP(i) do { } while (0)
#endif

// new cache of (stream, head) = the last `keep` rows of [cache ; this chunk's k / v] (:193-209): row r <- row nc + c - keep + r, written into the
// OTHER cache buffer (the attention blocks of the same launch still read the old one)
__device__ __forceinline__ void stream_cache_rotate(const float *__restrict__ qkv, const float *__restrict__ kcache, const float *__restrict__ vcache,
                                                    int cache_rows, int c, int nc, int d, int hd, int sidx, int h, int keep,
                                                    float *__restrict__ cache_k_out, float *__restrict__ cache_v_out, int tid, int nthr) {
    const int hd4 = hd / 4, kv = nc + c;
    for (int idx = tid; idx < keep * hd4; idx += nthr) {
        const int r = idx / hd4, e4 = idx % hd4, j = kv - keep + r;
        const int64_t src_c = ((int64_t)sidx * cache_rows + j) * d + h * hd + 4 * e4;
        const int64_t src_n = ((int64_t)sidx * c + (j - nc)) * 3 * d + h * hd + 4 * e4;
        const int64_t dst = ((int64_t)sidx * cache_rows + r) * d + h * hd + 4 * e4;
        *reinterpret_cast<float4 *>(cache_k_out + dst) = j < nc ? *reinterpret_cast<const float4 *>(kcache + src_c) : *reinterpret_cast<const float4 *>(qkv + src_n + d);
        *reinterpret_cast<float4 *>(cache_v_out + dst) = j < nc ? *reinterpret_cast<const float4 *>(vcache + src_c) : *reinterpret_cast<const float4 *>(qkv + src_n + 2 * d);
    }
}

// Blocks (stream * head, query row i < c): the attention of that row.  Blocks with blockIdx.y == c (when cache_k_out is set): the cache
// rotation of (stream, head) -- new cache = the last `keep` rows of [cache ; this chunk's k / v] (:193-209) written into the OTHER cache
// buffer, so it runs beside the attention blocks that still read the old one: one launch instead of three per layer.
__global__ __launch_bounds__(128) void stream_attention_kernel(const float *__restrict__ qkv, const float *__restrict__ kcache,
                                                               const float *__restrict__ vcache, int cache_rows, int c, int nc, int d, int H,
                                                               const float *__restrict__ pos, int P, const float *__restrict__ bias_u,
                                                               const float *__restrict__ bias_v, int att_left, int att_right, float scale,
                                                               float *__restrict__ ctx, float *__restrict__ cache_k_out,
                                                               float *__restrict__ cache_v_out, int keep, int ctx_sigma) {
    extern __shared__ float sm[];                                   // [hd] q+u, [hd] q+v, [kv] probabilities, [2] wave maxima
    const int hd = d / H, kv = nc + c;
    const int sidx = blockIdx.x / H, h = blockIdx.x % H, i = blockIdx.y;
    // one or two wavefronts (launcher: two when there are more than 64 keys -- a 70-row cache + the chunk: both halves of the key range run
    // their score chains side by side instead of one after the other)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    if (i == c) {                                                   // cache rotation of this (stream, head)
        stream_cache_rotate(qkv, kcache, vcache, cache_rows, c, nc, d, hd, sidx, h, keep, cache_k_out, cache_v_out, tid, nthr);
        return;
    }
    SA_STAMP(0);
    float *qu = sm, *qv = sm + hd, *pr = sm + 2 * hd, *red = pr + kv;
    const float *qrow = qkv + ((int64_t)sidx * c + i) * 3 * d + h * hd;
    for (int e = tid; e < hd; e += nthr) {
        const float q = qrow[e];
        qu[e] = q + bias_u[h * hd + e];                             // :209-212
        qv[e] = q + bias_v[h * hd + e];
    }
    __syncthreads();
    SA_STAMP(1);                                                    // q + biases in LDS
    auto krow = [&](int j) {                                        // key j: cached rows first, then this chunk's rows (:186-189)
        return j < nc ? kcache + ((int64_t)sidx * cache_rows + j) * d + h * hd : qkv + ((int64_t)sidx * c + (j - nc)) * 3 * d + d + h * hd;
    };
    auto vrow = [&](int j) {
        return j < nc ? vcache + ((int64_t)sidx * cache_rows + j) * d + h * hd : qkv + ((int64_t)sidx * c + (j - nc)) * 3 * d + 2 * d + h * hd;
    };
    const int off = P > kv ? P - kv : 0;                            // rightmost kv columns of the position scores, NOT rel-shifted (:215-224)
    const int abs_pos = kv - c + i;
    float m = -__builtin_huge_valf();
    for (int j = tid; j < kv; j += nthr) {
        // 16-byte loads of the key / position rows (each lane walks its own row), 32 / 16 / 2 of them in flight per round trip; the two chains
        // stay sequential in e
        const float4 *kr = reinterpret_cast<const float4 *>(krow(j)), *pp = reinterpret_cast<const float4 *>(pos + (int64_t)(off + j) * d + h * hd);
        const float4 *qu4 = reinterpret_cast<const float4 *>(qu), *qv4 = reinterpret_cast<const float4 *>(qv);
        float cs = 0.0f, ps = 0.0f;
        auto step = [&](const float4 &kk, const float4 &p4, int e) {
            const float4 a = qu4[e], bq = qv4[e];
            cs = __builtin_fmaf(a.x, kk.x, cs); cs = __builtin_fmaf(a.y, kk.y, cs); cs = __builtin_fmaf(a.z, kk.z, cs); cs = __builtin_fmaf(a.w, kk.w, cs);
            ps = __builtin_fmaf(bq.x, p4.x, ps); ps = __builtin_fmaf(bq.y, p4.y, ps); ps = __builtin_fmaf(bq.z, p4.z, ps); ps = __builtin_fmaf(bq.w, p4.w, ps);
        };
        int e0 = 0;
        for (; e0 + 16 <= hd / 4; e0 += 16) {
            float4 kk[16], p4[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { kk[u] = kr[e0 + u]; p4[u] = pp[e0 + u]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) step(kk[u], p4[u], e0 + u);
        }
        for (; e0 + 8 <= hd / 4; e0 += 8) {
            float4 kk[8], p4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { kk[u] = kr[e0 + u]; p4[u] = pp[e0 + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) step(kk[u], p4[u], e0 + u);
        }
        for (; e0 < hd / 4; ++e0) step(kr[e0], pp[e0], e0);
        float sc = (cs + ps) * scale;                               // :226
        const int dist = abs_pos - j;
        if ((att_left >= 0 || att_right >= 0) && (dist > att_left || -dist > att_right)) sc = -1e9f;   // masked_fill :231-247
        pr[j] = sc;
        m = fmaxf(m, sc);
    }
    SA_STAMP(2);                                                    // score chains of this wave's keys
    m = wave_max64(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (nthr > 64) m = fmaxf(red[0], red[1]);
    for (int j = tid; j < kv; j += nthr) pr[j] = dexpf_nonpos(pr[j] - m);
    __syncthreads();
    // the canonical sum (lane l adds elements l, l + 64, ... in index order, then the xor butterfly), by every wave for itself
    float p = 0.0f;
    for (int j = lane; j < kv; j += 64) p = p + pr[j];
    const float sum = wave_sum64(p);
    __syncthreads();                                                // (both waves have read the exponentials)
    for (int j = tid; j < kv; j += nthr) pr[j] = pr[j] / sum;
    __syncthreads();
    SA_STAMP(3);                                                    // softmax
    for (int eb = 0; eb < hd; eb += 2 * nthr) {                     // softmax(S) V, k = key index in natural order (:250); two output columns per thread
        const int e0 = eb + tid, e1 = eb + nthr + tid;
        const bool h0 = e0 < hd, h1 = e1 < hd;
        float acc0 = 0.0f, acc1 = 0.0f;
        int j = 0;
        for (; j + 24 <= kv; j += 24) {                             // 24 value rows in flight per round trip
            float v0[24], v1[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const float *vr = vrow(j + u);
                v0[u] = h0 ? vr[e0] : 0.0f;
                v1[u] = h1 ? vr[e1] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                acc0 = __builtin_fmaf(pr[j + u], v0[u], acc0);
                acc1 = __builtin_fmaf(pr[j + u], v1[u], acc1);
            }
        }
        for (; j + 8 <= kv; j += 8) {
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float *vr = vrow(j + u);
                v0[u] = h0 ? vr[e0] : 0.0f;
                v1[u] = h1 ? vr[e1] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_fmaf(pr[j + u], v0[u], acc0);
                acc1 = __builtin_fmaf(pr[j + u], v1[u], acc1);
            }
        }
        for (; j < kv; ++j) {
            const float *vr = vrow(j);
            if (h0) acc0 = __builtin_fmaf(pr[j], vr[e0], acc0);
            if (h1) acc1 = __builtin_fmaf(pr[j], vr[e1], acc1);
        }
        auto put = [&](int e, float acc) {                          // ctx_sigma: the out-projection reads its A operand in the sigma K layout
            const int col = h * hd + e;
            ctx[((int64_t)sidx * c + i) * d + (ctx_sigma ? ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)) : col)] = acc;
        };
        SA_STAMP(4);                                                // value rows loaded, chains done
        if (h0) put(e0, acc0);
        if (h1) put(e1, acc1);
    }
    SA_STAMP(5);
}

// The same attention for the steady state of a streaming session (a 70-row cache + the chunk: kv <= kSaKv keys, head size 64 / 128) with the
// key and position tiles staged through LDS.  The kernel above lets every lane walk its own key row with 16-byte loads: one load instruction
// touches 64 different 128-byte lines and needs the seven instructions after it to use them up, two tiles x two waves of that is the whole
// 32 KB L1 -- the phase stamps (tools/ubench/stream_att_bench.cpp) show six dependent round trips per launch (q, two batches of key / position
// rows, three of value rows: 11 of the launch's 14 us).  Here everything a workgroup reads is requested in ONE round trip before anything
// waits: the key / position tiles COALESCED (32 lanes per 512-byte row, four waves: 10 sixteen-byte loads per lane and tile) into registers
// and on into LDS at a pitch of hd + 4 floats (row-per-lane ds_read_b128 then hits 16 distinct bank groups: conflict-free), the value column
// of every output feature into registers.  The chains read their operands from LDS instead of L1: same products, same order, same bits.
constexpr int kSaKv = 80, kSaNew = 8;    // keys (cache + chunk) / rows of the chunk the LDS-tile form takes
// (amdgpu_waves_per_eu(1, 2): without it the scheduler protects an occupancy nobody needs -- 384 workgroups on 256 CUs -- by holding the
//  loads back until registers free up, i.e. by serialising exactly the round trips this kernel exists to merge)
template <int HD4 /* head size / 4: 16 or 32 */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void stream_attention_tiles_kernel(const float *__restrict__ qkv, const float *__restrict__ kcache,
                                                                     const float *__restrict__ vcache, int cache_rows, int c, int nc, int d, int H,
                                                                     const float *__restrict__ pos, int P, const float *__restrict__ bias_u,
                                                                     const float *__restrict__ bias_v, int att_left, int att_right, float scale,
                                                                     float *__restrict__ ctx, float *__restrict__ cache_k_out,
                                                                     float *__restrict__ cache_v_out, int keep, int ctx_sigma) {
    constexpr int HD = 4 * HD4, PITCH = HD + 4, NT = 256, TR = (kSaKv * HD4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [HD] q+u, [HD] q+v, [kSaKv] probabilities, [8] wave maxima, K tile [kv + 1][PITCH], P tile [kv + 1][PITCH]
    const int kv = nc + c;
    const int sidx = blockIdx.x / H, h = blockIdx.x % H, i = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (i == c) {                                                   // cache rotation of this (stream, head)
        stream_cache_rotate(qkv, kcache, vcache, cache_rows, c, nc, d, HD, sidx, h, keep, cache_k_out, cache_v_out, tid, NT);
        return;
    }
    SA_STAMP(0);
    float *qu = sm, *qv = sm + HD, *pr = sm + 2 * HD, *red = pr + kSaKv, *Kt = red + 8, *Pt = Kt + (kv + 1) * PITCH;   // (tiles of kv rows + one spare)
    const int off = P > kv ? P - kv : 0;                            // rightmost kv columns of the position scores, NOT rel-shifted (:215-224)
    const int abs_pos = kv - c + i;
    // ---- every global read of the workgroup, requested before anything waits.  No branch and no pointer select around a load (either makes
    // the compiler drain the queue at every one of them): the cached rows (keys 0 .. nc - 1, :186-189) come from the caches with the row index
    // clamped, the chunk's own c <= kSaNew rows from qkv by the first c * HD4 lanes; what was read past the end is never stored / used ----
    // (addresses as a workgroup-uniform base + a 32-bit byte offset per lane: the `saddr` form of global_load needs no 64-bit lane arithmetic --
    //  with it the allocator recycled load destinations as address temporaries and the loads waited on each other)
    auto ld1 = [](const float *base, unsigned byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); };
    auto ld4 = [](const float *base, unsigned byte_off) { return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off); };
    const int qe = tid < HD ? tid : 0;
    const float *kbase = kcache + (int64_t)sidx * cache_rows * d + h * HD, *vbase = vcache + (int64_t)sidx * cache_rows * d + h * HD;
    const float *nbase = qkv + (int64_t)sidx * c * 3 * d + h * HD;  // row u of the chunk: + u * 3 d (+ d: k, + 2 d: v)
    const float *pbase = pos + (int64_t)off * d + h * HD;
    const int nc1 = nc > 0 ? nc - 1 : 0;
    const int r0 = tid / HD4, c4 = tid % HD4;                       // tile element of load u: row r0 + (NT / HD4) u, float4 column c4
    float4 kreg[TR], preg[TR];
#pragma unroll
    for (int u = 0; u < TR; ++u) {
        const int row = r0 + (NT / HD4) * u;
        kreg[u] = ld4(kbase, (unsigned)((row < nc ? row : nc1) * d + 4 * c4) * 4u);
        preg[u] = ld4(pbase, (unsigned)((row < kv ? row : kv - 1) * d + 4 * c4) * 4u);
    }
    const int tn = tid < c * HD4 ? tid : c * HD4 - 1;               // (c <= kSaNew = 8: c * HD4 <= 256 lanes)
    const float4 knew = ld4(nbase, (unsigned)((tn / HD4) * 3 * d + d + 4 * (tn % HD4)) * 4u);
    __builtin_amdgcn_sched_barrier(0);                              // (key / position tiles first: the scores wait for them)
    // the value tile the same way (it takes the key tile's place in LDS once the scores are done: 35 loads per lane in all -- a wave's
    // load counter holds 63, a column of 72 four-byte loads per lane does not fit beside the tiles)
    float4 vreg[TR];
#pragma unroll
    for (int u = 0; u < TR; ++u) {
        const int row = r0 + (NT / HD4) * u;
        vreg[u] = ld4(vbase, (unsigned)((row < nc ? row : nc1) * d + 4 * c4) * 4u);
    }
    const float4 vnew = ld4(nbase, (unsigned)((tn / HD4) * 3 * d + 2 * d + 4 * (tn % HD4)) * 4u);
    const float q = ld1(nbase, (unsigned)(i * 3 * d + qe) * 4u), bu = ld1(bias_u + h * HD, (unsigned)qe * 4u), bv = ld1(bias_v + h * HD, (unsigned)qe * 4u);
    __builtin_amdgcn_sched_barrier(0);
    // (one basic block from the first load to the barrier: a branch in between lets the loads sink to their stores, one round trip each.
    //  Lanes past the head size repeat lane 0's element: the same value to the same address)
    qu[qe] = q + bu;                                                // :209-212
    qv[qe] = q + bv;
#pragma unroll
    for (int u = 0; u < TR; ++u) {
        const int row = r0 + (NT / HD4) * u;
        // (unconditional stores -- a store under a condition pulls its load in with it: what was read past the end goes to the spare row kv)
        *reinterpret_cast<float4 *>(Kt + (row < nc ? row : kv) * PITCH + 4 * c4) = kreg[u];
        *reinterpret_cast<float4 *>(Pt + (row < kv ? row : kv) * PITCH + 4 * c4) = preg[u];
    }
    *reinterpret_cast<float4 *>(Kt + (tid < c * HD4 ? nc + tid / HD4 : kv) * PITCH + 4 * (tid % HD4)) = knew;
    __syncthreads();
    SA_STAMP(1);                                                    // operands in LDS
    float m = -__builtin_huge_valf();
    if (tid < kv) {                                                 // one key per lane (kv <= kSaKv < 256); the two chains stay sequential in e
        const int j = tid;
        const float4 *kr = reinterpret_cast<const float4 *>(Kt + j * PITCH), *pp = reinterpret_cast<const float4 *>(Pt + j * PITCH);
        const float4 *qu4 = reinterpret_cast<const float4 *>(qu), *qv4 = reinterpret_cast<const float4 *>(qv);
        float cs = 0.0f, ps = 0.0f;
#pragma unroll 8
        for (int e = 0; e < HD4; ++e) {
            const float4 kk = kr[e], p4 = pp[e], a = qu4[e], bq = qv4[e];
            cs = __builtin_fmaf(a.x, kk.x, cs); cs = __builtin_fmaf(a.y, kk.y, cs); cs = __builtin_fmaf(a.z, kk.z, cs); cs = __builtin_fmaf(a.w, kk.w, cs);
            ps = __builtin_fmaf(bq.x, p4.x, ps); ps = __builtin_fmaf(bq.y, p4.y, ps); ps = __builtin_fmaf(bq.z, p4.z, ps); ps = __builtin_fmaf(bq.w, p4.w, ps);
        }
        float sc = (cs + ps) * scale;                               // :226
        const int dist = abs_pos - j;
        if ((att_left >= 0 || att_right >= 0) && (dist > att_left || -dist > att_right)) sc = -1e9f;   // masked_fill :231-247
        pr[j] = sc;
        m = sc;
    }
    SA_STAMP(2);                                                    // score chains
    m = wave_max64(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    // every score chain is done: the value tile takes the key tile's place (published by the barriers of the softmax below)
#pragma unroll
    for (int u = 0; u < TR; ++u) {
        const int row = r0 + (NT / HD4) * u;
        *reinterpret_cast<float4 *>(Kt + (row < nc ? row : kv) * PITCH + 4 * c4) = vreg[u];
    }
    *reinterpret_cast<float4 *>(Kt + (tid < c * HD4 ? nc + tid / HD4 : kv) * PITCH + 4 * (tid % HD4)) = vnew;
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));        // (waves without keys contribute -inf: the maximum of the same set)
    for (int j = tid; j < kv; j += NT) pr[j] = dexpf_nonpos(pr[j] - m);
    __syncthreads();
    // the canonical sum (lane l adds elements l, l + 64, ... in index order, then the xor butterfly), by every wave for itself
    float p = 0.0f;
    for (int j = lane; j < kv; j += 64) p = p + pr[j];
    const float sum = wave_sum64(p);
    __syncthreads();                                                // (every wave has read the exponentials)
    for (int j = tid; j < kv; j += NT) pr[j] = pr[j] / sum;
    __syncthreads();
    SA_STAMP(3);                                                    // softmax
    if (tid < HD) {                                                 // softmax(S) V on the prefetched column, k = key index in natural order (:250)
        float acc = 0.0f;
        for (int u = 0; u < kv; ++u) acc = __builtin_fmaf(pr[u], Kt[u * PITCH + tid], acc);    // (the value tile; consecutive lanes, consecutive banks)
        const int col = h * HD + tid;
        ctx[((int64_t)sidx * c + i) * d + (ctx_sigma ? ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)) : col)] = acc;
    }
    SA_STAMP(4);
    SA_STAMP(5);
}

void launch_stream_attention(const float *qkv_new, const float *kcache, const float *vcache, int cache_rows, int S, int c, int nc, int d,
                             int n_heads, const float *pos, int P, const float *bias_u, const float *bias_v, int att_left, int att_right,
                             float *ctx, hipStream_t s, float *cache_k_out, float *cache_v_out, int keep_max, int ctx_sigma) {
    const int hd = d / n_heads;
    // the kernel walks the head features and the cache rows as float4 chunks: fail loudly instead of dropping a tail
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (n_heads <= 0 || d % n_heads || hd % 4 || d % 4 || !al16(qkv_new) || !al16(kcache) || !al16(vcache) || !al16(pos) || !al16(ctx) ||
        !al16(cache_k_out) || !al16(cache_v_out))
        fail(PK_ERR_INVALID, "launch_stream_attention: head size %d / hidden size %d must be multiples of 4 and every buffer 16-byte aligned", hd, d);
    const float scale = 1.0f / sqrtf((float)hd);
    const size_t lds = (size_t)(2 * hd + nc + c + 2) * sizeof(float);
    const int kv = nc + c, keep = kv > keep_max ? keep_max : kv;
    const bool rotate = cache_k_out && cache_v_out && keep > 0;
    if (kv <= kSaKv && c <= kSaNew && (hd == 128 || hd == 64)) {    // steady state of a session: the LDS-tile form
        const size_t lds_t = (size_t)(2 * hd + kSaKv + 8 + 2 * (kv + 1) * (hd + 4)) * sizeof(float);
        const dim3 grid(S * n_heads, c + (rotate ? 1 : 0));
#define PK_SAT_LAUNCH(HD4V) do { \
            static DynLdsSlots slots; \
            ensure_dyn_lds(slots, reinterpret_cast<const void *>(&stream_attention_tiles_kernel<HD4V>), (size_t)(2 * hd + kSaKv + 8 + 2 * (kSaKv + 1) * (hd + 4)) * sizeof(float)); \
            hipLaunchKernelGGL((stream_attention_tiles_kernel<HD4V>), grid, dim3(256), lds_t, s, qkv_new, kcache, vcache, cache_rows, c, nc, d, n_heads, pos, P, bias_u, \
                               bias_v, att_left, att_right, scale, ctx, rotate ? cache_k_out : nullptr, rotate ? cache_v_out : nullptr, keep, ctx_sigma); \
        } while (0)
        if (hd == 128) PK_SAT_LAUNCH(32); else PK_SAT_LAUNCH(16);
#undef PK_SAT_LAUNCH
        return;
    }
    hipLaunchKernelGGL(stream_attention_kernel, dim3(S * n_heads, c + (rotate ? 1 : 0)), dim3(kv > 64 ? 128 : 64), lds, s, qkv_new, kcache, vcache, cache_rows, c, nc, d,
                       n_heads, pos, P, bias_u, bias_v, att_left, att_right, scale, ctx, rotate ? cache_k_out : nullptr, rotate ? cache_v_out : nullptr, keep, ctx_sigma);
}

__global__ void stream_cache_update_kernel(const float *__restrict__ cache_in, int nc, const float *__restrict__ qkv, int col0, int c, int d,
                                           int cache_rows, int keep, float *__restrict__ cache_out, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = (int)(idx % d), r = (int)((idx / d) % keep), sidx = (int)(idx / ((int64_t)d * keep));
    const int j = nc + c - keep + r;                                // row of [cache ; new]
    cache_out[((int64_t)sidx * cache_rows + r) * d + e] =
        j < nc ? cache_in[((int64_t)sidx * cache_rows + j) * d + e] : qkv[((int64_t)sidx * c + (j - nc)) * 3 * d + col0 + e];
}
void launch_stream_cache_update(const float *cache_in, int nc, const float *qkv_new, int col0, int S, int c, int d, int cache_rows,
                                int keep_max, float *cache_out, hipStream_t s) {
    const int kv = nc + c, keep = kv > keep_max ? keep_max : kv;
    if (keep <= 0) return;
    const int64_t n = (int64_t)S * keep * d;
    hipLaunchKernelGGL(stream_cache_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cache_in, nc, qkv_new, col0, c, d, cache_rows,
                       keep, cache_out, n);
}

// One thread per (stream, channel).  Everything the thread reads -- its column of [cache ; g] (KC - 1 + c values), the KC taps, bias and the four
// BatchNorm parameters -- is requested in ONE round trip (the first version walked the frames one by one: a dependent round trip per frame and one
// more for the cache copy; c <= CMAX, longer chunks take the frame loop below).  Same taps in the same order: same bits.
template <int KC, int CMAX>
__global__ void stream_dwconv_kernel(const float *__restrict__ g, const float *__restrict__ cache_in, int has_cache, int c, int d,
                                     const float *__restrict__ w /*[KC][d]*/, const float *__restrict__ bias, const float *__restrict__ bn_mean,
                                     const float *__restrict__ bn_rstd, const float *__restrict__ bn_g, const float *__restrict__ bn_b,
                                     float *__restrict__ out, float *__restrict__ cache_out, int64_t n, int out_sigma) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (stream, channel)
    if (idx >= n) return;
    const int ch = (int)(idx % d), sidx = (int)(idx / d);
    constexpr int CL = KC - 1;
    const int ocol = out_sigma ? ((ch & ~15) | ((ch & 3) << 2) | ((ch >> 2) & 3)) : ch;
    if (c <= CMAX) {
        // (no branch around a load: clamped indices, the first chunk's zero left padding (:55-63) applied after the loads)
        float cat[CL + CMAX], wk[KC];
#pragma unroll
        for (int r = 0; r < CL; ++r) cat[r] = cache_in[((int64_t)sidx * CL + r) * d + ch];
#pragma unroll
        for (int t = 0; t < CMAX; ++t) cat[CL + t] = g[((int64_t)sidx * c + (t < c ? t : c - 1)) * d + ch];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) wk[kk] = w[kk * d + ch];
        const float bs = bias[ch], mu = bn_mean[ch], rs = bn_rstd[ch], bg = bn_g[ch], bb = bn_b[ch];
        if (!has_cache) {
#pragma unroll
            for (int r = 0; r < CL; ++r) cat[r] = 0.0f;
        }
#pragma unroll
        for (int t = 0; t < CMAX; ++t) {
            if (t < c) {
                float acc = 0.0f;
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) acc = __builtin_fmaf(wk[kk], cat[t + kk], acc);      // depthwise, no padding (:71)
                float v = acc + bs;
                v = __builtin_fmaf((v - mu) * rs, bg, bb);
                out[((int64_t)sidx * c + t) * d + ocol] = dsiluf(v);
            }
        }
        // last KC - 1 rows of the concatenation (:66-69): rows c .. c + CL - 1
#pragma unroll
        for (int r = 0; r < CL; ++r) {
            float keep = cat[r];
#pragma unroll
            for (int t = 1; t <= CMAX; ++t)
                if (t == c) keep = cat[r + t];
            cache_out[((int64_t)sidx * CL + r) * d + ch] = keep;
        }
        return;
    }
    auto cat = [&](int r) {                                         // row r of [cache(CL rows) ; g(c rows)]
        if (r < CL) return has_cache ? cache_in[((int64_t)sidx * CL + r) * d + ch] : 0.0f;   // first chunk: zero left padding (:55-63)
        return g[((int64_t)sidx * c + (r - CL)) * d + ch];
    };
    for (int t = 0; t < c; ++t) {
        float acc = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) acc = __builtin_fmaf(w[kk * d + ch], cat(t + kk), acc);      // depthwise, no padding (:71)
        float v = acc + bias[ch];
        v = __builtin_fmaf((v - bn_mean[ch]) * bn_rstd[ch], bn_g[ch], bn_b[ch]);
        out[((int64_t)sidx * c + t) * d + ocol] = dsiluf(v);
    }
    float keep[CL];
#pragma unroll
    for (int r = 0; r < CL; ++r) keep[r] = cat(c + r);             // last KC-1 rows of the concatenation (:66-69)
#pragma unroll
    for (int r = 0; r < CL; ++r) cache_out[((int64_t)sidx * CL + r) * d + ch] = keep[r];
}
void launch_stream_dwconv(const float *g, const float *cache_in, int has_cache, int S, int c, int d, int kc, const float *w, const float *bias,
                          const float *bn_mean, const float *bn_rstd, const float *bn_g, const float *bn_b, float *out, float *cache_out,
                          hipStream_t s, int out_sigma) {
    const int64_t n = (int64_t)S * d;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (kc == 9) hipLaunchKernelGGL((stream_dwconv_kernel<9, 4>), grid, dim3(256), 0, s, g, cache_in, has_cache, c, d, w, bias, bn_mean, bn_rstd, bn_g, bn_b, out, cache_out, n, out_sigma);
    else if (kc == 31) hipLaunchKernelGGL((stream_dwconv_kernel<31, 2>), grid, dim3(256), 0, s, g, cache_in, has_cache, c, d, w, bias, bn_mean, bn_rstd, bn_g, bn_b, out, cache_out, n, out_sigma);
}

}  // namespace pk
