// parakeet.cpp_amd/csrc/kernels/decode.hip -- on-device greedy decoders.
//
// CTC  (reference src/ctc.cpp:12-25, :40-127): row log-softmax + first-max argmax, then the collapse /
//      timestamp rules, one thread per utterance (T <= a few hundred frames).
// TDT / RNNT (src/tdt.cpp:36-201, src/rnnt.cpp:56-177, src/lstm.cpp:11-29): the whole greedy loop stays on
//      the GPU -- the reference does 2 device->host scalar syncs per symbol.  All utterances of a batch
//      advance in lock-step; one "step" = one joint evaluation per live utterance:
//        gemm  GH = h W_hh^T            (MFMA)           -> lstm_cell   (gates, c', h' candidates)
//        gemm  PP = h' W_pred^T         (MFMA)           -> joint_act   z = relu(enc_proj[t_b] + PP)
//        gemm  LOG = z [W_label;W_dur]^T + b (MFMA)      -> tdt_decide  log-softmax, argmax, control,
//                                                                        commit / revert of the LSTM state
#include <cstdio>
#include <cstdlib>
#include "decode_dev.hpp"

namespace pk {

// ---- row log-softmax + first-max argmax (one wavefront per row) ---------------------------------------
// lsm = (x - max) - log(sum64(exp(x - max))).  argmax over lsm with strict '>' (lowest index wins ties).
__global__ __launch_bounds__(256) void logsoftmax_argmax_kernel(const float *__restrict__ logits, int64_t rows, int ld, int n,
                                                                float *__restrict__ lp_out, int *__restrict__ best_idx,
                                                                float *__restrict__ best_lp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const BestLP r = wave_logsoftmax_argmax(logits + row * ld, n, lp_out ? lp_out + row * n : nullptr, lane);
    if (lane == 0) {
        best_idx[row] = r.idx;
        best_lp[row] = r.lp;
    }
}
void launch_logsoftmax_argmax(const float *logits, int64_t rows, int ld, int n, float *lp_out, int *best_idx, float *best_lp, hipStream_t s) {
    hipLaunchKernelGGL(logsoftmax_argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, rows, ld, n, lp_out, best_idx, best_lp);
}

// ctc_greedy_decode(_with_timestamps): src/ctc.cpp:52-72 and :93-123, literally.
__global__ void ctc_collapse_kernel(const int *__restrict__ best_idx, const float *__restrict__ best_lp, int B, int T, int blank,
                                    int *__restrict__ ids, int *__restrict__ lens, int *__restrict__ start, int *__restrict__ end,
                                    float *__restrict__ conf, int pitch, SeqRag rg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t in0 = (int64_t)b * T;                                   // first frame of this utterance in the (packed) frame axis
    if (rg.T) { in0 = rg.T_off[b]; T = rg.T[b]; }                   // ragged batch: its own frame count
    const int64_t o0 = (int64_t)b * pitch;
    int prev = -1, n = 0;
    for (int t = 0; t < T; ++t) {
        const int best = best_idx[in0 + t];
        if (best != prev) {
            if (prev != -1 && prev != blank && n > 0) end[o0 + n - 1] = t - 1;
            if (best != blank) {
                ids[o0 + n] = best;
                start[o0 + n] = t;
                end[o0 + n] = t;
                conf[o0 + n] = dexpf(best_lp[in0 + t]);
                ++n;
            }
        }
        prev = best;
    }
    if (n > 0) end[o0 + n - 1] = T - 1;
    lens[b] = n;
}
void launch_ctc_collapse(const int *best_idx, const float *best_lp, int B, int T, int blank, int *ids, int *lens, int *start, int *end,
                         float *conf, hipStream_t s, int pitch, const SeqRag &rag) {
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3((B + 63) / 64), dim3(64), 0, s, best_idx, best_lp, B, T, blank, ids, lens, start, end, conf,
                       pitch > 0 ? pitch : T, rag);
}


// ctc_greedy_decode(_with_timestamps)_boosted (src/phrase_boost.cpp:70-171): the boosted argmax of frame t depends on the tokens
// emitted before t (the trie's active states), so the frames of one utterance are walked in order by one 256-thread workgroup;
// the utterances of the batch run side by side.  Outputs as ctc_collapse_kernel; confidence = exp(unboosted log-prob) (:151-152).
__global__ __launch_bounds__(256) void ctc_boosted_kernel(const float *__restrict__ logp, int B, int T, int V, int blank, TrieDev trie,
                                                          int *__restrict__ ids, int *__restrict__ lens, int *__restrict__ start,
                                                          int *__restrict__ end, float *__restrict__ conf, int pitch, SeqRag rg) {
    extern __shared__ __attribute__((aligned(16))) unsigned cb_sm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t in0 = (int64_t)b * T;                                   // first frame of this utterance in the (packed) frame axis
    if (rg.T) { in0 = rg.T_off[b]; T = rg.T[b]; }                   // ragged batch
    const int64_t o0 = (int64_t)b * pitch;
    const int MW = (V + 31) >> 5;
    unsigned *mask = cb_sm;                                         // [MW]
    int *acts = reinterpret_cast<int *>(mask + MW);                 // [2][1 + kTrieMaxActive] (count first), double-buffered
    float *red = reinterpret_cast<float *>(acts + 2 * (1 + kTrieMaxActive));   // [2][8]: best val, best idx per wavefront
    int cur = 0;
    if (tid == 0) { acts[0] = 1; acts[1] = 0; }
    auto rebuild = [&](const int *set) {                            // mask = union of the children of the active states
        for (int i = tid; i < MW; i += 256) mask[i] = 0u;
        __syncthreads();
        const int n = set[0];
        for (int a = 0; a < n; ++a) {
            const int sn = set[1 + a];
            const int c1 = trie.off[sn + 1];
            for (int c = trie.off[sn] + tid; c < c1; c += 256) {
                const int tk = trie.tok[c];
                if (tk >= 0 && tk < V) atomicOr(&mask[tk >> 5], 1u << (tk & 31));
            }
        }
        __syncthreads();
    };
    __syncthreads();
    rebuild(acts);
    int prev = -1, n = 0;
    for (int t = 0; t < T; ++t) {
        const float *frame = logp + (in0 + t) * V;
        float best = -__builtin_huge_valf();
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float v = frame[i] + (((mask[i >> 5] >> (i & 31)) & 1u) ? trie.boost : 0.0f);
            if (bi == 0x7fffffff || v > best) { best = v; bi = i; }
        }
        wave_butterfly([&](auto off) {
            const float ob = wave_xor<decltype(off)::value>(best);
            const int oi = wave_xor_i<decltype(off)::value>(bi);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        });
        float *rd = red + 8 * (t & 1);                               // double-buffered: one barrier per frame
        if (lane == 0) { rd[wave] = best; rd[4 + wave] = __int_as_float(bi); }
        __syncthreads();
        best = rd[0]; bi = __float_as_int(rd[4]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ob = rd[w];
            const int oi = __float_as_int(rd[4 + w]);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (bi != prev) {                                            // block-uniform control flow
            if (tid == 0 && prev != -1 && prev != blank && n > 0) end[o0 + n - 1] = t - 1;
            if (bi != blank) {
                if (tid == 0) {
                    ids[o0 + n] = bi;
                    start[o0 + n] = t;
                    end[o0 + n] = t;
                    conf[o0 + n] = dexpf(frame[bi]);
                }
                ++n;
                int *src = acts + cur * (1 + kTrieMaxActive), *dst = acts + (cur ^ 1) * (1 + kTrieMaxActive);
                if (tid == 0) { dst[0] = 1; dst[1] = 0; }
                __syncthreads();
                const int na = src[0];
                for (int a = 0; a < na; ++a) {
                    const int sn = src[1 + a];
                    const int c1 = trie.off[sn + 1];
                    for (int c = trie.off[sn] + tid; c < c1; c += 256)
                        if (trie.tok[c] == bi) {
                            const int slot = atomicAdd(&dst[0], 1);
                            if (slot < kTrieMaxActive) dst[1 + slot] = trie.node[c];
                        }
                }
                __syncthreads();
                if (tid == 0 && dst[0] > kTrieMaxActive) dst[0] = kTrieMaxActive;
                cur ^= 1;
                rebuild(dst);
            }
        }
        prev = bi;
    }
    if (tid == 0) {
        if (n > 0) end[o0 + n - 1] = T - 1;
        lens[b] = n;
    }
}
void launch_ctc_boosted(const float *logp, int B, int T, int V, int blank, const TrieDev &trie, int *ids, int *lens, int *start, int *end,
                        float *conf, hipStream_t s, int pitch, const SeqRag &rag) {
    const size_t lds = (size_t)((V + 31) / 32 + 2 * (1 + kTrieMaxActive) + 16) * sizeof(int);
    hipLaunchKernelGGL(ctc_boosted_kernel, dim3(B), dim3(256), lds, s, logp, B, T, V, blank, trie, ids, lens, start, end, conf,
                       pitch > 0 ? pitch : T, rag);
}

// ---- TDT / RNNT step kernels --------------------------------------------------------------------------
// LSTMCell::forward (src/lstm.cpp:11-29), gate order i,f,g,o.  gi = W_ih x + b (layer 0: row `token` of the
// precomputed table g1 = W_ih E + b), gh = W_hh h.  Writes CANDIDATE states hn/cn; tdt_decide commits them.
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float *__restrict__ gi, int gi_ld, const int *__restrict__ gi_row,
                                                        const float *__restrict__ gh, const float *__restrict__ c,
                                                        int B, int Hp, float *__restrict__ hn, float *__restrict__ cn) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * Hp) return;
    const int b = idx / Hp, j = idx % Hp;
    const float *gir = gi + (int64_t)(gi_row ? gi_row[b] : b) * gi_ld;
    const float *ghr = gh + (int64_t)b * 4 * Hp;
    const float ig = dsigmoidf(gir[j] + ghr[j]);
    const float fg = dsigmoidf(gir[Hp + j] + ghr[Hp + j]);
    const float gg = dtanhf(gir[2 * Hp + j] + ghr[2 * Hp + j]);
    const float og = dsigmoidf(gir[3 * Hp + j] + ghr[3 * Hp + j]);
    const float t1 = fg * c[idx];
    const float t2 = ig * gg;
    const float cnew = t1 + t2;
    cn[idx] = cnew;
    hn[idx] = og * dtanhf(cnew);
}
void launch_lstm_cell(const float *gi, int gi_ld, const int *gi_row, const float *gh, const float *c, int B, int Hp, float *hn,
                      float *cn, hipStream_t s) {
    hipLaunchKernelGGL(lstm_cell_kernel, dim3((B * Hp + 255) / 256), dim3(256), 0, s, gi, gi_ld, gi_row, gh, c, B, Hp, hn, cn);
}

// TDTJoint::forward first half (src/tdt.cpp:17-18): z = relu(enc_proj(enc_t) + pred_proj(pred) [+ bp, switch A5])
__global__ __launch_bounds__(256) void joint_act_kernel(const float *__restrict__ ep, const int *__restrict__ t, int T, int J,
                                                        const float *__restrict__ pp, const float *__restrict__ bp, int B,
                                                        float *__restrict__ z) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * J) return;
    const int b = idx / J, j = idx % J;
    int tt = t[b];
    tt = tt < T ? tt : T - 1;
    float p = pp[idx];
    if (bp) p = p + bp[j];
    const float s = ep[((int64_t)b * T + tt) * J + j] + p;
    z[idx] = s > 0.0f ? s : 0.0f;
}
void launch_joint_act(const float *ep, const int *t, int T, int J, const float *pp, const float *bp, int B, float *z, hipStream_t s) {
    hipLaunchKernelGGL(joint_act_kernel, dim3((B * J + 255) / 256), dim3(256), 0, s, ep, t, T, J, pp, bp, B, z);
}

// One 256-thread workgroup per utterance: heads' log-softmax + first-max argmax, then the control flow of
// tdt_greedy_decode (src/tdt.cpp:62-106; timestamps :157-187) or rnnt_greedy_decode (src/rnnt.cpp:75-107).
// The logits row is staged in LDS once; exp() is evaluated by all four wavefronts, the canonical sum64 by one
// (the summation ORDER is part of the numerics contract; who evaluates the terms is not).
// BOOST (phrase boosting, src/phrase_boost.cpp:177-350): the label argmax runs over log-prob + boost for the tokens that continue
// an active trie state (a V-bit mask in LDS, rebuilt from the CSR children every step); the confidence stays the raw log-prob
// and the active set advances on every emission.
template <bool BOOST, bool SCORE = false, bool FAST = false, int NC = 12>
__global__ __launch_bounds__(256) void tdt_decide_kernel(TdtState st) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // x[V+D], e[V+D], scratch[16] (+ BOOST: mask, active sets)
    tdt_decide_one<BOOST, false, SCORE, FAST, NC>(st, blockIdx.x, sm);
}
// The tolerance-class mode's plain greedy step on the register-resident form (decode_dev.hpp: FAST).  EXPERIMENTAL builds: PK_DEC_FAST=0 keeps the exact form.
static bool decide_fast_on() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_DEC_FAST"); return e ? atoi(e) != 0 : true; }();
    return on;
#else
    return true;
#endif
}
void launch_tdt_decide(const TdtState &st, hipStream_t s) {
    const size_t lds = (size_t)(((st.F > 1 ? st.F : 1) + 1) * (st.V + st.D) + 16) * sizeof(float);     // (frame window: its F rows in front of the scratch)
    if (st.F > 1 && (st.V + st.D > 5 * 256 || st.F > kDecWindowMax || st.trie.off || st.force_label || st.h_bf16)) {
        fprintf(stderr, "parakeet_amd: decode window outside its conditions -- engine bug\n"); abort();
    }
    if (st.trie.off) {
        const size_t extra = (size_t)((st.V + 31) / 32 + 2 * kTrieMaxActive + 1) * sizeof(int);
        hipLaunchKernelGGL(tdt_decide_kernel<true>, dim3(st.B), dim3(256), lds + extra, s, st);
    } else if (st.force_label) {
        hipLaunchKernelGGL((tdt_decide_kernel<false, true>), dim3(st.B), dim3(256), lds, s, st);      // pk_tdt_score
    } else if (st.h_bf16 && st.V + st.D <= 33 * 256 && st.V >= 2 && decide_fast_on()) {
        if (st.L * st.Hp <= 3 * 256) hipLaunchKernelGGL((tdt_decide_kernel<false, false, true, 3>), dim3(st.B), dim3(256), lds, s, st);
        else if (st.L * st.Hp <= 6 * 256) hipLaunchKernelGGL((tdt_decide_kernel<false, false, true, 6>), dim3(st.B), dim3(256), lds, s, st);
        else hipLaunchKernelGGL((tdt_decide_kernel<false, false, true>), dim3(st.B), dim3(256), lds, s, st);
    } else {
        if (st.L * st.Hp <= 3 * 256) hipLaunchKernelGGL((tdt_decide_kernel<false, false, false, 3>), dim3(st.B), dim3(256), lds, s, st);
        else if (st.L * st.Hp <= 6 * 256) hipLaunchKernelGGL((tdt_decide_kernel<false, false, false, 6>), dim3(st.B), dim3(256), lds, s, st);
        else hipLaunchKernelGGL(tdt_decide_kernel<false>, dim3(st.B), dim3(256), lds, s, st);
    }
}

__global__ void tdt_init_kernel(TdtState st) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) *st.done_count = 0;
    if (b >= st.B) return;
    if (!st.keep_state) st.token[b] = st.blank;        // SOS = blank (src/tdt.cpp:56-59); a streaming chunk carries its last token
    st.t[b] = 0;
    st.nsym[b] = 0;
    st.n_out[b] = 0;
    st.steps[b] = 0;
    st.done[b] = 0;
    st.lens[b] = 0;
    if (st.margin) st.margin[b] = __builtin_huge_valf();
    if (st.need) st.need[b] = 1;                       // the first step computes the prediction net for everyone
    if (st.trie.off) {                                 // active_states = {root} (phrase_boost.cpp:258)
        st.trie.n_act[b] = 1;
        st.trie.act[(int64_t)b * kTrieMaxActive] = 0;
    }
}
// decode tables of n utterances appended to a decode group (capi.cpp): frames and first enc_proj row of each (TdtState::Tb / row0)
__global__ void rag_decode_tables_kernel(const int *__restrict__ T, const int *__restrict__ T_off, int T_uniform, int n, int row_base,
                                         int *__restrict__ Tb_out, int *__restrict__ row0_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Tb_out[i] = T ? T[i] : T_uniform;
    row0_out[i] = row_base + (T ? T_off[i] : i * T_uniform);
}
void launch_rag_decode_tables(const int *T, const int *T_off, int T_uniform, int n, int row_base, int *Tb_out, int *row0_out, hipStream_t s) {
    hipLaunchKernelGGL(rag_decode_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, s, T, T_off, T_uniform, n, row_base, Tb_out, row0_out);
}

void launch_tdt_init(const TdtState &st, hipStream_t s) { hipLaunchKernelGGL(tdt_init_kernel, dim3((st.B + 63) / 64), dim3(64), 0, s, st); }

}  // namespace pk
