// parakeet.cpp_amd/csrc/engine.hpp -- the static execution plan behind the C ABI.
//
// Not a tensor library: a fixed pipeline for the Parakeet model family.  Weights are uploaded (and the few
// derived tables built) once in to_gpu(); activations live in a grow-only workspace; every stage is a
// short, fixed sequence of hand-written kernels on one HIP stream.
#pragma once
#include <map>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "kernels/kernels.hpp"
#include "safetensors.hpp"
#include "text.hpp"

namespace pk {

struct LayerW {   // reference names: SURVEY.md Appendix B / scripts/convert_nemo.py:134-184
    const float *ffn1_ng, *ffn1_nb, *ffn1_w1, *ffn1_b1, *ffn1_w2, *ffn1_b2;
    const float *ffn2_ng, *ffn2_nb, *ffn2_w1, *ffn2_b1, *ffn2_w2, *ffn2_b2;
    const float *att_ng, *att_nb, *wqkv, *bqkv, *wo, *bo, *wpos, *pos_u, *pos_v;
    const float *cv_ng, *cv_nb, *pw1_w, *pw1_b, *dw_w, *dw_b, *bn_g, *bn_b, *bn_mean, *bn_rstd, *pw2_w, *pw2_b;
    const float *fin_g, *fin_b;
};
struct SubW {
    const float *c1w, *c1b, *d1w, *d1b, *c2w, *c2b, *d2w, *d2b, *c3w, *c3b, *pw, *pb;
};
struct DecW {
    const float *embed;
    const float *wih[4], *bih[4], *whh[4];
    const float *g1;            // [V][4Hp] = W_ih0 * E[token] + b  (layer 0 input projection table)
    const float *we, *be, *wp, *bp, *wl, *bl, *wd, *bd;
    const float *ctc_w, *ctc_b;
};

MelTables make_mel_tables(int n_mels, bool window_centered, const std::function<const float *(const float *, size_t)> &upload);

// Per-kernel hipEvent timing: when a sink is attached every launch is bracketed by two events.
struct ProfileSink {
    struct Rec { std::string name; hipEvent_t e0, e1; double flops, bytes; };
    std::vector<Rec> recs;
    ~ProfileSink();
};

// Extents of one RAGGED batch -- clips of different lengths packed back to back along the time axis (kernels.hpp: MelRag / SubRag / SeqRag;
// the reference's "batch inference" roadmap item, README.md:513, without padding) -- as the host knows them, and the one int32 image of
// every table its kernels read.  build_* fill the host side; Workspace::set_ragged uploads the image and forms the device views.
struct RagBatch {
    int B = 0;
    int level = 0;                                // what the batch starts from: 0 PCM samples, 1 mel frames, 2 encoder frames
    std::vector<int64_t> pcm_off;                 // [B+1] (level 0)
    std::vector<int> Tm, H2, T;                   // per utterance: mel frames (level <= 1), rows after conv1 / dw1, encoder frames
    int64_t n_samples = 0;                        // totals and maxima
    int sum_Tm = 0, sum_H2 = 0, sum_T = 0, Tm_max = 0, T_max = 0;
    int sum_Tm_pad = 0;                           // frames of the row-padded log-mel blocks (mel_logmel_pitch per clip)
    int strip_rows = 0, dw_frames = 0, att_rows = 0;   // unit granularities the tables were built for
    std::vector<int32_t> image;
    size_t o_pcm_off = 0, o_Tm = 0, o_Tm_off = 0, o_Tm_pad_off = 0, o_H2 = 0, o_H2_off = 0, o_T = 0, o_T_off = 0, o_u_c1 = 0, o_u_row = 0, o_u_dw = 0, o_u_att = 0;
    int n_u_c1 = 0, n_u_row = 0, n_u_dw = 0, n_u_att = 0;
    void build_from_samples(const int64_t *lens, int B, int att_block_rows);
    void build_from_mel(const int *Tm, int B, int att_block_rows);
    void build_from_frames(const int *T, int B, int att_block_rows);
    // words of `image` a batch of <= max_clips clips with <= max_total_samples samples can need (capacity of the device copy)
    static size_t image_words_bound(int max_clips, int64_t max_total_samples);
  private:
    void finish(int att_block_rows);
};
struct RagDev {                                   // device views of a RagBatch's tables (kernels.hpp)
    MelRag mel; SubRag c1, dw2; SeqRag att, dwc, seq;
};

// Activation workspace of one resident batch (B clips of n_samples): sized once, reused every run.
struct Workspace {
    int B = 0; int64_t n_samples = 0; int Tm = 0, T = 0, max_tokens = 0;
    int T_run = 0;              // encoder frames per clip of the current UNIFORM run (= T unless the run is smaller than the capacity, set_uniform)
    // Ragged runs: `ragged` set, rag / rv describe the batch; B = its clip count, T = max_tokens / T = the CAPACITY the output arrays are pitched
    // for (fixed at size_ragged), rows = packed encoder rows.  Uniform runs: rows = B * T.
    bool ragged = false;
    // run_tdt's host poll ("all utterances finished"): called on the loop's stream right BEFORE the poll's copy is enqueued -- what it enqueues (a
    // streaming chunk's result copies into pinned memory) rides on the poll's synchronisation; poll_hit: the loop ended on such a poll (nothing was
    // enqueued after the hook's copies)
    std::function<void(hipStream_t)> before_poll;
    bool poll_hit = false;
    RagBatch rag;
    RagDev rv;
    DevBuf ragdev;                                // device copy of rag.image (+ the decode tables of a decode group)
    int32_t *rag_pinned = nullptr;                // pinned staging of the image (async upload on the run's stream)
    size_t rag_pinned_words = 0;
    hipEvent_t rag_copied = nullptr;              // the last upload out of rag_pinned has executed
    hipEvent_t poll_ev = nullptr;                 // run_tdt_loop: the poll's copy has executed (the host waits for it, not for the steps enqueued behind it)
    // teacher-forced scoring (pk_tdt_score, one utterance): device arrays of the given decisions and of the recorded joint outputs (TdtState)
    const int *force_label = nullptr, *force_dur = nullptr;
    int n_force = 0;
    float *score_lab = nullptr, *score_dur = nullptr;
    const int *n_force_b = nullptr;               // pk_stream_score: per-stream step counts; arrays of stream b start at b * force_stride
    int force_stride = 0;
    const int *dec_Tb = nullptr, *dec_row0 = nullptr;   // decode loop on a ragged batch: frames / first enc_proj row of every utterance (device; null = uniform)
    int64_t rows(int B_run) const { return ragged ? rag.sum_T : (int64_t)B_run * T_run; }   // packed encoder rows of the current run
    int t_max() const { return ragged ? rag.T_max : T_run; }
    DevBuf pcm, logmel, feats, a2, a3, a4, a5, flat, x, n, hbuf, qkv, ctx, g, dwb;
    DevBuf ctc_logits, ctc_lp, best_idx, best_lp;
    DevBuf ep, gh, gi, pp, z, logits, h, c, hn, cn, ints, ids, start, end, conf, lens, margin;
    DevBuf persist_bar;         // barrier words of the single-launch decode loop (kTdtBarrierWords)
    DevBuf trie_act;            // phrase boosting: per-utterance active trie states [B][kTrieMaxActive] + counts [B]
    // captured chunk of decode steps (Model::run_tdt): replayed while the step-invariant kernel arguments stay the same
    hipGraphExec_t dec_graph = nullptr;
    std::vector<unsigned char> dec_graph_key;
    ~Workspace() {
        if (dec_graph) (void)hipGraphExecDestroy(dec_graph);
        if (rag_pinned) (void)hipHostFree(rag_pinned);
        if (rag_copied) (void)hipEventDestroy(rag_copied);
        if (poll_ev) (void)hipEventDestroy(poll_ev);
    }
    Workspace() = default;
    Workspace(const Workspace &) = delete;
    Workspace &operator=(const Workspace &) = delete;
    // the per-utterance token arrays are pitched B x T x max_symbols: the only buffers that grow with (clips x longest clip); released when a
    // pipeline is re-sized for much shorter clips (capi.cpp model_pipeline), re-reserved by the next size_* call
    void release_tokens() { ids.release(); start.release(); end.release(); conf.release(); }
    size_t token_bytes() const { return ids.cap + start.cap + end.cap + conf.cap; }
    void size_for(const pk_config &cfg, int B, int64_t n_samples, int Tm);
    void size_decode(const pk_config &cfg, int B, int T, size_t rows_cap = 0);     // TDT / RNNT decode state only (rows_cap: enc_proj rows, default B * T)
    void reserve_decode(const pk_config &cfg);
    // Capacity for ragged batches of <= max_clips clips, <= max_total_samples samples in all, no clip longer than max_clip_samples
    // (own_pcm: with a PCM buffer of its own).  level: what the batches start from (RagBatch::level; 1: max_total_samples / max_clip_samples
    // count mel FRAMES, 2: encoder frames).
    void size_ragged(const pk_config &cfg, int max_clips, int64_t max_total, int64_t max_clip, bool own_pcm, int level = 0);
    // Makes `r` the batch of the next run: checks it against the capacity, uploads its tables on `s` (asynchronously, out of a pinned copy)
    // and forms the device views.  pos_T is patched in by Model::run_layers.
    void set_ragged(const RagBatch &r, hipStream_t s);
    void set_uniform(int B_, int64_t n_samples_);            // a uniform run inside the reserved capacity (n_samples_ = 0: keep)
    size_t rag_cap_rows = 0;                                  // packed encoder rows the buffers were reserved for (size_ragged / size_for)
    int rag_cap_clips = 0;
    int64_t rag_cap_samples = 0, rag_cap_clip = 0;
};

class StreamBatch;

class Model {
  public:
    Model(const std::string &weights_path, const std::string &vocab_path, const pk_config &cfg);
    // safetensors image in memory; borrow: no copy, the image must stay alive until to_gpu() has returned
    Model(const void *weights, size_t n_bytes, const std::string &vocab_path, const pk_config &cfg, bool borrow = false);
    ~Model();
    void to_gpu(int device);
    bool on_gpu() const { return device_ >= 0; }
    void require_gpu() const;

    pk_config cfg;
    Tokenizer tok;
    int device_ = -1;
    hipStream_t stream = nullptr;       // main stream (mel, encoder, and everything in the staged entry points)
    hipStream_t stream_dec = nullptr;   // high-priority stream of the decode loop in the pipelined batch path

    // device weights
    std::vector<void *> allocs_;
    MelTables mel{};
    SubW sub{};
    std::vector<LayerW> layers;
    DecW dec{};
    const float *wld = nullptr, *bld = nullptr;     // [V+D][J] label_proj rows then duration_proj rows (+ biases)
    // sigma-K-layout copies used by the decode GEMVs
    const float *wld_s = nullptr, *dec_wp_s = nullptr, *dec_whh_s[4] = {}, *dec_wih_s[4] = {};
    // bf16 copies (natural k order) for the decode GEMVs of the tolerance-class mode (kernels/decode_gemv_bf16.hip); null in fp32 mode
    const float *wld16 = nullptr, *dec_wp16 = nullptr, *dec_whh16[4] = {}, *dec_wih16[4] = {};

    // relative-position tables: sinusoidal pe [2T-1][d] (src/encoder.cpp:9-30, host float math) and the
    // per-layer pos_proj_(pe) [L][2T-1][d]; they depend on (T, weights) only, so they are rebuilt when T changes.
    // One resident table set per attention FORMAT (fp32 sigma columns / the bf16 kernel's): which kernel a batch runs depends on its longest
    // utterance (attn_bf16(T)), and a service alternating long and short batches in the bf16 mode must not rebuild every layer's table on
    // every batch (round-4 advisor finding).  A set holds 2 T - 1 rows; a shorter sequence of T' frames uses rows [T - T', T + T' - 1): identical values.
    struct PosTab {
        int T = 0;
        DevBuf proj;            // pos_proj_ of every layer [L][2T-1][d]
        DevBuf cvec;            // bf16 attention: (v_h - u_h) . P_p per (layer, head, p)
    } pos32, pos16;
    DevBuf pos_pe;
    const PosTab &pos_tab(int T) const { return attn_bf16(T) ? pos16 : pos32; }
    bool attn_bf16(int T) const;    // the bf16-MFMA attention kernel applies (gemm_bf16 mode, head size 64 / 128, strip + c band fit LDS)
    DevBuf att_scratch;         // score blocks of the attention kernel for sequences too long for LDS (grow-only)
    void ensure_pos_tables(int T, hipStream_t s);

    // stage drivers (device pointers, enqueue on `s`, never synchronise)
    void run_mel(const float *d_pcm, int B, int64_t n_samples, float *d_logmel, float *d_feats, hipStream_t s, const RagDev *rv = nullptr);
    // the whole path on a workspace whose batch (uniform or ragged) has been set: w.pcm-independent, PCM given by the caller
    void run_mel_ws(Workspace &w, const float *d_pcm, int B, hipStream_t s);
    void run_subsample(Workspace &w, const float *d_feats, int B, int Tm, float *d_x, hipStream_t s);
    void run_encoder(Workspace &w, const float *d_feats, int B, int Tm, int stop_layer, int stop_stage, hipStream_t s);  // -> w.x
    void run_layers(Workspace &w, int B, int first_layer, int stop_layer, int stop_stage, hipStream_t s);                 // w.x -> w.x
    void run_ctc(Workspace &w, const float *d_enc, int B, int T, bool want_logp, hipStream_t s);
    void run_tdt(Workspace &w, const float *d_enc, int B, int T, int max_tokens, hipStream_t s, bool keep_state = false);
    void run_enc_proj(const float *d_enc, int64_t rows, float *ep_out, hipStream_t s);                    // joint enc_proj_ of all frames
    void run_tdt_loop(Workspace &w, int B, int T, int max_tokens, hipStream_t s, bool keep_state = false);  // greedy loop over w.ep

    // Phrase boosting (reference include/parakeet/phrase_boost.hpp:22-57, TranscribeOptions.boost_phrases transcribe.hpp:41-42):
    // the ContextTrie of the tokenised phrases in CSR form on the device; CTC and TDT greedy decode then run boosted.
    // phrases.empty() switches boosting off.  Not to be changed while a pipelined batch is in flight.
    void set_boost(const std::vector<std::vector<int>> &phrases, float score);
    bool boost_on = false;
    float boost_score = 0.0f;
    int trie_nodes = 0;
    std::vector<std::vector<int>> boost_phrases;      // what the trie was built from (per-call overrides restore it)
    DevBuf trie_off, trie_tok, trie_node;
    TrieDev trie_dev(Workspace &w, int B);

    // Copies of the encoder layers' eight product weights tiled for the small-M chain kernel (kernels.hpp: GemmArgs::W_sig), built on first
    // use (streaming: StreamBatch) and shared by every stream of the model: + the size of the encoder weights in HBM.  Empty in gemm_bf16 mode
    // or when d / ffn are not multiples of 64.
    struct SigW { const float *ffn1_w1 = nullptr, *ffn1_w2 = nullptr, *ffn2_w1 = nullptr, *ffn2_w2 = nullptr, *wqkv = nullptr, *wo = nullptr, *pw1 = nullptr, *pw2 = nullptr; };
    const std::vector<SigW> &sigma_weights();
    std::vector<SigW> sig_layers_;
    DevBuf sig_buf_;
    bool sig_built_ = false;

    ProfileSink *prof = nullptr;
    void klaunch_begin(const char *name, double flops, double bytes, hipStream_t s);
    void klaunch_end(hipStream_t s);

    // grow-only scratch shared by the host-buffer entry points
    DevBuf io_in, io_out, io_tmp;
    Workspace ws;       // workspace of the host-buffer stage entry points
    int dec_nt_weights = 0;                    // decode GEMVs stream their weights with non-temporal loads (SkinnyArgs::nt_weights)
    int decode_loop = PK_DECODE_LOOP_PHASES;   // pk_model_set_decode_loop: how run_tdt_loop issues the greedy loop
    int *h_done = nullptr;   // pinned host word for the decode loop's "all utterances finished" poll
    // the two-stream batch pipeline of the one-call API (struct pk_batch, capi.cpp), owned by the model; freed first in ~Model
    void *pipe = nullptr;
    void (*pipe_free)(void *) = nullptr;

    const float *upload(const float *host, size_t n);
    const float *upload_tensor(const std::string &name, std::vector<int64_t> expect_shape);
    const float *upload_gemm_weight(const float *host, size_t n);                                   // fp32, or bf16 when cfg.gemm_bf16
    const float *upload_gemm_tensor(const std::string &name, std::vector<int64_t> expect_shape);
    void run_gemm(const char *name, const GemmArgs &g, int epi, hipStream_t s, bool fp32_weight = false);
    const HostTensor &host_tensor(const std::string &name, const std::vector<int64_t> &expect_shape);
    float *dev_alloc(size_t n_floats);

    friend class StreamBatch;

  private:
    std::unique_ptr<SafeTensors> st_;
    void validate_config();
    void build_mel_tables();
    void upload_weights();
    void gemm(const char *name, const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, float *out, int64_t ldo,
              int M, int N, int K, int epi, const float *resid, int64_t ldr, float alpha, hipStream_t s, int a_bf16 = 0, int out_bf16 = 0);
    // LayerNorm + product, folded into one launch where the fp32 chain kernel can (engine.cpp)
    bool ln_folds(const GemmArgs &g, int epi, int64_t rows) const;
    // ... or, large fp32 batches, applied from per-row statistics while the tile kernel stages A (GemmArgs::ln_stats; engine.cpp)
    bool ln_stats_folds(const GemmArgs &g, int epi, const float *ng, const float *nb, const float *x, const float *stats) const;
    void ln_gemm(const char *name, const GemmArgs &g, int epi, const float *ng, const float *nb, int norm_state, int ymode, const float *x, float *n,
                 int64_t rows, hipStream_t s);
    void ffn(Workspace &w, const LayerW &L, bool second, int64_t rows, hipStream_t s, int norm_state = 0, const SigW *sg = nullptr);
};

// thread-local error slot of the C ABI
void set_last_error(const std::string &msg);

}  // namespace pk

// the opaque handle of the C ABI (include/parakeet_amd.h)
struct pk_model {
    std::unique_ptr<pk::Model> m;
};
