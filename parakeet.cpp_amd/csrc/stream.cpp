// parakeet.cpp_amd/csrc/stream.cpp -- the streaming path (BASELINE configs[4]: N concurrent streams per GPU with cached
// encoder state): StreamingAudioPreprocessor::process_chunk (src/audio.cpp:195-259), StreamingFastConformerEncoder::
// forward_chunk (src/streaming_encoder.cpp:430-472) and rnnt_streaming_decode_chunk (src/eou.cpp:17-98), the three calls of
// NemotronTranscriber::transcribe_chunk / StreamingTranscriber::transcribe_chunk (src/nemotron.cpp:24-52, src/eou.cpp:113-146).
//
// One pk_stream = S streams advanced in LOCK-STEP (every push hands each stream the same number of samples, so all S share
// the chunk geometry: frames produced, leftovers, cache lengths) -- the batch dimension of every kernel.  All state lives in
// HBM (K/V caches [S][left][d] and conv caches [S][K-1][d] per layer, leftover mel frames, LSTM state, last token); the host
// keeps only the pre-emphasis carry and the < 560 overlap samples per stream.  The per-chunk work is tiny (1-3 encoder frames
// per stream): FFN / projection products run on the MFMA GEMM with M = S*c rows, the cached attention and causal conv are
// exact-chain VALU kernels (kernels/stream.hip).  Bit-identical to the oracle's Stream (tests/test_gpu_stream.py).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>

#include "stream.hpp"

namespace pk {

// Fold the LayerNorm of a product's input rows into the product (GemmArgs::ln_g; tolerance-class mode: kernels/gemm_smallm_bf16.hip, exact mode:
// gemm_smallm_ln_kernel in kernels/gemm_smallm.hip, bit for bit) -- four of a block's fifteen launches go.  EXPERIMENTAL builds: PK_STREAM_FUSE_LN=0 switches it off for the A/B of tools/experiments/stream_bf16_ab.sh.
// The depthwise conv + BatchNorm + SiLU of the conv module in the GLU epilogue of pw1 (kernels.hpp: DwTail) -- one launch less per block,
// bit-identical to the separate kernel (both modes).  EXPERIMENTAL builds: PK_STREAM_FUSE_DW=0 switches it off.
static bool stream_fuse_dw() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_STREAM_FUSE_DW"); return e ? atoi(e) != 0 : true; }();
    return on;
#else
    return true;
#endif
}
// Tolerance-class mode: fc1 -> fc2 activations in the small-M bf16 kernel's 8-row operand tiles.  EXPERIMENTAL builds: PK_STREAM_ACT_TILES=0 keeps rows.
static bool stream_act_tiles() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_STREAM_ACT_TILES"); return e ? atoi(e) != 0 : true; }();
    return on;
#else
    return true;
#endif
}
// A block's final_norm_ folded into the first product of the NEXT block (GemmArgs::pre_g: that product normalises twice and
// its first column tile writes the normalised rows -- the next block's residual stream -- into the other of two buffers): one launch less per block
// (both modes; the exact mode's kernel normalises exactly as the separate launch does: bit-identical).
// EXPERIMENTAL builds: PK_STREAM_FUSE_FIN=0 keeps the separate LayerNorm launch.
static bool stream_fuse_final() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_STREAM_FUSE_FIN"); return e ? atoi(e) != 0 : true; }();
    return on;
#else
    return true;
#endif
}
// The block chain of a steady-state chunk as a hipGraph (stream.hpp EncGraph): built, correct (the streaming fixtures pass on it) and SLOWER on this
// runtime -- median 1.79 -> 1.82 ms, p95 1.81 -> 2.6 ms per 16-session chunk (profiles/r05_stream_graph_ab.txt): hipGraphLaunch of ~250 kernel nodes costs
// more host time than the launches it replaces, the same finding as for the decode loop's graph in round 3.  EXPERIMENTAL builds: PK_STREAM_GRAPH=1.
static bool stream_graph() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_STREAM_GRAPH"); return e ? atoi(e) != 0 : false; }();
    return on;
#else
    return false;
#endif
}
static bool stream_fuse_ln() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_STREAM_FUSE_LN"); return e ? atoi(e) != 0 : true; }();
    return on;
#else
    return true;
#endif
}

StreamBatch::StreamBatch(Model &m, int n_streams, int att_left, int att_right) : S(n_streams), m_(m), left_(att_left), right_(att_right) {
    m_.require_gpu();
    if (S <= 0 || att_left < 0 || att_right < 0) fail(PK_ERR_INVALID, "n_streams / attention context");
    if (m_.cfg.rnnt_head) fail(PK_ERR_UNSUPPORTED, "streaming decode is the TDT-joint loop of src/eou.cpp (label + duration heads)");
    const int d = m_.cfg.hidden_size, F = m_.cfg.mel_bins, K = m_.cfg.conv_kernel_size;
    mel_cache_.reserve((size_t)S * 8 * F * 4);
    for (int l = 0; l < m_.cfg.num_layers; ++l) {
        layers_.push_back(std::make_unique<LayerState>());
        LayerState &L = *layers_.back();
        for (int i = 0; i < 2; ++i) {
            L.k[i].reserve((size_t)S * (left_ > 0 ? left_ : 1) * d * 4);
            L.v[i].reserve((size_t)S * (left_ > 0 ? left_ : 1) * d * 4);
            L.conv[i].reserve((size_t)S * (K - 1) * d * 4);
        }
    }
    // (the tiled weight copies are built by the first chunk whose row count takes the small-M kernels, encode_device: a session set that never
    //  qualifies, e.g. 128 streams x 2 frames in the bf16 mode, never pays for them -- round-5 advisor finding)
    dec_cap_frames_ = 64;                                           // encoder frames per chunk the decode workspace is sized for
    wd_.size_for(m_.cfg, S, 0, 8 * (dec_cap_frames_ - 1) + 1);
    reset();
}

void StreamBatch::reset() {
    m_.require_gpu();
    preemph_last_.assign(S, 0.0f);
    overlap_.assign(S, {});
    n_mel_cache_ = 0;
    frame_offset_ = 0;
    for (auto &L : layers_) { L->cur = L->ccur = 0; L->n_kv = 0; L->has_conv = 0; }
    const int Hp = m_.cfg.pred_hidden, Ln = m_.cfg.num_lstm_layers;
    PK_HIP(hipMemset(wd_.h.p, 0, (size_t)Ln * S * Hp * 4));
    PK_HIP(hipMemset(wd_.c.p, 0, (size_t)Ln * S * Hp * 4));
    std::vector<int> tok(S, m_.cfg.blank_id);                       // last_token = blank (src/eou.cpp:33-35)
    PK_HIP(hipMemcpy(wd_.ints.p, tok.data(), (size_t)S * 4, hipMemcpyHostToDevice));
}

// ---- StreamingAudioPreprocessor::process_chunk --------------------------------------------------------------------------------
int StreamBatch::mel(const float *pcm, int n, float *out, int cap_frames) { return mel_impl(pcm, n, out, cap_frames, true); }

int StreamBatch::mel_impl(const float *pcm, int n, float *out, int cap_frames, bool sync) {
    m_.require_gpu();
    if (n <= 0) fail(PK_ERR_INVALID, "n_samples");
    const int F = m_.cfg.mel_bins;
    const int total = (int)overlap_[0].size() + n;
    if ((size_t)S * total > pin_pcm_floats_) {                      // (no upload of an earlier call is in flight: every entry point ends synchronised)
        if (pin_pcm_) { PK_HIP(hipHostFree(pin_pcm_)); pin_pcm_ = nullptr; pin_pcm_floats_ = 0; }
        const size_t want = (size_t)S * total + (size_t)S * 1024;
        PK_HIP(hipHostMalloc(reinterpret_cast<void **>(&pin_pcm_), want * 4, hipHostMallocDefault));
        pin_pcm_floats_ = want;
    }
    float *const buf = pin_pcm_;
    for (int s = 0; s < S; ++s) {                                   // 1. pre-emphasis with the carried sample, 2. prepend the overlap (:204-214)
        float *b = buf + (size_t)s * total;
        memcpy(b, overlap_[s].data(), overlap_[s].size() * 4);
        float last = preemph_last_[s];
        const float *src = pcm + (size_t)s * n;
        for (int i = 0; i < n; ++i) {
            const float cur = src[i];
            const float t = 0.97f * last;
            b[overlap_[s].size() + i] = cur - t;
            last = cur;
        }
        preemph_last_[s] = last;
    }
    const int n_frames = total < 400 ? 0 : (total - 400) / 160 + 1;
    if (n_frames <= 0) {                                            // :216-228 buffer everything
        for (int s = 0; s < S; ++s) overlap_[s].assign(buf + (size_t)s * total, buf + (size_t)(s + 1) * total);
        return 0;
    }
    if (out && n_frames > cap_frames) fail(PK_ERR_INVALID, "mel output holds %d frames, chunk produces %d", cap_frames, n_frames);
    const int consumed = (n_frames - 1) * 160 + 400;               // :230-231
    for (int s = 0; s < S; ++s) overlap_[s].assign(buf + (size_t)s * total + consumed, buf + (size_t)(s + 1) * total);
    pre_.reserve((size_t)S * consumed * 4);
    mel_dev_.reserve((size_t)S * n_frames * F * 4);
    hipStream_t st = m_.stream;
    PK_HIP(hipMemcpy2DAsync(pre_.p, (size_t)consumed * 4, buf, (size_t)total * 4, (size_t)consumed * 4, S, hipMemcpyHostToDevice, st));
    launch_mel_stream(pre_.as<float>(), S, consumed, n_frames, m_.mel, mel_dev_.as<float>(), st);
    PK_CHECK_LAUNCH();
    if (out) PK_HIP(hipMemcpyAsync(out, mel_dev_.p, (size_t)S * n_frames * F * 4, hipMemcpyDeviceToHost, st));
    if (sync || out) PK_HIP(hipStreamSynchronize(st));              // (a push synchronises once, when it fetches the chunk's tokens)
    return n_frames;
}

// pos_proj_ of every layer for the position table of length Tp = att_left + c (src/streaming_encoder.cpp:452-454, :214), natural columns
const float *StreamBatch::pos_table(int Tp) {
    auto it = pos_tables_.find(Tp);
    if (it != pos_tables_.end()) return it->second->as<float>();
    const int d = m_.cfg.hidden_size, P = 2 * Tp - 1;
    std::vector<float> pe((size_t)P * d);
    for (int p = 0; p < P; ++p) {                                   // sinusoidal_position_embedding (src/encoder.cpp:9-30), float math
        const float position = (float)(Tp - 1 - p);
        for (int i = 0; i < d; i += 2) {
            const float div_term = std::exp((float)i * (-std::log(10000.0f) / (float)d));
            pe[(size_t)p * d + i] = std::sin(position * div_term);
            if (i + 1 < d) pe[(size_t)p * d + i + 1] = std::cos(position * div_term);
        }
    }
    DevBuf tmp;
    tmp.reserve(pe.size() * 4);
    auto tab = std::make_unique<DevBuf>();
    tab->reserve((size_t)m_.cfg.num_layers * P * d * 4);
    hipStream_t st = m_.stream;
    PK_HIP(hipMemcpyAsync(tmp.p, pe.data(), pe.size() * 4, hipMemcpyHostToDevice, st));
    for (int l = 0; l < m_.cfg.num_layers; ++l) {
        GemmArgs g{tmp.as<float>(), d, m_.layers[l].wpos, d, nullptr, tab->as<float>() + (size_t)l * P * d, d, nullptr, 0, 1.0f, P, d, d};
        m_.run_gemm("pos_proj", g, EPI_NONE, st);
    }
    PK_HIP(hipStreamSynchronize(st));
    const float *r = tab->as<float>();
    pos_tables_[Tp] = std::move(tab);
    return r;
}

// ---- StreamingFastConformerEncoder::forward_chunk ---------------------------------------------------------------------------------
int StreamBatch::encode_device(const float *d_mel, int n_frames) {
    const pk_config &cfg = m_.cfg;
    const int F = cfg.mel_bins, d = cfg.hidden_size, K = cfg.conv_kernel_size;
    hipStream_t st = m_.stream;
    // CausalConvSubsampling::forward_cached (:348-385): prepend the leftover frames, consume a multiple of 8, keep the rest
    const int total = n_mel_cache_ + n_frames, consumable = (total / 8) * 8, leftover = total - consumable;
    // The tiled weight copies (Model::sigma_weights: +1.2 / 2.4 GB for the 600M models) are built by the first chunk whose row count takes the small-M kernels
    // -- HERE, before any carried state (mel leftovers, caches) is touched: an allocation failure leaves the streams exactly where they were.
    // (rows of this chunk = S * consumable / 8: the three stride-2 stages of a multiple of 8 frames)
    if (!sig_ && consumable > 0 && (int64_t)S * (consumable / 8) <= (cfg.gemm_bf16 ? kSmallMRowsBf16 : kSmallMRows)) sig_ = &m_.sigma_weights();
    mel_all_.reserve((size_t)S * (total > 0 ? total : 1) * F * 4);
    const size_t rowb = (size_t)F * 4;
    if (n_mel_cache_ > 0)
        PK_HIP(hipMemcpy2DAsync(mel_all_.p, (size_t)total * rowb, mel_cache_.p, 8 * rowb, (size_t)n_mel_cache_ * rowb, S, hipMemcpyDeviceToDevice, st));
    PK_HIP(hipMemcpy2DAsync((char *)mel_all_.p + (size_t)n_mel_cache_ * rowb, (size_t)total * rowb, d_mel, (size_t)n_frames * rowb,
                            (size_t)n_frames * rowb, S, hipMemcpyDeviceToDevice, st));
    if (leftover > 0)
        PK_HIP(hipMemcpy2DAsync(mel_cache_.p, 8 * rowb, (char *)mel_all_.p + (size_t)consumable * rowb, (size_t)total * rowb, (size_t)leftover * rowb, S,
                                hipMemcpyDeviceToDevice, st));
    n_mel_cache_ = leftover;
    if (consumable == 0) return 0;
    // forward() on the consumable frames: the offline conv stack on a contiguous [S][consumable][F] tensor
    enc_in_.reserve((size_t)S * consumable * F * 4);
    PK_HIP(hipMemcpy2DAsync(enc_in_.p, (size_t)consumable * rowb, mel_all_.p, (size_t)total * rowb, (size_t)consumable * rowb, S, hipMemcpyDeviceToDevice, st));
    ws_.size_for(cfg, S, 0, consumable);
    const int c = ws_.T;
    float *x = ws_.x.as<float>(), *n = ws_.n.as<float>();
    m_.run_subsample(ws_, enc_in_.as<float>(), S, consumable, x, st);
    const int Tp = left_ + c, P = 2 * Tp - 1;
    const float *ptab = pos_table(Tp);
    const int64_t rows = (int64_t)S * c;
    const int cache_rows = left_ > 0 ? left_ : 1;
    // rows <= kSmallMRows: every product of the chunk is a gemm_smallm chain -- run them on the sigma-K weight copies with sigma-K activations (the
    // producers below write that layout; x, the residual stream, stays natural)
    const bool have_sig = sig_ && !sig_->empty();                  // (built at the top of this function by the first chunk that qualifies)
    const int sg = (!cfg.gemm_bf16 && rows <= kSmallMRows && have_sig) ? 1 : 0;
    // (tolerance-class mode: the copies are the bf16 operand tiles of the small-M bf16 kernel, GemmArgs::W_t16 -- a weight load reads one contiguous KB)
    const bool wt = cfg.gemm_bf16 && rows <= kSmallMRowsBf16 && have_sig;
    // Tolerance-class mode (pk_config.gemm_bf16; specification: the oracle's Stream in its gemm_bf16 mode): every product of the chunk takes bf16
    // operands (kernels/gemm_smallm_bf16.hip for these few rows); the rows that exist only as GEMM operands -- LayerNorm outputs, the fc1
    // activations -- are stored as bf16 by their producers (RNE, the rounding the GEMM would apply: same operand values, half the bytes);
    // attention, depthwise conv and the caches stay fp32 arithmetic on the products' fp32 outputs.
    const int a16 = cfg.gemm_bf16 ? 1 : 0;
    const int lnm = a16 ? 1 : (sg ? 2 : 0);                                                               // launch_layernorm's output mode
    const int f = cfg.ffn_intermediate;
    float *hb = ws_.hbuf.as<float>();
    // LayerNorm(x) -> n, then the product on n -- or, where the small-M bf16 kernel can fold the norm in, the product straight on x
    // the previous block's final norm, when it rides on this block's first product (set at the end of a block, consumed by the next ffn1 fc1)
    const float *pend_g = nullptr, *pend_b = nullptr;
    float *x_other = nullptr;
    auto ln_gemm = [&](const char *name, const GemmArgs &g, int epi, const float *ng, const float *nb, bool norm_done) {
        if (!norm_done && stream_fuse_ln()) {
            GemmArgs fg = g;
            fg.A = x; fg.lda = d; fg.a_bf16 = 0; fg.a_sigma = 0; fg.ln_g = ng; fg.ln_b = nb; fg.ln_eps = 1e-5f;
            if (pend_g) {                                            // (checked when it was set: gemm_smallm_bf16_pre_applies)
                fg.pre_g = pend_g; fg.pre_b = pend_b; fg.pre_out = x_other; fg.pre_ldo = d;
                m_.run_gemm(name, fg, epi, st);
                std::swap(x, x_other);                               // the normalised rows are the residual stream from here on
                pend_g = pend_b = nullptr;
                return;
            }
            // tolerance-class mode: gemm_smallm_bf16.hip; exact mode (tiled weight copies present): gemm_smallm_ln_kernel -- bit for bit norm + product
            if (a16 ? gemm_smallm_bf16_ln_applies(fg, epi) : (sg && gemm_smallm_ln_applies(fg, epi))) { m_.run_gemm(name, fg, epi, st); return; }
        }
        if (!norm_done) launch_layernorm(x, rows, d, ng, nb, 1e-5f, n, st, lnm);
        m_.run_gemm(name, g, epi, st);
    };
    bool ln_folds = false;                                                                                // the next block's ffn1 norm will be folded into its fc1
    if (stream_fuse_ln()) {
        GemmArgs pg{x, d, m_.layers[0].ffn1_w1, d, nullptr, hb, f, nullptr, 0, 1.0f, (int)rows, f, d};
        pg.ln_g = m_.layers[0].ffn1_ng; pg.ln_b = m_.layers[0].ffn1_nb; pg.out_bf16 = a16;
        if (sg) pg.W_sig = (*sig_)[0].ffn1_w1;
        if (wt) pg.W_t16 = (*sig_)[0].ffn1_w1;
        ln_folds = a16 ? gemm_smallm_bf16_ln_applies(pg, EPI_SILU) : (sg && gemm_smallm_ln_applies(pg, EPI_SILU));
    }
    bool fin_folds = false;                                                                               // ... and the block's final norm with it
    if (ln_folds && stream_fuse_final() && cfg.num_layers > 1) {
        x_alt_.reserve((size_t)rows * d * 4);
        x_other = x_alt_.as<float>();
        GemmArgs pg{x, d, m_.layers[1].ffn1_w1, d, nullptr, hb, f, nullptr, 0, 1.0f, (int)rows, f, d};
        pg.ln_g = m_.layers[1].ffn1_ng; pg.ln_b = m_.layers[1].ffn1_nb; pg.out_bf16 = a16; pg.ln_eps = 1e-5f;
        if (sg) pg.W_sig = (*sig_)[1].ffn1_w1;
        pg.pre_g = m_.layers[0].fin_g; pg.pre_b = m_.layers[0].fin_b; pg.pre_out = x_other; pg.pre_ldo = d;
        fin_folds = a16 ? gemm_smallm_bf16_pre_applies(pg, EPI_SILU) : (sg && gemm_smallm_pre_applies(pg, EPI_SILU));
    }
    auto ffn = [&](const LayerW &L, const Model::SigW &Ls, bool second, bool norm_done) {                // FeedForward (src/encoder.cpp:36-46)
        GemmArgs g1{n, d, second ? L.ffn2_w1 : L.ffn1_w1, d, second ? L.ffn2_b1 : L.ffn1_b1, hb, f, nullptr, 0, 1.0f, (int)rows, f, d};
        g1.a_sigma = sg; (wt ? g1.W_t16 : g1.W_sig) = second ? Ls.ffn2_w1 : Ls.ffn1_w1;
        g1.a_bf16 = a16; g1.out_bf16 = a16;
        // tolerance-class mode, both products on the small-M bf16 kernel: the fc1 activations in its 8-row operand tiles (GemmArgs::out_t8 / a_t8)
        GemmArgs p1 = g1, p2{hb, f, second ? L.ffn2_w2 : L.ffn1_w2, f, nullptr, x, d, x, d, 0.5f, (int)rows, d, f};
        p1.out_t8 = 1; p2.a_bf16 = 1; p2.a_t8 = 1;
        const bool t8 = a16 && stream_act_tiles() && gemm_smallm_bf16_applies(p1, EPI_SILU) && gemm_smallm_bf16_applies(p2, EPI_RESID);
        g1.out_t8 = t8;
        g1.sigma_cols = sg ? f : 0;                                                                       // h is fc2's A operand
        ln_gemm("ffn_fc1_silu", g1, EPI_SILU, second ? L.ffn2_ng : L.ffn1_ng, second ? L.ffn2_nb : L.ffn1_nb, norm_done);
        GemmArgs g2{hb, f, second ? L.ffn2_w2 : L.ffn1_w2, f, second ? L.ffn2_b2 : L.ffn1_b2, x, d, x, d, 0.5f, (int)rows, d, f};
        g2.a_sigma = sg; (wt ? g2.W_t16 : g2.W_sig) = second ? Ls.ffn2_w2 : Ls.ffn1_w2;
        g2.a_bf16 = a16; g2.a_t8 = t8;
        m_.run_gemm("ffn_fc2_resid", g2, EPI_RESID, st);
    };
    // ---- hipGraph of the block chain (stream.hpp: EncGraph) ----
    bool steady = stream_graph() && !m_.prof && left_ > 0 && cfg.num_layers > 0;
    int parity = 0;
    std::vector<uintptr_t> gkey;
    if (steady) {                                                   // (EXPERIMENTAL builds with PK_STREAM_GRAPH=1 only: a production chunk skips all of this)
        parity = layers_[0]->cur;
        for (const auto &Lp : layers_) steady = steady && Lp->n_kv == left_ && Lp->has_conv == 1 && Lp->cur == parity && Lp->ccur == parity;
    }
    if (steady) {
        const void *ptrs[] = {ws_.x.p, ws_.n.p, ws_.hbuf.p, ws_.qkv.p, ws_.ctx.p, ws_.g.p, ws_.dwb.p, x_alt_.p, ptab, sig_, st, x_other};
        for (const void *q : ptrs) gkey.push_back(reinterpret_cast<uintptr_t>(q));
        const int ints[] = {S, c, (int)rows, left_, right_, sg, wt ? 1 : 0, a16, ln_folds ? 1 : 0, fin_folds ? 1 : 0, parity};
        for (int v : ints) gkey.push_back((uintptr_t)v);
        EncGraph &G = enc_graph_[parity];
        if (G.exec && G.key == gkey) {
            PK_HIP(hipGraphLaunch(G.exec, st));
            for (auto &Lp : layers_) { Lp->cur ^= 1; Lp->ccur ^= 1; }     // what the captured chain does to the host-side state (full caches stay full)
            return c;
        }
        if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
        PK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    }
    struct CaptureGuard {                                           // an exception inside the captured region must not leave the stream capturing
        hipStream_t st; bool on;
        ~CaptureGuard() { if (on) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(st, &g); if (g) (void)hipGraphDestroy(g); } }
    } cap_guard{st, steady};
    bool ffn1_norm_done = false;
    for (int l = 0; l < cfg.num_layers; ++l) {
        const LayerW &L = m_.layers[l];
        LayerState &Ls = *layers_[l];
        static const Model::SigW no_sig{};
        const Model::SigW &Sg = (sg || wt) ? (*sig_)[l] : no_sig;
        ffn(L, Sg, false, ffn1_norm_done);                                                             // ffn1_ (:294)
        // StreamingConformerAttention::forward_cached (:162-272)
        {
            GemmArgs g{n, d, L.wqkv, d, L.bqkv, ws_.qkv.as<float>(), 3 * d, nullptr, 0, 1.0f, (int)rows, 3 * d, d};
            g.a_sigma = sg; (wt ? g.W_t16 : g.W_sig) = Sg.wqkv;
            g.a_bf16 = a16;
            ln_gemm("attn_qkv", g, EPI_NONE, L.att_ng, L.att_nb, false);                              // natural columns (no sigma layout here)
        }
        const float *kc = Ls.k[Ls.cur].as<float>(), *vc = Ls.v[Ls.cur].as<float>();
        // attention of the chunk's rows + (same launch, extra blocks) the rotation of the K / V caches into the other buffer pair
        launch_stream_attention(ws_.qkv.as<float>(), kc, vc, cache_rows, S, c, Ls.n_kv, d, cfg.num_heads, ptab + (size_t)l * P * d, P, L.pos_u, L.pos_v,
                                left_, right_, ws_.ctx.as<float>(), st, Ls.k[Ls.cur ^ 1].as<float>(), Ls.v[Ls.cur ^ 1].as<float>(), left_, sg);
        Ls.cur ^= 1;
        Ls.n_kv = (Ls.n_kv + c > left_) ? left_ : Ls.n_kv + c;
        {
            GemmArgs g{ws_.ctx.as<float>(), d, L.wo, d, L.bo, x, d, x, d, 1.0f, (int)rows, d, d};
            g.a_sigma = sg; (wt ? g.W_t16 : g.W_sig) = Sg.wo;
            m_.run_gemm("attn_out_resid", g, EPI_RESID, st);
        }
        // CausalConformerConvModule::forward_cached (:41-78)
        {
            GemmArgs g{n, d, L.pw1_w, d, L.pw1_b, ws_.g.as<float>(), d, nullptr, 0, 1.0f, (int)rows, d, d};
            g.a_sigma = sg; (wt ? g.W_t16 : g.W_sig) = Sg.pw1;
            g.a_bf16 = a16;
            // the depthwise conv in pw1's epilogue where the small-M bf16 kernel can (rows stream-major, c = 1 / 2 / 4 frames per stream)
            DwTail tail{Ls.conv[Ls.ccur].as<float>(), Ls.conv[Ls.ccur ^ 1].as<float>(), Ls.has_conv, c, L.dw_w, L.dw_b, L.bn_mean, L.bn_rstd, L.bn_g, L.bn_b, sg};
            GemmArgs probe = g;                                                                          // what ln_gemm will launch when the norm folds
            if (stream_fuse_ln()) { probe.A = x; probe.a_bf16 = 0; probe.a_sigma = 0; probe.ln_g = L.cv_ng; probe.ln_b = L.cv_nb; probe.ln_eps = 1e-5f; }
            // (tolerance-class mode: with or without the folded norm; exact mode: the tail lives in the kernel with the norm folded in)
            const bool fused_dw = stream_fuse_dw() && (a16 ? gemm_smallm_bf16_dw_applies(probe, EPI_GLU, c, K) && gemm_smallm_bf16_dw_applies(g, EPI_GLU, c, K)
                                                           : sg && stream_fuse_ln() && gemm_smallm_dw_applies(probe, EPI_GLU, c, K));
            if (fused_dw) { g.dw_tail = &tail; g.out = ws_.dwb.as<float>(); }
            ln_gemm("conv_pw1_glu", g, EPI_GLU, L.cv_ng, L.cv_nb, false);
            if (!fused_dw)
                launch_stream_dwconv(ws_.g.as<float>(), Ls.conv[Ls.ccur].as<float>(), Ls.has_conv, S, c, d, K, L.dw_w, L.dw_b, L.bn_mean, L.bn_rstd, L.bn_g, L.bn_b,
                                     ws_.dwb.as<float>(), Ls.conv[Ls.ccur ^ 1].as<float>(), st, sg);
        }
        {
            Ls.ccur ^= 1;
            Ls.has_conv = 1;
        }
        {
            GemmArgs g{ws_.dwb.as<float>(), d, L.pw2_w, d, L.pw2_b, x, d, x, d, 1.0f, (int)rows, d, d};
            g.a_sigma = sg; (wt ? g.W_t16 : g.W_sig) = Sg.pw2;
            m_.run_gemm("conv_pw2_resid", g, EPI_RESID, st);
        }
        ffn(L, Sg, true, false);                                                                       // ffn2_
        if (l + 1 < cfg.num_layers && !ln_folds) {   // final_norm_ and the next block's ffn1_ norm in one pass over the rows (as the offline encoder)
            launch_layernorm2(x, rows, d, L.fin_g, L.fin_b, m_.layers[l + 1].ffn1_ng, m_.layers[l + 1].ffn1_nb, 1e-5f, x, n, st, lnm);
            ffn1_norm_done = true;
        } else if (l + 1 < cfg.num_layers && fin_folds) {
            pend_g = L.fin_g; pend_b = L.fin_b;      // final_norm_ rides on the next block's fc1 (ln_gemm above)
        } else {                                     // (ln_folds: the next block's fc1 normalises its own input rows)
            launch_layernorm(x, rows, d, L.fin_g, L.fin_b, 1e-5f, ws_.x.as<float>(), st);            // final_norm_ (the last block's lands in ws_.x)
            x = ws_.x.as<float>();
        }
    }
    if (steady) {                                                   // the chain was recorded, not run: instantiate, keep, launch
        hipGraph_t graph = nullptr;
        cap_guard.on = false;
        PK_HIP(hipStreamEndCapture(st, &graph));
        EncGraph &G = enc_graph_[parity];
        const hipError_t e = hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { G.exec = nullptr; fail(PK_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
        G.key = gkey;
        PK_HIP(hipGraphLaunch(G.exec, st));
    }
    PK_CHECK_LAUNCH();
    return c;
}

int StreamBatch::encode(const float *mel_in, int n_frames, float *enc, int cap_frames) {
    m_.require_gpu();
    if (n_frames <= 0) fail(PK_ERR_INVALID, "n_frames");
    const int F = m_.cfg.mel_bins, d = m_.cfg.hidden_size;
    mel_dev_.reserve((size_t)S * n_frames * F * 4);
    PK_HIP(hipMemcpyAsync(mel_dev_.p, mel_in, (size_t)S * n_frames * F * 4, hipMemcpyHostToDevice, m_.stream));
    const int c = encode_device(mel_dev_.as<float>(), n_frames);
    if (c > 0) {
        if (c > cap_frames) fail(PK_ERR_INVALID, "encoder output holds %d frames, chunk produces %d", cap_frames, c);
        PK_HIP(hipMemcpyAsync(enc, ws_.x.p, (size_t)S * c * d * 4, hipMemcpyDeviceToHost, m_.stream));
    }
    PK_HIP(hipStreamSynchronize(m_.stream));
    return c;
}

int StreamBatch::encode_keep(const float *mel_in, int n_frames, const float **d_enc) {
    m_.require_gpu();
    if (n_frames <= 0) fail(PK_ERR_INVALID, "n_frames");
    const int F = m_.cfg.mel_bins;
    mel_dev_.reserve((size_t)S * n_frames * F * 4);
    PK_HIP(hipMemcpyAsync(mel_dev_.p, mel_in, (size_t)S * n_frames * F * 4, hipMemcpyHostToDevice, m_.stream));
    const int c = encode_device(mel_dev_.as<float>(), n_frames);
    *d_enc = ws_.x.as<float>();
    return c;
}

// ---- rnnt_streaming_decode_chunk -------------------------------------------------------------------------------------------------------
void StreamBatch::decode_device(const float *d_enc, int c, int max_tokens) {
    if (c > dec_cap_frames_) fail(PK_ERR_UNSUPPORTED, "chunk of %d encoder frames exceeds the stream's decode workspace (%d)", c, dec_cap_frames_);
    if (max_tokens > wd_.max_tokens) fail(PK_ERR_INVALID, "max_tokens %d > %d", max_tokens, wd_.max_tokens);
    wd_.T = c;
    m_.run_tdt(wd_, d_enc, S, c, max_tokens, m_.stream, /*keep_state=*/true);
}

StreamBatch::~StreamBatch() {
    for (EncGraph &G : enc_graph_) if (G.exec) (void)hipGraphExecDestroy(G.exec);
    if (pin_tok_) (void)hipHostFree(pin_tok_);
    if (pin_pcm_) (void)hipHostFree(pin_pcm_);
}
void *StreamBatch::pinned_tokens(size_t bytes) {
    if (bytes > pin_tok_bytes_) {
        if (pin_tok_) { PK_HIP(hipHostFree(pin_tok_)); pin_tok_ = nullptr; pin_tok_bytes_ = 0; }
        PK_HIP(hipHostMalloc(&pin_tok_, bytes, hipHostMallocDefault));
        pin_tok_bytes_ = bytes;
    }
    return pin_tok_;
}

// `pin`: pinned staging of S * (1 + 4 * max_tokens) words (StreamBatch::pinned_tokens).  The copies are enqueued by the decode loop's poll hook
// (Workspace::before_poll: they ride on the poll's synchronisation -- one host round trip per chunk) or, when the loop did not end on a poll, here.
static void enqueue_token_copies(Workspace &wd, int S, int max_tokens, bool start, bool end, bool conf, void *pin, hipStream_t st) {
    const size_t nt = (size_t)S * max_tokens, nb = nt * 4;
    int32_t *p_lens = static_cast<int32_t *>(pin), *p_ids = p_lens + S, *p_start = p_ids + nt, *p_end = p_start + nt;
    float *p_conf = reinterpret_cast<float *>(p_end + nt);
    PK_HIP(hipMemcpyAsync(p_lens, wd.lens.p, (size_t)S * 4, hipMemcpyDeviceToHost, st));
    PK_HIP(hipMemcpyAsync(p_ids, wd.ids.p, nb, hipMemcpyDeviceToHost, st));
    if (start) PK_HIP(hipMemcpyAsync(p_start, wd.start.p, nb, hipMemcpyDeviceToHost, st));
    if (end) PK_HIP(hipMemcpyAsync(p_end, wd.end.p, nb, hipMemcpyDeviceToHost, st));
    if (conf) PK_HIP(hipMemcpyAsync(p_conf, wd.conf.p, nb, hipMemcpyDeviceToHost, st));
}
static void fetch_tokens(Model &m, Workspace &wd, int S, int max_tokens, int frame_offset, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end,
                         float *conf, void *pin) {
    hipStream_t st = m.stream;
    const size_t nt = (size_t)S * max_tokens, nb = nt * 4;
    int32_t *p_lens = static_cast<int32_t *>(pin), *p_ids = p_lens + S, *p_start = p_ids + nt, *p_end = p_start + nt;
    float *p_conf = reinterpret_cast<float *>(p_end + nt);
    if (!wd.poll_hit) {                                             // (the results are not on the host yet)
        enqueue_token_copies(wd, S, max_tokens, start != nullptr, end != nullptr, conf != nullptr, pin, st);
        PK_HIP(hipStreamSynchronize(st));
    }
    memcpy(lens, p_lens, (size_t)S * 4);
    memcpy(ids, p_ids, nb);
    if (start) memcpy(start, p_start, nb);
    if (end) memcpy(end, p_end, nb);
    if (conf) memcpy(conf, p_conf, nb);
    for (int s = 0; s < S; ++s) {
        if (lens[s] < 0) fail(PK_ERR_DECODE_CAP, "stream %d: TDT loop hit the safety cap", s);
        for (int i = 0; i < max_tokens; ++i) {
            const size_t o = (size_t)s * max_tokens + i;
            if (i < lens[s]) {                                      // frames relative to the stream (src/eou.cpp:77-79)
                if (start) start[o] += frame_offset;
                if (end) end[o] += frame_offset;
            } else {                                                // never hand stale device memory to the caller
                ids[o] = 0;
                if (start) start[o] = 0;
                if (end) end[o] = 0;
                if (conf) conf[o] = 0.0f;
            }
        }
    }
}

void StreamBatch::decode(const float *enc, int c, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    m_.require_gpu();
    if (c <= 0 || max_tokens <= 0) fail(PK_ERR_INVALID, "c / max_tokens");
    const int d = m_.cfg.hidden_size;
    enc_in_.reserve((size_t)S * c * d * 4);
    PK_HIP(hipMemcpyAsync(enc_in_.p, enc, (size_t)S * c * d * 4, hipMemcpyHostToDevice, m_.stream));
    void *pin = pinned_tokens((size_t)S * (1 + 4 * (size_t)max_tokens) * 4);
    {
        struct Hook { Workspace &w; ~Hook() { w.before_poll = nullptr; } } hook{wd_};
        wd_.poll_hit = false;
        wd_.before_poll = [&](hipStream_t st) { enqueue_token_copies(wd_, S, max_tokens, start != nullptr, end != nullptr, conf != nullptr, pin, st); };
        decode_device(enc_in_.as<float>(), c, max_tokens);
    }
    fetch_tokens(m_, wd_, S, max_tokens, frame_offset_, ids, lens, start, end, conf, pin);
    frame_offset_ += c;
}

void StreamBatch::score(const float *enc, int c, const int32_t *labels, const int32_t *dur_idx, const int32_t *n_steps, int cap, float *label_logp,
                        float *dur_logp, int32_t *n_done) {
    m_.require_gpu();
    if (c <= 0 || cap <= 0 || !enc || !labels || !dur_idx || !n_steps) fail(PK_ERR_INVALID, "enc / c / labels / dur_idx / n_steps / cap");
    if (c > dec_cap_frames_) fail(PK_ERR_UNSUPPORTED, "chunk of %d encoder frames exceeds the stream's decode workspace (%d)", c, dec_cap_frames_);
    const int d = m_.cfg.hidden_size, V = m_.cfg.vocab_size, D = m_.cfg.rnnt_head ? 0 : m_.cfg.num_durations;
    if (V <= 0 || D <= 0) fail(PK_ERR_UNSUPPORTED, "pk_stream_score needs a TDT joint (label + duration heads)");
    for (int s = 0; s < S; ++s) {
        if (n_steps[s] < 0 || n_steps[s] > cap) fail(PK_ERR_INVALID, "n_steps[%d] = %d: 0 .. cap", s, n_steps[s]);
        for (int k = 0; k < n_steps[s]; ++k) {
            const int32_t l = labels[(size_t)s * cap + k], di = dur_idx[(size_t)s * cap + k];
            if (l < 0 || l >= V || di < 0 || di >= D) fail(PK_ERR_INVALID, "stream %d step %d: label / duration index out of range", s, k);
        }
    }
    hipStream_t st = m_.stream;
    const size_t nk = (size_t)S * cap;
    enc_in_.reserve((size_t)S * c * d * 4);
    force_.reserve((2 * nk + S) * sizeof(int));
    score_.reserve(nk * (size_t)(V + D) * 4);
    int *d_lab = force_.as<int>(), *d_dur = d_lab + nk, *d_n = d_dur + nk;
    float *d_sl = score_.as<float>(), *d_sd = d_sl + nk * V;
    PK_HIP(hipMemcpyAsync(enc_in_.p, enc, (size_t)S * c * d * 4, hipMemcpyHostToDevice, st));
    PK_HIP(hipMemcpyAsync(d_lab, labels, nk * 4, hipMemcpyHostToDevice, st));
    PK_HIP(hipMemcpyAsync(d_dur, dur_idx, nk * 4, hipMemcpyHostToDevice, st));
    PK_HIP(hipMemcpyAsync(d_n, n_steps, (size_t)S * 4, hipMemcpyHostToDevice, st));
    PK_HIP(hipMemsetAsync(d_sl, 0, nk * (size_t)(V + D) * 4, st));
    struct Scope {
        Workspace &w;
        ~Scope() { w.force_label = w.force_dur = w.n_force_b = nullptr; w.score_lab = w.score_dur = nullptr; w.n_force = w.force_stride = 0; }
    } scope{wd_};
    wd_.force_label = d_lab; wd_.force_dur = d_dur; wd_.n_force_b = d_n; wd_.force_stride = cap; wd_.n_force = cap;
    wd_.score_lab = d_sl; wd_.score_dur = d_sd;
    wd_.T = c;
    m_.run_tdt(wd_, enc_in_.as<float>(), S, c, wd_.max_tokens, st, /*keep_state=*/true);
    PK_CHECK_LAUNCH();
    if (n_done) PK_HIP(hipMemcpyAsync(n_done, wd_.ints.as<int>() + 4 * S, (size_t)S * 4, hipMemcpyDeviceToHost, st));      // TdtState::steps
    if (label_logp) PK_HIP(hipMemcpyAsync(label_logp, d_sl, nk * (size_t)V * 4, hipMemcpyDeviceToHost, st));
    if (dur_logp) PK_HIP(hipMemcpyAsync(dur_logp, d_sd, nk * (size_t)D * 4, hipMemcpyDeviceToHost, st));
    PK_HIP(hipStreamSynchronize(st));
    frame_offset_ += c;
}

void StreamBatch::push(const float *pcm, int n_samples, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    // Validate BEFORE any carried state (pre-emphasis carry, overlap samples, caches) is touched: a rejected push must leave the streams
    // exactly where they were.  Bound on the encoder frames this push can produce: < 8 leftover mel frames + (559 overlap + n_samples) / 160 + 1
    // new ones, / 8, + 1 -- at most dec_cap_frames_ when n_samples <= (dec_cap_frames_ - 2) * 8 hops.
    if (max_tokens <= 0 || max_tokens > wd_.max_tokens) fail(PK_ERR_INVALID, "max_tokens %d: 1 .. %d", max_tokens, wd_.max_tokens);
    if ((int64_t)n_samples > (int64_t)(dec_cap_frames_ - 2) * 8 * 160)
        fail(PK_ERR_UNSUPPORTED, "a push of %d samples exceeds the stream's decode workspace (%d encoder frames per chunk: at most %d samples)",
             n_samples, dec_cap_frames_, (dec_cap_frames_ - 2) * 8 * 160);
    for (int s = 0; s < S; ++s) lens[s] = 0;
    const int n_frames = mel_impl(pcm, n_samples, nullptr, 0, /*sync=*/false);
    if (n_frames == 0) return;
    const int c = encode_device(mel_dev_.as<float>(), n_frames);
    if (c == 0) { PK_HIP(hipStreamSynchronize(m_.stream)); return; }
    void *pin = pinned_tokens((size_t)S * (1 + 4 * (size_t)max_tokens) * 4);
    {
        struct Hook { Workspace &w; ~Hook() { w.before_poll = nullptr; } } hook{wd_};
        wd_.poll_hit = false;
        wd_.before_poll = [&](hipStream_t st) { enqueue_token_copies(wd_, S, max_tokens, start != nullptr, end != nullptr, conf != nullptr, pin, st); };
        decode_device(ws_.x.as<float>(), c, max_tokens);
    }
    fetch_tokens(m_, wd_, S, max_tokens, frame_offset_, ids, lens, start, end, conf, pin);
    frame_offset_ += c;
}

}  // namespace pk

using namespace pk;

struct pk_stream { std::unique_ptr<StreamBatch> s; };

template <class F>
static pk_status stream_guard(F &&fn) {
    try {
        fn();
        return PK_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return PK_ERR_INVALID;
    }
}

extern "C" {

pk_status pk_stream_create(pk_model *m, int n_streams, int att_context_left, int att_context_right, pk_stream **out) {
    return stream_guard([&] {
        if (!m || !out) fail(PK_ERR_INVALID, "invalid argument: model/out");
        auto h = std::make_unique<pk_stream>();
        h->s = std::make_unique<StreamBatch>(*m->m, n_streams, att_context_left, att_context_right);
        *out = h.release();
    });
}
void pk_stream_free(pk_stream *s) { delete s; }
pk_status pk_stream_reset(pk_stream *s) {
    return stream_guard([&] { if (!s) fail(PK_ERR_INVALID, "stream"); s->s->reset(); });
}
pk_status pk_stream_push(pk_stream *s, const float *pcm, int n_samples, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start,
                         int32_t *end, float *conf) {
    return stream_guard([&] {
        if (!s || !pcm || !ids || !lens || n_samples <= 0 || max_tokens <= 0) fail(PK_ERR_INVALID, "invalid argument: stream/pcm/ids/lens/sizes");
        s->s->push(pcm, n_samples, max_tokens, ids, lens, start, end, conf);
    });
}
pk_status pk_stream_mel(pk_stream *s, const float *pcm, int n_samples, float *out, int cap_frames, int *n_frames) {
    return stream_guard([&] {
        if (!s || !pcm || !out || !n_frames) fail(PK_ERR_INVALID, "invalid argument: stream/pcm/out/n_frames");
        *n_frames = s->s->mel(pcm, n_samples, out, cap_frames);
    });
}
pk_status pk_stream_encode(pk_stream *s, const float *mel, int n_frames, float *enc, int cap_frames, int *n_out) {
    return stream_guard([&] {
        if (!s || !mel || !enc || !n_out) fail(PK_ERR_INVALID, "invalid argument: stream/mel/enc/n_out");
        *n_out = s->s->encode(mel, n_frames, enc, cap_frames);
    });
}
pk_status pk_stream_decode(pk_stream *s, const float *enc, int n_frames, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start,
                           int32_t *end, float *conf) {
    return stream_guard([&] {
        if (!s || !enc || !ids || !lens) fail(PK_ERR_INVALID, "invalid argument: stream/enc/ids/lens");
        s->s->decode(enc, n_frames, max_tokens, ids, lens, start, end, conf);
    });
}
pk_status pk_stream_score(pk_stream *s, const float *enc, int n_frames, const int32_t *labels, const int32_t *dur_idx, const int32_t *n_steps,
                          int cap, float *label_logp, float *dur_logp, int32_t *n_done) {
    return stream_guard([&] {
        if (!s) fail(PK_ERR_INVALID, "stream");
        s->s->score(enc, n_frames, labels, dur_idx, n_steps, cap, label_logp, dur_logp, n_done);
    });
}

}  // extern "C"
