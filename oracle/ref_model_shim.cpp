// oracle/ref_model_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" wrappers around the REAL reference model code: /root/reference/src/{audio,encoder,lstm,rnnt,tdt,ctc,tdt_ctc,
// transformer,streaming_encoder,eou,nemotron,sortformer,phrase_boost,vocab,timestamp}.cpp and include/parakeet/transcribe.hpp,
// compiled where they lie by oracle/Makefile against the CPU stand-in for the un-vendored tensor library
// (oracle/axiom_stub/axiom/*.hpp) into oracle/_ref/libpk_ref_model.so.  Nothing here restates the reference: every function
// below converts plain arrays to tensors, calls the reference's own function / class, and copies the result out.
// Used by tests/test_*_vs_reference.py to pin oracle/pk_oracle.c (and through it the HIP path) to the reference's own code.
#include <cstdio>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>

#include "parakeet/audio.hpp"
#include "parakeet/ctc.hpp"
#include "parakeet/encoder.hpp"
#include "parakeet/eou.hpp"
#include "parakeet/nemotron.hpp"
#include "parakeet/phrase_boost.hpp"
#include "parakeet/rnnt.hpp"
#include "parakeet/sortformer.hpp"
#include "parakeet/streaming_encoder.hpp"
#include "parakeet/tdt.hpp"
#include "parakeet/tdt_ctc.hpp"
#include "parakeet/transcribe.hpp"
#include "parakeet/transformer.hpp"

#include <axiom/fft.hpp>

using namespace parakeet;
using axiom::Shape;
using axiom::Tensor;

namespace {

thread_local std::string g_err;
int fail(const std::exception &e) {
    g_err = e.what();
    return -1;
}

enum Kind { K_TDTCTC = 0, K_TDT = 1, K_RNNT = 2, K_NEMOTRON = 3, K_EOU = 4, K_SORTFORMER = 5 };

// flat int config (oracle/refmodel.py:cfg_array writes it)
struct Cfg {
    int kind, mel_bins, sub_channels, hidden, layers, heads, ffn, conv_k;
    int vocab, pred_hidden, lstm_layers, joint_hidden, n_dur, dur[8];
    int ctc_vocab, att_left, att_right, xscaling, sub_relu;
    int tf_hidden, tf_layers, tf_heads, tf_ffn, tf_pre_ln, tf_final_norm, max_speakers;
};
static_assert(sizeof(Cfg) == 33 * sizeof(int), "Cfg layout");

void fill_encoder(EncoderConfig &e, const Cfg &c) {
    e.mel_bins = c.mel_bins;
    e.subsampling_channels = c.sub_channels;
    e.hidden_size = c.hidden;
    e.num_layers = c.layers;
    e.num_heads = c.heads;
    e.ffn_intermediate = c.ffn;
    e.conv_kernel_size = c.conv_k;
}
void fill_streaming(StreamingEncoderConfig &e, const Cfg &c) {
    fill_encoder(e, c);
    e.att_context_left = c.att_left;
    e.att_context_right = c.att_right;
    e.xscaling = c.xscaling != 0;
    e.subsampling_activation = c.sub_relu ? SubsamplingActivation::ReLU : SubsamplingActivation::SiLU;
}
template <class C> void fill_heads(C &cfg, const Cfg &c) {
    cfg.prediction.vocab_size = c.vocab;
    cfg.prediction.pred_hidden = c.pred_hidden;
    cfg.prediction.num_lstm_layers = c.lstm_layers;
    cfg.joint.encoder_hidden = c.hidden;
    cfg.joint.pred_hidden = c.pred_hidden;
    cfg.joint.joint_hidden = c.joint_hidden;
    cfg.joint.vocab_size = c.vocab;
}
std::vector<int> durations_of(const Cfg &c) { return std::vector<int>(c.dur, c.dur + c.n_dur); }

TDTCTCConfig make_tdtctc(const Cfg &c) {
    TDTCTCConfig cfg;
    fill_encoder(cfg.encoder, c);
    fill_heads(cfg, c);
    cfg.durations = durations_of(c);
    cfg.ctc_vocab_size = c.ctc_vocab;
    return cfg;
}
TDTConfig make_tdt(const Cfg &c) {
    TDTConfig cfg;
    fill_encoder(cfg.encoder, c);
    fill_heads(cfg, c);
    cfg.durations = durations_of(c);
    return cfg;
}
RNNTConfig make_rnnt(const Cfg &c) {
    RNNTConfig cfg;
    fill_encoder(cfg.encoder, c);
    fill_heads(cfg, c);
    return cfg;
}
NemotronConfig make_nemotron(const Cfg &c) {
    NemotronConfig cfg;
    fill_streaming(cfg.encoder, c);
    fill_heads(cfg, c);
    cfg.durations = durations_of(c);
    cfg.latency_frames = c.att_right;
    return cfg;
}
EOUConfig make_eou(const Cfg &c) {
    EOUConfig cfg;
    fill_streaming(cfg.encoder, c);
    fill_heads(cfg, c);
    cfg.durations = durations_of(c);
    return cfg;
}
SortformerConfig make_sortformer(const Cfg &c) {
    SortformerConfig cfg;
    fill_streaming(cfg.nest_encoder, c);
    cfg.encoder_hidden = c.hidden;
    cfg.transformer_hidden = c.tf_hidden;
    cfg.transformer.hidden_size = c.tf_hidden;
    cfg.transformer.num_layers = c.tf_layers;
    cfg.transformer.num_heads = c.tf_heads;
    cfg.transformer.ffn_intermediate = c.tf_ffn;
    cfg.transformer.pre_ln = c.tf_pre_ln != 0;
    cfg.transformer.has_final_norm = c.tf_final_norm != 0;
    cfg.max_speakers = c.max_speakers;
    return cfg;
}

struct RefModel {
    Cfg cfg;
    std::map<std::string, Tensor> weights;
    std::unique_ptr<ParakeetTDTCTC> tdtctc;
    std::unique_ptr<ParakeetTDT> tdt;
    std::unique_ptr<ParakeetRNNT> rnnt;
    std::unique_ptr<ParakeetNemotron> nemotron;
    std::unique_ptr<ParakeetEOU> eou;
    std::unique_ptr<Sortformer> sortformer;
    axiom::nn::Module *root = nullptr;

    RNNTPrediction &prediction() {
        switch (cfg.kind) {
        case K_TDTCTC: return tdtctc->prediction();
        case K_TDT: return tdt->prediction();
        case K_RNNT: return rnnt->prediction();
        case K_NEMOTRON: return nemotron->prediction();
        case K_EOU: return eou->prediction();
        }
        throw std::runtime_error("model has no prediction network");
    }
    TDTJoint &tdt_joint() {
        switch (cfg.kind) {
        case K_TDTCTC: return tdtctc->tdt_joint();
        case K_TDT: return tdt->joint();
        case K_NEMOTRON: return nemotron->joint();
        case K_EOU: return eou->joint();
        }
        throw std::runtime_error("model has no TDT joint");
    }
    StreamingFastConformerEncoder &streaming_encoder() {
        switch (cfg.kind) {
        case K_NEMOTRON: return nemotron->encoder();
        case K_EOU: return eou->encoder();
        }
        throw std::runtime_error("model has no streaming encoder");
    }
    Tensor encode(const Tensor &feats) {
        switch (cfg.kind) {
        case K_TDTCTC: return tdtctc->encoder()(feats);
        case K_TDT: return tdt->encoder()(feats);
        case K_RNNT: return rnnt->encoder()(feats);
        case K_NEMOTRON: return nemotron->encoder()(feats);
        case K_EOU: return eou->encoder()(feats);
        }
        throw std::runtime_error("model has no stand-alone encoder");
    }
    const char *encoder_prefix() const { return cfg.kind == K_SORTFORMER ? "nest_encoder_." : "encoder_."; }
};

Tensor t3(const float *p, int a, int b, int c) { return Tensor::from_data(p, Shape{(size_t)a, (size_t)b, (size_t)c}, true); }
void copy_out(const Tensor &t, float *out) {
    Tensor c = t.ascontiguousarray();
    std::memcpy(out, c.typed_data<float>(), c.numel() * sizeof(float));
}
// flatten per-utterance results into ids[B][max_tokens] + lens[B] (+ optional start / end / conf)
int put_ids(const std::vector<std::vector<int>> &r, int max_tokens, int32_t *ids, int32_t *lens) {
    for (size_t b = 0; b < r.size(); ++b) {
        if ((int)r[b].size() > max_tokens) {
            g_err = "token buffer too small";
            return -1;
        }
        lens[b] = (int32_t)r[b].size();
        for (size_t i = 0; i < r[b].size(); ++i) ids[b * (size_t)max_tokens + i] = r[b][i];
    }
    return 0;
}
int put_ts(const std::vector<std::vector<TimestampedToken>> &r, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end,
           float *conf) {
    for (size_t b = 0; b < r.size(); ++b) {
        if ((int)r[b].size() > max_tokens) {
            g_err = "token buffer too small";
            return -1;
        }
        lens[b] = (int32_t)r[b].size();
        for (size_t i = 0; i < r[b].size(); ++i) {
            const size_t o = b * (size_t)max_tokens + i;
            ids[o] = r[b][i].token_id;
            if (start) start[o] = r[b][i].start_frame;
            if (end) end[o] = r[b][i].end_frame;
            if (conf) conf[o] = r[b][i].confidence;
        }
    }
    return 0;
}

struct RefStream {
    RefModel *m;
    std::unique_ptr<StreamingAudioPreprocessor> prep;
    EncoderCache cache;
    StreamingDecodeState dec;
    AOSCCache aosc{4};
};

}  // namespace

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }
void ref_set_window_centered(int on) { axiom::fft::window_centered() = on != 0; }

// ───────── models ─────────
void *ref_model_new(const int *cfg_ints, const char *weights_path) {
    try {
        auto m = std::make_unique<RefModel>();
        std::memcpy(&m->cfg, cfg_ints, sizeof(Cfg));
        m->weights = axiom::io::safetensors::load(weights_path);
        const Cfg &c = m->cfg;
        switch (c.kind) {
        case K_TDTCTC: m->tdtctc = std::make_unique<ParakeetTDTCTC>(make_tdtctc(c)); m->root = m->tdtctc.get(); break;
        case K_TDT: m->tdt = std::make_unique<ParakeetTDT>(make_tdt(c)); m->root = m->tdt.get(); break;
        case K_RNNT: m->rnnt = std::make_unique<ParakeetRNNT>(make_rnnt(c)); m->root = m->rnnt.get(); break;
        case K_NEMOTRON: m->nemotron = std::make_unique<ParakeetNemotron>(make_nemotron(c)); m->root = m->nemotron.get(); break;
        case K_EOU: m->eou = std::make_unique<ParakeetEOU>(make_eou(c)); m->root = m->eou.get(); break;
        case K_SORTFORMER: m->sortformer = std::make_unique<Sortformer>(make_sortformer(c)); m->root = m->sortformer.get(); break;
        default: throw std::runtime_error("unknown model kind");
        }
        m->root->load_state_dict(m->weights, "", false);  // exactly the reference's call (transcribe.hpp:63)
        return m.release();
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}
void ref_model_free(void *h) { delete static_cast<RefModel *>(h); }

// "missing\n<key>\n...unexpected\n<key>\n..." of the non-strict load; returns the length needed
int ref_model_load_report(void *h, char *buf, int len) {
    auto *m = static_cast<RefModel *>(h);
    std::ostringstream os;
    os << "missing\n";
    for (auto &k : m->root->missing_keys()) os << k << "\n";
    os << "unexpected\n";
    for (auto &k : m->root->unexpected_keys()) os << k << "\n";
    const std::string s = os.str();
    if (buf && len > 0) {
        std::strncpy(buf, s.c_str(), (size_t)len - 1);
        buf[len - 1] = 0;
    }
    return (int)s.size() + 1;
}

// ───────── front end: preprocess_audio (src/audio.cpp:100-158) ─────────
int ref_preprocess_audio(const float *pcm, long long n, int n_mels, int normalize, float *out, int max_frames) {
    try {
        AudioConfig ac;
        ac.n_mels = n_mels;
        ac.normalize = normalize != 0;
        Tensor f = preprocess_audio(Tensor::from_data(pcm, Shape{(size_t)n}, true), ac);  // (1, n_frames, n_mels)
        const int nf = (int)f.shape()[1];
        if (nf > max_frames) throw std::runtime_error("feature buffer too small");
        copy_out(f, out);
        return nf;
    } catch (const std::exception &e) { return fail(e); }
}

int ref_pos_emb(int seq_len, int d_model, float *out) {
    try {
        copy_out(sinusoidal_position_embedding(seq_len, d_model), out);
        return 2 * seq_len - 1;
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── encoder pieces ─────────
// ConvSubsampling::forward (src/encoder.cpp:219-241) as a stand-alone module loaded from the model's own weights
int ref_subsampling(void *h, const float *feats, int B, int Tm, float *out, int max_rows) {
    try {
        auto *m = static_cast<RefModel *>(h);
        ConvSubsampling sub(m->cfg.sub_channels);
        sub.load_state_dict(m->weights, std::string(m->encoder_prefix()) + "subsampling_.", false);
        Tensor y = sub(t3(feats, B, Tm, m->cfg.mel_bins));
        const int T = (int)y.shape()[1];
        if (B * T > max_rows) throw std::runtime_error("output buffer too small");
        copy_out(y, out);
        return T;
    } catch (const std::exception &e) { return fail(e); }
}
// one ConformerBlock (src/encoder.cpp:196-204) on x[B][T][d] with the reference's own position table
int ref_conformer_block(void *h, int layer, const float *x, int B, int T, float *out) {
    try {
        auto *m = static_cast<RefModel *>(h);
        EncoderConfig ec;
        fill_encoder(ec, m->cfg);
        ConformerBlock blk(ec);
        blk.load_state_dict(m->weights, std::string(m->encoder_prefix()) + "layers_." + std::to_string(layer) + ".", false);
        copy_out(blk(t3(x, B, T, m->cfg.hidden), sinusoidal_position_embedding(T, m->cfg.hidden)), out);
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}
// FastConformerEncoder::forward (src/encoder.cpp:253-271) / StreamingFastConformerEncoder::forward (streaming_encoder.cpp:395-422)
int ref_encoder(void *h, const float *feats, int B, int Tm, float *out, int max_rows) {
    try {
        auto *m = static_cast<RefModel *>(h);
        Tensor y = m->encode(t3(feats, B, Tm, m->cfg.mel_bins));
        const int T = (int)y.shape()[1];
        if (B * T > max_rows) throw std::runtime_error("output buffer too small");
        copy_out(y, out);
        return T;
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── CTC ─────────
int ref_ctc_logprobs(void *h, const float *enc, int B, int T, float *out) {
    try {
        auto *m = static_cast<RefModel *>(h);
        if (m->cfg.kind != K_TDTCTC) throw std::runtime_error("model has no CTC head");
        copy_out(m->tdtctc->ctc_decoder()(t3(enc, B, T, m->cfg.hidden)), out);
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}
int ref_ctc_greedy(const float *logp, int B, int T, int V, int blank_id, int timestamps, int max_tokens, int32_t *ids, int32_t *lens,
                   int32_t *start, int32_t *end, float *conf) {
    try {
        Tensor lp = t3(logp, B, T, V);
        if (timestamps) return put_ts(ctc_greedy_decode_with_timestamps(lp, blank_id), max_tokens, ids, lens, start, end, conf);
        return put_ids(ctc_greedy_decode(lp, blank_id), max_tokens, ids, lens);
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── prediction net / joint single steps ─────────
// RNNTPrediction::step from a given state (src/rnnt.cpp:22-28, lstm.cpp:11-49): h, c are [L][Hp], updated in place
int ref_prediction_step(void *h_, int token, float *hs, float *cs, float *out) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        const int L = m->cfg.lstm_layers;
        const size_t Hp = (size_t)m->cfg.pred_hidden;
        std::vector<LSTMState> st(L);
        for (int l = 0; l < L; ++l) st[l] = {Tensor::from_data(hs + l * Hp, Shape{1, Hp}, true), Tensor::from_data(cs + l * Hp, Shape{1, Hp}, true)};
        auto tok = Tensor(Shape{1}, axiom::DType::Int32);
        tok.fill(token);
        Tensor y = m->prediction().step(tok, st);
        copy_out(y, out);
        for (int l = 0; l < L; ++l) {
            copy_out(st[l].first, hs + l * Hp);
            copy_out(st[l].second, cs + l * Hp);
        }
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}
// TDTJoint::forward (src/tdt.cpp:15-24) / RNNTJoint::forward (rnnt.cpp:37-44) on one encoder frame + one prediction vector
int ref_joint(void *h_, const float *enc_t, const float *pred, float *label_lp, float *dur_lp) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        Tensor e = t3(enc_t, 1, 1, m->cfg.hidden), p = t3(pred, 1, 1, m->cfg.pred_hidden);
        if (m->cfg.kind == K_RNNT) {
            copy_out(m->rnnt->joint().forward(e, p), label_lp);
        } else {
            auto o = m->tdt_joint().forward(e, p);
            copy_out(o.label_log_probs, label_lp);
            if (dur_lp) copy_out(o.duration_log_probs, dur_lp);
        }
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── greedy decoders ─────────
// tdt_greedy_decode / _with_timestamps (src/tdt.cpp:36-201)
int ref_tdt_greedy(void *h_, const float *enc, int B, int T, int blank_id, int max_symbols, int timestamps, int max_tokens, int32_t *ids,
                   int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        Tensor e = t3(enc, B, T, m->cfg.hidden);
        const auto dur = durations_of(m->cfg);
        if (timestamps)
            return put_ts(tdt_greedy_decode_with_timestamps(m->prediction(), m->tdt_joint(), e, dur, blank_id, max_symbols), max_tokens, ids,
                          lens, start, end, conf);
        return put_ids(tdt_greedy_decode(m->prediction(), m->tdt_joint(), e, dur, blank_id, max_symbols), max_tokens, ids, lens);
    } catch (const std::exception &e) { return fail(e); }
}
// rnnt_greedy_decode / _with_timestamps (src/rnnt.cpp:56-177)
int ref_rnnt_greedy(void *h_, const float *enc, int B, int T, int blank_id, int max_symbols, int timestamps, int max_tokens, int32_t *ids,
                    int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        if (m->cfg.kind != K_RNNT) throw std::runtime_error("not an RNNT model");
        Tensor e = t3(enc, B, T, m->cfg.hidden);
        if (timestamps)
            return put_ts(rnnt_greedy_decode_with_timestamps(*m->rnnt, e, blank_id, max_symbols), max_tokens, ids, lens, start, end, conf);
        return put_ids(rnnt_greedy_decode(*m->rnnt, e, blank_id, max_symbols), max_tokens, ids, lens);
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── phrase boosting (src/phrase_boost.cpp) ─────────
void *ref_trie_new() { return new ContextTrie(); }
void ref_trie_free(void *t) { delete static_cast<ContextTrie *>(t); }
void ref_trie_insert(void *t, const int32_t *ids, int n) { static_cast<ContextTrie *>(t)->insert(std::vector<int>(ids, ids + n)); }
int ref_trie_size(void *t) { return (int)static_cast<ContextTrie *>(t)->size(); }
int ref_trie_boosted_tokens(void *t, const int32_t *states, int n, unsigned char *flag, int V) {
    auto s = static_cast<ContextTrie *>(t)->get_boosted_tokens(std::unordered_set<int>(states, states + n));
    std::memset(flag, 0, (size_t)V);
    for (int v : s)
        if (v >= 0 && v < V) flag[v] = 1;
    return (int)s.size();
}
int ref_trie_advance(void *t, const int32_t *states, int n, int tok, int32_t *out) {
    auto s = static_cast<ContextTrie *>(t)->advance(std::unordered_set<int>(states, states + n), tok);
    std::vector<int> v(s.begin(), s.end());
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}
int ref_ctc_greedy_boosted(const float *logp, int B, int T, int V, int blank_id, void *trie, float boost, int timestamps, int max_tokens,
                           int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    try {
        Tensor lp = t3(logp, B, T, V);
        auto &tr = *static_cast<ContextTrie *>(trie);
        if (timestamps) return put_ts(ctc_greedy_decode_with_timestamps_boosted(lp, tr, boost, blank_id), max_tokens, ids, lens, start, end, conf);
        return put_ids(ctc_greedy_decode_boosted(lp, tr, boost, blank_id), max_tokens, ids, lens);
    } catch (const std::exception &e) { return fail(e); }
}
int ref_tdt_greedy_boosted(void *h_, const float *enc, int B, int T, int blank_id, int max_symbols, void *trie, float boost, int timestamps,
                           int max_tokens, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        Tensor e = t3(enc, B, T, m->cfg.hidden);
        const auto dur = durations_of(m->cfg);
        auto &tr = *static_cast<ContextTrie *>(trie);
        if (timestamps)
            return put_ts(tdt_greedy_decode_with_timestamps_boosted(m->prediction(), m->tdt_joint(), e, dur, tr, boost, blank_id, max_symbols),
                          max_tokens, ids, lens, start, end, conf);
        return put_ids(tdt_greedy_decode_boosted(m->prediction(), m->tdt_joint(), e, dur, tr, boost, blank_id, max_symbols), max_tokens, ids,
                       lens);
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── end to end: parakeet::Transcriber / TDTTranscriber (include/parakeet/transcribe.hpp) ─────────
struct RefTranscriber {
    std::unique_ptr<Transcriber> a;
    std::unique_ptr<TDTTranscriber> b;
};
void *ref_transcriber_new(const int *cfg_ints, const char *weights_path, const char *vocab_path) {
    try {
        Cfg c;
        std::memcpy(&c, cfg_ints, sizeof(Cfg));
        auto t = std::make_unique<RefTranscriber>();
        if (c.kind == K_TDTCTC) t->a = std::make_unique<Transcriber>(weights_path, vocab_path, make_tdtctc(c));
        else if (c.kind == K_TDT) t->b = std::make_unique<TDTTranscriber>(weights_path, vocab_path, make_tdt(c));
        else throw std::runtime_error("Transcriber: tdt-ctc or tdt model expected");
        return t.release();
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}
void ref_transcriber_free(void *t) { delete static_cast<RefTranscriber *>(t); }
// decoder: 0 = CTC, 1 = TDT.  boost_phrases: '\n'-separated, may be empty.  Returns the token count (or -1).
int ref_transcribe(void *t_, const float *pcm, long long n, int decoder, int timestamps, const char *boost_phrases, float boost_score,
                   int max_tokens, int32_t *ids, int32_t *start, int32_t *end, float *conf, char *text, int text_len, int *n_words) {
    try {
        auto *t = static_cast<RefTranscriber *>(t_);
        TranscribeOptions o;
        o.decoder = decoder == 0 ? Decoder::CTC : Decoder::TDT;
        o.timestamps = timestamps != 0;
        o.boost_score = boost_score;
        if (boost_phrases && *boost_phrases) {
            std::istringstream is(boost_phrases);
            std::string line;
            while (std::getline(is, line))
                if (!line.empty()) o.boost_phrases.push_back(line);
        }
        Tensor s = Tensor::from_data(pcm, Shape{(size_t)n}, true);
        TranscribeResult r;
        if (t->a) {
            r = t->a->transcribe(s, o);
        } else {
            if (decoder == 0) throw std::runtime_error("TDTTranscriber has no CTC decoder");
            r = o.boost_phrases.empty() ? t->b->transcribe(s, o.timestamps) : t->b->transcribe(s, o);
        }
        if ((int)r.token_ids.size() > max_tokens) throw std::runtime_error("token buffer too small");
        for (size_t i = 0; i < r.token_ids.size(); ++i) ids[i] = r.token_ids[i];
        for (size_t i = 0; i < r.timestamped_tokens.size(); ++i) {
            if (start) start[i] = r.timestamped_tokens[i].start_frame;
            if (end) end[i] = r.timestamped_tokens[i].end_frame;
            if (conf) conf[i] = r.timestamped_tokens[i].confidence;
        }
        if (text && text_len > 0) {
            std::strncpy(text, r.text.c_str(), (size_t)text_len - 1);
            text[text_len - 1] = 0;
        }
        if (n_words) *n_words = (int)r.word_timestamps.size();
        return (int)r.token_ids.size();
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── streaming: StreamingAudioPreprocessor, forward_chunk, rnnt_streaming_decode_chunk ─────────
void *ref_stream_new(void *h_) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        auto s = std::make_unique<RefStream>();
        s->m = m;
        AudioConfig ac;
        ac.n_mels = m->cfg.mel_bins;
        s->prep = std::make_unique<StreamingAudioPreprocessor>(ac);
        s->aosc = AOSCCache(m->cfg.max_speakers > 0 ? m->cfg.max_speakers : 4);
        return s.release();
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}
void ref_stream_free(void *s) { delete static_cast<RefStream *>(s); }
// StreamingAudioPreprocessor::process_chunk (src/audio.cpp:195-259) -> frames written ([frames][n_mels]); 0 = buffered
int ref_stream_mel(void *s_, const float *pcm, int n, float *out, int max_frames) {
    try {
        auto *s = static_cast<RefStream *>(s_);
        Tensor f = s->prep->process_chunk(Tensor::from_data(pcm, Shape{(size_t)n}, true));
        if (!f.storage()) return 0;
        const int nf = (int)f.shape()[1];
        if (nf > max_frames) throw std::runtime_error("feature buffer too small");
        copy_out(f, out);
        return nf;
    } catch (const std::exception &e) { return fail(e); }
}
// StreamingFastConformerEncoder::forward_chunk (src/streaming_encoder.cpp:430-472) -> encoder frames written; 0 = buffered
int ref_stream_encode(void *s_, const float *mel, int n_frames, float *out, int max_rows) {
    try {
        auto *s = static_cast<RefStream *>(s_);
        auto *m = s->m;
        Tensor x = t3(mel, 1, n_frames, m->cfg.mel_bins);
        Tensor y = m->streaming_encoder().forward_chunk(x, s->cache);
        if (!y.storage() || y.shape().size() == 0) return 0;
        const int c = (int)y.shape()[1];
        if (c > max_rows) throw std::runtime_error("output buffer too small");
        copy_out(y, out);
        return c;
    } catch (const std::exception &e) { return fail(e); }
}
// rnnt_streaming_decode_chunk (src/eou.cpp:17-98) -> new tokens of this chunk; start / end / conf are those of the new tokens
int ref_stream_decode(void *s_, const float *enc, int c, int blank_id, int max_symbols, int max_tokens, int32_t *ids, int32_t *start,
                      int32_t *end, float *conf) {
    try {
        auto *s = static_cast<RefStream *>(s_);
        auto *m = s->m;
        const size_t before = s->dec.timestamped_tokens.size();
        auto nt = rnnt_streaming_decode_chunk(m->prediction(), m->tdt_joint(), t3(enc, 1, c, m->cfg.hidden), durations_of(m->cfg), s->dec,
                                              blank_id, max_symbols);
        if ((int)nt.size() > max_tokens) throw std::runtime_error("token buffer too small");
        for (size_t i = 0; i < nt.size(); ++i) {
            const auto &tt = s->dec.timestamped_tokens[before + i];
            ids[i] = nt[i];
            if (start) start[i] = tt.start_frame;
            if (end) end[i] = tt.end_frame;
            if (conf) conf[i] = tt.confidence;
        }
        return (int)nt.size();
    } catch (const std::exception &e) { return fail(e); }
}

// NemotronTranscriber (src/nemotron.cpp:14-66) end to end over chunks
void *ref_nemotron_new(const int *cfg_ints, const char *weights_path, const char *vocab_path) {
    try {
        Cfg c;
        std::memcpy(&c, cfg_ints, sizeof(Cfg));
        return new NemotronTranscriber(weights_path, vocab_path, make_nemotron(c));
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}
void ref_nemotron_free(void *t) { delete static_cast<NemotronTranscriber *>(t); }
// returns the cumulative token count after this chunk; ids/start/end/conf receive ALL tokens so far
int ref_nemotron_chunk(void *t_, const float *pcm, int n, int max_tokens, int32_t *ids, int32_t *start, int32_t *end, float *conf, char *text,
                       int text_len) {
    try {
        auto *t = static_cast<NemotronTranscriber *>(t_);
        std::string piece = t->transcribe_chunk(pcm, (size_t)n);
        const auto &tt = t->get_timestamped_tokens();
        if ((int)tt.size() > max_tokens) throw std::runtime_error("token buffer too small");
        for (size_t i = 0; i < tt.size(); ++i) {
            ids[i] = tt[i].token_id;
            if (start) start[i] = tt[i].start_frame;
            if (end) end[i] = tt[i].end_frame;
            if (conf) conf[i] = tt[i].confidence;
        }
        if (text && text_len > 0) {
            std::strncpy(text, t->get_text().c_str(), (size_t)text_len - 1);
            text[text_len - 1] = 0;
        }
        return (int)tt.size();
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── Sortformer (src/sortformer.cpp) ─────────
int ref_sortformer_forward(void *h_, const float *feats, int B, int Tm, float *probs, int max_rows) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        if (m->cfg.kind != K_SORTFORMER) throw std::runtime_error("not a Sortformer model");
        Tensor p = m->sortformer->forward(t3(feats, B, Tm, m->cfg.mel_bins));
        const int T = (int)p.shape()[1];
        if (B * T > max_rows) throw std::runtime_error("output buffer too small");
        copy_out(p, probs);
        return T;
    } catch (const std::exception &e) { return fail(e); }
}
int ref_sortformer_diarize(void *h_, const float *feats, int Tm, int max_seg, int32_t *spk, float *start, float *end) {
    try {
        auto *m = static_cast<RefModel *>(h_);
        if (m->cfg.kind != K_SORTFORMER) throw std::runtime_error("not a Sortformer model");
        auto segs = m->sortformer->diarize(t3(feats, 1, Tm, m->cfg.mel_bins));
        if ((int)segs.size() > max_seg) throw std::runtime_error("segment buffer too small");
        for (size_t i = 0; i < segs.size(); ++i) {
            spk[i] = segs[i].speaker_id;
            start[i] = segs[i].start;
            end[i] = segs[i].end;
        }
        return (int)segs.size();
    } catch (const std::exception &e) { return fail(e); }
}
// Sortformer::diarize_chunk (:123-150): segments of this chunk; speaker arrival order so far in `order` (n_order out)
int ref_sortformer_chunk(void *s_, const float *feats, int n_frames, int max_seg, int32_t *spk, float *start, float *end, int32_t *order,
                         int *n_order) {
    try {
        auto *s = static_cast<RefStream *>(s_);
        auto *m = s->m;
        if (m->cfg.kind != K_SORTFORMER) throw std::runtime_error("not a Sortformer model");
        auto segs = m->sortformer->diarize_chunk(t3(feats, 1, n_frames, m->cfg.mel_bins), s->cache, s->aosc);
        if ((int)segs.size() > max_seg) throw std::runtime_error("segment buffer too small");
        for (size_t i = 0; i < segs.size(); ++i) {
            spk[i] = segs[i].speaker_id;
            start[i] = segs[i].start;
            end[i] = segs[i].end;
        }
        auto ord = s->aosc.speaker_order();
        for (size_t i = 0; i < ord.size(); ++i) order[i] = ord[i];
        *n_order = (int)ord.size();
        return (int)segs.size();
    } catch (const std::exception &e) { return fail(e); }
}

// ───────── plain Transformer encoder (src/transformer.cpp:15-88), stand-alone, weights under `prefix` ─────────
int ref_transformer_forward(const char *weights_path, const char *prefix, int hidden, int layers, int heads, int ffn, int pre_ln,
                            int final_norm, const float *x, int B, int T, float *out) {
    try {
        TransformerConfig tc;
        tc.hidden_size = hidden;
        tc.num_layers = layers;
        tc.num_heads = heads;
        tc.ffn_intermediate = ffn;
        tc.pre_ln = pre_ln != 0;
        tc.has_final_norm = final_norm != 0;
        TransformerEncoder enc(tc);
        auto w = axiom::io::safetensors::load(weights_path);
        enc.load_state_dict(w, prefix, false);
        copy_out(enc(t3(x, B, T, hidden)), out);
        return 0;
    } catch (const std::exception &e) { return fail(e); }
}

}  // extern "C"
