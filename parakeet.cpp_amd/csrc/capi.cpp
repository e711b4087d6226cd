// parakeet.cpp_amd/csrc/capi.cpp -- the extern "C" boundary declared in include/parakeet_amd.h.
// Every entry point translates pk::Error / std::exception into a status code + thread-local message.
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <exception>
#include <functional>
#include <thread>

#include "engine.hpp"
#include "rccl_dyn.hpp"

namespace pk {
const std::string &last_error();
void read_wav(const std::string &path, std::vector<float> &mono, int &sample_rate, int *n_channels = nullptr);
void parse_wav(const uint8_t *bytes, size_t n_bytes, const char *what, std::vector<float> &mono, int &sample_rate, int *n_channels, bool info_only, size_t file_len = 0);
size_t wav_info_frames();
void sinc_resample(const float *input, size_t input_len, int src_rate, int dst_rate, std::vector<float> &output);
}

using namespace pk;

static pk_status guard(const std::function<void()> &fn) {
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

static void need(bool ok, const char *what) {
    if (!ok) fail(PK_ERR_INVALID, "invalid argument: %s", what);
}

// ContextTrie::build (src/phrase_boost.cpp:29-37): Tokenizer::encode of every phrase
static std::vector<std::vector<int>> encode_phrases(Model &m, const char *const *phrases, int n) {
    if (!m.tok.loaded()) fail(PK_ERR_INVALID, "boost phrases need a vocabulary (the model was loaded without one)");
    std::vector<std::vector<int>> ph;
    for (int i = 0; i < n; ++i) {
        need(phrases[i] != nullptr, "phrases[i]");
        auto v = m.tok.encode(phrases[i]);
        if (!v.empty()) ph.push_back(std::move(v));               // ContextTrie::build skips phrases that encode to nothing
    }
    if (ph.empty() && n > 0) ph.emplace_back();                   // a root-only trie: boosting on, nothing boosted
    return ph;
}

// Token arrays come back as whole [B][pitch] blocks; the device only writes the first lens[b] entries of a row.  Zero the rest
// on the host so that a caller comparing / hashing whole arrays sees deterministic contents (never stale device memory).
template <class T>
static void zero_tail(T *a, const int32_t *lens, int B, int pitch) {
    if (!a) return;
    for (int b = 0; b < B; ++b) {
        const int n = lens[b] < 0 ? 0 : (lens[b] < pitch ? lens[b] : pitch);
        for (int i = n; i < pitch; ++i) a[(size_t)b * pitch + i] = T(0);
    }
}

extern "C" {

const char *pk_version(void) { return "parakeet.cpp_amd 0.1 (gfx950)"; }

size_t pk_last_error(char *buf, size_t cap) {
    const std::string &e = last_error();
    if (buf && cap) {
        const size_t n = e.size() < cap - 1 ? e.size() : cap - 1;
        memcpy(buf, e.data(), n);
        buf[n] = 0;
    }
    return e.size();
}

int pk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

pk_status pk_config_preset(const char *name, pk_config *out) {
    return guard([&] {
        need(name && out, "name/out");
        pk_config c;
        memset(&c, 0, sizeof c);
        c.mel_bins = 80; c.subsampling_channels = 256; c.num_heads = 8; c.conv_kernel_size = 9;
        c.pred_hidden = 640; c.joint_hidden = 640; c.max_symbols_per_step = 10;
        const std::string n = name;
        if (n == "tdt-ctc-110m") {              // make_110m_config, config.hpp:77-95
            c.hidden_size = 512; c.num_layers = 17; c.ffn_intermediate = 2048; c.vocab_size = 1025; c.num_lstm_layers = 1;
            c.num_durations = 5; c.ctc_vocab_size = 1025; c.blank_id = 1024;
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "tdt_joint_.");
        } else if (n == "tdt-600m") {           // make_tdt_600m_config, config.hpp:98-116
            c.mel_bins = 128; c.hidden_size = 1024; c.num_layers = 24; c.ffn_intermediate = 4096; c.vocab_size = 8193;
            c.num_lstm_layers = 2; c.num_durations = 5; c.ctc_vocab_size = 0; c.blank_id = 8192;
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "joint_.");
        } else if (n == "rnnt-600m") {          // make_rnnt_600m_config, config.hpp:119-135
            c.hidden_size = 1024; c.num_layers = 24; c.ffn_intermediate = 4096; c.vocab_size = 1025; c.num_lstm_layers = 2;
            c.num_durations = 0; c.ctc_vocab_size = 0; c.blank_id = 1024; c.rnnt_head = 1;
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "joint_.");
        } else if (n == "eou-120m") {           // make_eou_120m_config, eou.hpp:34-56 (streaming: use with pk_stream_*, context 70 / 1)
            c.hidden_size = 512; c.num_layers = 17; c.ffn_intermediate = 2048; c.vocab_size = 1025; c.num_lstm_layers = 1;
            c.num_durations = 5; c.ctc_vocab_size = 0; c.blank_id = 1024;
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "joint_.");
        } else if (n == "nemotron-600m") {      // make_nemotron_600m_config, nemotron.hpp:31-52 (streaming: use with pk_stream_*)
            c.hidden_size = 1024; c.num_layers = 24; c.ffn_intermediate = 4096; c.vocab_size = 8193; c.num_lstm_layers = 2;
            c.num_durations = 5; c.ctc_vocab_size = 0;
            c.blank_id = 1024;                  // transcribe_chunk decodes with the DEFAULT blank_id of eou.hpp:91-94 (nemotron.cpp:40-42): kept literally
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "joint_.");
        } else {
            fail(PK_ERR_INVALID, "unknown preset '%s'", name);
        }
        for (int i = 0; i < c.num_durations; ++i) c.durations[i] = i;
        *out = c;
    });
}

pk_status pk_model_load(const char *safetensors_path, const char *vocab_path, const pk_config *cfg, pk_model **out) {
    return guard([&] {
        need(safetensors_path && cfg && out, "path/cfg/out");
        auto h = std::make_unique<pk_model>();
        h->m = std::make_unique<Model>(safetensors_path, vocab_path ? vocab_path : "", *cfg);
        *out = h.release();
    });
}

pk_status pk_model_load_buffer(const void *safetensors_image, size_t n_bytes, const char *vocab_path, const pk_config *cfg, pk_model **out) {
    return guard([&] {
        need(safetensors_image && n_bytes > 0 && cfg && out, "image/n_bytes/cfg/out");
        auto h = std::make_unique<pk_model>();
        h->m = std::make_unique<Model>(safetensors_image, n_bytes, vocab_path ? vocab_path : "", *cfg);
        *out = h.release();
    });
}

pk_status pk_model_to_gpu(pk_model *m, int device) {
    return guard([&] { need(m, "model"); m->m->to_gpu(device); });
}

void pk_model_free(pk_model *m) { delete m; }

pk_status pk_model_set_decode_loop(pk_model *m, int mode) {
    return guard([&] {
        need(m, "model");
        need(mode == PK_DECODE_LOOP_PHASES || mode == PK_DECODE_LOOP_PERSISTENT || mode == PK_DECODE_LOOP_GRAPH, "mode");
        m->m->decode_loop = mode;
    });
}

pk_status pk_model_config(const pk_model *m, pk_config *out) {
    return guard([&] { need(m && out, "model/out"); *out = m->m->cfg; });
}

int pk_mel_num_frames(int64_t n_samples) { return (int)(1 + n_samples / 160); }
int pk_encoder_num_frames(int n) {
    for (int i = 0; i < 3; ++i) n = (n - 1) / 2 + 1;
    return n;
}

pk_status pk_mel(pk_model *h, const float *pcm, int n_clips, int64_t n_samples, float *feats, float *logmel) {
    return guard([&] {
        need(h && pcm && feats && n_clips > 0, "model/pcm/feats/n_clips");
        need(n_samples > 256, "n_samples must exceed n_fft/2 (reflect padding)");
        Model &m = *h->m;
        m.require_gpu();
        const int nf = pk_mel_num_frames(n_samples), F = m.cfg.mel_bins;
        const size_t n_in = (size_t)n_clips * n_samples, n_lm = (size_t)n_clips * F * nf;
        const int pitch = mel_logmel_pitch(nf);                      // the device's log-mel rows are padded to 16 frames (kernels.hpp)
        m.io_in.reserve(n_in * 4);
        m.io_tmp.reserve((size_t)n_clips * F * pitch * 4);
        m.io_out.reserve(n_lm * 4);
        PK_HIP(hipMemcpyAsync(m.io_in.p, pcm, n_in * 4, hipMemcpyHostToDevice, m.stream));
        m.run_mel(m.io_in.as<float>(), n_clips, n_samples, m.io_tmp.as<float>(), m.io_out.as<float>(), m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(feats, m.io_out.p, n_lm * 4, hipMemcpyDeviceToHost, m.stream));
        if (logmel) PK_HIP(hipMemcpy2DAsync(logmel, (size_t)nf * 4, m.io_tmp.p, (size_t)pitch * 4, (size_t)nf * 4, (size_t)n_clips * F, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}


pk_status pk_subsample(pk_model *h, const float *feats, int B, int Tm, float *out) {
    return guard([&] {
        need(h && feats && out && B > 0 && Tm > 0, "model/feats/out/B/Tm");
        Model &m = *h->m;
        m.require_gpu();
        m.ws.size_for(m.cfg, B, 0, Tm);
        const size_t nin = (size_t)B * Tm * m.cfg.mel_bins, nout = (size_t)B * m.ws.T * m.cfg.hidden_size;
        PK_HIP(hipMemcpyAsync(m.ws.feats.p, feats, nin * 4, hipMemcpyHostToDevice, m.stream));
        m.run_subsample(m.ws, m.ws.feats.as<float>(), B, Tm, m.ws.x.as<float>(), m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(out, m.ws.x.p, nout * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}

pk_status pk_encode(pk_model *h, const float *feats, int B, int Tm, int stop_layer, int stop_stage, float *enc) {
    return guard([&] {
        need(h && feats && enc && B > 0 && Tm > 0, "model/feats/enc/B/Tm");
        need(stop_stage >= 0 && stop_stage <= 4, "stop_stage");
        Model &m = *h->m;
        m.require_gpu();
        m.ws.size_for(m.cfg, B, 0, Tm);
        const size_t nin = (size_t)B * Tm * m.cfg.mel_bins, nout = (size_t)B * m.ws.T * m.cfg.hidden_size;
        PK_HIP(hipMemcpyAsync(m.ws.feats.p, feats, nin * 4, hipMemcpyHostToDevice, m.stream));
        m.run_encoder(m.ws, m.ws.feats.as<float>(), B, Tm, stop_layer, stop_stage, m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(enc, m.ws.x.p, nout * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}

static void size_ws_for_T(Model &m, int B, int T);


pk_status pk_conformer_blocks(pk_model *h, const float *x_in, int B, int T, int first_layer, int n_layers, float *x_out) {
    return guard([&] {
        need(h && x_in && x_out && B > 0 && T > 0, "model/x_in/x_out/B/T");
        Model &m = *h->m;
        m.require_gpu();
        need(first_layer >= 0 && n_layers >= 0 && first_layer + n_layers <= m.cfg.num_layers, "layer range");
        size_ws_for_T(m, B, T);
        const size_t n = (size_t)B * T * m.cfg.hidden_size;
        PK_HIP(hipMemcpyAsync(m.ws.x.p, x_in, n * 4, hipMemcpyHostToDevice, m.stream));
        if (n_layers > 0) m.run_layers(m.ws, B, first_layer, first_layer + n_layers, 0, m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(x_out, m.ws.x.p, n * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}

static void size_ws_for_T(Model &m, int B, int T) {
    // a mel length that subsamples to exactly T frames: Tm = 8(T-1)+1
    m.ws.size_for(m.cfg, B, 0, 8 * (T - 1) + 1);
    if (m.ws.T != T) fail(PK_ERR_INVALID, "internal: workspace T %d != %d", m.ws.T, T);
}

pk_status pk_ctc_decode(pk_model *h, const float *enc, int B, int T, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end,
                        float *conf, float *logp) {
    return guard([&] {
        need(h && enc && ids && lens && B > 0 && T > 0, "model/enc/ids/lens/B/T");
        Model &m = *h->m;
        m.require_gpu();
        size_ws_for_T(m, B, T);
        const size_t rows = (size_t)B * T;
        PK_HIP(hipMemcpyAsync(m.ws.x.p, enc, rows * m.cfg.hidden_size * 4, hipMemcpyHostToDevice, m.stream));
        m.run_ctc(m.ws, m.ws.x.as<float>(), B, T, logp != nullptr, m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(ids, m.ws.ids.p, rows * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipMemcpyAsync(lens, m.ws.lens.p, (size_t)B * 4, hipMemcpyDeviceToHost, m.stream));
        if (start) PK_HIP(hipMemcpyAsync(start, m.ws.start.p, rows * 4, hipMemcpyDeviceToHost, m.stream));
        if (end) PK_HIP(hipMemcpyAsync(end, m.ws.end.p, rows * 4, hipMemcpyDeviceToHost, m.stream));
        if (conf) PK_HIP(hipMemcpyAsync(conf, m.ws.conf.p, rows * 4, hipMemcpyDeviceToHost, m.stream));
        if (logp) PK_HIP(hipMemcpyAsync(logp, m.ws.ctc_lp.p, rows * m.cfg.ctc_vocab_size * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
        zero_tail(ids, lens, B, T); zero_tail(start, lens, B, T); zero_tail(end, lens, B, T); zero_tail(conf, lens, B, T);
    });
}

pk_status pk_tdt_decode(pk_model *h, const float *enc, int B, int T, int max_tokens, int32_t *ids, int32_t *lens, int32_t *start,
                        int32_t *end, float *conf, int32_t *steps) {
    pk_status cap_hit = PK_OK;
    pk_status st = guard([&] {
        need(h && enc && ids && lens && B > 0 && T > 0 && max_tokens > 0, "model/enc/ids/lens/B/T/max_tokens");
        Model &m = *h->m;
        m.require_gpu();
        size_ws_for_T(m, B, T);
        need(max_tokens <= m.ws.max_tokens, "max_tokens exceeds T * max_symbols_per_step");
        PK_HIP(hipMemcpyAsync(m.ws.x.p, enc, (size_t)B * T * m.cfg.hidden_size * 4, hipMemcpyHostToDevice, m.stream));
        m.run_tdt(m.ws, m.ws.x.as<float>(), B, T, max_tokens, m.stream);
        PK_CHECK_LAUNCH();
        const size_t tok = (size_t)B * max_tokens;
        PK_HIP(hipMemcpyAsync(ids, m.ws.ids.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipMemcpyAsync(lens, m.ws.lens.p, (size_t)B * 4, hipMemcpyDeviceToHost, m.stream));
        if (start) PK_HIP(hipMemcpyAsync(start, m.ws.start.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (end) PK_HIP(hipMemcpyAsync(end, m.ws.end.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (conf) PK_HIP(hipMemcpyAsync(conf, m.ws.conf.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (steps) PK_HIP(hipMemcpyAsync(steps, m.ws.ints.as<int>() + 4 * B, (size_t)B * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
        zero_tail(ids, lens, B, max_tokens); zero_tail(start, lens, B, max_tokens); zero_tail(end, lens, B, max_tokens); zero_tail(conf, lens, B, max_tokens);
        for (int b = 0; b < B; ++b)
            if (lens[b] < 0) cap_hit = PK_ERR_DECODE_CAP;
    });
    if (st == PK_OK && cap_hit != PK_OK) {
        set_last_error("TDT decode hit the safety cap on joint evaluations for at least one utterance (lens = -1)");
        return cap_hit;
    }
    return st;
}


/* ---- ragged (mixed-length) forms of the stage entry points: every tensor PACKED along the time axis ---------------------------------- */
static int att_block_rows_of(Model &m, int T_max) {
    return m.attn_bf16(T_max) ? relpos_attention_bf16_block_rows(m.cfg.hidden_size / m.cfg.num_heads) : 32;
}

pk_status pk_mel_ragged(pk_model *h, const float *pcm, const int64_t *offsets, int n_clips, float *feats, float *logmel) {
    return guard([&] {
        need(h && pcm && offsets && feats && n_clips > 0, "model/pcm/offsets/feats/n_clips");
        Model &m = *h->m;
        m.require_gpu();
        std::vector<int64_t> lens(n_clips);
        int64_t longest = 0;
        for (int i = 0; i < n_clips; ++i) { lens[i] = offsets[i + 1] - offsets[i]; longest = std::max(longest, lens[i]); }
        RagBatch r;
        r.build_from_samples(lens.data(), n_clips, 32);
        m.ws.size_ragged(m.cfg, n_clips, r.n_samples, longest, /*own_pcm=*/true);
        m.ws.set_ragged(r, m.stream);
        const size_t n_lm = (size_t)r.sum_Tm * m.cfg.mel_bins;
        for (int i = 0; i < n_clips; ++i)        // (the clips need not be contiguous in the caller's buffer)
            PK_HIP(hipMemcpyAsync(m.ws.pcm.as<float>() + r.pcm_off[i], pcm + offsets[i], (size_t)lens[i] * 4, hipMemcpyHostToDevice, m.stream));
        m.run_mel_ws(m.ws, m.ws.pcm.as<float>(), n_clips, m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(feats, m.ws.feats.p, n_lm * 4, hipMemcpyDeviceToHost, m.stream));
        if (logmel) {                                               // per clip: [mel_bins][pitch] on the device -> [mel_bins][Tm] for the caller
            const int F = m.cfg.mel_bins;
            size_t dev_off = 0, host_off = 0;
            for (int i = 0; i < n_clips; ++i) {
                const int tm = r.Tm[i], pitch = mel_logmel_pitch(tm);
                PK_HIP(hipMemcpy2DAsync(logmel + host_off, (size_t)tm * 4, m.ws.logmel.as<float>() + dev_off, (size_t)pitch * 4, (size_t)tm * 4, (size_t)F,
                                        hipMemcpyDeviceToHost, m.stream));
                dev_off += (size_t)F * pitch; host_off += (size_t)F * tm;
            }
        }
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}

pk_status pk_encode_ragged(pk_model *h, const float *feats, const int32_t *n_mel_frames, int B, int stop_layer, int stop_stage, float *enc) {
    return guard([&] {
        need(h && feats && n_mel_frames && enc && B > 0, "model/feats/n_mel_frames/enc/B");
        need(stop_stage >= 0 && stop_stage <= 4, "stop_stage");
        Model &m = *h->m;
        m.require_gpu();
        int tm_max = 0;
        for (int i = 0; i < B; ++i) tm_max = std::max(tm_max, (int)n_mel_frames[i]);
        RagBatch r;
        r.build_from_mel(n_mel_frames, B, att_block_rows_of(m, pk_encoder_num_frames(tm_max)));
        m.ws.size_ragged(m.cfg, B, r.sum_Tm, tm_max, false, /*level=*/1);
        m.ws.set_ragged(r, m.stream);
        PK_HIP(hipMemcpyAsync(m.ws.feats.p, feats, (size_t)r.sum_Tm * m.cfg.mel_bins * 4, hipMemcpyHostToDevice, m.stream));
        m.run_encoder(m.ws, m.ws.feats.as<float>(), B, 0, stop_layer, stop_stage, m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(enc, m.ws.x.p, (size_t)r.sum_T * m.cfg.hidden_size * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}

// workspace of the host-buffer decode entry points for B utterances of n_frames[b] encoder frames (packed); returns the longest
static int size_ws_for_frames(Model &m, const int32_t *n_frames, int B) {
    int t_max = 0;
    for (int i = 0; i < B; ++i) t_max = std::max(t_max, (int)n_frames[i]);
    RagBatch r;
    r.build_from_frames(n_frames, B, att_block_rows_of(m, t_max));
    m.ws.size_ragged(m.cfg, B, r.sum_T, t_max, false, /*level=*/2);
    m.ws.set_ragged(r, m.stream);
    return t_max;
}

pk_status pk_conformer_blocks_ragged(pk_model *h, const float *x_in, const int32_t *n_frames, int B, int first_layer, int n_layers, float *x_out) {
    return guard([&] {
        need(h && x_in && x_out && n_frames && B > 0, "model/x_in/x_out/n_frames/B");
        Model &m = *h->m;
        m.require_gpu();
        need(first_layer >= 0 && n_layers >= 0 && first_layer + n_layers <= m.cfg.num_layers, "layer range");
        size_ws_for_frames(m, n_frames, B);
        const size_t n = (size_t)m.ws.rag.sum_T * m.cfg.hidden_size;
        PK_HIP(hipMemcpyAsync(m.ws.x.p, x_in, n * 4, hipMemcpyHostToDevice, m.stream));
        if (n_layers > 0) m.run_layers(m.ws, B, first_layer, first_layer + n_layers, 0, m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(x_out, m.ws.x.p, n * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}

pk_status pk_ctc_decode_ragged(pk_model *h, const float *enc, const int32_t *n_frames, int B, int32_t *ids, int32_t *lens, int32_t *start,
                               int32_t *end, float *conf, float *logp) {
    return guard([&] {
        need(h && enc && n_frames && ids && lens && B > 0, "model/enc/n_frames/ids/lens/B");
        Model &m = *h->m;
        m.require_gpu();
        const int T = size_ws_for_frames(m, n_frames, B);              // the token arrays are [B][T], T = the longest utterance
        const size_t rows = (size_t)m.ws.rag.sum_T, tok = (size_t)B * T;
        PK_HIP(hipMemcpyAsync(m.ws.x.p, enc, rows * m.cfg.hidden_size * 4, hipMemcpyHostToDevice, m.stream));
        m.run_ctc(m.ws, m.ws.x.as<float>(), B, T, logp != nullptr, m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(ids, m.ws.ids.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipMemcpyAsync(lens, m.ws.lens.p, (size_t)B * 4, hipMemcpyDeviceToHost, m.stream));
        if (start) PK_HIP(hipMemcpyAsync(start, m.ws.start.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (end) PK_HIP(hipMemcpyAsync(end, m.ws.end.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (conf) PK_HIP(hipMemcpyAsync(conf, m.ws.conf.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (logp) PK_HIP(hipMemcpyAsync(logp, m.ws.ctc_lp.p, rows * m.cfg.ctc_vocab_size * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
        zero_tail(ids, lens, B, T); zero_tail(start, lens, B, T); zero_tail(end, lens, B, T); zero_tail(conf, lens, B, T);
    });
}

pk_status pk_tdt_decode_ragged(pk_model *h, const float *enc, const int32_t *n_frames, int B, int max_tokens, int32_t *ids, int32_t *lens,
                               int32_t *start, int32_t *end, float *conf, int32_t *steps) {
    pk_status cap_hit = PK_OK;
    pk_status st = guard([&] {
        need(h && enc && n_frames && ids && lens && B > 0 && max_tokens > 0, "model/enc/n_frames/ids/lens/B/max_tokens");
        Model &m = *h->m;
        m.require_gpu();
        const int T = size_ws_for_frames(m, n_frames, B);
        need(max_tokens <= m.ws.max_tokens, "max_tokens exceeds (longest utterance) * max_symbols_per_step");
        PK_HIP(hipMemcpyAsync(m.ws.x.p, enc, (size_t)m.ws.rag.sum_T * m.cfg.hidden_size * 4, hipMemcpyHostToDevice, m.stream));
        m.run_tdt(m.ws, m.ws.x.as<float>(), B, T, max_tokens, m.stream);
        PK_CHECK_LAUNCH();
        const size_t tok = (size_t)B * max_tokens;
        PK_HIP(hipMemcpyAsync(ids, m.ws.ids.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipMemcpyAsync(lens, m.ws.lens.p, (size_t)B * 4, hipMemcpyDeviceToHost, m.stream));
        if (start) PK_HIP(hipMemcpyAsync(start, m.ws.start.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (end) PK_HIP(hipMemcpyAsync(end, m.ws.end.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (conf) PK_HIP(hipMemcpyAsync(conf, m.ws.conf.p, tok * 4, hipMemcpyDeviceToHost, m.stream));
        if (steps) PK_HIP(hipMemcpyAsync(steps, m.ws.ints.as<int>() + 4 * B, (size_t)B * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
        zero_tail(ids, lens, B, max_tokens); zero_tail(start, lens, B, max_tokens); zero_tail(end, lens, B, max_tokens); zero_tail(conf, lens, B, max_tokens);
        for (int b = 0; b < B; ++b)
            if (lens[b] < 0) cap_hit = PK_ERR_DECODE_CAP;
    });
    if (st == PK_OK && cap_hit != PK_OK) {
        set_last_error("TDT decode hit the safety cap on joint evaluations for at least one utterance (lens = -1)");
        return cap_hit;
    }
    return st;
}

/* tdt_greedy_decode's loop (src/tdt.cpp:62-106) along a GIVEN decision path, recording the joint's outputs (TDTJoint::forward, :15-24) */
pk_status pk_tdt_score(pk_model *h, const float *enc, int T, const int32_t *labels, const int32_t *dur_idx, int n_steps, float *label_logp,
                       float *dur_logp, int *n_done) {
    return guard([&] {
        need(h && enc && labels && dur_idx && T > 0 && n_steps > 0 && (label_logp || dur_logp), "model/enc/labels/dur_idx/T/n_steps/outputs");
        Model &m = *h->m;
        m.require_gpu();
        need(m.cfg.vocab_size > 0 && !m.cfg.rnnt_head && m.cfg.num_durations > 0, "pk_tdt_score needs a TDT joint (label + duration heads)");
        const int V = m.cfg.vocab_size, D = m.cfg.num_durations;
        for (int k = 0; k < n_steps; ++k)
            need(labels[k] >= 0 && labels[k] < V && dur_idx[k] >= 0 && dur_idx[k] < D, "labels[k] / dur_idx[k] out of range");
        size_ws_for_T(m, 1, T);
        Workspace &w = m.ws;
        const size_t nl = (size_t)n_steps * V, nd = (size_t)n_steps * D;
        m.io_in.reserve((size_t)2 * n_steps * sizeof(int));
        m.io_out.reserve((nl + nd) * 4);
        int *d_lab = m.io_in.as<int>(), *d_dur = d_lab + n_steps;
        float *d_sl = m.io_out.as<float>(), *d_sd = d_sl + nl;
        PK_HIP(hipMemcpyAsync(d_lab, labels, (size_t)n_steps * 4, hipMemcpyHostToDevice, m.stream));
        PK_HIP(hipMemcpyAsync(d_dur, dur_idx, (size_t)n_steps * 4, hipMemcpyHostToDevice, m.stream));
        PK_HIP(hipMemsetAsync(d_sl, 0, (nl + nd) * 4, m.stream));
        PK_HIP(hipMemcpyAsync(w.x.p, enc, (size_t)T * m.cfg.hidden_size * 4, hipMemcpyHostToDevice, m.stream));
        struct Scope { Workspace &w; ~Scope() { w.force_label = w.force_dur = nullptr; w.score_lab = w.score_dur = nullptr; w.n_force = 0; } } scope{w};
        w.force_label = d_lab; w.force_dur = d_dur; w.n_force = n_steps; w.score_lab = d_sl; w.score_dur = d_sd;
        m.run_tdt(w, w.x.as<float>(), 1, T, w.max_tokens, m.stream);
        PK_CHECK_LAUNCH();
        int steps = 0;
        PK_HIP(hipMemcpyAsync(&steps, w.ints.as<int>() + 4, sizeof(int), hipMemcpyDeviceToHost, m.stream));      // st.steps[0] (B = 1)
        if (label_logp) PK_HIP(hipMemcpyAsync(label_logp, d_sl, nl * 4, hipMemcpyDeviceToHost, m.stream));
        if (dur_logp) PK_HIP(hipMemcpyAsync(dur_logp, d_sd, nd * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
        if (n_done) *n_done = steps;
    });
}

pk_status pk_decode_margins(pk_model *h, float *min_margin, int B) {
    return guard([&] {
        need(h && min_margin && B > 0, "model/min_margin/B");
        Model &m = *h->m;
        m.require_gpu();
        need(B <= m.ws.B && m.ws.margin.p, "B exceeds the last pk_tdt_decode call");
        need(!m.boost_on, "margins are reported for unboosted decodes");
        PK_HIP(hipMemcpy(min_margin, m.ws.margin.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    });
}

/* ---- resident batch pipeline ---------------------------------------------------------------------------- */
// Two workspaces + two streams: the latency-bound decode loop of batch k (high-priority stream, a few small kernels
// per step) runs concurrently with the MFMA-bound mel + encoder of batch k+1 (main stream).  pk_batch_run(k) enqueues
// encoder(k) and then drives decode(k-1); pk_batch_sync / pk_batch_results flush the decode still pending.
// A stream of DISTINCT batches keeps the overlap with pk_batch_upload_async (PCM double-buffered, copied on its own stream
// under the running encoder) + pk_batch_results_done (the batch whose decode finished inside the last pk_batch_run; no flush).
struct pk_batch {
    Model *m;
    DevBuf pcm2[2];             // [max_clips][n_samples] x 2: the buffer being read by mel(k) and the one upload(k+1) fills
    int cur = 0;                // buffer the next pk_batch_run reads
    int staged = -1;            // buffer filled by pk_batch_upload_async and not yet consumed by a run
    int staged_clips = 0;
    // What each PCM buffer holds: a uniform batch (clips x n_samples) or a RAGGED one (clips of different lengths packed back to back,
    // pk_batch_upload_ragged).  A pipeline created with pk_batch_create_ragged takes both, run by run, inside its capacity.
    struct Held { bool ragged = false; int64_t n_samples = 0; RagBatch rag; } held[2];
    bool rag_capacity = false;  // created with pk_batch_create_ragged (capacity in ws[].rag_cap_*)
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done[2], mel_done[2];
    bool mel_used[2] = {false, false};
    int slot_clips[2] = {0, 0}; // clips of the run that owns each workspace
    int last_clips = 0;         // clips of the newest finished results
    Workspace ws[2];
    int n_clips = 0;
    int runs = 0;               // pk_batch_run calls so far
    int pending_slot = -1, pending_decoder = -1;   // decode not yet driven
    int last_slot = -1, last_decoder = -1;         // where the newest finished results live
    hipEvent_t ev[4];
    hipEvent_t enc_done[2], dec_done[2];
    bool used[2] = {false, false};
    bool ev_ok = false;
    // Decode groups (pk_batch_set_decode_group): the TDT loops of `group` consecutive runs are driven as ONE lock-step batch.  The loop is
    // launch-bound (4 launches per symbol step whatever the batch), and every launch on the decode stream costs the encoder of the
    // following run ~2 us (profiles/r02_decode_persistent.md): a group of G cuts that by G.  enc_proj of run k goes into its rows of
    // grp[fill].ep on the ENCODER stream right after encoder(k) (the encoder workspaces are then free again); a full group is decoded
    // under the encoder of the run after it.  Results of run k are available once its group is decoded (pk_batch_results_back).
    int group = 1;
    bool overlap = true;        // pk_batch_set_decode_overlap: false = the decode loop runs on the encoder's stream, after it
    hipStream_t dec_stream() const { return overlap ? m->stream_dec : m->stream; }
    struct Member { int clips, row0; int64_t seq; };
    struct Group {
        Workspace w;                    // decode state of group * max_clips utterances
        std::vector<Member> mem;        // runs in this group, oldest first
        int rows = 0;                   // utterances so far
        int64_t ep_rows = 0;            // enc_proj rows so far (ragged-capacity pipelines: runs of different row counts)
        int T_max = 0;                  // longest utterance among the members (bounds the lock-step loop)
        DevBuf tabs;                    // ragged-capacity pipelines: Tb[cap] then row0[cap] of the group's utterances (TdtState::Tb / row0)
        hipEvent_t ep_done = nullptr, dec_done = nullptr;
        bool decoded = false, used = false;
    } grp[2];
    struct Loc { Workspace *w; int row0, clips, decoder; hipEvent_t ev; int64_t seq; };
    // Finished (decode driven) runs, oldest first.  An entry stays readable until the buffers it points into are recycled: a pipeline
    // slot when run k+2 is encoded into it, a group buffer when the group after next starts to fill it.  Nothing else removes entries, so a
    // flush that drives a full group AND the partial group behind it keeps the runs of both (round-2 advisor finding).
    std::vector<Loc> done;
    int64_t slot_seq[2] = {-1, -1};     // run index that owns each pipeline slot
    void forget(const Workspace *w) {   // the buffers of `w` are about to be overwritten
        done.erase(std::remove_if(done.begin(), done.end(), [w](const Loc &l) { return l.w == w; }), done.end());
    }
    int fill = 0;                       // group collecting runs
    int ready = -1;                     // full group whose decode has not been driven yet
};

// makes the batch held by the current PCM buffer the run of workspace w
static void batch_set_run(pk_batch *b, Workspace &w, hipStream_t s) {
    const pk_batch::Held &H = b->held[b->cur];
    if (H.ragged) w.set_ragged(H.rag, s);
    else if (b->rag_capacity) w.set_uniform(b->n_clips, H.n_samples);
}

static void batch_encode(pk_batch *b, int slot) {
    Model &m = *b->m;
    Workspace &w = b->ws[slot];
    hipStream_t s = m.stream;
    b->forget(&w);                                           // the results of run k-2 live in this slot: no longer readable
    b->slot_seq[slot] = b->runs;
    if (b->used[slot]) PK_HIP(hipStreamWaitEvent(s, b->dec_done[slot], 0));   // decode(k-2) must be done with this slot
    if (b->staged >= 0) {                                    // a batch uploaded under the previous run: switch buffers
        b->cur = b->staged;
        b->n_clips = b->staged_clips;
        b->staged = -1;
        PK_HIP(hipStreamWaitEvent(s, b->copy_done[b->cur], 0));
    }
    batch_set_run(b, w, s);                                  // uniform or ragged: what the PCM buffer holds (tables uploaded on s)
    m.run_mel_ws(w, b->pcm2[b->cur].as<float>(), b->n_clips, s);
    PK_HIP(hipEventRecord(b->mel_done[b->cur], s));         // the PCM buffer is free again once the mel kernels have read it
    b->mel_used[b->cur] = true;
    m.run_encoder(w, w.feats.as<float>(), b->n_clips, w.Tm, -1, 0, s);
    PK_HIP(hipEventRecord(b->enc_done[slot], s));
    b->used[slot] = true;
    b->slot_clips[slot] = b->n_clips;
}

static void batch_decode(pk_batch *b, int slot, int decoder, hipStream_t s) {
    Model &m = *b->m;
    Workspace &w = b->ws[slot];
    if (s != m.stream) PK_HIP(hipStreamWaitEvent(s, b->enc_done[slot], 0));
    const int nc = b->slot_clips[slot] > 0 ? b->slot_clips[slot] : b->n_clips;
    if (decoder == PK_DECODER_CTC) m.run_ctc(w, w.x.as<float>(), nc, w.T_run, false, s);
    else m.run_tdt(w, w.x.as<float>(), nc, w.T_run, w.max_tokens, s);
    PK_HIP(hipEventRecord(b->dec_done[slot], s));
    b->last_slot = slot;
    b->last_decoder = decoder;
    b->last_clips = nc;
    b->forget(&w);                                            // (the timed / profiled single-slot paths decode into a slot they did not encode)
    b->done.push_back({&w, 0, nc, decoder, b->dec_done[slot], b->slot_seq[slot]});
}

// host-driven TDT loop of a whole decode group on the decode stream
static void group_drive(pk_batch *b, int gi) {
    Model &m = *b->m;
    auto &G = b->grp[gi];
    hipStream_t s = b->dec_stream();
    if (s != m.stream) PK_HIP(hipStreamWaitEvent(s, G.ep_done, 0));
    m.run_tdt_loop(G.w, G.rows, b->rag_capacity ? G.T_max : G.w.T, G.w.max_tokens, s);
    PK_HIP(hipEventRecord(G.dec_done, s));
    G.decoded = true;
    for (auto &mm : G.mem) b->done.push_back({&G.w, mm.row0, mm.clips, PK_DECODER_TDT, G.dec_done, mm.seq});
    b->last_slot = 0;                                         // (a result exists)
    b->last_decoder = PK_DECODER_TDT;
    b->last_clips = G.mem.back().clips;
}

// closes the group being filled (full, or partial at a flush) and makes the other buffer the one to fill
static void group_close(pk_batch *b) {
    auto &G = b->grp[b->fill];
    PK_HIP(hipEventRecord(G.ep_done, b->m->stream));
    b->ready = b->fill;
    b->fill ^= 1;
    b->grp[b->fill].mem.clear();
}

static void batch_flush(pk_batch *b) {
    if (b->pending_slot >= 0) {
        const int slot = b->pending_slot, dec = b->pending_decoder;
        b->pending_slot = -1;
        batch_decode(b, slot, dec, b->dec_stream());
    }
    if (b->ready >= 0) { const int r = b->ready; b->ready = -1; group_drive(b, r); }
    if (b->group > 1 && !b->grp[b->fill].mem.empty()) {      // a partial group: decode what there is
        group_close(b);
        const int r = b->ready;
        b->ready = -1;
        group_drive(b, r);
    }
    PK_HIP(hipStreamSynchronize(b->m->stream_dec));
    PK_HIP(hipStreamSynchronize(b->m->stream));
}

static void batch_run(pk_batch *b, int decoder) {
    Model &m = *b->m;
    m.require_gpu();
    need(b->n_clips > 0 || b->staged >= 0, "pk_batch_upload() first");
    need(decoder == PK_DECODER_CTC || decoder == PK_DECODER_TDT, "decoder");
    const int slot = b->runs & 1;
    const bool grouped = b->group > 1 && decoder == PK_DECODER_TDT;
    if (!grouped && b->group > 1 && (b->ready >= 0 || !b->grp[b->fill].mem.empty())) batch_flush(b);   // decoder switch inside a group
    batch_encode(b, slot);                                   // encoder(k) is queued first ...
    if (b->pending_slot >= 0) {                              // ... then the host drives decode(k-1) while it runs
        const int ps = b->pending_slot, pd = b->pending_decoder;
        b->pending_slot = -1;
        batch_decode(b, ps, pd, b->dec_stream());
    }
    if (grouped) {
        auto &G = b->grp[b->fill];
        Workspace &w = b->ws[slot];
        if (G.mem.empty()) {                                 // first run of a group: the buffer's previous decode must be done with it
            if (G.used) PK_HIP(hipStreamWaitEvent(m.stream, G.dec_done, 0));
            b->forget(&G.w);                                 // the runs of the group decoded two groups ago are overwritten from here on
            G.rows = 0;
            G.ep_rows = 0;
            G.T_max = 0;
            G.decoded = false;
        }
        const int nc = b->slot_clips[slot];
        const int64_t run_rows = w.rows(nc);
        m.run_enc_proj(w.x.as<float>(), run_rows, G.w.ep.as<float>() + (size_t)G.ep_rows * m.cfg.joint_hidden, m.stream);
        if (b->rag_capacity) {
            // the group's utterances have their own frame counts / first enc_proj rows: gathered behind the earlier members' (TdtState::Tb / row0)
            int *Tb = G.tabs.as<int>(), *row0 = Tb + G.w.B;
            launch_rag_decode_tables(w.ragged ? w.rv.seq.T : nullptr, w.ragged ? w.rv.seq.T_off : nullptr, w.T_run, nc, (int)G.ep_rows, Tb + G.rows, row0 + G.rows,
                                     m.stream);
            G.w.dec_Tb = Tb; G.w.dec_row0 = row0;
            G.T_max = std::max(G.T_max, w.t_max());
        }
        G.mem.push_back({nc, G.rows, b->slot_seq[slot]});
        G.rows += nc;
        G.ep_rows += run_rows;
        G.used = true;
        const bool full = (int)G.mem.size() == b->group;
        if (b->ready >= 0) {                                 // the group completed by an earlier run: decode it under this encoder
            const int r = b->ready;
            b->ready = -1;
            group_drive(b, r);
        }
        if (full) group_close(b);
    } else {
        b->pending_slot = slot;
        b->pending_decoder = decoder;
    }
    b->runs += 1;
    PK_CHECK_LAUNCH();
}

// (re)sizes the two pipeline slots for batches of up to max_clips clips of n_samples samples; buffers only ever grow
static void batch_size(pk_batch *b, int max_clips, int64_t n_samples) {
    Model &m = *b->m;
    for (auto &p : b->pcm2) p.reserve((size_t)max_clips * n_samples * 4);
    for (auto &w : b->ws) w.size_for(m.cfg, max_clips, -n_samples, pk_mel_num_frames(n_samples));
    for (auto &h : b->held) { h.ragged = false; h.n_samples = n_samples; }
    b->rag_capacity = false;
}
// capacity for ragged AND uniform batches of <= max_clips clips, <= max_total samples in all, <= max_clip per clip (buffers only ever grow)
static void batch_size_ragged(pk_batch *b, int max_clips, int64_t max_total, int64_t max_clip) {
    Model &m = *b->m;
    for (auto &p : b->pcm2) p.reserve((size_t)max_total * 4);
    for (auto &w : b->ws) w.size_ragged(m.cfg, max_clips, max_total, max_clip, /*own_pcm=*/false);
    for (auto &h : b->held) { h.ragged = false; h.n_samples = 0; }
    b->rag_capacity = true;
}

static std::unique_ptr<pk_batch> batch_new(Model &m, int max_clips, int64_t n_samples, int64_t rag_total = 0) {
    m.require_gpu();
    auto b = std::make_unique<pk_batch>();
    b->m = &m;
    if (rag_total > 0) batch_size_ragged(b.get(), max_clips, rag_total, n_samples);
    else batch_size(b.get(), max_clips, n_samples);
    for (auto &e : b->ev) PK_HIP(hipEventCreate(&e));
    PK_HIP(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        PK_HIP(hipEventCreateWithFlags(&b->enc_done[i], hipEventDisableTiming));
        PK_HIP(hipEventCreateWithFlags(&b->dec_done[i], hipEventDisableTiming));
        PK_HIP(hipEventCreateWithFlags(&b->copy_done[i], hipEventDisableTiming));
        PK_HIP(hipEventCreateWithFlags(&b->mel_done[i], hipEventDisableTiming));
    }
    b->ev_ok = true;
    return b;
}

pk_status pk_batch_create(pk_model *h, int max_clips, int64_t n_samples, pk_batch **out) {
    return guard([&] {
        need(h && out && max_clips > 0 && n_samples > 256, "model/out/max_clips/n_samples");
        *out = batch_new(*h->m, max_clips, n_samples).release();
    });
}

pk_status pk_batch_create_ragged(pk_model *h, int max_clips, int64_t max_total_samples, int64_t max_clip_samples, pk_batch **out) {
    return guard([&] {
        need(h && out && max_clips > 0 && max_clip_samples > 256 && max_total_samples >= max_clip_samples, "model/out/max_clips/max_total_samples/max_clip_samples");
        *out = batch_new(*h->m, max_clips, max_clip_samples, max_total_samples).release();
    });
}

void pk_batch_free(pk_batch *b) {
    if (!b) return;
    if (b->ev_ok) {
        (void)hipStreamSynchronize(b->m->stream_dec);
        (void)hipStreamSynchronize(b->m->stream);
        (void)hipStreamSynchronize(b->copy_stream);
        for (auto &e : b->ev) (void)hipEventDestroy(e);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy(b->enc_done[i]); (void)hipEventDestroy(b->dec_done[i]);
            (void)hipEventDestroy(b->copy_done[i]); (void)hipEventDestroy(b->mel_done[i]);
        }
        (void)hipStreamDestroy(b->copy_stream);
        for (auto &G : b->grp) {
            if (G.ep_done) (void)hipEventDestroy(G.ep_done);
            if (G.dec_done) (void)hipEventDestroy(G.dec_done);
        }
    }
    delete b;
}

pk_status pk_batch_upload(pk_batch *b, const float *pcm, int n_clips) {
    return guard([&] {
        need(b && pcm && n_clips > 0 && n_clips <= b->ws[0].B, "batch/pcm/n_clips");
        need(!b->rag_capacity, "a pipeline created with pk_batch_create_ragged takes pk_batch_upload_ragged (equal lengths are a special case of it)");
        b->m->require_gpu();
        batch_flush(b);
        PK_HIP(hipStreamSynchronize(b->copy_stream));
        b->staged = -1;
        PK_HIP(hipMemcpyAsync(b->pcm2[b->cur].p, pcm, (size_t)n_clips * b->ws[0].n_samples * 4, hipMemcpyHostToDevice, b->m->stream));
        PK_HIP(hipStreamSynchronize(b->m->stream));
        b->n_clips = n_clips;
    });
}

// what a PCM buffer holds after staging clips of the given lengths: a uniform batch when all lengths agree (the plain kernels: no tables),
// otherwise a ragged one
static void batch_hold(pk_batch *b, int buf, const int64_t *lens, int n_clips) {
    Model &m = *b->m;
    pk_batch::Held &H = b->held[buf];
    bool same = true;
    int64_t total = 0, longest = 0;
    for (int i = 0; i < n_clips; ++i) { same = same && lens[i] == lens[0]; total += lens[i]; longest = std::max(longest, lens[i]); }
    const Workspace &w = b->ws[0];
    if (n_clips > w.rag_cap_clips || total > w.rag_cap_samples || longest > w.rag_cap_clip)
        fail(PK_ERR_INVALID, "batch of %d clips / %lld samples (longest %lld) exceeds the pipeline's capacity (%d clips, %lld samples, %lld per clip)", n_clips,
             (long long)total, (long long)longest, w.rag_cap_clips, (long long)w.rag_cap_samples, (long long)w.rag_cap_clip);
    for (int i = 0; i < n_clips; ++i) need(lens[i] > 256, "every clip needs more than 256 samples");
    H.ragged = !same;
    H.n_samples = same ? lens[0] : 0;
    if (!same) {
        const int t_max = pk_encoder_num_frames(pk_mel_num_frames(longest));
        H.rag.build_from_samples(lens, n_clips, m.attn_bf16(t_max) ? relpos_attention_bf16_block_rows(m.cfg.hidden_size / m.cfg.num_heads) : 32);
    }
}

pk_status pk_batch_upload_ragged(pk_batch *b, const float *pcm, const int64_t *offsets, int n_clips) {
    return guard([&] {
        need(b && pcm && offsets && n_clips > 0, "batch/pcm/offsets/n_clips");
        need(b->rag_capacity, "pk_batch_upload_ragged needs a pipeline created with pk_batch_create_ragged");
        b->m->require_gpu();
        batch_flush(b);
        PK_HIP(hipStreamSynchronize(b->copy_stream));
        b->staged = -1;
        std::vector<int64_t> lens(n_clips);
        for (int i = 0; i < n_clips; ++i) lens[i] = offsets[i + 1] - offsets[i];
        batch_hold(b, b->cur, lens.data(), n_clips);
        int64_t o = 0;
        for (int i = 0; i < n_clips; ++i) {        // packed back to back in the device buffer, whatever the gaps in the caller's
            PK_HIP(hipMemcpyAsync(b->pcm2[b->cur].as<float>() + o, pcm + offsets[i], (size_t)lens[i] * 4, hipMemcpyHostToDevice, b->m->stream));
            o += lens[i];
        }
        PK_HIP(hipStreamSynchronize(b->m->stream));
        b->n_clips = n_clips;
    });
}

// stages the NEXT batch into the PCM buffer the running encoder does not read, on the copy stream.  clip(i) = host pointer of clip i;
// clips that follow each other in host memory go as one copy.
static void batch_stage(pk_batch *b, int n_clips, const std::function<const float *(int)> &clip, const int64_t *lens = nullptr) {
    b->m->require_gpu();
    const int nb = b->staged >= 0 ? b->staged : (b->cur ^ 1);      // re-staging before a run overwrites the staged batch
    std::vector<int64_t> uni;
    if (!lens) { uni.assign(n_clips, b->held[nb].n_samples > 0 ? b->held[nb].n_samples : b->ws[0].n_samples); lens = uni.data(); }
    if (b->rag_capacity) batch_hold(b, nb, lens, n_clips);         // (validates the batch against the capacity before anything is copied)
    PK_HIP(hipStreamSynchronize(b->copy_stream));                  // at most one copy in flight; the previous host buffer is released here
    if (b->mel_used[nb]) PK_HIP(hipStreamWaitEvent(b->copy_stream, b->mel_done[nb], 0));   // the last mel that read this buffer
    int64_t o = 0;
    for (int i = 0; i < n_clips;) {
        int j = i + 1;
        int64_t run = lens[i];
        while (j < n_clips && clip(j) == clip(j - 1) + lens[j - 1]) { run += lens[j]; ++j; }
        PK_HIP(hipMemcpyAsync(b->pcm2[nb].as<float>() + o, clip(i), (size_t)run * 4, hipMemcpyHostToDevice, b->copy_stream));
        o += run;
        i = j;
    }
    PK_HIP(hipEventRecord(b->copy_done[nb], b->copy_stream));
    b->staged = nb;
    b->staged_clips = n_clips;
}

pk_status pk_batch_upload_async(pk_batch *b, const float *pcm, int n_clips) {
    return guard([&] {
        need(b && pcm && n_clips > 0 && n_clips <= b->ws[0].B, "batch/pcm/n_clips");
        need(!b->rag_capacity, "a pipeline created with pk_batch_create_ragged takes pk_batch_upload_ragged_async");
        const int64_t n = b->ws[0].n_samples;
        batch_stage(b, n_clips, [&](int i) { return pcm + (size_t)i * n; });
    });
}

pk_status pk_batch_upload_ragged_async(pk_batch *b, const float *pcm, const int64_t *offsets, int n_clips) {
    return guard([&] {
        need(b && pcm && offsets && n_clips > 0, "batch/pcm/offsets/n_clips");
        need(b->rag_capacity, "pk_batch_upload_ragged_async needs a pipeline created with pk_batch_create_ragged");
        std::vector<int64_t> lens(n_clips);
        for (int i = 0; i < n_clips; ++i) lens[i] = offsets[i + 1] - offsets[i];
        batch_stage(b, n_clips, [&](int i) { return pcm + offsets[i]; }, lens.data());
    });
}

pk_status pk_batch_run(pk_batch *b, int decoder) {
    return guard([&] { need(b, "batch"); batch_run(b, decoder); });
}

pk_status pk_batch_sync(pk_batch *b) {
    return guard([&] { need(b, "batch"); b->m->require_gpu(); batch_flush(b); });
}

int pk_batch_max_tokens(const pk_batch *b) { return b ? b->ws[0].max_tokens : 0; }

// copies the results of one finished run: rows [row0, row0 + clips) of its workspace, presented as [clips][max_tokens]
static void copy_results(const pk_batch::Loc &L, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    Workspace &w = *L.w;
    const int B = L.clips, mt = w.max_tokens;
    const size_t src_w = (size_t)(L.decoder == PK_DECODER_TDT ? mt : w.T);        // CTC arrays are [B][T] on the device
    auto pitch = [&](void *dst, const void *src) {
        if (dst) PK_HIP(hipMemcpy2D(dst, (size_t)mt * 4, static_cast<const char *>(src) + (size_t)L.row0 * src_w * 4, src_w * 4, src_w * 4, B, hipMemcpyDeviceToHost));
    };
    PK_HIP(hipMemcpy(lens, w.lens.as<int>() + L.row0, (size_t)B * 4, hipMemcpyDeviceToHost));
    pitch(ids, w.ids.p); pitch(start, w.start.p); pitch(end, w.end.p); pitch(conf, w.conf.p);
    zero_tail(ids, lens, B, mt); zero_tail(start, lens, B, mt); zero_tail(end, lens, B, mt); zero_tail(conf, lens, B, mt);
}

pk_status pk_batch_results(pk_batch *b, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    return guard([&] {
        need(b && ids && lens, "batch/ids/lens");
        b->m->require_gpu();
        batch_flush(b);
        need(!b->done.empty(), "pk_batch_run() first");
        copy_results(b->done.back(), ids, lens, start, end, conf);
    });
}

pk_status pk_batch_results_back(pk_batch *b, int back, int *n_clips, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    return guard([&] {
        need(b && ids && lens, "batch/ids/lens");
        b->m->require_gpu();
        need(!b->done.empty(), "no decoded batch yet: the decode of run k finishes inside a later pk_batch_run (or pk_batch_sync)");
        need(back >= 0 && back < (int)b->done.size(), "back: only the runs of the newest decoded group are kept");
        const pk_batch::Loc &L = b->done[b->done.size() - 1 - (size_t)back];
        PK_HIP(hipEventSynchronize(L.ev));
        if (n_clips) *n_clips = L.clips;
        copy_results(L, ids, lens, start, end, conf);
    });
}

pk_status pk_batch_margins(pk_batch *b, int back, float *min_margin) {
    return guard([&] {
        need(b && min_margin, "batch/min_margin");
        b->m->require_gpu();
        need(back >= 0 && back < (int)b->done.size(), "back: 0 <= back < pk_batch_results_available()");
        const pk_batch::Loc &L = b->done[b->done.size() - 1 - (size_t)back];
        need(L.decoder == PK_DECODER_TDT && !b->m->boost_on, "margins are reported for unboosted TDT / RNNT decodes");
        PK_HIP(hipEventSynchronize(L.ev));
        PK_HIP(hipMemcpy(min_margin, L.w->margin.as<float>() + L.row0, (size_t)L.clips * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_batch_results_done(pk_batch *b, int *n_clips, int32_t *ids, int32_t *lens, int32_t *start, int32_t *end, float *conf) {
    return pk_batch_results_back(b, 0, n_clips, ids, lens, start, end, conf);
}

int pk_batch_results_available(const pk_batch *b) { return b ? (int)b->done.size() : 0; }

static void batch_set_group(pk_batch *b, int group) {
    need(group >= 1 && group <= 16, "decode group: 1 .. 16 runs");
    Model &m = *b->m;
    m.require_gpu();
    batch_flush(b);
    for (auto &G : b->grp) b->forget(&G.w);              // their buffers may be reallocated below: read results BEFORE changing the group size
    if (group > 1) {
        need(m.cfg.vocab_size > 0, "decode groups apply to the TDT / RNNT decoder; this model has none");
        for (auto &G : b->grp) {
            G.w.size_decode(m.cfg, group * b->ws[0].B, b->ws[0].T, (size_t)group * b->ws[0].rag_cap_rows);
            if (b->rag_capacity) G.tabs.reserve((size_t)2 * group * b->ws[0].B * sizeof(int));
            if (!G.ep_done) PK_HIP(hipEventCreateWithFlags(&G.ep_done, hipEventDisableTiming));
            if (!G.dec_done) PK_HIP(hipEventCreateWithFlags(&G.dec_done, hipEventDisableTiming));
            G.mem.clear();
            G.rows = 0;
            G.ep_rows = 0;
            G.T_max = 0;
            G.used = G.decoded = false;
        }
    }
    b->fill = 0;
    b->ready = -1;
    b->group = group;
}

pk_status pk_batch_set_decode_overlap(pk_batch *b, int on) {
    return guard([&] {
        need(b, "batch");
        b->m->require_gpu();
        batch_flush(b);
        b->overlap = on != 0;
    });
}

pk_status pk_batch_set_decode_group(pk_batch *b, int group) {
    return guard([&] { need(b, "batch"); batch_set_group(b, group); });
}

// One un-pipelined run on the main stream with hipEvents between the stages (mel / encoder / decode / total, ms).
pk_status pk_batch_run_timed(pk_batch *b, int decoder, float ms[4]) {
    return guard([&] {
        need(b && ms, "batch/ms");
        need(decoder == PK_DECODER_CTC || decoder == PK_DECODER_TDT, "decoder");
        Model &m = *b->m;
        m.require_gpu();
        need(b->n_clips > 0, "pk_batch_upload() first");
        batch_flush(b);
        Workspace &w = b->ws[0];
        hipStream_t s = m.stream;
        batch_set_run(b, w, s);
        PK_HIP(hipEventRecord(b->ev[0], s));
        m.run_mel_ws(w, b->pcm2[b->cur].as<float>(), b->n_clips, s);
        PK_HIP(hipEventRecord(b->ev[1], s));
        m.run_encoder(w, w.feats.as<float>(), b->n_clips, w.Tm, -1, 0, s);
        PK_HIP(hipEventRecord(b->ev[2], s));
        b->slot_clips[0] = b->n_clips;
        batch_decode(b, 0, decoder, s);
        PK_HIP(hipEventRecord(b->ev[3], s));
        PK_CHECK_LAUNCH();
        PK_HIP(hipStreamSynchronize(s));
        b->used[0] = true;
        PK_HIP(hipEventElapsedTime(&ms[0], b->ev[0], b->ev[1]));
        PK_HIP(hipEventElapsedTime(&ms[1], b->ev[1], b->ev[2]));
        PK_HIP(hipEventElapsedTime(&ms[2], b->ev[2], b->ev[3]));
        PK_HIP(hipEventElapsedTime(&ms[3], b->ev[0], b->ev[3]));
    });
}

void *pk_batch_dev_pcm(pk_batch *b) { return b ? b->pcm2[b->cur].p : nullptr; }
void *pk_batch_stream(pk_batch *b) { return b ? (void *)b->m->stream : nullptr; }

int pk_batch_profile(pk_batch *b, int decoder, pk_kernel_stat *out, int cap) {
    int n_out = -1;
    pk_status st = guard([&] {
        need(b && out && cap > 0, "batch/out/cap");
        need(decoder == 0 || decoder == 1, "decoder must be 0 (CTC) or 1 (TDT)");
        need(b->n_clips > 0, "no clips uploaded");
        Model &m = *b->m;
        m.require_gpu();
        ProfileSink sink;
        m.prof = &sink;
        batch_flush(b);
        try {
            Workspace &w = b->ws[0];
            batch_set_run(b, w, m.stream);
            m.run_mel_ws(w, b->pcm2[b->cur].as<float>(), b->n_clips, m.stream);
            m.run_encoder(w, w.feats.as<float>(), b->n_clips, w.Tm, -1, 0, m.stream);
            b->slot_clips[0] = b->n_clips;
            batch_decode(b, 0, decoder, m.stream);
            PK_HIP(hipStreamSynchronize(m.stream));
            b->used[0] = true;
        } catch (...) {
            m.prof = nullptr;
            throw;
        }
        m.prof = nullptr;
        std::vector<pk_kernel_stat> agg;
        for (auto &r : sink.recs) {
            float ms = 0.0f;
            PK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
            size_t i = 0;
            for (; i < agg.size(); ++i)
                if (r.name == agg[i].name) break;
            if (i == agg.size()) {
                pk_kernel_stat k;
                memset(&k, 0, sizeof k);
                snprintf(k.name, sizeof k.name, "%s", r.name.c_str());
                agg.push_back(k);
            }
            agg[i].launches += 1;
            agg[i].total_ms += ms;
            agg[i].flops += r.flops;
            agg[i].bytes += r.bytes;
        }
        n_out = (int)agg.size();
        for (int i = 0; i < n_out && i < cap; ++i) out[i] = agg[i];
    });
    return st == PK_OK ? n_out : (int)st;
}

// Brings a pipeline back to a defined idle state after an error inside a run (nothing pending, nothing readable).
static void batch_reset(pk_batch *b) {
    (void)hipStreamSynchronize(b->m->stream_dec);
    (void)hipStreamSynchronize(b->m->stream);
    (void)hipStreamSynchronize(b->copy_stream);
    (void)hipGetLastError();
    b->pending_slot = b->pending_decoder = -1;
    b->ready = -1;
    b->staged = -1;
    b->n_clips = 0;
    for (auto &G : b->grp) { G.mem.clear(); G.rows = 0; G.ep_rows = 0; G.T_max = 0; }
    b->done.clear();
}

static const size_t kTokenShrinkBytes = (size_t)512 << 20;   // token arrays of a pipeline (2 slots + 2 decode groups) above which a shorter call re-sizes them
// The pipeline a Model keeps for the one-call API (pk_transcribe_pcm, every rank of a pk_group): created on first use with ragged capacity
// (batches of mixed lengths AND uniform ones), re-sized when a call needs more (buffers only grow; the token arrays may shrink, see below), freed with the model.
static pk_batch *model_pipeline(Model &m, int max_clips, int64_t max_total, int64_t max_clip) {
    m.require_gpu();
    if (!m.pipe) {
        m.pipe = batch_new(m, max_clips, max_clip, max_total).release();
        m.pipe_free = [](void *p) { pk_batch_free(static_cast<pk_batch *>(p)); };
        return static_cast<pk_batch *>(m.pipe);
    }
    pk_batch *b = static_cast<pk_batch *>(m.pipe);
    batch_flush(b);
    PK_HIP(hipStreamSynchronize(b->copy_stream));
    b->done.clear();
    b->staged = -1;
    b->n_clips = 0;
    const Workspace &w = b->ws[0];
    // Buffers only grow -- except the token arrays, the one allocation pitched (clips x longest clip x max_symbols): when the pipeline was
    // sized for a clip at least twice as long as anything in this call and those arrays are large, they are released and re-reserved for this
    // call's longest clip, so that one long file does not make every later batch carry (and copy, and zero) its pitch (round-4 advisor finding).
    int64_t clip_cap = std::max(max_clip, w.rag_cap_clip);
    size_t tok = 0;
    for (auto &x : b->ws) tok += x.token_bytes();
    for (auto &G : b->grp) tok += G.w.token_bytes();
    // ... and whenever transcribe_clips' segment rule would have cut here (clips four times shorter than what the arrays are pitched for AND
    // one slot's arrays above kTokenShrinkBytes / 8): a segment cut is always followed by a re-pitch (round-5 advisor finding: between the two
    // thresholds a cut used to happen with no shrink behind it, and every batch of the new segment still carried the long file's pitch).
    const bool seg_rule = max_clip * 4 <= w.rag_cap_clip && w.token_bytes() > kTokenShrinkBytes / 8;
    if ((tok > kTokenShrinkBytes && max_clip * 2 <= w.rag_cap_clip) || seg_rule) {
        for (auto &x : b->ws) x.release_tokens();
        for (auto &G : b->grp) { b->forget(&G.w); G.w.release_tokens(); }
        clip_cap = max_clip;
    }
    batch_size_ragged(b, std::max(max_clips, w.rag_cap_clips), std::max(max_total, w.rag_cap_samples), clip_cap);
    return b;
}

/* ---- one-call API ------------------------------------------------------------------------------------------ */

namespace {
struct ResultStore {          // owns everything a pk_result array points into
    std::vector<pk_result> res;
    std::vector<std::string> text;
    std::vector<std::vector<int32_t>> ids, start, end;
    std::vector<std::vector<float>> conf;
    std::vector<std::vector<std::string>> word_text;
    std::vector<std::vector<pk_word>> words;
};
}  // namespace

// The packing policy of the one-call API (pure host logic; pk_plan_batches exposes it): clips sorted by length, longest first (stable), then
// cut greedily into batches of at most kMaxBatchClips clips and kBatchRows ENCODER ROWS (a single longer clip gets a batch of its own).
// Rows, not seconds, are what the batch costs: every product of the encoder is an M x N x K GEMM with M = the batch's packed rows, tiled 128
// (64) rows high, and 8192 rows are exactly the tile grids the kernels were tuned on -- fc2 / out_proj / pw2: 256 (512) tiles = ONE round of the
// 256 CUs, fc1: 1024 tiles = two rounds.  One tile row more starts another round of workgroups on every product: measured round 4
// (profiles/r04_mixed_bench_distributions.txt) 8272 rows cost fc2 +52 %, out_proj / pw2 +59 %, the encoder 27.3 instead of ~21 ms.
static const int kMaxBatchClips = 256;
static const int64_t kBatchRows = 8192;                          // 64 tile rows of 128; 64 x 10 s = 8064 rows, 65 x 10 s = 8190
static void plan_batches(const int64_t *len, int n, std::vector<int> &order, std::vector<int> &bstart) {
    order.resize(n);
    std::vector<int64_t> rows(n);
    for (int i = 0; i < n; ++i) {
        need(len[i] > 256, "every clip needs more than 256 samples");
        need(len[i] <= ((int64_t)1 << 30), "clip too long");
        order[i] = i;
        rows[i] = pk_encoder_num_frames(pk_mel_num_frames(len[i]));
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len[a] > len[b]; });
    bstart.clear();
    for (int i = 0; i < n;) {
        bstart.push_back(i);
        int64_t tot = 0;
        int j = i;
        while (j < n && j - i < kMaxBatchClips && (j == i || tot + rows[order[j]] <= kBatchRows)) tot += rows[order[j++]];
        i = j;
    }
    bstart.push_back(n);
}

pk_status pk_plan_batches(const int64_t *n_samples, int n_clips, int32_t *batch_of_clip, int32_t *pos_in_batch, int *n_batches) {
    return guard([&] {
        need(n_samples && n_clips > 0 && batch_of_clip, "n_samples/n_clips/batch_of_clip");
        std::vector<int> order, bstart;
        plan_batches(n_samples, n_clips, order, bstart);
        for (size_t k = 0; k + 1 < bstart.size(); ++k)
            for (int i = bstart[k]; i < bstart[k + 1]; ++i) {
                batch_of_clip[order[i]] = (int32_t)k;
                if (pos_in_batch) pos_in_batch[order[i]] = i - bstart[k];
            }
        if (n_batches) *n_batches = (int)bstart.size() - 1;
    });
}

pk_status pk_ragged_extents(const int64_t *n_samples, int n_clips, int32_t *n_mel_frames, int32_t *n_enc_frames, int64_t *totals) {
    return guard([&] {
        need(n_samples && n_clips > 0, "n_samples/n_clips");
        RagBatch r;
        r.build_from_samples(n_samples, n_clips, 32);
        for (int i = 0; i < n_clips; ++i) {
            if (n_mel_frames) n_mel_frames[i] = r.Tm[i];
            if (n_enc_frames) n_enc_frames[i] = r.T[i];
        }
        if (totals) { totals[0] = r.n_samples; totals[1] = r.sum_Tm; totals[2] = r.sum_H2; totals[3] = r.sum_T; totals[4] = r.n_u_att; totals[5] = r.n_u_dw; totals[6] = r.n_u_c1; }
    });
}

// Transcriber::transcribe (transcribe.hpp:99-179) of the clips listed in `clips` (global indices into offsets), results into the slots
// of the same indices of R.  One model, one device; called by pk_transcribe_pcm (all clips) and by every rank of a pk_group.
static void transcribe_clips(Model &m, const float *pcm, const int64_t *offsets, const std::vector<int> &clips, const pk_options *opt,
                             ResultStore &R) {
    m.require_gpu();
    const int decoder = opt ? opt->decoder : PK_DECODER_TDT;
    const bool ts = opt && opt->timestamps;
    need(decoder == PK_DECODER_CTC || decoder == PK_DECODER_TDT, "options.decoder");
    // per-call boost phrases (transcribe.hpp:110-115): the model-level setting comes back when the call ends
    struct BoostScope {
        Model &m; bool active = false; std::vector<std::vector<int>> saved; float saved_score = 0.0f;
        ~BoostScope() { if (active) { try { m.set_boost(saved, saved_score); } catch (...) {} } }
    } scope{m};
    if (opt && opt->n_boost_phrases > 0) {
        need(opt->boost_phrases != nullptr, "options.boost_phrases");
        auto ph = encode_phrases(m, opt->boost_phrases, opt->n_boost_phrases);
        scope.saved = m.boost_phrases; scope.saved_score = m.boost_score; scope.active = true;
        m.set_boost(ph, opt->boost_score);
    }
    // Mixed-length batching (the reference's roadmap item "batch inference: pad + length-mask", README.md:513 -- done by PACKING, no padding
    // and no masks: every clip keeps its own extents in every kernel and comes out bit-identical to a single-clip call).  The clips are
    // sorted by length, longest first (the position tables and the workspace are then sized once, by the first batch), and packed greedily
    // into batches of at most kMaxBatchClips clips and kBatchRows encoder rows -- neighbours in length share a batch, so the lock-step decode
    // loop of a batch ends for all of them at about the same step.  The batches go through the model's two-stream pipeline (struct
    // pk_batch): PCM of batch k+1 is staged on the copy stream and decode(k) -- or, from four batches on, the decode loops of four batches as
    // one lock-step group -- runs under encoder(k+1).  A batch whose clips all have the same length runs the plain uniform kernels.
    const int n_clips = (int)clips.size();
    if (n_clips == 0) return;
    std::vector<int64_t> clip_len(n_clips);
    for (int i = 0; i < n_clips; ++i) clip_len[i] = offsets[clips[i] + 1] - offsets[clips[i]];
    std::vector<int> order, bstart;                                // order: positions in `clips`, longest first; bstart: first position of every batch, + the end
    plan_batches(clip_len.data(), n_clips, order, bstart);
    for (auto &o : order) o = clips[o];                            // ... as global clip indices from here on
    auto len_of = [&](int i) { return offsets[order[i] + 1] - offsets[order[i]]; };
    const int nb_all = (int)bstart.size() - 1;
    // The output arrays of a pipeline are pitched (clips x longest clip x max_symbols).  With the clips sorted longest first, one very long
    // file in front of many short ones would make every batch of the call carry its pitch (a 1 h file and 256 short clips: ~0.5 GB per array,
    // per slot and decode group, copied and zeroed per batch -- round-4 advisor finding).  So the batch list is cut into SEGMENTS, each run
    // through the pipeline sized for its own longest clip: a new segment starts where the clips have become four times shorter than the
    // segment's first AND the segment's token arrays would be large.  Ordinary calls (10 s .. a few minutes per clip) are one segment.
    const int sym = m.cfg.max_symbols_per_step > 0 ? m.cfg.max_symbols_per_step : 10;
    auto frames_of = [&](int64_t n) { return (int64_t)pk_encoder_num_frames(pk_mel_num_frames(n)); };
    std::vector<int32_t> ids, st, en, lens;
    std::vector<float> cf;
    for (int kseg = 0; kseg < nb_all;) {
    const int seg0 = kseg;
    int64_t cap_total = 0;
    int cap_clips = 0;
    const int64_t seg_T = frames_of(len_of(bstart[seg0]));
    for (; kseg < nb_all; ++kseg) {
        const int nc = bstart[kseg + 1] - bstart[kseg];
        if (kseg > seg0 && len_of(bstart[kseg]) * 4 <= len_of(bstart[seg0]) &&
            (size_t)std::max(cap_clips, nc) * seg_T * sym * 4 * 4 > kTokenShrinkBytes / 8) break;
        int64_t tot = 0;
        for (int i = bstart[kseg]; i < bstart[kseg + 1]; ++i) tot += len_of(i);
        cap_total = std::max(cap_total, tot);
        cap_clips = std::max(cap_clips, nc);
    }
    const int nb = kseg - seg0;                                    // batches seg0 .. kseg-1 run as one pipeline pass
    pk_batch *b = model_pipeline(m, cap_clips, cap_total, len_of(bstart[seg0]));
    try {
        batch_set_group(b, (decoder == PK_DECODER_TDT && nb >= 4) ? 4 : 1);
        const int64_t first_seq = b->runs;
        const int mt = b->ws[0].max_tokens;
        std::vector<char> taken(nb, 0);
        std::vector<int64_t> blens;
        const int *bs = bstart.data() + seg0;                      // (the lambdas below index the segment's batches 0 .. nb-1)
        auto stage = [&](int k) {
            const int c0 = bs[k], nc = bs[k + 1] - c0;
            blens.resize(nc);
            for (int i = 0; i < nc; ++i) blens[i] = len_of(c0 + i);
            batch_stage(b, nc, [&](int i) { return pcm + offsets[order[c0 + i]]; }, blens.data());
        };
        auto drain = [&]() {                                  // every finished run of this call that has not been handed out yet
            for (const auto &L : b->done) {
                const int64_t k = L.seq - first_seq;
                if (k < 0 || k >= nb || taken[k]) continue;
                taken[k] = 1;
                const int B = L.clips;
                const size_t tok = (size_t)B * mt;
                ids.resize(tok); lens.resize(B);
                if (ts) { st.resize(tok); en.resize(tok); cf.resize(tok); }
                PK_HIP(hipEventSynchronize(L.ev));
                copy_results(L, ids.data(), lens.data(), ts ? st.data() : nullptr, ts ? en.data() : nullptr, ts ? cf.data() : nullptr);
                for (int i = 0; i < B; ++i) {
                    const int c = order[bs[k] + i];
                    if (lens[i] < 0) fail(PK_ERR_DECODE_CAP, "TDT decode hit the safety cap on clip %d", c);
                    const int n = lens[i];
                    R.ids[c].assign(ids.begin() + (size_t)i * mt, ids.begin() + (size_t)i * mt + n);
                    std::vector<int> iv(R.ids[c].begin(), R.ids[c].end());
                    if (m.tok.loaded()) R.text[c] = m.tok.decode(iv);                    // transcribe.hpp:149,172
                    if (ts) {
                        R.start[c].assign(st.begin() + (size_t)i * mt, st.begin() + (size_t)i * mt + n);
                        R.end[c].assign(en.begin() + (size_t)i * mt, en.begin() + (size_t)i * mt + n);
                        R.conf[c].assign(cf.begin() + (size_t)i * mt, cf.begin() + (size_t)i * mt + n);
                        if (m.tok.loaded()) {                                            // group_timestamps, transcribe.hpp:150-152
                            std::vector<TimestampedToken> tt(n);
                            for (int q = 0; q < n; ++q) tt[q] = {R.ids[c][q], R.start[c][q], R.end[c][q], R.conf[c][q]};
                            auto words = group_timestamps(tt, m.tok.pieces(), false);
                            for (auto &wd : words) R.word_text[c].push_back(wd.word);
                            for (size_t q = 0; q < words.size(); ++q)
                                R.words[c].push_back({R.word_text[c][q].c_str(), words[q].start, words[q].end, words[q].confidence});
                        }
                    }
                }
            }
        };
        stage(0);
        for (int k = 0; k < nb; ++k) {
            batch_run(b, decoder);                               // encoder(k) queued, then decode(k-1) / the finished group driven under it
            if (k + 1 < nb) stage(k + 1);
            drain();
        }
        batch_flush(b);
        drain();
        for (int k = 0; k < nb; ++k)
            if (!taken[k]) fail(PK_ERR_HIP, "internal: batch %d of the pipeline produced no result", seg0 + k);
    } catch (...) {
        batch_reset(b);
        throw;
    }
    }   // segments
}

static std::unique_ptr<ResultStore> new_store(int n_clips) {
    auto store = std::make_unique<ResultStore>();
    ResultStore &R = *store;
    R.res.resize(n_clips + 1);            // one hidden trailing slot keeps the store pointer
    R.text.resize(n_clips); R.ids.resize(n_clips); R.start.resize(n_clips); R.end.resize(n_clips); R.conf.resize(n_clips);
    R.word_text.resize(n_clips); R.words.resize(n_clips);
    return store;
}
// hands the store over to the caller as a pk_result array (freed by pk_results_free)
static pk_result *publish_store(std::unique_ptr<ResultStore> store, int n_clips, bool ts) {
    ResultStore &R = *store;
    for (int c = 0; c < n_clips; ++c) {
        pk_result &r = R.res[c];
        r.text = R.text[c].c_str();
        r.n_tokens = (int32_t)R.ids[c].size();
        r.token_ids = R.ids[c].data();
        r.start_frame = ts ? R.start[c].data() : nullptr;
        r.end_frame = ts ? R.end[c].data() : nullptr;
        r.confidence = ts ? R.conf[c].data() : nullptr;
        r.n_words = (int32_t)R.words[c].size();
        r.words = R.words[c].data();
    }
    memset(&R.res[n_clips], 0, sizeof(pk_result));
    R.res[n_clips].text = reinterpret_cast<const char *>(store.get());   // back-pointer for pk_results_free
    pk_result *out = R.res.data();
    store.release();
    return out;
}

pk_status pk_transcribe_pcm(pk_model *h, const float *pcm, const int64_t *offsets, int n_clips, const pk_options *opt,
                            pk_result **results) {
    return guard([&] {
        need(h && pcm && offsets && results && n_clips > 0, "model/pcm/offsets/results/n_clips");
        auto store = new_store(n_clips);
        std::vector<int> all(n_clips);
        for (int i = 0; i < n_clips; ++i) all[i] = i;
        transcribe_clips(*h->m, pcm, offsets, all, opt, *store);
        *results = publish_store(std::move(store), n_clips, opt && opt->timestamps);
    });
}

/* ---- one node, several GPUs ------------------------------------------------------------------------------------------------ */
struct pk_group {
    std::vector<int> devices;
    std::vector<std::unique_ptr<Model>> models;
    double wall_ms_max = 0.0, audio_s = 0.0;
    std::vector<int32_t> clips_per_rank;
    std::vector<double> wall_ms;
    std::vector<std::vector<int>> last_shard;  // clip indices each rank handled in the last call (pk_group_verify_exchange)
    // RCCL is only touched by pk_group_verify_exchange: communicators and streams are created on its first call
    const RcclApi *rccl = nullptr;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    ~pk_group() {
        for (size_t r = 0; r < streams.size(); ++r) {
            (void)hipSetDevice(devices[r]);
            if (streams[r]) (void)hipStreamDestroy(streams[r]);
        }
        models.clear();
        if (rccl) for (auto c : comms) if (c) (void)rccl->CommDestroy(c);
    }
};
#define PK_NCCL(api, call) do { ncclResult_t r_ = (api)->call; if (r_ != ncclSuccess) fail(PK_ERR_HIP, "RCCL: %s (%s)", (api)->GetErrorString(r_), #call); } while (0)

// runs fn(rank) on one host thread per device; the first exception of any rank is rethrown on the calling thread
static void for_each_rank(int n, const std::function<void(int)> &fn) {
    if (n == 1) { fn(0); return; }
    std::vector<std::exception_ptr> err(n);
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            try { fn(r); } catch (...) { err[r] = std::current_exception(); }
        });
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

pk_status pk_group_create(const char *weights, const char *vocab, const pk_config *cfg, const int *devices, int n_devices, pk_group **out) {
    return guard([&] {
        need(weights && cfg && out, "weights/cfg/out");
        int visible = 0;
        if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) fail(PK_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU path)");
        auto g = std::make_unique<pk_group>();
        if (!devices || n_devices <= 0) {
            for (int d = 0; d < visible; ++d) g->devices.push_back(d);
        } else {
            for (int i = 0; i < n_devices; ++i) {
                need(devices[i] >= 0 && devices[i] < visible, "devices[i] out of range");
                g->devices.push_back(devices[i]);
            }
        }
        const int G = (int)g->devices.size();
        // ONE disk read (mmap) shared by every rank: each replica is built from the same host image by its own host thread -- the per-tensor
        // layout transforms and uploads of the G devices run concurrently, each device over its own PCIe link; no copy of the image is made.
        SafeTensors image(weights);
        const void *base = image.image_base();
        const size_t len = image.image_bytes();
        g->models.resize(G);
        const std::string vp = vocab ? vocab : "";
        for_each_rank(G, [&](int r) {
            g->models[r] = std::make_unique<Model>(base, len, vp, *cfg, /*borrow=*/true);
            g->models[r]->to_gpu(g->devices[r]);
        });
        g->clips_per_rank.assign(G, 0);
        g->wall_ms.assign(G, 0.0);
        g->last_shard.assign(G, {});
        *out = g.release();
    });
}

void pk_group_free(pk_group *g) { delete g; }
int pk_group_size(const pk_group *g) { return g ? (int)g->devices.size() : 0; }

pk_status pk_group_transcribe_pcm(pk_group *g, const float *pcm, const int64_t *offsets, int n_clips, const pk_options *opt, pk_result **results) {
    return guard([&] {
        need(g && pcm && offsets && results && n_clips > 0, "group/pcm/offsets/results/n_clips");
        const int G = (int)g->devices.size();
        // partition by AUDIO: clips sorted by length, longest first, each dealt to the rank with the least audio so far (equal lengths: rank
        // r takes clips r, r+G, ...); every rank then packs its own clips into ragged batches (transcribe_clips)
        std::vector<int> order(n_clips);
        for (int i = 0; i < n_clips; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b]; });
        std::vector<std::vector<int>> shard(G);
        std::vector<int64_t> load(G, 0);
        double audio = 0.0;
        for (int i = 0; i < n_clips; ++i) {
            const int64_t len = offsets[order[i] + 1] - offsets[order[i]];
            const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            shard[r].push_back(order[i]);
            load[r] += len;
            audio += (double)len / 16000.0;
        }
        auto store = new_store(n_clips);
        ResultStore &R = *store;
        std::vector<double> wall_ms(G, 0.0);
        // No collective anywhere: utterances share nothing, every rank writes the result slots of its own clips, and the ranks never wait
        // for each other.  Each rank runs its batches through its replica's two-stream pipeline (transcribe_clips).
        for_each_rank(G, [&](int r) {
            if (shard[r].empty()) return;
            const auto t0 = std::chrono::steady_clock::now();
            transcribe_clips(*g->models[r], pcm, offsets, shard[r], opt, R);
            wall_ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
        g->wall_ms = wall_ms;
        g->wall_ms_max = *std::max_element(wall_ms.begin(), wall_ms.end());
        g->audio_s = audio;
        for (int r = 0; r < G; ++r) g->clips_per_rank[r] = (int32_t)shard[r].size();
        g->last_shard = shard;
        *results = publish_store(std::move(store), n_clips, opt && opt->timestamps);
    });
}

pk_status pk_group_last_stats(const pk_group *g, double *wall_ms_max, double *audio_seconds, int32_t *clips_per_rank) {
    return guard([&] {
        need(g, "group");
        if (wall_ms_max) *wall_ms_max = g->wall_ms_max;
        if (audio_seconds) *audio_seconds = g->audio_s;
        if (clips_per_rank) std::copy(g->clips_per_rank.begin(), g->clips_per_rank.end(), clips_per_rank);
    });
}

// Debug check, never part of a transcription: the token ids of the last pk_group_transcribe_pcm go rank by rank through device memory and
// ONE fixed-stride ncclAllGather ([clips_per_rank][2 + max_tokens] int32 -- the exchange a multi-PROCESS deployment ends with,
// parakeet.cpp_amd/shard.py) plus an ncclAllReduce(max) of the per-rank token maxima and wall times; every rank's copy of the gathered
// matrix must reproduce `results`.  RCCL is loaded here, on first use (rccl_dyn.hpp); without it: PK_ERR_UNSUPPORTED.
pk_status pk_group_verify_exchange(pk_group *g, const pk_result *results, int n_clips, int *rccl_ranks) {
    return guard([&] {
        need(g && results && n_clips > 0, "group/results/n_clips");
        const int G = (int)g->devices.size();
        size_t total = 0;
        for (auto &sh : g->last_shard) total += sh.size();
        need((int)total == n_clips, "results are not those of the last pk_group_transcribe_pcm");
        if (!g->rccl) {
            std::string why;
            g->rccl = rccl_api(&why);
            if (!g->rccl) fail(PK_ERR_UNSUPPORTED, "RCCL is not available on this host (%s)", why.c_str());
            g->comms.assign(G, nullptr);
            for (int r = 0; r < G; ++r) {                  // RCCL's init turns ANY pending HIP error into a failure: start from a clean slate on every device
                PK_HIP(hipSetDevice(g->devices[r]));
                PK_HIP(hipDeviceSynchronize());
                (void)hipGetLastError();
            }
            PK_NCCL(g->rccl, CommInitAll(g->comms.data(), G, g->devices.data()));
            g->streams.assign(G, nullptr);
            for (int r = 0; r < G; ++r) {
                PK_HIP(hipSetDevice(g->devices[r]));
                PK_HIP(hipStreamCreateWithFlags(&g->streams[r], hipStreamNonBlocking));
            }
        }
        const RcclApi *N = g->rccl;
        if (rccl_ranks) PK_NCCL(N, CommCount(g->comms[0], rccl_ranks));
        const auto &shard = g->last_shard;
        size_t cap = 1;
        int local_max = 0;
        std::vector<std::vector<int>> rank_max(G, std::vector<int>(2, 0));
        for (int r = 0; r < G; ++r) {
            cap = std::max(cap, shard[r].size());
            for (int c : shard[r]) rank_max[r][0] = std::max(rank_max[r][0], (int)results[c].n_tokens);
            rank_max[r][1] = (int)std::min(g->wall_ms[r] * 1000.0, 2.0e9);                    // microseconds
            local_max = std::max(local_max, rank_max[r][0]);
        }
        std::vector<int *> dmax(G, nullptr);
        std::vector<int32_t *> dmat(G, nullptr), dall(G, nullptr);
        struct Guard {
            std::vector<int *> &a; std::vector<int32_t *> &b, &c; std::vector<int> &dev;
            ~Guard() { for (size_t r = 0; r < dev.size(); ++r) { (void)hipSetDevice(dev[r]); if (a[r]) (void)hipFree(a[r]); if (b[r]) (void)hipFree(b[r]); if (c[r]) (void)hipFree(c[r]); } }
        } free_all{dmax, dmat, dall, g->devices};
        for (int r = 0; r < G; ++r) {
            PK_HIP(hipSetDevice(g->devices[r]));
            PK_HIP(hipMalloc(reinterpret_cast<void **>(&dmax[r]), 2 * sizeof(int)));
            PK_HIP(hipMemcpyAsync(dmax[r], rank_max[r].data(), 2 * sizeof(int), hipMemcpyHostToDevice, g->streams[r]));
        }
        PK_NCCL(N, GroupStart());
        for (int r = 0; r < G; ++r) {
            PK_HIP(hipSetDevice(g->devices[r]));
            PK_NCCL(N, AllReduce(dmax[r], dmax[r], 2, ncclInt32, ncclMax, g->comms[r], g->streams[r]));
        }
        PK_NCCL(N, GroupEnd());
        int reduced[2] = {0, 0};
        PK_HIP(hipSetDevice(g->devices[0]));
        PK_HIP(hipMemcpyAsync(reduced, dmax[0], sizeof(reduced), hipMemcpyDeviceToHost, g->streams[0]));
        PK_HIP(hipStreamSynchronize(g->streams[0]));
        const int max_tok = reduced[0];
        if (max_tok != local_max) fail(PK_ERR_HIP, "RCCL all-reduce(max) returned %d tokens, the ranks hold %d", max_tok, local_max);
        if (std::abs(reduced[1] / 1000.0 - g->wall_ms_max) > 1.0) fail(PK_ERR_HIP, "RCCL all-reduce(max) of the wall times returned %d us", reduced[1]);
        const size_t stride = 2 + (size_t)max_tok, per_rank = cap * stride;
        std::vector<std::vector<int32_t>> hmat(G);
        for (int r = 0; r < G; ++r) {                      // row = [global clip index, n_tokens, ids...] ; unused rows: index -1
            hmat[r].assign(per_rank, 0);
            for (size_t i = 0; i < cap; ++i) hmat[r][i * stride] = -1;
            for (size_t i = 0; i < shard[r].size(); ++i) {
                const int c = shard[r][i];
                int32_t *row = hmat[r].data() + i * stride;
                row[0] = c;
                row[1] = results[c].n_tokens;
                std::copy(results[c].token_ids, results[c].token_ids + results[c].n_tokens, row + 2);
            }
            PK_HIP(hipSetDevice(g->devices[r]));
            PK_HIP(hipMalloc(reinterpret_cast<void **>(&dmat[r]), per_rank * 4));
            PK_HIP(hipMalloc(reinterpret_cast<void **>(&dall[r]), per_rank * 4 * G));
            PK_HIP(hipMemcpyAsync(dmat[r], hmat[r].data(), per_rank * 4, hipMemcpyHostToDevice, g->streams[r]));
        }
        PK_NCCL(N, GroupStart());
        for (int r = 0; r < G; ++r) {
            PK_HIP(hipSetDevice(g->devices[r]));
            PK_NCCL(N, AllGather(dmat[r], dall[r], per_rank, ncclInt32, g->comms[r], g->streams[r]));
        }
        PK_NCCL(N, GroupEnd());
        std::vector<int32_t> all(per_rank * G);
        for (int r = 0; r < G; ++r) {                      // EVERY rank's copy of the gathered matrix is checked
            PK_HIP(hipSetDevice(g->devices[r]));
            PK_HIP(hipMemcpyAsync(all.data(), dall[r], all.size() * 4, hipMemcpyDeviceToHost, g->streams[r]));
            PK_HIP(hipStreamSynchronize(g->streams[r]));
            int seen = 0;
            for (size_t row = 0; row < (size_t)G * cap; ++row) {
                const int32_t *p = all.data() + row * stride;
                if (p[0] < 0) continue;
                need(p[0] < n_clips && p[1] >= 0 && p[1] <= max_tok, "gathered token matrix row");
                if (results[p[0]].n_tokens != p[1] || !std::equal(p + 2, p + 2 + p[1], results[p[0]].token_ids))
                    fail(PK_ERR_HIP, "RCCL all-gather: rank %d holds different token ids for clip %d", r, p[0]);
                ++seen;
            }
            if (seen != n_clips) fail(PK_ERR_HIP, "RCCL all-gather: rank %d holds %d of %d clips", r, seen, n_clips);
        }
    });
}

void pk_results_free(pk_result *results, int n_clips) {
    if (!results || n_clips < 0) return;
    delete reinterpret_cast<ResultStore *>(const_cast<char *>(results[n_clips].text));
}

pk_status pk_read_wav(const char *path, float **pcm, int64_t *n_samples, int *sample_rate) {
    return guard([&] {
        need(path && pcm && n_samples && sample_rate, "path/pcm/n_samples/sample_rate");
        std::vector<float> mono;
        int sr = 0;
        read_wav(path, mono, sr);
        float *p = static_cast<float *>(malloc((mono.size() ? mono.size() : 1) * sizeof(float)));
        if (!p) fail(PK_ERR_IO, "out of memory");
        memcpy(p, mono.data(), mono.size() * sizeof(float));
        *pcm = p;
        *n_samples = (int64_t)mono.size();
        *sample_rate = sr;
    });
}
/* read_audio(path, target_sample_rate) (audio_io.cpp:453-483) restricted to RIFF/WAVE: decode, mono downmix, resample to
 * target_rate with the reference's Kaiser-windowed sinc (sinc_resample, :123-195). */
pk_status pk_read_audio(const char *path, int target_rate, float **pcm, int64_t *n_samples, int *original_rate) {
    return guard([&] {
        need(path && pcm && n_samples && target_rate > 0, "path/pcm/n_samples/target_rate");
        std::vector<float> mono, out;
        int sr = 0;
        read_wav(path, mono, sr);
        if (original_rate) *original_rate = sr;
        sinc_resample(mono.data(), mono.size(), sr, target_rate, out);
        float *p = static_cast<float *>(malloc((out.size() ? out.size() : 1) * sizeof(float)));
        if (!p) fail(PK_ERR_IO, "out of memory");
        memcpy(p, out.data(), out.size() * sizeof(float));
        *pcm = p;
        *n_samples = (int64_t)out.size();
    });
}
/* resample() / read_audio(const float *pcm, n, sample_rate, target) (audio_io.cpp:250-262,506-514). */
static void hand_over(const std::vector<float> &v, float **pcm, int64_t *n) {
    float *p = static_cast<float *>(malloc((v.size() ? v.size() : 1) * sizeof(float)));
    if (!p) fail(PK_ERR_IO, "out of memory");
    memcpy(p, v.data(), v.size() * sizeof(float));
    *pcm = p;
    *n = (int64_t)v.size();
}
/* read_audio(const uint8_t *data, size_t len, target) (audio_io.cpp:485-493), RIFF/WAVE images */
pk_status pk_read_audio_memory(const void *data, size_t len, int target_rate, float **pcm, int64_t *n_samples, int *original_rate, int *n_channels) {
    return guard([&] {
        need(data && len > 0 && pcm && n_samples && target_rate > 0, "data/len/pcm/n_samples/target_rate");
        std::vector<float> mono, out;
        int sr = 0;
        parse_wav(static_cast<const uint8_t *>(data), len, nullptr, mono, sr, n_channels, false);
        if (original_rate) *original_rate = sr;
        sinc_resample(mono.data(), mono.size(), sr, target_rate, out);
        hand_over(out, pcm, n_samples);
    });
}
/* get_audio_duration (audio_io.cpp:527-586): header walk only */
pk_status pk_audio_info(const char *path, int *sample_rate, int *n_channels, int64_t *n_frames) {
    return guard([&] {
        need(path != nullptr, "path");
        FILE *f = fopen(path, "rb");
        if (!f) fail(PK_ERR_IO, "Failed to open audio file: %s", path);
        std::vector<uint8_t> head(1 << 16);
        fseek(f, 0, SEEK_END);
        const long total = ftell(f);
        fseek(f, 0, SEEK_SET);
        const size_t got = fread(head.data(), 1, head.size(), f);
        fclose(f);
        head.resize(got);
        // only the head of the file is read; the data chunk's declared size (clamped to the file length) gives the frame count
        std::vector<float> mono;
        int sr = 0, ch = 0;
        parse_wav(head.data(), head.size(), path, mono, sr, &ch, true, total > 0 ? (size_t)total : 0);
        if (sample_rate) *sample_rate = sr;
        if (n_channels) *n_channels = ch;
        if (n_frames) *n_frames = (int64_t)wav_info_frames();
    });
}

pk_status pk_resample(const float *pcm, int64_t n, int src_rate, int dst_rate, float **out, int64_t *n_out) {
    return guard([&] {
        need(pcm && out && n_out && n >= 0 && src_rate > 0 && dst_rate > 0, "pcm/out/n/rates");
        std::vector<float> r;
        sinc_resample(pcm, (size_t)n, src_rate, dst_rate, r);
        float *p = static_cast<float *>(malloc((r.size() ? r.size() : 1) * sizeof(float)));
        if (!p) fail(PK_ERR_IO, "out of memory");
        memcpy(p, r.data(), r.size() * sizeof(float));
        *out = p;
        *n_out = (int64_t)r.size();
    });
}
void pk_free(void *p) { free(p); }

/* ---- host-side text ---------------------------------------------------------------------------------------- */
int pk_vocab_size(const pk_model *m) { return m ? (int)m->m->tok.vocab_size() : 0; }

// The text entry points return a count (>= 0) or -1; like every other entry point no C++ exception may cross the ABI, so the
// bodies run under guard() and a failure (bad argument, std::bad_alloc) becomes -1 with pk_last_error() set.
int pk_detokenize(const pk_model *m, const int32_t *ids, int n, char *out, int cap) {
    int ret = -1;
    guard([&] {
        need(m && n >= 0 && (ids || n == 0), "model/ids/n");
        std::vector<int> v(ids, ids + n);
        const std::string s = m->m->tok.decode(v);
        if (out && cap > 0) {
            const int c = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
            memcpy(out, s.data(), c);
            out[c] = 0;
        }
        ret = (int)s.size();
    });
    return ret;
}

int pk_tokenize(const pk_model *m, const char *text, int32_t *ids, int cap) {
    int ret = -1;
    guard([&] {
        need(m && text && (ids || cap <= 0), "model/text/ids");
        const auto v = m->m->tok.encode(text);
        for (int i = 0; i < (int)v.size() && i < cap; ++i) ids[i] = v[i];
        ret = (int)v.size();
    });
    return ret;
}

pk_status pk_set_boost_tokens(pk_model *h, const int32_t *ids, const int32_t *offsets, int n_phrases, float boost_score) {
    return guard([&] {
        need(h && n_phrases >= 0 && (n_phrases == 0 || (ids && offsets)), "model/ids/offsets/n_phrases");
        std::vector<std::vector<int>> ph;
        for (int i = 0; i < n_phrases; ++i) {
            need(offsets[i + 1] >= offsets[i], "offsets must be non-decreasing");
            ph.emplace_back(ids + offsets[i], ids + offsets[i + 1]);
        }
        h->m->set_boost(ph, boost_score);
    });
}

pk_status pk_set_boost_phrases(pk_model *h, const char *const *phrases, int n_phrases, float boost_score) {
    return guard([&] {
        need(h && n_phrases >= 0 && (n_phrases == 0 || phrases), "model/phrases/n_phrases");
        h->m->set_boost(encode_phrases(*h->m, phrases, n_phrases), boost_score);
    });
}

int pk_boost_trie_size(const pk_model *m) { return m && m->m->boost_on ? m->m->trie_nodes : 0; }

int pk_group_timestamps(const pk_model *m, const int32_t *ids, const int32_t *start, const int32_t *end, const float *conf, int n,
                        int sentences, char *words, int cap, float *wstart, float *wend, float *wconf, int wcap) {
    int ret = -1;
    guard([&] {
        need(m && n >= 0 && (n == 0 || (ids && start && end)), "model/ids/start/end/n");
        std::vector<TimestampedToken> tt(n);
        for (int i = 0; i < n; ++i) tt[i] = {ids[i], start[i], end[i], conf ? conf[i] : 1.0f};
        const auto w = group_timestamps(tt, m->m->tok.pieces(), sentences != 0);
        std::string joined;
        for (size_t i = 0; i < w.size(); ++i) {
            if (i) joined += '\n';
            joined += w[i].word;
            if ((int)i < wcap) {
                if (wstart) wstart[i] = w[i].start;
                if (wend) wend[i] = w[i].end;
                if (wconf) wconf[i] = w[i].confidence;
            }
        }
        if (words && cap > 0) {
            const int c = (int)joined.size() < cap - 1 ? (int)joined.size() : cap - 1;
            memcpy(words, joined.data(), c);
            words[c] = 0;
        }
        ret = (int)w.size();
    });
    return ret;
}

/* ---- diagnostics ---------------------------------------------------------------------------------- */
namespace {
struct Scratch {   // device scratch for the model-less diagnostic entry points
    DevBuf a, b, c, d, e;
};
void diag_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(PK_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU path)");
}
}  // namespace

pk_status pk_diag_math(int fn, const float *in, float *out, int64_t n) {
    return guard([&] {
        need(in && out && n > 0, "in/out/n");
        diag_device();
        Scratch s;
        s.a.reserve(n * 4);
        s.b.reserve(n * 4);
        PK_HIP(hipMemcpy(s.a.p, in, n * 4, hipMemcpyHostToDevice));
        launch_math(fn, s.a.as<float>(), s.b.as<float>(), n, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.b.p, n * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_math_exhaustive(int fn, uint64_t *checked, uint64_t *mismatches, uint64_t *first_bad) {
    return guard([&] {
        need(fn == 3 || fn == 4 || fn == 13 || fn == 14, "fn must be 3, 4, 13 or 14");
        need(checked && mismatches && first_bad, "checked/mismatches/first_bad");
        diag_device();
        Scratch s;
        s.a.reserve(3 * sizeof(uint64_t));
        const uint64_t init[3] = {0, 0, 1ull << 32};
        PK_HIP(hipMemcpy(s.a.p, init, sizeof init, hipMemcpyHostToDevice));
        launch_math_exhaustive(fn, s.a.as<unsigned long long>(), nullptr);
        PK_CHECK_LAUNCH();
        uint64_t res[3];
        PK_HIP(hipMemcpy(res, s.a.p, sizeof res, hipMemcpyDeviceToHost));
        *checked = res[0]; *mismatches = res[1]; *first_bad = res[2];
    });
}

pk_status pk_diag_gemm(int M, int N, int K, const float *A, const float *W, const float *bias, int epi, const float *resid,
                       float alpha, float *out) {
    return guard([&] {
        need(A && W && out && M > 0 && N > 0 && K > 0, "A/W/out/M/N/K");
        need(K % 32 == 0, "K must be a multiple of 32");
        need(epi >= 0 && epi <= 4, "epi");
        need(epi != EPI_RESID || resid, "resid");
        diag_device();
        const int wrows = epi == EPI_GLU ? 2 * N : N;
        Scratch s;
        s.a.reserve((size_t)M * K * 4);
        s.b.reserve((size_t)wrows * K * 4);
        s.c.reserve((size_t)wrows * 4);
        s.d.reserve((size_t)M * N * 4);
        s.e.reserve((size_t)M * N * 4);
        PK_HIP(hipMemcpy(s.a.p, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.b.p, W, (size_t)wrows * K * 4, hipMemcpyHostToDevice));
        if (bias) PK_HIP(hipMemcpy(s.c.p, bias, (size_t)wrows * 4, hipMemcpyHostToDevice));
        if (resid) PK_HIP(hipMemcpy(s.d.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice));
        GemmArgs g{s.a.as<float>(), K, s.b.as<float>(), K, bias ? s.c.as<float>() : nullptr, s.e.as<float>(), N,
                   resid ? s.d.as<float>() : nullptr, N, alpha, M, N, K};
        launch_gemm(g, epi, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.e.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    });
}

// The bf16 diag products run as a streaming session runs them: weights also as operand tiles of the small-M kernel (GemmArgs::W_t16) where the
// shape allows (rows % 16 == 0, K % 32 == 0).  pk_diag_smallm_bf16_tiles(0) keeps the natural layout only (both are tested, bit for bit).
static std::atomic<int> g_diag_tiles{1};
pk_status pk_diag_smallm_bf16_tiles(int on) { g_diag_tiles.store(on ? 1 : 0); return PK_OK; }
static const float *diag_operand_tiles(DevBuf &buf, const float *w16, int64_t rows, int K) {
    if (!g_diag_tiles.load() || rows % 16 != 0 || K % 32 != 0) return nullptr;
    buf.reserve((size_t)rows * K * 2);
    launch_tile_copy_bf16(w16, buf.as<float>(), rows, K, K, nullptr);
    return buf.as<float>();
}

pk_status pk_diag_gemm_bf16(int M, int N, int K, const float *A, const float *W, const float *bias, int epi, const float *resid,
                            float alpha, float *out) {
    return guard([&] {
        need(A && W && out && M > 0 && N > 0 && K > 0, "A/W/out/M/N/K");
        need(K % 64 == 0, "K must be a multiple of 64");
        need(epi >= 0 && epi <= 4, "epi");
        need(epi != EPI_RESID || resid, "resid");
        diag_device();
        const int wrows = epi == EPI_GLU ? 2 * N : N;
        std::vector<uint16_t> w16((size_t)wrows * K);
        for (size_t i = 0; i < w16.size(); ++i) {               // weights: round to nearest even on the host, as at upload
            uint32_t u;
            memcpy(&u, &W[i], 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            w16[i] = (uint16_t)(u >> 16);
        }
        Scratch s;
        s.a.reserve((size_t)M * K * 4);
        s.b.reserve((size_t)wrows * K * 2);
        s.c.reserve((size_t)wrows * 4);
        s.d.reserve((size_t)M * N * 4);
        s.e.reserve((size_t)M * N * 4);
        PK_HIP(hipMemcpy(s.a.p, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.b.p, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
        if (bias) PK_HIP(hipMemcpy(s.c.p, bias, (size_t)wrows * 4, hipMemcpyHostToDevice));
        if (resid) PK_HIP(hipMemcpy(s.d.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice));
        GemmArgs g{s.a.as<float>(), K, s.b.as<float>(), K, bias ? s.c.as<float>() : nullptr, s.e.as<float>(), N,
                   resid ? s.d.as<float>() : nullptr, N, alpha, M, N, K};
        DevBuf wt_buf;
        g.W_t16 = diag_operand_tiles(wt_buf, s.b.as<float>(), wrows, K);
        launch_gemm_bf16(g, epi, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.e.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    });
}

/* the same product with the activations handed over as bf16 (GemmArgs::a_bf16: what the producing kernels of the bf16 mode store) */
pk_status pk_diag_gemm_bf16_a16(int M, int N, int K, const float *A, const float *W, const float *bias, int epi, const float *resid,
                                float alpha, float *out) {
    return guard([&] {
        need(A && W && out && M > 0 && N > 0 && K > 0, "A/W/out/M/N/K");
        need(K % 64 == 0, "K must be a multiple of 64");
        need(epi >= 0 && epi <= 4, "epi");
        need(epi != EPI_RESID || resid, "resid");
        diag_device();
        const int wrows = epi == EPI_GLU ? 2 * N : N;
        auto rne = [](float f) {
            uint32_t u;
            memcpy(&u, &f, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        std::vector<uint16_t> w16((size_t)wrows * K), a16((size_t)M * K);
        for (size_t i = 0; i < w16.size(); ++i) w16[i] = rne(W[i]);
        for (size_t i = 0; i < a16.size(); ++i) a16[i] = rne(A[i]);
        Scratch s;
        s.a.reserve((size_t)M * K * 2);
        s.b.reserve((size_t)wrows * K * 2);
        s.c.reserve((size_t)wrows * 4);
        s.d.reserve((size_t)M * N * 4);
        s.e.reserve((size_t)M * N * 4);
        PK_HIP(hipMemcpy(s.a.p, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.b.p, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
        if (bias) PK_HIP(hipMemcpy(s.c.p, bias, (size_t)wrows * 4, hipMemcpyHostToDevice));
        if (resid) PK_HIP(hipMemcpy(s.d.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice));
        GemmArgs g{s.a.as<float>(), K, s.b.as<float>(), K, bias ? s.c.as<float>() : nullptr, s.e.as<float>(), N,
                   resid ? s.d.as<float>() : nullptr, N, alpha, M, N, K};
        g.a_bf16 = 1;
        DevBuf wt_buf;
        g.W_t16 = diag_operand_tiles(wt_buf, s.b.as<float>(), wrows, K);
        launch_gemm_bf16(g, epi, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.e.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    });
}

/* ... and with the LayerNorm of the input rows folded into the product (the streaming chunks of the tolerance-class mode) */
pk_status pk_diag_ln_gemm_bf16(int M, int N, int K, const float *A, const float *gamma, const float *beta, float eps, const float *W,
                               const float *bias, int epi, const float *resid, float alpha, float *out) {
    return guard([&] {
        need(A && gamma && beta && W && out && M > 0 && N > 0 && K > 0, "A/gamma/beta/W/out/M/N/K");
        need(epi >= 0 && epi <= 4, "epi");
        need(epi != EPI_RESID || resid, "resid");
        diag_device();
        const int wrows = epi == EPI_GLU ? 2 * N : N;
        std::vector<uint16_t> w16((size_t)wrows * K);
        for (size_t i = 0; i < w16.size(); ++i) {
            uint32_t u;
            memcpy(&u, &W[i], 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            w16[i] = (uint16_t)(u >> 16);
        }
        Scratch s;
        DevBuf gb;
        s.a.reserve((size_t)M * K * 4);
        s.b.reserve((size_t)wrows * K * 2);
        s.c.reserve((size_t)wrows * 4);
        s.d.reserve((size_t)M * N * 4);
        s.e.reserve((size_t)M * N * 4);
        gb.reserve((size_t)2 * K * 4);
        PK_HIP(hipMemcpy(s.a.p, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.b.p, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(gb.p, gamma, (size_t)K * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy((char *)gb.p + (size_t)K * 4, beta, (size_t)K * 4, hipMemcpyHostToDevice));
        if (bias) PK_HIP(hipMemcpy(s.c.p, bias, (size_t)wrows * 4, hipMemcpyHostToDevice));
        if (resid) PK_HIP(hipMemcpy(s.d.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice));
        GemmArgs g{s.a.as<float>(), K, s.b.as<float>(), K, bias ? s.c.as<float>() : nullptr, s.e.as<float>(), N,
                   resid ? s.d.as<float>() : nullptr, N, alpha, M, N, K};
        g.ln_g = gb.as<float>(); g.ln_b = gb.as<float>() + K; g.ln_eps = eps;
        DevBuf wt_buf;
        g.W_t16 = diag_operand_tiles(wt_buf, s.b.as<float>(), wrows, K);
        if (!gemm_smallm_bf16_ln_applies(g, epi)) fail(PK_ERR_UNSUPPORTED, "pk_diag_ln_gemm_bf16: M <= %d, K = 256 * (1 .. 8; glu: .. 4)", kSmallMRowsBf16);
        launch_gemm_bf16(g, epi, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.e.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_ln2_gemm_bf16(int M, int N, int K, const float *A, const float *pre_gamma, const float *pre_beta, const float *gamma, const float *beta,
                                float eps, const float *W, const float *bias, float *out, float *pre_out) {
    return guard([&] {
        need(M > 0 && N > 0 && K > 0 && A && pre_gamma && pre_beta && gamma && beta && W && out && pre_out, "arguments");
        diag_device();
        std::vector<uint16_t> w16((size_t)N * K);
        for (size_t i = 0; i < w16.size(); ++i) {
            uint32_t u;
            memcpy(&u, &W[i], 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            w16[i] = (uint16_t)(u >> 16);
        }
        DevBuf a, w, b, gb, o, po, wt_buf;
        auto up = [&](DevBuf &buf, const void *src, size_t bytes) { buf.reserve(bytes); PK_HIP(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice)); };
        up(a, A, (size_t)M * K * 4);
        up(w, w16.data(), w16.size() * 2);
        if (bias) up(b, bias, (size_t)N * 4);
        gb.reserve((size_t)4 * K * 4);
        const float *four[4] = {pre_gamma, pre_beta, gamma, beta};
        for (int i = 0; i < 4; ++i) PK_HIP(hipMemcpy(gb.as<float>() + (size_t)i * K, four[i], (size_t)K * 4, hipMemcpyHostToDevice));
        o.reserve((size_t)M * N * 4);
        po.reserve((size_t)M * K * 4);
        GemmArgs g{a.as<float>(), K, w.as<float>(), K, bias ? b.as<float>() : nullptr, o.as<float>(), N, nullptr, 0, 1.0f, M, N, K};
        g.fast_act = 1;
        g.pre_g = gb.as<float>(); g.pre_b = gb.as<float>() + K; g.ln_g = gb.as<float>() + 2 * (size_t)K; g.ln_b = gb.as<float>() + 3 * (size_t)K; g.ln_eps = eps;
        g.pre_out = po.as<float>(); g.pre_ldo = K;
        g.W_t16 = diag_operand_tiles(wt_buf, w.as<float>(), N, K);
        if (!gemm_smallm_bf16_pre_applies(g, EPI_SILU)) fail(PK_ERR_UNSUPPORTED, "pk_diag_ln2_gemm_bf16: M <= %d, K = 256 * (1 .. 8)", kSmallMRowsBf16);
        launch_gemm_bf16(g, EPI_SILU, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, o.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        PK_HIP(hipMemcpy(pre_out, po.p, (size_t)M * K * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_glu_dwconv_bf16(int n_streams, int c, int d, const float *A, const float *gamma, const float *beta, float eps, const float *W,
                                  const float *bias, const float *cache_in, int has_cache, const float *dw_w, const float *dw_bias,
                                  const float *bn_mean, const float *bn_rstd, const float *bn_g, const float *bn_b, int fused, float *out,
                                  float *cache_out) {
    return guard([&] {
        need(n_streams > 0 && c > 0 && d > 0 && A && W && cache_in && dw_w && dw_bias && bn_mean && bn_rstd && bn_g && bn_b && out && cache_out, "arguments");
        need((gamma != nullptr) == (beta != nullptr), "gamma and beta: both or neither");
        diag_device();
        const int M = n_streams * c, K = d, N = d;
        std::vector<uint16_t> w16((size_t)2 * N * K);
        for (size_t i = 0; i < w16.size(); ++i) {
            uint32_t u;
            memcpy(&u, &W[i], 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            w16[i] = (uint16_t)(u >> 16);
        }
        DevBuf a, w, b, gb, ci, co, par, glu, o;
        auto up = [&](DevBuf &buf, const void *src, size_t bytes) { buf.reserve(bytes); PK_HIP(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice)); };
        up(a, A, (size_t)M * K * 4);
        up(w, w16.data(), w16.size() * 2);
        if (bias) up(b, bias, (size_t)2 * N * 4);
        if (gamma) { gb.reserve((size_t)2 * K * 4); PK_HIP(hipMemcpy(gb.p, gamma, (size_t)K * 4, hipMemcpyHostToDevice)); PK_HIP(hipMemcpy((char *)gb.p + (size_t)K * 4, beta, (size_t)K * 4, hipMemcpyHostToDevice)); }
        up(ci, cache_in, (size_t)n_streams * 8 * d * 4);
        co.reserve((size_t)n_streams * 8 * d * 4);
        par.reserve((size_t)(9 + 5) * d * 4);
        float *pp = par.as<float>();
        PK_HIP(hipMemcpy(pp, dw_w, (size_t)9 * d * 4, hipMemcpyHostToDevice));
        const float *five[5] = {dw_bias, bn_mean, bn_rstd, bn_g, bn_b};
        for (int i = 0; i < 5; ++i) PK_HIP(hipMemcpy(pp + (size_t)(9 + i) * d, five[i], (size_t)d * 4, hipMemcpyHostToDevice));
        glu.reserve((size_t)M * N * 4);
        o.reserve((size_t)M * N * 4);
        GemmArgs g{a.as<float>(), K, w.as<float>(), K, bias ? b.as<float>() : nullptr, glu.as<float>(), N, nullptr, 0, 1.0f, M, N, K};
        g.fast_act = 1;
        if (gamma) { g.ln_g = gb.as<float>(); g.ln_b = gb.as<float>() + K; g.ln_eps = eps; }
        if (!(gamma ? gemm_smallm_bf16_ln_applies(g, EPI_GLU) : gemm_smallm_bf16_applies(g, EPI_GLU)))
            fail(PK_ERR_UNSUPPORTED, "pk_diag_glu_dwconv_bf16: M <= %d, d = 256 * (1 .. 4)", kSmallMRowsBf16);
        DevBuf wt_buf;
        g.W_t16 = diag_operand_tiles(wt_buf, w.as<float>(), 2 * N, K);
        DwTail tail{ci.as<float>(), co.as<float>(), has_cache, c, pp, pp + 9 * (size_t)d, pp + 10 * (size_t)d, pp + 11 * (size_t)d, pp + 12 * (size_t)d, pp + 13 * (size_t)d};
        if (fused) {
            if (!gemm_smallm_bf16_dw_applies(g, EPI_GLU, c, 9)) fail(PK_ERR_UNSUPPORTED, "pk_diag_glu_dwconv_bf16: the fused tail takes c = 1, 2 or 4 frames per stream");
            g.dw_tail = &tail; g.out = o.as<float>();
            launch_gemm_bf16(g, EPI_GLU, nullptr);
        } else {
            launch_gemm_bf16(g, EPI_GLU, nullptr);
            launch_stream_dwconv(glu.as<float>(), tail.cache_in, has_cache, n_streams, c, d, 9, tail.w, tail.bias, tail.bn_mean, tail.bn_rstd, tail.bn_g, tail.bn_b,
                                 o.as<float>(), tail.cache_out, nullptr, 0);
        }
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, o.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        PK_HIP(hipMemcpy(cache_out, co.p, (size_t)n_streams * 8 * d * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_ffn_bf16_smallm(int M, int d, int f, const float *x, const float *gamma, const float *beta, float eps, const float *W1, const float *b1,
                                  const float *W2, const float *b2, int act_tiles, float *out) {
    return guard([&] {
        need(M > 0 && d > 0 && f > 0 && x && gamma && beta && W1 && b1 && W2 && b2 && out, "arguments");
        diag_device();
        auto to16 = [](const float *w, size_t n) {
            std::vector<uint16_t> v(n);
            for (size_t i = 0; i < n; ++i) {
                uint32_t u;
                memcpy(&u, &w[i], 4);
                u += 0x7fffu + ((u >> 16) & 1u);
                v[i] = (uint16_t)(u >> 16);
            }
            return v;
        };
        const std::vector<uint16_t> w1 = to16(W1, (size_t)f * d), w2 = to16(W2, (size_t)d * f);
        DevBuf xb, gb, w1b, w2b, b1b, b2b, hb, ob;
        auto up = [&](DevBuf &buf, const void *src, size_t bytes) { buf.reserve(bytes); PK_HIP(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice)); };
        up(xb, x, (size_t)M * d * 4); up(w1b, w1.data(), w1.size() * 2); up(w2b, w2.data(), w2.size() * 2); up(b1b, b1, (size_t)f * 4); up(b2b, b2, (size_t)d * 4);
        gb.reserve((size_t)2 * d * 4);
        PK_HIP(hipMemcpy(gb.p, gamma, (size_t)d * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy((char *)gb.p + (size_t)d * 4, beta, (size_t)d * 4, hipMemcpyHostToDevice));
        hb.reserve((size_t)((M + 7) / 8 * 8) * f * 2);
        up(ob, x, (size_t)M * d * 4);                                   // the residual stream: out = x + 0.5 * ffn(LN(x))
        GemmArgs g1{xb.as<float>(), d, w1b.as<float>(), d, b1b.as<float>(), hb.as<float>(), f, nullptr, 0, 1.0f, M, f, d};
        g1.ln_g = gb.as<float>(); g1.ln_b = gb.as<float>() + d; g1.ln_eps = eps; g1.out_bf16 = 1; g1.fast_act = 1; g1.out_t8 = act_tiles ? 1 : 0;
        GemmArgs g2{hb.as<float>(), f, w2b.as<float>(), f, b2b.as<float>(), ob.as<float>(), d, ob.as<float>(), d, 0.5f, M, d, f};
        g2.a_bf16 = 1; g2.a_t8 = act_tiles ? 1 : 0;
        DevBuf wt1, wt2;
        g1.W_t16 = diag_operand_tiles(wt1, w1b.as<float>(), f, d);
        g2.W_t16 = diag_operand_tiles(wt2, w2b.as<float>(), d, f);
        if (!gemm_smallm_bf16_ln_applies(g1, EPI_SILU) || !gemm_smallm_bf16_applies(g2, EPI_RESID))
            fail(PK_ERR_UNSUPPORTED, "pk_diag_ffn_bf16_smallm: M <= %d (act_tiles: M %% 8 == 0), d = 256 * (1 .. 8), f %% 256 == 0", kSmallMRowsBf16);
        launch_gemm_bf16(g1, EPI_SILU, nullptr);
        launch_gemm_bf16(g2, EPI_RESID, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, ob.p, (size_t)M * d * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_layernorm(const float *x, int64_t rows, int d, const float *gamma, const float *beta, float eps, float *y) {
    return guard([&] {
        need(x && gamma && beta && y && rows > 0 && d > 0 && d <= 1024, "x/gamma/beta/y/rows/d (d <= 1024)");
        diag_device();
        Scratch s;
        s.a.reserve((size_t)rows * d * 4);
        s.b.reserve((size_t)d * 4);
        s.c.reserve((size_t)d * 4);
        s.d.reserve((size_t)rows * d * 4);
        PK_HIP(hipMemcpy(s.a.p, x, (size_t)rows * d * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.b.p, gamma, (size_t)d * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.c.p, beta, (size_t)d * 4, hipMemcpyHostToDevice));
        launch_layernorm(s.a.as<float>(), rows, d, s.b.as<float>(), s.c.as<float>(), eps, s.d.as<float>(), nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(y, s.d.p, (size_t)rows * d * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_ln_gemm(int M, int N, int K, const float *A, const float *pre_gamma, const float *pre_beta, const float *gamma, const float *beta, float eps,
                          const float *W, const float *bias, int epi, int fold, float *out, float *y1) {
    return guard([&] {
        need(A && gamma && beta && W && out && M > 0 && N > 0 && K > 0 && K <= 1024, "A/gamma/beta/W/out/M/N/K (K <= 1024)");
        need((pre_gamma == nullptr) == (pre_beta == nullptr), "pre_gamma and pre_beta: both or neither");
        need(epi == EPI_NONE || epi == EPI_RELU || epi == EPI_SILU || epi == EPI_GLU, "epi: none / relu / silu / glu");
        diag_device();
        const int wrows = epi == EPI_GLU ? 2 * N : N;
        DevBuf a, w, b, gb, o, n, x1;
        auto up = [&](DevBuf &buf, const void *src, size_t bytes) { buf.reserve(bytes); PK_HIP(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice)); };
        up(a, A, (size_t)M * K * 4);
        up(w, W, (size_t)wrows * K * 4);
        if (bias) up(b, bias, (size_t)wrows * 4);
        gb.reserve((size_t)4 * K * 4);
        const float *four[4] = {gamma, beta, pre_gamma, pre_beta};
        for (int i = 0; i < 4; ++i) if (four[i]) PK_HIP(hipMemcpy(gb.as<float>() + (size_t)i * K, four[i], (size_t)K * 4, hipMemcpyHostToDevice));
        const float *dg = gb.as<float>(), *db = dg + K, *dpg = dg + 2 * (size_t)K, *dpb = dg + 3 * (size_t)K;
        o.reserve((size_t)M * N * 4);
        n.reserve((size_t)M * K * 4);
        x1.reserve((size_t)M * K * 4);
        const float *X = a.as<float>();
        GemmArgs g{n.as<float>(), K, w.as<float>(), K, bias ? b.as<float>() : nullptr, o.as<float>(), N, nullptr, 0, 1.0f, M, N, K};
        if (fold) {
            if (pre_gamma) { launch_layernorm_then_stats(X, M, K, dpg, dpb, eps, x1.as<float>(), n.as<float>(), nullptr); X = x1.as<float>(); }
            else launch_layernorm_stats(X, M, K, eps, n.as<float>(), nullptr);
            g.A = X; g.ln_g = dg; g.ln_b = db; g.ln_eps = eps; g.ln_stats = n.as<float>();
            if (!gemm_ln_stats_applies(g, epi)) fail(PK_ERR_UNSUPPORTED, "pk_diag_ln_gemm: fold = 1 needs M > %d, K %% 32 == 0 and a wide (N >= 1024) or glu product", kSmallMRows);
        } else {
            if (pre_gamma) launch_layernorm2(X, M, K, dpg, dpb, dg, db, eps, x1.as<float>(), n.as<float>(), nullptr);
            else launch_layernorm(X, M, K, dg, db, eps, n.as<float>(), nullptr);
        }
        launch_gemm(g, epi, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, o.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        if (y1 && pre_gamma) PK_HIP(hipMemcpy(y1, x1.p, (size_t)M * K * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_sum64(const float *x, int rows, int n, float *out) {
    return guard([&] {
        need(x && out && rows > 0 && n > 0, "x/out/rows/n");
        diag_device();
        Scratch s;
        s.a.reserve((size_t)rows * n * 4);
        s.b.reserve((size_t)rows * 4);
        PK_HIP(hipMemcpy(s.a.p, x, (size_t)rows * n * 4, hipMemcpyHostToDevice));
        launch_sum64_rows(s.a.as<float>(), rows, n, s.b.as<float>(), nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.b.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
