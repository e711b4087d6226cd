// parakeet.cpp_amd/csrc/sortformer.cpp -- Sortformer speaker diarization (reference src/sortformer.cpp:41-121,
// include/parakeet/sortformer.hpp:28-129): the NEST FastConformer (the offline Conformer path of engine.cpp with xscaling, weights
// under "nest_encoder_."), projection_ 512 -> 192, an 18-layer post-LN TransformerEncoder (transformer.cpp), and the speaker head
// relu -> first_hidden_ -> relu -> output_proj_ -> sigmoid.  Everything between the PCM / feature upload and the [B][T][S]
// probabilities stays on the device; probs_to_segments (:71-113) and diarize_transcription (src/diarize.cpp:10-48) are host loops.
// Bit-identical to the oracle's orc_sortformer_forward (natural-k fp32 chains, the same polynomial sigmoid).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "stream.hpp"
#include "transformer.hpp"

namespace pk {

class Sortformer {
  public:
    Sortformer(const std::string &weights_path, const pk_sortformer_config &c, int device);
    ~Sortformer();
    // Sortformer::forward (:50-69): feats[B][Tm][mel] (host) -> probs[B][T][S] (host); returns T
    int forward_feats(const float *feats, int B, int Tm, float *probs);
    // preprocess_audio(normalize = false) (src/main.cpp:513-517, src/diarize.cpp:81-88) + forward
    int forward_pcm(const float *pcm, int n_clips, int64_t n_samples, float *probs);
    // Sortformer::diarize_chunk (:123-150): forward_chunk of the NEST encoder on the session's caches, then projection / transformer /
    // head on this chunk's frames only.  feats[n_frames][mel] -> probs[c][S]; returns c (0: everything was buffered)
    int forward_chunk(const float *feats, int n_frames, float *probs, int cap_frames);
    void reset_stream();
    pk_sortformer_config cfg;
    std::unique_ptr<Model> nest;

  private:
    std::unique_ptr<TransformerEncoder> tr_;
    std::unique_ptr<StreamBatch> stream_;
    std::vector<void *> allocs_;
    const float *pw_, *pb_, *fw_, *fb_, *ow_, *ob_;
    DevBuf xt_, h_, lg_;
    const float *upload(const SafeTensors &st, const std::string &name, int64_t want);
    void head(const float *d_enc, int B, int T, float *probs_host);
};

const float *Sortformer::upload(const SafeTensors &st, const std::string &name, int64_t want) {
    const HostTensor *t = st.find(name);
    if (!t) fail(PK_ERR_WEIGHTS, "missing tensor '%s'", name.c_str());
    if (t->dtype != "F32" || t->numel() != want) fail(PK_ERR_WEIGHTS, "tensor '%s': expected %lld F32 elements", name.c_str(), (long long)want);
    void *p = nullptr;
    PK_HIP(hipMalloc(&p, (size_t)want * sizeof(float)));
    allocs_.push_back(p);
    PK_HIP(hipMemcpy(p, t->f32(), (size_t)want * sizeof(float), hipMemcpyHostToDevice));
    return static_cast<const float *>(p);
}

Sortformer::Sortformer(const std::string &weights_path, const pk_sortformer_config &c, int device) : cfg(c) {
    if (c.max_speakers <= 0 || c.max_speakers > 64) fail(PK_ERR_INVALID, "max_speakers must be 1..64");
    if (c.nest.vocab_size != 0 || c.nest.ctc_vocab_size != 0) fail(PK_ERR_INVALID, "the NEST encoder config must be encoder-only (vocab_size = ctc_vocab_size = 0)");
    nest = std::make_unique<Model>(weights_path, "", c.nest);
    nest->to_gpu(device);
    tr_ = std::make_unique<TransformerEncoder>(weights_path, "transformer_.", c.transformer, device);
    SafeTensors st(weights_path);
    const int d = c.nest.hidden_size, dt = c.transformer.hidden_size, S = c.max_speakers;
    pw_ = upload(st, "projection_.weight", (int64_t)dt * d);   pb_ = upload(st, "projection_.bias", dt);
    fw_ = upload(st, "first_hidden_.weight", (int64_t)dt * dt); fb_ = upload(st, "first_hidden_.bias", dt);
    ow_ = upload(st, "output_proj_.weight", (int64_t)S * dt);  ob_ = upload(st, "output_proj_.bias", S);
}

Sortformer::~Sortformer() {
    if (nest && nest->on_gpu()) {
        (void)hipSetDevice(nest->device_);
        for (void *p : allocs_) (void)hipFree(p);
    }
}

// projection_ -> transformer_ -> speaker head on encoder frames d_enc[B*T][d] (device)
void Sortformer::head(const float *d_enc, int B, int T, float *probs_host) {
    Model &m = *nest;
    hipStream_t s = m.stream;
    const int d = cfg.nest.hidden_size, dt = cfg.transformer.hidden_size, S = cfg.max_speakers;
    const int64_t rows = (int64_t)B * T;
    xt_.reserve(rows * dt * 4); h_.reserve(rows * dt * 4); lg_.reserve(rows * S * 4);
    float *x = xt_.as<float>(), *h = h_.as<float>(), *lg = lg_.as<float>();
    {   // :55 projection_
        GemmArgs g{d_enc, d, pw_, d, pb_, x, dt, nullptr, 0, 1.0f, (int)rows, dt, d};
        launch_gemm(g, EPI_NONE, s);
    }
    tr_->forward_dev(x, B, T, s);                                   // :58
    launch_math(7, x, x, rows * dt, s);                             // :62 relu
    {   // :63-64 first_hidden_ + relu
        GemmArgs g{x, dt, fw_, dt, fb_, h, dt, nullptr, 0, 1.0f, (int)rows, dt, dt};
        launch_gemm(g, EPI_RELU, s);
    }
    {   // :65 output_proj_
        GemmArgs g{h, dt, ow_, dt, ob_, lg, S, nullptr, 0, 1.0f, (int)rows, S, dt};
        launch_gemm(g, EPI_NONE, s);
    }
    launch_math(3, lg, lg, rows * S, s);                            // :68 sigmoid
    PK_CHECK_LAUNCH();
    PK_HIP(hipMemcpyAsync(probs_host, lg, rows * S * 4, hipMemcpyDeviceToHost, s));
    PK_HIP(hipStreamSynchronize(s));
}

int Sortformer::forward_feats(const float *feats, int B, int Tm, float *probs) {
    Model &m = *nest;
    m.require_gpu();
    m.ws.size_for(m.cfg, B, 0, Tm);
    PK_HIP(hipMemcpyAsync(m.ws.feats.p, feats, (size_t)B * Tm * m.cfg.mel_bins * 4, hipMemcpyHostToDevice, m.stream));
    m.run_encoder(m.ws, m.ws.feats.as<float>(), B, Tm, -1, 0, m.stream);
    head(m.ws.x.as<float>(), B, m.ws.T, probs);
    return m.ws.T;
}

int Sortformer::forward_pcm(const float *pcm, int n_clips, int64_t n_samples, float *probs) {
    Model &m = *nest;
    m.require_gpu();
    const int Tm = (int)(1 + n_samples / 160);
    m.ws.size_for(m.cfg, n_clips, n_samples, Tm);
    PK_HIP(hipMemcpyAsync(m.ws.pcm.p, pcm, (size_t)n_clips * n_samples * 4, hipMemcpyHostToDevice, m.stream));
    m.run_mel(m.ws.pcm.as<float>(), n_clips, n_samples, m.ws.logmel.as<float>(), m.ws.feats.as<float>(), m.stream);
    m.run_encoder(m.ws, m.ws.feats.as<float>(), n_clips, Tm, -1, 0, m.stream);
    head(m.ws.x.as<float>(), n_clips, m.ws.T, probs);
    return m.ws.T;
}

int Sortformer::forward_chunk(const float *feats, int n_frames, float *probs, int cap_frames) {
    if (!stream_) stream_ = std::make_unique<StreamBatch>(*nest, 1, cfg.att_context_left, cfg.att_context_right);
    const float *d_enc = nullptr;
    const int c = stream_->encode_keep(feats, n_frames, &d_enc);
    if (c <= 0) {
        PK_HIP(hipStreamSynchronize(nest->stream));
        return 0;
    }
    if (c > cap_frames) fail(PK_ERR_INVALID, "probs holds %d frames, the chunk produces %d", cap_frames, c);
    head(d_enc, 1, c, probs);
    return c;
}

void Sortformer::reset_stream() {
    if (stream_) stream_->reset();
}

}  // namespace pk

using namespace pk;

struct pk_sortformer {
    std::unique_ptr<Sortformer> s;
};

static pk_status sf_guard(const std::function<void()> &fn) {
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

void pk_sortformer_config_preset(pk_sortformer_config *out) {       // make_sortformer_117m_config (sortformer.hpp:43-76)
    if (!out) return;
    memset(out, 0, sizeof *out);
    pk_config &n = out->nest;
    n.mel_bins = 128; n.subsampling_channels = 256; n.hidden_size = 512; n.num_layers = 17; n.num_heads = 8; n.ffn_intermediate = 2048;
    n.conv_kernel_size = 9; n.xscaling = 1; n.mel_normalize_off = 1; n.max_symbols_per_step = 10;
    snprintf(n.encoder_prefix, sizeof n.encoder_prefix, "nest_encoder_.");
    out->transformer.hidden_size = 192; out->transformer.num_layers = 18; out->transformer.num_heads = 8;
    out->transformer.ffn_intermediate = 768; out->transformer.pre_ln = 0; out->transformer.has_final_norm = 0;
    out->transformer.layer_norm_eps = 1e-5f;
    out->max_speakers = 4;
    out->activity_threshold = 0.5f;
    out->att_context_left = 70;                                    // sortformer.hpp:53-54
    out->att_context_right = 0;
}

pk_status pk_sortformer_load(const char *safetensors_path, const pk_sortformer_config *cfg, int device, pk_sortformer **out) {
    return sf_guard([&] {
        if (!safetensors_path || !cfg || !out) fail(PK_ERR_INVALID, "invalid argument: path/cfg/out");
        auto h = std::make_unique<pk_sortformer>();
        h->s = std::make_unique<Sortformer>(safetensors_path, *cfg, device);
        *out = h.release();
    });
}

void pk_sortformer_free(pk_sortformer *s) { delete s; }

pk_status pk_sortformer_forward(pk_sortformer *s, const float *feats, int B, int Tm, float *probs, int *T_out) {
    return sf_guard([&] {
        if (!s || !feats || !probs || B <= 0 || Tm <= 0) fail(PK_ERR_INVALID, "invalid argument: sortformer/feats/probs/B/Tm");
        const int T = s->s->forward_feats(feats, B, Tm, probs);
        if (T_out) *T_out = T;
    });
}

pk_status pk_sortformer_forward_pcm(pk_sortformer *s, const float *pcm, int n_clips, int64_t n_samples, float *probs, int *T_out) {
    return sf_guard([&] {
        if (!s || !pcm || !probs || n_clips <= 0 || n_samples <= 256) fail(PK_ERR_INVALID, "invalid argument: sortformer/pcm/probs/n_clips/n_samples");
        const int T = s->s->forward_pcm(pcm, n_clips, n_samples, probs);
        if (T_out) *T_out = T;
    });
}

pk_status pk_sortformer_diarize_chunk(pk_sortformer *s, const float *feats, int n_frames, float *probs, int cap_frames, int *T_out) {
    return sf_guard([&] {
        if (!s || !feats || !probs || n_frames <= 0 || cap_frames <= 0) fail(PK_ERR_INVALID, "invalid argument: sortformer/feats/probs/n_frames/cap_frames");
        const int c = s->s->forward_chunk(feats, n_frames, probs, cap_frames);
        if (T_out) *T_out = c;
    });
}

pk_status pk_sortformer_stream_reset(pk_sortformer *s) {
    return sf_guard([&] {
        if (!s) fail(PK_ERR_INVALID, "invalid argument: sortformer");
        s->s->reset_stream();
    });
}

// Sortformer::probs_to_segments (src/sortformer.cpp:71-113)
int pk_sortformer_segments(const float *probs, int T, int S, float threshold, int32_t *speaker, float *start, float *end, int cap) {
    if (!probs || T <= 0 || S <= 0) return 0;
    struct Seg { int spk; float a, b; };
    std::vector<Seg> segs;
    for (int s = 0; s < S; ++s) {
        bool in_seg = false;
        int seg_start = 0;
        for (int t = 0; t < T; ++t) {
            const bool active = probs[(size_t)t * S + s] > threshold;
            if (active && !in_seg) { seg_start = t; in_seg = true; }
            else if (!active && in_seg) { segs.push_back({s, frame_to_seconds(seg_start), frame_to_seconds(t - 1)}); in_seg = false; }
        }
        if (in_seg) segs.push_back({s, frame_to_seconds(seg_start), frame_to_seconds(T - 1)});
    }
    std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.a < y.a; });   // by start time (:106-110)
    for (int i = 0; i < (int)segs.size() && i < cap; ++i) {
        if (speaker) speaker[i] = segs[i].spk;
        if (start) start[i] = segs[i].a;
        if (end) end[i] = segs[i].b;
    }
    return (int)segs.size();
}

}  // extern "C"
