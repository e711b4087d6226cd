// parakeet/sortformer.hpp -- Sortformer speaker diarization of the drop-in facade (reference: include/parakeet/sortformer.hpp:17-129,
// src/sortformer.cpp) on the MI355X engine's C ABI (pk_sortformer_*).
//
// Differences a caller can observe:
//  * the model is constructed from the weights FILE (the reference builds the module tree, then load_state_dict()s a map);
//  * `const Tensor &features` becomes (const float *features, int n_frames) -- [n_frames][mel_bins], the layout preprocess_audio
//    returns -- or raw PCM (the GPU runs the un-normalised mel front end itself, as run_sortformer does, src/main.cpp:513-517);
//  * forward() returns the probabilities as a flat [T][max_speakers] vector;
//  * diarize_chunk keeps the encoder cache inside the Sortformer object (one streaming session per object; reset_stream() starts a
//    new one) instead of taking an EncoderCache argument;
//  * there is no CPU path.
#pragma once

#include <algorithm>
#include <string>
#include <vector>

#include "audio.hpp"
#include "nemotron.hpp"

namespace parakeet {

struct DiarizationSegment {              // sortformer.hpp:19-23
    int speaker_id;
    float start;  // seconds
    float end;    // seconds
};

struct TransformerConfig {               // include/parakeet/transformer.hpp:13-22
    int hidden_size = 192;
    int num_layers = 18;
    int num_heads = 8;
    int ffn_intermediate = 768;
    float dropout = 0.1f;
    float layer_norm_eps = 1e-5f;
    bool pre_ln = true;
    bool has_final_norm = false;
};

struct SortformerConfig {                // sortformer.hpp:28-41
    StreamingEncoderConfig nest_encoder;
    int encoder_hidden = 512;
    int transformer_hidden = 192;
    TransformerConfig transformer;
    int max_speakers = 4;
    float activity_threshold = 0.5f;
};

inline SortformerConfig make_sortformer_117m_config() {   // sortformer.hpp:43-76
    SortformerConfig cfg;
    detail::set_encoder(cfg.nest_encoder, 128, 512, 17, 2048);
    cfg.nest_encoder.att_context_left = 70;
    cfg.nest_encoder.att_context_right = 0;
    cfg.nest_encoder.chunk_size = 20;
    cfg.encoder_hidden = 512;
    cfg.transformer_hidden = 192;
    cfg.transformer.hidden_size = 192;
    cfg.transformer.num_layers = 18;
    cfg.transformer.num_heads = 8;
    cfg.transformer.ffn_intermediate = 768;
    cfg.transformer.pre_ln = false;       // NeMo sortformer: post-norm
    cfg.transformer.has_final_norm = false;
    cfg.max_speakers = 4;
    cfg.activity_threshold = 0.5f;
    return cfg;
}

/// Arrival-Order Speaker Cache (sortformer.hpp:80-97, src/sortformer.cpp:11-38): speakers in the order their activity first exceeds 0.5.
class AOSCCache {
  public:
    explicit AOSCCache(int max_speakers = 4) : max_speakers_(max_speakers), speaker_active_(max_speakers, false) {}
    /// probs: [T][n_speakers] sigmoid probabilities
    void update(const float *probs, int T, int n_speakers) {
        for (int t = 0; t < T; ++t)
            for (int s = 0; s < n_speakers && s < max_speakers_; ++s)
                if (probs[(size_t)t * n_speakers + s] > 0.5f && !speaker_active_[s]) {
                    speaker_active_[s] = true;
                    arrival_order_.push_back(s);
                }
    }
    std::vector<int> speaker_order() const { return arrival_order_; }
    void reset() {
        std::fill(speaker_active_.begin(), speaker_active_.end(), false);
        arrival_order_.clear();
    }

  private:
    int max_speakers_;
    std::vector<bool> speaker_active_;
    std::vector<int> arrival_order_;
};

class Sortformer {
  public:
    explicit Sortformer(const std::string &weights_path, const SortformerConfig &config = make_sortformer_117m_config())
        : config_(config), weights_path_(weights_path) {}
    ~Sortformer() { pk_sortformer_free(h_); }
    Sortformer(const Sortformer &) = delete;
    Sortformer &operator=(const Sortformer &) = delete;

    void to_gpu(int device = 0) {
        if (h_) return;
        pk_sortformer_config c;
        pk_sortformer_config_preset(&c);
        const auto &e = config_.nest_encoder;
        c.nest.mel_bins = e.mel_bins; c.nest.subsampling_channels = e.subsampling_channels; c.nest.hidden_size = e.hidden_size;
        c.nest.num_layers = e.num_layers; c.nest.num_heads = e.num_heads; c.nest.ffn_intermediate = e.ffn_intermediate;
        c.nest.conv_kernel_size = e.conv_kernel_size;
        c.transformer.hidden_size = config_.transformer.hidden_size; c.transformer.num_layers = config_.transformer.num_layers;
        c.transformer.num_heads = config_.transformer.num_heads; c.transformer.ffn_intermediate = config_.transformer.ffn_intermediate;
        c.transformer.pre_ln = config_.transformer.pre_ln ? 1 : 0; c.transformer.has_final_norm = config_.transformer.has_final_norm ? 1 : 0;
        c.transformer.layer_norm_eps = config_.transformer.layer_norm_eps;
        c.max_speakers = config_.max_speakers; c.activity_threshold = config_.activity_threshold;
        c.att_context_left = e.att_context_left; c.att_context_right = e.att_context_right;
        detail::check(pk_sortformer_load(weights_path_.c_str(), &c, device, &h_));
    }

    /// Raw forward: features [n_frames][mel_bins] -> [T][max_speakers] sigmoid probabilities (sortformer.hpp:112)
    std::vector<float> forward(const float *features, int n_frames, int *T_out = nullptr) {
        to_gpu();
        std::vector<float> probs((size_t)pk_encoder_num_frames(n_frames) * config_.max_speakers);
        int T = 0;
        detail::check(pk_sortformer_forward(h_, features, 1, n_frames, probs.data(), &T));
        probs.resize((size_t)T * config_.max_speakers);
        if (T_out) *T_out = T;
        return probs;
    }
    /// Batch diarization from features (sortformer.hpp:104)
    std::vector<DiarizationSegment> diarize(const float *features, int n_frames) {
        int T = 0;
        const auto probs = forward(features, n_frames, &T);
        return probs_to_segments(probs.data(), T);
    }
    /// From preprocess_audio(samples, {.normalize = false}) -- the reference README's usage
    std::vector<DiarizationSegment> diarize(const Features &features) { return diarize(features.ptr(), features.n_frames); }
    /// From 16 kHz mono PCM: preprocess_audio(n_mels = mel_bins, normalize = false) + diarize (src/main.cpp:513-519)
    std::vector<DiarizationSegment> diarize_pcm(const float *pcm, size_t n) {
        to_gpu();
        const int T = pk_encoder_num_frames(pk_mel_num_frames((int64_t)n));
        std::vector<float> probs((size_t)T * config_.max_speakers);
        detail::check(pk_sortformer_forward_pcm(h_, pcm, 1, (int64_t)n, probs.data(), nullptr));
        return probs_to_segments(probs.data(), T);
    }

    /// Streaming: process a chunk of features [n_frames][mel_bins], update the arrival-order cache, return this chunk's segments
    /// (times relative to the chunk, as the reference's probs_to_segments(p) on the chunk; src/sortformer.cpp:123-150)
    std::vector<DiarizationSegment> diarize_chunk(const float *features, int n_frames, AOSCCache &aosc_cache) {
        to_gpu();
        const int cap = n_frames / 8 + 4;
        std::vector<float> probs((size_t)cap * config_.max_speakers);
        int T = 0;
        detail::check(pk_sortformer_diarize_chunk(h_, features, n_frames, probs.data(), cap, &T));
        if (T <= 0) return {};
        aosc_cache.update(probs.data(), T, config_.max_speakers);
        return probs_to_segments(probs.data(), T);
    }
    void reset_stream() { if (h_) detail::check(pk_sortformer_stream_reset(h_)); }

    /// probs [T][max_speakers] -> segments sorted by start (src/sortformer.cpp:71-113)
    std::vector<DiarizationSegment> probs_to_segments(const float *probs, int T) const {
        const int S = config_.max_speakers, cap = S * (T / 2 + 2);
        std::vector<int32_t> spk(cap);
        std::vector<float> a(cap), b(cap);
        const int n = pk_sortformer_segments(probs, T, S, config_.activity_threshold, spk.data(), a.data(), b.data(), cap);
        std::vector<DiarizationSegment> out;
        for (int i = 0; i < n && i < cap; ++i) out.push_back({spk[i], a[i], b[i]});
        return out;
    }

    const SortformerConfig &config() const { return config_; }

  private:
    SortformerConfig config_;
    std::string weights_path_;
    pk_sortformer *h_ = nullptr;
};

}  // namespace parakeet
