// parakeet/nemotron.hpp -- streaming drop-in: parakeet::NemotronTranscriber (reference include/parakeet/nemotron.hpp:54-133,
// src/nemotron.cpp:14-66) on the MI355X engine's streaming C ABI (pk_stream_*, include/parakeet_amd.h).
//
//   parakeet::NemotronTranscriber t("nemotron.safetensors", "vocab.txt", parakeet::make_nemotron_600m_config(/*latency_frames=*/1));
//   t.to_gpu();
//   while (have_audio) text += t.transcribe_chunk(pcm, n);      // new text of this chunk ("" while audio is being buffered)
//
// One object = one stream (the C ABI advances n lock-step streams per pk_stream for throughput: pk_stream_create(model, 16, ...)).
// Differences a caller can observe: the axiom::Tensor overload of transcribe_chunk becomes (const float*, size_t), which the
// reference also has (nemotron.hpp:93-96); there is no CPU path (the first chunk places the model on GPU 0 if to_gpu() was not
// called); weights load strictly.  The decode quirk of the reference is kept: transcribe_chunk decodes with the DEFAULT blank id
// of rnnt_streaming_decode_chunk, 1024 (nemotron.cpp:40-42, eou.hpp:91-94), whatever the vocabulary size.
#pragma once

#include <functional>
#include <string>
#include <vector>

#include "transcribe.hpp"

namespace parakeet {

struct StreamingEncoderConfig : EncoderConfig {   // include/parakeet/streaming_encoder.hpp:17-23
    int att_context_left = 70;
    int att_context_right = 0;
    int chunk_size = 20;
};

struct NemotronConfig {                            // include/parakeet/nemotron.hpp:20-29
    StreamingEncoderConfig encoder;
    PredictionConfig prediction;
    JointConfig joint;
    std::vector<int> durations = {0, 1, 2, 3, 4};
    int latency_frames = 0;
};

inline NemotronConfig make_nemotron_600m_config(int latency_frames = 0) {   // nemotron.hpp:31-52
    NemotronConfig cfg;
    detail::set_encoder(cfg.encoder, 80, 1024, 24, 4096);
    detail::set_decoder(cfg.prediction, cfg.joint, 1024, 8193, 2);
    cfg.encoder.att_context_left = 70;
    cfg.encoder.att_context_right = latency_frames;
    cfg.encoder.chunk_size = 20;
    cfg.latency_frames = latency_frames;
    return cfg;
}

using PartialResultCallback = std::function<void(const std::string &partial)>;

struct EOUConfig {                                 // include/parakeet/eou.hpp:25-32
    StreamingEncoderConfig encoder;
    PredictionConfig prediction;
    JointConfig joint;
    std::vector<int> durations = {0, 1, 2, 3, 4};
    int eou_token_id = -1;
    int ctc_vocab_size = 1025;
};

inline EOUConfig make_eou_120m_config() {          // eou.hpp:34-56
    EOUConfig cfg;
    detail::set_encoder(cfg.encoder, 80, 512, 17, 2048);
    detail::set_decoder(cfg.prediction, cfg.joint, 512, 1025, 1);
    cfg.encoder.att_context_left = 70;
    cfg.encoder.att_context_right = 1;
    cfg.encoder.chunk_size = 20;
    cfg.eou_token_id = 1024;
    return cfg;
}

namespace detail {

/// One streaming session on pk_stream_* : the body shared by NemotronTranscriber and StreamingTranscriber (both are
/// process_chunk -> forward_chunk -> rnnt_streaming_decode_chunk with the DEFAULT blank id 1024: nemotron.cpp:24-52, eou.cpp:113-146).
class StreamSession {
  public:
    StreamSession(const std::string &weights_path, const std::string &vocab_path, const StreamingEncoderConfig &enc, const PredictionConfig &pred,
                  const JointConfig &joint, const std::vector<int> &durations)
        : left_(enc.att_context_left), right_(enc.att_context_right),
          eng_(weights_path, vocab_path, flatten(enc, pred, joint, durations, 0, "joint_.", false, /*blank_id=*/1024)) {}
    ~StreamSession() { pk_stream_free(stream_); }
    StreamSession(const StreamSession &) = delete;
    StreamSession &operator=(const StreamSession &) = delete;

    void to_gpu() { eng_.to_gpu(0); on_gpu_ = true; }

    /// Process a chunk of raw float32 PCM (16 kHz mono) -> the new text of this chunk.
    std::string transcribe_chunk(const float *data, size_t num_samples) {
        if (num_samples == 0) return "";
        ensure_stream();
        const int mt = 256;
        std::vector<int32_t> ids(mt), start(mt), end(mt);
        std::vector<float> conf(mt);
        int32_t len = 0;
        detail::check(pk_stream_push(stream_, data, (int)num_samples, mt, ids.data(), &len, start.data(), end.data(), conf.data()));
        if (len <= 0) return "";
        std::vector<int> fresh(ids.begin(), ids.begin() + len);
        tokens_.insert(tokens_.end(), fresh.begin(), fresh.end());
        for (int i = 0; i < len; ++i) timestamped_.push_back({ids[i], start[i], end[i], conf[i]});
        if (!eng_.tokenizer().loaded()) return "";
        const std::string text = eng_.tokenizer().decode(fresh);
        if (partial_callback_) partial_callback_(text);
        return text;
    }
    /// int16 PCM convenience overload (nemotron.hpp:99-105)
    std::string transcribe_chunk(const int16_t *data, size_t num_samples) {
        std::vector<float> f(num_samples);
        for (size_t i = 0; i < num_samples; ++i) f[i] = static_cast<float>(data[i]) / 32768.0f;
        return transcribe_chunk(f.data(), num_samples);
    }

    /// Reset for a new utterance (nemotron.cpp:54-58)
    void reset() {
        if (stream_) detail::check(pk_stream_reset(stream_));
        tokens_.clear();
        timestamped_.clear();
    }
    /// Full transcription so far (nemotron.cpp:60-65)
    std::string get_text() const { return (eng_.tokenizer().loaded() && !tokens_.empty()) ? eng_.tokenizer().decode(tokens_) : ""; }
    void set_partial_callback(PartialResultCallback cb) { partial_callback_ = std::move(cb); }
    const std::vector<TimestampedToken> &get_timestamped_tokens() const { return timestamped_; }
    const Tokenizer &tokenizer() const { return eng_.tokenizer(); }

  private:
    void ensure_stream() {
        if (!on_gpu_) to_gpu();
        if (!stream_) detail::check(pk_stream_create(eng_.handle(), 1, left_, right_, &stream_));
    }
    int left_, right_;
    detail::Engine eng_;
    pk_stream *stream_ = nullptr;
    bool on_gpu_ = false;
    std::vector<int> tokens_;
    std::vector<TimestampedToken> timestamped_;
    PartialResultCallback partial_callback_;
};

}  // namespace detail

/// parakeet::NemotronTranscriber (reference include/parakeet/nemotron.hpp:73-133)
class NemotronTranscriber : public detail::StreamSession {
  public:
    NemotronTranscriber(const std::string &weights_path, const std::string &vocab_path, const NemotronConfig &config = make_nemotron_600m_config())
        : detail::StreamSession(weights_path, vocab_path, config.encoder, config.prediction, config.joint, config.durations), config_(config) {}

  private:
    NemotronConfig config_;
};

/// parakeet::StreamingTranscriber (reference include/parakeet/eou.hpp:101-160): the EOU model's streaming session.  As in the
/// reference's transcribe_chunk (src/eou.cpp:113-146) the end-of-utterance token id of the config takes no part in decoding.
class StreamingTranscriber : public detail::StreamSession {
  public:
    StreamingTranscriber(const std::string &weights_path, const std::string &vocab_path, const EOUConfig &config = make_eou_120m_config())
        : detail::StreamSession(weights_path, vocab_path, config.encoder, config.prediction, config.joint, config.durations), config_(config) {}

  private:
    EOUConfig config_;
};

}  // namespace parakeet
