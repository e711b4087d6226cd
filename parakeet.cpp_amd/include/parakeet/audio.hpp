// parakeet/audio.hpp -- preprocess_audio of the drop-in facade (reference: include/parakeet/audio.hpp:7-30, src/audio.cpp:100-158) on
// the MI355X mel kernels.  The waveform is a std::vector<float> (or pointer + length), the result a Features record holding the
// [n_frames][n_mels] matrix the reference returns as a (1, n_frames, n_mels) tensor.  dither is accepted and unused, as in the reference.
#pragma once

#include <map>
#include <memory>
#include <utility>
#include <vector>

#include "audio_io.hpp"

namespace parakeet {

struct AudioConfig {                     // audio.hpp:7-17
    int sample_rate = 16000;
    int n_fft = 512;
    int win_length = 400;
    int hop_length = 160;
    int n_mels = 80;
    float dither = 1e-5f;
    float f_min = 0.0f;
    float f_max = -1.0f;
    bool normalize = true;
};

struct Features {
    std::vector<float> data;             // [n_frames][n_mels], row-major
    int n_frames = 0;
    int n_mels = 0;
    const float *ptr() const { return data.data(); }
};

namespace detail {
struct FrontendDeleter { void operator()(pk_frontend *f) const { pk_frontend_free(f); } };
inline pk_frontend *frontend_for(int n_mels, bool normalize) {
    static std::map<std::pair<int, bool>, std::unique_ptr<pk_frontend, FrontendDeleter>> cache;   // one device context per (n_mels, normalize)
    auto &slot = cache[{n_mels, normalize}];
    if (!slot) {
        pk_frontend *f = nullptr;
        check(pk_frontend_create(n_mels, normalize ? 1 : 0, 0, 0, &f));
        slot.reset(f);
    }
    return slot.get();
}
}  // namespace detail

/// NeMo-compatible preprocessing: pre-emphasis -> STFT -> mel -> log -> (per-feature normalisation) -> [n_frames][n_mels]
inline Features preprocess_audio(const float *waveform, size_t n, const AudioConfig &config = {}) {
    if (config.sample_rate != 16000 || config.n_fft != 512 || config.win_length != 400 || config.hop_length != 160)
        throw std::runtime_error("preprocess_audio: this build implements the 16 kHz / n_fft 512 / win 400 / hop 160 front end of the shipped models");
    Features out;
    out.n_mels = config.n_mels;
    out.data.resize((size_t)pk_mel_num_frames((int64_t)n) * config.n_mels);
    detail::check(pk_frontend_features(detail::frontend_for(config.n_mels, config.normalize), waveform, (int64_t)n, out.data.data(), &out.n_frames));
    return out;
}
inline Features preprocess_audio(const std::vector<float> &waveform, const AudioConfig &config = {}) {
    return preprocess_audio(waveform.data(), waveform.size(), config);
}
/// Overload accepting AudioData; validates the sample rate (src/audio.cpp:160-167)
inline Features preprocess_audio(const AudioData &audio, const AudioConfig &config = {}) {
    if (audio.sample_rate != config.sample_rate)
        throw std::runtime_error("preprocess_audio: audio sample rate " + std::to_string(audio.sample_rate) + " does not match config " + std::to_string(config.sample_rate));
    return preprocess_audio(audio.samples, config);
}

}  // namespace parakeet
