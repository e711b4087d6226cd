// parakeet/audio_io.hpp -- audio loading of the drop-in facade (reference: include/parakeet/audio_io.hpp:10-49, src/audio_io.cpp).
// Same names and argument meaning; AudioData::samples is a std::vector<float> instead of an axiom::Tensor.  Containers: RIFF/WAVE
// (PCM 16 / 24 / 32, IEEE float32, any channel count and rate); FLAC / MP3 / OGG -- which the reference hands to dr_libs / stb_vorbis --
// raise std::runtime_error.  Downmix and the Kaiser-windowed sinc resampler are the reference's (pinned against its object code).
#pragma once

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "transcribe.hpp"

namespace parakeet {

enum class AudioFormat { Unknown, WAV, FLAC, MP3, OGG };

struct AudioData {
    std::vector<float> samples;   // float32 mono, [-1, 1]
    int sample_rate = 16000;      // always the target rate after read_audio
    int original_sample_rate = 0;
    int num_channels = 1;
    int num_samples = 0;
    float duration = 0.0f;        // seconds
    AudioFormat format = AudioFormat::Unknown;
};

inline AudioFormat detect_format_by_extension(const std::string &path) {          // src/audio_io.cpp:37-52
    const auto dot = path.rfind('.');
    if (dot == std::string::npos) return AudioFormat::Unknown;
    std::string ext = path.substr(dot + 1);
    std::transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    if (ext == "wav" || ext == "wave") return AudioFormat::WAV;
    if (ext == "flac") return AudioFormat::FLAC;
    if (ext == "mp3") return AudioFormat::MP3;
    if (ext == "ogg" || ext == "oga") return AudioFormat::OGG;
    return AudioFormat::Unknown;
}

inline AudioFormat detect_format_by_magic(const uint8_t *data, size_t len) {      // src/audio_io.cpp:54-94
    if (!data || len < 4) return AudioFormat::Unknown;
    if (len >= 12 && !std::memcmp(data, "RIFF", 4) && !std::memcmp(data + 8, "WAVE", 4)) return AudioFormat::WAV;
    if (!std::memcmp(data, "fLaC", 4)) return AudioFormat::FLAC;
    if (!std::memcmp(data, "OggS", 4)) return AudioFormat::OGG;
    if (!std::memcmp(data, "ID3", 3)) return AudioFormat::MP3;
    if (data[0] == 0xFF && (data[1] & 0xE0) == 0xE0) return AudioFormat::MP3;     // MPEG frame sync
    return AudioFormat::Unknown;
}

namespace detail {
inline AudioData finish(float *pcm, int64_t n, int target, int original, int channels, AudioFormat f) {
    AudioData a;
    a.samples.assign(pcm, pcm + n);
    pk_free(pcm);
    a.sample_rate = target; a.original_sample_rate = original; a.num_channels = channels;
    a.num_samples = (int)n; a.duration = (float)n / (float)target; a.format = f;
    return a;
}
}  // namespace detail

/// Memory buffer: encoded bytes (format detected by magic)
inline AudioData read_audio(const uint8_t *data, size_t len, int target_sample_rate = 16000) {
    float *pcm = nullptr;
    int64_t n = 0;
    int sr = 0, ch = 0;
    detail::check(pk_read_audio_memory(data, len, target_sample_rate, &pcm, &n, &sr, &ch));
    return detail::finish(pcm, n, target_sample_rate, sr, ch, detect_format_by_magic(data, len));
}

/// File loading (resamples to target_sample_rate)
inline AudioData read_audio(const std::string &path, int target_sample_rate = 16000) {
    float *pcm = nullptr;
    int64_t n = 0;
    int sr = 0, ch = 0;
    detail::check(pk_read_audio(path.c_str(), target_sample_rate, &pcm, &n, &sr));
    (void)pk_audio_info(path.c_str(), nullptr, &ch, nullptr);
    return detail::finish(pcm, n, target_sample_rate, sr, ch > 0 ? ch : 1, AudioFormat::WAV);
}

/// Memory buffer: raw float32 PCM (mono)
inline AudioData read_audio(const float *pcm, size_t num_samples, int sample_rate, int target_sample_rate = 16000) {
    float *out = nullptr;
    int64_t n = 0;
    detail::check(pk_resample(pcm, (int64_t)num_samples, sample_rate, target_sample_rate, &out, &n));
    return detail::finish(out, n, target_sample_rate, sample_rate, 1, AudioFormat::Unknown);
}

/// Memory buffer: raw int16 PCM (mono), / 32768 as the reference (src/audio_io.cpp:509-523)
inline AudioData read_audio(const int16_t *pcm, size_t num_samples, int sample_rate, int target_sample_rate = 16000) {
    std::vector<float> f(num_samples);
    for (size_t i = 0; i < num_samples; ++i) f[i] = static_cast<float>(pcm[i]) / 32768.0f;
    return read_audio(f.data(), num_samples, sample_rate, target_sample_rate);
}

/// Duration query (header only, no decode)
inline float get_audio_duration(const std::string &path) {
    int sr = 0, ch = 0;
    int64_t frames = 0;
    detail::check(pk_audio_info(path.c_str(), &sr, &ch, &frames));
    return sr > 0 ? (float)((double)frames / sr) : 0.0f;
}

/// Public resampler
inline std::vector<float> resample(const std::vector<float> &samples, int src_rate, int dst_rate) {
    float *out = nullptr;
    int64_t n = 0;
    detail::check(pk_resample(samples.data(), (int64_t)samples.size(), src_rate, dst_rate, &out, &n));
    std::vector<float> r(out, out + n);
    pk_free(out);
    return r;
}

}  // namespace parakeet
