// parakeet.cpp_amd/csrc/wav.cpp -- host-side audio ingestion: a RIFF/WAVE reader (PCM16 / PCM24 / PCM32 / IEEE float32), the
// mono downmix and the Kaiser-windowed sinc resampler of the reference's read_audio (src/audio_io.cpp:269-293,453-483: dr_wav ->
// float32 in [-1,1] -> downmix_to_mono :198-214 -> sinc_resample :123-195).  int16 -> float is /32768 as in the reference
// (tests/test_all.cpp:483-721 "int16 -> f32 /32768").  Pinned against the reference's own object code
// (oracle/_ref/libpk_ref_audio.so, tests/test_audio_vs_reference.py).  FLAC / MP3 / OGG containers are out of scope.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace pk {

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

thread_local size_t info_frames_ = 0;      // frame count of the last info-only parse
size_t wav_info_frames() { return info_frames_; }
void parse_wav(const uint8_t *bytes, size_t n_bytes, const char *what, std::vector<float> &mono, int &sample_rate, int *n_channels, bool info_only, size_t file_len = 0);

static void slurp(const std::string &path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) fail(PK_ERR_IO, "Failed to open audio file: %s", path.c_str());
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(len > 0 ? (size_t)len : 0);
    if (len <= 0 || fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); fail(PK_ERR_IO, "Failed to read audio file: %s", path.c_str()); }
    fclose(f);
}

void read_wav(const std::string &path, std::vector<float> &mono, int &sample_rate, int *n_channels) {
    std::vector<uint8_t> buf;
    slurp(path, buf);
    parse_wav(buf.data(), buf.size(), path.c_str(), mono, sample_rate, n_channels, false);
}

// read_audio(const uint8_t*, size_t) (src/audio_io.cpp:485-493): an encoded file image in memory.  info_only: header walk without
// decoding (get_audio_duration, :527-586) -- mono is resized to the frame count but not filled.
void parse_wav(const uint8_t *bytes, size_t n_bytes, const char *what, std::vector<float> &mono, int &sample_rate, int *n_channels, bool info_only, size_t file_len) {
    struct View { const uint8_t *p; size_t n; size_t size() const { return n; } const uint8_t *data() const { return p; } const uint8_t &operator[](size_t i) const { return p[i]; } } buf{bytes, n_bytes};
    const std::string path = what ? what : "<memory>";
    if (buf.size() < 12 || memcmp(buf.data(), "RIFF", 4) || memcmp(buf.data() + 8, "WAVE", 4))
        fail(PK_ERR_IO, "Unsupported audio format (only RIFF/WAVE is read natively): %s", path.c_str());
    int fmt = 0, channels = 0, bits = 0;
    sample_rate = 0;
    const uint8_t *data = nullptr;
    size_t data_len = 0;
    for (size_t pos = 12; pos + 8 <= buf.size();) {
        const uint32_t sz = rd32(&buf[pos + 4]);
        const uint8_t *body = &buf[pos + 8];
        if (pos + 8 + sz > buf.size() && memcmp(&buf[pos], "data", 4) != 0) break;
        if (!memcmp(&buf[pos], "fmt ", 4) && sz >= 16) {
            fmt = rd16(body); channels = rd16(body + 2); sample_rate = (int)rd32(body + 4); bits = rd16(body + 14);
            if (fmt == 0xFFFE && sz >= 26) fmt = rd16(body + 24);       // WAVE_FORMAT_EXTENSIBLE sub-format
        } else if (!memcmp(&buf[pos], "data", 4)) {
            data = body;
            const size_t flen = file_len > buf.size() ? file_len : buf.size();   // info mode: only the head of the file is in memory
            data_len = (pos + 8 + sz <= flen) ? sz : flen - pos - 8;
            break;
        }
        pos += 8 + sz + (sz & 1);
    }
    if (!data || channels <= 0 || !sample_rate) fail(PK_ERR_IO, "Failed to decode WAV file: %s", path.c_str());
    const int bps = bits / 8;
    if (!((fmt == 1 && (bits == 16 || bits == 24 || bits == 32)) || (fmt == 3 && bits == 32)))
        fail(PK_ERR_IO, "Unsupported WAV encoding (format %d, %d bits): %s", fmt, bits, path.c_str());
    const size_t frames = data_len / ((size_t)bps * channels);
    if (n_channels) *n_channels = channels;
    if (info_only) { mono.clear(); info_frames_ = frames; return; }
    mono.resize(frames);
    for (size_t i = 0; i < frames; ++i) {
        float acc = 0.0f;
        for (int c = 0; c < channels; ++c) {
            const uint8_t *p = data + (i * channels + c) * bps;
            float v;
            if (fmt == 3) { memcpy(&v, p, 4); }
            else if (bits == 16) v = (float)(int16_t)rd16(p) / 32768.0f;
            else if (bits == 24) v = (float)((int32_t)((p[0] << 8) | (p[1] << 16) | ((uint32_t)p[2] << 24)) >> 8) / 8388608.0f;
            else v = (float)((double)(int32_t)rd32(p) / 2147483648.0);
            acc += v;
        }
        mono[i] = channels == 1 ? acc : acc * (1.0f / (float)channels);       // downmix_to_mono: sum * inv_ch (:205-212)
    }
}

// ---- sinc_resample (src/audio_io.cpp:100-195): Kaiser (beta 7.857, 16-tap half width) windowed sinc, fp64, normalised by the
// sum of the weights; when downsampling the cutoff drops to dst/src and the window widens by src/dst.
static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 30; ++k) {
        term *= (x * x) / (4.0 * k * k);
        sum += term;
        if (term < 1e-12 * sum) break;
    }
    return sum;
}
static double kaiser_window(double n, double N, double beta) {
    const double arg = 2.0 * n / N - 1.0;
    double val = 1.0 - arg * arg;
    if (val < 0.0) val = 0.0;
    return bessel_i0(beta * std::sqrt(val)) / bessel_i0(beta);
}
static int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

void sinc_resample(const float *input, size_t input_len, int src_rate, int dst_rate, std::vector<float> &output) {
    if (src_rate == dst_rate) { output.assign(input, input + input_len); return; }
    const int g = gcd_int(src_rate, dst_rate), up = dst_rate / g, down = src_rate / g;
    const size_t output_len = (size_t)(((int64_t)input_len * up + down - 1) / down);
    output.resize(output_len);
    constexpr int HALF_WIDTH = 16;
    constexpr double BETA = 7.857;
    const double ratio = (double)src_rate / dst_rate;
    const double cutoff = std::min(1.0, 1.0 / std::max(ratio, 1.0));
    const double filter_scale = cutoff;
    const double sample_ratio = (double)dst_rate / src_rate;
    const double width_factor = std::max(1.0, ratio);
    for (size_t i = 0; i < output_len; ++i) {
        const double src_pos = (double)i / sample_ratio;
        const int center = (int)std::floor(src_pos);
        double sum = 0.0, weight_sum = 0.0;
        for (int j = center - HALF_WIDTH + 1; j <= center + HALF_WIDTH; ++j) {
            if (j < 0 || j >= (int)input_len) continue;
            const double dist = src_pos - j;
            const double window_pos = dist / width_factor;
            if (std::abs(window_pos) > HALF_WIDTH) continue;
            const double w = kaiser_window(window_pos + HALF_WIDTH, 2.0 * HALF_WIDTH, BETA);
            const double x = dist * cutoff * M_PI;
            const double sinc_val = std::abs(x) < 1e-10 ? 1.0 : std::sin(x) / x;
            const double weight = sinc_val * w * filter_scale;
            sum += input[j] * weight;
            weight_sum += weight;
        }
        output[i] = (weight_sum > 1e-10) ? (float)(sum / weight_sum) : 0.0f;
    }
}

}  // namespace pk
