// parakeet.cpp_amd/csrc/wav.cpp -- minimal RIFF/WAVE reader (PCM16 / PCM24 / PCM32 / IEEE float32), mono downmix.
// Stands in for the WAV branch of the reference's read_audio (src/audio_io.cpp:269-293,453-483: dr_wav ->
// float32 in [-1,1] -> downmix_to_mono :198-214).  int16 -> float is /32768 as in the reference
// (tests/test_all.cpp:483-721 "int16 -> f32 /32768").  Other containers / resampling are out of scope.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace pk {

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

void read_wav(const std::string &path, std::vector<float> &mono, int &sample_rate) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) fail(PK_ERR_IO, "Failed to open audio file: %s", path.c_str());
    std::vector<uint8_t> buf;
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(len > 0 ? (size_t)len : 0);
    if (len <= 0 || fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); fail(PK_ERR_IO, "Failed to read audio file: %s", path.c_str()); }
    fclose(f);
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
            data_len = (pos + 8 + sz <= buf.size()) ? sz : buf.size() - pos - 8;
            break;
        }
        pos += 8 + sz + (sz & 1);
    }
    if (!data || channels <= 0 || !sample_rate) fail(PK_ERR_IO, "Failed to decode WAV file: %s", path.c_str());
    const int bps = bits / 8;
    if (!((fmt == 1 && (bits == 16 || bits == 24 || bits == 32)) || (fmt == 3 && bits == 32)))
        fail(PK_ERR_IO, "Unsupported WAV encoding (format %d, %d bits): %s", fmt, bits, path.c_str());
    const size_t frames = data_len / ((size_t)bps * channels);
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
        mono[i] = channels == 1 ? acc : acc / (float)channels;
    }
}

}  // namespace pk
