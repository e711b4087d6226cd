// oracle/ref_audio_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" wrappers around the REAL reference audio loader (/root/reference/src/audio_io.cpp: dr_wav decode, downmix_to_mono
// :198-214, sinc_resample :123-195, read_audio :453-523), compiled by oracle/Makefile against oracle/axiom_stub (the file uses
// axiom only to wrap its output vector).  Pins the product's pk_read_audio / pk_resample (csrc/wav.cpp).
#include <cstdlib>
#include <cstring>
#include <string>

#include "parakeet/audio_io.hpp"

using namespace parakeet;

static float *dup(const AudioData &a, long long *n) {
    *n = a.num_samples;
    float *p = static_cast<float *>(std::malloc((size_t)(a.num_samples > 0 ? a.num_samples : 1) * sizeof(float)));
    std::memcpy(p, a.samples.typed_data<float>(), (size_t)a.num_samples * sizeof(float));
    return p;
}

extern "C" {

// read_audio(const float *pcm, n, sample_rate, target) -- the raw-PCM entry (audio_io.cpp:506-514): resample only
float *ref_resample(const float *pcm, long long n, int src_rate, int dst_rate, long long *n_out) {
    try {
        return dup(read_audio(pcm, (size_t)n, src_rate, dst_rate), n_out);
    } catch (...) {
        *n_out = -1;
        return nullptr;
    }
}
// read_audio(path, target): decode (WAV / FLAC / MP3 / OGG by extension or magic), downmix, resample
float *ref_read_audio(const char *path, int target_rate, long long *n_out, int *orig_rate, int *channels) {
    try {
        AudioData a = read_audio(std::string(path), target_rate);
        *orig_rate = a.original_sample_rate;
        *channels = a.num_channels;
        return dup(a, n_out);
    } catch (...) {
        *n_out = -1;
        return nullptr;
    }
}
void ref_audio_free(float *p) { std::free(p); }

}
