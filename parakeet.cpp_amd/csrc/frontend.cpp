// parakeet.cpp_amd/csrc/frontend.cpp -- preprocess_audio on its own (reference src/audio.cpp:100-158, include/parakeet/audio.hpp:7-30):
// the mel front end of engine.cpp without a model around it, for callers that feed features to Sortformer::diarize / forward
// themselves (README "Speaker Diarization": preprocess_audio(audio.samples, {.normalize = false})).  Same kernels, same bits.
#include <cstring>

#include "engine.hpp"

namespace pk {

class MelFrontend {
  public:
    MelFrontend(int n_mels, bool normalize, bool window_centered, int device) : n_mels_(n_mels), normalize_(normalize) {
        if (n_mels <= 0 || n_mels > 128 || n_mels % 8) fail(PK_ERR_UNSUPPORTED, "n_mels must be a multiple of 8 and <= 128");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(PK_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU path)");
        if (device < 0 || device >= n) fail(PK_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, n);
        PK_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        PK_HIP(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) fail(PK_ERR_NO_DEVICE, "device %d is %s; gfx950 (MI355X) code only", device, prop.gcnArchName);
        device_ = device;
        PK_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        tables_ = make_mel_tables(n_mels, window_centered, [this](const float *h, size_t cnt) {
            void *p = nullptr;
            PK_HIP(hipMalloc(&p, (cnt ? cnt : 1) * sizeof(float)));
            allocs_.push_back(p);
            PK_HIP(hipMemcpy(p, h, cnt * sizeof(float), hipMemcpyHostToDevice));
            return static_cast<const float *>(p);
        });
    }
    ~MelFrontend() {
        if (device_ >= 0) {
            (void)hipSetDevice(device_);
            for (void *p : allocs_) (void)hipFree(p);
            if (stream_) (void)hipStreamDestroy(stream_);
        }
    }
    int features(const float *pcm, int64_t n_samples, float *feats) {
        PK_HIP(hipSetDevice(device_));
        const int n_frames = (int)(1 + n_samples / 160);
        pcm_.reserve((size_t)n_samples * 4); logmel_.reserve((size_t)n_mels_ * mel_logmel_pitch(n_frames) * 4); feats_.reserve((size_t)n_mels_ * n_frames * 4);
        PK_HIP(hipMemcpyAsync(pcm_.p, pcm, (size_t)n_samples * 4, hipMemcpyHostToDevice, stream_));
        launch_mel_logmel(pcm_.as<float>(), 1, n_samples, n_frames, tables_, logmel_.as<float>(), stream_);
        launch_mel_normalize(logmel_.as<float>(), 1, n_mels_, n_frames, normalize_ ? 1 : 0, feats_.as<float>(), stream_);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(feats, feats_.p, (size_t)n_mels_ * n_frames * 4, hipMemcpyDeviceToHost, stream_));
        PK_HIP(hipStreamSynchronize(stream_));
        return n_frames;
    }
    int n_mels() const { return n_mels_; }

  private:
    int n_mels_, device_ = -1;
    bool normalize_;
    hipStream_t stream_ = nullptr;
    MelTables tables_{};
    std::vector<void *> allocs_;
    DevBuf pcm_, logmel_, feats_;
};

}  // namespace pk

using namespace pk;

struct pk_frontend {
    std::unique_ptr<MelFrontend> f;
};

extern "C" {

pk_status pk_frontend_create(int n_mels, int normalize, int stft_window_centered, int device, pk_frontend **out) {
    try {
        if (!out) fail(PK_ERR_INVALID, "invalid argument: out");
        auto h = std::make_unique<pk_frontend>();
        h->f = std::make_unique<MelFrontend>(n_mels, normalize != 0, stft_window_centered != 0, device);
        *out = h.release();
        return PK_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return PK_ERR_INVALID;
    }
}

pk_status pk_frontend_features(pk_frontend *f, const float *pcm, int64_t n_samples, float *feats, int *n_frames) {
    try {
        if (!f || !pcm || !feats || n_samples <= 256) fail(PK_ERR_INVALID, "invalid argument: frontend/pcm/feats/n_samples (> 256)");
        const int nf = f->f->features(pcm, n_samples, feats);
        if (n_frames) *n_frames = nf;
        return PK_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return PK_ERR_INVALID;
    }
}

void pk_frontend_free(pk_frontend *f) { delete f; }

}  // extern "C"
