// parakeet.cpp_amd/csrc/engine.cpp -- model lifetime, weight upload / derived tables, stage drivers.
#include "engine.hpp"

#include <cmath>
#include <cstring>

namespace pk {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error() { return g_last_error; }

Model::Model(const std::string &weights_path, const std::string &vocab_path, const pk_config &c) : cfg(c) {
    if (cfg.hidden_size <= 0 || cfg.num_heads <= 0 || cfg.hidden_size % cfg.num_heads) fail(PK_ERR_INVALID, "bad hidden_size / num_heads");
    if (cfg.hidden_size % 32 || cfg.ffn_intermediate % 32 || cfg.subsampling_channels % 32 || cfg.pred_hidden % 32 || cfg.joint_hidden % 32)
        fail(PK_ERR_UNSUPPORTED, "hidden / ffn / channel sizes must be multiples of 32 (MFMA K tile)");
    if (cfg.mel_bins > 128 || cfg.mel_bins % 8) fail(PK_ERR_UNSUPPORTED, "mel_bins must be a multiple of 8 and <= 128");
    st_ = std::make_unique<SafeTensors>(weights_path);
    if (!vocab_path.empty()) tok.load(vocab_path);
}

Model::~Model() {
    if (device_ >= 0) {
        (void)hipSetDevice(device_);
        for (void *p : allocs_) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
    }
}

void Model::require_gpu() const {
    if (device_ < 0) fail(PK_ERR_NO_DEVICE, "model is not on a GPU: call pk_model_to_gpu() / Transcriber::to_gpu() first (there is no CPU path)");
    PK_HIP(hipSetDevice(device_));
}

float *Model::dev_alloc(size_t n_floats) {
    void *p = nullptr;
    PK_HIP(hipMalloc(&p, (n_floats ? n_floats : 1) * sizeof(float)));
    allocs_.push_back(p);
    return static_cast<float *>(p);
}

const float *Model::upload(const float *host, size_t n) {
    float *d = dev_alloc(n);
    PK_HIP(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

const float *Model::upload_tensor(const std::string &name, std::vector<int64_t> expect) {
    const HostTensor *t = st_->find(name);
    if (!t) fail(PK_ERR_WEIGHTS, "missing tensor '%s'", name.c_str());
    if (t->dtype != "F32") fail(PK_ERR_WEIGHTS, "tensor '%s' has dtype %s, expected F32", name.c_str(), t->dtype.c_str());
    int64_t want = 1;
    for (auto e : expect) want *= e;
    if (t->numel() != want) {
        std::string got;
        for (auto s : t->shape) got += std::to_string(s) + " ";
        fail(PK_ERR_WEIGHTS, "tensor '%s' has shape [ %s], expected %lld elements", name.c_str(), got.c_str(), (long long)want);
    }
    return upload(t->f32(), (size_t)want);
}

// Slaney filterbank, fp64 build / fp32 store -- reference src/audio.cpp:24-94 (hz_to_mel_slaney, mel_to_hz_slaney,
// build_mel_filterbank); Hann window (periodic=false) :117; FFT twiddles per the FFT-512 specification in DESIGN.md.
void Model::build_mel_tables() {
    const int n_fft = 512, win = 400, n_freqs = 257, n_mels = cfg.mel_bins;
    const double sr = 16000.0, f_min = 0.0, f_max = sr / 2.0;
    auto hz2mel = [](double f) { return f < 1000.0 ? f / (200.0 / 3.0) : 15.0 + std::log(f / 1000.0) / 0.06875177742094912; };
    auto mel2hz = [](double m) { return m < 15.0 ? m * (200.0 / 3.0) : 1000.0 * std::exp((m - 15.0) * 0.06875177742094912); };
    std::vector<double> hz(n_mels + 2);
    const double m0 = hz2mel(f_min), m1 = hz2mel(f_max);
    for (int i = 0; i < n_mels + 2; ++i) hz[i] = mel2hz(m0 + (double)i * (m1 - m0) / (double)(n_mels + 1));
    std::vector<float> fb((size_t)n_freqs * n_mels, 0.0f);
    std::vector<int> lo(n_mels, 1), hi(n_mels, 0);
    for (int m = 0; m < n_mels; ++m) {
        const double left = hz[m], center = hz[m + 1], right = hz[m + 2];
        const double enorm = 2.0 / (right - left);
        bool any = false;
        for (int f = 0; f < n_freqs; ++f) {
            const double freq = (double)f * (double)(float)sr / (2.0 * (double)(n_freqs - 1));
            double v = 0.0;
            if (freq >= left && freq <= center && center > left) v = (freq - left) / (center - left);
            else if (freq > center && freq <= right && right > center) v = (right - freq) / (right - center);
            const float w = (float)(v * enorm);
            fb[(size_t)f * n_mels + m] = w;
            if (w != 0.0f) { if (!any) lo[m] = f; hi[m] = f; any = true; }
        }
    }
    std::vector<float> window(n_fft, 0.0f), twr(n_fft / 2), twi(n_fft / 2);
    const int off = (n_fft - win) / 2;   // switch A1 default: window centred in the FFT frame (torch.stft)
    for (int k = 0; k < win; ++k) window[off + k] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)k / (double)(win - 1)));
    for (int k = 0; k < n_fft / 2; ++k) {
        const double a = 2.0 * M_PI * (double)k / (double)n_fft;
        twr[k] = (float)std::cos(a);
        twi[k] = (float)(-std::sin(a));
    }
    mel.window = upload(window.data(), window.size());
    mel.tw_re = upload(twr.data(), twr.size());
    mel.tw_im = upload(twi.data(), twi.size());
    mel.fb = upload(fb.data(), fb.size());
    mel.f_lo = reinterpret_cast<const int *>(upload(reinterpret_cast<const float *>(lo.data()), lo.size()));
    mel.f_hi = reinterpret_cast<const int *>(upload(reinterpret_cast<const float *>(hi.data()), hi.size()));
    mel.n_mels = n_mels;
    mel.power_via_abs = 1;               // switch A2 default: abs() then square, as the reference writes it
}

void Model::to_gpu(int device) {
    if (device_ == device) return;
    if (device_ >= 0) fail(PK_ERR_INVALID, "model already lives on device %d", device_);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(PK_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU path)");
    if (device < 0 || device >= n) fail(PK_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, n);
    PK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PK_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fail(PK_ERR_NO_DEVICE, "device %d is %s; this library contains gfx950 (MI355X) code only", device, prop.gcnArchName);
    PK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    device_ = device;
    build_mel_tables();
    upload_weights();
}

void Model::run_mel(const float *d_pcm, int B, int64_t n_samples, float *d_logmel, float *d_feats, hipStream_t s) {
    const int n_frames = (int)(1 + n_samples / 160);
    launch_mel_logmel(d_pcm, B, n_samples, n_frames, mel, d_logmel, s);
    launch_mel_normalize(d_logmel, B, cfg.mel_bins, n_frames, 1, d_feats, s);
}

}  // namespace pk

namespace pk {
void Model::upload_weights() {}  // TEMP-STUB (replaced when the encoder lands)
}
