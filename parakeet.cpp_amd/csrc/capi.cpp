// parakeet.cpp_amd/csrc/capi.cpp -- the extern "C" boundary declared in include/parakeet_amd.h.
// Every entry point translates pk::Error / std::exception into a status code + thread-local message.
#include <cstring>
#include <functional>

#include "engine.hpp"

namespace pk {
const std::string &last_error();
}

using namespace pk;

struct pk_model {
    std::unique_ptr<Model> m;
};

static pk_status guard(const std::function<void()> &fn) {
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

static void need(bool ok, const char *what) {
    if (!ok) fail(PK_ERR_INVALID, "invalid argument: %s", what);
}

extern "C" {

const char *pk_version(void) { return "parakeet.cpp_amd 0.1 (gfx950)"; }

size_t pk_last_error(char *buf, size_t cap) {
    const std::string &e = last_error();
    if (buf && cap) {
        const size_t n = e.size() < cap - 1 ? e.size() : cap - 1;
        memcpy(buf, e.data(), n);
        buf[n] = 0;
    }
    return e.size();
}

int pk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

pk_status pk_config_preset(const char *name, pk_config *out) {
    return guard([&] {
        need(name && out, "name/out");
        pk_config c;
        memset(&c, 0, sizeof c);
        c.mel_bins = 80; c.subsampling_channels = 256; c.num_heads = 8; c.conv_kernel_size = 9;
        c.pred_hidden = 640; c.joint_hidden = 640; c.max_symbols_per_step = 10;
        const std::string n = name;
        if (n == "tdt-ctc-110m") {              // make_110m_config, config.hpp:77-95
            c.hidden_size = 512; c.num_layers = 17; c.ffn_intermediate = 2048; c.vocab_size = 1025; c.num_lstm_layers = 1;
            c.num_durations = 5; c.ctc_vocab_size = 1025; c.blank_id = 1024;
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "tdt_joint_.");
        } else if (n == "tdt-600m") {           // make_tdt_600m_config, config.hpp:98-116
            c.mel_bins = 128; c.hidden_size = 1024; c.num_layers = 24; c.ffn_intermediate = 4096; c.vocab_size = 8193;
            c.num_lstm_layers = 2; c.num_durations = 5; c.ctc_vocab_size = 0; c.blank_id = 8192;
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "joint_.");
        } else if (n == "rnnt-600m") {          // make_rnnt_600m_config, config.hpp:119-135
            c.hidden_size = 1024; c.num_layers = 24; c.ffn_intermediate = 4096; c.vocab_size = 1025; c.num_lstm_layers = 2;
            c.num_durations = 0; c.ctc_vocab_size = 0; c.blank_id = 1024; c.rnnt_head = 1;
            snprintf(c.joint_prefix, sizeof c.joint_prefix, "joint_.");
        } else {
            fail(PK_ERR_INVALID, "unknown preset '%s'", name);
        }
        for (int i = 0; i < c.num_durations; ++i) c.durations[i] = i;
        *out = c;
    });
}

pk_status pk_model_load(const char *safetensors_path, const char *vocab_path, const pk_config *cfg, pk_model **out) {
    return guard([&] {
        need(safetensors_path && cfg && out, "path/cfg/out");
        auto h = std::make_unique<pk_model>();
        h->m = std::make_unique<Model>(safetensors_path, vocab_path ? vocab_path : "", *cfg);
        *out = h.release();
    });
}

pk_status pk_model_to_gpu(pk_model *m, int device) {
    return guard([&] { need(m, "model"); m->m->to_gpu(device); });
}

void pk_model_free(pk_model *m) { delete m; }

pk_status pk_model_config(const pk_model *m, pk_config *out) {
    return guard([&] { need(m && out, "model/out"); *out = m->m->cfg; });
}

int pk_mel_num_frames(int64_t n_samples) { return (int)(1 + n_samples / 160); }
int pk_encoder_num_frames(int n) {
    for (int i = 0; i < 3; ++i) n = (n - 1) / 2 + 1;
    return n;
}

pk_status pk_mel(pk_model *h, const float *pcm, int n_clips, int64_t n_samples, float *feats, float *logmel) {
    return guard([&] {
        need(h && pcm && feats && n_clips > 0, "model/pcm/feats/n_clips");
        need(n_samples > 256, "n_samples must exceed n_fft/2 (reflect padding)");
        Model &m = *h->m;
        m.require_gpu();
        const int nf = pk_mel_num_frames(n_samples), F = m.cfg.mel_bins;
        const size_t n_in = (size_t)n_clips * n_samples, n_lm = (size_t)n_clips * F * nf;
        m.io_in.reserve(n_in * 4);
        m.io_tmp.reserve(n_lm * 4);
        m.io_out.reserve(n_lm * 4);
        PK_HIP(hipMemcpyAsync(m.io_in.p, pcm, n_in * 4, hipMemcpyHostToDevice, m.stream));
        m.run_mel(m.io_in.as<float>(), n_clips, n_samples, m.io_tmp.as<float>(), m.io_out.as<float>(), m.stream);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpyAsync(feats, m.io_out.p, n_lm * 4, hipMemcpyDeviceToHost, m.stream));
        if (logmel) PK_HIP(hipMemcpyAsync(logmel, m.io_tmp.p, n_lm * 4, hipMemcpyDeviceToHost, m.stream));
        PK_HIP(hipStreamSynchronize(m.stream));
    });
}

/* ---- diagnostics ---------------------------------------------------------------------------------- */
namespace {
struct Scratch {   // device scratch for the model-less diagnostic entry points
    DevBuf a, b, c, d, e;
};
void diag_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(PK_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU path)");
}
}  // namespace

pk_status pk_diag_math(int fn, const float *in, float *out, int64_t n) {
    return guard([&] {
        need(in && out && n > 0, "in/out/n");
        diag_device();
        Scratch s;
        s.a.reserve(n * 4);
        s.b.reserve(n * 4);
        PK_HIP(hipMemcpy(s.a.p, in, n * 4, hipMemcpyHostToDevice));
        launch_math(fn, s.a.as<float>(), s.b.as<float>(), n, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.b.p, n * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_gemm(int M, int N, int K, const float *A, const float *W, const float *bias, int epi, const float *resid,
                       float alpha, float *out) {
    return guard([&] {
        need(A && W && out && M > 0 && N > 0 && K > 0, "A/W/out/M/N/K");
        need(K % 32 == 0, "K must be a multiple of 32");
        need(epi >= 0 && epi <= 4, "epi");
        need(epi != EPI_RESID || resid, "resid");
        diag_device();
        const int wrows = epi == EPI_GLU ? 2 * N : N;
        Scratch s;
        s.a.reserve((size_t)M * K * 4);
        s.b.reserve((size_t)wrows * K * 4);
        s.c.reserve((size_t)wrows * 4);
        s.d.reserve((size_t)M * N * 4);
        s.e.reserve((size_t)M * N * 4);
        PK_HIP(hipMemcpy(s.a.p, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.b.p, W, (size_t)wrows * K * 4, hipMemcpyHostToDevice));
        if (bias) PK_HIP(hipMemcpy(s.c.p, bias, (size_t)wrows * 4, hipMemcpyHostToDevice));
        if (resid) PK_HIP(hipMemcpy(s.d.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice));
        GemmArgs g{s.a.as<float>(), K, s.b.as<float>(), K, bias ? s.c.as<float>() : nullptr, s.e.as<float>(), N,
                   resid ? s.d.as<float>() : nullptr, N, alpha, M, N, K};
        launch_gemm(g, epi, nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.e.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_layernorm(const float *x, int64_t rows, int d, const float *gamma, const float *beta, float eps, float *y) {
    return guard([&] {
        need(x && gamma && beta && y && rows > 0 && d > 0 && d <= 1024, "x/gamma/beta/y/rows/d (d <= 1024)");
        diag_device();
        Scratch s;
        s.a.reserve((size_t)rows * d * 4);
        s.b.reserve((size_t)d * 4);
        s.c.reserve((size_t)d * 4);
        s.d.reserve((size_t)rows * d * 4);
        PK_HIP(hipMemcpy(s.a.p, x, (size_t)rows * d * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.b.p, gamma, (size_t)d * 4, hipMemcpyHostToDevice));
        PK_HIP(hipMemcpy(s.c.p, beta, (size_t)d * 4, hipMemcpyHostToDevice));
        launch_layernorm(s.a.as<float>(), rows, d, s.b.as<float>(), s.c.as<float>(), eps, s.d.as<float>(), nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(y, s.d.p, (size_t)rows * d * 4, hipMemcpyDeviceToHost));
    });
}

pk_status pk_diag_sum64(const float *x, int rows, int n, float *out) {
    return guard([&] {
        need(x && out && rows > 0 && n > 0, "x/out/rows/n");
        diag_device();
        Scratch s;
        s.a.reserve((size_t)rows * n * 4);
        s.b.reserve((size_t)rows * 4);
        PK_HIP(hipMemcpy(s.a.p, x, (size_t)rows * n * 4, hipMemcpyHostToDevice));
        launch_sum64_rows(s.a.as<float>(), rows, n, s.b.as<float>(), nullptr);
        PK_CHECK_LAUNCH();
        PK_HIP(hipMemcpy(out, s.b.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
