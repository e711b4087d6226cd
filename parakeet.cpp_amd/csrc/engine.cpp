// parakeet.cpp_amd/csrc/engine.cpp -- model lifetime, weight upload / derived tables, stage drivers.
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <cstring>

namespace pk {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error() { return g_last_error; }

ProfileSink::~ProfileSink() {
    for (auto &r : recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
}

Model::Model(const std::string &weights_path, const std::string &vocab_path, const pk_config &c) : cfg(c) {
    validate_config();
    st_ = std::make_unique<SafeTensors>(weights_path);
    if (!vocab_path.empty()) tok.load(vocab_path);
}

Model::Model(const void *weights, size_t n_bytes, const std::string &vocab_path, const pk_config &c, bool borrow) : cfg(c) {
    validate_config();
    st_ = std::make_unique<SafeTensors>(weights, n_bytes, borrow);
    if (!vocab_path.empty()) tok.load(vocab_path);
}

void Model::validate_config() {
    // every size that is later a divisor, a modulus or a vector length is checked for > 0 BEFORE any arithmetic on it
    if (cfg.mel_bins <= 0 || cfg.subsampling_channels <= 0 || cfg.hidden_size <= 0 || cfg.num_heads <= 0 || cfg.num_layers < 0 ||
        cfg.ffn_intermediate <= 0 || cfg.conv_kernel_size <= 0)
        fail(PK_ERR_INVALID, "encoder sizes must be positive (mel_bins, subsampling_channels, hidden_size, num_heads, ffn_intermediate, conv_kernel_size; num_layers >= 0)");
    if (cfg.vocab_size < 0 || cfg.ctc_vocab_size < 0) fail(PK_ERR_INVALID, "vocab_size / ctc_vocab_size must be >= 0");
    if (cfg.vocab_size > 0 && (cfg.pred_hidden <= 0 || cfg.joint_hidden <= 0 || cfg.max_symbols_per_step <= 0 || cfg.blank_id < 0 || cfg.blank_id >= cfg.vocab_size))
        fail(PK_ERR_INVALID, "decoder sizes must be positive (pred_hidden, joint_hidden, max_symbols_per_step) and 0 <= blank_id < vocab_size");
    if (cfg.num_layers > 4096) fail(PK_ERR_INVALID, "num_layers out of range");
    if (cfg.hidden_size % cfg.num_heads) fail(PK_ERR_INVALID, "bad hidden_size / num_heads");
    const int hd = cfg.hidden_size / cfg.num_heads;
    if (hd % 32) fail(PK_ERR_UNSUPPORTED, "head_dim must be a multiple of 32");
    if (cfg.hidden_size % 32 || cfg.ffn_intermediate % 32 || cfg.subsampling_channels % 32 || cfg.pred_hidden % 32 || cfg.joint_hidden % 32)
        fail(PK_ERR_UNSUPPORTED, "hidden / ffn / channel sizes must be multiples of 32 (MFMA K tile)");
    if (256 % cfg.subsampling_channels) fail(PK_ERR_UNSUPPORTED, "subsampling_channels must divide 256");
    if (cfg.mel_bins > 128 || cfg.mel_bins % 8) fail(PK_ERR_UNSUPPORTED, "mel_bins must be a multiple of 8 and <= 128");
    if (cfg.conv_kernel_size != 9 && cfg.conv_kernel_size != 31) fail(PK_ERR_UNSUPPORTED, "conv_kernel_size must be 9 or 31");
    if (cfg.vocab_size > 0 && (cfg.num_lstm_layers < 1 || cfg.num_lstm_layers > 4)) fail(PK_ERR_UNSUPPORTED, "num_lstm_layers must be 1..4");
    if (cfg.hidden_size > 1024) fail(PK_ERR_UNSUPPORTED, "hidden_size > 1024");
    if (cfg.num_lstm_layers * cfg.pred_hidden > 12 * 256) fail(PK_ERR_UNSUPPORTED, "num_lstm_layers * pred_hidden > 3072");
    if (cfg.pred_hidden % 64 || cfg.joint_hidden % 64) fail(PK_ERR_UNSUPPORTED, "pred_hidden / joint_hidden must be multiples of 64 (decode GEMV K chunk)");
    if (cfg.num_durations < 0 || cfg.num_durations > 8) fail(PK_ERR_INVALID, "num_durations must be 0..8");
}

Model::~Model() {
    if (device_ >= 0) {
        (void)hipSetDevice(device_);
        if (pipe && pipe_free) pipe_free(pipe);           // uses the streams below
        for (void *p : allocs_) (void)hipFree(p);
        if (h_done) (void)hipHostFree(h_done);
        if (stream) (void)hipStreamDestroy(stream);
        if (stream_dec) (void)hipStreamDestroy(stream_dec);
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

// Strict weight lookup: the tensor must exist, be F32, and have exactly the expected extents.  Extents of 1 are ignored on both
// sides (a 1x1 conv weight [out][in][1][1] is the [out][in] matrix of the GEMM, a depthwise [C][1][k] is [C][k]); anything else
// -- a transposed matrix with the same element count, a different rank, a negative extent -- is an error naming the tensor.
const HostTensor &Model::host_tensor(const std::string &name, const std::vector<int64_t> &expect) {
    const HostTensor *t = st_->find(name);
    if (!t) fail(PK_ERR_WEIGHTS, "missing tensor '%s'", name.c_str());
    if (t->dtype != "F32") fail(PK_ERR_WEIGHTS, "tensor '%s' has dtype %s, expected F32", name.c_str(), t->dtype.c_str());
    std::vector<int64_t> a, b;
    for (auto v : t->shape) if (v != 1) a.push_back(v);
    for (auto v : expect) if (v != 1) b.push_back(v);
    if (a != b) {
        std::string got, want;
        for (auto v : t->shape) got += std::to_string(v) + " ";
        for (auto v : expect) want += std::to_string(v) + " ";
        fail(PK_ERR_WEIGHTS, "tensor '%s' has shape [ %s], expected [ %s]", name.c_str(), got.c_str(), want.c_str());
    }
    return *t;
}

// fp32 -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does to the activations on the device)
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Weight of a Linear / 1x1 conv that runs on the MFMA GEMM: fp32 as is, or (pk_config.gemm_bf16) rounded to bf16 once here.
// The returned pointer is only ever passed to Model::gemm, which knows which of the two it holds.
const float *Model::upload_gemm_weight(const float *host, size_t n) {
    if (!cfg.gemm_bf16) return upload(host, n);
    std::vector<uint16_t> h(n + (n & 1));
    for (size_t i = 0; i < n; ++i) h[i] = bf16_rne(host[i]);
    return upload(reinterpret_cast<const float *>(h.data()), (n + 1) / 2);
}
const float *Model::upload_gemm_tensor(const std::string &name, std::vector<int64_t> expect) {
    int64_t want = 1;
    for (auto e : expect) want *= e;
    return upload_gemm_weight(host_tensor(name, expect).f32(), (size_t)want);
}

const float *Model::upload_tensor(const std::string &name, std::vector<int64_t> expect) {
    int64_t want = 1;
    for (auto e : expect) want *= e;
    return upload(host_tensor(name, expect).f32(), (size_t)want);
}

// Slaney filterbank, fp64 build / fp32 store -- reference src/audio.cpp:24-94 (hz_to_mel_slaney, mel_to_hz_slaney,
// build_mel_filterbank); Hann window (periodic=false) :117; FFT twiddles per the FFT-512 specification in DESIGN.md.
// Filterbank (src/audio.cpp:40-94), Hann window, FFT twiddles and the packed filter bands of the mel kernels; `upload` puts a host
// array on the device and keeps ownership.  Shared by Model and the weight-free MelFrontend (preprocess_audio on its own).
MelTables make_mel_tables(int n_mels, bool window_centered, const std::function<const float *(const float *, size_t)> &upload) {
    MelTables mel{};
    const int n_fft = 512, win = 400, n_freqs = 257;
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
    // switch A1 (pk_config.stft_window_centered): left-aligned like the reference author's feature check, or centred (torch.stft)
    const int off = window_centered ? (n_fft - win) / 2 : 0;
    for (int k = 0; k < win; ++k) window[off + k] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)k / (double)(win - 1)));
    for (int k = 0; k < n_fft / 2; ++k) {
        const double a = 2.0 * M_PI * (double)k / (double)n_fft;
        twr[k] = (float)std::cos(a);
        twi[k] = (float)(-std::sin(a));
    }
    mel.window = upload(window.data(), window.size());
    {   // the streaming preprocessor's frames are win_length samples, zero-padded on the right (src/audio.cpp:222-241)
        std::vector<float> wl(n_fft, 0.0f);
        for (int k = 0; k < win; ++k) wl[k] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)k / (double)(win - 1)));
        mel.window_left = upload(wl.data(), wl.size());
    }
    mel.tw_re = upload(twr.data(), twr.size());
    mel.tw_im = upload(twi.data(), twi.size());
    mel.fb = upload(fb.data(), fb.size());
    mel.f_lo = reinterpret_cast<const int *>(upload(reinterpret_cast<const float *>(lo.data()), lo.size()));
    mel.f_hi = reinterpret_cast<const int *>(upload(reinterpret_cast<const float *>(hi.data()), hi.size()));
    {   // packed bands for the LDS-staged dot products of the mel kernel
        std::vector<int> off(n_mels + 1, 0);
        std::vector<float> packed;
        for (int m = 0; m < n_mels; ++m) {
            for (int f = lo[m]; f <= hi[m]; ++f) packed.push_back(fb[(size_t)f * n_mels + m]);
            off[m + 1] = (int)packed.size();
        }
        if ((int)packed.size() > kMelMaxTaps) fail(PK_ERR_UNSUPPORTED, "mel filterbank has %d taps (> %d)", (int)packed.size(), kMelMaxTaps);
        if (packed.empty()) packed.push_back(0.0f);
        mel.fb_nnz = off[n_mels];
        mel.fbc = upload(packed.data(), packed.size());
        mel.fb_off = reinterpret_cast<const int *>(upload(reinterpret_cast<const float *>(off.data()), off.size()));
    }
    mel.n_mels = n_mels;
    mel.power_via_abs = 1;               // switch A2 default: abs() then square, as the reference writes it
    return mel;
}

void Model::build_mel_tables() {
    mel = make_mel_tables(cfg.mel_bins, cfg.stft_window_centered != 0, [this](const float *h, size_t n) { return upload(h, n); });
}

// Upload every tensor the hot path reads, under the reference's names (scripts/convert_nemo.py:98-310;
// SURVEY.md Appendix B), strictly shape-checked.  Layout transforms done here, once:
//   * depthwise taps [C][1][3][3] -> [9][C], [d][1][K] -> [K][d]  (coalesced along channels)
//   * q/k/v projections stacked into one [3d][d] matrix (one GEMM instead of three)
//   * BatchNorm running_var -> rstd = 1/sqrt(var + eps) (the same fp32 expression the oracle evaluates)
//   * label_proj_ and duration_proj_ stacked into one [V+D][J] matrix
//   * g1[token] = W_ih0 E[token] + b : the layer-0 LSTM input projection of every possible token
void Model::upload_weights() {
    const int d = cfg.hidden_size, C = cfg.subsampling_channels, F = cfg.mel_bins, ffn = cfg.ffn_intermediate, K = cfg.conv_kernel_size;
    const int H = cfg.num_heads, hd = d / H;
    int f3 = F;
    for (int i = 0; i < 3; ++i) f3 = (f3 - 1) / 2 + 1;
    auto taps_last = [&](const std::string &name, int ch, int taps, bool two_d) {   // [ch][1][3][3] / [ch][1][taps] -> [taps][ch]
        const HostTensor &t = host_tensor(name, two_d ? std::vector<int64_t>{ch, 3, 3} : std::vector<int64_t>{ch, taps});
        std::vector<float> w((size_t)ch * taps);
        for (int c = 0; c < ch; ++c)
            for (int k = 0; k < taps; ++k) w[(size_t)k * ch + c] = t.f32()[(size_t)c * taps + k];
        return upload(w.data(), w.size());
    };
    const std::string ep = cfg.encoder_prefix[0] ? std::string(cfg.encoder_prefix) : std::string("encoder_.");
    const std::string sp = ep + "subsampling_.";
    sub.c1w = taps_last(sp + "conv1_.weight", C, 9, true);  sub.c1b = upload_tensor(sp + "conv1_.bias", {C});
    sub.d1w = taps_last(sp + "dw1_.weight", C, 9, true);    sub.d1b = upload_tensor(sp + "dw1_.bias", {C});
    sub.c2w = upload_gemm_tensor(sp + "conv2_.weight", {C, C}); sub.c2b = upload_tensor(sp + "conv2_.bias", {C});
    sub.d2w = taps_last(sp + "dw2_.weight", C, 9, true);    sub.d2b = upload_tensor(sp + "dw2_.bias", {C});
    sub.c3w = upload_gemm_tensor(sp + "conv3_.weight", {C, C}); sub.c3b = upload_tensor(sp + "conv3_.bias", {C});
    sub.pw = upload_gemm_tensor(sp + "proj_.weight", {d, (int64_t)C * f3}); sub.pb = upload_tensor(sp + "proj_.bias", {d});

    layers.resize(cfg.num_layers);
    for (int i = 0; i < cfg.num_layers; ++i) {
        LayerW &L = layers[i];
        const std::string q = ep + "layers_." + std::to_string(i) + ".";
        L.ffn1_ng = upload_tensor(q + "ffn1_.norm_.weight", {d}); L.ffn1_nb = upload_tensor(q + "ffn1_.norm_.bias", {d});
        L.ffn1_w1 = upload_gemm_tensor(q + "ffn1_.fc1_.weight", {ffn, d}); L.ffn1_b1 = upload_tensor(q + "ffn1_.fc1_.bias", {ffn});
        L.ffn1_w2 = upload_gemm_tensor(q + "ffn1_.fc2_.weight", {d, ffn}); L.ffn1_b2 = upload_tensor(q + "ffn1_.fc2_.bias", {d});
        L.ffn2_ng = upload_tensor(q + "ffn2_.norm_.weight", {d}); L.ffn2_nb = upload_tensor(q + "ffn2_.norm_.bias", {d});
        L.ffn2_w1 = upload_gemm_tensor(q + "ffn2_.fc1_.weight", {ffn, d}); L.ffn2_b1 = upload_tensor(q + "ffn2_.fc1_.bias", {ffn});
        L.ffn2_w2 = upload_gemm_tensor(q + "ffn2_.fc2_.weight", {d, ffn}); L.ffn2_b2 = upload_tensor(q + "ffn2_.fc2_.bias", {d});
        L.att_ng = upload_tensor(q + "attn_.norm_.weight", {d}); L.att_nb = upload_tensor(q + "attn_.norm_.bias", {d});
        {
            std::vector<float> w((size_t)3 * d * d), b((size_t)3 * d);
            const char *nm[3] = {"q_proj", "k_proj", "v_proj"};
            for (int j = 0; j < 3; ++j) {
                const HostTensor &tw = host_tensor(q + "attn_.mha_." + nm[j] + ".weight", {d, d});
                const HostTensor &tb = host_tensor(q + "attn_.mha_." + nm[j] + ".bias", {d});
                memcpy(w.data() + (size_t)j * d * d, tw.f32(), (size_t)d * d * 4);
                memcpy(b.data() + (size_t)j * d, tb.f32(), (size_t)d * 4);
            }
            L.wqkv = upload_gemm_weight(w.data(), w.size());
            L.bqkv = upload(b.data(), b.size());
        }
        L.wo = upload_gemm_tensor(q + "attn_.mha_.out_proj.weight", {d, d}); L.bo = upload_tensor(q + "attn_.mha_.out_proj.bias", {d});
        L.wpos = upload_gemm_tensor(q + "attn_.pos_proj_.weight", {d, d});
        L.pos_u = upload_tensor(q + "attn_.pos_bias_u_", {H, hd}); L.pos_v = upload_tensor(q + "attn_.pos_bias_v_", {H, hd});
        L.cv_ng = upload_tensor(q + "conv_.norm_.weight", {d}); L.cv_nb = upload_tensor(q + "conv_.norm_.bias", {d});
        L.pw1_w = upload_gemm_tensor(q + "conv_.pointwise_conv1_.weight", {2 * d, d}); L.pw1_b = upload_tensor(q + "conv_.pointwise_conv1_.bias", {2 * d});
        L.dw_w = taps_last(q + "conv_.depthwise_conv_.weight", d, K, false); L.dw_b = upload_tensor(q + "conv_.depthwise_conv_.bias", {d});
        L.bn_g = upload_tensor(q + "conv_.batch_norm_.weight", {d}); L.bn_b = upload_tensor(q + "conv_.batch_norm_.bias", {d});
        L.bn_mean = upload_tensor(q + "conv_.batch_norm_.running_mean", {d});
        {
            const HostTensor &tv = host_tensor(q + "conv_.batch_norm_.running_var", {d});
            std::vector<float> r(d);
            for (int c = 0; c < d; ++c) r[c] = 1.0f / sqrtf(tv.f32()[c] + 1e-5f);   // BatchNorm1d default eps (switch A3)
            L.bn_rstd = upload(r.data(), r.size());
        }
        L.pw2_w = upload_gemm_tensor(q + "conv_.pointwise_conv2_.weight", {d, d}); L.pw2_b = upload_tensor(q + "conv_.pointwise_conv2_.bias", {d});
        L.fin_g = upload_tensor(q + "final_norm_.weight", {d}); L.fin_b = upload_tensor(q + "final_norm_.bias", {d});
    }

    const int V = cfg.vocab_size, Hp = cfg.pred_hidden, J = cfg.joint_hidden, D = cfg.num_durations;
    // "sigma" K layout of the decode-loop weights (kernels/decode_gemv.hip): inside every block of 16 input features the
    // 4x4 index matrix is transposed, so one float4 holds a lane's k = 4s+kq operands of four consecutive MFMA steps.
    auto sigma = [](int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); };
    auto upload_sigma = [&](const float *w, int rows, int K) {
        std::vector<float> p((size_t)rows * K);
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < K; ++k) p[(size_t)r * K + sigma(k)] = w[(size_t)r * K + k];
        return upload(p.data(), p.size());
    };
    // bf16 decode weights (tolerance-class mode) in the LOAD ORDER of skinny_gemm_bf16_kernel: per (16-output tile, 32-k block) one 1 KB block
    // [lane][8] -- lane (col = lane & 15, kq = lane >> 4) holds W[row(tile, col)][32 blk + 8 kq .. + 7], so a wave's load instruction reads 1 KB of
    // consecutive addresses and a tile's weight stream is one contiguous run.  cell: the tile's columns are (gate, unit) pairs of the LSTM,
    // row = (col >> 2) * Hp + 4 tile + (col & 3); otherwise row = 16 tile + col (clamped to the last row: the kernel never stores those columns).
    auto upload_dec16 = [&](const float *w, int rows, int K, bool cell) {
        const int n_tiles = cell ? rows / 16 : (rows + 15) / 16, nblk = K / 32, hp = rows / 4;
        std::vector<float> p((size_t)n_tiles * nblk * 512);
        for (int t = 0; t < n_tiles; ++t)
            for (int blk = 0; blk < nblk; ++blk)
                for (int lane = 0; lane < 64; ++lane) {
                    const int col = lane & 15, kq = lane >> 4;
                    int row = cell ? (col >> 2) * hp + 4 * t + (col & 3) : 16 * t + col;
                    row = row < rows ? row : rows - 1;
                    const float *src = w + (size_t)row * K + 32 * blk + 8 * kq;
                    float *dst = p.data() + (((size_t)t * nblk + blk) * 64 + lane) * 8;
                    for (int e = 0; e < 8; ++e) dst[e] = src[e];
                }
        return upload_gemm_weight(p.data(), p.size());
    };
    if (cfg.ctc_vocab_size > 0) {
        dec.ctc_w = upload_gemm_tensor("ctc_decoder_.proj_.weight", {cfg.ctc_vocab_size, d});
        dec.ctc_b = upload_tensor("ctc_decoder_.proj_.bias", {cfg.ctc_vocab_size});
    }
    if (V <= 0) {                      // encoder-only model (Sortformer's NEST encoder): no prediction net, no joint
        PK_CHECK_LAUNCH();
        PK_HIP(hipStreamSynchronize(stream));
        st_.reset();
        return;
    }
    dec.embed = upload_tensor("prediction_.embed_.weight", {V, Hp});
    for (int l = 0; l < cfg.num_lstm_layers; ++l) {
        const std::string q = "prediction_.lstm_.cells_." + std::to_string(l) + ".";
        dec.wih[l] = upload_tensor(q + "input_proj_.weight", {4 * Hp, Hp});
        dec.bih[l] = upload_tensor(q + "input_proj_.bias", {4 * Hp});       // = b_ih + b_hh (convert_nemo.py:409-417)
        dec.whh[l] = upload_tensor(q + "hidden_proj_.weight", {4 * Hp, Hp});
        dec_whh_s[l] = upload_sigma(host_tensor(q + "hidden_proj_.weight", {4 * Hp, Hp}).f32(), 4 * Hp, Hp);
        dec_wih_s[l] = l ? upload_sigma(host_tensor(q + "input_proj_.weight", {4 * Hp, Hp}).f32(), 4 * Hp, Hp) : nullptr;
        if (cfg.gemm_bf16) {               // the decode GEMVs of the tolerance-class mode take bf16 weights (decode_gemv_bf16.hip)
            const bool t16 = Hp % 32 == 0 && J % 32 == 0;              // (run_tdt_loop's condition for the bf16 decode kernels)
            dec_whh16[l] = t16 ? upload_dec16(host_tensor(q + "hidden_proj_.weight", {4 * Hp, Hp}).f32(), 4 * Hp, Hp, true) : nullptr;
            dec_wih16[l] = (t16 && l) ? upload_dec16(host_tensor(q + "input_proj_.weight", {4 * Hp, Hp}).f32(), 4 * Hp, Hp, true) : nullptr;
        }
    }
    const std::string jp = cfg.joint_prefix;
    dec.we = upload_gemm_tensor(jp + "enc_proj_.weight", {J, d}); dec.be = upload_tensor(jp + "enc_proj_.bias", {J});
    dec.wp = upload_tensor(jp + "pred_proj_.weight", {J, Hp});
    dec_wp_s = upload_sigma(host_tensor(jp + "pred_proj_.weight", {J, Hp}).f32(), J, Hp);
    if (cfg.gemm_bf16 && Hp % 32 == 0 && J % 32 == 0) dec_wp16 = upload_dec16(host_tensor(jp + "pred_proj_.weight", {J, Hp}).f32(), J, Hp, false);
    dec.bp = (cfg.joint_pred_bias && st_->find(jp + "pred_proj_.bias")) ? upload_tensor(jp + "pred_proj_.bias", {J}) : nullptr;
    {
        std::vector<float> w((size_t)(V + D) * J), b((size_t)(V + D));
        const std::string ln = cfg.rnnt_head ? "out_proj_" : "label_proj_";
        memcpy(w.data(), host_tensor(jp + ln + ".weight", {V, J}).f32(), (size_t)V * J * 4);
        memcpy(b.data(), host_tensor(jp + ln + ".bias", {V}).f32(), (size_t)V * 4);
        if (D > 0) {
            memcpy(w.data() + (size_t)V * J, host_tensor(jp + "duration_proj_.weight", {D, J}).f32(), (size_t)D * J * 4);
            memcpy(b.data() + V, host_tensor(jp + "duration_proj_.bias", {D}).f32(), (size_t)D * 4);
        }
        wld = upload(w.data(), w.size());
        bld = upload(b.data(), b.size());
        wld_s = upload_sigma(w.data(), V + D, J);
        if (cfg.gemm_bf16 && Hp % 32 == 0 && J % 32 == 0) wld16 = upload_dec16(w.data(), V + D, J, false);
    }
    // g1 = E W_ih0^T + b  ([V][4Hp]) on the MFMA GEMM: the same natural-k chains the per-step projection would run
    float *g1 = dev_alloc((size_t)V * 4 * Hp);
    {   // decode-side table: always the fp32 chain (the per-step products it replaces are fp32)
        GemmArgs g{dec.embed, Hp, dec.wih[0], Hp, dec.bih[0], g1, 4 * Hp, nullptr, 0, 1.0f, V, 4 * Hp, Hp};
        run_gemm("g1_table", g, EPI_NONE, stream, /*fp32_weight=*/true);
    }
    PK_CHECK_LAUNCH();
    PK_HIP(hipStreamSynchronize(stream));
    dec.g1 = g1;
    st_.reset();   // host mapping no longer needed
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
    {
        int lo = 0, hi = 0;
        PK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));   // numerically lowest = highest priority
        int prio = hi;                                        // the decode loop of batch k-1 runs at the highest priority
#ifdef PK_EXPERIMENTAL
        if (const char *e = getenv("PK_DEC_PRIORITY")) {      // experiment builds only (tools/): "normal" / "low"
            if (!strcmp(e, "normal")) prio = 0;
            else if (!strcmp(e, "low")) prio = lo;
        }
#endif
        (void)lo;
#ifdef PK_EXPERIMENTAL
        if (const char *e = getenv("PK_DEC_NT")) dec_nt_weights = atoi(e);
#endif
        PK_HIP(hipStreamCreateWithPriority(&stream_dec, hipStreamNonBlocking, prio));
    }
    PK_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_done), sizeof(int), hipHostMallocDefault));
    // The model counts as resident only when EVERY weight made it: a failed upload (missing / mis-shaped / non-F32 tensor, HIP
    // out of memory) rolls everything back, so that a second to_gpu() reports the error again instead of returning PK_OK on a
    // half-initialised model.
    device_ = device;                       // dev_alloc / require_gpu inside the uploads need it
    try {
        build_mel_tables();
        upload_weights();
    } catch (...) {
        for (void *q : allocs_) (void)hipFree(q);
        allocs_.clear();
        if (h_done) { (void)hipHostFree(h_done); h_done = nullptr; }
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
        if (stream_dec) { (void)hipStreamDestroy(stream_dec); stream_dec = nullptr; }
        device_ = -1;
        throw;
    }
}

// ---- profiling hooks ----------------------------------------------------------------------------------------
void Model::klaunch_begin(const char *name, double flops, double bytes, hipStream_t s) {
    if (!prof) return;
    ProfileSink::Rec r;
    r.name = name; r.flops = flops; r.bytes = bytes;
    PK_HIP(hipEventCreate(&r.e0));
    PK_HIP(hipEventCreate(&r.e1));
    PK_HIP(hipEventRecord(r.e0, s));
    prof->recs.push_back(r);
}
void Model::klaunch_end(hipStream_t s) {
    if (!prof) return;
    PK_HIP(hipEventRecord(prof->recs.back().e1, s));
}
#define KL(name, flops, bytes, call) do { klaunch_begin(name, flops, bytes, s); call; klaunch_end(s); } while (0)

// every encoder-side product goes through here: fp32 chains, or bf16 operands when the model was loaded with gemm_bf16
void Model::run_gemm(const char *name, const GemmArgs &g, int epi, hipStream_t s, bool fp32_weight) {
    if (g.ln_stats) {                                               // large fp32 batches: the norm applied from per-row statistics while the tile kernel stages A
        if ((cfg.gemm_bf16 && !fp32_weight) || !gemm_ln_stats_applies(g, epi))
            fail(PK_ERR_INVALID, "%s: a LayerNorm from row statistics needs the fp32 tile kernel (gemm_ln_stats_applies)", name);
    } else if (g.ln_g && !((cfg.gemm_bf16 && !fp32_weight) ? gemm_smallm_bf16_ln_applies(g, epi) : gemm_smallm_ln_applies(g, epi)))
        fail(PK_ERR_INVALID, "%s: a folded LayerNorm needs one of the small-M kernels (gemm_smallm_ln_applies / gemm_smallm_bf16_ln_applies)", name);
    if (g.pre_g && !((cfg.gemm_bf16 && !fp32_weight) ? gemm_smallm_bf16_pre_applies(g, epi) : gemm_smallm_pre_applies(g, epi)))
        fail(PK_ERR_INVALID, "%s: a norm in front of the folded LayerNorm needs one of the small-M kernels (gemm_smallm_pre_applies / gemm_smallm_bf16_pre_applies)", name);
    if (cfg.gemm_bf16 && !fp32_weight) {
        if (g.K % 64) fail(PK_ERR_UNSUPPORTED, "gemm_bf16 needs K %% 64 == 0 (%s has K = %d)", name, g.K);
        if (g.out_bf16 && (g.remap_rows != 0 || (g.ldo & 3) != 0 || (g.N & 3) != 0 || g.sigma_cols != 0 || epi == EPI_RESID || epi == EPI_GLU))
            fail(PK_ERR_INVALID, "%s: a bf16 output needs the row-major wide epilogue", name);
        KL(name, gemm_flops(g, epi), 0.0, launch_gemm_bf16(g, epi, s));
    } else {
        if (g.a_bf16 || g.out_bf16) fail(PK_ERR_INVALID, "%s: bf16 activations outside the bf16 GEMM path", name);
        if (g.a_sigma && !(g.W_sig && g.M <= kSmallMRows && g.K % 64 == 0 && g.N % 16 == 0))
            fail(PK_ERR_INVALID, "%s: sigma-K activations need the small-M kernel and a sigma-K weight copy", name);
        KL(name, gemm_flops(g, epi), 0.0, launch_gemm(g, epi, s));
    }
}

void Model::gemm(const char *name, const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, float *out, int64_t ldo,
                 int M, int N, int K, int epi, const float *resid, int64_t ldr, float alpha, hipStream_t s, int a_bf16, int out_bf16) {
    GemmArgs g{A, lda, W, ldw, bias, out, ldo, resid, ldr, alpha, M, N, K};
    g.a_bf16 = a_bf16;
    g.out_bf16 = out_bf16;
    g.fast_act = (cfg.gemm_bf16 && (epi == EPI_SILU || epi == EPI_GLU)) ? 1 : 0;     // bf16 mode: tolerance-class activations (kernels.hpp)
    run_gemm(name, g, epi, s);
}

const std::vector<Model::SigW> &Model::sigma_weights() {
    if (sig_built_) return sig_layers_;
    require_gpu();
    const size_t d = cfg.hidden_size, f = cfg.ffn_intermediate;
    if (d % 64 || f % 64) { sig_built_ = true; return sig_layers_; }
    // (+ one more copy of the encoder's product weights in HBM: 0.4 GB for tdt-ctc-110m, 2.4 GB for the 600M models, half that in the bf16
    //  mode; pk_model_to_gpu's note)
    // gemm_bf16 mode: the same eight weights per layer as bf16 OPERAND TILES of the small-M bf16 kernel (GemmArgs::W_t16)
    const bool b16 = cfg.gemm_bf16 != 0;
    const size_t per_layer = 4 * f * d + 7 * d * d;
    try {
        sig_buf_.reserve((size_t)cfg.num_layers * per_layer * (b16 ? 2 : 4));
        float *p = sig_buf_.as<float>();
        sig_layers_.resize(cfg.num_layers);
        for (int l = 0; l < cfg.num_layers; ++l) {
            const LayerW &W = layers[l];
            auto sig = [&](const float *w, size_t rows, size_t K) {
                if (b16) launch_tile_copy_bf16(w, p, (int64_t)rows, (int)K, (int64_t)K, stream);
                else launch_sigma_copy(w, p, (int64_t)rows, (int)K, (int64_t)K, stream);
                const float *r = p;
                p += b16 ? rows * K / 2 : rows * K;                 // (bf16: two elements per float slot)
                return r;
            };
            SigW &S = sig_layers_[l];
            S.ffn1_w1 = sig(W.ffn1_w1, f, d); S.ffn1_w2 = sig(W.ffn1_w2, d, f);
            S.ffn2_w1 = sig(W.ffn2_w1, f, d); S.ffn2_w2 = sig(W.ffn2_w2, d, f);
            S.wqkv = sig(W.wqkv, 3 * d, d); S.wo = sig(W.wo, d, d);
            S.pw1 = sig(W.pw1_w, 2 * d, d); S.pw2 = sig(W.pw2_w, d, d);
        }
        PK_CHECK_LAUNCH();
        PK_HIP(hipStreamSynchronize(stream));
    } catch (...) {
        // nothing half-built survives: the next call tries again, and until then the natural-layout kernels are used
        sig_layers_.clear();
        if (sig_buf_.p) { (void)hipFree(sig_buf_.p); sig_buf_.p = nullptr; sig_buf_.cap = 0; }
        throw;
    }
    sig_built_ = true;                       // only now: every copy is complete
    return sig_layers_;
}

// ---- workspace ----------------------------------------------------------------------------------------------
static inline int sub_len(int n) { return (n - 1) / 2 + 1; }        // one stride-2 stage of the subsampling (src/encoder.cpp:208-217)

// encoder-side buffers for sum_Tm mel frames / sum_H2 rows after dw1 / sum_T encoder rows in all
static void reserve_encoder(Workspace &w, const pk_config &c, size_t pcm_samples, bool logmel, size_t sum_Tm, size_t sum_H2, size_t sum_T) {
    const int F = c.mel_bins, C = c.subsampling_channels, d = c.hidden_size;
    const int W2 = sub_len(sub_len(F)), W3 = sub_len(W2);
    const size_t f = sizeof(float), M = sum_T;
    if (pcm_samples) w.pcm.reserve(pcm_samples * f);
    if (logmel) w.logmel.reserve((size_t)F * (sum_Tm + 16 * (size_t)(w.rag_cap_clips > 0 ? w.rag_cap_clips : 1)) * f);   // rows padded to 16 frames per clip
    w.feats.reserve(sum_Tm * F * f);
    w.a2.reserve(sum_H2 * W2 * C * f); w.a3.reserve(sum_H2 * W2 * C * f);
    w.a4.reserve(sum_T * W3 * C * f); w.flat.reserve(sum_T * W3 * C * f);
    w.x.reserve(M * d * f); w.n.reserve(M * d * f); w.hbuf.reserve(M * c.ffn_intermediate * f); w.qkv.reserve(M * 3 * d * f);
    w.ctx.reserve(M * d * f); w.g.reserve(M * d * f); w.dwb.reserve(M * d * f);
    if (c.ctc_vocab_size > 0) { w.ctc_logits.reserve(M * c.ctc_vocab_size * f); }
    w.best_idx.reserve(M * sizeof(int)); w.best_lp.reserve(M * f);
}

void Workspace::size_for(const pk_config &c, int B_, int64_t n_samples_, int Tm_) {
    B = B_; Tm = Tm_;
    ragged = false;
    dec_Tb = dec_row0 = nullptr;
    const bool own_pcm = n_samples_ > 0;          // a negative length: the caller owns the PCM buffer (pipelined batch slots)
    n_samples = n_samples_ < 0 ? -n_samples_ : n_samples_;
    const int H2 = sub_len(sub_len(Tm));
    T = T_run = sub_len(H2);
    rag_cap_rows = (size_t)B * T; rag_cap_clips = B; rag_cap_samples = (int64_t)B * n_samples; rag_cap_clip = n_samples;
    reserve_encoder(*this, c, own_pcm ? (size_t)B * n_samples : 0, n_samples > 0, (size_t)B * Tm, (size_t)B * H2, (size_t)B * T);
    reserve_decode(c);
}

// the TDT / RNNT decode state of B utterances of <= T frames, rag_cap_rows frames in all (B, T, rag_cap_rows set by the caller)
void Workspace::reserve_decode(const pk_config &c) {
    const size_t M = rag_cap_rows ? rag_cap_rows : (size_t)B * T, f = sizeof(float);
    max_tokens = T * (c.max_symbols_per_step > 0 ? c.max_symbols_per_step : 10);
    const int Hp = c.pred_hidden, J = c.joint_hidden, L = c.num_lstm_layers, VD = c.vocab_size + c.num_durations;
    ep.reserve(M * J * f); gh.reserve((size_t)B * 4 * Hp * f); gi.reserve((size_t)B * 4 * Hp * f); pp.reserve((size_t)B * J * f);
    const size_t Bw = B < kDecWindowMax ? kDecWindowMax : B;         // a single utterance decodes through a frame window of kDecWindowMax rows (TdtState::F)
    z.reserve(Bw * J * f); logits.reserve(Bw * VD * f);
    h.reserve((size_t)L * B * Hp * f); this->c.reserve((size_t)L * B * Hp * f); hn.reserve((size_t)L * B * Hp * f); cn.reserve((size_t)L * B * Hp * f);
    ints.reserve((size_t)(8 * B + 8) * sizeof(int));
    const size_t tok = (size_t)B * max_tokens;
    ids.reserve(tok * sizeof(int)); start.reserve(tok * sizeof(int)); end.reserve(tok * sizeof(int)); conf.reserve(tok * f);
    lens.reserve((size_t)B * sizeof(int));
    margin.reserve((size_t)B * f);
}

// decode-only workspace (no encoder buffers): the lock-step TDT state of a GROUP of pipelined batches (capi.cpp)
void Workspace::size_decode(const pk_config &c, int B_, int T_, size_t rows_cap) {
    B = B_; T = T_run = T_;
    ragged = false;
    dec_Tb = dec_row0 = nullptr;
    rag_cap_rows = rows_cap ? rows_cap : (size_t)B * T;
    reserve_decode(c);
}

// ---- ragged batches -------------------------------------------------------------------------------------------
void RagBatch::build_from_samples(const int64_t *lens, int B_, int att_block_rows) {
    B = B_; level = 0;
    pcm_off.assign((size_t)B + 1, 0);
    Tm.resize(B);
    for (int b = 0; b < B; ++b) {
        if (lens[b] <= 256) fail(PK_ERR_INVALID, "clip %d has %lld samples: every clip needs more than 256 (one STFT frame with reflect padding)", b, (long long)lens[b]);
        if (lens[b] > ((int64_t)1 << 30)) fail(PK_ERR_UNSUPPORTED, "clip %d has %lld samples", b, (long long)lens[b]);
        pcm_off[b + 1] = pcm_off[b] + lens[b];
        Tm[b] = (int)(1 + lens[b] / 160);
    }
    n_samples = pcm_off[B];
    finish(att_block_rows);
}
void RagBatch::build_from_mel(const int *Tm_, int B_, int att_block_rows) {
    B = B_; level = 1;
    pcm_off.clear(); n_samples = 0;
    Tm.assign(Tm_, Tm_ + B);
    for (int b = 0; b < B; ++b) if (Tm[b] < 1) fail(PK_ERR_INVALID, "utterance %d has %d mel frames", b, Tm[b]);
    finish(att_block_rows);
}
void RagBatch::build_from_frames(const int *T_, int B_, int att_block_rows) {
    B = B_; level = 2;
    pcm_off.clear(); n_samples = 0; Tm.clear(); H2.clear();
    T.assign(T_, T_ + B);
    for (int b = 0; b < B; ++b) if (T[b] < 1) fail(PK_ERR_INVALID, "utterance %d has %d frames", b, T[b]);
    finish(att_block_rows);
}
void RagBatch::finish(int att_block_rows) {
    if (B <= 0) fail(PK_ERR_INVALID, "a ragged batch needs at least one utterance");
    if (level <= 1) {
        H2.resize(B); T.resize(B);
        for (int b = 0; b < B; ++b) { H2[b] = sub_len(sub_len(Tm[b])); T[b] = sub_len(H2[b]); }
    }
    int64_t sTm = 0, sH2 = 0, sT = 0;
    Tm_max = 0; T_max = 0;
    for (int b = 0; b < B; ++b) {
        if (level <= 1) { sTm += Tm[b]; sH2 += H2[b]; Tm_max = std::max(Tm_max, Tm[b]); }
        sT += T[b]; T_max = std::max(T_max, T[b]);
    }
    if (sTm > 0x7fffffff / 2 || sT > 0x7fffffff / 2) fail(PK_ERR_UNSUPPORTED, "ragged batch too large (%lld mel frames)", (long long)sTm);
    sum_Tm = (int)sTm; sum_H2 = (int)sH2; sum_T = (int)sT;
    strip_rows = sub_conv1_dw1_strip_rows(sum_H2);
    dw_frames = dwconv_strip_frames(sum_T);
    att_rows = att_block_rows;
    // ---- the image: every table as int32 words (pcm_off as int64 at word 0) ----
    image.clear();
    auto put_arr = [&](const std::vector<int> &v) { const size_t o = image.size(); image.insert(image.end(), v.begin(), v.end()); return o; };
    auto put_off = [&](const std::vector<int> &v) {                 // exclusive prefix sums, B + 1 entries
        const size_t o = image.size();
        int acc = 0;
        for (int b = 0; b < B; ++b) { image.push_back(acc); acc += v[b]; }
        image.push_back(acc);
        return o;
    };
    auto put_units = [&](const std::vector<int> &n, int gran, int &count) {   // one unit per strip of `gran` rows of one utterance
        if (image.size() & 1) image.push_back(0);
        const size_t o = image.size();
        count = 0;
        for (int b = 0; b < B; ++b)
            for (int r0 = 0; r0 < n[b]; r0 += gran) { image.push_back(b); image.push_back(r0); ++count; }
        return o;
    };
    o_pcm_off = 0;
    if (level == 0) {
        image.resize(2 * ((size_t)B + 1));
        memcpy(image.data(), pcm_off.data(), ((size_t)B + 1) * 8);
    }
    if (level <= 1) {
        o_Tm = put_arr(Tm); o_Tm_off = put_off(Tm); o_H2 = put_arr(H2); o_H2_off = put_off(H2);
        std::vector<int> padded(B);
        for (int b = 0; b < B; ++b) padded[b] = mel_logmel_pitch(Tm[b]);
        o_Tm_pad_off = put_off(padded);
        sum_Tm_pad = image[o_Tm_pad_off + B];
    }
    o_T = put_arr(T); o_T_off = put_off(T);
    if (level <= 1) {
        o_u_c1 = put_units(H2, strip_rows, n_u_c1);
        o_u_row = put_units(T, 1, n_u_row);
    } else { n_u_c1 = n_u_row = 0; }
    o_u_dw = put_units(T, dw_frames, n_u_dw);
    o_u_att = put_units(T, att_rows, n_u_att);
}
size_t RagBatch::image_words_bound(int max_clips, int64_t max_total) {
    // (max_total counted in samples: the frame counts below are upper bounds for every level)
    const size_t B = (size_t)max_clips, sTm = (size_t)(max_total / 160) + B, sH2 = sTm / 4 + 2 * B, sT = sH2 / 2 + B;
    return 2 * (B + 1) + 3 * (2 * B + 1) + (B + 1) + 2 * (sH2 / 2 + B + sT + sT / 2 + B + sT / 32 + B) + 16;
}

void Workspace::size_ragged(const pk_config &c, int max_clips, int64_t max_total, int64_t max_clip, bool own_pcm, int level) {
    if (max_clips <= 0 || max_total <= 0 || max_clip <= 0 || max_clip > max_total) fail(PK_ERR_INVALID, "ragged capacity: max_clips / max_total / max_clip");
    const size_t Bc = (size_t)max_clips;
    size_t sTm, sH2, sT;
    int T_cap;
    if (level == 0) {
        sTm = (size_t)(max_total / 160) + Bc; sH2 = sTm / 4 + 2 * Bc; sT = sH2 / 2 + Bc;      // sum of ceil(n/2) <= sum n / 2 + B, twice, once
        T_cap = sub_len(sub_len(sub_len((int)(1 + max_clip / 160))));
    } else if (level == 1) {
        sTm = (size_t)max_total; sH2 = sTm / 4 + 2 * Bc; sT = sH2 / 2 + Bc;
        T_cap = sub_len(sub_len(sub_len((int)max_clip)));
    } else {
        sTm = sH2 = 0; sT = (size_t)max_total;
        T_cap = (int)max_clip;
    }
    B = max_clips; T = T_run = T_cap; Tm = 0; n_samples = 0;
    rag_cap_rows = sT; rag_cap_clips = max_clips; rag_cap_samples = max_total; rag_cap_clip = max_clip;
    if (level <= 1) reserve_encoder(*this, c, (own_pcm && level == 0) ? (size_t)max_total : 0, level == 0, sTm, sH2, sT);
    else {
        const size_t f = sizeof(float), M = sT, d = c.hidden_size;
        x.reserve(M * d * f); n.reserve(M * d * f); hbuf.reserve(M * c.ffn_intermediate * f); qkv.reserve(M * 3 * d * f);
        ctx.reserve(M * d * f); g.reserve(M * d * f); dwb.reserve(M * d * f);
        if (c.ctc_vocab_size > 0) ctc_logits.reserve(M * c.ctc_vocab_size * f);
        best_idx.reserve(M * sizeof(int)); best_lp.reserve(M * f);
    }
    reserve_decode(c);
    const size_t words = RagBatch::image_words_bound(max_clips, level == 0 ? max_total : (level == 1 ? max_total * 160 : max_total * 1280));
    ragdev.reserve(words * 4);
    if (words > rag_pinned_words) {
        if (rag_copied) PK_HIP(hipEventSynchronize(rag_copied));
        if (rag_pinned) { PK_HIP(hipHostFree(rag_pinned)); rag_pinned = nullptr; rag_pinned_words = 0; }
        PK_HIP(hipHostMalloc(reinterpret_cast<void **>(&rag_pinned), words * 4, hipHostMallocDefault));
        rag_pinned_words = words;
    }
    if (!rag_copied) PK_HIP(hipEventCreateWithFlags(&rag_copied, hipEventDisableTiming));
}

void Workspace::set_uniform(int B_, int64_t n_samples_) {
    if (n_samples_ > 0) {
        if (B_ > rag_cap_clips || n_samples_ > rag_cap_clip || (int64_t)B_ * n_samples_ > rag_cap_samples)
            fail(PK_ERR_INVALID, "batch of %d clips x %lld samples exceeds the workspace (%d clips, %lld samples in all, %lld per clip)", B_, (long long)n_samples_,
                 rag_cap_clips, (long long)rag_cap_samples, (long long)rag_cap_clip);
        n_samples = n_samples_;
        Tm = (int)(1 + n_samples / 160);
        T_run = sub_len(sub_len(sub_len(Tm)));
    }
    ragged = false;
    dec_Tb = dec_row0 = nullptr;
    (void)B_;
}

void Workspace::set_ragged(const RagBatch &r, hipStream_t s) {
    if (r.B > rag_cap_clips || (size_t)r.sum_T > rag_cap_rows || r.T_max > T || (r.level == 0 && r.n_samples > rag_cap_samples))
        fail(PK_ERR_INVALID, "ragged batch (%d clips, %lld samples, %d encoder rows, longest %d frames) exceeds the workspace (%d clips, %lld samples, %zu rows, %d frames)",
             r.B, (long long)r.n_samples, r.sum_T, r.T_max, rag_cap_clips, (long long)rag_cap_samples, rag_cap_rows, T);
    rag = r;
    ragged = true;
    const size_t words = rag.image.size();
    ragdev.reserve(words * 4);                                      // (no-op inside the reserved capacity)
    if (words > rag_pinned_words) {
        if (rag_copied) PK_HIP(hipEventSynchronize(rag_copied));
        if (rag_pinned) { PK_HIP(hipHostFree(rag_pinned)); rag_pinned = nullptr; rag_pinned_words = 0; }
        PK_HIP(hipHostMalloc(reinterpret_cast<void **>(&rag_pinned), words * 4, hipHostMallocDefault));
        rag_pinned_words = words;
    }
    if (!rag_copied) PK_HIP(hipEventCreateWithFlags(&rag_copied, hipEventDisableTiming));
    PK_HIP(hipEventSynchronize(rag_copied));                        // the previous upload out of the pinned copy has executed (normally long ago)
    memcpy(rag_pinned, rag.image.data(), words * 4);
    PK_HIP(hipMemcpyAsync(ragdev.p, rag_pinned, words * 4, hipMemcpyHostToDevice, s));
    PK_HIP(hipEventRecord(rag_copied, s));
    const int32_t *dv = ragdev.as<int32_t>();
    rv = RagDev();
    const int *dT = dv + rag.o_T, *dToff = dv + rag.o_T_off;
    if (rag.level <= 1) {
        const int *dTm = dv + rag.o_Tm, *dTmoff = dv + rag.o_Tm_off, *dH2 = dv + rag.o_H2, *dH2off = dv + rag.o_H2_off;
        if (rag.level == 0) {
            rv.mel.pcm_off = reinterpret_cast<const int64_t *>(dv + rag.o_pcm_off); rv.mel.Tm = dTm; rv.mel.Tm_off = dTmoff; rv.mel.max_frames = rag.Tm_max;
            rv.mel.Tm_pad_off = dv + rag.o_Tm_pad_off;
        }
        rv.c1.strips = {reinterpret_cast<const RagUnit *>(dv + rag.o_u_c1), rag.n_u_c1};
        rv.c1.strip_rows = rag.strip_rows;
        rv.c1.Tm = dTm; rv.c1.Tm_off = dTmoff; rv.c1.H2 = dH2; rv.c1.H2_off = dH2off; rv.c1.T = dT; rv.c1.T_off = dToff;
        rv.dw2 = rv.c1;
        rv.dw2.strips = {reinterpret_cast<const RagUnit *>(dv + rag.o_u_row), rag.n_u_row};
        rv.dw2.strip_rows = 1;
    }
    rv.att.units = {reinterpret_cast<const RagUnit *>(dv + rag.o_u_att), rag.n_u_att};
    rv.att.T = dT; rv.att.T_off = dToff; rv.att.T_max = rag.T_max;
    rv.dwc = rv.att;
    rv.dwc.units = {reinterpret_cast<const RagUnit *>(dv + rag.o_u_dw), rag.n_u_dw};
    rv.seq = rv.att;
    rv.seq.units = RagUnits();
    dec_Tb = dT; dec_row0 = dToff;
}

// ---- stages ---------------------------------------------------------------------------------------------------
void Model::run_mel(const float *d_pcm, int B, int64_t n_samples, float *d_logmel, float *d_feats, hipStream_t s, const RagDev *rv) {
    if (rv) fail(PK_ERR_INVALID, "run_mel: ragged batches go through run_mel_ws");
    const int n_frames = (int)(1 + n_samples / 160);
    const double bytes_in = (double)B * n_samples * 4, bytes_lm = (double)B * cfg.mel_bins * n_frames * 4;
    KL("mel_logmel", 0.0, bytes_in + bytes_lm, launch_mel_logmel(d_pcm, B, n_samples, n_frames, mel, d_logmel, s));
    KL("mel_normalize", 0.0, 2.0 * bytes_lm, launch_mel_normalize(d_logmel, B, cfg.mel_bins, n_frames, cfg.mel_normalize_off ? 0 : 1, d_feats, s));
}

// preprocess_audio of the workspace's batch (uniform: w.B clips of w.n_samples; ragged: w.rag) -> w.logmel, w.feats
void Model::run_mel_ws(Workspace &w, const float *d_pcm, int B, hipStream_t s) {
    if (!w.ragged) { run_mel(d_pcm, B, w.n_samples, w.logmel.as<float>(), w.feats.as<float>(), s); return; }
    if (w.rag.level != 0) fail(PK_ERR_INVALID, "run_mel_ws: the ragged batch was not built from PCM lengths");
    const double bytes_in = (double)w.rag.n_samples * 4, bytes_lm = (double)cfg.mel_bins * w.rag.sum_Tm * 4;
    KL("mel_logmel", 0.0, bytes_in + bytes_lm, launch_mel_logmel(d_pcm, w.rag.B, 0, 0, mel, w.logmel.as<float>(), s, w.rv.mel));
    KL("mel_normalize", 0.0, 2.0 * bytes_lm,
       launch_mel_normalize(w.logmel.as<float>(), w.rag.B, cfg.mel_bins, 0, cfg.mel_normalize_off ? 0 : 1, w.feats.as<float>(), s, w.rv.mel));
}

// sinusoidal_position_embedding (src/encoder.cpp:9-30): float math on the host, exactly as the reference does,
// then pos_proj_ of every layer (src/encoder.cpp:148) -- batch-independent, so computed once per sequence length.
bool Model::attn_bf16(int T) const {
    if (!cfg.gemm_bf16) return false;
    const size_t lds = relpos_attention_bf16_lds_bytes(T, cfg.hidden_size / cfg.num_heads);
    return lds > 0 && lds <= (size_t)150 * 1024;
}

// The tables are kept for the LONGEST sequence seen so far: row p of the table of a T-frame sequence (position T - 1 - p) is row
// p + (pos_T - T) of the table built for pos_T >= T frames -- the same float position, hence the same sin / cos values and the same row of
// the row-wise pos_proj_ product.  Shorter sequences and the utterances of a ragged batch read their window of it (launch_relpos_attention:
// pos_row0 / SeqRag::pos_T); the tables are rebuilt only when a longer sequence arrives.
void Model::ensure_pos_tables(int T, hipStream_t s) {
    const bool want16 = attn_bf16(T);
    PosTab &tab_set = want16 ? pos16 : pos32;
    if (T <= tab_set.T) return;
    DevBuf &pos_proj = tab_set.proj, &pos_cvec = tab_set.cvec;
    int &pos_T = tab_set.T;
    const int d = cfg.hidden_size, P = 2 * T - 1;
    std::vector<float> pe((size_t)P * d);
    for (int p = 0; p < P; ++p) {
        const float position = (float)(T - 1 - p);
        for (int i = 0; i < d; i += 2) {
            const float div_term = std::exp((float)i * (-std::log(10000.0f) / (float)d));
            pe[(size_t)p * d + i] = std::sin(position * div_term);
            if (i + 1 < d) pe[(size_t)p * d + i + 1] = std::cos(position * div_term);
        }
    }
    PK_HIP(hipStreamSynchronize(s));       // previous users of the tables
    pos_pe.reserve(pe.size() * 4);
    pos_proj.reserve((size_t)cfg.num_layers * P * d * 4);
    PK_HIP(hipMemcpy(pos_pe.p, pe.data(), pe.size() * 4, hipMemcpyHostToDevice));
    if (attn_bf16(T)) {
        // tolerance-class mode: the table as bf16, natural columns (attention_bf16.hip), plus c[l][h][p] = (v_h - u_h) . P_p
        const int H = cfg.num_heads;
        pos_cvec.reserve((size_t)cfg.num_layers * H * P * 4);
        for (int l = 0; l < cfg.num_layers; ++l) {
            __bf16 *tab = reinterpret_cast<__bf16 *>(pos_proj.p) + (size_t)l * P * d;
            GemmArgs g{pos_pe.as<float>(), d, layers[l].wpos, d, nullptr, reinterpret_cast<float *>(tab), d, nullptr, 0, 1.0f, P, d, d};
            g.out_bf16 = 1;
            run_gemm("pos_proj", g, EPI_NONE, s);
            launch_pos_cvec(tab, layers[l].pos_u, layers[l].pos_v, P, d, H, pos_cvec.as<float>() + (size_t)l * H * P, s);
        }
        pos_T = T;
        return;
    }
    for (int l = 0; l < cfg.num_layers; ++l) {
        // written in the sigma column layout the attention kernel loads its MFMA operands in (kernels.hpp: GemmArgs::sigma_cols)
        GemmArgs g{pos_pe.as<float>(), d, layers[l].wpos, d, nullptr, pos_proj.as<float>() + (size_t)l * P * d, d, nullptr, 0, 1.0f, P, d, d};
        g.sigma_cols = d;
        run_gemm("pos_proj", g, EPI_NONE, s);
    }
    pos_T = T;
}

void Model::run_subsample(Workspace &w, const float *d_feats, int B, int Tm, float *d_x, hipStream_t s) {
    const int F = cfg.mel_bins, C = cfg.subsampling_channels, d = cfg.hidden_size;
    auto sl = [](int n) { return (n - 1) / 2 + 1; };
    const int H1 = sl(Tm), W1 = sl(F), H2 = sl(H1), W2 = sl(W1), H3 = sl(H2), W3 = sl(W2);
    // a ragged batch (w.ragged; B / Tm ignored): the same kernels on packed rows, every utterance with its own extents (kernels.hpp: SubRag)
    const bool rg = w.ragged;
    const double rows2 = rg ? (double)w.rag.sum_H2 : (double)B * H2, rows3 = rg ? (double)w.rag.sum_T : (double)B * H3;
    const double px2 = rows2 * W2, px3 = rows3 * W3, frames = rg ? (double)w.rag.sum_Tm : (double)B * Tm;
    // conv1 + ReLU + dw1 (fused)  src/encoder.cpp:223-226
    KL("sub_conv1_dw1", px2 * C * (81.0 + 9.0) * 2.0, frames * F * 4 + px2 * C * 4,
       launch_sub_conv1_dw1(d_feats, B, Tm, F, C, sub.c1w, sub.c1b, sub.d1w, sub.d1b, w.a2.as<float>(), s, rg ? w.rv.c1 : SubRag()));
    // conv2 (1x1) + ReLU  :227-228
    gemm("sub_pw", w.a2.as<float>(), C, sub.c2w, C, sub.c2b, w.a3.as<float>(), C, (int)px2, C, C, EPI_RELU, nullptr, 0, 1.0f, s);
    // dw2  :230
    KL("sub_dw2", px3 * C * 18.0, (px2 + px3) * C * 4, launch_sub_dw(w.a3.as<float>(), B, H2, W2, C, sub.d2w, sub.d2b, w.a4.as<float>(), s, rg ? w.rv.dw2 : SubRag()));
    // conv3 (1x1) + ReLU, written directly in permute(0,2,1,3)+reshape order (feature = c*W3 + f)  :231-238
    {
        GemmArgs g{w.a4.as<float>(), C, sub.c3w, C, sub.c3b, w.flat.as<float>(), C, nullptr, 0, 1.0f, (int)px3, C, C};
        g.remap_rows = W3; g.remap_gs = (int64_t)C * W3; g.remap_rs = 1; g.remap_cs = W3;
        run_gemm("sub_pw", g, EPI_RELU, s);
    }
    // proj_  :240
    gemm("sub_proj", w.flat.as<float>(), (int64_t)C * W3, sub.pw, (int64_t)C * W3, sub.pb, d_x, d, (int)rows3, d, C * W3, EPI_NONE, nullptr, 0, 1.0f, s);
    if (cfg.xscaling)                                               // streaming_encoder.cpp:402-406: x = x * sqrt(hidden_size)
        KL("xscale", 0.0, 2.0 * rows3 * d * 4, launch_scale(d_x, (int64_t)rows3 * d, sqrtf((float)d), s));
    (void)W1; (void)H1;
}

// y = LayerNorm(x) -> n, then the product g (A = n) -- or, for a handful of rows on the fp32 chain kernels, the product with the norm folded in
// (gemm_smallm_ln_kernel: bit for bit the same; one launch instead of two).  kLnFoldRows: beyond about one round of its workgroups the redundant
// normalisation (every 32-column workgroup normalises its 16 rows) costs more than the LayerNorm launch it saves.
static constexpr int kLnFoldRows = 256;
bool Model::ln_folds(const GemmArgs &g, int epi, int64_t rows) const {
    return !cfg.gemm_bf16 && rows <= kLnFoldRows && g.W_sig && gemm_smallm_ln_applies(g, epi);
}
// Large fp32 batches (round 6; round-5 verdict item 4): the LayerNorm in front of a wide product as a STATISTICS pass (launch_layernorm_stats: one read of x,
// {mean, rstd} per row into the buffer the normalised rows used to occupy) with the product's tile kernel normalising while it stages A (gemm_pipe.hpp: LNA)
// -- the same values bit for bit (tests/test_gpu_primitives.py), without the write and the re-read of the normalised tensor.
// MEASURED SLOWER, so OFF in production: the LayerNorm launches of a 64 x 10 s step go from 0.87 to 0.68 ms (statistics 7.2 us against 13 us per launch), but the
// products that take the norm pay more than that -- fc1 152 -> 163 us with the three operations per element in front of the staging stores, 167 us with
// them moved under the MFMAs of an earlier sub-step (the wait for the K tile in flight moves with them), qkv + 5 %, pw1 + 8-12 %: the step 18.46 -> 18.87 / 19.1 ms
// (profiles/r06_ln_stats_fold_ab.txt).  The global -> VGPR -> LDS staging is what the fp32 loop is bound by; anything added to it costs more
// than a memory-bound launch of 13 us.  EXPERIMENTAL builds: PK_LN_STATS=1 switches it on.
static bool ln_stats_on() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_LN_STATS"); return e ? atoi(e) != 0 : false; }();
    return on;
#else
    return false;
#endif
}
bool Model::ln_stats_folds(const GemmArgs &g, int epi, const float *ng, const float *nb, const float *x, const float *stats) const {
    if (cfg.gemm_bf16 || !ln_stats_on()) return false;
    GemmArgs fg = g;
    fg.A = x; fg.lda = cfg.hidden_size; fg.a_sigma = 0; fg.a_bf16 = 0; fg.ln_g = ng; fg.ln_b = nb; fg.ln_eps = 1e-5f; fg.ln_stats = stats;
    return fg.K == cfg.hidden_size && gemm_ln_stats_applies(fg, epi);
}
// norm_state: 0 = x is un-normalised, 1 = n holds the normalised rows (a previous kernel wrote them), 2 = n holds the rows' statistics for this norm
void Model::ln_gemm(const char *name, const GemmArgs &g, int epi, const float *ng, const float *nb, int norm_state, int ymode, const float *x, float *n,
                    int64_t rows, hipStream_t s) {
    const int d = cfg.hidden_size;
    if (norm_state != 1) {
        GemmArgs fg = g;
        fg.A = x; fg.lda = d; fg.a_sigma = 0; fg.a_bf16 = 0; fg.ln_g = ng; fg.ln_b = nb; fg.ln_eps = 1e-5f;
        if (norm_state == 2 || ln_stats_folds(g, epi, ng, nb, x, n)) {
            if (norm_state != 2) KL("layernorm_stats", 0.0, 1.0 * rows * d * 4, launch_layernorm_stats(x, rows, d, 1e-5f, n, s));
            fg.ln_stats = n;
            run_gemm(name, fg, epi, s);
            return;
        }
        if (ln_folds(fg, epi, rows)) { run_gemm(name, fg, epi, s); return; }
        KL("layernorm", 0.0, (cfg.gemm_bf16 ? 1.5 : 2.0) * rows * d * 4, launch_layernorm(x, rows, d, ng, nb, 1e-5f, n, s, ymode));
    }
    run_gemm(name, g, epi, s);
}

// FeedForward::forward (src/encoder.cpp:39-46): x += 0.5 * fc2(silu(fc1(LN(x))))
void Model::ffn(Workspace &w, const LayerW &L, bool second, int64_t rows, hipStream_t s, int norm_state, const SigW *sg) {
    const int d = cfg.hidden_size, f = cfg.ffn_intermediate;
    float *x = w.x.as<float>(), *n = w.n.as<float>(), *h = w.hbuf.as<float>();
    // bf16 mode: the normalised rows and the fc1 activations exist only as GEMM operands -- their producers round them to bf16 (RNE, the
    // rounding the GEMM's staging path would apply: same operand values) and store HALF the bytes in the same buffers.
    // sg (small batches, fp32): the same buffers in the sigma K layout, the products on the tiled weight copies (run_layers).
    const int a16 = cfg.gemm_bf16 ? 1 : 0;
    GemmArgs g1{n, d, second ? L.ffn2_w1 : L.ffn1_w1, d, second ? L.ffn2_b1 : L.ffn1_b1, h, f, nullptr, 0, 1.0f, (int)rows, f, d};
    g1.a_bf16 = a16; g1.out_bf16 = a16;
    g1.fast_act = a16;
    // bf16 mode, large batches: the fc1 activations live in 32 x 16 blocks between fc1's register epilogue and fc2's LDS-DMA
    // (GemmArgs::out_blocked / a_blocked: every store instruction of the epilogue writes one contiguous KB); h holds rows rounded up to 32
    const bool blocked = a16 && !sg && gemm_bf16_blocked_handoff((int)rows, f, d, EPI_SILU, true) && gemm_bf16_blocked_handoff((int)rows, d, f, EPI_RESID, false) &&
                         (size_t)((rows + 31) / 32 * 32) * f * 2 <= w.hbuf.cap;
    g1.out_blocked = blocked ? 1 : 0;
    if (sg) { g1.a_sigma = 1; g1.W_sig = second ? sg->ffn2_w1 : sg->ffn1_w1; g1.sigma_cols = f; }
    // (the first FFN's norm rides on the previous block's final_norm_ kernel, see run_layers -- unless the product folds it in: ln_gemm)
    ln_gemm("ffn_fc1_silu", g1, EPI_SILU, second ? L.ffn2_ng : L.ffn1_ng, second ? L.ffn2_nb : L.ffn1_nb, norm_state, sg ? 2 : a16, x, n, rows, s);
    GemmArgs g2{h, f, second ? L.ffn2_w2 : L.ffn1_w2, f, second ? L.ffn2_b2 : L.ffn1_b2, x, d, x, d, 0.5f, (int)rows, d, f};
    g2.a_bf16 = a16;
    g2.a_blocked = blocked ? 1 : 0;
    if (sg) { g2.a_sigma = 1; g2.W_sig = second ? sg->ffn2_w2 : sg->ffn1_w2; }
    run_gemm("ffn_fc2_resid", g2, EPI_RESID, s);
}

// FastConformerEncoder::forward (src/encoder.cpp:253-271) -> w.x [B][T][d]
void Model::run_encoder(Workspace &w, const float *d_feats, int B, int Tm, int stop_layer, int stop_stage, hipStream_t s) {
    float *x = w.x.as<float>();
    run_subsample(w, d_feats, B, Tm, x, s);
    run_layers(w, B, 0, stop_layer, stop_stage, s);
}

// ConformerBlock::forward x (layers first_layer ..) on w.x [B][T][d]  (src/encoder.cpp:196-204, :267-269)
void Model::run_layers(Workspace &w, int B, int first_layer, int stop_layer, int stop_stage, hipStream_t s) {
    // a ragged batch (w.ragged; B ignored): packed rows; T = its longest utterance (position tables, LDS sizing)
    const bool rg = w.ragged;
    const int d = cfg.hidden_size, T = w.t_max();
    if (rg) B = w.rag.B;
    const int64_t rows = w.rows(B);
    float *x = w.x.as<float>(), *n = w.n.as<float>();
    if (stop_layer < 0 || stop_layer > cfg.num_layers) { stop_layer = cfg.num_layers; stop_stage = 0; }
    if (stop_layer == 0 && stop_stage == 0) return;
    float *att_scratch_p = nullptr;
    if (!attn_bf16(T) && relpos_attention_lds_bytes(T, d / cfg.num_heads) > 160 * 1024) {
        // a [32][T] score block no longer fits LDS (> ~85 s of audio): the same kernel with its score blocks in global scratch
        const size_t need = rg ? relpos_attention_scratch_bytes_units(w.rag.n_u_att, T, cfg.num_heads, d / cfg.num_heads)
                               : relpos_attention_scratch_bytes(B, T, cfg.num_heads, d / cfg.num_heads);
        if (need == 0 || need > ((size_t)64 << 30))
            fail(PK_ERR_UNSUPPORTED, "utterance of %d encoder frames (%.1f s) x %d clips: the attention scratch would need %.1f GB -- use fewer / shorter clips per call",
                 T, T * 0.08, B, need / 1e9);
        att_scratch.reserve(need);
        att_scratch_p = att_scratch.as<float>();
    }
    ensure_pos_tables(T, s);
    const PosTab &ptab = pos_tab(T);                                 // the resident set of this batch's attention format
    const int pos_T = ptab.T, P = 2 * pos_T - 1;                     // its rows (built for pos_T >= T frames)
    const DevBuf &pos_proj = ptab.proj, &pos_cvec = ptab.cvec;
    const int a16 = cfg.gemm_bf16 ? 1 : 0;                           // bf16 mode: LayerNorm outputs stored as bf16 GEMM operands (ffn())
    const bool att16 = attn_bf16(T);                                 // ... and q / k / v as bf16 for the bf16-MFMA attention kernel
    if (rg) {
        const int want = att16 ? relpos_attention_bf16_block_rows(d / cfg.num_heads) : 32;
        if (w.rag.att_rows != want) fail(PK_ERR_INVALID, "internal: the ragged batch's attention blocks hold %d rows, the kernel takes %d", w.rag.att_rows, want);
        w.rv.att.pos_T = pos_T;
    }
    const SeqRag no_rag;
    // Small batches (rows <= kSmallMRows: one clip, the reference's own benchmark protocol): every product is a latency-bound chain of
    // dependent MFMAs (kernels/gemm_smallm.hip).  They run on the tiled sigma-K weight copies with sigma-K activations -- written that way by
    // their producers (LayerNorm mode 2, sigma_cols of fc1, the attention context, the depthwise conv) -- so nothing but the MFMAs is on the
    // chain; the residual stream x stays natural.  Same arithmetic, same bits.
    const std::vector<SigW> *sigv = (!a16 && rows <= kSmallMRows) ? &sigma_weights() : nullptr;
    const bool sgm = sigv && !sigv->empty();
    const int ymode = sgm ? 2 : a16;                                 // LayerNorm / attention / conv output mode: 0 fp32, 1 bf16, 2 fp32 sigma
    int ffn1_norm_state = 0;                                         // (ln_gemm: 1 = n holds the next ffn1's normalised rows, 2 = their statistics)
    for (int l = first_layer; l < cfg.num_layers; ++l) {
        if (l > stop_layer || (l == stop_layer && stop_stage == 0)) break;
        const LayerW &L = layers[l];
        const int stage_cap = (l == stop_layer) ? stop_stage : 5;
        const SigW *sg = sgm ? &(*sigv)[l] : nullptr;
        ffn(w, L, false, rows, s, ffn1_norm_state, sg);                              // ffn1_  :197
        ffn1_norm_state = 0;
        if (stage_cap == 1) break;
        // ConformerAttention::forward  :180-186
        {
            // q and k columns in the sigma layout (MFMA operands of the attention kernel), v natural
            GemmArgs g{n, d, L.wqkv, d, L.bqkv, w.qkv.as<float>(), 3 * d, nullptr, 0, 1.0f, (int)rows, 3 * d, d};
            g.sigma_cols = att16 ? 0 : 2 * d;
            g.a_bf16 = a16;
            g.out_bf16 = att16 ? 1 : 0;
            if (sg) { g.a_sigma = 1; g.W_sig = sg->wqkv; }
            ln_gemm("attn_qkv", g, EPI_NONE, L.att_ng, L.att_nb, 0, ymode, x, n, rows, s);
        }
        const int hd = d / cfg.num_heads;
        double fl = 0.0;                                             // QK^T + QP^T (needed band) + AV over every (utterance, head)
        if (rg) for (int tb : w.rag.T) fl += (double)cfg.num_heads * (2.0 * tb * tb * hd * 2 + 2.0 * tb * tb * hd);
        else fl = (double)B * cfg.num_heads * (2.0 * T * T * hd * 2 + 2.0 * T * T * hd);
        if (att16) {
            KL("relpos_attention", fl, 0.0,
               launch_relpos_attention_bf16(w.qkv.p, B, T, d, cfg.num_heads, reinterpret_cast<const __bf16 *>(pos_proj.p) + (size_t)l * P * d,
                                            pos_cvec.as<float>() + (size_t)l * cfg.num_heads * P, L.pos_u, w.ctx.p, s, pos_T, rg ? w.rv.att : no_rag));
        } else {
            KL("relpos_attention", fl, 0.0,
               launch_relpos_attention(w.qkv.as<float>(), B, T, d, cfg.num_heads, pos_proj.as<float>() + (size_t)l * P * d, L.pos_u, L.pos_v,
                                       w.ctx.as<float>(), s, 0.0f, att_scratch_p, ymode, pos_T - T, rg ? w.rv.att : no_rag));
        }
        {
            GemmArgs g{w.ctx.as<float>(), d, L.wo, d, L.bo, x, d, x, d, 1.0f, (int)rows, d, d};
            g.a_bf16 = a16;
            if (sg) { g.a_sigma = 1; g.W_sig = sg->wo; }
            run_gemm("attn_out_resid", g, EPI_RESID, s);
        }
        if (stage_cap == 2) break;
        // ConformerConvModule::forward  :59-75
        {
            GemmArgs g{n, d, L.pw1_w, d, L.pw1_b, w.g.as<float>(), d, nullptr, 0, 1.0f, (int)rows, d, d};
            g.a_bf16 = a16;
            g.fast_act = a16;
            if (sg) { g.a_sigma = 1; g.W_sig = sg->pw1; }
            ln_gemm("conv_pw1_glu", g, EPI_GLU, L.cv_ng, L.cv_nb, 0, ymode, x, n, rows, s);
        }
        KL("dwconv_bn_silu", (double)rows * d * cfg.conv_kernel_size * 2.0, 2.0 * rows * d * 4,
           launch_dwconv_bn_silu(w.g.as<float>(), rg ? 1 : B, rg ? (int)rows : T, d, cfg.conv_kernel_size, L.dw_w, L.dw_b, L.bn_mean, L.bn_rstd, L.bn_g, L.bn_b,
                                 w.dwb.as<float>(), s, ymode, rg ? w.rv.dwc : no_rag));
        {
            GemmArgs g{w.dwb.as<float>(), d, L.pw2_w, d, L.pw2_b, x, d, x, d, 1.0f, (int)rows, d, d};
            g.a_bf16 = a16;
            if (sg) { g.a_sigma = 1; g.W_sig = sg->pw2; }
            run_gemm("conv_pw2_resid", g, EPI_RESID, s);
        }
        if (stage_cap == 3) break;
        ffn(w, L, true, rows, s, 0, sg);                                             // ffn2_  :201
        if (stage_cap == 4) break;
        const bool next_runs = l + 1 < cfg.num_layers && !(l + 1 > stop_layer || (l + 1 == stop_layer && stop_stage == 0));
        bool next_folds = false;   // the next block's fc1 folds its own norm in (small fp32 batches): final_norm_ alone here
        if (next_runs && sg) {
            GemmArgs pg{x, d, layers[l + 1].ffn1_w1, d, nullptr, w.hbuf.as<float>(), cfg.ffn_intermediate, nullptr, 0, 1.0f, (int)rows, cfg.ffn_intermediate, d};
            pg.W_sig = (*sigv)[l + 1].ffn1_w1; pg.ln_g = layers[l + 1].ffn1_ng; pg.ln_b = layers[l + 1].ffn1_nb;
            next_folds = ln_folds(pg, EPI_SILU, rows);
        }
        bool next_stats = false;   // large fp32 batches: the next block's fc1 normalises from row statistics -- final_norm_ written, its rows' statistics beside it
        if (next_runs && !next_folds && !sg) {
            GemmArgs pg{n, d, layers[l + 1].ffn1_w1, d, layers[l + 1].ffn1_b1, w.hbuf.as<float>(), cfg.ffn_intermediate, nullptr, 0, 1.0f, (int)rows, cfg.ffn_intermediate, d};
            next_stats = ln_stats_folds(pg, EPI_SILU, layers[l + 1].ffn1_ng, layers[l + 1].ffn1_nb, x, n);
        }
        if (next_runs && next_stats) {
            KL("layernorm_then_stats", 0.0, 2.0 * rows * d * 4, launch_layernorm_then_stats(x, rows, d, L.fin_g, L.fin_b, 1e-5f, x, n, s));
            ffn1_norm_state = 2;
        } else if (next_runs && !next_folds) {   // final_norm_ :202 and the next block's ffn1_ norm :40 in one pass over the rows
            KL("layernorm", 0.0, 3.0 * rows * d * 4,
               launch_layernorm2(x, rows, d, L.fin_g, L.fin_b, layers[l + 1].ffn1_ng, layers[l + 1].ffn1_nb, 1e-5f, x, n, s, ymode));
            ffn1_norm_state = 1;
        } else {
            KL("layernorm", 0.0, 2.0 * rows * d * 4, launch_layernorm(x, rows, d, L.fin_g, L.fin_b, 1e-5f, x, s));   // final_norm_ :202
        }
    }
}

// CTCDecoder::forward + ctc_greedy_decode(_with_timestamps)  (src/ctc.cpp:12-25, :40-127)
// ContextTrie::insert / build (src/phrase_boost.cpp:11-37) on the host, flattened to CSR for the decode kernels.
void Model::set_boost(const std::vector<std::vector<int>> &phrases, float score) {
    require_gpu();
    boost_phrases = phrases;
    boost_score = score;
    boost_on = !phrases.empty();
    trie_nodes = 0;
    if (!boost_on) return;
    std::vector<std::map<int, int>> kids(1);                       // node -> (token -> child), root = 0
    for (const auto &ph : phrases) {
        if ((int)ph.size() >= kTrieMaxActive)
            fail(PK_ERR_INVALID, "boost phrase of %d tokens: at most %d are supported", (int)ph.size(), kTrieMaxActive - 1);
        int node = 0;
        for (int tk : ph) {
            auto it = kids[node].find(tk);
            if (it == kids[node].end()) {
                const int next = (int)kids.size();
                kids[node][tk] = next;
                kids.emplace_back();
                node = next;
            } else {
                node = it->second;
            }
        }
    }
    trie_nodes = (int)kids.size();
    std::vector<int> off(kids.size() + 1, 0), tok, nd;
    for (size_t i = 0; i < kids.size(); ++i) {
        for (const auto &kv : kids[i]) { tok.push_back(kv.first); nd.push_back(kv.second); }
        off[i + 1] = (int)tok.size();
    }
    if (tok.empty()) { tok.push_back(-1); nd.push_back(0); }       // root-only trie: keep the buffers non-empty
    PK_HIP(hipStreamSynchronize(stream));                          // nothing in flight may still read the old trie
    if (stream_dec) PK_HIP(hipStreamSynchronize(stream_dec));
    trie_off.reserve(off.size() * 4); trie_tok.reserve(tok.size() * 4); trie_node.reserve(nd.size() * 4);
    PK_HIP(hipMemcpy(trie_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    PK_HIP(hipMemcpy(trie_tok.p, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));
    PK_HIP(hipMemcpy(trie_node.p, nd.data(), nd.size() * 4, hipMemcpyHostToDevice));
}
TrieDev Model::trie_dev(Workspace &w, int B) {
    TrieDev t{};
    if (!boost_on) return t;
    w.trie_act.reserve((size_t)B * (kTrieMaxActive + 1) * sizeof(int));
    t.off = trie_off.as<int>(); t.tok = trie_tok.as<int>(); t.node = trie_node.as<int>();
    t.n_nodes = trie_nodes; t.boost = boost_score;
    t.act = w.trie_act.as<int>(); t.n_act = t.act + (size_t)B * kTrieMaxActive;
    return t;
}

void Model::run_ctc(Workspace &w, const float *d_enc, int B, int T, bool want_logp, hipStream_t s) {
    if (cfg.ctc_vocab_size <= 0) fail(PK_ERR_UNSUPPORTED, "this model has no ctc_decoder_ head");
    const int V = cfg.ctc_vocab_size, d = cfg.hidden_size;
    // a ragged batch (w.ragged; B / T ignored): packed rows, per-utterance frame counts; the token arrays are [B][w.T] (the capacity pitch)
    const bool rg = w.ragged;
    if (rg) B = w.rag.B;
    const int64_t rows = rg ? w.rows(B) : (int64_t)B * T;
    const int pitch = (rg || w.T >= T) ? w.T : T;                   // token arrays [B][pitch]: the workspace's capacity (= T for a full uniform run)
    const SeqRag seq = rg ? w.rv.seq : SeqRag();
    gemm("ctc_head", d_enc, d, dec.ctc_w, d, dec.ctc_b, w.ctc_logits.as<float>(), V, (int)rows, V, d, EPI_NONE, nullptr, 0, 1.0f, s);
    if (boost_on) want_logp = true;                                 // the boosted argmax needs the whole log-prob rows
    if (want_logp) w.ctc_lp.reserve((size_t)rows * V * 4);
    KL("logsoftmax_argmax", 0.0, (double)rows * V * 4,
       launch_logsoftmax_argmax(w.ctc_logits.as<float>(), rows, V, V, want_logp ? w.ctc_lp.as<float>() : nullptr, w.best_idx.as<int>(), w.best_lp.as<float>(), s));
    // CTC token arrays are [B][T]; they share the TDT output buffers (sized >= B*T)
    if (boost_on) {
        KL("ctc_boosted", 0.0, (double)rows * V * 4,
           launch_ctc_boosted(w.ctc_lp.as<float>(), B, T, V, cfg.blank_id < V ? cfg.blank_id : V - 1, trie_dev(w, B), w.ids.as<int>(), w.lens.as<int>(),
                              w.start.as<int>(), w.end.as<int>(), w.conf.as<float>(), s, pitch, seq));
        return;
    }
    KL("ctc_collapse", 0.0, 0.0,
       launch_ctc_collapse(w.best_idx.as<int>(), w.best_lp.as<float>(), B, T, cfg.blank_id < V ? cfg.blank_id : V - 1, w.ids.as<int>(), w.lens.as<int>(),
                           w.start.as<int>(), w.end.as<int>(), w.conf.as<float>(), s, pitch, seq));
}

// The decode loop's poll ahead of the chunk's end (run_tdt_loop).  EXPERIMENTAL builds: PK_DEC_POLL_AHEAD=0 keeps the synchronising poll.
static bool poll_ahead_on() {
#ifdef PK_EXPERIMENTAL
    static const bool on = [] { const char *e = getenv("PK_DEC_POLL_AHEAD"); return e ? atoi(e) != 0 : true; }();
    return on;
#else
    return true;
#endif
}

// Frame window of the small-batch decode loop (TdtState::F): the largest lock-step batch that gets one.  Measured (profiles/r06_dec_window_ab.txt, pk_transcribe_pcm,
// median of 100 calls): one clip 4.96-5.00 -> 4.84 ms on the benchmark's clips (81 tokens in 112 decisions: a blank-poor case); two clips no change; four clips
// + 4 % (the walked decisions of one utterance hold up the step of the other three, and a window of four frames ends at the first long blank) -> single
// utterances only.  EXPERIMENTAL builds: PK_DEC_WIN=<largest batch> (0: off).
static int decode_window_rows() {
#ifdef PK_EXPERIMENTAL
    static const int n = [] { const char *e = getenv("PK_DEC_WIN"); return e ? atoi(e) : 1; }();
    return n;
#else
    return 1;
#endif
}

// tdt_greedy_decode(_with_timestamps) / rnnt_greedy_decode  (src/tdt.cpp:36-201, src/rnnt.cpp:56-177)
// enc_proj_ hoisted out of the symbol loop: one GEMM over all frames (the reference recomputes it per symbol, src/tdt.cpp:17)
void Model::run_enc_proj(const float *d_enc, int64_t rows, float *ep_out, hipStream_t s) {
    const int d = cfg.hidden_size, J = cfg.joint_hidden;
    if (cfg.vocab_size <= 0) fail(PK_ERR_UNSUPPORTED, "this model has no prediction net / joint (encoder-only configuration)");
    gemm("joint_enc_proj", d_enc, d, dec.we, d, dec.be, ep_out, J, (int)rows, J, d, EPI_NONE, nullptr, 0, 1.0f, s);
}

void Model::run_tdt(Workspace &w, const float *d_enc, int B, int T, int max_tokens, hipStream_t s, bool keep_state) {
    if (w.ragged) { B = w.rag.B; T = w.rag.T_max; }                  // ragged batch: packed rows; T bounds the loop, w.dec_Tb / dec_row0 give the extents
    run_enc_proj(d_enc, w.ragged ? w.rows(B) : (int64_t)B * T, w.ep.as<float>(), s);
    run_tdt_loop(w, B, T, max_tokens, s, keep_state);
}

// The greedy loop over w.ep = enc_proj of B utterances of T frames (rows b*T + t).  B may span several batches of the pipelined path
// (capi.cpp: decode groups): the utterances are independent, a lock-step batch of 2B costs the same number of launches as one of B.
void Model::run_tdt_loop(Workspace &w, int B, int T, int max_tokens, hipStream_t s, bool keep_state) {
    w.poll_hit = false;
    const int Hp = cfg.pred_hidden, J = cfg.joint_hidden, V = cfg.vocab_size, D = cfg.rnnt_head ? 0 : cfg.num_durations;
    const int L = cfg.num_lstm_layers;
    if (V <= 0) fail(PK_ERR_UNSUPPORTED, "this model has no prediction net / joint (encoder-only configuration)");
    TdtState st{};
    st.B = B; st.T = T; st.V = V; st.D = D; st.L = L; st.Hp = Hp; st.blank = cfg.blank_id; st.max_symbols = cfg.max_symbols_per_step;
    st.max_tokens = max_tokens;
    st.keep_state = keep_state ? 1 : 0;
    st.Tb = w.dec_Tb; st.row0 = w.dec_row0;                          // ragged batch / decode group of ragged runs (null: uniform, T frames each)
    if (w.force_label) {                                             // pk_tdt_score: walk a given decision path, record the joint's outputs
        const bool batched = w.force_stride > 0 && w.n_force_b;          // pk_stream_score: every stream of a lock-step chunk, state carried
        if (boost_on || D <= 0 || (!batched && (B != 1 || keep_state)))
            fail(PK_ERR_UNSUPPORTED, "teacher-forced scoring takes one utterance (or the streams of one chunk) of a TDT model, unboosted");
        st.force_label = w.force_label; st.force_dur = w.force_dur; st.n_force = w.n_force;
        st.score_lab = w.score_lab; st.score_dur = w.score_dur;
        if (batched) { st.n_force_b = w.n_force_b; st.force_stride = w.force_stride; }
    }
    if (boost_on) {
        if (cfg.rnnt_head || keep_state) fail(PK_ERR_UNSUPPORTED, "phrase boosting applies to the CTC and TDT greedy decoders only (src/phrase_boost.cpp)");
        st.trie = trie_dev(w, B);
    }
    st.max_steps = T * (cfg.max_symbols_per_step + 1) + 16;          // safety cap (the reference has none)
    if (w.force_label && w.n_force + 1 > st.max_steps) st.max_steps = w.n_force + 1;
    for (int i = 0; i < D; ++i) st.durations[i] = cfg.durations[i];
    st.logits = w.logits.as<float>();
    st.h = w.h.as<float>(); st.c = w.c.as<float>(); st.hn = w.hn.as<float>(); st.cn = w.cn.as<float>();
    int *ib = w.ints.as<int>();
    st.token = ib; st.t = ib + B; st.nsym = ib + 2 * B; st.n_out = ib + 3 * B; st.steps = ib + 4 * B; st.done = ib + 5 * B;
    st.done_count = ib + 6 * B;
    st.lens = w.lens.as<int>();
    st.ids = w.ids.as<int>(); st.start = w.start.as<int>(); st.end = w.end.as<int>(); st.conf = w.conf.as<float>();
    st.margin = (boost_on || keep_state) ? nullptr : w.margin.as<float>();
    // Prediction-net caching (kernels.hpp TdtState::need): after a blank the cells and pred_proj of the next step would recompute, bit for bit,
    // what they produced the step before -- those rows are skipped (~2/3 of all utterance-steps on the benchmark's clips).  Per-phase loop only
    // (the single-launch loop keeps every row), lock-step batches up to kMaxListRows.
    bool pred_cache = B <= kMaxListRows && decode_loop != PK_DECODE_LOOP_PERSISTENT;
#ifdef PK_EXPERIMENTAL
    { static const bool off = [] { const char *e = getenv("PK_DEC_NOCACHE"); return e && atoi(e) != 0; }(); if (off) pred_cache = false; }
#endif
    if (pred_cache) {
        st.need = ib + 6 * B + 8;                                    // (behind done_count and the persistent loop's two spare words)
        st.pp = w.pp.as<float>();
        st.ep = w.ep.as<float>();
        st.J = J;
        // Frame window (TdtState::F): at B <= 8 the heads product's 16-row tile has room for the joint of the next frames under the unchanged prediction-net
        // state, and the decision kernel walks through the blanks inside it.  Offline greedy decode of the exact fp32 phases only.
        const bool dec16_ = cfg.gemm_bf16 && Hp % 32 == 0 && J % 32 == 0 && wld16;
        if (B <= decode_window_rows() && !keep_state && !boost_on && !w.force_label && !dec16_ && !dec_nt_weights && J <= 1024 && V + D <= 5 * 256)
            st.F = B <= 2 ? 8 : B <= 4 ? 4 : B <= 8 ? 2 : 1;
        w.z.reserve((size_t)B * st.F * J * sizeof(float));
        w.logits.reserve((size_t)B * st.F * (V + D) * sizeof(float));
        st.logits = w.logits.as<float>();
        st.z = w.z.as<float>();
    }
    if (!keep_state) {                                               // a streaming chunk continues from the carried LSTM state
        PK_HIP(hipMemsetAsync(w.h.p, 0, (size_t)L * B * Hp * 4, s));
        PK_HIP(hipMemsetAsync(w.c.p, 0, (size_t)L * B * Hp * 4, s));
    }
    launch_tdt_init(st, s);
    // Per step: [cell GEMV per LSTM layer] -> joint-activation GEMV -> heads GEMV -> decide.  h / h' / z live in the sigma
    // K layout (they are only ever GEMV operands); c, the g1 table, enc_proj and the logits are in natural order.
    const double f_hh = 2.0 * B * 4 * Hp * Hp, f_pp = 2.0 * B * J * Hp, f_hd = 2.0 * B * (V + D) * J;
    // the step-invariant arguments of every phase
    TdtPersist P{};
    P.st = st;
    P.L = L;
    for (int l = 0; l < L; ++l) {
        float *hl = w.h.as<float>() + (size_t)l * B * Hp, *cl = w.c.as<float>() + (size_t)l * B * Hp;
        float *hnl = w.hn.as<float>() + (size_t)l * B * Hp, *cnl = w.cn.as<float>() + (size_t)l * B * Hp;
        SkinnyArgs &a = P.cell[l];
        a.X = hl; a.W = dec_whh_s[l]; a.B = B; a.N = 4 * Hp; a.K = Hp; a.out = hnl; a.c = cl; a.cn = cnl; a.Hp = Hp;
        if (l == 0) {
            a.gi = dec.g1; a.gi_ld = 4 * Hp; a.gi_row = st.token;
        } else {
            // upper layer: the input projection W_ih h'(l-1) + b_ih runs inside the cell kernel (second chain of the same tile)
            a.X2 = w.hn.as<float>() + (size_t)(l - 1) * B * Hp; a.W2 = dec_wih_s[l]; a.bias2 = dec.bih[l];
            a.gi = nullptr; a.gi_ld = 4 * Hp; a.gi_row = nullptr;
        }
    }
    {
        SkinnyArgs &a = P.act;
        a.X = w.hn.as<float>() + (size_t)(L - 1) * B * Hp; a.W = dec_wp_s; a.B = B; a.N = J; a.K = Hp; a.bias = dec.bp;
        a.out = w.z.as<float>(); a.ep = w.ep.as<float>(); a.t = st.t; a.T = T; a.Tb = st.Tb; a.row0 = st.row0;
        a.F = st.F;
    }
    {
        SkinnyArgs &a = P.heads;
        a.X = w.z.as<float>(); a.W = wld_s; a.B = B * st.F; a.N = V + D; a.K = J; a.bias = bld; a.out = w.logits.as<float>(); a.ldo = V + D;   // (rows b * F + f)
    }
    for (int l = 0; l < L; ++l) P.cell[l].nt_weights = dec_nt_weights;
    P.act.nt_weights = P.heads.nt_weights = dec_nt_weights;
    if (pred_cache) {
        for (int l = 0; l < L; ++l) P.cell[l].need = st.need;
        P.act.need = st.need;
        P.act.pp_out = w.pp.as<float>();
    }
    // Tolerance-class mode: the same phases on bf16 operands (decode_gemv_bf16.hip).  h / h' / z are bf16 arrays in the same buffers (natural k
    // order), the weights are the bf16 copies; the layer-0 input projection stays the fp32 table g1.  Specification: the oracle's gemm_bf16 mode.
    const bool dec16 = cfg.gemm_bf16 && Hp % 32 == 0 && J % 32 == 0 && wld16;
    if (dec16) {
        st.h_bf16 = 1;
        P.st.h_bf16 = 1;
        __bf16 *h16 = reinterpret_cast<__bf16 *>(w.h.p), *hn16 = reinterpret_cast<__bf16 *>(w.hn.p);
        for (int l = 0; l < L; ++l) {
            SkinnyArgs &a = P.cell[l];
            a.X = reinterpret_cast<const float *>(h16 + (size_t)l * B * Hp);
            a.W = dec_whh16[l];
            a.out = reinterpret_cast<float *>(hn16 + (size_t)l * B * Hp);
            if (l > 0) {
                a.X2 = reinterpret_cast<const float *>(hn16 + (size_t)(l - 1) * B * Hp);
                a.W2 = dec_wih16[l];
            }
        }
        P.act.X = reinterpret_cast<const float *>(hn16 + (size_t)(L - 1) * B * Hp);
        P.act.W = dec_wp16;
        P.heads.W = wld16;
    }
    // ONE launch for the whole loop (kernels/decode_persist.hip): implemented, bit-identical to the per-phase loop (tests/test_gpu_decode.py),
    // and NOT faster on this hardware -- opt-in: pk_model_set_decode_loop(m, PK_DECODE_LOOP_PERSISTENT).  Measured in round 2 (profiles/r02_decode_persistent.md): the
    // grid barrier between the four all-to-all phases of a step costs ~20 us with a single system-scope arrival counter (160 arrivals
    // serialise at the memory side), 98 us per step against 46 us for four launches; next to the encoder of the following batch the
    // resident part of the grid spins while the rest waits for CU slots.  Eligible when the decide scratch is small enough for a
    // workgroup to sit beside the encoder's GEMM workgroups, at most two LSTM layers, no phrase boosting, no carried streaming state.
    {
        const bool want = decode_loop == PK_DECODE_LOOP_PERSISTENT && !dec16 && !w.force_label;     // (the single-launch loop exists for the fp32 phases only)
        const int G = Hp / 4;
        if (want && !boost_on && !keep_state && Hp % 4 == 0 && G >= 1 && G <= 200 && L <= 2 && tdt_persistent_lds_bytes(st) <= 12 * 1024) {
            w.persist_bar.reserve((size_t)kTdtBarrierWords * sizeof(unsigned));
            P.bar = w.persist_bar.as<unsigned>();
            P.abort = st.done_count + 2;                                       // (a spare word behind the per-utterance state)
            P.timeout_ticks = 200000000LL;                                     // 2 s of the 100 MHz wall clock
            PK_HIP(hipMemsetAsync(P.bar, 0, (size_t)kTdtBarrierWords * sizeof(unsigned), s));
            PK_HIP(hipMemsetAsync(P.abort, 0, sizeof(int), s));
            KL("tdt_persistent", (f_hh * L + f_pp + f_hd) * (T + 8), 0.0, launch_tdt_persistent(P, s));
            return;
        }
    }
    // host polls "all finished" every `chunk` steps: 16 for whole utterances (4 / 64 / 128 measured: no difference, profiles/r02_decode_persistent.md);
    // a streaming chunk of 1-3 frames is done after T + (symbols of its busiest stream) steps -- every step launched beyond that is four no-op
    // kernels of ~6 us each, more than the poll costs
    int chunk = T + 2 < 4 ? 4 : (T + 2 < 16 ? T + 2 : 16);
#ifdef PK_EXPERIMENTAL
    { static const int c = [] { const char *e = getenv("PK_DEC_CHUNK"); return e ? atoi(e) : 0; }(); if (c > 0) chunk = c; }
#endif
    auto skinny = [&](const SkinnyArgs &a, int epi) { if (dec16) launch_skinny_gemm_bf16(a, epi, s); else launch_skinny_gemm(a, epi, s); };
    auto enqueue_step = [&]() {
        for (int l = 0; l < L; ++l) {
            KL("lstm_hh_cell", l ? 2.0 * f_hh : f_hh, 0.0, skinny(P.cell[l], SK_CELL));
        }
        KL("joint_pred_act", f_pp, 0.0, skinny(P.act, SK_ACT));
        KL("joint_heads_gemv", f_hd, 0.0, skinny(P.heads, SK_BIAS));
        KL("tdt_decide", 0.0, 0.0, launch_tdt_decide(st, s));
    };
    // PK_DECODE_LOOP_GRAPH: the chunk of 16 steps is captured once into a hipGraph (every argument is step-invariant) and replayed.
    // (teacher-forced scoring launches another decision kernel on per-call arrays: it never shares a captured graph with the decode loop)
    const bool want_graph = decode_loop == PK_DECODE_LOOP_GRAPH && !w.force_label;
    if (want_graph && !prof) {
        // key = the pointer / size arguments the captured launches carry, field by field (no struct padding in the comparison) + the model's
        // weight generation (a graph never outlives the weights it was captured against: the workspace is keyed by model + stream)
        std::vector<unsigned char> key;
        auto put = [&key](const void *p, size_t n) { const unsigned char *c = static_cast<const unsigned char *>(p); key.insert(key.end(), c, c + n); };
        const void *ptrs[] = {this, s, st.logits, st.h, st.c, st.hn, st.cn, st.token, st.lens, st.ids, st.start, st.end, st.conf, P.act.ep, P.heads.W, P.act.W, P.cell[0].W, P.cell[0].gi, st.Tb, st.row0};
        const int ints[] = {B, T, V, D, L, Hp, J, max_tokens, st.blank, st.max_symbols, st.max_steps, st.keep_state, boost_on ? 1 : 0, pred_cache ? 1 : 0, st.F};
        put(ptrs, sizeof ptrs);
        put(ints, sizeof ints);
        if (!w.dec_graph || key != w.dec_graph_key) {
            if (w.dec_graph) { (void)hipGraphExecDestroy(w.dec_graph); w.dec_graph = nullptr; }
            hipGraph_t graph = nullptr;
            PK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < chunk; ++i) enqueue_step();
            PK_HIP(hipStreamEndCapture(s, &graph));
            PK_HIP(hipGraphInstantiate(&w.dec_graph, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            w.dec_graph_key = key;
        }
        for (int step = 0; step < st.max_steps; step += chunk) {
            PK_HIP(hipGraphLaunch(w.dec_graph, s));
            if (w.before_poll) w.before_poll(s);
            PK_HIP(hipMemcpyAsync(h_done, st.done_count, sizeof(int), hipMemcpyDeviceToHost, s));
            PK_HIP(hipStreamSynchronize(s));
            if (*h_done >= B) { w.poll_hit = true; break; }
        }
        return;
    }
    // Offline loops: the poll's copy is enqueued `ahead` steps before the end of the chunk and the host waits for THAT copy (an event), not for the stream -- the
    // steps behind it keep the GPU busy while the host wakes up and enqueues the next chunk (a wake-up measured at 20-24 us per poll in the single-clip
    // trace, profiles/r06_single_clip_kernel_stats.md; if the copy says "finished", the steps behind it were two no-op steps).  A streaming chunk's loop
    // (before_poll set: its result copies ride on the poll and must be the last thing enqueued) keeps the plain form.
    const int ahead = (!w.before_poll && !keep_state && chunk >= 8 && poll_ahead_on()) ? 2 : 0;
    if (ahead && !w.poll_ev) PK_HIP(hipEventCreateWithFlags(&w.poll_ev, hipEventDisableTiming));
    bool copy_pending = false;                                         // a poll's copy into h_done is enqueued and has not been waited for
    for (int step = 0; step < st.max_steps; ++step) {
        enqueue_step();
        const int k = (step + 1) % chunk;
        if (ahead && k == chunk - ahead) {
            PK_HIP(hipMemcpyAsync(h_done, st.done_count, sizeof(int), hipMemcpyDeviceToHost, s));
            PK_HIP(hipEventRecord(w.poll_ev, s));
            copy_pending = true;
        }
        if (k == 0) {                                                  // poll "all finished" once per chunk of steps
            if (ahead) {
                PK_HIP(hipEventSynchronize(w.poll_ev));
                copy_pending = false;
            } else {
                if (w.before_poll) w.before_poll(s);
                PK_HIP(hipMemcpyAsync(h_done, st.done_count, sizeof(int), hipMemcpyDeviceToHost, s));
                PK_HIP(hipStreamSynchronize(s));
            }
            if (*h_done >= B) { w.poll_hit = ahead == 0; break; }
        }
    }
    if (copy_pending) PK_HIP(hipEventSynchronize(w.poll_ev));           // (the step cap ended the loop between a copy and its wait: h_done is the model's one word -- no copy may outlive the loop)
}

}  // namespace pk
