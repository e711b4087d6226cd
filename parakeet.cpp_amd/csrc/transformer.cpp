// parakeet.cpp_amd/csrc/transformer.cpp -- TransformerEncoder / TransformerBlock of the reference (src/transformer.cpp:15-88,
// include/parakeet/transformer.hpp:12-21) on the same gfx950 kernels as the Conformer path: LayerNorm, the fp32-MFMA GEMM
// (bias / ReLU / residual epilogues) and the attention kernel in its "no position term" mode (scale applied to Q K^T).
// North-star file src/transformer.cpp; in the reference only Sortformer (diarization, out of scope) instantiates it, so it is
// exposed as a stage entry point of its own (pk_transformer_*), bit-identical to the oracle's orc_transformer_encoder.
//
// Heads narrower than the MFMA k-block (Sortformer: hidden 192 / 8 heads = 24) are zero-padded to a multiple of 32 at upload:
// the padded q / k / v output rows have zero weights and zero bias, the matching out_proj columns are zero, so every padded
// term is an exact fma(0, w, acc) == acc at the END of (or interleaved in) the reference's k-ordered chain: same bits.
#include <cmath>
#include <cstring>

#include "transformer.hpp"

namespace pk {

const float *TransformerEncoder::upload(const float *h, size_t n) {
    void *p = nullptr;
    PK_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(float)));
    allocs_.push_back(p);
    PK_HIP(hipMemcpy(p, h, n * sizeof(float), hipMemcpyHostToDevice));
    return static_cast<const float *>(p);
}

TransformerEncoder::TransformerEncoder(const std::string &weights_path, const std::string &prefix, const pk_transformer_config &c, int device)
    : cfg(c) {
    const int d = c.hidden_size, H = c.num_heads, f = c.ffn_intermediate;
    if (d <= 0 || H <= 0 || d % H || c.num_layers <= 0 || f <= 0) fail(PK_ERR_INVALID, "bad transformer config");
    if (d % 32 || f % 32) fail(PK_ERR_UNSUPPORTED, "hidden_size / ffn_intermediate must be multiples of 32 (MFMA K tile)");
    const int hd = d / H;
    hdp_ = (hd + 31) / 32 * 32;
    if (hdp_ > 128) fail(PK_ERR_UNSUPPORTED, "head_dim > 128");
    dp_ = H * hdp_;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(PK_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU path)");
    if (device < 0 || device >= n) fail(PK_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, n);
    PK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PK_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) fail(PK_ERR_NO_DEVICE, "device %d is %s; gfx950 (MI355X) code only", device, prop.gcnArchName);
    device_ = device;
    PK_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));

    SafeTensors st(weights_path);
    auto get = [&](const std::string &name, int64_t want) -> const float * {
        const HostTensor *t = st.find(prefix + name);
        if (!t) fail(PK_ERR_WEIGHTS, "missing tensor '%s%s'", prefix.c_str(), name.c_str());
        if (t->dtype != "F32" || t->numel() != want) fail(PK_ERR_WEIGHTS, "tensor '%s%s': expected %lld F32 elements", prefix.c_str(), name.c_str(), (long long)want);
        return t->f32();
    };
    layers_.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
        const std::string q = "layers_." + std::to_string(l) + ".";
        TransformerLayerW &L = layers_[l];
        L.n1g = upload(get(q + "norm1_.weight", d), d); L.n1b = upload(get(q + "norm1_.bias", d), d);
        L.n2g = upload(get(q + "norm2_.weight", d), d); L.n2b = upload(get(q + "norm2_.bias", d), d);
        // stacked q/k/v projection with every head padded from hd to hdp_ output rows
        std::vector<float> w((size_t)3 * dp_ * d, 0.0f), b((size_t)3 * dp_, 0.0f);
        const char *nm[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            const float *wj = get(q + "mha_." + nm[j] + ".weight", (int64_t)d * d), *bj = get(q + "mha_." + nm[j] + ".bias", d);
            for (int h = 0; h < H; ++h)
                for (int e = 0; e < hd; ++e) {
                    memcpy(&w[((size_t)j * dp_ + h * hdp_ + e) * d], &wj[(size_t)(h * hd + e) * d], (size_t)d * 4);
                    b[(size_t)j * dp_ + h * hdp_ + e] = bj[h * hd + e];
                }
        }
        L.wqkv = upload(w.data(), w.size()); L.bqkv = upload(b.data(), b.size());
        // out_proj [d][d] -> [d][dp_] with zero columns at the padded positions
        const float *wo = get(q + "mha_.out_proj.weight", (int64_t)d * d);
        std::vector<float> wop((size_t)d * dp_, 0.0f);
        for (int r = 0; r < d; ++r)
            for (int h = 0; h < H; ++h) memcpy(&wop[(size_t)r * dp_ + h * hdp_], &wo[(size_t)r * d + h * hd], (size_t)hd * 4);
        L.wo = upload(wop.data(), wop.size()); L.bo = upload(get(q + "mha_.out_proj.bias", d), d);
        L.w1 = upload(get(q + "fc1_.weight", (int64_t)f * d), (size_t)f * d); L.b1 = upload(get(q + "fc1_.bias", f), f);
        L.w2 = upload(get(q + "fc2_.weight", (int64_t)d * f), (size_t)d * f); L.b2 = upload(get(q + "fc2_.bias", d), d);
    }
    if (c.has_final_norm) { fin_g_ = upload(get("final_norm_.weight", d), d); fin_b_ = upload(get("final_norm_.bias", d), d); }
}

TransformerEncoder::~TransformerEncoder() {
    if (device_ >= 0) {
        (void)hipSetDevice(device_);
        for (void *p : allocs_) (void)hipFree(p);
        if (stream_) (void)hipStreamDestroy(stream_);
    }
}

void TransformerEncoder::forward(const float *x_host, int B, int T, float *y_host) {
    PK_HIP(hipSetDevice(device_));
    const int64_t rows = (int64_t)B * T;
    const int d = cfg.hidden_size;
    x_.reserve(rows * d * 4);
    PK_HIP(hipMemcpyAsync(x_.p, x_host, rows * d * 4, hipMemcpyHostToDevice, stream_));
    forward_dev(x_.as<float>(), B, T, stream_);
    PK_CHECK_LAUNCH();
    PK_HIP(hipMemcpyAsync(y_host, x_.p, rows * d * 4, hipMemcpyDeviceToHost, stream_));
    PK_HIP(hipStreamSynchronize(stream_));
}

void TransformerEncoder::forward_dev(float *x, int B, int T, hipStream_t s) {
    const int d = cfg.hidden_size, H = cfg.num_heads, f = cfg.ffn_intermediate, hd = d / H;
    const int64_t rows = (int64_t)B * T;
    const float eps = cfg.layer_norm_eps > 0.0f ? cfg.layer_norm_eps : 1e-5f;
    const float scale = 1.0f / sqrtf((float)hd);                              // src/transformer.cpp:27 (the REAL head dim)
    float *scratch = nullptr;
    if (relpos_attention_lds_bytes(T, hdp_) > 160 * 1024) {          // long sequence: score blocks in global scratch (attention.hip)
        const size_t need = relpos_attention_scratch_bytes(B, T, H, hdp_);
        if (need == 0 || need > ((size_t)64 << 30)) fail(PK_ERR_UNSUPPORTED, "sequence of %d frames x %d: attention scratch of %.1f GB", T, B, need / 1e9);
        att_scratch_.reserve(need);
        scratch = att_scratch_.as<float>();
    }
    n_.reserve(rows * d * 4); qkv_.reserve(rows * 3 * dp_ * 4); ctx_.reserve(rows * dp_ * 4); h_.reserve(rows * f * 4);
    float *n = n_.as<float>();
    for (const TransformerLayerW &L : layers_) {
        const float *in = x;
        if (cfg.pre_ln) { launch_layernorm(x, rows, d, L.n1g, L.n1b, eps, n, s); in = n; }        // :18
        {   // q, k, v (:20-22); q and k in the sigma column layout the attention kernel loads its MFMA operands in
            GemmArgs g{in, d, L.wqkv, d, L.bqkv, qkv_.as<float>(), 3 * dp_, nullptr, 0, 1.0f, (int)rows, 3 * dp_, d};
            g.sigma_cols = 2 * dp_;
            launch_gemm(g, EPI_NONE, s);
        }
        launch_relpos_attention(qkv_.as<float>(), B, T, dp_, H, nullptr, nullptr, nullptr, ctx_.as<float>(), s, scale, scratch);   // :38-45
        {   // out_proj + residual (:49-51)
            GemmArgs g{ctx_.as<float>(), dp_, L.wo, dp_, L.bo, x, d, x, d, 1.0f, (int)rows, d, dp_};
            launch_gemm(g, EPI_RESID, s);
        }
        if (!cfg.pre_ln) launch_layernorm(x, rows, d, L.n1g, L.n1b, eps, x, s);
        in = x;
        if (cfg.pre_ln) { launch_layernorm(x, rows, d, L.n2g, L.n2b, eps, n, s); in = n; }        // :54
        {   // fc1 + ReLU (:55-56)
            GemmArgs g{in, d, L.w1, d, L.b1, h_.as<float>(), f, nullptr, 0, 1.0f, (int)rows, f, d};
            launch_gemm(g, EPI_RELU, s);
        }
        {   // fc2 + residual (:58-61)
            GemmArgs g{h_.as<float>(), f, L.w2, f, L.b2, x, d, x, d, 1.0f, (int)rows, d, f};
            launch_gemm(g, EPI_RESID, s);
        }
        if (!cfg.pre_ln) launch_layernorm(x, rows, d, L.n2g, L.n2b, eps, x, s);
    }
    if (cfg.has_final_norm) launch_layernorm(x, rows, d, fin_g_, fin_b_, eps, x, s);              // :84-86
}

}  // namespace pk

using namespace pk;

struct pk_transformer {
    std::unique_ptr<TransformerEncoder> t;
};

extern "C" {

pk_status pk_transformer_load(const char *safetensors_path, const char *prefix, const pk_transformer_config *cfg, int device,
                              pk_transformer **out) {
    try {
        if (!safetensors_path || !cfg || !out) fail(PK_ERR_INVALID, "invalid argument: path/cfg/out");
        auto h = std::make_unique<pk_transformer>();
        h->t = std::make_unique<TransformerEncoder>(safetensors_path, prefix ? prefix : "", *cfg, device);
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

pk_status pk_transformer_forward(pk_transformer *t, const float *x, int B, int T, float *y) {
    try {
        if (!t || !x || !y || B <= 0 || T <= 0) fail(PK_ERR_INVALID, "invalid argument: transformer/x/y/B/T");
        t->t->forward(x, B, T, y);
        return PK_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return PK_ERR_INVALID;
    }
}

void pk_transformer_free(pk_transformer *t) { delete t; }

}  // extern "C"
