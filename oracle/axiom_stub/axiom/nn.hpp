// oracle/axiom_stub/axiom/nn.hpp -- TEST INFRASTRUCTURE ONLY.  See axiom.hpp in this directory for what this stand-in is.
// Module tree with name-based load_state_dict (the names the reference registers through AX_REGISTER_* ARE the on-disk weight
// names: `encoder_.layers_.3.attn_.mha_.q_proj.weight`, ... -- scripts/convert_nemo.py:98-310), and the layer types the
// reference instantiates (include/parakeet/{encoder,lstm,rnnt,ctc,transformer}.hpp).  Layers are shape-less until loaded, like
// the reference's (`Linear fc1_(true)`): every dimension comes from the weight tensors.
#pragma once
#include <array>
#include <sstream>

#include "axiom.hpp"

namespace axiom::nn {

class Module {
  public:
    Module() = default;
    Module(const Module &) = delete;             // children / parameters are registered by address
    Module &operator=(const Module &) = delete;
    virtual ~Module() = default;

    void register_module(const std::string &name, Module &m) { children_.emplace_back(name, &m); }
    void register_parameter(const std::string &name, Tensor &t) { params_.emplace_back(name, &t); }

    // Non-strict loading ignores missing and unexpected keys, as the reference asks for (`load_state_dict(w, "", false)`,
    // transcribe.hpp:63); the two lists are kept so that tests can look at them.
    template <class Map> void load_state_dict(const Map &sd, const std::string &prefix = "", bool strict = true) {
        missing_.clear();
        std::vector<std::string> seen;
        load_rec(sd, prefix, seen);
        unexpected_.clear();
        for (const auto &kv : sd) {
            if (kv.first.compare(0, prefix.size(), prefix) != 0) continue;
            if (std::find(seen.begin(), seen.end(), kv.first) == seen.end()) unexpected_.push_back(kv.first);
        }
        if (strict && (!missing_.empty() || !unexpected_.empty()))
            throw std::runtime_error("axiom stand-in: load_state_dict(strict): " + std::to_string(missing_.size()) + " missing, " +
                                     std::to_string(unexpected_.size()) + " unexpected keys");
    }
    const std::vector<std::string> &missing_keys() const { return missing_; }
    const std::vector<std::string> &unexpected_keys() const { return unexpected_; }

    Module &to(Device) { return *this; }
    void eval() {}

  protected:
    template <class Map> void load_rec(const Map &sd, const std::string &prefix, std::vector<std::string> &seen) {
        for (auto &p : params_) {
            auto it = sd.find(prefix + p.first);
            if (it == sd.end()) {
                missing_root().push_back(prefix + p.first);
                continue;
            }
            *p.second = it->second;
            seen.push_back(it->first);
        }
        for (auto &c : children_) {
            c.second->root_ = root_ ? root_ : this;
            c.second->load_rec(sd, prefix + c.first + ".", seen);
        }
        on_loaded();
    }
    virtual void on_loaded() {}
    std::vector<std::string> &missing_root() { return (root_ ? root_ : this)->missing_; }

    std::vector<std::pair<std::string, Module *>> children_;
    std::vector<std::pair<std::string, Tensor *>> params_;
    std::vector<std::string> missing_, unexpected_;
    Module *root_ = nullptr;
};

namespace detail {
inline std::vector<std::string> split_names(const char *s) {
    std::vector<std::string> out;
    std::string cur;
    for (const char *p = s; *p; ++p) {
        if (*p == ',') {
            out.push_back(cur);
            cur.clear();
        } else if (*p != ' ' && *p != '\t' && *p != '\n') {
            cur.push_back(*p);
        }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}
template <class... Ms> void reg_modules(Module &self, const char *names, Ms &...ms) {
    auto v = split_names(names);
    size_t i = 0;
    (self.register_module(v[i++], ms), ...);
}
template <class... Ts> void reg_params(Module &self, const char *names, Ts &...ts) {
    auto v = split_names(names);
    size_t i = 0;
    (self.register_parameter(v[i++], ts), ...);
}
}  // namespace detail

#define AX_REGISTER_MODULES(...) ::axiom::nn::detail::reg_modules(*this, #__VA_ARGS__, __VA_ARGS__)
#define AX_REGISTER_MODULE(m) ::axiom::nn::detail::reg_modules(*this, #m, m)
#define AX_REGISTER_PARAMETERS(...) ::axiom::nn::detail::reg_params(*this, #__VA_ARGS__, __VA_ARGS__)
#define AX_REGISTER_PARAMETER(p) ::axiom::nn::detail::reg_params(*this, #p, p)

// ───────────────────────────── ModuleList ─────────────────────────────
class ModuleList : public Module {
  public:
    template <class T, class... Args> T &emplace_back(Args &&...args) {
        auto p = std::make_unique<T>(std::forward<Args>(args)...);
        T &ref = *p;
        names_.push_back(std::to_string(items_.size()));
        items_.push_back(std::move(p));
        register_module(names_.back(), ref);
        return ref;
    }
    size_t size() const { return items_.size(); }

    template <class T> struct Range {
        const std::vector<std::unique_ptr<Module>> *v;
        struct It {
            typename std::vector<std::unique_ptr<Module>>::const_iterator i;
            T &operator*() const { return static_cast<T &>(**i); }
            It &operator++() {
                ++i;
                return *this;
            }
            bool operator!=(const It &o) const { return i != o.i; }
        };
        It begin() const { return {v->begin()}; }
        It end() const { return {v->end()}; }
    };
    template <class T> Range<const T> each() const { return {&items_}; }
    template <class T> Range<T> each() { return {&items_}; }

  private:
    std::vector<std::unique_ptr<Module>> items_;
    std::vector<std::string> names_;
};

// ───────────────────────────── layers ─────────────────────────────
class Linear : public Module {
  public:
    explicit Linear(bool bias = true) : has_bias_(bias) {
        register_parameter("weight", weight_);
        if (bias) register_parameter("bias", bias_);
    }
    // y[.., n] = (sum_k x[.., k] * W[n][k]) + b[n] : k-ordered fma chain from +0, bias afterwards
    Tensor forward(const Tensor &x_) const {
        if (!weight_.storage()) throw std::runtime_error("axiom stand-in: Linear used before its weight was loaded");
        Tensor x = x_.contig_f32();
        const size_t N = weight_.shape()[0], K = weight_.shape()[1];
        if (x.shape().back() != K) throw std::runtime_error("axiom stand-in: Linear input width mismatch");
        Shape os = x.shape();
        os.back() = N;
        Tensor out(os);
        const size_t M = x.numel() / K;
        ops::detail::gemm_kn(M, N, K, x.fdata(), wt_.data(), out.fdata());
        if (has_bias_ && bias_.storage()) {
            const float *b = bias_.fdata();
            float *o = out.fdata();
            for (size_t m = 0; m < M; ++m)
                for (size_t n = 0; n < N; ++n) o[m * N + n] += b[n];
        }
        return out;
    }
    Tensor operator()(const Tensor &x) const { return forward(x); }
    const Tensor &weight() const { return weight_; }
    const Tensor &bias() const { return bias_; }

  protected:
    void on_loaded() override {
        if (!weight_.storage()) return;
        weight_ = weight_.contig_f32();
        if (weight_.ndim() != 2) throw std::runtime_error("axiom stand-in: Linear weight must be 2-D");
        if (bias_.storage()) bias_ = bias_.contig_f32();
        const size_t N = weight_.shape()[0], K = weight_.shape()[1];
        wt_.resize(N * K);
        ops::detail::transpose2d(N, K, weight_.fdata(), wt_.data());
    }

  private:
    bool has_bias_;
    Tensor weight_, bias_;
    std::vector<float> wt_;  // [K][N]
};

class LayerNorm : public Module {
  public:
    explicit LayerNorm(float eps = 1e-5f) : eps_(eps) {
        register_parameter("weight", weight_);
        register_parameter("bias", bias_);
    }
    Tensor forward(const Tensor &x_) const {
        Tensor x = x_.contig_f32();
        const size_t d = x.shape().back(), rows = x.numel() / d;
        Tensor out(x.shape());
        const float *g = weight_.storage() ? weight_.fdata() : nullptr, *b = bias_.storage() ? bias_.fdata() : nullptr;
        for (size_t r = 0; r < rows; ++r) {
            const float *p = x.fdata() + r * d;
            float *o = out.fdata() + r * d;
            double s = 0.0;
            for (size_t i = 0; i < d; ++i) s += p[i];
            const float mean = (float)(s / (double)d);
            double v = 0.0;
            for (size_t i = 0; i < d; ++i) {
                const double c = (double)p[i] - (double)mean;
                v += c * c;
            }
            const float rstd = 1.0f / std::sqrt((float)(v / (double)d) + eps_);
            for (size_t i = 0; i < d; ++i) {
                float y = (p[i] - mean) * rstd;
                if (g) y *= g[i];
                if (b) y += b[i];
                o[i] = y;
            }
        }
        return out;
    }
    Tensor operator()(const Tensor &x) const { return forward(x); }

  protected:
    void on_loaded() override {
        if (weight_.storage()) weight_ = weight_.contig_f32();
        if (bias_.storage()) bias_ = bias_.contig_f32();
    }

  private:
    float eps_;
    Tensor weight_, bias_;
};

class Dropout : public Module {
  public:
    explicit Dropout(float p = 0.5f) : p_(p) {}
    Tensor forward(const Tensor &x) const { return x; }  // inference: identity (A4)
    Tensor operator()(const Tensor &x) const { return x; }

  private:
    float p_;
};

class Embedding : public Module {
  public:
    Embedding() { register_parameter("weight", weight_); }
    Tensor forward(const Tensor &idx_) const {
        if (!weight_.storage()) throw std::runtime_error("axiom stand-in: Embedding used before its weight was loaded");
        Tensor idx = idx_.ascontiguousarray();
        const size_t d = weight_.shape()[1], V = weight_.shape()[0];
        Shape os = idx.shape();
        os.push_back(d);
        Tensor out(os);
        for (size_t i = 0; i < idx.numel(); ++i) {
            int64_t id;
            switch (idx.dtype()) {
            case DType::Int32: id = idx.typed_data<int32_t>()[i]; break;
            case DType::Int64: id = idx.typed_data<int64_t>()[i]; break;
            case DType::Float32: id = (int64_t)idx.typed_data<float>()[i]; break;
            default: throw std::runtime_error("axiom stand-in: Embedding index dtype");
            }
            if (id < 0 || (size_t)id >= V) throw std::runtime_error("axiom stand-in: Embedding index out of range");
            std::memcpy(out.fdata() + i * d, weight_.fdata() + (size_t)id * d, d * sizeof(float));
        }
        return out;
    }
    Tensor operator()(const Tensor &idx) const { return forward(idx); }

  protected:
    void on_loaded() override {
        if (weight_.storage()) weight_ = weight_.contig_f32();
    }

  private:
    Tensor weight_;
};

// Conv1d over (batch, channels, length); weight (out, in/groups, k)
class Conv1d : public Module {
  public:
    explicit Conv1d(int stride = 1, int padding = 0, int dilation = 1, int groups = 1, bool bias = true)
        : stride_(stride), padding_(padding), dilation_(dilation), groups_(groups) {
        register_parameter("weight", weight_);
        if (bias) register_parameter("bias", bias_);
    }
    Tensor forward(const Tensor &x_) const {
        if (!weight_.storage()) throw std::runtime_error("axiom stand-in: Conv1d used before its weight was loaded");
        Tensor x = x_.contig_f32();
        const size_t B = x.shape()[0], Ci = x.shape()[1], L = x.shape()[2];
        const size_t Co = weight_.shape()[0], Cg = weight_.shape()[1], Kk = weight_.shape()[2];
        if (Cg * (size_t)groups_ != Ci) throw std::runtime_error("axiom stand-in: Conv1d channel mismatch");
        const int64_t Lo = ((int64_t)L + 2 * padding_ - dilation_ * ((int64_t)Kk - 1) - 1) / stride_ + 1;
        if (Lo <= 0) throw std::runtime_error("axiom stand-in: Conv1d output would be empty");
        Tensor out(Shape{B, Co, (size_t)Lo});
        const size_t opg = Co / (size_t)groups_;
        const float *w = weight_.fdata(), *bs = bias_.storage() ? bias_.fdata() : nullptr;
#pragma omp parallel for collapse(2) schedule(static)
        for (size_t b = 0; b < B; ++b)
            for (size_t co = 0; co < Co; ++co) {
                const size_t g = co / opg;
                float *o = out.fdata() + (b * Co + co) * (size_t)Lo;
                for (int64_t t = 0; t < Lo; ++t) o[t] = 0.0f;
                for (size_t ci = 0; ci < Cg; ++ci) {
                    const float *xi = x.fdata() + (b * Ci + g * Cg + ci) * L;
                    for (size_t k = 0; k < Kk; ++k) {
                        const float wv = w[(co * Cg + ci) * Kk + k];
                        for (int64_t t = 0; t < Lo; ++t) {
                            const int64_t it = t * stride_ - padding_ + (int64_t)k * dilation_;
                            if (it >= 0 && it < (int64_t)L) o[t] = std::fmaf(wv, xi[it], o[t]);
                        }
                    }
                }
                if (bs)
                    for (int64_t t = 0; t < Lo; ++t) o[t] += bs[co];
            }
        return out;
    }
    Tensor operator()(const Tensor &x) const { return forward(x); }

  protected:
    void on_loaded() override {
        if (weight_.storage()) weight_ = weight_.contig_f32();
        if (bias_.storage()) bias_ = bias_.contig_f32();
    }

  private:
    int stride_, padding_, dilation_, groups_;
    Tensor weight_, bias_;
};

// Conv2d over (batch, channels, H, W); weight (out, in/groups, kh, kw)
class Conv2d : public Module {
  public:
    using I2 = std::array<int, 2>;
    explicit Conv2d(I2 stride = {1, 1}, I2 padding = {0, 0}, I2 dilation = {1, 1}, int groups = 1, bool bias = true)
        : stride_(stride), padding_(padding), dilation_(dilation), groups_(groups) {
        register_parameter("weight", weight_);
        if (bias) register_parameter("bias", bias_);
    }
    Tensor forward(const Tensor &x_) const {
        if (!weight_.storage()) throw std::runtime_error("axiom stand-in: Conv2d used before its weight was loaded");
        Tensor x = x_.contig_f32();
        const size_t B = x.shape()[0], Ci = x.shape()[1], H = x.shape()[2], W = x.shape()[3];
        const size_t Co = weight_.shape()[0], Cg = weight_.shape()[1], KH = weight_.shape()[2], KW = weight_.shape()[3];
        if (Cg * (size_t)groups_ != Ci) throw std::runtime_error("axiom stand-in: Conv2d channel mismatch");
        const int64_t Ho = ((int64_t)H + 2 * padding_[0] - dilation_[0] * ((int64_t)KH - 1) - 1) / stride_[0] + 1;
        const int64_t Wo = ((int64_t)W + 2 * padding_[1] - dilation_[1] * ((int64_t)KW - 1) - 1) / stride_[1] + 1;
        if (Ho <= 0 || Wo <= 0) throw std::runtime_error("axiom stand-in: Conv2d output would be empty");
        Tensor out(Shape{B, Co, (size_t)Ho, (size_t)Wo});
        const size_t opg = Co / (size_t)groups_;
        const float *w = weight_.fdata(), *bs = bias_.storage() ? bias_.fdata() : nullptr;
#pragma omp parallel for collapse(2) schedule(static)
        for (size_t b = 0; b < B; ++b)
            for (size_t co = 0; co < Co; ++co) {
                const size_t g = co / opg;
                float *o = out.fdata() + (b * Co + co) * (size_t)(Ho * Wo);
                for (int64_t i = 0; i < Ho * Wo; ++i) o[i] = 0.0f;
                for (size_t ci = 0; ci < Cg; ++ci) {
                    const float *xi = x.fdata() + (b * Ci + g * Cg + ci) * H * W;
                    for (size_t kh = 0; kh < KH; ++kh)
                        for (size_t kw = 0; kw < KW; ++kw) {
                            const float wv = w[((co * Cg + ci) * KH + kh) * KW + kw];
                            for (int64_t oh = 0; oh < Ho; ++oh) {
                                const int64_t ih = oh * stride_[0] - padding_[0] + (int64_t)kh * dilation_[0];
                                if (ih < 0 || ih >= (int64_t)H) continue;
                                float *orow = o + oh * Wo;
                                const float *xrow = xi + ih * (int64_t)W;
                                for (int64_t ow = 0; ow < Wo; ++ow) {
                                    const int64_t iw = ow * stride_[1] - padding_[1] + (int64_t)kw * dilation_[1];
                                    if (iw >= 0 && iw < (int64_t)W) orow[ow] = std::fmaf(wv, xrow[iw], orow[ow]);
                                }
                            }
                        }
                }
                if (bs)
                    for (int64_t i = 0; i < Ho * Wo; ++i) o[i] += bs[co];
            }
        return out;
    }
    Tensor operator()(const Tensor &x) const { return forward(x); }

  protected:
    void on_loaded() override {
        if (weight_.storage()) weight_ = weight_.contig_f32();
        if (bias_.storage()) bias_ = bias_.contig_f32();
    }

  private:
    I2 stride_, padding_, dilation_;
    int groups_;
    Tensor weight_, bias_;
};

// inference-mode BatchNorm1d over (batch, channels, length): running statistics (A4), eps 1e-5 (A3)
class BatchNorm1d : public Module {
  public:
    explicit BatchNorm1d(float eps = 1e-5f) : eps_(eps) {
        register_parameter("weight", weight_);
        register_parameter("bias", bias_);
        register_parameter("running_mean", running_mean_);
        register_parameter("running_var", running_var_);
        register_parameter("num_batches_tracked", num_batches_tracked_);
    }
    Tensor forward(const Tensor &x_) const {
        Tensor x = x_.contig_f32();
        const size_t B = x.shape()[0], C = x.shape()[1], L = x.ndim() > 2 ? x.shape()[2] : 1;
        if (!running_mean_.storage() || !running_var_.storage())
            throw std::runtime_error("axiom stand-in: BatchNorm1d used before its statistics were loaded");
        Tensor out(x.shape());
        for (size_t b = 0; b < B; ++b)
            for (size_t c = 0; c < C; ++c) {
                const float mu = running_mean_.fdata()[c];
                const float rstd = 1.0f / std::sqrt(running_var_.fdata()[c] + eps_);
                const float g = weight_.storage() ? weight_.fdata()[c] : 1.0f, bb = bias_.storage() ? bias_.fdata()[c] : 0.0f;
                const float *p = x.fdata() + (b * C + c) * L;
                float *o = out.fdata() + (b * C + c) * L;
                for (size_t t = 0; t < L; ++t) o[t] = (p[t] - mu) * rstd * g + bb;
            }
        return out;
    }
    Tensor operator()(const Tensor &x) const { return forward(x); }

  protected:
    void on_loaded() override {
        for (Tensor *t : {&weight_, &bias_, &running_mean_, &running_var_})
            if (t->storage()) *t = t->contig_f32();
    }

  private:
    float eps_;
    Tensor weight_, bias_, running_mean_, running_var_, num_batches_tracked_;
};

// Only the projections are used by the reference (encoder.cpp:120-123, transformer.cpp:22-24): it bypasses forward().
class MultiHeadAttention : public Module {
  public:
    explicit MultiHeadAttention(int num_heads = 8) : num_heads_(num_heads), q_proj_(true), k_proj_(true), v_proj_(true), out_proj_(true) {
        register_module("q_proj", q_proj_);
        register_module("k_proj", k_proj_);
        register_module("v_proj", v_proj_);
        register_module("out_proj", out_proj_);
    }
    int num_heads() const { return num_heads_; }
    const Linear &q_proj() const { return q_proj_; }
    const Linear &k_proj() const { return k_proj_; }
    const Linear &v_proj() const { return v_proj_; }
    const Linear &out_proj() const { return out_proj_; }

  private:
    int num_heads_;
    Linear q_proj_, k_proj_, v_proj_, out_proj_;
};

}  // namespace axiom::nn
