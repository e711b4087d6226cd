// oracle/axiom_stub/axiom/axiom.hpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A small CPU stand-in for the reference's un-vendored tensor library (`Frikallo/axiom`, .gitmodules:1-3 of the reference;
// third_party/axiom/ is empty in the mount, so the real library cannot be built here).  It exists for ONE purpose: to let the
// REAL reference translation units -- src/{audio,encoder,lstm,rnnt,tdt,ctc,tdt_ctc,transformer,streaming_encoder,eou,nemotron,
// sortformer,phrase_boost,audio_io}.cpp and include/parakeet/transcribe.hpp -- compile WHERE THEY LIE (oracle/Makefile) into
// oracle/_ref/libpk_ref_model.so, so that the control flow, tensor plumbing, weight-name registration, module wiring and decode
// loops that the CPU oracle (oracle/pk_oracle.c) restates are checked against the reference's own object code.
//
// What this stand-in supplies is only the *surface the reference calls* (found by compiling it): a float/int strided tensor,
// the ~20 `ops::` functions, `fft::{hann_window,stft}`, the `nn::` modules with name-based `load_state_dict`, and a safetensors
// reader.  Arithmetic conventions (all are what PyTorch does, which is what the reference author checks the C++ against in
// scripts/compare_encoder.py / compare_features.py):
//   * Linear / matmul: fp32, each output is a k-ordered fused-multiply-add chain from +0, bias added afterwards;
//   * LayerNorm / BatchNorm1d: eps 1e-5, biased variance, inference statistics (assumption A3/A4 of SURVEY.md 8c);
//   * softmax / log_softmax: max-subtracted; argmax: first maximum (A6); glu: a * sigmoid(b), a = first half;
//   * stft: the win_length-tap window is placed at the START of the n_fft frame by default (A1; `fft::window_centered()` flips
//     it to torch.stft placement), `abs()` of a complex tensor is sqrt(re^2+im^2) (A2).
// These are ASSUMPTIONS about axiom, stated in DESIGN.md section 2; everything else is the reference's own code.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace axiom {

enum class DType { Float32, Int32, Int64, Bool, Complex64 };
enum class Device { CPU, GPU };

using Shape = std::vector<size_t>;

inline size_t dtype_size(DType d) {
    switch (d) {
    case DType::Float32: return 4;
    case DType::Int32: return 4;
    case DType::Int64: return 8;
    case DType::Bool: return 1;
    case DType::Complex64: return 8;
    }
    return 4;
}

struct Slice {
    int64_t start = 0;
    int64_t stop = std::numeric_limits<int64_t>::max();
    Slice() = default;
    Slice(int64_t s) : start(s) {}
    Slice(int64_t s, int64_t e) : start(s), stop(e) {}
};

namespace system {
inline bool is_metal_available() { return false; }
inline bool is_gpu_available() { return false; }
}  // namespace system

class Tensor {
  public:
    Tensor() = default;
    explicit Tensor(const Shape &shape, DType dt = DType::Float32) : shape_(shape), dtype_(dt) {
        size_t n = numel_of(shape);
        buf_ = std::make_shared<std::vector<unsigned char>>(std::max<size_t>(n, 1) * dtype_size(dt), 0);
        set_contiguous_strides();
    }
    static Tensor zeros(const Shape &shape, DType dt = DType::Float32) { return Tensor(shape, dt); }
    static Tensor from_data(const float *p, const Shape &s, bool /*copy*/ = true) {
        Tensor t(s, DType::Float32);
        if (t.numel()) std::memcpy(t.raw(), p, t.numel() * sizeof(float));
        return t;
    }

    // ---- metadata
    const Shape &shape() const { return shape_; }
    const std::vector<int64_t> &strides() const { return strides_; }
    DType dtype() const { return dtype_; }
    size_t ndim() const { return shape_.size(); }
    size_t numel() const { return buf_ ? numel_of(shape_) : 0; }
    size_t size() const { return numel(); }
    bool storage() const { return (bool)buf_; }
    Device device() const { return Device::CPU; }
    Tensor cpu() const { return *this; }
    Tensor gpu() const { return *this; }
    Tensor to(Device) const { return *this; }

    bool is_contiguous() const {
        int64_t s = 1;
        for (size_t i = shape_.size(); i-- > 0;) {
            if (shape_[i] != 1 && strides_[i] != s) return false;
            s *= (int64_t)shape_[i];
        }
        return true;
    }

    // ---- raw access (valid after ascontiguousarray(), which is how the reference uses it)
    template <class T> const T *typed_data() const { return reinterpret_cast<const T *>(raw()); }
    template <class T> T *typed_data() { return reinterpret_cast<T *>(raw()); }

    template <class T> T item() const {
        if (numel() != 1) throw std::runtime_error("axiom stand-in: item() on a tensor with numel != 1");
        switch (dtype_) {
        case DType::Float32: return (T) * reinterpret_cast<const float *>(raw());
        case DType::Int32: return (T) * reinterpret_cast<const int32_t *>(raw());
        case DType::Int64: return (T) * reinterpret_cast<const int64_t *>(raw());
        case DType::Bool: return (T) * reinterpret_cast<const unsigned char *>(raw());
        default: throw std::runtime_error("axiom stand-in: item() on complex");
        }
    }

    template <class V> void fill(V v) {
        Tensor c = is_contiguous() ? *this : ascontiguousarray();
        size_t n = numel();
        switch (dtype_) {
        case DType::Float32: std::fill_n(reinterpret_cast<float *>(c.raw()), n, (float)v); break;
        case DType::Int32: std::fill_n(reinterpret_cast<int32_t *>(c.raw()), n, (int32_t)v); break;
        case DType::Int64: std::fill_n(reinterpret_cast<int64_t *>(c.raw()), n, (int64_t)v); break;
        case DType::Bool: std::fill_n(reinterpret_cast<unsigned char *>(c.raw()), n, (unsigned char)(v != 0)); break;
        default: throw std::runtime_error("axiom stand-in: fill() on complex");
        }
        if (!is_contiguous()) *this = c;
    }

    // ---- views
    Tensor ascontiguousarray() const {
        if (!buf_ || is_contiguous()) return *this;   // raw() already includes the view's offset
        Tensor out(shape_, dtype_);
        copy_strided_to(out.raw());
        return out;
    }

    Tensor reshape(const Shape &ns) const {
        if (numel_of(ns) != numel()) throw std::runtime_error("axiom stand-in: reshape size mismatch");
        Tensor src = is_contiguous() ? *this : ascontiguousarray();
        Tensor out = src;
        out.shape_ = ns;
        out.set_contiguous_strides();
        return out;
    }
    Tensor flatten() const { return reshape({numel()}); }

    Tensor permute(const std::vector<int> &axes) const {
        if (axes.size() != shape_.size()) throw std::runtime_error("axiom stand-in: permute rank mismatch");
        Tensor out = *this;
        for (size_t i = 0; i < axes.size(); ++i) {
            size_t a = norm_axis(axes[i], shape_.size());
            out.shape_[i] = shape_[a];
            out.strides_[i] = strides_[a];
        }
        return out;
    }
    Tensor transpose(const std::vector<int> &axes) const { return permute(axes); }
    Tensor transpose() const {
        if (shape_.size() < 2) return *this;
        std::vector<int> ax(shape_.size());
        std::iota(ax.begin(), ax.end(), 0);
        std::reverse(ax.begin(), ax.end());
        return permute(ax);
    }

    Tensor squeeze(int axis) const {
        size_t a = norm_axis(axis, shape_.size());
        if (shape_[a] != 1) return *this;
        Tensor out = *this;
        out.shape_.erase(out.shape_.begin() + a);
        out.strides_.erase(out.strides_.begin() + a);
        return out;
    }
    Tensor unsqueeze(int axis) const {
        size_t a = norm_axis(axis, shape_.size() + 1);
        Tensor out = *this;
        out.shape_.insert(out.shape_.begin() + a, 1);
        out.strides_.insert(out.strides_.begin() + a, 1);
        return out;
    }

    Tensor slice(const std::vector<Slice> &sl) const {
        if (sl.size() > shape_.size()) throw std::runtime_error("axiom stand-in: too many slices");
        Tensor out = *this;
        for (size_t i = 0; i < sl.size(); ++i) {
            int64_t n = (int64_t)shape_[i];
            int64_t b = sl[i].start, e = sl[i].stop;
            if (b < 0) b += n;
            if (e < 0) e += n;
            b = std::clamp<int64_t>(b, 0, n);
            e = std::clamp<int64_t>(e, b, n);
            out.offset_ += (size_t)(b * strides_[i]);
            out.shape_[i] = (size_t)(e - b);
        }
        return out;
    }

    std::vector<Tensor> chunk(int n, int axis) const {
        size_t a = norm_axis(axis, shape_.size());
        size_t len = shape_[a], step = (len + n - 1) / n;
        std::vector<Tensor> out;
        for (size_t b = 0; b < len; b += step) {
            std::vector<Slice> sl(a + 1);
            sl[a] = Slice((int64_t)b, (int64_t)std::min(len, b + step));
            out.push_back(slice(sl));
        }
        return out;
    }

    static Tensor cat(const std::vector<Tensor> &ts, int axis) {
        if (ts.empty()) return Tensor();
        size_t a = norm_axis(axis, ts[0].shape_.size());
        Shape os = ts[0].shape_;
        size_t total = 0;
        for (const auto &t : ts) {
            if (t.shape_.size() != os.size() || t.dtype_ != ts[0].dtype_) throw std::runtime_error("axiom stand-in: cat mismatch");
            total += t.shape_[a];
        }
        os[a] = total;
        Tensor out(os, ts[0].dtype_);
        size_t pos = 0;
        for (const auto &t : ts) {
            std::vector<Slice> sl(a + 1);
            sl[a] = Slice((int64_t)pos, (int64_t)(pos + t.shape_[a]));
            Tensor dst = out.slice(sl);
            t.copy_into_view(dst);
            pos += t.shape_[a];
        }
        return out;
    }
    static Tensor stack(const std::vector<Tensor> &ts, int axis) {
        std::vector<Tensor> u;
        u.reserve(ts.size());
        for (const auto &t : ts) u.push_back(t.unsqueeze(axis < 0 ? axis + (int)t.shape_.size() + 1 : axis));
        return cat(u, axis < 0 ? axis + (int)ts[0].shape_.size() + 1 : axis);
    }

    // ---- helpers used by ops (public: the stand-in is one unit)
    static size_t numel_of(const Shape &s) {
        size_t n = 1;
        for (size_t d : s) n *= d;
        return n;
    }
    static size_t norm_axis(int axis, size_t nd) {
        int a = axis < 0 ? axis + (int)nd : axis;
        if (a < 0 || a >= (int)nd) throw std::runtime_error("axiom stand-in: axis out of range");
        return (size_t)a;
    }
    unsigned char *raw() const { return buf_ ? buf_->data() + offset_ * dtype_size(dtype_) : nullptr; }
    // contiguous float32 copy / view of this tensor
    Tensor contig_f32() const {
        if (dtype_ != DType::Float32) throw std::runtime_error("axiom stand-in: float32 tensor expected");
        return ascontiguousarray();
    }
    const float *fdata() const { return reinterpret_cast<const float *>(raw()); }
    float *fdata() { return reinterpret_cast<float *>(raw()); }

    // element offset (in elements, relative to raw()) of a multi-index given broadcast strides
    void copy_strided_to(unsigned char *dst) const {
        const size_t es = dtype_size(dtype_), nd = shape_.size();
        const size_t n = numel_of(shape_);
        if (n == 0) return;
        if (nd == 0) {
            std::memcpy(dst, raw(), es);
            return;
        }
        std::vector<size_t> idx(nd, 0);
        const size_t inner = shape_[nd - 1];
        const int64_t istr = strides_[nd - 1];
        const unsigned char *base = raw();
        size_t outer = n / std::max<size_t>(inner, 1);
        for (size_t o = 0; o < outer; ++o) {
            int64_t off = 0;
            for (size_t d = 0; d + 1 < nd; ++d) off += (int64_t)idx[d] * strides_[d];
            const unsigned char *src = base + off * (int64_t)es;
            if (istr == 1) {
                std::memcpy(dst, src, inner * es);
            } else {
                for (size_t i = 0; i < inner; ++i) std::memcpy(dst + i * es, src + (int64_t)i * istr * (int64_t)es, es);
            }
            dst += inner * es;
            for (size_t d = nd - 1; d-- > 0;) {
                if (++idx[d] < shape_[d]) break;
                idx[d] = 0;
            }
        }
    }
    // copy *this (any strides) into the view `dst` (same shape, any strides)
    void copy_into_view(Tensor &dst) const {
        Tensor src = ascontiguousarray();
        const size_t es = dtype_size(dtype_), nd = shape_.size();
        const size_t n = numel_of(shape_);
        if (n == 0) return;
        std::vector<size_t> idx(nd, 0);
        const unsigned char *s = src.raw();
        unsigned char *base = dst.raw();
        for (size_t e = 0; e < n; ++e) {
            int64_t off = 0;
            for (size_t d = 0; d < nd; ++d) off += (int64_t)idx[d] * dst.strides_[d];
            std::memcpy(base + off * (int64_t)es, s + e * es, es);
            for (size_t d = nd; d-- > 0;) {
                if (++idx[d] < shape_[d]) break;
                idx[d] = 0;
            }
        }
    }

  private:
    void set_contiguous_strides() {
        strides_.assign(shape_.size(), 1);
        int64_t s = 1;
        for (size_t i = shape_.size(); i-- > 0;) {
            strides_[i] = s;
            s *= (int64_t)shape_[i];
        }
    }
    std::shared_ptr<std::vector<unsigned char>> buf_;
    Shape shape_;
    std::vector<int64_t> strides_;
    size_t offset_ = 0;
    DType dtype_ = DType::Float32;
};

// ───────────────────────────── elementwise ─────────────────────────────
namespace detail {

inline Shape broadcast_shape(const Shape &a, const Shape &b) {
    size_t nd = std::max(a.size(), b.size());
    Shape o(nd);
    for (size_t i = 0; i < nd; ++i) {
        size_t da = i + a.size() >= nd ? a[i + a.size() - nd] : 1;
        size_t db = i + b.size() >= nd ? b[i + b.size() - nd] : 1;
        if (da != db && da != 1 && db != 1) throw std::runtime_error("axiom stand-in: shapes do not broadcast");
        o[i] = std::max(da, db);
        if (da == 0 || db == 0) o[i] = 0;
    }
    return o;
}
// strides (in elements) of contiguous tensor `s` viewed as broadcast to `o`
inline std::vector<int64_t> bstrides(const Shape &s, const Shape &o) {
    std::vector<int64_t> st(o.size(), 0);
    int64_t acc = 1;
    for (size_t i = s.size(); i-- > 0;) {
        size_t oi = i + o.size() - s.size();
        st[oi] = s[i] == 1 ? 0 : acc;
        acc *= (int64_t)s[i];
    }
    return st;
}

template <class F> Tensor binary(const Tensor &a_, const Tensor &b_, F f) {
    Tensor a = a_.contig_f32(), b = b_.contig_f32();
    Shape os = broadcast_shape(a.shape(), b.shape());
    Tensor out(os);
    const size_t n = Tensor::numel_of(os);
    if (n == 0) return out;
    const float *pa = a.fdata(), *pb = b.fdata();
    float *po = out.fdata();
    if (a.shape() == b.shape()) {
        for (size_t i = 0; i < n; ++i) po[i] = f(pa[i], pb[i]);
        return out;
    }
    const size_t nd = os.size();
    auto sa = bstrides(a.shape(), os), sb = bstrides(b.shape(), os);
    std::vector<size_t> idx(nd, 0);
    const size_t inner = nd ? os[nd - 1] : 1;
    const int64_t ia = nd ? sa[nd - 1] : 0, ib = nd ? sb[nd - 1] : 0;
    for (size_t o = 0; o < n / inner; ++o) {
        int64_t oa = 0, ob = 0;
        for (size_t d = 0; d + 1 < nd; ++d) {
            oa += (int64_t)idx[d] * sa[d];
            ob += (int64_t)idx[d] * sb[d];
        }
        for (size_t i = 0; i < inner; ++i) po[i] = f(pa[oa + (int64_t)i * ia], pb[ob + (int64_t)i * ib]);
        po += inner;
        for (size_t d = nd - 1; d-- > 0;) {
            if (++idx[d] < os[d]) break;
            idx[d] = 0;
        }
    }
    return out;
}
template <class F> Tensor unary(const Tensor &a_, F f) {
    Tensor a = a_.contig_f32();
    Tensor out(a.shape());
    const float *pa = a.fdata();
    float *po = out.fdata();
    const size_t n = a.numel();
    for (size_t i = 0; i < n; ++i) po[i] = f(pa[i]);
    return out;
}
inline Tensor scalar(float v) {
    Tensor t(Shape{1});
    t.fdata()[0] = v;
    return t;
}
}  // namespace detail

inline Tensor operator+(const Tensor &a, const Tensor &b) { return detail::binary(a, b, [](float x, float y) { return x + y; }); }
inline Tensor operator-(const Tensor &a, const Tensor &b) { return detail::binary(a, b, [](float x, float y) { return x - y; }); }
inline Tensor operator*(const Tensor &a, const Tensor &b) { return detail::binary(a, b, [](float x, float y) { return x * y; }); }
inline Tensor operator/(const Tensor &a, const Tensor &b) { return detail::binary(a, b, [](float x, float y) { return x / y; }); }
inline Tensor operator+(const Tensor &a, float s) { return detail::unary(a, [s](float x) { return x + s; }); }
inline Tensor operator-(const Tensor &a, float s) { return detail::unary(a, [s](float x) { return x - s; }); }
inline Tensor operator*(const Tensor &a, float s) { return detail::unary(a, [s](float x) { return x * s; }); }
inline Tensor operator/(const Tensor &a, float s) { return detail::unary(a, [s](float x) { return x / s; }); }
inline Tensor operator+(float s, const Tensor &a) { return a + s; }
inline Tensor operator*(float s, const Tensor &a) { return a * s; }
inline Tensor operator-(const Tensor &a) { return detail::unary(a, [](float x) { return -x; }); }

// ───────────────────────────── ops ─────────────────────────────
namespace ops {

namespace detail {
// C[M][N] = sum_k A[m][k] * Bt[k][n]  (natural-k fma chain from +0 per output; vectorised across n)
inline void gemm_kn(size_t M, size_t N, size_t K, const float *A, const float *Bkn, float *C) {
#pragma omp parallel for schedule(static) if (M * N * K > (1u << 18))
    for (size_t m = 0; m < M; ++m) {
        float *c = C + m * N;
        for (size_t n = 0; n < N; ++n) c[n] = 0.0f;
        const float *a = A + m * K;
        for (size_t k = 0; k < K; ++k) {
            const float av = a[k];
            const float *b = Bkn + k * N;
            for (size_t n = 0; n < N; ++n) c[n] = std::fmaf(av, b[n], c[n]);
        }
    }
}
inline void transpose2d(size_t R, size_t Cc, const float *in, float *out) {  // in[R][Cc] -> out[Cc][R]
    for (size_t r = 0; r < R; ++r)
        for (size_t c = 0; c < Cc; ++c) out[c * R + r] = in[r * Cc + c];
}
}  // namespace detail

// batched matmul with leading-dimension broadcasting; transpose flags act on the last two dims
inline Tensor matmul(const Tensor &a_, const Tensor &b_, bool ta = false, bool tb = false) {
    Tensor a = a_, b = b_;
    if (a.ndim() < 2 || b.ndim() < 2) throw std::runtime_error("axiom stand-in: matmul needs >= 2-D operands");
    if (ta) {
        std::vector<int> ax(a.ndim());
        std::iota(ax.begin(), ax.end(), 0);
        std::swap(ax[a.ndim() - 1], ax[a.ndim() - 2]);
        a = a.permute(ax);
    }
    if (tb) {
        std::vector<int> ax(b.ndim());
        std::iota(ax.begin(), ax.end(), 0);
        std::swap(ax[b.ndim() - 1], ax[b.ndim() - 2]);
        b = b.permute(ax);
    }
    a = a.contig_f32();
    b = b.contig_f32();  // [.., K, N]
    const size_t M = a.shape()[a.ndim() - 2], K = a.shape()[a.ndim() - 1], N = b.shape()[b.ndim() - 1];
    if (b.shape()[b.ndim() - 2] != K) throw std::runtime_error("axiom stand-in: matmul inner dimensions differ");
    Shape ba(a.shape().begin(), a.shape().end() - 2), bb(b.shape().begin(), b.shape().end() - 2);
    Shape bo = axiom::detail::broadcast_shape(ba, bb);
    Shape os = bo;
    os.push_back(M);
    os.push_back(N);
    Tensor out(os);
    const size_t nb = Tensor::numel_of(bo);
    auto sa = axiom::detail::bstrides(ba, bo), sb = axiom::detail::bstrides(bb, bo);
    std::vector<size_t> idx(bo.size(), 0);
    for (size_t i = 0; i < nb; ++i) {
        int64_t oa = 0, ob = 0;
        for (size_t d = 0; d < bo.size(); ++d) {
            oa += (int64_t)idx[d] * sa[d];
            ob += (int64_t)idx[d] * sb[d];
        }
        detail::gemm_kn(M, N, K, a.fdata() + oa * (int64_t)(M * K), b.fdata() + ob * (int64_t)(K * N), out.fdata() + i * M * N);
        for (size_t d = bo.size(); d-- > 0;) {
            if (++idx[d] < bo[d]) break;
            idx[d] = 0;
        }
    }
    return out;
}

inline Tensor relu(const Tensor &x) { return axiom::detail::unary(x, [](float v) { return v > 0.0f ? v : 0.0f; }); }
inline Tensor sigmoid(const Tensor &x) { return axiom::detail::unary(x, [](float v) { return 1.0f / (1.0f + std::exp(-v)); }); }
inline Tensor tanh(const Tensor &x) { return axiom::detail::unary(x, [](float v) { return std::tanh(v); }); }
inline Tensor silu(const Tensor &x) { return axiom::detail::unary(x, [](float v) { return v / (1.0f + std::exp(-v)); }); }
inline Tensor log(const Tensor &x) { return axiom::detail::unary(x, [](float v) { return std::log(v); }); }
inline Tensor exp(const Tensor &x) { return axiom::detail::unary(x, [](float v) { return std::exp(v); }); }
inline Tensor sqrt(const Tensor &x) { return axiom::detail::unary(x, [](float v) { return std::sqrt(v); }); }
inline Tensor abs(const Tensor &x) {
    if (x.dtype() == DType::Complex64) {
        Tensor c = x.ascontiguousarray();
        Tensor out(c.shape());
        const float *p = reinterpret_cast<const float *>(c.raw());
        float *o = out.fdata();
        for (size_t i = 0; i < out.numel(); ++i) o[i] = std::sqrt(std::fmaf(p[2 * i], p[2 * i], p[2 * i + 1] * p[2 * i + 1]));
        return out;
    }
    return axiom::detail::unary(x, [](float v) { return std::fabs(v); });
}

inline Tensor glu(const Tensor &x, int dim) {
    auto h = x.chunk(2, dim);
    return h[0] * sigmoid(h[1]);
}

namespace detail {
template <class F> Tensor lastdim(const Tensor &x_, int axis, F f) {
    size_t a = Tensor::norm_axis(axis, x_.ndim());
    Tensor x = x_;
    bool moved = a + 1 != x.ndim();
    std::vector<int> ax(x.ndim());
    std::iota(ax.begin(), ax.end(), 0);
    if (moved) {
        std::swap(ax[a], ax[x.ndim() - 1]);
        x = x.permute(ax);
    }
    x = x.contig_f32();
    Tensor out(x.shape());
    const size_t n = x.shape().back(), rows = n ? x.numel() / n : 0;
    for (size_t r = 0; r < rows; ++r) f(x.fdata() + r * n, out.fdata() + r * n, n);
    return moved ? out.permute(ax).ascontiguousarray() : out;
}
}  // namespace detail

inline Tensor softmax(const Tensor &x, int axis = -1) {
    return detail::lastdim(x, axis, [](const float *in, float *out, size_t n) {
        float mx = in[0];
        for (size_t i = 1; i < n; ++i) mx = std::max(mx, in[i]);
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) {
            out[i] = std::exp(in[i] - mx);
            s += out[i];
        }
        const float fs = (float)s;
        for (size_t i = 0; i < n; ++i) out[i] = out[i] / fs;
    });
}
inline Tensor log_softmax(const Tensor &x, int axis = -1) {
    return detail::lastdim(x, axis, [](const float *in, float *out, size_t n) {
        float mx = in[0];
        for (size_t i = 1; i < n; ++i) mx = std::max(mx, in[i]);
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += std::exp(in[i] - mx);
        const float lse = std::log((float)s);
        for (size_t i = 0; i < n; ++i) out[i] = (in[i] - mx) - lse;
    });
}

// first maximum along `axis`; result dtype Int64, the axis is removed
inline Tensor argmax(const Tensor &x_, int axis = -1) {
    size_t a = Tensor::norm_axis(axis, x_.ndim());
    Tensor x = x_;
    if (a + 1 != x.ndim()) {
        std::vector<int> ax(x.ndim());
        std::iota(ax.begin(), ax.end(), 0);
        ax.erase(ax.begin() + a);
        ax.push_back((int)a);
        x = x.permute(ax);
    }
    x = x.contig_f32();
    Shape os(x.shape().begin(), x.shape().end() - 1);
    Tensor out(os, DType::Int64);
    const size_t n = x.shape().back(), rows = x.numel() / n;
    int64_t *o = out.typed_data<int64_t>();
    for (size_t r = 0; r < rows; ++r) {
        const float *p = x.fdata() + r * n;
        size_t best = 0;
        for (size_t i = 1; i < n; ++i)
            if (p[i] > p[best]) best = i;
        o[r] = (int64_t)best;
    }
    return out;
}

inline Tensor pad(const Tensor &x, const std::vector<std::pair<int, int>> &widths, float value = 0.0f) {
    if (widths.size() != x.ndim()) throw std::runtime_error("axiom stand-in: pad widths rank mismatch");
    Shape os = x.shape();
    std::vector<Slice> sl(x.ndim());
    for (size_t i = 0; i < x.ndim(); ++i) {
        os[i] += (size_t)(widths[i].first + widths[i].second);
        sl[i] = Slice(widths[i].first, widths[i].first + (int64_t)x.shape()[i]);
    }
    Tensor out(os);
    if (value != 0.0f) out.fill(value);
    Tensor dst = out.slice(sl);
    x.copy_into_view(dst);
    return out;
}

inline Tensor masked_fill(const Tensor &x, const Tensor &mask, float value) {
    Tensor m = mask;
    if (m.dtype() != DType::Float32) {
        Tensor c = m.ascontiguousarray(), f(c.shape());
        for (size_t i = 0; i < c.numel(); ++i) {
            float v = 0.0f;
            switch (c.dtype()) {
            case DType::Bool: v = c.typed_data<unsigned char>()[i] ? 1.0f : 0.0f; break;
            case DType::Int32: v = c.typed_data<int32_t>()[i] ? 1.0f : 0.0f; break;
            case DType::Int64: v = c.typed_data<int64_t>()[i] ? 1.0f : 0.0f; break;
            default: break;
            }
            f.fdata()[i] = v;
        }
        m = f;
    }
    return axiom::detail::binary(x, m, [value](float a, float b) { return b != 0.0f ? value : a; });
}

namespace detail {
inline Tensor reduce(const Tensor &x_, const std::vector<int> &axes, bool mean) {
    Tensor x = x_.contig_f32();
    std::vector<bool> red(x.ndim(), false);
    for (int a : axes) red[Tensor::norm_axis(a, x.ndim())] = true;
    Shape os;
    size_t cnt = 1;
    for (size_t d = 0; d < x.ndim(); ++d) {
        if (red[d]) cnt *= x.shape()[d];
        else os.push_back(x.shape()[d]);
    }
    Tensor out(os);
    std::vector<double> acc(std::max<size_t>(out.numel(), 1), 0.0);
    std::vector<size_t> idx(x.ndim(), 0);
    const float *p = x.fdata();
    for (size_t e = 0; e < x.numel(); ++e) {
        size_t o = 0;
        for (size_t d = 0; d < x.ndim(); ++d)
            if (!red[d]) o = o * x.shape()[d] + idx[d];
        acc[o] += p[e];
        for (size_t d = x.ndim(); d-- > 0;) {
            if (++idx[d] < x.shape()[d]) break;
            idx[d] = 0;
        }
    }
    for (size_t o = 0; o < out.numel(); ++o) out.fdata()[o] = (float)(mean ? acc[o] / (double)cnt : acc[o]);
    return out;
}
}  // namespace detail
inline Tensor sum(const Tensor &x, const std::vector<int> &axes) { return detail::reduce(x, axes, false); }
inline Tensor mean(const Tensor &x, const std::vector<int> &axes) { return detail::reduce(x, axes, true); }

}  // namespace ops
}  // namespace axiom

// the reference's eou.cpp / nemotron.cpp call axiom::io::safetensors::load with only <axiom/axiom.hpp> in scope
#include "io/safetensors.hpp"
