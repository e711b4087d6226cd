// oracle/axiom_stub/axiom/axiom.hpp -- TEST INFRASTRUCTURE ONLY.
// The reference's src/audio_io.cpp (decoders via dr_libs / stb_vorbis, downmix, Kaiser-windowed sinc resampler) touches the
// un-vendored `axiom` library only to WRAP its final std::vector<float> in a tensor (axiom::Tensor::from_data, and
// ascontiguousarray / typed_data / shape in resample()).  This 40-line stand-in provides exactly that surface so the REAL
// audio_io.cpp compiles where it lies into oracle/_ref/libpk_ref_audio.so (oracle/Makefile) and the product's host-side
// resampler / WAV reader can be pinned against the reference's own object code.  It implements no arithmetic.
#pragma once
#include <cstddef>
#include <initializer_list>
#include <memory>
#include <vector>

namespace axiom {

struct Shape {
    std::vector<size_t> dims;
    Shape() = default;
    Shape(std::initializer_list<size_t> l) : dims(l) {}
    size_t operator[](size_t i) const { return dims[i]; }
    size_t size() const { return dims.size(); }
};

class Tensor {
  public:
    Tensor() = default;
    static Tensor from_data(const float *p, const Shape &s, bool /*copy*/ = true) {
        Tensor t;
        size_t n = 1;
        for (size_t d : s.dims) n *= d;
        t.data_ = std::make_shared<std::vector<float>>(p, p + n);
        t.shape_ = s;
        return t;
    }
    Tensor ascontiguousarray() const { return *this; }
    template <class T> const T *typed_data() const { return reinterpret_cast<const T *>(data_ ? data_->data() : nullptr); }
    const Shape &shape() const { return shape_; }
    bool storage() const { return (bool)data_; }

  private:
    std::shared_ptr<std::vector<float>> data_;
    Shape shape_;
};

}  // namespace axiom
