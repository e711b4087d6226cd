// parakeet.cpp_amd/csrc/transformer.hpp -- TransformerEncoder of the reference (src/transformer.cpp:15-88) on the gfx950 kernels;
// see transformer.cpp.  Used stand-alone (pk_transformer_*) and as the middle of Sortformer (sortformer.cpp).
#pragma once
#include <string>
#include <vector>

#include "engine.hpp"

namespace pk {

struct TransformerLayerW {
    const float *n1g, *n1b, *n2g, *n2b, *wqkv, *bqkv, *wo, *bo, *w1, *b1, *w2, *b2;
};

class TransformerEncoder {
  public:
    TransformerEncoder(const std::string &weights_path, const std::string &prefix, const pk_transformer_config &c, int device);
    ~TransformerEncoder();
    void forward(const float *x_host, int B, int T, float *y_host);
    // x[B*T][hidden] on the device, in place, enqueued on `s` (never synchronises)
    void forward_dev(float *x, int B, int T, hipStream_t s);
    pk_transformer_config cfg;

  private:
    int device_ = -1, hdp_ = 0, dp_ = 0;      // padded head dim / padded model width seen by the attention kernel
    hipStream_t stream_ = nullptr;
    std::vector<void *> allocs_;
    std::vector<TransformerLayerW> layers_;
    const float *fin_g_ = nullptr, *fin_b_ = nullptr;
    DevBuf x_, n_, qkv_, ctx_, h_, att_scratch_;
    const float *upload(const float *h, size_t n);
};

}  // namespace pk
