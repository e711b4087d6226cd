// parakeet.cpp_amd/csrc/engine.hpp -- the static execution plan behind the C ABI.
//
// Not a tensor library: a fixed pipeline for the Parakeet model family.  Weights are uploaded (and the few
// derived tables built) once in to_gpu(); activations live in a grow-only workspace; every stage is a
// short, fixed sequence of hand-written kernels on one HIP stream.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "kernels/kernels.hpp"
#include "safetensors.hpp"
#include "text.hpp"

namespace pk {

struct LayerW {   // reference names: SURVEY.md Appendix B / scripts/convert_nemo.py:134-184
    const float *ffn1_ng, *ffn1_nb, *ffn1_w1, *ffn1_b1, *ffn1_w2, *ffn1_b2;
    const float *ffn2_ng, *ffn2_nb, *ffn2_w1, *ffn2_b1, *ffn2_w2, *ffn2_b2;
    const float *att_ng, *att_nb, *wqkv, *bqkv, *wo, *bo, *wpos, *pos_u, *pos_v;
    const float *cv_ng, *cv_nb, *pw1_w, *pw1_b, *dw_w, *dw_b, *bn_g, *bn_b, *bn_mean, *bn_rstd, *pw2_w, *pw2_b;
    const float *fin_g, *fin_b;
};
struct SubW {
    const float *c1w, *c1b, *d1w, *d1b, *c2w, *c2b, *d2w, *d2b, *c3w, *c3b, *pw, *pb;
};
struct DecW {
    const float *embed;
    const float *wih[4], *bih[4], *whh[4];
    const float *g1;            // [V][4Hp] = W_ih0 * E[token] + b  (layer 0 input projection table)
    const float *we, *be, *wp, *bp, *wl, *bl, *wd, *bd;
    const float *ctc_w, *ctc_b;
};

struct ProfileSink;   // per-kernel hipEvent timing (engine.cpp)

class Model {
  public:
    Model(const std::string &weights_path, const std::string &vocab_path, const pk_config &cfg);
    ~Model();
    void to_gpu(int device);
    bool on_gpu() const { return device_ >= 0; }
    void require_gpu() const;

    pk_config cfg;
    Tokenizer tok;
    int device_ = -1;
    hipStream_t stream = nullptr;

    // device weights
    std::vector<void *> allocs_;
    MelTables mel{};
    SubW sub{};
    std::vector<LayerW> layers;
    DecW dec{};

    // stage drivers (device pointers, enqueue on `s`)
    void run_mel(const float *d_pcm, int B, int64_t n_samples, float *d_logmel, float *d_feats, hipStream_t s);

    // grow-only scratch shared by the host-buffer entry points
    DevBuf io_in, io_out, io_tmp;

    const float *upload(const float *host, size_t n);
    const float *upload_tensor(const std::string &name, std::vector<int64_t> expect_shape);
    float *dev_alloc(size_t n_floats);

  private:
    std::unique_ptr<SafeTensors> st_;
    void build_mel_tables();
    void upload_weights();
};

// thread-local error slot of the C ABI
void set_last_error(const std::string &msg);

}  // namespace pk
