// parakeet.cpp_amd/csrc/safetensors.hpp -- read-only safetensors container (what the reference loads with
// axiom::io::safetensors::load, transcribe.hpp:62).  The file is mmap'd; tensors are views into the mapping.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace pk {

struct HostTensor {
    std::string dtype;            // "F32" (convert_nemo.py:501 casts everything to fp32)
    std::vector<int64_t> shape;
    const uint8_t *data = nullptr;
    size_t nbytes = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
    const float *f32() const { return reinterpret_cast<const float *>(data); }
};

class SafeTensors {
  public:
    explicit SafeTensors(const std::string &path);   // throws pk::Error(PK_ERR_IO / PK_ERR_WEIGHTS)
    // the same container from memory: copied (e.g. an image received by a broadcast), or -- borrow -- a view of an image the caller keeps alive
    SafeTensors(const void *data, size_t len, bool borrow = false);
    ~SafeTensors();
    SafeTensors(const SafeTensors &) = delete;
    SafeTensors &operator=(const SafeTensors &) = delete;
    const HostTensor *find(const std::string &name) const;
    const std::map<std::string, HostTensor> &tensors() const { return tensors_; }
    // the whole container as it lies in memory (the file mapping or the owned copy): lets several views share one image
    const void *image_base() const { return map_ ? map_ : (own_.empty() ? nullptr : own_.data()); }
    size_t image_bytes() const { return map_ ? map_len_ : own_.size(); }

  private:
    void parse(const uint8_t *base, size_t len);
    void *map_ = nullptr;
    size_t map_len_ = 0;
    std::vector<uint8_t> own_;
    std::map<std::string, HostTensor> tensors_;
};

}  // namespace pk
