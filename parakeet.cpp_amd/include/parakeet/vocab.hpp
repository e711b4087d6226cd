// parakeet/vocab.hpp -- Tokenizer view of the drop-in facade (reference: include/parakeet/vocab.hpp).
// The piece table lives inside the pk_model; this class forwards to the C ABI.
#pragma once
#include <string>
#include <vector>

#include "../../../include/parakeet_amd.h"

namespace parakeet {

class Tokenizer {
  public:
    Tokenizer() = default;
    explicit Tokenizer(const pk_model *m) : m_(m) {}
    bool loaded() const { return m_ && pk_vocab_size(m_) > 1; }
    size_t vocab_size() const { return m_ ? (size_t)pk_vocab_size(m_) : 1; }
    std::string decode(const std::vector<int> &token_ids) const {
        if (!m_) return {};
        std::vector<int32_t> ids(token_ids.begin(), token_ids.end());
        const int need = pk_detokenize(m_, ids.data(), (int)ids.size(), nullptr, 0);
        std::string out((size_t)(need > 0 ? need : 0), '\0');
        if (need > 0) pk_detokenize(m_, ids.data(), (int)ids.size(), out.data(), need + 1);
        return out;
    }
    std::vector<int> encode(const std::string &text) const {
        if (!m_) return {};
        const int n = pk_tokenize(m_, text.c_str(), nullptr, 0);
        std::vector<int32_t> ids((size_t)(n > 0 ? n : 0));
        if (n > 0) pk_tokenize(m_, text.c_str(), ids.data(), n);
        return std::vector<int>(ids.begin(), ids.end());
    }

  private:
    const pk_model *m_ = nullptr;
};

}  // namespace parakeet
