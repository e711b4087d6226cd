// parakeet.cpp_amd/csrc/text.hpp -- host-side text post-processing (not kernels): the SentencePiece piece
// table (reference src/vocab.cpp:10-117) and word/sentence grouping of timestamped tokens
// (src/timestamp.cpp:24-111).  Re-written from the behaviour, pinned against the real reference
// objects by tests/test_text_vs_reference.py.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

namespace pk {

struct TimestampedToken {   // include/parakeet/timestamp.hpp:11-16
    int token_id;
    int start_frame;
    int end_frame;
    float confidence = 1.0f;
};
struct WordTimestamp {      // include/parakeet/timestamp.hpp:18-23
    std::string word;
    float start, end;
    float confidence = 1.0f;
};

constexpr float kFrameDurationS = 0.08f;   // 8 * 160 / 16000 (timestamp.hpp:31)
inline float frame_to_seconds(int frame) { return static_cast<float>(frame) * kFrameDurationS; }

class Tokenizer {
  public:
    void load(const std::string &vocab_path);                      // throws pk::Error(PK_ERR_IO)
    std::string decode(const std::vector<int> &ids) const;
    std::vector<int> encode(const std::string &text) const;
    bool loaded() const { return !pieces_.empty(); }
    size_t vocab_size() const { return pieces_.size() + 1; }       // + blank
    const std::vector<std::string> &pieces() const { return pieces_; }

  private:
    std::vector<std::string> pieces_;
    mutable std::unordered_map<std::string, int> lookup_;
    mutable size_t longest_ = 0;
};

std::vector<WordTimestamp> group_timestamps(const std::vector<TimestampedToken> &tokens,
                                            const std::vector<std::string> &pieces, bool sentences);

}  // namespace pk
