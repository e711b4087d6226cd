// parakeet/timestamp.hpp -- timestamp records of the drop-in facade (reference: include/parakeet/timestamp.hpp).
#pragma once
#include <string>
#include <vector>

namespace parakeet {

struct TimestampedToken {
    int token_id;
    int start_frame;          // encoder frame
    int end_frame;            // inclusive
    float confidence = 1.0f;  // exp(log-prob) of the emitted token
};

struct WordTimestamp {
    std::string word;
    float start;              // seconds
    float end;
    float confidence = 1.0f;  // min over the word's tokens
};

// one encoder frame = 8 (subsampling) * 160 (hop) / 16000 Hz
constexpr float FRAME_DURATION_S = 0.08f;
inline float frame_to_seconds(int frame) { return static_cast<float>(frame) * FRAME_DURATION_S; }

enum class TimestampMode { Words, Sentences };

}  // namespace parakeet
