// parakeet/diarize.hpp -- ASR + Sortformer fusion of the drop-in facade (reference: include/parakeet/diarize.hpp:17-80,
// src/diarize.cpp:10-106).  Host logic over Transcriber and Sortformer; both models run on the MI355X engine.
#pragma once

#include <string>
#include <unordered_map>
#include <vector>

#include "sortformer.hpp"
#include "transcribe.hpp"

namespace parakeet {

struct DiarizedWord {                    // diarize.hpp:19-25
    std::string word;
    float start = 0.0f;
    float end = 0.0f;
    int speaker_id = -1;                 // -1 = no overlapping segment
    float confidence = 1.0f;
};

struct DiarizedResult {                  // diarize.hpp:27-32
    std::string text;
    std::vector<DiarizedWord> words;
    std::vector<DiarizationSegment> segments;
    std::vector<WordTimestamp> word_timestamps;
};

/// Assign speaker ids to words by maximum total temporal overlap (src/diarize.cpp:10-48).  Ties between speakers: the reference
/// iterates an unordered_map (unspecified order); here the lowest speaker id wins.
inline std::vector<DiarizedWord> diarize_transcription(const std::vector<WordTimestamp> &words, const std::vector<DiarizationSegment> &segments) {
    std::vector<DiarizedWord> result;
    result.reserve(words.size());
    for (const auto &w : words) {
        DiarizedWord dw;
        dw.word = w.word; dw.start = w.start; dw.end = w.end; dw.confidence = w.confidence;
        std::unordered_map<int, float> overlap;
        int max_spk = -1;
        for (const auto &seg : segments) {
            const float o = std::min(w.end, seg.end) - std::max(w.start, seg.start);
            if (o > 0.0f) { overlap[seg.speaker_id] += o; max_spk = std::max(max_spk, seg.speaker_id); }
        }
        float best = 0.0f;
        for (int spk = 0; spk <= max_spk; ++spk) {
            auto it = overlap.find(spk);
            if (it != overlap.end() && it->second > best) { best = it->second; dw.speaker_id = spk; }
        }
        result.push_back(std::move(dw));
    }
    return result;
}

class DiarizedTranscriber {              // diarize.hpp:55-78
  public:
    DiarizedTranscriber(const std::string &asr_weights, const std::string &sortformer_weights, const std::string &vocab_path,
                        const TDTCTCConfig &config = make_110m_config(), const SortformerConfig &sf_config = make_sortformer_117m_config())
        : transcriber_(asr_weights, vocab_path, config), sortformer_(sortformer_weights, sf_config) {}

    void to_gpu() {
        transcriber_.to_gpu();
        sortformer_.to_gpu();
    }

    DiarizedResult transcribe(const std::string &audio_path, Decoder decoder = Decoder::TDT) {
        float *pcm = nullptr;
        int64_t n = 0;
        int sr = 0;
        detail::check(pk_read_audio(audio_path.c_str(), 16000, &pcm, &n, &sr));
        struct Free { float *p; ~Free() { pk_free(p); } } guard{pcm};
        return transcribe(pcm, (size_t)n, decoder);
    }
    DiarizedResult transcribe(const float *pcm, size_t n, Decoder decoder = Decoder::TDT) {
        auto asr = transcriber_.transcribe(pcm, n, decoder, /*timestamps=*/true);      // src/diarize.cpp:78-79
        auto segments = sortformer_.diarize_pcm(pcm, n);                                // :81-89 (128 mels, no normalisation)
        DiarizedResult r;
        r.text = asr.text;
        r.words = diarize_transcription(asr.word_timestamps, segments);                 // :92-93
        r.segments = std::move(segments);
        r.word_timestamps = std::move(asr.word_timestamps);
        return r;
    }
    DiarizedResult transcribe(const std::vector<float> &samples, Decoder decoder = Decoder::TDT) { return transcribe(samples.data(), samples.size(), decoder); }

  private:
    Transcriber transcriber_;
    Sortformer sortformer_;
};

}  // namespace parakeet
