// parakeet/transcribe.hpp -- the drop-in high-level API: parakeet::Transcriber / TDTTranscriber / transcribe() /
// to_gpu(), source-compatible with the reference's include/parakeet/transcribe.hpp:23-299, implemented on the
// MI355X engine's C ABI (include/parakeet_amd.h) instead of axiom tensors.
//
// Differences a caller can observe:
//  * the `const axiom::Tensor &samples` overloads become (const float *pcm, size_t n) / std::vector<float> -- the
//    reference itself uses (const float*, size_t) for raw PCM in read_audio() and transcribe_chunk();
//  * there is no CPU execution path: transcribe() places the model on GPU 0 on first use if to_gpu() was not called;
//  * audio files: RIFF/WAVE only (any sample rate: resampled to 16 kHz with the reference's sinc resampler; no FLAC/MP3/OGG);
//  * weights are loaded strictly (a missing / mis-shaped tensor throws instead of being ignored);
//  * boost_phrases / boost_score work as in the reference (the ContextTrie and the boosted argmax run on the GPU); the
//    Tensor-level free functions of phrase_boost.hpp (ctc_greedy_decode_boosted(Tensor, ...)) have C-ABI counterparts instead:
//    pk_set_boost_tokens / pk_set_boost_phrases + pk_ctc_decode / pk_tdt_decode;
//  * TDTTranscriber decodes with blank id = vocab_size - 1 (what the reference's CLI passes, src/main.cpp:252).  The reference CLASS
//    itself calls tdt_greedy_decode with its default blank_id = 1024 whatever the vocabulary (transcribe.hpp:274-279, tdt.hpp:78-80),
//    which differs for the 8193-token 600M vocabulary: construct TDTTranscriber(weights, vocab, config, /*blank_id=*/1024) to
//    reproduce the class literally;
//  * new: transcribe_batch() -- clips of ANY lengths are packed into ragged batches and decoded together, each clip bit-identical to
//    its single-clip result (the reference is batch-1 only; "batch inference" is a roadmap item, README.md:513).
// Errors surface as std::runtime_error with the reference's trigger conditions (unreadable vocab / audio, ...).
#pragma once

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/parakeet_amd.h"
#include "config.hpp"
#include "timestamp.hpp"
#include "vocab.hpp"

namespace parakeet {

struct TranscribeResult {
    std::string text;
    std::vector<int> token_ids;
    std::vector<TimestampedToken> timestamped_tokens;   // filled when timestamps = true
    std::vector<WordTimestamp> word_timestamps;
};

enum class Decoder { CTC, TDT };

struct TranscribeOptions {
    Decoder decoder = Decoder::TDT;
    bool timestamps = false;
    std::vector<std::string> boost_phrases;   // ContextTrie::build of these phrases biases the greedy argmax (phrase_boost.hpp)
    float boost_score = 5.0f;
};

namespace detail {

inline void check(pk_status st) {
    if (st == PK_OK) return;
    char msg[1024];
    pk_last_error(msg, sizeof msg);
    throw std::runtime_error(msg);
}

class Engine {   // owns one pk_model; shared by Transcriber and TDTTranscriber
  public:
    Engine(const std::string &weights_path, const std::string &vocab_path, const pk_config &cfg)
        : weights_path_(weights_path), vocab_path_(vocab_path), cfg_(cfg) {
        check(pk_model_load(weights_path.c_str(), vocab_path.empty() ? nullptr : vocab_path.c_str(), &cfg, &m_));
        tok_ = Tokenizer(m_);
    }
    ~Engine() {
        pk_group_free(g_);
        pk_model_free(m_);
    }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;

    void to_gpu(int device = 0) {
        check(pk_model_to_gpu(m_, device));
        on_gpu_ = true;
    }

    // New (the reference is single-device, README.md:513): one replica per GPU of this node, clips dealt to the devices in batches
    // (pk_group: one replica and one host thread + two-stream pipeline per device, no collective).  Empty list = every visible device.
    void to_all_gpus(const std::vector<int> &devices = {}) {
        pk_group_free(g_);
        g_ = nullptr;
        check(pk_group_create(weights_path_.c_str(), vocab_path_.empty() ? nullptr : vocab_path_.c_str(), &cfg_,
                              devices.empty() ? nullptr : devices.data(), (int)devices.size(), &g_));
    }
    int num_gpus() const { return g_ ? pk_group_size(g_) : (on_gpu_ ? 1 : 0); }

    std::vector<TranscribeResult> run(const std::vector<std::pair<const float *, size_t>> &clips, const TranscribeOptions &opts) {
        if (!g_ && !on_gpu_) to_gpu(0);
        std::vector<float> pcm;
        std::vector<int64_t> offsets{0};
        for (auto &c : clips) {
            pcm.insert(pcm.end(), c.first, c.first + c.second);
            offsets.push_back((int64_t)pcm.size());
        }
        pk_options o{};
        o.decoder = opts.decoder == Decoder::CTC ? PK_DECODER_CTC : PK_DECODER_TDT;
        o.timestamps = opts.timestamps ? 1 : 0;
        std::vector<const char *> phrases;
        for (auto &p : opts.boost_phrases) phrases.push_back(p.c_str());
        o.boost_phrases = phrases.data();
        o.n_boost_phrases = (int32_t)phrases.size();
        o.boost_score = opts.boost_score;
        pk_result *res = nullptr;
        if (g_) check(pk_group_transcribe_pcm(g_, pcm.data(), offsets.data(), (int)clips.size(), &o, &res));
        else check(pk_transcribe_pcm(m_, pcm.data(), offsets.data(), (int)clips.size(), &o, &res));
        std::vector<TranscribeResult> out(clips.size());
        for (size_t i = 0; i < clips.size(); ++i) {
            const pk_result &r = res[i];
            out[i].text = r.text ? r.text : "";
            out[i].token_ids.assign(r.token_ids, r.token_ids + r.n_tokens);
            if (opts.timestamps) {
                for (int k = 0; k < r.n_tokens; ++k)
                    out[i].timestamped_tokens.push_back({r.token_ids[k], r.start_frame[k], r.end_frame[k], r.confidence[k]});
                for (int k = 0; k < r.n_words; ++k)
                    out[i].word_timestamps.push_back({r.words[k].word, r.words[k].start, r.words[k].end, r.words[k].confidence});
            }
        }
        pk_results_free(res, (int)clips.size());
        return out;
    }

    TranscribeResult run_file(const std::string &audio_path, const TranscribeOptions &opts) {
        float *pcm = nullptr;
        int64_t n = 0;
        int sr = 0;
        check(pk_read_audio(audio_path.c_str(), 16000, &pcm, &n, &sr));   // read_audio(path): decode, downmix, resample to 16 kHz
        struct Free { float *p; ~Free() { pk_free(p); } } guard{pcm};
        return run({{pcm, (size_t)n}}, opts)[0];
    }

    const Tokenizer &tokenizer() const { return tok_; }
    pk_model *handle() { return m_; }

  private:
    std::string weights_path_, vocab_path_;
    pk_config cfg_;
    pk_model *m_ = nullptr;
    pk_group *g_ = nullptr;
    Tokenizer tok_;
    bool on_gpu_ = false;
};

}  // namespace detail

/// parakeet::Transcriber t("model.safetensors", "vocab.txt");  t.to_gpu();  auto r = t.transcribe("audio.wav");
class Transcriber {
  public:
    Transcriber(const std::string &weights_path, const std::string &vocab_path, const TDTCTCConfig &config = make_110m_config())
        : config_(config),
          eng_(weights_path, vocab_path,
               detail::flatten(config.encoder, config.prediction, config.joint, config.durations, config.ctc_vocab_size, "tdt_joint_.",
                               false, 1024)) {}   // blank_id: the decoders' default 1024 (tdt.hpp / ctc.hpp), as Transcriber uses it

    void to_gpu() { eng_.to_gpu(0); }
    void to_gpu(int device) { eng_.to_gpu(device); }
    /// New: a replica on every GPU of the node (or on `devices`); transcribe_batch() then shards its clips over them.
    void to_all_gpus(const std::vector<int> &devices = {}) { eng_.to_all_gpus(devices); }
    int num_gpus() const { return eng_.num_gpus(); }

    TranscribeResult transcribe(const std::string &audio_path, Decoder decoder = Decoder::TDT, bool timestamps = false) {
        return eng_.run_file(audio_path, options(decoder, timestamps));
    }
    TranscribeResult transcribe(const std::string &audio_path, const TranscribeOptions &opts) { return eng_.run_file(audio_path, opts); }
    TranscribeResult transcribe(const float *pcm, size_t n, Decoder decoder = Decoder::TDT, bool timestamps = false) {
        return eng_.run({{pcm, n}}, options(decoder, timestamps))[0];
    }
    TranscribeResult transcribe(const float *pcm, size_t n, const TranscribeOptions &opts) { return eng_.run({{pcm, n}}, opts)[0]; }
    TranscribeResult transcribe(const std::vector<float> &samples, Decoder decoder = Decoder::TDT, bool timestamps = false) {
        return transcribe(samples.data(), samples.size(), decoder, timestamps);
    }
    TranscribeResult transcribe(const std::vector<float> &samples, const TranscribeOptions &opts) {
        return transcribe(samples.data(), samples.size(), opts);
    }
    /// New: many clips at once; clips of equal length share a GPU batch.
    std::vector<TranscribeResult> transcribe_batch(const std::vector<std::vector<float>> &clips, const TranscribeOptions &opts = {}) {
        std::vector<std::pair<const float *, size_t>> v;
        for (auto &c : clips) v.emplace_back(c.data(), c.size());
        return eng_.run(v, opts);
    }

    const Tokenizer &tokenizer() const { return eng_.tokenizer(); }
    const TDTCTCConfig &config() const { return config_; }
    pk_model *model() { return eng_.handle(); }   // the engine handle (the reference returns its ParakeetTDTCTC module tree)

  private:
    static TranscribeOptions options(Decoder d, bool ts) {
        TranscribeOptions o;
        o.decoder = d;
        o.timestamps = ts;
        return o;
    }
    TDTCTCConfig config_;
    detail::Engine eng_;
};

/// TDT-only models (no CTC head), e.g. the 600M multilingual checkpoint.
class TDTTranscriber {
  public:
    // blank_id < 0 (default): vocab_size - 1, what the reference CLI passes (main.cpp:252).  blank_id = 1024 reproduces the reference
    // class literally: its transcribe() calls tdt_greedy_decode with the decoder's default blank (transcribe.hpp:274-279, tdt.hpp:78-80).
    TDTTranscriber(const std::string &weights_path, const std::string &vocab_path, const TDTConfig &config = make_tdt_600m_config(),
                   int blank_id = -1)
        : config_(config),
          eng_(weights_path, vocab_path,
               detail::flatten(config.encoder, config.prediction, config.joint, config.durations, 0, "joint_.", false,
                               blank_id >= 0 ? blank_id : config.joint.vocab_size - 1)) {}

    void to_gpu() { eng_.to_gpu(0); }
    void to_gpu(int device) { eng_.to_gpu(device); }
    void to_all_gpus(const std::vector<int> &devices = {}) { eng_.to_all_gpus(devices); }
    int num_gpus() const { return eng_.num_gpus(); }

    TranscribeResult transcribe(const std::string &audio_path, bool timestamps = false) { return eng_.run_file(audio_path, options(timestamps)); }
    TranscribeResult transcribe(const std::string &audio_path, const TranscribeOptions &opts) { return eng_.run_file(audio_path, tdt(opts)); }
    TranscribeResult transcribe(const float *pcm, size_t n, bool timestamps = false) { return eng_.run({{pcm, n}}, options(timestamps))[0]; }
    TranscribeResult transcribe(const float *pcm, size_t n, const TranscribeOptions &opts) { return eng_.run({{pcm, n}}, tdt(opts))[0]; }
    TranscribeResult transcribe(const std::vector<float> &samples, bool timestamps = false) {
        return transcribe(samples.data(), samples.size(), timestamps);
    }
    std::vector<TranscribeResult> transcribe_batch(const std::vector<std::vector<float>> &clips, const TranscribeOptions &opts = {}) {
        std::vector<std::pair<const float *, size_t>> v;
        for (auto &c : clips) v.emplace_back(c.data(), c.size());
        return eng_.run(v, tdt(opts));
    }

    const Tokenizer &tokenizer() const { return eng_.tokenizer(); }
    const TDTConfig &config() const { return config_; }
    pk_model *model() { return eng_.handle(); }

  private:
    static TranscribeOptions options(bool ts) {
        TranscribeOptions o;
        o.timestamps = ts;
        return o;
    }
    static TranscribeOptions tdt(TranscribeOptions o) {
        o.decoder = Decoder::TDT;
        return o;
    }
    TDTConfig config_;
    detail::Engine eng_;
};

}  // namespace parakeet
