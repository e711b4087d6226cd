// examples/parakeet_cli.cpp -- the reference's command line (src/main.cpp:12-37, :642-727) on the MI355X engine: same positional
// arguments, --model types and options; every model type runs through the drop-in facade classes.
//   parakeet_cli <model.safetensors> <audio.wav> [--model TYPE] [--ctc|--tdt] [--vocab PATH] [--timestamps] [--boost PHRASE]...
//                [--boost-score N] [--sortformer-weights PATH] [--latency N] [--streaming] [--gpu]
// Differences: --gpu is accepted and implied (there is no CPU path); --features (a .npy of pre-computed features) is not supported.
#include <chrono>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include <parakeet/parakeet.hpp>

using namespace parakeet;
using Clock = std::chrono::high_resolution_clock;

static void usage(const char *prog) {
    std::cerr << "Usage: " << prog << " <model.safetensors> <audio.wav> [options]\n"
              << "  --model TYPE   tdt-ctc-110m (default), tdt-600m, rnnt-600m, eou-120m, nemotron-600m, sortformer, diarized\n"
              << "  --ctc | --tdt  decoder (default: TDT)\n"
              << "  --boost PHRASE (repeatable), --boost-score N (default 5.0)\n"
              << "  --vocab PATH, --sortformer-weights PATH, --timestamps, --streaming, --latency N (0/1/6/13), --gpu\n";
}

static void print_result(const TranscribeResult &r, bool timestamps, double ms) {
    std::cout << "Inference: " << std::fixed << std::setprecision(1) << ms << " ms\n";
    std::cout << "\n--- Transcription ---\n" << r.text << "\n";
    std::cout << "Tokens (" << r.token_ids.size() << "):";
    for (int id : r.token_ids) std::cout << ' ' << id;
    std::cout << "\n";
    if (timestamps) {
        std::cout << "\n--- Word timestamps ---\n";
        for (const auto &w : r.word_timestamps)
            std::cout << "  [" << std::fixed << std::setprecision(2) << w.start << "s - " << w.end << "s] (" << std::setprecision(3) << w.confidence << ") " << w.word << "\n";
    }
}

template <class T>
static int run_stream(T &t, const std::string &audio_path, bool timestamps) {
    t.to_gpu();
    const auto audio = read_audio(audio_path);
    const size_t chunk = 2560;                                      // 160 ms at 16 kHz (main.cpp run_*_streaming)
    const auto t0 = Clock::now();
    for (size_t off = 0; off < audio.samples.size(); off += chunk) {
        const size_t n = std::min(chunk, audio.samples.size() - off);
        const std::string piece = t.transcribe_chunk(audio.samples.data() + off, n);
        if (!piece.empty()) std::cout << piece << std::flush;
    }
    const double ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    std::cout << "\nStreaming: " << std::fixed << std::setprecision(1) << ms << " ms for " << audio.duration << " s\n";
    std::cout << "\n--- Transcription ---\n" << t.get_text() << "\n";
    if (timestamps)
        for (const auto &tk : t.get_timestamped_tokens())
            std::cout << "  token " << tk.token_id << " frames [" << tk.start_frame << ", " << tk.end_frame << "]\n";
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { usage(argv[0]); return 1; }
    try {
        const std::string weights = argv[1], audio_path = argv[2];
        std::string model = "tdt-ctc-110m", vocab, sf_weights;
        bool use_ctc = false, timestamps = false;
        int latency = 0;
        std::vector<std::string> boost;
        float boost_score = 5.0f;
        for (int i = 3; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--model" && i + 1 < argc) model = argv[++i];
            else if (a == "--ctc") use_ctc = true;
            else if (a == "--tdt") use_ctc = false;
            else if (a == "--gpu" || a == "--streaming") {}
            else if (a == "--timestamps") timestamps = true;
            else if (a == "--latency" && i + 1 < argc) latency = std::stoi(argv[++i]);
            else if (a == "--vocab" && i + 1 < argc) vocab = argv[++i];
            else if (a == "--sortformer-weights" && i + 1 < argc) sf_weights = argv[++i];
            else if (a == "--boost" && i + 1 < argc) boost.push_back(argv[++i]);
            else if (a == "--boost-score" && i + 1 < argc) boost_score = std::stof(argv[++i]);
            else if (a == "--features") { std::cerr << "Error: --features is not supported by this build\n"; return 1; }
            else { std::cerr << "Unknown option: " << a << "\n"; usage(argv[0]); return 1; }
        }
        TranscribeOptions opts;
        opts.decoder = use_ctc ? Decoder::CTC : Decoder::TDT;
        opts.timestamps = timestamps;
        opts.boost_phrases = boost;
        opts.boost_score = boost_score;
        std::cout << "Loading model: " << model << std::endl;
        if (model == "tdt-ctc-110m") {
            Transcriber t(weights, vocab);
            t.to_gpu();
            if (!boost.empty()) std::cout << "Phrase boost: " << boost.size() << " phrases\n";
            const auto t0 = Clock::now();
            const auto r = t.transcribe(audio_path, opts);
            print_result(r, timestamps, std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
        } else if (model == "tdt-600m") {
            TDTTranscriber t(weights, vocab);
            t.to_gpu();
            const auto t0 = Clock::now();
            const auto r = t.transcribe(audio_path, opts);
            print_result(r, timestamps, std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
        } else if (model == "rnnt-600m") {                          // run_rnnt_600m (main.cpp:296-376): ParakeetRNNT + rnnt_greedy_decode
            pk_config cfg;
            detail::check(pk_config_preset("rnnt-600m", &cfg));
            pk_model *m = nullptr;
            detail::check(pk_model_load(weights.c_str(), vocab.empty() ? nullptr : vocab.c_str(), &cfg, &m));
            struct Free { pk_model *m; ~Free() { pk_model_free(m); } } guard{m};
            detail::check(pk_model_to_gpu(m, 0));
            const auto audio = read_audio(audio_path);
            const int64_t off[2] = {0, (int64_t)audio.samples.size()};
            pk_options o{};
            o.decoder = PK_DECODER_TDT;                             // the joint loop; the rnnt_head flag of the preset selects rnnt_greedy_decode
            o.timestamps = timestamps ? 1 : 0;
            pk_result *res = nullptr;
            const auto t0 = Clock::now();
            detail::check(pk_transcribe_pcm(m, audio.samples.data(), off, 1, &o, &res));
            TranscribeResult r;
            r.text = res[0].text ? res[0].text : "";
            r.token_ids.assign(res[0].token_ids, res[0].token_ids + res[0].n_tokens);
            for (int i = 0; i < res[0].n_words; ++i) r.word_timestamps.push_back({res[0].words[i].word, res[0].words[i].start, res[0].words[i].end, res[0].words[i].confidence});
            pk_results_free(res, 1);
            print_result(r, timestamps, std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
        } else if (model == "eou-120m") {
            StreamingTranscriber t(weights, vocab);
            return run_stream(t, audio_path, timestamps);
        } else if (model == "nemotron-600m") {
            NemotronTranscriber t(weights, vocab, make_nemotron_600m_config(latency));
            return run_stream(t, audio_path, timestamps);
        } else if (model == "sortformer") {
            Sortformer sf(weights);
            sf.to_gpu();
            const auto audio = read_audio(audio_path);
            const auto t0 = Clock::now();
            const auto segs = sf.diarize_pcm(audio.samples.data(), audio.samples.size());
            std::cout << "Diarization: " << std::fixed << std::setprecision(1) << std::chrono::duration<double, std::milli>(Clock::now() - t0).count() << " ms\n";
            std::cout << "\n--- Speaker Segments (" << segs.size() << " segments) ---\n";
            for (const auto &s : segs) std::cout << "  Speaker " << s.speaker_id << ": [" << std::fixed << std::setprecision(2) << s.start << "s - " << s.end << "s]\n";
        } else if (model == "diarized") {
            if (sf_weights.empty()) { std::cerr << "Error: --sortformer-weights required for diarized mode\n"; return 1; }
            DiarizedTranscriber dt(weights, sf_weights, vocab);
            dt.to_gpu();
            const auto r = dt.transcribe(audio_path, use_ctc ? Decoder::CTC : Decoder::TDT);
            std::cout << "\n--- Diarized transcription ---\n";
            int cur = -2;
            for (const auto &w : r.words) {
                if (w.speaker_id != cur) { cur = w.speaker_id; std::cout << "\nSpeaker " << cur << ":"; }
                std::cout << ' ' << w.word;
            }
            std::cout << "\n";
        } else {
            std::cerr << "Unknown model type: " << model << "\n";
            usage(argv[0]);
            return 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
