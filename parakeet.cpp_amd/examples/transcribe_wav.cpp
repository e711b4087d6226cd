// examples/transcribe_wav.cpp -- the reference README's three-line usage, unchanged, on the MI355X engine:
//     parakeet::Transcriber t("model.safetensors", "vocab.txt");  t.to_gpu();  auto r = t.transcribe("audio.wav");
// usage: transcribe_wav <model.safetensors> <vocab.txt> <audio.wav> [ctc|tdt] [--timestamps] [--boost PHRASE]... [--boost-score N] [--all-gpus]
// (--boost / --boost-score as the reference CLI, src/main.cpp:23-25)
// Prints one JSON object (text, token ids, optional word timestamps) -- tests/test_gpu_facade.py parses it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include <parakeet/parakeet.hpp>

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s model.safetensors vocab.txt audio.wav [ctc|tdt] [--timestamps]\n", argv[0]);
        return 2;
    }
    try {
        parakeet::TranscribeOptions opts;
        bool all_gpus = false;
        for (int i = 4; i < argc; ++i) {
            if (!std::strcmp(argv[i], "--all-gpus")) all_gpus = true;      // new: a replica on every GPU of the node (pk_group, RCCL)
            if (!std::strcmp(argv[i], "ctc")) opts.decoder = parakeet::Decoder::CTC;
            if (!std::strcmp(argv[i], "--timestamps")) opts.timestamps = true;
            if (!std::strcmp(argv[i], "--boost") && i + 1 < argc) opts.boost_phrases.push_back(argv[++i]);
            else if (!std::strcmp(argv[i], "--boost-score") && i + 1 < argc) opts.boost_score = std::strtof(argv[++i], nullptr);
        }
        parakeet::Transcriber t(argv[1], argv[2]);
        if (all_gpus) t.to_all_gpus();
        else t.to_gpu();
        const auto r = t.transcribe(std::string(argv[3]), opts);
        std::printf("{\"text\": \"");
        for (char c : r.text) { if (c == '"' || c == '\\') std::putchar('\\'); std::putchar(c); }
        std::printf("\", \"token_ids\": [");
        for (size_t i = 0; i < r.token_ids.size(); ++i) std::printf("%s%d", i ? ", " : "", r.token_ids[i]);
        std::printf("], \"words\": [");
        for (size_t i = 0; i < r.word_timestamps.size(); ++i)
            std::printf("%s[\"%s\", %.2f, %.2f]", i ? ", " : "", r.word_timestamps[i].word.c_str(), r.word_timestamps[i].start, r.word_timestamps[i].end);
        std::printf("]}\n");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
