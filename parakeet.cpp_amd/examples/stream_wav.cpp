// examples/stream_wav.cpp -- the reference's streaming usage (README "Streaming": NemotronTranscriber t(weights, vocab, cfg);
// t.to_gpu(); while (...) text += t.transcribe_chunk(pcm, n);) on the MI355X engine.
// usage: stream_wav <model.safetensors> <vocab.txt> <audio.wav> <layers> [chunk_samples=2560] [latency_frames=1]
// (<layers>: number of encoder layers of the weights file -- tests use a 2-layer cut of the nemotron-600m architecture).
// Prints one JSON object: the chunk texts joined, get_text(), and the token ids with their absolute frames.
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include <parakeet/parakeet.hpp>

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s model.safetensors vocab.txt audio.wav layers [chunk_samples] [latency_frames]\n", argv[0]);
        return 2;
    }
    try {
        const int chunk = argc > 5 ? std::atoi(argv[5]) : 2560, latency = argc > 6 ? std::atoi(argv[6]) : 1;
        auto cfg = parakeet::make_nemotron_600m_config(latency);
        cfg.encoder.num_layers = std::atoi(argv[4]);
        parakeet::NemotronTranscriber t(argv[1], argv[2], cfg);
        t.to_gpu();
        float *pcm = nullptr;
        int64_t n = 0;
        int sr = 0;
        parakeet::detail::check(pk_read_wav(argv[3], &pcm, &n, &sr));
        std::string joined;
        for (int64_t off = 0; off + chunk <= n; off += chunk) joined += t.transcribe_chunk(pcm + off, (size_t)chunk);
        pk_free(pcm);
        auto esc = [](const std::string &s) { std::string o; for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; } return o; };
        std::printf("{\"joined\": \"%s\", \"text\": \"%s\", \"tokens\": [", esc(joined).c_str(), esc(t.get_text()).c_str());
        const auto &tt = t.get_timestamped_tokens();
        for (size_t i = 0; i < tt.size(); ++i) std::printf("%s[%d, %d, %d]", i ? ", " : "", tt[i].token_id, tt[i].start_frame, tt[i].end_frame);
        std::printf("]}\n");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
