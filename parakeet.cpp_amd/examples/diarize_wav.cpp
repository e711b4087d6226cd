// examples/diarize_wav.cpp -- the reference's diarization usage on the MI355X engine (include/parakeet/diarize.hpp:44-52,
// src/main.cpp:497-538 run_sortformer / :545-640 run_diarized):
//     parakeet::DiarizedTranscriber dt("asr.safetensors", "sortformer.safetensors", "vocab.txt");  dt.to_gpu();  dt.transcribe("a.wav");
// usage: diarize_wav <sortformer.safetensors> <audio.wav>                              -> segments only
//        diarize_wav <sortformer.safetensors> <audio.wav> <asr.safetensors> <vocab.txt> -> speaker-attributed words
// Prints one JSON object -- tests/test_gpu_facade.py parses it.
#include <cstdio>
#include <iostream>

#include <parakeet/parakeet.hpp>

static void print_segments(const std::vector<parakeet::DiarizationSegment> &segs) {
    std::printf("\"segments\": [");
    for (size_t i = 0; i < segs.size(); ++i) std::printf("%s[%d, %.9g, %.9g]", i ? ", " : "", segs[i].speaker_id, segs[i].start, segs[i].end);
    std::printf("]");
}

int main(int argc, char **argv) {
    if (argc != 3 && argc != 5) {
        std::fprintf(stderr, "usage: %s sortformer.safetensors audio.wav [asr.safetensors vocab.txt]\n", argv[0]);
        return 2;
    }
    try {
        if (argc == 3) {
            parakeet::Sortformer sf(argv[1]);
            sf.to_gpu();
            float *pcm = nullptr;
            int64_t n = 0;
            int sr = 0;
            parakeet::detail::check(pk_read_audio(argv[2], 16000, &pcm, &n, &sr));
            const auto segs = sf.diarize_pcm(pcm, (size_t)n);
            pk_free(pcm);
            std::printf("{");
            print_segments(segs);
            std::printf("}\n");
        } else {
            parakeet::DiarizedTranscriber dt(argv[3], argv[1], argv[4]);
            dt.to_gpu();
            const auto r = dt.transcribe(std::string(argv[2]));
            std::printf("{");
            print_segments(r.segments);
            std::printf(", \"words\": [");
            for (size_t i = 0; i < r.words.size(); ++i)
                std::printf("%s[\"%s\", %d, %.9g, %.9g]", i ? ", " : "", r.words[i].word.c_str(), r.words[i].speaker_id, r.words[i].start, r.words[i].end);
            std::printf("]}\n");
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
