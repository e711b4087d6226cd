// oracle/ref_text_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" wrappers around the REAL reference tokenizer and timestamp grouping
// (/root/reference/src/vocab.cpp, src/timestamp.cpp -- the only hot-path-adjacent
// reference translation units that compile without the un-vendored `axiom`).
// Built by oracle/Makefile from the reference sources where they lie, into
// oracle/_ref/libpk_ref_text.so.  Used to pin the product's host-side
// detokenisation / word grouping against the reference itself.
#include <cstring>
#include <string>
#include <vector>

#include "parakeet/timestamp.hpp"
#include "parakeet/vocab.hpp"

using namespace parakeet;

extern "C" {

void *ref_tok_load(const char *path) {
    auto *t = new Tokenizer();
    try {
        t->load(path);
    } catch (...) {
        delete t;
        return nullptr;
    }
    return t;
}
void ref_tok_free(void *t) { delete static_cast<Tokenizer *>(t); }
int ref_tok_vocab_size(void *t) { return (int)static_cast<Tokenizer *>(t)->vocab_size(); }

// returns needed length (excluding NUL); writes at most cap-1 bytes + NUL
int ref_tok_decode(void *t, const int *ids, int n, char *out, int cap) {
    std::vector<int> v(ids, ids + n);
    std::string s = static_cast<Tokenizer *>(t)->decode(v);
    if (cap > 0) {
        int c = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
        std::memcpy(out, s.data(), c);
        out[c] = 0;
    }
    return (int)s.size();
}
int ref_tok_encode(void *t, const char *text, int *ids, int cap) {
    auto v = static_cast<Tokenizer *>(t)->encode(text);
    for (int i = 0; i < (int)v.size() && i < cap; ++i) ids[i] = v[i];
    return (int)v.size();
}

// group_timestamps on parallel arrays; words are returned '\n'-joined.
int ref_group_timestamps(void *t, const int *ids, const int *start, const int *end, const float *conf, int n,
                         int sentences, char *words, int cap, float *wstart, float *wend, float *wconf, int wcap) {
    std::vector<TimestampedToken> toks(n);
    for (int i = 0; i < n; ++i) toks[i] = {ids[i], start[i], end[i], conf[i]};
    auto w = group_timestamps(toks, static_cast<Tokenizer *>(t)->pieces(),
                              sentences ? TimestampMode::Sentences : TimestampMode::Words);
    std::string joined;
    for (size_t i = 0; i < w.size(); ++i) {
        if (i) joined += '\n';
        joined += w[i].word;
        if ((int)i < wcap) { wstart[i] = w[i].start; wend[i] = w[i].end; wconf[i] = w[i].confidence; }
    }
    if (cap > 0) {
        int c = (int)joined.size() < cap - 1 ? (int)joined.size() : cap - 1;
        std::memcpy(words, joined.data(), c);
        words[c] = 0;
    }
    return (int)w.size();
}
float ref_frame_to_seconds(int f) { return frame_to_seconds(f); }
}
