// parakeet.cpp_amd/csrc/text.cpp -- see text.hpp.
#include "text.hpp"

#include <algorithm>
#include <fstream>

#include "common.hpp"

namespace pk {

static const char kWordMark[] = "\xe2\x96\x81";   // U+2581, SentencePiece word-boundary marker

static bool has_mark(const std::string &s, size_t at = 0) { return s.compare(at, 3, kWordMark) == 0; }

void Tokenizer::load(const std::string &vocab_path) {
    std::ifstream in(vocab_path);
    if (!in) fail(PK_ERR_IO, "Cannot open vocab file: %s", vocab_path.c_str());
    pieces_.clear();
    lookup_.clear();
    longest_ = 0;
    for (std::string line; std::getline(in, line);) {
        const size_t tab = line.find('\t');             // "piece<TAB>score" or a bare piece; blank lines are skipped
        if (tab != std::string::npos) pieces_.emplace_back(line, 0, tab);
        else if (!line.empty()) pieces_.push_back(line);
    }
}

std::string Tokenizer::decode(const std::vector<int> &ids) const {
    std::string joined;
    for (int id : ids) {
        if (id >= 0 && id < (int)pieces_.size()) joined += pieces_[id];
        else joined += "[" + std::to_string(id) + "]";  // out-of-range ids print as [id]
    }
    std::string text;
    text.reserve(joined.size());
    for (size_t i = 0; i < joined.size();) {
        if (i + 3 <= joined.size() && has_mark(joined, i)) { text += ' '; i += 3; }
        else text += joined[i++];
    }
    if (!text.empty() && text.front() == ' ') text.erase(text.begin());
    return text;
}

std::vector<int> Tokenizer::encode(const std::string &text) const {
    std::vector<int> ids;
    if (pieces_.empty() || text.empty()) return ids;
    if (lookup_.empty())
        for (size_t i = 0; i < pieces_.size(); ++i) {
            lookup_[pieces_[i]] = (int)i;               // later duplicates win, like operator[] in the reference
            longest_ = std::max(longest_, pieces_[i].size());
        }
    std::string s = kWordMark;                          // leading marker, spaces become markers
    for (char c : text) { if (c == ' ') s += kWordMark; else s += c; }
    for (size_t pos = 0; pos < s.size();) {
        size_t take = 0;
        int id = -1;
        for (size_t len = std::min(longest_, s.size() - pos); len >= 1; --len) {   // greedy longest match
            auto it = lookup_.find(s.substr(pos, len));
            if (it != lookup_.end()) { take = len; id = it->second; break; }
        }
        if (id >= 0) { ids.push_back(id); pos += take; }
        else ++pos;                                     // unknown byte: skipped
    }
    return ids;
}

std::vector<WordTimestamp> group_timestamps(const std::vector<TimestampedToken> &tokens,
                                            const std::vector<std::string> &pieces, bool sentences) {
    std::vector<WordTimestamp> words;
    if (tokens.empty()) return words;
    std::string cur;
    int first = tokens[0].start_frame, last = tokens[0].end_frame;
    float conf = 1.0f;
    auto flush = [&]() { words.push_back({cur, frame_to_seconds(first), frame_to_seconds(last), conf}); };
    for (const auto &t : tokens) {
        if (t.token_id < 0 || t.token_id >= (int)pieces.size()) continue;
        const std::string &p = pieces[t.token_id];
        const bool starts = p.size() >= 3 && has_mark(p);
        if (starts && !cur.empty()) {
            flush();
            cur.clear();
            first = t.start_frame;
            conf = 1.0f;
        }
        cur += starts ? p.substr(3) : p;
        last = t.end_frame;
        conf = std::min(conf, t.confidence);
    }
    if (!cur.empty()) flush();
    if (!sentences) return words;
    std::vector<WordTimestamp> out;
    std::string sent;
    float s0 = 0.0f, s1 = 0.0f, sc = 1.0f;
    for (const auto &w : words) {
        if (sent.empty()) s0 = w.start; else sent += ' ';
        sent += w.word;
        s1 = w.end;
        sc = std::min(sc, w.confidence);
        const char e = w.word.empty() ? 0 : w.word.back();
        if (e == '.' || e == '?' || e == '!') { out.push_back({sent, s0, s1, sc}); sent.clear(); sc = 1.0f; }
    }
    if (!sent.empty()) out.push_back({sent, s0, s1, sc});
    return out;
}

}  // namespace pk
