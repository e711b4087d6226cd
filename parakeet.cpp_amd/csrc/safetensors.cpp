// parakeet.cpp_amd/csrc/safetensors.cpp -- minimal safetensors reader: 8-byte LE header length, a JSON
// object {name: {dtype, shape, data_offsets}}, then the raw little-endian tensor bytes.
#include "safetensors.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>

#include "common.hpp"

namespace pk {
namespace {

struct Json {  // just enough JSON for the header: objects, arrays, strings, integers
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    void need(char c) { if (!eat(c)) fail(PK_ERR_WEIGHTS, "safetensors header: expected '%c'", c); }
    std::string str() {
        ws();
        if (p >= end || *p != '"') fail(PK_ERR_WEIGHTS, "safetensors header: expected string");
        ++p;
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                case 'n': s += '\n'; break;
                case 't': s += '\t'; break;
                case 'u':
                    if (end - p < 5) fail(PK_ERR_WEIGHTS, "safetensors header: truncated \\u escape");
                    s += '?'; p += 4; break;
                default: s += *p;
                }
                ++p;
            } else {
                s += *p++;
            }
        }
        if (p >= end) fail(PK_ERR_WEIGHTS, "safetensors header: unterminated string");
        ++p;
        return s;
    }
    int64_t integer() {
        ws();
        bool neg = false;
        if (p < end && *p == '-') { neg = true; ++p; }
        if (p >= end || *p < '0' || *p > '9') fail(PK_ERR_WEIGHTS, "safetensors header: expected integer");
        int64_t v = 0;
        int digits = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            if (++digits > 18) fail(PK_ERR_WEIGHTS, "safetensors header: integer too long");
            v = v * 10 + (*p++ - '0');
        }
        return neg ? -v : v;
    }
    void skip(int depth = 0) {  // skip any value
        if (depth > 64) fail(PK_ERR_WEIGHTS, "safetensors header: nesting too deep");
        ws();
        if (p >= end) return;
        if (*p == '"') { (void)str(); return; }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = (*p == '{') ? '}' : ']';
            ++p;
            if (eat(close)) return;
            do {
                if (open == '{') { (void)str(); need(':'); }
                skip(depth + 1);
            } while (eat(','));
            need(close);
            return;
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
    }
};

}  // namespace

SafeTensors::SafeTensors(const std::string &path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) fail(PK_ERR_IO, "Cannot open weights file: %s", path.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { ::close(fd); fail(PK_ERR_IO, "Cannot stat weights file: %s", path.c_str()); }
    map_len_ = (size_t)st.st_size;
    map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED) { map_ = nullptr; fail(PK_ERR_IO, "mmap failed: %s", path.c_str()); }
    parse(static_cast<const uint8_t *>(map_), map_len_);
}

SafeTensors::SafeTensors(const void *data, size_t len, bool borrow) {
    if (!data || len < 8) fail(PK_ERR_WEIGHTS, "safetensors: buffer of %zu bytes is too short", len);
    if (borrow) {                       // the caller keeps the image alive for the lifetime of this view (pk_group_create)
        parse(static_cast<const uint8_t *>(data), len);
        return;
    }
    own_.assign(static_cast<const uint8_t *>(data), static_cast<const uint8_t *>(data) + len);
    parse(own_.data(), own_.size());
}

void SafeTensors::parse(const uint8_t *base, size_t len) {
    const size_t map_len_ = len;
    uint64_t hlen;
    memcpy(&hlen, base, 8);
    if (hlen > map_len_ - 8) fail(PK_ERR_WEIGHTS, "safetensors: header length %llu exceeds file", (unsigned long long)hlen);
    const uint8_t *payload = base + 8 + hlen;
    const size_t payload_len = map_len_ - 8 - hlen;
    Json j{reinterpret_cast<const char *>(base + 8), reinterpret_cast<const char *>(base + 8 + hlen)};
    j.need('{');
    if (!j.eat('}')) {
        do {
            const std::string name = j.str();
            j.need(':');
            if (name == "__metadata__") { j.skip(); continue; }
            HostTensor t;
            int64_t off0 = 0, off1 = 0;
            j.need('{');
            do {
                const std::string key = j.str();
                j.need(':');
                if (key == "dtype") {
                    t.dtype = j.str();
                } else if (key == "shape") {
                    j.need('[');
                    if (!j.eat(']')) {
                        do {
                            const int64_t dim = j.integer();
                            if (dim < 0) fail(PK_ERR_WEIGHTS, "safetensors: negative extent in the shape of %s", name.c_str());
                            t.shape.push_back(dim);
                        } while (j.eat(','));
                        j.need(']');
                    }
                } else if (key == "data_offsets") {
                    j.need('[');
                    off0 = j.integer();
                    j.need(',');
                    off1 = j.integer();
                    j.need(']');
                } else {
                    j.skip();
                }
            } while (j.eat(','));
            j.need('}');
            if (off0 < 0 || off1 < off0 || (size_t)off1 > payload_len) fail(PK_ERR_WEIGHTS, "safetensors: bad offsets for %s", name.c_str());
            t.data = payload + off0;
            t.nbytes = (size_t)(off1 - off0);
            {   // element count without overflow: no tensor can hold more elements than the payload has bytes
                uint64_t prod = 1;
                for (auto dim : t.shape) {
                    if (dim != 0 && prod > (uint64_t)payload_len / (uint64_t)dim) fail(PK_ERR_WEIGHTS, "safetensors: shape of %s exceeds the file", name.c_str());
                    prod *= (uint64_t)dim;
                }
            }
            {   // size / shape consistency for every dtype with a known element size (unknown dtypes are never read)
                static const struct { const char *n; size_t b; } sizes[] = {{"F64", 8}, {"F32", 4}, {"F16", 2}, {"BF16", 2}, {"I64", 8}, {"I32", 4}, {"I16", 2},
                                                                          {"I8", 1}, {"U8", 1}, {"BOOL", 1}, {"U16", 2}, {"U32", 4}, {"U64", 8}, {"F8_E4M3", 1}, {"F8_E5M2", 1}};
                for (const auto &e : sizes)
                    if (t.dtype == e.n && (size_t)t.numel() * e.b != t.nbytes) fail(PK_ERR_WEIGHTS, "safetensors: %s size/shape mismatch", name.c_str());
            }
            tensors_.emplace(name, std::move(t));
        } while (j.eat(','));
        j.need('}');
    }
}

SafeTensors::~SafeTensors() {
    if (map_) munmap(map_, map_len_);
}

const HostTensor *SafeTensors::find(const std::string &name) const {
    auto it = tensors_.find(name);
    return it == tensors_.end() ? nullptr : &it->second;
}

}  // namespace pk
