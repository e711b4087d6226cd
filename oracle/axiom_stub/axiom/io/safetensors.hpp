// oracle/axiom_stub/axiom/io/safetensors.hpp -- TEST INFRASTRUCTURE ONLY.  See ../axiom.hpp.
// Reads a safetensors file (8-byte little-endian header length, JSON header, raw tensor bytes) into name -> Tensor, which is
// what the reference does with `axiom::io::safetensors::load(path)` (transcribe.hpp:62, nemotron.cpp:19, eou.cpp:107).
// F32 tensors are kept; F64 / I64 / I32 / BF16 / F16 are converted to float32 (the reference's files are all fp32,
// scripts/convert_nemo.py:501).
#pragma once
#include <cstdio>
#include <fstream>

#include "../axiom.hpp"

namespace axiom::io::safetensors {

namespace detail {
struct P {
    const char *p, *e;
    void ws() {
        while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    [[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("axiom stand-in: safetensors header: ") + m); }
    void expect(char c) {
        ws();
        if (p >= e || *p != c) fail("unexpected character");
        ++p;
    }
    std::string str() {
        expect('"');
        std::string s;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                if (*p == 'u') {
                    if (e - p < 5) fail("bad escape");
                    p += 4;
                    s.push_back('?');
                } else {
                    s.push_back(*p);
                }
                ++p;
            } else {
                s.push_back(*p++);
            }
        }
        if (p >= e) fail("unterminated string");
        ++p;
        return s;
    }
    int64_t integer() {
        ws();
        int64_t v = 0;
        bool any = false, neg = false;
        if (p < e && *p == '-') {
            neg = true;
            ++p;
        }
        while (p < e && *p >= '0' && *p <= '9') {
            v = v * 10 + (*p++ - '0');
            any = true;
        }
        if (!any) fail("integer expected");
        return neg ? -v : v;
    }
    void skip(int depth = 0) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p >= e) fail("truncated");
        if (*p == '"') {
            str();
        } else if (*p == '{' || *p == '[') {
            const char close = *p == '{' ? '}' : ']';
            const bool obj = *p == '{';
            ++p;
            ws();
            if (p < e && *p == close) {
                ++p;
                return;
            }
            for (;;) {
                if (obj) {
                    str();
                    expect(':');
                }
                skip(depth + 1);
                ws();
                if (p < e && *p == ',') {
                    ++p;
                    continue;
                }
                expect(close);
                return;
            }
        } else {
            while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;
        }
    }
};
inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t s = (h >> 15) & 1u, ex = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (ex == 0) v = std::ldexp((float)m, -24);
    else if (ex == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m + 1024u), (int)ex - 25);
    return s ? -v : v;
}
}  // namespace detail

inline std::map<std::string, Tensor> load(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("axiom stand-in: cannot open safetensors file: " + path);
    f.seekg(0, std::ios::end);
    const uint64_t fsize = (uint64_t)f.tellg();
    f.seekg(0);
    unsigned char lenb[8];
    if (fsize < 8 || !f.read(reinterpret_cast<char *>(lenb), 8)) throw std::runtime_error("axiom stand-in: safetensors: truncated");
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | lenb[i];
    if (hlen > fsize - 8) throw std::runtime_error("axiom stand-in: safetensors: header length beyond file");
    std::string hdr((size_t)hlen, '\0');
    f.read(hdr.data(), (std::streamsize)hlen);
    std::vector<unsigned char> data((size_t)(fsize - 8 - hlen));
    f.read(reinterpret_cast<char *>(data.data()), (std::streamsize)data.size());

    std::map<std::string, Tensor> out;
    detail::P ps{hdr.data(), hdr.data() + hdr.size()};
    ps.expect('{');
    ps.ws();
    if (ps.p < ps.e && *ps.p == '}') return out;
    for (;;) {
        const std::string name = ps.str();
        ps.expect(':');
        if (name == "__metadata__") {
            ps.skip();
        } else {
            std::string dtype;
            Shape shape;
            int64_t b = -1, e = -1;
            ps.expect('{');
            for (;;) {
                const std::string key = ps.str();
                ps.expect(':');
                if (key == "dtype") {
                    dtype = ps.str();
                } else if (key == "shape") {
                    ps.expect('[');
                    ps.ws();
                    if (*ps.p == ']') {
                        ++ps.p;
                    } else {
                        for (;;) {
                            int64_t d = ps.integer();
                            if (d < 0) ps.fail("negative dimension");
                            shape.push_back((size_t)d);
                            ps.ws();
                            if (*ps.p == ',') {
                                ++ps.p;
                                continue;
                            }
                            ps.expect(']');
                            break;
                        }
                    }
                } else if (key == "data_offsets") {
                    ps.expect('[');
                    b = ps.integer();
                    ps.expect(',');
                    e = ps.integer();
                    ps.expect(']');
                } else {
                    ps.skip();
                }
                ps.ws();
                if (ps.p < ps.e && *ps.p == ',') {
                    ++ps.p;
                    continue;
                }
                ps.expect('}');
                break;
            }
            if (b < 0 || e < b || (uint64_t)e > data.size()) throw std::runtime_error("axiom stand-in: safetensors: bad offsets of " + name);
            const size_t n = Tensor::numel_of(shape);
            Tensor t(shape, DType::Float32);
            const unsigned char *src = data.data() + b;
            const size_t bytes = (size_t)(e - b);
            auto need = [&](size_t es) {
                if (bytes != n * es) throw std::runtime_error("axiom stand-in: safetensors: size mismatch of " + name);
            };
            if (dtype == "F32") {
                need(4);
                std::memcpy(t.raw(), src, bytes);
            } else if (dtype == "F64") {
                need(8);
                for (size_t i = 0; i < n; ++i) {
                    double v;
                    std::memcpy(&v, src + 8 * i, 8);
                    t.fdata()[i] = (float)v;
                }
            } else if (dtype == "I64") {
                need(8);
                for (size_t i = 0; i < n; ++i) {
                    int64_t v;
                    std::memcpy(&v, src + 8 * i, 8);
                    t.fdata()[i] = (float)v;
                }
            } else if (dtype == "I32") {
                need(4);
                for (size_t i = 0; i < n; ++i) {
                    int32_t v;
                    std::memcpy(&v, src + 4 * i, 4);
                    t.fdata()[i] = (float)v;
                }
            } else if (dtype == "BF16" || dtype == "F16") {
                need(2);
                for (size_t i = 0; i < n; ++i) {
                    uint16_t v;
                    std::memcpy(&v, src + 2 * i, 2);
                    t.fdata()[i] = dtype == "BF16" ? detail::bf16_to_f32(v) : detail::f16_to_f32(v);
                }
            } else {
                throw std::runtime_error("axiom stand-in: safetensors: unsupported dtype " + dtype + " of " + name);
            }
            out.emplace(name, t);
        }
        ps.ws();
        if (ps.p < ps.e && *ps.p == ',') {
            ++ps.p;
            continue;
        }
        ps.expect('}');
        break;
    }
    return out;
}

}  // namespace axiom::io::safetensors
