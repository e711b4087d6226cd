// parakeet.cpp_amd/csrc/common.hpp -- error plumbing and small HIP helpers shared by the engine.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/parakeet_amd.h"

namespace pk {

struct Error : std::runtime_error {
    pk_status code;
    Error(pk_status c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};

[[noreturn]] inline void fail(pk_status code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define PK_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            ::pk::fail(PK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define PK_CHECK_LAUNCH() PK_HIP(hipGetLastError())

// Device buffer that only grows (allocation stays out of the steady-state path).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) PK_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        PK_HIP(hipMalloc(&p, bytes));
        cap = bytes;
    }
    void release() {                      // give the memory back (the next reserve allocates afresh)
        if (p) PK_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace pk
