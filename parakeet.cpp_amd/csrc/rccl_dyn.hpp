// parakeet.cpp_amd/csrc/rccl_dyn.hpp -- RCCL resolved at run time (dlopen), so that libparakeet_amd.so has NO link-time dependency on
// librccl: a single-GPU user never loads it, and a host without RCCL gets a clear error only from the one entry point that needs it
// (pk_group_verify_exchange).  The declarations come from <rccl/rccl.h> (types and enums only; no symbol of it is referenced).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <string>

namespace pk {

struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;
};

// nullptr (and *why filled) when librccl cannot be loaded.  Search order: PK_RCCL_LIB; the librccl that sits NEXT TO the HIP runtime this
// library is actually running on (dladdr of a HIP entry point) -- a process may hold several ROCm stacks (PyTorch wheels bundle their own
// libamdhip64 / librccl), and an RCCL from another stack than the HIP runtime in use fails in ncclCommInitAll ("no ROCm-capable device");
// a bare soname would resolve to whichever copy happens to be loaded already; then $ROCM_PATH/lib, /opt/rocm/lib, and the soname last.
inline const RcclApi *rccl_api(std::string *why) {
    // resolved once, thread-safely (function-local static initialised by a lambda: C++11 guarantees a single, synchronised run)
    struct Loaded { RcclApi api; bool ok = false; std::string err; };
    static const Loaded L = [] {
        Loaded R;
        RcclApi &api = R.api;
        std::string &err = R.err;
        bool &ok = R.ok;
        std::string cand[8];
        int n = 0;
        if (const char *e = getenv("PK_RCCL_LIB")) cand[n++] = e;
        {
            Dl_info info;
            if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
                const std::string hip = info.dli_fname;
                const size_t slash = hip.rfind('/');
                if (slash != std::string::npos) {
                    cand[n++] = hip.substr(0, slash) + "/librccl.so.1";
                    cand[n++] = hip.substr(0, slash) + "/librccl.so";
                }
            }
        }
        if (const char *r = getenv("ROCM_PATH")) cand[n++] = std::string(r) + "/lib/librccl.so.1";
        cand[n++] = "/opt/rocm/lib/librccl.so.1";
        cand[n++] = "librccl.so.1";
        cand[n++] = "librccl.so";
        void *h = nullptr;
        for (int i = 0; i < n && !h; ++i) {
            h = dlopen(cand[i].c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) api.path = cand[i];
            else {
                const char *e = dlerror();                  // read ONCE: dlerror() clears the message it returns
                err += std::string(err.empty() ? "" : "; ") + (e ? e : cand[i].c_str());
            }
        }
        if (h) {
            bool all = true;
            auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) { all = false; err = std::string("librccl lacks ") + name; } return p; };
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
            api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
            ok = all;
        }
        return R;
    }();
    if (!L.ok && why) *why = L.err;
    return L.ok ? &L.api : nullptr;
}

}  // namespace pk
