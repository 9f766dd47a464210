// common.h -- internal helpers shared by the HIP translation units of libdcs_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dcs_abi.h"

namespace dcs {

void set_error(const char* fmt, ...);

#define DCS_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::dcs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return DCS_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define DCS_CHECK_LAUNCH() DCS_HIP(hipGetLastError())

int ensure_device();   // DCS_OK or DCS_ERR_NO_DEVICE (no CPU fallback anywhere)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int resize(size_t count) {
        if (count <= n) return DCS_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        DCS_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return DCS_OK;
    }
};

template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    int resize(size_t count) {
        if (count <= n) return DCS_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
        DCS_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
        n = count;
        return DCS_OK;
    }
};

// per-call device scratch of the host-buffer entry points (re-entrant: nothing is cached between calls)
struct Scratch {
    std::vector<void*> ptrs;
    Scratch() = default;
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T> int alloc(T** out, size_t n) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) { set_error("hipMalloc: %s", hipGetErrorString(e)); return DCS_ERR_HIP; }
        ptrs.push_back(p); *out = (T*)p; return DCS_OK;
    }
    template <typename T> int upload(T** out, const T* src, size_t n) {
        int rc = alloc(out, n);
        if (rc) return rc;
        if (n) DCS_HIP(hipMemcpy(*out, src, n * sizeof(T), hipMemcpyHostToDevice));
        return DCS_OK;
    }
    template <typename T> int upload(const T** out, const T* src, size_t n) {
        T* d = nullptr;
        int rc = upload(&d, src, n);
        *out = d;
        return rc;
    }
};

}  // namespace dcs
