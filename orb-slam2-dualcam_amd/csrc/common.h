// common.h -- internal helpers shared by the HIP translation units of libdcs_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

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

}  // namespace dcs
