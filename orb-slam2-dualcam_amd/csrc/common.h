// common.h -- internal helpers shared by the HIP translation units of libdcs_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
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

// Scratch of the host-buffer entry points (what the reference's Tracking / LocalMapping / LoopClosing threads call
// concurrently, dcs_abi.h "threading"). Everything a call needs comes out of a PER-THREAD context that lives across calls:
// a non-blocking stream (nothing touches the legacy null stream, so calls of different threads never serialise on it), a
// grow-only device arena and a grow-only pinned staging arena (no hipMalloc / hipFree / hipHostMalloc on the steady-state
// path: those cost more than the kernels of a 1000 x 1000 match). Uploads and downloads are asynchronous copies through the
// pinned arena; finish() is the call's single synchronisation and scatters the downloads into the caller's buffers.
struct ThreadArena {
    struct Block { char* p; size_t cap; };
    std::vector<Block> dev, pin;
    size_t dev_used = 0, pin_used = 0, dev_peak = 0, pin_peak = 0;      // offsets inside the LAST block / peak demand of a call
    size_t dev_call = 0, pin_call = 0;
    hipStream_t stream = nullptr;
    int device = -1;
    bool in_call = false;                                // a Scratch of this thread is alive: a second begin() would rewind ITS arena
    ~ThreadArena();
    void release();
    int begin();                                         // start of a call: rewind, coalesce fragmented blocks into one
    void* take(bool pinned, size_t bytes);               // 256-byte aligned; nullptr on allocation failure
};
ThreadArena& thread_arena();

struct Scratch {
    ThreadArena& a;
    hipStream_t st = nullptr;
    struct Pending { void* dst; const void* src; size_t bytes; };
    std::vector<Pending> pending;
    int rc0;
    bool owner = false, finished = false, touched = false;       // touched: something was enqueued on the stream
    // One Scratch per thread at a time: a nested one (an entry point calling another entry point) would rewind the outer call's arena
    // under its in-flight copies -- it fails with DCS_ERR_INVALID instead.
    Scratch() : a(thread_arena())
    {
        if (a.in_call) { set_error("nested scratch arena on one thread"); rc0 = DCS_ERR_INVALID; return; }
        a.in_call = owner = true;
        rc0 = a.begin(); st = a.stream;
    }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    // A call that returns early -- a failed launch, a late validation error, an allocation failure -- has not run finish(): its uploads
    // and kernels may still be reading the pinned staging and the arena that the thread's NEXT call rewinds, so the stream is drained here.
    ~Scratch()
    {
        if (owner) {
            if (!finished && touched && st) (void)hipStreamSynchronize(st);
            a.in_call = false;
        }
    }
    template <typename T> int alloc(T** out, size_t n) {
        if (rc0) return rc0;
        touched = true;                                   // the caller is about to launch on / copy through this memory
        void* p = a.take(false, std::max<size_t>(n, 1) * sizeof(T));
        if (!p) { set_error("device scratch: out of memory"); return DCS_ERR_HIP; }
        *out = (T*)p; return DCS_OK;
    }
    template <typename T> int upload(T** out, const T* src, size_t n) {
        int rc = alloc(out, n);
        if (rc || n == 0) return rc;
        void* h = a.take(true, n * sizeof(T));
        if (!h) { set_error("pinned scratch: out of memory"); return DCS_ERR_HIP; }
        memcpy(h, src, n * sizeof(T));
        DCS_HIP(hipMemcpyAsync(*out, h, n * sizeof(T), hipMemcpyHostToDevice, st));
        return DCS_OK;
    }
    template <typename T> int upload(const T** out, const T* src, size_t n) {
        T* d = nullptr;
        int rc = upload(&d, src, n);
        *out = d;
        return rc;
    }
    // host -> an already carved device range (several host arrays into one device array)
    template <typename T> int upload_into(T* d_dst, const T* src, size_t n) {
        if (rc0) return rc0;
        if (n == 0) return DCS_OK;
        touched = true;
        void* h = a.take(true, n * sizeof(T));
        if (!h) { set_error("pinned scratch: out of memory"); return DCS_ERR_HIP; }
        memcpy(h, src, n * sizeof(T));
        DCS_HIP(hipMemcpyAsync(d_dst, h, n * sizeof(T), hipMemcpyHostToDevice, st));
        return DCS_OK;
    }
    // asynchronous device -> caller copy: lands in `dst` at the next finish()
    template <typename T> int download(T* dst, const T* d_src, size_t n) {
        if (rc0) return rc0;
        if (n == 0) return DCS_OK;
        touched = true;
        void* h = a.take(true, n * sizeof(T));
        if (!h) { set_error("pinned scratch: out of memory"); return DCS_ERR_HIP; }
        DCS_HIP(hipMemcpyAsync(h, d_src, n * sizeof(T), hipMemcpyDeviceToHost, st));
        pending.push_back({dst, h, n * sizeof(T)});
        return DCS_OK;
    }
    int download_bytes(void* dst, const void* d_src, size_t bytes) { return download((char*)dst, (const char*)d_src, bytes); }
    int finish() {
        if (rc0) return rc0;
        DCS_HIP(hipStreamSynchronize(st));
        for (const Pending& q : pending) memcpy(q.dst, q.src, q.bytes);
        pending.clear();
        finished = true; touched = false;                 // a call may go on after a finish() (two-phase entry points): tracked again from here
        return DCS_OK;
    }
};

}  // namespace dcs
