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
    // take() opens a NEW hipMalloc block whenever the current one is full: consecutive take()s are contiguous only inside one block. A caller
    // that treats a run of arrays as one range (one memset, one copy) reserves the run first: after reserve(n) the next take()s whose 256-byte
    // padded sizes add up to at most n come from ONE block, back to back. false = allocation failure
    bool reserve(bool pinned, size_t bytes);
    bool same_block(bool pinned, const void* p, const void* q) const;   // both inside one block of the arena
};
ThreadArena& thread_arena();
int streams_share_queue(hipStream_t a, hipStream_t b, bool* shared, bool* conclusive = nullptr);   // measured: does work on b wait for work on a? *conclusive = false when a's backlog kept the probe from starting (common.cpp)
int create_stream_apart(hipStream_t* out, const hipStream_t* avoid, int n_avoid, bool* apart);   // non-blocking stream on another hardware queue than avoid[]
hipError_t create_cu_range_stream(hipStream_t* s, int first, int count);   // CUs [first, first + count) of the CU-mask bit order

struct Scratch {
    ThreadArena& a;
    hipStream_t raw_st = nullptr;
    // Transfers are COALESCED: uploads are staged in pinned memory and enqueued together the first time the stream is used (the proxy
    // below converts to hipStream_t and flushes first) -- staging blocks that are adjacent in pinned AND in device memory go as one copy;
    // downloads are enqueued by finish(), device ranges that lie within 4 KB of each other as one copy. A seam call with six small
    // arrays in and three out (dcs_match_bf) costs two or three DMA operations instead of nine: ~40 us of a 120 us call.
    struct StreamProxy {
        Scratch* s;
        operator hipStream_t() const { s->flush_uploads(); return s->raw_st; }
    } st{this};
    struct Up { char* d; const char* h; size_t bytes, padded; };
    struct Down { void* dst; const char* d_src; size_t bytes; };
    std::vector<Up> ups;
    std::vector<Down> downs;
    int rc0;
    int rc_sticky = DCS_OK;                                      // an upload that failed inside the stream proxy: reported by finish()
    bool owner = false, finished = false, touched = false;       // touched: something was enqueued on the stream
    // One Scratch per thread at a time: a nested one (an entry point calling another entry point) would rewind the outer call's arena
    // under its in-flight copies -- it fails with DCS_ERR_INVALID instead.
    Scratch() : a(thread_arena())
    {
        if (a.in_call) { set_error("nested scratch arena on one thread"); rc0 = DCS_ERR_INVALID; return; }
        a.in_call = owner = true;
        rc0 = a.begin(); raw_st = a.stream;
    }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    // A call that returns early -- a failed launch, a late validation error, an allocation failure -- has not run finish(): its uploads
    // and kernels may still be reading the pinned staging and the arena that the thread's NEXT call rewinds, so the stream is drained here.
    ~Scratch()
    {
        if (owner) {
            if (touched && raw_st) (void)hipStreamSynchronize(raw_st);      // finish() clears `touched`; work enqueued after a finish() sets it again
            a.in_call = false;
        }
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    // the next alloc()s of `bytes` in all (each padded to 256) are carved back to back from one block of the device arena
    int reserve(size_t bytes) {
        if (rc0) return rc0;
        if (!a.reserve(false, bytes)) { set_error("device scratch: out of memory"); return DCS_ERR_HIP; }
        return DCS_OK;
    }
    template <typename T> int alloc(T** out, size_t n) {
        if (rc0) return rc0;
        touched = true;                                   // the caller is about to launch on / copy through this memory
        void* p = a.take(false, std::max<size_t>(n, 1) * sizeof(T));
        if (!p) { set_error("device scratch: out of memory"); return DCS_ERR_HIP; }
        *out = (T*)p; return DCS_OK;
    }
    // device array + pinned staging of the same size, ONE pending copy: the caller fills the staging (several host arrays at their
    // final offsets) before the stream is first used
    template <typename T> int stage(T** d_out, T** h_out, size_t n) {
        int rc = alloc(d_out, n);
        if (rc) return rc;
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        void* h = a.take(true, bytes);
        if (!h) { set_error("pinned scratch: out of memory"); return DCS_ERR_HIP; }
        *h_out = (T*)h;
        ups.push_back({(char*)*d_out, (const char*)h, bytes, padded(bytes)});
        return DCS_OK;
    }
    template <typename T> int upload(T** out, const T* src, size_t n) {
        int rc = alloc(out, n);
        if (rc || n == 0) return rc;
        return upload_into(*out, src, n);
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
        ups.push_back({(char*)d_dst, (const char*)h, n * sizeof(T), padded(n * sizeof(T))});
        return DCS_OK;
    }
    int flush_uploads() {
        if (ups.empty() || rc_sticky) { ups.clear(); return rc_sticky; }
        size_t i = 0;
        while (i < ups.size()) {
            size_t j = i, bytes = ups[i].bytes;
            // the arena hands out 256-byte padded ranges back to back: neighbours in both address spaces merge (padding travels along)
            while (j + 1 < ups.size() && ups[j + 1].d == ups[j].d + ups[j].padded && ups[j + 1].h == ups[j].h + ups[j].padded) {
                bytes = (size_t)(ups[j + 1].d - ups[i].d) + ups[j + 1].bytes;
                ++j;
            }
            const hipError_t e = hipMemcpyAsync(ups[i].d, ups[i].h, bytes, hipMemcpyHostToDevice, raw_st);
            if (e != hipSuccess) { set_error("scratch upload: %s", hipGetErrorString(e)); rc_sticky = DCS_ERR_HIP; break; }
            i = j + 1;
        }
        ups.clear();
        return rc_sticky;
    }
    // device -> caller copy: enqueued by finish(), lands in `dst` there
    template <typename T> int download(T* dst, const T* d_src, size_t n) {
        if (rc0) return rc0;
        if (n == 0) return DCS_OK;
        touched = true;
        downs.push_back({(void*)dst, (const char*)d_src, n * sizeof(T)});
        return DCS_OK;
    }
    int download_bytes(void* dst, const void* d_src, size_t bytes) { return download((char*)dst, (const char*)d_src, bytes); }
    int finish() {
        if (rc0) return rc0;
        if (flush_uploads()) return rc_sticky;
        // one copy per cluster of device ranges (sorted by address; a gap of up to 4 KB -- alignment padding, a small array in between --
        // is cheaper to carry than a second DMA operation)
        std::vector<size_t> order(downs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return downs[x].d_src < downs[y].d_src; });
        struct Seg { const char* d0; const char* d1; char* h; };
        std::vector<Seg> segs;
        std::vector<size_t> seg_of(downs.size());
        for (size_t k = 0; k < order.size(); ++k) {
            const Down& q = downs[order[k]];
            // (ranges of two arena blocks never merge: what lies between two hipMalloc allocations is not ours to read)
            if (!segs.empty() && q.d_src <= segs.back().d1 + 4096 && a.same_block(false, segs.back().d0, q.d_src)) segs.back().d1 = std::max(segs.back().d1, q.d_src + q.bytes);
            else segs.push_back({q.d_src, q.d_src + q.bytes, nullptr});
            seg_of[order[k]] = segs.size() - 1;
        }
        for (Seg& sg : segs) {
            sg.h = (char*)a.take(true, (size_t)(sg.d1 - sg.d0));
            if (!sg.h) { set_error("pinned scratch: out of memory"); return DCS_ERR_HIP; }
            DCS_HIP(hipMemcpyAsync(sg.h, sg.d0, (size_t)(sg.d1 - sg.d0), hipMemcpyDeviceToHost, raw_st));
        }
        DCS_HIP(hipStreamSynchronize(raw_st));
        for (size_t i = 0; i < downs.size(); ++i) { const Seg& sg = segs[seg_of[i]]; memcpy(downs[i].dst, sg.h + (downs[i].d_src - sg.d0), downs[i].bytes); }
        downs.clear();
        finished = true; touched = false;                 // a call may go on after a finish() (two-phase entry points): tracked again from here
        return DCS_OK;
    }
};

}  // namespace dcs
