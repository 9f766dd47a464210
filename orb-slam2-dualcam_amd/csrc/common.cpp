// common.cpp -- error text, device probing, version.
#include "common.h"
#include "config.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dcs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s); libdcs_hip has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return DCS_ERR_NO_DEVICE;
    }
    return DCS_OK;
}

ThreadArena::~ThreadArena() { release(); }

void ThreadArena::release()
{
    for (Block& b : dev) (void)hipFree(b.p);
    for (Block& b : pin) (void)hipHostFree(b.p);
    dev.clear(); pin.clear();
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr; device = -1; dev_used = pin_used = 0;
}

int ThreadArena::begin()
{
    int d = 0;
    DCS_HIP(hipGetDevice(&d));
    if (device != d) release();
    device = d;
    if (!stream) DCS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    dev_peak = std::max(dev_peak, dev_call); pin_peak = std::max(pin_peak, pin_call);
    dev_call = pin_call = 0;
    // a previous call that outgrew its block left several behind: replace them by ONE of the peak size (nothing is in
    // flight: every call ends with a stream synchronisation)
    if (dev.size() > 1) { for (Block& b : dev) (void)hipFree(b.p); dev.clear(); }
    if (pin.size() > 1) { for (Block& b : pin) (void)hipHostFree(b.p); pin.clear(); }
    dev_used = pin_used = 0;
    return DCS_OK;
}

void* ThreadArena::take(bool pinned, size_t bytes)
{
    std::vector<Block>& blocks = pinned ? pin : dev;
    size_t& used = pinned ? pin_used : dev_used;
    size_t& call = pinned ? pin_call : dev_call;
    const size_t need = (bytes + 255) & ~(size_t)255;
    call += need;
    if (blocks.empty() || used + need > blocks.back().cap) {
        const size_t peak = pinned ? pin_peak : dev_peak;
        size_t cap = std::max<size_t>({need, peak + peak / 4, (size_t)1 << 20});
        if (!blocks.empty()) cap = std::max(cap, 2 * blocks.back().cap);
        Block b{nullptr, cap};
        const hipError_t e = pinned ? hipHostMalloc((void**)&b.p, cap, hipHostMallocDefault) : hipMalloc((void**)&b.p, cap);
        if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        blocks.push_back(b);
        used = 0;
    }
    void* p = blocks.back().p + used;
    used += need;
    return p;
}

bool ThreadArena::reserve(bool pinned, size_t bytes)
{
    std::vector<Block>& blocks = pinned ? pin : dev;
    size_t& used = pinned ? pin_used : dev_used;
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (!blocks.empty() && used + need <= blocks.back().cap) return true;
    // the same growth rule as take(), without handing anything out: the run starts at the top of a fresh block
    const size_t peak = pinned ? pin_peak : dev_peak;
    size_t cap = std::max<size_t>({need, peak + peak / 4, (size_t)1 << 20});
    if (!blocks.empty()) cap = std::max(cap, 2 * blocks.back().cap);
    Block b{nullptr, cap};
    const hipError_t e = pinned ? hipHostMalloc((void**)&b.p, cap, hipHostMallocDefault) : hipMalloc((void**)&b.p, cap);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    blocks.push_back(b);
    used = 0;
    return true;
}

bool ThreadArena::same_block(bool pinned, const void* p, const void* q) const
{
    const std::vector<Block>& blocks = pinned ? pin : dev;
    for (const Block& b : blocks) {
        const bool in_p = (const char*)p >= b.p && (const char*)p < b.p + b.cap, in_q = (const char*)q >= b.p && (const char*)q < b.p + b.cap;
        if (in_p || in_q) return in_p && in_q;
    }
    return false;                                            // memory that is not the arena's (a caller's own device array): never merged
}

ThreadArena& thread_arena() { static thread_local ThreadArena a; return a; }

// stream restricted to CUs [first, first + count) of the CU-mask bit order (bit k -> a CU of XCD k mod 8 on gfx950: a prefix spreads evenly)
hipError_t create_cu_range_stream(hipStream_t* s, int first, int count)
{
    int n_cu = 0, dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    if (count <= 0 || first < 0 || first + count > n_cu) return hipErrorInvalidValue;
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
    for (int k = first; k < first + count; ++k) mask[(size_t)k >> 5] |= 1u << (k & 31);
    return hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
}

// ---- hardware queues. HIP maps the streams of a process onto a handful of hardware queues (four per priority by default,
// GPU_MAX_HW_QUEUES; a new stream takes the queue with the fewest users), and two streams on ONE queue run strictly one after the other:
// whether the matcher runs underneath the next extraction, the solver next to the front end or an upload next to the kernels then
// depends on how many streams the process happened to have alive (bench.py's C5 leg measured three states from the same code: front end
// alone 106 k or 136 k kfeatures/s, the solver next to it at 0.85 or at 0.2 of its rate). There is no API to ask for a stream's queue,
// so it is measured: a one-wave kernel that waits for a host flag is parked on stream a, a one-thread kernel that sets a second flag is
// launched on stream b -- if b's flag does not arrive within a millisecond while a is parked, b sits behind a.
__global__ void k_queue_probe_wait(volatile int* go, int* parked)
{
    __hip_atomic_store(parked, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();                                  // 100 MHz: give up after 20 ms whatever happens
    while (__hip_atomic_load(const_cast<int*>(go), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < 2000000) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_queue_probe_set(int* arrived) { __hip_atomic_store(arrived, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int streams_share_queue(hipStream_t a, hipStream_t b, bool* shared, bool* conclusive)
{
    *shared = false;
    if (conclusive) *conclusive = true;
    if (a == b) { *shared = true; return DCS_OK; }
    int* w = nullptr;                                                      // [0] go, [1] parked, [2] arrived
    DCS_HIP(hipHostMalloc((void**)&w, 64, hipHostMallocDefault));
    w[0] = w[1] = w[2] = 0;
    auto rd = [&](int i) { return __atomic_load_n(&w[i], __ATOMIC_ACQUIRE); };
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(now() - t).count(); };
    // b is drained first (its own backlog must not read as "waits for a"); a is NOT: it may be a stream another thread keeps feeding
    // (the Tracking thread's, while a LocalMapping thread builds its solver context) -- the parked kernel simply takes its turn there
    hipError_t e = hipStreamSynchronize(b);
    hipEvent_t parked_done = nullptr;                                      // behind the parked kernel: the probe waits for ITS end, not for what another thread enqueues on a afterwards
    if (e == hipSuccess) e = hipEventCreateWithFlags(&parked_done, hipEventDisableTiming);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_queue_probe_wait, dim3(1), dim3(64), 0, a, (volatile int*)w, w + 1); e = hipGetLastError(); }
    bool recorded = false;
    if (e == hipSuccess) { e = hipEventRecord(parked_done, a); recorded = e == hipSuccess; }
    if (e == hipSuccess) {
        auto t0 = now();
        while (!rd(1) && us_since(t0) < 50000.0) { }                      // the parked kernel is running
        hipLaunchKernelGGL(k_queue_probe_set, dim3(1), dim3(1), 0, b, w + 2);
        e = hipGetLastError();
        t0 = now();
        while (!rd(2) && us_since(t0) < 1000.0) { }
        const bool parked = rd(1) != 0, passed = rd(2) != 0;                 // each flag read ONCE: it may flip between two reads
        *shared = parked && !passed;
        // the parked kernel never started (stream a has a backlog of more than 50 ms): nothing was measured -- b's flag arriving says
        // nothing about the queues. Reported as inconclusive, never as "apart".
        if (!parked && conclusive) *conclusive = false;
    }
    __atomic_store_n(&w[0], 1, __ATOMIC_RELEASE);
    if (!recorded || hipEventSynchronize(parked_done) != hipSuccess) (void)hipStreamSynchronize(a);     // the parked kernel reads w: it must have ended
    if (parked_done) (void)hipEventDestroy(parked_done);
    (void)hipStreamSynchronize(b);
    (void)hipHostFree(w);
    if (e != hipSuccess) { set_error("hardware-queue probe: %s", hipGetErrorString(e)); (void)hipGetLastError(); return DCS_ERR_HIP; }
    return DCS_OK;
}

// A non-blocking stream that shares its hardware queue with none of avoid[0 .. n_avoid): candidates are created (each new one lands on the
// least used queue) and probed until one is apart; the rejected ones are destroyed afterwards. With more streams to avoid than the
// process has queues there is no such stream: the last candidate is returned and *apart (optional) reports 0.
int create_stream_apart(hipStream_t* out, const hipStream_t* avoid, int n_avoid, bool* apart)
{
    const bool probing = true;
    std::vector<hipStream_t> rejected;
    hipStream_t s = nullptr;
    bool ok = false;
    int rc = DCS_OK;
    // every candidate lands on the queue with the fewest users and stays alive until the search ends, so the search walks towards a free
    // queue even when idle streams of the process have left the counts lopsided (a rejected candidate costs the probe's 1 ms time-out)
    const int max_attempts = n_avoid >= 4 ? 10 : 24;               // the default stream's queue + three more: four streams to avoid rarely leave one
    for (int attempt = 0; attempt < max_attempts && !ok; ++attempt) {
        hipStream_t c = nullptr;
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); set_error("hipStreamCreateWithFlags failed"); rc = DCS_ERR_HIP; break; }
        if (s) rejected.push_back(s);
        s = c;
        ok = true;
        bool undecided = false;
        for (int i = 0; i < n_avoid && ok && probing; ++i) {
            bool sh = false, sure = true;
            if ((rc = streams_share_queue(avoid[i], s, &sh, &sure))) break;
            if (!sure) { undecided = true; ok = false; }          // a backlogged stream cannot be probed: the candidate is NOT reported apart,
            else if (sh) ok = false;                              // and more candidates would only wait out the same backlog
        }
        if (rc || undecided) break;
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    if (rc) { if (s) (void)hipStreamDestroy(s); return rc; }
    *out = s;
    if (apart) *apart = ok;
    return DCS_OK;
}

}  // namespace dcs

extern "C" {

const char* dcs_last_error(void) { return dcs::g_err; }
const char* dcs_version(void) { return "dcs-hip 0.1 (gfx950)"; }
int dcs_stream_create_cu_range(int first_cu, int n_cus, void** stream)
{
    if (!stream) { dcs::set_error("null stream pointer"); return DCS_ERR_INVALID; }
    int rc = dcs::ensure_device();
    if (rc) return rc;
    hipStream_t s = nullptr;
    const hipError_t e = dcs::create_cu_range_stream(&s, first_cu, n_cus);
    if (e != hipSuccess) { dcs::set_error("dcs_stream_create_cu_range(%d, %d): %s", first_cu, n_cus, hipGetErrorString(e)); (void)hipGetLastError(); return e == hipErrorInvalidValue ? DCS_ERR_INVALID : DCS_ERR_HIP; }
    *stream = s;
    return DCS_OK;
}
int dcs_streams_share_queue(void* a, void* b, int* shared)
{
    if (!shared) { dcs::set_error("null output"); return DCS_ERR_INVALID; }
    int rc = dcs::ensure_device();
    if (rc) return rc;
    bool sh = false, sure = true;
    rc = dcs::streams_share_queue((hipStream_t)a, (hipStream_t)b, &sh, &sure);
    *shared = !sure ? -1 : sh ? 1 : 0;
    return rc;
}
int dcs_stream_create_apart(void* const* avoid, int n_avoid, void** stream, int* apart)
{
    if (!stream || n_avoid < 0 || (n_avoid && !avoid)) { dcs::set_error("bad argument"); return DCS_ERR_INVALID; }
    int rc = dcs::ensure_device();
    if (rc) return rc;
    hipStream_t s = nullptr;
    bool ok = false;
    rc = dcs::create_stream_apart(&s, reinterpret_cast<const hipStream_t*>(avoid), n_avoid, &ok);
    if (rc) return rc;
    *stream = s;
    if (apart) *apart = ok ? 1 : 0;
    return DCS_OK;
}
void dcs_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }

int dcs_host_alloc(void** ptr, size_t bytes)
{
    if (!ptr || !bytes) { dcs::set_error("dcs_host_alloc: bad argument"); return DCS_ERR_INVALID; }
    *ptr = nullptr;
    if (hipHostMalloc(ptr, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); dcs::set_error("dcs_host_alloc: %zu bytes of page-locked memory refused", bytes); return DCS_ERR_HIP; }
    return DCS_OK;
}
int dcs_host_free(void* ptr)
{
    if (ptr && hipHostFree(ptr) != hipSuccess) { (void)hipGetLastError(); dcs::set_error("dcs_host_free: not a dcs_host_alloc block"); return DCS_ERR_INVALID; }
    return DCS_OK;
}

int dcs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // extern "C"
