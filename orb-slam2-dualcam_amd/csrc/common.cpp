// common.cpp -- error text, device probing, version.
#include "common.h"

#include <algorithm>
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
