// hardware probe: what does a kernel boundary between two SHORT dependent kernels cost on gfx950, and what does it cost when the
// consumer is launched early on a second stream and waits on a device-side counter (a "gate") instead?
//
// A "step" = kChain short kernels, each depending on the previous one (80 workgroups x 256 threads: a few dependent L2 loads, a block
// reduction and an arrival on a counter -- the shape of the BA solver's narrow kernels).
//   mode 0  one stream, plain launches: the stream order is the dependency
//   mode 1  kernels alternate between two streams; every kernel spins on its predecessor's arrival counter (thread 0 of each workgroup,
//           s_sleep between polls, bounded), so its launch latency and ramp-up hide behind the predecessor
//   mode 2  one stream, gates on (cost of the gate itself with nothing to hide)
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/probe/gate_probe scratch/probe/gate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kBlocks = 80, kThreads = 256, kChain = 8;

__global__ __launch_bounds__(kThreads) void k_stage(const unsigned* __restrict__ chase, int hops, double* __restrict__ partial, unsigned* gate_in, unsigned target_in,
                                                    unsigned* gate_out, unsigned* abort_word, double* sink)
{
    __shared__ double s[kThreads];
    if (gate_in) {
        if (threadIdx.x == 0) {
            unsigned polls = 0;
            while (__hip_atomic_load(gate_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target_in) {
                __builtin_amdgcn_s_sleep(8);
                if (++polls > (1u << 22)) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    for (int h = 0; h < hops; ++h) i = chase[i & 0x3FFFF];
    s[threadIdx.x] = (double)(i & 0xFF) + partial[blockIdx.x];
    __syncthreads();
    for (int d = kThreads / 2; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s[0] * 1e-9;
        if (s[0] == 12345.678) sink[0] = s[0];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(gate_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 200, hops = argc > 2 ? atoi(argv[2]) : 4;
    unsigned* d_chase; double *d_partial, *d_sink; unsigned* d_gates;
    std::vector<unsigned> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)((i * 2654435761u + 12345u) & 0x3FFFF);
    CK(hipMalloc(&d_chase, h.size() * 4)); CK(hipMemcpy(d_chase, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_partial, kBlocks * 8)); CK(hipMemset(d_partial, 0, kBlocks * 8));
    CK(hipMalloc(&d_sink, 8));
    CK(hipMalloc(&d_gates, 64 * (kChain + 2)));
    hipStream_t st[2]; CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(d_gates, 0, 64 * (kChain + 2)));
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int s = 0; s < steps; ++s)
                for (int k = 0; k < kChain; ++k) {
                    // gate k counts the arrivals of stage k (kBlocks per step); stage k waits for stage k - 1 of this step, stage 0 for the last stage of the previous step
                    unsigned* gin = mode == 0 ? nullptr : d_gates + 16 * ((k + kChain - 1) % kChain);
                    const unsigned target = (unsigned)kBlocks * (k == 0 ? s : s + 1);
                    hipStream_t q = mode == 1 ? st[(s * kChain + k) & 1] : st[0];
                    hipLaunchKernelGGL(k_stage, dim3(kBlocks), dim3(kThreads), 0, q, d_chase, hops, d_partial, gin, target, d_gates + 16 * k, d_gates + 16 * kChain, d_sink);
                }
            const double us_enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            unsigned ab = 0; CK(hipMemcpy(&ab, d_gates + 16 * kChain, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s): %8.2f us per step of %d kernels = %6.2f us per kernel (host enqueue %6.2f us per kernel)%s\n", mode,
                   mode == 0 ? "one stream, plain" : mode == 1 ? "two streams, gated" : "one stream, gated", us / steps, kChain, us / steps / kChain, us_enq / steps / kChain, ab ? "  [GATE TIMEOUT]" : "");
        }
    }
    return 0;
}
