// Does gfx950 read LDS at 2-byte aligned addresses with ds_read_b32 / ds_read_b64 / ds_read2_b32, and at what rate?
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned_probe lds_unaligned_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k_check(uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (uint8_t)i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s + 2u + 8u * threadIdx.x % 128u;   // 2 (mod 4)
    unsigned v32; unsigned long long v64; unsigned p0, p1;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(a) : "memory");
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:22\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
    p0 = (unsigned)v64; p1 = (unsigned)(v64 >> 32);
    unsigned long long w64;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w64) : "v"(a) : "memory");
    out[4 * threadIdx.x] = v32; out[4 * threadIdx.x + 1] = (unsigned)w64; out[4 * threadIdx.x + 2] = (unsigned)(w64 >> 32); out[4 * threadIdx.x + 3] = p1;
}
template <int MODE> __global__ void k_rate(uint32_t* out, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = (uint8_t)i;
    __syncthreads();
    unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s + (MODE & 1 ? 2u : 0u) + 88u * (threadIdx.x & 63);
    unsigned long long acc = 0, v0, v1;
    for (int i = 0; i < iters; ++i) {
        if (MODE < 2) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a) : "memory");
        else asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a) : "memory");
        acc += v0 ^ v1; a ^= 16u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)acc;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 1 << 22);
    k_check<<<1, 64>>>(d); hipDeviceSynchronize();
    uint32_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 16; ++t) {
        const unsigned a = 2 + 8 * t % 128;
        uint32_t e32 = 0, e1 = 0, p1 = 0; uint8_t bytes[256]; for (int i = 0; i < 256; ++i) bytes[i] = (uint8_t)i;
        memcpy(&e32, bytes + a, 4); memcpy(&e1, bytes + a + 4, 4); memcpy(&p1, bytes + a + 88, 4);
        if (h[4 * t] != e32 || h[4 * t + 1] != e32 || h[4 * t + 2] != e1 || h[4 * t + 3] != p1) { ++bad; printf("lane %d addr %u: b32 %08x (want %08x) b64 %08x %08x (want %08x %08x) read2[22] %08x (want %08x)\n", t, a, h[4 * t], e32, h[4 * t + 1], h[4 * t + 2], e32, e1, h[4 * t + 3], p1); }
    }
    printf("unaligned LDS reads: %s\n", bad ? "WRONG" : "correct");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    auto run = [&](auto kern, const char* name) {
        kern<<<2048, 256>>>(d, iters); hipDeviceSynchronize();
        hipEventRecord(e0); kern<<<2048, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %.3f ms\n", name, ms);
    };
    run(k_rate<0>, "2 x ds_read_b64, 8-byte aligned");
    run(k_rate<1>, "2 x ds_read_b64, 2 (mod 4) address");
    run(k_rate<2>, "2 x ds_read2_b32, 4-byte aligned");
    run(k_rate<3>, "2 x ds_read2_b32, 2 (mod 4) address");
    return 0;
}
