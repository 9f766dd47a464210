// hardware probe (not product code): do D16 LDS loads preserve the other half on this GPU, and do unaligned ds_read_b32 work?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(unsigned* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (uint8_t)(i + 1);
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(s + threadIdx.x);
    unsigned r0 = 0xAAAAAAAAu, r1, r2;
    asm volatile("ds_read_u8_d16 %0, %3 offset:0\n\tds_read_u8_d16_hi %0, %3 offset:7\n\t"
                 "ds_read_b32 %1, %3 offset:0\n\tds_read_u16 %2, %3 offset:1\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(r0), "=&v"(r1), "=&v"(r2) : "v"(a));
    out[threadIdx.x * 3 + 0] = r0; out[threadIdx.x * 3 + 1] = r1; out[threadIdx.x * 3 + 2] = r2;
}
int main()
{
    unsigned* d; unsigned h[192];
    hipMalloc(&d, sizeof(h)); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 6; ++l) printf("lane %d: d16 lo+hi = %08x (preserving: %08x)  b32@%d = %08x  u16@%d = %08x\n", l, h[3 * l], ((l + 8) << 16) | (l + 1), l, h[3 * l + 1], l + 1, h[3 * l + 2]);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("arch %s\n", p.gcnArchName);
    return 0;
}
