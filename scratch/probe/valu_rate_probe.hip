// hardware probe: issue cost of the integer instructions the front-end kernels lean on, relative to v_mad_u32_u24 (one wave per SIMD, 4 independent
// chains, 4096 instructions each)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
template <int OP> __global__ void k(unsigned* out, long long* cyc, int iters)
{
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x * 5 + 2, d = threadIdx.x * 7 + 3, w = 0x12223137u + threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a = __umul24(a, w) + b; b = __umul24(b, w) + c; c = __umul24(c, w) + d; d = __umul24(d, w) + a; }
            if (OP == 1) { a = __builtin_amdgcn_udot4(a, w, b, false); b = __builtin_amdgcn_udot4(b, w, c, false); c = __builtin_amdgcn_udot4(c, w, d, false); d = __builtin_amdgcn_udot4(d, w, a, false); }
            if (OP == 2) { a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, w), b, false); b = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, b), __builtin_bit_cast(ushort2_t, w), c, false);
                           c = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, c), __builtin_bit_cast(ushort2_t, w), d, false); d = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, d), __builtin_bit_cast(ushort2_t, w), a, false); }
            if (OP == 3) { a = __builtin_amdgcn_alignbyte(a, b, 1); b = __builtin_amdgcn_alignbyte(b, c, 2); c = __builtin_amdgcn_alignbyte(c, d, 3); d = __builtin_amdgcn_alignbyte(d, a, 1); }
            if (OP == 4) { a = __builtin_amdgcn_perm(a, b, w); b = __builtin_amdgcn_perm(b, c, w); c = __builtin_amdgcn_perm(c, d, w); d = __builtin_amdgcn_perm(d, a, w); }
            if (OP == 5) { a = (a << 16) | b; b = (b << 16) | c; c = (c << 16) | d; d = (d << 16) | a; }
            if (OP == 6) { a = min(a, b) ; b = max(b, c); c = min(c, d); d = max(d, a); }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    unsigned* out; long long* cyc; (void)hipMalloc(&out, 4 * 4096); (void)hipMalloc(&cyc, 8);
    const char* names[] = {"v_mad_u32_u24", "v_dot4_u32_u8", "v_dot2_u32_u16", "v_alignbyte_b32", "v_perm_b32", "v_lshl_or_b32", "v_min/max_u32"};
    const int iters = 128;
    auto run = [&](auto kern, int op, int threads) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); }
        long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-16s %d waves/SIMD: %.2f cycles per instruction per wave\n", names[op], threads / 256, (double)h / (iters * 32.0));
    };
    run(k<0>, 0, 256); run(k<1>, 1, 256); run(k<2>, 2, 256); run(k<3>, 3, 256); run(k<4>, 4, 256); run(k<5>, 5, 256); run(k<6>, 6, 256);
    run(k<0>, 0, 512); run(k<1>, 1, 512); run(k<2>, 2, 512);
    return 0;
}
