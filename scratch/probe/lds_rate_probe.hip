// hardware probe: issue cost of LDS instructions on gfx950, per CU. All CUs busy, W waves per SIMD, every wave issues batches of 8 independent
// LDS instructions followed by one s_waitcnt. Reported: CU-cycles per wave-instruction = wall time x 2.39 GHz / (instructions per CU),
// next to the same figure for a VALU reference (v_mad_u32_u24: one SIMD is busy ~4.2 cycles per wave-instruction, i.e. ~1.05 CU-cycles
// with all four SIMDs busy). An LDS instruction that costs more CU-cycles than 1/4 of a SIMD's VALU budget per instruction makes the
// kernel LDS-issue bound before it is VALU bound.
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/probe/lds_rate_probe scratch/probe/lds_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr double kClock = 2.39e9;

#define DEF_LDS(NAME, STRIDE, BODY)                                                                                      \
    __global__ void k_##NAME(unsigned* out, int trips)                                                                   \
    {                                                                                                                    \
        extern __shared__ unsigned lds[];                                                                                \
        for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;                                                 \
        __syncthreads();                                                                                                 \
        unsigned a = (threadIdx.x & 63) * (STRIDE) + (threadIdx.x >> 6) * 256, acc = 0;                                  \
        unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;                                       \
        for (int i = 0; i < trips; ++i) { BODY; acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; }                          \
        if (acc == 0x12345678u) out[threadIdx.x] = acc;                                                                  \
    }
#define RD8(OP, OFFS) asm volatile(OP " %0, %8 offset:" #OFFS "*0\n" OP " %1, %8 offset:" #OFFS "*1\n" OP " %2, %8 offset:" #OFFS "*2\n" OP " %3, %8 offset:" #OFFS "*3\n" \
                                   OP " %4, %8 offset:" #OFFS "*4\n" OP " %5, %8 offset:" #OFFS "*5\n" OP " %6, %8 offset:" #OFFS "*6\n" OP " %7, %8 offset:" #OFFS "*7\n s_waitcnt lgkmcnt(0)" \
                                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a))
DEF_LDS(read_u8_consecutive, 1, RD8("ds_read_u8", 48))
DEF_LDS(read_u8_stride4, 4, RD8("ds_read_u8", 48))
DEF_LDS(read_u16_consecutive, 2, RD8("ds_read_u16", 48))
DEF_LDS(read_b32_consecutive, 4, RD8("ds_read_b32", 256))
DEF_LDS(read_b32_unaligned, 4, { unsigned a1 = a + 1; asm volatile("" : "+v"(a1)); unsigned a_ = a; a = a1; RD8("ds_read_b32", 256); a = a_; })
DEF_LDS(read_b32_same_addr, 0, RD8("ds_read_b32", 256))
// 64-bit and 128-bit reads: 4 per batch (register pressure of the asm), counted accordingly
#define RD4W(OP, OFFS, CONS) { unsigned long long q0, q1, q2, q3; asm volatile(OP " %0, %4 offset:" #OFFS "*0\n" OP " %1, %4 offset:" #OFFS "*1\n" OP " %2, %4 offset:" #OFFS "*2\n" OP " %3, %4 offset:" #OFFS "*3\n s_waitcnt lgkmcnt(0)" \
                                   : CONS(q0), CONS(q1), CONS(q2), CONS(q3) : "v"(a)); r0 ^= (unsigned)(q0 ^ q1 ^ q2 ^ q3); }
DEF_LDS(read_b64_consecutive, 8, RD4W("ds_read_b64", 512, "=v"))
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
#define RD4Q(OP, OFFS) { u4_t q0, q1, q2, q3; asm volatile(OP " %0, %4 offset:" #OFFS "*0\n" OP " %1, %4 offset:" #OFFS "*1\n" OP " %2, %4 offset:" #OFFS "*2\n" OP " %3, %4 offset:" #OFFS "*3\n s_waitcnt lgkmcnt(0)" \
                                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a)); r0 ^= q0.x ^ q1.y ^ q2.z ^ q3.w; }
DEF_LDS(read_b128_consecutive, 16, RD4Q("ds_read_b128", 1024))
#define WR8(OP, OFFS) asm volatile(OP " %0, %1 offset:" #OFFS "*0\n" OP " %0, %1 offset:" #OFFS "*1\n" OP " %0, %1 offset:" #OFFS "*2\n" OP " %0, %1 offset:" #OFFS "*3\n" \
                                   OP " %0, %1 offset:" #OFFS "*4\n" OP " %0, %1 offset:" #OFFS "*5\n" OP " %0, %1 offset:" #OFFS "*6\n" OP " %0, %1 offset:" #OFFS "*7\n s_waitcnt lgkmcnt(0)" \
                                   : : "v"(a), "v"(acc) : "memory")
DEF_LDS(write_b8_consecutive, 1, WR8("ds_write_b8", 64))
DEF_LDS(write_b16_consecutive, 2, WR8("ds_write_b16", 128))
DEF_LDS(write_b32_consecutive, 4, WR8("ds_write_b32", 256))
DEF_LDS(bpermute, 4, { asm volatile("ds_bpermute_b32 %0, %8, %9\n ds_bpermute_b32 %1, %8, %9\n ds_bpermute_b32 %2, %8, %9\n ds_bpermute_b32 %3, %8, %9\n ds_bpermute_b32 %4, %8, %9\n ds_bpermute_b32 %5, %8, %9\n ds_bpermute_b32 %6, %8, %9\n ds_bpermute_b32 %7, %8, %9\n s_waitcnt lgkmcnt(0)" \
                                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a), "v"(acc)); })
__global__ void k_valu_ref(unsigned* out, int trips)
{
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3, w = 0x123457u + a;
    for (int i = 0; i < trips; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w)); }
    }
    if ((a ^ b ^ c ^ d) == 0x12345678u) out[threadIdx.x] = a;
}
static hipEvent_t g_e0, g_e1;
static double time_launch(const void* fn, dim3 grid, dim3 block, void** args, size_t lds)
{
    float best = 1e30f;
    CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(g_e0, 0)); CK(hipLaunchKernel(fn, grid, block, args, lds, 0)); CK(hipEventRecord(g_e1, 0)); CK(hipEventSynchronize(g_e1));
        float ms; CK(hipEventElapsedTime(&ms, g_e0, g_e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best * 1e-3;
}
int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    unsigned* out; CK(hipMalloc(&out, 1 << 20));
    CK(hipEventCreate(&g_e0)); CK(hipEventCreate(&g_e1));
    int trips = 4096;
    void* args[] = {&out, &trips};
    printf("CU-cycles per wave64 LDS instruction (wall time x 2.39 GHz / instructions issued on the CU), all %d CUs busy; W = waves per SIMD\n%-26s %8s %8s %8s\n", cus, "", "W=1", "W=2", "W=4");
    struct Row { const char* name; const void* fn; int per_trip; };
    const Row rows[] = {{"v_mad_u32_u24 (VALU ref)", (const void*)k_valu_ref, 8}, {"ds_read_u8 consecutive", (const void*)k_read_u8_consecutive, 8}, {"ds_read_u8 stride 4 B", (const void*)k_read_u8_stride4, 8},
                        {"ds_read_u16 consecutive", (const void*)k_read_u16_consecutive, 8}, {"ds_read_b32 consecutive", (const void*)k_read_b32_consecutive, 8},
                        {"ds_read_b32 unaligned +1", (const void*)k_read_b32_unaligned, 8}, {"ds_read_b32 one address", (const void*)k_read_b32_same_addr, 8},
                        {"ds_read_b64 consecutive", (const void*)k_read_b64_consecutive, 4}, {"ds_read_b128 consecutive", (const void*)k_read_b128_consecutive, 4},
                        {"ds_write_b8 consecutive", (const void*)k_write_b8_consecutive, 8}, {"ds_write_b16 consecutive", (const void*)k_write_b16_consecutive, 8},
                        {"ds_write_b32 consecutive", (const void*)k_write_b32_consecutive, 8}, {"ds_bpermute_b32", (const void*)k_bpermute, 8}};
    for (const Row& r : rows) {
        printf("%-26s", r.name);
        for (int W = 1; W <= 4; W *= 2) {
            const double t = time_launch(r.fn, dim3(cus), dim3(256 * W), args, 96 * 1024);
            printf(" %8.2f", t * kClock / ((double)trips * r.per_trip * 4 * W));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
