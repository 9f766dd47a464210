// hardware probe: how long does one wave64 VALU instruction occupy a SIMD of gfx950?
//
// Method:
//  * every workgroup asks for 96 KB of LDS, so exactly ONE workgroup lands on a CU; a workgroup of 256 * W threads puts W waves on
//    each of the CU's 4 SIMDs; grid = number of CUs. Every SIMD of the chip then runs W waves of the same instruction stream.
//  * wall time of the launch comes from hipEvents; the shader clock under this very load from clock64() (s_memtime) read by a wave at
//    the start and the end of the same launch, divided by the same wall time -- printed, so that a throttled clock would show.
//    cycles per instruction = T * f / (W * N) with N instructions per wave: the time a SIMD is occupied by one wave-instruction.
//  * each op runs as ONE dependent chain and as FOUR independent chains (what a compiler-scheduled kernel looks like), at 1, 2, 4 and 8
//    waves per SIMD. At 1 wave / 1 chain the figure is the op's dependent-issue latency, at 4-8 waves it is the issue cost kernels pay.
//  * `s_nop 15` is in the list as a check of the unit (it is documented as 16 "wait states").
//  * v_fma_f32 / v_pk_fma_f32 are in the list to reconcile with the chip's quoted 157.3 TFLOP/s fp32 vector peak.
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/probe/valu_rate_probe scratch/probe/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 32;       // instructions per chain per loop trip

// d = chain variable (32-bit), dd = 64-bit chain variable, w / x = loop-invariant operands
#define DEF_OP(NAME, ...)                                                                                                              \
    template <int CHAINS> __global__ void k_##NAME(unsigned* out, int trips, long long* cyc)                                          \
    {                                                                                                                                  \
        extern __shared__ unsigned lds[];                                                                                              \
        unsigned a[4], w = 0x00010203u + threadIdx.x, x = 0x04050607u ^ threadIdx.x;                                                   \
        unsigned long long aa[4], ww = 0x3f8000003f800000ull;                                                                          \
        for (int c = 0; c < 4; ++c) { a[c] = threadIdx.x * (2 * c + 3) + 1; aa[c] = 0x3f8000003f800000ull + c; }                       \
        const long long t0 = clock64();                                                                                                \
        for (int i = 0; i < trips; ++i) {                                                                                              \
            _Pragma("unroll") for (int u = 0; u < kUnroll; ++u)                                                                        \
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { unsigned& d = a[c]; unsigned long long& dd = aa[c]; (void)d; (void)dd; __VA_ARGS__; } \
        }                                                                                                                              \
        const long long t1 = clock64();                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                                     \
        unsigned r = 0;                                                                                                                \
        for (int c = 0; c < 4; ++c) r ^= a[c] ^ (unsigned)aa[c] ^ (unsigned)(aa[c] >> 32);                                             \
        if (r == 0x12345678u) { lds[threadIdx.x] = r; out[blockIdx.x * blockDim.x + threadIdx.x] = lds[(threadIdx.x + 1) % blockDim.x]; } \
    }

#define A3(txt) asm volatile(txt : "+v"(d) : "v"(w), "v"(x))
#define A2(txt) asm volatile(txt : "+v"(d) : "v"(w))
DEF_OP(s_nop_15, asm volatile("s_nop 15"))
DEF_OP(v_mov_b32, A2("v_mov_b32 %0, %1"))
DEF_OP(v_add_u32, A2("v_add_u32 %0, %0, %1"))
DEF_OP(v_sub_u32, A2("v_sub_u32 %0, %0, %1"))
DEF_OP(v_and_b32, A2("v_and_b32 %0, %0, %1"))
DEF_OP(v_or_b32, A2("v_or_b32 %0, %0, %1"))
DEF_OP(v_xor_b32, A2("v_xor_b32 %0, %0, %1"))
DEF_OP(v_lshlrev_b32, A2("v_lshlrev_b32 %0, 1, %0"))
DEF_OP(v_lshrrev_b32, A2("v_lshrrev_b32 %0, 1, %0"))
DEF_OP(v_max_u32, A2("v_max_u32 %0, %0, %1"))
DEF_OP(v_min_u32, A2("v_min_u32 %0, %0, %1"))
DEF_OP(v_max_i32, A2("v_max_i32 %0, %0, %1"))
DEF_OP(v_max_u16, A2("v_max_u16 %0, %0, %1"))
DEF_OP(v_add_u16, A2("v_add_u16 %0, %0, %1"))
DEF_OP(v_mul_u32_u24, A2("v_mul_u32_u24 %0, %0, %1"))
DEF_OP(v_mad_u32_u24, A3("v_mad_u32_u24 %0, %0, %1, %2"))
DEF_OP(v_mul_lo_u32, A2("v_mul_lo_u32 %0, %0, %1"))
DEF_OP(v_add3_u32, A3("v_add3_u32 %0, %0, %1, %2"))
DEF_OP(v_lshl_add_u32, A2("v_lshl_add_u32 %0, %0, 1, %1"))
DEF_OP(v_lshl_or_b32, A2("v_lshl_or_b32 %0, %0, 16, %1"))
DEF_OP(v_and_or_b32, A3("v_and_or_b32 %0, %0, %1, %2"))
DEF_OP(v_or3_b32, A3("v_or3_b32 %0, %0, %1, %2"))
DEF_OP(v_bfe_u32, A2("v_bfe_u32 %0, %0, 1, 31"))
DEF_OP(v_bfi_b32, A3("v_bfi_b32 %0, %1, %0, %2"))
DEF_OP(v_perm_b32, A3("v_perm_b32 %0, %0, %1, %2"))
DEF_OP(v_alignbyte_b32, A2("v_alignbyte_b32 %0, %0, %1, 1"))
DEF_OP(v_alignbit_b32, A2("v_alignbit_b32 %0, %0, %1, 3"))
DEF_OP(v_dot4_u32_u8, A3("v_dot4_u32_u8 %0, %0, %1, %2"))
DEF_OP(v_dot2_u32_u16, A3("v_dot2_u32_u16 %0, %0, %1, %2"))
DEF_OP(v_sad_u8, A3("v_sad_u8 %0, %0, %1, %2"))
DEF_OP(v_pk_min_i16, A2("v_pk_min_i16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]"))
DEF_OP(v_pk_max_u16, A2("v_pk_max_u16 %0, %0, %1"))
DEF_OP(v_pk_add_u16, A2("v_pk_add_u16 %0, %0, %1"))
DEF_OP(v_pk_sub_i16, A2("v_pk_sub_i16 %0, %0, %1"))
DEF_OP(v_pk_mul_lo_u16, A2("v_pk_mul_lo_u16 %0, %0, %1"))
DEF_OP(v_pk_mad_i16, A3("v_pk_mad_i16 %0, %0, %1, %2"))
DEF_OP(v_pk_lshlrev_b16, A2("v_pk_lshlrev_b16 %0, 1, %0"))
DEF_OP(v_med3_u32, A3("v_med3_u32 %0, %0, %1, %2"))
DEF_OP(v_min3_u32, A3("v_min3_u32 %0, %0, %1, %2"))
DEF_OP(v_max3_u32, A3("v_max3_u32 %0, %0, %1, %2"))
DEF_OP(v_bcnt_u32_b32, A2("v_bcnt_u32_b32 %0, %1, %0"))
DEF_OP(v_mbcnt_lo, A2("v_mbcnt_lo_u32_b32 %0, %1, %0"))
DEF_OP(v_mbcnt_hi, A2("v_mbcnt_hi_u32_b32 %0, %1, %0"))
DEF_OP(v_cndmask_vcc, A2("v_cndmask_b32 %0, %0, %1, vcc"))
DEF_OP(v_cndmask_e64_vcc, A2("v_cndmask_b32_e64 %0, %0, %1, vcc"))
DEF_OP(v_cndmask_vcc_set, { asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(x), "v"(w) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(d) : "v"(w)); })
DEF_OP(v_addc_co_u32, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(d) : "v"(w) : "vcc"))
DEF_OP(v_lshlrev_b32_v, A2("v_lshlrev_b32 %0, %1, %0"))
DEF_OP(v_lshlrev_b32_e64, A2("v_lshlrev_b32_e64 %0, 1, %0"))
DEF_OP(v_add_u32_e64, A2("v_add_u32_e64 %0, %0, %1"))
DEF_OP(v_sub_u16, A2("v_sub_u16 %0, %0, %1"))
DEF_OP(v_min_u16, A2("v_min_u16 %0, %0, %1"))
DEF_OP(v_mul_lo_u16, A2("v_mul_lo_u16 %0, %0, %1"))
DEF_OP(v_lshrrev_b16, A2("v_lshrrev_b16 %0, 1, %0"))
DEF_OP(v_ashrrev_i32, A2("v_ashrrev_i32 %0, 1, %0"))
DEF_OP(v_max_f32, A2("v_max_f32 %0, %0, %1"))
DEF_OP(v_min_f32, A2("v_min_f32 %0, %0, %1"))
DEF_OP(v_sub_f32, A2("v_sub_f32 %0, %0, %1"))
DEF_OP(v_subrev_u32, A2("v_subrev_u32 %0, %0, %1"))
DEF_OP(v_xnor_b32, A2("v_xnor_b32 %0, %0, %1"))
DEF_OP(v_cndmask_sgpr, asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d) : "v"(w), "s"(0x5555aaaa5555aaaaull)))
DEF_OP(v_cmp_gt_u32_vcc, asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(d), "v"(w) : "vcc"))
DEF_OP(v_cmp_sgpr, { unsigned long long m; asm volatile("v_cmp_gt_u32 %0, %1, %2" : "=s"(m) : "v"(d), "v"(w)); })
DEF_OP(v_cmp_then_cndmask, { unsigned long long m; asm volatile("v_cmp_gt_u32 %0, %1, %2" : "=s"(m) : "v"(d), "v"(w)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "s"(m)); })
DEF_OP(v_add_co_u32, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(d) : "v"(w) : "vcc"))
DEF_OP(v_readlane, { unsigned s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(d)); })
DEF_OP(v_readfirstlane, { unsigned s_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s_) : "v"(d)); })
DEF_OP(v_mov_dpp_row_shr, A2("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_OP(v_add_dpp_row_shr, A2("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_OP(v_add_dpp_wave_shr, A2("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_OP(v_max_u32_sdwa_b, A2("v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2"))
DEF_OP(v_add_u32_sdwa_b, A2("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2"))
DEF_OP(v_cvt_f32_u32, A2("v_cvt_f32_u32 %0, %0"))
DEF_OP(v_cvt_i32_f32, A2("v_cvt_i32_f32 %0, %0"))
DEF_OP(v_cvt_f32_ubyte0, A2("v_cvt_f32_ubyte0 %0, %0"))
DEF_OP(v_rndne_f32, A2("v_rndne_f32 %0, %0"))
DEF_OP(v_add_f32, A2("v_add_f32 %0, %0, %1"))
DEF_OP(v_mul_f32, A2("v_mul_f32 %0, %0, %1"))
DEF_OP(v_fma_f32, A3("v_fma_f32 %0, %0, %1, %2"))
DEF_OP(v_mac_f32, A3("v_fmac_f32 %0, %1, %2"))
DEF_OP(v_rcp_f32, A2("v_rcp_f32 %0, %0"))
DEF_OP(v_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(dd) : "v"(ww)))
DEF_OP(v_pk_add_f32, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dd) : "v"(ww)))
DEF_OP(v_pk_mul_f32, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(dd) : "v"(ww)))
DEF_OP(v_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd) : "v"(ww)))
DEF_OP(v_add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd) : "v"(ww)))
DEF_OP(v_mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd) : "v"(ww)))
DEF_OP(v_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(dd)))
// round 4 additions: what the rewritten k_fast_cells / k_describe loops are made of
DEF_OP(v_bitop3_b32, A3("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x1e"))
DEF_OP(v_max_i16, A2("v_max_i16 %0, %0, %1"))
DEF_OP(v_max3_u16, A3("v_max3_u16 %0, %0, %1, %2"))
DEF_OP(v_mad_i32_i24, A3("v_mad_i32_i24 %0, %0, %1, %2"))
DEF_OP(v_mul_i32_i24, A2("v_mul_i32_i24 %0, %0, %1"))
DEF_OP(v_pk_min_i16_plain, A2("v_pk_min_i16 %0, %0, %1"))
DEF_OP(v_cmp_lt_i32_sdwa, { unsigned long long m; asm volatile("v_cmp_lt_i32_sdwa %0, %1, sext(%2) src0_sel:DWORD src1_sel:WORD_0" : "=s"(m) : "v"(w), "v"(d)); })
DEF_OP(v_cmp_gt_i16_sgpr, { unsigned long long m; asm volatile("v_cmp_gt_i16_e64 %0, %1, %2" : "=s"(m) : "v"(d), "v"(w)); })
DEF_OP(v_cmp_gt_u16_vcc, asm volatile("v_cmp_gt_u16 vcc, %0, %1" : : "v"(d), "v"(w) : "vcc"))
DEF_OP(v_add_u32_sgpr, asm volatile("v_add_u32 %0, %1, %0" : "+v"(d) : "s"(0x200u)))
DEF_OP(v_xor_b32_lit, asm volatile("v_xor_b32 %0, 0xff00ff, %0" : "+v"(d)))
// scalar unit: alone, and interleaved with vector instructions of the same wave (do they overlap across the waves of a SIMD?)
DEF_OP(s_add_u32, { unsigned s_ = 1; asm volatile("s_add_u32 %0, %0, 3" : "+s"(s_) : : "scc"); })
DEF_OP(s_and_b64, { unsigned long long s_ = 5; asm volatile("s_and_b64 %0, %0, exec" : "+s"(s_) : : "scc"); })
DEF_OP(s_bcnt1_i32_b64, { unsigned s_; asm volatile("s_bcnt1_i32_b64 %0, exec" : "=s"(s_) : : "scc"); })
DEF_OP(s_waitcnt_idle, asm volatile("s_waitcnt lgkmcnt(0)"))
DEF_OP(s_nop_0, asm volatile("s_nop 0"))
DEF_OP(s_saveexec_pair, { unsigned long long s_; asm volatile("s_and_saveexec_b64 %0, exec\n\ts_mov_b64 exec, %0" : "=&s"(s_) : : "scc"); })
DEF_OP(s_cmp_cbranch, asm volatile("s_cmp_eq_u32 s0, s0\n\ts_cbranch_scc0 1f\n\ts_nop 0\n1:" : : : "scc"))
DEF_OP(s_mov_b32, { unsigned s_; asm volatile("s_mov_b32 %0, 7" : "=s"(s_)); })
DEF_OP(s_add_indep4, { unsigned a_ = 1, b_ = 2, c_ = 3, e_ = 4; asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3" : "+s"(a_), "+s"(b_), "+s"(c_), "+s"(e_) : : "scc"); })
DEF_OP(mix_vslow_s, { unsigned s_ = 1; asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 3" : "+v"(d), "+s"(s_) : "v"(w), "v"(x) : "scc"); })
DEF_OP(mix_vfast_s, { unsigned s_ = 1; asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(d), "+s"(s_) : "v"(w) : "scc"); })
DEF_OP(mix_vfast_s_s, { unsigned s_ = 1; asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3\n\ts_lshl_b32 %1, %1, 1" : "+v"(d), "+s"(s_) : "v"(w) : "scc"); })
DEF_OP(mix_vslow_lds, { unsigned r_; asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n\tds_read_u8 %1, %4" : "+v"(d), "=v"(r_) : "v"(w), "v"(x), "v"(x & 0xfffu)); })
DEF_OP(ds_read_b32, { unsigned r_; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r_) : "v"((d & 0xffcu))); d ^= r_ & 4; })
DEF_OP(ds_read_u8, { unsigned r_; asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r_) : "v"((d & 0xfffu))); d ^= r_ & 4; })

static hipEvent_t g_e0, g_e1;
static double time_launch(const void* fn, dim3 grid, dim3 block, void** args, size_t lds)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {                                   // first repetition warms the clocks
        CK(hipEventRecord(g_e0, 0));
        CK(hipLaunchKernel(fn, grid, block, args, lds, 0));
        CK(hipEventRecord(g_e1, 0));
        CK(hipEventSynchronize(g_e1));
        float ms; CK(hipEventElapsedTime(&ms, g_e0, g_e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best * 1e-3;
}

struct Ctx { int cus; unsigned* out; long long* cyc; double f_hz; };
template <typename K1, typename K4> static void report(const Ctx& c, const char* name, K1 k1, K4 k4, int per_trip_scale = 1)
{
    const size_t lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int trips = 256;
    unsigned* out = c.out; long long* cyc = c.cyc;
    void* args[] = {&out, &trips, &cyc};
    printf("%-22s", name);
    for (int chains = 1; chains <= 4; chains += 3) {
        for (int W = 1; W <= 8; W *= 2) {
            if (W == 8 && 256 * W > 1024) { /* 2 workgroups of 4 waves per SIMD cannot be forced onto one CU with 96 KB each: use 48 KB */ }
            const int threads = W == 8 ? 1024 : 256 * W;
            const size_t l = W == 8 ? 64 * 1024 : lds;                    // W = 8: two 1024-thread workgroups per CU (64 KB each), grid = 2 x CUs
            const dim3 grid(W == 8 ? 2 * c.cus : c.cus);
            const void* fn = chains == 1 ? (const void*)k1 : (const void*)k4;
            const double t = time_launch(fn, grid, dim3(threads), args, l);
            const double n = (double)trips * kUnroll * chains * per_trip_scale;
            printf(" %6.2f", t * c.f_hz / (W * n));
        }
        printf(chains == 1 ? "  |" : "");
    }
    printf("\n");
    fflush(stdout);
}
#define REPORT(NAME) report(ctx, #NAME, k_##NAME<1>, k_##NAME<4>)

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    Ctx ctx{};
    ctx.cus = p.multiProcessorCount;
    CK(hipMalloc(&ctx.out, 4 * 4096 * 256)); CK(hipMalloc(&ctx.cyc, 8));
    CK(hipEventCreate(&g_e0)); CK(hipEventCreate(&g_e1));
    printf("device: %s, %d CUs, clockRate attribute %.0f MHz\n", p.gcnArchName, ctx.cus, p.clockRate * 1e-3);
    // shader clock under load: clock64() delta of one wave over a launch / wall time of that launch (v_mad chain, all SIMDs busy)
    {
        const size_t lds = 96 * 1024;
        CK(hipFuncSetAttribute((const void*)k_v_mad_u32_u24<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int trips = 8192;
        void* args[] = {&ctx.out, &trips, &ctx.cyc};
        const double t = time_launch((const void*)k_v_mad_u32_u24<4>, dim3(ctx.cus), dim3(1024), args, lds);
        long long h; CK(hipMemcpy(&h, ctx.cyc, 8, hipMemcpyDeviceToHost));
        ctx.f_hz = h / t;
        printf("clock64() advanced %lld in a %.3f ms launch (all SIMDs busy, 4 waves each): %.1f MHz -- used as the shader clock below\n", h, t * 1e3, ctx.f_hz * 1e-6);
    }
    printf("\ncycles a SIMD spends per wave64 instruction = wall time x clock / (waves per SIMD x instructions per wave); W = waves per SIMD, all %d SIMDs busy\n", 4 * ctx.cus);
    printf("%-22s %s\n", "", "1 dependent chain: W=1    W=2    W=4    W=8  | 4 independent chains: W=1  W=2    W=4    W=8");
    REPORT(s_nop_15);
    REPORT(v_mov_b32); REPORT(v_add_u32); REPORT(v_sub_u32); REPORT(v_and_b32); REPORT(v_or_b32); REPORT(v_xor_b32); REPORT(v_lshlrev_b32); REPORT(v_lshrrev_b32);
    REPORT(v_max_u32); REPORT(v_min_u32); REPORT(v_max_i32); REPORT(v_max_u16); REPORT(v_add_u16); REPORT(v_mul_u32_u24); REPORT(v_mad_u32_u24); REPORT(v_mul_lo_u32);
    REPORT(v_add3_u32); REPORT(v_lshl_add_u32); REPORT(v_lshl_or_b32); REPORT(v_and_or_b32); REPORT(v_or3_b32); REPORT(v_bfe_u32); REPORT(v_bfi_b32); REPORT(v_perm_b32);
    REPORT(v_alignbyte_b32); REPORT(v_alignbit_b32); REPORT(v_dot4_u32_u8); REPORT(v_dot2_u32_u16); REPORT(v_sad_u8);
    REPORT(v_pk_min_i16); REPORT(v_pk_max_u16); REPORT(v_pk_add_u16); REPORT(v_pk_sub_i16); REPORT(v_pk_mul_lo_u16); REPORT(v_pk_mad_i16); REPORT(v_pk_lshlrev_b16);
    REPORT(v_med3_u32); REPORT(v_min3_u32); REPORT(v_max3_u32); REPORT(v_bcnt_u32_b32); REPORT(v_mbcnt_lo); REPORT(v_mbcnt_hi);
    REPORT(v_cndmask_vcc); REPORT(v_cndmask_e64_vcc); report(ctx, "v_cmp vcc+v_cndmask vcc", k_v_cndmask_vcc_set<1>, k_v_cndmask_vcc_set<4>, 2); REPORT(v_addc_co_u32);
    REPORT(v_lshlrev_b32_v); REPORT(v_lshlrev_b32_e64); REPORT(v_add_u32_e64); REPORT(v_sub_u16); REPORT(v_min_u16); REPORT(v_mul_lo_u16); REPORT(v_lshrrev_b16); REPORT(v_ashrrev_i32);
    REPORT(v_max_f32); REPORT(v_min_f32); REPORT(v_sub_f32); REPORT(v_subrev_u32); REPORT(v_xnor_b32);
    REPORT(v_cndmask_sgpr); REPORT(v_cmp_gt_u32_vcc); REPORT(v_cmp_sgpr);
    report(ctx, "v_cmp+v_cndmask (pair)", k_v_cmp_then_cndmask<1>, k_v_cmp_then_cndmask<4>, 2);
    REPORT(v_add_co_u32); REPORT(v_readlane); REPORT(v_readfirstlane);
    REPORT(v_mov_dpp_row_shr); REPORT(v_add_dpp_row_shr); REPORT(v_add_dpp_wave_shr); REPORT(v_max_u32_sdwa_b); REPORT(v_add_u32_sdwa_b);
    REPORT(v_cvt_f32_u32); REPORT(v_cvt_i32_f32); REPORT(v_cvt_f32_ubyte0); REPORT(v_rndne_f32); REPORT(v_add_f32); REPORT(v_mul_f32); REPORT(v_fma_f32); REPORT(v_mac_f32); REPORT(v_rcp_f32);
    REPORT(v_pk_fma_f32); REPORT(v_pk_add_f32); REPORT(v_pk_mul_f32); REPORT(v_fma_f64); REPORT(v_add_f64); REPORT(v_mul_f64); REPORT(v_lshlrev_b64);
    REPORT(ds_read_b32); REPORT(ds_read_u8);
    REPORT(s_add_u32); REPORT(s_and_b64); REPORT(s_bcnt1_i32_b64); REPORT(s_waitcnt_idle); REPORT(s_nop_0); REPORT(s_mov_b32);
    report(ctx, "s_and_saveexec + s_mov exec (pair)", k_s_saveexec_pair<1>, k_s_saveexec_pair<4>, 1);
    report(ctx, "s_cmp + s_cbranch + s_nop (triple)", k_s_cmp_cbranch<1>, k_s_cmp_cbranch<4>, 1);
    report(ctx, "4 independent s_add (per four)", k_s_add_indep4<1>, k_s_add_indep4<4>, 1);
    report(ctx, "v_mad + s_add (per pair)", k_mix_vslow_s<1>, k_mix_vslow_s<4>, 1);
    report(ctx, "v_add + s_add (per pair)", k_mix_vfast_s<1>, k_mix_vfast_s<4>, 1);
    report(ctx, "v_add + 2 s (per triple)", k_mix_vfast_s_s<1>, k_mix_vfast_s_s<4>, 1);
    report(ctx, "v_mad + ds_read_u8 (pair)", k_mix_vslow_lds<1>, k_mix_vslow_lds<4>, 1);
    REPORT(v_bitop3_b32); REPORT(v_max_i16); REPORT(v_max3_u16); REPORT(v_mad_i32_i24); REPORT(v_mul_i32_i24); REPORT(v_pk_min_i16_plain);
    REPORT(v_cmp_lt_i32_sdwa); REPORT(v_cmp_gt_i16_sgpr); REPORT(v_cmp_gt_u16_vcc); REPORT(v_add_u32_sgpr); REPORT(v_xor_b32_lit);
    // what the fp32 peak works out to from the measured issue cost
    {
        const size_t lds = 96 * 1024;
        int trips = 2048;
        void* args[] = {&ctx.out, &trips, &ctx.cyc};
        const double t1 = time_launch((const void*)k_v_fma_f32<4>, dim3(ctx.cus), dim3(1024), args, lds);
        const double t2 = time_launch((const void*)k_v_pk_fma_f32<4>, dim3(ctx.cus), dim3(1024), args, lds);
        const double n = (double)trips * kUnroll * 4 * 4 * 4 * ctx.cus;   // wave instructions on the chip
        printf("\nfp32 rate on the whole chip (4 waves per SIMD, 4 chains): v_fma_f32 %.1f TFLOP/s, v_pk_fma_f32 %.1f TFLOP/s (quoted vector peak 157.3)\n",
               n * 64 * 2 / t1 * 1e-12, n * 64 * 4 / t2 * 1e-12);
    }
    return 0;
}
