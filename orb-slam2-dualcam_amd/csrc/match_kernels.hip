// match_kernels.hip -- Hamming matcher kernels for gfx950.
//
// Replaces the inner loop shared by every ORBmatcher::Search*/Fuse member of the reference
// (src/ORBmatcher.cc): DescriptorDistance :2015-2031 (256-bit XOR + popcount), best / second-best
// with strict "<" (:208-231), accept best <= TH and best < ratio * second (:233-236), 30-bin rotation
// histogram + ComputeThreeMaxima (:241-251, :272-290, :1969-2010).
//
// knn2 layout: a workgroup = 128 queries x 4 train splits (one wave per split, two queries per lane,
// the queries' 4 x u64 descriptor words live in VGPRs). Train descriptors are staged through LDS in
// tiles of 256 (8 KB, coalesced 16-B loads); every lane of a wave reads the SAME train word, an LDS
// broadcast, and spends 4 x (v_xor + popcount) per distance. The two smallest distances (with
// multiplicity) and the first index of the minimum are order-independent quantities, so the four
// partial results merge exactly into what the reference's sequential loop produces.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <vector>

#include "common.h"
#include "config.h"

namespace dcs {

struct Best { int b1, idx, b2; };

// branch-free form of: if (d < b1) { b2 = b1; b1 = d; idx = j; } else if (d < b2) b2 = d;   (ORBmatcher.cc:221-230)
__device__ __forceinline__ void best_update(Best& s, int dist, int j)
{
    const bool lt1 = dist < s.b1;
    s.b2 = lt1 ? s.b1 : min(s.b2, dist);
    s.idx = lt1 ? j : s.idx;
    s.b1 = min(s.b1, dist);
}

// merge of partial results over disjoint candidate sets (any order)
__device__ __forceinline__ Best best_merge(const Best& a, const Best& b)
{
    Best r;
    if (a.b1 < b.b1 || (a.b1 == b.b1 && (unsigned)a.idx <= (unsigned)b.idx)) { r.b1 = a.b1; r.idx = a.idx; r.b2 = min(a.b2, b.b1); }
    else { r.b1 = b.b1; r.idx = b.idx; r.b2 = min(b.b2, a.b1); }
    r.b2 = min(r.b2, min(a.b2, b.b2));
    return r;
}

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr int kTile = 256;
constexpr unsigned kKeyInit = (256u << 23) | 0x7FFFFFu;      // distance 256 = "none" (ORBmatcher.cc:218-219), index field all ones

__device__ __forceinline__ void load_query(const uint8_t* __restrict__ q, int qi, int nq, unsigned (&qw)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) qw[k] = 0;
    if (qi < nq) {
        const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi * 32);
        const uint4 a = qp[0], b = qp[1];
        qw[0] = a.x; qw[1] = a.y; qw[2] = a.z; qw[3] = a.w; qw[4] = b.x; qw[5] = b.y; qw[6] = b.z; qw[7] = b.w;
    }
}

// popcount(q ^ t) over 256 bits as one v_bcnt_u32_b32 accumulate chain (8 xor + 8 bcnt, no separate adds)
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc)
{
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ unsigned hamming256(const unsigned (&q)[8], const uint4& lo, const uint4& hi)
{
    unsigned d = bcnt_acc(q[0] ^ lo.x, 0u);
    d = bcnt_acc(q[1] ^ lo.y, d); d = bcnt_acc(q[2] ^ lo.z, d); d = bcnt_acc(q[3] ^ lo.w, d);
    d = bcnt_acc(q[4] ^ hi.x, d); d = bcnt_acc(q[5] ^ hi.y, d); d = bcnt_acc(q[6] ^ hi.z, d); d = bcnt_acc(q[7] ^ hi.w, d);
    return d;
}

// workgroup = 128 queries x 4 train splits: every lane keeps TWO queries in VGPRs (q_tile*128 + lane and + 64), so each
// LDS broadcast of a train descriptor (4 x ds_read_b64) feeds two distance computations.
__device__ __forceinline__ void knn2_tile(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                          const uint8_t* __restrict__ t_mask, int q_tile, int32_t* __restrict__ best_idx,
                                          int32_t* __restrict__ best_d, int32_t* __restrict__ second_d)
{
    __shared__ uint4 s_t[kTile * 2];                 // 256 descriptors x 32 B
    __shared__ unsigned s_k1[4][128], s_k2[4][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qa = q_tile * 128 + lane, qb = qa + 64;
    unsigned wa[8], wb[8];
    load_query(q, qa, nq, wa);
    load_query(q, qb, nq, wb);
    // best / second-best as packed keys (distance << 23 | train index): keys are unique, so the two smallest keys carry
    // exactly (smallest distance, first index attaining it) and (second-smallest distance counted with multiplicity) --
    // the same triple as best_update(), in 3 instructions per distance: key = lshl_or, k2 = med3(k1, k2, key), k1 = min.
    unsigned a1 = kKeyInit, a2 = kKeyInit, b1 = kKeyInit, b2 = kKeyInit;
    for (int t0 = 0; t0 < nt; t0 += kTile) {
        const int n_here = min(kTile, nt - t0);
        __syncthreads();
        {
            const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)t0 * 32);
            for (int i = tid; i < n_here * 2; i += 256) s_t[i] = tp[i];
        }
        __syncthreads();
        const int jb = wave * 64, je = min(jb + 64, n_here);             // wave-uniform (scalar) loop bounds
#pragma unroll 4
        for (int j = jb; j < je; ++j) {
            if (t_mask && t_mask[t0 + j]) continue;
            const uint4 lo = s_t[2 * j], hi = s_t[2 * j + 1];           // LDS broadcast
            const unsigned da = hamming256(wa, lo, hi), db = hamming256(wb, lo, hi);
            const unsigned ka = (da << 23) | (unsigned)(t0 + j), kb = (db << 23) | (unsigned)(t0 + j);
            a2 = umed3(a1, a2, ka); a1 = min(a1, ka);
            b2 = umed3(b1, b2, kb); b1 = min(b1, kb);
        }
    }
    s_k1[wave][lane] = a1; s_k2[wave][lane] = a2; s_k1[wave][lane + 64] = b1; s_k2[wave][lane + 64] = b2;
    __syncthreads();
    if (tid < 128) {
        const int qi = q_tile * 128 + tid;
        if (qi < nq) {
            unsigned k1 = s_k1[0][tid], k2 = s_k2[0][tid];
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {                             // two smallest of the union (keys are distinct)
                const unsigned o1 = s_k1[wv][tid], o2 = s_k2[wv][tid];
                k2 = min(min(k2, o2), max(k1, o1));
                k1 = min(k1, o1);
            }
            best_idx[qi] = (k1 >> 23) >= 256u ? -1 : (int)(k1 & 0x7FFFFFu); best_d[qi] = (int)(k1 >> 23); second_d[qi] = (int)(k2 >> 23);
        }
    }
}

__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                              const uint8_t* __restrict__ t_mask, int32_t* best_idx, int32_t* best_d, int32_t* second_d)
{
    knn2_tile(q, nq, t, nt, t_mask, blockIdx.x, best_idx, best_d, second_d);
}

// ---- Hamming knn2 on the i8 matrix cores ------------------------------------------------------------------------------
// popcount(q ^ t) = popcount(q) + popcount(t) - 2 <q, t> with the descriptors expanded to 256 bytes of 0 / 1: the inner
// product is an i8 GEMM, 4 x v_mfma_i32_16x16x64_i8 per 16 x 16 block of distances, on a pipe the rest of the front end
// leaves idle, and the vector ALU only keeps the best / second-best keys (5 instructions per distance instead of 19.5).
// A operand = 16 train descriptors (rows), B operand = 16 queries (columns): lane (c = l & 15, g = l >> 4) feeds, for
// K-slice s, the 16 bytes expanded from 16-bit chunk (4 s + g) of its row / column -- A and B use the same chunk-to-lane
// rule, so the products pair up bit by bit whatever the hardware's k ordering inside the instruction is. The accumulator
// puts <train 4 g + r, query c> in register r of lane (c, g): every lane tracks ITS query over its 4 train rows per block,
// and the four lanes of a query merge at the end (exact: keys are distinct).
// Workgroup = 4 waves x 64 queries (4 B fragments in VGPRs per wave); train tiles of 128 descriptors are expanded
// cooperatively into LDS (272-byte rows: 16 B of padding against bank conflicts) together with their popcounts.
typedef int v4i_t __attribute__((ext_vector_type(4)));
constexpr int kMTile = 128;                    // train descriptors per LDS tile
constexpr int kMRow = 272;                     // bytes per expanded descriptor in LDS (256 + 16 padding)

__device__ __forceinline__ unsigned spread4(unsigned nib)      // 4 bits -> 4 bytes of 0 / 1
{ return __umul24(nib, 0x204081u) & 0x01010101u; }

__device__ __forceinline__ v4i_t expand16(unsigned h16)        // 16 bits -> 16 bytes of 0 / 1 (bit b -> byte b)
{
    v4i_t v;
    v.x = (int)spread4(h16 & 15u); v.y = (int)spread4((h16 >> 4) & 15u); v.z = (int)spread4((h16 >> 8) & 15u); v.w = (int)spread4((h16 >> 12) & 15u);
    return v;
}

template <int QG, int WAVES>
__device__ __forceinline__ void knn2_tile_mfma(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt, int q_tile,
                                               int32_t* __restrict__ best_idx, int32_t* __restrict__ best_d, int32_t* __restrict__ second_d)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_t[2][kMTile * kMRow];     // double-buffered: one barrier per tile
    __shared__ __attribute__((aligned(16))) int s_pb[2][kMTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    constexpr int kParts = WAVES / 2;              // threads per train descriptor in the expansion (64 WAVES / 128)
    const int qbase = (q_tile * WAVES + wave) * (16 * QG);
    // B fragments + popcounts of this lane's 4 queries (query qbase + 16 qg + c, chunks 4 s + g)
    v4i_t bf[QG][4];
    int pa[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int qi = qbase + 16 * qg + c;
        uint4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (qi < nq) { const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi * 32); lo = qp[0]; hi = qp[1]; }
        const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        pa[qg] = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]) + __popc(w[4]) + __popc(w[5]) + __popc(w[6]) + __popc(w[7]);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            // chunk 4 sl + g = halfword (g & 1) of dword 2 sl + (g >> 1)
            // (two arrays and a select: written as w[2 sl + (g >> 1)] -- which is what the compiler makes of a select inside ONE array -- the words go to scratch memory)
            const unsigned w_even[4] = {lo.x, lo.z, hi.x, hi.z}, w_odd[4] = {lo.y, lo.w, hi.y, hi.w};
            const unsigned dw = (g >> 1) ? w_odd[sl] : w_even[sl];
            // queries enter as bytes 0 / -1: the accumulator holds -<q, t>, and the key is one v_lshl_add_u32
            const v4i_t e = expand16((g & 1) ? (dw >> 16) : (dw & 0xFFFFu));
            bf[qg][sl] = v4i_t{(e.x << 8) - e.x, (e.y << 8) - e.y, (e.z << 8) - e.z, (e.w << 8) - e.w};
        }
    }
    // Keys are tracked WITHOUT the query's own popcount (a constant per lane and query group that cannot change the order):
    // key = (popcount(t) - 2 <q, t> + 512) << 22 | train index -- 3 instructions per distance (lshl_add, med3, min): the train's
    // popcount and index are merged into one word when the tile is expanded.
    constexpr unsigned kInitM = (0x3FFu << 22) | 0x3FFFFFu;
    unsigned k1[QG], k2[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) { k1[qg] = kInitM; k2[qg] = kInitM; }
    auto expand_tile = [&](int t0, int buf) {
        // expand the tile: kParts threads per descriptor, 8 / kParts dwords (-> 16 bytes each per 16-bit chunk) per thread
        {
            constexpr int kDw = 8 / kParts;
            const int d = tid / kParts, part = tid % kParts, ti = t0 + d;
            unsigned w[kDw];
#pragma unroll
            for (int k = 0; k < kDw; ++k) w[k] = 0;
            if (ti < nt) {
                const unsigned* src = reinterpret_cast<const unsigned*>(t + (size_t)ti * 32) + part * kDw;
#pragma unroll
                for (int k = 0; k < kDw; ++k) w[k] = src[k];
            }
            uint8_t* row = s_t[buf] + d * kMRow + part * (32 * kDw);
            int pc = 0;
#pragma unroll
            for (int k = 0; k < kDw; ++k) {                    // dword k -> two chunks
                *reinterpret_cast<v4i_t*>(row + 32 * k) = expand16(w[k] & 0xFFFFu);
                *reinterpret_cast<v4i_t*>(row + 32 * k + 16) = expand16(w[k] >> 16);
                pc += __popc(w[k]);
            }
#pragma unroll
            for (int sh = 1; sh < kParts; sh <<= 1) pc += __shfl_xor(pc, sh);
            // everything of the key that does not depend on the query: biased popcount | train index (padding rows: largest distance, index >= nt)
            if (part == 0) s_pb[buf][d] = ((ti < nt ? pc + 512 : 0x3FF) << 22) | ti;
        }
    };
    expand_tile(0, 0);
    __syncthreads();
    for (int t0 = 0, buf = 0; t0 < nt; t0 += kMTile, buf ^= 1) {
        if (t0 + kMTile < nt) expand_tile(t0 + kMTile, buf ^ 1);   // next tile: its VALU / LDS work overlaps this tile's MFMAs
        const int n_grp = (min(kMTile, nt - t0) + 15) >> 4;
        v4i_t af[4], pbA, pbB;
        auto load_group = [&](int tg, v4i_t& pb) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) af[sl] = *reinterpret_cast<const v4i_t*>(s_t[buf] + (16 * tg + c) * kMRow + 16 * (4 * sl + g));
            pb = *reinterpret_cast<const v4i_t*>(&s_pb[buf][16 * tg + 4 * g]);              // biased popcounts of train rows 4 g + r
        };
        // one group of 16 train rows: 4 x QG matrix instructions, the NEXT group's operands requested behind them (into the other
        // popcount register set: the loop is unrolled by two so that no register copy is needed), then the keys of this group
        auto group = [&](int tg, const v4i_t& pb_now, v4i_t& pb_next) {
            v4i_t acc[QG];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = v4i_t{0, 0, 0, 0};
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)                       // slice-major: consecutive MFMAs hit different accumulators
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[sl], bf[qg][sl], acc[qg], 0, 0, 0);
            if (tg + 1 < n_grp) load_group(tg + 1, pb_next);     // next group's operands fly while this group's keys are ranked
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (popcount(t) + 512 - 2 <q, t>) << 22 | index; padding rows of the last group: acc = 0, largest key, index >= nt
                    const unsigned key = ((unsigned)acc[qg][r] << 23) + (unsigned)pb_now[r];       // one v_lshl_add_u32
                    k2[qg] = umed3(k1[qg], k2[qg], key); k1[qg] = min(k1[qg], key);
                }
            }
        };
        load_group(0, pbA);
        for (int tg = 0; tg < n_grp; tg += 2) {
            group(tg, pbA, pbB);
            if (tg + 1 < n_grp) group(tg + 1, pbB, pbA);
        }
        __syncthreads();
    }
    // the 4 lanes (g = 0..3) of a query hold disjoint train rows: two smallest of the union
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        unsigned a1 = k1[qg], a2 = k2[qg];
#pragma unroll
        for (int d = 16; d <= 32; d <<= 1) {
            const unsigned o1 = (unsigned)__shfl_xor((int)a1, d), o2 = (unsigned)__shfl_xor((int)a2, d);
            a2 = min(min(a2, o2), max(a1, o1));
            a1 = min(a1, o1);
        }
        const int qi = qbase + 16 * qg + c;
        if (g == 0 && qi < nq) {
            // un-bias: distance = key_distance - 512 + popcount(q); a key that never met a real train row keeps index >= nt
            const int i1 = (int)(a1 & 0x3FFFFFu), i2 = (int)(a2 & 0x3FFFFFu);
            const int d1 = i1 < nt ? (int)(a1 >> 22) - 512 + pa[qg] : 256, d2 = i2 < nt ? (int)(a2 >> 22) - 512 + pa[qg] : 256;
            best_idx[qi] = d1 >= 256 ? -1 : i1;
            best_d[qi] = min(d1, 256); second_d[qi] = min(d2, 256);
        }
    }
}

// ---- the same knn2 on the block-scaled FP4 matrix instruction (gfx950) --------------------------------------------------------
// v_mfma_scale_f32_16x16x128_f8f6f4 with E2M1 operands is an exact AND-popcount engine that also FINISHES the key
// (scratch/probe/fp4_probe.hip pins the operand maps and the exactness): a train bit enters as the nibble 0x2 (+1.0), a query bit as
// 0xC (-2.0), the A scale is 2^14, and the accumulator starts from the train row's key base (popcount(t) + 512) * 2^14 + index as a
// float -- so D = key base - 2 <q, t> * 2^14 = the finished key (< 2^24: exact in f32; positive floats order like integers). Against
// the i8 form: TWO instructions of K = 128 per 16 x 16 block of distances instead of four of K = 64 at about the same issue time each,
// no v_lshl_add per distance (2 vector instructions per distance: v_med3_f32 + v_min_f32), 128 instead of 256 operand bytes per
// descriptor in LDS (36 KB per workgroup instead of 70). The index field has 14 bits: problems of more than 16 383 descriptors per slot
// keep the i8 kernel.
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
constexpr int kIdxBits = 14;
constexpr int kMRow4 = 144;                    // bytes per expanded descriptor in LDS (128 + 16 padding: rows 36 banks apart)
constexpr int kFp4MaxCap = (1 << kIdxBits) - 1;

__device__ __forceinline__ unsigned spread8n(unsigned b)       // 8 bits -> 8 nibbles of 0 / 1 (bit i -> nibble i); b < 256
{
    unsigned x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    return (x | (x << 3)) & 0x11111111u;
}
template <bool QUERY>
__device__ __forceinline__ v4i_t expand32_fp4(unsigned w)       // 32 bits -> 32 E2M1 nibbles: +1.0 (train) or -2.0 (query) per set bit
{
    v4i_t v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned x = spread8n((w >> (8 * j)) & 0xFFu);
        v[j] = (int)(QUERY ? (x << 2) | (x << 3) : x << 1);
    }
    return v;
}

template <int QG, int WAVES>
__device__ __forceinline__ void knn2_tile_fp4(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt, int q_tile,
                                              int32_t* __restrict__ best_idx, int32_t* __restrict__ best_d, int32_t* __restrict__ second_d)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_t[2][kMTile * kMRow4];    // double-buffered: one barrier per tile
    __shared__ __attribute__((aligned(16))) float s_pb[2][kMTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    constexpr int kParts = WAVES / 2;              // threads per train descriptor in the expansion (64 WAVES / 128)
    const int qbase = (q_tile * WAVES + wave) * (16 * QG);
    // B fragments + popcounts of this lane's queries: K-slice sl of lane group g = dword 4 sl + g of the descriptor (A uses the same rule)
    v8i_t bf[QG][2];
    int pa[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int qi = qbase + 16 * qg + c;
        uint4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (qi < nq) { const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi * 32); lo = qp[0]; hi = qp[1]; }
        pa[qg] = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        const unsigned w0 = g == 0 ? lo.x : g == 1 ? lo.y : g == 2 ? lo.z : lo.w, w1 = g == 0 ? hi.x : g == 1 ? hi.y : g == 2 ? hi.z : hi.w;
        const v4i_t e0 = expand32_fp4<true>(w0), e1 = expand32_fp4<true>(w1);
        bf[qg][0] = v8i_t{e0.x, e0.y, e0.z, e0.w, 0, 0, 0, 0};
        bf[qg][1] = v8i_t{e1.x, e1.y, e1.z, e1.w, 0, 0, 0, 0};
    }
    // keys as floats, WITHOUT the query's own popcount (a constant per lane and query group that cannot change the order)
    constexpr float kInit = 16777215.0f;           // (1023 << 14) | 16383: index >= nt for every problem this kernel takes
    float k1[QG], k2[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) { k1[qg] = kInit; k2[qg] = kInit; }
    auto expand_tile = [&](int t0, int buf) {
        constexpr int kDw = 8 / kParts;
        const int d = tid / kParts, part = tid % kParts, ti = t0 + d;
        unsigned w[kDw];
#pragma unroll
        for (int k = 0; k < kDw; ++k) w[k] = 0;
        if (ti < nt) {
            const unsigned* src = reinterpret_cast<const unsigned*>(t + (size_t)ti * 32) + part * kDw;
#pragma unroll
            for (int k = 0; k < kDw; ++k) w[k] = src[k];
        }
        uint8_t* row = s_t[buf] + d * kMRow4 + part * (16 * kDw);
        int pc = 0;
#pragma unroll
        for (int k = 0; k < kDw; ++k) {                        // dword k -> 16 operand bytes
            *reinterpret_cast<v4i_t*>(row + 16 * k) = expand32_fp4<false>(w[k]);
            pc += __popc(w[k]);
        }
#pragma unroll
        for (int sh = 1; sh < kParts; sh <<= 1) pc += __shfl_xor(pc, sh);
        // everything of the key that does not depend on the query (padding rows: largest distance, index >= nt)
        if (part == 0) s_pb[buf][d] = (float)((((ti < nt ? pc + 512 : 0x3FF) << kIdxBits) | ti));
    };
    constexpr int kScaleA = 127 + kIdxBits, kScaleB = 127;      // E8M0 scales 2^14 and 2^0 (byte 0 of the scale operands: op_sel 0)
    expand_tile(0, 0);
    __syncthreads();
    for (int t0 = 0, buf = 0; t0 < nt; t0 += kMTile, buf ^= 1) {
#if !defined(DCS_KNN4_SKIP) || DCS_KNN4_SKIP != 1                  // timing-only side builds (wrong results): 1 = tiles are not re-expanded, 2 = no key ranking, 3 = no matrix instructions
#ifdef DCS_KNN4_DEPHASE
        // the two waves that share a SIMD (w and w + 4) work in opposite order inside a tile: one re-expands the next tile (vector + LDS writes) while
        // the other runs this tile's matrix instructions and key ranking -- the barrier per tile otherwise keeps every wave in the same phase
        const bool expand_late = ((wave >> 2) & 1) != 0;
        if (!expand_late && t0 + kMTile < nt) expand_tile(t0 + kMTile, buf ^ 1);
#else
        if (t0 + kMTile < nt) expand_tile(t0 + kMTile, buf ^ 1);   // next tile: its VALU / LDS work overlaps this tile's MFMAs
#endif
#endif
        const int n_grp = (min(kMTile, nt - t0) + 15) >> 4;
        v8i_t af[2];
        v4f_t pbA, pbB;
        auto load_group = [&](int tg, v4f_t& pb) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const v4i_t a = *reinterpret_cast<const v4i_t*>(s_t[buf] + (16 * tg + c) * kMRow4 + 16 * (4 * sl + g));
                af[sl] = v8i_t{a.x, a.y, a.z, a.w, 0, 0, 0, 0};
            }
            pb = *reinterpret_cast<const v4f_t*>(&s_pb[buf][16 * tg + 4 * g]);               // key bases of train rows 4 g + r
        };
        auto group = [&](int tg, const v4f_t& pb_now, v4f_t& pb_next) {
            v4f_t acc[QG];
#if defined(DCS_KNN4_SKIP) && DCS_KNN4_SKIP == 3
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = pb_now + v4f_t{(float)af[0][qg], (float)af[1][qg], (float)bf[qg][0][0], (float)bf[qg][1][1]};
#else
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[0], bf[qg][0], pb_now, 4, 4, 0, kScaleA, 0, kScaleB);
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[1], bf[qg][1], acc[qg], 4, 4, 0, kScaleA, 0, kScaleB);
#endif
            if (tg + 1 < n_grp) load_group(tg + 1, pb_next);     // next group's operands fly while this group's keys are ranked
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float key = acc[qg][r];
#if defined(DCS_KNN4_SKIP) && DCS_KNN4_SKIP == 2
                    if (r == 0) k1[qg] = fminf(k1[qg], (acc[qg][0] + acc[qg][1]) + (acc[qg][2] + acc[qg][3]));
#else
                    k2[qg] = __builtin_amdgcn_fmed3f(k1[qg], k2[qg], key); k1[qg] = fminf(k1[qg], key);
#endif
                }
            }
        };
        load_group(0, pbA);
        for (int tg = 0; tg < n_grp; tg += 2) {
            group(tg, pbA, pbB);
            if (tg + 1 < n_grp) group(tg + 1, pbB, pbA);
        }
#if defined(DCS_KNN4_DEPHASE) && (!defined(DCS_KNN4_SKIP) || DCS_KNN4_SKIP != 1)
        if (expand_late && t0 + kMTile < nt) expand_tile(t0 + kMTile, buf ^ 1);
#endif
        __syncthreads();
    }
    // the 4 lanes (g = 0..3) of a query hold disjoint train rows: two smallest of the union
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        float a1 = k1[qg], a2 = k2[qg];
#pragma unroll
        for (int d = 16; d <= 32; d <<= 1) {
            const float o1 = __shfl_xor(a1, d), o2 = __shfl_xor(a2, d);
            a2 = fminf(fminf(a2, o2), fmaxf(a1, o1));
            a1 = fminf(a1, o1);
        }
        const int qi = qbase + 16 * qg + c;
        if (g == 0 && qi < nq) {
            // un-bias: distance = key_distance - 512 + popcount(q); a key that never met a real train row keeps index >= nt
            const int u1 = (int)a1, u2 = (int)a2;
            const int i1 = u1 & ((1 << kIdxBits) - 1), i2 = u2 & ((1 << kIdxBits) - 1);
            const int d1 = i1 < nt ? (u1 >> kIdxBits) - 512 + pa[qg] : 256, d2 = i2 < nt ? (u2 >> kIdxBits) - 512 + pa[qg] : 256;
            best_idx[qi] = d1 >= 256 ? -1 : i1;
            best_d[qi] = min(d1, 256); second_d[qi] = min(d2, 256);
        }
    }
}

#ifndef DCS_KNN_QG                 // tuning hooks (scratch/ab builds): query groups of 16 per wave, waves per workgroup
#define DCS_KNN_QG 2
#endif
#ifndef DCS_KNN_WAVES
#define DCS_KNN_WAVES 8
#endif
constexpr int kKnnQG = DCS_KNN_QG, kKnnWaves = DCS_KNN_WAVES, kKnnQ = 16 * kKnnQG * kKnnWaves;      // 256 queries per workgroup
__global__ __launch_bounds__(64 * kKnnWaves) void k_knn2_pairs_mfma(const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_feat, int cap,
                                                         const int32_t* __restrict__ pairs, int32_t* best_idx, int32_t* best_d, int32_t* second_d)
{
    const int p = blockIdx.y;
    const int qs = pairs[2 * p], ts = pairs[2 * p + 1];
    const int nq = min(n_feat[qs], cap), nt = min(n_feat[ts], cap);
    if ((int)blockIdx.x * kKnnQ >= nq) return;
    knn2_tile_mfma<kKnnQG, kKnnWaves>(desc + (size_t)qs * cap * 32, nq, desc + (size_t)ts * cap * 32, nt, blockIdx.x,
                   best_idx + (size_t)p * cap, best_d + (size_t)p * cap, second_d + (size_t)p * cap);
}

#ifndef DCS_KNN4_QG
#define DCS_KNN4_QG 2
#endif
constexpr int kKnn4QG = DCS_KNN4_QG, kKnn4Q = 16 * kKnn4QG * kKnnWaves;
#ifndef DCS_KNN4_WPE               // tuning hook: minimum waves per SIMD the register allocation must allow (0 = compiler's choice)
#define DCS_KNN4_WPE 0
#endif
#if DCS_KNN4_WPE
__global__ __launch_bounds__(64 * kKnnWaves) __attribute__((amdgpu_waves_per_eu(DCS_KNN4_WPE))) void k_knn2_pairs_fp4(
#else
__global__ __launch_bounds__(64 * kKnnWaves) void k_knn2_pairs_fp4(
#endif
    const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_feat, int cap,
                                                        const int32_t* __restrict__ pairs, int32_t* best_idx, int32_t* best_d, int32_t* second_d)
{
    const int p = blockIdx.y;
    const int qs = pairs[2 * p], ts = pairs[2 * p + 1];
    const int nq = min(n_feat[qs], cap), nt = min(n_feat[ts], cap);
    if ((int)blockIdx.x * kKnn4Q >= nq) return;
    knn2_tile_fp4<kKnn4QG, kKnnWaves>(desc + (size_t)qs * cap * 32, nq, desc + (size_t)ts * cap * 32, nt, blockIdx.x,
                  best_idx + (size_t)p * cap, best_d + (size_t)p * cap, second_d + (size_t)p * cap);
}

// the matrix-core knn2 of n_pairs (query slot, train slot) problems: FP4 form when the 14-bit index field holds the slot, i8 form otherwise
// (DCS_KNN2_I8=1 forces the i8 form: the A/B and test hook)
static void launch_knn2_pairs_mfma(const uint8_t* desc, const int32_t* n_feat, int cap, const int32_t* pairs, int n_pairs, int32_t* best_idx,
                                   int32_t* best_d, int32_t* second_d, hipStream_t s)
{
    const bool force_i8 = opt(OPT_KNN2_I8) != 0;
    const int lds_pad = 0;
    if (cap <= kFp4MaxCap && !force_i8)
        hipLaunchKernelGGL(k_knn2_pairs_fp4, dim3((cap + kKnn4Q - 1) / kKnn4Q, n_pairs), dim3(64 * kKnnWaves), (size_t)lds_pad, s, desc, n_feat, cap, pairs, best_idx, best_d, second_d);
    else
        hipLaunchKernelGGL(k_knn2_pairs_mfma, dim3((cap + kKnnQ - 1) / kKnnQ, n_pairs), dim3(64 * kKnnWaves), 0, s, desc, n_feat, cap, pairs, best_idx, best_d, second_d);
}

// grouped (CSR buckets): one wave per group, one query per lane, candidates read through t_idx
__global__ __launch_bounds__(64) void k_knn2_grouped(const uint8_t* __restrict__ q, const uint8_t* __restrict__ t,
                                                     const int32_t* __restrict__ q_off, const int32_t* __restrict__ q_idx,
                                                     const int32_t* __restrict__ t_off, const int32_t* __restrict__ t_idx,
                                                     int32_t* best_idx, int32_t* best_d, int32_t* second_d)
{
    const int g = blockIdx.x;
    const int qb = q_off[g], qe = q_off[g + 1], tb = t_off[g], te = t_off[g + 1];
    for (int a = qb + threadIdx.x; a < qe; a += 64) {
        const int qi = q_idx[a];
        const unsigned long long* qp = reinterpret_cast<const unsigned long long*>(q + (size_t)qi * 32);
        const unsigned long long q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
        Best st{256, -1, 256};
        for (int b = tb; b < te; ++b) {
            const int j = t_idx[b];
            const unsigned long long* w = reinterpret_cast<const unsigned long long*>(t + (size_t)j * 32);
            const int dist = __popcll(q0 ^ w[0]) + __popcll(q1 ^ w[1]) + __popcll(q2 ^ w[2]) + __popcll(q3 ^ w[3]);
            best_update(st, dist, j);
        }
        best_idx[qi] = st.idx; best_d[qi] = st.b1; second_d[qi] = st.b2;
    }
}

__global__ void k_fill_knn(int32_t* best_idx, int32_t* best_d, int32_t* second_d, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { best_idx[i] = -1; best_d[i] = 256; second_d[i] = 256; }
}

// ---- accept + rotation histogram; one workgroup per problem
__device__ __forceinline__ int rot_bin(float aq, float at)
{
    const float factor = 1.0f / 30;                      // 1.0f/HISTO_LENGTH (:181)
    float rot = __fsub_rn(aq, at);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));       // round(): half away from zero (:246)
    if (bin == 30) bin = 0;
    return bin;
}

__device__ __forceinline__ bool accept(int idx, int b1, int b2, int th, int th_strict, float ratio)
{
    if (idx < 0) return false;
    if (th_strict ? !(b1 < th) : !(b1 <= th)) return false;
    return (float)b1 < __fmul_rn(ratio, (float)b2);
}

// note: `match` may alias `best_idx` (each thread reads best_idx[i] before it writes match[i])
__device__ void filter_problem(int nq, const int32_t* best_idx, const int32_t* __restrict__ best_d,
                               const int32_t* __restrict__ second_d, int th, int th_strict, float ratio, int check_ori,
                               const float* __restrict__ q_ang, int q_stride, const float* __restrict__ t_ang, int t_stride,
                               int32_t* match, int32_t* __restrict__ n_matches)
{
    __shared__ int s_hist[32];
    __shared__ int s_ind[3];
    __shared__ int s_count;
    const int tid = threadIdx.x;
    if (tid < 32) s_hist[tid] = 0;
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (check_ori) {
        for (int i = tid; i < nq; i += blockDim.x) {
            if (accept(best_idx[i], best_d[i], second_d[i], th, th_strict, ratio))
                atomicAdd(&s_hist[rot_bin(q_ang[(size_t)i * q_stride], t_ang[(size_t)best_idx[i] * t_stride])], 1);
        }
        __syncthreads();
        if (tid == 0) {                                  // ComputeThreeMaxima (:1969-2010)
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < 30; ++i) {
                const int s = s_hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
        }
        __syncthreads();
    }
    int local = 0;
    for (int i = tid; i < nq; i += blockDim.x) {
        int m = -1;
        if (accept(best_idx[i], best_d[i], second_d[i], th, th_strict, ratio)) {
            m = best_idx[i];
            if (check_ori) {
                const int b = rot_bin(q_ang[(size_t)i * q_stride], t_ang[(size_t)m * t_stride]);
                if (b != s_ind[0] && b != s_ind[1] && b != s_ind[2]) m = -1;
            }
        }
        match[i] = m;
        local += m >= 0;
    }
    atomicAdd(&s_count, local);
    __syncthreads();
    if (tid == 0) *n_matches = s_count;
}

__global__ __launch_bounds__(256) void k_filter(int nq, const int32_t* best_idx, const int32_t* best_d, const int32_t* second_d,
                                                int th, int th_strict, float ratio, int check_ori, const float* q_ang, int q_stride,
                                                const float* t_ang, int t_stride, int32_t* match, int32_t* n_matches)
{
    filter_problem(nq, best_idx, best_d, second_d, th, th_strict, ratio, check_ori, q_ang, q_stride, t_ang, t_stride, match, n_matches);
}

__global__ __launch_bounds__(256) void k_filter_pairs(const dcs_keypoint* __restrict__ kp, const int32_t* __restrict__ n_feat, int cap,
                                                      const int32_t* __restrict__ pairs, const int32_t* best_idx, const int32_t* best_d,
                                                      const int32_t* second_d, int th, float ratio, int check_ori, int32_t* match,
                                                      int32_t* n_matches)
{
    const int p = blockIdx.x;
    const int qs = pairs[2 * p], ts = pairs[2 * p + 1];
    const int nq = min(n_feat[qs], cap);
    const size_t o = (size_t)p * cap;
    filter_problem(nq, best_idx + o, best_d + o, second_d + o, th, 0, ratio, check_ori, &kp[(size_t)qs * cap].angle, 7,
                   &kp[(size_t)ts * cap].angle, 7, match + o, n_matches + p);
}


// ---- faithful SearchByBoWCrossCam (ORBmatcher.cc:162-294): one wave per shared vocabulary node. The
// reference is sequential over the KF features of a node because a query skips F features already
// claimed by an earlier query (:216); nodes are independent (every F feature lives in exactly one node),
// so the wave walks the node's queries in order and spreads each query's candidates over its lanes.
constexpr int kBowMaxCand = 4096;       // F features per node held as "claimed" bits in LDS

__device__ __forceinline__ Best wave_merge(Best s)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Best o;
        o.b1 = __shfl_xor(s.b1, d); o.idx = __shfl_xor(s.idx, d); o.b2 = __shfl_xor(s.b2, d);
        s = best_merge(s, o);
    }
    return s;
}

// The three BoW-guided matchers of the reference share this loop (queries of a vocabulary node in list order, candidates of
// the same node, candidates claimed by an earlier query are skipped); they differ in what is a valid candidate and when a
// query accepts:
//   kBowFKF   SearchByBoWCrossCam(F, cF, KF, cKF)        :162-294   best <= TH_LOW and best < ratio * second, first minimum
//   kBowKFKF  SearchByBoWCrossCam(KF1, c1, KF2, c2)      :297-414   best <  TH_LOW (strict) and ratio; candidates need a good MapPoint
//   kBowTri   SearchForTriangulation(KF1, KF2, F12, camS) :1253-1427 best only, dist <= TH_LOW, LAST minimum (`dist > bestDist` skips, a tie
//             replaces), candidates without a MapPoint, farther than 10 * sqrt(scale) px from the epipole and on the epipolar
//             line (CheckDistEpipolarLine :74-91, chi2 3.84 * sigma2 of the candidate's octave)
enum { kBowFKF = 0, kBowKFKF = 1, kBowTri = 2 };
struct EpipolarD {
    float F[9], ex, ey;                               // F12 row-major, epipole of KF1's camera centre in KF2's image
    const float *x1, *y1, *x2, *y2;                   // undistorted keypoint coordinates (mvvkeysUnTemp[camS])
    const int32_t* oct2;
    const float *sigma2, *scale;                      // mvLevelSigma2, mvScaleFactors
};

template <int MODE>
__global__ __launch_bounds__(64) void k_search_bow(const uint8_t* __restrict__ desc_kf, const float* __restrict__ ang_kf,
                                                   const uint8_t* __restrict__ kf_valid, const uint8_t* __restrict__ desc_f,
                                                   const float* __restrict__ ang_f, const uint8_t* __restrict__ f_valid,
                                                   const int32_t* __restrict__ node_pairs,
                                                   const int32_t* __restrict__ kf_off, const int32_t* __restrict__ kf_idx,
                                                   const int32_t* __restrict__ f_off, const int32_t* __restrict__ f_idx, float ratio,
                                                   int check_ori, EpipolarD ep, int32_t* __restrict__ match_f, int32_t* __restrict__ bin_f,
                                                   int32_t* __restrict__ hist, int32_t* __restrict__ overflow)
{
    __shared__ uint32_t s_claimed[kBowMaxCand / 32];
    const int a = node_pairs[2 * blockIdx.x], b = node_pairs[2 * blockIdx.x + 1], lane = threadIdx.x;
    const int qb = kf_off[a], qe = kf_off[a + 1], cb = f_off[b], ce = f_off[b + 1];
    if (ce - cb > kBowMaxCand) { if (lane == 0) *overflow = 1; return; }
    for (int i = lane; i < kBowMaxCand / 32; i += 64) s_claimed[i] = 0;
    __syncthreads();
    for (int qa = qb; qa < qe; ++qa) {
        const int ikf = kf_idx[qa];
        if (!kf_valid[ikf]) continue;                                      // wave-uniform
        const unsigned long long* qp = reinterpret_cast<const unsigned long long*>(desc_kf + (size_t)ikf * 32);
        const unsigned long long q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
        Best st{256, -1, 256};                                             // idx = position in the node's candidate list
        unsigned tri_key = 0xFFFFFFFFu;                                    // kBowTri: (dist << 16) | (0xFFFF - position): least distance, last position
        float la = 0.f, lb = 0.f, lc = 0.f;
        if (MODE == kBowTri) {                                             // epipolar line in the second image l = x1' F12 (:77-79), float, left to right
            const float x1 = ep.x1[ikf], y1 = ep.y1[ikf];
            la = __fadd_rn(__fadd_rn(__fmul_rn(x1, ep.F[0]), __fmul_rn(y1, ep.F[3])), ep.F[6]);
            lb = __fadd_rn(__fadd_rn(__fmul_rn(x1, ep.F[1]), __fmul_rn(y1, ep.F[4])), ep.F[7]);
            lc = __fadd_rn(__fadd_rn(__fmul_rn(x1, ep.F[2]), __fmul_rn(y1, ep.F[5])), ep.F[8]);
        }
        for (int c = cb + lane; c < ce; c += 64) {
            const int pos = c - cb;
            if (s_claimed[pos >> 5] & (1u << (pos & 31))) continue;
            const int jf = f_idx[c];
            if (MODE != kBowFKF && !f_valid[jf]) continue;
            const unsigned long long* w = reinterpret_cast<const unsigned long long*>(desc_f + (size_t)jf * 32);
            const int dist = __popcll(q0 ^ w[0]) + __popcll(q1 ^ w[1]) + __popcll(q2 ^ w[2]) + __popcll(q3 ^ w[3]);
            if (MODE != kBowTri) { best_update(st, dist, pos); continue; }
            if (dist > 50) continue;                                       // :1326 (the running bestDist only matters for the order: see the key)
            const float x2 = ep.x2[jf], y2 = ep.y2[jf];
            const int o2 = ep.oct2[jf];
            const float dex = __fsub_rn(ep.ex, x2), dey = __fsub_rn(ep.ey, y2);
            if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.0f, ep.scale[o2])) continue;      // :1331-1334
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, x2), __fmul_rn(lb, y2)), lc);
            const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
            if (den == 0.f) continue;
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)ep.sigma2[o2])) continue;                                             // :90, compared in double
            tri_key = min(tri_key, ((unsigned)dist << 16) | (unsigned)(0xFFFF - pos));
        }
        bool accept;
        int best_pos;
        if (MODE == kBowTri) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) tri_key = min(tri_key, (unsigned)__shfl_xor((int)tri_key, d));
            accept = tri_key != 0xFFFFFFFFu;
            best_pos = 0xFFFF - (int)(tri_key & 0xFFFFu);
        } else {
            st = wave_merge(st);                                           // first position of the minimum wins, like the loop
            accept = (MODE == kBowKFKF ? st.b1 < 50 : st.b1 <= 50) && (float)st.b1 < __fmul_rn(ratio, (float)st.b2);
            best_pos = st.idx;
        }
        if (accept) {
            if (lane == 0) {
                const int jf = f_idx[cb + best_pos];
                s_claimed[best_pos >> 5] |= 1u << (best_pos & 31);
                match_f[jf] = ikf;
                if (check_ori) { const int bn = rot_bin(ang_kf[ikf], ang_f[jf]); bin_f[jf] = bn; atomicAdd(&hist[bn], 1); }
            }
        }
        __syncthreads();
    }
}

// ComputeThreeMaxima + removal of matches outside the three dominant rotation bins (:272-290); single block
__global__ __launch_bounds__(256) void k_bow_finish(int n_f, int check_ori, const int32_t* __restrict__ hist, const int32_t* __restrict__ bin_f,
                                                    int32_t* __restrict__ match_f, int32_t* __restrict__ n_matches)
{
    __shared__ int s_ind[3];
    __shared__ int s_count;
    if (threadIdx.x == 0) {
        s_count = 0;
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; ++i) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
    }
    __syncthreads();
    int local = 0;
    for (int j = threadIdx.x; j < n_f; j += 256) {
        int m = match_f[j];
        if (m >= 0 && check_ori) {
            const int b = bin_f[j];
            if (b != s_ind[0] && b != s_ind[1] && b != s_ind[2]) { m = -1; match_f[j] = -1; }
        }
        local += m >= 0;
    }
    atomicAdd(&s_count, local);
    __syncthreads();
    if (threadIdx.x == 0) *n_matches = s_count;
}

__global__ void k_fill_i32(int32_t* p, int n, int32_t v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:270-340), one wave per map point. Lane i owns row i of the N x N
// distance table (rows i, i + 64, ... for N > 64) and never stores it: its descriptor sits in VGPRs, the others are
// broadcast loads, and the row median sorted[(int)(0.5 (N - 1))] is found by bisection on the distance value
// (9 counting passes over the row, distances in 0..256) instead of a sort. First row with the least median wins.
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ pool, const int32_t* __restrict__ off,
                                                    const int32_t* __restrict__ idx, int32_t* __restrict__ best)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    const int b0 = off[p], N = off[p + 1] - b0;
    if (N <= 0) { if (lane == 0) best[p] = -1; return; }
    const int32_t* id = idx + b0;
    const int k = (int)(0.5 * (N - 1));                      // rank of the median in the sorted row (:325)
    unsigned best_key = 0xFFFFFFFFu;                         // (median << 20) | row: minimum = least median, first row
    for (int i = lane; i < N; i += 64) {
        const uint4* qp = reinterpret_cast<const uint4*>(pool + (size_t)id[i] * 32);
        const uint4 qa = qp[0], qb = qp[1];
        const unsigned qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        int lo = 0, hi = 256;                                // smallest v with #(d_ij <= v) >= k + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; ++j) {
                const uint4* tp = reinterpret_cast<const uint4*>(pool + (size_t)id[j] * 32);
                const uint4 lo4 = tp[0], hi4 = tp[1];
                cnt += (int)hamming256(qw, lo4, hi4) <= mid;
            }
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        best_key = min(best_key, ((unsigned)lo << 20) | (unsigned)i);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best_key = min(best_key, (unsigned)__shfl_xor((int)best_key, d));
    if (lane == 0) best[p] = (int)(best_key & 0xFFFFFu);
}

}  // namespace dcs

using namespace dcs;

namespace {

// One (query set, train set) problem of the host-buffer entry points on the i8 matrix-core kernel: the two descriptor sets
// become slots 0 and 1 of a two-slot feature array (the layout dcs_match_bf_batch_device works on), pair (0, 1).
// Below this many distances the plain popcount kernel has less set-up to amortise.
constexpr long long kMfmaMinDistances = 64LL * 64LL;
struct TwoSlot {
    uint8_t* desc = nullptr; dcs_keypoint* kp = nullptr; int32_t *n = nullptr, *pairs = nullptr, *best_i = nullptr, *best_d = nullptr, *second_d = nullptr;
    int32_t* count = nullptr;                              // 16 spare ints right in front of best_i (a match count comes down in the same copy)
    int cap = 0;
};
int two_slot_upload(Scratch& s, const uint8_t* q, const dcs_keypoint* q_kp, int nq, const uint8_t* t, const dcs_keypoint* t_kp, int nt, TwoSlot& o)
{
    int rc;
    o.cap = (std::max(nq, nt) + 127) & ~127;               // whole train tiles: the kernel never reads past a slot
    // ONE staging image [descriptors slot 0 | slot 1 | counts, pair | key points slot 0 | slot 1] = one DMA operation; the unused tails
    // of the slots travel along (whatever the pinned block holds: the kernels read a slot only up to its count / mask the rest)
    const bool with_kp = q_kp && t_kp;
    const size_t desc_bytes = (size_t)2 * o.cap * 32, kp_bytes = with_kp ? sizeof(dcs_keypoint) * 2 * (size_t)o.cap : 0;
    uint8_t *d = nullptr, *h = nullptr;
    if ((rc = s.stage(&d, &h, desc_bytes + 16 + kp_bytes))) return rc;
    memcpy(h, q, (size_t)nq * 32);
    memcpy(h + (size_t)o.cap * 32, t, (size_t)nt * 32);
    const int32_t ints[4] = {nq, nt, 0, 1};
    memcpy(h + desc_bytes, ints, 16);
    o.desc = d; o.n = reinterpret_cast<int32_t*>(d + desc_bytes); o.pairs = o.n + 2;
    if (with_kp) {
        memcpy(h + desc_bytes + 16, q_kp, sizeof(dcs_keypoint) * (size_t)nq);
        memcpy(h + desc_bytes + 16 + sizeof(dcs_keypoint) * (size_t)o.cap, t_kp, sizeof(dcs_keypoint) * (size_t)nt);
        o.kp = reinterpret_cast<dcs_keypoint*>(d + desc_bytes + 16);
    }
    // results side by side: one download
    int32_t* res = nullptr;
    if ((rc = s.alloc(&res, (size_t)3 * o.cap + 16))) return rc;
    o.count = res; o.best_i = res + 16; o.best_d = o.best_i + o.cap; o.second_d = o.best_i + 2 * o.cap;
    return DCS_OK;
}

}  // namespace

extern "C" {

int dcs_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* t_mask, int32_t* best_idx,
                     int32_t* best_d, int32_t* second_d)
{
    if (nq < 0 || nt < 0 || (nq && (!q || !best_idx || !best_d || !second_d)) || (nt && !t)) { set_error("bad argument"); return DCS_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    if (nq == 0) return DCS_OK;
    if (nt >= (1 << 23)) { set_error("nt %d exceeds the 2^23 train descriptors of one knn2 problem", nt); return DCS_ERR_UNSUPPORTED; }
    Scratch s;
    if (!t_mask && (long long)nq * nt >= kMfmaMinDistances && nt < (1 << 22)) {         // the matrix-core kernel (no mask support)
        TwoSlot ts;
        if ((rc = two_slot_upload(s, q, nullptr, nq, t, nullptr, nt, ts))) return rc;
        launch_knn2_pairs_mfma(ts.desc, ts.n, ts.cap, ts.pairs, 1, ts.best_i, ts.best_d, ts.second_d, s.st);
        DCS_CHECK_LAUNCH();
        if ((rc = s.download_bytes(best_idx, ts.best_i, sizeof(int32_t) * nq)) || (rc = s.download_bytes(best_d, ts.best_d, sizeof(int32_t) * nq)) ||
            (rc = s.download_bytes(second_d, ts.second_d, sizeof(int32_t) * nq))) return rc;
        return s.finish();
    }
    uint8_t *dq, *dt, *dm = nullptr;
    int32_t *bi, *bd, *sd;
    if ((rc = s.upload(&dq, q, (size_t)nq * 32)) || (rc = s.upload(&dt, t, (size_t)nt * 32))) return rc;
    if (t_mask && (rc = s.upload(&dm, t_mask, (size_t)nt))) return rc;
    if ((rc = s.alloc(&bi, nq)) || (rc = s.alloc(&bd, nq)) || (rc = s.alloc(&sd, nq))) return rc;
    hipLaunchKernelGGL(k_knn2, dim3((nq + 127) / 128), dim3(256), 0, s.st, dq, nq, dt, nt, dm, bi, bd, sd);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(best_idx, bi, sizeof(int32_t) * nq))) return rc;
    if ((rc = s.download_bytes(best_d, bd, sizeof(int32_t) * nq))) return rc;
    if ((rc = s.download_bytes(second_d, sd, sizeof(int32_t) * nq))) return rc;
    return s.finish();
}

// CSR lists handed over by the caller (feature vectors, observation lists): offsets ascending from 0, indices inside [0, n_items)
static bool valid_csr(const int32_t* off, int n_groups, const int32_t* idx, int n_items)
{
    if (n_groups == 0) return true;
    if (off[0] != 0) return false;
    for (int g = 0; g < n_groups; ++g) if (off[g + 1] < off[g]) return false;
    if (off[n_groups] && !idx) return false;
    for (int i = 0; i < off[n_groups]; ++i) if (idx[i] < 0 || idx[i] >= n_items) return false;
    return true;
}

int dcs_hamming_knn2_grouped(const uint8_t* q, int nq, const uint8_t* t, int nt, int n_groups, const int32_t* q_off,
                             const int32_t* q_idx, const int32_t* t_off, const int32_t* t_idx, int32_t* best_idx,
                             int32_t* best_d, int32_t* second_d)
{
    if (nq < 0 || nt < 0 || n_groups < 0 || (n_groups && (!q_off || !t_off)) || (nq && (!q || !best_idx || !best_d || !second_d))) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    if (!valid_csr(q_off, n_groups, q_idx, nq) || !valid_csr(t_off, n_groups, t_idx, nt)) {
        set_error("dcs_hamming_knn2_grouped: offsets must ascend from 0 and indices lie inside the descriptor arrays"); return DCS_ERR_INVALID;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (nq == 0) return DCS_OK;
    Scratch s;
    uint8_t *dq, *dt;
    int32_t *bi, *bd, *sd, *dqo, *dqi, *dto, *dti;
    if ((rc = s.upload(&dq, q, (size_t)nq * 32)) || (rc = s.upload(&dt, t, (size_t)nt * 32))) return rc;
    if ((rc = s.alloc(&bi, nq)) || (rc = s.alloc(&bd, nq)) || (rc = s.alloc(&sd, nq))) return rc;
    hipLaunchKernelGGL(k_fill_knn, dim3((nq + 255) / 256), dim3(256), 0, s.st, bi, bd, sd, nq);
    DCS_CHECK_LAUNCH();
    if (n_groups) {
        const int nqi = q_off[n_groups], nti = t_off[n_groups];
        if ((rc = s.upload(&dqo, q_off, (size_t)n_groups + 1)) || (rc = s.upload(&dto, t_off, (size_t)n_groups + 1))) return rc;
        if ((rc = s.upload(&dqi, q_idx, (size_t)nqi)) || (rc = s.upload(&dti, t_idx, (size_t)nti))) return rc;
        hipLaunchKernelGGL(k_knn2_grouped, dim3(n_groups), dim3(64), 0, s.st, dq, dt, dqo, dqi, dto, dti, bi, bd, sd);
        DCS_CHECK_LAUNCH();
    }
    if ((rc = s.download_bytes(best_idx, bi, sizeof(int32_t) * nq))) return rc;
    if ((rc = s.download_bytes(best_d, bd, sizeof(int32_t) * nq))) return rc;
    if ((rc = s.download_bytes(second_d, sd, sizeof(int32_t) * nq))) return rc;
    return s.finish();
}

int dcs_match_filter(int nq, const int32_t* best_idx, const int32_t* best_d, const int32_t* second_d, int th, int th_strict,
                     float ratio, int check_ori, const float* q_angle, const float* t_angle, int32_t* match, int* n_matches)
{
    if (nq < 0 || !n_matches || (nq && (!best_idx || !best_d || !second_d || !match)) || (check_ori && nq && (!q_angle || !t_angle))) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    int rc = ensure_device();
    if (rc) return rc;
    *n_matches = 0;
    if (nq == 0) return DCS_OK;
    int max_t = 0;
    for (int i = 0; i < nq; ++i) max_t = std::max(max_t, best_idx[i] + 1);
    Scratch s;
    int32_t *bi, *bd, *sd, *dm, *dn;
    float *qa = nullptr, *ta = nullptr;
    if ((rc = s.upload(&bi, best_idx, nq)) || (rc = s.upload(&bd, best_d, nq)) || (rc = s.upload(&sd, second_d, nq))) return rc;
    if (check_ori && ((rc = s.upload(&qa, q_angle, nq)) || (rc = s.upload(&ta, t_angle, max_t)))) return rc;
    if ((rc = s.alloc(&dm, nq)) || (rc = s.alloc(&dn, 1))) return rc;
    hipLaunchKernelGGL(k_filter, dim3(1), dim3(256), 0, s.st, nq, bi, bd, sd, th, th_strict, ratio, check_ori, qa, 1, ta, 1, dm, dn);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(match, dm, sizeof(int32_t) * nq))) return rc;
    int32_t n32 = 0;
    if ((rc = s.download_bytes(&n32, dn, sizeof(int32_t)))) return rc;
    if ((rc = s.finish())) return rc;
    *n_matches = n32;
    return DCS_OK;
}

int dcs_match_bf(const uint8_t* q, const dcs_keypoint* q_kp, int nq, const uint8_t* t, const dcs_keypoint* t_kp, int nt,
                 int th, float ratio, int check_ori, int32_t* match, int* n_matches)
{
    if (nq < 0 || nt < 0 || !n_matches || (nq && (!q || !match)) || (nt && !t) || (check_ori && ((nq && !q_kp) || (nt && !t_kp)))) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    int rc = ensure_device();
    if (rc) return rc;
    *n_matches = 0;
    if (nq == 0) return DCS_OK;
    if (nt >= (1 << 23)) { set_error("nt %d exceeds the 2^23 train descriptors of one knn2 problem", nt); return DCS_ERR_UNSUPPORTED; }
    Scratch s;
    if ((long long)nq * nt >= kMfmaMinDistances && nt < (1 << 22) && (!check_ori || (q_kp && t_kp))) {   // matrix-core kernel + the batched filter
        TwoSlot ts;
        if ((rc = two_slot_upload(s, q, check_ori ? q_kp : nullptr, nq, t, check_ori ? t_kp : nullptr, nt, ts))) return rc;
        int32_t* dn1 = ts.count;
        if (!ts.kp && (rc = s.alloc(&ts.kp, 1))) return rc;                                  // never read without check_ori
        launch_knn2_pairs_mfma(ts.desc, ts.n, ts.cap, ts.pairs, 1, ts.best_i, ts.best_d, ts.second_d, s.st);
        DCS_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_filter_pairs, dim3(1), dim3(256), 0, s.st, ts.kp, ts.n, ts.cap, ts.pairs, ts.best_i, ts.best_d, ts.second_d, th, ratio, check_ori,
                           ts.best_i, dn1);
        DCS_CHECK_LAUNCH();
        int32_t n32 = 0;
        if ((rc = s.download_bytes(match, ts.best_i, sizeof(int32_t) * nq)) || (rc = s.download_bytes(&n32, dn1, sizeof(int32_t))) || (rc = s.finish())) return rc;
        *n_matches = n32;
        return DCS_OK;
    }
    uint8_t *dq, *dt;
    dcs_keypoint *kq = nullptr, *kt = nullptr;
    int32_t *bi, *bd, *sd, *dm, *dn;
    if ((rc = s.upload(&dq, q, (size_t)nq * 32)) || (rc = s.upload(&dt, t, (size_t)nt * 32))) return rc;
    if (check_ori && ((rc = s.upload(&kq, q_kp, nq)) || (rc = s.upload(&kt, t_kp, nt)))) return rc;
    if ((rc = s.alloc(&bi, nq)) || (rc = s.alloc(&bd, nq)) || (rc = s.alloc(&sd, nq)) || (rc = s.alloc(&dm, nq)) || (rc = s.alloc(&dn, 1))) return rc;
    hipLaunchKernelGGL(k_knn2, dim3((nq + 127) / 128), dim3(256), 0, s.st, dq, nq, dt, nt, (const uint8_t*)nullptr, bi, bd, sd);
    DCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_filter, dim3(1), dim3(256), 0, s.st, nq, bi, bd, sd, th, 0, ratio, check_ori,
                       kq ? &kq->angle : nullptr, 7, kt ? &kt->angle : nullptr, 7, dm, dn);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(match, dm, sizeof(int32_t) * nq))) return rc;
    int32_t n32 = 0;
    if ((rc = s.download_bytes(&n32, dn, sizeof(int32_t)))) return rc;
    if ((rc = s.finish())) return rc;
    *n_matches = n32;
    return DCS_OK;
}

int dcs_match_bf_batch_device(const uint8_t* d_desc, const dcs_keypoint* d_kp, const int32_t* d_n, int cap, const int32_t* d_pairs,
                              int n_pairs, int th, float ratio, int check_ori, int32_t* d_match, int32_t* d_n_matches,
                              int32_t* d_best_d, int32_t* d_second_d, void* stream)
{
    if (!d_desc || !d_kp || !d_n || !d_pairs || !d_match || !d_n_matches || !d_best_d || !d_second_d || cap < 1 || n_pairs < 0) {
        set_error("bad argument (d_best_d / d_second_d scratch is required)"); return DCS_ERR_INVALID;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (n_pairs == 0) return DCS_OK;
    if (cap >= (1 << 22)) { set_error("cap %d exceeds the 2^22 descriptors of one knn2 problem (22-bit index field of the matrix-core key)", cap); return DCS_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    // d_match doubles as the best-index buffer: the filter reads best_idx[i] and writes match[i] in the same thread
    launch_knn2_pairs_mfma(d_desc, d_n, cap, d_pairs, n_pairs, d_match, d_best_d, d_second_d, s);
    DCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_filter_pairs, dim3(n_pairs), dim3(256), 0, s, d_kp, d_n, cap, d_pairs, d_match, d_best_d, d_second_d, th,
                       ratio, check_ori, d_match, d_n_matches);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}


int dcs_distinctive_descriptors(const uint8_t* pool, int n_pool, const int32_t* off, const int32_t* idx, int n_points, int32_t* best)
{
    if (n_points < 0 || n_pool < 0 || (n_points && (!off || !best)) || (n_points && off[n_points] > 0 && (!idx || !pool))) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (n_points == 0) return DCS_OK;
    const int total = off[n_points];
    for (int p = 0; p < n_points; ++p) {
        if (off[p + 1] < off[p]) { set_error("offsets not ascending at point %d", p); return DCS_ERR_INVALID; }
        if (off[p + 1] - off[p] >= (1 << 20)) { set_error("more than 2^20 observations of one map point"); return DCS_ERR_UNSUPPORTED; }
    }
    for (int i = 0; i < total; ++i) if (idx[i] < 0 || idx[i] >= n_pool) { set_error("descriptor index %d out of range", idx[i]); return DCS_ERR_INVALID; }
    Scratch s;
    uint8_t* d_pool; int32_t *d_off, *d_idx, *d_best;
    if ((rc = s.upload(&d_pool, pool, (size_t)n_pool * 32)) || (rc = s.upload(&d_off, off, (size_t)n_points + 1)) ||
        (rc = s.upload(&d_idx, idx, (size_t)total)) || (rc = s.alloc(&d_best, n_points))) return rc;
    hipLaunchKernelGGL(k_distinctive, dim3(n_points), dim3(64), 0, s.st, d_pool, d_off, d_idx, d_best);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(best, d_best, sizeof(int32_t) * n_points))) return rc;
    if ((rc = s.finish())) return rc;
    return DCS_OK;
}

// shared host side of the three BoW-guided matchers: match_f[j] = query that took candidate j (or -1)
static int search_bow_impl(int mode, const uint8_t* desc_kf, const float* ang_kf, const uint8_t* kf_valid, int n_kf, const uint8_t* desc_f,
                           const float* ang_f, const uint8_t* f_valid, int n_f, const int32_t* kf_nodes, const int32_t* kf_off, const int32_t* kf_idx,
                           int kf_n_nodes, const int32_t* f_nodes, const int32_t* f_off, const int32_t* f_idx, int f_n_nodes, float ratio,
                           int check_ori, const dcs_epipolar* epi, int n_levels, int32_t* match_f, int* n_matches)
{
    if (n_kf < 0 || n_f < 0 || kf_n_nodes < 0 || f_n_nodes < 0 || !n_matches || (n_f && (!match_f || !desc_f)) || (n_kf && (!desc_kf || !kf_valid)) ||
        (kf_n_nodes && (!kf_nodes || !kf_off || !kf_idx)) || (f_n_nodes && (!f_nodes || !f_off || !f_idx)) || (check_ori && ((n_kf && !ang_kf) || (n_f && !ang_f))) ||
        (mode != kBowFKF && n_f && !f_valid)) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    if (!valid_csr(kf_off, kf_n_nodes, kf_idx, n_kf) || !valid_csr(f_off, f_n_nodes, f_idx, n_f)) {
        set_error("feature-vector offsets must ascend from 0 and indices lie inside the descriptor arrays"); return DCS_ERR_INVALID;
    }
    if (mode == kBowTri) {
        if (!epi || n_levels < 1 || n_levels > 16 || !epi->level_sigma2 || !epi->scale_factors || (n_kf && (!epi->kp1_x || !epi->kp1_y)) ||
            (n_f && (!epi->kp2_x || !epi->kp2_y || !epi->kp2_octave))) { set_error("bad epipolar geometry"); return DCS_ERR_INVALID; }
        for (int j = 0; j < n_f; ++j) if (epi->kp2_octave[j] < 0 || epi->kp2_octave[j] >= n_levels) { set_error("octave of feature %d outside the pyramid", j); return DCS_ERR_INVALID; }
    }
    int rc = ensure_device();
    if (rc) return rc;
    *n_matches = 0;
    for (int j = 0; j < n_f; ++j) match_f[j] = -1;
    // merge-join of the two sorted node lists (:189-269), on the host: O(nodes)
    std::vector<int32_t> np;
    for (int a = 0, b = 0; a < kf_n_nodes && b < f_n_nodes;) {
        if (kf_nodes[a] == f_nodes[b]) { np.push_back(a); np.push_back(b); ++a; ++b; }
        else if (kf_nodes[a] < f_nodes[b]) a = (int)(std::lower_bound(kf_nodes, kf_nodes + kf_n_nodes, f_nodes[b]) - kf_nodes);
        else b = (int)(std::lower_bound(f_nodes, f_nodes + f_n_nodes, kf_nodes[a]) - f_nodes);
    }
    const int n_shared = (int)np.size() / 2;
    if (n_shared == 0 || n_f == 0 || n_kf == 0) return DCS_OK;
    Scratch s;
    uint8_t *dkf, *df, *dval, *dfval = nullptr;
    float *akf = nullptr, *af = nullptr;
    int32_t *dnp, *dko, *dki, *dfo, *dfi, *dm, *dbin, *dhist, *dn;
    if ((rc = s.upload(&dkf, desc_kf, (size_t)n_kf * 32)) || (rc = s.upload(&df, desc_f, (size_t)n_f * 32)) || (rc = s.upload(&dval, kf_valid, (size_t)n_kf))) return rc;
    if (mode != kBowFKF && (rc = s.upload(&dfval, f_valid, (size_t)n_f))) return rc;
    if (check_ori && ((rc = s.upload(&akf, ang_kf, n_kf)) || (rc = s.upload(&af, ang_f, n_f)))) return rc;
    if ((rc = s.upload(&dnp, np.data(), np.size())) || (rc = s.upload(&dko, kf_off, (size_t)kf_n_nodes + 1)) || (rc = s.upload(&dki, kf_idx, (size_t)kf_off[kf_n_nodes])) ||
        (rc = s.upload(&dfo, f_off, (size_t)f_n_nodes + 1)) || (rc = s.upload(&dfi, f_idx, (size_t)f_off[f_n_nodes]))) return rc;
    if ((rc = s.alloc(&dm, n_f)) || (rc = s.alloc(&dbin, n_f)) || (rc = s.alloc(&dhist, 32)) || (rc = s.alloc(&dn, 2))) return rc;
    EpipolarD ep{};
    if (mode == kBowTri) {
        for (int i = 0; i < 9; ++i) ep.F[i] = epi->F12[i];
        ep.ex = epi->ex; ep.ey = epi->ey;
        float *x1, *y1, *x2, *y2, *sg, *sc; int32_t* o2;
        if ((rc = s.upload(&x1, epi->kp1_x, n_kf)) || (rc = s.upload(&y1, epi->kp1_y, n_kf)) || (rc = s.upload(&x2, epi->kp2_x, n_f)) ||
            (rc = s.upload(&y2, epi->kp2_y, n_f)) || (rc = s.upload(&o2, epi->kp2_octave, n_f)) || (rc = s.upload(&sg, epi->level_sigma2, n_levels)) ||
            (rc = s.upload(&sc, epi->scale_factors, n_levels))) return rc;
        ep.x1 = x1; ep.y1 = y1; ep.x2 = x2; ep.y2 = y2; ep.oct2 = o2; ep.sigma2 = sg; ep.scale = sc;
    }
    hipLaunchKernelGGL(k_fill_i32, dim3((n_f + 255) / 256), dim3(256), 0, s.st, dm, n_f, -1);
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, s.st, dhist, 32, 0);
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 0, s.st, dn, 2, 0);
    DCS_CHECK_LAUNCH();
    if (mode == kBowFKF)
        hipLaunchKernelGGL(k_search_bow<kBowFKF>, dim3(n_shared), dim3(64), 0, s.st, dkf, akf, dval, df, af, dfval, dnp, dko, dki, dfo, dfi, ratio, check_ori, ep, dm, dbin, dhist, dn + 1);
    else if (mode == kBowKFKF)
        hipLaunchKernelGGL(k_search_bow<kBowKFKF>, dim3(n_shared), dim3(64), 0, s.st, dkf, akf, dval, df, af, dfval, dnp, dko, dki, dfo, dfi, ratio, check_ori, ep, dm, dbin, dhist, dn + 1);
    else
        hipLaunchKernelGGL(k_search_bow<kBowTri>, dim3(n_shared), dim3(64), 0, s.st, dkf, akf, dval, df, af, dfval, dnp, dko, dki, dfo, dfi, ratio, check_ori, ep, dm, dbin, dhist, dn + 1);
    DCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, s.st, n_f, check_ori, dhist, dbin, dm, dn);
    DCS_CHECK_LAUNCH();
    int32_t h[2] = {0, 0};
    if ((rc = s.download_bytes(h, dn, sizeof(h)))) return rc;
    if ((rc = s.download_bytes(match_f, dm, sizeof(int32_t) * n_f))) return rc;
    if ((rc = s.finish())) return rc;
    if (h[1]) { set_error("a vocabulary node holds more than %d candidate features", kBowMaxCand); return DCS_ERR_UNSUPPORTED; }
    *n_matches = h[0];
    return DCS_OK;
}

// candidate-indexed result -> query-indexed (every query takes at most one candidate)
static void invert_matches(const std::vector<int32_t>& match_f, int n_q, int32_t* match_q)
{
    for (int i = 0; i < n_q; ++i) match_q[i] = -1;
    for (size_t j = 0; j < match_f.size(); ++j) if (match_f[j] >= 0) match_q[match_f[j]] = (int32_t)j;
}

int dcs_search_by_bow(const uint8_t* desc_kf, const float* ang_kf, const uint8_t* kf_valid, int n_kf, const uint8_t* desc_f,
                      const float* ang_f, int n_f, const int32_t* kf_nodes, const int32_t* kf_off, const int32_t* kf_idx,
                      int kf_n_nodes, const int32_t* f_nodes, const int32_t* f_off, const int32_t* f_idx, int f_n_nodes, float ratio,
                      int check_ori, int32_t* match_f, int* n_matches)
{
    return search_bow_impl(kBowFKF, desc_kf, ang_kf, kf_valid, n_kf, desc_f, ang_f, nullptr, n_f, kf_nodes, kf_off, kf_idx, kf_n_nodes, f_nodes, f_off, f_idx,
                           f_n_nodes, ratio, check_ori, nullptr, 0, match_f, n_matches);
}

int dcs_search_by_bow_kf(const uint8_t* desc1, const float* ang1, const uint8_t* valid1, int n1, const uint8_t* desc2, const float* ang2,
                         const uint8_t* valid2, int n2, const int32_t* nodes1, const int32_t* off1, const int32_t* idx1, int n_nodes1,
                         const int32_t* nodes2, const int32_t* off2, const int32_t* idx2, int n_nodes2, float ratio, int check_ori,
                         int32_t* match12, int* n_matches)
{
    if (n1 < 0 || n2 < 0 || (n1 && !match12)) { set_error("bad argument"); return DCS_ERR_INVALID; }
    std::vector<int32_t> mf((size_t)std::max(n2, 0), -1);
    const int rc = search_bow_impl(kBowKFKF, desc1, ang1, valid1, n1, desc2, ang2, valid2, n2, nodes1, off1, idx1, n_nodes1, nodes2, off2, idx2, n_nodes2,
                                   ratio, check_ori, nullptr, 0, mf.data(), n_matches);
    if (rc) return rc;
    invert_matches(mf, n1, match12);
    return DCS_OK;
}

int dcs_search_for_triangulation(const uint8_t* desc1, const float* ang1, const uint8_t* free1, int n1, const uint8_t* desc2, const float* ang2,
                                 const uint8_t* free2, int n2, const int32_t* nodes1, const int32_t* off1, const int32_t* idx1, int n_nodes1,
                                 const int32_t* nodes2, const int32_t* off2, const int32_t* idx2, int n_nodes2, const dcs_epipolar* epi,
                                 int check_ori, int32_t* match12, int* n_matches)
{
    if (n1 < 0 || n2 < 0 || (n1 && !match12) || !epi) { set_error("bad argument"); return DCS_ERR_INVALID; }
    std::vector<int32_t> mf((size_t)std::max(n2, 0), -1);
    const int rc = search_bow_impl(kBowTri, desc1, ang1, free1, n1, desc2, ang2, free2, n2, nodes1, off1, idx1, n_nodes1, nodes2, off2, idx2, n_nodes2,
                                   0.f, check_ori, epi, epi->n_levels, mf.data(), n_matches);
    if (rc) return rc;
    invert_matches(mf, n1, match12);
    return DCS_OK;
}

}  // extern "C"
