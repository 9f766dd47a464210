// orb_kernels.hip -- hand-written gfx950 kernels of the ORB front end.
//
// What each kernel replaces in the reference (src/ORBextractor.cc) and the OpenCV call it restates
// (OpenCV 3.3/3.4.0 non-IPP semantics):
//   k_resize        ComputePyramid :1107-1132        cv::resize INTER_LINEAR 8UC1 (11-bit fixed point)
//   k_fast_cells    ComputeKeyPointsOctTree :789-827  cv::FAST 9/16 + NMS per 30-px cell, iniTh -> minTh fallback
//   k_level_scan / k_gather                           vToDistributeKeys in emission order (cell-major, row-major)
//   k_blur          operator() :1085-1086             cv::GaussianBlur 7x7 sigma 2, REFLECT_101, legacy 8-bit path
//   k_describe      IC_Angle :77-104, computeOrbDescriptor :108-147, keypoint assembly :837-847,1095-1101
// All integer stages are bit-exact by construction; the float stages (fastAtan2 polynomial, rotated
// pattern coordinates) use explicitly rounded mul/add (no FMA contraction) and RNE conversions.
// Built with -ffp-contract=off.
#include "orb_kernels.h"

#include <cfloat>

#include "common.h"
#include "config.h"

namespace dcs {

__constant__ int8_t c_pattern[1024] = {
#include "brief_pattern.inc"
};

int upload_pattern() { return DCS_OK; }   // statically initialised __constant__ data: nothing to do

// ------------------------------------------------------------------------------------- resize
// cv::resize INTER_LINEAR 8UC1 (11-bit fixed point), streaming like the blur: a thread owns 4 adjacent destination
// pixels (one dword) of a strip of destination rows. Its 4 column taps {sx, a0, a1} are fixed for the whole strip, so
// each source row costs ONE 12-byte load (3 aligned dwords containing all 8 taps); per pixel the two taps are pulled
// out with one v_perm_b32 (as two zero-extended u16) and weighted with one v_dot2_u32_u16. The horizontally
// interpolated rows (already >> 4, as the vertical pass wants them) are cached in VGPRs because consecutive destination
// rows share source rows at scale 1.2. All row bookkeeping is wave-uniform (scalar registers and branches); the
// vertical pass is two 24-bit multiplies + one SDWA add of the high halves per pixel. No LDS, no barriers.
#ifndef DCS_RS_ROWS                          // tuning hook (scratch/ab builds); a power of two <= 64
#define DCS_RS_ROWS 16
#endif
constexpr int kRsRowsPerThread = DCS_RS_ROWS;
#ifndef DCS_RS_UPFRONT                       // tuning hook: request every source row of an interior strip before the first is used
#define DCS_RS_UPFRONT 1
#endif
#ifndef DCS_RS_SRCMAX
#define DCS_RS_SRCMAX 22
#endif
constexpr int kRsSrcMax = DCS_RS_SRCMAX;     // source rows of a strip held in registers (16 rows at scale <= 1.25: 21)

struct ResizeCol { int16_t sx, pad, a0, a1; };
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

struct ResizeTaps {                    // loop-invariant per thread
    bool upper[4];                     // taps of pixel k live in dwords (d1, d2) instead of (d0, d1)
    unsigned sel[4];                   // v_perm selector: byte0 = first tap, byte2 = second tap, bytes 1/3 = 0
    unsigned wgt[4];                   // a0 | a1 << 16
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct ResizeRaw { u32x4_t d; };                // x, y, z = the 12 source bytes one thread needs from one source row (one register tuple:
                                                // a conditionally reloaded row stays in place instead of being copied after the load)

__device__ __forceinline__ ResizeRaw resize_load(const uint8_t* __restrict__ row, bool aligned)
{
    ResizeRaw r;
    if (aligned) {
        const unsigned* p = reinterpret_cast<const unsigned*>(row);
        r.d = u32x4_t{p[0], p[1], p[2], 0u};
    } else {
        unsigned b[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] = row[k];
        r.d = u32x4_t{b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24),
                      b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24), 0u};
    }
    return r;
}

// horizontal pass of one source row for 4 destination pixels: (S0 a0 + S1 a1) >> 4
__device__ __forceinline__ void resize_hpass(const ResizeRaw& r, const ResizeTaps& t, unsigned (&h)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = t.upper[k] ? r.d.y : r.d.x, hi = t.upper[k] ? r.d.z : r.d.y;
        const unsigned taps = __builtin_amdgcn_perm(hi, lo, t.sel[k]);                  // (S0, S1) as two u16
        h[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, taps), __builtin_bit_cast(ushort2_t, t.wgt[k]), 0u, false) >> 4;
    }
}

__device__ __forceinline__ void resize_hrow(const uint8_t* __restrict__ row, bool aligned, const ResizeTaps& t, unsigned (&h)[4])
{
    resize_hpass(resize_load(row, aligned), t, h);
}

// vertical pass + store of one destination row: hA, hB <= 32640 and b <= 2048, so the 24-bit multiplies are exact; each product is
// truncated (>> 16) on its own like OpenCV's fixed-point VResizeLinear
__device__ __forceinline__ void resize_emit(uint8_t* __restrict__ out, const unsigned (&hA)[4], const unsigned (&hB)[4], unsigned b0, unsigned b1)
{
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned v = ((__umul24(b0, hA[k]) >> 16) + (__umul24(b1, hB[k]) >> 16) + 2u) >> 2;
        packed |= (v & 0xffu) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(out) = packed;
}

template <bool ALIGNED>      // source rows readable as dwords (base, image stride and pitch multiples of 4): decided on the host
__global__ __launch_bounds__(256) void k_resize(LevelView src, LevelView dst, const ResizeCol* __restrict__ cols,
                                                const int16_t* __restrict__ yofs, const int16_t* __restrict__ ya, int n_images, ResizeRects rects)
{
    // lanes run over (image, destination dword) pairs: every image of the batch has the same geometry and row tables, so a
    // wave stays uniform in y while its 64 lanes are all busy whatever the level width is (widths are not multiples of 256).
    // Round 5: the launch covers a list of destination rectangles (blockIdx.z; the whole level = one rectangle) -- with the FAST cells of
    // level l - 1 producing the inside of level l from their LDS tiles, this kernel only writes the frame no cell's ROI reaches.
    // (several rectangles: a 1-D grid, rectangle i owns blocks [blk_begin_i, blk_begin_{i+1}), nbx_i of them per 64-row strip group)
    int ri = 0, bx = blockIdx.x, by = blockIdx.y;
    if (rects.n > 1) {
        for (int i = 1; i < rects.n; ++i) if ((int)blockIdx.x >= rects.r[i].blk_begin) ri = i;
        const int b = (int)blockIdx.x - rects.r[ri].blk_begin;
        by = b / rects.r[ri].nbx; bx = b - by * rects.r[ri].nbx;
    }
    const ResizeRect rc = rects.r[ri];
    const int n_x4 = rc.x4_count, row_end = rc.row_end;
    const int li = bx * 64 + (int)threadIdx.x;
    const int img = li / max(n_x4, 1);
    const int dx0 = (rc.x4_begin + li - img * n_x4) * 4;
    const int dy0 = rc.row_begin + (by * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y)) * kRsRowsPerThread;   // wave-uniform
    if (dy0 >= row_end || n_x4 <= 0) return;
    // row tables of this strip: lane k (mod rows per thread) holds the entries of destination row dy0 + k; read back with v_readlane (the
    // loads are issued by every lane, before the out-of-range lanes of the last block leave)
    const int krow = min(dy0 + (int)(threadIdx.x & (kRsRowsPerThread - 1)), dst.h - 1);
    const int my_sy = yofs[krow];
    const unsigned my_a = *reinterpret_cast<const unsigned*>(ya + 2 * krow);           // b0 | b1 << 16 (both 0..2048)
    // the readlanes happen HERE, while all 64 lanes are still active: placed after the return below, the compiler sinks the two
    // loads behind it too and the lanes that left never fetch their entries (a wave with < 8 surviving lanes then read garbage)
    int sy_of[kRsRowsPerThread];
    unsigned a_of[kRsRowsPerThread];
#pragma unroll
    for (int k = 0; k < kRsRowsPerThread; ++k) { sy_of[k] = __builtin_amdgcn_readlane(my_sy, k); a_of[k] = (unsigned)__builtin_amdgcn_readlane((int)my_a, k); }
    if (img >= n_images) return;
    const uint8_t* S = src.base + (size_t)img * src.img_stride;
    uint8_t* D = const_cast<uint8_t*>(dst.base) + (size_t)img * dst.img_stride;
    constexpr bool aligned = ALIGNED;
    ResizeTaps t;
    int base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const ResizeCol cc = cols[min(dx0 + k, dst.w - 1)];
        // 12-byte window [base, base + 12) must hold every tap (offsets <= 11) and stay inside the row's storage
        if (k == 0) base = aligned ? min((int)cc.sx & ~3, src.pitch - 12) : max(0, min((int)cc.sx, src.w - 12));
        const int o = cc.sx - base;                          // 0..11
        t.upper[k] = o > 6;
        const unsigned o8 = (unsigned)(t.upper[k] ? o - 4 : o);              // offset inside the chosen 8-byte pair, 0..7
        // single-tap columns (sx == sw-1) carry a1 == 0; a second tap beyond the pair reads as constant zero (0x0c)
        t.sel[k] = 0x0c000c00u | o8 | ((o8 < 7 ? o8 + 1 : 0x0cu) << 16);
        t.wgt[k] = (unsigned)(uint16_t)cc.a0 | ((unsigned)(uint16_t)cc.a1 << 16);
    }
    const uint8_t* colbase = S + base;
    uint8_t* out = D + (size_t)dy0 * dst.pitch + dx0;                         // pitch is a multiple of 64: the dword stays in the row
    const int sy_first = sy_of[0], sy_last = sy_of[kRsRowsPerThread - 1];
#if DCS_RS_UPFRONT
    if (sy_first >= 0 && sy_last + 2 - sy_first <= kRsSrcMax) {
        // strip whose source rows fit the register window (every strip at scale <= 1.25): ALL of them are requested before the
        // first one is used -- the one-row look-ahead of the walk below keeps two loads in flight per wave, and with eight waves per SIMD
        // that is what bounded the kernel (Little: 8 x 2 x 768 B per ~1 us and SIMD = ~3 TB/s of requested bytes). The rows are consumed
        // in order (compile-time register names); the destination row that sits on source rows (j, j + 1) is emitted when j comes by --
        // its table entries fetched with v_readlane at the run-time row counter. The partial last strip of a level (its table entries
        // beyond the last row repeat that row's and are not emitted) and a strip that ends on the clamped last source row (the loads clamp the
        // row index: row j + 1 then IS row j) take this path too: the row-at-a-time loop below, one dependent load per destination row, made
        // the last workgroups of every launch its tail.
        const int n_rows = min(kRsRowsPerThread, row_end - dy0);
        ResizeRaw raw[kRsSrcMax];
#pragma unroll
        for (int j = 0; j < kRsSrcMax; ++j) raw[j] = resize_load(colbase + (size_t)min(sy_first + j, src.h - 1) * src.pitch, aligned);
        unsigned hP[4], hC[4];
        resize_hpass(raw[0], t, hP);
        int k = 0, sy_k = sy_first;
        unsigned a_k = a_of[0];
        uint8_t* o = out;
#pragma unroll
        for (int j = 0; j + 1 < kRsSrcMax; ++j) {
            resize_hpass(raw[j + 1], t, hC);
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                if (k < n_rows && sy_k == sy_first + j) {                    // wave-uniform
                    resize_emit(o, hP, hC, a_k & 0xffffu, a_k >> 16);
                    o += dst.pitch; ++k;
                    sy_k = __builtin_amdgcn_readlane(my_sy, k & (kRsRowsPerThread - 1));
                    a_k = (unsigned)__builtin_amdgcn_readlane((int)my_a, k & (kRsRowsPerThread - 1));
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) hP[c] = hC[c];
        }
        return;
    }
#endif
    if (dy0 + kRsRowsPerThread <= row_end && sy_first >= 0 && sy_last + 1 <= src.h - 1) {
        // interior strip (no clamped source row): walk the source rows once. hP / hC = horizontal passes of rows p, p + 1; the
        // raw bytes of row p + 2 are already in flight while the current destination row is produced. A destination row
        // advances p by 0..2 (scale <= 2), all of it wave-uniform (SGPR compares, constant-lane readlanes).
        int p = sy_first;
        unsigned hP[4], hC[4];
        const ResizeRaw r0 = resize_load(colbase + (size_t)p * src.pitch, aligned);
        const ResizeRaw r1 = resize_load(colbase + (size_t)(p + 1) * src.pitch, aligned);
        ResizeRaw nx = resize_load(colbase + (size_t)min(p + 2, src.h - 1) * src.pitch, aligned);
        resize_hpass(r0, t, hP);
        resize_hpass(r1, t, hC);
#pragma unroll
        for (int k = 0; k < kRsRowsPerThread; ++k) {
            const int sy = sy_of[k];
#pragma unroll
            for (int adv = 0; adv < 2; ++adv) {
                if (p < sy) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) hP[c] = hC[c];
                    resize_hpass(nx, t, hC);
                    ++p;
                    nx = resize_load(colbase + (size_t)min(p + 2, src.h - 1) * src.pitch, aligned);
                }
            }
            const unsigned a = a_of[k];
            resize_emit(out + (size_t)k * dst.pitch, hP, hC, a & 0xffffu, a >> 16);
        }
        return;
    }
    // boundary strips (clamped rows at the bottom edge, a partial last strip): one destination row at a time
    unsigned hA[4], hB[4];
    int rowA = -1, rowB = -1;                           // source rows currently held in hA / hB (wave-uniform)
    const int dy_end = min(dy0 + kRsRowsPerThread, row_end);
    for (int dy = dy0; dy < dy_end; ++dy) {
        const int sy = yofs[dy];
        const int sy0 = min(max(sy, 0), src.h - 1), sy1 = min(max(sy + 1, 0), src.h - 1);
        if (sy0 != rowA) {
            if (sy0 == rowB) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const unsigned tt = hA[k]; hA[k] = hB[k]; hB[k] = tt; }
                const int tt = rowA; rowA = rowB; rowB = tt;
            } else { resize_hrow(colbase + (size_t)sy0 * src.pitch, aligned, t, hA); rowA = sy0; }
        }
        if (sy1 == rowA) {
#pragma unroll
            for (int k = 0; k < 4; ++k) hB[k] = hA[k];
            rowB = rowA;
        } else if (sy1 != rowB) { resize_hrow(colbase + (size_t)sy1 * src.pitch, aligned, t, hB); rowB = sy1; }
        resize_emit(out + (size_t)(dy - dy0) * dst.pitch, hA, hB, (unsigned)(int)ya[2 * dy], (unsigned)(int)ya[2 * dy + 1]);
    }
}

int launch_resize(const LevelView& src, const LevelView& dst, const int16_t* d_cols, const int16_t* d_yofs, const int16_t* d_ya, int n_images, hipStream_t s,
                  const ResizeRects* rects)
{
    if ((double)src.w / dst.w > 2.0) { set_error("pyramid scale factor > 2 not supported by the resize kernel"); return DCS_ERR_UNSUPPORTED; }
    ResizeRects rr{};
    if (rects) rr = *rects;
    else { rr.n = 1; rr.r[0] = ResizeRect{0, (dst.w + 3) / 4, 0, dst.h}; }
    dim3 grid;
    if (rr.n == 1) {
        grid = dim3((n_images * rr.r[0].x4_count + 63) / 64, (rr.r[0].row_end - rr.r[0].row_begin + 4 * kRsRowsPerThread - 1) / (4 * kRsRowsPerThread));
        if (rr.r[0].x4_count <= 0 || rr.r[0].row_end <= rr.r[0].row_begin) return DCS_OK;
    } else {
        int total = 0;
        for (int i = 0; i < rr.n; ++i) {
            const int rows = rr.r[i].row_end - rr.r[i].row_begin;
            const bool empty = rr.r[i].x4_count <= 0 || rows <= 0;
            rr.r[i].blk_begin = total;
            rr.r[i].nbx = empty ? 1 : (n_images * rr.r[i].x4_count + 63) / 64;
            if (!empty) total += rr.r[i].nbx * ((rows + 4 * kRsRowsPerThread - 1) / (4 * kRsRowsPerThread));
        }
        if (total == 0) return DCS_OK;
        grid = dim3(total);
    }
    const bool aligned = ((reinterpret_cast<uintptr_t>(src.base) | (uintptr_t)src.img_stride | (uintptr_t)src.pitch) & 3) == 0 && src.pitch >= 12;
    if (aligned) hipLaunchKernelGGL(k_resize<true>, grid, dim3(64, 4), 0, s, src, dst, reinterpret_cast<const ResizeCol*>(d_cols), d_yofs, d_ya, n_images, rr);
    else hipLaunchKernelGGL(k_resize<false>, grid, dim3(64, 4), 0, s, src, dst, reinterpret_cast<const ResizeCol*>(d_cols), d_yofs, d_ya, n_images, rr);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// XCD-aware block mapping: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own
// 4 MB L2). Map a 1-D grid of n_images * per_image blocks so that all blocks of one image have ids congruent mod 8:
// neighbouring cells / tiles / keypoints of an image then share cache lines in ONE L2 instead of up to eight.
__device__ __forceinline__ void xcd_image_block(int b, int n_images, int per_image, int& img, int& item)
{
    const int full = n_images & ~7;                          // images forming complete groups of 8
    if (b < full * per_image) { const int j = b >> 3; img = (j / per_image) * 8 + (b & 7); item = j % per_image; }
    else { const int r = b - full * per_image; img = full + r / per_image; item = r % per_image; }
}

// ------------------------------------------------------------------------------------- FAST
typedef short short2_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

// FAST-9/16 score of ONE polarity of the pixel whose 7x7 neighbourhood starts at b (LDS bytes, row pitch P; the centre is
// b[3P + 3]): max over the 16 contiguous 9-arcs of min(sign * (ring - centre)), minus 1 (== OpenCV cornerScore<16> when the
// other polarity has no arc at the threshold, see k_fast_cells step 3). Ring pixel k and its opposite k + 8 share one
// register (lo = k, hi = k + 8): the arc k .. k + 8 is "k .. 7" of the low halves + "8 .. k + 8" of the high halves and the arc
// k + 8 .. k + 16 the same with the halves exchanged, so 7 suffix minima, 7 prefix minima and 8 half-swapped minima
// (v_pk_min_i16 op_sel) give all 16 arcs: 31 packed instructions per polarity instead of 59 for both polarities side by side.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)p; }
__device__ __forceinline__ unsigned pk_min_i16_swap(unsigned a, unsigned b)            // (min(a.lo, b.hi), min(a.hi, b.lo))
{
    unsigned r;
    asm("v_pk_min_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_min_i16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b)));
}
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b)));
}
// max over the 16 contiguous 9-arcs of the arc's minimum, for ring values X[k] = (x_k, x_{k+8}) in the (lo, hi) halves: 7 suffix + 7 prefix
// minima, 8 half-swapped minima, 7 maxima (all values are bytes: the signed packed forms order them like the unsigned ones)
__device__ __forceinline__ unsigned fast_arcs(const unsigned (&X)[8])
{
    unsigned S[8], Q[8];
    S[7] = X[7]; Q[0] = X[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { S[7 - k] = pk_min_i16(X[7 - k], S[8 - k]); Q[k] = pk_min_i16(Q[k - 1], X[k]); }
    unsigned best = pk_min_i16_swap(S[0], Q[0]);
#pragma unroll
    for (int k = 1; k < 8; ++k) best = pk_max_i16(best, pk_min_i16_swap(S[k], Q[k]));
    return best;                                         // (best over the arcs that start at 0..7, at 8..15): the caller takes the larger half
}
// Round 4: no subtraction per ring pixel and no packed compare. min over an arc of (r - v) = (min over the arc of r) - v, so the
// "brighter" score needs the arc minima of the RAW ring bytes; "darker" is the same on the complemented bytes x = 255 - r = r ^ 0xff
// (v - r = x - (255 - v)): the polarity enters as ONE xor per packed pair -- a plain 32-bit op, which issues at twice the rate of the
// v_pk_mad_i16 it replaces (profiles/r03_valu_rate_probe.txt) -- and the score is best(X) - (v ^ mask). Which polarity can hold an arc
// at the threshold is decided on the raw bytes with 16-bit min / max / sub (the fast class again) before anything is packed.
// D16Z: the eight "opposite" ring bytes come in through ds_read_u8_d16_hi, which on this hardware (gfx950 with SRAM ECC: "sramecc+" in the
// architecture string, scratch/probe/lds_probe.hip) writes the byte to bits 16-23 AND zeroes the low half -- the value arrives already
// shifted, and pack + polarity is ONE v_bitop3_b32 ((hi | lo) ^ mask, a fast-class op) per pair instead of a shift and a merge. The
// compiler neither emits that form by itself nor tracks the loads of an asm statement, so the 17 loads and their wait are one statement.
// The host picks D16Z = false (plain byte loads, shift + merge) on a device that does not report sramecc+.
template <int P, bool D16Z>
__device__ __forceinline__ int fast_score(const uint8_t* b, int th)
{
    constexpr int C = 3 * P + 3;
    // ring pixel k (OpenCV's order, starting below the centre) at b[off[k]], its opposite k + 8 at b[2C - off[k]]
    constexpr int off[8] = {C + 3 * P, C + 3 * P + 1, C + 2 * P + 2, C + P + 3, C + 3, C - P + 3, C - 2 * P + 2, C - 3 * P + 1};
    unsigned lo[8], hs[8], v;                            // hs[k] = ring byte k + 8, << 16
    if (D16Z) {
        const unsigned a = lds_addr(b);
        asm volatile("ds_read_u8 %0, %17 offset:%18\n\tds_read_u8 %1, %17 offset:%19\n\tds_read_u8 %2, %17 offset:%20\n\tds_read_u8 %3, %17 offset:%21\n\t"
                     "ds_read_u8 %4, %17 offset:%22\n\tds_read_u8 %5, %17 offset:%23\n\tds_read_u8 %6, %17 offset:%24\n\tds_read_u8 %7, %17 offset:%25\n\t"
                     "ds_read_u8_d16_hi %8, %17 offset:%26\n\tds_read_u8_d16_hi %9, %17 offset:%27\n\tds_read_u8_d16_hi %10, %17 offset:%28\n\tds_read_u8_d16_hi %11, %17 offset:%29\n\t"
                     "ds_read_u8_d16_hi %12, %17 offset:%30\n\tds_read_u8_d16_hi %13, %17 offset:%31\n\tds_read_u8_d16_hi %14, %17 offset:%32\n\tds_read_u8_d16_hi %15, %17 offset:%33\n\t"
                     "ds_read_u8 %16, %17 offset:%34\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(lo[6]), "=&v"(lo[7]),
                       "=&v"(hs[0]), "=&v"(hs[1]), "=&v"(hs[2]), "=&v"(hs[3]), "=&v"(hs[4]), "=&v"(hs[5]), "=&v"(hs[6]), "=&v"(hs[7]), "=&v"(v)
                     : "v"(a), "n"(off[0]), "n"(off[1]), "n"(off[2]), "n"(off[3]), "n"(off[4]), "n"(off[5]), "n"(off[6]), "n"(off[7]),
                       "n"(2 * C - off[0]), "n"(2 * C - off[1]), "n"(2 * C - off[2]), "n"(2 * C - off[3]), "n"(2 * C - off[4]), "n"(2 * C - off[5]),
                       "n"(2 * C - off[6]), "n"(2 * C - off[7]), "n"(C)
                     : "memory");
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo[k] = b[off[k]]; hs[k] = (unsigned)b[2 * C - off[k]] << 16; }
        v = b[C];
    }
    // a 9-arc covers two ADJACENT compass pixels (ring 0, 4, 8, 12): the test of k_fast_cells step 2, per polarity, on 16-bit lanes
    const uint16_t r0 = (uint16_t)lo[0], r4 = (uint16_t)lo[4], r8 = (uint16_t)(hs[0] >> 16), r12 = (uint16_t)(hs[4] >> 16), vc = (uint16_t)v;
    const uint16_t e = min(max(r0, r8), max(r4, r12)), f = max(min(r0, r8), min(r4, r12));
    // the two compares as SGPR masks (a ballot of the C++ bools is rebuilt by the compiler through v_cndmask + v_cmp_ne: two more vector
    // instructions per polarity); the selects below read those masks directly
    const uint16_t up = (uint16_t)(e - vc), down = (uint16_t)(vc - f);
    unsigned long long m_b, m_d;
    asm("v_cmp_gt_i16_e64 %0, %1, %2" : "=s"(m_b) : "v"(up), "v"((uint16_t)th));
    asm("v_cmp_gt_i16_e64 %0, %1, %2" : "=s"(m_d) : "v"(down), "v"((uint16_t)th));
    const unsigned long long m_both = m_b & m_d;
    unsigned m;
    asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(m) : "v"(0x00ff00ffu), "s"(m_b));       // brighter ? 0 : 0x00ff00ff
    unsigned X[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) X[k] = (hs[k] | lo[k]) ^ m;
    const short2_t bb = __builtin_bit_cast(short2_t, fast_arcs(X));
    int s = max((int)bb.x, (int)bb.y) - (int)(v ^ (m & 0xffu));
    if (m_both != 0) {                                   // wave-uniform, rare: some pixel passes the compass test in BOTH polarities -- its score is the larger one
#pragma unroll
        for (int k = 0; k < 8; ++k) X[k] ^= 0x00ff00ffu;                                  // (lanes that took the darker polarity above compute the brighter one here: unused)
        const short2_t dd = __builtin_bit_cast(short2_t, fast_arcs(X));
        const int s2 = max(s, max((int)dd.x, (int)dd.y) - (int)(v ^ 0xffu));
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(s) : "v"(s), "v"(s2), "s"(m_both));
    }
    return max(s, 1) - 1;
}

// Profiling builds (-DDCS_FAST_SECTIONS, scratch/fast_sections.sh) can end the kernel after a section to count the instructions
// of each one with the PMC counters; product builds compile the hook away.
#ifdef DCS_FAST_SECTIONS
#define DCS_FAST_SECTION(n) do { if (dbg_stop == (n)) { if (lane == 0) cell_count[(size_t)img * n_cells + cell] = 0; return; } } while (0)
#else
#define DCS_FAST_SECTION(n) do { } while (0)
#endif
// One wave per (cell, image) -- the reference's per-cell cv::FAST call (ORBextractor.cc:789-827):
//   1. ROI -> LDS with aligned dword loads (byte loads when the level is not 4-byte aligned)
//   2. necessary test on the 4 compass ring pixels at minTh (a 9-arc always covers 2 adjacent ones); survivors are
//      compacted IN ROW-MAJOR ORDER with ballot + popcount prefix
//   3. exact score only for the survivors (packed 16-bit min/max), written to an LDS score map
//   4. strict 8-neighbour NMS inside the ROI's detection area, iniTh -> minTh fallback when the cell has no
//      iniTh keypoint (vKeysCell.empty(), :812), ordered emission into the cell's fixed slot range.
// LDS: px[rh][P] + score[rh][P] bytes + survivor list (u16), sized by the host for the largest cell.
#ifndef DCS_FAST_CMPX_NOP
#define DCS_FAST_CMPX_NOP 4
#endif
#define DCS_STR2(x) #x
#define DCS_STR(x) DCS_STR2(x)
// ordered append of `yx` to a 16-bit list in LDS for the lanes with (int16) value > (int16) limit; end_addr = LDS byte address behind the
// last entry (scalar, advanced here). The compare is a v_cmpx: it writes EXEC itself, so the rank (v_mbcnt of exec), the store and the
// count run on it directly and ONE s_mov restores the wave (exec_full). Returns the compare's mask.
template <bool HW>      // HW = false: the same append through a plain ballot -- no EXEC games, no hand-placed wait states (the form the start-up probe falls back to)
__device__ __forceinline__ unsigned long long fast_append_gt(unsigned value, unsigned limit, unsigned yx, unsigned& end_addr, unsigned long long exec_full)
{
    if (!HW) {
        const bool pass = (short)(unsigned short)value > (short)(unsigned short)limit;
        const unsigned long long mb = __builtin_amdgcn_ballot_w64(pass);
        const unsigned at = end_addr + 2u * (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
        if (pass) asm volatile("ds_write_b16 %0, %1" : : "v"(at), "v"(yx) : "memory");
        end_addr += 2u * (unsigned)__popcll(mb);
        return mb;
    }
    unsigned long long m;
    unsigned at, cnt;
    asm volatile("v_cmpx_gt_i16_e64 %[m], %[v], %[lim]\n\t"
                 "s_nop " DCS_STR(DCS_FAST_CMPX_NOP) "\n\t"      // v_cmpx writes EXEC, the v_mbcnt behind it reads exec_lo as DATA: wait states the assembler does not insert
                 "v_mbcnt_lo_u32_b32 %[at], exec_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[at], exec_hi, %[at]\n\t"
                 "v_lshl_add_u32 %[at], %[at], 1, %[end]\n\t"
                 "ds_write_b16 %[at], %[yx]\n\t"
                 "s_bcnt1_i32_b64 %[cnt], exec\n\t"
                 "s_lshl1_add_u32 %[end], %[cnt], %[end]\n\t"
                 "s_mov_b64 exec, %[full]"
                 : [m] "=&s"(m), [at] "=&v"(at), [cnt] "=&s"(cnt), [end] "+s"(end_addr)
                 : [v] "v"(value), [lim] "v"(limit), [yx] "v"(yx), [full] "s"(exec_full)
                 : "memory", "scc");
    return m;
}

// EMIT (round 5): the cell also produces its part of the NEXT pyramid level from the ROI it holds in LDS -- cv::resize INTER_LINEAR with the
// tables and the per-pixel arithmetic of k_resize (resize_hpass / resize_emit), bit for bit. A lane owns one destination dword (4 pixels): its
// column taps are fixed, it walks the cell's destination rows in steps of eG rows (lane = column + enkx * row group). The 12 source bytes of a row
// come from LDS as three aligned dwords (the ROI row pitch P is a multiple of 4), the taps are picked out with the same v_perm selectors.
template <int P, bool D16Z, bool EMIT> // P: LDS row pitch in bytes (40 / 44: the ROI's own bytes; 48 / 64 / 128: aligned rows), compile-time so that the ring offsets are immediates; D16Z: see fast_score
__global__ __launch_bounds__(64) void k_fast_cells(LevelSet L, const CellDesc* __restrict__ cells, int n_cells, int ini_th, int min_th,
                                                   dcs_candidate* __restrict__ slots, size_t slots_per_image,
                                                   int32_t* __restrict__ cell_count, int map_bytes, int n_images, int sc_bytes, int dbg_stop,
                                                   int cell0, FastEmit em)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_px = smem;
    uint8_t* s_sc = smem + map_bytes;
    uint16_t* s_list = reinterpret_cast<uint16_t*>(smem + map_bytes + sc_bytes);
    // grid (8, n_cells, ceil(n_images / 8)): workgroups go to the 8 XCDs round-robin by linear id, so blockIdx.x IS the XCD and all
    // cells of an image run on one XCD -- neighbouring cells share their aprons' cache lines in ONE L2 (fabric fetches 1072 -> 219 MB
    // per 512-image launch) without any index arithmetic
    const int lane = threadIdx.x, cell = blockIdx.y + cell0, img = blockIdx.z * 8 + blockIdx.x;       // cell0: first cell of this launch
    if (img >= n_images) return;
    if (EMIT && (int)blockIdx.y >= em.n_cell_blocks) {
        // ---- the frame of the next level: one destination dword per lane, taps from global memory with k_resize's own column logic (12-byte
        // window of aligned dwords clamped into the row's storage, single-tap last column) and clamped source rows
        const int fb = (int)blockIdx.y - em.n_cell_blocks;
        int ri = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (fb >= em.fr[k].blk_begin) ri = k;                              // scalar
        const FrameRect fr = em.fr[ri];
        const int i = (fb - fr.blk_begin) * 64 + lane;
        if (i >= fr.count) return;
        const int q = (int)__umulhi((unsigned)i, fr.magic);                                           // i / x4_count (exact: i * x4_count < 2^32)
        constexpr int kFrameRows = 4;                                                                 // destination rows per lane: a frame wave costs its start-up and two load latencies whatever it does
        const int dy0 = fr.row_begin + kFrameRows * q, kx = fr.x4_begin + i - q * fr.x4_count;
        const int n_rows = min(kFrameRows, fr.row_end - dy0);
        const LevelView sv = L.lv[em.src_level];
        const uint8_t* S = sv.base + (size_t)img * sv.img_stride;
        const ResizeCol* cols = reinterpret_cast<const ResizeCol*>(em.cols);
        int sy[kFrameRows];
        unsigned ab[kFrameRows];
#pragma unroll
        for (int r = 0; r < kFrameRows; ++r) { const int dy = min(dy0 + r, fr.row_end - 1); sy[r] = em.rows[2 * dy]; ab[r] = (unsigned)em.rows[2 * dy + 1]; }
        ResizeTaps t;
        int base = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const ResizeCol cc = cols[min(4 * kx + k, em.dst.w - 1)];
            if (k == 0) base = min((int)cc.sx & ~3, sv.pitch - 12);
            const int o = cc.sx - base;                          // 0..11
            t.upper[k] = o > 6;
            const unsigned o8 = (unsigned)(t.upper[k] ? o - 4 : o);
            t.sel[k] = 0x0c000c00u | o8 | ((o8 < 7 ? o8 + 1 : 0x0cu) << 16);
            t.wgt[k] = (unsigned)(uint16_t)cc.a0 | ((unsigned)(uint16_t)cc.a1 << 16);
        }
        ResizeRaw ra[kFrameRows], rb[kFrameRows];
#pragma unroll
        for (int r = 0; r < kFrameRows; ++r) {
            const int sy0 = min(max(sy[r], 0), sv.h - 1), sy1 = min(max(sy[r] + 1, 0), sv.h - 1);
            ra[r] = resize_load(S + (size_t)sy0 * sv.pitch + base, true);
            rb[r] = resize_load(S + (size_t)sy1 * sv.pitch + base, true);
        }
        uint8_t* out = const_cast<uint8_t*>(em.dst.base) + (size_t)img * em.dst.img_stride + (size_t)dy0 * em.dst.pitch + 4 * kx;
#pragma unroll
        for (int r = 0; r < kFrameRows; ++r) {
            unsigned hA[4], hB[4];
            resize_hpass(ra[r], t, hA);
            resize_hpass(rb[r], t, hB);
            if (r < n_rows) resize_emit(out + (size_t)r * em.dst.pitch, hA, hB, ab[r] & 0xffffu, ab[r] >> 16);
        }
        return;
    }
    const CellDesc cd = cells[cell];
    const int rw = cd.rw, rh = cd.rh;
    if (rw < 7 || rh < 7) {
        if (lane == 0) cell_count[(size_t)img * n_cells + cell] = 0;
        return;
    }
    const LevelView lv = L.lv[cd.level];
    const uint8_t* img_base = lv.base + (size_t)img * lv.img_stride;
    // ROI columns start at byte `shift` of the LDS rows: global loads stay 16-byte (P = 64 / 128) or 8-byte (P = 48) aligned; the
    // exact classes (P = 40 / 44) read the ROI's own bytes with dword loads at its byte address (global loads need no alignment)
    constexpr bool kExact = P < 48;
    const int shift = kExact ? 0 : cd.x0 & (P == 48 ? 7 : 15);
    const int sc_pitch = rw - 4;                             // score map of THIS cell: detection width + 1-px rim, rows packed
    // ---- 0. (EMIT) this lane's column taps and the table entries of its first three destination rows: requested with the ROI
    constexpr int kEmitPre = 3;                              // rounds whose row entries are requested up front (a 30-px cell at scale 1.2: 25 rows, 9-10 groups)
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    int e_g = 0, e_kx = 0;
    bool e_lane = false;
    i32x4_t e_c01 = {0, 0, 0, 0}, e_c23 = {0, 0, 0, 0};
    i32x2_t e_row[kEmitPre] = {};
    if (EMIT && cd.enkx > 0) {
        e_g = (int)(((unsigned)lane * (unsigned)cd.emul) >> 16);                  // lane / enkx
        e_kx = lane - e_g * cd.enkx;
        e_lane = e_g < cd.eG;
        const i32x4_t* cp = reinterpret_cast<const i32x4_t*>(em.cols) + 2 * (cd.ekx0 + e_kx);   // 4 x {sx, 0, a0, a1}: 32 bytes
        e_c01 = cp[0]; e_c23 = cp[1];
#pragma unroll
        for (int r = 0; r < kEmitPre; ++r)
            e_row[r] = reinterpret_cast<const i32x2_t*>(em.rows)[cd.edy0 + min(e_g + r * cd.eG, cd.endy - 1)];
    }
    {   // ---- 1. ROI rows into LDS (pixel (x, y) of the ROI at s_px[y * P + shift + x]) and a zeroed score map
        const bool aligned = ((reinterpret_cast<uintptr_t>(img_base) | (uintptr_t)lv.pitch) & 3) == 0;
        const bool aligned16 = ((reinterpret_cast<uintptr_t>(img_base) | (uintptr_t)lv.pitch) & 15) == 0;
        const int ndw = (shift + rw + 3) >> 2;
        uint32_t* px_dw = reinterpret_cast<uint32_t*>(s_px);
        const int Pdw = P >> 2;
        const bool aligned8 = ((reinterpret_cast<uintptr_t>(img_base) | (uintptr_t)lv.pitch) & 7) == 0;
        if (kExact) {
            // P / 4 dwords per row, 64 / (P / 4) rows per wave pass; the dword that holds the ROI's last byte reads at most 3 bytes past it:
            // inside the image row (the ROI ends 13 pixels before the row does). All of a lane's loads are requested before its first store.
            constexpr int kDw = P / 4, kRows = 64 / kDw, kPasses = (44 + kRows - 1) / kRows;
            const uint8_t* src = img_base + (size_t)cd.y0 * lv.pitch + cd.x0;
            const int r0 = lane / kDw, c = lane - kDw * r0, nd = (rw + 3) >> 2;
            uint32_t v[kPasses];
            bool ok[kPasses];
#pragma unroll
            for (int i = 0; i < kPasses; ++i) {
                const int r = r0 + kRows * i;
                ok[i] = lane < kRows * kDw && c < nd && r < rh;
                if (ok[i]) __builtin_memcpy(&v[i], src + (size_t)r * lv.pitch + 4 * c, 4);
            }
#pragma unroll
            for (int i = 0; i < kPasses; ++i) if (ok[i]) px_dw[(r0 + kRows * i) * Pdw + c] = v[i];
            if (rh > kRows * kPasses && lane < kRows * kDw)    // taller cells (odd aspect ratios): the rest, row by row
                for (int r = r0 + kRows * kPasses; r < rh; r += kRows)
                    if (c < nd) { uint32_t w; __builtin_memcpy(&w, src + (size_t)r * lv.pitch + 4 * c, 4); px_dw[r * Pdw + c] = w; }
        } else if (P == 48 && aligned8) {                             // 10 rows x 6 x 8-byte columns per wave pass (rows of 48 bytes)
            // Round 4: buffer loads. The level image is a raw buffer of its valid bytes, a lane's address is ONE 32-bit offset (row r0 of the
            // ROI, 8-byte column c) that advances by 10 rows per pass with a scalar addend -- no 64-bit multiply-add per row -- and the loads need
            // no predicates: a lane without a column (lane >= 60, c beyond the ROI) starts out of range, a pass below the ROI reads image rows
            // that are simply not stored, and anything past the image's last byte returns 0 without touching memory. All five requests are
            // in flight before the first LDS store (the ROI has at most 44 rows, 38 in this size class); only the stores are predicated.
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img_base), 0, (lv.h - 1) * lv.pitch + lv.w, 0x00020000);
            const int r0 = lane / 6, c = lane - 6 * r0, nq = (shift + rw + 7) >> 3;
            const bool lane_ok = lane < 60 && c < nq;
            const unsigned voff = lane_ok ? (unsigned)((cd.y0 + r0) * lv.pitch + (cd.x0 - shift) + 8 * c) : 0x80000000u;
            const unsigned step = 10u * (unsigned)lv.pitch;
            const int rr = lane_ok ? r0 : 0x10000;                   // "row" of a lane that stores nothing
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            u32x2_t v[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(voff + (unsigned)i * step), 0, 0);
            uint8_t* const dst = s_px + r0 * P + 8 * c;
#pragma unroll
            for (int i = 0; i < 5; ++i) if (rr < rh - 10 * i) *reinterpret_cast<u32x2_t*>(dst + 10 * i * P) = v[i];
            if (rh > 50)                                             // taller cells (odd aspect ratios): the rest, row by row
                for (int r = r0 + 50; r < rh && lane_ok; r += 10)
                    *reinterpret_cast<u32x2_t*>(s_px + r * P + 8 * c) = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(voff + (unsigned)(r - r0) * (unsigned)lv.pitch), 0, 0);
        } else if (P != 48 && aligned16 && shift + rw <= 64 && (P & 15) == 0) {        // 16 rows x 4 x 16-byte columns per wave pass
            // the same form for the 64-byte class: three passes of 16 rows requested together (ROIs of up to 48 rows), buffer addressing
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img_base), 0, (lv.h - 1) * lv.pitch + lv.w, 0x00020000);
            const int r0 = lane >> 2, c = lane & 3, nq = (shift + rw + 15) >> 4;
            const bool lane_ok = c < nq;
            const unsigned voff = lane_ok ? (unsigned)((cd.y0 + r0) * lv.pitch + (cd.x0 - shift) + 16 * c) : 0x80000000u;
            const unsigned step = 16u * (unsigned)lv.pitch;
            const int rr = lane_ok ? r0 : 0x10000;
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t v[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + (unsigned)i * step), 0, 0);
            uint8_t* const dst = s_px + r0 * P + 16 * c;
#pragma unroll
            for (int i = 0; i < 3; ++i) if (rr < rh - 16 * i) *reinterpret_cast<u32x4_t*>(dst + 16 * i * P) = v[i];
            if (rh > 48)
                for (int r = r0 + 48; r < rh && lane_ok; r += 16)
                    *reinterpret_cast<u32x4_t*>(s_px + r * P + 16 * c) = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + (unsigned)(r - r0) * (unsigned)lv.pitch), 0, 0);
        } else if (aligned && ndw <= 16) {          // 4 rows x 16 dword columns per wave pass (no divisions)
            const uint8_t* src = img_base + (size_t)cd.y0 * lv.pitch + (cd.x0 - shift);
            const int c = lane & 15;
            for (int r = lane >> 4; r < rh; r += 4)
                if (c < ndw) px_dw[r * Pdw + c] = *reinterpret_cast<const uint32_t*>(src + (size_t)r * lv.pitch + 4 * c);
        } else if (aligned) {
            const uint8_t* src = img_base + (size_t)cd.y0 * lv.pitch + (cd.x0 - shift);
            for (int i = lane; i < rh * ndw; i += 64) {
                const int r = i / ndw, c = i - r * ndw;
                px_dw[r * Pdw + c] = *reinterpret_cast<const uint32_t*>(src + (size_t)r * lv.pitch + 4 * c);
            }
        } else {
            const uint8_t* src = img_base + (size_t)cd.y0 * lv.pitch + cd.x0;
            for (int i = lane; i < rh * rw; i += 64) { const int r = i / rw, c = i - r * rw; s_px[r * P + shift + c] = src[(size_t)r * lv.pitch + c]; }
        }
        uint4* sc_q = reinterpret_cast<uint4*>(s_sc);
        for (int i = lane; i < (sc_bytes >> 4); i += 64) sc_q[i] = uint4{0, 0, 0, 0};
    }
    __syncthreads();
    DCS_FAST_SECTION(1);
    if (EMIT && cd.enkx > 0) {   // ---- 1b. this cell's part of level `level + 1` (ComputePyramid :1120): see the comment above the kernel
        // The four pixels' taps span <= 7 source bytes (scale <= 1.33: the host checks it), so ONE 8-byte window per source row that starts
        // at the first tap holds them all, and pixel k's pair is one v_perm with a per-lane selector -- no choice between dword pairs as in k_resize.
        const int sx0 = (int)(short)e_c01.x;
        const int sxk[4] = {sx0, (int)(short)e_c01.z, (int)(short)e_c23.x, (int)(short)e_c23.z};
        const unsigned wg[4] = {(unsigned)e_c01.y, (unsigned)e_c01.w, (unsigned)e_c23.y, (unsigned)e_c23.w};      // a0 | a1 << 16
        unsigned sel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const unsigned o = (unsigned)(sxk[k] - sx0); sel[k] = 0x0c000c00u | o | ((o + 1) << 16); }   // o <= 6
        const int col0 = shift + sx0 - cd.x0;                                     // byte of the first tap inside an LDS row
        const unsigned mis = (unsigned)col0 & 3u;
#ifdef DCS_EMIT_UNALIGNED_LDS
        const unsigned col_addr = lds_addr(s_px) + (unsigned)(col0 - cd.y0 * P);
#else
        const uint8_t* const col_ptr = s_px + (col0 & ~3) - cd.y0 * P;
#endif
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(em.dst.base) + (size_t)img * em.dst.img_stride, 0,
                                                                              em.dst.h * em.dst.pitch, 0x00020000);
        const int G = cd.eG, rounds = cd.erounds;
        const unsigned dcol = (unsigned)(4 * (cd.ekx0 + e_kx) + cd.edy0 * em.dst.pitch);
        auto one = [&](int r, i32x2_t ent) {
            const int dyr = e_g + r * G;                                           // row inside the cell's rectangle
            const bool ok = e_lane && dyr < cd.endy;                               // (entries of lanes past the end were clamped to the last row: LDS reads stay in range)
            unsigned a0, a1, b0, b1;
#ifdef DCS_EMIT_UNALIGNED_LDS
            // measured: 299 us for level 0 against 184 with aligned reads -- a misaligned ds_read_b32 is not one access
            const unsigned a = col_addr + (unsigned)__mul24((int)ent.x, P);
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:%5\n\tds_read_b32 %3, %4 offset:%6\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(a), "n"(P), "n"(P + 4) : "memory");
#else
            // three aligned dwords per source row, realigned to the first tap's byte with two v_alignbyte (per-lane shift)
            const unsigned* pa = reinterpret_cast<const unsigned*>(col_ptr + __mul24((int)ent.x, P));
            const unsigned* pb = reinterpret_cast<const unsigned*>(reinterpret_cast<const uint8_t*>(pa) + P);
            const unsigned ra0 = pa[0], ra1 = pa[1], ra2 = pa[2], rb0 = pb[0], rb1 = pb[1], rb2 = pb[2];
            a0 = __builtin_amdgcn_alignbyte(ra1, ra0, mis); a1 = __builtin_amdgcn_alignbyte(ra2, ra1, mis);
            b0 = __builtin_amdgcn_alignbyte(rb1, rb0, mis); b1 = __builtin_amdgcn_alignbyte(rb2, rb1, mis);
#endif
            const unsigned wb0 = (unsigned)ent.y & 0xffffu, wb1 = (unsigned)ent.y >> 16;
            unsigned p0[4], p1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#ifdef DCS_EMIT_SKIP_H      // timing-only side build (WRONG pixels): the horizontal pass costs nothing -- an upper bound on what moving it to the matrix cores could save (NOTES R6.6)
                const unsigned hA = (k & 1 ? a1 : a0) & 0x7fffu, hB = (k & 1 ? b1 : b0) & 0x7fffu;
#else
                const unsigned hA = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, __builtin_amdgcn_perm(a1, a0, sel[k])), __builtin_bit_cast(ushort2_t, wg[k]), 0u, false) >> 4;
                const unsigned hB = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, __builtin_amdgcn_perm(b1, b0, sel[k])), __builtin_bit_cast(ushort2_t, wg[k]), 0u, false) >> 4;
#endif
                p0[k] = __umul24(wb0, hA) + 0x20000u;                              // the "+ 2" of the rounding rides on the first product's high half
                p1[k] = __umul24(wb1, hB);
            }
            // ((p0 >> 16) + (p1 >> 16) + 2) >> 2 for four pixels: the high halves are added by SDWA straight into the halves of two registers, one
            // 32-bit shift each (the bits that leak across the halves land above the byte that is kept), one v_perm packs the four bytes
            unsigned s01, s23;
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(s01) : "v"(p0[0]), "v"(p1[0]));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(s01) : "v"(p0[1]), "v"(p1[1]));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(s23) : "v"(p0[2]), "v"(p1[2]));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(s23) : "v"(p0[3]), "v"(p1[3]));
            const unsigned packed = __builtin_amdgcn_perm(s23 >> 2, s01 >> 2, 0x06040200u);
            // a lane without a row stores out of range: dropped by the buffer unit, no branch
            __builtin_amdgcn_raw_buffer_store_b32(packed, drs, ok ? (int)(dcol + (unsigned)__mul24(dyr, em.dst.pitch)) : (int)0x80000000u, 0, 0);
        };
#pragma unroll
        for (int r = 0; r < kEmitPre; ++r) if (r < rounds) one(r, e_row[r]);       // wave-uniform
        for (int r = kEmitPre; r < rounds; ++r)
            one(r, reinterpret_cast<const i32x2_t*>(em.rows)[cd.edy0 + min(e_g + r * G, cd.endy - 1)]);
    }
    DCS_FAST_SECTION(11);
    const uint8_t* px = s_px + shift;
    // score map: only the detection area and its 1-px rim exist, pixel (x, y) of the ROI at s_sc[(y - 2) * sc_pitch + (x - 2)] (the
    // smaller map buys LDS room for two more resident waves per SIMD, and FAST loses 13 % when it loses 1.25)
    uint8_t* sc = s_sc - 2 * sc_pitch - 2;
    const int dw = rw - 6, dh = rh - 6, ndet = dw * dh;
    // Steps 2-4 run for the reference's two thresholds in its own order (ORBextractor.cc:806-819): iniThFAST first; only a cell
    // that yields NO keypoint there is redone at minThFAST. A pass at threshold T only needs the pixels that pass the compass
    // test at T: every pixel with score >= T does, so the score map holds exactly what the non-maximum suppression at T compares
    // (a neighbour whose score is missing has score < T <= s(p) and would lose anyway). 91 % of the cells of the benchmark
    // scene stop after the first pass, which scores a third fewer pixels than the minThFAST pass.
    int n_list = 0, base = 0;                      // survivors of the compass test; keypoints emitted
    dcs_candidate* const out = slots + (size_t)img * slots_per_image + cd.slot_base;
    // position of this lane among the set bits of a ballot: v_mbcnt_lo + v_mbcnt_hi
    auto rank_in = [](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
    for (int pass = 0; pass < 2; ++pass) {
        const int th = pass == 0 ? ini_th : min_th;
        if (pass == 1 && min_th == ini_th) break;
        // ---- 2. compass test + ordered compaction; survivors are stored as (y << 8 | x), ROI coordinates
        // a 9-arc covers two ADJACENT compass pixels: (b0|b8)&(b4|b12) for "brighter", same for "darker". The two compares are
        // balloted one by one (ballot of a plain compare IS the compare's SGPR mask; a ballot of their OR would be rebuilt through
        // v_cndmask + v_cmp) and combined with scalar 64-bit logic.
        // b = the pixel 3 up and 3 left of the centre: every LDS offset is a non-negative immediate
        // 16-bit VOP2 min / max / sub issue at twice the rate of their 32-bit forms on gfx950 (profiles/r03_valu_rate_probe.txt), and the
        // two polarities fold into ONE compare: brighter <=> e - v > th, darker <=> v - f > th, so pass <=> max(e - v, v - f) > th.
        // max(e - v, v - f) of the pixel whose 7 x 7 neighbourhood starts at b: the pixel passes when it exceeds th
        auto compass_margin = [&](const uint8_t* b) -> uint16_t {
            const uint16_t v = b[3 * P + 3];
            const uint16_t r0 = b[6 * P + 3], r4 = b[3 * P + 6], r8 = b[3], r12 = b[3 * P];
            const uint16_t e = min(max(r0, r8), max(r4, r12)), f = max(min(r0, r8), min(r4, r12));
            const int16_t up = (int16_t)(uint16_t)(e - v), down = (int16_t)(uint16_t)(v - f);
            return (uint16_t)max(up, down);
        };
        auto compass = [&](const uint8_t* b) -> bool { return (int16_t)compass_margin(b) > (int16_t)th; };
        n_list = 0;
        const unsigned list_addr = lds_addr(s_list);
        unsigned long long exec_full;                                                           // the wave's execution mask in this region (restored after every append)
        asm volatile("s_mov_b64 %0, exec" : "=s"(exec_full));
        if (dw <= 32) {                                  // two detection rows per round: lanes 0-31 row y, lanes 32-63 row y + 1
            // Round 4: the CU's SCALAR unit is the kernel's second wall (profiles/r04_valu_rate_probe.txt: a two-operand scalar instruction
            // costs 2.4 ticks of SIMD time, 1.8 x a slow-class vector one; a v_mad + s_add pair runs at the scalar rate), and this loop spent
            // six scalar instructions per round on the append. Now: the columns outside the detection area are switched off through a per-lane
            // threshold (no mask to AND), and the compare is a v_cmpx -- it writes the result to EXEC itself, so the rank (v_mbcnt of exec), the
            // store and the count run on it directly and ONE s_mov restores the full wave: s_bcnt1 + s_lshl1_add + s_mov instead of
            // s_and + s_bcnt1 + s_lshl1_add + s_and_saveexec + s_mov.
            const uint16_t th_lane = (lane & 31) < dw ? (uint16_t)th : (uint16_t)0x7fff;
            const uint8_t* b = px + (lane >> 5) * P + (lane & 31);
            unsigned yx = (unsigned)(((3 + (lane >> 5)) << 8) | (3 + (lane & 31)));
            unsigned long long m = 0;
            unsigned end_addr = list_addr;                                                       // scalar, behind the last entry
            // Every lane evaluates the test (a lane outside the detection area reads the score map at worst: still inside
            // the LDS allocation). No branch: the append is one LDS store under exec = the lanes that pass.
            // When dh is odd the upper half-wave's last row lies below the detection area: whatever it appends comes after
            // every valid entry and is cut off by the count below.
            auto round = [&](const uint8_t* bb, unsigned yxr) { m = fast_append_gt<D16Z>(compass_margin(bb), th_lane, yxr, end_addr, exec_full); };
            // four rounds per trip: the later ones' LDS reads are immediate offsets of the first one's address, and the loop's
            // scalar bookkeeping is paid once per eight rows
            int yy = 0;
            for (; yy + 6 < dh; yy += 8, yx += 0x800u, b += 8 * P) {                          // wave-uniform trip counts
                round(b, yx); round(b + 2 * P, yx + 0x200u); round(b + 4 * P, yx + 0x400u); round(b + 6 * P, yx + 0x600u);
            }
            for (; yy < dh; yy += 2, yx += 0x200u, b += 2 * P) round(b, yx);
            n_list = (int)((end_addr - list_addr) >> 1);
            if (dh & 1) n_list -= __popc((unsigned)(m >> 32));
        } else {
            int y = 3 + lane / dw, x = 3 + lane % dw;            // pixel p = p0 + lane, advanced by 64 per round without dividing
            const int step_y = 64 / dw, step_x = 64 % dw;
            for (int p0 = 0; p0 < ndet; p0 += 64) {
                const bool pass_c = compass(px + (y - 3) * P + (x - 3));
                const bool in_range = p0 + lane < ndet;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(pass_c) & __builtin_amdgcn_ballot_w64(in_range);
                if (pass_c & in_range) s_list[n_list + rank_in(m)] = (uint16_t)((y << 8) | x);
                n_list += __popcll(m);
                y += step_y; x += step_x;
                if (x >= dw + 3) { x -= dw; ++y; }
            }
        }
        __syncthreads();
        // ---- 2b. a second necessary test on the SURVIVORS only (round 4): a 9-arc also covers two adjacent DIAGONAL ring pixels (2, 6, 10,
        // 14), so the same margin on those four must exceed th as well. It keeps 65 % of the compass survivors of the benchmark scene
        // (scratch/fast_stats3.py; 26 % hold a real arc): the list is compacted in place before the scoring, whose rounds of 64 cost six times
        // a round of this test (mean scoring rounds per cell 2.3 -> 1.6). Every pixel that scores >= th passes both tests, so the score map
        // still holds everything the non-maximum suppression at th compares, and the minThFAST pass's list is still a superset of this one.
#ifndef DCS_FAST_NO_DIAG
        if (__builtin_amdgcn_readfirstlane(n_list) > 64) {      // a list that already fits ONE scoring round gains nothing from being shorter
            const int n_in = __builtin_amdgcn_readfirstlane(n_list);
            unsigned end2 = list_addr;
            for (int i0 = 0; i0 < n_in; i0 += 64) {              // wave-uniform
                const int i = i0 + lane;
                const unsigned lim = i < n_in ? (unsigned)th : 0x7fffu;           // a lane past the end reads the last entry and appends nothing
                const unsigned yx = s_list[min(i, n_in - 1)];
                const uint8_t* b = px + __mul24((int)(yx >> 8) - 3, P) + ((int)(yx & 255u) - 3);
                const uint16_t v = b[3 * P + 3];
                const uint16_t r2 = b[5 * P + 5], r6 = b[P + 5], r10 = b[P + 1], r14 = b[5 * P + 1];
                const uint16_t e = min(max(r2, r10), max(r6, r14)), f = max(min(r2, r10), min(r6, r14));
                const int16_t up = (int16_t)(uint16_t)(e - v), down = (int16_t)(uint16_t)(v - f);
                // in place: the write position is never ahead of an entry not read yet (a wave's LDS operations complete in order)
                (void)fast_append_gt<D16Z>((unsigned)(uint16_t)max(up, down), lim, yx, end2, exec_full);
            }
            n_list = (int)((end2 - list_addr) >> 1);
        }
        __syncthreads();
#endif
        DCS_FAST_SECTION(2);
        // ---- 3. exact scores of the survivors. Only the polarity (or, rarely, both) that passed the compass test at this
        // threshold is evaluated: the other one has no 9-arc at th, so its arc minimum is < th + 1 and cannot be the maximum of a
        // pixel that scores >= th; a pixel that scores < th may be stored with a smaller value than cv::FAST's -- it is never
        // emitted and loses every comparison against a neighbour that is (s(q) >= th > both values). A second pass rewrites the
        // first pass's entries (its list is a superset), so every stored score >= min_th is exact then.
        // The list shrinks once more on the way: only the pixels that score >= th go on to the suppression (every entry of a round is read before
        // the round's survivors are written back, to positions that were read already).
        {
#ifdef DCS_FAST_ONE_ROUND    // timing-only side build (WRONG candidates): at most one scoring round per cell -- more than any packing of survivor lists across cells could save (NOTES R6.6)
            const int n_in = min(__builtin_amdgcn_readfirstlane(n_list), 64);
#else
            const int n_in = __builtin_amdgcn_readfirstlane(n_list);
#endif
            unsigned end3 = list_addr;
            for (int i0 = 0; i0 < n_in; i0 += 64) {              // wave-uniform
                const int i = i0 + lane;
                const bool valid = i < n_in;
                const unsigned yx = s_list[min(i, n_in - 1)];
                const int y = (int)(yx >> 8), x = (int)(yx & 255u);
                int o = (y - 3) * P + (x - 3);
                asm("" : "+v"(o));                          // opaque: keeps the 17 ring offsets non-negative immediates of ONE base address
                const int s = fast_score<P, D16Z>(px + o, th);
                if (valid) sc[__mul24(y, sc_pitch) + x] = (uint8_t)s;
                (void)fast_append_gt<D16Z>((unsigned)s, valid ? (unsigned)(th - 1) : 0x7fffu, yx, end3, exec_full);       // s >= th
            }
            n_list = (int)((end3 - list_addr) >> 1);
        }
        __syncthreads();
        DCS_FAST_SECTION(3);
        // ---- 4. NMS + ordered emission: keep(T) = { p : s(p) >= T and s(p) > s(q) for the 8 neighbours q } (neighbours below T
        // lose anyway). Keypoints go straight to the cell's slot range: a pass that finds none has emitted nothing, and only such
        // a pass is followed by another one (vKeysCell.empty(), :812).
        for (int i0 = 0; i0 < n_list; i0 += 64) {               // wave-uniform
            const int i = i0 + lane;
            bool keep = false;
            int y = 0, x = 0, s = 0;
            if (i < n_list) {
                const int yx = s_list[i];
                y = yx >> 8; x = yx & 255;
                const uint8_t* c = sc + __mul24(y, sc_pitch) + x;
                s = c[0];                                          // >= th: the scoring loop kept only those
                {
                    const uint8_t* cu = c - sc_pitch; const uint8_t* cd2 = c + sc_pitch;
                    // seven two-input 16-bit maxima: the compiler's own choice, v_max3_u16, occupies the SIMD 3.3 x as long as a v_max_u16
                    // (profiles/r04_valu_rate_probe.txt: 2.47 vs 0.75 ticks), so the pairs are pinned with asm
                    auto mx = [](unsigned a, unsigned b) { unsigned r; asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
                    const unsigned nb = mx(mx(mx(c[-1], c[1]), mx(cu[-1], cu[0])), mx(mx(cu[1], cd2[-1]), mx(cd2[0], cd2[1])));
                    keep = (unsigned)s > nb;
                }
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (m) {                                             // wave-uniform
                const int off = base + rank_in(m);
                if (keep && off < cd.cap) {
                    dcs_candidate o;
                    o.x = (int16_t)(x + cd.ox); o.y = (int16_t)(y + cd.oy); o.score = (uint8_t)s;
                    out[off] = o;
                }
                base += __popcll(m);
            }
        }
        DCS_FAST_SECTION(4);
        if (base) break;                               // vKeysCell not empty: no minThFAST retry (:812)
        __syncthreads();                               // the list is rebuilt by the next pass
    }
    if (lane == 0) cell_count[(size_t)img * n_cells + cell] = min(base, (int)cd.cap);
}

// ---- start-up probe of the two hardware behaviours k_fast_cells<P, true, .> relies on and no manual guarantees (round-4 verdict, item 6):
//  (1) ds_read_u8_d16_hi writes the byte to bits 16-23 AND clears the rest of the register (observed on gfx950 with SRAM ECC; the ISA text
//      only promises the high half);  (2) the v_cmpx / s_nop / v_mbcnt(exec) sequence of fast_append_gt<true> ranks the passing lanes
//      correctly with the shipped wait states. One wave runs the EXACT instruction forms on known data; the host compares with what the
//      plain forms must give, and a device that answers differently runs k_fast_cells<P, false, .> (byte loads, ballot append) -- bit-exact
//      either way, ~8 % slower.
__global__ __launch_bounds__(64) void k_fast_hw_probe(unsigned* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint8_t sm[256];
    __shared__ __attribute__((aligned(16))) uint16_t list[192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sm[i] = (uint8_t)(i * 37 + 11);
    for (int i = lane; i < 192; i += 64) list[i] = 0xffff;
    __syncthreads();
    unsigned r = 0xdeadbeefu;                                  // garbage in BOTH halves before the load
    asm volatile("ds_read_u8_d16_hi %0, %1 offset:5\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(lds_addr(sm) + (unsigned)lane) : "memory");
    out[lane] = r;
    unsigned end_addr = lds_addr(list);
    unsigned long long exec_full;
    asm volatile("s_mov_b64 %0, exec" : "=s"(exec_full));
    const unsigned long long m1 = fast_append_gt<true>((unsigned)(uint16_t)(short)(lane * 3 - 40), 17u, 0x100u + (unsigned)lane, end_addr, exec_full);
    const unsigned long long m2 = fast_append_gt<true>((unsigned)(uint16_t)(short)((lane * 7) % 23 - 5), (lane & 1) ? 0x7fffu : 3u, 0x200u + (unsigned)lane, end_addr, exec_full);
    __syncthreads();
    for (int i = lane; i < 128; i += 64) out[64 + i] = list[i];
    if (lane == 0) {
        out[192] = (end_addr - lds_addr(list)) >> 1;
        out[193] = (unsigned)m1; out[194] = (unsigned)(m1 >> 32); out[195] = (unsigned)m2; out[196] = (unsigned)(m2 >> 32);
    }
}
int fast_hw_probe(bool* ok, char* why, size_t why_len)
{
    *ok = false;
    unsigned* d = nullptr;
    unsigned h[200] = {0};
    DCS_HIP(hipMalloc(&d, sizeof(h)));
    hipLaunchKernelGGL(k_fast_hw_probe, dim3(1), dim3(64), 0, nullptr, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { set_error("fast_hw_probe: %s", hipGetErrorString(e)); return DCS_ERR_HIP; }
    // what the plain forms give
    for (int lane = 0; lane < 64; ++lane) {
        const unsigned want = (unsigned)(uint8_t)((lane + 5) * 37 + 11) << 16;
        if (h[lane] != want) { snprintf(why, why_len, "ds_read_u8_d16_hi: lane %d returned 0x%08x, the zeroing form gives 0x%08x", lane, h[lane], want); return DCS_OK; }
    }
    unsigned want_list[128];
    int n = 0;
    unsigned long long w1 = 0, w2 = 0;
    for (int lane = 0; lane < 64; ++lane) if ((short)(lane * 3 - 40) > 17) { want_list[n++] = 0x100u + (unsigned)lane; w1 |= 1ull << lane; }
    for (int lane = 0; lane < 64; ++lane) if ((short)((lane * 7) % 23 - 5) > (short)((lane & 1) ? 0x7fff : 3)) { want_list[n++] = 0x200u + (unsigned)lane; w2 |= 1ull << lane; }
    if ((int)h[192] != n) { snprintf(why, why_len, "v_cmpx append: %u entries, expected %d", h[192], n); return DCS_OK; }
    if ((((unsigned long long)h[194] << 32) | h[193]) != w1 || (((unsigned long long)h[196] << 32) | h[195]) != w2) { snprintf(why, why_len, "v_cmpx append: wrong compare mask"); return DCS_OK; }
    for (int i = 0; i < n; ++i) if (h[64 + i] != want_list[i]) { snprintf(why, why_len, "v_cmpx append: entry %d is 0x%x, expected 0x%x", i, h[64 + i], want_list[i]); return DCS_OK; }
    *ok = true;
    return DCS_OK;
}

// LDS of one cell's workgroup for a launch with the given footprint (the host groups levels by this figure, orb_extract.cpp)
static int fast_pitch(int max_rw)                                // exact rows, or shift (<= 7 / 15) + row
{
    // the exact classes (the ROI's own bytes, dword loads at its byte address): 44-byte rows for ROIs 41..44 wide instead of 64 (level 7 of
    // the 640 x 480 pyramid: 28 instead of 24 cells per CU, -3 us per 512 images); 40-byte rows for everything narrower would lift levels
    // 4-6 into the 32-per-CU class of levels 0-3 (one launch less), but seven dword loads per lane instead of five 8-byte ones cost more
    // than that gives back: FAST 530 vs 503 us. DCS_FAST_EXACT: bit 0 = P 40, bit 1 = P 44 (default 2).
    const int exact = (int)opt(OPT_FAST_EXACT);
    if ((exact & 1) && max_rw <= 40) return 40;
    if ((exact & 2) && max_rw > 40 && max_rw <= 44) return 44;
    return max_rw + 7 <= 48 ? 48 : max_rw + 15 <= 64 ? 64 : 128;
}
void fast_footprint_add(FastFootprint& f, int rw, int rh)
{
    f.max_rw = std::max(f.max_rw, rw); f.max_rh = std::max(f.max_rh, rh);
    f.list_entries = std::max(f.list_entries, (rw - 6) * (rh - 6) + 32);          // every detection pixel + 32 entries of an odd last row
    f.sc_bytes = std::max(f.sc_bytes, (rw - 4) * (rh - 4));                       // detection area + 1-px rim, rows packed at the CELL's width
}
int fast_cells_lds_bytes(const FastFootprint& f)
{
    const int map_bytes = ((f.max_rh * fast_pitch(f.max_rw)) + 15) & ~15;
    return map_bytes + ((f.sc_bytes + 15) & ~15) + ((2 * f.list_entries + 15) & ~15);
}

int launch_fast_cells(const LevelSet& levels, const CellDesc* d_cells, int n_cells, int n_images,
                      int ini_th, int min_th, dcs_candidate* d_slots, size_t slots_per_image,
                      int32_t* d_cell_count, const FastFootprint& fp, hipStream_t s, int cell0, int n_launch, const FastEmit* emit, bool fast_hw)
{
    if (n_launch < 0) n_launch = n_cells - cell0;             // cells [cell0, cell0 + n_launch) of the n_cells of the pyramid
    if (n_launch <= 0) return DCS_OK;
    const int P = fast_pitch(fp.max_rw);
    const int map_bytes = ((fp.max_rh * P) + 15) & ~15;
    const int sc_bytes = (fp.sc_bytes + 15) & ~15;
    const size_t shmem = (size_t)fast_cells_lds_bytes(fp);
#ifdef DCS_FAST_SECTIONS
    const char* stop_env = getenv("DCS_FAST_STOP");          // -DDCS_FAST_SECTIONS side builds only (scratch/fast_sections.sh): not an option of the library
    const int dbg_stop = stop_env ? atoi(stop_env) : 0;
#else
    const int dbg_stop = 0;
#endif
    FastEmit em = emit ? *emit : FastEmit{};
    em.n_cell_blocks = n_launch;
    const dim3 grid(8, n_launch + (emit ? em.n_frame_blocks : 0), (n_images + 7) / 8);
    // D16Z = fast_hw: BOTH hardware-specific forms of the kernel (the zeroing ds_read_u8_d16_hi of the score's ring loads, the v_cmpx append
    // with its hand-placed wait states) or neither -- decided per device by fast_hw_probe() when the handle is created, not by an architecture string
    const bool d16z = fast_hw;
#define DCS_FAST_LAUNCH2(PP, ZZ, EE) hipLaunchKernelGGL((k_fast_cells<PP, ZZ, EE>), grid, dim3(64), shmem, s, levels, d_cells, n_cells, ini_th, min_th, d_slots, slots_per_image, \
                                               d_cell_count, map_bytes, n_images, sc_bytes, dbg_stop, cell0, em)
#define DCS_FAST_LAUNCH(PP) do { if (d16z) { if (emit) DCS_FAST_LAUNCH2(PP, true, true); else DCS_FAST_LAUNCH2(PP, true, false); } \
                                 else { if (emit) DCS_FAST_LAUNCH2(PP, false, true); else DCS_FAST_LAUNCH2(PP, false, false); } } while (0)
    if (P == 40) DCS_FAST_LAUNCH(40);
    else if (P == 44) DCS_FAST_LAUNCH(44);
    else if (P == 48) DCS_FAST_LAUNCH(48);
    else if (P == 64) DCS_FAST_LAUNCH(64);
    else DCS_FAST_LAUNCH(128);
#undef DCS_FAST_LAUNCH
#undef DCS_FAST_LAUNCH2
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// ------------------------------------------------------------------------------------- compaction
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* s_tmp /*[5]*/, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int c = s_tmp[w]; if (w < wave) off += c; tot += c; }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

// grid (nlevels, n_images): exclusive offsets of the cells of one level + the level total
__global__ __launch_bounds__(256) void k_level_scan(const int32_t* __restrict__ level_cell_begin, int nlevels, int n_cells,
                                                    const int32_t* __restrict__ cell_count, int32_t* __restrict__ cell_off,
                                                    int32_t* __restrict__ lvl_total)
{
    __shared__ int s_tmp[4];
    const int l = blockIdx.x, img = blockIdx.y;
    const int cb = level_cell_begin[l], ce = level_cell_begin[l + 1];
    int running = 0;
    for (int c0 = cb; c0 < ce; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const int v = c < ce ? cell_count[(size_t)img * n_cells + c] : 0;
        int tot;
        const int ex = block_exclusive_scan_256(v, s_tmp, &tot);
        if (c < ce) cell_off[(size_t)img * n_cells + c] = running + ex;
        running += tot;
    }
    if (threadIdx.x == 0) lvl_total[img * nlevels + l] = running;
}

// single block: exclusive scan of the (image, level) totals -> lvl_off[n_all + 1]. Every thread scans a run of consecutive
// entries in registers, so 4096 totals (512 images x 8 levels) take ONE block scan instead of sixteen
__global__ __launch_bounds__(256) void k_lvl_offsets(const int32_t* __restrict__ lvl_total, int n_all, int32_t* __restrict__ lvl_off)
{
    __shared__ int s_tmp[4];
    constexpr int kRun = 16;
    int running = 0;
    for (int c0 = 0; c0 < n_all; c0 += 256 * kRun) {
        const int b = c0 + threadIdx.x * kRun;
        int v[kRun], sum = 0;
#pragma unroll
        for (int k = 0; k < kRun; ++k) { v[k] = b + k < n_all ? lvl_total[b + k] : 0; sum += v[k]; }
        int tot;
        int ex = running + block_exclusive_scan_256(sum, s_tmp, &tot);
#pragma unroll
        for (int k = 0; k < kRun; ++k) { if (b + k < n_all) lvl_off[b + k] = ex; ex += v[k]; }
        running += tot;
    }
    if (threadIdx.x == 0) lvl_off[n_all] = running;
}

// 16 lanes per (cell, image) -- a cell rarely holds more than a dozen candidates: copy the cell's slots to their place in
// the dense, emission-ordered array. The kernel is the latency of its four dependent index loads per cell, so every 16-lane group
// takes kGatherCells cells and issues their loads together (one cell per group: 104 k waves of a few microseconds each, 35 us per
// 512 images).
constexpr int kGatherCells = 4;
__global__ __launch_bounds__(256) void k_gather(const CellDesc* __restrict__ cells, int nlevels, int n_cells,
                                                const dcs_candidate* __restrict__ slots, size_t slots_per_image,
                                                const int32_t* __restrict__ cell_count, const int32_t* __restrict__ cell_off,
                                                const int32_t* __restrict__ lvl_off, dcs_candidate* __restrict__ dense, size_t dense_cap)
{
    const int c0 = blockIdx.x * (16 * kGatherCells) + (threadIdx.x >> 4), img = blockIdx.y, sub = threadIdx.x & 15;
    int cc[kGatherCells], n[kGatherCells], slot_base[kGatherCells], off[kGatherCells], lvl[kGatherCells];
#pragma unroll
    for (int u = 0; u < kGatherCells; ++u) {
        cc[u] = min(c0 + 16 * u, n_cells - 1);
        n[u] = c0 + 16 * u < n_cells ? cell_count[(size_t)img * n_cells + cc[u]] : 0;
        slot_base[u] = cells[cc[u]].slot_base; lvl[u] = cells[cc[u]].level;
        off[u] = cell_off[(size_t)img * n_cells + cc[u]];
    }
    int lo[kGatherCells];
#pragma unroll
    for (int u = 0; u < kGatherCells; ++u) lo[u] = lvl_off[img * nlevels + lvl[u]];
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(dense);
#pragma unroll
    for (int u = 0; u < kGatherCells; ++u) {
        const size_t dst0 = (size_t)lo[u] + off[u];
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(slots + (size_t)img * slots_per_image + slot_base[u]);
        for (int k = sub; k < n[u]; k += 16)
            if (dst0 + k < dense_cap) dst[dst0 + k] = src[k];
    }
}

// Small batches (what Frame::ExtractORB hands over: one or two images per call, src/Frame.cc:141-149): the three launches above are 4 us of
// work and 15 us of launch latency. ONE launch does all of it -- the dense, emission-ordered array is laid out (image, level, cell), which
// is exactly the order of the (image, cell) entries, so ONE exclusive scan over all n_images * n_cells counts gives every cell its place;
// the per-level offsets, totals and the (image, level) offsets the quadtree reads are differences of that scan. Same bytes in every
// output array as k_level_scan + k_lvl_offsets + k_gather.
constexpr int kCompactSmallThreads = 1024;
constexpr int kCompactSmallRun = 16;                                       // entries per thread: up to 16 384 (image, cell) entries
__global__ __launch_bounds__(kCompactSmallThreads) void k_compact_small(const CellDesc* __restrict__ cells, const int32_t* __restrict__ level_cell_begin, int nlevels,
                                                                        int n_images, int n_cells, const dcs_candidate* __restrict__ slots, size_t slots_per_image,
                                                                        const int32_t* __restrict__ cell_count, int32_t* __restrict__ cell_off,
                                                                        int32_t* __restrict__ lvl_total, int32_t* __restrict__ lvl_off,
                                                                        dcs_candidate* __restrict__ dense, size_t dense_cap)
{
    __shared__ int s_wave[kCompactSmallThreads / 64];
    extern __shared__ int s_pos[];                                         // [n_entries + 1]: exclusive prefix of every (image, cell) entry
    const int n_entries = n_images * n_cells, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = t * kCompactSmallRun;
    int v[kCompactSmallRun], sum = 0;
#pragma unroll
    for (int k = 0; k < kCompactSmallRun; ++k) { v[k] = b + k < n_entries ? cell_count[b + k] : 0; sum += v[k]; }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kCompactSmallThreads / 64; ++w) { const int c = s_wave[w]; if (w < wave) off += c; total += c; }
    int ex = off + inc - sum;
#pragma unroll
    for (int k = 0; k < kCompactSmallRun; ++k) { if (b + k < n_entries) s_pos[b + k] = ex; ex += v[k]; }
    if (t == 0) s_pos[n_entries] = total;
    __syncthreads();
    // (image, level) offsets and totals, per-level cell offsets: written by workgroup 0 (every workgroup repeats the scan -- 16 loads per
    // thread -- and gathers its own share of the entries: the gather is two dependent loads per entry, one workgroup alone would walk them
    // 64 at a time)
    const int n_all = n_images * nlevels;
    if (blockIdx.x == 0) {
    for (int i = t; i <= n_all; i += kCompactSmallThreads) {
        if (i == n_all) { lvl_off[n_all] = total; break; }
        const int img = i / nlevels, l = i - img * nlevels;
        const int first = s_pos[img * n_cells + level_cell_begin[l]], end = s_pos[img * n_cells + level_cell_begin[l + 1]];
        lvl_off[i] = first;
        lvl_total[i] = end - first;
    }
    for (int i = t; i < n_entries; i += kCompactSmallThreads) {
        const int img = i / n_cells, c = i - img * n_cells;
        cell_off[i] = s_pos[i] - s_pos[img * n_cells + level_cell_begin[cells[c].level]];
    }
    }
    // gather: 16 lanes per entry, like k_gather
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(dense);
    const int sub = t & 15;
    for (int i = blockIdx.x * (kCompactSmallThreads / 16) + (t >> 4); i < n_entries; i += gridDim.x * (kCompactSmallThreads / 16)) {
        const int n = s_pos[i + 1] - s_pos[i];
        if (n == 0) continue;
        const int img = i / n_cells, c = i - img * n_cells;
        const size_t dst0 = (size_t)s_pos[i];
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(slots + (size_t)img * slots_per_image + cells[c].slot_base);
        for (int k = sub; k < n; k += 16)
            if (dst0 + k < dense_cap) dst[dst0 + k] = src[k];
    }
}

int launch_compact(const CellDesc* d_cells, const int32_t* d_level_cell_begin, int nlevels, int n_images, int n_cells,
                   const dcs_candidate* d_slots, size_t slots_per_image, const int32_t* d_cell_count,
                   int32_t* d_cell_off, int32_t* d_lvl_total, int32_t* d_lvl_off, dcs_candidate* d_dense,
                   size_t dense_cap, hipStream_t s)
{
    const bool small_on = true;
    const long long n_entries = (long long)n_images * n_cells;
    if (small_on && n_cells > 0 && n_entries <= kCompactSmallThreads * kCompactSmallRun && (n_entries + 1) * sizeof(int) <= 64 * 1024) {
        const int n_wg = (int)std::min<long long>(64, (n_entries + 127) / 128);
        hipLaunchKernelGGL(k_compact_small, dim3(n_wg), dim3(kCompactSmallThreads), (size_t)(n_entries + 1) * sizeof(int), s, d_cells, d_level_cell_begin, nlevels,
                           n_images, n_cells, d_slots, slots_per_image, d_cell_count, d_cell_off, d_lvl_total, d_lvl_off, d_dense, dense_cap);
        DCS_CHECK_LAUNCH();
        return DCS_OK;
    }
    hipLaunchKernelGGL(k_level_scan, dim3(nlevels, n_images), dim3(256), 0, s, d_level_cell_begin, nlevels, n_cells,
                       d_cell_count, d_cell_off, d_lvl_total);
    DCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_lvl_offsets, dim3(1), dim3(256), 0, s, d_lvl_total, n_images * nlevels, d_lvl_off);
    DCS_CHECK_LAUNCH();
    if (n_cells) {
        hipLaunchKernelGGL(k_gather, dim3((n_cells + 16 * kGatherCells - 1) / (16 * kGatherCells), n_images), dim3(256), 0, s, d_cells, nlevels, n_cells, d_slots, slots_per_image,
                           d_cell_count, d_cell_off, d_lvl_off, d_dense, dense_cap);
        DCS_CHECK_LAUNCH();
    }
    return DCS_OK;
}

// ------------------------------------------------------------------------------------- Gaussian blur
// Separable 7-tap, integer kernel {18,34,49,55,49,34,18} (getGaussianKernel(7,2)*256 rounded), rows then columns in
// int32, out = sat_u8((acc + 32768) >> 16), BORDER_REFLECT_101.
// Streaming design, no LDS: a thread owns 4 horizontally adjacent output pixels (one dword) of a 16-row strip. For
// every input row it loads the 12 source bytes as 3 aligned dwords, forms the four 7-byte windows with
// v_alignbyte and reduces them with 2 x v_dot4_u32_u8 each; the last 7 horizontal results per pixel stay in
// VGPRs (fully unrolled ring), so the vertical pass is 7 v_mad per pixel and one packed dword store per row.
constexpr int kBlurW = 256;      // output pixels per workgroup row (64 lanes x 4 px)
#ifndef DCS_BLUR_ROWS                        // tuning hooks (scratch/ab builds)
#define DCS_BLUR_ROWS 32
#endif
#ifndef DCS_BLUR_WAVES
#define DCS_BLUR_WAVES 4
#endif
constexpr int kBlurR = DCS_BLUR_ROWS;       // output rows per thread
constexpr int kBlurWaves = DCS_BLUR_WAVES;    // waves per workgroup (stacked in y)

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// how a strip gets the 12 source bytes x0-4 .. x0+7 of a row:
//   kBlurInterior  all of them lie inside the image row: three aligned dword loads
//   kBlurEdge      the window crosses the left / right image border: the same three dword loads (shifted right by one
//                  dword at x0 == 0, predicated against the row's storage), then BORDER_REFLECT_101 is applied in
//                  registers with one v_perm_b32 per dword; the byte shuffle is fixed per lane for the whole strip
//   kBlurBytes     unaligned caller-owned level 0: byte gathers
enum { kBlurInterior = 0, kBlurEdge = 1, kBlurBytes = 2 };

struct BlurEdgeMap {                   // loop-invariant per lane (kBlurEdge)
    int ofs;                           // loaded window starts at x0 - 4 + ofs
    bool ld[3], use_b[3];              // dword t is loaded; target dword t permutes (L2, L1) instead of (L1, L0)
    unsigned sel[3];
};

__device__ __forceinline__ BlurEdgeMap blur_edge_map(int x0, int w, int pitch)
{
    BlurEdgeMap m;
    m.ofs = x0 == 0 ? 4 : 0;
    const int start = x0 - 4 + m.ofs;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        m.ld[t] = start + 4 * t >= 0 && start + 4 * t + 4 <= pitch;
        int q[4], qmin = 99;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int p = x0 - 4 + 4 * t + k;                      // reflect101 (w >= 8 here: one bounce is enough)
            p = p < 0 ? -p : (p >= w ? 2 * w - 2 - p : p);
            q[k] = p - start;                                // byte index inside the loaded 12-byte window, if 0..11
            if (q[k] >= 0 && q[k] <= 11) qmin = min(qmin, q[k]);
        }
        m.use_b[t] = qmin > 3 && qmin != 99;
        unsigned sel = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int qq = m.use_b[t] ? q[k] - 4 : q[k];     // bytes a valid output never needs fall outside: constant 0
            sel |= (unsigned)((q[k] >= 0 && q[k] <= 11 && qq >= 0 && qq <= 7) ? qq : 0x0c) << (8 * k);
        }
        m.sel[t] = sel;
    }
    return m;
}

// horizontal pass of the four pixels of a dword: the 7-byte windows by v_alignbyte, 2 x v_dot4_u32_u8 each (the second accumulates onto the first)
__device__ __forceinline__ void blur_hsum(unsigned d0, unsigned d1, unsigned d2, unsigned (&h)[4])
{
    constexpr unsigned KA = 18u | (34u << 8) | (49u << 16) | (55u << 24);   // taps 0..3
    constexpr unsigned KB = 49u | (34u << 8) | (18u << 16);                 // taps 4..6 (+0)
    h[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), KB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), KA, 0u, false), false);
    h[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), KB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), KA, 0u, false), false);
    h[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), KB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), KA, 0u, false), false);
    h[3] = __builtin_amdgcn_udot4(d2, KB, __builtin_amdgcn_udot4(d1, KA, 0u, false), false);
}

template <int MODE>
__device__ __forceinline__ void blur_hrow(const uint8_t* __restrict__ row, int x0, int w, const BlurEdgeMap& em, unsigned (&h)[4])
{
    unsigned d0, d1, d2;
    if (MODE == kBlurInterior) {
        const unsigned* p = reinterpret_cast<const unsigned*>(row + x0 - 4);
        d0 = p[0]; d1 = p[1]; d2 = p[2];
    } else if (MODE == kBlurEdge) {
        const unsigned* p = reinterpret_cast<const unsigned*>(row + x0 - 4 + em.ofs);
        const unsigned l0 = em.ld[0] ? p[0] : 0u, l1 = em.ld[1] ? p[1] : 0u, l2 = em.ld[2] ? p[2] : 0u;
        d0 = __builtin_amdgcn_perm(em.use_b[0] ? l2 : l1, em.use_b[0] ? l1 : l0, em.sel[0]);
        d1 = __builtin_amdgcn_perm(em.use_b[1] ? l2 : l1, em.use_b[1] ? l1 : l0, em.sel[1]);
        d2 = __builtin_amdgcn_perm(em.use_b[2] ? l2 : l1, em.use_b[2] ? l1 : l0, em.sel[2]);
    } else {                                       // unaligned level 0: byte gathers with reflection
        unsigned b[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] = row[reflect101(x0 - 4 + k, w)];
        d0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        d1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        d2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    }
    blur_hsum(d0, d1, d2, h);
}

// K * b + c with K an inline constant (the compiler splits this into v_mul_u32_u24 + v_add otherwise)
template <int K>
__device__ __forceinline__ unsigned kmad24(unsigned b, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "n"(K), "v"(b), "v"(c));
    return r;
}
template <int K>
__device__ __forceinline__ unsigned kmad24s(unsigned b, unsigned c_uniform)      // addend in a scalar register
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "n"(K), "v"(b), "s"(c_uniform));
    return r;
}

// 55 * centre row + rounding. The centre row's horizontal sum is a v_dot4 RESULT read directly, and gfx950 needs three wait states between
// a dot instruction's write and another VALU's read of it: the compiler pads its own instructions with s_nop but does not look inside
// an asm statement. From the second output row on the centre row is three rows old; in the FIRST output row of a strip (kk == 6) all
// seven rows have just been computed and the scheduler may place the dot right before the asm -- k_blur_fold returned pixels 100 too
// bright there, in runs that depended on the timing -- so that one row uses the plain expression (v_mul + v_add, hazards handled).
__device__ __forceinline__ unsigned blur_centre_term(unsigned centre, bool first_output_row)
{
    return first_output_row ? __umul24(centre, 55u) + 32768u : kmad24s<55>(centre, 32768u);
}

// one 4-pixel-wide, kBlurR-row strip. YEDGE = the strip touches the top / bottom of the image (row reflection and the
// y < h store predicate); interior strips walk a plain row pointer.
template <int MODE, bool YEDGE>
__device__ __forceinline__ void blur_strip(const uint8_t* __restrict__ S, uint8_t* __restrict__ D, const LevelView& sv, const LevelView& dv,
                                           int x0, int y0)
{
    BlurEdgeMap em{};
    if (MODE == kBlurEdge) em = blur_edge_map(x0, sv.w, sv.pitch);
    unsigned ring[7][4];
    const uint8_t* rowp = S + (ptrdiff_t)(y0 - 3) * sv.pitch;
    uint8_t* outp = D + (size_t)y0 * dv.pitch + x0;
    // The ring slot of every row must be a compile-time constant: hot strips (no row reflection) are unrolled completely,
    // the strips at the top / bottom of an image roll the rows in groups of 7 (the ring period) to keep the code small.
    constexpr int kGroup = YEDGE ? 7 : 42;
#pragma unroll 1
    for (int r0 = 0; r0 < kBlurR + 6; r0 += kGroup) {
#pragma unroll
        for (int kk = 0; kk < kGroup; ++kk) {
            const int k = kk % 7;
            const int r = r0 + kk;
            if (r >= kBlurR + 6) break;
            const uint8_t* row = YEDGE ? S + (size_t)reflect101(y0 + r - 3, sv.h) * sv.pitch : rowp + (size_t)r * sv.pitch;
            blur_hrow<MODE>(row, x0, sv.w, em, ring[k]);
            if (r >= 6) {                             // output row y0 + r - 6: window = input rows r-6 .. r
                unsigned packed = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // operands < 2^24: v_mad_u32_u24 chain (full rate), 32-bit accumulate (max 257 * 65535); rounding folded in
                    unsigned acc = blur_centre_term(ring[(k + 4) % 7][i], kk == 6);
                    acc = kmad24<18>(ring[(k + 1) % 7][i] + ring[k][i], acc);
                    acc = kmad24<34>(ring[(k + 2) % 7][i] + ring[(k + 6) % 7][i], acc);
                    acc = kmad24<49>(ring[(k + 3) % 7][i] + ring[(k + 5) % 7][i], acc);
                    packed |= min(255u, acc >> 16) << (8 * i);
                }
                if (!YEDGE || y0 + r - 6 < dv.h) *reinterpret_cast<unsigned*>(outp + (size_t)(r - 6) * dv.pitch) = packed;
            }
        }
    }
}

__device__ __forceinline__ bool level_aligned(const LevelView& v, int img)
{ return ((reinterpret_cast<uintptr_t>(v.base + (size_t)img * v.img_stride) | (uintptr_t)v.pitch) & 3) == 0; }

// interior strips: all 12 source bytes of every row lie inside the image row -> divergence-free dword path.
// (an unaligned caller-owned level 0 takes the byte path for every lane instead)
// Interior dwords (the 7-tap window of all 4 pixels lies inside the row): x0 = 4, 8, .. 4 n_int with n_int = (w - 8) / 4.
// Lanes run over (image, interior dword) pairs -- every image of the batch shares the geometry, so the waves are full at
// every level width and stay uniform in y; workgroup = 4 waves stacked in y (32 rows each). One launch for all levels:
// blockIdx.x -> (level, lane block, strip block).
__device__ __forceinline__ int blur_n_int(int w) { return w >= 12 ? (w - 8) / 4 : 0; }

__global__ __launch_bounds__(64 * kBlurWaves) void k_blur(LevelSet src, LevelSet dst, int n_images)
{
    int t = blockIdx.x, l = 0, nbx = 0;
    for (; l < src.nlevels; ++l) {
        nbx = (n_images * blur_n_int(src.lv[l].w) + 63) / 64;
        const int n = nbx * ((src.lv[l].h + kBlurR * kBlurWaves - 1) / (kBlurR * kBlurWaves));
        if (t < n) break;
        t -= n;
    }
    if (l >= src.nlevels) return;
    const LevelView sv = src.lv[l], dv = dst.lv[l];
    const int n_int = blur_n_int(sv.w);
    const int li = (t % nbx) * 64 + (int)threadIdx.x;
    const int img = li / n_int;
    const int x0 = 4 + 4 * (li - img * n_int);
    const int y0 = (t / nbx) * (kBlurR * kBlurWaves) + kBlurR * (int)threadIdx.y;
    if (y0 >= sv.h || img >= n_images) return;
    if (!level_aligned(sv, img)) return;                                          // k_blur_unaligned_l0
    const uint8_t* S = sv.base + (size_t)img * sv.img_stride;
    uint8_t* D = const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride;
    if (y0 < 3 || y0 + kBlurR + 3 > sv.h) blur_strip<kBlurInterior, true>(S, D, sv, dv, x0, y0);      // wave-uniform
    else blur_strip<kBlurInterior, false>(S, D, sv, dv, x0, y0);
}

// The dword columns whose window crosses the left / right image border (x0 = 0 and the last one or two dwords of a row,
// <= 2 % of the pixels): lane = (image, 32-row strip, edge column), the same streaming strip with the reflected window
// assembled in registers (v_perm selectors fixed per lane). Rows narrower than 8 pixels take the byte path.
__global__ __launch_bounds__(64) void k_blur_edge_cols(LevelSet src, LevelSet dst, int n_images)
{
    int t = blockIdx.x * 64 + threadIdx.x, l = 0, n_edge = 0, n_strip = 0;
    for (; l < src.nlevels; ++l) {
        const int w = src.lv[l].w, n_x4 = (w + 3) / 4;
        n_edge = n_x4 - blur_n_int(w);
        n_strip = (src.lv[l].h + kBlurR - 1) / kBlurR;
        const int n = n_images * n_strip * n_edge;
        if (t < n) break;
        t -= n;
    }
    if (l >= src.nlevels) return;
    const LevelView sv = src.lv[l], dv = dst.lv[l];
    const int e = t % n_edge, strip = (t / n_edge) % n_strip, img = t / (n_edge * n_strip);
    const int n_int = blur_n_int(sv.w);
    const int x0 = e == 0 ? 0 : 4 * (n_int + e);                                   // dwords after the interior run
    if (!level_aligned(sv, img)) return;
    const uint8_t* S = sv.base + (size_t)img * sv.img_stride;
    uint8_t* D = const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride;
    if (sv.w >= 8) blur_strip<kBlurEdge, true>(S, D, sv, dv, x0, strip * kBlurR);
    else blur_strip<kBlurBytes, true>(S, D, sv, dv, x0, strip * kBlurR);
}

// ---- folded form (the default when every level is >= 16 pixels wide and a level's images span < 4 GB): ONE launch, lanes run over
// (image, dword) pairs of ALL dwords of a row, the border dwords included, so the lines that hold a row's first and last bytes are
// fetched once, by the waves that stream the row anyway (k_blur_edge_cols fetched one line per 12 useful bytes: 8-16 x its algorithmic
// traffic, 30 % of the blur's time). A wave without a border dword runs the plain path; a wave with one (41 % of the waves of a
// 640-wide level, all of them from 256 pixels down) pays three v_perm_b32 per row: every lane loads three dwords la, lb, lc at
// per-lane offsets and forms d0 = perm(lb, la, s0), d1 = perm(lc, lb, s1), d2 = perm(lc, lb, s2) -- the operand assignment is the
// same for every lane, only offsets and selectors differ:
//   interior   (la, lb, lc) = dwords at x0 - 4, x0, x0 + 4, identity selectors
//   x0 == 0    (0, 4, 0): d0 = pixels (4, 3, 2, 1) out of (lb, la); d1 = lc, d2 = lb
//   right edge (x0 - 4, x0, x0 + 4 or, for the row's last dword, x0 - 4 again): BORDER_REFLECT_101 of the pixels >= w always lands
//              inside the pair (lc, lb); window bytes no valid output reads select constant zero
// Addresses are buffer-resource addresses: the level's base in a descriptor, the row offset in an SGPR (scalar arithmetic only) and a
// 32-bit per-lane offset that never changes -- no vector address arithmetic per row at all (the plain kernel above spends two 64-bit
// adds per row on it).
struct BlurFold { unsigned oa, ob, oc, s0, s1, s2; };

__device__ __forceinline__ BlurFold blur_fold_consts(int x0, int w, int last)          // wave-uniform arguments: scalar code
{
    BlurFold f;
    if (x0 == 0) { f.oa = 0; f.ob = 4; f.oc = 0; }
    else { f.oa = (unsigned)(x0 - 4); f.ob = (unsigned)x0; f.oc = (unsigned)(x0 + 4 <= last ? x0 + 4 : x0 - 4); }
    auto sel = [&](int t, int lo, int hi) {
        unsigned s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int p = x0 - 4 + 4 * t + k;
            p = p < 0 ? -p : (p >= w ? 2 * w - 2 - p : p);                              // w >= 16: one bounce
            const unsigned idx = (p >= lo && p < lo + 4) ? (unsigned)(p - lo) : (p >= hi && p < hi + 4) ? (unsigned)(4 + p - hi) : 0x0cu;
            s |= idx << (8 * k);
        }
        return s;
    };
    f.s0 = sel(0, (int)f.oa, (int)f.ob); f.s1 = sel(1, (int)f.ob, (int)f.oc); f.s2 = sel(2, (int)f.ob, (int)f.oc);
    return f;
}

// srs / drs: descriptors of the level's source / destination storage (wave-uniform); va / vb / vc / vd: per-lane byte offsets of the
// three source dwords and of the destination dword inside row 0 of the lane's image; y0 wave-uniform
typedef unsigned u32x3_t __attribute__((ext_vector_type(3)));
template <bool EDGE, bool YEDGE>
__device__ __forceinline__ void blur_strip_fold(__amdgpu_buffer_rsrc_t srs, __amdgpu_buffer_rsrc_t drs, const LevelView& sv, const LevelView& dv,
                                                unsigned va, unsigned vb, unsigned vc, unsigned s0, unsigned s1, unsigned s2, unsigned vd, int y0)
{
    unsigned ring[7][4];
    constexpr int kGroup = YEDGE ? 7 : 42;
#pragma unroll 1
    for (int r0 = 0; r0 < kBlurR + 6; r0 += kGroup) {
#pragma unroll
        for (int kk = 0; kk < kGroup; ++kk) {
            const int k = kk % 7;
            const int r = r0 + kk;
            if (r >= kBlurR + 6) break;
            const int ry = YEDGE ? reflect101(y0 + r - 3, sv.h) : y0 + r - 3;
            const int row = ry * sv.pitch;                                            // scalar (soffset of the loads)
            unsigned d0, d1, d2;
            if (!EDGE) {
                const u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(srs, (int)va, row, 0);
                d0 = v.x; d1 = v.y; d2 = v.z;
            } else {
                const unsigned la = __builtin_amdgcn_raw_buffer_load_b32(srs, (int)va, row, 0), lb = __builtin_amdgcn_raw_buffer_load_b32(srs, (int)vb, row, 0),
                               lc = __builtin_amdgcn_raw_buffer_load_b32(srs, (int)vc, row, 0);
                d0 = __builtin_amdgcn_perm(lb, la, s0); d1 = __builtin_amdgcn_perm(lc, lb, s1); d2 = __builtin_amdgcn_perm(lc, lb, s2);
            }
            blur_hsum(d0, d1, d2, ring[k]);
            if (r >= 6) {
                unsigned packed = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned acc = blur_centre_term(ring[(k + 4) % 7][i], kk == 6);
                    acc = kmad24<18>(ring[(k + 1) % 7][i] + ring[k][i], acc);
                    acc = kmad24<34>(ring[(k + 2) % 7][i] + ring[(k + 6) % 7][i], acc);
                    acc = kmad24<49>(ring[(k + 3) % 7][i] + ring[(k + 5) % 7][i], acc);
                    packed |= min(255u, acc >> 16) << (8 * i);
                }
                if (!YEDGE || y0 + r - 6 < dv.h) __builtin_amdgcn_raw_buffer_store_b32(packed, drs, (int)vd, (y0 + r - 6) * dv.pitch, 0);
            }
        }
    }
}

__global__ __launch_bounds__(64 * kBlurWaves) void k_blur_fold(LevelSet src, LevelSet dst, int n_images)
{
    int t = blockIdx.x, l = 0, nbx = 0;
    for (; l < src.nlevels; ++l) {
        nbx = (n_images * ((src.lv[l].w + 3) >> 2) + 63) / 64;
        const int n = nbx * ((src.lv[l].h + kBlurR * kBlurWaves - 1) / (kBlurR * kBlurWaves));
        if (t < n) break;
        t -= n;
    }
    if (l >= src.nlevels) return;
    const LevelView sv = src.lv[l], dv = dst.lv[l];
    const int n_x4 = (sv.w + 3) >> 2, last = 4 * (n_x4 - 1);
    const int li = (t % nbx) * 64 + (int)threadIdx.x;
    const int img = li / n_x4;
    const int x0 = 4 * (li - img * n_x4);
    const int y0 = (t / nbx) * (kBlurR * kBlurWaves) + kBlurR * __builtin_amdgcn_readfirstlane((int)threadIdx.y);       // wave-uniform
    if (y0 >= sv.h) return;
    const bool valid = img < n_images;
    const bool any_edge = __builtin_amdgcn_ballot_w64(valid && (x0 == 0 || x0 >= last - 4)) != 0;
    if (!valid || !level_aligned(sv, img)) return;                                   // unaligned images: k_blur_unaligned_l0
    const unsigned so = (unsigned)img * (unsigned)sv.img_stride, vd = (unsigned)img * (unsigned)dv.img_stride + (unsigned)x0;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sv.base), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(dv.base), 0, 0xffffffffu, 0x00020000);
    const bool yedge = y0 < 3 || y0 + kBlurR + 3 > sv.h;                             // wave-uniform
    if (!any_edge) {
        const unsigned va = so + (unsigned)(x0 - 4);
        if (yedge) blur_strip_fold<false, true>(srs, drs, sv, dv, va, 0, 0, 0, 0, 0, vd, y0);
        else blur_strip_fold<false, false>(srs, drs, sv, dv, va, 0, 0, 0, 0, 0, vd, y0);
        return;
    }
    const BlurFold f0 = blur_fold_consts(0, sv.w, last), f1 = blur_fold_consts(last - 4, sv.w, last), f2 = blur_fold_consts(last, sv.w, last);
    BlurFold f{(unsigned)(x0 - 4), (unsigned)x0, (unsigned)(x0 + 4), 0x03020100u, 0x03020100u, 0x07060504u};
    if (x0 == 0) f = f0;
    if (x0 == last - 4) f = f1;
    if (x0 == last) f = f2;
    if (yedge) blur_strip_fold<true, true>(srs, drs, sv, dv, so + f.oa, so + f.ob, so + f.oc, f.s0, f.s1, f.s2, vd, y0);
    else blur_strip_fold<true, false>(srs, drs, sv, dv, so + f.oa, so + f.ob, so + f.oc, f.s0, f.s1, f.s2, vd, y0);
}

// caller-owned level 0 whose base / stride is not 4-byte aligned: byte path for every strip (rare; correctness only)
__global__ __launch_bounds__(64) void k_blur_unaligned_l0(LevelSet src, LevelSet dst)
{
    const LevelView sv = src.lv[0], dv = dst.lv[0];
    const int img = blockIdx.z;
    if (level_aligned(sv, img)) return;
    const int x0 = 4 * (blockIdx.x * 64 + (int)threadIdx.x), y0 = kBlurR * (int)blockIdx.y;
    if (x0 >= sv.w || y0 >= sv.h) return;
    blur_strip<kBlurBytes, true>(sv.base + (size_t)img * sv.img_stride, const_cast<uint8_t*>(dv.base) + (size_t)img * dv.img_stride, sv, dv, x0, y0);
}

int launch_blur(const LevelSet& src, const LevelSet& dst, int n_images, hipStream_t s)
{
    // folded form (k_blur_fold): every level >= 16 pixels wide (one reflection bounce, the three border classes distinct) and 32-bit
    // per-lane offsets (a level's images span < 4 GB); DCS_BLUR_FOLD=0 keeps the round-2 pair k_blur + k_blur_edge_cols
    bool fold = opt(OPT_BLUR_FOLD) != 0 && n_images > 0;            // read per launch: the blur is not on the default pipeline's path
    int fold_blocks = 0;
    for (int l = 0; l < src.nlevels && fold; ++l) {
        const LevelView& a = src.lv[l];
        const LevelView& b = dst.lv[l];
        const unsigned long long span_a = (unsigned long long)n_images * a.img_stride + (unsigned long long)a.h * a.pitch + 16;
        const unsigned long long span_b = (unsigned long long)n_images * b.img_stride + (unsigned long long)b.h * b.pitch + 16;
        if (a.w < 16 || span_a >= (1ull << 32) || span_b >= (1ull << 32)) fold = false;
        fold_blocks += ((n_images * ((a.w + 3) / 4) + 63) / 64) * ((a.h + kBlurR * kBlurWaves - 1) / (kBlurR * kBlurWaves));
    }
    if (fold) {
        hipLaunchKernelGGL(k_blur_fold, dim3(fold_blocks), dim3(64, kBlurWaves), 0, s, src, dst, n_images); DCS_CHECK_LAUNCH();
    } else {
        int blocks = 0, edge_lanes = 0;
        for (int l = 0; l < src.nlevels; ++l) {
            const int w = src.lv[l].w, h = src.lv[l].h, n_int = w >= 12 ? (w - 8) / 4 : 0;
            blocks += ((n_images * n_int + 63) / 64) * ((h + kBlurR * kBlurWaves - 1) / (kBlurR * kBlurWaves));
            edge_lanes += n_images * ((h + kBlurR - 1) / kBlurR) * ((w + 3) / 4 - n_int);
        }
        if (blocks) { hipLaunchKernelGGL(k_blur, dim3(blocks), dim3(64, kBlurWaves), 0, s, src, dst, n_images); DCS_CHECK_LAUNCH(); }
        if (edge_lanes) { hipLaunchKernelGGL(k_blur_edge_cols, dim3((edge_lanes + 63) / 64), dim3(64), 0, s, src, dst, n_images); DCS_CHECK_LAUNCH(); }
    }
    if (((reinterpret_cast<uintptr_t>(src.lv[0].base) | (uintptr_t)src.lv[0].pitch | (uintptr_t)src.lv[0].img_stride) & 3) != 0) {
        hipLaunchKernelGGL(k_blur_unaligned_l0, dim3((src.lv[0].w + 255) / 256, (src.lv[0].h + kBlurR - 1) / kBlurR, n_images), dim3(64), 0, s, src, dst);
        DCS_CHECK_LAUNCH();
    }
    return DCS_OK;
}

// ------------------------------------------------------------------------------------- orientation + rBRIEF
// cv::fastAtan2 (OpenCV 3.x): 7th-order odd polynomial on min/max, degrees in [0, 360)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cvRound of a steered pattern coordinate (|x| < 2^22) WITHOUT leaving the float pipe: x + 1.5 * 2^23 rounds to the integer grid with the
// FPU's round-to-nearest-even -- v_rndne_f32's rule -- and its bit pattern is kRnBias + rint(x). The callers fold kRnBias into their
// LDS base addresses (32-bit wrap-around arithmetic), so the rounding costs one v_add_f32 instead of v_rndne_f32 + v_cvt_i32_f32.
constexpr unsigned kRnBias = 0x4B400000u;
#ifdef DCS_EXP_RN
__device__ __forceinline__ unsigned rn_biased(float x) { return (unsigned)__float2int_rn(x) + kRnBias; }
#else
__device__ __forceinline__ unsigned rn_biased(float x) { return __float_as_uint(__fadd_rn(x, 12582912.0f)); }
#endif
typedef __attribute__((address_space(3))) const uint8_t* lds_u8_ptr;
typedef __attribute__((address_space(3))) const uint32_t* lds_u32_ptr;
constexpr int kPatchR = 18;                 // rotated pattern reach: max radius 18.38 -> |coord| <= 18
constexpr int kPatchRows = 2 * kPatchR + 1; // 37
constexpr int kPatchDw = 16;                // dwords per staged row: 4 x 16 B cover 37 bytes + <= 15 alignment bytes
// fused variant (the Gaussian blur of the patch inside k_describe): the raw rows y - 21 .. y + 21 are staged in LDS like the blurred ones
// were, the HORIZONTAL 7-tap pass is an i8 GEMM H = Raw x T on the matrix cores (v_mfma_i32_16x16x64_i8: one K-window of 64 input
// columns covers the patch at any alignment; T = banded Toeplitz matrix of the taps, shifted by the patch's offset in its 16-byte
// aligned rows), its 16-bit sums land in LDS packed as ROW PAIRS (lo = even row, hi = odd row), and the VERTICAL pass runs on demand at
// the <= 512 pixels the tests read (4 v_dot2_u32_u16 each).
constexpr int kRawR = kPatchR + 3;          // 21
constexpr int kRawRows = 2 * kRawR + 1;     // 43
constexpr int kRawTileRows = 48;            // 3 GEMM row tiles (rows 43 .. 47: whatever LDS holds, never read back)
constexpr int kRawStoreRows = 44;           // rows a wave's LDS region really has: the last tile's rows 44 .. 47 are read from whatever follows it (the next wave's
                                            // region, the tables behind s_patch) and their row pairs 22 / 23 are not stored -- 1 280 B less per workgroup: 19 968 B = EIGHT
                                            // workgroups per CU instead of seven (the kernel gains with every resident workgroup: 565 / 479 / 399 / 382 us at 3 / 4 / 6 / 7)
constexpr int kRawPitch = 80;               // bytes per staged raw row: 64 + 16, so that the 16-byte A-operand reads of 8 consecutive rows hit 8 different bank groups
constexpr int kHPairs = 24;                 // row pairs of the 3 x 16 rows the GEMM produces (22 used)
constexpr int kHCols = 40;                  // columns of the horizontal sums (37 used); 2 * 40 dwords = 16 banks: the (g, g + 1) halves of a wave's store miss each other
constexpr int kBTabEntries = 24;            // B operand rows: the 7 taps at byte offset d = -7 .. 16 of a 16-byte k-group (d = -7 and 16: all zero)
typedef int v4i_t __attribute__((ext_vector_type(4)));
#ifndef DCS_DESC_KP                          // tuning hook (scratch/ab builds)
#define DCS_DESC_KP 16
#endif
#ifndef DCS_DESC_WAVES
#define DCS_DESC_WAVES 4
#endif
#ifndef DCS_DESC_WPE                         // minimum waves per SIMD the register allocation must allow
#define DCS_DESC_WPE 8
#endif
constexpr int kDescKp = DCS_DESC_KP;        // keypoints per workgroup
constexpr int kDescWaves = DCS_DESC_WAVES;  // waves per workgroup
constexpr int kDescPerWave = kDescKp / kDescWaves;
static_assert(kDescKp % (4 * kDescWaves) == 0 && kDescKp <= 64, "phase A works on 4 keypoints per wave at a time, phase B on one lane per keypoint");
constexpr int kIcCols = 8;                  // dwords covering x-15 .. x+16, read at the keypoint's own byte alignment (the 32nd byte is masked out)

// byte masks of the umax disc for the IC_Angle dword tasks: entry [row * 8 + col] selects the bytes b of the dword at
// u = -15 + 4 * col + b with |u| <= umax[|row - 15|]; row 31 (the partner of row 30 in the last round) is empty
void build_ic_mask(const int* umax, uint32_t* out /* kIcMaskWords */)
{
    for (int row = 0; row < 32; ++row)
        for (int col = 0; col < kIcCols; ++col) {
            uint32_t m = 0;
            for (int bb = 0; bb < 4; ++bb) {
                const int u = -kHalfPatch + 4 * col + bb;
                if (row < kPatchSize && std::abs(u) <= umax[std::abs(row - kHalfPatch)]) m |= 0xffu << (8 * bb);
            }
            out[row * kIcCols + col] = m;
        }
}

// cosf / sinf of the keypoint angle, bit for bit what glibc (>= 2.28, sysdeps/ieee754/flt-32/s_sincosf.h) returns for
// 0 <= x < 120: the reference's `cos(angle)` on a float resolves to the float overload (src/ORBextractor.cc:66-67, 112-113).
// Reduction by pi/2 and two minimax polynomials in double, no FMA contraction (this file is built with -ffp-contract=off;
// libm's contracted variant returns the same floats on this range). tests/test_oracle_extract.py holds the oracle's
// restatement of the algorithm against the host's libm, tests/test_gpu_extract.py holds this function against both.
// glibc's second table entry is the first with the cosine coefficients negated: every product and sum then flips its sign
// exactly, so the value is negated at the end instead.
__device__ __forceinline__ float glibc_sin_poly(double x, double x2)
{
    const double x3 = x * x2, s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13, x7 = x3 * x2, s = x + x3 * -0x1.555545995a603p-3;
    return (float)(s + x7 * s1);
}
__device__ __forceinline__ float glibc_cos_poly(double x2, bool negate)
{
    const double x4 = x2 * x2, c2 = -0x1.6c087e89a359dp-10 + x2 * 0x1.99343027bf8c3p-16, c1 = 0x1p0 + x2 * -0x1.ffffffd0c621cp-2;
    const double x6 = x4 * x2, c = c1 + x4 * 0x1.55553e1068f19p-5, r = c + x6 * c2;
    return (float)(negate ? -r : r);
}
__device__ __forceinline__ void glibc_sincosf(float y, float& c_out, float& s_out)
{
    double x = (double)y;
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ffu;
    if (top < 0x3f4u) {                                      // abstop12(pi/4 as float 0x3f490fdb)
        const double x2 = x * x;
        const bool tiny = top < 0x398u;                      // abstop12(0x1p-12f)
        c_out = tiny ? 1.0f : glibc_cos_poly(x2, false);
        s_out = tiny ? y : glibc_sin_poly(x, x2);
        return;
    }
    const double r = x * 0x1.45F306DC9C883p+23;              // 2/pi * 2^24
    const int n = ((int)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double xs = ((n & 3) == 1 || (n & 3) == 2) ? -x : x;   // sign[n & 3] = {1, -1, -1, 1}
    const double x2 = x * x;
    const bool neg = (n & 2) != 0;
    // sinf: polynomial picked by n, cosf: by n ^ 1 (odd -> cosine polynomial)
    s_out = (n & 1) ? glibc_cos_poly(x2, neg) : glibc_sin_poly(xs, x2);
    c_out = (n & 1) ? glibc_sin_poly(xs, x2) : glibc_cos_poly(x2, neg);
}
__global__ void k_debug_sincosf(const float* __restrict__ x, int n, float* __restrict__ c, float* __restrict__ s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) glibc_sincosf(x[i], c[i], s[i]);
}
int launch_debug_sincosf(const float* d_x, int n, float* d_c, float* d_s, hipStream_t st)
{
    hipLaunchKernelGGL(k_debug_sincosf, dim3((n + 255) / 256), dim3(256), 0, st, d_x, n, d_c, d_s);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

// Workgroup = 64 keypoints of one image, three phases:
//   A. IC_Angle moments (exact int32). A 16-lane row of a wave owns one keypoint; its 279 dword tasks (31 rows x 9
//      aligned dwords) are masked with the precomputed disc mask and reduced with v_dot4_u32_u8:
//      sum(val) and sum((u + 32) * val) per dword, m01 += v * sum(val).
//   B. one LANE per keypoint: fastAtan2 + libm's cosf / sinf (glibc_sincosf: the reference calls the float overloads,
//      ORBextractor.cc:112-113) -- issued once per 64 keypoints instead of once per keypoint-wave -- and the cv::KeyPoint.
//   C. one wave per keypoint: 37x64 B blurred neighbourhood -> LDS (16-byte loads), 4 rounds of 64 rBRIEF tests.
// FUSED keeps four workgroups per CU: <= 40 960 B of LDS each and <= 128 registers per lane (amdgpu_waves_per_eu makes that the compiler's budget)
template <bool FUSED>       // FUSED: no blurred pyramid exists, phase C blurs the raw patch itself
__global__ __launch_bounds__(64 * kDescWaves) __attribute__((amdgpu_waves_per_eu(DCS_DESC_WPE))) void k_describe(LevelSet raw, LevelSet blurred, DescribeParams prm,
                                                  const SelKp* __restrict__ sel, const int32_t* __restrict__ img_off,
                                                  const int32_t* __restrict__ lvl_cnt, dcs_keypoint* __restrict__ kp_out,
                                                  uint8_t* __restrict__ desc_out, int cap, int32_t* __restrict__ n_out, int n_images, int chunks,
                                                  const int32_t* __restrict__ dense_total, int dense_cap)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_patch[kDescWaves][FUSED ? kRawStoreRows * kRawPitch / 4 : kPatchRows * kPatchDw];
    __shared__ __attribute__((aligned(16))) uint32_t s_btab[FUSED ? kBTabEntries * 4 : 4];
    __shared__ float4 s_pattern[256];
    __shared__ uint32_t s_mask[kIcMaskWords];
    __shared__ SelKp s_sel[kDescKp];
    __shared__ int s_m10[kDescKp], s_m01[kDescKp];
    __shared__ float s_cos[kDescKp], s_sin[kDescKp];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int img, chunk;
    xcd_image_block(blockIdx.x, n_images, chunks, img, chunk);
    const int i0 = chunk * kDescKp;
    // The batch's FAST candidates did not fit the handle's dense buffer (k_gather dropped the excess): nothing downstream of
    // it is the reference's result, so every image of the call reports DCS_ERR_CAPACITY in place of a count -- the
    // asynchronous API has no other way to fail, and a negative count cannot be mistaken for features.
    if (dense_total && *dense_total > dense_cap) {
        if (chunk == 0 && tid == 0) n_out[img] = DCS_ERR_CAPACITY;
        return;
    }
    int n_img;
    if (lvl_cnt) {                                           // device quadtree: per-level slots, level-major output order
        int acc = 0;
        for (int l = 0; l < prm.nlevels; ++l) acc += lvl_cnt[img * prm.nlevels + l];
        n_img = min(acc, cap);
    } else n_img = min(img_off[img + 1] - img_off[img], cap);
    if (chunk == 0 && tid == 0) n_out[img] = n_img;
    if (i0 >= n_img) return;                                 // block-uniform
    {
        for (int e = tid; e < 256; e += 64 * kDescWaves) {
            const char4 pt = reinterpret_cast<const char4*>(c_pattern)[e];
            s_pattern[e] = float4{(float)pt.x, (float)pt.y, (float)pt.z, (float)pt.w};
        }
        for (int e = tid; e < kIcMaskWords; e += 64 * kDescWaves) s_mask[e] = prm.ic_mask[e];
        if constexpr (FUSED) {
            for (int e = tid; e < kBTabEntries * 4; e += 64 * kDescWaves) {      // dword w of entry e: byte j = 4 w + b holds tap[j - d], d = e - 7
                const int d = (e >> 2) - 7;
                unsigned v = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int t = 4 * (e & 3) + bb - d;
                    const unsigned tap = t == 0 || t == 6 ? 18u : t == 1 || t == 5 ? 34u : t == 2 || t == 4 ? 49u : t == 3 ? 55u : 0u;
                    v |= tap << (8 * bb);
                }
                s_btab[e] = v;
            }
        }
        if (tid < kDescKp) {
            const int i = i0 + tid;
            int src;
            if (lvl_cnt) {
                int acc = 0;
                src = -1;
                for (int l = 0; l < prm.nlevels; ++l) {
                    const int cnt = lvl_cnt[img * prm.nlevels + l];
                    if (src < 0 && i < acc + cnt) src = img * prm.out_per_image + prm.out_base[l] + (i - acc);
                    acc += cnt;
                }
                if (src < 0) src = img * prm.out_per_image;
            } else src = img_off[img] + (i < n_img ? i : 0);
            s_sel[tid] = sel[src];
        }
    }
    __syncthreads();
    const int n_here = min(kDescKp, n_img - i0);             // keypoints of this workgroup

    // ---- A. IC_Angle moments on the unblurred level
#if !defined(DCS_DESCRIBE_SKIP) || DCS_DESCRIBE_SKIP != 1        // profiling side builds: 1 = without phase A, 2 = without phase C
    {
        const int grp = lane >> 4, sub = lane & 15;
#pragma unroll 1
        for (int batch = 0; batch < kDescPerWave / 4; ++batch) {
            const int kq = wave * kDescPerWave + batch * 4 + grp;
            if (wave * kDescPerWave + batch * 4 >= n_here) break;      // wave-uniform
            const SelKp k = s_sel[min(kq, n_here - 1)];
            const int x = k.x, y = k.y;
            const LevelView rv = raw.lv[k.level];
            const uint8_t* rimg = rv.base + (size_t)img * rv.img_stride;
            int m10 = 0, m01 = 0;
            {
                // Round 3: the 31 x 31 disc is read as 8 dwords per row AT THE KEYPOINT'S OWN BYTE ALIGNMENT (global loads need none; the
                // level may even be a caller-owned, unaligned level 0): a 16-lane group is two rows x eight dwords, so a lane keeps its
                // column -- its weight word (u + 32 per byte) is a constant, its mask a fixed stride through the table, its pointer two
                // rows further every round; 16 rounds of 8 vector instructions. (Until then: aligned dwords, 9 per row, 18 rounds of 16
                // with the column / row / weight bookkeeping of a 9-column walk on 16 lanes, and a byte-load path for unaligned input:
                // 120 vector instructions per keypoint in this phase.)
                const int r2 = sub >> 3, col = sub & 7;
                const uint8_t* pp = rimg + (size_t)(y - kHalfPatch + r2) * rv.pitch + (x - kHalfPatch + 4 * col);
                const long step = 2 * (long)rv.pitch;
                const unsigned ub = (unsigned)(17 + 4 * col);                                  // u + 32 of byte 0
                const unsigned w = (__umul24(ub, 0x010101u) + 0x03020100u) + (ub << 24);      // u + 32 per byte, no carries
                const uint32_t* mk = s_mask + sub;                                             // entry (2 j + r2) * 8 + col = sub + 16 j
                unsigned s_all = 0, s_u = 0;                 // sum(val), sum((u + 32) * val)
                int rowm = r2 - kHalfPatch;
#pragma unroll
                for (int j = 0; j < 16; ++j) {               // row 31 (j = 15, r2 = 1) lies inside the image (y + 16 <= h - 4) and has an empty mask
                    unsigned val;
                    __builtin_memcpy(&val, pp, 4);
                    val &= mk[16 * j];
                    const unsigned rs = __builtin_amdgcn_udot4(val, 0x01010101u, 0u, false);
                    s_u = __builtin_amdgcn_udot4(val, w, s_u, false);
                    s_all += rs;
                    m01 += __mul24(rowm, (int)rs);
                    pp += step; rowm += 2;
                }
                m10 = (int)s_u - 32 * (int)s_all;
            }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) { m10 += __shfl_xor(m10, d); m01 += __shfl_xor(m01, d); }
            if (sub == 0 && kq < n_here) { s_m10[kq] = m10; s_m01[kq] = m01; }
        }
    }
#endif
    __syncthreads();
    // ---- B. orientation, steering coefficients a = cosf(rad), b = sinf(rad) exactly as libm computes them (glibc_sincosf), keypoint
    if (tid < n_here) {
        const SelKp k = s_sel[tid];
        const float angle = fast_atan2_deg((float)s_m01[tid], (float)s_m10[tid]);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float rad = __fmul_rn(angle, factorPI);
        glibc_sincosf(rad, s_cos[tid], s_sin[tid]);
        dcs_keypoint o;
        const int level = k.level;
        const float sc = prm.scale[level];
        o.x = level ? __fmul_rn((float)k.x, sc) : (float)k.x;
        o.y = level ? __fmul_rn((float)k.y, sc) : (float)k.y;
        o.size = (float)prm.scaled_patch[level];
        o.angle = angle; o.response = (float)k.score; o.octave = level; o.class_id = -1;
        kp_out[(size_t)img * cap + i0 + tid] = o;
    }
    __syncthreads();
    // ---- C. steered BRIEF, one wave per keypoint
#if defined(DCS_DESCRIBE_SKIP) && DCS_DESCRIBE_SKIP == 2
    return;
#endif
    if constexpr (!FUSED) {
    uint32_t* patch = s_patch[wave];
    const int r_lane = lane >> 2;                            // 16 rows x 4 x 16 B per wave pass; 37 rows = 3 passes (last: 5 rows)
    uint4 q0, q1, q2;
    int shift = 0;
    auto fetch = [&](int kq) {                               // blurred neighbourhood of keypoint kq -> registers
        const SelKp k = s_sel[kq];
        const LevelView bv = blurred.lv[k.level];
        const uint8_t* bimg = bv.base + (size_t)img * bv.img_stride;
        const int xs = (k.x - kPatchR) & ~15;                // blurred slab: 256-B aligned levels, pitch % 64 == 0
        shift = (k.x - kPatchR) & 15;
        const uint8_t* bsrc = bimg + (size_t)(k.y - kPatchR) * bv.pitch + xs + 16 * (lane & 3);
        q0 = *reinterpret_cast<const uint4*>(bsrc + (size_t)r_lane * bv.pitch);
        q1 = *reinterpret_cast<const uint4*>(bsrc + (size_t)(r_lane + 16) * bv.pitch);
        q2 = *reinterpret_cast<const uint4*>(bsrc + (size_t)min(r_lane + 32, kPatchRows - 1) * bv.pitch);
    };
    if (wave * kDescPerWave < n_here) fetch(wave * kDescPerWave);
#pragma unroll 1
    for (int kk = 0; kk < kDescPerWave; ++kk) {
        const int kq = wave * kDescPerWave + kk;
        if (kq >= n_here) break;                              // wave-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // previous keypoint's LDS reads are done
        __builtin_amdgcn_wave_barrier();
        reinterpret_cast<uint4*>(patch)[lane] = q0;
        reinterpret_cast<uint4*>(patch)[lane + 64] = q1;
        if (lane + 128 < kPatchRows * 4) reinterpret_cast<uint4*>(patch)[lane + 128] = q2;
        // address of patch pixel (r, c) = pb + 64 r + c; with biased coordinates the two biases go into the base
        const unsigned pb = lds_addr(patch) + (unsigned)(kPatchR * (kPatchDw * 4) + kPatchR + shift) - (kPatchDw * 4 + 1) * kRnBias;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (kq + 1 < min(n_here, wave * kDescPerWave + kDescPerWave)) fetch(kq + 1);        // next keypoint's loads fly during this one's tests
        const float a = s_cos[kq], b = s_sin[kq];
        uint8_t* dout = desc_out + ((size_t)img * cap + i0 + kq) * 32;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float4 pt = s_pattern[it * 64 + lane];
            const unsigned r0 = rn_biased(__fadd_rn(__fmul_rn(pt.x, b), __fmul_rn(pt.y, a)));
            const unsigned c0 = rn_biased(__fsub_rn(__fmul_rn(pt.x, a), __fmul_rn(pt.y, b)));
            const unsigned r1 = rn_biased(__fadd_rn(__fmul_rn(pt.z, b), __fmul_rn(pt.w, a)));
            const unsigned c1 = rn_biased(__fsub_rn(__fmul_rn(pt.z, a), __fmul_rn(pt.w, b)));
            const int t0 = *(lds_u8_ptr)(uintptr_t)(pb + r0 * (kPatchDw * 4) + c0), t1 = *(lds_u8_ptr)(uintptr_t)(pb + r1 * (kPatchDw * 4) + c1);
            const unsigned long long m = __ballot(t0 < t1);  // bit j of m = test 64*it + j  (LSB-first bytes)
            if (lane == 0) *reinterpret_cast<unsigned long long*>(dout + 8 * it) = m;
        }
    }
    } else {
    uint32_t* patch = s_patch[wave];
    uint32_t* hp = patch;                                    // the GEMM works in place: 16 staged rows (16 x 80 B) become 8 row pairs (8 x 160 B)
    static_assert(16 * kRawPitch == 8 * kHCols * 4 && kRawTileRows * kRawPitch == kHPairs * kHCols * 4 && kRawStoreRows >= kRawRows && kRawStoreRows % 2 == 0, "in-place horizontal pass");
    const int r_lane = lane >> 2, c4 = lane & 3;             // staging: 16 rows x 4 x 16 B per wave pass; 43 rows = 3 passes
    const int mc = lane & 15, mg = lane >> 4;                // GEMM: A row / B column mc, k-group mg (k = 16 mg + byte); D: column mc, rows 4 mg + r
    uint4 q0, q1, q2;
    int shift = 0;
    bool wide = true, xedge = false;                         // this keypoint: 16-byte row loads possible; patch crosses the left / right border
    int kx = 0, kw = 0;
    auto fetch = [&](int kq) {                               // raw neighbourhood of keypoint kq -> registers (rows reflected at the level's top / bottom)
        const SelKp k = s_sel[kq];
        const LevelView rv = raw.lv[k.level];
        const uint8_t* rimg = rv.base + (size_t)img * rv.img_stride;
        kx = k.x; kw = rv.w;
        wide = ((reinterpret_cast<uintptr_t>(rimg) | (uintptr_t)rv.pitch) & 15) == 0;       // a caller-owned level 0 may not be
        xedge = k.x - kRawR < 0 || k.x + kRawR >= rv.w;
        const int xs = (k.x - kRawR) & ~15;
        shift = (k.x - kRawR) - xs;
        if (!wide) return;
        // 16-byte chunks that lie completely outside the row are loaded from the nearest chunk inside it (never used: the columns
        // they stand for are rewritten by the reflection fix-up)
        const int col = min(max(xs + 16 * c4, 0), (rv.w - 1) & ~15);
        const uint8_t* src = rimg + col;
        // rows: reflected only when the neighbourhood crosses the level's top / bottom (y within two rows of the 19-px border) -- a
        // wave-uniform test instead of three reflections per lane
        int ra = k.y - kRawR + r_lane, rb = ra + 16, rc = k.y - kRawR + min(r_lane + 32, kRawRows - 1);
        if (k.y - kRawR < 0 || k.y + kRawR >= rv.h) { ra = reflect101(ra, rv.h); rb = reflect101(rb, rv.h); rc = reflect101(rc, rv.h); }
        q0 = *reinterpret_cast<const uint4*>(src + (size_t)ra * rv.pitch);
        q1 = *reinterpret_cast<const uint4*>(src + (size_t)rb * rv.pitch);
        q2 = *reinterpret_cast<const uint4*>(src + (size_t)rc * rv.pitch);
    };
    if (wave * kDescPerWave < n_here) fetch(wave * kDescPerWave);
    // u8 -> i8: x ^ 0x80 = x - 128, and sum_k tap[k] * 128 = 257 * 128 = 32896 comes back through the accumulator's initial value
    const v4i_t c_init = {32896, 32896, 32896, 32896};
#pragma unroll 1
    for (int kk = 0; kk < kDescPerWave; ++kk) {
        const int kq = wave * kDescPerWave + kk;
        if (kq >= n_here) break;                              // wave-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // previous keypoint's LDS reads are done
        __builtin_amdgcn_wave_barrier();
        uint8_t* praw = reinterpret_cast<uint8_t*>(patch);
#ifdef DCS_DESC_SHARE_BOUND       // measurement aid (wrong descriptors): staging + horizontal pass only for a wave's FIRST keypoint = the bound of any scheme that shares horizontal sums
        const bool do_h = kk == 0;
#else
        constexpr bool do_h = true;
#endif
        if (do_h) {
        if (wide) {                                           // wave-uniform
            *reinterpret_cast<uint4*>(praw + r_lane * kRawPitch + 16 * c4) = q0;
            *reinterpret_cast<uint4*>(praw + (r_lane + 16) * kRawPitch + 16 * c4) = q1;
            if (r_lane + 32 < kRawRows) *reinterpret_cast<uint4*>(praw + (r_lane + 32) * kRawPitch + 16 * c4) = q2;
            if (xedge) {                                      // at most 2 columns per side lie outside the row (19 <= x <= w - 20): BORDER_REFLECT_101
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (lane < kRawRows) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int jj = e < 2 ? e : kRawRows - 4 + e, g = kx - kRawR + jj;      // patch columns 0, 1, 41, 42
                        if (g < 0 || g >= kw) praw[lane * kRawPitch + shift + jj] = praw[lane * kRawPitch + shift + (reflect101(g, kw) - (kx - kRawR))];
                    }
                }
            }
        } else {                                              // unaligned caller-owned level 0: byte loads, reflection applied on the way
            const SelKp k = s_sel[kq];
            const LevelView rv = raw.lv[k.level];
            const uint8_t* rimg = rv.base + (size_t)img * rv.img_stride;
            for (int i = lane; i < kRawRows * kRawRows; i += 64) {
                const int r = i / kRawRows, j = i - r * kRawRows;
                praw[r * kRawPitch + shift + j] = rimg[(size_t)reflect101(k.y - kRawR + r, rv.h) * rv.pitch + reflect101(k.x - kRawR + j, rv.w)];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int shift_now = shift;
#ifndef DCS_DESC_SHARE_BOUND
        if (kq + 1 < min(n_here, wave * kDescPerWave + kDescPerWave)) fetch(kq + 1);        // next keypoint's loads fly during this one's work
#endif
        // ---- horizontal pass: h[row][c] = sum_k tap[k] raw[row][c + k] (<= 65 535) for 48 rows x 37 columns as 3 x 3 matrix instructions.
        // A = 16 staged bytes of row 16 mt + mc, k-group mg; B (column tile nt) = the taps at byte offset shift + 16 nt + mc - 16 mg of that
        // k-group, read from the 24-entry table. Register r of lane (mc, mg) holds row 16 mt + 4 mg + r, column 16 nt + mc: rows
        // (4 mg, 4 mg + 1) and (4 mg + 2, 4 mg + 3) are two row pairs, stored with one ds_write2_b32 -- IN PLACE: the 8 row pairs of a
        // row tile take exactly the bytes of its 16 staged rows, every lane has read its A operand by then (LDS operations of a wave
        // complete in order), and column tile 2 (columns 32 .. 47) stores only its first 8 columns, so nothing reaches the next tile's rows.
        {
            v4i_t bt[3];
            // (a 111-entry table without the clamps costs 1.4 KB of LDS: 42.2 KB per workgroup = three workgroups per CU instead of four, +15 % kernel time)
            const int e0 = shift_now + mc - 16 * mg + 7;
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) bt[nt] = *reinterpret_cast<const v4i_t*>(s_btab + 4 * min(max(e0 + 16 * nt, 0), kBTabEntries - 1));
            uint32_t* hrow = hp + (2 * mg) * kHCols + mc;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                v4i_t a = *reinterpret_cast<const v4i_t*>(praw + (16 * mt + mc) * kRawPitch + 16 * mg);
                a ^= (v4i_t){(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
                const v4i_t d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bt[0], c_init, 0, 0, 0);
                const v4i_t d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bt[1], c_init, 0, 0, 0);
                const v4i_t d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bt[2], c_init, 0, 0, 0);
                uint32_t* o = hrow + (8 * mt) * kHCols;
                if (mt == 2 && mg == 3) continue;                // row pairs 22 / 23 (rows 44 .. 47): beyond the wave's region, never read
                o[0] = (unsigned)d0.x | ((unsigned)d0.y << 16); o[kHCols] = (unsigned)d0.z | ((unsigned)d0.w << 16);
                o[16] = (unsigned)d1.x | ((unsigned)d1.y << 16); o[16 + kHCols] = (unsigned)d1.z | ((unsigned)d1.w << 16);
                if (mc < 8) { o[32] = (unsigned)d2.x | ((unsigned)d2.y << 16); o[32 + kHCols] = (unsigned)d2.z | ((unsigned)d2.w << 16); }
            }
        }
        }       // do_h
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- vertical pass ON DEMAND: only the (at most) 512 pixels the tests read are blurred -- rows q .. q + 6 of the horizontal
        // sums = 4 row pairs, taps arranged by the parity of q: 4 v_dot2_u32_u16 per pixel
        const float a = s_cos[kq], b = s_sin[kq];
        uint8_t* dout = desc_out + ((size_t)img * cap + i0 + kq) * 32;
        // biased coordinates (kRnBias + r, kRnBias + c; kRnBias is even): row pair (r + 18) >> 1 = (rb >> 1) - kRnBias / 2 + 9, parity = rb & 1,
        // column c + 18; the constants go into the base address
        // (the row-pair index goes through a 24-bit multiply-add: only the low 24 bits of kRnBias / 2 take part in the product)
        const unsigned hp_base = lds_addr(hp) + (unsigned)(9 * kHCols * 4 + kPatchR * 4) - ((kRnBias / 2) & 0xFFFFFFu) * (kHCols * 4) - kRnBias * 4;
        unsigned hp_pitch_v = (unsigned)(kHCols * 4), hp_base_v = hp_base;
        asm volatile("" : "+v"(hp_pitch_v), "+v"(hp_base_v));                  // loop-invariant VGPR copies
        auto blurred_at = [&](unsigned rb, unsigned cb) {
            const bool odd = rb & 1;
            const unsigned W0 = odd ? (18u << 16) : (18u | (34u << 16)), W1 = odd ? (34u | (49u << 16)) : (49u | (55u << 16)),
                           W2 = odd ? (55u | (49u << 16)) : (49u | (34u << 16)), W3 = odd ? (34u | (18u << 16)) : 18u;
            // two instructions, spelled out (the compiler's own choice is v_mul_u32_u24 + v_lshlrev + v_add3): pitch and base sit in VGPRs
            // because a VOP3 instruction reads at most one scalar register on gfx9
            unsigned rowaddr, coladdr;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(rowaddr) : "v"(rb >> 1), "v"(hp_pitch_v), "v"(hp_base_v));
            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(coladdr) : "v"(cb), "v"(rowaddr));
            const lds_u32_ptr col = (lds_u32_ptr)(uintptr_t)coladdr;
            unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, col[0]), __builtin_bit_cast(ushort2_t, W0), 32768u, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, col[kHCols]), __builtin_bit_cast(ushort2_t, W1), acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, col[2 * kHCols]), __builtin_bit_cast(ushort2_t, W2), acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, col[3 * kHCols]), __builtin_bit_cast(ushort2_t, W3), acc, false);
            return min(255u, acc >> 16);
        };
        unsigned long long bits[4];                          // bit j of bits[it] = test 64 it + j (LSB-first bytes)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float4 pt = s_pattern[it * 64 + lane];
            const unsigned r0 = rn_biased(__fadd_rn(__fmul_rn(pt.x, b), __fmul_rn(pt.y, a)));
            const unsigned c0 = rn_biased(__fsub_rn(__fmul_rn(pt.x, a), __fmul_rn(pt.y, b)));
            const unsigned r1 = rn_biased(__fadd_rn(__fmul_rn(pt.z, b), __fmul_rn(pt.w, a)));
            const unsigned c1 = rn_biased(__fsub_rn(__fmul_rn(pt.z, a), __fmul_rn(pt.w, b)));
            const unsigned t0 = blurred_at(r0, c0), t1 = blurred_at(r1, c1);
            bits[it] = __ballot(t0 < t1);
        }
        if (lane == 0) {                                     // the 32 bytes in two 16-byte stores under ONE exec region (one store per ballot cost 3 scalar instructions each)
            uint4* d4 = reinterpret_cast<uint4*>(dout);
            d4[0] = uint4{(unsigned)bits[0], (unsigned)(bits[0] >> 32), (unsigned)bits[1], (unsigned)(bits[1] >> 32)};
            d4[1] = uint4{(unsigned)bits[2], (unsigned)(bits[2] >> 32), (unsigned)bits[3], (unsigned)(bits[3] >> 32)};
        }
    }
    }
}

int launch_describe(const LevelSet& raw, const LevelSet& blurred, const DescribeParams& prm, const SelKp* d_sel,
                    const int32_t* d_img_off, const int32_t* d_lvl_cnt, int n_images, int max_per_image, dcs_keypoint* d_kp,
                    uint8_t* d_desc, int cap, int32_t* d_n_out, hipStream_t s, const int32_t* d_dense_total, int dense_cap, bool fused)
{
    const int gx = max_per_image > 0 ? (max_per_image + kDescKp - 1) / kDescKp : 1;
    const size_t lds_pad = 0;
    if (fused) hipLaunchKernelGGL(k_describe<true>, dim3(gx * n_images), dim3(64 * kDescWaves), lds_pad, s, raw, blurred, prm, d_sel, d_img_off, d_lvl_cnt, d_kp,
                                  d_desc, cap, d_n_out, n_images, gx, d_dense_total, dense_cap);
    else hipLaunchKernelGGL(k_describe<false>, dim3(gx * n_images), dim3(64 * kDescWaves), 0, s, raw, blurred, prm, d_sel, d_img_off, d_lvl_cnt, d_kp,
                            d_desc, cap, d_n_out, n_images, gx, d_dense_total, dense_cap);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

}  // namespace dcs
