// proj_kernels.hip -- projection-guided matching on gfx950: ORBmatcher::SearchByProjection (src/ORBmatcher.cc:539-624),
// SearchByProjectionOnCam (:954-1113) and Frame::GetFeaturesInArea (src/Frame.cc:316-376).
//
// The reference walks the queries (map points) one after the other; a feature matched by an earlier query is skipped by
// the later ones, so the result depends on the order. Two kernels keep that semantics and still do the heavy part in
// parallel:
//   k_proj_collect  one wave per query: visits the grid cells of the window in the reference's (ix, iy, j) order, applies
//                   the octave and |dx|,|dy| < r tests, drops features that were taken BEFORE the call, computes the
//                   Hamming distance of the survivors (64 candidates per round, one per lane) and appends
//                   (distance, octave, feature) words to the query's candidate list with an ordered ballot compaction
//   k_proj_resolve  one wave per call: walks the queries in order; a query's candidates sit one per lane, the lanes whose
//                   feature has been taken meanwhile (LDS byte map) drop out, best / second are the two smallest
//                   (distance, position) keys of the wave -- the first two elements of a stable sort by distance, which
//                   is what the reference's strict "<" updates produce -- then the TH_HIGH / same-level ratio rules,
//                   the rotation histogram and ComputeThreeMaxima.
// Candidate lists hold up to kProjCap entries; a query with a wider window (relocalisation-sized radii) is re-walked
// inside the resolver with the same visiting code (correct, slower).
#include <hip/hip_runtime.h>

#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "config.h"
#include "track.h"

namespace dcs {
namespace {

constexpr int kProjCap = 64;                    // candidates kept per query (one per lane of the resolver)
constexpr int kHisto = 30;                      // HISTO_LENGTH (ORBmatcher.cc:59)

struct ProjFrameD {
    int n_cams, N;
    const int32_t* cam_off; const float *kp_x, *kp_y; const int32_t* kp_octave; const float* kp_angle;
    const uint8_t *desc, *taken; const float *min_x, *min_y, *w_inv, *h_inv; const int32_t *grid_off, *grid_idx;
    // window searches on a KeyFrame (dcs_search_in_window): KeyFrame::GetFeaturesInArea reads the position of
    // mvTotalKeysUn[local index] (KeyFrame.cc:756) and has no level test -- the octave gate sits in the caller's loop
    // (min <= octave <= max, ORBmatcher.cc:1497, 1662, 757) -- and Fuse adds e2 * mvInvLevelSigma2[octave] > 5.99 (:1503-1509)
    int kf_area, loop_levels; const float* chi2_inv_sigma2;
};
struct ProjQueriesD {
    int n;
    const uint8_t* valid; const int32_t* cam; const float *u, *v, *radius; const int32_t *min_level, *max_level;
    const uint8_t* desc; const float* angle;
};

// A kernel whose first two arguments are the frame and query descriptors reads them through the kernarg segment, where they are used: as
// by-value arguments they are 30 + 21 scalar registers loaded at the top and held to the end, and the register allocator spills
// (tools/check_codeobj.py: 18-36 spilled scalar registers per resolver before, none after).
struct ProjArgsFQ { ProjFrameD f; ProjQueriesD q; };
#define DCS_PROJ_ARGS_FROM_KERNARG(f, q)                                                                                                        \
    const __attribute__((address_space(4))) char* ka_ = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(ka_));                                                                                                               \
    const ProjFrameD& f = ((const ProjArgsFQ*)ka_)->f;                                                                                          \
    const ProjQueriesD& q = ((const ProjArgsFQ*)ka_)->q;

__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc)
{
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// Visits the window of query qi exactly like Frame::GetFeaturesInArea (Frame.cc:316-376) + the candidate loop of
// ORBmatcher.cc:581-603: cells ix (outer), iy (inner), entries in insertion order, 64 entries per round (lane = entry).
// For every round calls emit(pass, word, pos0): pass = this lane holds a candidate, word = packed (dist << 23 |
// octave << 19 | camera-local index), pos0 = candidates emitted before this round. `taken` is the byte map to honour.
// how the "already matched" byte of a feature is read: plain loads for a map nobody writes (k_proj_collect) or that lives
// in LDS, device-coherent loads for the resolver's working copy in HBM (the L1 of the CU may hold a stale line)
struct TakenPlain { const uint8_t* p; __device__ bool operator()(int g) const { return p[g] != 0; } };
struct TakenCoherent {
    const uint8_t* p;
    __device__ bool operator()(int g) const { return __hip_atomic_load(p + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
};

template <typename Taken, typename Emit>
__device__ __forceinline__ int proj_visit(const ProjFrameD& f, const ProjQueriesD& q, int qi, Taken taken, Emit emit)
{
    const int lane = threadIdx.x & 63;
    const int c = q.cam[qi];
    const float x = q.u[qi], y = q.v[qi], r = q.radius[qi];
    const int minLevel = q.min_level[qi], maxLevel = q.max_level[qi];
    const float mx = f.min_x[c], my = f.min_y[c], wi = f.w_inv[c], hi = f.h_inv[c];
    const int x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, mx), r), wi)));
    if (x0 >= DCS_GRID_COLS) return 0;
    const int x1 = min(DCS_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, mx), r), wi)));
    if (x1 < 0) return 0;
    const int y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, my), r), hi)));
    if (y0 >= DCS_GRID_ROWS) return 0;
    const int y1 = min(DCS_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, my), r), hi)));
    if (y1 < 0) return 0;
    const bool check_levels = (minLevel > 0) || (maxLevel >= 0);
    const int base = f.cam_off[c];
    const uint4* qp = reinterpret_cast<const uint4*>(q.desc + (size_t)qi * 32);
    const uint4 qa = qp[0], qb = qp[1];
    int count = 0;
    for (int ix = x0; ix <= x1; ++ix) {
        // the cells (ix, y0..y1) are adjacent in the CSR: one contiguous run of entries, already in (iy, j) order
        const int cell0 = (c * DCS_GRID_COLS + ix) * DCS_GRID_ROWS + y0;
        const int j0 = f.grid_off[cell0], j1 = f.grid_off[cell0 + (y1 - y0) + 1];
        for (int jb = j0; jb < j1; jb += 64) {
            const int j = jb + lane;
            bool pass = false;
            unsigned word = 0;
            if (j < j1) {
                const int local = f.grid_idx[j], g = base + local;
                const int oct = f.kp_octave[g];
                bool ok = true;
                if (f.loop_levels) ok = !(oct < minLevel || oct > maxLevel);
                else if (check_levels) ok = !(oct < minLevel) && !(maxLevel >= 0 && oct > maxLevel);
                const int gp = f.kf_area ? local : g;
                const float dx = __fsub_rn(f.kp_x[gp], x), dy = __fsub_rn(f.kp_y[gp], y);
                ok = ok && fabsf(dx) < r && fabsf(dy) < r && !taken(g);
                if (ok && f.chi2_inv_sigma2) {
                    const float ex = __fsub_rn(x, f.kp_x[g]), ey = __fsub_rn(y, f.kp_y[g]);
                    const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                    ok = !((double)__fmul_rn(e2, f.chi2_inv_sigma2[oct]) > 5.99);
                }
                if (ok) {
                    const uint4* tp = reinterpret_cast<const uint4*>(f.desc + (size_t)g * 32);
                    const uint4 ta = tp[0], tb = tp[1];
                    unsigned d = bcnt_acc(qa.x ^ ta.x, 0u);
                    d = bcnt_acc(qa.y ^ ta.y, d); d = bcnt_acc(qa.z ^ ta.z, d); d = bcnt_acc(qa.w ^ ta.w, d);
                    d = bcnt_acc(qb.x ^ tb.x, d); d = bcnt_acc(qb.y ^ tb.y, d); d = bcnt_acc(qb.z ^ tb.z, d); d = bcnt_acc(qb.w ^ tb.w, d);
                    pass = true;
                    word = (d << 23) | ((unsigned)oct << 19) | (unsigned)local;
                }
            }
            const unsigned long long m = __ballot(pass);
            emit(pass, word, count + __popcll(m & ((1ull << lane) - 1ull)));
            count += __popcll(m);
        }
    }
    return count;
}

__global__ __launch_bounds__(256) void k_proj_collect(ProjFrameD f, ProjQueriesD q, unsigned* __restrict__ cand, int32_t* __restrict__ cand_n)
{
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= q.n) return;
    if (!q.valid[qi]) { if ((threadIdx.x & 63) == 0) cand_n[qi] = 0; return; }
    unsigned* out = cand + (size_t)qi * kProjCap;
    const int n = proj_visit(f, q, qi, TakenPlain{f.taken}, [&](bool pass, unsigned word, int pos) { if (pass && pos < kProjCap) out[pos] = word; });
    if ((threadIdx.x & 63) == 0) cand_n[qi] = n;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, d));
    return v;
}

// one wave, queries in order. The live "taken" map sits in LDS when the frame has <= 64 K features (always, in practice);
// otherwise in HBM with device-coherent accesses.
template <bool LDS_MAP>
__global__ __launch_bounds__(64) void k_proj_resolve(ProjFrameD f_arg, ProjQueriesD q_arg, const unsigned* __restrict__ cand,
                                                    const int32_t* __restrict__ cand_n, uint8_t* __restrict__ taken_hbm /* [N] working copy */,
                                                    int th_high, float nn_ratio, int check_ori, int32_t* __restrict__ match_of_query,
                                                    int32_t* __restrict__ query_of_feature, int32_t* __restrict__ bin_of_query,
                                                    int32_t* __restrict__ n_matches)
{
    DCS_PROJ_ARGS_FROM_KERNARG(f, q)
    __shared__ int s_hist[kHisto];
    __shared__ int s_ind[3];
    extern __shared__ uint8_t s_taken[];
    const int lane = threadIdx.x;
    for (int i = lane; i < f.N; i += 64) { query_of_feature[i] = -1; if (LDS_MAP) s_taken[i] = f.taken[i]; }
    if (lane < kHisto) s_hist[lane] = 0;
    __syncthreads();
    uint8_t* const taken = LDS_MAP ? s_taken : taken_hbm;
    auto is_taken = [&](int g) -> bool {
        if (LDS_MAP) return taken[g] != 0;
        return __hip_atomic_load(taken + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    };
    int nmatches = 0;
    for (int qi = 0; qi < q.n; ++qi) {
        int n = cand_n[qi];                                   // wave-uniform
        unsigned best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;    // (dist << 23 | octave << 19 | local) of the two winners
        if (n > 0 && n <= kProjCap) {
            unsigned w = 0, key = 0xFFFFFFFFu;
            if (lane < n) {
                w = cand[(size_t)qi * kProjCap + lane];
                const int g = f.cam_off[q.cam[qi]] + (int)(w & 0x7FFFFu);
                if (!is_taken(g)) key = ((w >> 23) << 8) | (unsigned)lane;         // distance, then visiting order
            }
            const unsigned k1 = wave_min_u32(key);
            if (k1 != 0xFFFFFFFFu) {
                const unsigned k2 = wave_min_u32(key == k1 ? 0xFFFFFFFFu : key);
                best = (unsigned)__builtin_amdgcn_readlane((int)w, (int)(k1 & 63u));
                if (k2 != 0xFFFFFFFFu) second = (unsigned)__builtin_amdgcn_readlane((int)w, (int)(k2 & 63u));
            }
        } else if (n > kProjCap) {                           // window wider than the list: walk it again, honouring the live map
            unsigned long long b1 = ~0ull, b2 = ~0ull;       // (dist << 40 | position << 8... ) keys: dist, then visiting position
            unsigned w1 = 0, w2 = 0;
            auto track = [&](bool pass, unsigned word, int pos) {
                const unsigned long long key = pass ? (((unsigned long long)(word >> 23) << 32) | (unsigned)pos) : ~0ull;
                const bool lt1 = key < b1, lt2 = key < b2;             // (selects, not branches: as if / else-if the compiler chose between the ADDRESSES of the four values and kept them in scratch memory)
                b2 = lt1 ? b1 : (lt2 ? key : b2); w2 = lt1 ? w1 : (lt2 ? word : w2);
                b1 = lt1 ? key : b1; w1 = lt1 ? word : w1;
            };
            if (LDS_MAP) (void)proj_visit(f, q, qi, TakenPlain{taken}, track);
            else (void)proj_visit(f, q, qi, TakenCoherent{taken}, track);
            // merge the per-lane (best, second) pairs: two smallest keys of the wave
            unsigned long long m1 = b1;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m1, d); m1 = o < m1 ? o : m1; }
            unsigned long long c2 = (b1 == m1) ? b2 : b1, m2 = c2;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m2, d); m2 = o < m2 ? o : m2; }
            if (m1 != ~0ull) {
                const unsigned long long has1 = __ballot(b1 == m1);
                best = (unsigned)__builtin_amdgcn_readlane((int)w1, __ffsll((long long)has1) - 1);
                if (m2 != ~0ull) {
                    const unsigned mine = (b1 == m2) ? w1 : w2;
                    const unsigned long long has2 = __ballot(b1 == m2 || b2 == m2);
                    second = (unsigned)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)has2) - 1);
                }
            }
        }
        int matched = -1;
        if (best != 0xFFFFFFFFu) {
            const int bestDist = (int)(best >> 23), bestLevel = (int)((best >> 19) & 15u);
            const int bestDist2 = second != 0xFFFFFFFFu ? (int)(second >> 23) : 256;
            const int bestLevel2 = second != 0xFFFFFFFFu ? (int)((second >> 19) & 15u) : -1;
            if (bestDist <= th_high &&
                !(nn_ratio > 0.f && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nn_ratio, (float)bestDist2)))
                matched = f.cam_off[q.cam[qi]] + (int)(best & 0x7FFFFu);
        }
        if (lane == 0) {
            match_of_query[qi] = matched;
            if (matched >= 0) {
                if (LDS_MAP) taken[matched] = 1; else __hip_atomic_store(taken + matched, (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                query_of_feature[matched] = qi;
                if (check_ori) {                              // :1072-1084
                    float rot = __fsub_rn(q.angle[qi], f.kp_angle[matched]);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bin = (int)roundf(__fmul_rn(rot, 1.0f / kHisto));
                    if (bin == kHisto) bin = 0;
                    bin_of_query[qi] = bin;
                    ++s_hist[bin];
                }
            }
        }
        nmatches += matched >= 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the taken byte must be visible to the next query's lanes
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    if (check_ori) {
        __syncthreads();
        if (lane == 0) {                                      // ComputeThreeMaxima (:1969-2010)
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < kHisto; ++i) {
                const int s = s_hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
        }
        __syncthreads();
        int removed = 0;
        for (int qi = lane; qi < q.n; qi += 64) {
            const int g = match_of_query[qi];
            if (g < 0) continue;
            const int b = bin_of_query[qi];
            if (b != s_ind[0] && b != s_ind[1] && b != s_ind[2]) { match_of_query[qi] = -1; query_of_feature[g] = -1; ++removed; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) removed += __shfl_xor(removed, d);
        nmatches -= removed;
    }
    if (lane == 0) *n_matches = nmatches;
}

// Parallel resolver with the same order-dependent result: thread = query, rounds until every query is final.
// A query may be decided once no EARLIER undecided query can still take a candidate its decision depends on: per round every
// undecided query posts its index with atomicMin on the features of its list (minq), and the queries that own their best live
// candidate (and, under the ratio test, their second: minq == own index) are decided together -- two of them never take the same
// feature, none can be overtaken by an earlier one, and the earliest undecided query always qualifies, so the rounds terminate.
// (Round 6: "owns EVERY live candidate" before; and the first kResOwn queries of a thread now keep their state and their first kResReg
// candidate words in registers -- a round touched global memory five times per query, 4.5 us a round, 76 us for a motion-model frame.) Windows of different map points rarely overlap, so a frame takes a
// handful of rounds instead of one step per query. A query whose window overflowed its list is decided alone by wave 0
// (re-walk of the window) when it becomes the earliest undecided one, and blocks the later ones until then.
constexpr int kResT = 1024;
constexpr int kResMaxN = 16384;                 // features whose minq / taken maps fit in LDS (80 KB)
constexpr int kResOwn = 4, kResReg = 4;         // queries per thread whose state stays on chip (the first 4 096 with anything to decide): index, list length and base in registers, the first kResReg candidate words in LDS (64 KB; 2 x 8 measured the same at 2 000 queries and slower beyond)

__device__ __forceinline__ void proj_resolve_par_body(const ProjFrameD& f, const ProjQueriesD& q, const unsigned* __restrict__ cand,
                                                      const int32_t* __restrict__ cand_n, uint8_t* __restrict__ state /* [n] 0 = undecided */,
                                                      int th_high, float nn_ratio, int check_ori, int32_t* __restrict__ match_of_query,
                                                      int32_t* __restrict__ query_of_feature, int32_t* __restrict__ bin_of_query,
                                                      int32_t* __restrict__ n_matches, int cam_filter = -1)
{
    // cam_filter >= 0 (round 5): only the queries and the features of ONE camera -- ORBmatcher::SearchByProjection(Fcur, Flast, th) is a loop of
    // SearchByProjectionOnCam calls (ORBmatcher.cc:954-1113), each with a rotation histogram of its own and each confined to its camera's features;
    // the workgroups of a frame's cameras then run side by side and touch disjoint entries of every array.
    __shared__ int s_minq[kResMaxN];
    __shared__ uint8_t s_taken[kResMaxN];
    __shared__ unsigned s_w[kResReg][kResOwn * kResT];        // candidate word k of query qi < kResOwn * kResT at [k][qi]: a thread's reads fall on its own bank
    __shared__ int s_hist[kHisto], s_ind[3];
    __shared__ int s_first[2], s_firstovf[2], s_undecided, s_nm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int f_lo = cam_filter >= 0 ? f.cam_off[cam_filter] : 0, f_hi = cam_filter >= 0 ? f.cam_off[cam_filter + 1] : f.N;
    auto mine = [&](int qi) { return cam_filter < 0 || q.cam[qi] == cam_filter; };
    for (int i = f_lo + tid; i < f_hi; i += kResT) { s_taken[i] = f.taken[i]; query_of_feature[i] = -1; }
    // The queries that have anything to decide -- this camera's, with a non-empty list -- are counted off in index order; the first kResOwn * kResT of
    // them keep their state on chip: thread t owns ranks t, t + kResT (query index, list length and base in registers; n = -1: none / decided), the
    // first kResReg words of their lists in LDS at [k][rank]. Later ones (a local map of many thousand points IN VIEW) stay in the global arrays.
    // (Round 6, late: the on-chip slots went to the first 2 048 query INDICES before -- of a 6 000-point local map two thirds of the visible points
    // sat behind them, on the slow path: 60 us instead of 23; 148 instead of 40 at 12 000.)
    int* const s_act = reinterpret_cast<int*>(&s_w[0][0]);        // (rank k's query index: read by the thread that then writes s_w[.][k], nobody else)
    constexpr int kResChunks = 64;                              // chunks of kResT queries whose activity bits a thread holds (65 536 queries)
    __shared__ int s_pre[kResChunks * (kResT / 64)];            // active queries per (chunk, wave), then their exclusive prefix in index order
    __shared__ int s_part[kResT / 64];
    __shared__ int s_run, s_rest;
    const int chunks = (q.n + kResT - 1) / kResT, wave_id = tid >> 6;
    if (tid == 0) { s_run = 0; s_rest = q.n; }
    // (three barriers per chunk before: a 12 000-point local map spent a dozen barrier-and-load round trips here. Now: every thread reads the
    // activity of ALL its queries, the per-(chunk, wave) counts are scanned as one table, and the ranks follow from the ballots.)
    unsigned long long act_bits = 0;                            // bit c: query c * kResT + tid has something to decide
    if (chunks <= kResChunks) {
#pragma unroll 8
        for (int c = 0; c < chunks; ++c) {                      // (nothing in this loop waits for a load but the bit it sets: eight chunks' loads fly together)
            const int qi = c * kResT + tid;
            const bool m = qi < q.n && mine(qi);
            const int n = m ? cand_n[qi] : 0;
            if (m) { match_of_query[qi] = -1; if (n <= 0) state[qi] = 1; }   // (empty window / invalid: decided)
            if (n > 0) act_bits |= 1ull << c;
        }
        for (int c = 0; c < chunks; ++c) {
            const unsigned long long bal = __ballot((act_bits >> c) & 1ull);
            if (lane == 0) s_pre[c * (kResT / 64) + wave_id] = __popcll(bal);
        }
        __syncthreads();
        {   // exclusive scan of the table (chunk-major, wave-minor = query index order), one entry per thread
            const int n_ent = chunks * (kResT / 64);
            const int v = tid < n_ent ? s_pre[tid] : 0;
            int inc = v;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const int t = __shfl_up(inc, dd); if (lane >= dd) inc += t; }
            if (lane == 63) s_part[wave_id] = inc;
            __syncthreads();
            int before = inc - v;
            for (int w = 0; w < wave_id; ++w) before += s_part[w];
            if (tid < n_ent) s_pre[tid] = before;
            if (tid == n_ent - 1) s_run = before + v;
        }
        __syncthreads();
        for (int c = 0; c < chunks; ++c) {
            const bool act = (act_bits >> c) & 1ull;
            const unsigned long long bal = __ballot(act);
            if (!act) continue;
            const int qi = c * kResT + tid, r = s_pre[c * (kResT / 64) + wave_id] + __popcll(bal & ((1ull << lane) - 1ull));
            if (r < kResOwn * kResT) s_act[r] = qi;
            else { state[qi] = 0; atomicMin(&s_rest, qi); }
        }
    } else {                                                    // more queries than the bits hold (never seen): everything on the global path
        for (int qi = tid; qi < q.n; qi += kResT) if (mine(qi)) { match_of_query[qi] = -1; state[qi] = cand_n[qi] == 0; }
        if (tid == 0) s_rest = 0;
    }
    __syncthreads();
    const int n_on = min(s_run, kResOwn * kResT), rest = s_rest;      // on-chip ranks [0, n_on); the global path takes the queries from index `rest` on
    int own_n[kResOwn], own_base[kResOwn], own_qi[kResOwn];
#pragma unroll
    for (int u = 0; u < kResOwn; ++u) {
        const int k = tid + u * kResT;
        own_n[u] = -1; own_base[u] = 0; own_qi[u] = 0x7FFFFFFF;
        if (k < n_on) {
            const int qi = s_act[k];
            own_qi[u] = qi; own_n[u] = cand_n[qi]; own_base[u] = f.cam_off[q.cam[qi]];
            // the first kResReg words of the query's 64-word row, two 16-byte loads (words past the list's end are never read back)
            static_assert(kResReg == 4 && kProjCap % 4 == 0, "one uint4 per row");
            const uint4 a = *reinterpret_cast<const uint4*>(cand + (size_t)qi * kProjCap);
            s_w[0][k] = a.x; s_w[1][k] = a.y; s_w[2][k] = a.z; s_w[3][k] = a.w;
        }
    }
    if (tid < kHisto) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    auto decide = [&](int qi, int base, unsigned best, unsigned second) {      // ORBmatcher.cc:606-613 / :1066-1084 (the rotation histogram: after the rounds)
        int matched = -1;
        if (best != 0xFFFFFFFFu) {
            const int bestDist = (int)(best >> 23), bestLevel = (int)((best >> 19) & 15u);
            const int bestDist2 = second != 0xFFFFFFFFu ? (int)(second >> 23) : 256;
            const int bestLevel2 = second != 0xFFFFFFFFu ? (int)((second >> 19) & 15u) : -1;
            if (bestDist <= th_high &&
                !(nn_ratio > 0.f && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nn_ratio, (float)bestDist2)))
                matched = base + (int)(best & 0x7FFFFu);
        }
        match_of_query[qi] = matched;
        if (matched >= 0) { s_taken[matched] = 1; query_of_feature[matched] = qi; }
    };
    // A round = rank + post | barrier | check + decide | barrier. s_minq is never reset: a post carries the round in its upper bits, counting DOWN, so
    // that this round's posts win every atomicMin against older ones (a feature nobody posts on this round keeps a stale word nobody reads); the two
    // "earliest undecided" words alternate between two copies, the one of the next round being reset between this round's barriers. (Round 6: three
    // barriers, a reset pass and every list read twice per round before: 4 500 cycles a round, now ~1 500.)
    struct Two { unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, w1 = 0, w2 = 0; };
    auto rank = [&](Two& t, unsigned w, int k, bool live) {
        const unsigned key = live ? ((w >> 23) << 8) | (unsigned)k : 0xFFFFFFFFu;           // (dist, position): strict "<" keeps the first of equal distances
        const bool lt1 = key < t.k1, lt2 = key < t.k2;
        t.k2 = lt1 ? t.k1 : (lt2 ? key : t.k2); t.w2 = lt1 ? t.w1 : (lt2 ? w : t.w2);
        t.k1 = lt1 ? key : t.k1; t.w1 = lt1 ? w : t.w1;
    };
    for (int i = f_lo + tid; i < f_hi; i += kResT) s_minq[i] = 0x7FFFFFFF;
    if (tid == 0) { s_first[0] = s_first[1] = 0x7FFFFFFF; s_firstovf[0] = s_firstovf[1] = 0x7FFFFFFF; s_undecided = 0; }
    __syncthreads();
    const bool tagged = q.n <= 0x7FFF;                         // (a round decides at least one query: at most q.n rounds, and a query index fits the lower 16 bits)
    for (int round = 0;; ++round) {
        const int par = round & 1, tag = tagged ? (0x7FFF - round) << 16 : 0;
        if (!tagged) {                                          // more queries than the tag holds (never seen): the reset pass of before
            __syncthreads();
            for (int i = f_lo + tid; i < f_hi; i += kResT) s_minq[i] = 0x7FFFFFFF;
            __syncthreads();
        }
        Two own_t[kResOwn];
        {   // the earliest undecided query: one atomic per wave
            int mine_first = 0x7FFFFFFF;
#pragma unroll
            for (int u = kResOwn - 1; u >= 0; --u) if (own_n[u] >= 0) mine_first = own_qi[u];
            const unsigned wf = wave_min_u32((unsigned)mine_first);
            if (lane == 0 && wf != 0x7FFFFFFFu) atomicMin(&s_first[par], (int)wf);
        }
#pragma unroll
        for (int u = 0; u < kResOwn; ++u) {
            const int qi = own_qi[u], n = own_n[u], slot = tid + u * kResT;
            if (n < 0) continue;
            if (n > kProjCap) { atomicMin(&s_firstovf[par], qi); continue; }
            {   // (loads first, all of them, then the tests, then the atomics: one LDS latency per stage instead of three per candidate)
                unsigned w[kResReg]; int g[kResReg]; bool lv[kResReg];
#pragma unroll
                for (int k = 0; k < kResReg; ++k) w[k] = s_w[k][slot];
#pragma unroll
                for (int k = 0; k < kResReg; ++k) g[k] = k < n ? own_base[u] + (int)(w[k] & 0x7FFFFu) : f_lo;
#pragma unroll
                for (int k = 0; k < kResReg; ++k) lv[k] = !s_taken[g[k]] && k < n;
#pragma unroll
                for (int k = 0; k < kResReg; ++k) if (lv[k]) atomicMin(&s_minq[g[k]], tag | qi);
#pragma unroll
                for (int k = 0; k < kResReg; ++k) rank(own_t[u], w[k], k, lv[k]);
            }
            for (int k = kResReg; k < n; ++k) {
                const unsigned w = cand[(size_t)qi * kProjCap + k];
                const int g = own_base[u] + (int)(w & 0x7FFFFu);
                const bool live = !s_taken[g];
                if (live) atomicMin(&s_minq[g], tag | qi);
                rank(own_t[u], w, k, live);
            }
        }
        for (int qi = rest + tid; qi < q.n; qi += kResT) {
            if (!mine(qi) || state[qi]) continue;
            atomicMin(&s_first[par], qi);
            const int n = cand_n[qi];
            if (n > kProjCap) { atomicMin(&s_firstovf[par], qi); continue; }
            const int base = f.cam_off[q.cam[qi]];
            for (int k = 0; k < n; ++k) { const int g = base + (int)(cand[(size_t)qi * kProjCap + k] & 0x7FFFFu); if (!s_taken[g]) atomicMin(&s_minq[g], tag | qi); }
        }
        __syncthreads();
        const int first = s_first[par], first_ovf = s_firstovf[par];
        if (tid == 0) { s_first[par ^ 1] = 0x7FFFFFFF; s_firstovf[par ^ 1] = 0x7FFFFFFF; }     // (the next round's copies: nobody touches them before the barrier below)
        if (first == 0x7FFFFFFF) break;                        // everything decided
        // final as soon as the candidates the decision READS cannot be taken before this query's turn: the live set only shrinks, so the best live
        // candidate stays the best if no earlier undecided query lists it, and with it the decision when there is no ratio test
        // (SearchByProjectionOnCam); the ratio test also reads the second
        auto settle = [&](const Two& t, int qi, int base) -> bool {
            bool safe = t.k1 == 0xFFFFFFFFu || s_minq[base + (int)(t.w1 & 0x7FFFFu)] == (tag | qi);
            if (safe && nn_ratio > 0.f && t.k2 != 0xFFFFFFFFu) safe = s_minq[base + (int)(t.w2 & 0x7FFFFu)] == (tag | qi);
            if (safe) decide(qi, base, t.k1 != 0xFFFFFFFFu ? t.w1 : 0xFFFFFFFFu, t.k2 != 0xFFFFFFFFu ? t.w2 : 0xFFFFFFFFu);
            return safe;
        };
#pragma unroll
        for (int u = 0; u < kResOwn; ++u) {
            const int qi = own_qi[u], n = own_n[u];
            if (n < 0 || n > kProjCap || qi > first_ovf) continue;
            if (settle(own_t[u], qi, own_base[u])) own_n[u] = -1;
        }
        for (int qi = rest + tid; qi < q.n; qi += kResT) {
            if (!mine(qi) || state[qi]) continue;
            const int n = cand_n[qi];
            if (n > kProjCap || qi > first_ovf) continue;
            const int base = f.cam_off[q.cam[qi]];
            Two t;
            for (int k = 0; k < n; ++k) { const unsigned w = cand[(size_t)qi * kProjCap + k]; rank(t, w, k, !s_taken[base + (int)(w & 0x7FFFFu)]); }
            if (settle(t, qi, base)) state[qi] = 1;
        }
        __syncthreads();
        if (first_ovf == first) {                               // the earliest undecided query has an oversized window: wave 0 walks it
            if (tid < 64) {
                const int qi = first_ovf;
                unsigned long long b1 = ~0ull, b2 = ~0ull;
                unsigned w1 = 0, w2 = 0;
                (void)proj_visit(f, q, qi, TakenPlain{s_taken}, [&](bool pass, unsigned word, int pos) {
                    const unsigned long long key = pass ? (((unsigned long long)(word >> 23) << 32) | (unsigned)pos) : ~0ull;
                    const bool lt1 = key < b1, lt2 = key < b2;             // (selects, not branches: as if / else-if the compiler chose between the ADDRESSES of the four values and kept them in scratch memory)
                    b2 = lt1 ? b1 : (lt2 ? key : b2); w2 = lt1 ? w1 : (lt2 ? word : w2);
                    b1 = lt1 ? key : b1; w1 = lt1 ? word : w1;
                });
                unsigned long long m1 = b1;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m1, d); m1 = o < m1 ? o : m1; }
                const unsigned long long c2 = (b1 == m1) ? b2 : b1;
                unsigned long long m2 = c2;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m2, d); m2 = o < m2 ? o : m2; }
                unsigned best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
                if (m1 != ~0ull) {
                    const unsigned long long has1 = __ballot(b1 == m1);
                    best = (unsigned)__builtin_amdgcn_readlane((int)w1, __ffsll((long long)has1) - 1);
                    if (m2 != ~0ull) {
                        const unsigned mine = (b1 == m2) ? w1 : w2;
                        const unsigned long long has2 = __ballot(b1 == m2 || b2 == m2);
                        second = (unsigned)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)has2) - 1);
                    }
                }
                if (lane == 0) { decide(qi, f.cam_off[q.cam[qi]], best, second); if (qi >= rest) state[qi] = 1; }
            }
#pragma unroll
            for (int u = 0; u < kResOwn; ++u) if (own_qi[u] == first_ovf) own_n[u] = -1;             // (its owner: decided)
            __syncthreads();
        }
    }
    // the matches and, with the orientation check, their rotation histogram (:1072-1084): counted here, once, for every match -- inside the rounds
    // the two angle loads of a decided query were on every round's critical path
    // (only queries that had something to decide can hold a match: a thread's on-chip ones -- their results read back together -- and the global
    // path's from index `rest` on; a walk over every query of a 12 000-point local map was a dozen dependent global loads)
    __syncthreads();
    int my_matches = 0;
    auto count = [&](int qi, int g) {
        if (g < 0) return;
        ++my_matches;
        if (check_ori) {
            float rot = __fsub_rn(q.angle[qi], f.kp_angle[g]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / kHisto));
            if (bin == kHisto) bin = 0;
            bin_of_query[qi] = bin;
            atomicAdd(&s_hist[bin], 1);
        }
    };
    int own_g[kResOwn];
#pragma unroll
    for (int u = 0; u < kResOwn; ++u) own_g[u] = own_qi[u] != 0x7FFFFFFF ? match_of_query[own_qi[u]] : -1;
#pragma unroll
    for (int u = 0; u < kResOwn; ++u) count(own_qi[u], own_g[u]);
    for (int qi = rest + tid; qi < q.n; qi += kResT) if (mine(qi)) count(qi, match_of_query[qi]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) my_matches += __shfl_xor(my_matches, d);
    if (lane == 0 && my_matches) atomicAdd(&s_nm, my_matches);
    __syncthreads();
    int nm = s_nm;
    if (check_ori) {
        if (tid == 0) {                                        // ComputeThreeMaxima (:1969-2010)
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < kHisto; ++i) {
                const int sz = s_hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
            s_undecided = 0;                                    // reused: number of removed matches
        }
        __syncthreads();
        auto prune = [&](int qi, int g) {
            if (g < 0) return;
            const int b = bin_of_query[qi];
            if (b != s_ind[0] && b != s_ind[1] && b != s_ind[2]) { match_of_query[qi] = -1; query_of_feature[g] = -1; atomicAdd(&s_undecided, 1); }
        };
#pragma unroll
        for (int u = 0; u < kResOwn; ++u) prune(own_qi[u], own_g[u]);
        for (int qi = rest + tid; qi < q.n; qi += kResT) if (mine(qi)) prune(qi, match_of_query[qi]);
        __syncthreads();
        nm -= s_undecided;
    }
    if (tid == 0) *n_matches = nm;
}
__global__ __launch_bounds__(kResT) void k_proj_resolve_par(ProjFrameD f_arg, ProjQueriesD q_arg, const unsigned* __restrict__ cand,
                                                           const int32_t* __restrict__ cand_n, uint8_t* __restrict__ state, int th_high, float nn_ratio,
                                                           int check_ori, int32_t* __restrict__ match_of_query, int32_t* __restrict__ query_of_feature,
                                                           int32_t* __restrict__ bin_of_query, int32_t* __restrict__ n_matches)
{
    DCS_PROJ_ARGS_FROM_KERNARG(f, q)
    proj_resolve_par_body(f, q, cand, cand_n, state, th_high, nn_ratio, check_ori, match_of_query, query_of_feature, bin_of_query, n_matches);
}

// ---- Frame::isInFrustum + PredictScale + search window, one lane per map point (Frame.cc:244-312, MapPoint.cc:440-455,
// ORBmatcher.cc:65-71, 557-565). Same arithmetic as the reference's cv::Mat expressions (see oracle/match_oracle.cpp): float dot
// products left to right without contraction, the translation added in double, norm / dot accumulated in double.
// Window searches whose queries do not see each other (Fuse x2, SearchBySim3CrossCam, SearchByProjection(KF, vpMapPoints, ...)):
// one wave per query walks the window like the reference's loop and keeps the smallest (distance, visiting position) key -- strict
// `dist < bestDist` updates keep the FIRST of equal distances.
__global__ __launch_bounds__(256) void k_window_static(ProjFrameD f, ProjQueriesD q, int th, int32_t* __restrict__ match_of_query,
                                                        int32_t* __restrict__ best_dist, int32_t* __restrict__ n_acc)
{
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= q.n) return;
    int matched = -1, bd = 256;
    if (q.valid[qi]) {
        unsigned long long b1 = ~0ull;
        unsigned w1 = 0;
        (void)proj_visit(f, q, qi, TakenPlain{f.taken}, [&](bool pass, unsigned word, int pos) {
            const unsigned long long key = pass ? (((unsigned long long)(word >> 23) << 32) | (unsigned)pos) : ~0ull;
            if (key < b1) { b1 = key; w1 = word; }
        });
        unsigned long long m1 = b1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m1, d); m1 = o < m1 ? o : m1; }
        if (m1 != ~0ull) {
            const unsigned long long has = __ballot(b1 == m1);
            const unsigned best = (unsigned)__builtin_amdgcn_readlane((int)w1, __ffsll((long long)has) - 1);
            bd = (int)(best >> 23);
            if (bd <= th) matched = f.cam_off[q.cam[qi]] + (int)(best & 0x7FFFFu);
        }
    }
    if (lane == 0) {
        match_of_query[qi] = matched; best_dist[qi] = bd;
        if (matched >= 0) atomicAdd(n_acc, 1);
    }
}

// SearchForInitialization (ORBmatcher.cc:1117-1251): one wave, queries in order. A candidate is skipped while an earlier query
// holds its feature with a distance <= this one's (vMatchedDistance, :1176); an accepted query takes the feature away from its
// previous owner (vnMatches21, :1194-1198). Candidate lists (distance, octave, feature) come from k_proj_collect (nothing taken).
// The distance map lives in LDS as u16 (0xFFFF = INT_MAX) for frames of <= 30 000 features, else in HBM.
template <bool LDS_MAP>
__global__ __launch_bounds__(64) void k_init_resolve(ProjFrameD f_arg, ProjQueriesD q_arg, const unsigned* __restrict__ cand, const int32_t* __restrict__ cand_n,
                                                    uint16_t* __restrict__ md_hbm /* [N] */, float nn_ratio, int check_ori,
                                                    int32_t* __restrict__ match12, int32_t* __restrict__ owner /* [N] vnMatches21 */,
                                                    int32_t* __restrict__ bin_of_query, int32_t* __restrict__ n_matches)
{
    DCS_PROJ_ARGS_FROM_KERNARG(f, q)
    __shared__ int s_hist[kHisto];
    __shared__ int s_ind[3];
    extern __shared__ uint16_t s_md[];
    const int lane = threadIdx.x;
    uint16_t* const md = LDS_MAP ? s_md : md_hbm;
    for (int i = lane; i < f.N; i += 64) { owner[i] = -1; md[i] = 0xFFFFu; }
    for (int i = lane; i < q.n; i += 64) { match12[i] = -1; bin_of_query[i] = -1; }
    if (lane < kHisto) s_hist[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    auto held = [&](int g) -> unsigned {
        if (LDS_MAP) return md[g];
        return __hip_atomic_load(md + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (int qi = 0; qi < q.n; ++qi) {
        const int n = cand_n[qi];                            // wave-uniform (0 for invalid queries)
        if (n == 0) continue;
        const int base = f.cam_off[q.cam[qi]];
        unsigned best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;
        if (n <= kProjCap) {
            unsigned w = 0, key = 0xFFFFFFFFu;
            if (lane < n) {
                w = cand[(size_t)qi * kProjCap + lane];
                const int g = base + (int)(w & 0x7FFFFu);
                if (!(held(g) <= (w >> 23))) key = ((w >> 23) << 8) | (unsigned)lane;      // vMatchedDistance[g] <= dist: skip
            }
            const unsigned k1 = wave_min_u32(key);
            if (k1 != 0xFFFFFFFFu) {
                const unsigned k2 = wave_min_u32(key == k1 ? 0xFFFFFFFFu : key);
                best = (unsigned)__builtin_amdgcn_readlane((int)w, (int)(k1 & 63u));
                if (k2 != 0xFFFFFFFFu) second = (unsigned)__builtin_amdgcn_readlane((int)w, (int)(k2 & 63u));
            }
        } else {                                             // window wider than the list: walk it again
            unsigned long long b1 = ~0ull, b2 = ~0ull;
            unsigned w1 = 0, w2 = 0;
            (void)proj_visit(f, q, qi, TakenPlain{f.taken}, [&](bool pass, unsigned word, int pos) {
                if (pass && held(base + (int)(word & 0x7FFFFu)) <= (word >> 23)) pass = false;
                const unsigned long long key = pass ? (((unsigned long long)(word >> 23) << 32) | (unsigned)pos) : ~0ull;
                const bool lt1 = key < b1, lt2 = key < b2;             // (selects, not branches: as if / else-if the compiler chose between the ADDRESSES of the four values and kept them in scratch memory)
                b2 = lt1 ? b1 : (lt2 ? key : b2); w2 = lt1 ? w1 : (lt2 ? word : w2);
                b1 = lt1 ? key : b1; w1 = lt1 ? word : w1;
            });
            unsigned long long m1 = b1;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m1, d); m1 = o < m1 ? o : m1; }
            unsigned long long c2 = (b1 == m1) ? b2 : b1, m2 = c2;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m2, d); m2 = o < m2 ? o : m2; }
            if (m1 != ~0ull) {
                const unsigned long long has1 = __ballot(b1 == m1);
                best = (unsigned)__builtin_amdgcn_readlane((int)w1, __ffsll((long long)has1) - 1);
                if (m2 != ~0ull) {
                    const unsigned mine = (b1 == m2) ? w1 : w2;
                    const unsigned long long has2 = __ballot(b1 == m2 || b2 == m2);
                    second = (unsigned)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)has2) - 1);
                }
            }
        }
        if (best != 0xFFFFFFFFu) {
            const int bestDist = (int)(best >> 23);
            const float d2 = second != 0xFFFFFFFFu ? (float)(int)(second >> 23) : 2147483648.0f;      // (float)INT_MAX
            if (bestDist <= 50 && (float)bestDist < __fmul_rn(d2, nn_ratio)) {                       // TH_LOW, :1190-1192
                const int g = base + (int)(best & 0x7FFFFu);
                if (lane == 0) {
                    const int prev = owner[g];
                    if (prev >= 0) match12[prev] = -1;
                    match12[qi] = g; owner[g] = qi;
                    if (LDS_MAP) md[g] = (uint16_t)bestDist; else __hip_atomic_store(md + g, (uint16_t)bestDist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (check_ori) {
                        float rot = __fsub_rn(q.angle[qi], f.kp_angle[g]);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, 1.0f / kHisto));
                        if (bin == kHisto) bin = 0;
                        bin_of_query[qi] = bin;
                        ++s_hist[bin];                       // a robbed query stays in the histogram (:1206 vs :1194-1198)
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    // the rotation filter removes the LIVE matches outside the three maxima (:1236-1240)
    if (check_ori) {
        if (lane == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < kHisto; ++i) {
                const int sgm = s_hist[i];
                if (sgm > max1) { max3 = max2; max2 = max1; max1 = sgm; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sgm > max2) { max3 = max2; max2 = sgm; ind3 = ind2; ind2 = i; }
                else if (sgm > max3) { max3 = sgm; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
        }
        __syncthreads();
        for (int qi = lane; qi < q.n; qi += 64) {
            const int b = bin_of_query[qi];
            if (b >= 0 && match12[qi] >= 0 && b != s_ind[0] && b != s_ind[1] && b != s_ind[2]) match12[qi] = -1;
        }
        __syncthreads();
    }
    // nmatches of the reference = accepted - robbed - filtered = the queries that still hold a feature
    int live = 0;
    for (int qi = lane; qi < q.n; qi += 64) live += match12[qi] >= 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) live += __shfl_xor(live, d);
    if (lane == 0) *n_matches = live;
}

constexpr int kFrMaxCams = 8;
struct FrustumDev {
    int n_cams, n_levels;
    float R[kFrMaxCams][9], t[kFrMaxCams][3], O[kFrMaxCams][3];
    float fx[kFrMaxCams], fy[kFrMaxCams], cx[kFrMaxCams], cy[kFrMaxCams], min_x[kFrMaxCams], max_x[kFrMaxCams], min_y[kFrMaxCams], max_y[kFrMaxCams];
    float log_scale;
    const float* scale_factors;
};

__device__ __forceinline__ void frustum_point(const FrustumDev& F, int i, const float* __restrict__ pos, const float* __restrict__ normal,
                                              const float* __restrict__ min_dist, const float* __restrict__ max_dist,
                                              const uint8_t* __restrict__ candidate, float cos_limit, float th, uint8_t* __restrict__ in_view,
                                              int32_t* __restrict__ cam, float* __restrict__ u_out, float* __restrict__ v_out,
                                              float* __restrict__ view_cos, int32_t* __restrict__ level, float* __restrict__ radius)
{
    uint8_t ok = 0; int c_out = -1, lvl = 0; float uo = 0, vo = 0, vc = 0, rad = 0;
    if (!candidate || candidate[i]) {
        const float P0 = pos[3 * i], P1 = pos[3 * i + 1], P2 = pos[3 * i + 2];
        const float N0 = normal[3 * i], N1 = normal[3 * i + 1], N2 = normal[3 * i + 2];
        const float mind = __fmul_rn(0.8f, min_dist[i]), maxd = __fmul_rn(1.2f, max_dist[i]), maxd_raw = max_dist[i];
        for (int ic = 0; ic < F.n_cams; ++ic) {
            float Pic[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float t0 = __fadd_rn(__fadd_rn(__fmul_rn(F.R[ic][3 * r], P0), __fmul_rn(F.R[ic][3 * r + 1], P1)), __fmul_rn(F.R[ic][3 * r + 2], P2));
                Pic[r] = (float)((double)t0 + (double)F.t[ic][r]);
            }
            if (Pic[2] < 0.0f) continue;
            const float invz = __fdiv_rn(1.0f, Pic[2]);
            const float u = __fadd_rn(__fmul_rn(__fmul_rn(F.fx[ic], Pic[0]), invz), F.cx[ic]);
            const float v = __fadd_rn(__fmul_rn(__fmul_rn(F.fy[ic], Pic[1]), invz), F.cy[ic]);
            if (u < F.min_x[ic] || u > F.max_x[ic]) continue;
            if (v < F.min_y[ic] || v > F.max_y[ic]) continue;
            const float d0 = __fsub_rn(P0, F.O[ic][0]), d1 = __fsub_rn(P1, F.O[ic][1]), d2 = __fsub_rn(P2, F.O[ic][2]);
            const double s = __dadd_rn(__dadd_rn(__dmul_rn((double)d0, (double)d0), __dmul_rn((double)d1, (double)d1)), __dmul_rn((double)d2, (double)d2));
            const float dist = (float)sqrt(s);
            if (dist < mind || dist > maxd) continue;
            const double dot = __dadd_rn(__dadd_rn(__dmul_rn((double)d0, (double)N0), __dmul_rn((double)d1, (double)N1)), __dmul_rn((double)d2, (double)N2));
            const float viewCos = (float)(dot / (double)dist);
            if (viewCos < cos_limit) continue;
            const float ratio = __fdiv_rn(maxd_raw, dist);
            const float lg = (float)log((double)ratio);                   // correctly rounded logf for all but ~2^-29 of the inputs
            int nScale = (int)ceilf(__fdiv_rn(lg, F.log_scale));
            nScale = nScale < 0 ? 0 : (nScale >= F.n_levels ? F.n_levels - 1 : nScale);
            float r = ((double)viewCos > 0.998) ? 2.5f : 4.0f;
            if (th != 1.0f) r = __fmul_rn(r, th);
            ok = 1; c_out = ic; uo = u; vo = v; vc = viewCos; lvl = nScale; rad = __fmul_rn(r, F.scale_factors[nScale]);
            break;
        }
    }
    in_view[i] = ok; cam[i] = c_out; u_out[i] = uo; v_out[i] = vo; if (view_cos) view_cos[i] = vc; level[i] = lvl; radius[i] = rad;
}
__global__ __launch_bounds__(256) void k_frustum(FrustumDev F, int n, const float* __restrict__ pos, const float* __restrict__ normal,
                                                 const float* __restrict__ min_dist, const float* __restrict__ max_dist,
                                                 const uint8_t* __restrict__ candidate, float cos_limit, float th, uint8_t* __restrict__ in_view,
                                                 int32_t* __restrict__ cam, float* __restrict__ u_out, float* __restrict__ v_out,
                                                 float* __restrict__ view_cos, int32_t* __restrict__ level, float* __restrict__ radius)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) frustum_point(F, i, pos, normal, min_dist, max_dist, candidate, cos_limit, th, in_view, cam, u_out, v_out, view_cos, level, radius);
}

// ---- the tracking chain, batched over frames (dcs_track_local_map): blockIdx.y = frame, every per-frame pointer in a TrackItem in HBM
struct TrackItem {
    FrustumDev F;
    ProjFrameD f; ProjQueriesD q;                  // q.* = what the frustum stage writes (the mutable aliases follow)
    int n_points;
    const float *pos, *normal, *min_dist, *max_dist; const uint8_t* candidate;
    uint8_t* q_valid; int32_t *q_cam, *q_level, *q_min, *q_max; float *q_u, *q_v, *q_radius;
    unsigned* cand; int32_t* cand_n; uint8_t* state; int32_t *mq, *qf, *bin, *nm;
    const uint8_t* has_point; const float* point_xw;
    int edge_base;                                  // first edge slot of this frame (= its first feature's global number in the batch)
    int32_t *edge_feature, *point_of_feature; uint8_t* feat_outlier;
};
__device__ __forceinline__ void track_frustum_query(const TrackItem& it, int i, float cos_limit, float th)
{
    if (i >= it.n_points) return;
    frustum_point(it.F, i, it.pos, it.normal, it.min_dist, it.max_dist, it.candidate, cos_limit, th, it.q_valid, it.q_cam, it.q_u, it.q_v, nullptr, it.q_level, it.q_radius);
    const int lvl = it.q_level[i];
    it.q_min[i] = lvl - 1; it.q_max[i] = lvl + 1;                    // ORBmatcher.cc:567-569: GetFeaturesInArea(..., nPredictedLevel - 1, nPredictedLevel + 1)
}
__global__ __launch_bounds__(256) void k_track_frustum(const TrackItem* __restrict__ items, float cos_limit, float th)
{
    track_frustum_query(items[blockIdx.y], blockIdx.x * 256 + threadIdx.x, cos_limit, th);
}
__global__ __launch_bounds__(256) void k_track_collect(const TrackItem* __restrict__ items)
{
    const TrackItem& it = items[blockIdx.y];
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= it.q.n) return;
    if (!it.q.valid[qi]) { if ((threadIdx.x & 63) == 0) it.cand_n[qi] = 0; return; }
    unsigned* out = it.cand + (size_t)qi * kProjCap;
    const int n = proj_visit(it.f, it.q, qi, TakenPlain{it.f.taken}, [&](bool pass, unsigned word, int pos) { if (pass && pos < kProjCap) out[pos] = word; });
    if ((threadIdx.x & 63) == 0) it.cand_n[qi] = n;
}
__global__ __launch_bounds__(kResT) void k_track_resolve(const TrackItem* __restrict__ items, int th_high, float nn_ratio)
{
    const TrackItem& it = items[blockIdx.x];
    proj_resolve_par_body(it.f, it.q, it.cand, it.cand_n, it.state, th_high, nn_ratio, 0, it.mq, it.qf, it.bin, it.nm);
}
// Optimizer::PoseOptimization's edge list (Optimizer.cc:288-350): every feature that holds a map point, ascending feature index -- the
// point the search has just assigned (ORBmatcher.cc:617: F.mvpMapPoints[bestIdx] = pMP), else the one it held before the call.
constexpr int kEdgesT = 1024;                   // (round 6: 256 before -- a 2 096-slot frame took nine passes of three barriers each, 16 us; now three)
__global__ __launch_bounds__(kEdgesT) void k_track_edges(const TrackItem* __restrict__ items, const float* __restrict__ inv_level_sigma2, int n_levels,
                                                     double* __restrict__ xw, double* __restrict__ obs, double* __restrict__ w, int32_t* __restrict__ ecam,
                                                     int32_t* __restrict__ edge_cnt)
{
    __shared__ int s_wave[kEdgesT / 64];
    __shared__ int s_run;
    const TrackItem& it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, N = it.f.N;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += kEdgesT) {
        const int i = i0 + tid;
        int src = -1;                                              // >= 0: local map point of the new match, -2: the point held before, -1: none
        if (i < N) {
            const int qn = it.qf[i];
            src = qn >= 0 ? qn : (it.has_point[i] ? -2 : -1);
            it.point_of_feature[i] = src;
            it.feat_outlier[i] = 0;
        }
        const bool has = src != -1;
        const unsigned long long m = __ballot(has);
        const int rank_w = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = s_run;
        for (int k = 0; k < wave; ++k) before += s_wave[k];
        if (has) {
            const int e = it.edge_base + before + rank_w;
            const float* X = src >= 0 ? it.pos + 3 * (size_t)src : it.point_xw + 3 * (size_t)i;
            xw[3 * (size_t)e] = (double)X[0]; xw[3 * (size_t)e + 1] = (double)X[1]; xw[3 * (size_t)e + 2] = (double)X[2];   // Converter::toVector3d
            obs[2 * (size_t)e] = (double)it.f.kp_x[i]; obs[2 * (size_t)e + 1] = (double)it.f.kp_y[i];
            const int oct = min(max(it.f.kp_octave[i], 0), n_levels - 1);
            w[e] = (double)inv_level_sigma2[oct];
            int c = 0;
            while (c + 1 < it.f.n_cams && i >= it.f.cam_off[c + 1]) ++c;                                                  // keypointToCam
            ecam[e] = c;
            it.edge_feature[before + rank_w] = i;
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int k = 0; k < kEdgesT / 64; ++k) t += s_wave[k]; s_run += t; }
        __syncthreads();
    }
    if (tid == 0) edge_cnt[blockIdx.x] = s_run;
}
__global__ __launch_bounds__(256) void k_track_finish(const TrackItem* __restrict__ items, const int32_t* __restrict__ edge_cnt, const uint8_t* __restrict__ outlier)
{
    const TrackItem& it = items[blockIdx.x];
    for (int k = threadIdx.x; k < edge_cnt[blockIdx.x]; k += 256) it.feat_outlier[it.edge_feature[k]] = outlier[it.edge_base + k];   // mvbOutlier[i]
}

// cv::undistortPoints(src, dst, K, distCoeffs, cv::Mat(), K) for ONE point, as Frame::UndistortKeyPoints calls it (src/Frame.cc:410-441): OpenCV 3.3 / 3.4.0
// cvUndistortPoints (modules/imgproc/src/undistort.cpp) in its own order of double operations -- normalise with ifx = 1 / fx, the identity tilt
// compensation, five fixed-point iterations of the radial / tangential model, re-projection with RR = K (the zero products stay in: they add exact zeros).
// c = {fx, fy, cx, cy, k1, k2, p1, p2, k3}. Built without contraction (-ffp-contract=off) on both sides of the compiler.
__host__ __device__ inline void undistort_point(const double c[9], float u, float v, float& uo, float& vo)
{
    const double fx = c[0], fy = c[1], cx = c[2], cy = c[3], k0 = c[4], k1 = c[5], k2 = c[6], k3 = c[7], k4 = c[8];
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = u, y = v;
    x = (x - cx) * ifx; y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0.0 * r2 + 0.0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    uo = (float)(xx * ww); vo = (float)(yy * ww);
}

// ---- device-resident frames (dcs_track_frame_device): what the Frame constructor does with the extractor's output (src/Frame.cc:141-196)
struct DevAsm {                                    // one per frame: where the extractor left the features, and the assembled arrays
    const dcs_keypoint* kp; const uint8_t* desc; const int32_t* n; int cap, first_slot, n_cams;
    int32_t* cam_off; float *kp_x, *kp_y, *kp_angle; int32_t* kp_octave; uint8_t* desc_out; int32_t *grid_off, *grid_idx;
    const float *min_x, *min_y, *w_inv, *h_inv;    // device copies of the per-camera grid constants
    int32_t* n_features;                           // [n_cams] output
    int32_t* n_matches;                            // [kFrMaxCams] the frame's match counters (TrackItem::nm): zeroed by k_dev_assemble
    const int32_t *q_cam_in, *q_octave_in;         // mode 1: camera and octave of every query (the last frame's key point)
    double und[kFrMaxCams][9];                     // {fx, fy, cx, cy, k1, k2, p1, p2, k3} per camera
    uint8_t und_on[kFrMaxCams];                    // Frame::UndistortKeyPoints: distCoef[0] != 0 (Frame.cc:414)
};
// mvTotalKeysUn / mvDescriptors concatenated over the cameras (Frame.cc:166-181): feature i of camera c at cam_off[c] + i
__global__ __launch_bounds__(256) void k_dev_assemble(const DevAsm* __restrict__ das)
{
    const DevAsm& d = das[blockIdx.y];
    int off[kFrMaxCams + 1];
    off[0] = 0;
    for (int c = 0; c < d.n_cams; ++c) off[c + 1] = off[c] + min(max(d.n[d.first_slot + c], 0), d.cap);      // (a negative count is the extractor's error code: no features)
    if (blockIdx.x == 0 && threadIdx.x <= d.n_cams) {
        d.cam_off[threadIdx.x] = off[threadIdx.x];
        if (threadIdx.x < d.n_cams) d.n_features[threadIdx.x] = off[threadIdx.x + 1] - off[threadIdx.x];
    }
    if (blockIdx.x == 0 && threadIdx.x < kFrMaxCams) d.n_matches[threadIdx.x] = 0;
    const int t = blockIdx.x * 256 + threadIdx.x;               // thread = (feature slot, 8-byte piece of its descriptor): 4 threads per feature
    const int s = t >> 2, piece = t & 3;
    // the key-point arrays are read up to the frame's CAPACITY (all the host knows): the slots behind the last real feature are zeros. Written here,
    // by the threads that have no feature to copy (round 6: two hipMemsetAsync in front of the upload before -- each a dispatch of its own with
    // ~10 us of idle queue around it, 25 us of a 0.45-ms frame)
    if (piece == 0 && s >= off[d.n_cams] && s < d.n_cams * d.cap) { d.kp_x[s] = 0.f; d.kp_y[s] = 0.f; d.kp_angle[s] = 0.f; d.kp_octave[s] = 0; }
    const int c = s / d.cap, i = s - c * d.cap;
    if (c >= d.n_cams || i >= off[c + 1] - off[c]) return;
    const size_t src = (size_t)(d.first_slot + c) * d.cap + i;
    const int g = off[c] + i;
    reinterpret_cast<unsigned long long*>(d.desc_out + (size_t)g * 32)[piece] = reinterpret_cast<const unsigned long long*>(d.desc + src * 32)[piece];
    if (piece == 0) {
        const dcs_keypoint k = d.kp[src];
        float ux = k.x, uy = k.y;
        if (d.und_on[c]) undistort_point(d.und[c], k.x, k.y, ux, uy);            // mvvkeysUnTemp (Frame.cc:410-441)
        d.kp_x[g] = ux; d.kp_y[g] = uy; d.kp_angle[g] = k.angle; d.kp_octave[g] = k.octave;
    }
}
// The geometry of SearchByProjectionOnCam (ORBmatcher.cc:962-968, 990-1036) for every query (a feature of the last frame with a good map
// point): x3Ds = Rsw x3Dw + tsw in the reference's cv::Mat arithmetic (float dot product left to right, the translation added through
// double: the small-matrix path of cv::gemm, as in frustum_point), zs < 0 -> skip, invzs = 1.0 / zs (a DOUBLE division narrowed to float,
// :1003), u, v, the image-bounds test, radius = th * mvScaleFactors[last octave], levels octave - 1 .. octave + 1.
__device__ __forceinline__ void track_mm_query(const TrackItem& it, const DevAsm& d, int i, float th)
{
    if (i >= it.n_points) return;
    const FrustumDev& F = it.F;
    const int c = d.q_cam_in[i], oct = min(max(d.q_octave_in[i], 0), F.n_levels - 1);
    const float P0 = it.pos[3 * i], P1 = it.pos[3 * i + 1], P2 = it.pos[3 * i + 2];
    float Ps[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float t0 = __fadd_rn(__fadd_rn(__fmul_rn(F.R[c][3 * r], P0), __fmul_rn(F.R[c][3 * r + 1], P1)), __fmul_rn(F.R[c][3 * r + 2], P2));
        Ps[r] = (float)((double)t0 + (double)F.t[c][r]);
    }
    uint8_t ok = 0;
    float u = 0, v = 0;
    if (!(Ps[2] < 0.0f) && Ps[2] != 0.0f) {                       // (zs == 0 would project to infinity: treated as not visible)
        const float invz = (float)(1.0 / (double)Ps[2]);
        u = __fadd_rn(__fmul_rn(__fmul_rn(F.fx[c], Ps[0]), invz), F.cx[c]);
        v = __fadd_rn(__fmul_rn(__fmul_rn(F.fy[c], Ps[1]), invz), F.cy[c]);
        ok = !(u < F.min_x[c] || u > F.max_x[c]) && !(v < F.min_y[c] || v > F.max_y[c]);
    }
    it.q_valid[i] = ok; it.q_cam[i] = c; it.q_u[i] = u; it.q_v[i] = v;
    it.q_radius[i] = __fmul_rn(th, F.scale_factors[oct]);
    it.q_level[i] = oct; it.q_min[i] = d.q_octave_in[i] - 1; it.q_max[i] = d.q_octave_in[i] + 1;
}
__global__ __launch_bounds__(256) void k_track_mm_queries(const TrackItem* __restrict__ items, const DevAsm* __restrict__ das, float th)
{
    track_mm_query(items[blockIdx.y], das[blockIdx.y], blockIdx.x * 256 + threadIdx.x, th);
}
// Frame::PosInGrid + the grid fill (Frame.cc:183-190, 380-390) as the CSR dcs_frame_grid builds on the host: cells (c, ix, iy), entries =
// camera-local indices in insertion (= ascending) order. One workgroup per frame: counts in LDS, scan, placement, then every cell orders
// its (few) entries.
constexpr int kGridT = 1024;
// (round 6: the launch also carries the frame's QUERIES -- blocks y >= 1, a thread per map point: isInFrustum (mode 0) or SearchByProjectionOnCam's
// geometry (mode 1). They depend on nothing the grid build does; as a launch of their own they were 5-8 us plus a kernel boundary on a 0.39-ms chain.)
__global__ __launch_bounds__(kGridT) void k_dev_grid(const DevAsm* __restrict__ das, int cells_max, const TrackItem* __restrict__ items, int mode, float cos_limit, float th)
{
    if (blockIdx.y > 0) {
        const int i = ((int)blockIdx.y - 1) * kGridT + (int)threadIdx.x;
        if (mode == 0) track_frustum_query(items[blockIdx.x], i, cos_limit, th);
        else track_mm_query(items[blockIdx.x], das[blockIdx.x], i, th);
        return;
    }
    extern __shared__ int s_dyn[];                              // [cells_max] counts, then write cursors; [capacity] the cells' entries until they are ordered
    __shared__ int s_scan[kGridT / 64];
    const DevAsm& d = das[blockIdx.x];
    int* const s_cell = s_dyn;
    int* const s_idx = s_dyn + cells_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cells = d.n_cams * DCS_GRID_COLS * DCS_GRID_ROWS, N = d.cam_off[d.n_cams];
    for (int i = tid; i < cells; i += kGridT) s_cell[i] = 0;
    __syncthreads();
    auto cell_of = [&](int g, int c) {
        const int px = __float2int_rn(__fmul_rn(__fsub_rn(d.kp_x[g], d.min_x[c]), d.w_inv[c]));              // cvRound: round half to even
        const int py = __float2int_rn(__fmul_rn(__fsub_rn(d.kp_y[g], d.min_y[c]), d.h_inv[c]));
        return (px < 0 || px >= DCS_GRID_COLS || py < 0 || py >= DCS_GRID_ROWS) ? -1 : (c * DCS_GRID_COLS + px) * DCS_GRID_ROWS + py;
    };
    auto cam_of = [&](int g) { int c = 0; while (c + 1 < d.n_cams && g >= d.cam_off[c + 1]) ++c; return c; };
    // (round 6: a thread keeps the cell and the local index of its first two features -- every feature of a frame of up to 2 048 -- from the count to
    // the placement, and the entries are placed and ordered in LDS: the cells' insertion sorts ran on global memory, a round trip per step, 20 us)
    int my_cl[2] = {-1, -1}, my_loc[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int g = tid + u * kGridT;
        if (g < N) { const int c = cam_of(g); my_cl[u] = cell_of(g, c); my_loc[u] = g - d.cam_off[c]; if (my_cl[u] >= 0) atomicAdd(&s_cell[my_cl[u]], 1); }
    }
    for (int g = tid + 2 * kGridT; g < N; g += kGridT) { const int cl = cell_of(g, cam_of(g)); if (cl >= 0) atomicAdd(&s_cell[cl], 1); }
    __syncthreads();
    // exclusive scan over the cells: every thread owns a run of consecutive cells
    const int per = (cells + kGridT - 1) / kGridT, c0 = tid * per, c1 = min(c0 + per, cells);
    int sum = 0;
    for (int i = c0; i < c1; ++i) sum += s_cell[i];
    int inc = sum;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const int t = __shfl_up(inc, dd); if (lane >= dd) inc += t; }
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    int before = inc - sum;
    for (int w = 0; w < wave; ++w) before += s_scan[w];
    const int run_begin = before;                               // first entry of this thread's run of cells
    for (int i = c0; i < c1; ++i) { const int n = s_cell[i]; d.grid_off[i] = before; s_cell[i] = before; before += n; }
    if (tid == kGridT - 1) d.grid_off[cells] = before;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) if (my_cl[u] >= 0) s_idx[atomicAdd(&s_cell[my_cl[u]], 1)] = my_loc[u];
    for (int g = tid + 2 * kGridT; g < N; g += kGridT) {
        const int c = cam_of(g), cl = cell_of(g, c);
        if (cl >= 0) s_idx[atomicAdd(&s_cell[cl], 1)] = g - d.cam_off[c];
    }
    __syncthreads();
    for (int i = c0, b = run_begin; i < c1; ++i) {              // s_cell[i] is now the END of cell i; insertion sort of its entries (ascending local index)
        const int e = s_cell[i];
        for (int a = b + 1; a < e; ++a) {
            const int v = s_idx[a];
            int k = a - 1;
            while (k >= b && s_idx[k] > v) { s_idx[k + 1] = s_idx[k]; --k; }
            s_idx[k + 1] = v;
        }
        b = e;
    }
    __syncthreads();
    const int placed = s_cell[cells - 1];                       // (the last cell's end = entries placed: features outside the grid have none)
    for (int k = tid; k < placed; k += kGridT) d.grid_idx[k] = s_idx[k];
}
// mode 1: one workgroup per (camera, frame) -- the SearchByProjectionOnCam calls of SearchByProjection(Fcur, Flast, th)
__global__ __launch_bounds__(kResT) void k_track_resolve_cam(const TrackItem* __restrict__ items, int th_high, int check_ori)
{
    const TrackItem& it = items[blockIdx.y];
    if ((int)blockIdx.x >= it.f.n_cams) return;
    if (blockIdx.x == 0) for (int i = it.f.cam_off[it.f.n_cams] + (int)threadIdx.x; i < it.f.N; i += kResT) it.qf[i] = -1;      // the slots behind the last real feature (k_track_edges walks the capacity)
    proj_resolve_par_body(it.f, it.q, it.cand, it.cand_n, it.state, th_high, 0.f, check_ori, it.mq, it.qf, it.bin, it.nm + blockIdx.x, (int)blockIdx.x);
}



}  // namespace
}  // namespace dcs

using namespace dcs;

extern "C" {

int dcs_frame_grid(int n_cams, const int32_t* cam_off, const float* kp_x, const float* kp_y, const float* min_x, const float* min_y,
                   const float* grid_w_inv, const float* grid_h_inv, int32_t* grid_off, int32_t* grid_idx, int* n_entries)
{
    if (n_cams < 1 || !cam_off || !min_x || !min_y || !grid_w_inv || !grid_h_inv || !grid_off || (cam_off[n_cams] > 0 && (!kp_x || !kp_y || !grid_idx))) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    const int cells = n_cams * DCS_GRID_COLS * DCS_GRID_ROWS, N = cam_off[n_cams];
    std::vector<int32_t> cell_of((size_t)std::max(N, 1), -1), cnt((size_t)cells + 1, 0);
    for (int c = 0; c < n_cams; ++c)
        for (int i = cam_off[c]; i < cam_off[c + 1]; ++i) {
            const int px = (int)nearbyintf((kp_x[i] - min_x[c]) * grid_w_inv[c]);      // cvRound: round half to even
            const int py = (int)nearbyintf((kp_y[i] - min_y[c]) * grid_h_inv[c]);
            if (px < 0 || px >= DCS_GRID_COLS || py < 0 || py >= DCS_GRID_ROWS) continue;   // PosInGrid (Frame.cc:386-388)
            cell_of[i] = (c * DCS_GRID_COLS + px) * DCS_GRID_ROWS + py;
            ++cnt[cell_of[i] + 1];
        }
    for (int k = 0; k < cells; ++k) cnt[k + 1] += cnt[k];
    for (int k = 0; k <= cells; ++k) grid_off[k] = cnt[k];
    std::vector<int32_t> cur(cnt.begin(), cnt.end() - 1);
    for (int c = 0; c < n_cams; ++c)
        for (int i = cam_off[c]; i < cam_off[c + 1]; ++i)
            if (cell_of[i] >= 0) grid_idx[cur[cell_of[i]]++] = i - cam_off[c];         // ascending i inside a cell = push_back order
    if (n_entries) *n_entries = cnt[cells];
    return DCS_OK;
}

// validation + upload shared by the window searches: a malformed grid would index features / LDS maps out of bounds on the device
struct ProjInputs { ProjFrameD f{}; ProjQueriesD q{}; int C = 0, N = 0, nq = 0; };
// the frame half of every projection search: validation of the caller's arrays (the kernels index with them) and upload
static int proj_frame_check(const dcs_proj_frame* fr, bool need_taken, bool need_angles)
{
    if (!fr || fr->n_cams < 1 || !fr->cam_off) { set_error("bad argument"); return DCS_ERR_INVALID; }
    const int C = fr->n_cams, N = fr->cam_off[C];
    const int cells = C * DCS_GRID_COLS * DCS_GRID_ROWS;
    if (N < 0 || N >= (1 << 19)) { set_error("feature count %d outside 0 .. 2^19", N); return DCS_ERR_UNSUPPORTED; }
    if (!fr->min_x || !fr->min_y || !fr->grid_w_inv || !fr->grid_h_inv || !fr->grid_off || (N && (!fr->kp_x || !fr->kp_y || !fr->kp_octave ||
        !fr->desc || (need_taken && !fr->taken) || !fr->grid_idx)) || (need_angles && N && !fr->kp_angle)) { set_error("null array"); return DCS_ERR_INVALID; }
    const int n_entries = fr->grid_off[cells];
    if (n_entries < 0 || n_entries > N) { set_error("grid CSR inconsistent"); return DCS_ERR_INVALID; }
    for (int c = 0; c <= C; ++c) if (fr->cam_off[c] < 0 || (c && fr->cam_off[c] < fr->cam_off[c - 1])) { set_error("cam_off not ascending"); return DCS_ERR_INVALID; }
    for (int c = 0; c < C; ++c) {                       // offsets ascending, every entry a local index of its camera
        const int n_cam = fr->cam_off[c + 1] - fr->cam_off[c], c0 = c * DCS_GRID_COLS * DCS_GRID_ROWS;
        for (int k = c0; k < c0 + DCS_GRID_COLS * DCS_GRID_ROWS; ++k) {
            const int a = fr->grid_off[k], b = fr->grid_off[k + 1];
            if (a < 0 || b < a || b > n_entries) { set_error("grid_off not ascending at cell %d", k); return DCS_ERR_INVALID; }
            for (int e = a; e < b; ++e)
                if (fr->grid_idx[e] < 0 || fr->grid_idx[e] >= n_cam) { set_error("grid_idx[%d] = %d outside camera %d (%d features)", e, fr->grid_idx[e], c, n_cam); return DCS_ERR_INVALID; }
        }
    }
    for (int i = 0; i < N; ++i) if (fr->kp_octave[i] < 0 || fr->kp_octave[i] > 15) { set_error("octave of feature %d outside 0..15", i); return DCS_ERR_INVALID; }
    return DCS_OK;
}
static int proj_frame_upload(Scratch& s, const dcs_proj_frame* fr, bool need_taken, bool need_angles, ProjFrameD& f)
{
    const int C = fr->n_cams, N = fr->cam_off[C];
    const int cells = C * DCS_GRID_COLS * DCS_GRID_ROWS, n_entries = fr->grid_off[cells];
    int rc;
    f.n_cams = C; f.N = N;
    static const float zero_f = 0.f;
    const std::vector<uint8_t> none(need_taken || fr->taken ? 0 : (size_t)std::max(N, 1), 0);        // "nothing taken" when the caller has no map
    if ((rc = s.upload(&f.cam_off, fr->cam_off, (size_t)C + 1)) || (rc = s.upload(&f.kp_x, fr->kp_x, (size_t)N)) || (rc = s.upload(&f.kp_y, fr->kp_y, (size_t)N)) ||
        (rc = s.upload(&f.kp_octave, fr->kp_octave, (size_t)N)) || (rc = s.upload(&f.kp_angle, need_angles ? fr->kp_angle : &zero_f, need_angles ? (size_t)N : 1)) ||
        (rc = s.upload(&f.desc, fr->desc, (size_t)N * 32)) || (rc = s.upload(&f.taken, fr->taken ? fr->taken : none.data(), (size_t)N)) ||
        (rc = s.upload(&f.min_x, fr->min_x, (size_t)C)) || (rc = s.upload(&f.min_y, fr->min_y, (size_t)C)) ||
        (rc = s.upload(&f.w_inv, fr->grid_w_inv, (size_t)C)) || (rc = s.upload(&f.h_inv, fr->grid_h_inv, (size_t)C)) ||
        (rc = s.upload(&f.grid_off, fr->grid_off, (size_t)cells + 1)) || (rc = s.upload(&f.grid_idx, fr->grid_idx, (size_t)n_entries))) return rc;
    return DCS_OK;
}

static int proj_prepare(Scratch& s, const dcs_proj_frame* fr, const dcs_proj_queries* qs, bool need_taken, bool need_angles, ProjInputs& in)
{
    if (!fr || !qs || fr->n_cams < 1 || !fr->cam_off || qs->n < 0) { set_error("bad argument"); return DCS_ERR_INVALID; }
    const int C = fr->n_cams, N = fr->cam_off[C], nq = qs->n;
    if ((nq && (!qs->valid || !qs->cam || !qs->u || !qs->v || !qs->radius || !qs->min_level || !qs->max_level || !qs->desc)) ||
        (need_angles && nq && !qs->angle)) { set_error("null array"); return DCS_ERR_INVALID; }
    for (int i = 0; i < nq; ++i) if (qs->valid[i] && (qs->cam[i] < 0 || qs->cam[i] >= C)) { set_error("query %d: camera out of range", i); return DCS_ERR_INVALID; }
    int rc;
    if ((rc = proj_frame_check(fr, need_taken, need_angles)) || (rc = ensure_device())) return rc;
    in.C = C; in.N = N; in.nq = nq;
    if (nq == 0) return DCS_OK;
    if ((rc = proj_frame_upload(s, fr, need_taken, need_angles, in.f))) return rc;
    ProjQueriesD& q = in.q;
    static const float zero_f = 0.f;
    q.n = nq;
    if ((rc = s.upload(&q.valid, qs->valid, (size_t)nq)) || (rc = s.upload(&q.cam, qs->cam, (size_t)nq)) || (rc = s.upload(&q.u, qs->u, (size_t)nq)) ||
        (rc = s.upload(&q.v, qs->v, (size_t)nq)) || (rc = s.upload(&q.radius, qs->radius, (size_t)nq)) ||
        (rc = s.upload(&q.min_level, qs->min_level, (size_t)nq)) || (rc = s.upload(&q.max_level, qs->max_level, (size_t)nq)) ||
        (rc = s.upload(&q.desc, qs->desc, (size_t)nq * 32)) || (rc = s.upload(&q.angle, need_angles ? qs->angle : &zero_f, need_angles ? (size_t)nq : 1))) return rc;
    return DCS_OK;
}

// kf_area: the window of KeyFrame::GetFeaturesInArea (camera-local index read as a global one, KeyFrame.cc:756; no level argument: the
// octave gate min <= octave <= max sits in the caller's loop, ORBmatcher.cc:510-513) instead of Frame::GetFeaturesInArea
static int search_by_projection_impl(const dcs_proj_frame* fr, const dcs_proj_queries* qs, int th_high, float nn_ratio, int check_orientation, int kf_area,
                                     int32_t* match_of_query, int32_t* query_of_feature, int* n_matches)
{
    if (!n_matches || (fr && fr->cam_off && fr->n_cams >= 1 && fr->cam_off[fr->n_cams] > 0 && !query_of_feature) || (qs && qs->n > 0 && !match_of_query)) {
        set_error("bad argument"); return DCS_ERR_INVALID;
    }
    Scratch s;
    ProjInputs in;
    int rc = proj_prepare(s, fr, qs, true, check_orientation != 0, in);
    if (rc) return rc;
    const int N = in.N, nq = in.nq;
    in.f.kf_area = kf_area != 0; in.f.loop_levels = kf_area != 0;
    const ProjFrameD& f = in.f;
    const ProjQueriesD& q = in.q;
    *n_matches = 0;
    if (nq == 0) { for (int i = 0; i < N; ++i) query_of_feature[i] = -1; return DCS_OK; }
    unsigned* d_cand; int32_t *d_cn, *d_mq, *d_qf, *d_bin, *d_nm; uint8_t* d_taken;
    if ((rc = s.alloc(&d_cand, (size_t)nq * kProjCap)) || (rc = s.alloc(&d_cn, (size_t)nq)) || (rc = s.alloc(&d_mq, (size_t)nq)) ||
        (rc = s.alloc(&d_qf, (size_t)N)) || (rc = s.alloc(&d_bin, (size_t)nq)) || (rc = s.alloc(&d_nm, 1)) || (rc = s.upload(&d_taken, fr->taken, (size_t)N))) return rc;
    hipLaunchKernelGGL(k_proj_collect, dim3((nq + 3) / 4), dim3(256), 0, s.st, f, q, d_cand, d_cn);
    DCS_CHECK_LAUNCH();
    if (N <= kResMaxN) {                               // (larger frames: the one-wave resolver, reference order step by step)
        uint8_t* d_state;
        if ((rc = s.alloc(&d_state, (size_t)nq))) return rc;
        hipLaunchKernelGGL(k_proj_resolve_par, dim3(1), dim3(kResT), 0, s.st, f, q, d_cand, d_cn, d_state, th_high, nn_ratio, check_orientation,
                           d_mq, d_qf, d_bin, d_nm);
    } else if (N <= 65536)
        hipLaunchKernelGGL(k_proj_resolve<true>, dim3(1), dim3(64), (size_t)std::max(N, 1), s.st, f, q, d_cand, d_cn, d_taken, th_high, nn_ratio,
                           check_orientation, d_mq, d_qf, d_bin, d_nm);
    else
        hipLaunchKernelGGL(k_proj_resolve<false>, dim3(1), dim3(64), 0, s.st, f, q, d_cand, d_cn, d_taken, th_high, nn_ratio, check_orientation,
                           d_mq, d_qf, d_bin, d_nm);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(match_of_query, d_mq, sizeof(int32_t) * nq))) return rc;
    if (N) if ((rc = s.download_bytes(query_of_feature, d_qf, sizeof(int32_t) * N))) return rc;
    int32_t nm = 0;
    if ((rc = s.download_bytes(&nm, d_nm, sizeof(int32_t)))) return rc;
    if ((rc = s.finish())) return rc;
    *n_matches = nm;
    return DCS_OK;
}

int dcs_search_by_projection(const dcs_proj_frame* fr, const dcs_proj_queries* qs, int th_high, float nn_ratio, int check_orientation,
                             int32_t* match_of_query, int32_t* query_of_feature, int* n_matches)
{
    return search_by_projection_impl(fr, qs, th_high, nn_ratio, check_orientation, 0, match_of_query, query_of_feature, n_matches);
}

int dcs_search_by_projection_kf(const dcs_proj_frame* fr, const dcs_proj_queries* qs, int th, int32_t* match_of_query, int32_t* query_of_feature, int* n_matches)
{
    return search_by_projection_impl(fr, qs, th, 0.f, 0, 1, match_of_query, query_of_feature, n_matches);
}

int dcs_search_in_window(const dcs_proj_frame* fr, const dcs_proj_queries* qs, int th, int kf_area, const float* chi2_inv_sigma2, int n_levels,
                         int32_t* match_of_query, int32_t* best_dist, int* n_matches)
{
    if (!n_matches || (qs && qs->n > 0 && !match_of_query) || (chi2_inv_sigma2 && (n_levels < 1 || n_levels > 16))) { set_error("bad argument"); return DCS_ERR_INVALID; }
    Scratch s;
    ProjInputs in;
    int rc = proj_prepare(s, fr, qs, false, false, in);
    if (rc) return rc;
    *n_matches = 0;
    const int nq = in.nq;
    if (nq == 0) return DCS_OK;
    if (chi2_inv_sigma2)
        for (int i = 0; i < in.N; ++i) if (fr->kp_octave[i] >= n_levels) { set_error("octave of feature %d beyond the %d sigma levels", i, n_levels); return DCS_ERR_INVALID; }
    in.f.kf_area = kf_area != 0; in.f.loop_levels = 1;
    if (chi2_inv_sigma2 && (rc = s.upload(&in.f.chi2_inv_sigma2, chi2_inv_sigma2, (size_t)n_levels))) return rc;
    int32_t *d_mq, *d_bd, *d_nm;
    if ((rc = s.alloc(&d_mq, (size_t)nq)) || (rc = s.alloc(&d_bd, (size_t)nq)) || (rc = s.alloc(&d_nm, 1))) return rc;
    DCS_HIP(hipMemsetAsync(d_nm, 0, sizeof(int32_t), s.st));
    hipLaunchKernelGGL(k_window_static, dim3((nq + 3) / 4), dim3(256), 0, s.st, in.f, in.q, th, d_mq, d_bd, d_nm);
    DCS_CHECK_LAUNCH();
    int32_t nm = 0;
    if ((rc = s.download_bytes(match_of_query, d_mq, sizeof(int32_t) * nq))) return rc;
    if (best_dist && (rc = s.download_bytes(best_dist, d_bd, sizeof(int32_t) * nq))) return rc;
    if ((rc = s.download_bytes(&nm, d_nm, sizeof(int32_t)))) return rc;
    if ((rc = s.finish())) return rc;
    *n_matches = nm;
    return DCS_OK;
}

int dcs_search_for_initialization(const dcs_proj_frame* f2, const dcs_proj_queries* qs, float nn_ratio, int check_orientation,
                                  int32_t* match12, int* n_matches)
{
    if (!n_matches || (qs && qs->n > 0 && !match12)) { set_error("bad argument"); return DCS_ERR_INVALID; }
    Scratch s;
    ProjInputs in;
    int rc = proj_prepare(s, f2, qs, false, check_orientation != 0, in);
    if (rc) return rc;
    *n_matches = 0;
    const int N = in.N, nq = in.nq;
    if (nq == 0) return DCS_OK;
    if (f2->taken) {                                     // SearchForInitialization has no "already matched" input: ignore the caller's map
        const std::vector<uint8_t> none((size_t)std::max(N, 1), 0);
        if ((rc = s.upload(&in.f.taken, none.data(), (size_t)N))) return rc;
    }
    unsigned* d_cand; int32_t *d_cn, *d_m12, *d_owner, *d_bin, *d_nm; uint16_t* d_md;
    if ((rc = s.alloc(&d_cand, (size_t)nq * kProjCap)) || (rc = s.alloc(&d_cn, (size_t)nq)) || (rc = s.alloc(&d_m12, (size_t)nq)) ||
        (rc = s.alloc(&d_owner, (size_t)N)) || (rc = s.alloc(&d_bin, (size_t)nq)) || (rc = s.alloc(&d_nm, 1)) || (rc = s.alloc(&d_md, (size_t)N))) return rc;
    hipLaunchKernelGGL(k_proj_collect, dim3((nq + 3) / 4), dim3(256), 0, s.st, in.f, in.q, d_cand, d_cn);
    DCS_CHECK_LAUNCH();
    if (N <= 30000)                                      // u16 distance map + the static arrays stay below 64 KB of LDS
        hipLaunchKernelGGL(k_init_resolve<true>, dim3(1), dim3(64), sizeof(uint16_t) * (size_t)std::max(N, 1), s.st, in.f, in.q, d_cand, d_cn, d_md, nn_ratio,
                           check_orientation, d_m12, d_owner, d_bin, d_nm);
    else
        hipLaunchKernelGGL(k_init_resolve<false>, dim3(1), dim3(64), 0, s.st, in.f, in.q, d_cand, d_cn, d_md, nn_ratio, check_orientation, d_m12, d_owner,
                           d_bin, d_nm);
    DCS_CHECK_LAUNCH();
    int32_t nm = 0;
    if ((rc = s.download_bytes(match12, d_m12, sizeof(int32_t) * nq))) return rc;
    if ((rc = s.download_bytes(&nm, d_nm, sizeof(int32_t)))) return rc;
    if ((rc = s.finish())) return rc;
    *n_matches = nm;
    return DCS_OK;
}

int dcs_is_in_frustum(const dcs_frustum_frame* f, int n, const float* pos, const float* normal, const float* min_dist, const float* max_dist,
                      const uint8_t* candidate, float viewing_cos_limit, float th, uint8_t* in_view, int32_t* cam, float* u, float* v,
                      float* view_cos, int32_t* level, float* radius)
{
    if (!f || n < 0 || f->n_cams < 1 || f->n_cams > kFrMaxCams || f->n_scale_levels < 1 || !f->Rsw || !f->tsw || !f->Ow || !f->fx || !f->fy || !f->cx ||
        !f->cy || !f->min_x || !f->max_x || !f->min_y || !f->max_y || !f->scale_factors ||
        (n && (!pos || !normal || !min_dist || !max_dist || !in_view || !cam || !u || !v || !view_cos || !level || !radius))) {
        set_error("dcs_is_in_frustum: bad argument"); return DCS_ERR_INVALID;
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) return DCS_OK;
    FrustumDev F{};
    F.n_cams = f->n_cams; F.n_levels = f->n_scale_levels; F.log_scale = f->log_scale_factor;
    for (int c = 0; c < f->n_cams; ++c) {
        for (int k = 0; k < 9; ++k) F.R[c][k] = f->Rsw[9 * c + k];
        for (int k = 0; k < 3; ++k) { F.t[c][k] = f->tsw[3 * c + k]; F.O[c][k] = f->Ow[3 * c + k]; }
        F.fx[c] = f->fx[c]; F.fy[c] = f->fy[c]; F.cx[c] = f->cx[c]; F.cy[c] = f->cy[c];
        F.min_x[c] = f->min_x[c]; F.max_x[c] = f->max_x[c]; F.min_y[c] = f->min_y[c]; F.max_y[c] = f->max_y[c];
    }
    Scratch s;
    const float *d_pos, *d_nrm, *d_min, *d_max, *d_sf;
    float *d_u, *d_v, *d_vc, *d_rad;
    const uint8_t* d_cand = nullptr;
    uint8_t* d_in;
    int32_t *d_cam, *d_lvl;
    if ((rc = s.upload(&d_pos, pos, (size_t)3 * n)) || (rc = s.upload(&d_nrm, normal, (size_t)3 * n)) || (rc = s.upload(&d_min, min_dist, n)) ||
        (rc = s.upload(&d_max, max_dist, n)) || (rc = s.upload(&d_sf, f->scale_factors, f->n_scale_levels)) || (candidate && (rc = s.upload(&d_cand, candidate, n))) ||
        (rc = s.alloc(&d_in, n)) || (rc = s.alloc(&d_cam, n)) || (rc = s.alloc(&d_u, n)) || (rc = s.alloc(&d_v, n)) || (rc = s.alloc(&d_vc, n)) ||
        (rc = s.alloc(&d_lvl, n)) || (rc = s.alloc(&d_rad, n))) return rc;
    F.scale_factors = d_sf;
    hipLaunchKernelGGL(k_frustum, dim3((n + 255) / 256), dim3(256), 0, s.st, F, n, d_pos, d_nrm, d_min, d_max, d_cand, viewing_cos_limit, th, d_in, d_cam, d_u, d_v,
                       d_vc, d_lvl, d_rad);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(in_view, d_in, n))) return rc;
    if ((rc = s.download_bytes(cam, d_cam, sizeof(int32_t) * n))) return rc;
    if ((rc = s.download_bytes(u, d_u, sizeof(float) * n))) return rc;
    if ((rc = s.download_bytes(v, d_v, sizeof(float) * n))) return rc;
    if ((rc = s.download_bytes(view_cos, d_vc, sizeof(float) * n))) return rc;
    if ((rc = s.download_bytes(level, d_lvl, sizeof(int32_t) * n))) return rc;
    if ((rc = s.download_bytes(radius, d_rad, sizeof(float) * n))) return rc;
    return s.finish();
}

// The steady-state per-frame chain of the Tracking thread -- SearchLocalPoints (src/Tracking.cc:1617-1680: isInFrustum for every local map
// point, then SearchByProjection) followed by PoseOptimization (:1321) -- for a BATCH of frames (one per stream / rig), device-resident:
// the frustum stage writes the queries of the search, the search's assignment becomes the optimiser's edge list, nothing returns to the
// host in between. Five launches for the whole batch (blockIdx.y / .x = frame) + k_pose_opt.
int dcs_track_local_map(int n_frames, const dcs_track_frame* frames, const dcs_track_params* prm, dcs_track_result* res)
{
    const int F = n_frames;
    if (F < 0 || (F && (!frames || !prm || !res || !res->poses || !res->n_inliers || !res->n_matches || !res->match_of_point || !res->point_of_feature ||
        !res->outlier)) || (F && (prm->n_levels < 1 || !prm->inv_level_sigma2 || !prm->cams))) { set_error("dcs_track_local_map: bad argument"); return DCS_ERR_INVALID; }
    if (F == 0) return DCS_OK;
    // the optimiser holds its cameras by value (DCams, ba_solver.hip): checked here, before anything is enqueued
    if (prm->n_cams < 1 || prm->n_cams > kPoseMaxCams) { set_error("dcs_track_local_map: prm->n_cams must be 1..%d", kPoseMaxCams); return DCS_ERR_INVALID; }
    int rc;
    long long n_feat = 0;
    int max_pts = 0, max_q4 = 0, max_feat = 0;
    for (int k = 0; k < F; ++k) {
        const dcs_track_frame& t = frames[k];
        const dcs_frustum_frame* v = &t.view;
        if ((rc = proj_frame_check(&t.features, true, false))) return rc;
        // k_track_edges labels an edge with its feature's camera: every one of them must have intrinsics in prm->cams
        if (t.features.n_cams > prm->n_cams) { set_error("dcs_track_local_map: frame %d has %d cameras, prm->cams holds %d", k, t.features.n_cams, prm->n_cams); return DCS_ERR_INVALID; }
        const int N = t.features.cam_off[t.features.n_cams];
        if (N > kResMaxN) { set_error("frame %d: %d features (the chain's resolver holds %d)", k, N, kResMaxN); return DCS_ERR_UNSUPPORTED; }
        if (t.n_points < 0 || !t.pose || v->n_cams != t.features.n_cams || v->n_cams > kFrMaxCams || v->n_scale_levels < 1 || !v->Rsw || !v->tsw || !v->Ow ||
            !v->fx || !v->fy || !v->cx || !v->cy || !v->min_x || !v->max_x || !v->min_y || !v->max_y || !v->scale_factors ||
            (t.n_points && (!t.pos || !t.normal || !t.min_dist || !t.max_dist || !t.desc)) || (N && (!t.point_xw || !(t.has_point || t.features.taken))) ||
            !res->match_of_point[k] || !res->point_of_feature[k] || !res->outlier[k]) { set_error("dcs_track_local_map: frame %d: bad argument", k); return DCS_ERR_INVALID; }
        n_feat += N; max_feat = std::max(max_feat, N);
        max_pts = std::max(max_pts, t.n_points); max_q4 = std::max(max_q4, (t.n_points + 3) / 4);
    }
    if ((rc = ensure_device())) return rc;
    Scratch s;
    std::vector<TrackItem> items((size_t)F);
    std::vector<int32_t> edge_off((size_t)F);
    const size_t Etot = (size_t)std::max<long long>(n_feat, 1);
    int base = 0;
    for (int k = 0; k < F; ++k) {
        const dcs_track_frame& t = frames[k];
        TrackItem& it = items[(size_t)k];
        const dcs_frustum_frame* v = &t.view;
        FrustumDev& Fd = it.F;
        Fd = FrustumDev{};
        Fd.n_cams = v->n_cams; Fd.n_levels = v->n_scale_levels; Fd.log_scale = v->log_scale_factor;
        for (int c = 0; c < v->n_cams; ++c) {
            for (int j = 0; j < 9; ++j) Fd.R[c][j] = v->Rsw[9 * c + j];
            for (int j = 0; j < 3; ++j) { Fd.t[c][j] = v->tsw[3 * c + j]; Fd.O[c][j] = v->Ow[3 * c + j]; }
            Fd.fx[c] = v->fx[c]; Fd.fy[c] = v->fy[c]; Fd.cx[c] = v->cx[c]; Fd.cy[c] = v->cy[c];
            Fd.min_x[c] = v->min_x[c]; Fd.max_x[c] = v->max_x[c]; Fd.min_y[c] = v->min_y[c]; Fd.max_y[c] = v->max_y[c];
        }
        if ((rc = s.upload(&Fd.scale_factors, v->scale_factors, (size_t)v->n_scale_levels))) return rc;
        if ((rc = proj_frame_upload(s, &t.features, true, false, it.f))) return rc;
        it.f.kf_area = 0; it.f.loop_levels = 0; it.f.chi2_inv_sigma2 = nullptr;
        const int N = it.f.N, np = t.n_points;
        it.n_points = np;
        const uint8_t* hp = t.has_point ? t.has_point : t.features.taken;
        // uploads copy the REAL counts (an empty local map / a frame without features may pass NULL arrays: upload() of 0 elements only reserves)
        const size_t npu = (size_t)np, Nu = (size_t)N;
        if ((rc = s.upload(&it.pos, t.pos, 3 * npu)) || (rc = s.upload(&it.normal, t.normal, 3 * npu)) || (rc = s.upload(&it.min_dist, t.min_dist, npu)) ||
            (rc = s.upload(&it.max_dist, t.max_dist, npu)) || (rc = s.upload(&it.has_point, hp, Nu)) || (rc = s.upload(&it.point_xw, t.point_xw, 3 * Nu))) return rc;
        it.candidate = nullptr;
        if (t.candidate && np && (rc = s.upload(&it.candidate, t.candidate, npu))) return rc;
        ProjQueriesD& q = it.q;
        q.n = np;
        if ((rc = s.upload(&q.desc, t.desc, 32 * npu))) return rc;
        q.angle = nullptr;
        it.edge_base = base; edge_off[(size_t)k] = base;
        base += N;
    }
    // Everything that goes UP is carved first and back to back (the frames' arrays above, then the four small tables, staged here and filled once the
    // device pointers they hold are known), everything that comes DOWN last and back to back: one DMA operation each way whatever the number of frames
    // (with the scratch arrays in between a batch of 16 frames cost ~100 copies of 4 us each: a third of its time)
    TrackItem* d_items; TrackItem* h_items; int32_t* d_edge_off; int32_t* h_edge_off; float* d_sig; float* h_sig; double* d_pose_in; double* h_pose_in;
    if ((rc = s.stage(&d_items, &h_items, (size_t)F)) || (rc = s.stage(&d_edge_off, &h_edge_off, (size_t)F)) || (rc = s.stage(&d_sig, &h_sig, (size_t)prm->n_levels)) ||
        (rc = s.stage(&d_pose_in, &h_pose_in, (size_t)7 * F))) return rc;
    for (int k = 0; k < F; ++k) {                             // scratch of the searches
        TrackItem& it = items[(size_t)k];
        const size_t npe = (size_t)std::max(it.n_points, 1), Ne = (size_t)std::max(it.f.N, 1);
        if ((rc = s.alloc(&it.q_valid, npe)) || (rc = s.alloc(&it.q_cam, npe)) || (rc = s.alloc(&it.q_level, npe)) || (rc = s.alloc(&it.q_min, npe)) ||
            (rc = s.alloc(&it.q_max, npe)) || (rc = s.alloc(&it.q_u, npe)) || (rc = s.alloc(&it.q_v, npe)) || (rc = s.alloc(&it.q_radius, npe)) ||
            (rc = s.alloc(&it.cand, npe * kProjCap)) || (rc = s.alloc(&it.cand_n, npe)) || (rc = s.alloc(&it.state, npe)) ||
            (rc = s.alloc(&it.qf, Ne)) || (rc = s.alloc(&it.bin, npe)) || (rc = s.alloc(&it.edge_feature, Ne))) return rc;
        ProjQueriesD& q = it.q;
        q.valid = it.q_valid; q.cam = it.q_cam; q.u = it.q_u; q.v = it.q_v; q.radius = it.q_radius; q.min_level = it.q_min; q.max_level = it.q_max;
    }
    double *d_xw, *d_obs, *d_w, *d_err, *d_out; int32_t *d_ecam, *d_cnt, *d_ninl; uint8_t *d_level, *d_outl;
    if ((rc = s.alloc(&d_xw, 3 * Etot)) || (rc = s.alloc(&d_obs, 2 * Etot)) || (rc = s.alloc(&d_w, Etot)) || (rc = s.alloc(&d_err, 2 * Etot)) ||
        (rc = s.alloc(&d_ecam, Etot)) || (rc = s.alloc(&d_cnt, (size_t)F)) || (rc = s.alloc(&d_level, Etot)) || (rc = s.alloc(&d_outl, Etot))) return rc;
    if ((rc = s.alloc(&d_out, (size_t)7 * F)) || (rc = s.alloc(&d_ninl, (size_t)F))) return rc;      // results: from here to the end of the arena
    for (int k = 0; k < F; ++k) {
        TrackItem& it = items[(size_t)k];
        const size_t npe = (size_t)std::max(it.n_points, 1), Ne = (size_t)std::max(it.f.N, 1);
        if ((rc = s.alloc(&it.mq, npe)) || (rc = s.alloc(&it.point_of_feature, Ne)) || (rc = s.alloc(&it.feat_outlier, Ne)) || (rc = s.alloc(&it.nm, 1))) return rc;
    }
    memcpy(h_items, items.data(), sizeof(TrackItem) * (size_t)F);
    memcpy(h_edge_off, edge_off.data(), sizeof(int32_t) * (size_t)F);
    memcpy(h_sig, prm->inv_level_sigma2, sizeof(float) * (size_t)prm->n_levels);
    for (int k = 0; k < F; ++k) memcpy(h_pose_in + (size_t)7 * k, frames[k].pose, sizeof(double) * 7);
    hipStream_t st = s.st;
    if (max_pts > 0) {
        hipLaunchKernelGGL(k_track_frustum, dim3((max_pts + 255) / 256, F), dim3(256), 0, st, d_items, prm->viewing_cos_limit, prm->th);
        hipLaunchKernelGGL(k_track_collect, dim3(max_q4, F), dim3(256), 0, st, d_items);
    }
    hipLaunchKernelGGL(k_track_resolve, dim3(F), dim3(kResT), 0, st, d_items, prm->th_high, prm->nn_ratio);
    hipLaunchKernelGGL(k_track_edges, dim3(F), dim3(kEdgesT), 0, st, d_items, d_sig, prm->n_levels, d_xw, d_obs, d_w, d_ecam, d_cnt);
    DCS_CHECK_LAUNCH();
    PoseOptDevice po{};
    po.poses = d_pose_in; po.edge_off = d_edge_off; po.edge_cnt = d_cnt; po.xw = d_xw; po.obs = d_obs; po.w = d_w; po.cam = d_ecam;
    po.huber = prm->huber_delta;
    for (int i = 0; i < 4; ++i) { po.chi2_th[i] = prm->chi2_th[i]; po.its[i] = prm->its[i]; }
    po.err = d_err; po.level = d_level; po.out_poses = d_out; po.outlier = d_outl; po.n_inliers = d_ninl; po.edge_chi2 = nullptr; po.n_iters = nullptr;
    if ((rc = launch_pose_opt_device(po, prm->cams, prm->n_cams, F, max_feat, st))) return rc;      // a frame's edges <= its features
    hipLaunchKernelGGL(k_track_finish, dim3(F), dim3(256), 0, st, d_items, (const int32_t*)d_cnt, (const uint8_t*)d_outl);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(res->poses, d_out, sizeof(double) * 7 * F)) || (rc = s.download_bytes(res->n_inliers, d_ninl, sizeof(int32_t) * F))) return rc;
    for (int k = 0; k < F; ++k) {
        const TrackItem& it = items[(size_t)k];
        if (it.n_points && (rc = s.download_bytes(res->match_of_point[k], it.mq, sizeof(int32_t) * it.n_points))) return rc;
        if (it.f.N && ((rc = s.download_bytes(res->point_of_feature[k], it.point_of_feature, sizeof(int32_t) * it.f.N)) ||
                       (rc = s.download_bytes(res->outlier[k], it.feat_outlier, (size_t)it.f.N)))) return rc;
        if ((rc = s.download_bytes(&res->n_matches[k], it.nm, sizeof(int32_t)))) return rc;
    }
    return s.finish();
}


int dcs_undistort_points(int n, const float* xy, const float K4[4], const float dist5[5], float* out)
{
    if (n < 0 || (n && (!xy || !out)) || !K4 || !dist5) { set_error("dcs_undistort_points: bad argument"); return DCS_ERR_INVALID; }
    const double c[9] = {K4[0], K4[1], K4[2], K4[3], dist5[0], dist5[1], dist5[2], dist5[3], dist5[4]};
    for (int i = 0; i < n; ++i) {
        if (dist5[0] == 0.0f) { out[2 * i] = xy[2 * i]; out[2 * i + 1] = xy[2 * i + 1]; }       // Frame.cc:414-418: the key points are taken as they are
        else undistort_point(c, xy[2 * i], xy[2 * i + 1], out[2 * i], out[2 * i + 1]);
    }
    return DCS_OK;
}

int dcs_track_frame_device(int n_frames, const dcs_track_dev_frame* frames, const dcs_track_params* prm, int mode, int check_orientation,
                           dcs_track_dev_result* res, void* stream)
{
    const int F = n_frames;
    if (F < 0 || (mode != 0 && mode != 1) || (F && (!frames || !prm || !res || !res->n_features || !res->r.poses || !res->r.n_inliers || !res->r.n_matches ||
        !res->r.match_of_point || !res->r.point_of_feature || !res->r.outlier)) || (F && (prm->n_levels < 1 || !prm->inv_level_sigma2 || !prm->cams))) {
        set_error("dcs_track_frame_device: bad argument"); return DCS_ERR_INVALID;
    }
    if (F == 0) return DCS_OK;
    if (prm->n_cams < 1 || prm->n_cams > kPoseMaxCams) { set_error("dcs_track_frame_device: prm->n_cams must be 1..%d", kPoseMaxCams); return DCS_ERR_INVALID; }
    int rc, max_pts = 0, max_q4 = 0, max_cap = 0, max_cams = 0;
    long long n_feat = 0;
    for (int k = 0; k < F; ++k) {
        const dcs_track_dev_frame& t = frames[k];
        const dcs_dev_frame& df = t.features;
        const dcs_frustum_frame* v = &t.view;
        const int Ncap = df.n_cams * df.cap;
        if (df.n_cams < 1 || df.n_cams > prm->n_cams || df.cap < 1 || df.first_slot < 0 || !df.d_kp || !df.d_desc || !df.d_n || !df.min_x || !df.min_y || !df.grid_w_inv ||
            !df.grid_h_inv || Ncap > kResMaxN || t.n_points < 0 || !t.pose || v->n_cams != df.n_cams || v->n_scale_levels < 1 || !v->Rsw || !v->tsw || !v->Ow || !v->fx || !v->fy ||
            !v->cx || !v->cy || !v->min_x || !v->max_x || !v->min_y || !v->max_y || !v->scale_factors || t.n_held < 0 || t.n_held > Ncap ||
            (t.n_held && (!t.taken || !t.point_xw)) || (t.n_points && (!t.pos || !t.desc)) ||
            (mode == 0 && t.n_points && (!t.normal || !t.min_dist || !t.max_dist)) || (mode == 1 && t.n_points && (!t.q_cam || !t.q_octave || (check_orientation && !t.q_angle))) ||
            !res->r.match_of_point[k] || !res->r.point_of_feature[k] || !res->r.outlier[k]) { set_error("dcs_track_frame_device: frame %d: bad argument", k); return DCS_ERR_INVALID; }
        if (mode == 1) for (int i = 0; i < t.n_points; ++i) if (t.q_cam[i] < 0 || t.q_cam[i] >= df.n_cams) { set_error("frame %d: query %d: camera out of range", k, i); return DCS_ERR_INVALID; }
        n_feat += Ncap; max_cap = std::max(max_cap, Ncap); max_cams = std::max(max_cams, df.n_cams);
        max_pts = std::max(max_pts, t.n_points); max_q4 = std::max(max_q4, (t.n_points + 3) / 4);
    }
    if ((rc = ensure_device())) return rc;
    Scratch s;
    std::vector<TrackItem> items((size_t)F);
    std::vector<DevAsm> das((size_t)F);
    std::vector<int32_t> edge_off((size_t)F);
    const size_t Etot = (size_t)std::max<long long>(n_feat, 1);
    int base = 0;
    int32_t* d_nfeat = nullptr;
    // The arena is carved in four runs so that a call costs one DMA operation up and one down whatever the number of frames: (1) everything
    // that goes up, (2) the arrays the kernels read past a frame's real feature count (its capacity is all the host knows: k_dev_assemble zeroes
    // the slots behind it), (3) scratch, (4) everything that comes down, the per-camera match counters (zeroed by k_dev_assemble too) first.
    for (int k = 0; k < F; ++k) {
        const dcs_track_dev_frame& t = frames[k];
        const dcs_dev_frame& df = t.features;
        TrackItem& it = items[(size_t)k];
        DevAsm& d = das[(size_t)k];
        const dcs_frustum_frame* v = &t.view;
        const int C = df.n_cams, Ncap = C * df.cap, cells = C * DCS_GRID_COLS * DCS_GRID_ROWS, np = t.n_points;
        FrustumDev& Fd = it.F;
        Fd = FrustumDev{};
        Fd.n_cams = C; Fd.n_levels = v->n_scale_levels; Fd.log_scale = v->log_scale_factor;
        for (int c = 0; c < C; ++c) {
            for (int j = 0; j < 9; ++j) Fd.R[c][j] = v->Rsw[9 * c + j];
            for (int j = 0; j < 3; ++j) { Fd.t[c][j] = v->tsw[3 * c + j]; Fd.O[c][j] = v->Ow[3 * c + j]; }
            Fd.fx[c] = v->fx[c]; Fd.fy[c] = v->fy[c]; Fd.cx[c] = v->cx[c]; Fd.cy[c] = v->cy[c];
            Fd.min_x[c] = v->min_x[c]; Fd.max_x[c] = v->max_x[c]; Fd.min_y[c] = v->min_y[c]; Fd.max_y[c] = v->max_y[c];
        }
        if ((rc = s.upload(&Fd.scale_factors, v->scale_factors, (size_t)v->n_scale_levels))) return rc;
        d = DevAsm{};
        d.kp = df.d_kp; d.desc = df.d_desc; d.n = df.d_n; d.cap = df.cap; d.first_slot = df.first_slot; d.n_cams = C;
        for (int c = 0; c < C; ++c) {
            d.und_on[c] = df.dist && df.K && df.dist[5 * c] != 0.0f;
            if (d.und_on[c]) { for (int j = 0; j < 4; ++j) d.und[c][j] = (double)df.K[4 * c + j]; for (int j = 0; j < 5; ++j) d.und[c][4 + j] = (double)df.dist[5 * c + j]; }
        }
        uint8_t *taken_w, *hasp_w; float* pxw_w;
        if ((rc = s.upload(&d.min_x, df.min_x, (size_t)C)) || (rc = s.upload(&d.min_y, df.min_y, (size_t)C)) ||
            (rc = s.upload(&d.w_inv, df.grid_w_inv, (size_t)C)) || (rc = s.upload(&d.h_inv, df.grid_h_inv, (size_t)C))) return rc;
        {   // what the features hold: the WHOLE capacity goes up from a zero-filled host image. (Uploading only n_held entries into a zeroed device
            // array is not enough: Scratch merges staged uploads that are adjacent at their 256-byte padded sizes into one copy, and the padding of
            // the pinned block -- stale bytes -- would land on the zeros behind a short array: frames of < 256 features then grew phantom edges.)
            std::vector<uint8_t> h_tk((size_t)Ncap, 0), h_hp((size_t)Ncap, 0);
            std::vector<float> h_xw((size_t)3 * Ncap, 0.f);
            if (t.n_held) {
                memcpy(h_tk.data(), t.taken, (size_t)t.n_held);
                memcpy(h_hp.data(), t.has_point ? t.has_point : t.taken, (size_t)t.n_held);
                memcpy(h_xw.data(), t.point_xw, sizeof(float) * 3 * (size_t)t.n_held);
            }
            if ((rc = s.upload(&taken_w, h_tk.data(), h_tk.size())) || (rc = s.upload(&hasp_w, h_hp.data(), h_hp.size())) || (rc = s.upload(&pxw_w, h_xw.data(), h_xw.size()))) return rc;
        }
        ProjFrameD& f = it.f;
        f = ProjFrameD{};
        f.n_cams = C; f.N = Ncap;                                   // (the real count lives in cam_off on the device; entries beyond it are inert zeros)
        f.taken = taken_w; f.min_x = d.min_x; f.min_y = d.min_y; f.w_inv = d.w_inv; f.h_inv = d.h_inv;
        it.has_point = hasp_w; it.point_xw = pxw_w;
        it.n_points = np;
        const size_t npu = (size_t)np;
        (void)cells;
        it.normal = it.min_dist = it.max_dist = nullptr; it.candidate = nullptr;
        if ((rc = s.upload(&it.pos, t.pos, 3 * npu))) return rc;
        if (mode == 0) {
            if ((rc = s.upload(&it.normal, t.normal, 3 * npu)) || (rc = s.upload(&it.min_dist, t.min_dist, npu)) || (rc = s.upload(&it.max_dist, t.max_dist, npu))) return rc;
            if (t.candidate && np && (rc = s.upload(&it.candidate, t.candidate, npu))) return rc;
        } else {
            if ((rc = s.upload(&d.q_cam_in, t.q_cam, npu)) || (rc = s.upload(&d.q_octave_in, t.q_octave, npu))) return rc;
        }
        ProjQueriesD& q = it.q;
        q.n = np;
        if ((rc = s.upload(&q.desc, t.desc, 32 * npu))) return rc;
        q.angle = nullptr;
        if (mode == 1 && check_orientation && (rc = s.upload(&q.angle, t.q_angle, npu))) return rc;
        it.edge_base = base; edge_off[(size_t)k] = base;
        base += Ncap;
    }
    TrackItem* d_items; TrackItem* h_items; DevAsm* d_das; DevAsm* h_das; int32_t* d_edge_off; int32_t* h_edge_off; float* d_sig; float* h_sig; double* d_pose_in; double* h_pose_in;
    if ((rc = s.stage(&d_items, &h_items, (size_t)F)) || (rc = s.stage(&d_das, &h_das, (size_t)F)) || (rc = s.stage(&d_edge_off, &h_edge_off, (size_t)F)) ||
        (rc = s.stage(&d_sig, &h_sig, (size_t)prm->n_levels)) || (rc = s.stage(&d_pose_in, &h_pose_in, (size_t)7 * F))) return rc;
    // (2) the key-point arrays: read up to a frame's capacity, so the slots behind its last real feature must be zeros -- k_dev_assemble writes them
    // (round 5 filled the run with a hipMemsetAsync, and filled foreign memory when a cold thread's arena opened a second block inside the run; round 6
    // reserved the run; now there is no fill)
    for (int k = 0; k < F; ++k) {
        DevAsm& d = das[(size_t)k];
        const size_t Ncap = (size_t)frames[k].features.n_cams * frames[k].features.cap;
        if ((rc = s.alloc(&d.kp_x, Ncap)) || (rc = s.alloc(&d.kp_y, Ncap)) || (rc = s.alloc(&d.kp_angle, Ncap)) || (rc = s.alloc(&d.kp_octave, Ncap))) return rc;
    }
    for (int k = 0; k < F; ++k) {                                   // (3) scratch
        TrackItem& it = items[(size_t)k];
        DevAsm& d = das[(size_t)k];
        const int C = frames[k].features.n_cams, Ncap = C * frames[k].features.cap, cells = C * DCS_GRID_COLS * DCS_GRID_ROWS;
        const size_t npe = (size_t)std::max(it.n_points, 1), Ne = (size_t)std::max(Ncap, 1);
        if ((rc = s.alloc(&d.cam_off, (size_t)C + 1)) || (rc = s.alloc(&d.desc_out, (size_t)Ncap * 32)) || (rc = s.alloc(&d.grid_off, (size_t)cells + 1)) || (rc = s.alloc(&d.grid_idx, (size_t)Ncap)) ||
            (rc = s.alloc(&it.q_valid, npe)) || (rc = s.alloc(&it.q_cam, npe)) || (rc = s.alloc(&it.q_level, npe)) || (rc = s.alloc(&it.q_min, npe)) ||
            (rc = s.alloc(&it.q_max, npe)) || (rc = s.alloc(&it.q_u, npe)) || (rc = s.alloc(&it.q_v, npe)) || (rc = s.alloc(&it.q_radius, npe)) ||
            (rc = s.alloc(&it.cand, npe * kProjCap)) || (rc = s.alloc(&it.cand_n, npe)) || (rc = s.alloc(&it.state, npe)) ||
            (rc = s.alloc(&it.qf, Ne)) || (rc = s.alloc(&it.bin, npe)) || (rc = s.alloc(&it.edge_feature, Ne))) return rc;
        ProjFrameD& f = it.f;
        f.cam_off = d.cam_off; f.kp_x = d.kp_x; f.kp_y = d.kp_y; f.kp_octave = d.kp_octave; f.kp_angle = d.kp_angle; f.desc = d.desc_out;
        f.grid_off = d.grid_off; f.grid_idx = d.grid_idx;
        ProjQueriesD& q = it.q;
        q.valid = it.q_valid; q.cam = it.q_cam; q.u = it.q_u; q.v = it.q_v; q.radius = it.q_radius; q.min_level = it.q_min; q.max_level = it.q_max;
    }
    double *d_xw, *d_obs, *d_w, *d_err, *d_out; int32_t *d_ecam, *d_cnt, *d_ninl; uint8_t *d_level, *d_outl;
    if ((rc = s.alloc(&d_xw, 3 * Etot)) || (rc = s.alloc(&d_obs, 2 * Etot)) || (rc = s.alloc(&d_w, Etot)) || (rc = s.alloc(&d_err, 2 * Etot)) ||
        (rc = s.alloc(&d_ecam, Etot)) || (rc = s.alloc(&d_cnt, (size_t)F)) || (rc = s.alloc(&d_level, Etot)) || (rc = s.alloc(&d_outl, Etot))) return rc;
    {   // (4) what comes down: the match counters of every frame (zeroed by k_dev_assemble) first, then the rest
        for (int k = 0; k < F; ++k) if ((rc = s.alloc(&items[(size_t)k].nm, (size_t)kFrMaxCams))) return rc;
        if ((rc = s.alloc(&d_nfeat, (size_t)F * kFrMaxCams)) || (rc = s.alloc(&d_out, (size_t)7 * F)) || (rc = s.alloc(&d_ninl, (size_t)F))) return rc;
        for (int k = 0; k < F; ++k) {
            TrackItem& it = items[(size_t)k];
            const size_t npe = (size_t)std::max(it.n_points, 1), Ne = (size_t)std::max(it.f.N, 1);
            if ((rc = s.alloc(&it.mq, npe)) || (rc = s.alloc(&it.point_of_feature, Ne)) || (rc = s.alloc(&it.feat_outlier, Ne))) return rc;
        }
    }
    for (int k = 0; k < F; ++k) { das[(size_t)k].n_features = d_nfeat + (size_t)k * kFrMaxCams; das[(size_t)k].n_matches = items[(size_t)k].nm; }
    memcpy(h_items, items.data(), sizeof(TrackItem) * (size_t)F);
    memcpy(h_das, das.data(), sizeof(DevAsm) * (size_t)F);
    memcpy(h_edge_off, edge_off.data(), sizeof(int32_t) * (size_t)F);
    memcpy(h_sig, prm->inv_level_sigma2, sizeof(float) * (size_t)prm->n_levels);
    for (int k = 0; k < F; ++k) memcpy(h_pose_in + (size_t)7 * k, frames[k].pose, sizeof(double) * 7);
    hipStream_t st = s.st;                                          // (flushes the staged uploads)
    // the features were produced on the caller's stream: this call's stream waits for it -- when there is something to wait for. A stream that has
    // drained (the second stage of a frame: the extraction ended a stage ago) needs no event: recording one and waiting for it put 8 us of idle queue in
    // front of the first kernel and four runtime calls on the host's way to it. The event is the calling thread's own, made once.
    if (hipStreamQuery((hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();                                // (hipErrorNotReady is what the query is for)
        thread_local hipEvent_t tl_ev = nullptr;
        if (!tl_ev) DCS_HIP(hipEventCreateWithFlags(&tl_ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(tl_ev, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, tl_ev, 0);
        if (e != hipSuccess) { set_error("dcs_track_frame_device: cannot order behind the caller's stream: %s", hipGetErrorString(e)); return DCS_ERR_HIP; }
    }
    const int cells_max = max_cams * DCS_GRID_COLS * DCS_GRID_ROWS;
    hipLaunchKernelGGL(k_dev_assemble, dim3((4 * max_cap + 255) / 256, F), dim3(256), 0, st, d_das);
    {
        const size_t grid_lds = sizeof(int) * ((size_t)cells_max + (size_t)max_cap);
        static std::atomic<size_t> grid_lds_set{65536};            // what every launch may ask for without saying so
        if (grid_lds > grid_lds_set.load()) {
            DCS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dev_grid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)grid_lds));
            grid_lds_set.store(grid_lds);
        }
        hipLaunchKernelGGL(k_dev_grid, dim3(F, 1 + (max_pts + kGridT - 1) / kGridT), dim3(kGridT), grid_lds, st, d_das, cells_max, (const TrackItem*)d_items, mode,
                           prm->viewing_cos_limit, prm->th);                      // the grids (y = 0) and the queries (y >= 1) in one launch
    }
    if (max_pts > 0) hipLaunchKernelGGL(k_track_collect, dim3(max_q4, F), dim3(256), 0, st, d_items);
    if (mode == 0) hipLaunchKernelGGL(k_track_resolve, dim3(F), dim3(kResT), 0, st, d_items, prm->th_high, prm->nn_ratio);
    else hipLaunchKernelGGL(k_track_resolve_cam, dim3(max_cams, F), dim3(kResT), 0, st, d_items, prm->th_high, check_orientation ? 1 : 0);
    hipLaunchKernelGGL(k_track_edges, dim3(F), dim3(kEdgesT), 0, st, d_items, d_sig, prm->n_levels, d_xw, d_obs, d_w, d_ecam, d_cnt);
    DCS_CHECK_LAUNCH();
    PoseOptDevice po{};
    po.poses = d_pose_in; po.edge_off = d_edge_off; po.edge_cnt = d_cnt; po.xw = d_xw; po.obs = d_obs; po.w = d_w; po.cam = d_ecam;
    po.huber = prm->huber_delta;
    for (int i = 0; i < 4; ++i) { po.chi2_th[i] = prm->chi2_th[i]; po.its[i] = prm->its[i]; }
    po.err = d_err; po.level = d_level; po.out_poses = d_out; po.outlier = d_outl; po.n_inliers = d_ninl; po.edge_chi2 = nullptr; po.n_iters = nullptr;
    if ((rc = launch_pose_opt_device(po, prm->cams, prm->n_cams, F, max_cap, st))) return rc;
    hipLaunchKernelGGL(k_track_finish, dim3(F), dim3(256), 0, st, d_items, (const int32_t*)d_cnt, (const uint8_t*)d_outl);
    DCS_CHECK_LAUNCH();
    std::vector<int32_t> nfeat((size_t)F * kFrMaxCams), nm((size_t)F * kFrMaxCams);
    if ((rc = s.download_bytes(res->r.poses, d_out, sizeof(double) * 7 * F)) || (rc = s.download_bytes(res->r.n_inliers, d_ninl, sizeof(int32_t) * F)) ||
        (rc = s.download_bytes(nfeat.data(), d_nfeat, sizeof(int32_t) * nfeat.size()))) return rc;
    for (int k = 0; k < F; ++k) {
        const TrackItem& it = items[(size_t)k];
        if (it.n_points && (rc = s.download_bytes(res->r.match_of_point[k], it.mq, sizeof(int32_t) * it.n_points))) return rc;
        if ((rc = s.download_bytes(res->r.point_of_feature[k], it.point_of_feature, sizeof(int32_t) * it.f.N)) ||
            (rc = s.download_bytes(res->r.outlier[k], it.feat_outlier, (size_t)it.f.N))) return rc;
        if ((rc = s.download_bytes(&nm[(size_t)k * kFrMaxCams], it.nm, sizeof(int32_t) * kFrMaxCams))) return rc;
    }
    if ((rc = s.finish())) return rc;
    size_t nf_at = 0;                                               // n_features: the frames' counts one after the other (frame k at the sum of n_cams of frames 0..k-1)
    for (int k = 0; k < F; ++k) {
        const int C = frames[k].features.n_cams;
        for (int c = 0; c < C; ++c) res->n_features[nf_at + c] = nfeat[(size_t)k * kFrMaxCams + c];
        nf_at += (size_t)C;
        int total = 0;
        for (int c = 0; c < (mode == 1 ? C : 1); ++c) total += nm[(size_t)k * kFrMaxCams + c];
        res->r.n_matches[k] = total;
    }
    return DCS_OK;
}

}  // extern "C"
