// octree_kernels.hip -- DistributeOctTree (reference src/ORBextractor.cc:481-763) on the GPU.
//
// Device version of the sort/scan formulation documented and validated on the host in octree.cpp
// (tests/test_octree_host.py compares that formulation with the oracle's literal std::list restatement).
// One workgroup per (image, level):
//   1. path code per candidate (initial node, then one quadrant per depth; children use ceil(w/2))
//   2. bitonic sort of (code << 20 | input index) -- in LDS when it fits, in HBM scratch otherwise
//   3. lcp[i] = common path elements of sorted neighbours; two 16-bin LDS histograms give the list size and
//      the number of multi-key nodes at every depth -> stopping depth D of the breadth-first phase and whether
//      the "expand the fullest nodes first" tail (:673-738) runs
//   4. tail passes: sort the expandable nodes by (size desc, list order | creation seq desc), prefix-sum the
//      list growth, cut at the first node that reaches N
//   5. final nodes -> (list-order key, best candidate = max response, first in input order) -> bitonic sort ->
//      selected keypoints in the reference's list order.
// Ties between equal-size nodes: creation sequence (Q3, same as the oracle). Output is bit-exact vs the oracle.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "config.h"
#include "orb_kernels.h"

namespace dcs {

namespace {

constexpr int kT = 256;                 // threads per workgroup
constexpr int kLdsKeys = 4096;          // sort in LDS up to this many (padded) keys
constexpr int kMaxDepth = 14;           // 8-bit initial node + 14 x 2-bit quadrants = 36-bit path code
constexpr unsigned long long kM20 = (1ull << 20) - 1;
constexpr unsigned long long kM36 = (1ull << 36) - 1;

__device__ __forceinline__ int common_prefix(unsigned long long a, unsigned long long b)
{
    const unsigned long long x = a ^ b;
    if (x == 0) return kMaxDepth + 1;
    const int hb = 63 - __clzll(x);
    if (hb >= 2 * kMaxDepth) return 0;
    return (2 * kMaxDepth + 1 - hb) / 2;
}

// list-order key of a depth-`depth` node: quadrant j complemented when (depth - j) is even, the initial-node
// index complemented when depth is odd (derivation in octree.cpp)
__device__ __forceinline__ unsigned long long order_key(unsigned long long code, int depth)
{
    const unsigned long long k = code >> (2 * kMaxDepth);
    unsigned long long out = (depth & 1) ? (~k & 0xFF) : k;
#pragma unroll
    for (int j = 1; j <= kMaxDepth; ++j) {
        unsigned long long q = 0;
        if (j <= depth) {
            q = (code >> (2 * (kMaxDepth - j))) & 3;
            if (((depth - j) & 1) == 0) q = 3 - q;
        }
        out = (out << 2) | q;
    }
    return out;
}

__device__ __forceinline__ int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// in-place ascending bitonic sort of npow2 keys (+ optional payload); every thread of the block calls it
__device__ __forceinline__ void bitonic_sort(unsigned long long* keys, unsigned* vals, int npow2)
{
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npow2 >> 1); t += kT) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) {
                    keys[lo] = b; keys[hi] = a;
                    if (vals) { const unsigned va = vals[lo]; vals[lo] = vals[hi]; vals[hi] = va; }
                }
            }
            __syncthreads();
        }
    }
}

// in-place exclusive scan of a[0..n); returns the total. Every thread of the block calls it.
__device__ __forceinline__ int block_scan_inplace(int* a, int n)
{
    __shared__ int s_run, s_w[kT / 64];
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += kT) {
        const int i = base + threadIdx.x;
        const int v = i < n ? a[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int off = s_run, tot = 0;
#pragma unroll
        for (int w = 0; w < kT / 64; ++w) { const int c = s_w[w]; if (w < wave) off += c; tot += c; }
        if (i < n) a[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_run += tot;
        __syncthreads();
    }
    return s_run;
}

}  // namespace

// All scratch arrays have 2 * dense_cap elements; the task whose candidates start at b0 uses [2*b0, 2*b0 + 2n).
__global__ __launch_bounds__(kT) void k_octree(const dcs_candidate* __restrict__ dense, const int32_t* __restrict__ lvl_off,
                                               OctLevels P, OctScratch G, int dense_cap, const int32_t* __restrict__ only_flagged, SelKp* __restrict__ sel,
                                               int32_t* __restrict__ lvl_cnt)
{
    __shared__ unsigned long long s_keys[kLdsKeys];
    __shared__ unsigned s_vals[kLdsKeys];
    __shared__ int s_hist[kMaxDepth + 2], s_exp[kMaxDepth + 3];
    __shared__ int s_D, s_tail, s_size, s_nfin, s_nnext, s_rstar, s_seq;
    const int task = blockIdx.x, tid = threadIdx.x;
    if (only_flagged && !only_flagged[task]) return;          // the histogram fast path already did this task
    const int l = task % P.nlevels, img = task / P.nlevels;
    const OctLevel lp = P.lv[l];
    const int b0 = lvl_off[task], n = min(lvl_off[task + 1], dense_cap) - b0;   // k_gather drops what does not fit
    SelKp* out = sel + (size_t)img * P.out_per_image + lp.out_base;
    if (n <= 0 || lp.height <= 0) { if (tid == 0) lvl_cnt[task] = 0; return; }
    const int n_ini = (int)roundf(__fdiv_rn((float)lp.width, (float)lp.height));
    if (n_ini < 1 || n_ini > 255) { if (tid == 0) lvl_cnt[task] = 0; return; }
    const float hX = __fdiv_rn((float)lp.width, (float)n_ini);
    const dcs_candidate* c = dense + b0;
    const size_t o = (size_t)2 * b0;
    unsigned long long* gk = G.keys + o;          // sorted keys
    unsigned char* lcp = G.lcp + o;
    int* A0 = G.i0 + o; int* A1 = G.i1 + o; int* A2 = G.i2 + o;
    int* todo_b = G.i3 + o; int* todo_e = G.i4 + o; int* next_b = G.i5 + o; int* next_e = G.i6 + o; int* next_seq = G.i7 + o;
    unsigned long long* fkey = G.fkey + o; unsigned* fbeg = G.fval + o;
    unsigned long long* tkey = G.tkey + o; unsigned* tval = G.tval + o;

    // ---- 1. path codes (same float arithmetic as the reference: kp.pt.x / hX, (int)(hX * i))
    const int npow2 = next_pow2(n);
    unsigned long long* keys = npow2 <= kLdsKeys ? s_keys : gk;
    for (int i = tid; i < npow2; i += kT) {
        unsigned long long key = ~0ull;
        if (i < n) {
            const int x = c[i].x, y = c[i].y;
            int k = (int)__fdiv_rn((float)x, hX);
            if (k >= n_ini) k = n_ini - 1;
            int ulx = (int)__fmul_rn(hX, (float)k), urx = (int)__fmul_rn(hX, (float)(k + 1)), uly = 0, bry = lp.height;
            unsigned long long code = (unsigned long long)k;
#pragma unroll
            for (int d = 1; d <= kMaxDepth; ++d) {
                const int mx = ulx + (urx - ulx + 1) / 2, my = uly + (bry - uly + 1) / 2;
                unsigned q = 0;
                if (x < mx) urx = mx; else { ulx = mx; q |= 1; }
                if (y < my) bry = my; else { uly = my; q |= 2; }
                code = (code << 2) | q;
            }
            key = (code << 20) | (unsigned long long)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    // ---- 2. sort by (path code, input index)
    bitonic_sort(keys, nullptr, npow2);
    if (keys != gk) for (int i = tid; i < n; i += kT) gk[i] = keys[i];
    if (tid <= kMaxDepth + 1) s_hist[tid] = 0;
    if (tid <= kMaxDepth + 2) s_exp[tid] = 0;
    __syncthreads();
    // ---- 3. common-prefix lengths, list size and expandable-node count per depth
    for (int i = tid; i <= n; i += kT) {
        int v = 0;
        if (i > 0 && i < n) { v = common_prefix(gk[i - 1] >> 20, gk[i] >> 20); atomicAdd(&s_hist[v], 1); }
        lcp[i] = (unsigned char)v;
    }
    __syncthreads();
    for (int i = tid; i + 1 < n; i += kT) {        // the run starting at i is a multi-key node for depths [lcp[i], lcp[i+1])
        const int a = lcp[i], b = lcp[i + 1];
        if (a < b) { atomicAdd(&s_exp[a], 1); atomicSub(&s_exp[b], 1); }
    }
    __syncthreads();
    if (tid == 0) {                                 // breadth-first phase (:594-673)
        int cum = 1, e = 0, prev = 0, D = kMaxDepth, tail = 0, size_D = 0;
        bool done = false;
        for (int d = 0; d <= kMaxDepth; ++d) {
            cum += s_hist[d]; e += s_exp[d];
            if (done) continue;
            if (d == 0) { prev = cum; size_D = cum; continue; }
            size_D = cum; D = d;
            if (cum >= lp.n_target || cum == prev) { done = true; }
            else if (cum + 3 * e > lp.n_target) { tail = 1; done = true; }
            prev = cum;
        }
        s_D = D; s_tail = tail; s_size = size_D; s_nfin = 0; s_nnext = 0; s_seq = 1;
    }
    __syncthreads();
    const int D = s_D, tail = s_tail;
    // ---- 4. nodes of the breadth-first list L_D = runs of keys sharing D+1 path elements
    for (int i = tid; i < n; i += kT) A0[i] = (i == 0 || lcp[i] <= D) ? 1 : 0;
    __syncthreads();
    const int M0 = block_scan_inplace(A0, n);
    for (int i = tid; i < n; i += kT) {
        if (i == 0 || lcp[i] <= D) { const int m = A0[i]; A1[m] = i; if (m > 0) A2[m - 1] = i; }
    }
    if (tid == 0) A2[M0 - 1] = n;
    __syncthreads();
    int* fend = A0;                                  // scan flags are dead: A0 becomes the end index of final nodes
    for (int m = tid; m < M0; m += kT) {
        const int b = A1[m], e = A2[m];
        const int depth = (e - b == 1) ? min(D, max((int)lcp[b], (int)lcp[e])) : D;   // single keys froze at a shallower depth
        const unsigned long long ok = order_key(gk[b] >> 20, depth);
        if (tail && e - b > 1) {
            const int t = atomicAdd(&s_nnext, 1);
            todo_b[t] = b; todo_e[t] = e;
            tkey[t] = ((kM20 - (unsigned long long)(e - b)) << 36) | ok;      // fullest first, then list order
            tval[t] = (unsigned)t;
        } else {
            const int f = atomicAdd(&s_nfin, 1);
            fkey[f] = (1ull << 60) | ((unsigned long long)(D - depth) << 36) | ok;
            fbeg[f] = (unsigned)b; fend[f] = e;
        }
    }
    __syncthreads();
    // ---- 5. "expand the fullest nodes first" passes (:673-738)
    if (tail) {
        int* nch = A1; int* inc = A2;               // node_b / node_e are dead from here on
        int depth_cur = D;
        bool first_pass = true;
        for (;;) {
            const int T = s_nnext, prev_size = s_size, seq_base = s_seq;
            __syncthreads();
            if (T == 0) break;
            const int tp = next_pow2(T);
            unsigned long long* sk = tp <= kLdsKeys ? s_keys : tkey;
            unsigned* sv = tp <= kLdsKeys ? s_vals : tval;
            for (int i = tid; i < tp; i += kT) {
                if (sk != tkey) { sk[i] = i < T ? tkey[i] : ~0ull; sv[i] = i < T ? tval[i] : 0u; }
                else if (i >= T) { sk[i] = ~0ull; sv[i] = 0u; }
            }
            __syncthreads();
            bitonic_sort(sk, sv, tp);                // processing order r = 0..T-1
            for (int r = tid; r < T; r += kT) {
                const int t = (int)sv[r];
                int cnt = 1;
                for (int i = todo_b[t] + 1; i < todo_e[t]; ++i) cnt += (lcp[i] == depth_cur + 1);
                nch[r] = cnt; inc[r] = cnt;
            }
            __syncthreads();
            (void)block_scan_inplace(inc, T);        // inc[r] = children created before node r
            if (tid == 0) s_rstar = T;
            __syncthreads();
            for (int r = tid; r < T; r += kT)
                if (prev_size + inc[r] + nch[r] - (r + 1) >= lp.n_target) atomicMin(&s_rstar, r);
            __syncthreads();
            const bool full = s_rstar < T;
            const int last_r = full ? s_rstar : T - 1;
            const int created = inc[last_r] + nch[last_r];
            const int new_size = prev_size + created - (last_r + 1);
            const bool stop = full || new_size == prev_size;
            __syncthreads();
            if (tid == 0) { s_nnext = 0; s_size = new_size; s_seq = seq_base + created; }
            __syncthreads();
            for (int r = tid; r < T; r += kT) {
                const int t = (int)sv[r];
                const int b = todo_b[t], e = todo_e[t];
                if (r > last_r) {                    // never reached: the node stays where it is in the list
                    const int f = atomicAdd(&s_nfin, 1);
                    fkey[f] = first_pass ? ((1ull << 60) | (sk[r] & kM36)) : ((sk[r] & kM20) << 36);
                    fbeg[f] = (unsigned)b; fend[f] = e;
                    continue;
                }
                int child = 0;
                for (int cb = b; cb < e; ++child) {
                    int ce = cb + 1;
                    while (ce < e && lcp[ce] > depth_cur + 1) ++ce;
                    const unsigned long long seq = (unsigned long long)(seq_base + inc[r] + child);
                    if (!stop && ce - cb > 1 && depth_cur + 1 < kMaxDepth) {
                        const int s2 = atomicAdd(&s_nnext, 1);
                        next_b[s2] = cb; next_e[s2] = ce; next_seq[s2] = (int)seq;
                    } else {
                        const int f = atomicAdd(&s_nfin, 1);
                        fkey[f] = (kM20 - seq) << 36;               // group 0: pushed to the front, latest first
                        fbeg[f] = (unsigned)cb; fend[f] = ce;
                    }
                    cb = ce;
                }
            }
            __syncthreads();
            if (stop) break;
            const int T2 = s_nnext;
            for (int s2 = tid; s2 < T2; s2 += kT) {  // next pass: fullest first, then latest created first
                todo_b[s2] = next_b[s2]; todo_e[s2] = next_e[s2];
                tkey[s2] = ((kM20 - (unsigned long long)(next_e[s2] - next_b[s2])) << 36) | (kM20 - (unsigned long long)next_seq[s2]);
                tval[s2] = (unsigned)s2;
            }
            depth_cur += 1;
            first_pass = false;
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- 6. best candidate per final node (max response, first in input order), sort into list order, emit
    const int F = s_nfin;
    const int fp = next_pow2(F);
    unsigned long long* sk = fp <= kLdsKeys ? s_keys : fkey;
    unsigned* sv = fp <= kLdsKeys ? s_vals : fbeg;
    for (int f = tid; f < fp; f += kT) {
        if (f < F) {
            const int b = (int)fbeg[f], e = fend[f];
            int best = (int)(gk[b] & kM20);
            for (int i = b + 1; i < e; ++i) {
                const int id = (int)(gk[i] & kM20);
                if (c[id].score > c[best].score || (c[id].score == c[best].score && id < best)) best = id;
            }
            sk[f] = fkey[f]; sv[f] = (unsigned)best;
        } else { sk[f] = ~0ull; sv[f] = 0u; }
    }
    __syncthreads();
    bitonic_sort(sk, sv, fp);
    const int n_out = min(F, lp.out_cap);
    for (int f = tid; f < n_out; f += kT) {
        const dcs_candidate cc = c[sv[f]];
        SelKp s;
        s.x = (int16_t)(cc.x + kMinBorder); s.y = (int16_t)(cc.y + kMinBorder); s.score = (int16_t)cc.score; s.level = (int8_t)l; s.pad = 0;
        out[f] = s;
    }
    if (tid == 0) lvl_cnt[task] = n_out;
}

// ------------------------------------------------------------------------------------------------------------
// Fast path: the quadtree as a histogram pyramid in LDS (no sort of the candidates at all).
// A depth-d node is the bin "initial node, d quadrants"; bins of one depth are contiguous ranges of the depth-6 bins, so
//   * key counts per bin: LDS atomics at depth 6, then a bottom-up pyramid (sum of 4 children, wave shuffles),
//   * the best candidate of any node = max over a contiguous range of a packed (response << 20 | ~index) word,
//   * list size / expandable nodes per depth = non-empty / multi-key bins -> stopping depth D exactly as in k_octree,
//   * the "fullest nodes first" tail only needs the 4 child counts of each expandable node.
// The kernel is latency-bound (a few thousand instructions per task), so it runs 1024 threads per task and keeps the
// number of workgroup barriers small: node lists (<= 512 entries) are ordered by a rank sort (rank = number of smaller
// keys, counted by all threads in parallel; keys are distinct) instead of a bitonic network.
// Valid while everything the reference touches lies within 6 levels (true for the usual quotas); otherwise the task is
// flagged and the general sort-based kernel k_octree redoes it. Output is identical by construction and by test.
#ifdef DCS_OCT_PROF_ALL                       // schedule of every task (scratch/oct_sched.py): start / end on the 100 MHz wall clock, HW_ID, XCC_ID
__device__ long long g_oct_all[4 * 8192];
#define OPA(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_oct_all[4 * blockIdx.x + (k)] = (v); } while (0)
#else
#define OPA(k, v) do { } while (0)
#endif
#ifdef DCS_OCT_PROF                           // phase timestamps of one task (scratch: -DDCS_OCT_PROF side build + dcs_debug_oct_prof)
__device__ long long g_oct_prof[64];
#define OP(k) do { if (blockIdx.x == DCS_OCT_PROF && threadIdx.x == 0) g_oct_prof[k] = clock64(); } while (0)
#else
#define OP(k)
#endif
constexpr int kTH = 512;
constexpr int kHD = 6;                                  // deepest histogram level
constexpr int kCapH = 512;                              // final / expandable nodes handled in LDS

// out[rank] = in[i] for n <= kCapH distinct keys; every thread of the 1024-thread block calls it
template <typename V>
__device__ __forceinline__ void rank_sort(const unsigned long long* kin, const V* vin, unsigned long long* kout, V* vout, int n, int* s_rank)
{
    const int tid = threadIdx.x;
    const int np = next_pow2(max(n, 1)), parts = kTH / np;       // np <= 512: every element gets `parts` >= 2 threads
    if (tid < n) s_rank[tid] = 0;
    __syncthreads();
    const int i = tid & (np - 1), part = tid / np;
    if (i < n) {
        const unsigned long long key = kin[i];
        const int per = (n + parts - 1) / parts;
        const int j0 = part * per, j1 = min(n, j0 + per);
        int r = 0;
#pragma unroll 8
        for (int j = j0; j < j1; ++j) r += kin[j] < key;         // independent LDS reads: keep several in flight
        if (r) atomicAdd(&s_rank[i], r);
    }
    __syncthreads();
    if (tid < n) { const int r = s_rank[tid]; kout[r] = kin[tid]; vout[r] = vin[tid]; }
    __syncthreads();
}

// in-place exclusive scan of a[0..n), n <= kTH; returns the total. Every thread calls it.
__device__ __forceinline__ int scan_block(int* a, int n, int* s_w /* [kTH / 64] */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = tid < n ? a[tid] : 0;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kTH / 64; ++w) { const int c = s_w[w]; if (w < wave) off += c; tot += c; }
    if (tid < n) a[tid] = off + inc - v;
    __syncthreads();
    return tot;
}

template <int NINI>
__global__ __launch_bounds__(kTH, NINI == 1 ? 8 : 4) void k_octree_hist(const dcs_candidate* __restrict__ dense, const int32_t* __restrict__ lvl_off,
                                                     OctLevels P, int dense_cap, int force_general, SelKp* __restrict__ sel,
                                                     int32_t* __restrict__ lvl_cnt, int32_t* __restrict__ need_general)
{
    constexpr int NB6 = NINI << (2 * kHD), NB5 = NINI << (2 * kHD - 2);
    // depth 6 bins: count (8 bits) | best candidate (response << 16 | ~index, 24 bits) in ONE word, updated with a CAS loop
    __shared__ __attribute__((aligned(16))) unsigned s_bin6[NB6];
    __shared__ unsigned short s_cnt[NINI * 1365 + 8];   // depths 0..5: NINI * (1 + 4 + ... + 1024)
    __shared__ unsigned long long s_ka[kCapH], s_kb[kCapH], s_fkey[kCapH];    // expandable nodes (unsorted / sorted), final nodes
    // bins (< 8192) and candidate indices (< 65536) as 16-bit values, child counts as bytes: 39.2 KB for NINI = 1 = FOUR tasks per CU
    // (43.8 KB held three; the kernel is latency-bound, so resident tasks are what counts)
    __shared__ unsigned short s_va[kCapH], s_vb[kCapH], s_fval[kCapH];
    __shared__ unsigned char s_nch[kCapH];
    __shared__ int s_inc[kCapH], s_rank[kCapH];
    __shared__ int s_size[kHD + 2], s_nexp[kHD + 2], s_w[kTH / 64];
    __shared__ int s_D, s_tail, s_cursize, s_nfin, s_nnext, s_rstar, s_seq, s_bail;
    // Workgroup ids are dealt round-robin to the 8 XCDs: with task = blockIdx and 8 levels, XCD k would run ALL tasks of level k -- a third
    // of the kernel's work (level 0) on an eighth of the chip. Level-major order instead: a level's tasks spread over the XCDs by image, and
    // the long tasks (level 0: most candidates, largest quota) start first while the short ones fill in behind them.
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_img = (int)gridDim.x / P.nlevels;
    const int l = (int)blockIdx.x / n_img, img = (int)blockIdx.x - l * n_img;
    const int task = img * P.nlevels + l;
    OPA(0, (long long)wall_clock64()); OPA(2, (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))); OPA(3, (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));
    const OctLevel lp = P.lv[l];
    const int b0 = lvl_off[task], n = min(lvl_off[task + 1], dense_cap) - b0;
    SelKp* out = sel + (size_t)img * P.out_per_image + lp.out_base;
    if (tid == 0) need_general[task] = 0;
    if (n <= 0 || lp.height <= 0) { if (tid == 0) lvl_cnt[task] = 0; return; }
    const int n_ini = (int)roundf(__fdiv_rn((float)lp.width, (float)lp.height));
    if (n_ini < 1 || n_ini > 255) { if (tid == 0) lvl_cnt[task] = 0; return; }
    if (n_ini > NINI || n > 65535 || force_general) { if (tid == 0) need_general[task] = 1; return; }   // 16-bit counts above depth 6
    const float hX = __fdiv_rn((float)lp.width, (float)n_ini);
    const dcs_candidate* c = dense + b0;
    auto lvl_base = [](int d) { return NINI * (((1 << (2 * d)) - 1) / 3); };     // offset of depth d inside s_cnt
    auto count_at = [&](int d, int bin) -> int { return d == kHD ? (int)(s_bin6[bin] & 255u) : (int)s_cnt[lvl_base(d) + bin]; };

    OP(0);
    for (int i = tid; i < NB6; i += kTH) s_bin6[i] = 0;
    if (tid < kHD + 2) { s_size[tid] = 0; s_nexp[tid] = 0; }
    if (tid == 0) { s_nfin = 0; s_nnext = 0; s_seq = 1; s_bail = 0; }
    __syncthreads();
    OP(1);
    // ---- 1. histogram of the depth-6 bins + best candidate per bin
    const unsigned long long* c8 = reinterpret_cast<const unsigned long long*>(c);        // {x, y, score}: one 8-byte load
#pragma unroll 2
    for (int i = tid; i < n; i += kTH) {
        const unsigned long long cw = c8[i];
        const int x = (int)(short)(cw & 0xffff), y = (int)(short)((cw >> 16) & 0xffff);
        const unsigned score = (unsigned)(cw >> 32);
        int k = (int)__fdiv_rn((float)x, hX);
        if (k >= n_ini) k = n_ini - 1;
        int ulx = (int)__fmul_rn(hX, (float)k), urx = (int)__fmul_rn(hX, (float)(k + 1)), uly = 0, bry = lp.height;
        unsigned code = (unsigned)k;
#pragma unroll
        for (int d = 1; d <= kHD; ++d) {
            const int mx = ulx + (urx - ulx + 1) / 2, my = uly + (bry - uly + 1) / 2;
            unsigned q = 0;
            if (x < mx) urx = mx; else { ulx = mx; q |= 1; }
            if (y < my) bry = my; else { uly = my; q |= 2; }
            code = (code << 2) | q;
        }
        const unsigned mine = (score << 16) | (0xFFFFu - (unsigned)i);           // max response, then first in input order
        unsigned old = s_bin6[code];
        for (;;) {
            if ((old & 255u) == 255u) { s_bail = 1; break; }                    // cannot happen for FAST + NMS output; be safe
            const unsigned nw = (max(old >> 8, mine) << 8) | ((old & 255u) + 1u);
            const unsigned prev = atomicCAS(&s_bin6[code], old, nw);
            if (prev == old) break;
            old = prev;
        }
    }
    __syncthreads();
    OP(2);
    // ---- 2. pyramid of counts (depth 5..2 with wave shuffles: a wave owns 64 consecutive depth-5 bins = one depth-2 bin)
    //         and, per depth, the list size (non-empty bins) and the expandable nodes (bins with more than one key)
    for (int b5 = tid; b5 < NB5; b5 += kTH) {
        int ne6 = 0, nm6 = 0, c5 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int v = (int)(s_bin6[4 * b5 + q] & 255u); c5 += v; ne6 += v > 0; nm6 += v > 1; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { ne6 += __shfl_xor(ne6, d); nm6 += __shfl_xor(nm6, d); }
        int c4 = c5 + __shfl_xor(c5, 1); c4 += __shfl_xor(c4, 2);
        int c3 = c4 + __shfl_xor(c4, 4); c3 += __shfl_xor(c3, 8);
        int c2 = c3 + __shfl_xor(c3, 16); c2 += __shfl_xor(c2, 32);
        s_cnt[lvl_base(5) + b5] = (unsigned short)c5;
        if ((lane & 3) == 0) s_cnt[lvl_base(4) + (b5 >> 2)] = (unsigned short)c4;
        if ((lane & 15) == 0) s_cnt[lvl_base(3) + (b5 >> 4)] = (unsigned short)c3;
        const int ne5 = __popcll(__ballot(c5 > 0)), nm5 = __popcll(__ballot(c5 > 1));
        const int ne4 = __popcll(__ballot((lane & 3) == 0 && c4 > 0)), nm4 = __popcll(__ballot((lane & 3) == 0 && c4 > 1));
        const int ne3 = __popcll(__ballot((lane & 15) == 0 && c3 > 0)), nm3 = __popcll(__ballot((lane & 15) == 0 && c3 > 1));
        if (lane == 0) {
            s_cnt[lvl_base(2) + (b5 >> 6)] = (unsigned short)c2;
            atomicAdd(&s_size[6], ne6); atomicAdd(&s_nexp[6], nm6);
            atomicAdd(&s_size[5], ne5); atomicAdd(&s_nexp[5], nm5);
            atomicAdd(&s_size[4], ne4); atomicAdd(&s_nexp[4], nm4);
            atomicAdd(&s_size[3], ne3); atomicAdd(&s_nexp[3], nm3);
            atomicAdd(&s_size[2], c2 > 0); atomicAdd(&s_nexp[2], c2 > 1);
        }
    }
    __syncthreads();
    OP(3);
    if (tid < NINI * 4) {                                // depth 1 (and depth 0 by its first child)
        int c1 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) c1 += (int)s_cnt[lvl_base(2) + 4 * tid + q];
        s_cnt[lvl_base(1) + tid] = (unsigned short)c1;
        atomicAdd(&s_size[1], c1 > 0); atomicAdd(&s_nexp[1], c1 > 1);
    }
    if (tid >= 64 && tid < 64 + NINI) {                  // depth 0 from the 16 depth-2 bins of the initial node
        const int k0 = tid - 64;
        int c0 = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) c0 += (int)s_cnt[lvl_base(2) + 16 * k0 + q];
        s_cnt[lvl_base(0) + k0] = (unsigned short)c0;
        atomicAdd(&s_size[0], c0 > 0); atomicAdd(&s_nexp[0], c0 > 1);
    }
    __syncthreads();
    OP(4);
    if (tid == 0) {                                     // breadth-first phase (:594-673), same decisions as k_octree
        int prev = s_size[0], D = -1, tail = 0;
        for (int d = 1; d <= kHD; ++d) {
            const int sz = s_size[d];
            if (sz >= lp.n_target || sz == prev) { D = d; break; }
            if (sz + 3 * s_nexp[d] > lp.n_target) { D = d; tail = 1; break; }
            prev = sz;
        }
        if (D < 0 || (tail && D + 1 > kHD)) s_bail = 1;   // deeper than the pyramid: general kernel
        s_D = D; s_tail = tail; s_cursize = D >= 0 ? s_size[D] : 0;
    }
    __syncthreads();
    if (s_bail) { if (tid == 0) need_general[task] = 1; return; }
    const int D = s_D, tail = s_tail;
    auto code14 = [](unsigned bin, int d) -> unsigned long long { return (unsigned long long)bin << (2 * (kMaxDepth - d)); };
    auto best_of = [&](int d, int bin) -> unsigned {    // best candidate index of a depth-d node
        const int span = 1 << (2 * (kHD - d));
        unsigned m = 0;
        if (span >= 4) {                                 // 16-byte LDS reads, several in flight
            const uint4* p4 = reinterpret_cast<const uint4*>(s_bin6 + bin * span);
#pragma unroll 4
            for (int q = 0; q < (span >> 2); ++q) { const uint4 v = p4[q]; m = max(max(m, max(v.x, v.y)), max(v.z, v.w)); }   // count bits are below the response
        } else m = s_bin6[bin];
        return 0xFFFFu - ((m >> 8) & 0xFFFFu);
    };
    OP(5);
    // ---- 3. nodes of the breadth-first list L_D
    {
        const int nb = NINI << (2 * D);
        for (int b = tid; b < nb; b += kTH) {
            const int cc = count_at(D, b);
            if (cc == 0) continue;
            if (tail && cc > 1) {
                const int t = atomicAdd(&s_nnext, 1);
                if (t < kCapH) { s_ka[t] = ((kM20 - (unsigned long long)cc) << 36) | order_key(code14(b, D), D); s_va[t] = (unsigned short)b; }
                else s_bail = 1;
                continue;
            }
            int depth = D;
            if (cc == 1) { while (depth > 0 && count_at(depth - 1, b >> (2 * (D - depth + 1))) == 1) --depth; }   // froze at the first depth it was alone
            const int f = atomicAdd(&s_nfin, 1);
            if (f < kCapH) {
                s_fkey[f] = (1ull << 60) | ((unsigned long long)(D - depth) << 36) | order_key(code14(b >> (2 * (D - depth)), depth), depth);
                s_fval[f] = (unsigned short)best_of(D, b);
            } else s_bail = 1;
        }
    }
    __syncthreads();
    OP(6);
    // ---- 4. "expand the fullest nodes first" passes (:673-738)
    if (tail && !s_bail) {
        int depth_cur = D;
        bool first_pass = true;
        for (;;) {
            const int T = s_nnext, prev_size = s_cursize, seq_base = s_seq;
            __syncthreads();
            if (T == 0 || s_bail) break;
            rank_sort(s_ka, s_va, s_kb, s_vb, T, s_rank);    // processing order: fullest first, then list order / latest first
            if (tid < T) {
                int cnt = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) cnt += count_at(depth_cur + 1, 4 * (int)s_vb[tid] + q) > 0;
                s_nch[tid] = (unsigned char)cnt; s_inc[tid] = cnt;
            }
            if (tid == 0) s_rstar = T;
            __syncthreads();
            (void)scan_block(s_inc, T, s_w);                 // s_inc[r] = children created before node r
            if (tid < T && prev_size + s_inc[tid] + s_nch[tid] - (tid + 1) >= lp.n_target) atomicMin(&s_rstar, tid);
            __syncthreads();
            const bool full = s_rstar < T;
            const int last_r = full ? s_rstar : T - 1;
            const int created = s_inc[last_r] + s_nch[last_r];
            const int new_size = prev_size + created - (last_r + 1);
            const bool stop = full || new_size == prev_size;
            __syncthreads();
            if (tid == 0) { s_nnext = 0; s_cursize = new_size; s_seq = seq_base + created; }
            __syncthreads();
            if (tid < T) {
                const int r = tid;
                const unsigned bin = s_vb[r];
                if (r > last_r) {                            // never reached: stays where it is in the list
                    const int f = atomicAdd(&s_nfin, 1);
                    if (f < kCapH) {
                        s_fkey[f] = first_pass ? ((1ull << 60) | (s_kb[r] & kM36)) : ((s_kb[r] & kM20) << 36);
                        s_fval[f] = (unsigned short)best_of(depth_cur, (int)bin);
                    } else s_bail = 1;
                } else {
                    int child = 0;
                    for (int q = 0; q < 4; ++q) {
                        const int cb = 4 * (int)bin + q, cc = count_at(depth_cur + 1, cb);
                        if (cc == 0) continue;
                        const unsigned long long seq = (unsigned long long)(seq_base + s_inc[r] + child);
                        ++child;
                        if (!stop && cc > 1) {
                            if (depth_cur + 2 > kHD) { s_bail = 1; continue; }       // grandchildren would leave the pyramid
                            const int s2 = atomicAdd(&s_nnext, 1);                   // next pass: fullest first, then latest created first
                            if (s2 < kCapH) { s_ka[s2] = ((kM20 - (unsigned long long)cc) << 36) | (kM20 - seq); s_va[s2] = (unsigned short)cb; }
                            else s_bail = 1;
                        } else {
                            const int f = atomicAdd(&s_nfin, 1);
                            if (f < kCapH) { s_fkey[f] = (kM20 - seq) << 36; s_fval[f] = (unsigned short)best_of(depth_cur + 1, cb); } else s_bail = 1;
                        }
                    }
                }
            }
            __syncthreads();
            if (stop || s_bail) break;
            depth_cur += 1;
            first_pass = false;
        }
    }
    __syncthreads();
    if (s_bail) { if (tid == 0) need_general[task] = 1; return; }
    OP(7);
    // ---- 5. sort the final nodes into list order, emit
    const int F = s_nfin;
    rank_sort(s_fkey, s_fval, s_ka, s_va, F, s_rank);
    OP(8);
    const int n_out = min(F, lp.out_cap);
    if (tid < n_out) {
        const dcs_candidate cc = c[s_va[tid]];
        SelKp s;
        s.x = (int16_t)(cc.x + kMinBorder); s.y = (int16_t)(cc.y + kMinBorder); s.score = (int16_t)cc.score; s.level = (int8_t)l; s.pad = 0;
        out[tid] = s;
    }
    if (tid == 0) lvl_cnt[task] = n_out;
    OP(9);
    OPA(1, (long long)wall_clock64());
#ifdef DCS_OCT_PROF
    if (blockIdx.x == DCS_OCT_PROF && threadIdx.x == 0) { g_oct_prof[10] = n; g_oct_prof[11] = F; g_oct_prof[12] = s_D; g_oct_prof[13] = s_tail; }
#endif
}

int launch_octree(const dcs_candidate* d_dense, const int32_t* d_lvl_off, const OctLevels& levels, const OctScratch& scratch,
                  int n_tasks, int dense_cap, SelKp* d_sel, int32_t* d_lvl_cnt, int32_t* d_need_general, hipStream_t s)
{
    if (n_tasks <= 0) return DCS_OK;
    const int force_general = opt(OPT_OCTREE_FORCE_GENERAL) != 0 ? 1 : 0;      // test hook: exercise the sort-based kernel
    int max_ini = 1;
    for (int l = 0; l < levels.nlevels; ++l)
        if (levels.lv[l].height > 0) max_ini = max(max_ini, (int)roundf((float)levels.lv[l].width / (float)levels.lv[l].height));
    if (max_ini <= 1)
        hipLaunchKernelGGL(k_octree_hist<1>, dim3(n_tasks), dim3(kTH), 0, s, d_dense, d_lvl_off, levels, dense_cap, force_general, d_sel, d_lvl_cnt, d_need_general);
    else
        hipLaunchKernelGGL(k_octree_hist<2>, dim3(n_tasks), dim3(kTH), 0, s, d_dense, d_lvl_off, levels, dense_cap, force_general, d_sel, d_lvl_cnt, d_need_general);
    DCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_octree, dim3(n_tasks), dim3(kT), 0, s, d_dense, d_lvl_off, levels, scratch, dense_cap, d_need_general, d_sel, d_lvl_cnt);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

}  // namespace dcs

#ifdef DCS_OCT_PROF_ALL
extern "C" int dcs_debug_oct_all(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcs::g_oct_all), sizeof(long long) * 4 * 8192) == hipSuccess ? 0 : -1; }
#endif
#ifdef DCS_OCT_PROF
extern "C" int dcs_debug_oct_prof(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcs::g_oct_prof), sizeof(long long) * 64) == hipSuccess ? 0 : -1; }
#endif
