// bow_kernels.hip -- BoW front half on the GPU: the DBoW2 vocabulary tree resident in HBM,
// transform(features) -> BowVector + FeatureVector for a batch of images, L1 score against a BowVector database.
//
// Reference (Thirdparty/DBoW2/DBoW2): TemplatedVocabulary.h:1151-1228 (transform of an image), :1242-1283 (greedy descent of one
// descriptor: first child with the least FORB::distance wins), BowVector.cpp:34-88 (addWeight / addIfNotExist / normalize),
// FeatureVector.cpp:31-45, ScoringObject.cpp:23-67 (L1 score); called from Frame::ComputeBoW with levelsup = 4 (src/Frame.cc:393-406).
//
// HBM layout: the tree is renumbered into "child slots" in breadth-first order; the children of a node are consecutive slots, so
// one step of the descent reads one contiguous block of n_children x 32 B descriptors. Per slot: original node id, first child
// slot, child count, word id, weight (f64). Top levels (1 + k + k^2 descriptors) stay L2-resident; the leaves' blocks are random
// 320-byte reads -- the kernel is bound by HBM/L2 latency, not by the 60 popcount distances per descriptor.
//
// k_bow_descend : one lane per descriptor of the (image, slot) batch.
// k_bow_assemble: one workgroup per image. std::map semantics are reproduced with two LDS sorts (by word id, by node id; ties
//                 keep feature order). Floating-point results are bit-exact vs the reference's order of additions: a word's value
//                 is w added count times (all addends equal), the norm is ONE lane's sequential sum in ascending word order.
// k_bow_score_l1: one lane per database entry, merge-join in ascending word order (same order of additions as the reference).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

namespace dcs {
namespace {

constexpr int kAsmT = 256;            // threads of k_bow_assemble
constexpr int kAsmMax = 4096;         // features per image handled in LDS
constexpr int kChunk = 10;            // children whose descriptors one lane loads before it compares (k = 10 in ORBvoc)

struct VocabDev {
    const uint4* kid_desc;            // [n_slots][2]
    const int32_t* node_id;           // [n_slots] original node id (row + 1)
    const int32_t* first_kid;         // [n_slots + 1]; entry n_slots = the root
    const uint8_t* n_kids;            // [n_slots + 1]
    const int32_t* word_id;           // [n_slots]
    const double* weight;             // [n_slots]
    int n_slots, L;
};

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ __launch_bounds__(256) void k_bow_descend(VocabDev V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_feat,
                                                     int n_images, int cap, int levelsup, int32_t* __restrict__ word,
                                                     int32_t* __restrict__ node, double* __restrict__ wt)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)n_images * cap) return;
    const int img = (int)(g / cap), j = (int)(g - (long long)img * cap);
    if (j >= n_feat[img]) return;
    const uint4* dp = reinterpret_cast<const uint4*>(desc + (size_t)g * 32);
    const uint4 f0 = dp[0], f1 = dp[1];
    const int nid_level = V.L - levelsup;
    int cur = V.n_slots;                                   // root
    int level = 0, nid_slot = -1;
    int nk = V.n_kids[cur];
    while (nk > 0) {
        ++level;
        const int base = V.first_kid[cur];
        int best_d = 1 << 30, best = base;
        for (int c0 = 0; c0 < nk; c0 += kChunk) {          // blocks of kChunk children: 2 kChunk independent 16-byte loads in flight
            uint4 k0[kChunk], k1[kChunk];
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {
                const int c = min(c0 + u, nk - 1);
                k0[u] = V.kid_desc[2 * (size_t)(base + c)]; k1[u] = V.kid_desc[2 * (size_t)(base + c) + 1];
            }
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {
                const int d = hamming256(f0, f1, k0[u], k1[u]);
                if (c0 + u < nk && d < best_d) { best_d = d; best = base + c0 + u; }      // strict <: the first child wins ties (:1266)
            }
        }
        cur = best;
        if (level == nid_level) nid_slot = cur;
        nk = V.n_kids[cur];
    }
    // leaf above nid_level: the reference leaves *nid unset (:1175, :1275); canonical choice = the leaf
    const int nid = nid_level <= 0 ? 0 : V.node_id[nid_slot >= 0 ? nid_slot : cur];
    const double w = V.weight[cur];
    const bool keep = w > 0;                               // stopped words are skipped (:1181)
    word[g] = keep ? V.word_id[cur] : -1;
    node[g] = keep ? nid : -1;
    wt[g] = w;
}

__device__ __forceinline__ int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int npow2)
{
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npow2 >> 1); t += kAsmT) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
}

// slot[i] = number of segment heads before i for the sorted keys (head = id differs from the previous key); returns the number of
// segments. Each thread owns a run of `per` consecutive elements. Every thread calls it.
__device__ __forceinline__ int segment_slots(const unsigned long long* keys, int m, int per, int* s_part, int& my_first_slot)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = tid * per, e = min(m, b + per);
    int heads = 0;
    for (int i = b; i < e; ++i) heads += (i == 0 || (keys[i] >> 16) != (keys[i - 1] >> 16));
    int inc = heads;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kAsmT / 64; ++w) { const int c = s_part[w]; if (w < wave) off += c; tot += c; }
    my_first_slot = off + inc - heads;
    __syncthreads();
    return tot;
}

// weighting: 0 TF_IDF, 1 TF (accumulate), 2 IDF, 3 BINARY (first only). norm: 0 none (TF variants divide by the number of words), 1 L1, 2 L2.
__global__ __launch_bounds__(kAsmT) void k_bow_assemble(const int32_t* __restrict__ word, const int32_t* __restrict__ node,
                                                        const double* __restrict__ wt, const int32_t* __restrict__ n_feat, int cap,
                                                        int weighting, int norm_kind, int32_t* __restrict__ bow_word,
                                                        double* __restrict__ bow_val, int32_t* __restrict__ bow_n,
                                                        int32_t* __restrict__ fv_node, int32_t* __restrict__ fv_off,
                                                        int32_t* __restrict__ fv_idx, int32_t* __restrict__ fv_n)
{
    __shared__ unsigned long long s_keys[kAsmMax];
    __shared__ double s_val[kAsmMax];
    __shared__ int s_part[kAsmT / 64];
    __shared__ int s_m;
    __shared__ double s_norm;
    const int img = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_feat[img], cap);
    const size_t o = (size_t)img * cap;
    const int npow2 = next_pow2(max(n, 1));
    const int per = (npow2 + kAsmT - 1) / kAsmT;
    const bool tf = weighting <= 1;

    // ---------------- BowVector: sort by (word id, feature index)
    for (int i = tid; i < npow2; i += kAsmT) {
        unsigned long long key = ~0ull;
        if (i < n) { const int w = word[o + i]; if (w >= 0) key = ((unsigned long long)(unsigned)w << 16) | (unsigned)i; }
        s_keys[i] = key;
    }
    if (tid == 0) s_m = 0;
    __syncthreads();
    bitonic_sort_u64(s_keys, npow2);
    for (int i = tid; i < npow2; i += kAsmT)
        if (s_keys[i] != ~0ull && (i + 1 == npow2 || s_keys[i + 1] == ~0ull)) s_m = i + 1;
    __syncthreads();
    const int m = s_m;                                     // features that are not stopped
    int slot;
    const int n_words = segment_slots(s_keys, m, per, s_part, slot);
    {
        const int b = tid * per, e = min(m, b + per);
        for (int i = b; i < e; ++i) {
            if (i != 0 && (s_keys[i] >> 16) == (s_keys[i - 1] >> 16)) continue;
            const double w = wt[o + (s_keys[i] & 0xFFFF)];     // every feature of a word carries the same weight
            double v = w;
            if (tf) for (int r = i + 1; r < m && (s_keys[r] >> 16) == (s_keys[i] >> 16); ++r) v += w;     // addWeight, one feature at a time
            bow_word[o + slot] = (int32_t)(s_keys[i] >> 16);
            s_val[slot] = v;
            ++slot;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double nrm = 0.0;
        if (norm_kind == 1) { for (int a = 0; a < n_words; ++a) nrm += fabs(s_val[a]); }                  // BowVector.cpp:67-71, map order
        else if (norm_kind == 2) { for (int a = 0; a < n_words; ++a) nrm += s_val[a] * s_val[a]; nrm = sqrt(nrm); }
        else if (tf) nrm = (double)n_words;                                                                  // :1188-1194
        s_norm = nrm;
        bow_n[img] = n_words;
    }
    __syncthreads();
    {
        const double nrm = s_norm;
        const bool divide = norm_kind ? nrm > 0.0 : (tf && n_words > 0);
        for (int a = tid; a < n_words; a += kAsmT) bow_val[o + a] = divide ? s_val[a] / nrm : s_val[a];
    }
    __syncthreads();

    // ---------------- FeatureVector: sort by (node id, feature index)
    for (int i = tid; i < npow2; i += kAsmT) {
        unsigned long long key = ~0ull;
        if (i < n) { const int nd = node[o + i]; if (nd >= 0) key = ((unsigned long long)(unsigned)nd << 16) | (unsigned)i; }
        s_keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_u64(s_keys, npow2);
    const int n_nodes = segment_slots(s_keys, m, per, s_part, slot);
    {
        const int b = tid * per, e = min(m, b + per);
        for (int i = b; i < e; ++i) {
            fv_idx[o + i] = (int32_t)(s_keys[i] & 0xFFFF);
            if (i != 0 && (s_keys[i] >> 16) == (s_keys[i - 1] >> 16)) continue;
            fv_node[o + slot] = (int32_t)(s_keys[i] >> 16);
            fv_off[(size_t)img * (cap + 1) + slot] = i;
            ++slot;
        }
    }
    if (tid == 0) { fv_off[(size_t)img * (cap + 1) + n_nodes] = m; fv_n[img] = n_nodes; }
}

__global__ __launch_bounds__(256) void k_bow_score_l1(const int32_t* __restrict__ q_word, const double* __restrict__ q_val, int nq,
                                                      const int32_t* __restrict__ db_off, const int32_t* __restrict__ db_word,
                                                      const double* __restrict__ db_val, int n_db, double* __restrict__ score)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_db) return;
    int a = 0, b = db_off[k];
    const int be = db_off[k + 1];
    double s = 0.0;
    while (a < nq && b < be) {
        const int wa = q_word[a], wb = db_word[b];
        if (wa == wb) { const double vi = q_val[a], wi = db_val[b]; s += fabs(vi - wi) - fabs(vi) - fabs(wi); ++a; ++b; }
        else if (wa < wb) ++a;
        else ++b;
    }
    score[k] = -s / 2.0;
}

// KeyFrameDatabase query: one lane per database entry, the same merge-join as k_bow_score_l1 (identical order of additions) that
// also counts the shared words -- what the walk over the inverted files accumulates in mnLoopWords / mnRelocWords
// (src/KeyFrameDatabase.cc:128-149, 257-272) -- and remembers the first one, which fixes the entry's place in lKFsSharingWords.
__global__ __launch_bounds__(256) void k_kfdb_query(const int32_t* __restrict__ q_word, const double* __restrict__ q_val, int nq,
                                                    const int64_t* __restrict__ db_off, const int32_t* __restrict__ db_word,
                                                    const double* __restrict__ db_val, const uint8_t* __restrict__ dead, int n_db,
                                                    int32_t* __restrict__ common, int32_t* __restrict__ first_word, float* __restrict__ score)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_db) return;
    int cnt = 0, first = -1;
    double s = 0.0;
    if (!dead[k]) {
        int a = 0;
        int64_t b = db_off[k];
        const int64_t be = db_off[k + 1];
        while (a < nq && b < be) {
            const int wa = q_word[a], wb = db_word[b];
            if (wa == wb) {
                const double vi = q_val[a], wi = db_val[b];
                s += fabs(vi - wi) - fabs(vi) - fabs(wi);
                if (!cnt) first = wa;
                ++cnt; ++a; ++b;
            } else if (wa < wb) ++a;
            else ++b;
        }
    }
    common[k] = cnt; first_word[k] = first; score[k] = (float)(-s / 2.0);          // `float si = mpVoc->score(...)`
}

}  // namespace
}  // namespace dcs

using namespace dcs;

// KeyFrameDatabase of ONE camera (mvvInvertedFiles[c], src/KeyFrameDatabase.cc:46-110) kept in HBM as the entries' BowVectors in
// order of insertion: an inverted list is then "the entries that hold the word, ascending entry id" -- the reference's push_back
// order -- without being stored. Grow-only arrays, erased entries stay as tombstones.
struct dcs_kfdb {
    int device = 0, n = 0;
    int64_t n_words = 0;
    std::vector<int64_t> h_off{0};
    std::vector<uint8_t> h_dead;
    int32_t* d_word = nullptr; double* d_val = nullptr; size_t cap_word = 0, cap_val = 0;
    int64_t* d_off = nullptr; uint8_t* d_dead = nullptr; size_t cap_off = 0, cap_dead = 0;
    hipStream_t st = nullptr;
    PinnedBuf<char> stage;                           // one add = one pinned image of {words, values, end offset, alive byte}
    // the arrays live on the device the handle was created on; the calling thread's scratch arena lives on ITS current device
    int check_device(const char* who) const
    {
        int cur = -1;
        DCS_HIP(hipGetDevice(&cur));
        if (cur != device) { set_error("%s: the database lives on device %d, the calling thread's current device is %d", who, device, cur); return DCS_ERR_INVALID; }
        return DCS_OK;
    }
    ~dcs_kfdb()
    {
        if (d_word) (void)hipFree(d_word);
        if (d_val) (void)hipFree(d_val);
        if (d_off) (void)hipFree(d_off);
        if (d_dead) (void)hipFree(d_dead);
        if (st) (void)hipStreamDestroy(st);
    }
    template <typename T> int grow(T*& p, size_t& cap, size_t need, size_t used)
    {
        if (need <= cap) return DCS_OK;
        size_t nc = std::max<size_t>(need, std::max<size_t>(2 * cap, 1024));
        T* q = nullptr;
        DCS_HIP(hipMalloc((void**)&q, nc * sizeof(T)));
        if (p && used) DCS_HIP(hipMemcpyAsync(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, st));
        DCS_HIP(hipStreamSynchronize(st));
        if (p) (void)hipFree(p);
        p = q; cap = nc;
        return DCS_OK;
    }
};

struct dcs_vocab {
    int k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0, n_slots = 0, device = 0;
    DevBuf<uint4> d_desc;
    DevBuf<int32_t> d_node_id, d_first_kid, d_word;
    DevBuf<uint8_t> d_n_kids;
    DevBuf<double> d_weight;
    VocabDev dev() const { return VocabDev{d_desc.p, d_node_id.p, d_first_kid.p, d_n_kids.p, d_word.p, d_weight.p, n_slots, L}; }
    int norm_kind() const { return scoring == 5 ? 0 : (scoring == 1 ? 2 : 1); }      // mustNormalize, ScoringObject.h:76-91
};

extern "C" {

int dcs_vocab_create(int k, int L, int scoring, int weighting, int n_rows, const int32_t* parent, const uint8_t* is_leaf,
                     const uint8_t* desc, const double* weight, dcs_vocab** out)
{
    if (!out) { set_error("dcs_vocab_create: null out"); return DCS_ERR_INVALID; }
    *out = nullptr;
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) {      // loadFromTextFile :1383
        set_error("dcs_vocab_create: not a DBoW2 vocabulary header (k %d, L %d, scoring %d, weighting %d)", k, L, scoring, weighting); return DCS_ERR_INVALID;
    }
    if (n_rows < 1 || !parent || !is_leaf || !desc || !weight) { set_error("dcs_vocab_create: empty vocabulary"); return DCS_ERR_INVALID; }
    const int n_nodes = n_rows + 1;
    std::vector<int32_t> n_kids(n_nodes, 0);
    for (int i = 0; i < n_rows; ++i) {
        if (parent[i] < 0 || parent[i] > i) { set_error("dcs_vocab_create: row %d names parent %d (a parent precedes its children)", i, parent[i]); return DCS_ERR_INVALID; }
        ++n_kids[parent[i]];
    }
    for (int v = 0; v < n_nodes; ++v) {
        if (n_kids[v] > 255) { set_error("dcs_vocab_create: node %d has %d children", v, n_kids[v]); return DCS_ERR_UNSUPPORTED; }
        if (v > 0 && (is_leaf[v - 1] != 0) != (n_kids[v] == 0)) {
            set_error("dcs_vocab_create: node %d: leaf flag %d but %d children", v, (int)is_leaf[v - 1], n_kids[v]); return DCS_ERR_INVALID;
        }
    }
    // children lists in row order (:1416), then breadth-first slot numbering
    std::vector<int32_t> kid_begin(n_nodes + 1, 0), kids(n_rows), fill(n_nodes, 0);
    for (int v = 0; v < n_nodes; ++v) kid_begin[v + 1] = kid_begin[v] + n_kids[v];
    for (int i = 0; i < n_rows; ++i) { const int p = parent[i]; kids[kid_begin[p] + fill[p]++] = i + 1; }
    std::vector<int32_t> slot_node; slot_node.reserve(n_rows);                 // slot -> node id, BFS: queue of nodes whose children get slots
    std::vector<int32_t> first_of_node(n_nodes, 0);
    {
        std::vector<int32_t> queue; queue.reserve(n_nodes); queue.push_back(0);
        for (size_t qi = 0; qi < queue.size(); ++qi) {
            const int v = queue[qi];
            first_of_node[v] = (int32_t)slot_node.size();
            for (int c = kid_begin[v]; c < kid_begin[v + 1]; ++c) { slot_node.push_back(kids[c]); queue.push_back(kids[c]); }
        }
        if ((int)slot_node.size() != n_rows) { set_error("dcs_vocab_create: %d rows are not reachable from the root", n_rows - (int)slot_node.size()); return DCS_ERR_INVALID; }
    }
    const int S = n_rows;
    std::vector<uint8_t> h_desc((size_t)S * 32), h_nk(S + 1);
    std::vector<int32_t> h_first(S + 1), h_word(S), word_of_node(n_nodes, -1);
    std::vector<double> h_w(S);
    int n_words = 0;
    for (int i = 0; i < n_rows; ++i) if (is_leaf[i]) word_of_node[i + 1] = n_words++;          // :1432-1437
    for (int s = 0; s < S; ++s) {
        const int v = slot_node[s];
        std::memcpy(&h_desc[(size_t)s * 32], desc + (size_t)(v - 1) * 32, 32);
        h_first[s] = first_of_node[v]; h_nk[s] = (uint8_t)n_kids[v]; h_word[s] = word_of_node[v]; h_w[s] = weight[v - 1];
    }
    h_first[S] = first_of_node[0]; h_nk[S] = (uint8_t)n_kids[0];
    int rc = ensure_device();
    if (rc) return rc;
    dcs_vocab* V = new dcs_vocab();
    V->k = k; V->L = L; V->scoring = scoring; V->weighting = weighting; V->n_nodes = n_nodes; V->n_words = n_words; V->n_slots = S;
    (void)hipGetDevice(&V->device);
    auto fail = [&](int r) { delete V; return r; };
    if ((rc = V->d_desc.resize((size_t)S * 2)) || (rc = V->d_node_id.resize(S)) || (rc = V->d_first_kid.resize(S + 1)) || (rc = V->d_word.resize(S)) ||
        (rc = V->d_n_kids.resize(S + 1)) || (rc = V->d_weight.resize(S))) return fail(rc);
    if (hipMemcpy(V->d_desc.p, h_desc.data(), (size_t)S * 32, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(V->d_node_id.p, slot_node.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(V->d_first_kid.p, h_first.data(), sizeof(int32_t) * (S + 1), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(V->d_word.p, h_word.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(V->d_n_kids.p, h_nk.data(), S + 1, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(V->d_weight.p, h_w.data(), sizeof(double) * S, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("dcs_vocab_create: upload failed"); return fail(DCS_ERR_HIP);
    }
    *out = V;
    return DCS_OK;
}

void dcs_vocab_destroy(dcs_vocab* v) { delete v; }

int dcs_vocab_info(const dcs_vocab* v, int* k, int* L, int* n_nodes, int* n_words)
{
    if (!v) { set_error("null vocabulary"); return DCS_ERR_INVALID; }
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    return DCS_OK;
}

int dcs_bow_transform_device(const dcs_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int n_images, int cap, int levelsup,
                             int32_t* d_word, int32_t* d_node, double* d_weight, int32_t* d_bow_word, double* d_bow_val, int32_t* d_bow_n,
                             int32_t* d_fv_node, int32_t* d_fv_off, int32_t* d_fv_idx, int32_t* d_fv_n, void* stream)
{
    if (!v || n_images < 0 || cap < 1 || levelsup < 0 || (n_images && (!d_desc || !d_n || !d_word || !d_node || !d_weight || !d_bow_word || !d_bow_val ||
                                                                          !d_bow_n || !d_fv_node || !d_fv_off || !d_fv_idx || !d_fv_n))) {
        set_error("dcs_bow_transform_device: bad argument"); return DCS_ERR_INVALID;
    }
    if (cap > kAsmMax) { set_error("dcs_bow_transform_device: cap %d > %d features per image", cap, kAsmMax); return DCS_ERR_UNSUPPORTED; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n_images == 0) return DCS_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n_images * cap;
    hipLaunchKernelGGL(k_bow_descend, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, v->dev(), d_desc, d_n, n_images, cap, levelsup, d_word, d_node, d_weight);
    DCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bow_assemble, dim3(n_images), dim3(kAsmT), 0, st, d_word, d_node, d_weight, d_n, cap, v->weighting, v->norm_kind(), d_bow_word, d_bow_val,
                       d_bow_n, d_fv_node, d_fv_off, d_fv_idx, d_fv_n);
    DCS_CHECK_LAUNCH();
    return DCS_OK;
}

int dcs_bow_transform(const dcs_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* word, int32_t* node, int32_t* bow_word,
                      double* bow_val, int* n_words, int32_t* fv_node, int32_t* fv_off, int32_t* fv_idx, int* n_nodes)
{
    if (!v || n < 0 || levelsup < 0 || !n_words || !n_nodes || !fv_off || (n && (!desc || !bow_word || !bow_val || !fv_node || !fv_idx))) {
        set_error("dcs_bow_transform: bad argument"); return DCS_ERR_INVALID;
    }
    *n_words = 0; *n_nodes = 0; fv_off[0] = 0;
    if (n > kAsmMax) { set_error("dcs_bow_transform: %d > %d features per image", n, kAsmMax); return DCS_ERR_UNSUPPORTED; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) return DCS_OK;
    Scratch s;
    uint8_t* dd; int32_t *dn, *dw, *dnd, *dbw, *dbn, *dfn, *dfo, *dfi, *dfc; double *dwt, *dbv;
    if ((rc = s.upload(&dd, desc, (size_t)n * 32)) || (rc = s.upload(&dn, &n, 1)) || (rc = s.alloc(&dw, n)) || (rc = s.alloc(&dnd, n)) || (rc = s.alloc(&dwt, n)) ||
        (rc = s.alloc(&dbw, n)) || (rc = s.alloc(&dbv, n)) || (rc = s.alloc(&dbn, 1)) || (rc = s.alloc(&dfn, n)) || (rc = s.alloc(&dfo, n + 1)) ||
        (rc = s.alloc(&dfi, n)) || (rc = s.alloc(&dfc, 1))) return rc;
    if ((rc = dcs_bow_transform_device(v, dd, dn, 1, n, levelsup, dw, dnd, dwt, dbw, dbv, dbn, dfn, dfo, dfi, dfc, s.st))) return rc;
    // one round trip: the counts and the arrays they delimit come back together, then only the valid prefixes are handed over
    std::vector<int32_t> h_bw(n), h_fn(n), h_fo(n + 1), h_fi(n);
    std::vector<double> h_bv(n);
    int32_t nw = 0, nn = 0;
    if ((rc = s.download_bytes(&nw, dbn, 4)) || (rc = s.download_bytes(&nn, dfc, 4))) return rc;
    if (word && (rc = s.download_bytes(word, dw, sizeof(int32_t) * n))) return rc;
    if (node && (rc = s.download_bytes(node, dnd, sizeof(int32_t) * n))) return rc;
    if ((rc = s.download_bytes(h_bw.data(), dbw, sizeof(int32_t) * n)) || (rc = s.download_bytes(h_bv.data(), dbv, sizeof(double) * n)) ||
        (rc = s.download_bytes(h_fn.data(), dfn, sizeof(int32_t) * n)) || (rc = s.download_bytes(h_fo.data(), dfo, sizeof(int32_t) * (n + 1))) ||
        (rc = s.download_bytes(h_fi.data(), dfi, sizeof(int32_t) * n))) return rc;
    if ((rc = s.finish())) return rc;
    if (nw) { memcpy(bow_word, h_bw.data(), sizeof(int32_t) * nw); memcpy(bow_val, h_bv.data(), sizeof(double) * nw); }
    memcpy(fv_off, h_fo.data(), sizeof(int32_t) * (nn + 1));
    if (nn) { memcpy(fv_node, h_fn.data(), sizeof(int32_t) * nn); memcpy(fv_idx, h_fi.data(), sizeof(int32_t) * h_fo[nn]); }
    *n_words = nw; *n_nodes = nn;
    return DCS_OK;
}

int dcs_bow_score_l1(const int32_t* q_word, const double* q_val, int nq, const int32_t* db_off, const int32_t* db_word, const double* db_val,
                     int n_db, double* score)
{
    if (nq < 0 || n_db < 0 || (nq && (!q_word || !q_val)) || (n_db && (!db_off || !score))) { set_error("dcs_bow_score_l1: bad argument"); return DCS_ERR_INVALID; }
    if (n_db) {
        if (db_off[0] != 0) { set_error("dcs_bow_score_l1: db_off must start at 0"); return DCS_ERR_INVALID; }
        for (int k = 0; k < n_db; ++k) if (db_off[k + 1] < db_off[k]) { set_error("dcs_bow_score_l1: db_off must ascend"); return DCS_ERR_INVALID; }
        if (db_off[n_db] && (!db_word || !db_val)) { set_error("dcs_bow_score_l1: bad argument"); return DCS_ERR_INVALID; }
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (n_db == 0) return DCS_OK;
    Scratch s;
    int32_t *dqw, *dof, *dbw; double *dqv, *dbv, *dsc;
    if ((rc = s.upload(&dqw, q_word, nq)) || (rc = s.upload(&dqv, q_val, nq)) || (rc = s.upload(&dof, db_off, (size_t)n_db + 1)) ||
        (rc = s.upload(&dbw, db_word, db_off[n_db])) || (rc = s.upload(&dbv, db_val, db_off[n_db])) || (rc = s.alloc(&dsc, n_db))) return rc;
    hipLaunchKernelGGL(k_bow_score_l1, dim3((n_db + 255) / 256), dim3(256), 0, s.st, dqw, dqv, nq, dof, dbw, dbv, n_db, dsc);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(score, dsc, sizeof(double) * n_db))) return rc;
    return s.finish();
}

int dcs_kfdb_create(dcs_kfdb** out)
{
    if (!out) { set_error("dcs_kfdb_create: bad argument"); return DCS_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    dcs_kfdb* d = new (std::nothrow) dcs_kfdb;
    if (!d) { set_error("out of memory"); return DCS_ERR_HIP; }
    if (hipGetDevice(&d->device) != hipSuccess || hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking) != hipSuccess) { delete d; set_error("dcs_kfdb_create: no stream"); return DCS_ERR_HIP; }
    *out = d;
    return DCS_OK;
}

void dcs_kfdb_destroy(dcs_kfdb* d) { delete d; }

int dcs_kfdb_size(const dcs_kfdb* d, int* n_entries)
{
    if (!d || !n_entries) { set_error("dcs_kfdb_size: bad argument"); return DCS_ERR_INVALID; }
    *n_entries = d->n;
    return DCS_OK;
}

int dcs_kfdb_add(dcs_kfdb* d, const int32_t* word, const double* val, int n, int* entry_id)
{
    if (!d || n < 0 || (n && (!word || !val)) || !entry_id) { set_error("dcs_kfdb_add: bad argument"); return DCS_ERR_INVALID; }
    for (int i = 1; i < n; ++i) if (word[i] <= word[i - 1]) { set_error("dcs_kfdb_add: word ids must ascend strictly (a BowVector is a std::map)"); return DCS_ERR_INVALID; }
    if (n && word[0] < 0) { set_error("dcs_kfdb_add: negative word id"); return DCS_ERR_INVALID; }
    int rc;
    if ((rc = d->check_device("dcs_kfdb_add"))) return rc;
    const bool first = d->cap_off == 0;
    if ((rc = d->grow(d->d_off, d->cap_off, (size_t)d->n + 2, first ? 0 : (size_t)d->n + 1)) || (rc = d->grow(d->d_dead, d->cap_dead, (size_t)d->n + 1, (size_t)d->n)) ||
        (rc = d->grow(d->d_word, d->cap_word, (size_t)d->n_words + n, (size_t)d->n_words)) || (rc = d->grow(d->d_val, d->cap_val, (size_t)d->n_words + n, (size_t)d->n_words))) return rc;
    // one pinned staging image per add: [val (8 n) | off[n_entries], off[n_entries + 1] (16) | word (4 n) | alive byte], asynchronous copies into
    // the four arrays, ONE synchronisation (the first offset is rewritten with every add: it also initialises off[0] = 0)
    const int64_t end = d->n_words + n;
    const size_t o_off = sizeof(double) * (size_t)n, o_word = o_off + 16, o_alive = o_word + sizeof(int32_t) * (size_t)n;
    if ((rc = d->stage.resize(o_alive + 16))) return rc;
    char* h = d->stage.p;
    if (n) { memcpy(h, val, sizeof(double) * n); memcpy(h + o_word, word, sizeof(int32_t) * n); }
    const int64_t offs[2] = {d->n_words, end};
    memcpy(h + o_off, offs, 16);
    h[o_alive] = 0;
    if (n) {
        DCS_HIP(hipMemcpyAsync(d->d_word + d->n_words, h + o_word, sizeof(int32_t) * n, hipMemcpyHostToDevice, d->st));
        DCS_HIP(hipMemcpyAsync(d->d_val + d->n_words, h, sizeof(double) * n, hipMemcpyHostToDevice, d->st));
    }
    DCS_HIP(hipMemcpyAsync(d->d_off + d->n, h + o_off, 16, hipMemcpyHostToDevice, d->st));
    DCS_HIP(hipMemcpyAsync(d->d_dead + d->n, h + o_alive, 1, hipMemcpyHostToDevice, d->st));
    DCS_HIP(hipStreamSynchronize(d->st));
    d->h_off.push_back(end); d->h_dead.push_back(0);
    *entry_id = d->n;
    d->n += 1; d->n_words = end;
    return DCS_OK;
}

int dcs_kfdb_erase(dcs_kfdb* d, int entry_id)
{
    if (!d || entry_id < 0 || entry_id >= d->n) { set_error("dcs_kfdb_erase: no such entry"); return DCS_ERR_INVALID; }
    if (int rc = d->check_device("dcs_kfdb_erase")) return rc;
    const uint8_t dead = 1;
    DCS_HIP(hipMemcpyAsync(d->d_dead + entry_id, &dead, 1, hipMemcpyHostToDevice, d->st));
    DCS_HIP(hipStreamSynchronize(d->st));
    d->h_dead[entry_id] = 1;
    return DCS_OK;
}

int dcs_kfdb_clear(dcs_kfdb* d)
{
    if (!d) { set_error("dcs_kfdb_clear: bad argument"); return DCS_ERR_INVALID; }
    d->n = 0; d->n_words = 0; d->h_off.assign(1, 0); d->h_dead.clear();
    return DCS_OK;
}

int dcs_kfdb_query(dcs_kfdb* d, const int32_t* q_word, const double* q_val, int nq, int32_t* common, int32_t* first_word, float* score)
{
    if (!d || nq < 0 || (nq && (!q_word || !q_val)) || (d->n && (!common || !first_word || !score))) { set_error("dcs_kfdb_query: bad argument"); return DCS_ERR_INVALID; }
    for (int i = 1; i < nq; ++i) if (q_word[i] <= q_word[i - 1]) { set_error("dcs_kfdb_query: word ids must ascend strictly"); return DCS_ERR_INVALID; }
    if (d->n == 0) return DCS_OK;
    int rc;
    if ((rc = d->check_device("dcs_kfdb_query"))) return rc;
    Scratch s;
    int32_t *dqw, *dcm, *dfw; double* dqv; float* dsc;
    if ((rc = s.upload(&dqw, q_word, nq)) || (rc = s.upload(&dqv, q_val, nq)) || (rc = s.alloc(&dcm, d->n)) || (rc = s.alloc(&dfw, d->n)) || (rc = s.alloc(&dsc, d->n))) return rc;
    hipLaunchKernelGGL(k_kfdb_query, dim3((d->n + 255) / 256), dim3(256), 0, s.st, dqw, dqv, nq, d->d_off, d->d_word, d->d_val, d->d_dead, d->n, dcm, dfw, dsc);
    DCS_CHECK_LAUNCH();
    if ((rc = s.download_bytes(common, dcm, sizeof(int32_t) * d->n)) || (rc = s.download_bytes(first_word, dfw, sizeof(int32_t) * d->n)) ||
        (rc = s.download_bytes(score, dsc, sizeof(float) * d->n))) return rc;
    return s.finish();
}

}  // extern "C"
