// octree.cpp -- DistributeOctTree (reference src/ORBextractor.cc:481-763) as a sort/scan problem.
//
// The reference grows a std::list of nodes by repeated 4-way splits. The geometry of every
// possible node is a pure function of the root rectangle (children use ceil(w/2), :483-484), so
// each keypoint's path through the tree (initial node, then one quadrant id per depth) can be
// computed independently. After sorting the keys by path code:
//   * a depth-d node is a run of keys sharing the first d+1 path elements,
//   * list size after breadth-first iteration d = number of distinct d-prefixes (single-key nodes
//     freeze, :528-535, but still count one each),
//   * nToExpand = number of runs longer than one key,
// which gives the stopping depth D of the breadth-first phase (:594-673) from a histogram of
// adjacent common-prefix lengths. The "expand the fullest nodes first" tail (:673-738) touches a few
// hundred nodes and is done with a small sort per pass. The list order the reference produces by
// push_front is recovered in closed form: nodes created at iteration d appear in reverse creation
// order, creation order follows the previous list order, hence the order key of a depth-d node is
// its path with quadrant j complemented when (d - j) is even (and the initial-node index when d is
// odd: the depth-0 list is built by push_back); frozen leaves of shallower depth come after deeper nodes. This file is the host version (used by the round-1 pipeline and as the model
// of the device version); it is NOT the oracle: tests compare it against oracle/orb_oracle.cpp.
//
// Q3: the reference breaks ties of equal-size nodes by heap address; here
// (as in the oracle) the creation sequence number is used.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "orb_host.h"

namespace dcs {

namespace {

constexpr int kMaxDepth = 14;            // 2 bits per depth below an 8-bit initial-node index
constexpr int kCodeBits = 8 + 2 * kMaxDepth;

struct Key { uint64_t code; int idx; };

inline int common_prefix(uint64_t a, uint64_t b)
{
    const uint64_t x = a ^ b;
    if (x == 0) return kMaxDepth + 1;
    const int hb = 63 - __builtin_clzll(x);
    if (hb >= 2 * kMaxDepth) return 0;
    return (2 * kMaxDepth + 1 - hb) / 2;
}

// order key of a node = its (depth+1)-element prefix with element j complemented when (depth-j) is even
inline uint64_t order_key(uint64_t code, int depth)
{
    uint64_t out = 0;
    const uint64_t k = code >> (2 * kMaxDepth);
    out = ((depth % 2 == 1) ? (~k & 0xFF) : k);          // depth-0 list is in ascending push_back order
    for (int j = 1; j <= kMaxDepth; ++j) {
        uint64_t q = 0;
        if (j <= depth) {
            q = (code >> (2 * (kMaxDepth - j))) & 3;
            if (((depth - j) & 1) == 0) q = 3 - q;
        }
        out = (out << 2) | q;
    }
    return out;
}

struct FinalNode {
    int b, e;             // key range in the sorted array
    int group;            // 0: created in the fullest-first tail, 1: breadth-first list
    int64_t k1;           // group 0: -seq ; group 1: D - depth
    uint64_t k2;          // group 1: order key
};

struct Todo { int b, e, depth; int64_t seq; uint64_t okey; };

}  // namespace

int distribute_octree(const dcs_candidate* c, int n, int width, int height, int n_target,
                      std::vector<dcs_candidate>& out)
{
    out.clear();
    if (n <= 0 || height <= 0) return 0;
    const int n_ini = (int)std::round(static_cast<float>(width) / height);
    if (n_ini < 1 || n_ini > 255) return 0;
    const float hX = static_cast<float>(width) / n_ini;

    std::vector<Key> keys(n);
    for (int i = 0; i < n; ++i) {
        const int x = c[i].x, y = c[i].y;
        int k = (int)((float)x / hX);
        if (k >= n_ini) k = n_ini - 1;
        int ulx = (int)(hX * static_cast<float>(k)), urx = (int)(hX * static_cast<float>(k + 1));
        int uly = 0, bry = height;
        uint64_t code = (uint64_t)k;
        for (int d = 1; d <= kMaxDepth; ++d) {
            const int mx = ulx + (urx - ulx + 1) / 2;        // UL.x + ceil((UR.x-UL.x)/2)
            const int my = uly + (bry - uly + 1) / 2;
            unsigned q = 0;
            if (x < mx) urx = mx; else { ulx = mx; q |= 1; }
            if (y < my) bry = my; else { uly = my; q |= 2; }
            code = (code << 2) | q;
        }
        keys[i] = {code, i};
    }
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.code != b.code ? a.code < b.code : a.idx < b.idx; });

    // lcp[i] = common path elements of sorted keys i-1 and i; lcp[0] = lcp[n] = 0
    std::vector<int> lcp(n + 1, 0);
    int hist[kMaxDepth + 2] = {0};
    int expand_diff[kMaxDepth + 3] = {0};
    for (int i = 1; i < n; ++i) { lcp[i] = common_prefix(keys[i - 1].code, keys[i].code); ++hist[lcp[i]]; }
    for (int i = 0; i + 1 < n; ++i) {                        // run starting at i is multi-key for depths [lcp[i], lcp[i+1]-1]
        if (lcp[i] <= lcp[i + 1] - 1) { ++expand_diff[lcp[i]]; --expand_diff[lcp[i + 1]]; }
    }
    int size_at[kMaxDepth + 1], nexp_at[kMaxDepth + 1];
    {
        int cum = 1, e = 0;
        for (int d = 0; d <= kMaxDepth; ++d) { cum += hist[d]; size_at[d] = cum; e += expand_diff[d]; nexp_at[d] = e; }
    }

    // breadth-first phase (:594-673)
    int D = kMaxDepth;
    bool tail = false;
    {
        int prev = size_at[0];
        for (int d = 1; d <= kMaxDepth; ++d) {
            const int sz = size_at[d];
            if (sz >= n_target || sz == prev) { D = d; break; }
            if (sz + 3 * nexp_at[d] > n_target) { D = d; tail = true; break; }
            prev = sz;
            D = d;
        }
    }

    std::vector<FinalNode> fin;
    fin.reserve(size_at[D] + 64);
    std::vector<Todo> todo;
    // nodes of the breadth-first list L_D
    for (int b = 0; b < n;) {
        int e = b + 1;
        while (e < n && lcp[e] > D) ++e;
        const int depth = (e - b == 1) ? std::min(D, std::max(lcp[b], lcp[e])) : D;
        const uint64_t ok = order_key(keys[b].code, depth);
        if (tail && e - b > 1) todo.push_back({b, e, D, 0, ok});
        else fin.push_back({b, e, 1, (int64_t)(D - depth), ok});
        b = e;
    }

    if (tail) {
        // pass 1 processes the multi-key nodes of depth D: fullest first; equal sizes in list order
        // (= descending creation sequence). Later passes: fullest first, then latest created first.
        std::sort(todo.begin(), todo.end(), [](const Todo& a, const Todo& b) {
            const int ca = a.e - a.b, cb = b.e - b.b;
            return ca != cb ? ca > cb : a.okey < b.okey;
        });
        int size = size_at[D];
        int64_t seq = 1;
        bool first_pass = true;
        while (true) {
            const int prev = size;
            std::vector<Todo> next;
            size_t r = 0;
            bool full = false;
            for (; r < todo.size(); ++r) {
                const Todo& t = todo[r];
                int nch = 0;
                for (int b = t.b; b < t.e;) {
                    int e = b + 1;
                    while (e < t.e && lcp[e] > t.depth + 1) ++e;
                    Todo ch{b, e, t.depth + 1, seq++, 0};
                    next.push_back(ch);
                    ++nch;
                    b = e;
                }
                size += nch - 1;
                if (size >= n_target) { full = true; ++r; break; }
            }
            // nodes of this pass that were not reached stay in the list where they are
            for (size_t k = r; k < todo.size(); ++k) {
                const Todo& t = todo[k];
                if (first_pass) fin.push_back({t.b, t.e, 1, 0, t.okey});
                else fin.push_back({t.b, t.e, 0, -t.seq, 0});
            }
            if (full || size == prev) {
                for (const Todo& ch : next) fin.push_back({ch.b, ch.e, 0, -ch.seq, 0});
                break;
            }
            todo.clear();
            for (const Todo& ch : next) {
                if (ch.e - ch.b > 1 && ch.depth < kMaxDepth) todo.push_back(ch);
                else fin.push_back({ch.b, ch.e, 0, -ch.seq, 0});
            }
            std::sort(todo.begin(), todo.end(), [](const Todo& a, const Todo& b) {
                const int ca = a.e - a.b, cb = b.e - b.b;
                return ca != cb ? ca > cb : a.seq > b.seq;
            });
            first_pass = false;
            if (todo.empty()) break;
        }
    }

    std::sort(fin.begin(), fin.end(), [](const FinalNode& a, const FinalNode& b) {
        if (a.group != b.group) return a.group < b.group;
        if (a.k1 != b.k1) return a.k1 < b.k1;
        return a.k2 < b.k2;
    });
    out.reserve(fin.size());
    for (const FinalNode& f : fin) {
        int best = keys[f.b].idx;
        for (int i = f.b + 1; i < f.e; ++i) {
            const int id = keys[i].idx;
            if (c[id].score > c[best].score || (c[id].score == c[best].score && id < best)) best = id;
        }
        out.push_back(c[best]);
    }
    return (int)out.size();
}

}  // namespace dcs
