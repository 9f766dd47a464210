/*
 * match_oracle.cpp -- CPU restatement of the ORBmatcher inner kernels (TEST INFRASTRUCTURE ONLY).
 * Follows /root/reference/src/ORBmatcher.cc: DescriptorDistance :2015-2031, best/second loop
 * :208-231, accept + ratio :233-236, rotation histogram :241-251/:272-290, ComputeThreeMaxima
 * :1969-2010, constants :57-59. PARITY UNPINNED (the reference has no tests); these functions are
 * dependency-free integer code once cv::Mat is replaced by a pointer.
 */
#include "oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
const int TH_LOW = 50;          /* ORBmatcher.cc:58 */
const int HISTO_LENGTH = 30;    /* ORBmatcher.cc:59 */

/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:2015-2031): SWAR popcount over 8 x int32 */
inline int descriptor_distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1969-2010) on bin sizes */
void three_maxima(const int* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; ++i) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

/* rotation bin (ORBmatcher.cc:243-248): rot = a_q - a_t (+360 if <0), bin = round(rot/30) mod 30.
   `round` is C round() on a float promoted to double: half away from zero. */
inline int rot_bin(float a_q, float a_t)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a_q - a_t;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}
}  // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

/* ComputeThreeMaxima on a histogram given as bin sizes; ind[3] start at -1 like every caller's (ORBmatcher.cc:274-276) */
void orc_three_maxima(const int32_t* sizes, int L, int32_t* ind)
{
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(sizes, L, i1, i2, i3);
    ind[0] = i1; ind[1] = i2; ind[2] = i3;
}

void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* t_mask,
              int32_t* best_idx, int32_t* best_d, int32_t* second_d)
{
    for (int i = 0; i < nq; ++i) {
        int best1 = 256, best2 = 256, idx = -1;               /* ORBmatcher.cc:208-210 */
        const uint8_t* dq = q + (size_t)i * 32;
        for (int j = 0; j < nt; ++j) {
            if (t_mask && t_mask[j]) continue;
            const int dist = descriptor_distance(dq, t + (size_t)j * 32);
            if (dist < best1) { best2 = best1; best1 = dist; idx = j; }
            else if (dist < best2) { best2 = dist; }
        }
        best_idx[i] = idx; best_d[i] = best1; second_d[i] = best2;
    }
}

void orc_knn2_grouped(const uint8_t* q, int nq, const uint8_t* t, int nt, int n_groups,
                      const int32_t* q_off, const int32_t* q_idx, const int32_t* t_off, const int32_t* t_idx,
                      int32_t* best_idx, int32_t* best_d, int32_t* second_d)
{
    (void)nt;
    for (int i = 0; i < nq; ++i) { best_idx[i] = -1; best_d[i] = 256; second_d[i] = 256; }
    for (int g = 0; g < n_groups; ++g) {
        for (int a = q_off[g]; a < q_off[g + 1]; ++a) {
            const int i = q_idx[a];
            int best1 = 256, best2 = 256, idx = -1;
            for (int b = t_off[g]; b < t_off[g + 1]; ++b) {
                const int j = t_idx[b];
                const int dist = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
                if (dist < best1) { best2 = best1; best1 = dist; idx = j; }
                else if (dist < best2) { best2 = dist; }
            }
            best_idx[i] = idx; best_d[i] = best1; second_d[i] = best2;
        }
    }
}

int orc_ratio_rot_filter(int nq, const int32_t* best_idx, const int32_t* best_d, const int32_t* second_d,
                         int th, int th_strict, float ratio, int check_ori,
                         const float* q_angle, const float* t_angle, int32_t* match)
{
    int nmatches = 0;
    std::vector<int> bin_of(nq, -1);
    int histo[HISTO_LENGTH] = {0};
    for (int i = 0; i < nq; ++i) {
        match[i] = -1;
        if (best_idx[i] < 0) continue;
        const bool ok_th = th_strict ? (best_d[i] < th) : (best_d[i] <= th);
        if (!ok_th) continue;
        if (!(static_cast<float>(best_d[i]) < ratio * static_cast<float>(second_d[i]))) continue;
        match[i] = best_idx[i];
        if (check_ori) {
            const int bin = rot_bin(q_angle[i], t_angle[best_idx[i]]);
            bin_of[i] = bin;
            ++histo[bin];
        }
        ++nmatches;
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < nq; ++i) {
            const int b = bin_of[i];
            if (b < 0 || b == ind1 || b == ind2 || b == ind3) continue;
            match[i] = -1;
            --nmatches;
        }
    }
    return nmatches;
}

/* Faithful SearchByBoWCrossCam(F,cF,KF,cKF) (ORBmatcher.cc:162-294): sequential over KF features of
   each shared vocabulary node; F candidates already claimed by an earlier query are skipped
   (:216), so the result depends on processing order -- kept exactly. */
int orc_search_by_bow_crosscam(const uint8_t* desc_kf, const float* ang_kf, const uint8_t* kf_valid, int n_kf,
                               const uint8_t* desc_f, const float* ang_f, int n_f,
                               const int32_t* kf_nodes, const int32_t* kf_off, const int32_t* kf_idx, int kf_n_nodes,
                               const int32_t* f_nodes, const int32_t* f_off, const int32_t* f_idx, int f_n_nodes,
                               float ratio, int check_ori, int32_t* match_f)
{
    (void)n_kf;
    for (int j = 0; j < n_f; ++j) match_f[j] = -1;
    int nmatches = 0;
    std::vector<std::vector<int>> rot_hist(HISTO_LENGTH);
    int a = 0, b = 0;
    while (a < kf_n_nodes && b < f_n_nodes) {
        if (kf_nodes[a] == f_nodes[b]) {
            for (int ia = kf_off[a]; ia < kf_off[a + 1]; ++ia) {
                const int ikf = kf_idx[ia];
                if (!kf_valid[ikf]) continue;                       /* !pMP || pMP->isBad() */
                int best1 = 256, best2 = 256, best_f = -1;
                for (int ib = f_off[b]; ib < f_off[b + 1]; ++ib) {
                    const int jf = f_idx[ib];
                    if (match_f[jf] >= 0) continue;                 /* :216 */
                    const int dist = descriptor_distance(desc_kf + (size_t)ikf * 32, desc_f + (size_t)jf * 32);
                    if (dist < best1) { best2 = best1; best1 = dist; best_f = jf; }
                    else if (dist < best2) { best2 = dist; }
                }
                if (best1 <= TH_LOW) {
                    if (static_cast<float>(best1) < ratio * static_cast<float>(best2)) {
                        match_f[best_f] = ikf;
                        if (check_ori) rot_hist[rot_bin(ang_kf[ikf], ang_f[best_f])].push_back(best_f);
                        ++nmatches;
                    }
                }
            }
            ++a; ++b;
        } else if (kf_nodes[a] < f_nodes[b]) {
            a = (int)(std::lower_bound(kf_nodes, kf_nodes + kf_n_nodes, f_nodes[b]) - kf_nodes);
        } else {
            b = (int)(std::lower_bound(f_nodes, f_nodes + f_n_nodes, kf_nodes[a]) - f_nodes);
        }
    }
    if (check_ori) {
        int histo[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) histo[i] = (int)rot_hist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int jf : rot_hist[i]) { match_f[jf] = -1; --nmatches; }
        }
    }
    return nmatches;
}

/* SearchByBoWCrossCam(KF1, c1, KF2, c2, vpMatches12) (ORBmatcher.cc:297-414): both sides key frames. valid = "has a MapPoint
   that is not bad" (:336, :348); vbMatched2 stays set when the rotation histogram drops the match; best < TH_LOW is strict
   (:364). match12[i] = camera-local KF2 feature or -1. */
int orc_search_by_bow_kfkf(const uint8_t* desc1, const float* ang1, const uint8_t* valid1, int n1,
                           const uint8_t* desc2, const float* ang2, const uint8_t* valid2, int n2,
                           const int32_t* nodes1, const int32_t* off1, const int32_t* idx1, int n_nodes1,
                           const int32_t* nodes2, const int32_t* off2, const int32_t* idx2, int n_nodes2,
                           float ratio, int check_ori, int32_t* match12)
{
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    std::vector<uint8_t> matched2((size_t)std::max(n2, 1), 0);
    std::vector<std::vector<int>> rot_hist(HISTO_LENGTH);
    int nmatches = 0, a = 0, b = 0;
    while (a < n_nodes1 && b < n_nodes2) {
        if (nodes1[a] == nodes2[b]) {
            for (int ia = off1[a]; ia < off1[a + 1]; ++ia) {
                const int i1 = idx1[ia];
                if (!valid1[i1]) continue;
                int best1 = 256, best2 = 256, best_i2 = -1;
                for (int ib = off2[b]; ib < off2[b + 1]; ++ib) {
                    const int i2 = idx2[ib];
                    if (matched2[i2] || !valid2[i2]) continue;                                       /* :348 */
                    const int dist = descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                    if (dist < best1) { best2 = best1; best1 = dist; best_i2 = i2; }
                    else if (dist < best2) { best2 = dist; }
                }
                if (best1 < TH_LOW) {                                                                /* :364 */
                    if (static_cast<float>(best1) < ratio * static_cast<float>(best2)) {
                        match12[i1] = best_i2;
                        matched2[best_i2] = 1;
                        if (check_ori) rot_hist[rot_bin(ang1[i1], ang2[best_i2])].push_back(i1);
                        ++nmatches;
                    }
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) a = (int)(std::lower_bound(nodes1, nodes1 + n_nodes1, nodes2[b]) - nodes1);
        else b = (int)(std::lower_bound(nodes2, nodes2 + n_nodes2, nodes1[a]) - nodes2);
    }
    if (check_ori) {
        int histo[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) histo[i] = (int)rot_hist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int i1 : rot_hist[i]) { match12[i1] = -1; --nmatches; }
        }
    }
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (ORBmatcher.cc:74-91): float arithmetic left to right, the final comparison against
   3.84 * mvLevelSigma2[octave] in double (3.84 is a double literal). */
static bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float* F12, float sigma2_oct)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2_oct;
}

/* SearchForTriangulation (ORBmatcher.cc:1253-1427) for one camera: features without a MapPoint (free1 / free2), per shared
   vocabulary node, best candidate only; bestDist starts at TH_LOW and `dist > bestDist` skips (:1326), so of equal distances
   the last accepted one wins; epipole gate (:1331-1334) and epipolar line gate (:1337) before a candidate may lower bestDist. */
int orc_search_for_triangulation(const uint8_t* desc1, const float* ang1, const uint8_t* free1, int n1,
                                 const uint8_t* desc2, const float* ang2, const uint8_t* free2, int n2,
                                 const int32_t* nodes1, const int32_t* off1, const int32_t* idx1, int n_nodes1,
                                 const int32_t* nodes2, const int32_t* off2, const int32_t* idx2, int n_nodes2,
                                 const float* F12, float ex, float ey, const float* x1, const float* y1, const float* x2, const float* y2,
                                 const int32_t* oct2, const float* level_sigma2, const float* scale_factors, int check_ori, int32_t* match12)
{
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    std::vector<uint8_t> matched2((size_t)std::max(n2, 1), 0);
    std::vector<std::vector<int>> rot_hist(HISTO_LENGTH);
    int nmatches = 0, a = 0, b = 0;
    while (a < n_nodes1 && b < n_nodes2) {
        if (nodes1[a] == nodes2[b]) {
            for (int ia = off1[a]; ia < off1[a + 1]; ++ia) {
                const int i1 = idx1[ia];
                if (!free1[i1]) continue;                                                            /* :1296-1297 */
                int best_dist = TH_LOW, best_i2 = -1;
                for (int ib = off2[b]; ib < off2[b + 1]; ++ib) {
                    const int i2 = idx2[ib];
                    if (matched2[i2] || !free2[i2]) continue;                                        /* :1314 */
                    const int dist = descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                    if (dist > TH_LOW || dist > best_dist) continue;                                 /* :1326 */
                    const float distex = ex - x2[i2], distey = ey - y2[i2];
                    if (distex * distex + distey * distey < 100 * scale_factors[oct2[i2]]) continue;  /* :1333 */
                    if (check_dist_epipolar_line(x1[i1], y1[i1], x2[i2], y2[i2], F12, level_sigma2[oct2[i2]])) { best_i2 = i2; best_dist = dist; }
                }
                if (best_i2 >= 0) {
                    match12[i1] = best_i2;
                    matched2[best_i2] = 1;
                    ++nmatches;
                    if (check_ori) rot_hist[rot_bin(ang1[i1], ang2[best_i2])].push_back(i1);
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) a = (int)(std::lower_bound(nodes1, nodes1 + n_nodes1, nodes2[b]) - nodes1);
        else b = (int)(std::lower_bound(nodes2, nodes2 + n_nodes2, nodes1[a]) - nodes2);
    }
    if (check_ori) {
        int histo[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) histo[i] = (int)rot_hist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int i1 : rot_hist[i]) { match12[i1] = -1; --nmatches; }
        }
    }
    return nmatches;
}

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:270-340) for a batch of map points: point p owns the
   descriptors pool[idx[off[p] .. off[p+1])]. Full N x N distance table, every row sorted, median = sorted[(int)(0.5 (N-1))],
   first row with the least median wins (:318-331). */
void orc_distinctive_descriptors(const uint8_t* pool, const int32_t* off, const int32_t* idx, int n_points, int32_t* best)
{
    std::vector<int> dist, row;
    for (int p = 0; p < n_points; ++p) {
        const int N = off[p + 1] - off[p];
        if (N <= 0) { best[p] = -1; continue; }
        const int32_t* id = idx + off[p];
        dist.assign((size_t)N * N, 0);
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j) {
                const int d = descriptor_distance(pool + (size_t)id[i] * 32, pool + (size_t)id[j] * 32);
                dist[(size_t)i * N + j] = d; dist[(size_t)j * N + i] = d;
            }
        int best_median = INT_MAX, best_idx = 0;
        for (int i = 0; i < N; ++i) {
            row.assign(dist.begin() + (size_t)i * N, dist.begin() + (size_t)(i + 1) * N);
            std::sort(row.begin(), row.end());
            const int median = row[(size_t)(0.5 * (N - 1))];
            if (median < best_median) { best_median = median; best_idx = i; }
        }
        best[p] = best_idx;
    }
}

/* Frame::GetFeaturesInArea (Frame.cc:316-376) */
static void features_in_area(const orc_proj_frame* f, int c, float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out)
{
    out.clear();
    const int nMinCellX = std::max(0, (int)std::floor((x - f->min_x[c] - r) * f->grid_w_inv[c]));
    if (nMinCellX >= ORC_GRID_COLS) return;
    const int nMaxCellX = std::min((int)ORC_GRID_COLS - 1, (int)std::ceil((x - f->min_x[c] + r) * f->grid_w_inv[c]));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - f->min_y[c] - r) * f->grid_h_inv[c]));
    if (nMinCellY >= ORC_GRID_ROWS) return;
    const int nMaxCellY = std::min((int)ORC_GRID_ROWS - 1, (int)std::ceil((y - f->min_y[c] + r) * f->grid_h_inv[c]));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const int base = f->cam_off[c];
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
        for (int iy = nMinCellY; iy <= nMaxCellY; ++iy) {
            const int cell = (c * ORC_GRID_COLS + ix) * ORC_GRID_ROWS + iy;
            for (int j = f->grid_off[cell]; j < f->grid_off[cell + 1]; ++j) {
                const int local = f->grid_idx[j], g = base + local;
                if (bCheckLevels) {
                    if (f->kp_octave[g] < minLevel) continue;
                    if (maxLevel >= 0 && f->kp_octave[g] > maxLevel) continue;
                }
                const float distx = f->kp_x[g] - x, disty = f->kp_y[g] - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(local);
            }
        }
}

int orc_features_in_area(const orc_proj_frame* f, int c, float x, float y, float r, int min_level, int max_level, int32_t* out, int cap)
{
    std::vector<int> v;
    features_in_area(f, c, x, y, r, min_level, max_level, v);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

void orc_search_by_projection(const orc_proj_frame* f, const orc_proj_queries* q, int th_high, float nn_ratio, int check_orientation,
                              int32_t* match_of_query, int32_t* query_of_feature, int32_t* n_matches)
{
    const int N = f->cam_off[f->n_cams];
    std::vector<uint8_t> taken(f->taken, f->taken + N);
    for (int i = 0; i < N; ++i) query_of_feature[i] = -1;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    std::vector<int> vIndices;
    for (int i = 0; i < q->n; ++i) {
        match_of_query[i] = -1;
        if (!q->valid[i]) continue;
        const int c = q->cam[i];
        features_in_area(f, c, q->u[i], q->v[i], q->radius[i], q->min_level[i], q->max_level[i], vIndices);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = q->desc + (size_t)i * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int local : vIndices) {
            const int g = f->cam_off[c] + local;
            if (taken[g]) continue;                                        /* mvpMapPoints[g] && Observations() > 0 */
            const int dist = descriptor_distance(dMP, f->desc + (size_t)g * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = f->kp_octave[g]; bestIdx = g; }
            else if (dist < bestDist2) { bestLevel2 = f->kp_octave[g]; bestDist2 = dist; }
        }
        if (bestDist <= th_high) {
            if (nn_ratio > 0 && bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;   /* :609-610 */
            taken[bestIdx] = 1; query_of_feature[bestIdx] = i; match_of_query[i] = bestIdx;
            ++nmatches;
            if (check_orientation) {                                       /* :1072-1084 */
                float rot = q->angle[i] - f->kp_angle[bestIdx];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx);
            }
        }
    }
    if (check_orientation) {                                               /* :1087-1101 */
        int histo[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int b = 0; b < HISTO_LENGTH; ++b) histo[b] = (int)rotHist[b].size();
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; ++b)
            if (b != ind1 && b != ind2 && b != ind3)
                for (int g : rotHist[b]) { match_of_query[query_of_feature[g]] = -1; query_of_feature[g] = -1; --nmatches; }
    }
    *n_matches = nmatches;
}

/* KeyFrame::GetFeaturesInArea (KeyFrame.cc:728-765): the cell arithmetic of the Frame version without the level test, and the
   |dx|,|dy| < r test reads mvTotalKeysUn[vCell[j]] -- the camera-LOCAL index taken as a global one (:756), i.e. for c > 0 the
   position of another keypoint decides. Kept: Fuse / SearchBySim3CrossCam / SearchByProjection(KF, ...) all go through it. */
static void kf_features_in_area(const orc_proj_frame* f, int c, float x, float y, float r, std::vector<int>& out)
{
    out.clear();
    const int nMinCellX = std::max(0, (int)std::floor((x - f->min_x[c] - r) * f->grid_w_inv[c]));
    if (nMinCellX >= ORC_GRID_COLS) return;
    const int nMaxCellX = std::min((int)ORC_GRID_COLS - 1, (int)std::ceil((x - f->min_x[c] + r) * f->grid_w_inv[c]));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - f->min_y[c] - r) * f->grid_h_inv[c]));
    if (nMinCellY >= ORC_GRID_ROWS) return;
    const int nMaxCellY = std::min((int)ORC_GRID_ROWS - 1, (int)std::ceil((y - f->min_y[c] + r) * f->grid_h_inv[c]));
    if (nMaxCellY < 0) return;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
        for (int iy = nMinCellY; iy <= nMaxCellY; ++iy) {
            const int cell = (c * ORC_GRID_COLS + ix) * ORC_GRID_ROWS + iy;
            for (int j = f->grid_off[cell]; j < f->grid_off[cell + 1]; ++j) {
                const int local = f->grid_idx[j];
                const float distx = f->kp_x[local] - x, disty = f->kp_y[local] - y;          /* mvTotalKeysUn[vCell[j]] */
                if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(local);
            }
        }
}

/* ORBmatcher::SearchByProjection(KF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536, loop closing), the candidate
   loop :494-529 on flat inputs: candidates = KeyFrame::GetFeaturesInArea (kf_features_in_area above, with its index quirk),
   vpMatched[idxLocal] skips (:506; taken[] is indexed globally here, the caller maps the reference's camera-local vpMatched), the
   octave gate nPredictedLevel - 1 .. nPredictedLevel (:510-513) = min_level .. max_level of the query, dist < bestDist (:519),
   bestDist <= th accepts and marks the feature for the queries that follow (:525-529). */
void orc_search_by_projection_kf(const orc_proj_frame* f, const orc_proj_queries* q, int th, int32_t* match_of_query, int32_t* query_of_feature,
                                 int32_t* n_matches)
{
    const int N = f->cam_off[f->n_cams];
    std::vector<uint8_t> taken(f->taken, f->taken + N);
    for (int i = 0; i < N; ++i) query_of_feature[i] = -1;
    int nmatches = 0;
    std::vector<int> cand;
    for (int i = 0; i < q->n; ++i) {
        match_of_query[i] = -1;
        if (!q->valid[i]) continue;
        const int c = q->cam[i];
        kf_features_in_area(f, c, q->u[i], q->v[i], q->radius[i], cand);
        if (cand.empty()) continue;
        const uint8_t* dMP = q->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (int local : cand) {
            const int g = f->cam_off[c] + local;
            if (taken[g]) continue;
            const int kpLevel = f->kp_octave[g];
            if (kpLevel < q->min_level[i] || kpLevel > q->max_level[i]) continue;
            const int dist = descriptor_distance(dMP, f->desc + (size_t)g * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = g; }
        }
        if (bestDist <= th) { taken[bestIdx] = 1; query_of_feature[bestIdx] = i; match_of_query[i] = bestIdx; ++nmatches; }
    }
    *n_matches = nmatches;
}

/* The window searches whose queries do not see each other's results: Fuse(KF, vpMapPoints, th) (ORBmatcher.cc:1431-1556),
   Fuse(KF, Scw, ...) (:1560-1706), both directions of SearchBySim3CrossCam (:1713-1965), SearchByProjection(KF, vpMapPoints,
   sAlreadyFound, th, ORBdist) (:693-799). Per query: candidates = GetFeaturesInArea of a KeyFrame (kf_area != 0, see above) or
   of a Frame; octave gate min_level <= octave <= max_level inside the loop (:1497, :1662, :1847, :757); Fuse's reprojection gate
   e2 * mvInvLevelSigma2[octave] > 5.99 (:1503-1509; float product, double comparison) when chi2_inv_sigma2 != NULL; features
   flagged in f->taken are skipped (vpMatched[idx] of :416-536 read as a snapshot; all zero for the others); strict `dist <
   bestDist` from 256 / INT_MAX: the first of equal distances wins; accepted when bestDist <= th.
   match_of_query[i] = GLOBAL feature index or -1. Returns the number of accepted queries. */
int orc_search_in_window(const orc_proj_frame* f, const orc_proj_queries* q, int th, int kf_area, const float* chi2_inv_sigma2,
                         int32_t* match_of_query, int32_t* best_dist)
{
    int n_acc = 0;
    std::vector<int> cand;
    for (int i = 0; i < q->n; ++i) {
        match_of_query[i] = -1;
        if (best_dist) best_dist[i] = 256;
        if (!q->valid[i]) continue;
        const int c = q->cam[i];
        const float u = q->u[i], v = q->v[i];
        if (kf_area) kf_features_in_area(f, c, u, v, q->radius[i], cand);
        else features_in_area(f, c, u, v, q->radius[i], -1, -1, cand);
        int bestDist = 256, bestIdx = -1;
        for (int local : cand) {
            const int g = f->cam_off[c] + local;
            if (f->taken && f->taken[g]) continue;
            const int lvl = f->kp_octave[g];
            if (lvl < q->min_level[i] || lvl > q->max_level[i]) continue;
            if (chi2_inv_sigma2) {
                const float ex = u - f->kp_x[g], ey = v - f->kp_y[g];
                const float e2 = ex * ex + ey * ey;
                if (e2 * chi2_inv_sigma2[lvl] > 5.99) continue;
            }
            const int dist = descriptor_distance(q->desc + (size_t)i * 32, f->desc + (size_t)g * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = g; }
        }
        if (best_dist) best_dist[i] = bestDist;
        if (bestDist <= th) { match_of_query[i] = bestIdx; ++n_acc; }
    }
    return n_acc;
}

/* SearchForInitialization (ORBmatcher.cc:1117-1251). Queries = F1's key points in global order; valid[i] = "camera CAP and
   octave 0" (:1142-1149); (u, v) = vbPrevMatched[i], radius = windowSize, min_level = max_level = the key point's octave
   (GetFeaturesInArea(CAP, x, y, windowSize, level1, level1), :1152). In-loop state: vMatchedDistance[g] -- a candidate is skipped
   when an earlier query holds it with a distance <= this one's (:1176) -- and vnMatches21: a later, closer query steals the
   feature (:1194-1198). The rotation histogram keeps the entries of robbed queries (they still count in ComputeThreeMaxima,
   :1228) and only live matches outside the three maxima are removed (:1236-1240). match12[i] = global F2 index or -1. */
int orc_search_for_initialization(const orc_proj_frame* f2, const orc_proj_queries* q, float nn_ratio, int check_orientation,
                                  int32_t* match12)
{
    const int N2 = f2->cam_off[f2->n_cams];
    std::vector<int> vMatchedDistance((size_t)std::max(N2, 1), INT_MAX), vnMatches21((size_t)std::max(N2, 1), -1);
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    std::vector<int> vIndices2;
    for (int i1 = 0; i1 < q->n; ++i1) match12[i1] = -1;
    for (int i1 = 0; i1 < q->n; ++i1) {
        if (!q->valid[i1]) continue;
        const int c = q->cam[i1];
        features_in_area(f2, c, q->u[i1], q->v[i1], q->radius[i1], q->min_level[i1], q->max_level[i1], vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = q->desc + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int local : vIndices2) {
            const int g = f2->cam_off[c] + local;
            const int dist = descriptor_distance(d1, f2->desc + (size_t)g * 32);
            if (vMatchedDistance[g] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = g; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nn_ratio) {
                if (vnMatches21[bestIdx2] >= 0) { match12[vnMatches21[bestIdx2]] = -1; --nmatches; }
                match12[i1] = bestIdx2; vnMatches21[bestIdx2] = i1; vMatchedDistance[bestIdx2] = bestDist;
                ++nmatches;
                if (check_orientation) {
                    float rot = q->angle[i1] - f2->kp_angle[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {
        int histo[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int b = 0; b < HISTO_LENGTH; ++b) histo[b] = (int)rotHist[b].size();
        three_maxima(histo, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; ++b) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int i1 : rotHist[b]) if (match12[i1] >= 0) { match12[i1] = -1; --nmatches; }
        }
    }
    return nmatches;
}

int orc_frame_grid(int n_cams, const int32_t* cam_off, const float* kp_x, const float* kp_y, const float* min_x, const float* min_y,
                   const float* grid_w_inv, const float* grid_h_inv, int32_t* grid_off, int32_t* grid_idx)
{
    const int cells = n_cams * ORC_GRID_COLS * ORC_GRID_ROWS;
    std::vector<std::vector<int>> g((size_t)cells);
    for (int c = 0; c < n_cams; ++c)
        for (int i = cam_off[c]; i < cam_off[c + 1]; ++i) {
            const int px = (int)std::nearbyint((kp_x[i] - min_x[c]) * grid_w_inv[c]);      /* cvRound (RNE) */
            const int py = (int)std::nearbyint((kp_y[i] - min_y[c]) * grid_h_inv[c]);
            if (px < 0 || px >= ORC_GRID_COLS || py < 0 || py >= ORC_GRID_ROWS) continue;
            g[(size_t)(c * ORC_GRID_COLS + px) * ORC_GRID_ROWS + py].push_back(i - cam_off[c]);
        }
    int n = 0;
    for (int k = 0; k < cells; ++k) { grid_off[k] = n; for (int v : g[k]) grid_idx[n++] = v; }
    grid_off[cells] = n;
    return n;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// Frame::isInFrustum (src/Frame.cc:244-312), one map point at a time, in the arithmetic the reference's cv::Mat expressions
// perform (OpenCV 3.x, CV_32F; restated -- OpenCV is not in this image, parity unpinned):
//   * Pic = Rsw * P + tsw (:258): MatExpr folds it into ONE gemm(Rsw, P, 1, tsw, 1); 3x3 by 3x1 takes gemm's small-matrix
//     path (matmul.cpp, "len <= 4"): float dot products left to right, then d = (float)(t * alpha + c * beta) in double
//   * u = fx * PicX * invz + cx (:271-272): float, left to right, invz = 1.0f / PicZ
//   * PO = P - Ow (:286) float; dist = cv::norm(PO) (:287): L2 with double accumulation, sqrt in double, narrowed to float
//   * viewCos = PO.dot(Pn) / dist (:294): Mat::dot accumulates (double)a * b; the quotient is double, narrowed to float
//   * PredictScale (src/MapPoint.cc:440-455): ratio = mfMaxDistance / dist (float), ceil(log(ratio) / mfLogScaleFactor) with
//     the float overloads (`using namespace std` is in scope: logf, float division, ceilf), clamped to [0, nScaleLevels - 1]
//   * window (src/ORBmatcher.cc:65-71, 557-565): r = viewCos > 0.998 (double compare) ? 2.5f : 4.0f; r *= th when th != 1;
//     radius = r * mvScaleFactors[level]
extern "C" void orc_is_in_frustum(const orc_frustum_frame* f, int n, const float* pos, const float* normal, const float* min_dist,
                                  const float* max_dist, const uint8_t* candidate, float viewing_cos_limit, float th, uint8_t* in_view,
                                  int32_t* cam, float* u_out, float* v_out, float* view_cos, int32_t* level, float* radius)
{
    const bool bFactor = th != 1.0f;                                       // :545 (th is a float compared with 1.0)
    for (int i = 0; i < n; ++i) {
        in_view[i] = 0; cam[i] = -1; u_out[i] = 0; v_out[i] = 0; view_cos[i] = 0; level[i] = 0; radius[i] = 0;
        if (candidate && !candidate[i]) continue;
        const float* P = pos + 3 * i; const float* Pn = normal + 3 * i;
        for (int ic = 0; ic < f->n_cams; ++ic) {
            const float* R = f->Rsw + 9 * ic; const float* t = f->tsw + 3 * ic; const float* O = f->Ow + 3 * ic;
            float Pic[3];
            for (int r = 0; r < 3; ++r) {
                const float t0 = R[3 * r] * P[0] + R[3 * r + 1] * P[1] + R[3 * r + 2] * P[2];
                Pic[r] = (float)((double)t0 * 1.0 + (double)t[r] * 1.0);
            }
            if (Pic[2] < 0.0f) continue;                                   // :265
            const float invz = 1.0f / Pic[2];
            const float u = f->fx[ic] * Pic[0] * invz + f->cx[ic];
            const float v = f->fy[ic] * Pic[1] * invz + f->cy[ic];
            if (u < f->min_x[ic] || u > f->max_x[ic]) continue;            // :274 (NaN passes, like the reference)
            if (v < f->min_y[ic] || v > f->max_y[ic]) continue;
            const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
            const float PO[3] = {P[0] - O[0], P[1] - O[1], P[2] - O[2]};
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)PO[k] * (double)PO[k];
            const float dist = (float)std::sqrt(s);
            if (dist < minDistance || dist > maxDistance) continue;       // :289
            double dot = 0;
            for (int k = 0; k < 3; ++k) dot += (double)PO[k] * (double)Pn[k];
            const float viewCos = (float)(dot / (double)dist);
            if (viewCos < viewing_cos_limit) continue;                     // :296
            const float ratio = max_dist[i] / dist;
            int nScale = (int)std::ceil(std::log(ratio) / f->log_scale_factor);      // float overloads
            if (nScale < 0) nScale = 0; else if (nScale >= f->n_scale_levels) nScale = f->n_scale_levels - 1;
            float r = ((double)viewCos > 0.998) ? 2.5f : 4.0f;
            if (bFactor) r *= th;
            in_view[i] = 1; cam[i] = ic; u_out[i] = u; v_out[i] = v; view_cos[i] = viewCos; level[i] = nScale;
            radius[i] = r * f->scale_factors[nScale];
            break;                                                         // return true (:307)
        }
    }
}


// ---- ORBmatcher::SearchByProjectionOnCam (src/ORBmatcher.cc:954-1113), the geometry in front of its candidate loop, for every feature i of
// the last frame that holds a good map point (the caller passes those, in ascending feature order: :989-995 skip the rest):
//   * Tsw = Tsc * Tcw, Rsw / tsw its blocks (:962-968) come from the caller as cv::Mat forms them (orc_frustum_frame);
//   * x3Ds = Rsw * x3Dw + tsw (:997): cv::gemm's small-matrix path for CV_32F -- a float dot product left to right, then
//     (float)(t0 * alpha + c * beta) with double alpha = beta = 1 (the arithmetic orc_is_in_frustum restates);
//   * zs < 0 -> skip (:1002); invzs = 1.0 / zs, a DOUBLE division narrowed to float (:1003); u = fx * xs * invzs + cx in float (:1005-1006);
//   * outside [mvMinX, mvMaxX] x [mvMinY, mvMaxY] -> skip (:1008-1011);
//   * radius = th * mvScaleFactors[nLastOctave] (:1031), GetFeaturesInArea(query, u, v, radius, nLastOctave - 1, nLastOctave + 1) (:1036).
// zs == 0 (a point in the camera's focal plane) projects to infinity in the reference; here it is reported as not visible.
extern "C" void orc_motion_model_queries(const orc_frustum_frame* f, int n, const float* pos, const int32_t* q_cam, const int32_t* q_octave, float th,
                                         uint8_t* valid, float* u_out, float* v_out, float* radius, int32_t* min_level, int32_t* max_level)
{
    for (int i = 0; i < n; ++i) {
        const int c = q_cam[i];
        const float* R = f->Rsw + 9 * c; const float* t = f->tsw + 3 * c; const float* P = pos + 3 * i;
        float Ps[3];
        for (int r = 0; r < 3; ++r) {
            const float t0 = R[3 * r] * P[0] + R[3 * r + 1] * P[1] + R[3 * r + 2] * P[2];
            Ps[r] = (float)((double)t0 * 1.0 + (double)t[r] * 1.0);
        }
        int oct = q_octave[i];
        if (oct < 0) oct = 0; else if (oct >= f->n_scale_levels) oct = f->n_scale_levels - 1;
        valid[i] = 0; u_out[i] = 0; v_out[i] = 0;
        radius[i] = th * f->scale_factors[oct];
        min_level[i] = q_octave[i] - 1; max_level[i] = q_octave[i] + 1;
        if (Ps[2] < 0.0f || Ps[2] == 0.0f) continue;
        const float invzs = (float)(1.0 / (double)Ps[2]);
        const float u = f->fx[c] * Ps[0] * invzs + f->cx[c];
        const float v = f->fy[c] * Ps[1] * invzs + f->cy[c];
        u_out[i] = u; v_out[i] = v;
        if (u < f->min_x[c] || u > f->max_x[c]) continue;
        if (v < f->min_y[c] || v > f->max_y[c]) continue;
        valid[i] = 1;
    }
}


// ---- cv::undistortPoints(src, dst, K, distCoeffs, cv::Mat(), K) as Frame::UndistortKeyPoints / ComputeImageBounds call it (src/Frame.cc:410-441, 454-476).
// EXT: OpenCV 3.3 / 3.4.0 modules/imgproc/src/undistort.cpp, cvUndistortPoints (not under /root/reference: parity unpinned): everything in double --
// A = K and the coefficients widened from float, ifx = 1 / fx, (x, y) = ((u - cx) ifx, (v - cy) ify), the tilt compensation with the identity matrix
// (x0 = x = invProj * x with invProj = 1 / 1), FIVE fixed-point iterations
//   r2 = x x + y y;  icdist = (1 + ((k7 r2 + k6) r2 + k5) r2) / (1 + ((k4 r2 + k1) r2 + k0) r2)
//   dX = 2 k2 x y + k3 (r2 + 2 x x) + k8 r2 + k9 r2 r2;  dY = k2 (r2 + 2 y y) + 2 k3 x y + k10 r2 + k11 r2 r2;  x = (x0 - dX) icdist; y = (y0 - dY) icdist
// (k0 k1 k2 k3 k4 = k1 k2 p1 p2 k3 of the rig file, the rest 0), then RR = P * R = K: xx = fx x + 0 y + cx, ww = 1 / (0 x + 0 y + 1), (float)(xx ww).
extern "C" void orc_undistort_points(int n, const float* xy, const float K4[4], const float* dist, int n_dist, float* out)
{
    double k[14] = {0};
    for (int i = 0; i < n_dist && i < 14; ++i) k[i] = (double)dist[i];
    const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3], ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; ++i) {
        double x = xy[2 * i], y = xy[2 * i + 1];
        x = (x - cx) * ifx; y = (y - cy) * ify;
        const double invProj = 1. / 1.0;
        const double x0 = x = invProj * x, y0 = y = invProj * y;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
        out[2 * i] = (float)(xx * ww); out[2 * i + 1] = (float)(yy * ww);
    }
}
