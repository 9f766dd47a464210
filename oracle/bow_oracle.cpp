// bow_oracle.cpp -- TEST INFRASTRUCTURE (parity oracle, never part of the product path).
// CPU restatement of the DBoW2 pieces ORB-SLAM2-DualCam uses per frame (reference Thirdparty/DBoW2/DBoW2):
//   * vocabulary tree as loaded by loadFromTextFile                      TemplatedVocabulary.h:1362-1446
//   * transform(feature, word, weight, nid, levelsup): greedy descent    TemplatedVocabulary.h:1242-1283
//   * transform(features, BowVector, FeatureVector, levelsup)            TemplatedVocabulary.h:1151-1228
//   * BowVector::addWeight / addIfNotExist / normalize                   BowVector.cpp:34-88
//   * FeatureVector::addFeature                                          FeatureVector.cpp:31-45
//   * FORB::distance                                                     FORB.cpp:82-102
//   * L1Scoring::score                                                   ScoringObject.cpp:23-67
// Called from Frame::ComputeBoW with levelsup = 4 (src/Frame.cc:393-406). std::map is used exactly as DBoW2 does, so
// the order of the floating-point additions (per-word accumulation, norm, score) is the reference's.
// PARITY UNPINNED: the reference ships no vocabulary (Vocabulary/download_link.txt) and no test vectors for this path.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "oracle.h"

namespace {

struct Node {
    std::vector<int> children;
    int parent = 0, word_id = -1;
    double weight = 0;
    unsigned char desc[32] = {0};
};

int forb_distance(const unsigned char* a, const unsigned char* b)     // FORB.cpp:82-102
{
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        unsigned int x, y;
        std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
        unsigned int v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// BowVector::addWeight (BowVector.cpp:34-46) / addIfNotExist (:50-58)
void bow_add(std::map<unsigned, double>& bow, unsigned wid, double w, bool if_not_exist)
{
    auto it = bow.lower_bound(wid);
    if (!if_not_exist) { if (it != bow.end() && it->first == wid) it->second += w; else bow.insert(it, {wid, w}); }
    else if (it == bow.end() || it->first != wid) bow.insert(it, {wid, w});
}
// BowVector::normalize (BowVector.cpp:62-88): sum in ascending word order, then one division per entry
void bow_normalize(std::map<unsigned, double>& bow, bool l2)
{
    double norm = 0.0;
    if (!l2) for (auto& kv : bow) norm += std::fabs(kv.second);
    else { for (auto& kv : bow) norm += kv.second * kv.second; norm = std::sqrt(norm); }
    if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
}

}  // namespace

struct orc_vocab {
    int k, L, scoring, weighting;
    std::vector<Node> nodes;
    int n_words = 0;
};

extern "C" {

orc_vocab* orc_vocab_create(int k, int L, int scoring, int weighting, int n_rows, const int32_t* parent, const uint8_t* is_leaf,
                            const uint8_t* desc, const double* weight)
{
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) return nullptr;   // :1383
    orc_vocab* v = new orc_vocab{k, L, scoring, weighting, {}, 0};
    v->nodes.resize((size_t)n_rows + 1);
    for (int i = 0; i < n_rows; ++i) {
        const int nid = i + 1, pid = parent[i];
        if (pid < 0 || pid >= nid) { delete v; return nullptr; }         // the file lists a parent before its children
        Node& nd = v->nodes[nid];
        nd.parent = pid;
        v->nodes[pid].children.push_back(nid);                           // :1416
        std::memcpy(nd.desc, desc + (size_t)i * 32, 32);
        nd.weight = weight[i];
        if (is_leaf[i]) nd.word_id = v->n_words++;                        // :1432-1437
    }
    return v;
}

void orc_vocab_destroy(orc_vocab* v) { delete v; }
int  orc_vocab_words(const orc_vocab* v) { return v ? v->n_words : 0; }

int orc_bow_transform(const orc_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* word, int32_t* node,
                      int32_t* bow_word, double* bow_val, int* n_words, int32_t* fv_node, int32_t* fv_off, int32_t* fv_idx, int* n_nodes)
{
    *n_words = 0; *n_nodes = 0;
    if (fv_off) fv_off[0] = 0;
    if (!v || v->nodes.size() <= 1) return 0;                             // empty(): nothing
    // mustNormalize (ScoringObject.h:76-91): L1 for L1_NORM / CHI_SQUARE / KL / BHATTACHARYYA, L2 for L2_NORM, none for DOT_PRODUCT
    const bool must = v->scoring != 5, l2 = v->scoring == 1;
    const bool tf = v->weighting == 0 || v->weighting == 1;               // TF_IDF or TF: accumulate; IDF / BINARY: first only
    std::map<unsigned, double> bow;
    std::map<unsigned, std::vector<unsigned>> fv;
    const int nid_level = v->L - levelsup;
    for (int i = 0; i < n; ++i) {
        const unsigned char* f = desc + (size_t)i * 32;
        // :1242-1283. When the leaf sits above nid_level the reference leaves *nid untouched (an uninitialised local at the
        // call site, :1175); canonical choice here and on the GPU: the leaf itself (Q12 in DESIGN.md).
        int nid = 0, final_id = 0, level = 0;
        bool have_nid = nid_level <= 0;
        do {
            ++level;
            const std::vector<int>& ch = v->nodes[final_id].children;
            final_id = ch[0];
            int best_d = forb_distance(f, v->nodes[final_id].desc);
            for (size_t c = 1; c < ch.size(); ++c) {
                const int d = forb_distance(f, v->nodes[ch[c]].desc);
                if (d < best_d) { best_d = d; final_id = ch[c]; }
            }
            if (level == nid_level) { nid = final_id; have_nid = true; }
        } while (!v->nodes[final_id].children.empty());
        if (!have_nid) nid = final_id;
        const double w = v->nodes[final_id].weight;
        const int wid = v->nodes[final_id].word_id;
        if (w > 0) {                                                      // not stopped (:1181)
            bow_add(bow, (unsigned)wid, w, !tf);
            fv[(unsigned)nid].push_back((unsigned)i);                     // addFeature (FeatureVector.cpp:31-45)
            if (word) word[i] = wid;
            if (node) node[i] = nid;
        } else {
            if (word) word[i] = -1;
            if (node) node[i] = -1;
        }
    }
    if (tf && !bow.empty() && !must) {                                    // :1188-1194
        const double nd = (double)bow.size();
        for (auto& kv : bow) kv.second /= nd;
    }
    if (must) bow_normalize(bow, l2);
    int a = 0;
    for (auto& kv : bow) { bow_word[a] = (int32_t)kv.first; bow_val[a] = kv.second; ++a; }
    *n_words = a;
    int b = 0, o = 0;
    for (auto& kv : fv) {
        fv_node[b] = (int32_t)kv.first; fv_off[b] = o;
        for (unsigned id : kv.second) fv_idx[o++] = (int32_t)id;
        ++b;
    }
    fv_off[b] = o;
    *n_nodes = b;
    return 0;
}

/* the accumulation of transform() alone (same helpers as orc_bow_transform): for the checks against oracle/_ref */
int orc_bow_vector(int n, const uint32_t* word, const double* weight, int if_not_exist, int norm, int32_t* out_word, double* out_val)
{
    std::map<unsigned, double> bow;
    for (int i = 0; i < n; ++i) if (weight[i] > 0) bow_add(bow, word[i], weight[i], if_not_exist != 0);
    if (norm) bow_normalize(bow, norm == 2);
    int a = 0;
    for (auto& kv : bow) { out_word[a] = (int32_t)kv.first; out_val[a] = kv.second; ++a; }
    return a;
}

void orc_bow_score_l1(const int32_t* q_word, const double* q_val, int nq, const int32_t* db_off, const int32_t* db_word,
                      const double* db_val, int n_db, double* score)
{
    for (int k = 0; k < n_db; ++k) {
        const int32_t* w = db_word + db_off[k]; const double* x = db_val + db_off[k];
        const int nw = db_off[k + 1] - db_off[k];
        int a = 0, b = 0;
        double s = 0;
        while (a < nq && b < nw) {                                        // ScoringObject.cpp:34-59 (lower_bound = skip ahead)
            if (q_word[a] == w[b]) { const double vi = q_val[a], wi = x[b]; s += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi); ++a; ++b; }
            else if (q_word[a] < w[b]) ++a;
            else ++b;
        }
        score[k] = -s / 2.0;                                              // :64
    }
}

/* KeyFrameDatabase::DetectLoopCandidatesForCam (src/KeyFrameDatabase.cc:111-235, loop != 0) and DetectRelocalizationCandidates
   (:237-372, loop == 0) for one camera pair on flat arrays, statement by statement, with REAL inverted files: the database is
   given as the BowVectors of its entries in insertion order (CSR) + a dead flag per entry (erase()); the lists
   mvvInvertedFiles[word] are built from it in push_back order. st_query / st_words / st_score are the key frames' members
   (mnLoopQuery | mnRelocQuery, mnLoopWords | mnRelocWords, mLoopScore | mRelocScore): in/out, they persist across calls.
   connected[k]: entry k is in pKF->GetConnectedKeyFrames() (loop only). covis: GetBestCovisibilityKeyFrames(10) of every entry
   as CSR of entry ids. Returns the number of candidates written to out (entry ids, the reference's order). */
int orc_detect_candidates(int loop, int query_id, const int32_t* q_word, const double* q_val, int nq, int n_db, const int32_t* db_off,
                          const int32_t* db_word, const double* db_val, const uint8_t* dead, const uint8_t* connected, float minScore,
                          const int32_t* covis_off, const int32_t* covis_idx, int32_t* st_query, int32_t* st_words, float* st_score,
                          int32_t* out, int cap)
{
    std::map<int, std::vector<int>> inverted;                           /* word -> entries, insertion order */
    for (int k = 0; k < n_db; ++k)
        if (!dead[k]) for (int j = db_off[k]; j < db_off[k + 1]; ++j) inverted[db_word[j]].push_back(k);
    std::vector<int> lKFsSharingWords;
    for (int a = 0; a < nq; ++a) {                                      /* :128-149 / :257-272 */
        auto it = inverted.find(q_word[a]);
        if (it == inverted.end()) continue;
        for (int pKFi : it->second) {
            if (st_query[pKFi] != query_id) {
                st_words[pKFi] = 0;
                if (!loop || !connected[pKFi]) { st_query[pKFi] = query_id; lKFsSharingWords.push_back(pKFi); }
            }
            st_words[pKFi]++;
        }
    }
    if (lKFsSharingWords.empty()) return 0;
    int maxCommonWords = 0;
    for (int k : lKFsSharingWords) if (st_words[k] > maxCommonWords) maxCommonWords = st_words[k];
    const int minCommonWords = (int)(maxCommonWords * 0.8f);
    std::vector<std::pair<float, int>> lScoreAndMatch;
    for (int k : lKFsSharingWords) {
        if (st_words[k] > minCommonWords) {
            double sc;
            orc_bow_score_l1(q_word, q_val, nq, db_off + k, db_word, db_val, 1, &sc);      /* db_off + k: entry k as a one-entry CSR */
            const float si = (float)sc;
            st_score[k] = si;
            if (!loop || si >= minScore) lScoreAndMatch.push_back(std::make_pair(si, k));
        }
    }
    if (lScoreAndMatch.empty()) return 0;
    std::vector<std::pair<float, int>> lAccScoreAndMatch;
    float bestAccScore = loop ? minScore : 0.0f;
    for (auto& sm : lScoreAndMatch) {
        const int pKFi = sm.second;
        float bestScore = sm.first, accScore = sm.first;
        int pBestKF = pKFi;
        for (int j = covis_off[pKFi]; j < covis_off[pKFi + 1]; ++j) {
            const int pKF2 = covis_idx[j];
            if (loop) { if (!(st_query[pKF2] == query_id && st_words[pKF2] > minCommonWords)) continue; }
            else if (st_query[pKF2] != query_id) continue;
            accScore += st_score[pKF2];
            if (st_score[pKF2] > bestScore) { pBestKF = pKF2; bestScore = st_score[pKF2]; }
        }
        lAccScoreAndMatch.push_back(std::make_pair(accScore, pBestKF));
        if (accScore > bestAccScore) bestAccScore = accScore;
    }
    const float minScoreToRetain = 0.75f * bestAccScore;
    std::vector<int> added;
    int n_out = 0;
    for (auto& am : lAccScoreAndMatch) {
        if (am.first > minScoreToRetain) {
            const int k = am.second;
            if (std::find(added.begin(), added.end(), k) == added.end()) { if (n_out < cap) out[n_out] = k; ++n_out; added.push_back(k); }
        }
    }
    return n_out;
}

}  // extern "C"
