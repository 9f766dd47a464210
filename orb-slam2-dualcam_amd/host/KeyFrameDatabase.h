// KeyFrameDatabase.h -- header-only mirror of ORB_SLAM2::KeyFrameDatabase (include/KeyFrameDatabase.h, src/KeyFrameDatabase.cc) for ONE
// camera over the C ABI (dcs_kfdb_*): add / erase / clear and the two candidate searches, DetectLoopCandidatesForCam (:111-235) and
// DetectRelocalizationCandidates (:237-372). The GPU does what touches every key frame of the map -- the walk over the inverted
// files (shared-word counts, order of first encounter) and the L1 scores; the thresholds, the covisibility accumulation and the
// final cut run here on those three arrays, statement by statement like the reference. Key frames are entry ids (the order of
// add()); the members the reference keeps in KeyFrame (mnLoopQuery, mnLoopWords, mLoopScore, mnRelocQuery, mnRelocWords,
// mRelocScore) live in a State that several cameras' databases may share, as the key frames are shared in the reference.
#ifndef DCS_HOST_KEYFRAMEDATABASE_H
#define DCS_HOST_KEYFRAMEDATABASE_H

#include <algorithm>
#include <cstdint>
#include <list>
#include <numeric>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dcs_abi.h"

namespace ORB_SLAM2 {

class KeyFrameDatabase {
public:
    struct State {                                           // per key frame (entry id), persists across queries
        std::vector<long> mnQuery; std::vector<int> mnWords; std::vector<float> mScore;
        void grow(size_t n) { if (mnQuery.size() < n) { mnQuery.resize(n, -1); mnWords.resize(n, 0); mScore.resize(n, 0.f); } }
    };

    KeyFrameDatabase() { check(dcs_kfdb_create(&h_), "dcs_kfdb_create"); }
    ~KeyFrameDatabase() { dcs_kfdb_destroy(h_); }
    KeyFrameDatabase(const KeyFrameDatabase&) = delete;
    KeyFrameDatabase& operator=(const KeyFrameDatabase&) = delete;

    // add(pKF): pKF->mvBowVec[c] as ascending (word, value) columns; returns the entry id
    int add(const std::vector<int32_t>& words, const std::vector<double>& values)
    {
        int id = -1;
        check(dcs_kfdb_add(h_, words.data(), values.data(), (int)words.size(), &id), "dcs_kfdb_add");
        return id;
    }
    void erase(int entry) { check(dcs_kfdb_erase(h_, entry), "dcs_kfdb_erase"); }
    void clear() { check(dcs_kfdb_clear(h_), "dcs_kfdb_clear"); }
    int size() const { int n = 0; check(dcs_kfdb_size(h_, &n), "dcs_kfdb_size"); return n; }

    // DetectLoopCandidatesForCam(pKF, queryC, respC, minScore): query = pKF->mvBowVec[queryC], this = mvvInvertedFiles[respC];
    // connected[k] != 0: entry k is in pKF->GetConnectedKeyFrames(); covis[k] = GetBestCovisibilityKeyFrames(10) of entry k
    std::vector<int> DetectLoopCandidates(long kfId, const std::vector<int32_t>& qWords, const std::vector<double>& qValues,
                                          const std::vector<uint8_t>& connected, const std::vector<std::vector<int>>& covis, float minScore, State& st) const
    { return detect(true, kfId, qWords, qValues, &connected, covis, minScore, st); }

    // DetectRelocalizationCandidates(F, queryC, respC): query = F->mvBowVec[queryC]
    std::vector<int> DetectRelocalizationCandidates(long frameId, const std::vector<int32_t>& qWords, const std::vector<double>& qValues,
                                                    const std::vector<std::vector<int>>& covis, State& st) const
    { return detect(false, frameId, qWords, qValues, nullptr, covis, 0.f, st); }

private:
    std::vector<int> detect(bool loop, long id, const std::vector<int32_t>& qw, const std::vector<double>& qv, const std::vector<uint8_t>* connected,
                            const std::vector<std::vector<int>>& covis, float minScore, State& st) const
    {
        const int n = size();
        st.grow((size_t)n);
        std::vector<int32_t> common((size_t)std::max(n, 1)), first((size_t)std::max(n, 1));
        std::vector<float> score((size_t)std::max(n, 1));
        check(dcs_kfdb_query(h_, qw.data(), qv.data(), (int)qw.size(), common.data(), first.data(), score.data()), "dcs_kfdb_query");
        // the walk over the inverted files (:128-149 / :257-272): entries in (first shared word, entry id) order
        std::vector<int> order;
        for (int k = 0; k < n; ++k) if (common[k] > 0) order.push_back(k);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
        std::list<int> lKFsSharingWords;
        for (int k : order) {
            if (st.mnQuery[k] != id) {
                if (loop && (*connected)[k]) st.mnWords[k] = 1;            // never marked: the count restarts at every word
                else { st.mnQuery[k] = id; st.mnWords[k] = common[k]; lKFsSharingWords.push_back(k); }
            } else st.mnWords[k] += common[k];                              // marked by an earlier camera pair of the same query
        }
        if (lKFsSharingWords.empty()) return {};
        int maxCommonWords = 0;
        for (int k : lKFsSharingWords) if (st.mnWords[k] > maxCommonWords) maxCommonWords = st.mnWords[k];
        const int minCommonWords = (int)(maxCommonWords * 0.8f);
        std::list<std::pair<float, int>> lScoreAndMatch;
        for (int k : lKFsSharingWords) {
            if (st.mnWords[k] > minCommonWords) {
                const float si = score[k];
                st.mScore[k] = si;
                if (!loop || si >= minScore) lScoreAndMatch.push_back(std::make_pair(si, k));
            }
        }
        if (lScoreAndMatch.empty()) return {};
        std::list<std::pair<float, int>> lAccScoreAndMatch;
        float bestAccScore = loop ? minScore : 0.f;
        for (const auto& sm : lScoreAndMatch) {
            const int pKFi = sm.second;
            float bestScore = sm.first, accScore = sm.first;
            int pBestKF = pKFi;
            for (int pKF2 : covis[(size_t)pKFi]) {
                if (loop) { if (!(st.mnQuery[pKF2] == id && st.mnWords[pKF2] > minCommonWords)) continue; }
                else if (st.mnQuery[pKF2] != id) continue;
                accScore += st.mScore[pKF2];
                if (st.mScore[pKF2] > bestScore) { pBestKF = pKF2; bestScore = st.mScore[pKF2]; }
            }
            lAccScoreAndMatch.push_back(std::make_pair(accScore, pBestKF));
            if (accScore > bestAccScore) bestAccScore = accScore;
        }
        const float minScoreToRetain = 0.75f * bestAccScore;
        std::vector<int> out;
        for (const auto& am : lAccScoreAndMatch)
            if (am.first > minScoreToRetain && std::find(out.begin(), out.end(), am.second) == out.end()) out.push_back(am.second);
        return out;
    }
    static void check(int rc, const char* what) { if (rc != DCS_OK) throw std::runtime_error(std::string(what) + ": " + dcs_last_error()); }
    dcs_kfdb* h_ = nullptr;
};

}  // namespace ORB_SLAM2
#endif
