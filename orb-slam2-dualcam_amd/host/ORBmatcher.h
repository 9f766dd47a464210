// ORBmatcher.h -- header-only C++ mirror of the numeric core of ORB_SLAM2::ORBmatcher (reference
// include/ORBmatcher.h:45-309, src/ORBmatcher.cc) over the C ABI. The 14 Search*/Fuse members of the
// reference differ in how they pick candidates (BoW node, grid window); what they share -- and what runs on
// the GPU -- is the Hamming best/second-best loop, the TH_LOW/TH_HIGH + ratio accept test and the rotation
// histogram. Stateless and re-entrant like the reference's stack-constructed matcher.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "dcs_abi.h"

namespace ORB_SLAM2 {

// flat DBoW2::FeatureVector (Thirdparty/DBoW2/DBoW2/FeatureVector.h:23-24): ascending node ids + CSR of local feature indices
struct FeatureVectorCSR { std::vector<int32_t> nodes, off, idx; };

class ORBmatcher {
public:
    static const int TH_LOW = DCS_TH_LOW, TH_HIGH = DCS_TH_HIGH, HISTO_LENGTH = DCS_HISTO_LENGTH;

    ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

#ifdef DCS_WITH_REFERENCE_MODEL
    // The reference's own signatures (include/ORBmatcher.h:65-67, 79-82, 121-124, 196-200), bodies in ReferenceAdapters.h: the reference's gating per
    // map point, one call into the library for the candidate loops, the reference's assignment of the matches.
    int SearchByProjection(FramePtr pF, const std::vector<MapPointPtr>& vpMapPoints, const float th = 3);
    int SearchByProjection(FramePtr pCurrentFrame, const FramePtr pLastFrame, const float th, bool bMapScaled);
    int SearchByProjectionOnCam(FramePtr pFcur, const int& query, FramePtr pFlast, const float th);
    int SearchByBoWCrossCam(FramePtr F, const int& cF, KeyFramePtr pKF, const int& cKF, std::vector<MapPointPtr>& vpMapPointMatches);
#endif

    // SearchByBoWCrossCam(F, cF, KF, cKF, vpMapPointMatches) (ORBmatcher.cc:162-294) on flat per-camera arrays:
    // kfValid[i] = "KF feature i has a good MapPoint"; matchF[j] = matched KF feature or -1. Returns nmatches.
    int SearchByBoWCrossCam(const std::vector<uint8_t>& descF, const std::vector<float>& angF, const FeatureVectorCSR& fvF,
                            const std::vector<uint8_t>& descKF, const std::vector<float>& angKF, const std::vector<uint8_t>& kfValid,
                            const FeatureVectorCSR& fvKF, std::vector<int32_t>& matchF) const
    {
        const int nF = (int)angF.size(), nKF = (int)angKF.size();
        matchF.assign(nF, -1);
        int n = 0;
        check(dcs_search_by_bow(descKF.data(), angKF.data(), kfValid.data(), nKF, descF.data(), angF.data(), nF,
                                fvKF.nodes.data(), fvKF.off.data(), fvKF.idx.data(), (int)fvKF.nodes.size(),
                                fvF.nodes.data(), fvF.off.data(), fvF.idx.data(), (int)fvF.nodes.size(),
                                mfNNratio, mbCheckOrientation, matchF.data(), &n), "dcs_search_by_bow");
        return n;
    }

    // brute-force "Hamming BF + ratio test across the two camera streams": knn2 + TH + ratio + rotation histogram
    int MatchBruteForce(const std::vector<uint8_t>& descQ, const std::vector<dcs_keypoint>& kpQ, const std::vector<uint8_t>& descT,
                        const std::vector<dcs_keypoint>& kpT, std::vector<int32_t>& match, int th = TH_LOW) const
    {
        match.assign(kpQ.size(), -1);
        int n = 0;
        check(dcs_match_bf(descQ.data(), kpQ.data(), (int)kpQ.size(), descT.data(), kpT.data(), (int)kpT.size(), th, mfNNratio,
                           mbCheckOrientation, match.data(), &n), "dcs_match_bf");
        return n;
    }

    // SearchByProjection(pF, vpMapPoints, th) (ORBmatcher.cc:539-624): the caller fills `queries` from the map points that
    // passed Frame::isInFrustum (cam = mTrackProjCamera, u/v = mTrackProjX/Y, radius = RadiusByViewingCos(mTrackViewCos) [* th]
    // * mvScaleFactors[mnTrackScaleLevel], levels = mnTrackScaleLevel -/+ 1) and `frame` from pF (dcs_frame_grid builds the
    // CSR of mvGrids). matchOfQuery[i] = global feature index that now holds map point i, or -1. Returns nmatches.
    int SearchByProjection(const dcs_proj_frame& frame, const dcs_proj_queries& queries, std::vector<int32_t>& matchOfQuery,
                           std::vector<int32_t>& queryOfFeature) const
    {
        matchOfQuery.assign(queries.n > 0 ? queries.n : 1, -1); queryOfFeature.assign(frame.cam_off[frame.n_cams] > 0 ? frame.cam_off[frame.n_cams] : 1, -1);
        int n = 0;
        check(dcs_search_by_projection(&frame, &queries, TH_HIGH, mfNNratio, 0, matchOfQuery.data(), queryOfFeature.data(), &n), "dcs_search_by_projection");
        matchOfQuery.resize(queries.n); queryOfFeature.resize(frame.cam_off[frame.n_cams]);
        return n;
    }

    // SearchByProjectionOnCam(pFcurt, query, pFlast, th) (ORBmatcher.cc:954-1113): queries = the last frame's features of
    // camera `query` that hold a map point and project inside the image (:990-1013), radius = th * mvScaleFactors[octave],
    // levels = octave -/+ 1, angle = the last frame's keypoint angle. Best candidate only, rotation histogram if enabled.
    int SearchByProjectionOnCam(const dcs_proj_frame& frame, const dcs_proj_queries& queries, std::vector<int32_t>& matchOfQuery,
                                std::vector<int32_t>& queryOfFeature) const
    {
        matchOfQuery.assign(queries.n > 0 ? queries.n : 1, -1); queryOfFeature.assign(frame.cam_off[frame.n_cams] > 0 ? frame.cam_off[frame.n_cams] : 1, -1);
        int n = 0;
        check(dcs_search_by_projection(&frame, &queries, TH_HIGH, 0.f, mbCheckOrientation, matchOfQuery.data(), queryOfFeature.data(), &n),
              "dcs_search_by_projection");
        matchOfQuery.resize(queries.n); queryOfFeature.resize(frame.cam_off[frame.n_cams]);
        return n;
    }

    // SearchByProjectionOnCam(pF, query, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:812-951): queries = pKF's map points that
    // are not bad / not already found and project into camera `query` (:849-876), radius = th * mvScaleFactors[nPredictedLevel],
    // levels = nPredictedLevel -/+ 1, angle = pKF->mvTotalKeysUn[i].angle, frame.taken = (pF->mvpMapPoints[g] != NULL).
    int SearchByProjectionOnCam(const dcs_proj_frame& frame, const dcs_proj_queries& queries, int ORBdist, std::vector<int32_t>& matchOfQuery,
                                std::vector<int32_t>& queryOfFeature) const
    {
        matchOfQuery.assign(queries.n > 0 ? queries.n : 1, -1); queryOfFeature.assign(frame.cam_off[frame.n_cams] > 0 ? frame.cam_off[frame.n_cams] : 1, -1);
        int n = 0;
        check(dcs_search_by_projection(&frame, &queries, ORBdist, 0.f, mbCheckOrientation, matchOfQuery.data(), queryOfFeature.data(), &n),
              "dcs_search_by_projection");
        matchOfQuery.resize(queries.n); queryOfFeature.resize(frame.cam_off[frame.n_cams]);
        return n;
    }

    // SearchByProjection(pKF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536, loop closing): queries = the candidate
    // map points that pass :448-490 (depth, IsInImage, distance range, viewing angle), radius = th * mvScaleFactors[nPredictedLevel],
    // levels nPredictedLevel - 1 .. nPredictedLevel, best only, TH_LOW; a matched feature is taken for the queries that follow
    // (:506, :527). The window is the key frame's own (KeyFrame::GetFeaturesInArea with its camera-local index quirk, KeyFrame.cc:756,
    // reproduced by dcs_search_by_projection_kf). keyFrame.taken[cam_off[query] + l] = (vpMatched[l] != NULL): the reference indexes
    // vpMatched with the camera-local index l, and vpMatched[l] = the query matched to queryOfFeature[cam_off[query] + l].
    int SearchByProjection(const dcs_proj_frame& keyFrame, const dcs_proj_queries& queries, int th, std::vector<int32_t>& matchOfQuery,
                           std::vector<int32_t>& queryOfFeature) const
    {
        (void)th;                                            // folded into queries.radius by the caller
        matchOfQuery.assign(queries.n > 0 ? queries.n : 1, -1); queryOfFeature.assign(keyFrame.cam_off[keyFrame.n_cams] > 0 ? keyFrame.cam_off[keyFrame.n_cams] : 1, -1);
        int n = 0;
        check(dcs_search_by_projection_kf(&keyFrame, &queries, TH_LOW, matchOfQuery.data(), queryOfFeature.data(), &n), "dcs_search_by_projection_kf");
        matchOfQuery.resize(queries.n); queryOfFeature.resize(keyFrame.cam_off[keyFrame.n_cams]);
        return n;
    }

    // The candidate loop of Fuse(pKF, vpMapPoints, th) (ORBmatcher.cc:1431-1556; reprojGate = true: e2 * mvInvLevelSigma2[octave]
    // > 5.99 skips, :1503-1509) and of Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1560-1706; reprojGate = false): queries = the
    // map points that pass the projection tests of their camera (:1451-1480), levels nPredictedLevel - 1 .. nPredictedLevel.
    // bestIdx[i] = global key point whose map point is to be replaced / that receives the observation (:1530-1552), or -1.
    // Queries are independent of each other; the Replace / AddObservation bookkeeping stays with the caller. Returns nFused.
    int Fuse(const dcs_proj_frame& keyFrame, const dcs_proj_queries& queries, const std::vector<float>& invLevelSigma2, bool reprojGate,
             std::vector<int32_t>& bestIdx) const
    {
        bestIdx.assign(queries.n > 0 ? queries.n : 1, -1);
        int n = 0;
        check(dcs_search_in_window(&keyFrame, &queries, TH_LOW, 1, reprojGate ? invLevelSigma2.data() : nullptr, (int)invLevelSigma2.size(), bestIdx.data(),
                                   nullptr, &n), "dcs_search_in_window");
        bestIdx.resize(queries.n);
        return n;
    }

    // SearchByProjection(pKF, vpMapPoints, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:693-799), one camera per call: levels
    // nPredictedLevel -/+ 1, best only, bestDist <= ORBdist.
    int SearchByProjection(const dcs_proj_frame& keyFrame, const dcs_proj_queries& queries, int ORBdist, std::vector<int32_t>& bestIdx) const
    {
        bestIdx.assign(queries.n > 0 ? queries.n : 1, -1);
        int n = 0;
        check(dcs_search_in_window(&keyFrame, &queries, ORBdist, 1, nullptr, 0, bestIdx.data(), nullptr, &n), "dcs_search_in_window");
        bestIdx.resize(queries.n);
        return n;
    }

    // SearchBySim3CrossCam(pKF1, cam1, pKF2, cam2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1713-1965): queries1in2[i1] = map
    // point of KF1's local feature i1 projected into KF2's camera (:1793-1832), queries2in1 the other way (:1872-1911); both carry
    // levels nPredictedLevel - 1 .. nPredictedLevel and cam = the searched key frame's camera. match12[i1] = camera-local KF2
    // feature after the agreement check (:1950-1964), or -1. Returns nFound.
    int SearchBySim3CrossCam(const dcs_proj_frame& keyFrame1, const dcs_proj_queries& queries2in1, const dcs_proj_frame& keyFrame2,
                             const dcs_proj_queries& queries1in2, std::vector<int32_t>& match12) const
    {
        std::vector<int32_t> m1(queries1in2.n > 0 ? queries1in2.n : 1, -1), m2(queries2in1.n > 0 ? queries2in1.n : 1, -1);
        int n = 0;
        check(dcs_search_in_window(&keyFrame2, &queries1in2, TH_HIGH, 1, nullptr, 0, m1.data(), nullptr, &n), "dcs_search_in_window");
        check(dcs_search_in_window(&keyFrame1, &queries2in1, TH_HIGH, 1, nullptr, 0, m2.data(), nullptr, &n), "dcs_search_in_window");
        match12.assign(queries1in2.n, -1);
        int nFound = 0;
        for (int i1 = 0; i1 < queries1in2.n; ++i1) {
            if (m1[i1] < 0) continue;
            const int i2 = m1[i1] - keyFrame2.cam_off[queries1in2.cam[i1]];              // vnMatch1[i1] = bestIdx2local
            if (i2 < queries2in1.n && m2[i2] >= 0 && m2[i2] - keyFrame1.cam_off[queries2in1.cam[i2]] == i1) { match12[i1] = i2; ++nFound; }
        }
        return nFound;
    }

    // SearchForInitialization(pF1, pF2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:1117-1251): frame2 = pF2, queries =
    // pF1's key points in global order (valid = camera CAP and octave 0; u, v = vbPrevMatched; radius = windowSize; min = max
    // level = octave). vnMatches12[i] = global key point of pF2 or -1; the caller then refreshes vbPrevMatched (:1244-1246).
    int SearchForInitialization(const dcs_proj_frame& frame2, const dcs_proj_queries& queries, std::vector<int>& vnMatches12) const
    {
        std::vector<int32_t> m(queries.n > 0 ? queries.n : 1, -1);
        int n = 0;
        check(dcs_search_for_initialization(&frame2, &queries, mfNNratio, mbCheckOrientation, m.data(), &n), "dcs_search_for_initialization");
        vnMatches12.assign(m.begin(), m.begin() + queries.n);
        return n;
    }

    // Frame::isInFrustum (Frame.cc:244-312) for all local map points of Tracking::SearchLocalPoints (Tracking.cc:1617-1680) in one
    // call, fused with PredictScale and the window of SearchByProjection (:557-565). `frame` carries Tsw = mvExtrinsics[c] * mTcw
    // and GetCameraCenter(c) as the caller's cv::Mat code forms them; the outputs fill the valid / cam / u / v / radius /
    // level -+ 1 columns of dcs_proj_queries (and pMP->mbTrackInView, mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos).
    struct FrustumResult { std::vector<uint8_t> inView; std::vector<int32_t> cam, level; std::vector<float> u, v, viewCos, radius; };
    static void IsInFrustum(const dcs_frustum_frame& frame, const std::vector<float>& worldPos, const std::vector<float>& normal,
                            const std::vector<float>& minDistance, const std::vector<float>& maxDistance, const std::vector<uint8_t>& candidate,
                            float viewingCosLimit, float th, FrustumResult& r)
    {
        const int n = (int)minDistance.size();
        const size_t m = n > 0 ? (size_t)n : 1;
        r.inView.assign(m, 0); r.cam.assign(m, -1); r.level.assign(m, 0); r.u.assign(m, 0.f); r.v.assign(m, 0.f); r.viewCos.assign(m, 0.f); r.radius.assign(m, 0.f);
        check(dcs_is_in_frustum(&frame, n, worldPos.data(), normal.data(), minDistance.data(), maxDistance.data(), candidate.empty() ? nullptr : candidate.data(),
                                viewingCosLimit, th, r.inView.data(), r.cam.data(), r.u.data(), r.v.data(), r.viewCos.data(), r.level.data(), r.radius.data()),
              "dcs_is_in_frustum");
    }

    // best / second-best distances of every query (the loop at ORBmatcher.cc:208-231 over all candidates)
    static void Knn2(const std::vector<uint8_t>& q, const std::vector<uint8_t>& t, std::vector<int32_t>& bestIdx,
                     std::vector<int32_t>& bestDist, std::vector<int32_t>& secondDist)
    {
        const int nq = (int)q.size() / 32, nt = (int)t.size() / 32;
        bestIdx.assign(nq, -1); bestDist.assign(nq, 256); secondDist.assign(nq, 256);
        check(dcs_hamming_knn2(q.data(), nq, t.data(), nt, nullptr, bestIdx.data(), bestDist.data(), secondDist.data()), "dcs_hamming_knn2");
    }

protected:
    static void check(int rc, const char* what) { if (rc != DCS_OK) throw std::runtime_error(std::string(what) + ": " + dcs_last_error()); }
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2
