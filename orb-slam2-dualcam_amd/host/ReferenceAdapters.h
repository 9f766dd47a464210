// ReferenceAdapters.h -- the reference's OWN member signatures on the matcher and optimiser seams (-DDCS_WITH_REFERENCE_MODEL).
//
// host/ORBmatcher.h and host/Optimizer.h expose the C ABI on flat arrays; north_star asks for "the existing ORBextractor::operator(),
// ORBmatcher and Optimizer C++ interfaces ... so Tracking / LocalMapping threads are unchanged". This header is that layer, written as real
// C++ against the accessors of the reference's data model (include/Frame.h, KeyFrame.h, MapPoint.h, Map.h, Cameras.h): every member below has
// the reference's exact parameter list (include/Optimizer.h:47-74, include/ORBmatcher.h:45-281), its body is the reference's own gather with
// the g2o graph / the candidate loop replaced by ONE call into the library, and the write-back the reference performs afterwards. It is
// compiled wherever the reference's headers are (define DCS_WITH_REFERENCE_MODEL and include Frame.h / KeyFrame.h / MapPoint.h / Map.h / Cameras.h
// before ORBmatcher.h / Optimizer.h); in this repository it goes through `g++ -fsyntax-only -Werror` against a declaration-only stand-in of
// those classes (tests/cpp/slam_model_stub.h, tests/test_reference_model_syntax.py) together with the reference's own call lines
// (LocalMapping.cc:103; Tracking.cc:822, 1321, 1387, 1680). It stays ON the seam: no Tracking / Map logic is re-implemented, and nothing here
// computes -- the numerics are the library's.
//
// Index conventions: poses ascending KeyFrame::mnId and points ascending MapPoint::mnId (g2o orders its vertices by id);
// edges in the order the reference inserts them (map points in list order, observations in std::map order); a frame's features in the global
// order of mvTotalKeysUn (camera after camera: Frame::GetGlobalIdxByLocal, Frame.cc:444-450).
#pragma once
#ifdef DCS_WITH_REFERENCE_MODEL

#include "ORBmatcher.h"
#include "Optimizer.h"

#include <algorithm>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <unordered_map>
#include <vector>

namespace ORB_SLAM2 {

namespace dcs_adapters {

inline dcs_ba_camera rigCamera(float fx, float fy, float cx, float cy, const cv::Mat& camExt, const cv::Mat& camExtAdj)
{
    dcs_ba_camera cam{};
    cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy;
    dcs_pose_from_matrix(reinterpret_cast<const float*>(camExt.data), cam.ext);        // Converter::toSE3Quat(camExt) (Converter.cc:58-68)
    const float* a = reinterpret_cast<const float*>(camExtAdj.data);                   // Converter::toMatrix6d(camExtAdj) (:104-112): the 6x6 CV_32F as it is
    for (int i = 0; i < 36; ++i) cam.adj[i] = (double)a[i];
    return cam;
}
inline void poseOf(const cv::Mat& Tcw, std::vector<double>& out)
{
    double p[7];
    dcs_pose_from_matrix(reinterpret_cast<const float*>(Tcw.data), p);                  // Converter::toSE3Quat
    out.insert(out.end(), p, p + 7);
}
inline cv::Mat poseMat(const double* p7)
{
    float M[16];
    dcs_pose_to_matrix(p7, M);                                                          // Converter::toCvMat(SE3Quat) (:70-74)
    return cv::Mat(4, 4, CV_32F, M).clone();
}
inline cv::Mat pointMat(const double* p3)
{
    float v[3] = {(float)p3[0], (float)p3[1], (float)p3[2]};                            // Converter::toCvMat(Eigen::Vector3d) (:124-131)
    return cv::Mat(3, 1, CV_32F, v).clone();
}

// a frame's side of the projection searches: key points, descriptors, "already holds an observed map point", image bounds, grid
struct FrameArrays {
    std::vector<int32_t> camOff, octave, gridOff, gridIdx;
    std::vector<float> x, y, angle, minX, minY, wInv, hInv;
    std::vector<uint8_t> desc, taken;
    dcs_proj_frame view{};
    void fill(const FramePtr& pF)
    {
        const int C = pF->mnCams;
        camOff.assign(1, 0);
        for (int c = 0; c < C; ++c) camOff.push_back(camOff.back() + pF->mvN[c]);
        const int N = camOff.back();
        x.resize(N); y.resize(N); angle.resize(N); octave.resize(N); desc.resize((size_t)N * 32); taken.resize(N);
        for (int c = 0; c < C; ++c) {
            const std::vector<cv::KeyPoint>& keys = pF->mvvkeysUnTemp[c];
            const unsigned char* rows = pF->mvDescriptors[c].data;
            for (int i = 0; i < pF->mvN[c]; ++i) {
                const int g = camOff[c] + i;
                x[g] = keys[i].pt.x; y[g] = keys[i].pt.y; angle[g] = keys[i].angle; octave[g] = keys[i].octave;
                std::copy(rows + (size_t)i * 32, rows + (size_t)i * 32 + 32, desc.begin() + (size_t)g * 32);
                const MapPointPtr& held = pF->mvpMapPoints[g];
                taken[g] = (held && held->Observations() > 0) ? 1 : 0;                  // ORBmatcher.cc:588-590, 1046-1048
            }
            minX.push_back(Frame::mvMinX[c]); minY.push_back(Frame::mvMinY[c]);
            wInv.push_back(Frame::mvfGridElementWidthInv[c]); hInv.push_back(Frame::mvfGridElementHeightInv[c]);
        }
        gridOff.resize((size_t)C * DCS_GRID_COLS * DCS_GRID_ROWS + 1); gridIdx.resize(N > 0 ? N : 1);
        int entries = 0;
        if (dcs_frame_grid(C, camOff.data(), x.data(), y.data(), minX.data(), minY.data(), wInv.data(), hInv.data(), gridOff.data(), gridIdx.data(), &entries) != DCS_OK)
            throw std::runtime_error(std::string("dcs_frame_grid: ") + dcs_last_error());
        view.n_cams = C; view.cam_off = camOff.data(); view.kp_x = x.data(); view.kp_y = y.data(); view.kp_octave = octave.data(); view.kp_angle = angle.data();
        view.desc = desc.data(); view.taken = taken.data(); view.min_x = minX.data(); view.min_y = minY.data(); view.grid_w_inv = wInv.data();
        view.grid_h_inv = hInv.data(); view.grid_off = gridOff.data(); view.grid_idx = gridIdx.data();
    }
};

struct QueryArrays {
    std::vector<uint8_t> valid, desc;
    std::vector<int32_t> cam, minLevel, maxLevel;
    std::vector<float> u, v, radius, angle;
    dcs_proj_queries view{};
    void reserve(size_t n) { valid.assign(n, 0); cam.assign(n, 0); minLevel.assign(n, 0); maxLevel.assign(n, 0); u.assign(n, 0.f); v.assign(n, 0.f); radius.assign(n, 0.f); angle.assign(n, 0.f); desc.assign(n * 32, 0); }
    void finish()
    {
        view.n = (int32_t)valid.size(); view.valid = valid.data(); view.cam = cam.data(); view.u = u.data(); view.v = v.data(); view.radius = radius.data();
        view.min_level = minLevel.data(); view.max_level = maxLevel.data(); view.desc = desc.data(); view.angle = angle.data();
    }
    void setDescriptor(size_t i, const cv::Mat& d) { std::copy(d.data, d.data + 32, desc.begin() + i * 32); }
};

inline FeatureVectorCSR flatten(const DBoW2::FeatureVector& fv)
{
    FeatureVectorCSR out;
    out.off.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {      // ascending node id, indices in insertion order
        out.nodes.push_back((int32_t)it->first);
        for (size_t k = 0; k < it->second.size(); ++k) out.idx.push_back((int32_t)it->second[k]);
        out.off.push_back((int32_t)out.idx.size());
    }
    return out;
}

// the flat problem of LocalBundleAdjustment / BundleAdjustment from key frames and map points + the write-back keys
struct BaGather {
    std::vector<KeyFramePtr> kfs;               // ascending mnId
    std::vector<MapPointPtr> mps;               // ascending mnId
    std::vector<KeyFramePtr> edgeKF; std::vector<MapPointPtr> edgeMP;
    LocalBAProblem pb;
    void run(const std::vector<KeyFramePtr>& keyFrames, const std::vector<uint8_t>& fixedFlag, const std::vector<MapPointPtr>& points, const KeyFramePtr& rigOwner)
    {
        std::vector<size_t> order(keyFrames.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return keyFrames[a]->mnId < keyFrames[b]->mnId; });
        std::unordered_map<KeyFrame*, int32_t> kfIndex;
        for (size_t k = 0; k < order.size(); ++k) {
            const KeyFramePtr& kf = keyFrames[order[k]];
            kfIndex[kf.get()] = (int32_t)kfs.size();
            kfs.push_back(kf);
            poseOf(kf->GetPose(), pb.poses);                                                // vSE3->setEstimate(Converter::toSE3Quat(pKFi->GetPose())) (:479, :492)
            pb.poseFixed.push_back(fixedFlag[order[k]]);
        }
        const int C = rigOwner->mnCams;
        for (int c = 0; c < C; ++c)                                                           // e->fx .. e->setExtrinsic(ExtSE3, ExtAdj) (:561-571), once per camera
            pb.cams.push_back(rigCamera(rigOwner->mvfx[c], rigOwner->mvfy[c], rigOwner->mvcx[c], rigOwner->mvcy[c], rigOwner->mpCameras->getExtrinsici(c),
                                        rigOwner->mpCameras->getExtrinsicAdji(c)));
        std::vector<size_t> porder(points.size());
        for (size_t i = 0; i < porder.size(); ++i) porder[i] = i;
        std::sort(porder.begin(), porder.end(), [&](size_t a, size_t b) { return points[a]->mnId < points[b]->mnId; });
        std::unordered_map<MapPoint*, int32_t> mpIndex;
        for (size_t k = 0; k < porder.size(); ++k) {
            const MapPointPtr& mp = points[porder[k]];
            mpIndex[mp.get()] = (int32_t)mps.size();
            mps.push_back(mp);
            const cv::Mat Xw = mp->GetWorldPos();                                            // Converter::toVector3d (:523)
            for (int d = 0; d < 3; ++d) pb.points.push_back((double)Xw.at<float>(d));
        }
        for (size_t i = 0; i < points.size(); ++i) {                                          // the reference's edge order: points as listed, observations in map order
            const MapPointPtr& mp = points[i];
            const std::map<KeyFramePtr, size_t> observations = mp->GetObservations();
            for (std::map<KeyFramePtr, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
                const KeyFramePtr& kf = mit->first;
                const std::unordered_map<KeyFrame*, int32_t>::const_iterator at = kfIndex.find(kf.get());
                if (kf->isBad() || at == kfIndex.end()) continue;                            // (:536; BundleAdjustment: pKF->mnId > maxKFid, :150)
                const cv::KeyPoint& kpUn = kf->mvTotalKeysUn[mit->second];
                pb.edgePose.push_back(at->second); pb.edgePoint.push_back(mpIndex[mp.get()]);
                pb.edgeCam.push_back(kf->keypointToCam[mit->second]);                       // viewedkpCam (:541)
                pb.obs.push_back((double)kpUn.pt.x); pb.obs.push_back((double)kpUn.pt.y);
                pb.invSigma2.push_back((double)kf->mvInvLevelSigma2[kpUn.octave]);
                edgeKF.push_back(kf); edgeMP.push_back(mp);
            }
        }
    }
};

}  // namespace dcs_adapters

// ------------------------------------------------------------------------------------------------------------------ Optimizer
// void static LocalBundleAdjustment(KeyFramePtr pKF, bool *pbStopFlag, MapPtr pMap, size_t fixId)   (include/Optimizer.h:55, src/Optimizer.cc:407-696)
inline void Optimizer::LocalBundleAdjustment(KeyFramePtr pKF, bool* pbStopFlag, MapPtr pMap, size_t fixId)
{
    // ---- :409-458 as they are: local key frames, their map points, the fixed cameras
    std::list<KeyFramePtr> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    const std::vector<KeyFramePtr> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (int i = 0, iend = (int)vNeighKFs.size(); i < iend; i++) {
        KeyFramePtr pKFi = vNeighKFs[i];
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    std::list<MapPointPtr> lLocalMapPoints;
    for (std::list<KeyFramePtr>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
        std::vector<MapPointPtr> vpMPs = (*lit)->GetMapPointMatches();
        for (std::vector<MapPointPtr>::iterator vit = vpMPs.begin(), vend = vpMPs.end(); vit != vend; vit++) {
            MapPointPtr pMP = *vit;
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
        }
    }
    std::list<KeyFramePtr> lFixedCameras;
    for (std::list<MapPointPtr>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
        std::map<KeyFramePtr, size_t> observations = (*lit)->GetObservations();
        for (std::map<KeyFramePtr, size_t>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
            KeyFramePtr pKFi = mit->first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
            }
        }
    }
    // ---- :460-580 (the g2o graph) -> the flat problem
    std::vector<KeyFramePtr> keyFrames(lLocalKeyFrames.begin(), lLocalKeyFrames.end());
    std::vector<uint8_t> fixedFlag;
    for (size_t i = 0; i < keyFrames.size(); ++i) fixedFlag.push_back(keyFrames[i]->mnId == fixId ? 1 : 0);                      // vSE3->setFixed(pKFi->mnId == fixId) (:483)
    for (std::list<KeyFramePtr>::iterator lit = lFixedCameras.begin(); lit != lFixedCameras.end(); ++lit) { keyFrames.push_back(*lit); fixedFlag.push_back(1); }   // :496
    dcs_adapters::BaGather g;
    g.run(keyFrames, fixedFlag, std::vector<MapPointPtr>(lLocalMapPoints.begin(), lLocalMapPoints.end()), pKF);
    if (pbStopFlag && *pbStopFlag) return;                                                                                             // :582-584
    // ---- :586-621: optimize(5), level-1 classification, optimize(10) -- on the device, pbStopFlag polled like g2o's forceStopFlag
    LocalBAResult out;
    LocalBundleAdjustment(g.pb, pbStopFlag, out);
    // ---- :641-693 as they are: erase the outlier observations, write the estimates back under the map mutex
    std::vector<std::pair<KeyFramePtr, MapPointPtr> > vToErase;
    for (size_t i = 0; i < g.edgeKF.size(); ++i) {
        if (g.edgeMP[i]->isBad()) continue;
        if (out.edgeOutlier[i]) vToErase.push_back(std::make_pair(g.edgeKF[i], g.edgeMP[i]));                                          // e->chi2() > 5.991 || !e->isDepthPositive() (:653)
    }
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (size_t i = 0; i < vToErase.size(); i++) {
        KeyFramePtr pKFi = vToErase[i].first;
        MapPointPtr pMPi = vToErase[i].second;
        pKFi->EraseMapPointMatch(pMPi);
        pMPi->EraseObservation(pKFi);
    }
    for (size_t k = 0; k < g.kfs.size(); ++k)
        if (g.kfs[k]->mnBALocalForKF == pKF->mnId) g.kfs[k]->SetPose(dcs_adapters::poseMat(&out.poses[7 * k]));                       // the local key frames (:676-682); fixed cameras keep theirs
    for (size_t k = 0; k < g.mps.size(); ++k) {
        g.mps[k]->SetWorldPos(dcs_adapters::pointMat(&out.points[3 * k]));
        g.mps[k]->UpdateNormalAndDepth();
    }
}

// void static BundleAdjustment(const std::vector<KeyFramePtr>&, const std::vector<MapPointPtr>&, unsigned long fixId, int nIterations, bool*, const unsigned long nLoopKF, const bool bRobust)
// (include/Optimizer.h:49-51, src/Optimizer.cc:70-248)
inline void Optimizer::BundleAdjustment(const std::vector<KeyFramePtr>& vpKFs, const std::vector<MapPointPtr>& vpMP, unsigned long fixId, int nIterations,
                                        bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust)
{
    std::vector<KeyFramePtr> keyFrames;
    std::vector<uint8_t> fixedFlag;
    for (size_t i = 0; i < vpKFs.size(); i++) {
        if (vpKFs[i]->isBad()) continue;                                                                                               // :93-94
        keyFrames.push_back(vpKFs[i]); fixedFlag.push_back(vpKFs[i]->mnId == fixId ? 1 : 0);
    }
    if (keyFrames.empty()) return;
    std::vector<MapPointPtr> points;
    for (size_t i = 0; i < vpMP.size(); i++) {
        if (vpMP[i]->isBad()) continue;                                                                                                // :116-117
        // a point no included key frame observes is left out (:184-192: removeVertex)
        const std::map<KeyFramePtr, size_t> observations = vpMP[i]->GetObservations();
        bool seen = false;
        for (std::map<KeyFramePtr, size_t>::const_iterator mit = observations.begin(); mit != observations.end() && !seen; ++mit)
            seen = !mit->first->isBad() && std::find(keyFrames.begin(), keyFrames.end(), mit->first) != keyFrames.end();
        if (seen) points.push_back(vpMP[i]);
    }
    dcs_adapters::BaGather g;
    g.run(keyFrames, fixedFlag, points, keyFrames[0]);
    LocalBAResult out;
    BundleAdjustment(g.pb, nIterations, pbStopFlag, bRobust, out);                                                                     // optimizer.optimize(nIterations) (:196)
    for (size_t k = 0; k < g.kfs.size(); ++k) {                                                                                        // :201-218
        if (nLoopKF == 0) g.kfs[k]->SetPose(dcs_adapters::poseMat(&out.poses[7 * k]));
        else { g.kfs[k]->mTcwGBA = dcs_adapters::poseMat(&out.poses[7 * k]); g.kfs[k]->mnBAGlobalForKF = nLoopKF; }
    }
    for (size_t k = 0; k < g.mps.size(); ++k) {                                                                                        // :221-245
        if (nLoopKF == 0) { g.mps[k]->SetWorldPos(dcs_adapters::pointMat(&out.points[3 * k])); g.mps[k]->UpdateNormalAndDepth(); }
        else { g.mps[k]->mPosGBA = dcs_adapters::pointMat(&out.points[3 * k]); g.mps[k]->mnBAGlobalForKF = nLoopKF; }
    }
}

// void static GlobalBundleAdjustemnt(MapPtr pMap, int nIterations, unsigned long fixId, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust)   (:61-67)
inline void Optimizer::GlobalBundleAdjustemnt(MapPtr pMap, int nIterations, unsigned long fixId, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust)
{
    std::vector<KeyFramePtr> vpKFs = pMap->GetAllKeyFrames();
    std::vector<MapPointPtr> vpMP = pMap->GetAllMapPoints();
    BundleAdjustment(vpKFs, vpMP, fixId, nIterations, pbStopFlag, nLoopKF, bRobust);
}

// int static PoseOptimization(FramePtr pFrame)   (include/Optimizer.h:56, src/Optimizer.cc:250-405)
inline int Optimizer::PoseOptimization(FramePtr pFrame)
{
    PoseProblem pp;
    dcs_adapters::poseOf(pFrame->mTcw, pp.poses);                                                                                      // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw)) (:265)
    pp.edgeOff.push_back(0);
    const int N = pFrame->totalN;
    std::vector<size_t> vnIndexEdgeMono;
    int nInitialCorrespondences = 0;
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);                                                                     // :286
        for (int i = 0; i < N; i++) {
            MapPointPtr pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            nInitialCorrespondences++;
            pFrame->mvbOutlier[i] = false;
            const cv::KeyPoint& kpUn = pFrame->mvTotalKeysUn[i];
            const cv::Mat Xw = pMP->GetWorldPos();
            for (int d = 0; d < 3; ++d) pp.xw.push_back((double)Xw.at<float>(d));                                                      // e->Xw[d] = Xw.at<float>(d) (:317-319)
            pp.obs.push_back((double)kpUn.pt.x); pp.obs.push_back((double)kpUn.pt.y);
            pp.invSigma2.push_back((double)pFrame->mvInvLevelSigma2[kpUn.octave]);
            pp.edgeCam.push_back(pFrame->keypointToCam[i]);
            vnIndexEdgeMono.push_back((size_t)i);
        }
    }
    pp.edgeOff.push_back((int32_t)pp.edgeCam.size());
    if (nInitialCorrespondences < 3) return 0;                                                                                         // :343-344
    for (int c = 0; c < pFrame->mnCams; ++c)                                                                                           // e->fx .. e->setExtrinsic (:311-328)
        pp.cams.push_back(dcs_adapters::rigCamera(Frame::mvfx[c], Frame::mvfy[c], Frame::mvcx[c], Frame::mvcy[c], pFrame->mvExtrinsics[c], pFrame->mvExtAdj[c]));
    std::vector<double> poseOut;
    std::vector<uint8_t> outlier;
    const std::vector<int> inliers = PoseOptimization(pp, poseOut, outlier);                                                           // the four rounds of :354-393 on the device
    for (size_t e = 0; e < vnIndexEdgeMono.size(); ++e) pFrame->mvbOutlier[vnIndexEdgeMono[e]] = outlier[e] != 0;
    pFrame->SetPose(dcs_adapters::poseMat(poseOut.data()));                                                                             // :396-399
    return inliers[0];                                                                                                                  // nInitialCorrespondences - nBad
}

// ------------------------------------------------------------------------------------------------------------------ ORBmatcher
// int SearchByProjection(FramePtr pF, const std::vector<MapPointPtr>& vpMapPoints, const float th)   (include/ORBmatcher.h:65-67, src/ORBmatcher.cc:539-624)
inline int ORBmatcher::SearchByProjection(FramePtr pF, const std::vector<MapPointPtr>& vpMapPoints, const float th)
{
    dcs_adapters::FrameArrays fr;
    fr.fill(pF);
    dcs_adapters::QueryArrays q;
    q.reserve(vpMapPoints.size());
    const bool bFactor = th != 1.0;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPointPtr pMP = vpMapPoints[iMP];
        if (!pMP || !pMP->mbTrackInView || pMP->isBad()) continue;                                                                     // :549-552
        const int nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = pMP->mTrackViewCos > 0.998 ? 2.5f : 4.0f;                                                                            // RadiusByViewingCos (:65-71)
        if (bFactor) r *= th;
        q.valid[iMP] = 1; q.cam[iMP] = pMP->mTrackProjCamera; q.u[iMP] = pMP->mTrackProjX; q.v[iMP] = pMP->mTrackProjY;
        q.radius[iMP] = r * pF->mvScaleFactors[nPredictedLevel];                                                                       // GetFeaturesInArea(.., r * mvScaleFactors[level], level - 1, level + 1) (:563-567)
        q.minLevel[iMP] = nPredictedLevel - 1; q.maxLevel[iMP] = nPredictedLevel + 1;
        q.setDescriptor(iMP, pMP->GetDescriptor());
    }
    q.finish();
    std::vector<int32_t> matchOfQuery, queryOfFeature;
    const int nmatches = SearchByProjection(fr.view, q.view, matchOfQuery, queryOfFeature);                                            // the candidate loop of :575-620, order dependence included
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++)
        if (matchOfQuery[iMP] >= 0) pF->mvpMapPoints[matchOfQuery[iMP]] = vpMapPoints[iMP];                                           // :619
    return nmatches;
}

// int SearchByProjectionOnCam(FramePtr pFcur, const int& query, FramePtr pFlast, const float th)   (include/ORBmatcher.h:121-124, src/ORBmatcher.cc:954-1113)
inline int ORBmatcher::SearchByProjectionOnCam(FramePtr pFcurt, const int& query, FramePtr pFlast, const float th)
{
    const cv::Mat Tsc = pFcurt->mvExtrinsics[query];                                                                                   // :962-968 as they are
    const cv::Mat Tcw = pFcurt->mTcw;
    const cv::Mat Tsw = Tsc * Tcw;
    const cv::Mat Rsw = Tsw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tsw = Tsw.rowRange(0, 3).col(3);
    dcs_adapters::FrameArrays fr;
    fr.fill(pFcurt);
    const std::vector<MapPointPtr> vplastMPs = pFlast->mvpMapPoints;
    dcs_adapters::QueryArrays q;
    q.reserve(vplastMPs.size());
    for (size_t i = 0, iend = vplastMPs.size(); i < iend; i++) {                                                                       // :988-1036 without the debug drawing
        if (pFlast->keypointToCam[i] != query) continue;
        MapPointPtr pMP = vplastMPs[i];
        if (!pMP || pMP->isBad()) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Ds = Rsw * x3Dw + tsw;
        const float xs = x3Ds.at<float>(0), ys = x3Ds.at<float>(1), zs = x3Ds.at<float>(2);
        if (zs < 0) continue;
        const float invzs = 1.0 / x3Ds.at<float>(2);
        const float u = Frame::mvfx[query] * xs * invzs + Frame::mvcx[query];
        const float v = Frame::mvfy[query] * ys * invzs + Frame::mvcy[query];
        if (u < Frame::mvMinX[query] || u > Frame::mvMaxX[query]) continue;
        if (v < Frame::mvMinY[query] || v > Frame::mvMaxY[query]) continue;
        const int nLastOctave = pFlast->mvTotalKeysUn[i].octave;
        q.valid[i] = 1; q.cam[i] = query; q.u[i] = u; q.v[i] = v;
        q.radius[i] = th * pFcurt->mvScaleFactors[nLastOctave];
        q.minLevel[i] = nLastOctave - 1; q.maxLevel[i] = nLastOctave + 1;
        q.angle[i] = pFlast->mvTotalKeysUn[i].angle;
        q.setDescriptor(i, pMP->GetDescriptor());
    }
    q.finish();
    std::vector<int32_t> matchOfQuery, queryOfFeature;
    const int nmatches = SearchByProjectionOnCam(fr.view, q.view, matchOfQuery, queryOfFeature);                                        // :1038-1098: best candidate, TH_HIGH, rotation histogram
    for (size_t i = 0; i < vplastMPs.size(); i++)
        if (matchOfQuery[i] >= 0) pFcurt->mvpMapPoints[matchOfQuery[i]] = vplastMPs[i];                                                // :1062 (the histogram's removals are already applied)
    return nmatches;
}

// int SearchByProjection(FramePtr pCurrentFrame, const FramePtr pLastFrame, const float th, bool bMapScaled)   (include/ORBmatcher.h:79-82, src/ORBmatcher.cc:634-690)
inline int ORBmatcher::SearchByProjection(FramePtr pCurrentFrame, const FramePtr pLastFrame, const float th, bool bMapScaled)
{
    int nmatches = 0;
    const int nCams = pCurrentFrame->mnCams;
    for (int ic = 0; ic < nCams; ic++) {
        if (ic != 0 && !bMapScaled) continue;
        const int nmatch = SearchByProjectionOnCam(pCurrentFrame, ic, pLastFrame, th);
        if (nmatch <= 20) { nmatches = nmatch; break; }
        nmatches += nmatch;
    }
    return nmatches;
}

// int SearchByBoWCrossCam(FramePtr F, const int& cF, KeyFramePtr pKF, const int& cKF, std::vector<MapPointPtr>& vpMapPointMatches)   (include/ORBmatcher.h:196-200, src/ORBmatcher.cc:162-294)
inline int ORBmatcher::SearchByBoWCrossCam(FramePtr pF, const int& cF, KeyFramePtr pKF, const int& cKF, std::vector<MapPointPtr>& vpMapPointMatches)
{
    const std::vector<MapPointPtr> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPointPtr>(pF->mvN[cF], static_cast<MapPointPtr>(NULL));
    const int nF = pF->mvN[cF], nKF = pKF->mvN[cKF];
    std::vector<uint8_t> descF(pF->mvDescriptors[cF].data, pF->mvDescriptors[cF].data + (size_t)nF * 32);
    std::vector<uint8_t> descKF(pKF->mvDescriptors[cKF].data, pKF->mvDescriptors[cKF].data + (size_t)nKF * 32);
    std::vector<float> angF(nF), angKF(nKF);
    std::vector<uint8_t> kfValid(nKF);
    for (int i = 0; i < nF; ++i) angF[i] = pF->mvvkeysUnTemp[cF][i].angle;
    for (int i = 0; i < nKF; ++i) {
        angKF[i] = pKF->mvvkeysUnTemp[cKF][i].angle;
        const MapPointPtr& pMP = vpMapPointsKF[pKF->GetGlobalIdxByLocal(i, cKF)];                                                      // :197-202
        kfValid[i] = (pMP && !pMP->isBad()) ? 1 : 0;
    }
    std::vector<int32_t> matchF;
    const int nmatches = SearchByBoWCrossCam(descF, angF, dcs_adapters::flatten(pF->mvFeatVec[cF]), descKF, angKF, kfValid, dcs_adapters::flatten(pKF->mvFeatVec[cKF]), matchF);
    for (int j = 0; j < nF; ++j)
        if (matchF[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[pKF->GetGlobalIdxByLocal(matchF[j], cKF)];                            // :236
    return nmatches;
}

}  // namespace ORB_SLAM2

#endif  // DCS_WITH_REFERENCE_MODEL
