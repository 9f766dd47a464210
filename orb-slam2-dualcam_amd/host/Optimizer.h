// Optimizer.h -- header-only C++ mirror of Optimizer::LocalBundleAdjustment (reference include/Optimizer.h:55,
// src/Optimizer.cc:407-696) over the C ABI. The reference gathers KeyFrames / MapPoints into a g2o graph
// (:409-580), optimises (:586-621) and writes back (:641-693); the gather and write-back stay with the caller
// (they walk the SLAM data model under its mutexes), the numerics run on the GPU from a flat problem.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "dcs_abi.h"

namespace ORB_SLAM2 {

struct LocalBAProblem {
    std::vector<double> poses;        // [P][7] from dcs_pose_from_matrix(KF->GetPose()), ascending mnId
    std::vector<uint8_t> poseFixed;   // mnId == fixId, or member of lFixedCameras (:483, :496)
    std::vector<double> points;       // [L][3] MapPoint::GetWorldPos
    std::vector<int32_t> edgePose, edgePoint, edgeCam;   // keypointToCam[idx] (:541)
    std::vector<double> obs;          // [E][2] mvTotalKeysUn[idx].pt
    std::vector<double> invSigma2;    // [E] mvInvLevelSigma2[octave]
    std::vector<dcs_ba_camera> cams;  // per camera: mvfx.., dcs_rig_adjoint(getExtrinsici(c))
};

struct LocalBAResult {
    std::vector<double> poses, points;
    std::vector<uint8_t> edgeOutlier;  // -> vToErase (:653-657)
    int iterations[2];
};

// flat input of PoseOptimization for a batch of frames (one per stream); see dcs_pose_problem
struct PoseProblem {
    std::vector<double> poses;        // [F][7] dcs_pose_from_matrix(pFrame->mTcw)
    std::vector<int32_t> edgeOff;     // [F+1]
    std::vector<double> xw, obs, invSigma2;   // [E][3] MapPoint::GetWorldPos, [E][2] mvTotalKeysUn[i].pt, [E]
    std::vector<int32_t> edgeCam;     // [E] keypointToCam[i]
    std::vector<dcs_ba_camera> cams;
};

class Optimizer {
public:
    // Optimizer::PoseOptimization (src/Optimizer.cc:250-405), batched over frames. outlier[e] = pFrame->mvbOutlier of the
    // edge's feature; returns per frame nInitialCorrespondences - nBad (0 and an untouched pose below 3 correspondences).
    static std::vector<int> PoseOptimization(const PoseProblem& in, std::vector<double>& posesOut, std::vector<uint8_t>& outlier)
    {
        dcs_pose_problem p{};
        p.n_frames = (int)in.poses.size() / 7; p.n_cams = (int)in.cams.size();
        p.poses = in.poses.data(); p.edge_off = in.edgeOff.data(); p.xw = in.xw.data(); p.obs = in.obs.data();
        p.inv_sigma2 = in.invSigma2.data(); p.edge_cam = in.edgeCam.data(); p.cams = in.cams.data();
        p.huber_delta = (double)(float)2.447651936;                       // const float deltaMono = sqrt(5.991) (:284)
        for (int i = 0; i < 4; ++i) { p.chi2_th[i] = 5.991f; p.its[i] = 10; }   // :352, :354
        posesOut.resize(in.poses.size()); outlier.resize(in.edgeCam.size() ? in.edgeCam.size() : 1);
        std::vector<int> inliers(p.n_frames);
        dcs_pose_result r{};
        r.poses = posesOut.data(); r.outlier = outlier.data(); r.n_inliers = inliers.data();
        const int rc = dcs_pose_optimization(&p, &r);
        if (rc != DCS_OK) throw std::runtime_error(std::string("dcs_pose_optimization: ") + dcs_last_error());
        outlier.resize(in.edgeCam.size());
        return inliers;
    }

#ifdef DCS_WITH_REFERENCE_MODEL
    // The reference's own signatures (include/Optimizer.h:49-56), bodies in ReferenceAdapters.h: the reference's gather, one call into the library,
    // the reference's write-back. Needs Frame.h / KeyFrame.h / MapPoint.h / Map.h / Cameras.h included first.
    void static BundleAdjustment(const std::vector<KeyFramePtr>& vpKF, const std::vector<MapPointPtr>& vpMP, unsigned long fixId, int nIterations = 5,
                                 bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    void static GlobalBundleAdjustemnt(MapPtr pMap, int nIterations = 5, unsigned long fixId = 0, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                       const bool bRobust = true);
    void static LocalBundleAdjustment(KeyFramePtr pKF, bool* pbStopFlag, MapPtr pMap, size_t fixId);
    int static PoseOptimization(FramePtr pFrame);
#endif

    // pbStopFlag as in the reference (:471-472, :582-593); thHuber = sqrt(5.991), chi2 gate 5.991, 5 + 10 iterations
    static void LocalBundleAdjustment(const LocalBAProblem& in, bool* pbStopFlag, LocalBAResult& out)
    {
        solve(in, pbStopFlag, out, (double)(float)2.447651936 /* const float thHuberMono = sqrt(5.991) (:515) */, 5, 10);
    }

    // Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (src/Optimizer.cc:61-248): same dual-camera edges and solver,
    // ONE round of nIterations, Huber delta = (float)sqrt(3.99) (:107) when bRobust, no outlier re-classification.
    // `in.poseFixed` marks only fixId; map points without observations are left out by the caller (:184-192).
    static void BundleAdjustment(const LocalBAProblem& in, int nIterations, bool* pbStopFlag, bool bRobust, LocalBAResult& out)
    {
        solve(in, pbStopFlag, out, bRobust ? (double)(float)1.9974984 : 0.0, nIterations, 0);
    }

private:
    static void solve(const LocalBAProblem& in, bool* pbStopFlag, LocalBAResult& out, double huber, int iters1, int iters2)
    {
        dcs_ba_problem p{};
        p.n_poses = (int)in.poseFixed.size(); p.n_points = (int)in.points.size() / 3; p.n_edges = (int)in.edgePose.size();
        p.n_cams = (int)in.cams.size();
        p.poses = in.poses.data(); p.pose_fixed = in.poseFixed.data(); p.points = in.points.data();
        p.edge_pose = in.edgePose.data(); p.edge_point = in.edgePoint.data(); p.edge_cam = in.edgeCam.data();
        p.obs = in.obs.data(); p.inv_sigma2 = in.invSigma2.data(); p.cams = in.cams.data();
        p.huber_delta = huber;
        p.chi2_th = 5.991; p.iters1 = iters1; p.iters2 = iters2;
        out.poses.resize(in.poses.size()); out.points.resize(in.points.size()); out.edgeOutlier.resize(p.n_edges);
        dcs_ba_result r{};
        r.poses = out.poses.data(); r.points = out.points.data(); r.edge_outlier = out.edgeOutlier.data();
        static_assert(sizeof(bool) == 1, "stop flag is polled as a byte");
        const int rc = dcs_ba_local(&p, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), &r);
        if (rc != DCS_OK) throw std::runtime_error(std::string("dcs_ba_local: ") + dcs_last_error());
        out.iterations[0] = r.n_iters[0]; out.iterations[1] = r.n_iters[1];
    }
};

}  // namespace ORB_SLAM2
