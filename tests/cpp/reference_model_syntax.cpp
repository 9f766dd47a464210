// reference_model_syntax.cpp -- parsed with `g++ -fsyntax-only -Werror -DDCS_WITH_OPENCV -DDCS_WITH_REFERENCE_MODEL` against
// tests/cpp/slam_model_stub.h + cv_syntax_stub.h (syntax stand-ins that pin nothing): the reference's OWN call lines on the matcher and
// optimiser seams must resolve to the reference-typed members of the mirrors (host/ReferenceAdapters.h), with the reference's parameter lists.
#include <memory>
#include <set>
#include <vector>

#include "slam_model_stub.h"              // where the reference is: Frame.h, KeyFrame.h, MapPoint.h, Map.h, Cameras.h

#include "ORBextractor.h"
#include "ReferenceAdapters.h"            // ORBmatcher.h + Optimizer.h + the reference-typed members

namespace {

using namespace ORB_SLAM2;

// the members of LocalMapping / Tracking these lines touch (include/LocalMapping.h, include/Tracking.h)
struct LocalMappingLike {
    KeyFramePtr mpCurrentKeyFrame;
    bool mbAbortBA;
    MapPtr mpMap;
    void Run(size_t fixId)
    {
        Optimizer::LocalBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpMap, fixId);                          // src/LocalMapping.cc:103
    }
};

struct TrackingLike {
    FramePtr mpCurrentFrame, mpLastFrame;
    KeyFramePtr mpReferenceKF;
    std::vector<MapPointPtr> mvpLocalMapPoints;
    MapPtr mpMap;
    bool mbIsMapScaled;

    int Relocalization(KeyFramePtr pKF, int camS, int CAP, std::vector<std::vector<MapPointPtr> >& vvpMapPointMatches, int i)
    {
        ORBmatcher matcher(0.75, true);                                                                             // src/Tracking.cc:797
        int nmatches = matcher.SearchByBoWCrossCam(mpCurrentFrame, camS, pKF, CAP, vvpMapPointMatches[i]);           // src/Tracking.cc:822
        return nmatches;
    }
    int TrackReferenceKeyFrame()
    {
        mpCurrentFrame->SetPose(mpReferenceKF->GetPose());                                                           // src/Tracking.cc:1319
        return Optimizer::PoseOptimization(mpCurrentFrame);                                                          // src/Tracking.cc:1321
    }
    int TrackWithMotionModel(int th)
    {
        ORBmatcher matcher(0.8, true);                                                                               // src/Tracking.cc:1387
        int nmatches = matcher.SearchByProjection(mpCurrentFrame, mpLastFrame, th, mbIsMapScaled);                   // src/Tracking.cc:1406
        if (nmatches < 20) nmatches = matcher.SearchByProjection(mpCurrentFrame, mpLastFrame, 2 * th, mbIsMapScaled);   // :1415
        Optimizer::PoseOptimization(mpCurrentFrame);                                                                 // :1427
        return nmatches;
    }
    int SearchLocalPoints(int th)
    {
        ORBmatcher matcher(0.8);                                                                                     // src/Tracking.cc:1672
        return matcher.SearchByProjection(mpCurrentFrame, mvpLocalMapPoints, th);                                    // src/Tracking.cc:1680
    }
    void GlobalBA(unsigned long nLoopKF, bool* pbStop)
    {
        Optimizer::GlobalBundleAdjustemnt(mpMap, 10, 0, pbStop, nLoopKF, false);                                     // src/LoopClosing.cc (RunGlobalBundleAdjustment)
    }
};

// the overloads those calls resolve to have exactly the reference's parameter lists (include/Optimizer.h:49-56, include/ORBmatcher.h:65-67, 79-82, 121-124, 196-200)
using RefLocalBA = void (*)(KeyFramePtr, bool*, MapPtr, size_t);
using RefBA = void (*)(const std::vector<KeyFramePtr>&, const std::vector<MapPointPtr>&, unsigned long, int, bool*, const unsigned long, const bool);
using RefGBA = void (*)(MapPtr, int, unsigned long, bool*, const unsigned long, const bool);
using RefPoseOpt = int (*)(FramePtr);
using RefSearchLocal = int (ORBmatcher::*)(FramePtr, const std::vector<MapPointPtr>&, const float);
using RefSearchLast = int (ORBmatcher::*)(FramePtr, const FramePtr, const float, bool);
using RefSearchOnCam = int (ORBmatcher::*)(FramePtr, const int&, FramePtr, const float);
using RefSearchBoW = int (ORBmatcher::*)(FramePtr, const int&, KeyFramePtr, const int&, std::vector<MapPointPtr>&);
constexpr RefLocalBA kLocalBA = &Optimizer::LocalBundleAdjustment;
constexpr RefBA kBA = &Optimizer::BundleAdjustment;
constexpr RefGBA kGBA = &Optimizer::GlobalBundleAdjustemnt;
constexpr RefPoseOpt kPoseOpt = &Optimizer::PoseOptimization;
constexpr RefSearchLocal kSearchLocal = &ORBmatcher::SearchByProjection;
constexpr RefSearchLast kSearchLast = &ORBmatcher::SearchByProjection;
constexpr RefSearchOnCam kSearchOnCam = &ORBmatcher::SearchByProjectionOnCam;
constexpr RefSearchBoW kSearchBoW = &ORBmatcher::SearchByBoWCrossCam;

}  // namespace

int reference_model_syntax_anchor()
{
    LocalMappingLike lm{};
    TrackingLike tr{};
    std::vector<std::vector<MapPointPtr> > vv(1);
    lm.Run(0);
    return tr.Relocalization(KeyFramePtr(), 0, 1, vv, 0) + tr.TrackReferenceKeyFrame() + tr.TrackWithMotionModel(7) + tr.SearchLocalPoints(1) +
           (kLocalBA != nullptr) + (kBA != nullptr) + (kGBA != nullptr) + (kPoseOpt != nullptr) + (kSearchLocal != nullptr) + (kSearchLast != nullptr) +
           (kSearchOnCam != nullptr) + (kSearchBoW != nullptr);
}
