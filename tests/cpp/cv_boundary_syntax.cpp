// cv_boundary_syntax.cpp -- parsed with `g++ -fsyntax-only -DDCS_WITH_OPENCV` against tests/cpp/cv_syntax_stub.h (a syntax stand-in that
// pins nothing): the reference's own call of the extraction seam, Frame::ExtractORB (src/Frame.cc:210-213)
//     (*(mvpORBextractor[c]))(mvImages[c], cv::Mat(), vKeys, Descriptor);
// with the reference's argument list (include/ORBextractor.h:59-61: cv::InputArray image, cv::InputArray mask,
// std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors) must resolve to the mirror's DCS_WITH_OPENCV overload, and
// cv::KeyPoint must be the 28-byte record the C ABI writes.
#include <cstddef>
#include <memory>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"
#include "ORBVocabulary.h"
#include "KeyFrameDatabase.h"

static_assert(sizeof(cv::KeyPoint) == sizeof(dcs_keypoint) && sizeof(dcs_keypoint) == 28, "cv::KeyPoint is the 28-byte record of the C ABI");
static_assert(offsetof(cv::KeyPoint, pt) == offsetof(dcs_keypoint, x) && offsetof(cv::KeyPoint, pt) + offsetof(cv::Point2f, y) == offsetof(dcs_keypoint, y), "pt");
static_assert(offsetof(cv::KeyPoint, size) == offsetof(dcs_keypoint, size), "size");
static_assert(offsetof(cv::KeyPoint, angle) == offsetof(dcs_keypoint, angle), "angle");
static_assert(offsetof(cv::KeyPoint, response) == offsetof(dcs_keypoint, response), "response");
static_assert(offsetof(cv::KeyPoint, octave) == offsetof(dcs_keypoint, octave), "octave");
static_assert(offsetof(cv::KeyPoint, class_id) == offsetof(dcs_keypoint, class_id), "class_id");

namespace {

// the members of the reference's Frame that ExtractORB touches (include/Frame.h), with the mirror in place of ORB_SLAM2::ORBextractor
struct FrameLike {
    std::vector<std::shared_ptr<ORB_SLAM2::ORBextractor>> mvpORBextractor;
    std::vector<cv::Mat> mvImages;

    void ExtractORB(const int& c, std::vector<cv::KeyPoint>& vKeys, cv::Mat& Descriptor)
    {
        (*(mvpORBextractor[c]))(mvImages[c], cv::Mat(), vKeys, Descriptor);
    }
};

// the overload that call resolves to has exactly the reference's parameter list
using RefCallOperator = void (ORB_SLAM2::ORBextractor::*)(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray);
constexpr RefCallOperator kRefCall = &ORB_SLAM2::ORBextractor::operator();

}  // namespace

int cv_boundary_syntax_anchor()
{
    FrameLike f;
    std::vector<cv::KeyPoint> keys;
    cv::Mat desc;
    f.mvpORBextractor.push_back(std::make_shared<ORB_SLAM2::ORBextractor>(1000, 1.2f, 8, 20, 7));      // the reference's five constructor arguments (Tracking.cc:204-207)
    f.mvImages.resize(1);
    f.ExtractORB(0, keys, desc);
    return (int)keys.size() + (kRefCall != nullptr);
}
