// ORBextractor.h -- header-only C++ mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-113)
// over the C ABI (include/dcs_abi.h). Same constructor arguments, getters and call operator; the
// reference's Frame::ExtractORB (src/Frame.cc:210-213) compiles against it unchanged where OpenCV exists
// (define DCS_WITH_OPENCV). Without OpenCV the call operator takes a plain image view.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "dcs_abi.h"
#ifdef DCS_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace ORB_SLAM2 {

struct ImageView { const uint8_t* data; int rows, cols, stride; };   // 8-bit single channel

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxImages = 2)
        : nfeatures_(nfeatures), scaleFactor_(scaleFactor), nlevels_(nlevels)
    {
        dcs_orb_params p{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, -1, maxImages, 0};
        if (dcs_orb_create(&p, &h_) != DCS_OK) throw std::runtime_error(std::string("dcs_orb_create: ") + dcs_last_error());
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        dcs_orb_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), nullptr);
    }
    ~ORBextractor() { dcs_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // keypoints: cv::KeyPoint-compatible 28-byte records; descriptors: N x 32 bytes, row-contiguous.
    // Empty image -> returns with empty outputs (reference ORBextractor.cc:1046-1047).
    void operator()(const ImageView& image, std::vector<dcs_keypoint>& keypoints, std::vector<uint8_t>& descriptors)
    {
        const int cap = capacity(image.rows, image.cols);
        keypoints.resize(cap); descriptors.resize((size_t)cap * 32);
        int n = 0;
        const int rc = dcs_orb_extract(h_, image.data, image.rows, image.cols, image.stride, keypoints.data(), descriptors.data(), cap, &n);
        if (rc != DCS_OK) throw std::runtime_error(std::string("dcs_orb_extract: ") + dcs_last_error());
        keypoints.resize(n); descriptors.resize((size_t)n * 32);
    }

#ifdef DCS_WITH_OPENCV
    static_assert(sizeof(cv::KeyPoint) == sizeof(dcs_keypoint), "cv::KeyPoint layout");
    // exact reference signature: mask is ignored (ORBextractor.h:57-61)
    void operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        const int cap = capacity(image.rows, image.cols);
        _keypoints.resize(cap);
        cv::Mat desc(cap, 32, CV_8U);
        int n = 0;
        const int rc = dcs_orb_extract(h_, image.data, image.rows, image.cols, (int)image.step, reinterpret_cast<dcs_keypoint*>(_keypoints.data()),
                                       desc.data, cap, &n);
        if (rc != DCS_OK) throw std::runtime_error(std::string("dcs_orb_extract: ") + dcs_last_error());
        _keypoints.resize(n);
        if (n == 0) _descriptors.release(); else desc.rowRange(0, n).copyTo(_descriptors);
    }
#endif

    // output slots one image may need: the quadtree keeps up to max(N_level + 3, 4 * round(w/h)) keypoints per level
    int capacity(int rows, int cols) const
    {
        int cap = nfeatures_ + 4 * nlevels_ + 64;
        int need = 0;
        if (rows > 0 && cols > 0 && dcs_orb_required_cap(h_, rows, cols, &need) == DCS_OK && need > cap) cap = need;
        return cap;
    }

    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return (float)scaleFactor_; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
    dcs_orb* handle() { return h_; }

protected:
    dcs_orb* h_ = nullptr;
    int nfeatures_;
    double scaleFactor_;
    int nlevels_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2
