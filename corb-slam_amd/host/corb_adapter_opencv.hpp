// corb_adapter_opencv.hpp -- restores the reference's EXACT extractor signature on top of corb_host.hpp:
//   ORB_SLAM2::ORBextractor::operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray)
// (the ORBmatcher / Optimizer signatures -- KeyFrame*, Frame&, vector<MapPoint*>&, Cache*, bool* pbStopFlag, nLoopKF -- live in
// corb_adapter_orbslam.hpp as templates that ARE compiled and run here, against test doubles) so that corbslam_client/src/{Frame,Tracking,LocalMapping,LoopClosing}.cc and corbslam_server/src/*.cpp
// compile unchanged against it.  It needs OpenCV headers, which do not exist in the
// build container or on the GPU box: this file is compiled out unless <opencv2/core/core.hpp> is found and
// is therefore UNTESTED here (stated in DESIGN.md).  All arithmetic lives below the C-ABI either way.
#pragma once
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "corb_host.hpp"
#include <memory>

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : nf_(nfeatures), sf_(scaleFactor), nl_(nlevels), ini_(iniThFAST), min_(minThFAST) {}
    ~ORBextractor() {}
    // ORBextractor.cc:1043-1105.  static_assert: cv::KeyPoint and CorbKeyPoint are the same 28-byte POD.
    void operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors)
    {
        static_assert(sizeof(cv::KeyPoint) == sizeof(CorbKeyPoint), "cv::KeyPoint layout");
        if (image.empty()) return;
        cv::Mat im = image.getMat();
        CV_Assert(im.type() == CV_8UC1);
        if (!impl_ || w_ != im.cols || h_ != im.rows) { impl_.reset(new corb::ORBextractor(nf_, sf_, nl_, ini_, min_, im.cols, im.rows)); w_ = im.cols; h_ = im.rows; }
        std::vector<corb::KeyPoint> k; corb::Descriptors d;
        (*impl_)(im.data, im.cols, im.rows, (int)im.step, k, d);
        keypoints.resize(k.size());
        if (!k.empty()) std::memcpy((void*)keypoints.data(), k.data(), k.size() * sizeof(CorbKeyPoint));
        if (k.empty()) { descriptors.release(); return; }
        descriptors.create((int)k.size(), 32, CV_8U);
        std::memcpy(descriptors.getMat().data, d.data.data(), d.data.size());
        // mvImagePyramid (public in the reference, ORBextractor.h:85) is read by the UNMODIFIED Frame::ComputeStereoMatches (Frame.cc:477, 556-600: .rows of level 0
        // and the SAD refinement's patches), so a header-swap-only integration needs it filled after every call: that is the default.  The copy
        // (~1.4 MB per KITTI image) is pure overhead once the stereo matching runs on the device too (corb_stereo_frames): such callers set fillImagePyramid = false
        // and call ImagePyramid() where levels are still wanted (fetched once per image).
        pyramid_fresh_ = false;
        if (fillImagePyramid) (void)ImagePyramid(); else mvImagePyramid.clear();
    }
    // the pyramid of the LAST image passed to operator(), downloaded on first use
    const std::vector<cv::Mat>& ImagePyramid()
    {
        if (!pyramid_fresh_ && impl_) {
            mvImagePyramid.resize(nl_);
            for (int l = 0; l < nl_; l++) { int w, h; std::vector<uint8_t> px = impl_->ImagePyramidLevel(l, &w, &h); mvImagePyramid[l] = cv::Mat(h, w, CV_8UC1, px.data()).clone(); }
            pyramid_fresh_ = true;
        }
        return mvImagePyramid;
    }
    bool fillImagePyramid = true;              // the drop-in default; false = the fast path (see operator())
    int inline GetLevels() { return nl_; }
    float inline GetScaleFactor() { return sf_; }
    std::vector<float> inline GetScaleFactors() { return need().GetScaleFactors(); }
    std::vector<float> inline GetInverseScaleFactors() { return need().GetInverseScaleFactors(); }
    std::vector<float> inline GetScaleSigmaSquares() { return need().GetScaleSigmaSquares(); }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return need().GetInverseScaleSigmaSquares(); }
    std::vector<cv::Mat> mvImagePyramid;             // public in the reference (ORBextractor.h:85), read by Frame::ComputeStereoMatches
private:
    corb::ORBextractor& need() { if (!impl_) { impl_.reset(new corb::ORBextractor(nf_, sf_, nl_, ini_, min_, 1241, 376)); w_ = 1241; h_ = 376; } return *impl_; }
    int nf_; float sf_; int nl_, ini_, min_; int w_ = 0, h_ = 0; bool pyramid_fresh_ = false;
    std::unique_ptr<corb::ORBextractor> impl_;
};

}  // namespace ORB_SLAM2
#endif
#endif
