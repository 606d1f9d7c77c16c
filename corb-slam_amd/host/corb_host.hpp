// corb_host.hpp -- C++ host-side mirror of the reference's operator interface for the hot path, over the
// C-ABI of libcorb_accel.so (include/corb_accel.h).  Header-only, no OpenCV / Eigen / ROS: containers are
// std::vector and the POD structs of the C-ABI.  Same class and method names, argument meaning and
// "error" behaviour as the reference (which has no error path: an empty image yields no keypoints; a
// failure inside the accelerator throws corb::Error instead of silently falling back to the CPU).
//
//   ORB_SLAM2::ORBextractor  (corbslam_client/include/ORBextractor.h:45-114)   -> corb::ORBextractor
//   ORB_SLAM2::ORBmatcher    (corbslam_client/include/ORBmatcher.h:41-107)     -> corb::ORBmatcher
//   ORB_SLAM2::Optimizer     (corbslam_client/include/Optimizer.h:42-46)       -> corb::Optimizer
//   Frame::ComputeStereoMatches (corbslam_client/src/Frame.cc:470-644)          -> corb::StereoFrontend
//
// corb_adapter_opencv.hpp layers the exact cv::/KeyFrame signatures on top of this where OpenCV exists.
#pragma once
#include <corb_accel.h>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include <cmath>

namespace corb {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what + ": " + corb_last_error()), code(c) {}
};
inline void check(int rc, const char* what) { if (rc != CORB_OK) throw Error(rc, what); }
// The structs of corb_accel.h carry no size fields: the library must have been built from the header this translation unit was compiled against.
// Checked by Warmup() and by every class of this header on its first use (one comparison of a function-local static).
inline void check_abi()
{
    static const int v = corb_abi_version();
    if (v != CORB_ABI_VERSION) throw std::runtime_error("libcorb_accel.so: struct layout version " + std::to_string(v) + ", this program was compiled against " + std::to_string(CORB_ABI_VERSION));
}
// once per process and device at start-up: the per-device workspace lanes are created now, not inside the first bundle adjustment
inline void Warmup(int device = 0) { check_abi(); check(corb_warmup(device), "corb_warmup"); }

using KeyPoint = CorbKeyPoint;                       // bit-identical to cv::KeyPoint as the reference fills it
struct Descriptors { std::vector<uint8_t> data; int rows() const { return (int)(data.size() / 32); } const uint8_t* row(int i) const { return &data[(size_t)i * 32]; } };

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };       // ORBextractor.h:49 (unused by the reference as well)
    // ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) + the image size
    // the handle is built for (the reference sizes its pyramid lazily per call; the device arena is static)
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int width, int height, int device = 0)
        : nlevels_(nlevels), width_(width), height_(height)
    {
        check_abi();
        CorbOrbConfig cfg{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 1, device};
        check(corb_orb_create(&cfg, &h_), "corb_orb_create");
        cap_ = nfeatures + 16 * nlevels + 256;
    }
    ~ORBextractor() { corb_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>&, cv::OutputArray descriptors)
    // (mask is ignored by the reference, ORBextractor.cc:1043)
    void operator()(const uint8_t* image, int width, int height, int stride, std::vector<KeyPoint>& keypoints, Descriptors& descriptors)
    {
        keypoints.resize(cap_); descriptors.data.resize((size_t)cap_ * 32);
        int n = 0;
        check(corb_orb_extract(h_, image, width, height, stride, keypoints.data(), descriptors.data.data(), cap_, &n), "corb_orb_extract");
        keypoints.resize(n); descriptors.data.resize((size_t)n * 32);
    }
    int GetLevels() const { return nlevels_; }
    float GetScaleFactor() const { return table(0)[nlevels_ > 1 ? 1 : 0]; }
    std::vector<float> GetScaleFactors() const { return table(0); }
    std::vector<float> GetInverseScaleFactors() const { return table(1); }
    std::vector<float> GetScaleSigmaSquares() const { return table(2); }
    std::vector<float> GetInverseScaleSigmaSquares() const { return table(3); }
    // mvImagePyramid[level] (public member of the reference, read by Frame::ComputeStereoMatches)
    std::vector<uint8_t> ImagePyramidLevel(int level, int* w, int* h) const
    {
        check(corb_orb_pyramid_level(h_, 0, level, 0, nullptr, 0, w, h), "corb_orb_pyramid_level");
        std::vector<uint8_t> out((size_t)*w * *h);
        check(corb_orb_pyramid_level(h_, 0, level, 0, out.data(), out.size(), w, h), "corb_orb_pyramid_level");
        return out;
    }
    CorbOrb* handle() const { return h_; }
private:
    std::vector<float> table(int which) const
    {
        std::vector<float> t[4]; for (auto& v : t) v.resize(nlevels_);
        check(corb_orb_tables(h_, t[0].data(), t[1].data(), t[2].data(), t[3].data(), nullptr, nullptr), "corb_orb_tables");
        return t[which];
    }
    CorbOrb* h_ = nullptr; int nlevels_, width_, height_, cap_;
};

// Frame::Frame(stereo) hot path: left + right ORBextractor::operator() and Frame::ComputeStereoMatches
class StereoFrontend {
public:
    struct FrameResult { std::vector<KeyPoint> mvKeys, mvKeysRight; Descriptors mDescriptors, mDescriptorsRight; std::vector<float> mvuRight, mvDepth; };
    StereoFrontend(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int width, int height,
                   float fx, float bf, int max_frames = 1, int device = 0) : max_frames_(max_frames)
    {
        CorbStereoConfig cfg{{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 0, device}, max_frames, fx, bf};
        check(corb_stereo_create(&cfg, &h_), "corb_stereo_create");
        cap_ = nfeatures + 16 * nlevels + 256;
    }
    ~StereoFrontend() { corb_stereo_destroy(h_); }
    void Upload(int frame, const uint8_t* left, const uint8_t* right, int stride) { check(corb_stereo_upload(h_, frame, left, right, stride), "corb_stereo_upload"); }
    // whole batches with one copy each (pinned host memory = DMA): leftRight = per frame the left image then the right image, tightly packed;
    // results strided by Capacity(): keypoints / descriptors of image 2f (left) and 2f+1 (right), mvuRight / mvDepth per frame
    void UploadBatch(int firstFrame, int nFrames, const uint8_t* leftRight) { check(corb_stereo_upload_batch(h_, firstFrame, nFrames, leftRight), "corb_stereo_upload_batch"); }
    int Capacity() const { return corb_orb_capacity(corb_stereo_orb(h_)); }
    void FetchBatch(int firstFrame, int nFrames, KeyPoint* keys, uint8_t* descriptors, int32_t* counts, float* uRight, float* depth, int32_t* nMatched)
    {
        check(corb_orb_fetch_batch(corb_stereo_orb(h_), 2 * firstFrame, 2 * nFrames, keys, descriptors, counts), "corb_orb_fetch_batch");
        check(corb_stereo_fetch_matches_batch(h_, firstFrame, nFrames, uRight, depth, nMatched), "corb_stereo_fetch_matches_batch");
    }
    void Run(int n_frames) { check(corb_stereo_run(h_, n_frames), "corb_stereo_run"); }
    // Frame::Frame(stereo) (Frame.cc:61-117) as ONE call (corb_stereo_frames): nFrames x {left, right} images tightly packed in, nFrames result blocks out
    // (page-locked buffers from corb_pinned_alloc make both transfers DMA); one synchronisation.  FromBlock() copies a block into the containers of Fetch().
    CorbStereoFrameLayout FrameLayout() const { CorbStereoFrameLayout l; check(corb_stereo_frame_layout(h_, &l), "corb_stereo_frame_layout"); return l; }
    void Frames(int nFrames, const uint8_t* leftRight, void* resultBlocks, CorbStereoFrameTiming* timing = nullptr) { check(corb_stereo_frames(h_, nFrames, leftRight, resultBlocks, timing), "corb_stereo_frames"); }
    static FrameResult FromBlock(const CorbStereoFrameLayout& l, const void* block)
    {
        const uint8_t* b = static_cast<const uint8_t*>(block); const int32_t* hd = reinterpret_cast<const int32_t*>(b);
        FrameResult r; const size_t nl = (size_t)hd[0], nr = (size_t)hd[1];
        r.mvKeys.assign(reinterpret_cast<const KeyPoint*>(b + l.off_kp_left), reinterpret_cast<const KeyPoint*>(b + l.off_kp_left) + nl);
        r.mvKeysRight.assign(reinterpret_cast<const KeyPoint*>(b + l.off_kp_right), reinterpret_cast<const KeyPoint*>(b + l.off_kp_right) + nr);
        r.mDescriptors.data.assign(b + l.off_desc_left, b + l.off_desc_left + 32 * nl); r.mDescriptorsRight.data.assign(b + l.off_desc_right, b + l.off_desc_right + 32 * nr);
        r.mvuRight.assign(reinterpret_cast<const float*>(b + l.off_u_right), reinterpret_cast<const float*>(b + l.off_u_right) + nl);
        r.mvDepth.assign(reinterpret_cast<const float*>(b + l.off_depth), reinterpret_cast<const float*>(b + l.off_depth) + nl);
        return r;
    }
    CorbStereo* handle() const { return h_; }
    void Sync() { check(corb_stereo_sync(h_), "corb_stereo_sync"); }
    FrameResult Fetch(int frame)
    {
        FrameResult r; int n = 0, nm = 0;
        auto get = [&](int slot, std::vector<KeyPoint>& k, Descriptors& d) {
            k.resize(cap_); d.data.resize((size_t)cap_ * 32);
            check(corb_orb_fetch(corb_stereo_orb(h_), slot, k.data(), d.data.data(), cap_, &n), "corb_orb_fetch");
            k.resize(n); d.data.resize((size_t)n * 32);
        };
        get(2 * frame, r.mvKeys, r.mDescriptors); get(2 * frame + 1, r.mvKeysRight, r.mDescriptorsRight);
        r.mvuRight.resize(cap_); r.mvDepth.resize(cap_);
        check(corb_stereo_fetch_matches(h_, frame, r.mvuRight.data(), r.mvDepth.data(), cap_, &n, &nm), "corb_stereo_fetch_matches");
        r.mvuRight.resize(n); r.mvDepth.resize(n);
        return r;
    }
private:
    CorbStereo* h_ = nullptr; int max_frames_, cap_;
};

// DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned>>, flattened in ascending node order
struct FeatureVector {
    std::vector<uint32_t> node_id, idx; std::vector<int32_t> offset{0};
    void add(uint32_t node, const std::vector<unsigned>& features) { node_id.push_back(node); idx.insert(idx.end(), features.begin(), features.end()); offset.push_back((int32_t)idx.size()); }
    CorbFeatVec c() const { return CorbFeatVec{(int32_t)node_id.size(), node_id.data(), offset.data(), idx.data()}; }
};

// what the matchers read from a KeyFrame / Frame
struct FeatureSet {
    Descriptors desc; std::vector<KeyPoint> keysUn; std::vector<float> uRight;
    std::vector<uint8_t> hasGoodMapPoint;           // vpMapPoints[i] && !isBad()
    FeatureVector featVec;
    std::vector<float> angles() const { std::vector<float> a(keysUn.size()); for (size_t i = 0; i < a.size(); i++) a[i] = keysUn[i].angle; return a; }
};

class ORBmatcher {
public:
    static constexpr int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;       // ORBmatcher.cc:37-39
    ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b, int device = 0) { int32_t d = 0; check(corb_descriptor_distance(a, b, 1, &d, device), "corb_descriptor_distance"); return d; }
    // int SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches): matches[iF] = KF feature index or -1
    int SearchByBoW(const FeatureSet& kf, const FeatureSet& frame, std::vector<int32_t>& matches) const { return bow(0, kf, frame, matches); }
    int SearchByBoWInServer(const FeatureSet& kf, const FeatureSet& f, std::vector<int32_t>& matches) const { return bow(0, kf, f, matches); }
    // int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12): matches12[i1] = idx2 or -1
    int SearchByBoW_KF(const FeatureSet& kf1, const FeatureSet& kf2, std::vector<int32_t>& matches12) const { return bow(1, kf1, kf2, matches12); }
    // int SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, vector<pair<size_t,size_t>>&, bool bOnlyStereo)
    int SearchForTriangulation(const FeatureSet& kf1, const FeatureSet& kf2, const float F12[9], float ex, float ey,
                               const std::vector<float>& scaleFactors2, const std::vector<float>& levelSigma2_2,
                               std::vector<std::pair<size_t, size_t>>& vMatchedPairs, bool bOnlyStereo) const
    {
        CorbTriSide a{kf1.desc.data.data(), kf1.keysUn.data(), kf1.uRight.data(), kf1.hasGoodMapPoint.data(), (int32_t)kf1.keysUn.size(), kf1.featVec.c()};
        CorbTriSide b{kf2.desc.data.data(), kf2.keysUn.data(), kf2.uRight.data(), kf2.hasGoodMapPoint.data(), (int32_t)kf2.keysUn.size(), kf2.featVec.c()};
        std::vector<int32_t> pairs(2 * kf1.keysUn.size() + 2); int n = 0;
        check(corb_search_for_triangulation(&a, &b, F12, ex, ey, scaleFactors2.data(), levelSigma2_2.data(), (int)scaleFactors2.size(),
                                            bOnlyStereo, mbCheckOrientation, pairs.data(), &n, device_), "corb_search_for_triangulation");
        vMatchedPairs.clear();
        for (int i = 0; i < n; i++) vMatchedPairs.emplace_back((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]);
        return n;
    }
    // int SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, float th) (ORBmatcher.cc:45-131): `tracked` holds the
    // points with mbTrackInView (their cached isInFrustum() outputs); matches[iF] = index into `tracked` or -1
    int SearchByProjection(const CorbFrameView& F, const std::vector<CorbTrackedPoint>& tracked, const uint8_t* pointDescriptors, float th,
                           std::vector<int32_t>& matches) const
    {
        matches.assign(F.n > 0 ? F.n : 1, -1); int n = 0;
        check(corb_search_by_projection_map(&F, tracked.data(), pointDescriptors, (int)tracked.size(), th, mfNNratio, matches.data(), &n, device_), "corb_search_by_projection_map");
        matches.resize(F.n);
        return n;
    }
    // int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th, bool bMono) (ORBmatcher.cc:1470-1614)
    int SearchByProjection(const CorbFrameView& Cur, const float Tcw[16], const float Tlw[16], float fx, float fy, float cx, float cy, float mbf, float mb,
                           const std::vector<CorbLastPoint>& last, const uint8_t* lastDescriptors, float th, bool bMono, std::vector<int32_t>& matches) const
    {
        matches.assign(Cur.n > 0 ? Cur.n : 1, -1); int n = 0;
        check(corb_search_by_projection_frame(&Cur, Tcw, Tlw, fx, fy, cx, cy, mbf, mb, last.data(), lastDescriptors, (int)last.size(), th, bMono ? 1 : 0,
                                              mbCheckOrientation ? 1 : 0, matches.data(), &n, device_), "corb_search_by_projection_frame");
        matches.resize(Cur.n);
        return n;
    }

    // int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1616-1744):
    // points[i] = the MapPoint of keyframe feature i (valid = non-NULL && !isBad() && !sAlreadyFound.count()); matches[iF] = i or -1
    int SearchByProjection(const CorbKeyFrameView& CurrentFrame, const std::vector<uint8_t>& hasMapPoint, const float Tcw[16],
                           const std::vector<CorbMapPointView>& points, const uint8_t* pointDescriptors, float th, int ORBdist, std::vector<int32_t>& matches) const
    {
        matches.assign(CurrentFrame.n > 0 ? CurrentFrame.n : 1, -1); int n = 0;
        check(corb_search_by_projection_reloc(&CurrentFrame, hasMapPoint.data(), Tcw, points.data(), pointDescriptors, (int)points.size(), th, ORBdist,
                                              mbCheckOrientation ? 1 : 0, matches.data(), &n, device_), "corb_search_by_projection_reloc");
        matches.resize(CurrentFrame.n);
        return n;
    }
    // int SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize) (ORBmatcher.cc:540-655):
    // prevMatched = vbPrevMatched as n(F1) x 2 floats, read and written; matches12[i1] = feature of F2 or -1
    int SearchForInitialization(const CorbFrameView& F1, const CorbFrameView& F2, std::vector<float>& prevMatched, std::vector<int32_t>& matches12, int windowSize = 10) const
    {
        matches12.assign(F1.n > 0 ? F1.n : 1, -1); int n = 0;
        prevMatched.resize((size_t)2 * (F1.n > 0 ? F1.n : 1));
        check(corb_search_for_initialization(&F1, &F2, prevMatched.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, matches12.data(), &n, device_), "corb_search_for_initialization");
        matches12.resize(F1.n); prevMatched.resize((size_t)2 * F1.n);
        return n;
    }
    // int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th) (ORBmatcher.cc:425-538):
    // hasMatch[idx] = vpMatched[idx] != NULL on entry; points[i].valid = !isBad() && !spAlreadyFound.count(); matches[idx] = i (the point written into vpMatched[idx]) or -1
    int SearchByProjection(const CorbKeyFrameView& pKF, const std::vector<uint8_t>& hasMatch, const float Scw[16], const std::vector<CorbMapPointView>& points,
                           const uint8_t* pointDescriptors, int th, std::vector<int32_t>& matches) const
    {
        matches.assign(pKF.n > 0 ? pKF.n : 1, -1); int n = 0;
        check(corb_search_by_projection_scw(&pKF, hasMatch.data(), Scw, points.data(), pointDescriptors, (int)points.size(), (float)th, matches.data(), &n, device_),
              "corb_search_by_projection_scw");
        matches.resize(pKF.n);
        return n;
    }
    // int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) (:960-1116): bestIdx[i] = feature of pKF point i fuses into, or -1.
    // The caller then runs the reference's tail per point: Replace() by observation count if pKF->GetMapPoint(bestIdx) exists, else AddObservation/AddMapPoint.
    int Fuse(const CorbKeyFrameView& pKF, const float Tcw[16], const float Ow[3], const std::vector<CorbMapPointView>& points, const uint8_t* pointDescriptors,
             float th, std::vector<int32_t>& bestIdx) const
    { return fuse(pKF, Tcw, Ow, 0, points, pointDescriptors, th, bestIdx); }
    // int Fuse(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, th, vector<MapPoint*>& vpReplacePoint) (:1118-1241)
    int Fuse(const CorbKeyFrameView& pKF, const float Scw[16], const std::vector<CorbMapPointView>& points, const uint8_t* pointDescriptors,
             float th, std::vector<int32_t>& bestIdx) const
    { return fuse(pKF, Scw, nullptr, 1, points, pointDescriptors, th, bestIdx); }
    // int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, s12, R12, t12, th) (:1244-1468): matches12[i1] = feature of KF2 or -1
    int SearchBySim3(const CorbKeyFrameView& kf1, const CorbKeyFrameView& kf2, const float T1w[16], const float T2w[16],
                     const std::vector<CorbMapPointView>& points1, const uint8_t* desc1, const std::vector<CorbMapPointView>& points2, const uint8_t* desc2,
                     float s12, const float R12[9], const float t12[3], float th, std::vector<int32_t>& matches12) const
    {
        matches12.assign(kf1.n > 0 ? kf1.n : 1, -1); int n = 0;
        check(corb_search_by_sim3(&kf1, &kf2, T1w, T2w, points1.data(), desc1, points2.data(), desc2, s12, R12, t12, th, matches12.data(), &n, device_), "corb_search_by_sim3");
        matches12.resize(kf1.n);
        return n;
    }

private:
    int fuse(const CorbKeyFrameView& K, const float* T, const float* Ow, int sim3, const std::vector<CorbMapPointView>& pts, const uint8_t* desc, float th,
             std::vector<int32_t>& bestIdx) const
    {
        bestIdx.assign(pts.size() ? pts.size() : 1, -1); std::vector<int32_t> bestDist(bestIdx.size()); int n = 0;
        check(corb_fuse(&K, T, Ow, sim3, pts.data(), desc, (int)pts.size(), th, bestIdx.data(), bestDist.data(), &n, device_), "corb_fuse");
        bestIdx.resize(pts.size());
        return n;
    }
    int bow(int variant, const FeatureSet& A, const FeatureSet& B, std::vector<int32_t>& out) const
    {
        std::vector<float> a1 = A.angles(), a2 = B.angles();
        std::vector<uint8_t> ones(B.keysUn.size(), 1);
        CorbBowSide a{A.desc.data.data(), a1.data(), A.hasGoodMapPoint.data(), (int32_t)A.keysUn.size(), A.featVec.c()};
        CorbBowSide b{B.desc.data.data(), a2.data(), variant == 1 ? B.hasGoodMapPoint.data() : ones.data(), (int32_t)B.keysUn.size(), B.featVec.c()};
        out.assign(variant == 0 ? B.keysUn.size() : A.keysUn.size(), -1);
        int n = 0;
        check(corb_search_by_bow(variant, &a, &b, mfNNratio, mbCheckOrientation, out.data(), &n, device_), "corb_search_by_bow");
        return n;
    }
    float mfNNratio; bool mbCheckOrientation; int device_;
};

class Optimizer {
public:
    struct Graph {                                  // what Optimizer::BundleAdjustment reads from KeyFrames / MapPoints
        std::vector<float> Tcw;                     // K x 16, pKF->GetPose()
        std::vector<uint8_t> kfFixed;               // pKF->mnId==1 || pKF->getFixed()
        std::vector<float> worldPos;                // M x 3, pMP->GetWorldPos()
        std::vector<uint8_t> mpFixed;               // pMP->getFixed()
        std::vector<CorbBAEdge> observations;       // one per (MapPoint, KeyFrame) observation
        float fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;  // shared camera (used when intr is empty)
        std::vector<float> intr;                    // K x 5: pKF->fx, fy, cx, cy, mbf of every keyframe (e->fx = pKF->fx ..., Optimizer.cc:160-163, 189-193)
        const float* intrPtr() const { return intr.empty() ? nullptr : intr.data(); }
    };
    // static void GlobalBundleAdjustemnt(Cache*, int nIterations=5, bool* pbStopFlag=NULL, unsigned long nLoopKF=0, bool bRobust=true)
    // (sic -- the reference's spelling, Optimizer.h:45).  The caller applies the nLoopKF write-back policy
    // (SetPose/SetWorldPos when nLoopKF==0, else mTcwGBA/mPosGBA; Optimizer.cc:226-262) from the returned arrays.
    static CorbBAResult GlobalBundleAdjustemnt(const Graph& g, std::vector<float>& TcwOut, std::vector<float>& posOut,
                                               int nIterations = 5, volatile int* pbStopFlag = nullptr, bool bRobust = true, int device = 0)
    { return BundleAdjustment(g, TcwOut, posOut, nIterations, pbStopFlag, bRobust, device); }
    static CorbBAResult BundleAdjustment(const Graph& g, std::vector<float>& TcwOut, std::vector<float>& posOut,
                                         int nIterations, volatile int* pbStopFlag, bool bRobust, int device = 0)
    {
        CorbBAProblem p{(int32_t)(g.Tcw.size() / 16), (int32_t)(g.worldPos.size() / 3), (int32_t)g.observations.size(), g.Tcw.data(), g.kfFixed.data(),
                        g.worldPos.data(), g.mpFixed.data(), g.observations.data(), g.fx, g.fy, g.cx, g.cy, g.bf, g.intrPtr()};
        TcwOut.resize(g.Tcw.size()); posOut.resize(g.worldPos.size());
        CorbBAResult r{}; r.poses = TcwOut.data(); r.points = posOut.data();
        check(corb_ba_solve(&p, nIterations, bRobust, pbStopFlag, &r, device), "corb_ba_solve");
        return r;
    }

    // void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Cache* pCache) (Optimizer.cc:487-838): the graph holds the local
    // keyframes (free), the fixed keyframes and the local map points; vToErase[i] = observation i failed the chi2 / depth test
    // after the second optimize().  The caller erases those observations and writes the estimates back (Optimizer.cc:799-837).
    static CorbBAResult LocalBundleAdjustment(const Graph& g, std::vector<float>& TcwOut, std::vector<float>& posOut, std::vector<uint8_t>& vToErase,
                                              volatile int* pbStopFlag = nullptr, int device = 0)
    {
        const float hm = std::sqrt(5.991f), hs = std::sqrt(7.815f);
        const CorbBAStage st[2] = { {5, 1, 5.991f, 7.815f, 1, 0, 0, 0, 0, hm, hs}, {10, 0, 5.991f, 7.815f, 1, 0, 1, 0, 0, hm, hs} };   // final test: every edge (allow_reactivate)
        return staged(g, st, 2, TcwOut, posOut, vToErase, pbStopFlag, device);
    }

    // int PoseOptimization(Frame* pFrame) (Optimizer.cc:272-485): Tcw in/out, mvbOutlier out, returns nInitialCorrespondences - nBad.
    // One observation per matched MapPoint: world position, mvKeysUn[i].pt, mvuRight[i] (<0 = monocular), mvInvLevelSigma2[octave].
    struct FrameObservations { std::vector<float> worldPos, u, v, uRight, invSigma2; float fx, fy, cx, cy, bf; };
    static int PoseOptimization(float Tcw[16], const FrameObservations& f, std::vector<uint8_t>& mvbOutlier, int device = 0)
    {
        CorbPoseOptFrame F{Tcw, (int32_t)f.u.size(), f.worldPos.data(), f.u.data(), f.v.data(), f.uRight.data(), f.invSigma2.data(), f.fx, f.fy, f.cx, f.cy, f.bf};
        mvbOutlier.assign(f.u.size() ? f.u.size() : 1, 0);
        uint8_t* op = mvbOutlier.data(); float out[16]; int32_t ninl = 0;
        check(corb_pose_optimization_batch(&F, 1, out, &op, &ninl, device), "corb_pose_optimization_batch");
        mvbOutlier.resize(f.u.size());
        for (int i = 0; i < 16; i++) Tcw[i] = out[i];
        return ninl;
    }

    // int OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12, float th2, bool bFixScale) (Optimizer.cc:1119-1311).
    // One entry per correspondence the reference turns into an edge pair (pMP1, pMP2 good, i2 >= 0): camera-frame points, keypoints, sigmas.
    struct Sim3Correspondences { std::vector<float> P3D1c, P3D2c, obs1, obs2, invSigma2_1, invSigma2_2; float fx1, fy1, cx1, cy1, fx2, fy2, cx2, cy2; };
    // R12 (row-major 3x3), t12, s12 = g2oS12 in / out; removed[i] = 1 -> vpMatches1[vnIndexEdge[i]] = NULL; returns nIn
    static int OptimizeSim3(const Sim3Correspondences& c, double R12[9], double t12[3], double& s12, float th2, bool bFixScale, std::vector<uint8_t>& removed, int device = 0)
    {
        CorbSim3Problem P{(int32_t)c.invSigma2_1.size(), c.P3D1c.data(), c.P3D2c.data(), c.obs1.data(), c.obs2.data(), c.invSigma2_1.data(), c.invSigma2_2.data(),
                          c.fx1, c.fy1, c.cx1, c.cy1, c.fx2, c.fy2, c.cx2, c.cy2};
        removed.assign(P.n ? P.n : 1, 0); uint8_t* rp = removed.data(); int32_t nIn = 0;
        check(corb_optimize_sim3(&P, 1, R12, t12, &s12, th2, bFixScale ? 1 : 0, &rp, &nIn, nullptr, device), "corb_optimize_sim3");
        removed.resize(P.n);
        return nIn;
    }

    // void OptimizeEssentialGraph(Cache*, KeyFrame* pLoopKF, KeyFrame* pCurKF, NonCorrectedSim3, CorrectedSim3, LoopConnections, bFixScale) (Optimizer.cc:840-1117).
    // The caller flattens lines 866-1037: one Sim3 (quaternion x y z w, t, s) per keyframe, fixed flags, EdgeSim3 list (vertex 0, vertex 1, Sji).
    struct EssentialGraph { std::vector<double> S; std::vector<uint8_t> fixed; std::vector<int32_t> vi, vj; std::vector<double> Sji;
                            std::vector<int32_t> pointRef; std::vector<float> points; };
    // S updated in place; TiwOut = K x 16 corrected SE3 poses [R | t/s]; g.points corrected through their reference keyframe
    static int OptimizeEssentialGraph(EssentialGraph& g, std::vector<float>& TiwOut, bool bFixScale, int nIterations = 20, int device = 0)
    {
        const int K = (int)(g.S.size() / 8); TiwOut.resize((size_t)16 * K); int32_t its = 0;
        check(corb_optimize_essential_graph(K, g.S.data(), g.fixed.data(), (int)g.vi.size(), g.vi.data(), g.vj.data(), g.Sji.data(), nIterations, bFixScale ? 1 : 0,
                                            TiwOut.data(), (int)g.pointRef.size(), g.pointRef.data(), g.points.data(), nullptr, &its, device), "corb_optimize_essential_graph");
        return its;
    }

private:
    static CorbBAResult staged(const Graph& g, const CorbBAStage* st, int n, std::vector<float>& TcwOut, std::vector<float>& posOut, std::vector<uint8_t>& outlier,
                               volatile int* pbStopFlag, int device)
    {
        CorbBAProblem p{(int32_t)(g.Tcw.size() / 16), (int32_t)(g.worldPos.size() / 3), (int32_t)g.observations.size(), g.Tcw.data(), g.kfFixed.data(),
                        g.worldPos.data(), g.mpFixed.data(), g.observations.data(), g.fx, g.fy, g.cx, g.cy, g.bf, g.intrPtr()};
        TcwOut.resize(g.Tcw.size()); posOut.resize(g.worldPos.size()); outlier.assign(g.observations.size() ? g.observations.size() : 1, 0);
        CorbBAResult r{}; r.poses = TcwOut.data(); r.points = posOut.data();
        check(corb_ba_solve_staged(&p, st, n, pbStopFlag, &r, outlier.data(), device, nullptr), "corb_ba_solve_staged");
        outlier.resize(g.observations.size());
        return r;
    }
};

// MapPoint::ComputeDistinctiveDescriptors for a batch (MapPoint.cc:337-402): bestRow[p] relative to offset[p]
inline std::vector<int32_t> ComputeDistinctiveDescriptors(const std::vector<uint8_t>& stackedDescriptors, const std::vector<int32_t>& offset, int device = 0)
{
    std::vector<int32_t> best(offset.size() > 1 ? offset.size() - 1 : 1);
    check(corb_distinctive_descriptors(stackedDescriptors.data(), offset.data(), (int)offset.size() - 1, best.data(), device), "corb_distinctive_descriptors");
    best.resize(offset.size() - 1);
    return best;
}
// MapFusion::insertServerMapToGlobleMap arithmetic (S/src/MapFusion.cpp:622-658): Tcw <- Tcw * To2n, p <- Rwc (p - tcw), in place
inline void RebaseMap(const float To2n[16], std::vector<float>& Tcw, std::vector<float>& worldPos, int device = 0)
{ check(corb_rebase_map(To2n, Tcw.data(), (int)(Tcw.size() / 16), worldPos.data(), (int)(worldPos.size() / 3), device), "corb_rebase_map"); }

}  // namespace corb
