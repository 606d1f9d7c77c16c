// corb_adapter_orbslam.hpp -- the reference's ORBmatcher / Optimizer SIGNATURES on top of corb_host.hpp.
//
//   int  ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)                        corbslam_client/include/ORBmatcher.h:57
//   int  ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)                     :58
//   int  ORBmatcher::SearchByBoWInServer(KeyFrame*, KeyFrame*, vector<MapPoint*>&)             :60
//   int  ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, vector<pair<size_t,size_t>>&, bool)   :66-67
//   int  ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>&, int th)  :52-53 (loop closing / map fusion)
//   int  ORBmatcher::SearchForInitialization(Frame&, Frame&, vector<cv::Point2f>&, vector<int>&, int windowSize)         :62-63 (monocular initialisation)
//   void Optimizer::BundleAdjustment(const vector<KeyFrame*>&, const vector<MapPoint*>&, int, bool*, unsigned long, bool)   include/Optimizer.h:42-44
//   void Optimizer::GlobalBundleAdjustemnt(Cache*, int, bool*, unsigned long, bool)            :45-46
//   int  Optimizer::PoseOptimization(Frame*)                                                   :51
//
// The adapters FLATTEN the reference's pointer graph into the arrays of the C-ABI (KeyFrame* / Frame& -> descriptors, keypoints, mvuRight,
// "has a good MapPoint" flags, DBoW2::FeatureVector; Cache* -> poses, per-keyframe intrinsics, points, observations), call the accelerator,
// and MAP the indices BACK (match index -> MapPoint*, estimates -> SetPose / SetWorldPos or mTcwGBA / mPosGBA by the nLoopKF policy of
// Optimizer.cc:216-262).  They are templates over the reference's class types and touch them only through the member names the reference
// itself uses at the cited lines, so that
//   * on a tree with the reference's headers (OpenCV, Boost, ROS present) the aliases at the bottom of this file instantiate them with
//     ORB_SLAM2::KeyFrame / Frame / MapPoint / Cache and cv::Mat -- a drop-in for ORBmatcher.cc / Optimizer.cc on this path;
//   * in this repository (none of those dependencies exist) the SAME template code is compiled and run against small test doubles that carry
//     those member names (tests/host/mock_orbslam.hpp, driven by tests/host/adapter_main.cpp on the GPU box) and its outputs are compared
//     with the oracle (tests/test_gpu_host.py).
#pragma once
#include "corb_host.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <set>
#include <thread>
#include <utility>
#include <vector>
#include <cmath>
#include <type_traits>

namespace corb {
namespace adapt {

// ---- the only operations used on the reference's matrix / keypoint types ----
template <class Mat> inline float matf(const Mat& m, int r, int c) { return m.template at<float>(r, c); }
template <class Mat> inline float matf(const Mat& m, int i) { return m.template at<float>(i); }
template <class Mat> inline const uint8_t* desc_row(const Mat& m, int r) { return m.template ptr<uint8_t>(r); }            // cv::Mat::ptr<uchar>(row)
template <class Mat> struct MatFactory;              // MatFactory<Mat>::from_floats(rows, cols, const float*) -> an owning CV_32F matrix
template <class KP> inline CorbKeyPoint to_kp(const KP& k)
{ CorbKeyPoint o; o.x = k.pt.x; o.y = k.pt.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id; return o; }

// DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned int>>: ascending node ids, as the merge-walk of the matchers needs them
template <class FV> inline FeatureVector flatten_featvec(const FV& fv)
{ FeatureVector o; for (auto it = fv.begin(); it != fv.end(); ++it) o.add((uint32_t)it->first, it->second); return o; }

// What SearchByBoW / SearchForTriangulation read from a KeyFrame: mDescriptors, mvKeysUn, mvuRight, GetMapPointMatches(), mFeatVec.
// good_only: vpMapPoints[i] && !isBad() (BoW matchers, ORBmatcher.cc:194-199, 693-699); otherwise any non-NULL entry (GetMapPoint(idx) != NULL,
// SearchForTriangulation :835-838, :858-862)
template <class KeyFrame> FeatureSet flatten_keyframe(KeyFrame* pKF, bool good_only, bool angle_from_mvKeys = false)
{
    FeatureSet s; const int N = pKF->N;
    s.desc.data.resize((size_t)N * 32); s.keysUn.resize(N); s.uRight.resize(N); s.hasGoodMapPoint.assign(N, 0);
    const auto vpMP = pKF->GetMapPointMatches();
    for (int i = 0; i < N; i++) {
        std::memcpy(&s.desc.data[(size_t)i * 32], desc_row(pKF->mDescriptors, i), 32);
        s.keysUn[i] = to_kp(pKF->mvKeysUn[i]);
        if (angle_from_mvKeys) s.keysUn[i].angle = pKF->mvKeys[i].angle;       // the "frame" side of SearchByBoWInServer reads F->mvKeys[..].angle (:375)
        s.uRight[i] = pKF->mvuRight[i];
        auto* pMP = i < (int)vpMP.size() ? vpMP[i] : nullptr;
        s.hasGoodMapPoint[i] = pMP && (!good_only || !pMP->isBad()) ? 1 : 0;
    }
    s.featVec = flatten_featvec(pKF->mFeatVec);
    return s;
}
// The Frame side of SearchByBoW(KeyFrame*, Frame&): mDescriptors, mFeatVec and F.mvKeys[..].angle (ORBmatcher.cc:241)
template <class Frame> FeatureSet flatten_frame(Frame& F)
{
    FeatureSet s; const int N = F.N;
    s.desc.data.resize((size_t)N * 32); s.keysUn.resize(N); s.uRight.resize(N); s.hasGoodMapPoint.assign(N, 1);
    for (int i = 0; i < N; i++) {
        std::memcpy(&s.desc.data[(size_t)i * 32], desc_row(F.mDescriptors, i), 32);
        s.keysUn[i] = to_kp(F.mvKeysUn[i]); s.keysUn[i].angle = F.mvKeys[i].angle;
        s.uRight[i] = F.mvuRight[i];
    }
    s.featVec = flatten_featvec(F.mFeatVec);
    return s;
}

template <class KeyFrame, class Frame, class MapPoint, class Mat>
class ORBmatcherT {
public:
    static constexpr int TH_LOW = corb::ORBmatcher::TH_LOW, TH_HIGH = corb::ORBmatcher::TH_HIGH, HISTO_LENGTH = corb::ORBmatcher::HISTO_LENGTH;
    ORBmatcherT(float nnratio = 0.6f, bool checkOri = true, int device = 0) : m_(nnratio, checkOri, device) {}

    static int DescriptorDistance(const Mat& a, const Mat& b) { return corb::ORBmatcher::DescriptorDistance(desc_row(a, 0), desc_row(b, 0)); }

    // ORBmatcher.cc:162-291
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
    {
        const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(nullptr));
        std::vector<int32_t> m;
        const int n = m_.SearchByBoW(flatten_keyframe(pKF, true), flatten_frame(F), m);
        for (int iF = 0; iF < F.N; iF++) if (m[iF] >= 0) vpMapPointMatches[iF] = vpMapPointsKF[m[iF]];
        return n;
    }
    // ORBmatcher.cc:294-423 (server-side map fusion: the "frame" is a KeyFrame of the other map)
    int SearchByBoWInServer(KeyFrame* pKF, KeyFrame* F, std::vector<MapPoint*>& vpMapPointMatches)
    {
        const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint*>(F->N, static_cast<MapPoint*>(nullptr));
        FeatureSet fs = flatten_keyframe(F, true, true); std::fill(fs.hasGoodMapPoint.begin(), fs.hasGoodMapPoint.end(), 1);
        std::vector<int32_t> m;
        const int n = m_.SearchByBoWInServer(flatten_keyframe(pKF, true), fs, m);
        for (int iF = 0; iF < F->N; iF++) if (m[iF] >= 0) vpMapPointMatches[iF] = vpMapPointsKF[m[iF]];
        return n;
    }
    // ORBmatcher.cc:657-790 (loop closing / Sim3): vpMatches12[i1] = the MapPoint of the matched feature of pKF2
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12)
    {
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(nullptr));
        std::vector<int32_t> m;
        const int n = m_.SearchByBoW_KF(flatten_keyframe(pKF1, true), flatten_keyframe(pKF2, true), m);
        for (size_t i1 = 0; i1 < vpMatches12.size() && i1 < m.size(); i1++) if (m[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m[i1]];
        return n;
    }
    // ORBmatcher.cc:792-958.  The epipole (:799-808): C2 = R2w*Cw + t2w in float matrices (cv::gemm accumulates a float product in double)
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo)
    {
        const Mat Cw = pKF1->GetCameraCenter(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
        float C2[3];
        for (int i = 0; i < 3; i++) {
            double acc = 0; for (int k = 0; k < 3; k++) acc += (double)matf(R2w, i, k) * (double)matf(Cw, k);
            C2[i] = (float)acc + matf(t2w, i);
        }
        const float invz = 1.0f / C2[2];
        const float ex = pKF2->fx * C2[0] * invz + pKF2->cx, ey = pKF2->fy * C2[1] * invz + pKF2->cy;
        float F[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F[3 * i + j] = matf(F12, i, j);
        return m_.SearchForTriangulation(flatten_keyframe(pKF1, false), flatten_keyframe(pKF2, false), F, ex, ey, pKF2->mvScaleFactors, pKF2->mvLevelSigma2,
                                         vMatchedPairs, bOnlyStereo);
    }
    // ORBmatcher.cc:425-538 -- LoopClosing::ComputeSim3 (C/src/LoopClosing.cc:377) and the server's map fusion (S/src/GlobalOptimize.cpp:199) call it with the loop
    // keyframe's covisible points right before CorrectLoop / the global BA.  What the reference reads: pKF->fx.., mnMinX.., mvScaleFactors, mfLogScaleFactor, mvKeysUn,
    // mDescriptors (GetFeaturesInArea + the loop body); pMP->isBad / GetWorldPos / GetNormal / Get{Min,Max}DistanceInvariance / PredictScale / GetDescriptor.  The two
    // raw distances are read through GetMinDistance() / GetMaxDistance() (the accessors MapPoint.h gains), the getters' 0.8f / 1.2f are applied on the device.
    int SearchByProjection(KeyFrame* pKF, Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th)
    {
        const int N = pKF->N;
        std::vector<CorbKeyPoint> keys(N); std::vector<uint8_t> desc((size_t)N * 32), hasMatch(N, 0);
        for (int i = 0; i < N; i++) { keys[i] = to_kp(pKF->mvKeysUn[i]); std::memcpy(&desc[(size_t)i * 32], desc_row(pKF->mDescriptors, i), 32); hasMatch[i] = (i < (int)vpMatched.size() && vpMatched[i]) ? 1 : 0; }
        CorbKeyFrameView K; std::memset(&K, 0, sizeof(K));
        K.keys_un = keys.data(); K.u_right = pKF->mvuRight.data(); K.desc = desc.data(); K.n = N;
        K.min_x = (float)pKF->mnMinX; K.min_y = (float)pKF->mnMinY; K.max_x = (float)pKF->mnMaxX; K.max_y = (float)pKF->mnMaxY;
        K.scale = pKF->mvScaleFactors.data(); K.inv_level_sigma2 = pKF->mvInvLevelSigma2.data(); K.nlevels = (int32_t)pKF->mvScaleFactors.size();
        K.log_scale_factor = pKF->mfLogScaleFactor; K.fx = pKF->fx; K.fy = pKF->fy; K.cx = pKF->cx; K.cy = pKF->cy; K.bf = pKF->mbf;
        std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());       // :441-442
        spAlreadyFound.erase(static_cast<MapPoint*>(nullptr));
        std::vector<CorbMapPointView> pts(vpPoints.size()); std::vector<uint8_t> pdesc(vpPoints.size() * 32 + 32, 0);
        for (size_t i = 0; i < vpPoints.size(); i++) {
            MapPoint* pMP = vpPoints[i]; CorbMapPointView& v = pts[i]; std::memset(&v, 0, sizeof(v));
            if (!pMP || pMP->isBad() || spAlreadyFound.count(pMP)) continue;          // :452 (valid stays 0)
            const Mat X = pMP->GetWorldPos(), Nn = pMP->GetNormal(), D = pMP->GetDescriptor();
            for (int a = 0; a < 3; a++) { v.world[a] = matf(X, a); v.normal[a] = matf(Nn, a); }
            v.min_distance = pMP->GetMinDistance(); v.max_distance = pMP->GetMaxDistance(); v.valid = 1;
            std::memcpy(&pdesc[i * 32], desc_row(D, 0), 32);
        }
        float S[16]; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) S[4 * r + c] = matf(Scw, r, c);
        std::vector<int32_t> m;
        const int n = m_.SearchByProjection(K, hasMatch, S, pts, pdesc.data(), th, m);
        if ((int)vpMatched.size() < N) vpMatched.resize(N, static_cast<MapPoint*>(nullptr));
        for (int idx = 0; idx < N; idx++) if (m[idx] >= 0) vpMatched[idx] = vpPoints[m[idx]];      // :530
        return n;
    }
    // ORBmatcher.cc:540-655 (Tracking::MonocularInitialization, C/src/Tracking.cc:606).  Point2f = cv::Point2f (anything with float x, y).  What the reference reads:
    // F1.mvKeysUn (octave, angle), F1.mDescriptors, F2.GetFeaturesInArea (mvKeysUn, the grid over mnMinX .. mnMaxY), F2.mDescriptors, F2.mvKeysUn[..].angle / .pt
    template <class Point2f>
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10)
    {
        auto view = [](Frame& F, std::vector<CorbKeyPoint>& k, std::vector<uint8_t>& d) {
            k.resize(F.N); d.resize((size_t)F.N * 32);
            for (int i = 0; i < F.N; i++) { k[i] = to_kp(F.mvKeysUn[i]); std::memcpy(&d[(size_t)i * 32], desc_row(F.mDescriptors, i), 32); }
            CorbFrameView v; std::memset(&v, 0, sizeof(v));
            v.keys_un = k.data(); v.desc = d.data(); v.n = F.N; v.min_x = Frame::mnMinX; v.min_y = Frame::mnMinY; v.max_x = Frame::mnMaxX; v.max_y = Frame::mnMaxY;
            return v;
        };
        std::vector<CorbKeyPoint> k1, k2; std::vector<uint8_t> d1, d2;
        const CorbFrameView v1 = view(F1, k1, d1), v2 = view(F2, k2, d2);
        std::vector<float> pm((size_t)2 * F1.N);
        for (int i = 0; i < F1.N; i++) { pm[2 * i] = vbPrevMatched[i].x; pm[2 * i + 1] = vbPrevMatched[i].y; }
        std::vector<int32_t> m;
        const int n = m_.SearchForInitialization(v1, v2, pm, m, windowSize);
        vnMatches12 = std::vector<int>(F1.N, -1);                                   // :543
        for (int i = 0; i < F1.N; i++) { vnMatches12[i] = m[i]; if (m[i] >= 0) { vbPrevMatched[i].x = pm[2 * i]; vbPrevMatched[i].y = pm[2 * i + 1]; } }      // :650-653
        return n;
    }
private:
    corb::ORBmatcher m_;
};

// ---- Converter::toSE3Quat -> SE3Quat -> Converter::toCvMat round trip of a float pose (what the reference writes back for a keyframe whose vertex
// was fixed by id only: mnId == 1 is fixed in the graph but not skipped by the write-back loop, Optimizer.cc:94, 222) ----
inline void se3_roundtrip(const float* T, float* out)
{
    const double R[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
    double q[4]; double t = R[0] + R[4] + R[8];
    if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
    if (q[3] < 0) for (double& v : q) v = -v;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= n;
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2], twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    const double Ro[9] = { 1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy) };
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) out[4 * r + c] = (float)Ro[3 * r + c]; out[4 * r + 3] = T[4 * r + 3]; }
    out[12] = out[13] = out[14] = 0; out[15] = 1;
}

// bool* pbStopFlag (the reference's type, polled by g2o between LM trials) -> the C-ABI's volatile int*: a watcher copies it while the call runs
struct StopBridge {
    volatile int flag = 0; std::atomic<bool> done{false}; std::thread th;
    explicit StopBridge(bool* pb) { if (pb) { flag = *pb ? 1 : 0; th = std::thread([this, pb] { while (!done.load()) { if (*pb) flag = 1; std::this_thread::sleep_for(std::chrono::microseconds(200)); } }); } }
    ~StopBridge() { done.store(true); if (th.joinable()) th.join(); }
    volatile int* ptr(bool* pb) { return pb ? &flag : nullptr; }
};

template <class KeyFrame, class Frame, class MapPoint, class Cache, class Mat>
class OptimizerT {
public:
    // Optimizer.cc:43-51
    static void GlobalBundleAdjustemnt(Cache* pCache, int nIterations = 5, bool* pbStopFlag = nullptr, const unsigned long nLoopKF = 0, const bool bRobust = true, int device = 0)
    {
        std::vector<KeyFrame*> vpKFs = pCache->getAllKeyFramesInMap();
        std::vector<MapPoint*> vpMP = pCache->GetAllMapPointsFromMap();
        BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust, device);
    }

    // Optimizer.cc:54-270.  Graph build :84-203 (vertex ids = mnId, so the solver's ordering is ascending mnId: the arrays are sorted that way);
    // write-back :216-262.
    static void BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations, bool* pbStopFlag,
                                 const unsigned long nLoopKF, const bool bRobust, int device = 0)
    {
        unsigned long maxKFid = 0;
        std::vector<KeyFrame*> kfs;                                       // non-bad keyframes, ascending mnId
        for (KeyFrame* pKF : vpKFs) { if (pKF && pKF->mnId > maxKFid) maxKFid = pKF->mnId; if (pKF->isBad()) continue; kfs.push_back(pKF); }
        std::sort(kfs.begin(), kfs.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
        kfs.erase(std::unique(kfs.begin(), kfs.end()), kfs.end());
        std::map<KeyFrame*, int> kfIndex; for (size_t i = 0; i < kfs.size(); i++) kfIndex[kfs[i]] = (int)i;
        corb::Optimizer::Graph g;
        g.Tcw.resize(16 * kfs.size()); g.kfFixed.resize(kfs.size()); g.intr.resize(5 * kfs.size());
        for (size_t i = 0; i < kfs.size(); i++) {
            KeyFrame* pKF = kfs[i];
            const Mat T = pKF->GetPose();
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) g.Tcw[16 * i + 4 * r + c] = matf(T, r, c);
            g.kfFixed[i] = (pKF->mnId == 1 || pKF->getFixed()) ? 1 : 0;       // :94
            float* in = &g.intr[5 * i]; in[0] = pKF->fx; in[1] = pKF->fy; in[2] = pKF->cx; in[3] = pKF->cy; in[4] = pKF->mbf;   // e->fx = pKF->fx ... e->bf = pKF->mbf (:160-163, :189-193)
        }
        if (!kfs.empty()) { g.fx = kfs[0]->fx; g.fy = kfs[0]->fy; g.cx = kfs[0]->cx; g.cy = kfs[0]->cy; g.bf = kfs[0]->mbf; }
        std::vector<MapPoint*> mps;                                       // non-bad map points, ascending mnId
        for (MapPoint* pMP : vpMP) if (pMP && !pMP->isBad()) mps.push_back(pMP);
        std::sort(mps.begin(), mps.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
        mps.erase(std::unique(mps.begin(), mps.end()), mps.end());
        std::vector<int> nEdges(mps.size(), 0);
        g.worldPos.resize(3 * mps.size()); g.mpFixed.resize(mps.size());
        for (size_t m = 0; m < mps.size(); m++) {
            MapPoint* pMP = mps[m];
            const Mat X = pMP->GetWorldPos();
            for (int a = 0; a < 3; a++) g.worldPos[3 * m + a] = matf(X, a);
            g.mpFixed[m] = pMP->getFixed() ? 1 : 0;                       // :120
            const std::map<KeyFrame*, size_t> observations = pMP->GetObservations();
            for (auto mit = observations.begin(); mit != observations.end(); ++mit) {
                KeyFrame* pKF = mit->first;
                if (pKF->isBad() || pKF->mnId > maxKFid) continue;        // :131-132
                auto it = kfIndex.find(pKF); if (it == kfIndex.end()) continue;   // :134-135 (allKFId)
                nEdges[m]++;
                const auto& kpUn = pKF->mvKeysUn[mit->second];
                CorbBAEdge e; e.pose = it->second; e.point = (int32_t)m; e.u = kpUn.pt.x; e.v = kpUn.pt.y;
                e.u_right = pKF->mvuRight[mit->second];                   // < 0: EdgeSE3ProjectXYZ, else EdgeStereoSE3ProjectXYZ (:141, :166)
                e.inv_sigma2 = pKF->mvInvLevelSigma2[kpUn.octave];
                g.observations.push_back(e);
            }
        }
        std::vector<float> Tout, Xout;
        {
            StopBridge stop(pbStopFlag);
            corb::Optimizer::BundleAdjustment(g, Tout, Xout, nIterations, stop.ptr(pbStopFlag), bRobust, device);
        }
        // Keyframes (:216-237): every non-bad keyframe that is not getFixed() -- including mnId == 1, whose estimate did not move
        for (size_t i = 0; i < kfs.size(); i++) {
            KeyFrame* pKF = kfs[i];
            if (pKF->getFixed()) continue;
            float T[16];
            if (g.kfFixed[i]) se3_roundtrip(&g.Tcw[16 * i], T); else std::memcpy(T, &Tout[16 * i], sizeof(T));
            if (nLoopKF == 0) { pKF->SetPose(MatFactory<Mat>::from_floats(4, 4, T)); pKF->mpCacher->addUpdateKeyframe(pKF); }
            else { pKF->mTcwGBA = MatFactory<Mat>::from_floats(4, 4, T); pKF->mnBAGlobalForKF = nLoopKF; }
        }
        // Points (:240-262): those that got at least one edge (vbNotIncludedMP) and are not getFixed()
        for (size_t m = 0; m < mps.size(); m++) {
            MapPoint* pMP = mps[m];
            if (nEdges[m] == 0 || pMP->getFixed()) continue;
            if (nLoopKF == 0) { pMP->SetWorldPos(MatFactory<Mat>::from_floats(3, 1, &Xout[3 * m])); pMP->getCache()->addUpdateMapPoint(pMP); pMP->UpdateNormalAndDepth(); }
            else { pMP->mPosGBA = MatFactory<Mat>::from_floats(3, 1, &Xout[3 * m]); pMP->mnBAGlobalForKF = nLoopKF; }
        }
    }

    // Optimizer.cc:272-485
    static int PoseOptimization(Frame* pFrame, int device = 0)
    {
        corb::Optimizer::FrameObservations f;
        f.fx = pFrame->fx; f.fy = pFrame->fy; f.cx = pFrame->cx; f.cy = pFrame->cy; f.bf = pFrame->mbf;
        std::vector<size_t> vnIndexEdge;
        const int N = pFrame->N;
        for (int i = 0; i < N; i++) {
            MapPoint* pMP = pFrame->mvpMapPoints[i].getMapPoint();        // (CORB-SLAM keeps LightMapPoint handles in the Frame, :314)
            if (!pMP) continue;
            pFrame->mvbOutlier[i] = false;                                // :321, :352
            const auto& kpUn = pFrame->mvKeysUn[i];
            const Mat Xw = pMP->GetWorldPos();
            for (int a = 0; a < 3; a++) f.worldPos.push_back(matf(Xw, a));
            f.u.push_back(kpUn.pt.x); f.v.push_back(kpUn.pt.y); f.uRight.push_back(pFrame->mvuRight[i]);
            f.invSigma2.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
            vnIndexEdge.push_back((size_t)i);
        }
        const int nInitialCorrespondences = (int)vnIndexEdge.size();
        if (nInitialCorrespondences < 3) return 0;                        // :396-397
        float Tcw[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw[4 * r + c] = matf(pFrame->mTcw, r, c);
        std::vector<uint8_t> outl;
        const int nGood = corb::Optimizer::PoseOptimization(Tcw, f, outl, device);
        for (size_t k = 0; k < vnIndexEdge.size(); k++) pFrame->mvbOutlier[vnIndexEdge[k]] = outl[k] != 0;
        pFrame->SetPose(MatFactory<Mat>::from_floats(4, 4, Tcw));        // :480-482
        return nGood;
    }
};

// ---- KeyFrame / MapPoint objects <-> device-resident store records (what the reference moves as boost text archives) ----
// The client side of Cache::runUpdateToServer (C/src/Cache.cc:322-375) files its new keyframes and map points in the stores and calls corb_map_push_ex;
// the server side of MapFusion::insertKeyFrameToMap / insertMapPointToMap (S/src/MapFusion.cpp:31-190) finds them in its own stores at the destination
// slots, re-bases them (corb_rebase_map_store) and runs the fused global BA on the records (corb_ba_solve_store); ReadBack* applies what the
// records hold to the objects with the policy of Optimizer.cc:216-262 (nLoopKF == 0: SetPose / SetWorldPos + cache marks; else mTcwGBA / mPosGBA).
template <class KeyFrame, class MapPoint, class Mat>
class MapStoreT {
public:
    // the record of one keyframe: mvKeysUn (the keypoints the optimiser and the matchers read; rectified stereo: = mvKeys), mDescriptors, mvuRight, mvDepth,
    // mFeatVec, mvpMapPoints as ids, and the header fields of KeyFrame.h:65-79
    static void PutKeyFrame(CorbKfStore* store, int slot, KeyFrame* pKF, int clientId = 0)
    {
        const int N = pKF->N;
        std::vector<CorbKeyPoint> kp(N); std::vector<uint8_t> desc((size_t)N * 32); std::vector<float> depth(N, -1.f); std::vector<uint64_t> ids(N, CORB_NO_MAP_POINT);
        const auto vpMP = pKF->GetMapPointMatches();
        for (int i = 0; i < N; i++) {
            kp[i] = to_kp(pKF->mvKeysUn[i]);
            if (!pKF->mDescriptors.empty()) std::memcpy(&desc[(size_t)i * 32], desc_row(pKF->mDescriptors, i), 32);
            auto* pMP = i < (int)vpMP.size() ? vpMP[i] : nullptr;
            if (pMP) ids[i] = (uint64_t)pMP->mnId;
        }
        // mvDepth travels with the keyframe like in the reference's serialisation (KeyFrame.h:68-79; UnprojectStereo reads it on the server); a keyframe
        // without the vector (monocular) files -1
        const float* pdepth = (int)pKF->mvDepth.size() == N ? pKF->mvDepth.data() : depth.data();
        check(corb_kf_store_put_host(store, slot, kp.data(), desc.data(), pKF->mvuRight.data(), pdepth, N, (uint64_t)pKF->mnId), "corb_kf_store_put_host");
        CorbKeyFrameMeta m; std::memset(&m, 0, sizeof(m));
        m.id = (uint64_t)pKF->mnId; m.client_id = clientId; m.flags = (pKF->isBad() ? CORB_KF_BAD : 0u) | (pKF->getFixed() ? CORB_KF_FIXED : 0u);
        m.fx = pKF->fx; m.fy = pKF->fy; m.cx = pKF->cx; m.cy = pKF->cy; m.bf = pKF->mbf;
        m.nlevels = (int32_t)std::min<size_t>(pKF->mvInvLevelSigma2.size(), 16);
        for (int l = 0; l < m.nlevels; l++) m.inv_level_sigma2[l] = pKF->mvInvLevelSigma2[l];
        const Mat T = pKF->GetPose();
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { m.Tcw[4 * r + c] = matf(T, r, c); m.TcwGBA[4 * r + c] = r == c ? 1.f : 0.f; }
        check(corb_kf_store_set_meta(store, slot, &m), "corb_kf_store_set_meta");
        check(corb_kf_store_set_map_points(store, slot, ids.data()), "corb_kf_store_set_map_points");
        const FeatureVector fv = flatten_featvec(pKF->mFeatVec);
        if (!fv.node_id.empty()) { CorbFeatVec c = fv.c(); check(corb_kf_store_set_bow(store, slot, &c), "corb_kf_store_set_bow"); }
    }
    // records first .. first + n of the map-point store: header fields of MapPoint.h:52-72 and mObservations as (keyframe mnId, feature index), ascending in
    // the keyframe id like the reference's std::map<LightKeyFrame, size_t>
    static void PutMapPoints(CorbMpStore* store, int first, const std::vector<MapPoint*>& vpMP, int clientId = 0)
    {
        std::vector<CorbMapPointRecord> rec(vpMP.size()); std::vector<int32_t> off(vpMP.size() + 1, 0); std::vector<uint64_t> okf; std::vector<uint32_t> oidx;
        for (size_t m = 0; m < vpMP.size(); m++) {
            MapPoint* pMP = vpMP[m]; CorbMapPointRecord& r = rec[m]; std::memset(&r, 0, sizeof(r));
            r.id = (uint64_t)pMP->mnId; r.client_id = clientId; r.flags = (pMP->isBad() ? CORB_MP_BAD : 0u) | (pMP->getFixed() ? CORB_MP_FIXED : 0u);
            if (auto* pRef = pMP->GetReferenceKeyFrame()) r.ref_kf_id = (uint64_t)pRef->mnId;      // mpRefKF (MapPoint.h:66)
            const Mat X = pMP->GetWorldPos();
            for (int a = 0; a < 3; a++) r.world_pos[a] = matf(X, a);
            std::vector<std::pair<uint64_t, uint32_t>> obs;
            const auto observations = pMP->GetObservations();
            for (auto it = observations.begin(); it != observations.end(); ++it) obs.emplace_back((uint64_t)it->first->mnId, (uint32_t)it->second);
            std::sort(obs.begin(), obs.end());
            for (auto& o : obs) { okf.push_back(o.first); oidx.push_back(o.second); }
            r.n_obs = (int32_t)obs.size(); off[m + 1] = (int32_t)okf.size();
        }
        check(corb_mp_store_put_host(store, first, (int)vpMP.size(), rec.data(), off.data(), okf.data(), oidx.data()), "corb_mp_store_put_host");
        // the public tracking / local-mapping / loop-closing fields the reference's archive moves as well (MapPoint.h:53-58, 151-174): they ride behind the lists
        std::vector<CorbMapPointScratch> sc(vpMP.size());
        for (size_t m = 0; m < vpMP.size(); m++) {
            MapPoint* pMP = vpMP[m]; CorbMapPointScratch& c = sc[m]; std::memset(&c, 0, sizeof(c));
            c.first_kf_id = (int64_t)pMP->mnFirstKFid; c.first_frame = (int64_t)pMP->mnFirstFrame; c.n_obs_weight = pMP->nObs;
            c.track_proj_x = pMP->mTrackProjX; c.track_proj_y = pMP->mTrackProjY; c.track_proj_xr = pMP->mTrackProjXR; c.track_view_cos = pMP->mTrackViewCos;
            c.track_in_view = pMP->mbTrackInView ? 1 : 0; c.track_scale_level = pMP->mnTrackScaleLevel;
            c.track_reference_for_frame = (uint64_t)pMP->mnTrackReferenceForFrame; c.last_frame_seen = (uint64_t)pMP->mnLastFrameSeen;
            c.ba_local_for_kf = (uint64_t)pMP->mnBALocalForKF; c.fuse_candidate_for_kf = (uint64_t)pMP->mnFuseCandidateForKF; c.loop_point_for_kf = (uint64_t)pMP->mnLoopPointForKF;
            c.corrected_by_kf = (uint64_t)pMP->mnCorrectedByKF; c.corrected_reference = (uint64_t)pMP->mnCorrectedReference;
        }
        if (!sc.empty()) check(corb_mp_store_set_scratch(store, first, (int)sc.size(), sc.data()), "corb_mp_store_set_scratch");
    }
    // the same fields back into the objects (the receiving side of a push: what boost's load leaves in a MapPoint)
    static void ReadBackScratch(CorbMpStore* store, int first, const std::vector<MapPoint*>& vpMP)
    {
        std::vector<CorbMapPointScratch> sc(vpMP.size());
        if (sc.empty()) return;
        check(corb_mp_store_get_scratch(store, first, (int)sc.size(), sc.data()), "corb_mp_store_get_scratch");
        for (size_t m = 0; m < vpMP.size(); m++) {
            MapPoint* pMP = vpMP[m]; const CorbMapPointScratch& c = sc[m];
            pMP->mnFirstKFid = (long int)c.first_kf_id; pMP->mnFirstFrame = (long int)c.first_frame; pMP->nObs = c.n_obs_weight;
            pMP->mTrackProjX = c.track_proj_x; pMP->mTrackProjY = c.track_proj_y; pMP->mTrackProjXR = c.track_proj_xr; pMP->mTrackViewCos = c.track_view_cos;
            pMP->mbTrackInView = c.track_in_view != 0; pMP->mnTrackScaleLevel = c.track_scale_level;
            pMP->mnTrackReferenceForFrame = (long unsigned int)c.track_reference_for_frame; pMP->mnLastFrameSeen = (long unsigned int)c.last_frame_seen;
            pMP->mnBALocalForKF = (long unsigned int)c.ba_local_for_kf; pMP->mnFuseCandidateForKF = (long unsigned int)c.fuse_candidate_for_kf; pMP->mnLoopPointForKF = (long unsigned int)c.loop_point_for_kf;
            pMP->mnCorrectedByKF = (long unsigned int)c.corrected_by_kf; pMP->mnCorrectedReference = (long unsigned int)c.corrected_reference;
        }
    }
    // Optimizer::GlobalBundleAdjustemnt(pCache, nIterations, pbStopFlag, nLoopKF, bRobust) on records (GlobalOptimize.cpp:444 on the server rank)
    // scaleFactor (ORBextractor's, KeyFrame::mfScaleFactor) > 0 and nLoopKF == 0: the write-back also does pMP->UpdateNormalAndDepth() on the records (Optimizer.cc:254-256)
    static CorbBAResult GlobalBundleAdjustemnt(CorbKfStore* kf, const std::vector<int32_t>& kfSlots, CorbMpStore* mp, const std::vector<int32_t>& mpSlots, int nIterations = 5,
                                               bool* pbStopFlag = nullptr, const unsigned long nLoopKF = 0, const bool bRobust = true, float scaleFactor = 0.f)
    {
        CorbBAResult r{};
        StopBridge stop(pbStopFlag);
        CorbBAOptions opt{}; opt.scale_factor = scaleFactor;
        check(corb_ba_solve_store(kf, kfSlots.data(), (int)kfSlots.size(), mp, mpSlots.data(), (int)mpSlots.size(), nIterations, bRobust ? 1 : 0, stop.ptr(pbStopFlag),
                                  (uint64_t)nLoopKF, &r, &opt), "corb_ba_solve_store");
        return r;
    }
    // void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Cache* pCache) (Optimizer.cc:487-838) on records: the caller selects the window as the
    // reference does (:493-546) and names it by slots -- lLocalKeyFrames, lFixedCameras, lLocalMapPoints; scaleFactor = pKF->mfScaleFactor.  The records receive the
    // poses / positions, vToErase (with SetBadFlag below three observations) and UpdateNormalAndDepth; vToErase (optional) lists the erased observations as
    // (index into localKfSlots ++ fixedKfSlots, index into mpSlots) for ReadBackLocalBA.
    static CorbBAResult LocalBundleAdjustment(CorbKfStore* kf, const std::vector<int32_t>& localKfSlots, const std::vector<int32_t>& fixedKfSlots, CorbMpStore* mp,
                                              const std::vector<int32_t>& mpSlots, float scaleFactor, bool* pbStopFlag = nullptr,
                                              std::vector<std::pair<int32_t, int32_t>>* vToErase = nullptr, bool applyErase = true)
    {
        CorbBAResult r{};
        StopBridge stop(pbStopFlag);
        std::vector<int32_t> slots(localKfSlots); slots.insert(slots.end(), fixedKfSlots.begin(), fixedKfSlots.end());
        const float hm = std::sqrt(5.991f), hs = std::sqrt(7.815f);
        const CorbBAStage st[2] = { {5, 1, 5.991f, 7.815f, 1, 0, 0, 0, 0, hm, hs}, {10, 0, 5.991f, 7.815f, 1, 0, 1, 0, 0, hm, hs} };      // optimize(5) / classify / optimize(10) / final test on every edge (:711-790)
        std::vector<int32_t> pairs(2 * std::max<size_t>(1, slots.size() * mpSlots.size())); int n = 0;
        check(corb_local_ba_store(kf, slots.data(), (int)localKfSlots.size(), (int)slots.size(), mp, mpSlots.data(), (int)mpSlots.size(), st, 2, scaleFactor, applyErase ? 1 : 0,
                                  stop.ptr(pbStopFlag), &r, pairs.data(), (int)(pairs.size() / 2), &n, nullptr), "corb_local_ba_store");
        if (vToErase) { vToErase->clear(); for (int k = 0; k < n; k++) vToErase->emplace_back(pairs[2 * k], pairs[2 * k + 1]); }
        return r;
    }
    // what LocalBundleAdjustment leaves in the records -> the objects (Optimizer.cc:796-836): the erased observations first, then the local keyframes' poses and the
    // local points' positions (+ UpdateNormalAndDepth on the object, as the reference calls it)
    static void ReadBackLocalBA(CorbKfStore* kf, const std::vector<int32_t>& localKfSlots, const std::vector<KeyFrame*>& vpLocalAndFixedKF, CorbMpStore* mp, int first,
                                const std::vector<MapPoint*>& vpMP, const std::vector<std::pair<int32_t, int32_t>>& vToErase)
    {
        for (const auto& e : vToErase) { KeyFrame* pKFi = vpLocalAndFixedKF[e.first]; MapPoint* pMPi = vpMP[e.second]; pKFi->EraseMapPointMatch(pMPi); pMPi->EraseObservation(pKFi); }
        for (size_t k = 0; k < localKfSlots.size(); k++) ReadBackKeyFrame(kf, localKfSlots[k], vpLocalAndFixedKF[k], 0);
        std::vector<CorbMapPointRecord> rec(vpMP.size());
        check(corb_mp_store_get(mp, first, (int)vpMP.size(), rec.data(), nullptr, nullptr), "corb_mp_store_get");
        for (size_t m = 0; m < vpMP.size(); m++) {
            MapPoint* pMP = vpMP[m];
            if (pMP->getFixed()) continue;                                            // if (!pMP->getFixed()) (:826)
            pMP->SetWorldPos(MatFactory<Mat>::from_floats(3, 1, rec[m].world_pos)); pMP->getCache()->addUpdateMapPoint(pMP); pMP->UpdateNormalAndDepth();
        }
    }
    // the record's estimate -> the object (Optimizer.cc:216-237): a record the solve did not write keeps what PutKeyFrame filed, so the copy is idempotent
    static void ReadBackKeyFrame(CorbKfStore* store, int slot, KeyFrame* pKF, const unsigned long nLoopKF)
    {
        CorbKeyFrameMeta m; check(corb_kf_store_get_meta(store, slot, &m), "corb_kf_store_get_meta");
        if (pKF->isBad() || pKF->getFixed()) return;
        if (nLoopKF == 0) { pKF->SetPose(MatFactory<Mat>::from_floats(4, 4, m.Tcw)); pKF->mpCacher->addUpdateKeyframe(pKF); }
        else if (m.ba_global_for_kf == (uint64_t)nLoopKF) { pKF->mTcwGBA = MatFactory<Mat>::from_floats(4, 4, m.TcwGBA); pKF->mnBAGlobalForKF = nLoopKF; }
    }
    // (:240-262) -- `optimised[m]`: the point had an edge in the solve (ba_global_for_kf == nLoopKF says so for nLoopKF != 0; for nLoopKF == 0 the caller
    // knows it from the observation lists it filed: a point without observations among the solve's keyframes is not touched)
    static void ReadBackMapPoints(CorbMpStore* store, int first, const std::vector<MapPoint*>& vpMP, const unsigned long nLoopKF, const std::vector<uint8_t>& optimised)
    {
        std::vector<CorbMapPointRecord> rec(vpMP.size());
        check(corb_mp_store_get(store, first, (int)vpMP.size(), rec.data(), nullptr, nullptr), "corb_mp_store_get");
        for (size_t m = 0; m < vpMP.size(); m++) {
            MapPoint* pMP = vpMP[m];
            if (pMP->isBad() || pMP->getFixed()) continue;
            if (nLoopKF == 0) { if (!optimised[m]) continue; pMP->SetWorldPos(MatFactory<Mat>::from_floats(3, 1, rec[m].world_pos)); pMP->getCache()->addUpdateMapPoint(pMP); pMP->UpdateNormalAndDepth(); }
            else if (rec[m].ba_global_for_kf == (uint64_t)nLoopKF) { pMP->mPosGBA = MatFactory<Mat>::from_floats(3, 1, rec[m].pos_gba); pMP->mnBAGlobalForKF = nLoopKF; }
        }
    }
};


// ---- the tracking thread on device-resident records (corb_track_*, include/corb_accel.h): a Frame lives in a slot of a keyframe store, the client's map in a
// map-point store; the calls below are Tracking::TrackWithMotionModel / TrackLocalMap's matcher and optimiser calls with the reference's argument meaning.
// MapPoint needs two accessors the reference does not have (the raw mfMinDistance / mfMaxDistance next to Get*DistanceInvariance()): GetMinDistance(), GetMaxDistance().
template <class Frame, class MapPoint, class Mat> struct FrameStoreT {
    static CorbTrackCamera Camera(const Frame& F)
    {
        CorbTrackCamera c; std::memset(&c, 0, sizeof(c));
        c.fx = F.fx; c.fy = F.fy; c.cx = F.cx; c.cy = F.cy; c.bf = F.mbf; c.mb = F.mb;
        c.min_x = Frame::mnMinX; c.max_x = Frame::mnMaxX; c.min_y = Frame::mnMinY; c.max_y = Frame::mnMaxY;
        c.nlevels = (int32_t)std::min<size_t>(F.mvScaleFactors.size(), 16);
        for (int l = 0; l < c.nlevels; l++) c.scale[l] = F.mvScaleFactors[l];
        return c;
    }
    // the Frame's features, pose, mvInvLevelSigma2, mvpMapPoints (as ids) and mvbOutlier -> record `slot`
    static void PutFrame(CorbKfStore* store, int slot, const Frame& F)
    {
        const int N = F.N;
        std::vector<CorbKeyPoint> kp(N); std::vector<uint8_t> desc((size_t)N * 32), fl(N, 0); std::vector<uint64_t> ids(N, CORB_NO_MAP_POINT);
        for (int i = 0; i < N; i++) {
            const auto& k = F.mvKeysUn[i];
            kp[i] = CorbKeyPoint{k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id};
            std::memcpy(&desc[(size_t)i * 32], F.mDescriptors.template ptr<uint8_t>(i), 32);
            MapPoint* pMP = F.mvpMapPoints[i].getMapPoint();
            if (pMP) ids[i] = (uint64_t)pMP->mnId;
            if (i < (int)F.mvbOutlier.size() && F.mvbOutlier[i]) fl[i] = 2;                    // record feature flag bit 1 = mvbOutlier
        }
        CorbKeyFrameMeta m; std::memset(&m, 0, sizeof(m));
        m.id = (uint64_t)F.mnId; m.fx = F.fx; m.fy = F.fy; m.cx = F.cx; m.cy = F.cy; m.bf = F.mbf;
        m.nlevels = (int32_t)std::min<size_t>(F.mvInvLevelSigma2.size(), 16);
        for (int l = 0; l < m.nlevels; l++) m.inv_level_sigma2[l] = F.mvInvLevelSigma2[l];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { m.Tcw[4 * r + c] = F.mTcw.empty() ? (r == c ? 1.f : 0.f) : matf(F.mTcw, r, c); m.TcwGBA[4 * r + c] = r == c ? 1.f : 0.f; }
        check(corb_kf_store_put_frame(store, slot, kp.data(), desc.data(), F.mvuRight.data(), nullptr, N, &m), "corb_kf_store_put_frame");
        check(corb_kf_store_set_map_points(store, slot, ids.data()), "corb_kf_store_set_map_points");
        check(corb_kf_store_set_flags(store, slot, fl.data()), "corb_kf_store_set_flags");
    }
    // the client's map: MapStoreT::PutMapPoints' fields plus what the tracking calls read (normal, distance range, descriptor), then the id index
    static void PutMap(CorbMpStore* store, const std::vector<MapPoint*>& vpMP)
    {
        std::vector<CorbMapPointRecord> rec(vpMP.size()); std::vector<int32_t> off(vpMP.size() + 1, 0); std::vector<uint64_t> okf; std::vector<uint32_t> oidx;
        for (size_t m = 0; m < vpMP.size(); m++) {
            MapPoint* pMP = vpMP[m]; CorbMapPointRecord& r = rec[m]; std::memset(&r, 0, sizeof(r));
            r.id = (uint64_t)pMP->mnId; r.flags = pMP->isBad() ? CORB_MP_BAD : 0u;
            const Mat X = pMP->GetWorldPos(), Nn = pMP->GetNormal(), D = pMP->GetDescriptor();
            for (int a = 0; a < 3; a++) { r.world_pos[a] = matf(X, a); r.normal[a] = Nn.empty() ? 0.f : matf(Nn, a); }
            r.min_distance = pMP->GetMinDistance(); r.max_distance = pMP->GetMaxDistance();
            if (!D.empty()) std::memcpy(r.descriptor, D.template ptr<uint8_t>(0), 32);
            const auto observations = pMP->GetObservations();
            std::vector<std::pair<uint64_t, uint32_t>> obs;
            for (auto it = observations.begin(); it != observations.end(); ++it) obs.emplace_back((uint64_t)it->first->mnId, (uint32_t)it->second);
            std::sort(obs.begin(), obs.end());
            for (auto& o : obs) { okf.push_back(o.first); oidx.push_back(o.second); }
            r.n_obs = (int32_t)obs.size(); off[m + 1] = (int32_t)okf.size();
        }
        check(corb_mp_store_put_host(store, 0, (int)vpMP.size(), rec.data(), off.data(), okf.data(), oidx.data()), "corb_mp_store_put_host");
        check(corb_mp_store_build_index(store, 0, (int)vpMP.size()), "corb_mp_store_build_index");
    }
    // int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono): CurrentFrame.mTcw = the predicted pose
    static int SearchByProjection(CorbKfStore* store, int curSlot, int lastSlot, CorbMpStore* map, const Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono,
                                  float nnratio = 0.9f, bool checkOri = true)
    {
        float Tcw[16], Tlw[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { Tcw[4 * r + c] = matf(CurrentFrame.mTcw, r, c); Tlw[4 * r + c] = matf(LastFrame.mTcw, r, c); }
        const CorbTrackCamera cam = Camera(CurrentFrame);
        int n = 0;
        check(corb_track_search_last_frame(store, curSlot, lastSlot, map, Tcw, Tlw, &cam, th, bMono ? 1 : 0, nnratio, checkOri ? 1 : 0, nullptr, &n), "corb_track_search_last_frame");
        return n;
    }
    // int Optimizer::PoseOptimization(Frame *pFrame) (+ the caller's "Discard outliers" loop when discardOutliers): pose and mvbOutlier land in the Frame as well
    static int PoseOptimization(CorbKfStore* store, int slot, CorbMpStore* map, Frame* pFrame, bool discardOutliers)
    {
        float Tin[16], Tout[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tin[4 * r + c] = matf(pFrame->mTcw, r, c);
        const CorbTrackCamera cam = Camera(*pFrame);
        std::vector<uint8_t> o((size_t)std::max(pFrame->N, 1)); int32_t inl = 0;
        check(corb_track_pose_optimization(store, slot, map, &cam, Tin, Tout, discardOutliers ? 1 : 0, o.data(), &inl), "corb_track_pose_optimization");
        pFrame->SetPose(MatFactory<Mat>::from_floats(4, 4, Tout));
        pFrame->mvbOutlier.assign(pFrame->N, false);
        typedef typename std::decay<decltype(pFrame->mvpMapPoints[0])>::type LMP;
        for (int i = 0; i < pFrame->N; i++) if (o[i]) { if (discardOutliers) pFrame->mvpMapPoints[i] = LMP{nullptr}; else pFrame->mvbOutlier[i] = true; }
        return inl;
    }
    // void Tracking::SearchLocalPoints(): returns the matches; *nToMatch = the points that passed isInFrustum
    static int SearchLocalPoints(CorbKfStore* store, int slot, CorbMpStore* map, const Frame& F, const std::vector<MapPoint*>& vpLocalMapPoints, const float th, float nnratio = 0.8f, int* nToMatch = nullptr)
    {
        std::vector<uint64_t> ids(vpLocalMapPoints.size());
        for (size_t i = 0; i < ids.size(); i++) ids[i] = (uint64_t)vpLocalMapPoints[i]->mnId;
        float Tcw[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw[4 * r + c] = matf(F.mTcw, r, c);
        const CorbTrackCamera cam = Camera(F);
        int n = 0, nv = 0;
        check(corb_track_search_local_points(store, slot, map, ids.data(), (int)ids.size(), &cam, Tcw, F.mfLogScaleFactor, th, nnratio, nullptr, nullptr, &n, &nv), "corb_track_search_local_points");
        if (nToMatch) *nToMatch = nv;
        return n;
    }
    // the record's mvpMapPoints -> the Frame (ids resolved through the caller's id -> MapPoint* map; features whose point was discarded hold none)
    template <class Lookup> static void ReadBackMapPoints(CorbKfStore* store, int slot, Frame* pFrame, const Lookup& byId)
    {
        const int N = pFrame->N;
        std::vector<uint64_t> ids((size_t)std::max(N, 1)); std::vector<uint8_t> fl((size_t)std::max(N, 1));
        check(corb_kf_store_get_map_points(store, slot, ids.data(), N), "corb_kf_store_get_map_points");
        int n = 0; uint64_t kid = 0; int32_t nn = 0;
        check(corb_kf_store_get(store, slot, nullptr, nullptr, nullptr, nullptr, fl.data(), N, &n, &kid, nullptr, nullptr, nullptr, &nn), "corb_kf_store_get");
        typedef typename std::decay<decltype(pFrame->mvpMapPoints[0])>::type LMP;
        for (int i = 0; i < N; i++) {
            pFrame->mvpMapPoints[i] = LMP{nullptr};
            if (ids[i] != CORB_NO_MAP_POINT && !(fl[i] & 4)) { auto it = byId.find((unsigned long)ids[i]); if (it != byId.end()) pFrame->mvpMapPoints[i] = LMP{it->second}; }
        }
    }
};
}  // namespace adapt
}  // namespace corb

// ---- on a tree with the reference's headers: the reference's own class names ----
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && __has_include("KeyFrame.h") && __has_include("Frame.h") && __has_include("MapPoint.h") && __has_include("Cache.h")
#include <opencv2/core/core.hpp>
#include "KeyFrame.h"
#include "Frame.h"
#include "MapPoint.h"
#include "Cache.h"
namespace corb { namespace adapt {
template <> struct MatFactory<cv::Mat> { static cv::Mat from_floats(int rows, int cols, const float* p) { return cv::Mat(rows, cols, CV_32F, const_cast<float*>(p)).clone(); } };
} }
namespace ORB_SLAM2 {
namespace accel {
using ORBmatcher = corb::adapt::ORBmatcherT<KeyFrame, Frame, MapPoint, cv::Mat>;
using Optimizer = corb::adapt::OptimizerT<KeyFrame, Frame, MapPoint, Cache, cv::Mat>;
using MapStore = corb::adapt::MapStoreT<KeyFrame, MapPoint, cv::Mat>;
using FrameStore = corb::adapt::FrameStoreT<Frame, MapPoint, cv::Mat>;
}  // namespace accel
}  // namespace ORB_SLAM2
#endif
#endif
