// mock_orbslam.hpp -- TEST DOUBLES (test infrastructure, not product code) carrying the member names through which
// corb-slam_amd/host/corb_adapter_orbslam.hpp touches the reference's classes (corbslam_client/include/{KeyFrame,Frame,MapPoint,Cache}.h, cv::Mat,
// cv::KeyPoint, DBoW2::FeatureVector).  They hold plain data and count the cache notifications; they are NOT stand-ins for building the reference.
#pragma once
#include <cstdint>
#include <map>
#include <vector>

namespace mock {
struct Mat {
    int rows = 0, cols = 0; std::vector<float> f; std::vector<uint8_t> b;            // float matrix or byte matrix (descriptors)
    template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T&>(f[(size_t)r * cols + c]); }
    template <class T> const T& at(int i) const { return reinterpret_cast<const T&>(f[(size_t)i]); }
    template <class T> const T* ptr(int r) const { return reinterpret_cast<const T*>(b.data() + (size_t)r * cols); }
    bool empty() const { return f.empty() && b.empty(); }
};
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;           // DBoW2::FeatureVector
struct KeyFrame; struct MapPoint;
struct Cache {
    std::vector<KeyFrame*> kfs; std::vector<MapPoint*> mps; int nUpdKF = 0, nUpdMP = 0;
    std::vector<KeyFrame*> getAllKeyFramesInMap() { return kfs; }
    std::vector<MapPoint*> GetAllMapPointsFromMap() { return mps; }
    void addUpdateKeyframe(KeyFrame*) { nUpdKF++; }
    void addUpdateMapPoint(MapPoint*) { nUpdMP++; }
};
struct MapPoint {
    unsigned long mnId = 0; bool bad = false, fixed = false; Mat pos; std::map<KeyFrame*, size_t> obs; Cache* cache = nullptr;
    Mat mPosGBA; unsigned long mnBAGlobalForKF = 0; int nNormalUpdates = 0;
    Mat normal, descriptor; float minDistance = 0, maxDistance = 0;      // mNormalVector, mDescriptor, mfMinDistance, mfMaxDistance
    Mat GetNormal() { return normal; }
    Mat GetDescriptor() { return descriptor; }
    float GetMinDistance() { return minDistance; }          // (the two raw distances: accessors the adapter asks MapPoint.h for)
    float GetMaxDistance() { return maxDistance; }
    int Observations() { return (int)obs.size(); }
    bool isBad() { return bad; }
    bool getFixed() { return fixed; }
    Mat GetWorldPos() { return pos; }
    void SetWorldPos(const Mat& m) { pos = m; }
    std::map<KeyFrame*, size_t> GetObservations() { return obs; }
    Cache* getCache() { return cache; }
    void UpdateNormalAndDepth() { nNormalUpdates++; }
    KeyFrame* refKF = nullptr; KeyFrame* GetReferenceKeyFrame() { return refKF; }
    int nObs = 0;                                          // MapPoint::nObs: 2 per stereo observation, 1 per monocular one (MapPoint.cc:170-190)
    long int mnFirstKFid = 0, mnFirstFrame = 0;             // MapPoint.h:151-174: the public scratch of the tracking / local-mapping / loop-closing threads
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0; bool mbTrackInView = false; int mnTrackScaleLevel = 0;
    long unsigned int mnTrackReferenceForFrame = 0, mnLastFrameSeen = 0, mnBALocalForKF = 0, mnFuseCandidateForKF = 0, mnLoopPointForKF = 0, mnCorrectedByKF = 0, mnCorrectedReference = 0;
    inline void EraseObservation(KeyFrame* pKF);            // MapPoint.cc:192-217 (+ SetBadFlag, :255-269); defined below KeyFrame
};
struct LightMapPoint { MapPoint* p = nullptr; MapPoint* getMapPoint() const { return p; } };
struct KeyFrame {
    unsigned long mnId = 0; int N = 0; Mat mDescriptors; std::vector<KeyPoint> mvKeys, mvKeysUn; std::vector<float> mvuRight, mvDepth;
    std::vector<MapPoint*> mps; FeatureVector mFeatVec; std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0; Mat Tcw, Ow; bool bad = false, fixed = false; Cache* mpCacher = nullptr;
    Mat mTcwGBA; unsigned long mnBAGlobalForKF = 0;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0; float mfLogScaleFactor = 0;       // KeyFrame.h:271-280
    std::vector<MapPoint*> GetMapPointMatches() { return mps; }
    MapPoint* GetMapPoint(size_t i) { return mps[i]; }
    bool isBad() { return bad; }
    bool getFixed() { return fixed; }
    Mat GetPose() { return Tcw; }
    void SetPose(const Mat& m) { Tcw = m; }
    Mat GetCameraCenter() { return Ow; }
    Mat GetRotation() { Mat R; R.rows = R.cols = 3; R.f.resize(9); for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.f[3 * r + c] = Tcw.f[4 * r + c]; return R; }
    Mat GetTranslation() { Mat t; t.rows = 3; t.cols = 1; t.f = { Tcw.f[3], Tcw.f[7], Tcw.f[11] }; return t; }
    void EraseMapPointMatch(MapPoint* pMP) { for (auto& q : mps) if (q == pMP) q = nullptr; }      // KeyFrame.cc: idx = pMP->GetIndexInKeyFrame(this); mvpMapPoints[idx] = NULL
};
inline void MapPoint::EraseObservation(KeyFrame* pKF)
{
    auto it = obs.find(pKF);
    if (it == obs.end()) return;
    nObs -= pKF->mvuRight[it->second] >= 0 ? 2 : 1;
    obs.erase(it);
    if (nObs <= 2) { bad = true; for (auto& o : obs) o.first->mps[o.second] = nullptr; obs.clear(); }
}
struct Frame {
    int N = 0; Mat mDescriptors; std::vector<KeyPoint> mvKeys, mvKeysUn; std::vector<float> mvuRight; FeatureVector mFeatVec;
    std::vector<LightMapPoint> mvpMapPoints; std::vector<bool> mvbOutlier; std::vector<float> mvInvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0; Mat mTcw;
    long unsigned int mnId = 0; float mb = 0, mfLogScaleFactor = 0; int mnScaleLevels = 0; std::vector<float> mvScaleFactors;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    void SetPose(const Mat& m) { mTcw = m; }
};
inline float Frame::mnMinX = 0, Frame::mnMaxX = 0, Frame::mnMinY = 0, Frame::mnMaxY = 0;
}  // namespace mock
