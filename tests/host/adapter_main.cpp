// adapter_main.cpp -- TEST DRIVER (test infrastructure): runs the signature-preserving adapters of
// corb-slam_amd/host/corb_adapter_orbslam.hpp -- ORBmatcher::SearchByBoW x3 / SearchForTriangulation, Optimizer::GlobalBundleAdjustemnt /
// PoseOptimization, SearchByProjection(KeyFrame*, Scw, ...), SearchForInitialization, and the store adapter (MapStoreT: objects -> device records -> global BA on the records -> objects) -- on the test doubles of tests/host/mock_orbslam.hpp, from a scene file written by tests/test_gpu_host.py, and dumps what
// the reference's callers would observe (MapPoint* matches as feature indices, poses / points after the nLoopKF write-back, mvbOutlier).
// Usage: adapter_main <scene.bin> <out.bin>.   Records are [u32 bytes][payload], read / written in a fixed order.
#include "corb_adapter_orbslam.hpp"
#include "mock_orbslam.hpp"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <map>

namespace corb { namespace adapt {
template <> struct MatFactory<mock::Mat> { static mock::Mat from_floats(int rows, int cols, const float* p) { mock::Mat m; m.rows = rows; m.cols = cols; m.f.assign(p, p + (size_t)rows * cols); return m; } };
} }
using Matcher = corb::adapt::ORBmatcherT<mock::KeyFrame, mock::Frame, mock::MapPoint, mock::Mat>;
using Optim = corb::adapt::OptimizerT<mock::KeyFrame, mock::Frame, mock::MapPoint, mock::Cache, mock::Mat>;

struct Reader {
    std::ifstream f; explicit Reader(const char* p) : f(p, std::ios::binary) {}
    template <class T> std::vector<T> arr() { uint32_t nb = 0; f.read((char*)&nb, 4); std::vector<T> v(nb / sizeof(T)); if (nb) f.read((char*)v.data(), nb); return v; }
    template <class T> T one() { return arr<T>().at(0); }
};
struct Writer {
    std::ofstream f; explicit Writer(const char* p) : f(p, std::ios::binary) {}
    template <class T> void arr(const std::vector<T>& v) { uint32_t nb = (uint32_t)(v.size() * sizeof(T)); f.write((const char*)&nb, 4); if (nb) f.write((const char*)v.data(), nb); }
};
static mock::Mat desc_mat(const std::vector<uint8_t>& d) { mock::Mat m; m.rows = (int)(d.size() / 32); m.cols = 32; m.b = d; return m; }
static mock::Mat fmat(int r, int c, const float* p) { mock::Mat m; m.rows = r; m.cols = c; m.f.assign(p, p + (size_t)r * c); return m; }
static std::vector<mock::KeyPoint> keys(const std::vector<CorbKeyPoint>& k)
{ std::vector<mock::KeyPoint> o(k.size()); for (size_t i = 0; i < k.size(); i++) o[i] = mock::KeyPoint{{k[i].x, k[i].y}, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id}; return o; }
static mock::FeatureVector featvec(const std::vector<uint32_t>& node, const std::vector<int32_t>& off, const std::vector<uint32_t>& idx)
{ mock::FeatureVector fv; for (size_t n = 0; n < node.size(); n++) fv[node[n]] = std::vector<unsigned int>(idx.begin() + off[n], idx.begin() + off[n + 1]); return fv; }

// a keyframe whose feature i holds: flag 0 -> no MapPoint, 1 -> a good one, 2 -> a bad one
static void fill_kf(Reader& in, mock::KeyFrame& K, std::vector<std::unique_ptr<mock::MapPoint>>& pool)
{
    K.mDescriptors = desc_mat(in.arr<uint8_t>()); K.N = K.mDescriptors.rows;
    K.mvKeysUn = keys(in.arr<CorbKeyPoint>()); K.mvKeys = K.mvKeysUn;
    K.mvuRight = in.arr<float>();
    const std::vector<uint8_t> flag = in.arr<uint8_t>();
    const std::vector<uint32_t> node = in.arr<uint32_t>(); const std::vector<int32_t> off = in.arr<int32_t>(); const std::vector<uint32_t> idx = in.arr<uint32_t>();
    K.mFeatVec = featvec(node, off, idx);
    K.mps.assign(K.N, nullptr);
    for (int i = 0; i < K.N; i++) if (flag[i]) { pool.emplace_back(new mock::MapPoint()); pool.back()->bad = flag[i] == 2; pool.back()->mnId = pool.size(); K.mps[i] = pool.back().get(); }
}
static std::vector<int32_t> to_index(const std::vector<mock::MapPoint*>& m, const mock::KeyFrame& owner)
{
    std::map<mock::MapPoint*, int> where; for (int i = 0; i < owner.N; i++) if (owner.mps[i]) where[owner.mps[i]] = i;
    std::vector<int32_t> o(m.size(), -1); for (size_t i = 0; i < m.size(); i++) if (m[i]) o[i] = where.at(m[i]);
    return o;
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 2; }
    Reader in(argv[1]); Writer out(argv[2]);
    try {
        std::vector<std::unique_ptr<mock::MapPoint>> pool;
        // ---- A. the three BoW matchers ----
        {
            mock::KeyFrame K1, K2; fill_kf(in, K1, pool); fill_kf(in, K2, pool);
            const float ratio = in.one<float>(); const int ori = in.one<int32_t>();
            mock::Frame F; F.N = K2.N; F.mDescriptors = K2.mDescriptors; F.mvKeys = K2.mvKeys; F.mvKeysUn = K2.mvKeysUn; F.mvuRight = K2.mvuRight; F.mFeatVec = K2.mFeatVec;
            Matcher m(ratio, ori != 0);
            std::vector<mock::MapPoint*> a, b, c;
            const int na = m.SearchByBoW(&K1, F, a), nb = m.SearchByBoW(&K1, &K2, b), nc = m.SearchByBoWInServer(&K1, &K2, c);
            out.arr(to_index(a, K1)); out.arr(to_index(b, K2)); out.arr(to_index(c, K1)); out.arr(std::vector<int32_t>{na, nb, nc});
        }
        // ---- B. SearchForTriangulation ----
        {
            mock::KeyFrame K1, K2; fill_kf(in, K1, pool); fill_kf(in, K2, pool);
            const std::vector<float> T2 = in.arr<float>(), Ow1 = in.arr<float>(), F12 = in.arr<float>(), cam = in.arr<float>();
            K2.Tcw = fmat(4, 4, T2.data()); K1.Ow = fmat(3, 1, Ow1.data());
            K2.fx = cam[0]; K2.fy = cam[1]; K2.cx = cam[2]; K2.cy = cam[3];
            K2.mvScaleFactors = in.arr<float>(); K2.mvLevelSigma2 = in.arr<float>();
            const int only_stereo = in.one<int32_t>();
            Matcher m(0.6f, true);
            std::vector<std::pair<size_t, size_t>> pairs;
            const int n = m.SearchForTriangulation(&K1, &K2, fmat(3, 3, F12.data()), pairs, only_stereo != 0);
            std::vector<int32_t> flat; for (auto& pr : pairs) { flat.push_back((int32_t)pr.first); flat.push_back((int32_t)pr.second); }
            out.arr(flat); out.arr(std::vector<int32_t>{n});
        }
        // ---- C. Optimizer::GlobalBundleAdjustemnt(Cache*, 10, NULL, nLoopKF, false): nLoopKF = 0, then nLoopKF = 7 on a fresh copy ----
        {
            const std::vector<float> poses = in.arr<float>(), intr = in.arr<float>(), pts = in.arr<float>();
            const std::vector<uint8_t> kf_fixed = in.arr<uint8_t>(), kf_bad = in.arr<uint8_t>(), mp_fixed = in.arr<uint8_t>(), mp_bad = in.arr<uint8_t>();
            const std::vector<CorbBAEdge> edges = in.arr<CorbBAEdge>();
            const std::vector<int32_t> octave = in.arr<int32_t>();
            const int K = (int)(poses.size() / 16), M = (int)(pts.size() / 3);
            for (int pass = 0; pass < 2; pass++) {
                mock::Cache cache; std::vector<std::unique_ptr<mock::KeyFrame>> kfs; std::vector<std::unique_ptr<mock::MapPoint>> mps;
                for (int k = 0; k < K; k++) {
                    kfs.emplace_back(new mock::KeyFrame()); mock::KeyFrame& kf = *kfs.back();
                    kf.mnId = (unsigned long)k + 1; kf.Tcw = fmat(4, 4, &poses[16 * (size_t)k]); kf.fixed = kf_fixed[k] != 0; kf.bad = kf_bad[k] != 0; kf.mpCacher = &cache;
                    kf.fx = intr[5 * k]; kf.fy = intr[5 * k + 1]; kf.cx = intr[5 * k + 2]; kf.cy = intr[5 * k + 3]; kf.mbf = intr[5 * k + 4];
                    kf.mvInvLevelSigma2.resize(8); for (int l = 0; l < 8; l++) kf.mvInvLevelSigma2[l] = 1.0f / (float)std::pow(1.44, l);
                }
                for (int m = 0; m < M; m++) { mps.emplace_back(new mock::MapPoint()); mock::MapPoint& mp = *mps.back(); mp.mnId = (unsigned long)m; mp.pos = fmat(3, 1, &pts[3 * (size_t)m]); mp.fixed = mp_fixed[m] != 0; mp.bad = mp_bad[m] != 0; mp.cache = &cache; }
                for (size_t e = 0; e < edges.size(); e++) {      // observation e: feature slot = a new keypoint of its keyframe
                    mock::KeyFrame& kf = *kfs[edges[e].pose];
                    kf.mvKeysUn.push_back(mock::KeyPoint{{edges[e].u, edges[e].v}, 31.f, 0.f, 0.f, octave[e], -1}); kf.mvuRight.push_back(edges[e].u_right);
                    kf.mvInvLevelSigma2[octave[e]] = edges[e].inv_sigma2;
                    mps[edges[e].point]->obs[&kf] = kf.mvKeysUn.size() - 1;
                }
                // the cache hands the keyframes over in REVERSE order and with a NULL map point in between: the adapter sorts by mnId / skips NULLs
                for (int k = K - 1; k >= 0; k--) cache.kfs.push_back(kfs[k].get());
                for (int m = 0; m < M; m++) { cache.mps.push_back(mps[m].get()); if (m == 3) cache.mps.push_back(nullptr); }
                bool stop = false;
                Optim::GlobalBundleAdjustemnt(&cache, 10, &stop, pass == 0 ? 0ul : 7ul, false);
                std::vector<float> Tout, Xout; std::vector<int32_t> marks;
                for (int k = 0; k < K; k++) { const mock::Mat& T = pass == 0 ? kfs[k]->Tcw : kfs[k]->mTcwGBA; if (T.f.size() == 16) Tout.insert(Tout.end(), T.f.begin(), T.f.end()); else Tout.insert(Tout.end(), 16, -777.f);
                                              marks.push_back((int32_t)kfs[k]->mnBAGlobalForKF); }
                for (int m = 0; m < M; m++) { const mock::Mat& X = pass == 0 ? mps[m]->pos : mps[m]->mPosGBA; if (X.f.size() == 3) Xout.insert(Xout.end(), X.f.begin(), X.f.end()); else Xout.insert(Xout.end(), 3, -777.f);
                                              marks.push_back((int32_t)mps[m]->mnBAGlobalForKF + 1000 * mps[m]->nNormalUpdates); }
                out.arr(Tout); out.arr(Xout); out.arr(marks); out.arr(std::vector<int32_t>{cache.nUpdKF, cache.nUpdMP});
            }
        }
        // ---- D. Optimizer::PoseOptimization(Frame*) ----
        {
            const std::vector<float> Tcw = in.arr<float>(), P = in.arr<float>(), obs = in.arr<float>(), w = in.arr<float>(), cam = in.arr<float>();
            const std::vector<uint8_t> has = in.arr<uint8_t>();
            mock::Frame F; F.N = (int)has.size(); F.mTcw = fmat(4, 4, Tcw.data()); F.fx = cam[0]; F.fy = cam[1]; F.cx = cam[2]; F.cy = cam[3]; F.mbf = cam[4];
            F.mvKeysUn.resize(F.N); F.mvKeys.resize(F.N); F.mvuRight.resize(F.N); F.mvpMapPoints.resize(F.N); F.mvbOutlier.assign(F.N, true);
            F.mvInvLevelSigma2.resize(F.N);                      // (one "level" per feature: its weight)
            std::vector<std::unique_ptr<mock::MapPoint>> mps;
            for (int i = 0; i < F.N; i++) {
                F.mvKeysUn[i] = mock::KeyPoint{{obs[3 * i], obs[3 * i + 1]}, 31.f, 0.f, 0.f, i, -1}; F.mvuRight[i] = obs[3 * i + 2]; F.mvInvLevelSigma2[i] = w[i];
                if (has[i]) { mps.emplace_back(new mock::MapPoint()); mps.back()->pos = fmat(3, 1, &P[3 * (size_t)i]); F.mvpMapPoints[i].p = mps.back().get(); }
            }
            const int n = Optim::PoseOptimization(&F);
            std::vector<uint8_t> o(F.N); for (int i = 0; i < F.N; i++) o[i] = F.mvbOutlier[i] ? 1 : 0;
            out.arr(F.mTcw.f); out.arr(o); out.arr(std::vector<int32_t>{n});
        }
        // ---- E. the same map through the device-resident stores: MapStoreT::PutKeyFrame / PutMapPoints -> GlobalBundleAdjustemnt on the records -> ReadBack* ----
        {
            const std::vector<float> poses = in.arr<float>(), intr = in.arr<float>(), pts = in.arr<float>();
            const std::vector<uint8_t> kf_fixed = in.arr<uint8_t>(), kf_bad = in.arr<uint8_t>(), mp_fixed = in.arr<uint8_t>(), mp_bad = in.arr<uint8_t>();
            const std::vector<CorbBAEdge> edges = in.arr<CorbBAEdge>();
            const std::vector<int32_t> octave = in.arr<int32_t>();
            const int K = (int)(poses.size() / 16), M = (int)(pts.size() / 3);
            using Store = corb::adapt::MapStoreT<mock::KeyFrame, mock::MapPoint, mock::Mat>;
            for (int pass = 0; pass < 2; pass++) {
                mock::Cache cache; std::vector<std::unique_ptr<mock::KeyFrame>> kfs; std::vector<std::unique_ptr<mock::MapPoint>> mps;
                for (int k = 0; k < K; k++) {
                    kfs.emplace_back(new mock::KeyFrame()); mock::KeyFrame& kf = *kfs.back();
                    kf.mnId = (unsigned long)k + 1; kf.Tcw = fmat(4, 4, &poses[16 * (size_t)k]); kf.fixed = kf_fixed[k] != 0; kf.bad = kf_bad[k] != 0; kf.mpCacher = &cache;
                    kf.fx = intr[5 * k]; kf.fy = intr[5 * k + 1]; kf.cx = intr[5 * k + 2]; kf.cy = intr[5 * k + 3]; kf.mbf = intr[5 * k + 4];
                    kf.mvInvLevelSigma2.resize(8); for (int l = 0; l < 8; l++) kf.mvInvLevelSigma2[l] = 1.0f / (float)std::pow(1.44, l);
                }
                for (int m = 0; m < M; m++) { mps.emplace_back(new mock::MapPoint()); mock::MapPoint& mp = *mps.back(); mp.mnId = (unsigned long)m + 1000; mp.pos = fmat(3, 1, &pts[3 * (size_t)m]); mp.fixed = mp_fixed[m] != 0; mp.bad = mp_bad[m] != 0; mp.cache = &cache; }
                std::vector<uint8_t> has_edge(M, 0);
                for (size_t e = 0; e < edges.size(); e++) {
                    mock::KeyFrame& kf = *kfs[edges[e].pose];
                    kf.mvKeysUn.push_back(mock::KeyPoint{{edges[e].u, edges[e].v}, 31.f, 0.f, 0.f, octave[e], -1}); kf.mvuRight.push_back(edges[e].u_right);
                    kf.mvInvLevelSigma2[octave[e]] = edges[e].inv_sigma2;
                    kf.mps.push_back(mps[edges[e].point].get());
                    mps[edges[e].point]->obs[&kf] = kf.mvKeysUn.size() - 1;
                    if (!kf.bad) has_edge[edges[e].point] = 1;
                }
                int F = 1; for (auto& k : kfs) { k->N = (int)k->mvKeysUn.size(); k->mvKeys = k->mvKeysUn; F = std::max(F, k->N); }
                CorbKfStore* KS = nullptr; CorbMpStore* MS = nullptr;
                corb::check(corb_kf_store_create(0, K, F, &KS), "corb_kf_store_create"); corb::check(corb_mp_store_create(0, M, 16, &MS), "corb_mp_store_create");
                std::vector<int32_t> ks(K), ms(M); std::vector<mock::MapPoint*> vmp(M);
                for (int k = 0; k < K; k++) { Store::PutKeyFrame(KS, k, kfs[k].get(), 1 + k / 10); ks[k] = k; }
                for (int m = 0; m < M; m++) { vmp[m] = mps[m].get(); ms[m] = m; }
                // the public scratch fields of MapPoint's archive ride with the records (CorbMapPointScratch) and come back through ReadBackScratch
                for (int m = 0; m < M; m++) { mock::MapPoint& q = *mps[m]; q.mnFirstKFid = 3 + m; q.mnFirstFrame = -m; q.mTrackProjX = 0.5f * m; q.mTrackProjXR = -1.25f * m; q.mTrackViewCos = 0.001f * m;
                                              q.mbTrackInView = (m & 1) != 0; q.mnTrackScaleLevel = m % 8; q.mnLastFrameSeen = 100ul + m; q.mnBALocalForKF = 7ul * m; q.mnCorrectedReference = (1ul << 40) + m; }
                Store::PutMapPoints(MS, 0, vmp, 1);
                for (int m = 0; m < M; m++) { mock::MapPoint& q = *mps[m]; q.mnFirstKFid = q.mnFirstFrame = 0; q.mTrackProjX = q.mTrackProjXR = q.mTrackViewCos = 0; q.mbTrackInView = false; q.mnTrackScaleLevel = 0;
                                              q.mnLastFrameSeen = q.mnBALocalForKF = q.mnCorrectedReference = 0; }
                bool stop = false; const unsigned long loop = pass == 0 ? 0ul : 7ul;
                const CorbBAResult r = Store::GlobalBundleAdjustemnt(KS, ks, MS, ms, 10, &stop, loop, false);
                Store::ReadBackScratch(MS, 0, vmp);
                for (int m = 0; m < M; m++) {
                    const mock::MapPoint& q = *mps[m];
                    if (q.mnFirstKFid != 3 + m || q.mnFirstFrame != -m || q.mTrackProjX != 0.5f * m || q.mTrackProjXR != -1.25f * m || q.mTrackViewCos != 0.001f * m || q.mbTrackInView != ((m & 1) != 0) ||
                        q.mnTrackScaleLevel != m % 8 || q.mnLastFrameSeen != 100ul + m || q.mnBALocalForKF != 7ul * m || q.mnCorrectedReference != (1ul << 40) + m || q.nObs != (int)0 + q.nObs) {
                        std::fprintf(stderr, "CorbMapPointScratch did not round-trip for map point %d\n", m); return 4;
                    }
                }
                for (int k = 0; k < K; k++) Store::ReadBackKeyFrame(KS, k, kfs[k].get(), loop);
                Store::ReadBackMapPoints(MS, 0, vmp, loop, has_edge);
                std::vector<float> Tout, Xout; std::vector<int32_t> marks;
                for (int k = 0; k < K; k++) { const mock::Mat& T = pass == 0 ? kfs[k]->Tcw : kfs[k]->mTcwGBA; if (T.f.size() == 16) Tout.insert(Tout.end(), T.f.begin(), T.f.end()); else Tout.insert(Tout.end(), 16, -777.f);
                                              marks.push_back((int32_t)kfs[k]->mnBAGlobalForKF); }
                for (int m = 0; m < M; m++) { const mock::Mat& X = pass == 0 ? mps[m]->pos : mps[m]->mPosGBA; if (X.f.size() == 3) Xout.insert(Xout.end(), X.f.begin(), X.f.end()); else Xout.insert(Xout.end(), 3, -777.f);
                                              marks.push_back((int32_t)mps[m]->mnBAGlobalForKF + 1000 * mps[m]->nNormalUpdates); }
                out.arr(Tout); out.arr(Xout); out.arr(marks); out.arr(std::vector<int32_t>{cache.nUpdKF, cache.nUpdMP, r.iters_done, r.active_edges});
                corb_kf_store_destroy(KS); corb_mp_store_destroy(MS);
            }
        }
        // ---- F. the tracking thread on records: FrameStoreT::PutFrame / PutMap -> SearchByProjection(Cur, Last) -> PoseOptimization + discard -> SearchLocalPoints -> PoseOptimization ----
        {
            using FS = corb::adapt::FrameStoreT<mock::Frame, mock::MapPoint, mock::Mat>;
            const std::vector<float> camv = in.arr<float>(), scale = in.arr<float>();          // fx fy cx cy bf mb minx maxx miny maxy logs
            const std::vector<int64_t> mpid = in.arr<int64_t>();
            const std::vector<float> X = in.arr<float>(), Nn = in.arr<float>(), dmin = in.arr<float>(), dmax = in.arr<float>();
            const std::vector<uint8_t> mdesc = in.arr<uint8_t>(), mbad = in.arr<uint8_t>(); const std::vector<int32_t> mobs = in.arr<int32_t>();
            const int M = (int)mpid.size();
            mock::KeyFrame someKF; someKF.mnId = 1;
            std::vector<std::unique_ptr<mock::MapPoint>> mps; std::vector<mock::MapPoint*> vmp(M); std::map<unsigned long, mock::MapPoint*> byId;
            for (int m = 0; m < M; m++) {
                mps.emplace_back(new mock::MapPoint()); mock::MapPoint& p = *mps.back(); vmp[m] = &p;
                p.mnId = (unsigned long)mpid[m]; p.pos = fmat(3, 1, &X[3 * (size_t)m]); p.normal = fmat(3, 1, &Nn[3 * (size_t)m]); p.minDistance = dmin[m]; p.maxDistance = dmax[m];
                p.descriptor = desc_mat(std::vector<uint8_t>(mdesc.begin() + 32 * (size_t)m, mdesc.begin() + 32 * (size_t)m + 32)); p.bad = mbad[m] != 0;
                if (mobs[m] > 0) p.obs[&someKF] = 0;
                byId[p.mnId] = &p;
            }
            mock::Frame::mnMinX = camv[6]; mock::Frame::mnMaxX = camv[7]; mock::Frame::mnMinY = camv[8]; mock::Frame::mnMaxY = camv[9];
            auto read_frame = [&](mock::Frame& F, unsigned long id) {
                F.mDescriptors = desc_mat(in.arr<uint8_t>()); F.N = F.mDescriptors.rows; F.mvKeysUn = keys(in.arr<CorbKeyPoint>()); F.mvKeys = F.mvKeysUn; F.mvuRight = in.arr<float>();
                const std::vector<int32_t> held = in.arr<int32_t>(); const std::vector<uint8_t> outl = in.arr<uint8_t>(); const std::vector<float> T = in.arr<float>();
                F.mvpMapPoints.assign(F.N, mock::LightMapPoint{}); F.mvbOutlier.assign(F.N, false);
                for (int i = 0; i < F.N; i++) { if (held[i] >= 0) F.mvpMapPoints[i].p = vmp[held[i]]; F.mvbOutlier[i] = outl[i] != 0; }
                F.mTcw = fmat(4, 4, T.data()); F.mnId = id;
                F.fx = camv[0]; F.fy = camv[1]; F.cx = camv[2]; F.cy = camv[3]; F.mbf = camv[4]; F.mb = camv[5]; F.mfLogScaleFactor = camv[10];
                F.mvScaleFactors = scale; F.mnScaleLevels = (int)scale.size(); F.mvInvLevelSigma2.resize(scale.size());
                for (size_t l = 0; l < scale.size(); l++) F.mvInvLevelSigma2[l] = 1.0f / (scale[l] * scale[l]);
            };
            mock::Frame Last, Cur; read_frame(Last, 1); read_frame(Cur, 2);
            const std::vector<int32_t> local = in.arr<int32_t>();
            CorbKfStore* KS = nullptr; CorbMpStore* MS = nullptr;
            corb::check(corb_kf_store_create(0, 2, 2048, &KS), "corb_kf_store_create"); corb::check(corb_mp_store_create(0, M, 2, &MS), "corb_mp_store_create");
            FS::PutMap(MS, vmp); FS::PutFrame(KS, 0, Last); FS::PutFrame(KS, 1, Cur);
            const int n1 = FS::SearchByProjection(KS, 1, 0, MS, Cur, Last, 7.0f, false);
            const int i1 = FS::PoseOptimization(KS, 1, MS, &Cur, true);
            std::vector<mock::MapPoint*> vlocal; for (int32_t l : local) vlocal.push_back(vmp[l]);
            int nToMatch = 0;
            const int n2 = FS::SearchLocalPoints(KS, 1, MS, Cur, vlocal, 1.0f, 0.8f, &nToMatch);
            const int i2 = FS::PoseOptimization(KS, 1, MS, &Cur, false);
            FS::ReadBackMapPoints(KS, 1, &Cur, byId);
            std::vector<int64_t> held(Cur.N, -1); std::vector<uint8_t> o(Cur.N);
            for (int i = 0; i < Cur.N; i++) { if (Cur.mvpMapPoints[i].getMapPoint()) held[i] = (int64_t)Cur.mvpMapPoints[i].getMapPoint()->mnId; o[i] = Cur.mvbOutlier[i] ? 1 : 0; }
            out.arr(std::vector<int32_t>{n1, i1, n2, i2, nToMatch}); out.arr(held); out.arr(o); out.arr(Cur.mTcw.f);
            corb_kf_store_destroy(KS); corb_mp_store_destroy(MS);
        }
        // ---- G. Optimizer::LocalBundleAdjustment on records: MapStoreT::LocalBundleAdjustment on the map of E (the first n_local keyframes local, the rest fixed cameras) -> ReadBackLocalBA ----
        {
            const std::vector<float> poses = in.arr<float>(), intr = in.arr<float>(), pts = in.arr<float>();
            const std::vector<uint8_t> kf_fixed = in.arr<uint8_t>(), kf_bad = in.arr<uint8_t>(), mp_fixed = in.arr<uint8_t>(), mp_bad = in.arr<uint8_t>();
            const std::vector<CorbBAEdge> edges = in.arr<CorbBAEdge>();
            const std::vector<int32_t> octave = in.arr<int32_t>(), nloc = in.arr<int32_t>();
            const int K = (int)(poses.size() / 16), M = (int)(pts.size() / 3), n_local = nloc[0];
            using Store = corb::adapt::MapStoreT<mock::KeyFrame, mock::MapPoint, mock::Mat>;
            mock::Cache cache; std::vector<std::unique_ptr<mock::KeyFrame>> kfs; std::vector<std::unique_ptr<mock::MapPoint>> mps;
            for (int k = 0; k < K; k++) {
                kfs.emplace_back(new mock::KeyFrame()); mock::KeyFrame& kf = *kfs.back();
                kf.mnId = (unsigned long)k + 1; kf.Tcw = fmat(4, 4, &poses[16 * (size_t)k]); kf.fixed = kf_fixed[k] != 0; kf.bad = kf_bad[k] != 0; kf.mpCacher = &cache;
                kf.fx = intr[5 * k]; kf.fy = intr[5 * k + 1]; kf.cx = intr[5 * k + 2]; kf.cy = intr[5 * k + 3]; kf.mbf = intr[5 * k + 4];
                kf.mvInvLevelSigma2.resize(8); for (int l = 0; l < 8; l++) kf.mvInvLevelSigma2[l] = 1.0f / (float)std::pow(1.44, l);
            }
            for (int m = 0; m < M; m++) { mps.emplace_back(new mock::MapPoint()); mock::MapPoint& mp = *mps.back(); mp.mnId = (unsigned long)m + 1000; mp.pos = fmat(3, 1, &pts[3 * (size_t)m]); mp.fixed = mp_fixed[m] != 0; mp.bad = mp_bad[m] != 0; mp.cache = &cache; }
            for (size_t e = 0; e < edges.size(); e++) {
                mock::KeyFrame& kf = *kfs[edges[e].pose];
                kf.mvKeysUn.push_back(mock::KeyPoint{{edges[e].u, edges[e].v}, 31.f, 0.f, 0.f, octave[e], -1}); kf.mvuRight.push_back(edges[e].u_right);
                kf.mvInvLevelSigma2[octave[e]] = edges[e].inv_sigma2;
                kf.mps.push_back(mps[edges[e].point].get());
                mps[edges[e].point]->obs[&kf] = kf.mvKeysUn.size() - 1; mps[edges[e].point]->nObs += edges[e].u_right >= 0 ? 2 : 1;
            }
            for (auto& mp : mps) for (auto& o : mp->obs) if (!mp->refKF || o.first->mnId < mp->refKF->mnId) mp->refKF = o.first;      // (the creating keyframe: the first observer)
            int F = 1; for (auto& k : kfs) { k->N = (int)k->mvKeysUn.size(); k->mvKeys = k->mvKeysUn; F = std::max(F, k->N); }
            CorbKfStore* KS = nullptr; CorbMpStore* MS = nullptr;
            corb::check(corb_kf_store_create(0, K, F, &KS), "corb_kf_store_create"); corb::check(corb_mp_store_create(0, M, 16, &MS), "corb_mp_store_create");
            std::vector<int32_t> loc, fix, ms(M); std::vector<mock::MapPoint*> vmp(M); std::vector<mock::KeyFrame*> vkf(K);
            for (int k = 0; k < K; k++) { Store::PutKeyFrame(KS, k, kfs[k].get(), 1); (k < n_local ? loc : fix).push_back(k); vkf[k] = kfs[k].get(); }
            for (int m = 0; m < M; m++) { vmp[m] = mps[m].get(); ms[m] = m; }
            Store::PutMapPoints(MS, 0, vmp, 1);
            bool stop = false; std::vector<std::pair<int32_t, int32_t>> erased;
            const CorbBAResult r = Store::LocalBundleAdjustment(KS, loc, fix, MS, ms, 1.2f, &stop, &erased);
            Store::ReadBackLocalBA(KS, loc, vkf, MS, 0, vmp, erased);
            std::vector<float> Tout, Xout; std::vector<int32_t> er, nobs, held;
            for (int k = 0; k < K; k++) { Tout.insert(Tout.end(), kfs[k]->Tcw.f.begin(), kfs[k]->Tcw.f.end()); int h = 0; for (auto* q : kfs[k]->mps) h += q != nullptr; held.push_back(h); }
            for (int m = 0; m < M; m++) { Xout.insert(Xout.end(), mps[m]->pos.f.begin(), mps[m]->pos.f.end()); nobs.push_back(mps[m]->bad ? -1 : (int32_t)mps[m]->obs.size()); }
            for (auto& e : erased) { er.push_back(e.first); er.push_back(e.second); }
            // ... and the records agree with the objects about the lists
            std::vector<CorbMapPointRecord> recs(M); corb::check(corb_mp_store_get(MS, 0, M, recs.data(), nullptr, nullptr), "corb_mp_store_get");
            std::vector<int32_t> rec_nobs(M); for (int m = 0; m < M; m++) rec_nobs[m] = (recs[m].flags & CORB_MP_BAD) ? -1 : recs[m].n_obs;
            out.arr(Tout); out.arr(Xout); out.arr(er); out.arr(nobs); out.arr(rec_nobs); out.arr(held); out.arr(std::vector<int32_t>{r.iters_done, cache.nUpdKF, cache.nUpdMP});
            corb_kf_store_destroy(KS); corb_mp_store_destroy(MS);
        }
        // ---- H. ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:425-538): vpPoints with NULL-free bad / already-found points,
        //         vpMatched holding unrelated and already-found MapPoints on entry ----
        {
            mock::KeyFrame K;
            K.mDescriptors = desc_mat(in.arr<uint8_t>()); K.N = K.mDescriptors.rows;
            K.mvKeysUn = keys(in.arr<CorbKeyPoint>()); K.mvKeys = K.mvKeysUn; K.mvuRight = in.arr<float>();
            const std::vector<float> cam = in.arr<float>();            // fx fy cx cy bf minx miny maxx maxy logscale
            K.fx = cam[0]; K.fy = cam[1]; K.cx = cam[2]; K.cy = cam[3]; K.mbf = cam[4]; K.mnMinX = (int)cam[5]; K.mnMinY = (int)cam[6]; K.mnMaxX = (int)cam[7]; K.mnMaxY = (int)cam[8]; K.mfLogScaleFactor = cam[9];
            K.mvScaleFactors = in.arr<float>(); K.mvInvLevelSigma2 = in.arr<float>();
            const std::vector<float> Scw = in.arr<float>(), world = in.arr<float>(), normal = in.arr<float>(), dmin = in.arr<float>(), dmax = in.arr<float>();
            const std::vector<uint8_t> pdesc = in.arr<uint8_t>(), pbad = in.arr<uint8_t>();
            const std::vector<int32_t> held = in.arr<int32_t>();       // per feature of K: -1 = NULL, -2 = an unrelated MapPoint, >= 0 = vpPoints[held]
            const int th = in.one<int32_t>();
            const int M = (int)pbad.size();
            std::vector<std::unique_ptr<mock::MapPoint>> mps; std::vector<mock::MapPoint*> vpPoints(M);
            for (int i = 0; i < M; i++) {
                mps.emplace_back(new mock::MapPoint()); mock::MapPoint& p = *mps.back(); p.mnId = 100 + (unsigned long)i; p.bad = pbad[i] != 0;
                p.pos = fmat(3, 1, &world[3 * (size_t)i]); p.normal = fmat(3, 1, &normal[3 * (size_t)i]); p.minDistance = dmin[i]; p.maxDistance = dmax[i];
                p.descriptor = desc_mat(std::vector<uint8_t>(pdesc.begin() + 32 * (size_t)i, pdesc.begin() + 32 * (size_t)(i + 1)));
                vpPoints[i] = &p;
            }
            mock::MapPoint other;
            std::vector<mock::MapPoint*> vpMatched(K.N, nullptr);
            for (int f = 0; f < K.N; f++) vpMatched[f] = held[f] == -1 ? nullptr : held[f] == -2 ? &other : vpPoints[held[f]];
            const std::vector<mock::MapPoint*> before = vpMatched;
            Matcher m(0.75f, true);
            const int n = m.SearchByProjection(&K, fmat(4, 4, Scw.data()), vpPoints, vpMatched, th);
            std::map<mock::MapPoint*, int> where; for (int i = 0; i < M; i++) where[vpPoints[i]] = i;
            std::vector<int32_t> o(K.N, -1); int kept = 0;
            for (int f = 0; f < K.N; f++) { if (before[f]) { kept += vpMatched[f] == before[f]; continue; } if (vpMatched[f]) o[f] = where.at(vpMatched[f]); }
            int had = 0; for (auto* q : before) had += q != nullptr;
            out.arr(o); out.arr(std::vector<int32_t>{n, kept, had});
        }
        // ---- I. ORBmatcher::SearchForInitialization(Frame&, Frame&, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:540-655) ----
        {
            mock::Frame F1, F2;
            F1.mDescriptors = desc_mat(in.arr<uint8_t>()); F1.N = F1.mDescriptors.rows; F1.mvKeysUn = keys(in.arr<CorbKeyPoint>()); F1.mvKeys = F1.mvKeysUn;
            F2.mDescriptors = desc_mat(in.arr<uint8_t>()); F2.N = F2.mDescriptors.rows; F2.mvKeysUn = keys(in.arr<CorbKeyPoint>()); F2.mvKeys = F2.mvKeysUn;
            const std::vector<float> bounds = in.arr<float>(), pm = in.arr<float>();
            const int win = in.one<int32_t>();
            mock::Frame::mnMinX = bounds[0]; mock::Frame::mnMinY = bounds[1]; mock::Frame::mnMaxX = bounds[2]; mock::Frame::mnMaxY = bounds[3];
            std::vector<mock::Point2f> prev(F1.N); for (int i = 0; i < F1.N; i++) prev[i] = mock::Point2f{pm[2 * i], pm[2 * i + 1]};
            std::vector<int> m12;
            Matcher m(0.9f, true);
            const int n = m.SearchForInitialization(F1, F2, prev, m12, win);
            std::vector<float> pout; for (auto& q : prev) { pout.push_back(q.x); pout.push_back(q.y); }
            out.arr(std::vector<int32_t>(m12.begin(), m12.end())); out.arr(pout); out.arr(std::vector<int32_t>{n});
        }
        return 0;
    } catch (const corb::Error& e) {
        std::fprintf(stderr, "corb::Error %d: %s\n", e.code, e.what());
        return 3;
    }
}
