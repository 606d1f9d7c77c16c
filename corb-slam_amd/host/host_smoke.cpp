// host_smoke.cpp -- C++ host-side use of the reference-shaped interface (corb_host.hpp) over the C-ABI.
// Built by tests/test_cabi.py (compile + link check on CPU) and run on the GPU box by tests/test_gpu_host.py.
// Usage: host_smoke <left.raw> <right.raw> <width> <height>   -> prints counts and an FNV-1a hash of the outputs
#include "corb_host.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{ const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }

int main(int argc, char** argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s left.raw right.raw width height\n", argv[0]); return 2; }
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    auto slurp = [](const char* path) { std::ifstream f(path, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); };
    std::vector<uint8_t> L = slurp(argv[1]), R = slurp(argv[2]);
    if ((int)L.size() != w * h || (int)R.size() != w * h) { std::fprintf(stderr, "bad image size\n"); return 2; }
    try {
        // the reference's Tracking ctor: two extractors, KITTI settings (Tracking.cc:112-124)
        corb::ORBextractor left(2000, 1.2f, 8, 20, 7, w, h), right(2000, 1.2f, 8, 20, 7, w, h);
        std::vector<corb::KeyPoint> kl, kr; corb::Descriptors dl, dr;
        left(L.data(), w, h, w, kl, dl);
        right(R.data(), w, h, w, kr, dr);
        corb::StereoFrontend sf(2000, 1.2f, 8, 20, 7, w, h, 718.856f, 386.1448f, 1);
        sf.Upload(0, L.data(), R.data(), w); sf.Run(1); sf.Sync();
        corb::StereoFrontend::FrameResult fr = sf.Fetch(0);
        int matched = 0; for (float u : fr.mvuRight) matched += u >= 0;
        // the per-frame call of a client (corb_stereo_frames): the same frame in one call, byte for byte the same results
        const CorbStereoFrameLayout lay = sf.FrameLayout();
        std::vector<uint8_t> both(L); both.insert(both.end(), R.begin(), R.end());
        std::vector<uint8_t> block((size_t)lay.frame_bytes);
        sf.Frames(1, both.data(), block.data());
        const corb::StereoFrontend::FrameResult f1 = corb::StereoFrontend::FromBlock(lay, block.data());
        const bool one_call = f1.mvKeys.size() == fr.mvKeys.size() && f1.mvKeysRight.size() == fr.mvKeysRight.size() &&
                              fnv(f1.mvKeys.data(), f1.mvKeys.size() * sizeof(corb::KeyPoint)) == fnv(fr.mvKeys.data(), fr.mvKeys.size() * sizeof(corb::KeyPoint)) &&
                              f1.mDescriptors.data == fr.mDescriptors.data && f1.mDescriptorsRight.data == fr.mDescriptorsRight.data &&
                              fnv(f1.mvuRight.data(), f1.mvuRight.size() * 4) == fnv(fr.mvuRight.data(), fr.mvuRight.size() * 4) &&
                              fnv(f1.mvDepth.data(), f1.mvDepth.size() * 4) == fnv(fr.mvDepth.data(), fr.mvDepth.size() * 4);
        if (!one_call) { std::fprintf(stderr, "corb_stereo_frames differs from upload / run / fetch\n"); return 1; }
        const bool same = fr.mvKeys.size() == kl.size() && fnv(fr.mvKeys.data(), kl.size() * sizeof(corb::KeyPoint)) == fnv(kl.data(), kl.size() * sizeof(corb::KeyPoint)) &&
                          fnv(fr.mDescriptorsRight.data.data(), fr.mDescriptorsRight.data.size()) == fnv(dr.data.data(), dr.data.size());
        std::printf("n_left=%zu n_right=%zu matched=%d consistent=%d kp_hash=%016llx desc_hash=%016llx\n", kl.size(), kr.size(), matched, (int)same,
                    (unsigned long long)fnv(kl.data(), kl.size() * sizeof(corb::KeyPoint)), (unsigned long long)fnv(dl.data.data(), dl.data.size()));
        return same ? 0 : 1;
    } catch (const corb::Error& e) {
        std::fprintf(stderr, "corb::Error %d: %s\n", e.code, e.what());
        return 3;
    }
}
