// example_track.cpp — the reference's per-frame call order (Frame ctor -> TrackWithMotionModel, Tracking.cc:906-967)
// written against the C++ mirror classes.  Reads two raw 640x480 gray frames (+ a constant depth) from files given
// on the command line and prints keypoints / matches / inliers / pose; tests/test_host_cpp_gpu.py feeds it
// synthetic frames and compares the printed numbers with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "sgx_host.hpp"

static std::vector<uint8_t> read_file(const char *p, size_t n)
{
    std::vector<uint8_t> b(n);
    FILE *f = fopen(p, "rb");
    if (!f || fread(b.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", p); exit(2); }
    fclose(f);
    return b;
}

static void fill_frame(sgx::FrameView &F, sgx::ORBextractor &ex, const std::vector<uint8_t> &gray, float z, const sgx_camera &cam)
{
    sgx::GrayView g{gray.data(), 480, 640, 640};
    ex(g, nullptr, F.mvKeysUn, F.mDescriptors);
    F.N = (int)F.mvKeysUn.size();
    F.mvuRight.assign(F.N, -1.f); F.mvDepth.assign(F.N, -1.f);
    for (int i = 0; i < F.N; i++) { F.mvDepth[i] = z; F.mvuRight[i] = F.mvKeysUn[i].x - cam.bf / z; }   // Frame::ComputeStereoFromRGBD on a constant-depth image
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s frame0.raw frame1.raw\n", argv[0]); return 2; }
    const sgx_camera cam{535.4f, 539.2f, 320.1f, 247.6f, 40.0f, 0.f, 640.f, 0.f, 480.f};
    const float z = 2.0f;
    sgx::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    sgx::FrameView last, cur;
    fill_frame(last, ex, read_file(argv[1], 640 * 480), z, cam);
    fill_frame(cur, ex, read_file(argv[2], 640 * 480), z, cam);
    // UpdateLastFrame-style map points: unproject every last-frame keypoint with its depth at identity pose
    last.mpValid.assign(last.N, 1); last.mpObservations.assign(last.N, 0); last.mpDescriptor = last.mDescriptors;
    last.mpWorldPos.resize((size_t)last.N * 3);
    for (int i = 0; i < last.N; i++) {
        last.mpWorldPos[3 * i] = (last.mvKeysUn[i].x - cam.cx) * z * (1.0f / cam.fx);
        last.mpWorldPos[3 * i + 1] = (last.mvKeysUn[i].y - cam.cy) * z * (1.0f / cam.fy);
        last.mpWorldPos[3 * i + 2] = z;
    }
    sgx::ORBmatcher matcher(0.9f, true);
    int nmatches = matcher.SearchByProjection(cur, last, 15, false, cam, ex.GetScaleFactors());
    if (nmatches < 20) nmatches = matcher.SearchByProjection(cur, last, 30, false, cam, ex.GetScaleFactors());   // Tracking.cc:927-931
    const int ninl = sgx::Optimizer::PoseOptimization(&cur, last, cam, ex.GetInverseScaleSigmaSquares());
    // TrackLocalMap-style second search (Tracking.cc:969-1013): the local map = the last frame's points as MapPoints created from that frame
    // (MapPoint.cc:45-67: normal from the camera centre, mfMaxDistance = dist * scale[octave], mfMinDistance = mfMaxDistance / scale[nlevels - 1])
    sgx::LocalMapView lm; lm.N = last.N;
    const std::vector<float> sf = ex.GetScaleFactors();
    lm.mWorldPos = last.mpWorldPos; lm.mDescriptor = last.mDescriptors; lm.nObs.assign(lm.N, 1); lm.skip.assign(lm.N, 0);
    lm.mNormalVector.resize((size_t)lm.N * 3); lm.mfMinDistance.resize(lm.N); lm.mfMaxDistance.resize(lm.N);
    for (int i = 0; i < lm.N; i++) {
        const float *X = &lm.mWorldPos[3 * (size_t)i];
        const float d = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);       // camera centre of the last frame = origin
        for (int a = 0; a < 3; a++) lm.mNormalVector[3 * (size_t)i + a] = X[a] / d;
        lm.mfMaxDistance[i] = d * sf[last.mvKeysUn[i].octave]; lm.mfMinDistance[i] = lm.mfMaxDistance[i] / sf[sf.size() - 1];
    }
    std::vector<int32_t> held(cur.N, -1), local;
    for (int k = 0; k < cur.N; k++) if (cur.mvpMapPoints[k] >= 0 && !cur.mvbOutlier[k]) held[k] = 1;     // keypoints that kept their point from the first stage
    sgx::ORBmatcher lmatcher(0.8f);
    const int nlocal = lmatcher.SearchByProjection(cur, lm, 3, cam, sf, local, &held);
    int inview = 0; for (uint8_t v : lm.mbTrackInView) inview += v;
    printf("N0 %d N1 %d matches %d inliers %d local %d inview %d\n", last.N, cur.N, nmatches, ninl, nlocal, inview);
    printf("Tcw");
    for (int i = 0; i < 16; i++) printf(" %.9g", cur.mTcw[i]);
    printf("\n");
    return 0;
}
