// example_track.cpp — the reference's per-frame call order (Frame ctor -> TrackWithMotionModel, Tracking.cc:906-967)
// written against the C++ mirror classes.  Reads two raw 640x480 gray frames (+ a constant depth) from files given
// on the command line and prints keypoints / matches / inliers / pose; tests/test_host_cpp_gpu.py feeds it
// synthetic frames and compares the printed numbers with the oracle.
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
    printf("N0 %d N1 %d matches %d inliers %d\n", last.N, cur.N, nmatches, ninl);
    printf("Tcw");
    for (int i = 0; i < 16; i++) printf(" %.9g", cur.mTcw[i]);
    printf("\n");
    return 0;
}
