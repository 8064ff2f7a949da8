// example_pipeline.cpp — the pipelined tracking host driven from C++ (sgx::TrackingPipeline over sgx_tracker_*): what Examples/rgbd_tum.cc's main loop does for one
// sequence (rgbd_tum.cc:108-148: imread, TrackRGBD per frame), here for S sequences in lock-step.  Reads raw frames written by the test
//   <dir>/s<stream>_f<frame>.bgr   640 x 480 x 3 bytes (interleaved, as cv::imread delivers)      <dir>/s<stream>_f<frame>.depth   640 x 480 uint16
// and <dir>/poses0.f32 (streams x 16 floats: the first frame's Tcw), tracks frames 0 .. F-1 of every stream and prints one line per (frame, stream):
//   frame stream nkeys nmatches ninliers Tcw[16]
// tests/test_host_cpp_gpu.py compares the lines bit for bit with the Python binding of the same library on the same files.
#include <cstdio>
#include <cstdlib>
#include <string>
#include "sgx_host.hpp"

static void read_into(const std::string &p, void *dst, size_t n)
{
    FILE *f = fopen(p.c_str(), "rb");
    if (!f || fread(dst, 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", p.c_str()); exit(2); }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s dir streams frames\n", argv[0]); return 2; }
    const std::string dir = argv[1]; const int S = atoi(argv[2]), F = atoi(argv[3]);
    const sgx_camera cam{535.4f, 539.2f, 320.1f, 247.6f, 40.0f, 0.f, 640.f, 0.f, 480.f};       // Examples/TUM3.yaml
    try {
        sgx::TrackingPipeline trk(S, cam, 5000.f);
        std::vector<float> T0((size_t)16 * S); read_into(dir + "/poses0.f32", T0.data(), T0.size() * 4);
        trk.SetInitialPose(T0);
        std::vector<float> Tcw; std::vector<int32_t> nk, nm, ni;
        for (int f = 0; f < F; f++) {
            uint8_t *bgr; int pitch; uint16_t *depth;
            trk.HostBuffers(f & 1, &bgr, &pitch, &depth);
            for (int s = 0; s < S; s++) {
                std::vector<uint8_t> tight((size_t)640 * 480 * 3);
                read_into(dir + "/s" + std::to_string(s) + "_f" + std::to_string(f) + ".bgr", tight.data(), tight.size());
                for (int y = 0; y < 480; y++) memcpy(bgr + ((size_t)s * 480 + y) * pitch, tight.data() + (size_t)y * 640 * 3, 640 * 3);
                read_into(dir + "/s" + std::to_string(s) + "_f" + std::to_string(f) + ".depth", depth + (size_t)s * 480 * 640, (size_t)640 * 480 * 2);
            }
            trk.GrabImagesRGBD(f & 1, true);                  // asynchronous; the read below synchronises (a real caller reads poses a few frames later)
            trk.Pose(Tcw, &nk, &nm, &ni);
            for (int s = 0; s < S; s++) {
                printf("%d %d %d %d %d", f, s, nk[s], nm[s], ni[s]);
                for (int i = 0; i < 16; i++) printf(" %.9g", Tcw[(size_t)16 * s + i]);
                printf("\n");
            }
        }
    } catch (const std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
