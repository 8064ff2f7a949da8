// example_backend.cpp — the LocalMapping / detector-thread side of the C++ mirror (sgx_host.hpp): Optimizer::LocalBundleAdjustment on a flattened local
// graph and Detector2D::detect, driven from files so tests/test_host_cpp_gpu.py can compare against the oracle / the Python mirror.
//   example_backend ba  graph.bin                       graph.bin = int32 np, nl, ne | poses f32 | fixed u8 | points f32 | edge_pose i32 | edge_point i32 | obs f32 | info f32
//   example_backend det model.param model.bin frame.raw   frame.raw = 480 x 640 x 3 u8 (BGR)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "sgx_host.hpp"

static std::vector<uint8_t> slurp(const char *p)
{
    FILE *f = fopen(p, "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", p); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", p); exit(2); }
    fclose(f);
    return b;
}

int main(int argc, char **argv)
{
    const sgx_camera cam{535.4f, 539.2f, 320.1f, 247.6f, 40.0f, 0.f, 640.f, 0.f, 480.f};
    if (argc >= 3 && !strcmp(argv[1], "ba")) {
        const std::vector<uint8_t> b = slurp(argv[2]);
        const int32_t *hdr = (const int32_t *)b.data(); const int np = hdr[0], nl = hdr[1], ne = hdr[2];
        const uint8_t *p = b.data() + 12;
        sgx::Optimizer::LocalGraph g;
        auto take = [&](auto &v, size_t n) { v.resize(n); memcpy(v.data(), p, n * sizeof(v[0])); p += n * sizeof(v[0]); };
        take(g.poses, (size_t)np * 16); take(g.pose_fixed, (size_t)np); take(g.points, (size_t)nl * 3); take(g.edge_pose, (size_t)ne); take(g.edge_point, (size_t)ne);
        take(g.edge_obs, (size_t)ne * 3); take(g.edge_info, (size_t)ne);
        sgx_ba_stats st;
        const std::vector<uint8_t> erase = sgx::Optimizer::LocalBundleAdjustment(g, cam, nullptr, &st);
        int nerase = 0; for (uint8_t e : erase) nerase += e;
        printf("iterations %d %d chi2 %.17g erased %d\n", st.iterations_first, st.iterations_second, st.chi2_second, nerase);
        printf("pose1"); for (int i = 0; i < 16; i++) printf(" %.9g", g.poses[16 + i]); printf("\n");
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "det")) {
        const std::vector<uint8_t> param = slurp(argv[2]), bin = slurp(argv[3]), img = slurp(argv[4]);
        sgx::Detector2D det(0.9f, 0.01f, std::string(param.begin(), param.end()), bin);
        det.detect(img.data(), 640 * 3);
        printf("raw %zu objects %zu map_boxes %zu rm_boxes %zu dyn_map %d dyn_rm %d\n", det.raw.size(), det.mvObjects2D.size(), det.mvPotentialDynamicBorderForMapping.size(),
               det.mvPotentialDynamicBorderForRmDynamicFeature.size(), (int)det.mbHaveDynamicObjectForMapping, (int)det.mbHaveDynamicObjectForRmDynamicFeature);
        for (size_t i = 0; i < det.raw.size() && i < 5; i++) printf("row %g %.9g %.9g %.9g %.9g %.9g\n", det.raw[i].label, det.raw[i].score, det.raw[i].xmin, det.raw[i].ymin, det.raw[i].xmax, det.raw[i].ymax);
        return 0;
    }
    fprintf(stderr, "usage: %s ba graph.bin | det model.param model.bin frame.raw\n", argv[0]);
    return 2;
}
