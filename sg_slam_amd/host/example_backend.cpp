// example_backend.cpp — the LocalMapping / detector-thread side of the C++ mirror (sgx_host.hpp): Optimizer::LocalBundleAdjustment on a flattened local
// graph and Detector2D::detect, driven from files so tests/test_host_cpp_gpu.py can compare against the oracle / the Python mirror.
//   example_backend ba  graph.bin                       graph.bin = int32 np, nl, ne | poses f32 | fixed u8 | points f32 | edge_pose i32 | edge_point i32 | obs f32 | info f32
//   example_backend det model.param model.bin frame.raw   frame.raw = 480 x 640 x 3 u8 (BGR)
//   example_backend flow cur.raw prev.raw pts.bin          two 480 x 640 u8 frames, pts.bin = int32 n | n x 2 f32: calcOpticalFlowPyrLK + findFundamentalMat (Frame.cc:445, :469-472)
//   example_backend sim3 pairs.bin                        int32 n, fix_scale | p1c p2c (n x 3 f32) obs1 obs2 (n x 2 f32) info1 info2 (n f32) | K1 K2 (4 f32) | S12 (8 f64): OptimizeSim3
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
    if (argc >= 5 && !strcmp(argv[1], "flow")) {
        const std::vector<uint8_t> cur = slurp(argv[2]), prev = slurp(argv[3]), pb = slurp(argv[4]);
        const int n = *(const int32_t *)pb.data();
        std::vector<float> pts((size_t)2 * n), next; std::vector<uint8_t> st;
        memcpy(pts.data(), pb.data() + 4, sizeof(float) * 2 * (size_t)n);
        sgx::OpticalFlowLK lk(640, 480);
        lk(cur.data(), prev.data(), pts, next, st);
        std::vector<float> a, b; int ntr = 0;
        for (int i = 0; i < n; i++) if (st[(size_t)i]) { ntr++; a.push_back(pts[2 * (size_t)i]); a.push_back(pts[2 * (size_t)i + 1]); b.push_back(next[2 * (size_t)i]); b.push_back(next[2 * (size_t)i + 1]); }
        double F[9]; const bool ok = sgx::findFundamentalMat(a, b, F);
        printf("tracked %d ok %d\n", ntr, (int)ok);
        printf("first"); for (int i = 0; i < 8 && i < 2 * n; i++) printf(" %.9g", next[(size_t)i]); printf("\n");
        printf("F"); for (int i = 0; i < 9; i++) printf(" %.17g", F[i]); printf("\n");
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "sim3")) {
        const std::vector<uint8_t> b = slurp(argv[2]);
        const int32_t *hdr = (const int32_t *)b.data(); const int n = hdr[0], fix = hdr[1];
        const uint8_t *p = b.data() + 8;
        sgx::Optimizer::Sim3Pairs c; float K1[4], K2[4]; sgx::Optimizer::Sim3 S;
        auto take = [&](std::vector<float> &v, size_t m) { v.resize(m); memcpy(v.data(), p, m * 4); p += m * 4; };
        take(c.p1c, (size_t)n * 3); take(c.p2c, (size_t)n * 3); take(c.obs1, (size_t)n * 2); take(c.obs2, (size_t)n * 2); take(c.info1, (size_t)n); take(c.info2, (size_t)n);
        memcpy(K1, p, 16); p += 16; memcpy(K2, p, 16); p += 16; memcpy(S.v, p, 64);
        std::vector<uint8_t> inl;
        const int nin = sgx::Optimizer::OptimizeSim3(c, K1, K2, S, 10.0f, fix != 0, inl);
        int kept = 0; for (uint8_t e : inl) kept += e;
        printf("nin %d kept %d\n", nin, kept);
        printf("S12"); for (int i = 0; i < 8; i++) printf(" %.17g", S.v[i]); printf("\n");
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "voc")) {                     // Frame::ComputeBoW: vocabulary file (.txt = text, else binary), descriptors (N x 32 bytes)
        sgx::ORBVocabulary voc;
        const std::string vf(argv[2]);
        const bool okload = vf.size() >= 4 && vf.compare(vf.size() - 4, 4, ".txt") == 0 ? voc.loadFromTextFile(vf) : voc.loadFromBinaryFile(vf);
        if (!okload) { fprintf(stderr, "Wrong path to vocabulary.\n"); return 1; }        // System.cc:74-79
        const std::vector<uint8_t> d = slurp(argv[3]);
        sgx::ORBVocabulary::BowVector v; std::vector<int32_t> fn;
        voc.transform(d, v, fn, 4);
        double wsum = 0; long long idsum = 0, nodesum = 0; for (auto &e : v) { wsum += e.second; idsum += e.first; } for (int32_t x : fn) nodesum += x;
        printf("words %u bow %zu idsum %lld nodesum %lld wsum %.17g self %.17g\n", voc.size(), v.size(), idsum, nodesum, wsum, voc.score(v, v));
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "s3solver")) {               // Sim3Solver: n, fix, then x3dc1, x3dc2 (n x 3), maxErr1, maxErr2 (n), K1, K2 (4 floats each)
        const std::vector<uint8_t> b = slurp(argv[2]);
        const int32_t *hdr = (const int32_t *)b.data(); const int n = hdr[0], fix = hdr[1];
        const uint8_t *p = b.data() + 8;
        std::vector<float> x1, x2, e1, e2; float K1[4], K2[4];
        auto take = [&](std::vector<float> &v, size_t m) { v.resize(m); memcpy(v.data(), p, m * 4); p += m * 4; };
        take(x1, (size_t)n * 3); take(x2, (size_t)n * 3); take(e1, (size_t)n); take(e2, (size_t)n); memcpy(K1, p, 16); p += 16; memcpy(K2, p, 16);
        std::vector<int32_t> idx((size_t)n); for (int i = 0; i < n; i++) idx[(size_t)i] = 2 * i;          // pretend every second keypoint of pKF1 carries a usable pair
        sgx::Sim3Solver solver(x1, x2, e1, e2, K1, K2, idx, 2 * n, fix != 0, 7);
        solver.SetRansacParameters(0.99, 20, 300);
        bool bNoMore = false; std::vector<bool> vbInliers; int nInliers = 0, calls = 0; float T12[16]; bool found = false;
        while (!found && !bNoMore) { found = solver.iterate(5, bNoMore, vbInliers, nInliers, T12); calls++; }     // the loop of LoopClosing::ComputeSim3 (:296-335) for one candidate
        int marked = 0, odd = 0; for (size_t i = 0; i < vbInliers.size(); i++) { marked += vbInliers[i]; if ((i & 1) && vbInliers[i]) odd++; }
        printf("found %d calls %d inliers %d marked %d odd %d\n", (int)found, calls, nInliers, marked, odd);
        printf("T12"); for (int i = 0; i < 16; i++) printf(" %.9g", found ? T12[i] : 0.f); printf("\n");
        return 0;
    }
    fprintf(stderr, "usage: %s ba graph.bin | det model.param model.bin frame.raw | flow cur.raw prev.raw pts.bin | sim3 pairs.bin | voc voc.txt desc.bin | s3solver pairs.bin\n", argv[0]);
    return 2;
}
