#!/usr/bin/env python3
"""pin_third_party.py — the route from "parity unpinned" to pinned.

The reference's tracking path calls arithmetic that lives in libraries absent from /root/reference and from this image: OpenCV 3.4.15 (cv::FAST, cv::resize,
cv::GaussianBlur, cv::fastAtan2, cv::pyrDown, cv::calcOpticalFlowPyrLK, cv::findFundamentalMat, cv::cvtColor — call sites ORBextractor.cc:104,810,1087,1121,
Frame.cc:445,469-472, Tracking.cc:214-227) and ncnn (Detector2D.cc:20-45 on Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.param).  oracle/ restates them from the
published algorithms.  On ANY machine that has the real libraries (pip: `opencv-python==3.4.15.55` or a source build of 3.4.x, and the `ncnn` wheel), run

    python tools/pin_third_party.py            # writes tests/golden/third_party/*.npz

and commit the files: tests/test_third_party_pins.py then checks the oracle against them primitive by primitive (bit-exact for the integer / fixed-point ones,
stated tolerances for LK / RANSAC / the network) and is skipped while they are absent.  Inputs come from the repo's own seeded generators, so the fixtures are
small and reproducible; the script records library versions and build information in every file.  It never runs on the GPU box and nothing in the product imports it."""
import argparse
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden', 'third_party')


def frames(n=3):
    from sg_slam_amd import synth
    gen = synth.LayeredStream(seed=1234)
    return [gen.frame(10 + i)[0] for i in range(n)]


def pin_opencv():
    import cv2
    ver = cv2.__version__
    if not ver.startswith('3.4'):
        print(f'WARNING: OpenCV {ver} is not 3.4.x (the reference pins 3.4.15, README.md:86-97): fixtures are written, but label them accordingly', file=sys.stderr)
    info = dict(opencv_version=ver, build=cv2.getBuildInformation()[:4000])
    cv2.setNumThreads(1)
    g = frames(3)
    out = {}
    # cv::resize chain of ORBextractor::ComputePyramid (ORBextractor.cc:1108-1133): level l from level l-1, INTER_LINEAR
    sizes = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    lv = g[0]; pyr = []
    for (w, h) in sizes[1:]:
        lv = cv2.resize(lv, (w, h), interpolation=cv2.INTER_LINEAR); pyr.append(lv)
    for i, p in enumerate(pyr): out[f'resize_l{i + 1}'] = p
    out['resize_src'] = g[0]
    # cv::FAST(img, kps, t, true) at both thresholds on a cell-sized crop and on the whole frame (ORBextractor.cc:810-815)
    for t in (20, 7):
        det = cv2.FastFeatureDetector_create(threshold=t, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        for name, img in (('full', g[1]), ('crop', np.ascontiguousarray(g[1][100:136, 200:236]))):
            k = det.detect(img, None)
            out[f'fast_{name}_t{t}'] = np.array([[p.pt[0], p.pt[1], p.response] for p in k], np.float32).reshape(-1, 3)
    out['fast_src'] = g[1]
    # cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) (ORBextractor.cc:1087)
    out['blur_src'] = g[2]; out['blur'] = cv2.GaussianBlur(g[2], (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    # cv::fastAtan2 (ORBextractor.cc:104) on a grid of moment pairs incl. axes and octant borders
    rng = np.random.RandomState(1)
    yx = np.concatenate([rng.randint(-50000, 50000, (4000, 2)), [[0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, 1], [1, -1], [-1, -1], [0, 0]]]).astype(np.float32)
    out['atan2_yx'] = yx; out['atan2'] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in yx], np.float32)
    # cv::cvtColor RGB2GRAY / BGR2GRAY (Tracking.cc:214-227)
    col = rng.randint(0, 256, (64, 96, 3)).astype(np.uint8)
    out['cvt_src'] = col; out['cvt_rgb2gray'] = cv2.cvtColor(col, cv2.COLOR_RGB2GRAY); out['cvt_bgr2gray'] = cv2.cvtColor(col, cv2.COLOR_BGR2GRAY)
    # cv::pyrDown chain of buildOpticalFlowPyramid + cv::calcOpticalFlowPyrLK exactly as Frame.cc:445 calls it
    p1 = cv2.pyrDown(g[0]); out['pyrdown_1'] = p1; out['pyrdown_2'] = cv2.pyrDown(p1)
    det = cv2.FastFeatureDetector_create(threshold=20, nonmaxSuppression=True)
    pts = np.array([p.pt for p in det.detect(g[1], None)][:1500], np.float32).reshape(-1, 1, 2)
    nxt, st, err = cv2.calcOpticalFlowPyrLK(g[1], g[0], pts, None, winSize=(21, 21), maxLevel=3, criteria=(cv2.TERM_CRITERIA_EPS | cv2.TERM_CRITERIA_COUNT, 30, 0.01))
    out['lk_cur'] = g[1]; out['lk_prev'] = g[0]; out['lk_pts'] = pts.reshape(-1, 2); out['lk_next'] = nxt.reshape(-1, 2); out['lk_status'] = st.reshape(-1)
    # cv::findFundamentalMat(cur, prev, FM_RANSAC, 1.0, 0.99) (Frame.cc:469-472)
    good = st.reshape(-1) > 0
    F, mask = cv2.findFundamentalMat(pts.reshape(-1, 2)[good], nxt.reshape(-1, 2)[good], cv2.FM_RANSAC, 1.0, 0.99)
    out['fm_p1'] = pts.reshape(-1, 2)[good]; out['fm_p2'] = nxt.reshape(-1, 2)[good]; out['fm_F'] = np.zeros((3, 3)) if F is None else F[:3]; out['fm_mask'] = np.zeros(0, np.uint8) if mask is None else mask.reshape(-1)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'opencv.npz'), info=np.array([repr(info)]), **out)
    print('wrote', os.path.join(OUT, 'opencv.npz'), 'OpenCV', ver)


def pin_ncnn(bin_path):
    import ncnn
    from sg_slam_amd import synth
    param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
    layers = synth.parse_ncnn_param(param)
    if bin_path:
        blob = open(bin_path, 'rb').read(); wnote = os.path.basename(bin_path)
    else:
        _, blob = synth.synth_ncnn_weights(layers, seed=7); wnote = 'synthetic N(0, 2/fan_in) seed 7 + batch-norm-style calibration fold (sg_slam_amd.synth.synth_ncnn_weights)'
    tmp = os.path.join(OUT, '_weights.bin'); os.makedirs(OUT, exist_ok=True)
    open(tmp, 'wb').write(blob)
    net = ncnn.Net(); net.opt.use_vulkan_compute = False; net.opt.num_threads = 1
    for k in ('use_fp16_packed', 'use_fp16_storage', 'use_fp16_arithmetic', 'use_int8_inference', 'use_packing_layout', 'use_winograd_convolution', 'use_sgemm_convolution'):
        if hasattr(net.opt, k): setattr(net.opt, k, False)          # plain fp32 reference kernels (record the options below)
    net.load_param(param); net.load_model(tmp)
    img = frames(1)[0]; bgr = np.repeat(img[..., None], 3, -1)
    m = ncnn.Mat.from_pixels_resize(bgr, ncnn.Mat.PixelType.PIXEL_RGB, 640, 480, 300, 300)           # Detector2D.cc:39
    m.substract_mean_normalize([123.675, 116.28, 103.53], [1.0, 1.0, 1.0])                              # Detector2D.cc:40
    out = {'input_bgr': bgr, 'input_blob': np.array(m)}
    for name in ('587', '632', '672', '849', '908', '944', 'mbox_loc', 'mbox_conf_softmax', 'mbox_priorbox', 'detection_out'):
        ex = net.create_extractor(); ex.input('input', m)
        ret, o = ex.extract(name)
        if ret == 0: out['blob_' + name] = np.array(o)
    info = dict(ncnn_version=getattr(ncnn, '__version__', '?'), weights=wnote, options='fp32, no packing / winograd / sgemm / fp16 / int8, 1 thread')
    os.remove(tmp)
    np.savez_compressed(os.path.join(OUT, 'ncnn.npz'), info=np.array([repr(info)]), **out)
    print('wrote', os.path.join(OUT, 'ncnn.npz'))


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--bin', default='', help='real mobilenetv3_ssdlite_voc.bin when available (else the synthetic weights of the tests)')
    ap.add_argument('--only', choices=['opencv', 'ncnn'], default=None)
    a = ap.parse_args()
    done = 0
    if a.only in (None, 'opencv'):
        try: pin_opencv(); done += 1
        except ImportError as e: print('OpenCV (cv2) not importable here:', e, file=sys.stderr)
    if a.only in (None, 'ncnn'):
        try: pin_ncnn(a.bin); done += 1
        except ImportError as e: print('ncnn not importable here:', e, file=sys.stderr)
    sys.exit(0 if done else 3)
