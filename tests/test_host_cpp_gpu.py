"""GPU: the C++ host mirror classes (sg_slam_amd/host/sgx_host.hpp) driven by example_track.cpp — a
reference-style Frame -> SearchByProjection -> PoseOptimization sequence — against the oracle."""
import os
import subprocess
import numpy as np
import pytest
from scenes import CAM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_example_matches_oracle(gpulib, oracle, stream_frames, tmp_path):
    exe = os.path.join(ROOT, 'sg_slam_amd', 'host', 'example_track')
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    g0, _, _ = stream_frames.frame(10); g1, _, _ = stream_frames.frame(11)
    f0 = tmp_path / 'f0.raw'; f1 = tmp_path / 'f1.raw'
    g0.tofile(f0); g1.tofile(f1)
    out = subprocess.check_output([exe, str(f0), str(f1)], text=True).splitlines()
    vals = out[0].split()
    n0, n1, nm, ninl = int(vals[1]), int(vals[3]), int(vals[5]), int(vals[7])
    T = np.array([float(v) for v in out[1].split()[1:]], 'f4').reshape(4, 4)
    # oracle, chained the same way (constant depth 2 m, identity last pose)
    k0, d0 = oracle.orb_extract(g0); k1, d1 = oracle.orb_extract(g1)
    z = np.float32(2.0)
    ur0 = (k0['x'] - np.float32(CAM['bf']) / z).astype('f4'); ur1 = (k1['x'] - np.float32(CAM['bf']) / z).astype('f4')
    xw = np.stack([(k0['x'] - np.float32(CAM['cx'])) * z * (np.float32(1) / np.float32(CAM['fx'])),
                   (k0['y'] - np.float32(CAM['cy'])) * z * (np.float32(1) / np.float32(CAM['fy'])), np.full(len(k0), z)], 1).astype('f4')
    I = np.eye(4, dtype='f4')
    last = dict(keys=k0, has_mp=np.ones(len(k0), np.uint8), outlier=np.zeros(len(k0), np.uint8), xw=xw, obs=np.zeros(len(k0), 'i4'), mpdesc=d0, Tcw=I)
    cur = dict(keys=k1, desc=d1, uright=ur1, Tcw=I)
    sf = oracle.orb_params()['scale']; is2 = oracle.orb_params()['inv_sigma2']
    m, en = oracle.search_by_projection_frame(cur, last, CAM, sf, th=15)
    if en < 20:
        m, en = oracle.search_by_projection_frame(cur, last, CAM, sf, th=30)
    fr = dict(keys=k1, uright=ur1, has_mp=(m >= 0).astype(np.uint8), Tcw=I, xw=np.where((m >= 0)[:, None], xw[np.maximum(m, 0)], 0).astype('f4'))
    einl, eT, _ = oracle.pose_optimization(fr, CAM, is2)
    assert (n0, n1, nm, ninl) == (len(k0), len(k1), en, einl)
    assert np.abs(T - eT).max() <= 1e-5 * max(1.0, np.abs(eT).max())
    # TrackLocalMap-style second search through ORBmatcher::SearchByProjection(F, vpMapPoints, th): chained from the pose the C++ run printed
    nlocal, ninview = int(vals[9]), int(vals[11])
    _, _, eout = oracle.pose_optimization(fr, CAM, is2)
    dist = np.sqrt(xw[:, 0] * xw[:, 0] + xw[:, 1] * xw[:, 1] + xw[:, 2] * xw[:, 2]).astype('f4')
    sf32 = np.asarray(sf, 'f4')
    mx = (dist * sf32[k0['octave']]).astype('f4')
    lm = dict(xw=xw, normal=(xw / dist[:, None]).astype('f4'), max_dist=mx, min_dist=(mx / sf32[-1]).astype('f4'), desc=d0, obs=np.ones(len(k0), 'i4'), skip=np.zeros(len(k0), np.uint8))
    cur2 = dict(keys=k1, desc=d1, uright=ur1, Tcw=T, mp_obs=np.where((m >= 0) & (eout == 0), 1, -1).astype('i4'))
    res = oracle.search_by_projection_local(cur2, lm, CAM, sf, th=3.0, nnratio=0.8)
    assert nlocal == res[1] and ninview == int(np.asarray(res[2]).sum())


def _exe(name):
    exe = os.path.join(ROOT, 'sg_slam_amd', 'host', name)
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    return exe


def test_cpp_local_bundle_adjustment_matches_oracle(gpulib, oracle, tmp_path):
    """sgx::Optimizer::LocalBundleAdjustment (C++ mirror) on a flattened local graph: same LM iteration counts, erase count, chi2 and poses as the oracle."""
    from scenes import make_ba_problem
    prob, _, _ = make_ba_problem(oracle, n_free=12, n_fixed=6, n_points=700, seed=31)
    eposes, epoints, eerase, etrace, eiters = oracle.local_ba(prob, CAM)
    f = tmp_path / 'graph.bin'
    with open(f, 'wb') as fh:
        fh.write(np.array([len(prob['poses']), len(prob['points']), len(prob['edge_pose'])], 'i4').tobytes())
        for k, dt in (('poses', 'f4'), ('pose_fixed', 'u1'), ('points', 'f4'), ('edge_pose', 'i4'), ('edge_point', 'i4'), ('edge_obs', 'f4'), ('edge_info', 'f4')):
            fh.write(np.ascontiguousarray(prob[k], dt).tobytes())
    out = subprocess.check_output([_exe('example_backend'), 'ba', str(f)], text=True).splitlines()
    v = out[0].split()
    assert (int(v[1]), int(v[2])) == tuple(eiters) and int(v[6]) == int(eerase.sum())
    chi = etrace[1, eiters[1] - 1, 0]
    assert abs(float(v[4]) - chi) <= 1e-5 * max(1.0, chi)
    T1 = np.array([float(x) for x in out[1].split()[1:]], 'f4').reshape(4, 4)
    assert np.abs(T1 - eposes[1]).max() <= 1e-5 * max(1.0, np.abs(eposes[1]).max())


def test_cpp_detector_matches_python_mirror(gpulib, tmp_path):
    """sgx::Detector2D::detect (C++ mirror) gives the same detection rows and dynamic-object flags as the Python mirror of the same C ABI."""
    from oracle import detector_oracle as D
    from sg_slam_amd.detector import Detector2D
    param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
    layers = D.parse_param(param); W, blob = D.synth_weights(layers, seed=7)
    rng = np.random.RandomState(9)
    img = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    fb = tmp_path / 'model.bin'; fi = tmp_path / 'frame.raw'
    fb.write_bytes(blob); img.tofile(fi)
    out = subprocess.check_output([_exe('example_backend'), 'det', param, str(fb), str(fi)], text=True).splitlines()
    det = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, lib=gpulib)
    r = det.detect_batch(img)[0]
    v = out[0].split()
    assert (int(v[1]), int(v[3]), int(v[5]), int(v[7]), int(v[9]), int(v[11])) == (r.n_raw, r.n_objects, r.n_map_boxes, r.n_rm_boxes, r.have_dynamic_for_mapping, r.have_dynamic_for_rm_feature)
    for i, line in enumerate(out[1:]):
        x = [float(t) for t in line.split()[1:]]
        d = r.raw[i]
        assert np.allclose(x, [d.label, d.score, d.xmin, d.ymin, d.xmax, d.ymax], rtol=1e-6, atol=1e-7)
    det.close()


def test_cpp_flow_and_fundamental_match_python_mirror(gpulib, oracle, tmp_path):
    """sgx::OpticalFlowLK / sgx::findFundamentalMat (C++ mirror of Frame.cc:445, :469-472): the same tracks and F as the Python mirror of the same C ABI (which the parity tests pin to the oracle)."""
    from sg_slam_amd import synth
    from sg_slam_amd.flow import OpticalFlowLK, find_fundamental_mat
    L = synth.LayeredStream(seed=1234)
    prev, cur = L.frame(20)[0], L.frame(21)[0]
    k, _ = oracle.orb_extract(cur)
    pts = np.stack([k['x'], k['y']], 1).astype('f4')[:400]
    fc = tmp_path / 'cur.raw'; fp = tmp_path / 'prev.raw'; fb = tmp_path / 'pts.bin'
    cur.tofile(fc); prev.tofile(fp)
    with open(fb, 'wb') as fh: fh.write(np.array([len(pts)], 'i4').tobytes()); fh.write(pts.tobytes())
    out = subprocess.check_output([_exe('example_backend'), 'flow', str(fc), str(fp), str(fb)], text=True).splitlines()
    fl = OpticalFlowLK(lib=gpulib); nxt, st = fl(cur, prev, pts); fl.close()
    sel = st > 0
    ok, F, _ = find_fundamental_mat(pts[sel], nxt[sel], lib=gpulib)
    v = out[0].split()
    assert int(v[1]) == int(sel.sum()) and int(v[3]) == ok
    assert np.allclose([float(x) for x in out[1].split()[1:]], nxt.reshape(-1)[:8], rtol=0, atol=1e-6)
    assert np.allclose([float(x) for x in out[2].split()[1:]], F.reshape(-1), rtol=1e-12, atol=0)


def test_cpp_optimize_sim3_matches_oracle(gpulib, oracle, tmp_path):
    """sgx::Optimizer::OptimizeSim3 (C++ mirror): same return value, inlier count and similarity as the oracle."""
    import sim3_cases as sc
    pr = sc.make_problem(21, n=150, outliers=0.15)
    f = tmp_path / 'pairs.bin'
    with open(f, 'wb') as fh:
        fh.write(np.array([len(pr['p1c']), 0], 'i4').tobytes())
        for key in ('p1c', 'p2c', 'obs1', 'obs2', 'info1', 'info2'): fh.write(np.ascontiguousarray(pr[key], 'f4').tobytes())
        fh.write(sc.K.tobytes()); fh.write(sc.K.tobytes()); fh.write(np.ascontiguousarray(pr['S0'], 'f8').tobytes())
    out = subprocess.check_output([_exe('example_backend'), 'sim3', str(f)], text=True).splitlines()
    en, eS, einl, _ = oracle.optimize_sim3(pr['p1c'], pr['p2c'], pr['obs1'], pr['obs2'], pr['info1'], pr['info2'], sc.K, sc.K, pr['S0'], 10.0, False)
    v = out[0].split()
    assert int(v[1]) == en and int(v[3]) == int(einl.sum())
    assert sc.sim3_close([float(x) for x in out[1].split()[1:]], eS, 1e-5)


def test_cpp_vocabulary_matches_python_mirror(gpulib, oracle, tmp_path):
    """sgx::ORBVocabulary (C++ mirror): load a text vocabulary, transform(features, BowVector, FeatureVector, 4), score — same numbers as the Python mirror and the oracle."""
    import voc_cases as vc
    from sg_slam_amd.vocabulary import ORBVocabulary
    voc = vc.make_vocabulary(17, k=6, L=5)
    f = tmp_path / 'voc.txt'; vc.write_text(voc, str(f))
    d = vc.make_features(voc, 3, 900)
    fd = tmp_path / 'desc.bin'; d.tofile(fd)
    out = subprocess.check_output([_exe('example_backend'), 'voc', str(f), str(fd)], text=True).split()
    V = ORBVocabulary(gpulib); assert V.loadFromTextFile(str(f))
    ids, w, fn, _ = V.transform(d, 4)
    O = oracle.Vocabulary(path=str(f)); oi, ow, ofn, _ = O.transform(d, 4)
    assert (ids == oi).all() and (w == ow).all() and (fn == ofn).all()
    assert int(out[1]) == V.size() and int(out[3]) == len(ids) and int(out[5]) == int(ids.astype('i8').sum()) and int(out[7]) == int(fn.astype('i8').sum())
    assert float(out[9]) == float(np.sum(w[np.arange(len(w))].tolist())) or abs(float(out[9]) - 1.0) < 1e-12
    assert abs(float(out[11]) - 1.0) < 1e-12


def test_cpp_sim3_solver_matches_python_mirror(gpulib, oracle, tmp_path):
    """sgx::Sim3Solver (C++ mirror): the iterate(5) loop of LoopClosing::ComputeSim3 with the solver's own rand() replica — same call count, inliers and T12 as the Python mirror."""
    import sim3solver_cases as sc
    from sg_slam_amd.sim3solver import Sim3Solver
    X1, X2, e1, e2, R, t, s, bad = sc.make_pairs(41, 180, 0.35)
    f = tmp_path / 'pairs.bin'
    with open(f, 'wb') as fh:
        fh.write(np.array([len(X1), 1], 'i4').tobytes())
        for a in (X1, X2, e1, e2, sc.K, sc.K): fh.write(np.ascontiguousarray(a, 'f4').tobytes())
    out = subprocess.check_output([_exe('example_backend'), 's3solver', str(f)], text=True).splitlines()
    S = Sim3Solver(X1, X2, e1, e2, sc.K, sc.K, True, rand_seed=7, lib=gpulib); S.SetRansacParameters(0.99, 20, 300)
    calls = 0; T = None; nm = False
    while T is None and not nm:
        T, nm, inl, ni, run = S.iterate(5); calls += 1
    v = out[0].split()
    assert int(v[1]) == int(T is not None) and int(v[3]) == calls and int(v[5]) == ni and int(v[7]) == int(inl.sum()) and int(v[9]) == 0
    if T is not None:
        assert np.abs(np.array([float(x) for x in out[1].split()[1:]], 'f4').reshape(4, 4) - T).max() == 0


def test_cpp_pipelined_host_matches_python_binding(gpulib, tmp_path):
    """example_pipeline.cpp (sgx::TrackingPipeline over sgx_tracker_*: host frames -> upload stream -> the three-stream chain) prints the same keypoint / match /
    inlier counts and poses, bit for bit, as the Python binding of the same library fed the same frames; the trajectory follows the synthetic ground truth."""
    from sg_slam_amd import synth
    from sg_slam_amd.tracker_native import TrackerNative
    exe = os.path.join(ROOT, 'sg_slam_amd', 'host', 'example_pipeline')
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    S, F = 2, 4
    gen = synth.LayeredStream(seed=1234); offs = [3, 57]
    T0 = np.stack([gen.Tcw(o) for o in offs]).astype('f4')
    T0.reshape(S, 16).tofile(tmp_path / 'poses0.f32')
    frames = []
    for f in range(F):
        row = []
        for s, o in enumerate(offs):
            g, d, _ = gen.frame(o + f)
            bgr = np.repeat(g[..., None], 3, -1)
            bgr.tofile(tmp_path / f's{s}_f{f}.bgr'); d.tofile(tmp_path / f's{s}_f{f}.depth')
            row.append((bgr, d))
        frames.append(row)
    out = subprocess.check_output([exe, str(tmp_path), str(S), str(F)], text=True).splitlines()
    assert len(out) == S * F
    tr = TrackerNative(gpulib, S, CAM, dynamic_mask=True)
    tr.set_initial_pose(T0)
    k = 0
    for f in range(F):
        hb, hd = tr.host_buffers(f & 1)
        for s in range(S):
            hb[s, :, :640 * 3] = frames[f][s][0].reshape(480, 640 * 3); hd[s] = frames[f][s][1]
        tr.step_host(f & 1, rgb_order=True)
        r = tr.read()
        for s in range(S):
            v = out[k].split(); k += 1
            assert (int(v[0]), int(v[1])) == (f, s)
            assert int(v[2]) == r['nkeys'][s] and int(v[3]) == r['nmatch'][s] and int(v[4]) == r['ninl2'][s]
            assert (np.array([float(x) for x in v[5:]], 'f4').view(np.uint32) == r['Tcw'][s].view(np.uint32)).all()
    assert np.abs(r['Tcw'].reshape(S, 4, 4) - np.stack([gen.Tcw(o + F - 1) for o in offs])).max() < 0.05
    tr.close()
