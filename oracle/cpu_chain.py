"""CPU baseline leg of bench.py (test infrastructure, like everything under oracle/): the oracle chained exactly like the device tracker —
orb_extract -> [calcOpticalFlowPyrLK -> pair selection -> findFundamentalMat(RANSAC) -> dynamic-feature mask + erase] -> ComputeStereoFromRGBD ->
SearchByProjection(cur,last) -> PoseOptimization -> [local-map SearchByProjection -> PoseOptimization] -> UnprojectStereo -> new map points —
on the synthetic streams.  Used in-process for the 1-core figure and as `python -m oracle.cpu_chain` workers (one per host core, frames-parallel)
for the all-cores figure.  Never part of the measured product path."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_chain(frames, depth_img, cam, Tstart, order, n, use_lm=True, use_mask=True, boxes=None, want_traj=False, restart=True, stage_times=None, detector=None):
    """frames: list/array of distinct gray frames of ONE stream (replayed in ping-pong `order`), depth_img: one u16 depth image or a list (one per distinct frame);
    returns seconds for n tracked frames
    (and the list of Tcw per frame when want_traj).  boxes (optional): per replayed frame index i a (k,4) array of person boxes (x, y, w, h) — the
    detector results of that frame; have_dynamic = k > 0.  restart: re-initialise the stream every 2*len(order) frames (bench timing) or run on.
    stage_times (optional dict): accumulates seconds per stage.  detector (optional callable gray -> anything): Detector2D::detect's forward of every frame on this core
    (pre-processing + network; timing only — the boxes used are `boxes`)."""
    from oracle import oracle as orc
    sf = orc.orb_params()['scale']; is2 = orc.orb_params()['inv_sigma2']
    orc.orb_extract(frames[0])                                   # warm-up (library load, tables)

    def map_points(fr):        # MapPoint(Pos, pMap, pFrame, idx) glue (numpy)
        Tm = fr['Tcw'].astype('f8'); Ow = (-(Tm[:3, :3].T @ Tm[:3, 3])).astype('f4')
        PO = fr['xw'] - Ow[None]; nrm = np.sqrt((PO.astype('f8') ** 2).sum(1)); nrm[nrm == 0] = 1
        mx = (nrm * sf[fr['keys']['octave']]).astype('f4')
        return dict(xw=fr['xw'], normal=(PO / nrm[:, None]).astype('f4'), min_dist=(mx / sf[-1]).astype('f4'), max_dist=mx, desc=fr['desc'],
                    skip=(fr['has_mp'] == 0).astype(np.uint8), obs=np.ones(len(mx), 'i4'))

    def cat(ds):
        return {k2: np.concatenate([d_[k2] for d_ in ds]) for k2 in ds[0]}

    def tick(name, t0):
        if stage_times is not None:
            stage_times[name] = stage_times.get(name, 0.0) + time.perf_counter() - t0

    traj = []
    c0 = time.perf_counter()
    done = 0
    while done < n:
        Tcur = np.asarray(Tstart, 'f4').copy(); ring = []; last = None; prev_gray = None
        pre_boxes = np.zeros((0, 4), 'f4'); pre_have = False
        Tl = Tcur.copy(); Tll = Tcur.copy()
        for i in range(n - done if not restart else min(len(order) * 2, n - done)):
            g = frames[order[i % len(order)]]
            dimg = depth_img[order[i % len(order)]] if isinstance(depth_img, (list, tuple)) else depth_img
            if detector is not None:
                t0 = time.perf_counter()
                detector(g)
                tick('detector_forward', t0)
            t0 = time.perf_counter()
            k, d = orc.orb_extract(g)
            tick('orb_extract', t0)
            if use_mask and prev_gray is not None:               # Frame::RmDynamicPointWithSemanticAndGeometry (Frame.cc:430-612)
                t0 = time.perf_counter()
                pts = np.stack([k['x'], k['y']], 1)
                prev_pts, _ = orc.lk_pyr(g, prev_gray, pts, acc_mode=0)            # acc_mode 0: OpenCV's x86 accumulation type (float)
                tick('lk_flow', t0); t0 = time.perf_counter()
                c, p = orc.fm_select(pts, prev_pts, pre_have, pre_boxes)
                ok, F, _, _ = orc.find_fundamental_ransac(c, p)
                tick('fundamental_ransac', t0); t0 = time.perf_counter()
                bx = np.zeros((0, 4), 'f4') if boxes is None else np.asarray(boxes[i], 'f4').reshape(-1, 4)
                have = len(bx) > 0
                if ok == 1:
                    keep, restored = orc.dynamic_mask(pts, prev_pts, F, bx, have)
                    if not restored:
                        k = k[keep]; d = d[keep]
                pre_boxes, pre_have = bx, have
                tick('dynamic_mask', t0)
            prev_gray = g
            t0 = time.perf_counter()
            ur, z = orc.compute_stereo_from_rgbd(k, dimg, cam['bf'], cam['depth_factor'])
            if last is not None and i > 0:
                if i > 1:                                        # constant-velocity prediction (Tracking.cc:463-470, :914); frame 1 has no velocity yet
                    Twc = np.eye(4, dtype='f4'); Twc[:3, :3] = Tll[:3, :3].T; Twc[:3, 3] = (-(Tll[:3, :3].T.astype('f8') @ Tll[:3, 3].astype('f8'))).astype('f4')
                    Tcur = ((Tl @ Twc).astype('f4') @ Tl).astype('f4')
                cur = dict(keys=k, desc=d, uright=ur, Tcw=Tcur)
                m, _ = orc.search_by_projection_frame(cur, last, cam, sf, th=15)
                tick('stereo+search_by_projection', t0); t0 = time.perf_counter()
                fr2 = dict(keys=k, uright=ur, has_mp=(m >= 0).astype(np.uint8), Tcw=Tcur,
                           xw=np.where((m >= 0)[:, None], last['xw'][np.maximum(m, 0)], 0).astype('f4'))
                _, Tcur, out1 = orc.pose_optimization(fr2, cam, is2)
                tick('pose_optimization', t0)
                if use_lm:
                    t0 = time.perf_counter()
                    keep = (m >= 0) & (out1 == 0)
                    merged_xw = np.where(keep[:, None], fr2['xw'], 0).astype('f4'); has2 = keep.copy()
                    if ring:
                        lm = cat(ring[-2:])
                        ml, _, _ = orc.search_by_projection_local(dict(keys=k, desc=d, uright=ur, Tcw=Tcur, mp_obs=np.where(keep, 0, -1).astype('i4')),
                                                                  lm, cam, sf, th=3.0, nnratio=0.8, viewing_cos_limit=0.5)
                        merged_xw = np.where((ml >= 0)[:, None], lm['xw'][np.maximum(ml, 0)], merged_xw).astype('f4'); has2 |= ml >= 0
                    tick('search_local', t0); t0 = time.perf_counter()
                    _, Tcur, _ = orc.pose_optimization(dict(keys=k, uright=ur, has_mp=has2.astype(np.uint8), Tcw=Tcur, xw=merged_xw), cam, is2)
                    tick('pose_optimization', t0)
                    ring.append(map_points(last))
            t0 = time.perf_counter()
            xw, has = orc.unproject_stereo(k, z, Tcur, cam)
            tick('unproject', t0)
            last = dict(keys=k, desc=d, uright=ur, Tcw=Tcur, has_mp=has, outlier=np.zeros(len(k), np.uint8), xw=xw, obs=np.zeros(len(k), 'i4'), mpdesc=d)
            Tll = Tl; Tl = np.asarray(Tcur, 'f4').copy()
            if want_traj:
                traj.append(Tl.copy())
            done += 1
    dt = time.perf_counter() - c0
    return (dt, traj) if want_traj else dt


def make_detector_fn(param_path, seed=7, person_logit=-0.5):
    """gray frame -> (loc, conf) of Detector2D::detect's forward on ONE CPU thread: from_pixels_resize + mean subtraction (numpy, integer) and the shipped graph with torch's
    float32 CPU operators (oracle.detector_oracle.TorchForward) on the harness's synthetic weights.  DetectionOutput is not included (its numpy restatement is not a fair timing)."""
    from oracle import detector_oracle as D
    layers = D.parse_param(param_path); W, _ = D.synth_weights(layers, seed=seed, person_logit=person_logit)
    tf = D.TorchForward(layers, W, threads=1)
    def fn(gray):
        return tf(D.preprocess(np.repeat(gray[:, :, None], 3, 2)))
    fn(np.zeros((480, 640), np.uint8))
    return fn


def ping_pong(T):
    return list(range(T)) + list(range(T - 2, 0, -1)) if T > 1 else [0]


def main():
    """worker: own stream (time offset 37*index), T distinct frames, n tracked frames; waits for the start file so all workers overlap; prints seconds"""
    ap = argparse.ArgumentParser()
    ap.add_argument('--index', type=int, default=0); ap.add_argument('--frames', type=int, default=6); ap.add_argument('--n', type=int, default=40)
    ap.add_argument('--no-local-map', action='store_true'); ap.add_argument('--no-mask', action='store_true'); ap.add_argument('--start-file', default='')
    ap.add_argument('--scene', default='LayeredStream')
    ap.add_argument('--detector-param', default='', help='ncnn .param: also run the detector forward of every frame (torch CPU float32, one thread; synthetic weights)')
    a = ap.parse_args()
    from sg_slam_amd import synth
    gen = getattr(synth, a.scene)(seed=1234); cam = dict(synth.TUM3)
    t0 = 37 * a.index
    fr = [gen.frame(t0 + t) for t in range(a.frames)]
    frames = [f[0] for f in fr]; depth_img = [f[1] for f in fr]
    from oracle import oracle as orc
    orc.orb_extract(frames[0])
    det_fn = make_detector_fn(a.detector_param) if a.detector_param else None
    print('READY', flush=True)
    while a.start_file and not os.path.exists(a.start_file):
        time.sleep(0.01)
    dt = run_chain(frames, depth_img, cam, gen.Tcw(t0), ping_pong(a.frames), a.n, use_lm=not a.no_local_map, use_mask=not a.no_mask, detector=det_fn)
    print(f'SECONDS {dt:.6f} {a.n}', flush=True)


if __name__ == '__main__':
    main()
