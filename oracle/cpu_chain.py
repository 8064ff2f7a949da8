"""CPU baseline leg of bench.py (test infrastructure, like everything under oracle/): the oracle chained exactly like the device tracker —
orb_extract -> ComputeStereoFromRGBD -> SearchByProjection(cur,last) -> PoseOptimization -> [local-map SearchByProjection -> PoseOptimization]
-> UnprojectStereo -> new map points — on the synthetic streams.  Used in-process for the 1-core figure and as `python -m oracle.cpu_chain`
workers (one per host core, frames-parallel) for the all-cores figure.  Never part of the measured product path."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_chain(frames, depth_img, cam, Tstart, order, n, use_lm=True):
    """frames: list/array of distinct gray frames of ONE stream (replayed in ping-pong `order`); returns seconds for n tracked frames."""
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

    c0 = time.perf_counter()
    done = 0
    while done < n:
        Tcur = np.asarray(Tstart, 'f4').copy(); ring = []; last = None
        for i in range(min(len(order) * 2, n - done)):
            g = frames[order[i % len(order)]]
            k, d = orc.orb_extract(g)
            ur, z = orc.compute_stereo_from_rgbd(k, depth_img, cam['bf'], cam['depth_factor'])
            if last is not None and i > 0:
                cur = dict(keys=k, desc=d, uright=ur, Tcw=Tcur)
                m, _ = orc.search_by_projection_frame(cur, last, cam, sf, th=15)
                fr2 = dict(keys=k, uright=ur, has_mp=(m >= 0).astype(np.uint8), Tcw=Tcur,
                           xw=np.where((m >= 0)[:, None], last['xw'][np.maximum(m, 0)], 0).astype('f4'))
                _, Tcur, out1 = orc.pose_optimization(fr2, cam, is2)
                if use_lm:
                    keep = (m >= 0) & (out1 == 0)
                    merged_xw = np.where(keep[:, None], fr2['xw'], 0).astype('f4'); has2 = keep.copy()
                    if ring:
                        lm = cat(ring[-2:])
                        ml, _, _ = orc.search_by_projection_local(dict(keys=k, desc=d, uright=ur, Tcw=Tcur, mp_obs=np.where(keep, 0, -1).astype('i4')),
                                                                  lm, cam, sf, th=3.0, nnratio=0.8, viewing_cos_limit=0.5)
                        merged_xw = np.where((ml >= 0)[:, None], lm['xw'][np.maximum(ml, 0)], merged_xw).astype('f4'); has2 |= ml >= 0
                    _, Tcur, _ = orc.pose_optimization(dict(keys=k, uright=ur, has_mp=has2.astype(np.uint8), Tcw=Tcur, xw=merged_xw), cam, is2)
                    ring.append(map_points(last))
            xw, has = orc.unproject_stereo(k, z, Tcur, cam)
            last = dict(keys=k, desc=d, uright=ur, Tcw=Tcur, has_mp=has, outlier=np.zeros(len(k), np.uint8), xw=xw, obs=np.zeros(len(k), 'i4'), mpdesc=d)
            done += 1
    return time.perf_counter() - c0


def ping_pong(T):
    return list(range(T)) + list(range(T - 2, 0, -1)) if T > 1 else [0]


def main():
    """worker: own stream (time offset 37*index), T distinct frames, n tracked frames; waits for the start file so all workers overlap; prints seconds"""
    ap = argparse.ArgumentParser()
    ap.add_argument('--index', type=int, default=0); ap.add_argument('--frames', type=int, default=6); ap.add_argument('--n', type=int, default=40)
    ap.add_argument('--no-local-map', action='store_true'); ap.add_argument('--start-file', default='')
    a = ap.parse_args()
    from sg_slam_amd import synth
    gen = synth.PlaneStream(seed=1234); cam = dict(synth.TUM3)
    t0 = 37 * a.index
    frames = [gen.frame(t0 + t)[0] for t in range(a.frames)]
    depth_img = np.full((480, 640), int(round(gen.z0 * cam['depth_factor'])), np.uint16)
    from oracle import oracle as orc
    orc.orb_extract(frames[0])
    print('READY', flush=True)
    while a.start_file and not os.path.exists(a.start_file):
        time.sleep(0.01)
    dt = run_chain(frames, depth_img, cam, gen.Tcw(t0), ping_pong(a.frames), a.n, use_lm=not a.no_local_map)
    print(f'SECONDS {dt:.6f} {a.n}', flush=True)


if __name__ == '__main__':
    main()
