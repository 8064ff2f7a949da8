#!/usr/bin/env python3
"""bench.py — tracked frames/s of the MI355X tracking hot path on synthetic 640x480 RGB-D streams.

One "step" = one frame from each of S independent streams on this GPU pushed through the whole
per-frame path (sg_slam_amd/tracker.py): ORB extract -> stereo-from-RGBD -> motion model ->
SearchByProjection(cur,last) -> PoseOptimization -> [TrackLocalMap: SearchByProjection(cur, local points) ->
PoseOptimization] -> unproject -> new map points.  Frames are resident in HBM before the
timed region.  N>1: one process per GPU (torch.distributed over RCCL); streams are sharded across ranks
(weak scaling, no data-path collective); after the timed region the per-frame records (pose + counts) are
gathered to rank 0 with one RCCL all_gather (BASELINE config 5), outside the timing.
value = all ranks' frames / max-over-ranks time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel by summed HIP-event time inside the timed
region) and `cpu_baseline` (the oracle — CPU restatement of the reference path — timed on host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
LEVELS = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
PIX = [w * h for w, h in LEVELS]


def algorithmic_bytes_per_frame(nkp=1000, ncand=6000, nmatch=600):
    """Compulsory HBM bytes per frame per kernel class."""
    return {
        'pyramid_resize': sum(PIX[:-1]) + sum(PIX[1:]),            # read levels 0..6 once, write levels 1..7
        'fast_cells': sum(PIX) + 4 * ncand,                        # read every level once, write packed candidates
        'octree': 4 * ncand + 4 * nkp,                             # read candidates, write selected
        'orient_desc': sum(PIX) + nkp * (28 + 32),                 # fused bound: read every level at most once, write kp + desc (the whole-level blur design moves ~1.9x)
        'stereo_from_rgbd': nkp * (28 + 2 + 8),                    # keypoint + one depth texel in, uright/z out
        'motion_model': 3 * 64,
        'match_project_frame': 2 * nkp * (28 + 32) + nkp * (4 + 12 + 1 + 1 + 4) + nkp * 4,   # both frames' kp+desc, uright/xw/flags, match out
        'pose_opt': nkp * (28 + 4 + 4) + nmatch * 12 + nkp + 64,   # keypoints, uright, match index, matched map points, outlier flags, pose
        'unproject': nkp * (28 + 4 + 12 + 1),
        'match_project_local': nkp * (28 + 32 + 4 + 4) + 2 * nkp * (12 + 12 + 4 + 4 + 32 + 4 + 1) + nkp * 4 + 2 * nkp,   # cur kp/desc/uright/obs, 2 frames of map points, match + in_view out
        'dynamic_mask': nkp * (28 + 8 + 1) / 2 + nkp * (1 + 2 * (28 + 32)) / 2,      # per launch: mask (keypoint + prev point in, flag out) | compaction (records in and out)
        'map_point_glue': nkp * (28 + 12 + 1 + 32 + 12 + 12 + 8 + 32 + 1) / 4 + nkp * (4 + 1 + 4 + 4 + 4) / 2 + 3 * nkp * 24 / 4,   # per launch: make_map_points | merge (x2) | gather
    }


def ping_pong(T):
    return list(range(T)) + list(range(T - 2, 0, -1)) if T > 1 else [0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--streams', type=int, default=256, help='independent streams per GPU (frames per step)')
    ap.add_argument('--frames', type=int, default=6, help='distinct frames kept per stream (ping-pong replay)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cpu-all-cores', action='store_true', help='skip the frames-parallel all-host-cores CPU baseline (keeps the 1-core figure)')
    ap.add_argument('--no-pipeline', action='store_true', help='single HIP stream (no overlap of extraction with match/pose-opt)')
    ap.add_argument('--no-local-map', action='store_true', help='skip the TrackLocalMap stage (local-map SearchByProjection + second PoseOptimization)')
    ap.add_argument('--groups', type=int, default=1, help='split the streams of this GPU into G independently pipelined groups (own extraction / tracking HIP streams each): '
                    'the drain of one group between kernels overlaps the body of another')
    ap.add_argument('--detector', action='store_true', help='also run Detector2D::detect (MobileNetV3-SSDLite forward + DetectionOutput + filtering, all on the device; synthetic weights) on '
                    'every frame, on a third HIP stream; its boxes land in device arrays of the mask stage\'s layout, but the mask keeps using the synthetic person box '
                    '(random-weight detections would erase random features)')
    ap.add_argument('--no-mask', action='store_true', help='skip the dynamic-feature mask + erase stage (Frame::RmDynamicPointWithSemanticAndGeometry)')
    ap.add_argument('--cpu-sample', type=int, default=400, help='frames timed on the CPU oracle')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        print('bench.py needs a GPU (the product has no CPU path)', file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    import sg_slam_amd
    from sg_slam_amd import synth
    from sg_slam_amd.tracker import TrackerBatch
    lib = sg_slam_amd.load()
    cam = dict(synth.TUM3)

    S, T = args.streams, args.frames
    # synthetic streams: stream s of rank r = the plane stream starting at time offset 37*(r*S+s)
    gen = synth.PlaneStream(seed=1234)
    host = np.empty((T, S, 480, 640), np.uint8)
    from sg_slam_amd import dist as sdist
    t0s = sdist.stream_offsets(rank, S)
    for s in range(S):
        for t in range(T):
            host[t, s] = gen.frame(t0s[s] + t)[0]
    d_frames = torch.from_numpy(host).cuda()
    depth_val = int(round(gen.z0 * cam['depth_factor']))
    d_depth = torch.full((S, 480, 640), depth_val, dtype=torch.int16, device='cuda')      # the plane: constant raw depth (u16 bits)
    order = ping_pong(T)

    class TrackerGroups:
        """G TrackerBatch instances over contiguous slices of the S streams, stepped together (same interface as one TrackerBatch for what bench.py reads)"""
        def __init__(self, G):
            self.bounds = [(g * S // G, (g + 1) * S // G) for g in range(G)]
            self.trs = [TrackerBatch(lib, b - a, cam, xp='torch', pipelined=not args.no_pipeline, local_map=not args.no_local_map) for a, b in self.bounds]
            self.max_boxes, self.cap, self.ex = self.trs[0].max_boxes, self.trs[0].cap, self.trs[0].ex
        def set_initial_pose(self, T0):
            for (a, b), t in zip(self.bounds, self.trs): t.set_initial_pose(T0[a:b])
        def step(self, gray, depth, stream=None, mask=None):
            for (a, b), t in zip(self.bounds, self.trs):
                t.step(gray[a:b], depth[a:b], stream=stream, mask=None if mask is None else {k: v[a:b] for k, v in mask.items()})
        def synchronize(self):
            for t in self.trs: t.synchronize()
        def _cat(self, parts): return tuple(np.concatenate(x) for x in zip(*parts))
        def last_counts(self): return self._cat([t.last_counts() for t in self.trs])
        def last_local_counts(self): return self._cat([t.last_local_counts() for t in self.trs])
        def last_pose(self): return np.concatenate([t.last_pose() for t in self.trs])
        @property
        def rn(self): return torch.cat([t.rn for t in self.trs])
        @property
        def ninl(self): return torch.cat([t.ninl for t in self.trs])
        @property
        def nmatch(self): return torch.cat([t.nmatch for t in self.trs])
        @property
        def Tcw(self): return [torch.cat([t.Tcw[i] for t in self.trs]) for i in range(3)]

    tr = TrackerGroups(max(1, min(args.groups, S)))
    tr.set_initial_pose(np.stack([gen.Tcw(t0) for t0 in t0s]))
    stream = torch.cuda.current_stream().cuda_stream

    # Inputs of the dynamic-feature mask.  The reference obtains them on the host (LK flow + RANSAC F, Frame.cc:445-472; person boxes from
    # Detector2D): here they come from the synthetic ground truth (synth.flow_affine / synth.fundamental), with one "person" box per stream whose
    # content moves 6 px across the epipolar lines, so the erase path does real work (~8 % of the keypoints go).
    masks = {}
    if not args.no_mask:
        boxes = torch.zeros((S, tr.max_boxes, 4), dtype=torch.float32, device='cuda'); boxes[:, 0] = torch.tensor([200.0, 120.0, 160.0, 240.0])
        nboxes = torch.ones((S,), dtype=torch.int32, device='cuda'); have_dyn = torch.ones((S,), dtype=torch.int32, device='cuda')
        for i in range(1, len(order) + 1):
            a, b = order[(i - 1) % len(order)], order[i % len(order)]
            if (a, b) in masks: continue
            A = np.stack([synth.flow_affine(gen, t0 + b, t0 + a).reshape(6) for t0 in t0s]).astype('f4')
            F = np.stack([synth.fundamental(gen, t0 + b, t0 + a).reshape(9) for t0 in t0s])
            d = np.stack([A[:, 0] * 280 + A[:, 1] * 240 + A[:, 2] - 280, A[:, 3] * 280 + A[:, 4] * 240 + A[:, 5] - 240], 1)
            sh = 6.0 * np.stack([-d[:, 1], d[:, 0]], 1) / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
            masks[(a, b)] = dict(A=torch.from_numpy(A).cuda(), F=torch.from_numpy(F).cuda(), boxes=boxes, nboxes=nboxes, have_dynamic=have_dyn,
                                 shift=torch.from_numpy(sh.astype('f4')).cuda())

    det = None
    if args.detector:
        # BASELINE config 3: the detector forward of every frame runs beside extraction and tracking (the reference runs Detector2D::detect on its own thread).
        from sg_slam_amd.detector import Detector2D
        from oracle import detector_oracle as D_                  # only to synthesise the weight blob (the reference's .bin is absent)
        import ctypes as C_
        from sg_slam_amd.capi import _vp as _vp_
        param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
        _, blob = D_.synth_weights(D_.parse_param(param))
        det = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=lib)
        d_bgr = d_frames.unsqueeze(-1).expand(T, S, 480, 640, 3).contiguous()          # gray replicated to 3 channels (SURVEY §8(d) input 2)
        sD = torch.cuda.Stream(); sD.wait_stream(torch.cuda.current_stream())
        from sg_slam_amd.capi import DetResult as DetResult_
        d_det_res = torch.zeros((S, C_.sizeof(DetResult_)), dtype=torch.uint8, device='cuda')
        d_det_boxes = torch.zeros((S, 4, 4), dtype=torch.float32, device='cuda'); d_det_nb = torch.zeros(S, dtype=torch.int32, device='cuda'); d_det_have = torch.zeros(S, dtype=torch.int32, device='cuda')
        def det_step(i):
            det.detect_batch_dev(d_bgr[order[i % len(order)]], 640 * 3, S, d_det_res, d_det_boxes, d_det_nb, 4, d_det_have, stream=sD.cuda_stream)

    def step(i):
        if det is not None: det_step(i)
        m = masks.get((order[(i - 1) % len(order)], order[i % len(order)])) if (i > 0 and masks) else None
        tr.step(d_frames[order[i % len(order)]], d_depth, stream=stream, mask=m)

    for i in range(args.warmup):
        step(i)
    tr.synchronize()
    tr.ex.last_status(stream=stream)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    lib.profile_read(reset=True)
    lib.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    tr.synchronize()
    if det is not None: sD.synchronize()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.profile_enable(False)
    prof = lib.profile_read()
    tr.ex.last_status(stream=stream)
    nkp, nmatch, ninl = tr.last_counts()
    nmatch_local, ninl2 = tr.last_local_counts()
    n_raw = tr.rn.cpu().numpy() if not args.no_mask else nkp
    if not args.no_local_map:
        ninl = ninl2
    poses = tr.last_pose()
    # accuracy vs the synthetic ground truth of the last tracked frame (translation of the camera centre)
    last_i = args.warmup + args.steps - 1
    gt = np.stack([gen.Tcw(t0 + order[last_i % len(order)]) for t0 in t0s])
    def centre(Tm): return -np.einsum('sij,sj->si', np.transpose(Tm[:, :3, :3], (0, 2, 1)), Tm[:, :3, 3])
    err = np.linalg.norm(centre(poses.astype('f8')) - centre(gt), axis=1)
    ate_rmse = float(np.sqrt((err ** 2).mean()))
    tracked = int((ninl >= 10).sum())

    if dist:
        dt = sdist.max_over_ranks(dist, dt, 'cuda')
        # BASELINE config 5: gather per-frame records (pose + counts) over RCCL/xGMI (outside the timed region)
        allrec = sdist.gather_frame_records(dist, tr.Tcw[1], tr.ninl, tr.nmatch)
        assert allrec.shape == (world, S, 18)
        sq = sdist.sum_over_ranks(dist, [float((err ** 2).sum()), float(len(err)), float(tracked)], 'cuda')
        ate_rmse = float(np.sqrt(sq[0] / sq[1])); tracked = int(sq[2])

    if rank != 0:
        if dist: dist.destroy_process_group()
        return
    frames_total = S * args.steps * world
    fps = frames_total / dt

    alg = algorithmic_bytes_per_frame(nkp=int(round(float(n_raw.mean()))), nmatch=int(round(float(nmatch.mean()))))
    SL = S / len(tr.trs)                 # frames per launch (streams of one group)
    per_kernel = {}
    for k, (ms, n) in prof.items():
        if n == 0: continue
        avg_ms = ms / n
        per_kernel[k] = {'avg_ms_per_launch': round(avg_ms, 5), 'launches': n, 'total_ms': round(ms, 3), 'alg_bytes_per_launch': alg[k] * SL,
                         'achieved_GBs': round(alg[k] * SL / (avg_ms * 1e-3) / 1e9, 3)}
    dom = max(per_kernel, key=lambda k: per_kernel[k]['total_ms'])
    dk = per_kernel[dom]
    traffic = None
    try:        # HBM traffic of the same kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/)
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'r1_traffic.json')))
        if tj['frames_per_launch'] == SL and dom in tj['bytes_per_launch']:
            traffic = tj['bytes_per_launch'][dom]
    except Exception:
        traffic = None
    roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': dk['achieved_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': dk['achieved_GBs'] / HBM_PEAK_GBS, 'traffic': traffic,
                'avg_launch_ms': dk['avg_ms_per_launch'], 'alg_bytes_per_launch': dk['alg_bytes_per_launch'],
                'per_kernel': per_kernel,
                'orb_stage_frac': 1.96e6 * (fps / world) / 1e9 / HBM_PEAK_GBS}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        # the oracle chained exactly like the tracker (checker/baseline leg only — never the measured product path): 1 core in-process, then
        # frames-parallel on all host cores (one worker process per core, each its own stream), as SURVEY.md §8(d) asks
        from oracle import cpu_chain
        n = args.cpu_sample
        depth_img = np.full((480, 640), depth_val, np.uint16)
        use_lm = not args.no_local_map
        cdt = cpu_chain.run_chain([host[t, 0] for t in range(T)], depth_img, cam, gen.Tcw(t0s[0]), order, n, use_lm)
        cpu = {'value': n / cdt, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
               'sample': f'{n} frames of one synthetic stream through the oracle chain (orb_extract + stereo + SearchByProjection + '
                         f'PoseOptimization + {"local-map SearchByProjection + PoseOptimization + " if use_lm else ""}unproject; the mask stage and the '
                         f'LK / RANSAC of the reference are not included), 1 thread; host has {os.cpu_count()} cores'}
        if not args.no_cpu_all_cores:
            import subprocess, tempfile
            P = max(1, min(os.cpu_count() or 1, 128)); n_per = max(12, args.cpu_sample // 10)
            start = os.path.join(tempfile.mkdtemp(prefix='sgx_cpu_'), 'go')
            env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
            procs = [subprocess.Popen([sys.executable, '-m', 'oracle.cpu_chain', '--index', str(k), '--frames', str(T), '--n', str(n_per), '--start-file', start] +
                                      (['--no-local-map'] if not use_lm else []), cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(P)]
            try:
                for pr in procs:
                    pr.stdout.readline()                                 # READY (frames synthesised, library loaded)
                open(start, 'w').close()
                w0 = time.perf_counter()
                secs = []
                for pr in procs:
                    line = pr.stdout.readline().split()
                    if len(line) == 3 and line[0] == 'SECONDS': secs.append(float(line[1]))
                wall = time.perf_counter() - w0
                if len(secs) == P:
                    cpu.update({'value_all_cores': P * n_per / wall, 'cores_all': P,
                                'sample_all_cores': f'{P} worker processes (one per core, each its own stream) x {n_per} frames, started together; {P * n_per} frames / wall time'})
            finally:
                for pr in procs:
                    try: pr.wait(timeout=5)
                    except Exception: pr.kill()

    out = {
        'metric': 'tracked frames/sec (640x480 RGB-D)', 'value': fps, 'unit': 'frames/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': 'Single MI355X: ORB extract+match HIP kernels, 640x480 synthetic stream, 1000 feats/frame',
                   'detector_detect_concurrent': bool(args.detector), 'stream_groups': len(tr.trs),
                   'stages': ['orb_extract'] + ([] if args.no_mask else ['dynamic_mask+erase (LK/F inputs from synthetic ground truth)']) + ['stereo_from_rgbd', 'motion_model', 'search_by_projection(cur,last)', 'pose_optimization'] +
                             ([] if args.no_local_map else ['search_by_projection(cur,local_map th=3)', 'pose_optimization#2']) + ['unproject'] +
                             ([] if args.no_local_map else ['make_map_points']),
                   'local_map_points': 0 if args.no_local_map else 2 * tr.cap, 'mean_local_map_matches': None if args.no_local_map else float(nmatch_local.mean()),
                   'streams_per_gpu': S, 'frames_per_step': S, 'distinct_frames_per_stream': T,
                   'mean_keypoints': float(nkp.mean()), 'mean_keypoints_before_mask': float(n_raw.mean()), 'mean_matches': float(nmatch.mean()), 'mean_inliers': float(ninl.mean()),
                   'tracked_streams_last_frame': tracked, 'ate_rmse_m_vs_synthetic_gt': ate_rmse,
                   'nfeatures': 1000, 'nlevels': 8, 'scale_factor': 1.2, 'parallelism': f'streams-sharded x{world}', 'hip_streams': 1 if args.no_pipeline else 2,
                   'pose_dtype': 'f64 LM, f32 boundary'},
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    print(json.dumps(out))
    if dist: dist.destroy_process_group()


if __name__ == '__main__':
    main()
