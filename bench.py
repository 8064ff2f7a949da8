#!/usr/bin/env python3
"""bench.py — tracked frames/s of the MI355X tracking hot path on 640x480 RGB-D streams.

One "step" = one frame from each of S independent streams on this GPU pushed through the whole per-frame path by the C++ pipelined host behind the C-ABI
(sgx_tracker_step_dev, sg_slam_amd/csrc/sgx_tracker.cpp — one ctypes call per step; Python only indexes the resident frame tensors):
  Detector2D::detect (MobileNetV3-SSDLite forward on the fp32 matrix cores + DetectionOutput + filtering, own HIP stream)   [--no-detector to drop]
  ORB extract -> pyramidal LK flow into the previous frame -> RANSAC fundamental matrix -> wait for the detector -> dynamic-feature mask + erase
  -> stereo-from-RGBD -> motion model -> SearchByProjection(cur,last) -> PoseOptimization -> [TrackLocalMap: SearchByProjection(cur, local points)
  -> PoseOptimization] -> unproject -> new map points.
Frames are resident in HBM before the timed region.  N>1: one process per GPU (torch.distributed over RCCL); streams are sharded across ranks
(weak scaling, no data-path collective); the per-frame records {n, keypoints, descriptors, pose} of every step are packed by one kernel and gathered to
rank 0 with one RCCL gather per step on a side stream inside the timed region (BASELINE config 5).  value = all ranks' frames / max-over-ranks time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel class by summed HIP-event time inside the timed region), `cpu_baseline` (the oracle —
CPU restatement of the reference path — timed on host cores) and `config2` (the ORB extract + match + pose-opt chain without detector / LK / RANSAC:
BASELINE configs[1]), `host_input` (the same full chain fed from pinned host buffers: BGR + depth uploaded over PCIe and converted to gray inside the timed region)
and `config4` (BASELINE configs[3]: bundle adjustment of 2 000 keyframes / 50 000 landmarks, seconds per call with a per-kernel-class roofline table).
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
MFMA_F32_PEAK_TFS = 157.3        # fp32-input MFMA = fp32 vector peak
MFMA_BF16_PEAK_TFS = 2516.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s); a bf16x3 product costs six bf16 MFMAs -> 419 TFLOP/s of fp32-equivalent flops
FP64_PEAK_TFS = 78.6
LEVELS = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
PIX = [w * h for w, h in LEVELS]
LKPIX = [640 * 480, 320 * 240, 160 * 120, 80 * 60]


def algorithmic_bytes_per_frame(nkp=1000, ncand=6000, nmatch=600):
    """Compulsory HBM bytes per frame per kernel class."""
    return {
        'pyramid_resize': sum(PIX[:-1]) + sum(PIX[1:]),            # read levels 0..6 once, write levels 1..7
        'fast_cells': sum(PIX) + 4 * ncand,                        # read every level once, write packed candidates
        'octree': 4 * ncand + 4 * nkp,                             # read candidates, write selected
        'orient_desc': sum(PIX) + nkp * (28 + 32),                 # fused bound: read every level at most once, write kp + desc
        'stereo_from_rgbd': nkp * (28 + 2 + 8),                    # keypoint + one depth texel in, uright/z out
        'motion_model': 3 * 64,
        'match_project_frame': 2 * nkp * (28 + 32) + nkp * (4 + 12 + 1 + 1 + 4) + nkp * 4,   # both frames' kp+desc, uright/xw/flags, match out
        'pose_opt': nkp * (28 + 4 + 4) + nmatch * 12 + nkp + 64,   # keypoints, uright, match index, matched map points, outlier flags, pose
        'unproject': nkp * (28 + 4 + 12 + 1),
        'match_project_local': nkp * (28 + 32 + 4 + 4) + 2 * nkp * (12 + 12 + 4 + 4 + 32 + 4 + 1) + nkp * 4 + 2 * nkp,
        'dynamic_mask': nkp * (28 + 8 + 1) / 2 + nkp * (1 + 2 * (28 + 32)) / 2,      # per launch: mask | compaction
        'map_point_glue': nkp * (28 + 12 + 1 + 32 + 12 + 12 + 8 + 32 + 1) / 4 + nkp * (4 + 1 + 4 + 4 + 4) / 2 + 3 * nkp * 24 / 4,
        'lk_pyramid': LKPIX[0] + sum(LKPIX),                       # read the frame once, write the four LK levels (incl. the kept copy of level 0)
        'lk_track': 2 * sum(LKPIX) + nkp * (28 + 9),               # both pyramids read once, keypoints in, tracked position + status out
        'fm_ransac': nkp * (28 + 8) + 80,                          # point pairs in, F + flag out
        'det_output': 2268 * (4 + 21) * 4 + 4852,                  # loc + softmax conf of every prior in, result struct out
    }


def ping_pong(T):
    return list(range(T)) + list(range(T - 2, 0, -1)) if T > 1 else [0]


def ate_pooled(est, ref):
    """est, ref: (N, S, 4, 4) Tcw; per-stream Horn alignment, pooled translational RMSE over all streams and frames.  A stream whose estimated poses are not finite (tracking
    diverged: the harness has no relocalisation) is left out of the pool and COUNTED: the JSON line carries ate_streams_excluded (an ATE that silently improves when
    tracking diverges is not a measurement)"""
    from sg_slam_amd import tum
    sq = 0.0; cnt = 0; excluded = 0
    for s in range(est.shape[1]):
        if not np.isfinite(est[:, s]).all(): excluded += 1; continue
        a, b = tum.camera_centres(est[:, s]), tum.camera_centres(ref[:, s])
        try:
            r = tum.ate_rmse(a, b)
        except np.linalg.LinAlgError:
            excluded += 1; continue
        sq += r * r * len(a); cnt += len(a)
    return float(np.sqrt(sq / max(cnt, 1))), sq, cnt, excluded


def latest_profile(name):
    """newest committed profiles/rN_<name> (the PMC / standalone passes are collected by tools/collect_profiles.sh, not inside this run)"""
    for r in ('r6', 'r5', 'r4', 'r3', 'r2'):
        f = os.path.join(ROOT, 'profiles', f'{r}_{name}')
        if os.path.exists(f): return f
    raise FileNotFoundError(name)


def needs_spawn(gpus, environ):
    """`python bench.py --gpus N` outside torchrun (no WORLD_SIZE): this process becomes the launcher of N ranks.  Under the driver's own
    `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is set and the process is a rank."""
    if 'WORLD_SIZE' in environ: return False
    return gpus > 1 or bool(environ.get('SGX_BENCH_FORCE_SPAWN'))


def spawn_command(gpus, argv, environ=None, port=None):
    """the re-exec command: one rank per GPU of this node under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)"""
    environ = dict(os.environ if environ is None else environ)
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]
    env = dict(environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SGX_BENCH_SPAWNED='1')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bench.py')] + list(argv)
    return cmd, env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None, help='GPUs of this node (default: WORLD_SIZE under torch.distributed.run, else 1)')
    ap.add_argument('--steps', type=int, default=240)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--streams', type=int, default=512, help='independent streams per GPU (frames per step)')
    ap.add_argument('--groups', type=int, default=1, help='independent pipelines per GPU: the streams of a GPU are cut into this many contiguous slices, each stepped by its own '
                    'C++ host (three HIP streams each), so one slice\'s detector graph runs beside another\'s extraction / tracking kernels; results are identical for any value')
    ap.add_argument('--scene', choices=('layered', 'dynamic'), default='layered', help='synthetic scene: two static textured layers (parallax), or the same plus an independently moving textured '
                    '"walker" at 1 m that crosses the view (sg_slam_amd.synth.DynamicStream): dynamic keypoints for LK + RANSAC + the mask to erase.  A stress scene, not the default: measured '
                    'at 512 streams (profiles/r5_scene_dynamic.txt) 27.6 k frames/s, 24 RANSAC iterations, 612 of 1 006 keypoints kept, all streams tracked but 14 cm ATE — the harness has no '
                    'relocalisation or keyframe map to recover from the frames in which walker + random person boxes cover most of the view')
    ap.add_argument('--frames', type=int, default=6, help='distinct frames kept per stream (ping-pong replay)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cpu-all-cores', action='store_true', help='skip the frames-parallel all-host-cores CPU baseline (keeps the 1-core figure)')
    ap.add_argument('--no-pipeline', action='store_true', help='single HIP stream (no overlap of extraction with match/pose-opt)')
    ap.add_argument('--no-local-map', action='store_true', help='skip the TrackLocalMap stage (local-map SearchByProjection + second PoseOptimization)')
    ap.add_argument('--no-detector', action='store_true', help='drop Detector2D::detect (the mask then sees no person boxes)')
    ap.add_argument('--no-config2', action='store_true', help='skip the secondary ORB extract + match + pose-opt measurement (BASELINE configs[1])')
    ap.add_argument('--no-config4', action='store_true', help='skip the bundle-adjustment measurement (BASELINE configs[3]: 2 000 keyframes / 50 000 landmarks)')
    ap.add_argument('--no-host-input', action='store_true', help='skip the host-input measurement (frames uploaded from pinned host memory inside the timed region)')
    ap.add_argument('--host-steps', type=int, default=40, help='timed steps of the host-input measurement')
    ap.add_argument('--config2-steps', type=int, default=200, help='timed steps of the configs[1] measurement')
    ap.add_argument('--config2-only', action='store_true', help='measure only the configs[1] chain (no detector, no LK / RANSAC; mask inputs from the synthetic ground truth)')
    ap.add_argument('--param', default=os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param'), help='ncnn .param of the detector (the graph the reference ships)')
    ap.add_argument('--bin', default='', help='ncnn .bin weights of the detector (absent from the reference tree; default: synthetic weights in .bin order)')
    ap.add_argument('--person-logit', type=float, default=-0.5, help='synthetic detector weights only: offset of the person-class logit of the calibrated synthetic network '
                    '(sg_slam_amd.synth.synth_ncnn_weights).  0: three to six person boxes per frame; -0.5 (default): one or two (measured on the bench streams: 1.4 per frame, each about a '
                    'third of the image — what a walking person covers in TUM fr3/walking_xyz — and all 512 streams keep tracking); -1: a box in every second or third frame; -3: none')
    ap.add_argument('--tum', default='', help='TUM RGB-D sequence directory (rgb/ depth/ associations.txt [groundtruth.txt]): the streams are consecutive chunks of the sequence')
    ap.add_argument('--save-trajectory', default='', help='write stream 0 of rank 0 as a TUM trajectory file (System::SaveTrajectoryTUM format)')
    ap.add_argument('--cpu-sample', type=int, default=120, help='frames timed on the CPU oracle')
    args = ap.parse_args()

    # SGX_BENCH_EMU_TEST=1 (tests/test_bench_gloo.py ONLY): the harness itself — rank set-up, stream sharding, the per-step record gather, max-over-ranks timing, the JSON line —
    # on the tests' kernel-logic emulator with the gloo backend, so that the N > 1 code of THIS file runs in a container without GPUs.  Never a measurement: the line says so.
    EMU = os.environ.get('SGX_BENCH_EMU_TEST') == '1'
    explicit_gpus = args.gpus is not None
    if args.gpus is None: args.gpus = int(os.environ.get('WORLD_SIZE', '1'))      # `torchrun --nproc-per-node N bench.py` without --gpus is a valid launch
    if needs_spawn(args.gpus, os.environ):
        # `--gpus N` used to be parsed and ignored.  Without a torchrun environment this process launches the N ranks itself and relays their output
        # (rank 0 prints the JSON line); the exit code is the launcher's.
        import subprocess
        cmd, env = spawn_command(args.gpus, sys.argv[1:])
        sys.exit(subprocess.call(cmd, env=env, cwd=ROOT))

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not EMU and not torch.cuda.is_available():
        print('bench.py needs a GPU (the product has no CPU path)', file=sys.stderr)
        sys.exit(2)
    if explicit_gpus and world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}, '
              f'or plain `python bench.py --gpus {args.gpus}`, which spawns the ranks itself)', file=sys.stderr)
        sys.exit(2)
    if not EMU:
        if local >= torch.cuda.device_count():
            print(f'bench.py: rank {rank} wants GPU {local} but this node shows {torch.cuda.device_count()} GPU(s)', file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local)
    DEV = 'cpu' if EMU else 'cuda'
    dsync = (lambda: None) if EMU else torch.cuda.synchronize
    dist = None
    if world > 1 or os.environ.get('SGX_BENCH_FORCE_DIST') or os.environ.get('SGX_BENCH_SPAWNED'):       # SGX_BENCH_FORCE_DIST=1: run the RCCL gather path with one rank too (smoke test of the multi-GPU code on a 1-GPU box)
        import torch.distributed as dist_
        dist = dist_
        if EMU: dist.init_process_group('gloo')
        else: dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    import sg_slam_amd
    from sg_slam_amd import synth, tum
    from sg_slam_amd import dist as sdist
    from sg_slam_amd.tracker_native import TrackerNative
    from sg_slam_amd.capi import _vp
    if EMU:
        from sg_slam_amd.capi import SgxLib
        lib = SgxLib(os.path.join(ROOT, 'tests', 'emu', 'libsgx_emu.so')); assert 'EMULATOR' in lib.version()
    elif os.environ.get('SGX_BENCH_TAPS_LIB') == '1':
        # A/B tool runs only (tools/ab_*.sh): the tap build of the same sources, whose SGX_* environment switches select kernel variants; the line names the library
        from sg_slam_amd.capi import SgxLib
        lib = SgxLib(os.path.join(ROOT, 'tests', 'taps', 'libsgx_taps.so')); assert lib.has_taps
    else:
        lib = sg_slam_amd.load()
    cam = dict(synth.TUM3)

    S, T = args.streams, args.frames
    # synthetic scene: a textured background plane with a second textured layer in front of it (parallax, occlusion boundaries); ground-truth poses known
    SCENE = 'DynamicStream' if args.scene == 'dynamic' else 'LayeredStream'
    gen = getattr(synth, SCENE)(seed=1234)
    t0s = sdist.stream_offsets(rank, S)
    order = ping_pong(T)
    stamps = None; gt_tum = None
    if args.tum:
        # BASELINE configs 1 / 3: a real TUM sequence.  Stream s = frames [s*T, (s+1)*T) of the sequence (consecutive chunks), replayed ping-pong like the synthetic ones.
        st_all, rgbf, depf = tum.load_associations(os.path.join(args.tum, 'associations.txt'))
        S = min(S, max(1, len(st_all) // T)); t0s = [s * T for s in range(S)]
        bgr = np.empty((T, S, 480, 640, 3), np.uint8); dep = np.empty((T, S, 480, 640), np.uint16)
        for s in range(S):
            for t in range(T):
                bgr[t, s], dep[t, s] = tum.load_frame(args.tum, rgbf[s * T + t], depf[s * T + t])
        stamps = np.array(st_all[:S * T]).reshape(S, T)
        d_bgr = torch.from_numpy(bgr).to(DEV)
        d_frames = torch.empty((T, S, 480, 640), dtype=torch.uint8, device=DEV)
        for t in range(T):      # Tracking::GrabImageRGBD's cvtColor: Camera.RGB = 1 in TUM3.yaml applies the RGB weights to imread's BGR data (Tracking.cc:216-217)
            lib.check(lib.dll.sgx_frame_gray_from_color_batch_dev(S, 640, 480, _vp(d_bgr[t]), 640 * 3, 3, 0, _vp(d_frames[t]), 640, None), 'gray')
        dsync()
        host = d_frames.cpu().numpy()
        d_depth_t = torch.from_numpy(dep.view(np.int16)).to(DEV)
        gtp = os.path.join(args.tum, 'groundtruth.txt')
        if os.path.exists(gtp):
            gt_tum = tum.load_trajectory_tum(gtp)
    else:
        host, host_depth = synth.synth_streams(SCENE, 1234, t0s, T, workers=max(1, min(32, (os.cpu_count() or 2) // (2 * world))))   # the ranks of a node share its cores
        d_frames = torch.from_numpy(host).to(DEV)
        d_depth_t = torch.from_numpy(host_depth.view(np.int16)).to(DEV)          # raw u16 depth (DepthMapFactor 5000) as int16 bits
        d_bgr = None
    def depth_of(fi): return d_depth_t[fi if d_depth_t.shape[0] > 1 else 0]
    stream = None if EMU else torch.cuda.current_stream().cuda_stream
    MB = 100                                                                                      # SGX_DET_MAX person boxes per frame

    def initial_poses():
        return np.stack([np.eye(4) for _ in t0s]) if args.tum else np.stack([gen.Tcw(t0) for t0 in t0s])

    # ------------------------------------------------------------------------------------------------ secondary: BASELINE configs[1] chain
    def run_config2(steps, warmup):
        """ORB extract -> stereo -> motion model -> match -> pose-opt (-> local map) -> unproject: BASELINE configs[1], last round's headline chain (that one also ran the
        mask + erase kernels on ground-truth flow; they are part of the full path now)"""
        tr2 = TrackerNative(lib, S, cam, pipelined=not args.no_pipeline, local_map=not args.no_local_map, dynamic_mask=False)
        tr2.set_initial_pose(initial_poses())
        def st2(i):
            tr2.step(d_frames[order[i % len(order)]], depth_of(order[i % len(order)]), stream=stream)
        for i in range(warmup): st2(i)
        tr2.synchronize(); dsync()
        if dist: dist.barrier()
        dsync()
        c0 = time.perf_counter()
        for i in range(steps): st2(warmup + i)
        tr2.synchronize(); dsync()
        if dist: dist.barrier()
        dsync()
        dt2 = time.perf_counter() - c0
        if dist: dt2 = sdist.max_over_ranks(dist, dt2, DEV)
        r2 = tr2.read(); nk, nm = r2['nkeys'], r2['nmatch']
        tr2.close()
        return {'workload': 'Single MI355X: ORB extract+match HIP kernels, 640x480 synthetic stream, 1000 feats/frame', 'value': S * steps * world / dt2, 'unit': 'frames/s',
                'ms_per_step': dt2 / steps * 1e3, 'steps': steps, 'warmup': warmup, 'mean_keypoints': float(nk.mean()), 'mean_matches': float(nm.mean()),
                'stages': 'orb_extract, stereo, motion model, SearchByProjection, PoseOptimization, local-map SearchByProjection, PoseOptimization, unproject, make_map_points; '
                          'no detector, no LK / RANSAC / mask'}

    if args.config2_only:
        c2 = run_config2(args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({'metric': 'tracked frames/sec (640x480 RGB-D)', 'value': c2['value'], 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': c2['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8/f64', 'data': 'synthetic',
                              'config': c2, 'roofline': None, 'cpu_baseline': None}))
        if dist: dist.destroy_process_group()
        return

    # ------------------------------------------------------------------------------------------------ the full per-frame path
    det = None
    if not args.no_detector:
        # Detector2D::detect of every frame on its own HIP stream inside the tracker (the reference runs it on its own thread, Detector2D::Run); its person boxes are read by the
        # mask stage of the SAME frame after an event wait (Frame.cc:478-500) and by the RANSAC pair selection of the NEXT frame (Frame.cc:454-467).
        from sg_slam_amd.detector import Detector2D
        layers = synth.parse_ncnn_param(args.param)
        if args.bin:
            blob = open(args.bin, 'rb').read(); weights_note = f'weights from {os.path.basename(args.bin)}'
        else:
            _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=args.person_logit)
            weights_note = f'synthetic weights (N(0, 2/fan_in) seed 7 + per-layer batch-norm-style calibration fold, ncnn .bin order), person-class logit {args.person_logit:+g} (the reference tree does not contain the .bin)'
        _param_text = open(args.param).read()
        make_det = lambda n: Detector2D(0.9, 0.01, param_text=_param_text, bin_bytes=blob, max_batch=n, lib=lib)
        det = make_det(S // max(1, args.groups) if args.groups > 1 else S)      # with --groups G > 1 this instance only answers metadata questions (TrackerGroups builds its own per pipeline): no S-batch of activations for it
        if d_bgr is None:
            d_bgr = d_frames.unsqueeze(-1).expand(T, S, 480, 640, 3).contiguous()          # gray replicated to 3 channels (SURVEY §8(d) input 2)
        det_gflop = 2.0 * det.gmac
        # share of the graph's multiply-adds that the plan runs as bf16x3 on the bf16 matrix pipes (k_conv_pw3 steps are marked in the plan's step descriptions); the roofline peak
        # of det_forward is the harmonic blend of the two pipes' peaks over that split ("frac recomputed against the peak of the pipe actually used")
        import re as _re
        mac3 = 0.0
        for desc, _ in ([] if det.gemm != 'bf16x3' else [(d, 0) for d in det.op_descriptions()]):
            m = _re.match(r'pw \S+ c(\d+)->(\d+) k1 s1 (\d+)x(\d+)->', desc)
            if m and desc.rstrip().endswith('bf16x3'): mac3 += int(m.group(1)) * int(m.group(2)) * int(m.group(3)) * int(m.group(4))
            m = _re.match(r'block \S+ c(\d+)->(\d+)->(\d+) k\d+ s\d+ (\d+)x(\d+)->(\d+)x(\d+)', desc)      # k_hrb (round 6): the block's expand and project convolutions are bf16x3, its depthwise is vector fp32
            if m and ' hrb' in desc: mac3 += int(m.group(1)) * int(m.group(2)) * int(m.group(4)) * int(m.group(5)) + int(m.group(2)) * int(m.group(3)) * int(m.group(6)) * int(m.group(7))
        det_bf16x3_share = mac3 / (det.gmac * 1e9)
        det_peak_tfs = 1.0 / (det_bf16x3_share / (MFMA_BF16_PEAK_TFS / 6.0) + (1.0 - det_bf16x3_share) / MFMA_F32_PEAK_TFS)
    G = max(1, args.groups); SL = S // G          # frames per kernel launch
    if G > 1:
        from sg_slam_amd.tracker_native import TrackerGroups
        tr = TrackerGroups(lib, S, cam, G, make_detector=make_det if det is not None else None, pipelined=not args.no_pipeline, local_map=not args.no_local_map, dynamic_mask=True, max_boxes=MB)
    else:
        tr = TrackerNative(lib, S, cam, pipelined=not args.no_pipeline, local_map=not args.no_local_map, dynamic_mask=True, max_boxes=MB, detector=det)
    tr.set_initial_pose(initial_poses())

    gather = sdist.FrameRecordGather(dist, S, tr.cap, DEV) if dist else None
    NSTEP = args.warmup + args.steps
    traj = torch.zeros((NSTEP, S, 16), dtype=torch.float32, device=DEV)                     # pose snapshots per step (device copies on the tracking stream)
    NBOX = min(64, NSTEP)
    box_log = torch.zeros((NBOX, MB, 4), dtype=torch.float32, device=DEV); nbox_log = torch.zeros((NBOX, 1), dtype=torch.int32, device=DEV)      # stream 0, for the oracle-chain comparison
    dsync()          # the snapshot copies run on the tracker's own (non-blocking) streams: the zero-fills above must have landed first

    def step(i):
        fi = order[i % len(order)]
        tr.step(d_frames[fi], depth_of(fi), d_bgr=d_bgr[fi] if det is not None else None, stream=stream)
        tr.snapshot_pose(traj[i])
        if det is not None and i < NBOX:
            tr.snapshot_boxes(0, box_log[i], nbox_log[i])
        if gather is not None:
            gather.submit_tracker(tr)

    for i in range(args.warmup):
        step(i)
    tr.synchronize()
    if gather is not None: gather.wait()
    tr.last_status(stream=stream)
    dsync()
    if dist: dist.barrier()
    dsync()
    lib.profile_read(reset=True)
    lib.profile_enable(True)
    gather_bytes0 = gather.bytes_moved if gather else 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    tr.synchronize()
    if gather is not None: gather.wait()
    dsync()
    if dist: dist.barrier()
    dsync()
    dt = time.perf_counter() - t0
    lib.profile_enable(False)
    prof = lib.profile_read()
    tr.last_status(stream=stream)
    res = tr.read()
    nkp, nmatch, ninl, nmatch_local, ninl2, n_raw, f_ok, f_stats = (res[k] for k in ('nkeys', 'nmatch', 'ninl', 'nmatch_local', 'ninl2', 'nkeys_raw', 'f_ok', 'f_stats'))
    if not args.no_local_map:
        ninl = ninl2
    tracked = int((ninl >= 10).sum())
    if det is not None:
        bx_last = torch.zeros((MB, 4), dtype=torch.float32, device=DEV); nb_all = torch.zeros(S, dtype=torch.int32, device=DEV)
        dsync()                                                                  # the fills above run on torch's stream, the snapshots on the tracker's detector stream
        for s_ in range(0, S, max(1, S // 64)):                                                   # person-box count of a sample of streams in the last step
            tr.snapshot_boxes(s_, bx_last, nb_all[s_:s_ + 1])
        tr.synchronize(); dsync()
        nbx = nb_all[::max(1, S // 64)].cpu().numpy()
    else:
        nbx = np.zeros(1, 'i4')

    # ---- accuracy over the WHOLE run (warm-up + timed steps): per-stream Horn-aligned ATE against the synthetic ground truth (or groundtruth.txt)
    N = NSTEP
    est = traj.cpu().numpy().reshape(N, S, 4, 4).astype('f8')
    ate_gt = None; ate_sq = 0.0; ate_cnt = 0; ate_excl = 0
    if not args.tum:
        gt = np.stack([np.stack([gen.Tcw(t0 + order[i % len(order)]) for t0 in t0s]) for i in range(N)])
        ate_gt, ate_sq, ate_cnt, ate_excl = ate_pooled(est, gt)
    elif gt_tum is not None:
        gst, gxyz, _ = gt_tum
        for s in range(S):
            st_s = np.array([stamps[s, order[i % len(order)]] for i in range(N)])
            pairs = tum.associate(st_s, gst)
            if len(pairs) >= 3:
                a = tum.camera_centres(est[[i for i, _ in pairs], s]); b = gxyz[[j for _, j in pairs]]
                r = tum.ate_rmse(a, b); ate_sq += r * r * len(a); ate_cnt += len(a)
        ate_gt = float(np.sqrt(ate_sq / ate_cnt)) if ate_cnt else None

    if dist:
        dt = sdist.max_over_ranks(dist, dt, DEV)
        sq = sdist.sum_over_ranks(dist, [ate_sq, float(ate_cnt), float(tracked), float(ate_excl)], DEV)
        ate_gt = float(np.sqrt(sq[0] / sq[1])) if sq[1] else None; tracked = int(sq[2]); ate_excl = int(sq[3])
        assert (gather.recv[0] is None and gather.recv[1] is None) == (rank != 0)      # only the destination rank holds receive buffers (gather, not all_gather)
        if rank == 0:
            last_rec = gather.unpack(gather.last())
            assert last_rec['n'].shape == (world, S) and (last_rec['n'][0] == nkp).all() and (last_rec['Tcw'][0].reshape(S, 16) == res['Tcw']).all()
    if args.save_trajectory and rank == 0:
        st0 = [stamps[0, order[i % len(order)]] if stamps is not None else float(i) / 30.0 for i in range(N)]
        tum.save_trajectory_tum(args.save_trajectory, st0, [est[i, 0] for i in range(N)])

    c2 = None
    if not args.no_config2 and not args.tum:
        del traj
        c2 = run_config2(args.config2_steps, 4)

    # ------------------------------------------------------------------------------------------------ host-input mode: the same chain fed from host memory
    host_in = None
    if not args.no_host_input and not args.tum and world == 1:
        # what a caller that holds cv::imread's images does (rgbd_tum.cc:114-115): BGR + 16-bit depth of every stream are uploaded from pinned host buffers on the tracker's
        # upload stream and converted to gray on the device (Tracking.cc:214-227) INSIDE the timed region.  The two staging slots are filled once (decode / disk are the
        # caller's); every step moves S x (921 600 + 614 400) bytes over PCIe.
        th = TrackerNative(lib, S, cam, pipelined=True, local_map=not args.no_local_map, dynamic_mask=True, max_boxes=MB, detector=det)
        th.set_initial_pose(initial_poses())
        for slot in range(2):
            hb, hd = th.host_buffers(slot)
            f = order[slot % len(order)]
            hb[:, :, :640 * 3] = np.repeat(host[f][..., None], 3, -1).reshape(S, 480, 640 * 3)
            hd[...] = host_depth[f if host_depth.shape[0] > 1 else 0]
        K = max(2, args.host_steps)
        for i in range(4): th.step_host(i & 1)
        th.synchronize(); dsync()
        h0 = time.perf_counter()
        for i in range(K): th.step_host(i & 1)
        th.synchronize(); dsync()
        dth = time.perf_counter() - h0
        rh = th.read(); th.close()
        up_bytes = S * (480 * 640 * 3 + 480 * 640 * 2)
        host_in = {'value': S * K / dth, 'unit': 'frames/s', 'ms_per_step': dth / K * 1e3, 'steps': K, 'upload_bytes_per_step': up_bytes, 'upload_GBs': up_bytes * K / dth / 1e9,
                   'pcie_gen5_x16_spec_GBs': 63.0, 'tracked_streams_last_frame': int(((rh['ninl2'] if not args.no_local_map else rh['ninl']) >= 10).sum()),
                   'note': 'full chain (detector + ORB + LK + RANSAC + mask + matching + 2 x PoseOptimization) with per-step upload of BGR (3 B/px) + depth (2 B/px) from pinned host '
                           'memory and BGR2GRAY on the device inside the timed region; staging slots pre-filled with two consecutive frames (image decode is the caller\'s)'}

    # ------------------------------------------------------------------------------------------------ BASELINE configs[3]: bundle adjustment, 2 000 keyframes / 50 000 landmarks
    c4 = None
    if not args.no_config4 and not args.tum and world == 1:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from scenes import make_big_ba_problem, CAM as BA_CAM          # the seeded generator the parity tests use (SURVEY.md §8(d) input 4)
        from sg_slam_amd.optimizer import Optimizer
        prob, Ts, poses0 = make_big_ba_problem(2000, 50000)
        ne = int(len(prob['edge_pose']))
        p1 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in prob.items()}
        t4 = time.perf_counter(); Optimizer.LocalBundleAdjustment(p1, BA_CAM, lib=lib); dt4_first = time.perf_counter() - t4      # first call: code objects + device arena
        lib.profile_read(reset=True); lib.profile_enable(True)
        t4 = time.perf_counter(); er4, st4 = Optimizer.LocalBundleAdjustment(prob, BA_CAM, lib=lib); dt4 = time.perf_counter() - t4
        lib.profile_enable(False); prof4 = lib.profile_read()
        its = int(sum(st4['iterations'])); NP4 = 6 * int(st4['free_poses'])
        pk4 = {}
        for k in ('ba_linearize', 'ba_schur', 'ba_solve', 'ba_update'):
            ms, nl = prof4.get(k, (0.0, 0))
            if nl: pk4[k] = {'launch_groups': nl, 'total_ms': round(ms, 3), 'avg_ms': round(ms / nl, 4)}
        if 'ba_linearize' in pk4:      # J^T J / J^T r block accumulation (k_ba_linearize_points + k_ba_linearize_poses): SURVEY.md §8(d): ~192 algorithmic bytes per edge
            e = pk4['ba_linearize']; e.update({'bound': 'hbm', 'alg_bytes_per_launch': 192 * ne, 'achieved_GBs': round(192 * ne / (e['avg_ms'] * 1e-3) / 1e9, 2)})
            e['frac'] = e['achieved_GBs'] / HBM_PEAK_GBS
        if 'ba_solve' in pk4:          # reduced camera system (envelope Cholesky): latency-bound column steps; no flop rate is quoted — a dense-equivalent count would price
            pk4['ba_solve'].update({'bound': 'latency (dependent column steps of the envelope factorisation)', 'solver': os.environ.get('SGX_BA_SOLVER', 'default')})      # structural zeros it never computes
        err_before = float(np.abs(poses0[:, :3, 3] - Ts[:, :3, 3]).max()); err_after = float(np.abs(prob['poses'].astype('f8')[:, :3, 3] - Ts[:, :3, 3]).max())
        c4 = {'workload': 'Single MI355X: g2o PoseOptimization/LocalBA HIP kernels, 2000 keyframes / 50k landmarks synthetic', 'value': dt4, 'unit': 's per bundle adjustment', 'higher_is_better': False,
              'seconds_first_call': dt4_first, 'keyframes': 2000, 'landmarks': 50000, 'edges': ne, 'reduced_system_unknowns': NP4, 'lm_iterations': its,
              'chi2_passes': [float(x) for x in st4['chi2']], 'edges_per_s': ne * its / dt4, 'max_abs_translation_error_before': err_before, 'max_abs_translation_error_after': err_after,
              'erased_edge_frac': float(np.asarray(er4).mean()), 'per_kernel_class': pk4,
              'note': 'sgx_local_bundle_adjustment (two LM passes + outlier classification, Optimizer.cc:453-778) on the seeded generator of tests/scenes.py; steady-state second call timed, '
                      'host flattening + upload + LM control included'}

    if rank != 0:
        if dist: dist.destroy_process_group()
        return
    frames_total = S * args.steps * world
    fps = frames_total / dt

    alg = algorithmic_bytes_per_frame(nkp=int(round(float(n_raw.mean()))), nmatch=int(round(float(nmatch.mean()))))
    insts = {}
    try:        # wave-level instruction counts per frame from the committed PMC passes (tools/collect_profiles.sh -> profiles/r2_pmc_insts.json)
        insts = json.load(open(latest_profile('pmc_insts.json')))
    except Exception:
        insts = {}
    per_kernel = {}
    for k, (ms, n) in prof.items():
        if n == 0: continue
        avg_ms = ms / n
        e = {'avg_ms_per_launch': round(avg_ms, 5), 'launches': n, 'total_ms': round(ms, 3)}
        if k == 'det_forward':
            e.update({'bound': 'mfma', 'alg_gflop_per_launch': det_gflop * SL, 'achieved_TFLOPs': round(det_gflop * SL / (avg_ms * 1e-3) / 1e3, 3)})
            e['frac'] = round(e['achieved_TFLOPs'] / det_peak_tfs, 4)
        else:
            e.update({'bound': 'hbm', 'alg_bytes_per_launch': alg[k] * SL, 'achieved_GBs': round(alg[k] * SL / (avg_ms * 1e-3) / 1e9, 3)})
            e['hbm_frac'] = round(e['achieved_GBs'] / HBM_PEAK_GBS, 4)
        ik = insts.get('kernels', {}).get(k)
        if ik:   # share of the chip's VALU issue capacity this class used while it ran: wave-VALU instructions x measured cycles per instruction / (SIMDs x clock x time)
            e['valu_frac'] = round(ik['valu_insts_per_frame'] * SL * insts['cycles_per_valu_inst'] / (1024 * insts['clock_ghz'] * 1e9 * avg_ms * 1e-3), 3)
            if 'fp64_gflop_per_frame' in ik:
                e['fp64_frac'] = round(ik['fp64_gflop_per_frame'] * SL / (avg_ms * 1e-3) / 1e3 / FP64_PEAK_TFS, 4)
            # VERDICT r4 next #4a: say what bounds the kernel.  A class whose vector-issue time is the larger share of its duration is VALU-issue bound — its `frac` is the issue
            # fraction (the actionable number: fewer instructions), with the HBM fraction kept beside it (`hbm_frac`); the others keep `frac` = HBM fraction.
            if e['bound'] == 'hbm' and e['valu_frac'] > max(0.25, 2 * e['hbm_frac']):
                e['bound'] = 'valu'; e['frac'] = e['valu_frac']
                e['valu_insts_per_frame'] = ik['valu_insts_per_frame']
        if e['bound'] == 'hbm': e['frac'] = e['hbm_frac']
        # classes that neither stream memory nor saturate vector issue: name what they wait for instead of calling them HBM-bound with a fraction of 0.002
        hint = {'pose_opt': 'fp64 issue of one wave per SIMD', 'match_project_frame': 'LDS / popcount + lock-sweep latency', 'match_project_local': 'LDS / popcount + lock-sweep latency',
                'octree': 'latency (dependent list passes)', 'fm_ransac': 'latency (replayed sequential accept rule)', 'det_output': 'latency (per-class NMS chains)'}.get(k)
        if hint and e['bound'] == 'hbm':
            e['bound'] = hint; e['frac'] = e.get('fp64_frac', e.get('valu_frac', e['hbm_frac']))
        per_kernel[k] = e
    try:        # launch durations with nothing else on the GPU (the timed region above runs three streams at once: every kernel there shares CUs with the detector graph)
        sj_path = latest_profile('standalone.json')
        sj = json.load(open(sj_path))
        if sj['frames_per_launch'] == SL:
            for k, v in sj['avg_ms_per_launch'].items():
                if k in per_kernel: per_kernel[k]['standalone_avg_ms_per_launch'] = round(v, 5)
    except Exception:
        pass
    roofline = None
    if per_kernel:          # empty only in the emulator plumbing test (no HIP events there)
        dom = max(per_kernel, key=lambda k: per_kernel[k]['total_ms'])
        dk = per_kernel[dom]
        traffic = None
        try:        # HBM traffic of the same kernel class from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), not measured in this run
            tj_path = latest_profile('traffic.json')
            tj = json.load(open(tj_path))
            if tj['frames_per_launch'] == SL and dom in tj['bytes_per_launch']:
                traffic = tj['bytes_per_launch'][dom]
        except Exception:
            traffic = None
        if dk['bound'] == 'mfma':
            roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': dk['achieved_TFLOPs'], 'peak': round(det_peak_tfs, 1), 'unit': 'TFLOP/s', 'frac': dk['achieved_TFLOPs'] / det_peak_tfs,
                        'peak_fp32_matrix': MFMA_F32_PEAK_TFS, 'frac_of_fp32_matrix_peak': dk['achieved_TFLOPs'] / MFMA_F32_PEAK_TFS, 'bf16x3_share_of_macs': round(det_bf16x3_share, 4),
                        'traffic': traffic, 'avg_launch_ms': dk['avg_ms_per_launch'], 'alg_gflop_per_launch': dk['alg_gflop_per_launch'],
                        'note': f'det_forward = pre-processing + the {det.num_kernels}-launch hipGraph of the MobileNetV3-SSDLite plan; flops = 2 x MACs of the whole graph.  Matrix products: {det.gemm} '
                                f'({det_bf16x3_share:.0%} of the MACs as bf16x3 = six v_mfma_f32_32x32x16_bf16 per product on the bf16 pipes at {MFMA_BF16_PEAK_TFS / 6:.0f} TFLOP/s fp32-equivalent; the rest — the fused '
                                'inverted-residual blocks and the short-k layers — as exact fp32 on v_mfma_f32_32x32x2_f32 or packed fp32 FMAs); peak = harmonic blend of the two pipes over that split; '
                                'per-launch rocprof table in profiles/'}
        else:
            roofline = {'bound': dk.get('bound', 'hbm'), 'kernel': dom, 'achieved': dk['achieved_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dk['achieved_GBs'] / HBM_PEAK_GBS, 'traffic': traffic,
                        'avg_launch_ms': dk['avg_ms_per_launch'], 'alg_bytes_per_launch': dk['alg_bytes_per_launch']}
        if 'standalone_avg_ms_per_launch' in dk:      # the same kernel class with nothing else on the GPU (committed profile, not measured in this run)
            sa = dk['standalone_avg_ms_per_launch']
            ach = (dk['alg_gflop_per_launch'] / (sa * 1e-3) / 1e3) if dk['bound'] == 'mfma' else (dk['alg_bytes_per_launch'] / (sa * 1e-3) / 1e9)
            roofline['standalone'] = {'avg_launch_ms': sa, 'achieved': round(ach, 3), 'frac': ach / roofline['peak'], 'source': 'profiles/' + os.path.basename(sj_path)}
        if dom == 'det_forward':
            try:        # both matrix-product schemes stand-alone, from one committed session ("report both in the line")
                gj = json.load(open(latest_profile('detector_gemm_schemes.json')))
                if gj['frames_per_launch'] == SL:
                    roofline['standalone_by_scheme'] = {k: {'avg_launch_ms': v, 'achieved': round(dk['alg_gflop_per_launch'] / (v * 1e-3) / 1e3, 3),
                                                            'frac_of_fp32_matrix_peak': dk['alg_gflop_per_launch'] / (v * 1e-3) / 1e3 / MFMA_F32_PEAK_TFS} for k, v in gj['det_forward_ms_per_launch'].items()}
                    roofline['standalone_by_scheme']['source'] = 'profiles/' + os.path.basename(latest_profile('detector_gemm_schemes.json'))
            except (OSError, KeyError, ValueError):
                pass
            # det_forward is a ~100-node hipGraph, timed as one HIP-event span; the sum of its node kernels' own durations from the committed rocprofv3 kernel statistics of this same
            # command (profiles/r2_bench_kernel_stats.csv) is reported next to it (the two agree when the graph's nodes run back to back).
            try:
                import csv
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                from pmc_classes import classify
                from pmc_classes import unclassified_share
                ks_path = latest_profile('bench_kernel_stats.csv')
                rows = list(csv.DictReader(open(ks_path)))
                share, unknown = unclassified_share(rows)
                if share > 0.01:      # a plan kernel the classifier does not know would silently shrink every per-class sum (round 3: k_irb / k_se_gate -> 0.276 instead of 0.198);
                    # reported in the line, not raised: the timed run above is valid whatever the committed bookkeeping file says (tests/test_measurement_tools.py fails on it)
                    roofline['graph_kernel_time_error'] = f'{share:.1%} of the kernel time in {os.path.basename(ks_path)} belongs to kernels without a class in tools/pmc_classes.py: {unknown}'
                    raise KeyError('unclassified kernels')
                tot_ns = sum(float(r['TotalDurationNs']) for r in rows if classify(r['Name']) == 'det_forward')
                nl = max([int(r['Calls']) for r in rows if r['Name'].startswith(('k_det_preprocess', 'k_stem_pre'))] or [0])
                if nl and SL == 512:
                    kms = tot_ns / nl / 1e6
                    roofline['graph_kernel_time'] = {'sum_of_node_kernel_ms_per_launch': round(kms, 3), 'achieved': round(dk['alg_gflop_per_launch'] / (kms * 1e-3) / 1e3, 3),
                                                     'frac': dk['alg_gflop_per_launch'] / (kms * 1e-3) / 1e3 / det_peak_tfs, 'source': 'profiles/' + os.path.basename(ks_path) + ' (rocprofv3 --kernel-trace --stats of this command at 512 streams)'}
            except (OSError, ImportError, KeyError, StopIteration):
                pass
        roofline['traffic_source'] = ('profiles/' + os.path.basename(tj_path) + ' (separate rocprofv3 --pmc passes of this command)') if traffic is not None else None
        roofline['per_kernel'] = per_kernel
        # the ORB stage (north_star: ">= 60 % HBM roofline on the ORB kernel"): 1.96 MB of algorithmic traffic per frame over its four kernel classes — measured in THIS run inside the
        # three-stream pipeline (every kernel shares the CUs with the detector graph) and, from the committed one-stream profile, with nothing else on the GPU
        ORB_K = ('pyramid_resize', 'fast_cells', 'octree', 'orient_desc')
        orb_ms = sum(per_kernel[k]['avg_ms_per_launch'] * per_kernel[k]['launches'] / args.steps for k in ORB_K if k in per_kernel)
        roofline['orb_stage'] = {'ms_per_step_in_pipeline_sum': round(orb_ms, 4), 'alg_bytes_per_frame': 1.96e6,
                                 'frac_of_hbm_peak_in_pipeline': (1.96e6 * S / (orb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if orb_ms > 0 else None}
        if insts.get('kernels') and all(k in insts['kernels'] for k in ORB_K) and orb_ms > 0:
            # the stage is bound by vector-instruction issue, not by HBM (profiles/r*_pmc_kernels.md): wave-VALU instructions of the four classes x cycles per instruction over the
            # chip's 1 024 SIMDs = the time the stage needs at 100 % issue; both fractions are reported, only this one is actionable
            issue_ms = sum(insts['kernels'][k]['valu_insts_per_frame'] for k in ORB_K) * S * insts['cycles_per_valu_inst'] / (1024 * insts['clock_ghz'] * 1e9) * 1e3
            roofline['orb_stage'].update({'bound': 'valu', 'valu_issue_ms_per_step': round(issue_ms, 4), 'frac_of_valu_issue_in_pipeline': round(issue_ms / orb_ms, 4)})
        if all('standalone_avg_ms_per_launch' in per_kernel.get(k, {}) for k in ORB_K):
            orb_sa = sum(per_kernel[k]['standalone_avg_ms_per_launch'] * per_kernel[k]['launches'] / args.steps for k in ORB_K)
            roofline['orb_stage'].update({'ms_per_step_standalone_sum': round(orb_sa, 4), 'frac_of_hbm_peak_standalone': 1.96e6 * S / (orb_sa * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          'standalone_source': 'profiles/' + os.path.basename(sj_path)})
            if 'valu_issue_ms_per_step' in roofline['orb_stage']:
                roofline['orb_stage']['frac_of_valu_issue_standalone'] = round(roofline['orb_stage']['valu_issue_ms_per_step'] / orb_sa, 4)

    cpu = None; ate_oracle = None
    if not args.no_cpu_baseline and world == 1 and not args.tum:
        # the oracle chained exactly like the tracker (checker/baseline leg only — never the measured product path): 1 core in-process, then
        # frames-parallel on all host cores (one worker process per core, each its own stream), as SURVEY.md §8(d) asks
        from oracle import cpu_chain
        n = args.cpu_sample
        depth_img = [host_depth[t, 0] for t in range(T)]
        use_lm = not args.no_local_map
        stage = {}
        # the GPU figure includes Detector2D::detect, so the CPU figure does too — its forward runs on the same core per frame (pre-processing in numpy
        # integers + the shipped graph on torch's float32 CPU operators, one thread: oracle.detector_oracle.TorchForward, checked against the numpy oracle in the tests)
        det_fn = cpu_chain.make_detector_fn(args.param, person_logit=args.person_logit) if det is not None and not args.bin else None
        cdt = cpu_chain.run_chain([host[t, 0] for t in range(T)], depth_img, cam, gen.Tcw(t0s[0]), order, n, use_lm, stage_times=stage, detector=det_fn)
        det_s = stage.get('detector_forward', 0.0)
        cpu = {'value': n / cdt, 'unit': 'frames/s', 'cores': 1, 'kind': 'port', 'value_without_detector': n / (cdt - det_s),
               'sample': f'{n} frames of one synthetic stream through the oracle chain ({"Detector2D forward (from_pixels_resize + the shipped graph, float32 CPU operators of torch, no DetectionOutput) + " if det_fn else ""}'
                         f'orb_extract + calcOpticalFlowPyrLK + findFundamentalMat RANSAC + dynamic mask + stereo + SearchByProjection + PoseOptimization + '
                         f'{"local-map SearchByProjection + PoseOptimization + " if use_lm else ""}unproject), 1 thread; host has {os.cpu_count()} cores',
               'ms_per_frame_by_stage': {k: round(v / n * 1e3, 3) for k, v in stage.items()}}
        # trajectory of the device path against the oracle chain on the same frames and the same detector boxes ("ATE vs ref"): stream 0, first M frames
        M = min(N, 32, NBOX if det is not None else N)
        if M >= 3:
            bl_h, nb_h = box_log.cpu().numpy(), nbox_log.cpu().numpy()
            bxs = [bl_h[i][:int(nb_h[i, 0])] for i in range(M)] if det is not None else None
            _, otraj = cpu_chain.run_chain([host[t, 0] for t in range(T)], depth_img, cam, gen.Tcw(t0s[0]), order, M, use_lm, boxes=bxs, want_traj=True, restart=False)
            oc = tum.camera_centres(np.stack(otraj)); dc = tum.camera_centres(est[:M, 0])
            ate_oracle = {'frames': M, 'ate_rmse_m': tum.ate_rmse(dc, oc), 'max_abs_pose_entry_diff': float(np.abs(np.stack(otraj).astype('f8') - est[:M, 0]).max()),
                          'note': 'stream 0: device trajectory vs the oracle chain run on the same frames with the same detector boxes (the oracle uses OpenCV\'s float accumulation '
                                  'order in LK, the device the exact-sum variant; RANSAC / mask decisions can differ on knife-edge points)'}
        if not args.no_cpu_all_cores:
            import subprocess, tempfile
            # one worker per PHYSICAL core: the box reports 256 logical CPUs = 128 cores x 2 threads; measured in round 4: 256 workers give 115 frames/s, 128 give 166-168
            P = max(1, min(os.cpu_count() or 1, 128)); n_per = max(12, args.cpu_sample // 10)
            start = os.path.join(tempfile.mkdtemp(prefix='sgx_cpu_'), 'go')
            env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
            procs = [subprocess.Popen([sys.executable, '-m', 'oracle.cpu_chain', '--index', str(k), '--frames', str(T), '--n', str(n_per), '--start-file', start, '--scene', SCENE] +
                                      (['--no-local-map'] if not use_lm else []) + (['--detector-param', args.param] if det_fn else []), cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(P)]
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
                                'sample_all_cores': f'{P} worker processes (one per core, each its own stream{", detector forward included" if det_fn else ""}) x {n_per} frames, started together; {P * n_per} frames / wall time'})
            finally:
                for pr in procs:
                    try: pr.wait(timeout=5)
                    except Exception: pr.kill()

    workload = ('Single MI355X: + NCNN detector fwd (MFMA convs) and dynamic-feature mask' if det is not None else 'Single MI355X: ORB extract+match HIP kernels + LK / RANSAC mask inputs') + \
               (f', TUM sequence {os.path.basename(os.path.normpath(args.tum))}' if args.tum else f', 640x480 synthetic streams ({SCENE})') + ', 1000 feats/frame'
    # arithmetic types of the path: u8 / integer fixed point (ORB, LK, Hamming matching), f32 (detector forward; its scheme is named by the detector), f64 (LM pose / BA solvers)
    DTYPE = ('u8/f32(bf16x3 matrix products)/f64' if det.gemm == 'bf16x3' else 'u8/f32/f64') if det is not None else 'u8/f64'
    out = {
        'metric': 'tracked frames/sec (640x480 RGB-D)', 'value': fps, 'unit': 'frames/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'EMULATOR PLUMBING TEST (SGX_BENCH_EMU_TEST=1): NOT A MEASUREMENT' if EMU else ('tum' if args.tum else 'synthetic'),
        'config': {'workload': workload,
                   'timed_region': ['detector_detect (forward + DetectionOutput + filtering, own stream)' if det is not None else None, 'orb_extract', 'lk_pyramid + lk_track (calcOpticalFlowPyrLK)',
                                    'fm_ransac (pair selection + findFundamentalMat)', 'wait for detector boxes', 'dynamic_mask + erase', 'stereo_from_rgbd', 'motion_model',
                                    'search_by_projection(cur,last)', 'pose_optimization'] + ([] if args.no_local_map else ['search_by_projection(cur,local_map th=3)', 'pose_optimization#2']) +
                                   ['unproject'] + ([] if args.no_local_map else ['make_map_points']) + (['gather of frame records to rank 0'] if dist else []),
                   'detector': None if det is None else {'graph': os.path.basename(args.param), 'weights': weights_note, 'gflop_per_frame': det_gflop, 'matrix_products': det.gemm, 'mean_person_boxes_last_step': float(nbx.mean()),
                                                         'boxes_feed_mask_of_same_frame_and_ransac_selection_of_next': True},
                   'local_map_points': 0 if args.no_local_map else 2 * tr.cap, 'mean_local_map_matches': None if args.no_local_map else float(nmatch_local.mean()),
                   'streams_per_gpu': S, 'frames_per_step': S, 'pipelines_per_gpu': G, 'frames_per_launch': SL, 'distinct_frames_per_stream': T,
                   'mean_keypoints': float(nkp.mean()), 'mean_keypoints_before_mask': float(n_raw.mean()), 'mean_matches': float(nmatch.mean()), 'mean_inliers': float(ninl.mean()),
                   'fundamental_ok_frac': float(f_ok.mean()), 'mean_ransac_iterations': float(f_stats[:, 0].mean()),
                   'tracked_streams_last_frame': tracked, 'trajectory_frames_per_stream': N,
                   'ate_rmse_m_vs_ground_truth': ate_gt, 'ate_streams_excluded': ate_excl, 'ate_vs_oracle_chain': ate_oracle,
                   'frame_record_gather': None if gather is None else {'collective': 'gather to rank 0 (torch.distributed over RCCL), one per step, records packed by one kernel (sgx_tracker_pack_records_dev)',
                                                                       'bytes_per_step': gather._step_bytes(), 'record_bytes': gather.rec_bytes, 'records_per_step': gather.world * S,
                                                                       'GBs_into_rank0_over_xgmi': (gather.bytes_moved - gather_bytes0) / dt / 1e9, 'inside_timed_region': True},
                   'nfeatures': 1000, 'nlevels': 8, 'scale_factor': 1.2, 'parallelism': f'streams-sharded x{world}', 'hip_streams': 1 if args.no_pipeline else (3 if det is not None else 2), 'host': 'C++ pipelined host behind the C-ABI (sgx_tracker_step_dev): one ctypes call per step',
                   'pose_dtype': 'f64 LM, f32 boundary'},
        'value_host_input': None if host_in is None else host_in['value'],      # the same step with BGR + depth uploaded from pinned host memory inside the timed region (never `value`)
        'roofline': roofline, 'cpu_baseline': cpu, 'config2': c2, 'host_input': host_in, 'config4': c4,
        'library': os.path.relpath(lib.path, ROOT) + ('' if not lib.has_taps else ' (TAP BUILD: A/B tool run, switches: ' + ' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith(('SGX_DET', 'SGX_IRB', 'SGX_TUNE', 'SGX_TRK', 'SGX_LK', 'SGX_PW', 'SGX_DW', 'SGX_FB')) and k != 'SGX_BENCH_TAPS_LIB') + ')'),
    }
    out['config']['timed_region'] = [x for x in out['config']['timed_region'] if x]
    print(json.dumps(out))
    if dist: dist.destroy_process_group()


if __name__ == '__main__':
    main()
