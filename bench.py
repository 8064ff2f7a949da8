#!/usr/bin/env python3
"""bench.py — tracked frames/s of the MI355X tracking hot path on synthetic 640x480 RGB-D streams.

One "step" = one frame from each of S independent streams on this GPU, pushed through the hot path
(stages implemented so far are listed in config.stages).  Frames are resident in HBM before the
timed region.  N>1: one process per GPU (torch.distributed / RCCL), streams are sharded across
ranks (weak scaling, no data-path collective); value = all ranks' frames / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel,
HIP-event timed inside the timed region) and `cpu_baseline` (the oracle timed on host cores).
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

# level geometry of the 640x480 / 1.2 / 8-level pyramid
LEVELS = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
PIX = [w * h for w, h in LEVELS]


def algorithmic_bytes_per_frame(nkp=1000, ncand=6000):
    """Compulsory HBM bytes per frame per kernel class."""
    return {
        'pyramid_resize': sum(PIX[:-1]) + sum(PIX[1:]),            # read levels 0..6 once, write levels 1..7
        'fast_cells': sum(PIX) + 4 * ncand,                        # read every level once, write packed candidates
        'octree': 4 * ncand + 4 * nkp,                             # read candidates, write selected
        'orient_desc': sum(PIX) + nkp * (28 + 32),                 # read every level at most once, write kp + desc
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--streams', type=int, default=64, help='independent streams per GPU (frames per step)')
    ap.add_argument('--frames', type=int, default=4, help='distinct frames kept per stream (ping-pong replay)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=120, help='frames timed on the CPU oracle')
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
    from sg_slam_amd.orb import ORBextractor
    lib = sg_slam_amd.load()

    S, T = args.streams, args.frames
    # synthetic streams: stream s of rank r = the plane stream starting at time offset 37*(r*S+s)
    gen = synth.PlaneStream(seed=1234)
    host = np.empty((T, S, 480, 640), np.uint8)
    for s in range(S):
        t0 = 37 * (rank * S + s)
        for t in range(T):
            host[t, s] = gen.frame(t0 + t)[0]
    d_frames = torch.from_numpy(host).cuda()
    order = list(range(T)) + list(range(T - 2, 0, -1)) if T > 1 else [0]      # ping-pong: motion stays continuous

    ex = ORBextractor(lib=lib, max_batch=S)
    cap = ex.capacity
    d_kps = torch.zeros((S, cap, 28), dtype=torch.uint8, device='cuda')
    d_desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device='cuda')
    d_cnt = torch.zeros(S, dtype=torch.int32, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        fr = d_frames[order[i % len(order)]]
        ex.extract_batch_dev(fr, 640, S, d_kps, d_desc, d_cnt, stream=stream)

    for i in range(args.warmup):
        step(i)
    ex.last_status(stream=stream)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    ex.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ex.profile_enable(False)
    prof = ex.profile_read()
    ex.last_status(stream=stream)
    nkp_mean = float(d_cnt.float().mean().item())
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank != 0:
        if dist: dist.destroy_process_group()
        return
    frames_total = S * args.steps * world
    fps = frames_total / dt

    # dominant kernel (largest summed HIP-event time in the timed region)
    alg = algorithmic_bytes_per_frame(nkp=int(round(nkp_mean)))
    dom = max(prof, key=lambda k: prof[k][0])
    per_kernel = {}
    for k, (ms, n) in prof.items():
        if n == 0: continue
        avg_ms = ms / n
        per_kernel[k] = {'avg_ms_per_launch': avg_ms, 'launches': n, 'alg_bytes_per_launch': alg[k] * S,
                         'achieved_GBs': alg[k] * S / (avg_ms * 1e-3) / 1e9}
    dk = per_kernel[dom]
    roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': dk['achieved_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': dk['achieved_GBs'] / HBM_PEAK_GBS, 'traffic': None,
                'avg_launch_ms': dk['avg_ms_per_launch'], 'alg_bytes_per_launch': dk['alg_bytes_per_launch'],
                'per_kernel': per_kernel,
                'orb_stage_frac': 1.96e6 * (fps / world) / 1e9 / HBM_PEAK_GBS}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as orc            # checker/baseline leg only — never the measured product path
        n = args.cpu_sample
        orc.orb_extract(host[0, 0])
        c0 = time.perf_counter()
        for i in range(n):
            orc.orb_extract(host[i % T, (i // T) % S])
        cdt = time.perf_counter() - c0
        cpu = {'value': n / cdt, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
               'sample': f'{n} frames of the same synthetic stream through oracle.orb_extract (ORB stage), 1 thread, host has {os.cpu_count()} cores'}

    out = {
        'metric': 'tracked frames/sec (640x480 RGB-D)', 'value': fps, 'unit': 'frames/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': 'Single MI355X: ORB extract+match HIP kernels, 640x480 synthetic stream, 1000 feats/frame',
                   'stages': ['orb_extract'], 'streams_per_gpu': S, 'frames_per_step': S, 'mean_keypoints': nkp_mean,
                   'nfeatures': 1000, 'nlevels': 8, 'scale_factor': 1.2, 'parallelism': f'streams-sharded x{world}'},
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    print(json.dumps(out))
    if dist: dist.destroy_process_group()


if __name__ == '__main__':
    main()
