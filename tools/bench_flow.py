#!/usr/bin/env python3
"""Stand-alone timing of the mask-input kernels (LK pyramid + track, RANSAC F) at S frames per launch on synthetic streams."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--streams', type=int, default=256); ap.add_argument('--reps', type=int, default=10)
    a = ap.parse_args()
    import torch
    import sg_slam_amd
    from sg_slam_amd import synth
    from sg_slam_amd.orb import ORBextractor
    from sg_slam_amd.flow import OpticalFlowLK, fundamental_ransac_batch_dev
    from _campaign_lib import tool_lib; lib = tool_lib()
    S = a.streams
    gen = synth.PlaneStream(seed=1234)
    fr = np.stack([np.stack([gen.frame(37 * s + t)[0] for s in range(min(S, 16))]) for t in range(2)])
    fr = np.tile(fr, (1, (S + 15) // 16, 1, 1))[:, :S]
    d = torch.from_numpy(fr).cuda()
    ex = ORBextractor(max_batch=S, lib=lib); cap = ex.capacity
    keys = torch.zeros((S, cap, 28), dtype=torch.uint8, device='cuda'); desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device='cuda'); n = torch.zeros(S, dtype=torch.int32, device='cuda')
    ex.extract_batch_dev(d[1], 640, S, keys, desc, n)
    fl = OpticalFlowLK(max_batch=S, lib=lib)
    out = torch.zeros((S, cap, 2), dtype=torch.float32, device='cuda'); st = torch.zeros((S, cap), dtype=torch.uint8, device='cuda')
    F = torch.zeros((S, 9), dtype=torch.float64, device='cuda'); ok = torch.zeros(S, dtype=torch.int32, device='cuda'); stats = torch.zeros((S, 4), dtype=torch.int32, device='cuda')
    lib.profile_enable(True)
    for r in range(a.reps + 2):
        if r == 2: lib.profile_read(reset=True)
        fl.reset()
        fl.lk_batch_dev(d[0], 640, S, keys, n, cap, out, st)
        fl.lk_batch_dev(d[1], 640, S, keys, n, cap, out, st)
        fundamental_ransac_batch_dev(lib, S, cap, keys, n, out, F, ok, stats)
    torch.cuda.synchronize()
    prof = {k: v for k, v in lib.profile_read().items() if v[1]}
    res = {k: {'ms_per_launch': ms / c, 'launches': c} for k, (ms, c) in prof.items()}
    res['mean_keypoints'] = float(n.float().mean()); res['ransac_iters_mean'] = float(stats[:, 0].float().mean()); res['streams'] = S
    print(json.dumps(res))


if __name__ == '__main__':
    main()
