"""Runs the batched ORB extraction a few times on synthetic frames (for rocprofv3 --pmc passes)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sg_slam_amd
from sg_slam_amd import synth
from sg_slam_amd.orb import ORBextractor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
from _campaign_lib import tool_lib; lib = tool_lib()
gen = synth.PlaneStream(seed=1234)
host = np.stack([gen.frame(37 * s)[0] for s in range(B)])
d = torch.from_numpy(host).cuda()
ex = ORBextractor(lib=lib, max_batch=B)
cap = ex.capacity
dk = torch.zeros((B, cap, 28), dtype=torch.uint8, device='cuda'); dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device='cuda'); dc = torch.zeros(B, dtype=torch.int32, device='cuda')
for _ in range(n):
    ex.extract_batch_dev(d, 640, B, dk, dd, dc, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print('keypoints', float(dc.float().mean()))
