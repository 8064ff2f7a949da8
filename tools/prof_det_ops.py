"""Per-plan-step HIP-event timing of the detector forward against each step's own roofline (max(bytes / HBM peak, flops / fp32-MFMA peak)).
usage: python tools/prof_det_ops.py [batch] > profiles/<name>.txt"""
import os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def step_roof_ms(desc, B):
    """max(bytes / HBM peak, flops / fp32 peak) of one plan step; fused blocks: the block's input (+ residual) and output tensors, the MACs of its three convolutions"""
    m = re.match(r'(pw|kxk) \S+ c(\d+)->(\d+) k(\d+) s(\d+) (dw )?(\d+)x(\d+)->(\d+)x(\d+)', desc)
    if m:
        c, oc, k, s = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)); dw = m.group(6) is not None
        h, w, ho, wo = (int(m.group(i)) for i in (7, 8, 9, 10))
        by = 4.0 * (c * h * w + oc * ho * wo) * B; fl = 2.0 * ho * wo * oc * (1 if dw else c) * k * k * B
        return max(by / 8e12, fl / 157.3e12) * 1e3
    m = re.match(r'block \S+ c(\d+)->(\d+)->(\d+) k(\d+) s(\d+) (\d+)x(\d+)->(\d+)x(\d+)', desc)
    if m:
        c, cm, oc, k, s, h, w, ho, wo = (int(m.group(i)) for i in range(1, 10))
        by = 4.0 * (c * h * w * (2 if '+res' in desc else 1) + oc * ho * wo) * B
        fl = 2.0 * (h * w * c * cm + ho * wo * cm * k * k + ho * wo * cm * oc) * B
        return max(by / 8e12, fl / 157.3e12) * 1e3
    m = re.match(r'irb \S+ c(\d+)->(\d+)->(\d+) q(\d+) k(\d+) s(\d+) (\d+)x(\d+)->(\d+)x(\d+)', desc)
    if m:      # matrix-core inverted-residual block / SSD head (sgx_det_irb.h): the block's input (+ residual) and output in HBM, the MACs of all its convolutions
        c, cm, oc, cq, k, s, h, w, ho, wo = (int(m.group(i)) for i in range(1, 11))
        noexp = ' noexp' in desc
        by = 4.0 * ((cm if noexp else c) * h * w + oc * ho * wo * (2 if '+res' in desc else 1)) * B
        fl = 2.0 * ((0 if noexp else h * w * c * cm) + ho * wo * cm * k * k + ho * wo * cm * oc + 2 * ho * wo * oc * cq) * B
        return max(by / 8e12, fl / 157.3e12) * 1e3
    m = re.match(r'se_gate \S+ c(\d+)->(\d+)->(\d+) (\d+)x(\d+)', desc)
    if m:      # squeeze-excite tail as one kernel: input, output (+ residual) in HBM, 2 x C x Cq multiply-adds per pixel
        c, cq, _, h, w = (int(m.group(i)) for i in range(1, 6))
        by = 4.0 * c * h * w * (3 if '+res' in desc else 2) * B; fl = 2.0 * 2 * c * cq * h * w * B
        return max(by / 8e12, fl / 157.3e12) * 1e3
    return 0.0


def report(B, rows):
    tot = sum(ms for _, ms in rows); troof = 0.0
    print(f'detector plan, batch {B}: {len(rows)} launches, sum of per-launch times {tot:.3f} ms  ({B / tot * 1e3:.0f} frames/s)')
    print(f'{"ms":>8} {"roof_ms":>8} {"frac":>6}  step')
    for desc, ms in rows:
        roof = step_roof_ms(desc, B); troof += roof
        print(f'{ms:8.4f} {roof:8.4f} {roof / ms if ms else 0:6.2f}  {desc}')
    print(f'sum of step rooflines {troof:.3f} ms -> plan at {troof / tot:.2f} of its roofline')


if len(sys.argv) > 2 and sys.argv[1] == '--reprice':          # re-derive the roofline columns of an existing table (no GPU): prof_det_ops.py --reprice table.txt
    lines = open(sys.argv[2]).read().splitlines()
    while lines and not lines[0].startswith('detector plan'): lines.pop(0)          # stderr noise of the GPU box ahead of the table
    B = int(re.search(r'batch (\d+)', lines[0]).group(1))
    rows = []
    for l in lines[2:]:
        m = re.match(r'\s*([\d.]+)\s+[\d.]+\s+[\d.]+\s+(\S.*)', l)
        if m: rows.append((m.group(2), float(m.group(1))))
    report(B, rows)
    sys.exit(0)

import torch
import sg_slam_amd
from sg_slam_amd.detector import Detector2D
from sg_slam_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PARAM = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
from _campaign_lib import taps_lib
lib = taps_lib()          # the tap build (include/sgx_debug.h): plan selection / per-step timing / blob read-back are not in the product library
layers = synth.parse_ncnn_param(PARAM); W, blob = synth.synth_ncnn_weights(layers)
det = Detector2D(0.9, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=B, lib=lib, legacy_kernels=bool(int(os.environ.get('SGX_PROF_LEGACY', '0'))),
                 block_fusion=bool(int(os.environ.get('SGX_PROF_BLOCKS', '0'))))
img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device='cuda')
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
report(B, det.time_ops(img, B, reps=REPS))
