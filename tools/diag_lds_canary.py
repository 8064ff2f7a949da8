"""diagnostic (round 6): an LDS canary kernel (tools/lds_pollute: every workgroup fills its LDS with a pattern and keeps re-reading it) beside ONE detector plan step launched over and over
(tap build): does a kernel write LDS that is not its own?  usage: python tools/diag_lds_canary.py <step> [canary LDS in units of 256 B = 37] [step batch = 32] [lib = taps]"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sg_slam_amd import synth
from sg_slam_amd.capi import SgxLib
from sg_slam_amd.detector import Detector2D
STEP = int(sys.argv[1]); UNITS = int(sys.argv[2]) if len(sys.argv) > 2 else 37; SB = int(sys.argv[3]) if len(sys.argv) > 3 else 32
libp = sys.argv[4] if len(sys.argv) > 4 else 'taps'
lib = SgxLib(os.path.join(ROOT, 'tests', 'taps', 'libsgx_taps.so') if libp == 'taps' else os.path.join(ROOT, libp))
can = C.CDLL(os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so'))
param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
layers = synth.parse_ncnn_param(param); _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=-0.5)
det = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=SB, lib=lib); st = torch.cuda.Stream()
img = torch.randint(0, 256, (SB, 480, 640, 3), dtype=torch.uint8, device='cuda')
det.time_ops(img, SB, reps=1)
print('suspect:', det.op_descriptions()[STEP] if STEP >= 0 else 'none', '| canary LDS', UNITS * 256, 'bytes per workgroup')
ev = (C.c_uint32 * 320)(); total = 0; c2 = (C.c_int * 8)(); tot2 = np.zeros(8, np.int64)
for r in range(40):
    assert can.lds_canary_launch(UNITS, 2048, 400, C.c_uint32(0x5a5a0000 + r)) == 0
    assert can.lds_canary2_launch(1024, 60, C.c_uint32(0x1234 + r)) == 0
    if os.environ.get('EXT_KIND'): assert can.corun_launch(1024, 2000, int(os.environ['EXT_KIND']), 8, 3) == 0
    if STEP >= 0: lib.check(lib.tap('sgx_det_debug_run_step')(det.h, C.c_void_p(img.data_ptr()), 640 * 3, SB, STEP, 6, C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    n = can.lds_canary_read(ev, 1); can.lds_canary2_read(c2, 1); tot2 += np.array(list(c2))
    if n and total < 3:
        e = np.frombuffer(ev, dtype=np.uint32).reshape(64, 5)[:min(n, 8)]
        for row in e: print('   rep %d block %d word %d (byte %d) expected %08x found %08x (as float %g) round %d' % (r, row[0], row[1], 4 * row[1], row[2], row[3], np.array([row[3]], np.uint32).view(np.float32)[0], row[4]))
    total += n
print('canary words changed by someone else:', total, '| LK-style accesses, wrong bytes by lane quarter: tile reads', tot2[:4].tolist(), 'bpermute', tot2[4:].tolist())
