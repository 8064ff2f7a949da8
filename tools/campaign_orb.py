"""Differential campaign (CPU, minutes to hours): random images of six kinds (texture crops, low contrast, noise, half-flat, salt & pepper, blocky) through
the kernel-logic emulator and the oracle; full extraction must agree bit for bit.  usage: python tools/campaign_orb.py <seed> <seconds>
Round 1: 4 seeds x 1200 s + 3 seeds x 2400 s = 34 649 images, 0 mismatches."""
import sys, time; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from sg_slam_amd import synth
from sg_slam_amd.orb import ORBextractor
from sg_slam_amd.capi import SgxLib
from oracle import oracle as orc
from _campaign_lib import campaign_lib
lib, XP = campaign_lib()
ex = ORBextractor(lib=lib, max_batch=1)
rng = np.random.RandomState(int(sys.argv[1]))
t0 = time.time(); n = 0; bad = 0
MAXC = int(sys.argv[3]) if len(sys.argv) > 3 else None          # optional: stop after this many cases (deterministic runs)
while time.time() - t0 < float(sys.argv[2]) and (MAXC is None or n < MAXC):
    kind = rng.randint(0, 6)
    tex = synth.world_texture(int(rng.randint(0, 10000)), 1000, 800)
    y0, x0 = rng.randint(0, 300), rng.randint(0, 300)
    img = tex[y0:y0 + 480, x0:x0 + 640].astype(np.float32)
    if kind == 1: img = img * rng.uniform(0.05, 0.4) + rng.uniform(0, 150)          # low contrast
    elif kind == 2: img = img + rng.randn(480, 640) * rng.uniform(2, 40)             # noise
    elif kind == 3: img[:, rng.randint(100, 500):] = rng.randint(0, 256)            # flat half
    elif kind == 4: img = np.where(rng.rand(480, 640) < 0.5, 0, 255)               # salt & pepper
    elif kind == 5:                                                                  # blocky
        b = rng.randint(2, 12); img = np.kron(rng.randint(0, 256, (480 // b + 1, 640 // b + 1)), np.ones((b, b)))[:480, :640]
    img = np.ascontiguousarray(np.clip(img, 0, 255).astype(np.uint8))
    k, d = ex(img)
    ko, do = orc.orb_extract(img)
    ok = len(k) == len(ko) and (k == ko).all() and (d == do).all()
    n += 1
    if not ok:
        bad += 1; np.save('/tmp/orb_bad_%d_%d.npy' % (int(sys.argv[1]), n), img); print('MISMATCH kind', kind, len(k), len(ko), flush=True)
print('seed', sys.argv[1], 'frames', n, 'bad', bad, flush=True)
