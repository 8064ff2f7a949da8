"""Differential campaign (CPU): ORBextractor with random image sizes (any width, incl. odd), feature budgets, pyramid scale factors / depths and FAST thresholds —
a new extractor per case, so the host-side tables (cell grids, pyramid tiles, octree classes, blur tiles, candidate capacities) are exercised as well — through the
kernel-logic emulator and the oracle; full extraction bit-identical.  Geometries the product refuses (a pyramid level too small for one FAST cell: the reference
divides by zero there) are counted, not compared.  usage: python tools/campaign_orb_geometry.py <seed> <seconds>
Round 1 (6 seeds x 600 s + 2 x 3000 s): 53 921 accepted cases, all bit-identical; 28 665 refused geometries (levels smaller than one FAST cell with deep / steep pyramids, portrait levels).
The first run of this campaign found the nIni = 0 case: the product returned such levels without keypoints and the oracle crashed like the reference would."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from sg_slam_amd import synth
from sg_slam_amd.orb import ORBextractor
from sg_slam_amd.capi import SgxLib
from oracle import oracle as orc
from _campaign_lib import campaign_lib
lib, XP = campaign_lib()
rng = np.random.RandomState(int(sys.argv[1])); t0 = time.time(); n = bad = refused = 0
tex = synth.world_texture(int(sys.argv[1]), 1500, 1000)
MAXC = int(sys.argv[3]) if len(sys.argv) > 3 else None          # optional: stop after this many cases (deterministic runs)
while time.time() - t0 < float(sys.argv[2]) and (MAXC is None or n < MAXC):
    w = int(rng.randint(160, 1301)); h = int(rng.randint(120, 801))
    nf = int(rng.choice([300, 500, 1000, 1200, 2000, 3000])); sf = float(rng.choice([1.1, 1.2, 1.3, 1.5])); nl = int(rng.randint(3, 11))
    ini = int(rng.choice([10, 20, 30])); mn = int(rng.choice([3, 7, 10]))
    x0 = int(rng.randint(0, 1500 - w)); y0 = int(rng.randint(0, 1000 - h))
    img = np.ascontiguousarray(tex[y0:y0 + h, x0:x0 + w])
    try:
        e = ORBextractor(lib=lib, nfeatures=nf, scaleFactor=sf, nlevels=nl, iniThFAST=ini, minThFAST=mn, width=w, height=h)
    except Exception as ex:
        refused += 1
        if 'unsupported' not in str(ex): bad += 1; print('CREATE FAILED', w, h, nf, sf, nl, ex, flush=True)
        continue
    try:
        k, d = e(img)
    except Exception as ex:
        bad += 1; print('EXTRACT FAILED', w, h, nf, sf, nl, ini, mn, ex, flush=True); e.close(); n += 1; continue
    e.close()
    try:
        ko, do = orc.orb_extract(img, nfeatures=nf, scale=sf, nlevels=nl, ini_th=ini, min_th=mn)
    except ValueError:
        bad += 1; print('ORACLE REFUSED what the product accepted', w, h, nf, sf, nl, flush=True); n += 1; continue
    n += 1
    if not (len(k) == len(ko) and (k == ko).all() and (d == do).all()):
        bad += 1; print('MISMATCH', w, h, nf, sf, nl, ini, mn, len(k), len(ko), flush=True)
print('seed', sys.argv[1], 'cases', n, 'refused geometries', refused, 'bad', bad, flush=True)
