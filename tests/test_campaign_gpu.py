"""A fixed-seed slice of every differential campaign (tools/campaign_*.py) on the DEVICE build: the random images / geometries / matcher cases / pose and BA
problems / DetectionOutput cases / tracker runs that the long CPU campaigns feed to the g++ emulator also reach the hipcc code on the MI355X.
The tools run side by side (they are mostly oracle-bound on the host)."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = '4321'
SLICES = [('campaign_orb.py', 150), ('campaign_orb_geometry.py', 150), ('campaign_match.py', 120), ('campaign_solvers.py', 1200), ('campaign_ba_large.py', 30),
          ('campaign_detection_output.py', 200), ('campaign_flow.py', 60)]


def test_campaign_slices_on_device(gpulib, oracle):
    env = dict(os.environ, SGX_CAMPAIGN_LIB='device', OMP_NUM_THREADS='2', OPENBLAS_NUM_THREADS='2', MKL_NUM_THREADS='2')
    procs = [(tool, subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', tool), SEED, '150', str(cases)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
             for tool, cases in SLICES]
    for tool, p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, (tool, err[-2000:])
        last = out.strip().splitlines()[-1]
        assert last.startswith('seed') and last.endswith('bad 0'), (tool, out[-2000:])


def test_campaign_tracker_slice_on_device(gpulib, oracle):
    """tools/campaign_tracker.py's loop, in this process (the chained harness on random streams / offsets / lengths against the chained oracle)"""
    import numpy as np
    from test_tracker_emu import run_tracker
    rng = np.random.RandomState(int(SEED))
    for _ in range(3):
        ss = int(rng.randint(0, 100000)); offs = (int(rng.randint(0, 80)), int(rng.randint(0, 80))); nf = int(rng.randint(4, 8))
        run_tracker(gpulib, oracle, 'torch', stream_seed=ss, offs=offs, nframes=nf)
