"""Two identical trackers + detectors side by side on the same frames must agree bit for bit in every buffer of the extraction stage.  Round 6: the LK tracker differed for a few
keypoints whenever its kernel shared a CU with the detector's bf16 blocks — compiler-generated packed fp32 with a half select goes wrong in lanes 48-63 beside another wave's bf16
matrix products (profiles/r6_lk_priority_diagnosis.md; the library is built with -fno-slp-vectorize, tests/test_isa_rules.py and test_flow_gpu.py guard it).  Mixed stream priorities,
the first suspect, only made the two kernels meet more often; the library and TrackerBatch create all their streams at the default priority.  The comparison itself is
tools/diag_two_trackers.py (MODE=prio: the round 2-5 priorities)."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_co_running_trackers_agree_bit_for_bit():
    env = dict(os.environ); env.pop('MODE', None); env.pop('POLLUTE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'diag_two_trackers.py'), '24'], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    last = [l for l in out.stdout.splitlines() if l.startswith('reps')][-1]
    assert last.startswith('reps 24 bad 0'), out.stdout[-3000:]


def test_library_streams_have_no_priority():
    """the source-level half of the same contract (runs without a GPU)"""
    src = open(os.path.join(ROOT, 'sg_slam_amd', 'csrc', 'sgx_tracker.cpp')).read()
    assert 'sgx_getenv("SGX_TRK_PRIO") ? atoi(sgx_getenv("SGX_TRK_PRIO")) : 0;' in src
    from sg_slam_amd.tracker import TrackerBatch
    assert TrackerBatch.stream_priorities == (0, 0)
