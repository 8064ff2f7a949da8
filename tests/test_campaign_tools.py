"""The randomised differential campaigns under tools/ (emulator vs oracle) run for a few seconds each, with a fixed seed: keeps the tools working (their long runs
and totals are recorded in DESIGN.md; the solver campaigns can report benign degenerate problems, so the test does not draw fresh seeds)."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = '1234'


@pytest.mark.parametrize('tool,cases', [('campaign_orb.py', 12), ('campaign_orb_geometry.py', 12), ('campaign_match.py', 4), ('campaign_solvers.py', 120),
                                        ('campaign_ba_large.py', 5), ('campaign_tracker.py', 1), ('campaign_flow.py', 4)])
def test_campaign_tool_runs_clean(emu, oracle, tool, cases):
    # a fixed number of cases from a fixed seed: the same inputs on every machine
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool), SEED, '120', str(cases)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith('seed') and last.endswith('bad 0'), out.stdout[-2000:]
