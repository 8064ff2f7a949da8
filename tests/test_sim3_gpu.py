"""GPU (MI355X): Optimizer::OptimizeSim3 through the C-ABI against the oracle."""
import pytest
import sim3_cases as sc

pytestmark = pytest.mark.gpu


def test_sim3_gpu(gpulib, oracle):
    sc.check_sim3(gpulib, oracle, n_cases=6)


def test_essential_graph_gpu(gpulib, oracle):
    sc.check_eg(gpulib, oracle, n_cases=4)


def test_essential_graph_full_size_properties(gpulib):
    """BASELINE config 4's size (2 000 keyframes) is beyond what the dense oracle finishes in seconds: size-independent properties instead — with exact relative measurements chi2
    collapses, every keyframe returns to the generating pose, the fixed keyframe is untouched, and two runs give the same bits."""
    import time
    import numpy as np
    from sg_slam_amd.optimizer import Optimizer
    g = sc.make_graph(7, 2000, noise=0.002)
    t0 = time.time()
    S, st = Optimizer.OptimizeEssentialGraph(g['S0'], g['fixed'], g['ei'], g['ej'], g['meas'], True, 20, lib=gpulib)
    dt = time.time() - t0
    assert st[1] > 1e-2 and st[2] < 1e-9 * st[1] and st[0] >= 3, st
    assert (S[0] == g['S0'][0]).all()
    for v in range(0, 2000, 37):
        assert sc.sim3_close(S[v], g['truth'][v], 1e-6), v
    S2, st2 = Optimizer.OptimizeEssentialGraph(g['S0'], g['fixed'], g['ei'], g['ej'], g['meas'], True, 20, lib=gpulib)
    assert (S2 == S).all() and (st2 == st).all()
    print(f'essential graph, 2000 keyframes / {len(g["ei"])} edges / 13 993 unknowns: {st[0]:.0f} LM iterations in {dt:.2f} s')
