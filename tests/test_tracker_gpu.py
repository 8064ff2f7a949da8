"""GPU (MI355X): the batched tracking harness stage-by-stage against the chained oracle."""
import pytest
from test_tracker_emu import run_tracker

pytestmark = pytest.mark.gpu


def test_tracker_two_streams_gpu(gpulib, oracle):
    run_tracker(gpulib, oracle, 'torch')


def test_tracker_mask_gpu(gpulib_taps):
    from test_tracker_emu import run_tracker_mask
    run_tracker_mask(gpulib_taps, 'torch')          # the synthetic affine flow stand-in (sgx_debug_flow_affine_batch_dev) is a tap


def test_tracker_full_batch_properties(gpulib):
    """A large batch (256 streams per launch; the bench runs 512): a stream's result must not depend on its slot or on the batch it runs in.  Eight distinct streams are
    tiled over the 256 slots; every copy must return the same bits as the first, slot results must equal a batch-of-one run of the same frames, and the
    trajectories must follow the synthetic ground truth."""
    import numpy as np
    import torch
    from sg_slam_amd import synth
    from sg_slam_amd.tracker import TrackerBatch
    from scenes import CAM
    S, D, NF = 256, 8, 4
    gen = synth.PlaneStream(seed=1234)
    offs = [13 * d for d in range(D)]
    frames = [[gen.frame(o + t) for o in offs] for t in range(NF)]
    T0 = np.stack([gen.Tcw(o) for o in offs])
    slot = np.arange(S) % D
    dev = lambda a: torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).cuda()

    def run(idx):
        tr = TrackerBatch(gpulib, len(idx), CAM, xp='torch')
        tr.set_initial_pose(T0[idx])
        out = []
        for t in range(NF):
            gray = np.stack([frames[t][i][0] for i in idx]); depth = np.stack([frames[t][i][1] for i in idx])
            tr.step(dev(gray), dev(depth))
            n, nm, ninl = tr.last_counts(); nml, ninl2 = tr.last_local_counts()
            out.append((n.copy(), nm.copy(), ninl.copy(), nml.copy(), ninl2.copy(), tr.last_pose().copy()))
        tr.synchronize()
        return out

    full = run(slot)
    for t in range(NF):
        for a in full[t][:5]:
            assert (a.reshape(S // D, D) == a[:D][None]).all(), t
        P = full[t][5].reshape(S // D, D, 16).view(np.uint32)
        assert (P == P[:1]).all(), t
    for d in (0, 5):
        one = run(np.array([d]))
        for t in range(NF):
            for a, b in zip(full[t][:5], one[t][:5]):
                assert a[d] == b[0], (d, t)
            assert (full[t][5][d].view(np.uint32) == one[t][5][0].view(np.uint32)).all(), (d, t)
    for d in range(D):                                        # ground truth: centimetre-level on the synthetic plane
        Tgt = gen.Tcw(offs[d] + NF - 1)
        assert np.abs(full[NF - 1][5][d].reshape(4, 4)[:3, 3] - Tgt[:3, 3]).max() < 0.02
        assert full[NF - 1][2][d] > 100


def test_tracker_lk_gpu(gpulib, oracle):
    from test_tracker_emu import run_tracker_lk
    run_tracker_lk(gpulib, oracle, 'torch', nframes=5)
