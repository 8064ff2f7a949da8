"""The C++ pipelined host (sgx_tracker_*, sg_slam_amd/csrc/sgx_tracker.cpp) against the Python orchestration of the same C-ABI stages (TrackerBatch, whose every
stage the other tracker tests check against the oracle): same frames in, identical bits out — poses, keypoint / match / inlier counts, RANSAC statistics, the
packed frame records."""
import numpy as np
from scenes import CAM
from sg_slam_amd import synth
from sg_slam_amd.tracker import TrackerBatch
from sg_slam_amd.tracker_native import TrackerNative


def run_native_equals_python(lib, xp, dynamic_mask, nframes=4, S=2):
    gen = synth.LayeredStream(seed=1234) if dynamic_mask else synth.PlaneStream(seed=1234)
    offs = [3, 57][:S]
    def D(a):
        if xp != 'torch':
            return a
        import torch
        return torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).cuda()
    H = (lambda a: a.cpu().numpy()) if xp == 'torch' else (lambda a: np.asarray(a))
    py = TrackerBatch(lib, S, CAM, xp=xp, lk=dynamic_mask)
    nat = TrackerNative(lib, S, CAM, dynamic_mask=dynamic_mask, pipelined=(xp == 'torch'))
    assert nat.cap == py.cap
    T0 = np.stack([gen.Tcw(o) for o in offs])
    py.set_initial_pose(T0); nat.set_initial_pose(T0)
    held = []
    for t in range(nframes):
        fr = [gen.frame(o + t) for o in offs]
        gray = D(np.stack([f[0] for f in fr])); depth = D(np.stack([f[1] for f in fr]))
        held.append((gray, depth))
        py.step(gray, depth); nat.step(gray, depth)
        r = nat.read()
        n, nm, ninl = py.last_counts(); nml, ninl2 = py.last_local_counts()
        assert (r['nkeys'] == n).all(), t
        assert (r['Tcw'].view(np.uint32) == py.last_pose().reshape(S, 16).view(np.uint32)).all(), t
        if t > 0:
            assert (r['nmatch'] == nm).all() and (r['ninl'] == ninl).all() and (r['nmatch_local'] == nml).all() and (r['ninl2'] == ninl2).all(), t
            if dynamic_mask:
                assert (r['nkeys_raw'] == H(py.rn)).all() and (r['f_ok'] == H(py.f_ok)).all() and (r['f_stats'] == H(py.f_stats)).all(), t
        # the packed frame record of BASELINE config 5: header, keypoints, descriptors, pose of the frame just tracked
        if xp == 'torch':
            import torch
            rec = torch.zeros((S, nat.rec_bytes), dtype=torch.uint8, device='cuda')
            nat.pack_records(rec, stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
            rec = rec.cpu().numpy()
        else:
            rec = np.zeros((S, nat.rec_bytes), np.uint8); nat.pack_records(rec)
        cap = nat.cap; c = py.cur
        assert (rec[:, 0:4].copy().view(np.int32)[:, 0] == n).all() and (rec[:, 4:16] == 0).all()
        assert (rec[:, 16:16 + cap * 28] == H(py.keys[c]).reshape(S, cap * 28)).all()
        assert (rec[:, 16 + cap * 28:16 + cap * 60] == H(py.desc[c]).reshape(S, cap * 32)).all()
        assert (rec[:, 16 + cap * 60:].copy().view(np.float32) == r['Tcw']).all()
    assert np.abs(r['Tcw'].reshape(S, 4, 4) - np.stack([gen.Tcw(o + nframes - 1) for o in offs])).max() < 0.03
    nat.close()


def test_native_tracker_config2_chain_emu(emu):
    run_native_equals_python(emu, 'numpy', dynamic_mask=False, nframes=3)


def test_native_tracker_full_chain_emu(emu):
    run_native_equals_python(emu, 'numpy', dynamic_mask=True, nframes=3)


def test_tracker_groups_equal_one_pipeline_emu(emu):
    """TrackerGroups (bench.py --groups) = one pipeline, bit for bit (emulator: 2 streams as 2 groups of 1, 3 frames)"""
    from sg_slam_amd.tracker_native import TrackerGroups
    S, NF = 2, 3
    gen = synth.LayeredStream(seed=1234); offs = [5, 14]
    T0 = np.stack([gen.Tcw(o) for o in offs])
    one = TrackerNative(emu, S, CAM, dynamic_mask=True, pipelined=False); grp = TrackerGroups(emu, S, CAM, 2, dynamic_mask=True, pipelined=False)
    one.set_initial_pose(T0); grp.set_initial_pose(T0)
    for t in range(NF):
        fr = [gen.frame(o + t) for o in offs]
        gray = np.stack([f[0] for f in fr]); depth = np.stack([f[1] for f in fr])
        one.step(gray, depth); grp.step(gray, depth)
        a, b = one.read(), grp.read()
        for k in a:
            assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all(), (t, k)
    ra = np.zeros((S, one.rec_bytes), np.uint8); rb = np.zeros_like(ra)
    one.pack_records(ra); grp.pack_records(rb)
    assert (ra == rb).all()
    one.close(); grp.close()
