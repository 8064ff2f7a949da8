"""A 30-frame synthetic sequence written in TUM layout, read back through the loader, tracked by the device path (LK + RANSAC mask inputs, no detector) and
written as a TUM trajectory file; the same frames go through the oracle chain; the two trajectory files must agree (ATE) and follow the ground truth."""
import os
import numpy as np
import pytest
from scenes import CAM

pytestmark = pytest.mark.gpu


def test_tum_sequence_trajectory_matches_oracle(gpulib, oracle, tmp_path):
    import torch
    from oracle import cpu_chain
    from sg_slam_amd import synth, tum
    from sg_slam_amd.capi import _vp
    from sg_slam_amd.tracker import TrackerBatch
    NF = 30
    gen = synth.LayeredStream(seed=1234)
    fr = [gen.frame(20 + t) for t in range(NF)]
    stamps = [1341847980.72 + t / 30.0 for t in range(NF)]
    root = str(tmp_path)
    tum.write_sequence(root, stamps, [f[0] for f in fr], [f[1] for f in fr])
    st, rgbf, depf = tum.load_associations(os.path.join(root, 'associations.txt'))
    assert len(st) == NF
    tr = TrackerBatch(gpulib, 1, CAM, xp='torch', lk=True)
    tr.set_initial_pose(gen.Tcw(20)[None])
    poses = []; grays = []; depths = []
    for i in range(NF):
        bgr, dep = tum.load_frame(root, rgbf[i], depf[i])
        d_bgr = torch.from_numpy(bgr[None]).cuda(); d_gray = torch.empty((1, 480, 640), dtype=torch.uint8, device='cuda')
        # Tracking::GrabImageRGBD: cvtColor with the RGB weights on imread's BGR data (Camera.RGB = 1, Tracking.cc:216-217)
        gpulib.check(gpulib.dll.sgx_frame_gray_from_color_batch_dev(1, 640, 480, _vp(d_bgr), 640 * 3, 3, 0, _vp(d_gray), 640, None), 'gray')
        tr.step(d_gray, torch.from_numpy(dep[None].view(np.int16)).cuda())
        poses.append(tr.last_pose()[0].copy())
        grays.append(d_gray.cpu().numpy()[0]); depths.append(dep)
    assert (grays[3] == fr[3][0]).all()                           # gray of a three-equal-channel image is the channel itself: (4899 + 9617 + 1868) / 16384 = 1
    dev_file = os.path.join(root, 'CameraTrajectory_dev.txt'); tum.save_trajectory_tum(dev_file, st, poses)
    _, otraj = cpu_chain.run_chain(grays, depths, CAM, gen.Tcw(20), list(range(NF)), NF, use_lm=True, use_mask=True, want_traj=True, restart=False)
    orc_file = os.path.join(root, 'CameraTrajectory_oracle.txt'); tum.save_trajectory_tum(orc_file, st, otraj)
    sd, xd, qd = tum.load_trajectory_tum(dev_file); so, xo, qo = tum.load_trajectory_tum(orc_file)
    assert len(sd) == len(so) == NF and np.allclose(sd, so)
    assert tum.ate_rmse(xd, xo) < 0.003 and np.abs(xd - xo).max() < 0.01            # device vs oracle chain ("ATE vs ref")
    gt = tum.camera_centres(np.stack([gen.Tcw(20 + t) for t in range(NF)]))
    gt = (np.linalg.inv(gen.Tcw(20))[:3, :3].T @ (gt - gt[0]).T).T                    # relative to the first pose, like SaveTrajectoryTUM
    assert tum.ate_rmse(xd, gt) < 0.01 and tum.ate_rmse(xo, gt) < 0.01
