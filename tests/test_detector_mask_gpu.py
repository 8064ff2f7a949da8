"""D3 on the device: the detector's person boxes of a frame reach the dynamic-feature mask of the SAME frame through an event between the detector stream and
the extraction stream (Frame.cc:478-500), and the RANSAC pair selection of the NEXT frame (Frame.cc:454-467) — no host round trip."""
import ctypes as C
import os
import numpy as np
import pytest
from scenes import CAM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tracker_with_device_detector_boxes(gpulib, oracle):
    import torch
    from sg_slam_amd import synth
    from sg_slam_amd.capi import DetResult
    from sg_slam_amd.detector import Detector2D
    from sg_slam_amd.tracker import TrackerBatch
    param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
    layers = synth.parse_ncnn_param(param)
    _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=-0.5)       # a handful of "person" detections per frame
    S, MB, NF = 2, 100, 5
    det = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=gpulib)
    gen = synth.PlaneStream(seed=1234); offs = [3, 57]
    tr = TrackerBatch(gpulib, S, CAM, xp='torch', lk=True, max_boxes=MB)
    tr.set_initial_pose(np.stack([gen.Tcw(o) for o in offs]))
    sD = torch.cuda.Stream()
    res = [torch.zeros((S, C.sizeof(DetResult)), dtype=torch.uint8, device='cuda') for _ in range(2)]
    boxes = [torch.zeros((S, MB, 4), dtype=torch.float32, device='cuda') for _ in range(2)]
    nb = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]; have = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    prev_gray = None; pre = [(False, np.zeros((0, 4), 'f4'))] * S
    total_person = 0
    for t in range(NF):
        fr = [gen.frame(o + t) for o in offs]
        gray = np.stack([f[0] for f in fr]); depth = np.stack([f[1] for f in fr])
        d_gray = torch.from_numpy(gray).cuda(); d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
        d_bgr = d_gray.unsqueeze(-1).expand(S, 480, 640, 3).contiguous()
        b = t & 1
        sD.wait_stream(torch.cuda.current_stream())
        if t >= 2: sD.wait_event(tr.ev_extract[(t - 2) % 3])
        det.detect_batch_dev(d_bgr, 640 * 3, S, res[b], boxes[b], nb[b], MB, have[b], stream=sD.cuda_stream)
        ev[b].record(sD)
        tr.step(d_gray, d_depth, mask=dict(boxes=boxes[b], nboxes=nb[b], have_dynamic=have[b], event=ev[b]))
        tr.synchronize(); sD.synchronize()
        # what the detector reported for this frame (host view of the device result structs)
        R = (DetResult * S).from_buffer_copy(res[b].cpu().numpy().tobytes())
        hb, hn, hh = boxes[b].cpu().numpy(), nb[b].cpu().numpy(), have[b].cpu().numpy()
        cur = []
        for s in range(S):
            rm = np.array([[o.x, o.y, o.w, o.h] for o in R[s].rm_boxes[:R[s].n_rm_boxes]], 'f4').reshape(-1, 4)
            assert hn[s] == len(rm) and (hb[s, :len(rm)] == rm).all() and hh[s] == R[s].have_dynamic_for_rm_feature == int(len(rm) > 0)
            cur.append((bool(hh[s]), rm)); total_person += len(rm)
        if t > 0:
            n, _, ninl = tr.last_counts()
            rn, keep, pxy, F = tr.rn.cpu().numpy(), tr.keep.cpu().numpy(), tr.prev_xy.cpu().numpy(), tr.F.cpu().numpy()
            for s in range(S):
                k, _ = oracle.orb_extract(gray[s]); pts = np.stack([k['x'], k['y']], 1)
                ref, _ = oracle.lk_pyr(gray[s], prev_gray[s], pts)
                assert (pxy[s, :rn[s]].view(np.uint32) == ref.view(np.uint32)).all()
                # the previous frame's boxes selected the RANSAC pairs (frame 1 has no "previous" detector state, like the reference)
                c, p = oracle.fm_select(pts, ref, pre[s][0], pre[s][1])
                rok, rF, _, _ = oracle.find_fundamental_ransac(c, p)
                assert rok == 1 and np.abs(F[s].reshape(3, 3) - rF).max() <= 1e-9 * np.abs(rF).max()
                # this frame's boxes set the 0.2 px / 1.0 px thresholds of the mask
                ek, restored = oracle.dynamic_mask(pts, ref, F[s].reshape(3, 3), cur[s][1], cur[s][0])
                assert (keep[s, :rn[s]].astype(bool) == ek).all()
                assert n[s] == (rn[s] if restored else ek.sum())
            Tg = tr.last_pose()
            for s in range(S):
                assert np.abs(Tg[s] - gen.Tcw(offs[s] + t)).max() < 0.03 and ninl[s] > 100
            pre = cur
        else:
            pre = [(False, np.zeros((0, 4), 'f4'))] * S
        prev_gray = gray
    assert total_person > 0          # the synthetic weights really produced person boxes, so the 0.2 px branch and the pair selection were exercised
