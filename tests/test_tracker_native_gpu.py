"""GPU (MI355X): the C++ pipelined host (sgx_tracker_*) — three HIP streams chained by events inside the library — against the Python orchestration of the same
stages, with and without the detector, from device frames and from host frames through the pinned staging buffers."""
import ctypes as C
import os
import numpy as np
import pytest
from scenes import CAM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_tracker_config2_chain_gpu(gpulib):
    from test_tracker_native_emu import run_native_equals_python
    run_native_equals_python(gpulib, 'torch', dynamic_mask=False, nframes=5)


def test_native_tracker_full_chain_gpu(gpulib):
    from test_tracker_native_emu import run_native_equals_python
    run_native_equals_python(gpulib, 'torch', dynamic_mask=True, nframes=5)


def _detector(gpulib, S, person_logit=-0.5):
    from sg_slam_amd import synth
    from sg_slam_amd.detector import Detector2D
    param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
    layers = synth.parse_ncnn_param(param)
    _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=person_logit)       # a handful of "person" detections per frame
    return Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=gpulib)


def test_native_tracker_with_detector_equals_python_orchestration(gpulib):
    """Detector2D::detect on its own stream, its boxes feeding the mask of the same frame and the RANSAC selection of the next: the library-side event chain gives
    the same bits as the Python-side one (tests/test_detector_mask_gpu.py checks that one against the oracle)."""
    import torch
    from sg_slam_amd import synth
    from sg_slam_amd.capi import DetResult
    from sg_slam_amd.tracker import TrackerBatch
    from sg_slam_amd.tracker_native import TrackerNative
    S, MB, NF = 2, 100, 5
    gen = synth.PlaneStream(seed=1234); offs = [3, 57]
    T0 = np.stack([gen.Tcw(o) for o in offs])
    det_py, det_nat = _detector(gpulib, S), _detector(gpulib, S)
    py = TrackerBatch(gpulib, S, CAM, xp='torch', lk=True, max_boxes=MB); py.set_initial_pose(T0)
    nat = TrackerNative(gpulib, S, CAM, dynamic_mask=True, max_boxes=MB, detector=det_nat); nat.set_initial_pose(T0)
    sD = torch.cuda.Stream()
    res = [torch.zeros((S, C.sizeof(DetResult)), dtype=torch.uint8, device='cuda') for _ in range(2)]
    boxes = [torch.zeros((S, MB, 4), dtype=torch.float32, device='cuda') for _ in range(2)]
    nb = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]; have = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    held = []; total = 0
    for t in range(NF):
        fr = [gen.frame(o + t) for o in offs]
        d_gray = torch.from_numpy(np.stack([f[0] for f in fr])).cuda(); d_depth = torch.from_numpy(np.stack([f[1] for f in fr]).view(np.int16)).cuda()
        d_bgr = d_gray.unsqueeze(-1).expand(S, 480, 640, 3).contiguous()
        held.append((d_gray, d_depth, d_bgr))
        b = t & 1
        sD.wait_stream(torch.cuda.current_stream())
        if t >= 2: sD.wait_event(py.ev_extract[(t - 2) % 3])
        det_py.detect_batch_dev(d_bgr, 640 * 3, S, res[b], boxes[b], nb[b], MB, have[b], stream=sD.cuda_stream)
        ev[b].record(sD)
        py.step(d_gray, d_depth, mask=dict(boxes=boxes[b], nboxes=nb[b], have_dynamic=have[b], event=ev[b]))
        nat.step(d_gray, d_depth, d_bgr=d_bgr, stream=torch.cuda.current_stream().cuda_stream)
        bx = torch.zeros((MB, 4), dtype=torch.float32, device='cuda'); bn = torch.zeros(1, dtype=torch.int32, device='cuda')
        torch.cuda.current_stream().synchronize()      # the zero-fills run on torch's stream, the snapshot copies on the tracker's detector stream (non-blocking): without this the fill can land AFTER the copy
        nat.snapshot_boxes(1, bx, bn)
        r = nat.read(); py.synchronize(); sD.synchronize(); torch.cuda.synchronize()
        assert (bn.cpu().numpy()[0] == nb[b].cpu().numpy()[1]) and (bx.cpu().numpy() == boxes[b][1].cpu().numpy()).all()
        total += int(nb[b].sum().item())
        n, nm, ninl = py.last_counts(); nml, ninl2 = py.last_local_counts()
        assert (r['nkeys'] == n).all() and (r['Tcw'].view(np.uint32) == py.last_pose().reshape(S, 16).view(np.uint32)).all(), t
        if t > 0:
            assert (r['nkeys_raw'] == py.rn.cpu().numpy()).all() and (r['f_stats'] == py.f_stats.cpu().numpy()).all() and (r['ninl2'] == ninl2).all(), t
    assert total > 0
    nat.close()


def test_native_tracker_host_input_equals_device_input(gpulib):
    """sgx_tracker_step_host (pinned staging -> upload stream -> BGR2GRAY on the device -> step) against sgx_tracker_step_dev on frames converted up front."""
    import torch
    from sg_slam_amd import synth
    from sg_slam_amd.capi import _vp
    from sg_slam_amd.tracker_native import TrackerNative
    S, NF = 2, 5
    gen = synth.LayeredStream(seed=1234); offs = [3, 57]
    T0 = np.stack([gen.Tcw(o) for o in offs])
    rng = np.random.RandomState(3)
    a = TrackerNative(gpulib, S, CAM, dynamic_mask=True); b = TrackerNative(gpulib, S, CAM, dynamic_mask=True)
    a.set_initial_pose(T0); b.set_initial_pose(T0)
    held = []
    for t in range(NF):
        fr = [gen.frame(o + t) for o in offs]
        gray = np.stack([f[0] for f in fr]); depth = np.stack([f[1] for f in fr])
        # a colour image whose RGB2GRAY value is not simply one channel: gray + per-channel noise
        bgr = np.clip(gray[..., None].astype(np.int16) + rng.randint(-20, 21, gray.shape + (3,)), 0, 255).astype(np.uint8)
        hb, hd = a.host_buffers(t & 1)
        hb[:, :, :640 * 3] = bgr.reshape(S, 480, 640 * 3); hd[...] = depth
        a.step_host(t & 1, rgb_order=True)
        d_bgr = torch.from_numpy(bgr).cuda(); d_gray = torch.empty((S, 480, 640), dtype=torch.uint8, device='cuda')
        gpulib.check(gpulib.dll.sgx_frame_gray_from_color_batch_dev(S, 640, 480, _vp(d_bgr), 640 * 3, 3, 0, _vp(d_gray), 640, None))
        d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
        held.append((d_bgr, d_gray, d_depth))
        b.step(d_gray, d_depth, stream=torch.cuda.current_stream().cuda_stream)
        ra, rb = a.read(), b.read()
        for k in ra:
            assert (ra[k].view(np.uint32) == rb[k].view(np.uint32)).all(), (t, k)
    assert np.abs(ra['Tcw'].reshape(S, 4, 4) - np.stack([gen.Tcw(o + NF - 1) for o in offs])).max() < 0.05
    a.close(); b.close()


@pytest.mark.parametrize('pipelined', [False, True])
def test_host_slot_refill_contract_gpu(gpulib, pipelined):
    """sgx_tracker_host_buffers(slot) returns once the slot's pending upload has left the pinned buffers — in the non-pipelined mode too (its copies out of pinned
    memory on the null stream are asynchronous to the host).  The caller here scribbles over a slot IMMEDIATELY after asking for it again, without any other synchronisation,
    while the same frames go through a second tracker that is synchronised after every step: identical results."""
    from sg_slam_amd import synth
    from sg_slam_amd.tracker_native import TrackerNative
    S, NF = 8, 6
    gen = synth.LayeredStream(seed=1234); offs = [3 + 11 * s for s in range(S)]
    T0 = np.stack([gen.Tcw(o) for o in offs])
    a = TrackerNative(gpulib, S, CAM, dynamic_mask=True, pipelined=pipelined); b = TrackerNative(gpulib, S, CAM, dynamic_mask=True, pipelined=pipelined)
    a.set_initial_pose(T0); b.set_initial_pose(T0)
    frames = []
    for t in range(NF):
        fr = [gen.frame(o + t) for o in offs]
        frames.append((np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])))
    ref = []
    for t in range(NF):                      # b: one slot, full synchronisation after every step
        hb, hd = b.host_buffers(0)
        hb[:, :, :640 * 3] = np.repeat(frames[t][0][..., None], 3, 3).reshape(S, 480, 640 * 3); hd[...] = frames[t][1]
        b.step_host(0); ref.append(b.read())
    for t in range(NF):                      # a: refill slot t & 1 as soon as host_buffers hands it out again, then garbage into the OTHER slot's successor right away
        hb, hd = a.host_buffers(t & 1)
        hb[:, :, :640 * 3] = np.repeat(frames[t][0][..., None], 3, 3).reshape(S, 480, 640 * 3); hd[...] = frames[t][1]
        a.step_host(t & 1)
        hb2, hd2 = a.host_buffers(t & 1)     # must block until the upload just issued has been read out of the pinned buffers
        hb2[...] = 0x5a; hd2[...] = 0x5a5a
    got = a.read()
    for k in got:
        assert (got[k].view(np.uint32) == ref[-1][k].view(np.uint32)).all(), k
    a.close(); b.close()


def test_tracker_groups_equal_one_pipeline_gpu(gpulib):
    """TrackerGroups (bench.py --groups: several independent C++ pipelines over contiguous slices of a GPU's streams) returns the bits of ONE pipeline over all streams"""
    import torch
    from sg_slam_amd import synth
    from sg_slam_amd.tracker_native import TrackerNative, TrackerGroups
    S, NF = 8, 4
    gen = synth.LayeredStream(seed=1234); offs = [5 + 9 * s for s in range(S)]
    T0 = np.stack([gen.Tcw(o) for o in offs])
    one = TrackerNative(gpulib, S, CAM, dynamic_mask=True); grp = TrackerGroups(gpulib, S, CAM, 4, dynamic_mask=True)
    one.set_initial_pose(T0); grp.set_initial_pose(T0)
    held = []
    for t in range(NF):
        fr = [gen.frame(o + t) for o in offs]
        d_gray = torch.from_numpy(np.stack([f[0] for f in fr])).cuda(); d_depth = torch.from_numpy(np.stack([f[1] for f in fr]).view(np.int16)).cuda()
        held.append((d_gray, d_depth))
        st = torch.cuda.current_stream().cuda_stream
        one.step(d_gray, d_depth, stream=st); grp.step(d_gray, d_depth, stream=st)
        a, b = one.read(), grp.read()
        for k in a:
            assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all(), (t, k)
    rec_a = torch.zeros((S, one.rec_bytes), dtype=torch.uint8, device='cuda'); rec_b = torch.zeros_like(rec_a)
    one.pack_records(rec_a, stream=torch.cuda.current_stream().cuda_stream); grp.pack_records(rec_b, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (rec_a == rec_b).all()
    one.close(); grp.close()
