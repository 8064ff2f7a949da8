"""TrackerNative — the C++ pipelined per-frame host (sg_slam_amd/csrc/sgx_tracker.cpp, C-ABI sgx_tracker_*) seen from Python: one ctypes call per step.
Mirrors Tracking::GrabImageRGBD -> Frame::Frame -> TrackWithMotionModel / TrackLocalMap for S streams in lock-step (src/sg-slam/src/Tracking.cc:206-251, :906-1013);
the stream / event orchestration that sg_slam_amd/tracker.py (TrackerBatch, kept for the kernel-logic emulator tests) does in Python lives in the library here."""
import ctypes as C
import numpy as np
from .capi import TrackerConfig, _vp
from .matcher import camera_struct


class TrackerNative:
    def __init__(self, lib, streams, cam, width=640, height=480, nfeatures=1000, th=15.0, pipelined=True, local_map=True, dynamic_mask=True, max_boxes=8, detector=None):
        self.lib, self.S, self.W, self.H = lib, int(streams), int(width), int(height)
        cfg = TrackerConfig()
        cfg.streams, cfg.width, cfg.height = self.S, width, height
        cfg.nfeatures, cfg.scale_factor, cfg.nlevels, cfg.ini_th_fast, cfg.min_th_fast = nfeatures, 1.2, 8, 20, 7          # TUM3.yaml:41-54
        cfg.cam = camera_struct(cam, width, height); cfg.depth_map_factor = float(cam['depth_factor']); cfg.th_projection = float(th)
        cfg.local_map, cfg.dynamic_mask, cfg.max_boxes, cfg.pipelined = int(local_map), int(dynamic_mask), int(max_boxes), int(pipelined)
        self.detector = detector                                  # keeps the Detector2D wrapper (and its handle) alive
        h = C.c_void_p()
        lib.check(lib.dll.sgx_tracker_create(C.byref(cfg), None if detector is None else detector.h, C.byref(h)), 'sgx_tracker_create')
        self.h = h
        self.cap = lib.dll.sgx_tracker_keypoint_capacity(h)
        self.rec_bytes = lib.dll.sgx_tracker_record_bytes(h)
        self.max_boxes = int(max_boxes)
        self.frame_idx = 0

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dll.sgx_tracker_destroy(self.h); self.h = None

    def __del__(self):
        try: self.close()
        except Exception: pass

    def set_initial_pose(self, Tcw_host):
        T = np.ascontiguousarray(Tcw_host, 'f4').reshape(self.S, 16)
        self.lib.check(self.lib.dll.sgx_tracker_set_initial_pose(self.h, _vp(T)))

    def step(self, d_gray, d_depth, d_bgr=None, gray_pitch=None, bgr_pitch=None, stream=None):
        self.lib.check(self.lib.dll.sgx_tracker_step_dev(self.h, _vp(d_gray), gray_pitch or self.W, _vp(d_depth), _vp(d_bgr), bgr_pitch or 3 * self.W,
                                                         None if stream is None else C.c_void_p(stream)), 'sgx_tracker_step_dev')
        self.frame_idx += 1

    def host_buffers(self, slot):
        """numpy views of the pinned staging buffers of `slot`: bgr (S, H, pitch) u8 — pixels in [..., :3 * W] — and depth (S, H, W) u16"""
        b, d, p = C.c_void_p(), C.c_void_p(), C.c_int()
        self.lib.check(self.lib.dll.sgx_tracker_host_buffers(self.h, slot, C.byref(b), C.byref(p), C.byref(d)), 'sgx_tracker_host_buffers')
        bgr = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), shape=(self.S, self.H, p.value))
        dep = np.ctypeslib.as_array(C.cast(d, C.POINTER(C.c_uint16)), shape=(self.S, self.H, self.W))
        return bgr, dep

    def step_host(self, slot, rgb_order=True):
        self.lib.check(self.lib.dll.sgx_tracker_step_host(self.h, slot, int(rgb_order)), 'sgx_tracker_step_host')
        self.frame_idx += 1

    def synchronize(self):
        self.lib.check(self.lib.dll.sgx_tracker_sync(self.h))

    def wait_inputs(self, steps_back=0):
        """host-blocks until the device has read the input images of the step issued `steps_back` calls ago (then they may be rewritten from the host)"""
        self.lib.check(self.lib.dll.sgx_tracker_wait_inputs(self.h, steps_back), 'sgx_tracker_wait_inputs')

    def read(self):
        S = self.S
        out = dict(Tcw=np.zeros((S, 16), 'f4'), nkeys=np.zeros(S, 'i4'), nmatch=np.zeros(S, 'i4'), ninl=np.zeros(S, 'i4'), nmatch_local=np.zeros(S, 'i4'), ninl2=np.zeros(S, 'i4'),
                   nkeys_raw=np.zeros(S, 'i4'), f_ok=np.zeros(S, 'i4'), f_stats=np.zeros((S, 4), 'i4'))
        self.lib.check(self.lib.dll.sgx_tracker_read(self.h, *[_vp(out[k]) for k in ('Tcw', 'nkeys', 'nmatch', 'ninl', 'nmatch_local', 'ninl2', 'nkeys_raw', 'f_ok', 'f_stats')]))
        return out

    def last_pose(self):
        return self.read()['Tcw'].reshape(self.S, 4, 4)

    def snapshot_pose(self, d_out):
        self.lib.check(self.lib.dll.sgx_tracker_snapshot_pose_dev(self.h, _vp(d_out)))

    def snapshot_boxes(self, stream_index, d_boxes, d_nboxes):
        self.lib.check(self.lib.dll.sgx_tracker_snapshot_boxes_dev(self.h, stream_index, _vp(d_boxes), _vp(d_nboxes)))

    def pack_records(self, d_records, stream=None):
        self.lib.check(self.lib.dll.sgx_tracker_pack_records_dev(self.h, _vp(d_records), None if stream is None else C.c_void_p(stream)))

    def frame_dev(self):
        """device pointers (ints) of the frame tracked last: n, keys, desc, Tcw, xw, has"""
        p = [C.c_void_p() for _ in range(6)]
        self.lib.check(self.lib.dll.sgx_tracker_frame_dev(self.h, *[C.byref(x) for x in p]))
        return dict(zip(('n', 'keys', 'desc', 'Tcw', 'xw', 'has'), [x.value for x in p]))

    def last_status(self, stream=None):
        self.lib.check(self.lib.dll.sgx_orb_last_status(self.lib.dll.sgx_tracker_extractor(self.h), None if stream is None else C.c_void_p(stream)))


class TrackerGroups:
    """G independent TrackerNative pipelines over contiguous slices of the S streams of a GPU, behind the interface of one: a step issues the G slices one after the other, each on
    its own three HIP streams, so the detector graph of one slice runs beside the extraction / tracking kernels of another (streams are independent: SURVEY.md §8(e) shards at
    stream granularity, inside a GPU as well as across GPUs).  Results are bit-identical to one pipeline over all S streams (every kernel works per frame).
    Reduced interface: step / synchronize / read / snapshot_* / pack_records / last_status with device-resident inputs (what bench.py --groups uses); the host-input
    entries of TrackerNative (host_buffers, step_host, wait_inputs) and frame_dev are not offered."""

    def __init__(self, lib, streams, cam, groups, make_detector=None, **kw):
        assert streams % groups == 0, 'streams must divide into equal groups'
        self.lib, self.S, self.G, self.Sg = lib, int(streams), int(groups), int(streams) // int(groups)
        self.detectors = [make_detector(self.Sg) if make_detector else None for _ in range(self.G)]
        self.tr = [TrackerNative(lib, self.Sg, cam, detector=d, **kw) for d in self.detectors]
        self.cap, self.rec_bytes, self.max_boxes = self.tr[0].cap, self.tr[0].rec_bytes, self.tr[0].max_boxes
        self.W, self.H = self.tr[0].W, self.tr[0].H

    def _sl(self, g, a):
        """slice g of a per-stream input.  Inputs are tensors / arrays with the stream axis first and contiguous in it (a raw device pointer or a strided view would give the
        pipelines wrong addresses without any error: ADVICE r5)"""
        if a is None: return None
        assert hasattr(a, 'shape') and len(a.shape) >= 1 and a.shape[0] == self.S, 'TrackerGroups takes tensors / arrays of shape (streams, ...), not raw pointers'
        contiguous = a.is_contiguous() if hasattr(a, 'is_contiguous') else a.flags['C_CONTIGUOUS']
        assert contiguous, 'TrackerGroups inputs must be contiguous'
        return a[g * self.Sg:(g + 1) * self.Sg]

    def set_initial_pose(self, Tcw_host):
        T = np.ascontiguousarray(Tcw_host, 'f4').reshape(self.S, 16)
        for g, t in enumerate(self.tr): t.set_initial_pose(T[g * self.Sg:(g + 1) * self.Sg])

    def step(self, d_gray, d_depth, d_bgr=None, stream=None, **kw):
        for g, t in enumerate(self.tr): t.step(self._sl(g, d_gray), self._sl(g, d_depth), d_bgr=self._sl(g, d_bgr), stream=stream, **kw)

    def synchronize(self):
        for t in self.tr: t.synchronize()

    def read(self):
        parts = [t.read() for t in self.tr]
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def snapshot_pose(self, d_out):
        for g, t in enumerate(self.tr): t.snapshot_pose(self._sl(g, d_out))

    def snapshot_boxes(self, stream_index, d_boxes, d_nboxes):
        self.tr[stream_index // self.Sg].snapshot_boxes(stream_index % self.Sg, d_boxes, d_nboxes)

    def pack_records(self, d_records, stream=None):
        for g, t in enumerate(self.tr): t.pack_records(self._sl(g, d_records), stream=stream)

    def last_status(self, stream=None):
        for t in self.tr: t.last_status(stream=stream)

    def close(self):
        for t in self.tr: t.close()
