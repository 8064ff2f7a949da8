"""TrackerBatch — S independent RGB-D streams tracked in lock-step on one GPU, one frame per stream per
step, every stage device-resident and asynchronous on one HIP stream:

  ORBextractor::operator()          -> sgx_orb_extract_batch_dev           (ORBextractor.cc:1045-1106)
  Frame::RmDynamicPoint... (mask)   -> sgx_flow_lk_batch_dev (calcOpticalFlowPyrLK, Frame.cc:445) + sgx_fundamental_ransac_batch_dev (pair selection +
                                       findFundamentalMat, :454-472) + wait for the detector (:478-500) + sgx_dynamic_mask_batch_dev +
                                       sgx_frame_compact_keys_batch_dev (:556-604)          [lk=True; with lk=False the LK / F inputs are given by the caller]
  Frame::ComputeStereoFromRGBD      -> sgx_frame_stereo_from_rgbd_batch_dev (Frame.cc:893-914)
  constant-velocity prediction      -> sgx_frame_motion_model_batch_dev     (Tracking.cc:463-470, :914)
  ORBmatcher::SearchByProjection    -> sgx_match_project_frame_batch_dev    (ORBmatcher.cc:1332-1472, th=15)
  Optimizer::PoseOptimization       -> sgx_pose_optimization_batch_dev      (Optimizer.cc:239-451)
  Tracking::TrackLocalMap           -> sgx_match_project_local_batch_dev    (isInFrustum + SearchByProjection(F, local points, th=3), Tracking.cc:1262-1312)
                                       + sgx_pose_optimization_batch_dev    (second PoseOptimization, Tracking.cc:979)
  Frame::UnprojectStereo            -> sgx_frame_unproject_batch_dev        (Frame.cc:916-930)
  MapPoint(Pos, pMap, pFrame, idx)  -> sgx_frame_make_map_points_batch_dev  (MapPoint.cc:45-67; the local map is the points of frames t-2, t-3)

This mirrors the call order of Tracking::GrabImageRGBD -> Frame() -> TrackWithMotionModel
(Tracking.cc:206-251, :906-967) in "visual odometry" form: the map points a frame is tracked against
are the previous frame's keypoints unprojected with their measured depth (UpdateLastFrame, :840-904),
Observations()==0.  It is the bench / integration harness, not a re-implementation of Tracking.
Arrays are torch CUDA tensors (product) or numpy arrays (kernel-logic emulator in tests).
"""
import ctypes as C
import numpy as np
from .capi import _vp
from .matcher import camera_struct
from .orb import ORBextractor
from .flow import OpticalFlowLK, fundamental_ransac_batch_dev


class TrackerBatch:
    stream_priorities = (0, 0)      # (extraction, tracking) HIP stream priorities; see __init__
    def __init__(self, lib, streams, cam, width=640, height=480, nfeatures=1000, xp='torch', th=15.0, pipelined=True, local_map=True, debug_taps=False, lk=False, max_boxes=8):
        self.lib, self.S, self.cam, self.W, self.H, self.th = lib, streams, dict(cam), width, height, th
        self.ex = ORBextractor(nfeatures=nfeatures, width=width, height=height, max_batch=streams, lib=lib)
        self.cap = self.ex.capacity
        self.cs = camera_struct(cam, width, height)
        self.scale = np.ascontiguousarray(self.ex.mvScaleFactor, 'f4')
        self.inv_sigma2 = np.ascontiguousarray(self.ex.mvInvLevelSigma2, 'f4')
        self.xp = xp
        S, cap = streams, self.cap
        z = self._zeros
        # triple-buffered per-frame state: frame t lives in slot t % 3 (current / last / being overwritten by the next extract)
        NB = 3
        self.keys = [z((S, cap, 28), 'u1') for _ in range(NB)]
        self.desc = [z((S, cap, 32), 'u1') for _ in range(NB)]
        self.n = [z((S,), 'i4') for _ in range(NB)]
        self.uright = [z((S, cap), 'f4') for _ in range(NB)]
        self.zdepth = [z((S, cap), 'f4') for _ in range(NB)]
        self.xw = [z((S, cap, 3), 'f4') for _ in range(NB)]
        self.has = [z((S, cap), 'u1') for _ in range(NB)]
        self.Tcw = [z((S, 16), 'f4') for _ in range(3)]          # cur, last, last-last (rotating)
        self.match = z((S, cap), 'i4')
        self.nmatch = z((S,), 'i4')
        self.outlier = z((S, cap), 'u1')
        self.ninl = z((S,), 'i4')
        self.zero_u8 = z((S, cap), 'u1')
        self.zero_i4 = z((S, cap), 'i4')
        self.vel_valid = z((S,), 'u1')
        # TrackLocalMap stage (Tracking.cc:969-1013): local map = the visual-odometry points of frames t-2 and t-3 (ring of 2*cap records)
        self.local_map = bool(local_map)
        self.debug_taps = bool(debug_taps)          # tests: keep the pose between the two PoseOptimization calls
        self.Tcw_mm = z((S, 16), 'f4')
        self.log_scale = float(np.log(np.float32(self.scale[1])))
        self.lm_xw = z((S, 2 * cap, 3), 'f4'); self.lm_normal = z((S, 2 * cap, 3), 'f4')
        self.lm_min = z((S, 2 * cap), 'f4'); self.lm_max = z((S, 2 * cap), 'f4')
        self.lm_desc = z((S, 2 * cap, 32), 'u1'); self.lm_skip = z((S, 2 * cap), 'u1')
        self.lm_obs = z((S, 2 * cap), 'i4'); self.lm_n = z((S,), 'i4')
        self.lm_skip[...] = 1; self.lm_obs[...] = 1; self.lm_n[...] = 2 * cap
        self.match_local = z((S, cap), 'i4'); self.nmatch_local = z((S,), 'i4'); self.in_view = z((S, 2 * cap), 'u1')
        self.merged = z((S, cap), 'i4'); self.cur_mp_obs = z((S, cap), 'i4'); self.xw_all = z((S, 3 * cap, 3), 'f4')
        self.outlier2 = z((S, cap), 'u1'); self.ninl2 = z((S,), 'i4')
        # dynamic-feature mask stage (Frame::RmDynamicPointWithSemanticAndGeometry): raw extraction buffers, keep flags, previous positions
        self.rkeys = z((S, cap, 28), 'u1'); self.rdesc = z((S, cap, 32), 'u1'); self.rn = z((S,), 'i4')
        self.keep = z((S, cap), 'u1'); self.prev_xy = z((S, cap, 2), 'f4')
        self.max_boxes = int(max_boxes)
        self.no_boxes = z((S, self.max_boxes, 4), 'f4'); self.no_nboxes = z((S,), 'i4')
        self.nfeatures = int(nfeatures)
        # the mask's real inputs (lk=True): LK flow into the previous frame's pyramid, RANSAC F, the previous frame's person boxes (Frame.cc:31-33 globals)
        self.lk = bool(lk)
        if self.lk:
            self.flow = OpticalFlowLK(width=width, height=height, max_batch=S, lib=lib)
            self.F = z((S, 9), 'f8'); self.f_ok = z((S,), 'i4'); self.f_stats = z((S, 4), 'i4'); self.lk_status = z((S, cap), 'u1')
            self.pre_boxes = z((S, self.max_boxes, 4), 'f4'); self.pre_nboxes = z((S,), 'i4'); self.pre_have = z((S,), 'i4')     # vPreFramePotentialDynamicBorder, bPreFrameHavePotentialDynamicObj
            self.no_have = z((S,), 'i4')
        self._held = [None] * NB                     # inputs of the last NB steps stay referenced until their slot is reused (their readers run on other streams)
        self.cur = 0
        self.frame_idx = 0
        # Two HIP streams: E = extraction (+stereo) of frame t+1 overlaps T = match / pose-opt / unproject of frame t.  The wide
        # kernels (FAST, descriptors) fill the chip while the one-workgroup-per-frame kernels (matcher, LM) run beside them.
        self.pipelined = bool(pipelined and xp == 'torch')
        if self.pipelined:
            import torch
            # Both streams at the same priority.  Rounds 2-5 gave the tracking stream the high priority (latency of a single camera).  Round 6 first blamed mixed priorities for a
            # rare LK difference between co-running trackers; they only changed how often the LK kernel met the detector's bf16 blocks on a CU — the cause was compiler-generated
            # packed fp32 (profiles/r6_lk_priority_diagnosis.md).  `stream_priorities` is the switch tools/diag_two_trackers.py uses (MODE=prio).
            pe, pt = self.stream_priorities
            self.sE, self.sT = torch.cuda.Stream(priority=pe), torch.cuda.Stream(priority=pt)
            self.ev_extract = [torch.cuda.Event() for _ in range(NB)]
            self.ev_track = [torch.cuda.Event() for _ in range(NB)]

    def _zeros(self, shape, dt):
        if self.xp == 'torch':
            import torch
            tdt = {'u1': torch.uint8, 'i4': torch.int32, 'f4': torch.float32, 'f8': torch.float64}[dt]
            return torch.zeros(shape, dtype=tdt, device='cuda')
        return np.zeros(shape, {'u1': np.uint8, 'i4': np.int32, 'f4': np.float32, 'f8': np.float64}[dt])

    def set_initial_pose(self, Tcw_host):
        """Tcw of the first frame of every stream (S,4,4) — the reference starts at identity
        (Tracking::StereoInitialization, Tracking.cc:547-603); synthetic streams start at ground truth."""
        T = np.ascontiguousarray(Tcw_host, 'f4').reshape(self.S, 16)
        for b in self.Tcw:
            if self.xp == 'torch':
                import torch
                b.copy_(torch.from_numpy(T))
            else:
                b[...] = T

    def _prev_points(self, keys_raw, A, shift, boxes, st):
        """prev_xy[s, i] = A[s] * (x, y, 1) of keypoint i: the stand-in for the LK tracker on synthetic streams (see synth.flow_affine);
        `shift` (S,2) moves the previous position of keypoints inside the first box of `boxes` (an independently moving object)."""
        L = self.lib
        L.check(L.tap('sgx_debug_flow_affine_batch_dev')(self.S, self.cap, _vp(keys_raw), _vp(self.rn), _vp(A), _vp(shift), _vp(boxes if shift is not None else None),
                                                      self.max_boxes, _vp(self.prev_xy), st), 'flow affine')

    def step(self, d_gray, d_depth, stream=None, gray_pitch=None, mask=None):
        """Track the next frame of every stream.  d_gray: S x H x W u8, d_depth: S x H x W u16 (raw, DepthMapFactor 5000).
        Asynchronous; with pipelined=True the extraction runs on its own stream (call synchronize() before reading results).
        mask (optional, frames t > 0): dict with 'A' (S,6) f32 flow map and 'F' (S,9) f64 fundamental matrices [the LK / RANSAC outputs the
        reference computes on the host, Frame.cc:445-472], optionally 'boxes' (S,max_boxes,4) f32, 'nboxes' (S,) i32, 'have_dynamic' (S,) i32
        [Detector2D results] and 'shift' (S,2): runs the dynamic-feature mask + erase step between extraction and ComputeStereoFromRGBD.
        With lk=True (constructor) the flow and F are computed here (every keypoint tracked into the previous frame, RANSAC F) and `mask` only carries the
        detector's results for THIS frame: 'boxes', 'nboxes', 'have_dynamic' (device arrays in sgx_det_detect_batch_dev's layout) and optionally 'event' —
        a torch event recorded after the detector launch, which the mask stage waits for (Frame.cc:478 `while(!isDetectImageFinished())`).
        Input lifetime: the tensors passed to step() are read asynchronously on the tracker's streams; the tracker keeps references to them for three
        steps, so callers may drop (but must not overwrite) them earlier."""
        L, S, cap, cam = self.lib, self.S, self.cap, self.cam
        t = self.frame_idx
        c, l = t % 3, (t - 1) % 3
        Tc, Tl, Tll = self.Tcw[0], self.Tcw[1], self.Tcw[2]
        if self.pipelined:
            import torch
            cur_stream = torch.cuda.current_stream()
            sE, sT = self.sE, self.sT
            sE.wait_stream(cur_stream)                                # the frames (and mask inputs) may have been produced on the caller's stream
            if t == 0:
                sT.wait_stream(cur_stream)
            if t >= 2:
                sE.wait_event(self.ev_track[(t - 2) % 3])            # slot c was "last" of step t-2+1: its readers must be done
            stE, stT = sE.cuda_stream, sT.cuda_stream
        else:
            stE = stT = stream
        self._held[c] = (d_gray, d_depth, mask)
        use_mask = (mask is not None or self.lk) and t > 0
        if self.lk:
            mask = mask or {}
            have_boxes = 'boxes' in mask
            boxes = mask.get('boxes', self.no_boxes); nboxes = mask.get('nboxes', self.no_nboxes); have_dyn = mask.get('have_dynamic', self.no_have)
            if t == 0:
                self.ex.extract_batch_dev(d_gray, gray_pitch or self.W, S, self.keys[c], self.desc[c], self.n[c], stream=stE)
                self.flow.lk_batch_dev(d_gray, gray_pitch or self.W, S, None, None, cap, None, None, stream=stE)       # first frame: pyramid only (imGrayPre empty, Frame.cc:155-163)
            else:
                self.ex.extract_batch_dev(d_gray, gray_pitch or self.W, S, self.rkeys, self.rdesc, self.rn, stream=stE)
                self.flow.lk_batch_dev(d_gray, gray_pitch or self.W, S, self.rkeys, self.rn, cap, self.prev_xy, self.lk_status, stream=stE)
                fundamental_ransac_batch_dev(L, S, cap, self.rkeys, self.rn, self.prev_xy, self.F, self.f_ok, self.f_stats, pre_have=self.pre_have, pre_boxes=self.pre_boxes,
                                             pre_nboxes=self.pre_nboxes, max_boxes=self.max_boxes, stream=stE)
                if self.pipelined and mask.get('event') is not None:
                    self.sE.wait_event(mask['event'])               # Frame.cc:478: the detector thread must have finished this image
                L.check(L.dll.sgx_dynamic_mask_batch_dev(S, cap, _vp(self.rkeys), _vp(self.rn), _vp(self.prev_xy), _vp(self.F), _vp(boxes), _vp(nboxes),
                                                         self.max_boxes, _vp(self.keep), _vp(stE)), 'dynamic mask')
                L.check(L.dll.sgx_frame_compact_keys_batch_dev(S, cap, _vp(self.rkeys), _vp(self.rdesc), _vp(self.rn), _vp(self.keep), _vp(have_dyn),
                                                               self.nfeatures, _vp(self.keys[c]), _vp(self.desc[c]), _vp(self.n[c]), _vp(stE)), 'compact keys')
                # Frame.cc:482-499: this frame's detector results become the "previous frame" state of the next call (frame 0 never gets here, like the reference)
                if self.xp == 'torch':
                    import torch
                    with torch.cuda.stream(self.sE) if self.pipelined else torch.cuda.stream(torch.cuda.current_stream()):
                        self.pre_boxes.copy_(boxes); self.pre_nboxes.copy_(nboxes); self.pre_have.copy_(have_dyn)
                else:
                    self.pre_boxes[...] = boxes; self.pre_nboxes[...] = nboxes; self.pre_have[...] = have_dyn
        elif not use_mask:
            self.ex.extract_batch_dev(d_gray, gray_pitch or self.W, S, self.keys[c], self.desc[c], self.n[c], stream=stE)
        else:
            self.ex.extract_batch_dev(d_gray, gray_pitch or self.W, S, self.rkeys, self.rdesc, self.rn, stream=stE)
            boxes = mask.get('boxes', self.no_boxes); nboxes = mask.get('nboxes', self.no_nboxes)
            self._prev_points(self.rkeys, mask['A'], mask.get('shift'), boxes, _vp(stE))
            L.check(L.dll.sgx_dynamic_mask_batch_dev(S, cap, _vp(self.rkeys), _vp(self.rn), _vp(self.prev_xy), _vp(mask['F']), _vp(boxes), _vp(nboxes),
                                                     self.max_boxes, _vp(self.keep), _vp(stE)), 'dynamic mask')
            L.check(L.dll.sgx_frame_compact_keys_batch_dev(S, cap, _vp(self.rkeys), _vp(self.rdesc), _vp(self.rn), _vp(self.keep), _vp(mask.get('have_dynamic')),
                                                           self.nfeatures, _vp(self.keys[c]), _vp(self.desc[c]), _vp(self.n[c]), _vp(stE)), 'compact keys')
        L.check(L.dll.sgx_frame_stereo_from_rgbd_batch_dev(S, cap, _vp(self.keys[c]), _vp(self.n[c]), _vp(d_depth), self.W, self.H,
                                                           float(cam['depth_factor']), float(cam['bf']), _vp(self.uright[c]), _vp(self.zdepth[c]), _vp(stE)), 'stereo')
        if self.pipelined:
            self.ev_extract[c].record(self.sE)
            self.sT.wait_event(self.ev_extract[c])
        st = _vp(stT)
        if t > 0:
            # Tc <- predicted pose from (Tl, Tll); frame 1 has no velocity yet -> uses the last pose
            L.check(L.dll.sgx_frame_motion_model_batch_dev(S, _vp(Tl), _vp(Tll), _vp(self.vel_valid), _vp(Tc), st), 'motion model')
            L.check(L.dll.sgx_match_project_frame_batch_dev(
                S, cap, _vp(self.keys[c]), _vp(self.desc[c]), _vp(self.uright[c]), _vp(self.n[c]), _vp(Tc),
                _vp(self.keys[l]), _vp(self.n[l]), _vp(self.has[l]), _vp(self.zero_u8), _vp(self.xw[l]), _vp(self.zero_i4), _vp(self.desc[l]), _vp(Tl),
                C.byref(self.cs), _vp(self.scale), len(self.scale), float(self.th), 0, 1, _vp(self.match), _vp(self.nmatch), st), 'match')
            L.check(L.dll.sgx_pose_optimization_batch_dev(
                S, cap, _vp(self.keys[c]), _vp(self.uright[c]), _vp(self.n[c]), _vp(self.match), None, _vp(self.xw[l]), cap,
                _vp(self.inv_sigma2), len(self.inv_sigma2), C.byref(self.cs), _vp(Tc), _vp(self.outlier), _vp(self.ninl), st), 'pose opt')
            if t == 1:
                if self.pipelined:
                    import torch
                    with torch.cuda.stream(self.sT):
                        self.vel_valid.fill_(1)
                else:
                    self.vel_valid[...] = 1
        if t > 0 and self.debug_taps:
            if self.pipelined:
                import torch
                with torch.cuda.stream(self.sT):
                    self.Tcw_mm.copy_(Tc)
            else:
                self.Tcw_mm[...] = Tc
        if t > 0 and self.local_map:
            # ---- TrackLocalMap: SearchLocalPoints (isInFrustum + SearchByProjection th=3) and the second PoseOptimization
            L.check(L.dll.sgx_frame_merge_matches_batch_dev(S, cap, _vp(self.n[c]), _vp(self.match), _vp(self.outlier), None, None, None,
                                                            None, _vp(self.cur_mp_obs), None, st), 'mp_obs')
            L.check(L.dll.sgx_match_project_local_batch_dev(
                S, cap, _vp(self.keys[c]), _vp(self.desc[c]), _vp(self.uright[c]), _vp(self.n[c]), _vp(Tc), _vp(self.cur_mp_obs),
                2 * cap, _vp(self.lm_n), _vp(self.lm_xw), _vp(self.lm_normal), _vp(self.lm_min), _vp(self.lm_max), _vp(self.lm_desc), _vp(self.lm_obs), _vp(self.lm_skip),
                C.byref(self.cs), _vp(self.scale), len(self.scale), self.log_scale, 3.0, 0.8, 0.5, _vp(self.match_local), _vp(self.nmatch_local), _vp(self.in_view), st), 'match local')
            L.check(L.dll.sgx_frame_merge_matches_batch_dev(S, cap, _vp(self.n[c]), _vp(self.match), _vp(self.outlier), _vp(self.match_local),
                                                            _vp(self.xw[l]), _vp(self.lm_xw), _vp(self.merged), None, _vp(self.xw_all), st), 'merge')
            L.check(L.dll.sgx_pose_optimization_batch_dev(
                S, cap, _vp(self.keys[c]), _vp(self.uright[c]), _vp(self.n[c]), _vp(self.merged), None, _vp(self.xw_all), 3 * cap,
                _vp(self.inv_sigma2), len(self.inv_sigma2), C.byref(self.cs), _vp(Tc), _vp(self.outlier2), _vp(self.ninl2), st), 'pose opt 2')
        L.check(L.dll.sgx_frame_unproject_batch_dev(S, cap, _vp(self.keys[c]), _vp(self.n[c]), _vp(self.zdepth[c]), _vp(Tc), C.byref(self.cs),
                                                    _vp(self.xw[c]), _vp(self.has[c]), st), 'unproject')
        if t > 0 and self.local_map:
            # the last frame's points join the local map for the NEXT frames (ring slice (t-1) % 2 <- frame t-1, i.e. at step t+1 the ring holds t-1 and t-2)
            L.check(L.dll.sgx_frame_make_map_points_batch_dev(S, cap, (t - 1) % 2, _vp(self.keys[l]), _vp(self.n[l]), _vp(self.xw[l]), _vp(self.has[l]), _vp(self.desc[l]),
                                                              _vp(Tl), _vp(self.scale), len(self.scale), _vp(self.lm_xw), _vp(self.lm_normal), _vp(self.lm_min),
                                                              _vp(self.lm_max), _vp(self.lm_desc), _vp(self.lm_skip), st), 'make map points')
        if self.pipelined:
            self.ev_track[c].record(self.sT)
        # rotate poses: cur -> last, last -> last-last
        self.Tcw = [Tll, Tc, Tl]
        self.cur = c
        self.frame_idx += 1

    def snapshot_pose(self):
        """device copy of the pose of the frame just tracked (S,16), ordered on the tracking stream — for trajectory files / ATE without a host sync per frame"""
        T = self.Tcw[1]
        if self.xp != 'torch':
            return T.copy()
        if self.pipelined:
            import torch
            with torch.cuda.stream(self.sT):
                return T.clone()
        return T.clone()

    def synchronize(self):
        if self.pipelined:
            import torch
            self.sE.synchronize(); self.sT.synchronize()
            torch.cuda.current_stream().wait_stream(self.sT)

    def last_pose(self):
        """(S,4,4) float32 Tcw of the most recently tracked frame (synchronises)."""
        self.synchronize()
        T = self.Tcw[1]
        return (T.cpu().numpy() if self.xp == 'torch' else T.copy()).reshape(self.S, 4, 4)

    def last_counts(self):
        self.synchronize()
        g = (lambda a: a.cpu().numpy()) if self.xp == 'torch' else (lambda a: a.copy())
        return g(self.n[self.cur]), g(self.nmatch), g(self.ninl)

    def last_local_counts(self):
        """(local-map matches, inliers of the second PoseOptimization) of the last tracked frame"""
        self.synchronize()
        g = (lambda a: a.cpu().numpy()) if self.xp == 'torch' else (lambda a: a.copy())
        return g(self.nmatch_local), g(self.ninl2)
