"""Detector2D — host-side mirror of ORB_SLAM2::Detector2D (reference: src/sg-slam/include/Detector2D.h:45-67,
src/sg-slam/src/Detector2D.cc:16-89) over the C-ABI.  detect(bgr) fills the same public members the reference's
Frame reads (Frame.cc:482-500): mvObjects2D, mbHaveDynamicObjectFor{Mapping,RmDynamicFeature},
mvPotentialDynamicBorderFor{Mapping,RmDynamicFeature}."""
import ctypes as C
import numpy as np
from .capi import DetResult, _vp
from ._lib import load

CLASS_NAMES = ["background", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
               "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]        # Detector2D.cc:8-14


class Detector2D:
    def __init__(self, detection_confidence_threshold, dynamic_detection_confidence_threshold, param_path=None, bin_path=None,
                 param_text=None, bin_bytes=None, width=640, height=480, max_batch=1, lib=None, fuse=True, legacy_kernels=False, block_fusion=False, irb=None, gemm=None):
        self.lib = lib if lib is not None else load()
        if param_text is None:
            param_text = open(param_path or './Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.param').read()      # Detector2D.cc:24
        if bin_bytes is None:
            bin_bytes = open(bin_path or './Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.bin', 'rb').read()      # Detector2D.cc:25
        self._bin = bin_bytes
        h = C.c_void_p()
        # plan selection = test / tuning taps (include/sgx_debug.h, libsgx_taps.so / emulator only): fuse=False one kernel per ncnn layer with every blob kept; legacy_kernels the
        # simple reference kernels; block_fusion the opt-in k_fused_block plan; irb None = default (the shapes where the matrix-core block kernel wins), True / 2 = every supported
        # shape, False / 0 = off; gemm None = default, 'f32' = exact fp32 matrix products, 'bf16x3' = three-term bf16 split on the bf16 matrix pipes
        taps = dict(fusion=1 if fuse else 0, legacy_kernels=1 if legacy_kernels else 0, block_fusion=1 if block_fusion else 0,
                    irb=-1 if irb is None else (2 if irb is True else int(irb)), gemm=-1 if gemm is None else (0 if gemm in (0, 'f32') else 1))
        default = dict(fusion=1, legacy_kernels=0, block_fusion=0, irb=-1, gemm=-1)
        if taps != default or self.lib.has_taps:
            for k, v in taps.items(): self.lib.tap('sgx_det_debug_set_' + k)(v)
        self.lib.check(self.lib.dll.sgx_det_create(param_text.encode(), bin_bytes, len(bin_bytes), width, height, max_batch,
                                                   float(detection_confidence_threshold), float(dynamic_detection_confidence_threshold), C.byref(h)), 'sgx_det_create')
        if self.lib.has_taps:
            for k, v in default.items(): self.lib.tap('sgx_det_debug_set_' + k)(v)
        self.h = h; self.width, self.height, self.max_batch = width, height, max_batch
        npri, ncls, nk = C.c_int32(), C.c_int32(), C.c_int32(); g = C.c_double()
        self.lib.check(self.lib.dll.sgx_det_info(self.h, C.byref(npri), C.byref(ncls), C.byref(nk), C.byref(g)))
        self.num_priors, self.num_class, self.num_kernels, self.gmac = npri.value, ncls.value, nk.value, g.value
        self.gemm = 'bf16x3' if self.lib.dll.sgx_det_gemm_mode(self.h) == 1 else 'f32'
        self.mvObjects2D = []; self.mbHaveDynamicObjectForMapping = False; self.mbHaveDynamicObjectForRmDynamicFeature = False
        self.mvPotentialDynamicBorderForMapping = []; self.mvPotentialDynamicBorderForRmDynamicFeature = []

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dll.sgx_det_destroy(self.h); self.h = None

    def __del__(self):
        try: self.close()
        except Exception: pass

    def detect_batch(self, images):
        imgs = np.ascontiguousarray(images, np.uint8)
        if imgs.ndim == 3: imgs = imgs[None]
        B = imgs.shape[0]
        res = (DetResult * B)()
        self.lib.check(self.lib.dll.sgx_det_detect(self.h, _vp(imgs), imgs.shape[2] * 3, B, res), 'sgx_det_detect')
        return res

    def detect_batch_dev(self, d_img, pitch, batch, d_results, d_boxes=None, d_nboxes=None, max_boxes=0, d_have_dynamic=None, stream=None):
        """device-resident detect(): forward + DetectionOutput + filtering, asynchronous on `stream`; d_results = batch DetResult structs in device memory;
        d_boxes / d_nboxes / d_have_dynamic (optional) are the mask stage's inputs (sgx_dynamic_mask_batch_dev / sgx_frame_compact_keys_batch_dev layout)"""
        self.lib.check(self.lib.dll.sgx_det_detect_batch_dev(self.h, _vp(d_img), pitch, batch, _vp(d_results), None if d_boxes is None else _vp(d_boxes),
                                                             None if d_nboxes is None else _vp(d_nboxes), max_boxes, None if d_have_dynamic is None else _vp(d_have_dynamic),
                                                             None if stream is None else C.c_void_p(stream)), 'sgx_det_detect_batch_dev')

    def detect(self, bgr):
        r = self.detect_batch(bgr)[0]
        rect = lambda o: (o.x, o.y, o.w, o.h)
        self.raw = np.array([[d.label, d.score, d.xmin, d.ymin, d.xmax, d.ymax] for d in r.raw[:r.n_raw]], np.float32).reshape(-1, 6)
        self.mvObjects2D = [(o.id, CLASS_NAMES[o.id], o.prob, rect(o)) for o in r.objects[:r.n_objects]]
        self.mbHaveDynamicObjectForMapping = bool(r.have_dynamic_for_mapping)
        self.mbHaveDynamicObjectForRmDynamicFeature = bool(r.have_dynamic_for_rm_feature)
        self.mvPotentialDynamicBorderForMapping = [rect(o) for o in r.map_boxes[:r.n_map_boxes]]
        self.mvPotentialDynamicBorderForRmDynamicFeature = [rect(o) for o in r.rm_boxes[:r.n_rm_boxes]]

    def has_blob(self, name):
        n = C.c_int(0)
        return self.lib.tap('sgx_det_debug_read_blob')(self.h, name.encode(), 0, None, 0, C.byref(n)) == 0

    def time_ops(self, d_img, batch, reps=5):
        """[(description, ms per launch)] of every plan step on device images (tuning tap)"""
        ms = np.zeros(self.num_kernels, 'f4'); n = C.c_int(0)
        self.lib.check(self.lib.tap('sgx_det_debug_time_ops')(self.h, _vp(d_img), self.width * 3, batch, reps, _vp(ms), len(ms), C.byref(n)))
        out = []
        for i in range(n.value):
            buf = C.create_string_buffer(256); self.lib.check(self.lib.dll.sgx_det_plan_step(self.h, i, buf, 256)); out.append((buf.value.decode(), float(ms[i])))
        return out

    def op_descriptions(self):
        """one line per plan step (kind, layers, shapes, ' bf16x3' behind the steps that run on the bf16 matrix pipes)"""
        out = []
        for i in range(self.num_kernels):
            buf = C.create_string_buffer(256)
            if self.lib.dll.sgx_det_plan_step(self.h, i, buf, 256) != 0: break
            out.append(buf.value.decode())
        return out

    def debug_blob(self, name, image=0):
        n = C.c_int(0)
        self.lib.check(self.lib.tap('sgx_det_debug_read_blob')(self.h, name.encode(), image, None, 0, C.byref(n)))
        out = np.zeros(n.value, 'f4')
        self.lib.check(self.lib.tap('sgx_det_debug_read_blob')(self.h, name.encode(), image, _vp(out), n.value, C.byref(n)))
        return out
