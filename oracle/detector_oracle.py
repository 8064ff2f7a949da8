"""oracle/detector_oracle.py — CPU ORACLE (numpy, fp32) for Detector2D::detect.  TEST INFRASTRUCTURE ONLY.

Restates  src/sg-slam/src/Detector2D.cc:34-89  (pre-processing, ncnn forward of the shipped graph
src/sg-slam/Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.param, box post-processing) and
Frame::RmDynamicPointWithSemanticAndGeometry's keep/erase predicate (src/sg-slam/src/Frame.cc:563-627).

ncnn is NOT vendored in the reference tree (un-pinned git master, README.md:144-152) and the weight blob is absent:
==> PARITY UNPINNED at the ncnn boundary <==  The layer semantics below follow ncnn's published layer definitions
(SURVEY.md Appendix A.7): Convolution / ConvolutionDepthWise (0=outc 1=k 3=stride 4=pad 5=bias 6=weights 7=group),
BinaryOp (0 add, 2 mul, 3 div), Clip, ReLU, Permute(order 3 = HWC), Flatten, Concat, Reshape, Softmax, PriorBox (Caffe-SSD
style; the mmdetection flags 14/15 of the shipped graph are NOT modelled — noted in DESIGN.md), DetectionOutput.
Weights are synthetic (the .bin is missing): N(0, 2/fan_in), seed 7, then a synthetic batch-norm fold per convolution (sg_slam_amd.synth._calibrate: unit-variance
blobs, active gates, separated class scores — a network as well conditioned as a trained one), in ncnn .bin order.  They are INPUT data shared by harness, tests and oracle.
"""
import numpy as np

MEAN = np.array([123.675, 116.28, 103.53], np.float32)
TARGET = 300


# ---------------------------------------------------------------- param parsing / synthetic weights
# (shared with the harness: the graph description is data, the synthetic blob is an input, neither is oracle arithmetic)
from sg_slam_amd.synth import parse_ncnn_param as parse_param, synth_ncnn_weights as synth_weights  # noqa: E402,F401


# ---------------------------------------------------------------- pre-processing
_RESIZE_TABS = {}


def _resize_tabs(s, d):
    if (s, d) not in _RESIZE_TABS:
        scale = float(s) / d
        ofs = np.zeros(d, np.int32); a = np.zeros((d, 2), np.int32)
        for i in range(d):
            f = np.float32((i + 0.5) * scale - 0.5)
            si = int(np.floor(f)); f = np.float32(f - si)
            if si < 0: si, f = 0, np.float32(0)
            if si >= s - 1: si, f = s - 2, np.float32(1)
            ofs[i] = si
            a[i, 0] = int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048))); a[i, 1] = int(np.rint(f * np.float32(2048)))
        _RESIZE_TABS[(s, d)] = (ofs, a)
    return _RESIZE_TABS[(s, d)]


def resize_bilinear_c3(src, dw, dh):
    """ncnn resize_bilinear_c3 (src/mat_pixel_resize.cpp): 11-bit fixed point, like OpenCV's but clamping to (n-2, 1.0).  Only the rows / columns the output touches are formed."""
    sh, sw, _ = src.shape
    xo, xa = _resize_tabs(sw, dw); yo, ya = _resize_tabs(sh, dh)
    rows_needed = np.unique(np.concatenate([yo, yo + 1]))
    s = src[rows_needed].astype(np.int32)
    rows = s[:, xo, :] * xa[None, :, 0, None] + s[:, xo + 1, :] * xa[None, :, 1, None]            # (rows needed, dw, 3)
    i0 = np.searchsorted(rows_needed, yo); i1 = np.searchsorted(rows_needed, yo + 1)
    r0 = rows[i0]; r1 = rows[i1]
    out = (((ya[:, 0, None, None] * (r0 >> 4)) >> 16) + ((ya[:, 1, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def preprocess(img_u8):
    """from_pixels_resize(..., PIXEL_RGB, w, h, 300, 300) + substract_mean_normalize(mean, norm=1): -> float32 CHW"""
    r = resize_bilinear_c3(img_u8, TARGET, TARGET)
    x = r.astype(np.float32).transpose(2, 0, 1)
    return (x - MEAN[:, None, None]) * np.float32(1.0)


# ---------------------------------------------------------------- forward
def conv2d(x, w, b, outc, k, stride, pad, group, dt=np.float32):
    x = x.astype(dt); w = w.astype(dt); b = b.astype(dt)
    inc, H, W = x.shape
    Ho = (H + 2 * pad - k) // stride + 1; Wo = (W + 2 * pad - k) // stride + 1
    xp = np.zeros((inc, H + 2 * pad, W + 2 * pad), dt); xp[:, pad:pad + H, pad:pad + W] = x
    if group == 1:
        w = w.reshape(outc, inc, k, k)
        if k == 1:
            y = w.reshape(outc, inc) @ xp[:, ::stride, ::stride].reshape(inc, -1)
        else:
            cols = np.stack([xp[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride] for i in range(k) for j in range(k)], 1)   # (inc, k*k, Ho, Wo)
            y = w.reshape(outc, inc * k * k) @ cols.reshape(inc * k * k, -1)
        y = y.reshape(outc, Ho, Wo).astype(dt)
    else:
        assert group == inc == outc
        w = w.reshape(outc, k, k)
        y = np.zeros((outc, Ho, Wo), dt)
        for i in range(k):
            for j in range(k):
                y += w[:, i, j, None, None] * xp[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride]
    return (y + b[:, None, None]).astype(dt)


def prior_box(fh, fw, img, p):
    mins, maxs, ars = p.get(0, []), p.get(1, []), p.get(2, [])
    flip, clip, offset = p.get(7, 1), p.get(8, 0), np.float32(p.get(13, 0.0))
    step_w = np.float32(img) / np.float32(fw); step_h = np.float32(img) / np.float32(fh)
    boxes = []
    for i in range(fh):
        for j in range(fw):
            cx = np.float32((np.float32(j) + offset) * step_w); cy = np.float32((np.float32(i) + offset) * step_h)
            def add(bw, bh):
                boxes.append([(cx - bw * np.float32(0.5)) / img, (cy - bh * np.float32(0.5)) / img, (cx + bw * np.float32(0.5)) / img, (cy + bh * np.float32(0.5)) / img])
            for kk, ms in enumerate(mins):
                ms = np.float32(ms); add(ms, ms)
                if maxs:
                    s = np.float32(np.sqrt(ms * np.float32(maxs[kk]))); add(s, s)
                for ar in ars:
                    sq = np.float32(np.sqrt(np.float32(ar)))
                    add(ms * sq, ms / sq)
                    if flip: add(ms / sq, ms * sq)
    b = np.array(boxes, np.float32)
    if clip: b = np.clip(b, 0, 1)
    var = np.tile(np.array([p.get(3, .1), p.get(4, .1), p.get(5, .2), p.get(6, .2)], np.float32), len(b))
    return np.stack([b.reshape(-1), var])


def detection_output(loc, conf, priors, p, margins=None, extra=0):
    """ncnn DetectionOutput.  margins (optional dict, filled): how far the decisions behind the returned rows are from flipping — 'order' = the smallest score gap between
    consecutive rows of the final list (incl. the first row cut off by keep_top_k), 'iou' = the smallest |IoU - nms_threshold| over the suppression tests of candidates that score
    at least as high as the last returned row.  An image whose margins are inside fp32 noise has no well-defined row list (tests/test_detector.py::run_rows_identical).
    extra > 0 (tests): also return the first `extra` rows BEHIND the keep_top_k cut — a score tie across the cut decides which row is the last one."""
    ncls, nms_th, nms_topk, keep_topk, conf_th = p[0], np.float32(p[1]), p[2], p[3], np.float32(p[4])
    var = np.array([p.get(5, .1), p.get(6, .1), p.get(7, .2), p.get(8, .2)], np.float32)
    pb = priors[0].reshape(-1, 4); n = len(pb); loc = loc.reshape(n, 4); conf = conf.reshape(n, ncls)
    pw = pb[:, 2] - pb[:, 0]; ph = pb[:, 3] - pb[:, 1]; pcx = (pb[:, 0] + pb[:, 2]) * np.float32(0.5); pcy = (pb[:, 1] + pb[:, 3]) * np.float32(0.5)
    cx = var[0] * loc[:, 0] * pw + pcx; cy = var[1] * loc[:, 1] * ph + pcy
    w = np.exp(var[2] * loc[:, 2]).astype(np.float32) * pw; h = np.exp(var[3] * loc[:, 3]).astype(np.float32) * ph
    boxes = np.stack([cx - w * np.float32(0.5), cy - h * np.float32(0.5), cx + w * np.float32(0.5), cy + h * np.float32(0.5)], 1).astype(np.float32)
    allr = []; tests = []
    for c in range(1, ncls):
        sc = conf[:, c]; idx = np.nonzero(sc > conf_th)[0]
        idx = idx[np.argsort(-sc[idx], kind='stable')][:nms_topk]
        keep = []
        for i in idx:
            ok = True
            for k in keep:
                a, b = boxes[i], boxes[k]
                iw = min(a[2], b[2]) - max(a[0], b[0]); ih = min(a[3], b[3]) - max(a[1], b[1])
                inter = np.float32(iw * ih) if (iw > 0 and ih > 0) else np.float32(0)
                union = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
                if margins is not None: tests.append((float(sc[i]), abs(float(inter / union) - float(nms_th))))
                if inter / union > nms_th: ok = False; break
            if ok: keep.append(i)
        allr += [(c, sc[i], *boxes[i]) for i in keep]
    allr.sort(key=lambda r: -r[1])
    if margins is not None:
        top = [float(r[1]) for r in allr[:keep_topk + 1]]
        margins['order'] = min([a_ - b_ for a_, b_ in zip(top, top[1:])], default=1.0)
        cut = top[min(len(top), keep_topk) - 1] if top else 0.0
        margins['iou'] = min([m for s_, m in tests if s_ >= cut], default=1.0)
    return np.array(allr[:keep_topk + extra], np.float32).reshape(-1, 6)


def forward(layers, W, x, dt=np.float32):
    """x: CHW input blob. dt = np.float32 (the reference's precision) or np.float64 (error yardstick for the fp32 tolerance).
    Returns (detection_out rows[n,6], blobs dict)."""
    blobs = {'input': x.astype(dt)}
    for L in layers:
        t, p, ins, outs = L['type'], L['p'], L['ins'], L['outs']
        if t == 'Input': continue
        if t == 'MemoryData': blobs[outs[0]] = W[L['name']].astype(dt); continue
        if t == 'Split':
            for o in outs: blobs[o] = blobs[ins[0]]
            continue
        a = blobs[ins[0]]
        if t in ('Convolution', 'ConvolutionDepthWise'):
            w, b = W[L['name']]
            y = conv2d(a, w, b, p[0], p[1], p.get(3, 1), p.get(4, 0), p.get(7, 1), dt)
        elif t == 'BinaryOp':
            b = blobs[ins[1]]; op = p.get(0, 0)
            if b.size == 1: b = b.reshape(())
            y = (a + b if op == 0 else a * b if op == 2 else a / b if op == 3 else None).astype(dt)
        elif t == 'Clip': y = np.clip(a, dt(p[0]), dt(p[1]))
        elif t == 'ReLU': y = np.maximum(a, dt(0))
        elif t == 'Permute': assert p[0] == 3; y = np.ascontiguousarray(a.transpose(1, 2, 0))
        elif t == 'Flatten': y = a.reshape(-1)
        elif t == 'Concat':
            xs = [blobs[i] for i in ins]
            y = np.concatenate(xs, 0) if p.get(0, 0) == 0 else np.concatenate(xs, 1)
        elif t == 'Reshape': y = a.reshape(-1, p[0])
        elif t == 'Softmax':
            e = np.exp(a - a.max(1, keepdims=True)).astype(dt); y = (e / e.sum(1, keepdims=True)).astype(dt)
        elif t == 'PriorBox':
            y = prior_box(a.shape[1], a.shape[2], TARGET, p)
        elif t == 'DetectionOutput':
            y = detection_output(blobs[ins[0]].astype(np.float32), blobs[ins[1]].astype(np.float32), blobs[ins[2]], p)
        else:
            raise NotImplementedError(t)
        blobs[outs[0]] = y
    return blobs['detection_out'], blobs


class TorchForward:
    """The same graph with torch's CPU operators in float32 (MKL-DNN / oneDNN convolutions): the CPU-baseline leg's stand-in for ncnn's SIMD forward (bench.py cpu_baseline) —
    the numpy forward() above is a readable restatement (~4 s per frame), not a fair CPU timing.  Checked against forward() in tests/test_detector.py (fp32 drift).  Runs up to
    mbox_loc / mbox_conf_softmax; DetectionOutput stays the numpy routine.  TEST / BASELINE INFRASTRUCTURE like the rest of oracle/."""

    def __init__(self, layers, W, threads=1):
        import torch
        self.torch, self.layers, self.threads = torch, layers, threads
        self.P = {}
        for L in layers:
            if L['type'] in ('Convolution', 'ConvolutionDepthWise'):
                w, b = W[L['name']]; p = L['p']
                self.P[L['name']] = (torch.from_numpy(np.ascontiguousarray(w, np.float32).reshape(p[0], -1, p[1], p[1]).copy()), torch.from_numpy(np.ascontiguousarray(b, np.float32).copy()))
            elif L['type'] == 'MemoryData':
                self.P[L['name']] = float(W[L['name']][0])

    def __call__(self, x):
        """x: (3, 300, 300) float32 from preprocess() -> (mbox_loc flat, mbox_conf_softmax (n, 21)) as numpy float32"""
        torch = self.torch; F = torch.nn.functional
        old = torch.get_num_threads(); torch.set_num_threads(self.threads)
        try:
            with torch.no_grad():
                blobs = {'input': torch.from_numpy(np.ascontiguousarray(x, np.float32))[None]}
                for L in self.layers:
                    t, p, ins, outs = L['type'], L['p'], L['ins'], L['outs']
                    if t == 'Input': continue
                    if t == 'MemoryData': blobs[outs[0]] = self.P[L['name']]; continue
                    if t == 'Split':
                        for o in outs: blobs[o] = blobs[ins[0]]
                        continue
                    if t in ('PriorBox', 'DetectionOutput') or ins[0] not in blobs: continue          # the prior-box branch is input-independent
                    a = blobs[ins[0]]
                    if t in ('Convolution', 'ConvolutionDepthWise'):
                        w, b = self.P[L['name']]
                        y = F.conv2d(a, w, b, stride=p.get(3, 1), padding=p.get(4, 0), groups=p.get(7, 1))
                    elif t == 'BinaryOp':
                        b = blobs[ins[1]]; op = p.get(0, 0)
                        y = a + b if op == 0 else a * b if op == 2 else a / b
                    elif t == 'Clip': y = torch.clamp(a, float(p[0]), float(p[1]))
                    elif t == 'ReLU': y = torch.relu(a)
                    elif t == 'Permute': y = a.permute(0, 2, 3, 1).contiguous()
                    elif t == 'Flatten': y = a.reshape(1, -1)
                    elif t == 'Concat': y = torch.cat([blobs[i] for i in ins], 1)
                    elif t == 'Reshape': y = a.reshape(-1, p[0])
                    elif t == 'Softmax': y = torch.softmax(a, 1)
                    else: raise NotImplementedError(t)
                    blobs[outs[0]] = y
                return blobs['mbox_loc'].reshape(-1).numpy(), blobs['mbox_conf_softmax'].numpy()
        finally:
            torch.set_num_threads(old)


def eval_layer(L, W, args, dt=np.float32):
    """one layer of forward() on explicit input arrays (same arithmetic; used by forward_cut)"""
    t, p = L['type'], L['p']
    a = args[0]
    if t in ('Convolution', 'ConvolutionDepthWise'):
        w, b = W[L['name']]
        return conv2d(a, w, b, p[0], p[1], p.get(3, 1), p.get(4, 0), p.get(7, 1), dt)
    if t == 'BinaryOp':
        b = args[1]; op = p.get(0, 0)
        if b.size == 1: b = b.reshape(())
        return (a + b if op == 0 else a * b if op == 2 else a / b).astype(dt)
    if t == 'Clip': return np.clip(a, dt(p[0]), dt(p[1]))
    if t == 'ReLU': return np.maximum(a, dt(0))
    if t == 'Permute': return np.ascontiguousarray(a.transpose(1, 2, 0))
    if t == 'Flatten': return a.reshape(-1)
    if t == 'Concat': return np.concatenate(args, 0) if p.get(0, 0) == 0 else np.concatenate(args, 1)
    if t == 'Reshape': return a.reshape(-1, p[0])
    if t == 'Softmax':
        e = np.exp(a - a.max(1, keepdims=True)).astype(dt); return (e / e.sum(1, keepdims=True)).astype(dt)
    raise NotImplementedError(t)


def forward_cut(layers, W, given, shapes, targets, dt=np.float64):
    """Per-step isolation (VERDICT r4 next #1c): every blob in `targets` is evaluated in `dt` from the NEAREST blobs of `given` upstream of it (given[target] itself is not
    used for that target), i.e. exactly the layers one plan step of the device fuses, fed with the device's own step inputs.  given: name -> flat array (device blobs, plus
    'input' from preprocess()); shapes: name -> shape (from one forward() run).  Returns name -> array."""
    producer = {o: L for L in layers for o in L['outs']}
    cache = {}

    def value(name, root):
        if not root:
            if name in given: return np.asarray(given[name], dt).reshape(shapes[name])
            if name in cache: return cache[name]
        L = producer[name]
        if L['type'] == 'MemoryData': v = W[L['name']].astype(dt)
        elif L['type'] == 'Split': v = value(L['ins'][0], False)
        elif L['type'] == 'Input': raise KeyError('the network input must be given')
        else: v = eval_layer(L, W, [value(i, False) for i in L['ins']], dt)
        if not root and name not in given: cache[name] = v
        return v

    return {t: value(t, True) for t in targets}


def detect(layers, W, img_u8, det_th=0.90, dyn_th=0.01):
    """Detector2D::detect: returns (objects [(id, prob, x, y, w, h)], person boxes for mapping, person boxes for the dynamic-feature mask)."""
    img_h, img_w = img_u8.shape[:2]
    out, _ = forward(layers, W, preprocess(img_u8))
    objs, map_boxes, rm_boxes = [], [], []
    T = np.float32(TARGET)
    def cl(v): return min(max(np.float32(v * T), np.float32(0)), np.float32(TARGET - 1)) / T
    for v in out:
        if v[1] > np.float32(det_th) or (v[1] > np.float32(dyn_th) and int(v[0]) == 15):
            x1 = cl(v[2]) * img_w; y1 = cl(v[3]) * img_h; x2 = cl(v[4]) * img_w; y2 = cl(v[5]) * img_h
            rect = (np.float32(x1), np.float32(y1), np.float32(x2 - x1), np.float32(y2 - y1))
            if int(v[0]) == 15:
                map_boxes.append(rect)
                if v[1] > np.float32(0.2): rm_boxes.append(rect)
            else:
                objs.append((int(v[0]), float(v[1]), *rect))
    return objs, map_boxes, rm_boxes


# ---------------------------------------------------------------- dynamic-feature mask predicate
def dynamic_mask(kps_xy, prev_xy, F, boxes, have_dynamic, nfeatures=1000):
    """Frame.cc:556-604 keep/erase decision per keypoint.  Returns (keep[N] bool, restored flag).
    CheckEpiLineDistToRmDynamicPoint (:613-627, fp64) with threshold 0.2 inside a person box (isInDynamicRegion :629-652, strict),
    1.0 elsewhere; if a person is present and fewer than 0.1*nFeatures survive, all keypoints are restored (:599-604)."""
    F = np.asarray(F, np.float64)
    keep = np.zeros(len(kps_xy), bool)
    for i, ((x, y), (px, py)) in enumerate(zip(kps_xy, prev_xy)):
        x = np.float32(x); y = np.float32(y)
        a = x * F[0, 0] + y * F[0, 1] + F[0, 2]; b = x * F[1, 0] + y * F[1, 1] + F[1, 2]; c = x * F[2, 0] + y * F[2, 1] + F[2, 2]
        dist = abs(a * np.float32(px) + b * np.float32(py) + c) / np.sqrt(a * a + b * b)
        inbox = have_dynamic and any((x > bx and x < bx + bw and y > by and y < by + bh) for (bx, by, bw, bh) in boxes)
        keep[i] = dist < (0.2 if inbox else 1.0)
    restored = bool(have_dynamic and keep.sum() < nfeatures * 0.1)
    return keep, restored
