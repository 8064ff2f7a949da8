"""Detector2D forward + post-processing + dynamic-feature mask: emulator vs the numpy oracle on the shipped graph with
synthetic weights (the reference's .bin is absent).

Weights (round 5): the He draw followed by a synthetic batch-norm fold per convolution (sg_slam_amd.synth._calibrate) — unit-variance blobs, active gates,
separated class scores.  On that network the oracle's own fp32 run is <= 1e-5 from its float64 run at every blob (asserted below), so the criteria have teeth:
  * every tapped blob of the device is within max(4 x the oracle's own fp32 drift, 6e-6) of the float64 run (run_compare);
  * DetectionOutput rows of the device equal the oracle's fp32 rows — same labels in the same order, scores / boxes to 2e-5 (run_compare, run_rows_identical);
  * every plan step in isolation: the float64 oracle evaluated on the device's OWN step inputs gives the device's step output to 2e-6 (fp32 products) /
    4e-6 (bf16x3 products) of the blob's magnitude (run_steps_isolated) — covers every k_conv_pw3 / k_irb / k_fused_block2 instantiation of the plan."""
import os
import numpy as np
import pytest
from oracle import detector_oracle as D
from sg_slam_amd.detector import Detector2D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAM = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
RTOL = 1e-4


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


@pytest.fixture(scope='module')
def model():
    layers = D.parse_param(PARAM)
    W, blob = D.synth_weights(layers, seed=7)
    return layers, W, blob


def make_image(seed, person=False):
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    img[100:300, 200:400] = (img[100:300, 200:400] // 4 + 150).astype(np.uint8)
    return img


def run_compare(lib, model, seeds=(0, 1), fuse=False, gemm=None, report=None):
    """gemm: None = the library's default matrix-product scheme (bf16x3 on the device, exact fp32 in the emulator), 'f32' / 'bf16x3' force one.  The criterion is the same for
    both: it never asked for the ascending-k fmaf chain, only for fp32-grade drift.  report (dict): per-blob (device drift, oracle-fp32 drift) for the caller."""
    layers, W, blob = model
    det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=2, lib=lib, fuse=fuse, irb=fuse, gemm=gemm)
    assert gemm is None or det.gemm == gemm
    assert det.num_kernels == (282 if not fuse else 39), det.num_kernels      # 39: the inverted-residual blocks run as one k_irb each, the two SSD heads of a feature map as one, the pre-processing inside the stem
    assert det.num_priors == 2268 and det.num_class == 21 and abs(det.gmac - 0.5574) < 1e-3
    imgs = np.stack([make_image(s) for s in seeds])
    res = det.detect_batch(imgs)
    for b, s in enumerate(seeds):
        x = D.preprocess(imgs[b])
        out, blobs = D.forward(layers, W, x)
        _, blobs64 = D.forward(layers, W, x, dt=np.float64)
        if det.has_blob('input'):
            assert (det.debug_blob('input', b).reshape(3, 300, 300) == x).all()             # integer resize + exact fp32 subtract
        else:
            assert fuse                                                                     # k_stem_pre resizes into LDS: the stem's output ('587' below) is the first blob in HBM
        for name in ('580', '587', '603'):
            if fuse and not det.has_blob(name):
                assert name == '580'                              # the stem's raw output lives only in registers of the fused h-swish epilogue
                continue
            assert rel_err(det.debug_blob(name, b), np.asarray(blobs[name], np.float32).reshape(-1)) < 1e-5, name
        for name in ('620', '632', '672', '849', '908', '944', 'mbox_loc', 'mbox_conf_softmax'):
            if fuse and not det.has_blob(name):
                assert name == '620'                              # project output ahead of the squeeze-excite gate: lives only in the accumulators of k_irb
                continue
            got = det.debug_blob(name, b); ref = np.asarray(blobs64[name]).reshape(-1); np32 = np.asarray(blobs[name], np.float64).reshape(-1)
            e_dev, e_np = rel_err(got.astype(np.float64), ref), rel_err(np32, ref)
            if report is not None: report[(s, name)] = (e_dev, e_np)
            assert e_np <= 1e-5, (name, e_np)                         # the calibrated network is well conditioned: the oracle's own fp32 run stays at rounding level
            # accumulated drift: the device's matrix products are ascending-k fp32 chains (what v_mfma_f32_32x32x2_f32 computes), numpy's are blocked BLAS sums, so a CORRECT device run
            # sits at 1 - 4.2 x the oracle's own drift (tools/campaign_detector.py on random weight draws and images, round 5: 420 cases; the largest ratios at the 128-value blob '944',
            # where max-error / max-value is a noisy statistic).  4 x with a 6e-6 floor;
            # the sharp criterion is the per-step one (run_steps_isolated: 2e-6 / 4e-6 absolute per plan step)
            assert got.shape == ref.shape and e_dev <= max(4 * e_np, 6e-6), (name, e_dev, e_np)
        r = res[b]
        got_rows = np.array([[d.label, d.score, d.xmin, d.ymin, d.xmax, d.ymax] for d in r.raw[:r.n_raw]], np.float32).reshape(-1, 6)
        # post-processing (DetectionOutput + Detector2D::detect filtering) checked exactly on the DEVICE's own loc/conf
        p = [L for L in layers if L['type'] == 'DetectionOutput'][0]['p']
        exp_rows = D.detection_output(det.debug_blob('mbox_loc', b), det.debug_blob('mbox_conf_softmax', b), blobs['mbox_priorbox'], p)
        assert got_rows.shape == exp_rows.shape and (got_rows[:, 0] == exp_rows[:, 0]).all()
        assert np.abs(got_rows[:, 1:] - exp_rows[:, 1:]).max() < 1e-5
        # ... and END TO END against the oracle's own fp32 run: same detections in the same order (what feeds Detector2D::detect's filter and the mask) wherever the oracle's
        # decisions are outside fp32 noise (run_rows_identical explains the margins); 95 % of the rows otherwise
        mg = {}
        out_ext = D.detection_output(np.asarray(blobs['mbox_loc'], np.float32).reshape(-1), np.asarray(blobs['mbox_conf_softmax'], np.float32).reshape(-1), blobs['mbox_priorbox'], p, margins=mg, extra=8)
        assert got_rows.shape == out.shape and len(out) > 0 and (out_ext[:len(out)] == out).all()
        if mg['iou'] > 1e-4:
            assert rows_identical(got_rows, out_ext) >= 1, (s, mg)
        else:
            assert sum(1 for r in got_rows if ((out[:, 0] == r[0]) & (np.abs(out[:, 1:] - r[1:]).max(1) < 2e-5)).any()) >= 0.95 * len(out), (s, mg)
        keep = [v for v in exp_rows if v[1] > np.float32(0.90) or (v[1] > np.float32(0.01) and int(v[0]) == 15)]
        assert r.n_objects == sum(int(v[0]) != 15 for v in keep) and r.n_map_boxes == sum(int(v[0]) == 15 for v in keep)
        assert r.n_rm_boxes == sum(int(v[0]) == 15 and v[1] > np.float32(0.2) for v in keep)
        assert bool(r.have_dynamic_for_mapping) == (r.n_map_boxes > 0) and bool(r.have_dynamic_for_rm_feature) == (r.n_rm_boxes > 0)
    det.close()


def device_rows(res, b):
    r = res[b]
    return np.array([[d.label, d.score, d.xmin, d.ymin, d.xmax, d.ymax] for d in r.raw[:r.n_raw]], np.float32).reshape(-1, 6)


def rows_identical(a, b, tol=2e-5, tie=2e-5):
    """DetectionOutput rows a (device) against b (oracle; may carry a few extra rows from behind the keep_top_k cut, detection_output(extra=...)): 2 = identical — same labels in
    the same order, scores / boxes within tol; 1 = identical up to the order of rows whose SCORES TIE within `tie` (fp32 noise: any correct fp32 evaluation of the heads is within
    max(4 x the oracle's own drift, 6e-6) ~ 1e-5 of the float64 run — run_compare —, so two rows less than 2e-5 apart may legitimately swap, also across the cut; the stable sort
    orders only EXACT ties): every device row has its own oracle row (same label, score / box within tol) and the oracle row sitting at the device row's position scores within
    `tie` of it; 0 = different detections."""
    n = len(a)
    if n > len(b) or (len(b) > n and n == 0): return 0
    if n == 0: return 2
    if (a[:, 0] == b[:n, 0]).all() and float(np.abs(a[:, 1:] - b[:n, 1:]).max()) < tol: return 2
    used = np.zeros(len(b), bool)
    for i, r in enumerate(a):
        cand = np.nonzero(~used & (b[:, 0] == r[0]) & (np.abs(b[:, 1:] - r[1:]).max(1) < tol))[0]
        if len(cand) == 0: return 0
        j = cand[np.argmin(np.abs(cand - i))]
        if j != i and abs(float(b[j, 1]) - float(b[i, 1])) > tie: return 0
        used[j] = True
    return 1


def run_rows_identical(lib, model, seeds=tuple(range(8)), gemms=('f32', 'bf16x3'), plans=(None,), min_exact=None):
    """VERDICT r4 next #1b — the condition under which bf16x3 may be the default: DetectionOutput rows (label, order, score / box <= 2e-5) IDENTICAL between the oracle's
    fp32 run, the device's exact-fp32 plan and the device's bf16x3 plan, on >= 8 images.
    What "identical" can mean between two fp32 evaluations: the heads of any correct one are within ~8e-6 of a float64 run (run_compare), so (a) two rows whose oracle scores are
    less than 2e-5 apart may swap places — rows_identical's tie rule, return value 1 — and (b) a suppression decision whose IoU is within 1e-4 of the NMS threshold may flip
    (the oracle reports that margin: detection_output(margins=...)).  Every image without a knife-edge IoU must match up to (a); the others (rare) in 95 % of their rows.  The
    EXACT matches (return value 2: same order everywhere) are counted and must be the majority — the tie rule is the exception, not how the test passes."""
    layers, W, blob = model
    imgs = np.stack([make_image(s) for s in seeds])
    p = [L for L in layers if L['type'] == 'DetectionOutput'][0]['p']
    ref = []; iou_ok = []; nrows = []
    for im in imgs:
        out, blobs = D.forward(layers, W, D.preprocess(im))
        m = {}
        again = D.detection_output(np.asarray(blobs['mbox_loc'], np.float32).reshape(-1), np.asarray(blobs['mbox_conf_softmax'], np.float32).reshape(-1), blobs['mbox_priorbox'], p, margins=m, extra=8)
        assert (again[:len(out)] == out).all()
        ref.append(again); iou_ok.append(m['iou'] > 1e-4); nrows.append(len(out))
    n_person = 0; n_exact = 0; n_cmp = 0
    for gemm in gemms:
        for irb in plans:
            det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=len(seeds), lib=lib, fuse=True, irb=irb, gemm=gemm)
            assert gemm is None or det.gemm == gemm
            res = det.detect_batch(imgs)
            for b in range(len(seeds)):
                got = device_rows(res, b)
                assert nrows[b] >= 20 and len(got) == nrows[b]
                same = rows_identical(got, ref[b])
                if iou_ok[b]:
                    assert same >= 1, (gemm, irb, seeds[b], got[:4], ref[b][:4])
                else:
                    hit = sum(1 for r in got if ((ref[b][:, 0] == r[0]) & (np.abs(ref[b][:, 1:] - r[1:]).max(1) < 2e-5)).any())
                    assert hit >= 0.95 * len(got), (gemm, irb, seeds[b], hit)
                n_exact += same == 2; n_cmp += 1
                n_person += res[b].n_rm_boxes
            det.close()
    assert n_person > 0                         # person boxes (what the dynamic-feature mask consumes) are among the compared rows
    if min_exact is None: min_exact = (n_cmp + 1) // 2
    assert n_exact >= min_exact and sum(iou_ok) >= len(seeds) - max(1, len(seeds) // 4), (n_exact, n_cmp, iou_ok)
    return n_person, n_exact, n_cmp


def steps_isolated_errors(layers, W, x, given):
    """the judgement of run_steps_isolated on a set of device blobs: name -> |device - float64 oracle on the device's own inputs| / max|oracle|, for every blob that a plan step
    produced (everything in `given` except the network input and pure aliases)"""
    _, blobs64 = D.forward(layers, W, x, dt=np.float64)
    shapes = {k: np.asarray(v).shape for k, v in blobs64.items()}
    producer = {o: L for L in layers for o in L['outs']}
    targets = [nm for nm in given if nm != 'input' and producer[nm]['type'] not in ('Split', 'Reshape')]
    ref = D.forward_cut(layers, W, given, shapes, targets)
    worst = {}
    for nm in targets:
        r = np.asarray(ref[nm]).reshape(-1); g = given[nm].astype(np.float64)
        worst[nm] = float(np.abs(g - r).max() / max(np.abs(r).max(), 1e-30))
    return worst


def resident_blobs(det, layers, x, image=0):
    """name -> flat array of every blob the plan keeps in device memory (plus the network input from the oracle's exact integer pre-processing)"""
    producer = {o: L for L in layers for o in L['outs']}
    sizes = {}
    for L in layers:                                             # element counts from the graph's shapes are not needed: a blob whose device size differs from the oracle's is skipped below
        pass
    given = {'input': x}
    for L in layers:
        if L['type'] in ('Input', 'MemoryData', 'PriorBox', 'DetectionOutput', 'Permute', 'Flatten'): continue      # Permute / Flatten: layout-only, in the fused plans these names alias a slice of the concat buffer
        for nm in L['outs']:
            if det.has_blob(nm): given[nm] = det.debug_blob(nm, image)
    return given


def run_steps_isolated(lib, model, seed=2, gemm=None, irb=None, fuse=True, tol=None, block_fusion=False):
    """VERDICT r4 next #1c: for EVERY plan step, the float64 oracle evaluated on the device's own step inputs (the nearest device-resident blobs upstream) must give the
    device's step output to tol x max|out|: 2e-6 for fp32 matrix products, 4e-6 for bf16x3 (dropped terms <= 3 x 2^-24 |a||b|).  A kernel bug in ONE k_conv_pw3 / k_irb /
    k_fused_block2 instantiation cannot hide behind accumulated drift: each step is judged on its own inputs."""
    layers, W, blob = model
    det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=1, lib=lib, fuse=fuse, irb=irb, gemm=gemm, block_fusion=block_fusion)
    if tol is None: tol = 4e-6 if det.gemm == 'bf16x3' else 2e-6
    img = make_image(seed)
    det.detect_batch(img[None])
    x = D.preprocess(img)
    given = resident_blobs(det, layers, x)
    _, blobs64 = D.forward(layers, W, x, dt=np.float64)
    given = {k: v for k, v in given.items() if v.size == int(np.prod(np.asarray(blobs64[k]).shape))}
    worst = steps_isolated_errors(layers, W, x, given)
    assert len(worst) >= det.num_kernels - 13, (len(worst), det.num_kernels)       # 12 Permute steps + the pre-processing have no target of their own
    descs = det.op_descriptions()
    det.close()
    bad = {nm: e for nm, e in worst.items() if not e <= tol}
    assert not bad, (det.gemm, irb, tol, bad)            # every failing step at once (one GPU run names them all)
    return worst, descs


def test_steps_isolated_criterion_has_teeth(emu, model):
    """The per-step checker is itself checked: a fault of 1e-4 of a blob's magnitude injected into ONE element of ONE device blob (what a wrong lane, tap or k-step of one kernel
    instantiation produces) is reported at that step — and at the steps that read the blob, whose device outputs no longer follow from their inputs — while every other step
    stays below 2e-6; and the checker does not peek at the blob it judges (forward_cut never uses given[target] for the target)."""
    layers, W, blob = model
    det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=1, lib=emu, fuse=True, gemm='f32')
    img = make_image(2); det.detect_batch(img[None]); x = D.preprocess(img)
    given = resident_blobs(det, layers, x); det.close()
    clean = steps_isolated_errors(layers, W, x, given)
    assert max(clean.values()) <= 2e-6
    victim = '849' if '849' in given else sorted(k for k in given if k.isdigit())[len(given) // 2]
    hurt = dict(given); v = hurt[victim].copy(); i = int(np.argmax(np.abs(v))) // 2; v[i] += 1e-4 * np.abs(v).max(); hurt[victim] = v
    faulty = steps_isolated_errors(layers, W, x, hurt)
    flagged = {nm for nm, e in faulty.items() if e > 2e-6}
    assert victim in flagged and faulty[victim] > 5e-5, (victim, faulty[victim])
    # (its readers see a one-element input fault of 1e-4 attenuated by their own weights: they may or may not cross 2e-6; nothing else may)
    # ... i.e. every flagged step must lie DOWNSTREAM of the victim in the graph (the former check "everything not flagged is below the threshold" was true by
    # construction); a step upstream of the victim or on a sibling branch that crossed the threshold would be the checker peeking at the wrong blob
    down = {victim}; grew = True
    while grew:
        grew = False
        for L in layers:
            if any(i in down for i in L['ins']) and not all(o in down for o in L['outs']): down.update(L['outs']); grew = True
    assert flagged <= down and len(flagged) <= 8, (sorted(flagged), sorted(flagged - down))
    assert all(faulty[nm] <= 2e-6 for nm in faulty if nm not in down), sorted(nm for nm in faulty if nm not in down and faulty[nm] > 2e-6)
    # a larger fault in a whole channel (a wrong weight row) must also surface downstream
    hurt2 = dict(given); v = given[victim].copy(); v[:361] *= 1.01
    for nm in given:                                             # the Split outputs are views of the same device buffer: a real fault shows under every name
        if nm == victim or nm.startswith(victim + '_splitncnn'): hurt2[nm] = v
    faulty2 = steps_isolated_errors(layers, W, x, hurt2)
    assert faulty2[victim] > 1e-3 and len({nm for nm, e in faulty2.items() if e > 2e-6}) >= 2


def test_detector_emu_matches_oracle(emu, model):
    run_compare(emu, model)


def test_detector_emu_bf16x3_plan_matches_oracle(emu, model):
    """The bf16x3 plan on the CPU tier: the planner's branch (host-split weights in the matrix-core layout, zero padding, the k >= 64 rule) with the pointwise layers run by a software
    model of k_conv_pw3 that reads those very operands (sgx_det.cpp::sgx_pw3_emu) — same criterion as the exact-fp32 plan; the device kernel itself is checked on the GPU."""
    run_compare(emu, model, seeds=(0,), gemm='bf16x3')
    run_compare(emu, model, seeds=(1,), fuse=True, gemm='bf16x3')


def test_detector_emu_fused_matches_oracle(emu, model):
    run_compare(emu, model, seeds=(0,), fuse=True)


def test_rows_identical_rule():
    """the comparison itself: identical -> 2; two rows whose scores tie within 2e-6 swapped -> 1; a swap of distinguishable rows, a moved box, a changed label -> 0"""
    r = np.array([[4, 0.9, .1, .1, .5, .5], [7, 0.5275831, .2, .2, .4, .4], [1, 0.5275830, .4, .8, .7, .9], [15, 0.3, 0, 0, 1, 1]], np.float32)
    assert rows_identical(r, r.copy()) == 2 and rows_identical(r + np.float32(3e-6) * (np.arange(6) > 0), r) == 2
    assert rows_identical(r[[0, 2, 1, 3]], r) == 1 and rows_identical(r[[0, 2, 1, 3]], r, tie=5e-8) == 0
    assert rows_identical(r[[1, 0, 2, 3]], r) == 0
    ext = np.vstack([r, np.array([[9, 0.2999995, .3, .3, .6, .6]], np.float32)])          # the oracle's list with the first row behind the cut
    assert rows_identical(r, ext) == 2 and rows_identical(np.vstack([r[:3], ext[4:]]), ext) == 1 and rows_identical(np.vstack([r[:3], ext[4:]]), r) == 0
    q = r.copy(); q[3, 4] += 1e-3
    assert rows_identical(q, r) == 0
    q = r.copy(); q[3, 0] = 14
    assert rows_identical(q, r) == 0 and rows_identical(r, r[:3]) == 0          # (a shorter device list than the oracle's is the callers' length check)


def test_detector_emu_rows_identical_to_oracle(emu, model):
    _, n_exact, n_cmp = run_rows_identical(emu, model, seeds=(0, 1), gemms=('f32', 'bf16x3'))
    assert n_exact == n_cmp == 4


def test_detector_emu_steps_isolated(emu, model):
    """the emulator models the block kernels' arithmetic order on the host (no MFMA): the per-step criterion itself is exercised here, the device kernels on the GPU tier"""
    for gemm, fuse in (('f32', True), ('bf16x3', True), ('f32', False)):
        worst, _ = run_steps_isolated(emu, model, gemm=gemm, fuse=fuse)
        assert len(worst) >= 30
        print(gemm, fuse, len(worst), 'steps, worst', max(worst.items(), key=lambda kv: kv[1]))


def run_fused_equals_unfused(lib, model):
    """(All plans with the EXACT fp32 matrix products, gemm='f32': the bf16x3 default of the device build has another summation order and is compared with tolerances,
    run_bf16x3_against_f32.)  The fused plan (conv epilogue programs, HWC head stores) and the tuned kernels (streaming MFMA GEMM, LDS-tiled depthwise, stem)
    apply the same fp32 operations in the same order as the one-kernel-per-layer reference plan: bit-identical outputs."""
    layers, W, blob = model
    imgs = np.stack([make_image(3), make_image(4)])
    outs = []
    for fuse, legacy, blocks, irb in ((False, True, False, False), (True, True, False, False), (True, False, False, False), (False, False, False, False), (True, False, True, False), (True, False, False, True), (True, False, False, 1)):
        # the fifth plan additionally runs the six expand -> depthwise -> project triples as one k_fused_block each (opt-in: correct but slower at batch 256);
        # the last two run inverted-residual blocks (with their squeeze-excite gates) and SSD heads as one matrix-core kernel each (k_irb): every supported shape / the default plan (the shapes where it wins)
        det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=2, lib=lib, fuse=fuse, legacy_kernels=legacy, block_fusion=blocks, irb=irb, gemm='f32')
        assert det.num_kernels == (52 if irb == 1 and irb is not True else 39 if irb else 90 if blocks else 96 if (fuse and not legacy) else 103 if fuse else 282), det.num_kernels     # 96: the three high-resolution blocks run as k_fused_block2
        det.detect_batch(imgs)
        outs.append([np.stack([det.debug_blob(nm, b) for b in range(2)]) for nm in ('587', '632', '672', '849', '908', '944', 'mbox_loc', 'mbox_conf_softmax')])
        det.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert (a == b).all()


def run_bf16x3_against_f32(lib, model, seeds=(0, 1, 3, 4)):
    """bf16x3 is the default scheme of the pointwise layers only under these conditions: blob by blob its distance to the oracle's float64 run is at most
    2 x the exact-fp32 plan's (floor 2e-6), and DetectionOutput rows are IDENTICAL (label, order, score / box <= 2e-5) between the two device plans and the oracle's fp32
    run — for the default plan and for the plan with every supported shape on the matrix-core block kernel.  Device only."""
    layers, W, blob = model
    imgs = np.stack([make_image(s) for s in seeds])
    taps = ('603', '632', '672', '849', '908', '944', 'mbox_loc', 'mbox_conf_softmax')
    got = {}
    rows = {}
    for gemm in ('f32', 'bf16x3'):
        for fuse_irb in (None, True):             # the default plan and every shape on the matrix-core block kernel
            det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=len(seeds), lib=lib, fuse=True, irb=fuse_irb, gemm=gemm)
            assert det.gemm == gemm
            res = det.detect_batch(imgs)
            got[(gemm, fuse_irb)] = {nm: np.stack([det.debug_blob(nm, b) for b in range(len(seeds))]) for nm in taps if det.has_blob(nm)}
            rows[(gemm, fuse_irb)] = [device_rows(res, b) for b in range(len(seeds))]
            det.close()
    worst = {}
    for b, s in enumerate(seeds):
        x = D.preprocess(imgs[b])
        out32, _ = D.forward(layers, W, x)
        _, blobs64 = D.forward(layers, W, x, dt=np.float64)
        for plan in (None, True):
            for nm in taps:
                if nm not in got[('f32', plan)] or nm not in got[('bf16x3', plan)]: continue
                ref = np.asarray(blobs64[nm]).reshape(-1)
                e32 = rel_err(got[('f32', plan)][nm][b].astype(np.float64), ref); e3 = rel_err(got[('bf16x3', plan)][nm][b].astype(np.float64), ref)
                worst[(plan, nm)] = max(worst.get((plan, nm), (0, 0)), (e3, e32))
                assert e3 <= max(2.0 * e32, 2e-6), (plan, nm, s, e3, e32)
            ra, rb = rows[('f32', plan)][b], rows[('bf16x3', plan)][b]
            assert rows_identical(ra, rb) and rows_identical(ra, out32) and len(out32) >= 20, (plan, s)
    return worst


def test_detector_emu_fused_equals_unfused(emu, model):
    run_fused_equals_unfused(emu, model)


def run_mask(lib, to_dev=lambda a: a, to_host=lambda a: a):
    from sg_slam_amd.capi import KP_DTYPE, _vp
    rng = np.random.RandomState(2)
    cap, n = 1024, 900
    keys = np.zeros((1, cap), KP_DTYPE); keys['x'][0, :n] = rng.uniform(0, 640, n); keys['y'][0, :n] = rng.uniform(0, 480, n)
    prev = np.zeros((1, cap, 2), 'f4'); prev[0, :n, 0] = keys['x'][0, :n] + 3 + rng.randn(n) * 0.6; prev[0, :n, 1] = keys['y'][0, :n] + rng.randn(n) * 0.6
    F = np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]], 'f8')                               # pure x-translation: epipolar lines are rows
    boxes = np.zeros((1, 4, 4), 'f4'); boxes[0, 0] = (200, 100, 150, 250); boxes[0, 1] = (400, 50, 60, 60)
    nb = np.array([2], 'i4'); cnt = np.array([n], 'i4'); keep = np.zeros((1, cap), np.uint8)
    dk, dc, dp, dF, db, dn, dkeep = map(to_dev, (keys, cnt, prev, F.reshape(1, 9).copy(), boxes, nb, keep))
    lib.check(lib.dll.sgx_dynamic_mask_batch_dev(1, cap, _vp(dk), _vp(dc), _vp(dp), _vp(dF), _vp(db), _vp(dn), 4, _vp(dkeep), None))
    got = to_host(dkeep)[0, :n].astype(bool)
    exp, restored = D.dynamic_mask(list(zip(keys['x'][0, :n], keys['y'][0, :n])), prev[0, :n], F, [tuple(b) for b in boxes[0, :2]], True)
    assert (got == exp).all() and 0.2 < exp.mean() < 0.95 and not restored
    assert (to_host(dkeep)[0, n:] == 0).all()


def test_dynamic_mask_emu(emu):
    run_mask(emu)


def run_compact(lib, to_dev=lambda a: a, to_host=lambda a: a):
    """Erase + restore rule (Frame.cc:556-604) vs numpy: order-preserving compaction of keypoints and descriptor rows."""
    from sg_slam_amd import frame as fr
    from sg_slam_amd.capi import KP_DTYPE
    rng = np.random.RandomState(11)
    B, cap = 5, 1024
    n = np.array([1000, 987, 1000, 0, 300], 'i4')
    keys = np.zeros((B, cap), KP_DTYPE); keys['x'] = rng.rand(B, cap) * 640; keys['y'] = rng.rand(B, cap) * 480; keys['octave'] = rng.randint(0, 8, (B, cap))
    keys['response'] = rng.rand(B, cap) * 100; keys['class_id'] = -1
    desc = rng.randint(0, 256, (B, cap, 32)).astype(np.uint8)
    keep = (rng.rand(B, cap) < 0.8).astype(np.uint8)
    keep[1] = rng.rand(cap) < 0.05          # 5 % survive, dynamic object present  -> restore everything
    keep[2] = rng.rand(cap) < 0.05          # 5 % survive, no dynamic object       -> erase anyway
    keep[4, :300] = 0; keep[4, :99] = 1     # 99 survive (< 100 = 0.1 * nFeatures), dynamic -> restore; exactly at the boundary
    have = np.array([1, 1, 0, 1, 1], 'i4')
    ko, do, no = to_dev(np.zeros_like(keys)), to_dev(np.zeros_like(desc)), to_dev(np.zeros(B, 'i4'))
    fr.compact_keys_batch(lib, B, cap, to_dev(keys), to_dev(desc), to_dev(n), to_dev(keep), to_dev(have), 1000, ko, do, no)
    ko, do, no = to_host(ko), to_host(do), to_host(no)
    if ko.dtype != KP_DTYPE: ko = ko.view(KP_DTYPE).reshape(B, cap)
    for b in range(B):
        sel = np.nonzero(keep[b, :n[b]])[0]
        if have[b] and len(sel) < 100.0: sel = np.arange(n[b])
        assert no[b] == len(sel), (b, no[b], len(sel))
        assert (ko[b, :len(sel)] == keys[b, sel]).all() and (do[b, :len(sel)] == desc[b, sel]).all()
    assert no[1] == 987 and no[2] < 100 and no[4] == 300 and no[0] < 1000


def test_compact_emu(emu):
    run_compact(emu)


def run_detect_dev(lib, model, to_dev=lambda a: a, to_host=lambda a: a):
    """sgx_det_detect_batch_dev: results struct and the mask-stage arrays it fills on the device (person rectangles, count, have-dynamic flag) against the
    host entry (which copies the same device results back; its rows are checked against the oracle in run_compare)"""
    from sg_slam_amd.capi import DetResult
    import ctypes as C
    layers, W, blob = model
    det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=2, lib=lib)
    imgs = np.stack([make_image(5), make_image(6)])
    host = det.detect_batch(imgs)
    MB = 4
    d_img = to_dev(np.ascontiguousarray(imgs)); d_res = to_dev(np.zeros((2, C.sizeof(DetResult)), np.uint8))
    d_boxes = to_dev(np.full((2, MB, 4), -1, 'f4')); d_nb = to_dev(np.full(2, -1, 'i4')); d_have = to_dev(np.full(2, -1, 'i4'))
    det.detect_batch_dev(d_img, 640 * 3, 2, d_res, d_boxes, d_nb, MB, d_have)
    res = (DetResult * 2).from_buffer_copy(np.ascontiguousarray(to_host(d_res)).tobytes())
    boxes, nb, have = to_host(d_boxes), to_host(d_nb), to_host(d_have)
    for b in range(2):
        assert bytes(res[b]) == bytes(host[b])
        r = res[b]
        assert nb[b] == min(r.n_rm_boxes, MB) and have[b] == r.have_dynamic_for_rm_feature
        for q in range(nb[b]):
            o = r.rm_boxes[q]
            assert (boxes[b, q] == np.array([o.x, o.y, o.w, o.h], 'f4')).all()
        assert r.n_raw > 0 and r.n_map_boxes >= r.n_rm_boxes
    det.close()


def test_detect_dev_emu(emu, model):
    run_detect_dev(emu, model)


def run_detection_output_stress(lib, model):
    """DetectionOutput alone on adversarial head outputs: clusters of heavily overlapping boxes (long suppression chains across the 64-candidate chunks of the
    device scan), exactly tied scores (stable order = prior index, then class), more than nms_top_k candidates per class, classes with none"""
    from sg_slam_amd.capi import DetResult
    layers, W, blob = model
    det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=2, lib=lib)
    n, nc = det.num_priors, det.num_class
    _, blobs = D.forward(layers, W, D.preprocess(make_image(0)))
    priors = blobs['mbox_priorbox']
    p = [L for L in layers if L['type'] == 'DetectionOutput'][0]['p']
    rng = np.random.RandomState(11)
    loc = np.zeros((2, n, 4), 'f4'); conf = np.zeros((2, n, nc), 'f4')
    for b in range(2):
        loc[b] = rng.randn(n, 4).astype('f4') * (0.3 if b == 0 else 2.0)           # small offsets: neighbouring priors overlap heavily
        raw = rng.rand(n, nc).astype('f4') ** 3
        raw[:, 3] = 0.0                                                               # a class without candidates
        raw[:, 5] = np.round(raw[:, 5] * 8) / 8                                       # heavy score ties
        raw[:, 15] = np.maximum(raw[:, 15], 0.02)                                     # every prior is a "person" candidate (> nms_top_k)
        if b == 1: raw[rng.rand(n) < 0.7] *= 0.001
        conf[b] = raw
    res = (DetResult * 2)()
    lib.check(lib.tap('sgx_det_debug_detection_output')(det.h, loc.ctypes.data, conf.ctypes.data, 2, res), 'detection_output')
    for b in range(2):
        exp = D.detection_output(loc[b].reshape(-1), conf[b].reshape(-1), priors, p)
        r = res[b]
        got = np.array([[d.label, d.score, d.xmin, d.ymin, d.xmax, d.ymax] for d in r.raw[:r.n_raw]], np.float32).reshape(-1, 6)
        assert got.shape == exp.shape and len(exp) == 100
        assert (got[:, :2] == exp[:, :2]).all() and np.abs(got[:, 2:] - exp[:, 2:]).max() < 1e-5
    det.close()


def test_detection_output_stress_emu(emu, model):
    run_detection_output_stress(emu, model)


def test_oracle_torch_forward_matches_numpy_forward(model):
    """oracle.detector_oracle.TorchForward (torch's float32 CPU operators: the detector leg of bench.py's cpu_baseline) computes the same network as the numpy oracle: head outputs
    within fp32 drift of the float64 run, and the harness's per-frame callable runs"""
    layers, W, _ = model
    x = D.preprocess(make_image(6))
    _, b64 = D.forward(layers, W, x, dt=np.float64)
    loc, conf = D.TorchForward(layers, W, threads=1)(x)
    for nm, g in (('mbox_loc', loc), ('mbox_conf_softmax', conf)):
        r = np.asarray(b64[nm]).reshape(-1)
        assert g.size == r.size and rel_err(g.reshape(-1).astype(np.float64), r) < 1e-5, nm
    from oracle import cpu_chain
    fn = cpu_chain.make_detector_fn(PARAM)
    loc2, conf2 = fn(make_image(6)[:, :, 0])
    assert loc2.size == loc.size and np.isfinite(conf2).all()


def test_oracle_convolutions_match_torch(model):
    """pins the ORACLE's convolution semantics (padding, stride, depthwise grouping, ncnn weight order [outc][inc/group][kh][kw]) to an independent
    implementation: every convolution of the shipped graph is evaluated in float64 on the oracle's own input blob, once with the oracle's numpy routine and
    once with torch.nn.functional.conv2d; the outputs must agree to 1e-12 of the magnitude bound sum |w||x| (a wrong stride / pad / weight order is off by O(1))."""
    import torch
    import torch.nn.functional as F
    layers, W, _ = model
    x = D.preprocess(make_image(4))
    _, ref = D.forward(layers, W, x, dt=np.float64)

    def conv_torch(x, w, b, outc, k, stride, pad, group):
        inc = x.shape[0]
        wt = torch.from_numpy(np.ascontiguousarray(w, np.float64).reshape(outc, inc // group, k, k))
        y = F.conv2d(torch.from_numpy(np.ascontiguousarray(x, np.float64))[None], wt, torch.from_numpy(np.ascontiguousarray(b, np.float64)), stride=stride, padding=pad, groups=group)
        return y[0].numpy()

    nconv = 0; kinds = set()
    for L in layers:
        if L['type'] not in ('Convolution', 'ConvolutionDepthWise'): continue
        p = L['p']; w, b = W[L['name']]; a = ref[L['ins'][0]]
        args = (p[0], p[1], p.get(3, 1), p.get(4, 0), p.get(7, 1))
        y0 = D.conv2d(a, w, b, *args, dt=np.float64)
        y1 = conv_torch(a, w, b, *args)
        bound = D.conv2d(np.abs(a), np.abs(w), np.abs(b), *args, dt=np.float64).max()
        assert y0.shape == y1.shape and np.abs(y0 - y1).max() <= 1e-12 * bound, L['name']
        assert np.array_equal(y0, ref[L['outs'][0]])
        nconv += 1; kinds.add(args[1:])
    assert nconv > 90 and len(kinds) >= 6          # 1x1, 3x3 s1 / s2, 5x5 s1 / s2, full and depthwise
