"""GPU (MI355X): detector forward (fp32-MFMA pointwise convolutions) + post-processing + dynamic mask vs the oracle."""
import numpy as np
import pytest
from test_detector import run_compare, run_mask, model   # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_detector_gpu_matches_oracle(gpulib_taps, model):
    run_compare(gpulib_taps, model)


def test_detector_gpu_exact_fp32_plan_matches_oracle(gpulib_taps, model):
    """the exact-fp32 matrix products (SGX_DET_GEMM=f32): the anchor plan of the plan-equality tests"""
    run_compare(gpulib_taps, model, seeds=(0,), gemm='f32')
    run_compare(gpulib_taps, model, seeds=(0,), fuse=True, gemm='f32')


def test_detector_gpu_bf16x3_plan_matches_oracle(gpulib_taps, model):
    """the bf16x3 matrix products (three-term bf16 split, six cross products on v_mfma_f32_32x32x16_bf16): same criterion as the fp32 plan, per-layer and fused"""
    run_compare(gpulib_taps, model, gemm='bf16x3')
    run_compare(gpulib_taps, model, fuse=True, gemm='bf16x3')


def test_detector_gpu_bf16x3_same_detections_as_fp32_and_oracle(gpulib_taps, model):
    from test_detector import run_bf16x3_against_f32
    worst = run_bf16x3_against_f32(gpulib_taps, model)
    print('worst (bf16x3, fp32) distance to the float64 run per blob:', worst)


def test_detector_gpu_rows_identical_to_oracle_on_8_images(gpulib_taps, model):
    """DetectionOutput rows: oracle fp32 == device f32 plan == device bf16x3 plan (label, order, score / box <= 1e-5) on eight images, default plan and all-shapes block plan"""
    from test_detector import run_rows_identical
    run_rows_identical(gpulib_taps, model, seeds=tuple(range(10, 18)), gemms=('f32', 'bf16x3'), plans=(None, True))


@pytest.mark.parametrize('gemm,irb,block_fusion', [('bf16x3', None, False), ('f32', None, False), ('bf16x3', True, False), ('f32', True, False), ('f32', False, False), ('f32', None, True)])
def test_detector_gpu_every_plan_step_isolated(gpulib_taps, model, gemm, irb, block_fusion):
    """every plan step against the float64 oracle ON THE DEVICE'S OWN STEP INPUTS (2e-6 fp32 / 4e-6 bf16x3 of the blob's magnitude): each k_conv_pw3<*>, k_irb<*>,
    k_fused_block2<*>, k_se_gate, depthwise and stem instantiation of the default plan, of the all-shapes block plan, of the plan without block kernels and of the
    opt-in k_fused_block plan is judged on its own inputs"""
    from test_detector import run_steps_isolated
    worst, descs = run_steps_isolated(gpulib_taps, model, gemm=gemm, irb=irb, block_fusion=block_fusion)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print(gemm, irb, block_fusion, len(worst), 'steps; largest step errors:', top)


def test_detector_gpu_product_library_rows_identical_to_oracle(gpulib, model):
    """the PRODUCT library (no taps: its one and only plan, the default bf16x3 scheme): DetectionOutput rows and person boxes equal the oracle's fp32 run on eight images"""
    from test_detector import run_rows_identical
    run_rows_identical(gpulib, model, seeds=tuple(range(20, 28)), gemms=(None,), plans=(None,))


def test_dynamic_mask_gpu(gpulib):
    import torch
    run_mask(gpulib, to_dev=lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda(), to_host=lambda t: t.cpu().numpy())


def test_detector_gpu_fused_matches_oracle(gpulib_taps, model):
    run_compare(gpulib_taps, model, seeds=(0,), fuse=True)


def test_detector_gpu_fused_equals_unfused(gpulib_taps, model):
    from test_detector import run_fused_equals_unfused
    run_fused_equals_unfused(gpulib_taps, model)


def test_compact_gpu(gpulib):
    import torch
    from test_detector import run_compact
    run_compact(gpulib, to_dev=lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda(), to_host=lambda t: t.cpu().numpy())


def test_forward_graph_replay_equals_plain_launches(gpulib, model):
    """sgx_det_forward_batch_dev on a non-default stream captures the plan into a hipGraph and replays it: same loc / conf as individual launches."""
    import ctypes as C
    import torch
    from sg_slam_amd.capi import _vp
    from sg_slam_amd.detector import Detector2D
    from test_detector import PARAM, make_image
    layers, W, blob = model
    det = Detector2D(0.9, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=2, lib=gpulib)
    imgs = torch.from_numpy(np.stack([make_image(5), make_image(6)])).cuda()
    dl, dc = C.c_void_p(), C.c_void_p()
    nl, nc = det.num_priors * 4, det.num_priors * det.num_class
    def grab():
        torch.cuda.synchronize()
        out = []
        for ptr, n in ((dl, nl), (dc, nc)):
            host = np.zeros(2 * n, 'f4'); t = torch.zeros(2 * n, dtype=torch.float32, device='cuda')
            C.cdll.LoadLibrary('libamdhip64.so').hipMemcpy(C.c_void_p(t.data_ptr()), ptr, C.c_size_t(8 * n), 3)
            out.append(t.cpu().numpy())
        return out
    gpulib.check(gpulib.dll.sgx_det_forward_batch_dev(det.h, _vp(imgs), 640 * 3, 2, C.byref(dl), C.byref(dc), None))
    plain = grab()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    for _ in range(2):       # first call captures + launches, second replays
        gpulib.check(gpulib.dll.sgx_det_forward_batch_dev(det.h, _vp(imgs), 640 * 3, 2, C.byref(dl), C.byref(dc), C.c_void_p(s.cuda_stream)))
    s.synchronize()
    graph = grab()
    for a, b in zip(plain, graph):
        assert (a == b).all()
    det.close()


def test_detect_dev_gpu(gpulib, model):
    import torch
    from test_detector import run_detect_dev
    run_detect_dev(gpulib, model, to_dev=lambda a: torch.from_numpy(a).cuda(), to_host=lambda t: t.cpu().numpy())


def test_detection_output_stress_gpu(gpulib_taps, model):
    from test_detector import run_detection_output_stress
    run_detection_output_stress(gpulib_taps, model)


def test_xcd_work_order_is_a_permutation_of_the_work(gpulib_taps, model):
    """The tuned kernels deal frames out to the eight XCDs (sgx_xcd_order): a batch of 19 frames (two full groups of eight + three frames in the plain order)
    gives, frame by frame, the bytes the batch-of-two plan gives (which run_compare pins to the oracle), and the same with the plain order (SGX_DET_XCD=0)."""
    import os
    from sg_slam_amd.detector import Detector2D
    from test_detector import PARAM, make_image
    layers, W, blob = model
    imgs = np.stack([make_image(s) for s in range(19)])

    def run(batch, env):
        old = os.environ.get('SGX_DET_XCD')
        if env is None: os.environ.pop('SGX_DET_XCD', None)
        else: os.environ['SGX_DET_XCD'] = env
        try:
            det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=batch, lib=gpulib_taps, fuse=True)
        finally:
            if old is None: os.environ.pop('SGX_DET_XCD', None)
            else: os.environ['SGX_DET_XCD'] = old
        outs = []
        for i in range(0, 19, batch):
            n = min(batch, 19 - i)
            det.detect_batch(imgs[i:i + n])
            outs += [(det.debug_blob('mbox_loc', b).copy(), det.debug_blob('mbox_conf_softmax', b).copy()) for b in range(n)]
        det.close()
        return outs

    ref = run(2, None)
    for env in (None, '0'):
        got = run(19, env)
        for f in range(19):
            assert (got[f][0] == ref[f][0]).all() and (got[f][1] == ref[f][1]).all(), (env, f)


def test_forked_graph_capture_equals_plain_launches_gpu(gpulib_taps, model, tmp_path):
    """capture_forked (sgx_det.cpp; a tap since round 5: SGX_DET_FORK=3, SGX_DET_EXECS=2): the plan captured with its parallel branches — dependencies from the blobs every step
    reads and writes, cross-lane edges as capture events — replays to the same bytes as the plan launched step by step.  The switch is read once per process: a subprocess."""
    import subprocess, sys, os
    from test_detector import PARAM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'fork.py'
    script.write_text(f'''
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, "tests")!r}); sys.path.insert(0, {os.path.join(root, "tools")!r})
from _campaign_lib import taps_lib
from sg_slam_amd import synth
from sg_slam_amd.capi import _vp
from sg_slam_amd.detector import Detector2D
from test_detector import make_image
lib = taps_lib()
layers = synth.parse_ncnn_param({PARAM!r}); _, blob = synth.synth_ncnn_weights(layers, seed=7)
det = Detector2D(0.9, 0.01, param_text=open({PARAM!r}).read(), bin_bytes=blob, max_batch=3, lib=lib)
imgs = torch.from_numpy(np.stack([make_image(s) for s in (5, 6, 7)])).cuda()
dl, dc = C.c_void_p(), C.c_void_p()
nl, nc = det.num_priors * 4, det.num_priors * det.num_class
hip = C.cdll.LoadLibrary('libamdhip64.so')
def grab():
    torch.cuda.synchronize(); out = []
    for ptr, n in ((dl, nl), (dc, nc)):
        t = torch.zeros(3 * n, dtype=torch.float32, device='cuda'); hip.hipMemcpy(C.c_void_p(t.data_ptr()), ptr, C.c_size_t(12 * n), 3); out.append(t.cpu().numpy())
    return out
lib.check(lib.dll.sgx_det_forward_batch_dev(det.h, _vp(imgs), 640 * 3, 3, C.byref(dl), C.byref(dc), None))
plain = grab()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
for rep in range(3):
    lib.check(lib.dll.sgx_det_forward_batch_dev(det.h, _vp(imgs), 640 * 3, 3, C.byref(dl), C.byref(dc), C.c_void_p(s.cuda_stream)))
    s.synchronize()
    g = grab()
    assert all((a == b).all() for a, b in zip(plain, g)), rep
print("FORK_OK", float(np.abs(plain[0]).max()))
''')
    env = dict(os.environ, SGX_DET_FORK='3', SGX_DET_EXECS='2')
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'FORK_OK' in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_detector_gpu_irb3_block_kernel_steps_isolated(gpulib_taps, tmp_path):
    """k_irb3 — the inverted-residual block kernel with EVERY matrix product as bf16x3 (built in round 4, slower than k_irb, opt-in through the tap SGX_DET_IRB3=1; the product
    never selects it): each of its plan steps against the float64 oracle on the device's own step inputs, like the default plans.  The switch is read once per process: a subprocess."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'irb3.py'
    script.write_text(f'''
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, "tests")!r}); sys.path.insert(0, {os.path.join(root, "tools")!r})
import torch
from _campaign_lib import taps_lib
from oracle import detector_oracle as D
import test_detector as T
layers = D.parse_param(T.PARAM); W, blob = D.synth_weights(layers, seed=7)
worst, descs = T.run_steps_isolated(taps_lib(), (layers, W, blob), gemm='bf16x3', irb=None)
n3 = sum(1 for d in descs if d.startswith('irb') and d.rstrip().endswith('bf16x3'))
print('IRB3_OK', n3, len(worst), max(worst.values()))
assert n3 >= 6
''')
    env = dict(os.environ, SGX_DET_IRB3='1')
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'IRB3_OK' in out.stdout, (out.stdout[-800:], out.stderr[-2500:])
