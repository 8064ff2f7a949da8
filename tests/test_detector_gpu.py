"""GPU (MI355X): detector forward (fp32-MFMA pointwise convolutions) + post-processing + dynamic mask vs the oracle."""
import numpy as np
import pytest
from test_detector import run_compare, run_mask, model   # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_detector_gpu_matches_oracle(gpulib, model):
    run_compare(gpulib, model)


def test_dynamic_mask_gpu(gpulib):
    import torch
    run_mask(gpulib, to_dev=lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda(), to_host=lambda t: t.cpu().numpy())


def test_detector_gpu_fused_matches_oracle(gpulib, model):
    run_compare(gpulib, model, seeds=(0,), fuse=True)


def test_detector_gpu_fused_equals_unfused(gpulib, model):
    from test_detector import run_fused_equals_unfused
    run_fused_equals_unfused(gpulib, model)


def test_compact_gpu(gpulib):
    import torch
    from test_detector import run_compact
    run_compact(gpulib, to_dev=lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda(), to_host=lambda t: t.cpu().numpy())
