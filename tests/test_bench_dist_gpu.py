"""GPU (MI355X): the multi-GPU code path of bench.py on ONE GPU — torch.distributed over RCCL with world_size 1 (SGX_BENCH_FORCE_DIST=1): process-group set-up, the
one-kernel record pack on the side stream, the per-step gather to rank 0 inside the timed region, the max-over-ranks reduce.  The driver only has 8-GPU nodes now
and then; this keeps the path exercised every round (the 2-rank semantics are covered on CPU by tests/test_dist_gloo.py)."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0)); return so.getsockname()[1]


def test_bench_rccl_gather_path_single_rank(gpulib):
    env = dict(os.environ, SGX_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '3', '--streams', '8', '--no-cpu-baseline', '--no-config2', '--no-config4',
                          '--no-host-input'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    j = json.loads(line)
    g = j['config']['frame_record_gather']
    assert j['n_gpus'] == 1 and j['value'] > 0 and g is not None and g['inside_timed_region'] and g['records_per_step'] == 8 and g['bytes_per_step'] == 0      # one rank: its own slice is a local copy, nothing arrives over a link
    assert j['config']['tracked_streams_last_frame'] == 8


def test_bench_gpus_flag_spawns_the_ranks(gpulib):
    """`python bench.py --gpus N` with no torchrun environment launches N ranks itself (torch.distributed.run, 127.0.0.1 rendezvous).  A 1-GPU box can
    only run N = 1 through that launcher (SGX_BENCH_FORCE_SPAWN=1 makes --gpus 1 take the same route as --gpus 8); a request for more GPUs than the node has fails loudly."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(SGX_BENCH_FORCE_SPAWN='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '6', '--warmup', '3', '--streams', '8', '--no-cpu-baseline', '--no-config2',
                          '--no-config4', '--no-host-input'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert j['n_gpus'] == 1 and j['value'] > 0 and j['config']['frame_record_gather'] is not None       # it ran as a torch.distributed rank
    import torch
    n = torch.cuda.device_count() + 1
    env.pop('SGX_BENCH_FORCE_SPAWN')
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1', '--streams', '8', '--no-cpu-baseline', '--no-config2',
                          '--no-config4', '--no-host-input'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith('{')]
