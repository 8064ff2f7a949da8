"""bench.py ITSELF with two ranks: the driver's multi-GPU launch line — python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 — on the
tests' kernel-logic emulator with the gloo backend (SGX_BENCH_EMU_TEST=1: the only thing that switch changes is library, device and backend; rank set-up, stream sharding,
the per-step record gather to rank 0, max-over-ranks timing and the JSON line are the code the 8-GPU run executes).  Tiny streams / steps; the line is labelled as not a measurement."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0)); return so.getsockname()[1]


def _run(nproc, extra, gpus_flag=True):
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'sg_slam_amd', 'csrc'), 'emu'])
    env = dict(os.environ, SGX_BENCH_EMU_TEST='1', OMP_NUM_THREADS='2', OPENBLAS_NUM_THREADS='2', MKL_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py')] + (['--gpus', str(nproc)] if gpus_flag else []) + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]          # exactly one JSON line, printed by rank 0
    return json.loads(lines[0])


ARGS = ['--steps', '2', '--warmup', '1', '--streams', '2', '--frames', '3', '--no-detector', '--no-config2', '--no-config4', '--no-host-input', '--no-cpu-baseline']


def test_bench_two_ranks_gloo():
    out = _run(2, ARGS)
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['warmup'] == 1 and out['scaling'] == 'weak' and out['higher_is_better'] is True
    assert 'NOT A MEASUREMENT' in out['data']
    c = out['config']
    S, world = 2, 2
    assert c['streams_per_gpu'] == S and c['parallelism'] == 'streams-sharded x2'
    # whole-job value: all ranks' frames over the max-over-ranks time of the timed region
    assert abs(out['value'] - S * world * out['steps'] / (out['ms_per_step'] * out['steps'] / 1e3)) < 1e-6 * out['value']
    g = c['frame_record_gather']
    rec = 16 + 1024 * 28 + 1024 * 32 + 64          # header + cv::KeyPoint[cap] + descriptors[cap][32] + pose, cap = 1024 at 1000 features (sgx_tracker_record_bytes)
    assert g['record_bytes'] == rec and g['records_per_step'] == world * S and g['inside_timed_region'] is True
    assert g['bytes_per_step'] == (world - 1) * S * rec          # what ARRIVES at rank 0 per step: the other rank's records (its own slice is a local copy)
    assert abs(g['GBs_into_rank0_over_xgmi'] - g['bytes_per_step'] * out['steps'] / (out['ms_per_step'] * out['steps'] / 1e3) / 1e9) < 1e-9
    # bench.py asserts on its own, per rank: receive buffers exist on rank 0 only, and rank 0's slice of the last gathered step equals its own tracker's read-back
    assert c['tracked_streams_last_frame'] == world * S          # summed over ranks: every stream of both ranks still tracks
    assert c['ate_rmse_m_vs_ground_truth'] < 0.01 and c['mean_keypoints'] > 800


def test_bench_rank_without_gpus_flag_adopts_world_size():
    """`torchrun --nproc-per-node N bench.py` without an explicit --gpus N is a valid launch (the default follows WORLD_SIZE)"""
    out = _run(1, ARGS + ['--scene', 'dynamic'], gpus_flag=False)          # also: the scene with an independently moving object (synth.DynamicStream)
    assert out['n_gpus'] == 1 and out['config']['frame_record_gather'] is None
    c = out['config']
    assert 'DynamicStream' in c['workload'] and c['tracked_streams_last_frame'] == 2 and c['mean_keypoints'] < c['mean_keypoints_before_mask'] - 50      # the mask erased the walker's keypoints


def test_bench_explicit_gpus_mismatch_fails_loudly():
    env = dict(os.environ, SGX_BENCH_EMU_TEST='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + ARGS, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 2 and 'WORLD_SIZE=1' in p.stderr
