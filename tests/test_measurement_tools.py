"""CPU: the measurement plumbing that bench.py and tools/collect_profiles.sh rely on.
* every kernel of the per-frame chain has a class in tools/pmc_classes.py (detector kernels by membership of the sgx_det*.h headers), so that no rocprof kernel
  time can fall into class None unnoticed;
* bench.py --gpus N without a torchrun environment builds the right re-exec command (one rank per GPU, 127.0.0.1 rendezvous)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_every_chain_kernel_has_a_class():
    import pmc_classes as pc
    ks = pc.source_kernels()
    assert len(ks) > 60
    chain_files = ('sgx_orb_kernels.h', 'sgx_match_kernels.h', 'sgx_poseopt_kernels.h', 'sgx_det_kernels.h', 'sgx_det_block.h', 'sgx_det_irb.h', 'sgx_det_bf16.h')
    flow_taps = ('k_flow_affine',)                       # test tap of the mask kernels, never launched by the tracker
    missing = [k for k, f in ks.items() if f in chain_files and k not in flow_taps and pc.classify(k) is None]
    assert not missing, missing
    for k in ('k_lk_copy', 'k_lk_pyrdown', 'k_lk_track', 'k_lk_trackN', 'k_fm_ransac'):
        assert pc.classify(k) is not None
    det = [k for k, f in ks.items() if f.startswith('sgx_det')]
    assert 'k_irb' in det and 'k_se_gate' in det
    for k in det:
        assert pc.classify(k) in ('det_forward', 'det_output', 'dynamic_mask'), k
    assert pc.classify('void k_irb<3, 2, true>(SgxIrb)') == 'det_forward'


def test_unclassified_share_flags_unknown_kernels():
    import pmc_classes as pc
    rows = [{'Name': 'k_fast_cells(int)', 'TotalDurationNs': '90'}, {'Name': 'k_brand_new(int)', 'TotalDurationNs': '10'}, {'Name': '__amd_rocclr_copyBuffer', 'TotalDurationNs': '1000'}]
    share, names = pc.unclassified_share(rows)
    assert abs(share - 0.1) < 1e-12 and names == ['k_brand_new']


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_bench_gpus_builds_a_torchrun_command():
    b = _bench()
    cmd, env = b.spawn_command(8, ['--gpus', '8', '--steps', '20', '--warmup', '3'], environ={})
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '20', '--warmup', '3'] and os.path.basename(cmd[-7]) == 'bench.py'
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and env['SGX_BENCH_SPAWNED'] == '1'
    # under torchrun (WORLD_SIZE present) nothing is spawned
    assert b.needs_spawn(8, {'WORLD_SIZE': '8'}) is False and b.needs_spawn(8, {}) is True and b.needs_spawn(1, {}) is False
    assert b.needs_spawn(1, {'SGX_BENCH_FORCE_SPAWN': '1'}) is True


def test_isa_census_counts_select_runs_and_lane_reads():
    """tools/isa_census.py: a run of v_cndmask_b32_e32 is broken by a vector / memory instruction, not by scalar instructions or s_nop (profiles/r4_ubench_snop_cost*.txt)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import isa_census
    asm = """_Z3foov:
	v_cmp_lt_u32 vcc, v0, v1
	s_nop 1
	v_cndmask_b32_e32 v1, v1, v2, vcc
	s_mov_b32 s0, 0
	v_cndmask_b32_e32 v3, v3, v2, vcc
	v_cndmask_b32_e32 v4, v4, v2, vcc
	v_mov_b32_e32 v5, v1
	v_cndmask_b32_e32 v4, v4, v2, vcc
	v_readlane_b32 s1, v9, 3
	s_cbranch_vccnz .LBB0_1
	s_endpgm
_Z3barv:
	v_cndmask_b32_e64 v1, 0, v2, s[2:3]
	s_endpgm
"""
    c = isa_census.census(asm)
    assert c['_Z3foov'] == dict(valu=7, cnd_e32=4, runs={3: 1, 1: 1}, lanes=1, s_nop=1, branches=1)
    assert c['_Z3barv']['cnd_e32'] == 0 and c['_Z3barv']['valu'] == 1 and not c['_Z3barv']['runs']
