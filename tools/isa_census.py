"""Static census of the product's device code (no GPU needed): per kernel the vector-instruction count, v_cndmask_b32_e32 selects and their run lengths (a run of three or more
issues one per 20 cycles on gfx950, profiles/r4_ubench_snop_cost*.txt), v_readlane / v_writelane (scalar registers spilled into vector lanes), s_nop and scalar branches.
usage: python tools/isa_census.py [sgx_det sgx_flow ...]     (default: every sg_slam_amd/csrc/sgx_*.cpp; compiles each with hipcc --cuda-device-only -S, ~1-2 min per file)"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); CSRC = os.path.join(ROOT, 'sg_slam_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt', '-fno-slp-vectorize', '-Wno-everything', '-x', 'hip', '--cuda-device-only', '-S']

def census(asm_text):
    """{kernel: dict(valu, cnd_e32, runs{len: count}, lanes, s_nop, branches)} from the text of a gfx950 assembly file"""
    out, k, run = {}, None, 0
    def flush():
        nonlocal run
        if k is not None and run: out[k]['runs'][run] += 1
        run = 0
    for line in asm_text.split('\n'):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            flush(); k = m.group(1); out[k] = dict(valu=0, cnd_e32=0, runs=Counter(), lanes=0, s_nop=0, branches=0); continue
        t = line.strip()
        if k is None or not t or t.startswith(';'): continue
        op = t.split()[0]
        if op == 's_endpgm': flush(); k = None; continue
        d = out[k]
        if op == 's_nop': d['s_nop'] += 1
        if op.startswith('s_cbranch') or op == 's_branch': d['branches'] += 1
        breaks = op.startswith(('v_', 'ds_', 'global_', 'buffer_', 'scratch_', 'flat_', 's_cbranch', 's_branch')) or op.endswith(':')   # scalar ALU ops and s_nop do not break a run
        if op.startswith('v_'):
            d['valu'] += 1
            if op in ('v_readlane_b32', 'v_writelane_b32'): d['lanes'] += 1
        if op == 'v_cndmask_b32_e32': d['cnd_e32'] += 1; run += 1
        elif breaks: flush()
    flush()
    return out

def main():
    names = sys.argv[1:] or sorted(f[:-4] for f in os.listdir(CSRC) if f.startswith('sgx_') and f.endswith('.cpp'))
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    with tempfile.TemporaryDirectory() as tmp:
        for n in names:
            s = os.path.join(tmp, n + '.s')
            r = subprocess.run([hipcc] + FLAGS + ['-I' + os.path.join(ROOT, 'include'), n + '.cpp', '-o', s], cwd=CSRC, capture_output=True, text=True)
            if r.returncode: print(n, 'did not compile:', r.stderr[-300:]); continue
            print('==', n)
            for k, d in sorted(census(open(s).read()).items()):
                slow = {L: c for L, c in sorted(d['runs'].items()) if L >= 3}
                print(f"{k[:72]:72s} valu {d['valu']:6d}  cnd_e32 {d['cnd_e32']:4d}  runs>=3 {str(slow) if slow else '-':14s} lanes {d['lanes']:5d}  s_nop {d['s_nop']:4d}  branches {d['branches']:4d}")

if __name__ == '__main__':
    main()
