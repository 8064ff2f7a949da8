"""Write the two PMC summaries kept under profiles/ (pmc_kernels.md, pmc_detector.md) from the per-kernel counter means tools/collect_profiles.sh leaves in a run directory.
usage: pmc_markdown.py <run dir> <frames per launch>"""
import os, re, json, sys, collections
O, S = sys.argv[1], int(sys.argv[2])

def parse(fn):
    d = collections.OrderedDict(); cur = None
    try:
        for l in open(fn):
            if not l.startswith(' '):
                cur = l.strip(); d[cur] = {}
            else:
                m = re.match(r'\s+(\S+)\s+([\d.]+)\s+\(n=(\d+)\)', l)
                if m: d[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    except FileNotFoundError:
        pass
    return d

def M(x): return '%.2f M' % (x / 1e6) if x >= 1e5 else '%.0f' % x
g = lambda d, c: d.get(c, (0, 0))[0]
a, b, f, w = (parse(O + '/pmc_%s.txt' % n) for n in ('sq_a', 'sq_b', 'fetch', 'write'))
try: st = json.load(open(O + '/standalone.json'))['avg_ms_per_launch']
except Exception: st = {}
issue = lambda valu: valu * 4.0 / (1024 * 2.4e9) * 1e3            # ms of pure issue time at 4 cycles per wave instruction on 1 024 SIMDs at 2.4 GHz
TAG = os.path.basename(os.path.normpath(sys.argv[1]))          # the collection's tag (tools/collect_profiles.sh <tag>) names the round: no round number hard-coded here
out = ['# %s — PMC evidence for the kernels of the tracking chain (MI355X, %d frames per launch)\n' % (TAG, S),
       'Collected by `tools/collect_profiles.sh` with separate `rocprofv3 --pmc … --kernel-trace` passes over `bench.py --no-detector --no-config2 --steps 3 --warmup 1` (no sys / runtime trace',
       'domains; FETCH_SIZE and WRITE_SIZE each in a pass of their own), per the recipe in `MI355X_MICROARCH.md`.  Values are means per launch over the dispatches of the profiled steps (the',
       'tracking-stage kernels run in 3 of the 4 steps: the first frame of a run has no predecessor; `tools/pmc_insts.py` / `pmc_traffic.py` normalise per class by the steps in which it ran).',
       'SQ_INSTS_* count wave-level instructions.  FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so the "read" column doubles it.',
       '"issue ms" = VALU instructions × 4 cycles ÷ (1 024 SIMDs × 2.4 GHz): the time the launch needs for instruction issue alone at the price `tools/ubench/valu_issue*.hip` measures for the',
       'integer / packed / DPP instructions these kernels are made of (≈ 2.2–2.6 cycles only for fp32 fma / add / mul, integer add / sub, bitwise ops, right shifts, moves).\n',
       '| kernel | waves | VALU | issue ms | SALU | LDS | VMEM rd | fp64 (add+mul+fma) | SQ_WAVE_CYCLES | SQ_WAIT_INST_ANY | LDS bank-conflict cycles | read MB | written MB |',
       '|---|---|---|---|---|---|---|---|---|---|---|---|---|']
for k in a:
    if not (k.startswith('k_') or k.startswith('void k_')): continue
    A, B, F, W = a[k], b.get(k, {}), f.get(k, {}), w.get(k, {})
    f64 = g(B, 'SQ_INSTS_VALU_ADD_F64') + g(B, 'SQ_INSTS_VALU_MUL_F64') + g(B, 'SQ_INSTS_VALU_FMA_F64')
    out.append('| `%s` | %s | %s | %.3f | %s | %s | %s | %s | %s | %s | %s | %.1f | %.1f |' % (k.replace('void ', ''), M(g(A, 'SQ_WAVES')), M(g(A, 'SQ_INSTS_VALU')), issue(g(A, 'SQ_INSTS_VALU')),
               M(g(A, 'SQ_INSTS_SALU')), M(g(A, 'SQ_INSTS_LDS')), M(g(A, 'SQ_INSTS_VMEM_RD')), M(f64), M(g(A, 'SQ_WAVE_CYCLES')), M(g(B, 'SQ_WAIT_INST_ANY')), M(g(B, 'SQ_LDS_BANK_CONFLICT')),
               2 * g(F, 'FETCH_SIZE') / 1024, g(W, 'WRITE_SIZE') / 1024))
if st:
    out.append('\nStandalone launch durations (nothing else on the GPU, `standalone.json`), ms per %d frames: ' % S + ', '.join('%s %.3f' % kv for kv in st.items()) + '.\n')
out.append('''Reading.  Every heavy kernel of the chain is **bound by VALU issue, not by memory**: compare the "issue ms" column with the standalone durations (LK tracker, `k_fast_cells`, blur + descriptor
kernels: issue time is 80–95 % of the launch).  `bench.py` turns the counts into the `valu_frac` column of its per-kernel table (`pmc_insts.json`, 3.0 cycles per instruction as the mix
average — a lower bound for the integer kernels).  HBM-side traffic equals the algorithmic bytes of DESIGN.md §4 / §4b for the pyramid, FAST and LK kernels; the blur + descriptor pair moves
3.9 MB per frame (the blurred pyramid makes a round trip and the descriptor kernel gathers from both planes) against the 1.0 MB a fused kernel would need — a deliberate trade on a VALU-bound
stage (DESIGN.md §4).  `k_pose_opt` spends about two thirds of its VALU instructions in fp64 (`fp64_frac` in the bench table prices them against the 78.6 TFLOP/s vector peak: ≈ 0.02 — the kernel
is latency-bound, one wave per SIMD).''')
open(O + '/pmc_kernels.md', 'w').write('\n'.join(out) + '\n')

ds, dm, df, dw = (parse(O + '/pmc_%s.txt' % n) for n in ('det_sq', 'det_mfma', 'det_fetch', 'det_write'))
out = ['# %s — PMC evidence for the detector kernels (MI355X, batch %d, every plan step launched on its own: `tools/prof_det_ops.py %d 2`)\n' % (TAG, S, S),
       'Means per launch over all launches of a kernel instantiation (n = launches averaged; 3 launches per plan step).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 × SQ_BUSY_CU_CYCLES):',
       'share of the four matrix pipes of a CU that was busy while the CU was.  read MB = 2 × FETCH_SIZE (gfx950 correction), written MB = WRITE_SIZE.\n',
       '| kernel | n | waves | VALU | SALU | LDS | VMEM rd | MFMA MOPS f32 | MFMA MOPS bf16 | MFMA busy | read MB | written MB |', '|---|---|---|---|---|---|---|---|---|---|---|---|']
for k in ds:
    if not (k.startswith('k_') or k.startswith('void k_')): continue
    A, Mx, F, W = ds[k], dm.get(k, {}), df.get(k, {}), dw.get(k, {})
    busy = g(Mx, 'SQ_VALU_MFMA_BUSY_CYCLES') / (4 * g(Mx, 'SQ_BUSY_CU_CYCLES')) if g(Mx, 'SQ_BUSY_CU_CYCLES') else 0
    out.append('| `%s` | %d | %s | %s | %s | %s | %s | %s | %s | %.0f %% | %.1f | %.1f |' % (k.replace('void ', ''), A.get('SQ_WAVES', (0, 0))[1], M(g(A, 'SQ_WAVES')), M(g(A, 'SQ_INSTS_VALU')), M(g(A, 'SQ_INSTS_SALU')),
               M(g(A, 'SQ_INSTS_LDS')), M(g(A, 'SQ_INSTS_VMEM_RD')), M(g(A, 'SQ_INSTS_VALU_MFMA_MOPS_F32')), M(g(Mx, 'SQ_INSTS_VALU_MFMA_MOPS_BF16')), 100 * busy, 2 * g(F, 'FETCH_SIZE') / 1024, g(W, 'WRITE_SIZE') / 1024))
try: tj = json.load(open(O + '/traffic.json'))['bytes_per_launch']
except Exception: tj = {}
out.append('\nWhole forward (pre-processing + 102 plan steps), %d frames: %.1f GB of HBM-side traffic (`traffic.json`) against %.1f GB of algorithmic activation + weight bytes summed over the steps '
           '(`detector_ops.txt`): the input of every step comes from HBM again (at this batch size nothing survives in the Infinity Cache from one step to the next), nothing is re-read.  Standalone: forward %.2f ms, DetectionOutput + filtering %.2f ms '
           '(`standalone.json`).\n' % (S, tj.get('det_forward', 0) / 1e9, 27.2 * S / 256, st.get('det_forward', 0), st.get('det_output', 0)))
out.append('''Reading.  The pointwise kernels keep the fp32 matrix pipes 30–60 % busy (the deep-K 19 × 19 / 10 × 10 layers most), the depthwise and stem kernels are VALU / LDS work.  The per-step
table prices every step against max(bytes / 8 TB/s, flops / 157.3 TFLOP/s); `tools/ubench/stream_bw.hip` shows that the pointwise access shape (a wave reads two 128-byte row segments per
load) reaches 5.2 TB/s with nothing else in the kernel and a plain 16-byte-per-lane copy 6.0–6.1 TB/s, so "0.39 of the step rooflines" is ≈ 0.55 of what the access shapes can attain on this chip.''')
open(O + '/pmc_detector.md', 'w').write('\n'.join(out) + '\n')
print('wrote', O + '/pmc_kernels.md', O + '/pmc_detector.md')
