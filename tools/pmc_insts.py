"""Build profiles/<name>_pmc_insts.json (read by bench.py for valu_frac / fp64_frac) from rocprofv3 --pmc passes.
usage: pmc_insts.py out.json frames_per_launch steps_profiled sq_pass.csv [more.csv ...] [--det det_sq.csv det_frames det_reps]
Every CSV is a counter_collection.csv; counters are summed over all dispatches of a class and divided by frames x steps.  SQ_INSTS_* are
wave-level instruction counts (one per wave64 instruction issued).  fp64 flops per frame assume all 64 lanes active: 64 x (ADD + MUL + 2 FMA)."""
import csv, json, sys, collections
from pmc_classes import classify, steps_ran, base_name

def collect(path):
    """per class: counter totals over all dispatches, and the number of steps in which the class ran (from the dispatch counts of its kernels)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for r in csv.DictReader(open(path)):
        cls = classify(r['Kernel_Name'])
        if cls is None: continue
        acc[cls][r['Counter_Name']] += float(r['Counter_Value']); disp[cls][base_name(r['Kernel_Name'])].add(r['Dispatch_Id'])
    ran = {cls: steps_ran({k: len(v) for k, v in d.items()}) for cls, d in disp.items()}
    return acc, ran

args = sys.argv[1:]
det = None
if '--det' in args:
    i = args.index('--det'); det = (args[i + 1], int(args[i + 2]), int(args[i + 3])); args = args[:i]
out, S, steps = args[0], int(args[1]), int(args[2])
tot = collections.defaultdict(dict)
for p in args[3:]:
    acc, ran = collect(p)
    for cls, cs in acc.items():
        if cls.startswith('det_') and det: continue
        for c, v in cs.items(): tot[cls][c] = v / (S * min(steps, ran[cls]))
if det:
    acc, _ = collect(det[0])
    for cls, cs in acc.items():
        if not cls.startswith('det_'): continue
        for c, v in cs.items(): tot[cls][c] = v / (det[1] * det[2])
kern = {}
for cls, cs in sorted(tot.items()):
    e = {'valu_insts_per_frame': round(cs.get('SQ_INSTS_VALU', 0.0), 1)}
    for c, key in (('SQ_INSTS_SALU', 'salu_insts_per_frame'), ('SQ_INSTS_LDS', 'lds_insts_per_frame'), ('SQ_INSTS_VMEM_RD', 'vmem_rd_insts_per_frame'), ('SQ_WAVES', 'waves_per_frame'),
                   ('SQ_INSTS_VALU_MFMA_MOPS_F32', 'mfma_mops_f32_per_frame'), ('SQ_WAVE_CYCLES', 'wave_cycles_per_frame'), ('SQ_BUSY_CYCLES', 'busy_cycles_per_frame'),
                   ('SQ_WAIT_INST_ANY', 'wait_inst_any_per_frame'), ('SQ_ACTIVE_INST_VALU', 'active_inst_valu_per_frame'), ('SQ_LDS_BANK_CONFLICT', 'lds_bank_conflict_per_frame')):
        if c in cs: e[key] = round(cs[c], 1)
    f64 = [cs.get('SQ_INSTS_VALU_ADD_F64'), cs.get('SQ_INSTS_VALU_MUL_F64'), cs.get('SQ_INSTS_VALU_FMA_F64')]
    if any(v is not None for v in f64):
        a, m, f = (v or 0.0 for v in f64)
        e['fp64_insts_per_frame'] = round(a + m + f + cs.get('SQ_INSTS_VALU_TRANS_F64', 0.0), 1)
        e['fp64_gflop_per_frame'] = 64.0 * (a + m + 2.0 * f) / 1e9
    kern[cls] = e
json.dump({'note': 'wave-level instruction counts per frame from rocprofv3 --pmc passes over bench.py (--no-detector; detector classes from tools/prof_det_ops.py), summed over all '
                   'dispatches of a class / (frames x steps).  cycles_per_valu_inst = the chip-wide issue cost of the instruction mix these kernels use, measured by '
                   'tools/ubench/valu_issue*.hip (2.1-2.5 cycles for fp32 fma/add/mul, int add/sub/logic/right shifts, mov; 4.1 for everything else); 3.0 is the mix average '
                   'used for valu_frac.  fp64 flops assume 64 active lanes.',
           'frames_per_launch': S, 'steps_profiled': steps, 'cycles_per_valu_inst': 3.0, 'clock_ghz': 2.4, 'kernels': kern}, open(out, 'w'), indent=1)
for k, e in kern.items(): print(k, e)
