"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-28s %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
