"""Summarise a rocprofv3 counter_collection CSV: mean counter value per kernel name."""
import collections, csv, sys
path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        k = r['Kernel_Name']
        if filt and filt not in k:
            continue
        # distinguish instantiations by grid size too
        key = (k[:70], r.get('Grid_Size', ''))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, cs in acc.items():
    print(key)
    for c, v in sorted(cs.items()):
        print(f'   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}')
