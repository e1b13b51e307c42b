#!/usr/bin/env python3
"""Per-kernel mean / min / max of the counters in a rocprofv3 --pmc run (counter_collection.csv)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
per = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    per[(r['Dispatch_Id'], r['Kernel_Name'], r['Counter_Name'])] += float(r['Counter_Value'])
for (_, k, c), v in per.items():
    k = k.replace('void ', '').replace('(anonymous namespace)::', '')
    acc[(k[:44], c)].append(v)
for (k, c), v in sorted(acc.items(), key=lambda t: (t[0][1], t[0][0])):
    print(f"{k:46s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
