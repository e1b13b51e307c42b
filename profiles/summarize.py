#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv / kernel_trace.csv summary (helper for profiles/)."""
import csv, sys, glob
d = sys.argv[1]
st = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(st)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:n]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f} pct={r['Percentage']}")
if len(sys.argv) > 3:
    tr = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    xs = [(int(r['Start_Timestamp']), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][:70]) for r in csv.DictReader(open(tr)) if sys.argv[3] in r['Kernel_Name']]
    xs.sort()
    for x in xs[-int(sys.argv[4]) if len(sys.argv) > 4 else -12:]: print(f"{x[1]:10.1f} us  {x[2]}")
