#!/usr/bin/env python3
"""Per-kernel sums of ONE counter over the LAST FULL training step of a rocprofv3 --pmc run of bench.py (window: from one
quantiser launch to the next, as scripts/step_trace.py).  Usage: pmc_step.py <rocprof output dir> <counter> <out.json>"""
import collections
import csv
import glob
import json
import sys

d, counter, out = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != counter:
        continue
    k = int(r["Dispatch_Id"])
    name = r["Kernel_Name"]
    if k not in per:
        per[k] = [name, 0.0]
    per[k][1] += float(r["Counter_Value"])
ids = sorted(per)
marks = [i for i in ids if "vq_main" in per[i][0]]
a, b = marks[-2], marks[-1]
acc = collections.OrderedDict()
for i in ids:
    if a <= i < b:
        name = per[i][0].replace("void ", "").replace("(anonymous namespace)::", "")
        short = name.split("(")[0][:90]
        e = acc.setdefault(short, [0, 0.0])
        e[0] += 1
        e[1] += per[i][1]
json.dump({"counter": counter, "unit": "KiB as rocprofv3 reports FETCH_SIZE / WRITE_SIZE", "kernels": {k: {"launches": v[0], "sum": v[1]} for k, v in acc.items()}},
          open(out, "w"), indent=1)
