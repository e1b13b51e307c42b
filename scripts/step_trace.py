"""Measurement helper: the kernel sequence of ONE training step (the last full one) from a rocprofv3 --kernel-trace CSV,
with the idle gap in front of every kernel (window: from one quantiser launch to the next).  Usage: python scripts/step_trace.py <kernel_trace.csv> <out.txt>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = [i for i, r in enumerate(rows) if "vq_main" in r["Kernel_Name"]]  # one per step: a step-long window, cyclically shifted
a, b = mark[-2], mark[-1]
prev, busy, idle = None, 0.0, 0.0
with open(sys.argv[2], "w") as out:
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        busy += (e - s) / 1e3
        idle += max(gap, 0.0)
        out.write(f"{gap:8.1f} us gap | {(e - s) / 1e3:9.1f} us | {r['Kernel_Name'][:110]}\n")
        prev = e
    out.write(f"# {b - a} kernels, busy {busy / 1e3:.3f} ms, idle between kernels {idle / 1e3:.3f} ms\n")
