#!/usr/bin/env python3
"""Print the kernel sequence of the LAST bench step from a rocprofv3 kernel_trace.csv (name, duration in us)."""
import csv, glob, sys
tr = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(tr))]
rows.sort()
# a step starts at its first spmm_rows_kernel<32> (layer-1 aggregation)
starts = [i for i, r in enumerate(rows) if 'spmm_rows_kernel<32>' in r[2]]
lo = starts[-1]
hi = len(rows)
t0 = rows[lo][0]
for s, e, n in rows[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n[:110]}")
