#!/usr/bin/env python3
"""development aid: from a rocprofv3 --kernel-trace csv directory of bench.py, print the kernel sequence (start, duration,
gap to the previous kernel's end, name) around the loss / hook section of the last traced step"""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "softmax_kl_fwd" in r[2]]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
lo, hi = max(0, i0 - 45), min(len(rows), i0 + 75)
t0 = rows[lo][0]
prev_end = rows[lo][0]
for s, e, n in rows[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n[:100]}")
    prev_end = max(prev_end, e)
