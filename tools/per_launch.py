#!/usr/bin/env python3
"""durations (us) of the last N launches of the kernels whose name contains a pattern, from a rocprofv3 kernel trace:
    python tools/per_launch.py <profile dir> <pattern> [N]"""
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]][-n:]
print(" ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}" for r in rows))
print(" ".join(str(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows))
