#!/usr/bin/env python3
"""the kernels of the last step of a rocprofv3 kernel trace, in start order, with gap / duration / workgroups:
    python tools/step_dump.py <profile dir> <launches per step>"""
import csv
import glob
import re
import subprocess
import sys

d, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
names = sorted(set(r["Kernel_Name"] for r in rows))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
m = dict(zip(names, dem))
t0 = prev = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void ", "", m[r["Kernel_Name"]])
    name = re.sub(r"\(.*", "", name)[:100]
    wg = 1
    for ax in "XYZ":
        wg *= int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"]))
    print(f"{(s - t0) / 1e3:8.1f} gap {(s - prev) / 1e3:5.1f} dur {(e - s) / 1e3:6.1f} wg {wg:6d} {name}")
    prev = e
