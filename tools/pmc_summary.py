#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv)."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"void ", "", name)[:80]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    n = len(next(iter(cs.values())))
    print(f"{k}  (dispatches {n})")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}")
