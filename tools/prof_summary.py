#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals, short names."""
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<.*>)?", name)
    base = m.group(1) if m else name
    targs = ""
    if m and m.group(2):
        targs = m.group(2)
        targs = targs.replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16")
        if len(targs) > 60:
            targs = targs[:57] + "...>"
    return (base + targs)[:100]


def main(d, top=45):
    f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)
    if not f:
        sys.exit(f"no kernel_stats.csv under {d}")
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print(f"# {f[0]}\n# total kernel time {tot / 1e6:.3f} ms over {calls} launches")
    print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print(f"{short(r['Name']):100s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:9.3f} "
              f"{float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
