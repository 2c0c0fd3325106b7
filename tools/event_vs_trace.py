#!/usr/bin/env python3
"""Matches bench.py's per-launch event durations (stderr lines of CY_BENCH_DUMP_EVENTS=1: the first counted step of
the instrumented pass) with the rocprofv3 kernel trace of the same process (tools/event_vs_trace.sh):
    python tools/event_vs_trace.py gpurun_out/evt
Prints, per conv3x3_fwd_dgrad launch, the event interval, the traced kernels inside it and their summed duration."""
import csv
import glob
import re
import sys


def main(d):
    ev = []
    for line in open(f"{d}/bench.err"):
        m = re.match(r"\[bench\] (\S+)\s+([\d.]+) GFLOP\s+([\d.]+) us", line)
        if m:
            ev.append((m.group(1), float(m.group(2)), float(m.group(3))))
    f = glob.glob(f"{d}/trace/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    is_main = lambda n: ("conv3x3_" in n and "first" not in n) or "wgrad" in n or "splitk" in n  # noqa: E731
    # the instrumented pass is the LAST epoch: 6 steps of identical launch sequences; the dump is its 4th step
    seq = [r for r in rows if is_main(r["Kernel_Name"])]
    n_conv_ev = sum(1 for e in ev if e[0] == "conv3x3_fwd_dgrad")
    conv_rows = [r for r in seq if "wgrad" not in r["Kernel_Name"] and "splitk" not in r["Kernel_Name"]]
    per_step = n_conv_ev
    step_rows = conv_rows[-3 * per_step:-2 * per_step]  # 4th of 6 steps
    by_start = {int(r["Start_Timestamp"]): r for r in rows}
    starts = sorted(by_start)
    tot_e = tot_k = 0.0
    i = 0
    for kind, gf, us in ev:
        if kind != "conv3x3_fwd_dgrad":
            continue
        r = step_rows[i]
        i += 1
        k_us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        # gap to the previous traced kernel of any kind
        j = starts.index(int(r["Start_Timestamp"]))
        prev = by_start[starts[j - 1]]
        gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3
        name = re.search(r"conv3x3_(\w+?)_kernelI\w+?(Li\w+?)Ev", r["Kernel_Name"])
        tot_e += us
        tot_k += k_us
        print(f"{i:3d} {gf:7.2f} GF  event {us:6.1f}  kernel {k_us:6.1f}  diff {us - k_us:5.1f}  idle before {gap:6.1f}  "
              f"{name.group(1) if name else r['Kernel_Name'][:40]} {name.group(2) if name else ''}")
    print(f"sum events {tot_e:.1f} us, sum traced main kernels {tot_k:.1f} us (split-K finish kernels not in the latter)")


if __name__ == "__main__":
    main(sys.argv[1])
