#!/usr/bin/env python3
"""Per-layer durations of the BatchNorm backward triple (reduce, finalize, apply) of one C2 step, from a rocprofv3
kernel trace directory (tools/profile_round.sh: gpurun_out/profile_rNN/c2_single):  python tools/bn_layers.py <dir>"""
import csv
import glob
import sys


def main(d):
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda x: (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3  # noqa: E731
    tr = []
    for i, r in enumerate(rows):
        if "bn_relu_bwd_reduce" in r["Kernel_Name"]:
            fin = next(x for x in rows[i + 1:i + 6] if "bn_bwd_finalize" in x["Kernel_Name"])
            ap = next(x for x in rows[i + 1:i + 8] if "bn_relu_bwd_apply" in x["Kernel_Name"])
            tr.append((dur(r), dur(fin), dur(ap), (int(ap["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                       int(r["Grid_Size_X"]) // 256))
    n = 32
    steps = len(tr) // n
    step = tr[(steps // 2) * n:(steps // 2 + 1) * n]
    for t in step:
        print("reduce %5.1f  finalize %4.1f  apply %5.1f  first start -> last end %6.1f  reduce workgroups %d" % t)
    print(f"sum of kernels {sum(t[0] + t[1] + t[2] for t in step):.1f} us, of spans {sum(t[3] for t in step):.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
