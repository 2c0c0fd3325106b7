#!/usr/bin/env python3
"""HBM bytes and time of every kernel family of the step, side by side: which families sit at the memory roof
and which do not.  Bytes from the two --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB units, as
tools/traffic_summary.py), time from the single-stream kernel trace of the same command.

    python tools/family_roofline.py <profile root> <workload> [steps of the single pass] [steps of the pmc passes]
"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

FAMILIES = OrderedDict([
    ("conv fwd/dgrad (flow)", ("conv3x3_flow_kernel",)),
    ("conv fwd/dgrad (stream)", ("conv3x3_stream_kernel",)),
    ("conv fwd/dgrad (plane, split-K, stem)", ("conv3x3_plane_kernel", "conv_splitk", "conv3x3_first_mfma")),
    ("wgrad", ("wgrad12s_kernel", "wgrad12_kernel", "first_wgrad")),
    ("wgrad reduce", ("wgrad_reduce",)),
    ("BN+ReLU apply (+pool)", ("bn_relu_apply_kernel", "bn_relu_apply_pool_kernel")),
    ("BN bwd sums", ("bn_relu_bwd_reduce",)),
    ("BN bwd apply", ("bn_relu_bwd_apply",)),
    ("maxpool bwd + sums", ("maxpool2_bwd",)),
    ("upsample bwd + sums", ("upsample2_bwd",)),
    ("head 1x1 (fwd, bwd)", ("head_fwd", "head_bwd")),
    ("losses, projector, optimizer, glue", ("",)),
])


def family_of(name):
    for fam, keys in FAMILIES.items():
        if any(k in name for k in keys):
            return fam


def pmc(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[family_of(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


def main(root, wl, steps_single=20, steps_pmc=8):
    f = glob.glob(f"{root}/{wl}_single/**/*kernel_trace.csv", recursive=True)[0]
    t, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        fam = family_of(r["Kernel_Name"])
        t[fam] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        n[fam] += 1
    fe, wr = pmc(f"{root}/{wl}_fetch", "FETCH_SIZE"), pmc(f"{root}/{wl}_write", "WRITE_SIZE")
    print(f"{'family':40s} {'launches':>8s} {'us/step':>9s} {'MB/step':>9s} {'GB/s':>7s} {'of 8 TB/s':>9s}")
    tt = bb = 0.0
    for fam in FAMILIES:
        us = t[fam] / steps_single
        mb = (2.0 * fe[fam] + wr[fam]) * 1024 / steps_pmc / 1e6
        tt, bb = tt + us, bb + mb
        print(f"{fam:40s} {n[fam] / steps_single:8.1f} {us:9.1f} {mb:9.1f} {mb / us * 1e3 if us else 0:7.0f} {mb / us / 8e3 * 1e3 if us else 0:9.2f}")
    print(f"{'step':40s} {sum(n.values()) / steps_single:8.1f} {tt:9.1f} {bb:9.1f} {bb / tt * 1e3:7.0f} {bb / tt / 8:9.2f}")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], a[1], *(int(x) for x in a[2:4]))
