#!/usr/bin/env python3
"""HBM traffic per launch of a kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
corrected as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact.

    python tools/traffic_summary.py <fetch_dir> <write_dir> <out.json>
"""
import csv
import glob
import json
import sys
from collections import defaultdict

# bench.py's name of a kernel family -> substrings of the kernel names that belong to it
FAMILIES = {"conv3x3_fwd_dgrad": ("conv3x3_igemm_kernel", "conv3x3_plane_kernel", "conv3x3_stream_kernel", "conv3x3_flow_kernel"),
            "conv3x3_wgrad": ("wgrad_kernel", "wgrad12_kernel", "wgrad12s_kernel")}


def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for fam, keys in FAMILIES.items():
            if any(k in r["Kernel_Name"] for k in keys) and "reduce" not in r["Kernel_Name"] \
                    and "first" not in r["Kernel_Name"]:
                acc[fam][0] += float(r["Counter_Value"])
                acc[fam][1] += 1
    return acc


def main(fd, wd, out):
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {}
    for fam in FAMILIES:
        if fe[fam][1] == 0:
            continue
        fetch = 2.0 * fe[fam][0] * 1024 / fe[fam][1]
        write = wr[fam][0] * 1024 / max(wr[fam][1], 1)
        res[fam] = {"launches_sampled": fe[fam][1], "fetch_bytes_per_launch": round(fetch),
                    "write_bytes_per_launch": round(write), "hbm_bytes_per_launch": round(fetch + write),
                    "note": "FETCH_SIZE x2 (gfx950 half-count of 16B/lane reads), WRITE_SIZE exact, KiB units"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
