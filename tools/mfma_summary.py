#!/usr/bin/env python3
"""MFMA-busy fraction per conv kernel family from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE` pass (counter_collection.csv):

    mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles)
    kernel cycles = GRBM_GUI_ACTIVE / 8   (rocprofv3 reports the sum over the 8 XCDs, MI355X_MICROARCH.md "DVFS")
    SIMDs = 256 CUs x 4

(SQ_VALU_MFMA_BUSY_CYCLES counts cycles: 32 per v_mfma_f32_32x32x16_bf16, same guide, cycle-constants table.)
Also the MFMA FLOPs the hardware executed (MOPS x 512), to set beside the algorithmic FLOPs: the plane kernels
compute 16 halo columns for 14 outputs.

    python tools/mfma_summary.py <pmc_dir> <out.json>
"""
import csv
import glob
import json
import sys
from collections import defaultdict

FAMILIES = {"conv3x3_fwd_dgrad": ("conv3x3_igemm_kernel", "conv3x3_plane_kernel", "conv3x3_stream_kernel", "conv3x3_flow_kernel"),
            "conv3x3_wgrad": ("wgrad_kernel", "wgrad12_kernel", "wgrad12s_kernel")}
SIMDS = 256 * 4


def main(d, out):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    per = defaultdict(lambda: defaultdict(float))  # (family, dispatch) -> counter -> value
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "reduce" in name or "first" in name:
            continue
        for fam, keys in FAMILIES.items():
            if any(k in name for k in keys):
                per[(fam, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {}
    for fam in FAMILIES:
        rows = [v for (f_, _), v in per.items() if f_ == fam]
        if not rows:
            continue
        busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in rows)
        cyc = sum(v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 for v in rows)
        mops = sum(v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) for v in rows)
        res[fam] = {"launches_sampled": len(rows), "mfma_busy": round(busy / (cyc * SIMDS), 4) if cyc else None,
                    "kernel_cycles_per_launch": round(cyc / len(rows)),
                    "mfma_flops_executed_per_launch": round(mops * 512 / len(rows)),
                    "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
