#!/usr/bin/env python3
"""registers / spills / occupancy of the kernels of one csrc file whose mangled name contains a pattern:
    python tools/kernel_regs.py cy_conv3x3 flow_kernelIDF16b"""
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
stem, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO}/include", f"-I{REPO}/contrast-you_amd/csrc",
       "-c", f"{REPO}/contrast-you_amd/csrc/{stem}.hip", "-o", "/tmp/_regs.o", "-Rpass-analysis=kernel-resource-usage"]
log = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = None
rows = {}
for line in log.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    for key in ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(rf"remark: +{re.escape(key)}: (\d+)", line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for name, r in rows.items():
    if pat in name:
        print(f"{name[:110]:110s} v{r.get('VGPRs', -1):4d} a{r.get('AGPRs', -1):4d} spill {r.get('VGPRs Spill', -1):4d} occ {r.get('Occupancy [waves/SIMD]', -1)} lds {r.get('LDS Size [bytes/block]', -1)}")
