#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks read from stdin:
one line per kernel with VGPR/AGPR/SGPR counts, spills, scratch and occupancy."""
import re
import subprocess
import sys

cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = {"name": body.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = r["name"]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)[:110]
    print(f"{name:110s} v={r.get('VGPRs','?'):>4} a={r.get('AGPRs','?'):>4} s={r.get('TotalSGPRs','?'):>4} "
          f"vsp={r.get('VGPRs Spill','?'):>3} ssp={r.get('SGPRs Spill','?'):>3} scr={r.get('ScratchSize [bytes/lane]','?'):>4} "
          f"occ={r.get('Occupancy [waves/SIMD]','?')}")
