#!/usr/bin/env python3
"""side-by-side table of tools/flow_sweep.sh logs: python tools/sweep_table.py gpurun_out/r3/sw1"""
import sys
from pathlib import Path

d = Path(sys.argv[1])
for n in (32, 16):
    cols = {}
    for v in ("plane", "big", "small", "auto"):
        f = d / f"bl{n}_{v}.log"
        if not f.exists():
            continue
        for line in f.read_text().splitlines():
            p = line.split("|")
            if len(p) < 5 or p[0].startswith("layer"):
                continue
            name = p[0].split()[0]
            fwd, dg = float(p[1].split()[0]), float(p[2].split()[0])
            tags = p[4].split()
            cols.setdefault(name, {})[v] = (fwd * 1e3, dg * 1e3, tags[0], tags[1])
    print(f"N={n}  (us)   fwd: plane big small auto | dgrad: plane big small auto")
    tot = {}
    for name, c in cols.items():
        def g(v, i):
            return c[v][i] if v in c else float("nan")
        print(f"{name:10s} f {g('plane',0):6.1f} {g('big',0):6.1f} {g('small',0):6.1f} {g('auto',0):6.1f} | d {g('plane',1):6.1f} {g('big',1):6.1f} {g('small',1):6.1f} {g('auto',1):6.1f}"
              f" | big {c.get('big',('','','',''))[2]} {c.get('big',('','','',''))[3]} small {c.get('small',('','','',''))[2]} {c.get('small',('','','',''))[3]}")
        for v in c:
            for i, k in ((0, "f"), (1, "d")):
                tot[(v, k)] = tot.get((v, k), 0) + c[v][i]
        tot[("best", "f")] = tot.get(("best", "f"), 0) + min(c[v][0] for v in c)
        tot[("best", "d")] = tot.get(("best", "d"), 0) + min(c[v][1] for v in c)
    print("totals:", {f"{v}.{k}": round(x) for (v, k), x in sorted(tot.items())})
