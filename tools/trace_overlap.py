#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: wall time, sum of kernel durations, busy (union) time and the
time at concurrency >= 2, over the last `steps` steps (step boundary = radam_kernel launches)."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
marks = [i for i, r in enumerate(rows) if "radam_kernel" in r[2]]
# two radam launches per step (two parameter groups): take the even ones as step ends
ends = marks[1::2]
if len(ends) < 6:
    print("not enough steps", len(ends)); sys.exit(0)
lo, hi = ends[-6] + 1, ends[-1] + 1
win = rows[lo:hi]
steps = 5
t0, t1 = win[0][0], max(r[1] for r in win)
ev = []
for s, e, *_ in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = conc2 = 0
cur = 0
last = ev[0][0]
for t, dlt in ev:
    if cur >= 1: busy += t - last
    if cur >= 2: conc2 += t - last
    cur += dlt
    last = t
tot = sum(e - s for s, e, *_ in win)
print(f"steps {steps}: wall {(t1 - t0) / steps / 1e6:.3f} ms/step, kernel-sum {tot / steps / 1e6:.3f}, "
      f"busy {busy / steps / 1e6:.3f}, concurrency>=2 {conc2 / steps / 1e6:.3f}, idle {(t1 - t0 - busy) / steps / 1e6:.3f}")
streams = {}
for s, e, n, q in win:
    streams.setdefault(q, [0, 0])
    streams[q][0] += 1
    streams[q][1] += e - s
for q, (c, t) in sorted(streams.items()):
    print(f"  stream/queue {q}: {c / steps:.0f} launches/step, {t / steps / 1e6:.3f} ms/step")
# biggest idle gaps
gaps = []
cur = 0
last = None
for t, dlt in ev:
    if cur == 0 and last is not None and t > last:
        gaps.append((t - last, last))
    cur += dlt
    if cur == 0:
        last = t
gaps.sort(reverse=True)
print("largest idle gaps (us):", [round(g / 1e3, 1) for g, _ in gaps[:12]])
