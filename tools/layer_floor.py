#!/usr/bin/env python3
"""Per-layer mixed roofline of a per-layer table (tools/bench_layers.py output): for every 3x3 layer the floor
max(FLOPs / MFMA peak, algorithmic bytes / HBM rate) of its forward, data gradient and weight gradient, next to the
measured time -- at the nominal peaks (2.5 PFLOP/s bf16 dense, 8 TB/s) and at what the part holds under load
(MI355X_MICROARCH.md: 6.3 TB/s achievable; the MFMA pipes at the ~1.9 GHz a long matrix workload settles at = 0.79 x).

    python tools/layer_floor.py profiles/r04_layers_n32.txt 32
"""
import re
import sys

PEAK_F, PEAK_B = 2500e12, 8e12
HELD_F, HELD_B = 2500e12 * 1.9 / 2.4, 6.3e12


def main(path, n):
    rows = []
    for line in open(path):
        m = re.match(r"(\S+)\s+(\d+)\s+(\d+)\s+(\d+) \|\s+([\d.]+)\s+[\d.]+ \|\s+([\d.]+)\s+[\d.]+ \|\s+([\d.]+)", line)
        if m:
            rows.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(6)), float(m.group(7))))
    print(f"{'layer':10s} {'HxW':>4s} {'Cin':>4s} {'Cout':>4s} | {'fwd us':>7s} {'floor':>6s} {'held':>6s} {'x':>5s} | {'dgrad':>7s} {'floor':>6s} {'held':>6s} {'x':>5s} | "
          f"{'wgrad':>7s} {'floor':>6s} {'held':>6s} {'x':>5s}")
    tot = [0.0] * 9
    for name, hw, cin, cout, tf, td, tw in rows:
        px = n * hw * hw
        flops = 2.0 * px * 9 * cin * cout
        b_f = px * (cin + cout) * 2.0                       # forward / data gradient: the input and the output once (16-bit)
        b_w = px * (cin + cout) * 2.0 + 9.0 * cin * cout * 4  # weight gradient: input and dy once, dW once (f32)
        out = []
        for k, (t_ms, byts) in enumerate(((tf, b_f), (td, b_f), (tw, b_w))):
            floor = max(flops / PEAK_F, byts / PEAK_B) * 1e6
            held = max(flops / HELD_F, byts / HELD_B) * 1e6
            t = t_ms * 1e3
            out.append(f"{t:7.1f} {floor:6.1f} {held:6.1f} {t / held:5.2f}")
            tot[3 * k] += t
            tot[3 * k + 1] += floor
            tot[3 * k + 2] += held
        print(f"{name:10s} {hw:4d} {cin:4d} {cout:4d} | " + " | ".join(out))
    print(f"{'total':25s} | " + " | ".join(f"{tot[3 * k]:7.1f} {tot[3 * k + 1]:6.1f} {tot[3 * k + 2]:6.1f} {tot[3 * k] / tot[3 * k + 2]:5.2f}" for k in range(3)))
    print("(us; floor = max(FLOPs / 2.5 PFLOP/s, bytes / 8 TB/s); held = the same at 1.98 PFLOP/s and 6.3 TB/s; x = measured / held)")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
