#!/usr/bin/env python3
"""development aid (library built with -DP8_LATENCY_PROBE): load-to-use latency of one weight item in the
eight-wave plane kernel: first touch, same address again, another tap, again"""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.argv = [sys.argv[0], "Up4", "32"]
src = (REPO / "tools" / "p8_stamps.py").read_text().split("buf = stamps.cpu().tolist()")[0]
exec(compile(src, "p8_stamps", "exec"))
buf = stamps.cpu().tolist()  # noqa: F821
for wv in (0, 1, 4, 5):
    c = buf[wv * 128:wv * 128 + 8]
    print(wv, [c[k + 1] - c[k] for k in range(4)])
