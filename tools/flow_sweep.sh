#!/bin/bash
# per-layer A/B of the conv kernels on one box: plane kernel, flow big tile, flow 16-row tile (run from the repo root)
#   tools/flow_sweep.sh <out-subdir> [extra bench_layers.py arguments, e.g. --only Conv2a,Conv3a]
set -e -o pipefail
OUT=gpurun_out/${1:-sweep}
shift || true
mkdir -p $OUT
for n in 32 16; do
  CY_FLOW=0 python tools/bench_layers.py --n $n "$@" > $OUT/bl${n}_plane.log 2>&1
  CY_FLOW_CFG=1 python tools/bench_layers.py --n $n "$@" > $OUT/bl${n}_big.log 2>&1
  CY_FLOW_CFG=2 python tools/bench_layers.py --n $n "$@" > $OUT/bl${n}_small.log 2>&1
  python tools/bench_layers.py --n $n "$@" > $OUT/bl${n}_auto.log 2>&1
done
