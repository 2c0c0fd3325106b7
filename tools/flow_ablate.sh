#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/${1:-abl}
mkdir -p $OUT
L=Up4,Up3,Conv4b,Up_conv4b
for n in 32 16; do
for d in 0 1 2 3 4 8 12 15; do
  echo "== N=$n debug=$d" >> $OUT/abl.log
  CY_FLOW_DEBUG=$d python tools/bench_layers.py --n $n --only $L --iters 20 2>&1 | grep -v amdgpu >> $OUT/abl.log
done
done
