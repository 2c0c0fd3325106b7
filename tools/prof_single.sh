#!/bin/bash
# per-kernel durations of the eager single-stream C2 step (one rocprofv3 process): tools/prof_single.sh <tag>
set -e -o pipefail
TAG=${1:-tmp}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CY_GRAPH_STEP=0 CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -- python3 $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/single.log 2>&1
cd $ROOT
python tools/prof_summary.py $OUT/single 40 > $OUT/kernel_stats.txt
