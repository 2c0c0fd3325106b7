#!/bin/bash
# One process, both clocks: bench.py's per-launch HIP-event durations (instrumented pass, CY_BENCH_DUMP_EVENTS=1)
# and rocprofv3's kernel trace of the same launches.  Run on the GPU box from the repo root.
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-evt}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CY_GRAPH_STEP=0 CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0 CY_BENCH_DUMP_EVENTS=1
rocprofv3 -M --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err
cd $ROOT
python tools/prof_summary.py $OUT/trace 40 > $OUT/kernel_stats.txt
ls $OUT/trace/*/
