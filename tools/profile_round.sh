#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag> [workload]     e.g.  tools/profile_round.sh r03        (C2, every summary)
#                                                     tools/profile_round.sh r03 c4     (PMC summaries of C4 only)
# Kernel names are kept MANGLED (-M): rocprofv3's own demangling truncates template arguments
# ("conv3x3_stream_kernel<bool _Accum, int, E, 1, true, true>"), which tests/c2_layers.py cannot match to a plan.
# Every rocprofv3 run is its own process; the counter passes (--pmc) carry no other trace domain.
set -e -o pipefail
TAG=${1:-r01}
WL=${2:-c2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
# one directory per (workload, pass), emptied first: the summaries below read whatever CSVs they find there
for d in single overlap fetch write mfma; do rm -rf $OUT/${WL}_$d; done
BENCH_ARGS="--workload $WL --steps 16 --warmup 4 --no-cpu-baseline --no-roofline"
PMC_ARGS="--workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-roofline"
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel durations, eager single-stream step (what bench.py's roofline leg must agree with)
( export CY_GRAPH_STEP=0 CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0
  rocprofv3 -M --kernel-trace --stats --output-format csv -d $OUT/${WL}_single -- python3 $ROOT/bench.py $BENCH_ARGS > $OUT/${WL}_single.log 2>&1 )
echo "[profile] single-stream kernel stats done"
# 2. eager three-stream step: stream overlap
( export CY_GRAPH_STEP=0
  rocprofv3 -M --kernel-trace --stats --output-format csv -d $OUT/${WL}_overlap -- python3 $ROOT/bench.py $BENCH_ARGS > $OUT/${WL}_overlap.log 2>&1 )
echo "[profile] overlapped kernel trace done"
# 3./4. HBM traffic counters, one pass each
( export CY_GRAPH_STEP=0 CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0
  rocprofv3 -M --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${WL}_fetch -- python3 $ROOT/bench.py $PMC_ARGS > $OUT/${WL}_fetch.log 2>&1 )
echo "[profile] FETCH_SIZE pass done"
( export CY_GRAPH_STEP=0 CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0
  rocprofv3 -M --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${WL}_write -- python3 $ROOT/bench.py $PMC_ARGS > $OUT/${WL}_write.log 2>&1 )
echo "[profile] WRITE_SIZE pass done"
# 5. matrix-core busy cycles (north_star: "rocprof ... MFMA-busy reported against MI355X peak"), its own pass
( export CY_GRAPH_STEP=0 CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0
  rocprofv3 -M --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/${WL}_mfma -- python3 $ROOT/bench.py $PMC_ARGS > $OUT/${WL}_mfma.log 2>&1 )
echo "[profile] MFMA-busy pass done"
cd $ROOT
python tools/prof_summary.py $OUT/${WL}_single 60 > $OUT/${TAG}_bench_${WL}_kernel_stats_single_stream.txt
python tools/prof_summary.py $OUT/${WL}_overlap 60 > $OUT/${TAG}_bench_${WL}_kernel_stats_overlapped.txt
python tools/trace_overlap.py $OUT/${WL}_overlap > $OUT/${TAG}_bench_${WL}_stream_overlap.txt
python tools/pmc_summary.py $OUT/${WL}_fetch > $OUT/${TAG}_${WL}_pmc_fetch_size.txt
python tools/pmc_summary.py $OUT/${WL}_write > $OUT/${TAG}_${WL}_pmc_write_size.txt
python tools/traffic_summary.py $OUT/${WL}_fetch $OUT/${WL}_write $OUT/${TAG}_${WL}_traffic.json
python tools/pmc_summary.py $OUT/${WL}_mfma > $OUT/${TAG}_${WL}_pmc_mfma.txt
python tools/mfma_summary.py $OUT/${WL}_mfma $OUT/${TAG}_${WL}_mfma.json
if [ "$WL" = c2 ]; then python tools/step_timeline.py > $OUT/${TAG}_step_timeline.txt 2>&1; fi
echo "[profile] summaries written to $OUT"
