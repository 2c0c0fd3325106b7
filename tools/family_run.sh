#!/bin/bash
# per-family kernel time of the default step (eager, single stream) + the graph-replayed bench line:
#   tools/family_run.sh <tag>      -> gpurun_out/fam_<tag>/{family.txt,bench.json}
set -e -o pipefail
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/fam_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( export CY_GRAPH_STEP=0 CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0
  rocprofv3 -M --kernel-trace --stats --output-format csv -d $OUT/single -- python3 $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline > $OUT/single.log 2>&1 )
cd $ROOT
python tools/family_times.py $OUT/single 20 > $OUT/family.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/family.txt
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], " conv frac", d["roofline"]["frac"])
PY
