#!/bin/bash
# same-box A/B of the default bench under two settings of an environment switch, alternating:
#   tools/ab.sh CY_DGRAD_BN 0 1 [repetitions] [extra bench.py args]
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4 2>/dev/null
mkdir -p gpurun_out/ab
for r in $(seq 1 $REPS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-cpu-baseline "$@" > gpurun_out/ab/b_$v.json 2> gpurun_out/ab/b_$v.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab/b_$v.json").read().strip().splitlines()[-1])
print("$VAR=$v", d["ms_per_step"], "ms/step  conv frac", d["roofline"]["frac"], " conv launch ms", d["roofline"]["avg_launch_ms"])
PY
  done
done
