#!/bin/bash
# same-box: round 3's code (git worktree _r03 of 7558423: `git worktree add _r03 7558423 && (cd _r03 && python contrast-you_amd/build.py)`) against the working tree, alternating
mkdir -p gpurun_out/ab_r03
for r in 1 2 3; do
  for d in _r03 .; do
    ( cd $d && python bench.py --no-cpu-baseline > $OLDPWD/gpurun_out/ab_r03/b.json 2> $OLDPWD/gpurun_out/ab_r03/b.err )
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_r03/b.json").read().strip().splitlines()[-1])
print("$d", d["ms_per_step"], "ms/step", d["value"], "slices/s  conv frac", d["roofline"]["frac"])
PY
  done
done
