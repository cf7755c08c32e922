#!/bin/bash
# GPU box: the default bench under different environment settings, two rounds (A/B on one box).
# usage: env_bench.sh "VAR=val" "VAR=val2" ...   ("-" = no setting)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e python bench.py --no-cpu-baseline --steps 300 --warmup 20 > gpurun_out/envb.json 2> gpurun_out/envb.err
  python - <<PY
import json
b=json.loads(open("gpurun_out/envb.json").read().strip().splitlines()[-1])
print("$v", round(b["ms_per_step"],5), b["kernels_us"], round(b["roofline"]["avg_launch_us"],2))
PY
done
done
