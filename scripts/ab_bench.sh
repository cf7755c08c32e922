#!/bin/bash
# GPU box: the default bench with each library variant under ab/ in turn (A/B timing on one box).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
cp psba_amd/libpsba_hip.so /tmp/lib_keep.so
for round in 1 2; do
for v in "$@"; do
  cp ab/lib$v.so psba_amd/libpsba_hip.so
  python bench.py --no-cpu-baseline --steps 300 --warmup 20 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
b=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("$v", round(b["ms_per_step"],5), b["kernels_us"], round(b["roofline"]["avg_launch_us"],2))
PY
done
done
cp /tmp/lib_keep.so psba_amd/libpsba_hip.so
