#!/bin/bash
# K2 (venice-shaped, default bench workload) under an environment switch: `k2_env_sweep.sh VAR v1 v2 ...` prints the
# HIP-event times of k_schur_lds, k_schur_reduce and of the pair for each value (two runs each).
export PSBA_BENCH_NO_CFG5=1 PSBA_BENCH_NO_CLUSTERED=1
VAR=$1; shift
for v in "$@"; do
  for rep in 1 2; do
  env $VAR=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', 'schur %.2f reduce %.2f pair %.2f' % (b['kernels_us']['schur'], b['kernels_us']['schur_reduce'], b['roofline']['avg_launch_us']), 'ms/iter %.4f' % b['ms_per_step'])"
  done
done
