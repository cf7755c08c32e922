#!/bin/bash
# K2 with and without pair items (PSBA_SCHUR_PAIRS), default bench workload, per-kernel HIP-event times.
# Extra arguments: libraries to compare (PSBA_LIB), e.g. build variants of kernels_schur.hip.
export PSBA_BENCH_NO_CFG5=1 PSBA_BENCH_NO_CLUSTERED=1
run() {
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/iter %.4f' % b['ms_per_step'], 'schur %.2f reduce %.2f pair %.2f' % (b['kernels_us']['schur'], b['kernels_us']['schur_reduce'], b['roofline']['avg_launch_us']), 'cost', b['final_cost'])"
}
for rep in 1 2; do
  PSBA_SCHUR_PAIRS=0 run "pairs=0"
  if [ $# -eq 0 ]; then PSBA_SCHUR_PAIRS=1 run "pairs=1"; fi
  for lib in "$@"; do PSBA_LIB=$lib PSBA_SCHUR_PAIRS=1 run "pairs=1 $lib"; done
done
