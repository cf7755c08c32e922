#!/bin/bash
# A/B on one box: K2's plan defaults against PSBA_SCHUR_WINDOW / PSBA_SCHUR_LDS_KB candidates (pair time, HIP events).
export PSBA_BENCH_NO_CFG5=1 PSBA_BENCH_NO_CLUSTERED=1
run() {
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'schur %.2f reduce %.2f pair %.2f' % (b['kernels_us']['schur'], b['kernels_us']['schur_reduce'], b['roofline']['avg_launch_us']), 'ms/iter %.4f' % b['ms_per_step'])"
}
for rep in 1 2; do
  run "default      "
  PSBA_SCHUR_WINDOW=2 run "window 2     "
  PSBA_SCHUR_WINDOW=2 PSBA_SCHUR_LDS_KB=140 run "window 2, 140"
  PSBA_SCHUR_LDS_KB=140 run "140          "
done
