#!/bin/bash
# cfg5's dense factorization (12 000 x 12 000) under an environment switch: `chol_cfg5_sweep.sh VAR v1 v2 ...`
# prints kernels_us.cholesky (HIP events, ms) of bench.py --workload cfg5 with 200 k points for each value.
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --workload cfg5 --cfg5-points 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', 'cholesky %.3f ms' % (b['kernels_us']['cholesky']/1e3), 'ms/iter %.3f' % b['ms_per_step'], 'cost %.10g' % b['final_cost'])"
done
