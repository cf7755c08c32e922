#!/bin/bash
# K2 at full-size cfg5 under PSBA_SCHUR_SPLIT (slabs per block-range group = point stretches): kernels_us of
# bench.py --workload cfg5 --cfg5-points 2000000.
export PSBA_SCHUR_SLAB_MAX_GB=${SLAB_MAX_GB:-16}
for v in "$@"; do
  PSBA_SCHUR_SPLIT=$v python bench.py --workload cfg5 --cfg5-points 2000000 --steps 3 --warmup 0 --segment 3 --spread-segments 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=b['kernels_us']
print('split=$v', 'schur %.2f ms reduce %.2f ms pair %.2f ms' % (k['schur']/1e3, k['schur_reduce']/1e3, b['roofline']['avg_launch_us']/1e3), 'ms/iter %.2f' % b['ms_per_step'], 'cost %.10g' % b['final_cost'])"
done
