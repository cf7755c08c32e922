#!/bin/bash
# Development helper: K2 time vs LDS budget per workgroup (number of camera-row groups).
for kb in 159 100 80 56 40; do
  echo "== PSBA_SCHUR_LDS_KB=$kb"
  PSBA_SCHUR_PLAN_INFO=1 PSBA_SCHUR_LDS_KB=$kb timeout -k 10 120 python scripts/k2_modes.py 0 4 2>&1 | tail -3 || exit 1
done
