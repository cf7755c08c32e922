#!/bin/bash
# Development helper: K2 time vs the dealing window of the static schedule.
for w in 1 4 8 16 32 64; do
  echo "== PSBA_SCHUR_WINDOW=$w"
  PSBA_SCHUR_PLAN_INFO=1 PSBA_SCHUR_WINDOW=$w timeout -k 10 120 python scripts/k2_modes.py 0 1 3 2 2>&1 | tail -5 || exit 1
done
