#!/bin/bash
# Development helper: K2 time vs the dealing window of the static schedule, with and without
# letting a bank pair be used twice per row (PSBA_SCHUR_DUPS).
for d in 0 1; do
for w in 1 2 3 4 8; do
  echo "== PSBA_SCHUR_DUPS=$d PSBA_SCHUR_WINDOW=$w"
  PSBA_SCHUR_PLAN_INFO=1 PSBA_SCHUR_DUPS=$d PSBA_SCHUR_WINDOW=$w timeout -k 10 120 python scripts/k2_modes.py 0 1 2>&1 | grep -E "plan|mode" | tail -3 || exit 1
done
done
