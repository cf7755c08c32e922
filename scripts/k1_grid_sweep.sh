#!/bin/bash
# Development helper: K1 (linearize + camera reduce) time vs number of persistent workgroups.
for g in 256 512 768 1024 1536 2048; do
  echo "== PSBA_LIN_GRID=$g"
  PSBA_LIN_GRID=$g timeout -k 10 120 python scripts/k13_modes.py 2>&1 | grep "LIN_MODE=0" || exit 1
done
