#!/bin/bash
# Development helper: K2 time vs (LDS budget per workgroup, number of workgroups): smaller
# partitions let several workgroups share a CU.
for cfg in "100 256" "72 512" "56 512" "48 768" "36 1024"; do
  set -- $cfg
  echo "== PSBA_SCHUR_LDS_KB=$1 PSBA_SCHUR_NWG=$2"
  PSBA_SCHUR_LDS_KB=$1 PSBA_SCHUR_NWG=$2 timeout -k 10 120 python scripts/k2_modes.py 0 4 2>&1 | tail -2 || exit 1
done
