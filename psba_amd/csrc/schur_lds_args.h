// schur_lds_args.h -- arguments of K2's LDS-partition kernel (kernels_schur.hip; the instrumented fork of the
// kernel, kernels_schur_modes.hip, an experiments-build file, takes the same)
#pragma once
#include "psba_internal.h"

namespace psba {

constexpr int SCHUR_THREADS = 1024;
constexpr int BLK_STRIDE = 37;

struct SchurLdsArgs {
  const double *W, *PV;
  const SchurWg *wg;
  const unsigned long long *items;
  double *slab;
  int *status;
  double *dbg_Y, *dbg_Vinv;
  double mu;
  int nWg, try_id;
  // single rank: while flushing, a workgroup also adds its copies of the blocks (j, k), j <= 5
  // (the first 32x32 diagonal block of S) into diag0 with global atomics, so that the S-reduce
  // kernel can factor that block without gathering it from all the slabs
  double *diag0;           // nullptr: off
  int diag_grp[21], diag_pos[21];
};

}  // namespace psba
