// kernels_chol.hip -- dense reduced-camera solve  S dpa = ea  by one blocked Cholesky
// factorisation and two triangular solves; S^-1 is never formed.
//
// Replaces SPDinv = cholesky -> trigMat_inv -> trigMat_mul (+ kern_fill_rest) and matVec_mul
// (reference PSBA/cl_spdinv.cpp:18-204, CL_files/SPD_inv.cl:20-411, PSBA/cl_linearalg.cpp:19,
// CL_files/matVec_mul.cl:7-18): the reference chains ~nA device-enqueued launches of 3x3
// blocks; here one workgroup runs a right-looking blocked factorisation with the forward
// solve folded in (ea is carried as an extra row of the matrix) and a blocked backward
// solve.  A non-positive or non-finite pivot sets status[1] (the reference's ret = 1.0,
// SPD_inv.cl:35-38,66).
//
// v1: one workgroup, vector fp64 (no MFMA yet), n <= CHOL_MAX_N.
#include "psba_internal.h"

namespace psba {

constexpr int NB = 16;
constexpr int CHOL_THREADS = 1024;
constexpr int CHOL_MAX_N = 1024;

__global__ __launch_bounds__(CHOL_THREADS) void k_chol_solve(double *S, double *ea, double *x,
                                                             int n, int *status) {
  __shared__ double sD[NB][NB + 1];
  __shared__ double sX[NB];
  __shared__ int sFail;
  extern __shared__ double sP[];  // [(n+1)][NB+1] panel below the diagonal block (+ ea row)
  const int tid = threadIdx.x;
  if (tid == 0) sFail = 0;
  __syncthreads();

  for (int jb = 0; jb < n; jb += NB) {
    const int nb = (n - jb) < NB ? (n - jb) : NB;
    // 1. diagonal block -> LDS
    if (tid < NB * NB) {
      const int r = tid / NB, c = tid % NB;
      if (r < nb && c <= r) sD[r][c] = S[(size_t)(jb + r) * n + jb + c];
    }
    __syncthreads();
    // 2. factor it in one wave: lane = row
    if (tid < 64) {
      const int r = tid;
      for (int j = 0; j < nb; j++) {
        const double d = sD[j][j];
        const double l = sqrt(d);
        if (!(d > 0.0) || !isfinite(l)) {
          if (r == 0) sFail = 1;
        }
        __builtin_amdgcn_wave_barrier();
        if (r == j) sD[j][j] = l;
        if (r > j && r < nb) sD[r][j] = sD[r][j] / l;
        __builtin_amdgcn_wave_barrier();
        if (r > j && r < nb) {
          const double lrj = sD[r][j];
          for (int c = j + 1; c <= r; c++) sD[r][c] -= lrj * sD[c][j];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    if (sFail) break;
    if (tid < NB * NB) {
      const int r = tid / NB, c = tid % NB;
      if (r < nb && c <= r) S[(size_t)(jb + r) * n + jb + c] = sD[r][c];
    }
    // 3. panel solve: rows below the block and the ea row; one thread per row
    const int r0 = jb + nb;          // first trailing row / column
    const int m = n - r0 + 1;        // trailing rows including the ea row
    for (int t = tid; t < m; t += CHOL_THREADS) {
      const int R = r0 + t;
      double *row = (R < n) ? (S + (size_t)R * n + jb) : (ea + jb);
      double a[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) a[c] = (c < nb) ? row[c] : 0.0;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c < nb) {
          double v = a[c];
#pragma unroll
          for (int k = 0; k < NB; k++)
            if (k < c) v -= a[k] * sD[c][k];
          a[c] = v / sD[c][c];
        }
      }
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c < nb) row[c] = a[c];
        sP[t * (NB + 1) + c] = a[c];
      }
    }
    __syncthreads();
    // 4. trailing update  A[R][C] -= P[R] . P[C]  for r0 <= C <= R (lower), R up to the ea row
    const int mc = m - 1;  // trailing columns
    if (mc > 0) {
      const int total = m * mc;
      for (int t = tid; t < total; t += CHOL_THREADS) {
        const int rr = t / mc, cc = t % mc;
        if (cc > rr) continue;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NB; k++) acc += sP[rr * (NB + 1) + k] * sP[cc * (NB + 1) + k];
        double *dst = (r0 + rr < n) ? (S + (size_t)(r0 + rr) * n + r0 + cc) : (ea + r0 + cc);
        *dst -= acc;
      }
    }
    __syncthreads();
  }

  if (sFail) {
    if (tid == 0) atomicOr(&status[1], 1);
    for (int t = tid; t < n; t += CHOL_THREADS) x[t] = 0.0;
    return;
  }

  // backward solve  L^T x = y  (y is in ea), blocked from the last panel up
  const int last = ((n - 1) / NB) * NB;
  for (int jb = last; jb >= 0; jb -= NB) {
    const int nb = (n - jb) < NB ? (n - jb) : NB;
    if (tid < NB * NB) {
      const int r = tid / NB, c = tid % NB;
      if (r < nb && c <= r) sD[r][c] = S[(size_t)(jb + r) * n + jb + c];
    }
    __syncthreads();
    if (tid < 64) {
      const int c = tid;
      double y = (c < nb) ? ea[jb + c] : 0.0;
      for (int k = nb - 1; k >= 0; k--) {
        if (c == k) sX[k] = y / sD[k][k];
        __builtin_amdgcn_wave_barrier();
        if (c < k) y -= sD[k][c] * sX[k];
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    if (tid < nb) x[jb + tid] = sX[tid];
    for (int c = tid; c < jb; c += CHOL_THREADS) {
      double acc = 0.0;
      for (int k = 0; k < nb; k++) acc += S[(size_t)(jb + k) * n + c] * sX[k];
      ea[c] -= acc;
    }
    __syncthreads();
  }
  // non-finite solution counts as failure, as in the reference's isfinite checks
  int bad = 0;
  for (int t = tid; t < n; t += CHOL_THREADS)
    if (!isfinite(x[t])) bad = 1;
  if (bad) atomicOr(&status[1], 1);
}

int launch_chol_solve(psba_ctx *h) {
  const Dims &d = h->d;
  if (d.nA > CHOL_MAX_N)
    return fail(h, PSBA_E_INVALID, "dense solve supports 6*nCams <= %d for now (got %d)",
                CHOL_MAX_N, d.nA);
  const size_t lds = sizeof(double) * (size_t)(d.nA + 1) * (NB + 1);
  {
    ProfScope ps(h, PSBA_K_CHOLESKY);
    hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(CHOL_THREADS), lds, h->stream, h->red,
                       h->red + (size_t)d.nA * d.nA, h->dp, d.nA, h->status);
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
