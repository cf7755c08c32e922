// kernels_chol.hip -- dense reduced-camera solve  S dpa = ea  by one blocked Cholesky
// factorisation and two triangular solves; S^-1 is never formed.
//
// Replaces SPDinv = cholesky -> trigMat_inv -> trigMat_mul (+ kern_fill_rest) and matVec_mul
// (reference PSBA/cl_spdinv.cpp:18-204, CL_files/SPD_inv.cl:20-411, PSBA/cl_linearalg.cpp:19,
// CL_files/matVec_mul.cl:7-18): the reference chains ~nA device-enqueued launches of 3x3
// blocks and then multiplies by an explicit inverse.  A non-positive or non-finite pivot
// sets status[1] (the reference's ret = 1.0, SPD_inv.cl:35-38,66).
//
// Data: the reduce buffer is the padded matrix Lw[(n32+16)][n32] (row-major, n32 = n rounded
// up to 32): rows < n hold S, rows n..n32-1 identity padding, row n32 holds e_a, the rest
// zeros -- so every load is branch-free and e_a rides along as one more row (the forward
// solve L y = e_a comes for free).
//
// One workgroup of 8 waves, left-looking by block columns of 32:
//   A  the 32 rows of the block column's own L are staged in LDS (the B operand); every wave
//      owns row tiles (16 rows) and applies all previous block columns with
//      v_mfma_f64_16x16x4_f64 (C -= L_rows L_cols^T), accumulators in registers, A operands
//      prefetched from the L2-resident matrix four k-steps ahead;
//   B  the 32x32 diagonal block is factored by one wave, rows in registers, columns passed
//      between lanes with v_readlane;
//   C  the rows below are finished by forward substitution against the block, one thread
//      per row.
// The backward solve L^T x = y is blocked the same way (substitution in one wave per block,
// then a rank-32 update of the remaining right-hand side by all threads).
#include <cstdlib>

#include "psba_internal.h"

namespace psba {

constexpr int CB = 16;  // block column width (one MFMA column tile)
constexpr int CHOL_THREADS = 512;
constexpr int CHOL_WAVES = CHOL_THREADS / 64;

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double dcb __attribute__((ext_vector_type(CB)));  // SSA vector: never demoted to scratch

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

// Phase A for one wave: NT row tiles (tile index T0, T0+8, ...) of block column j.
// acc = C - L[rows, 0:j] L[j:j+16, 0:j]^T with two accumulators per tile (even / odd k-steps:
// a dependent v_mfma_f64_16x16x4 chain costs ~138 cycles per link, independent ones issue
// every 32), A operands fetched two 16-column groups ahead, B operands from LDS.  No branch
// sits between the MFMAs: the number of tiles is a template parameter chosen by a
// wave-uniform switch.  The finished tiles go back in place, the diagonal tile into sD.
template <int NT>
__device__ __forceinline__ void chol_update_tiles(double *Lw, const double *sB, double (*sD)[CB + 1],
                                                  int ld, int ldb, int j, int T0, int firstTile,
                                                  int li, int lk, long long *st) {
  long long s0 = st ? (long long)__builtin_amdgcn_s_memtime() : 0;
  d4 acc0[NT], acc1[NT], acc2[NT], acc3[NT];
  // k-slot assignment: in the group of 16 columns starting at k0, MFMA step t (0..3) pairs
  // lane slot lk with column k0 + 4 lk + t.  Any pairing is valid as long as A and B agree,
  // and this one lets every lane fetch its four A values as ONE 32-byte piece of its row, so
  // a wave-load covers 16 rows x 128 B exactly (8-byte pieces at a 32-byte stride moved four
  // times the bytes through L1 and made this loop bandwidth-bound).
  const double4 *arow[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    const int T = T0 + q * CHOL_WAVES;
    arow[q] = reinterpret_cast<const double4 *>(Lw + (size_t)(16 * T + li) * ld + 4 * lk);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      acc0[q][r] = Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + li];
      acc1[q][r] = 0.0;
      acc2[q][r] = 0.0;
      acc3[q][r] = 0.0;
    }
  }
  const int nGrp = j / 16;
  const double4 zero4 = {0.0, 0.0, 0.0, 0.0};
  double4 a1[NT], a2[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    a1[q] = (nGrp > 0) ? arow[q][0] : zero4;
    a2[q] = (nGrp > 1) ? arow[q][4] : zero4;  // + 16 columns
  }
  if (st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long s1 = (long long)__builtin_amdgcn_s_memtime();
    st[0] += s1 - s0;
    s0 = s1;
  }
  for (int gI = 0; gI < nGrp; gI++) {
    const int k0 = 16 * gI;
    double4 ac[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) {
      ac[q] = a1[q];
      a1[q] = a2[q];
    }
    if (gI + 2 < nGrp) {
#pragma unroll
      for (int q = 0; q < NT; q++) a2[q] = arow[q][4 * (gI + 2)];
    }
    const double *bp = sB + li * ldb + k0 + 4 * lk;
    const double b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
    // four independent accumulators per tile (one per k-slot step): a dependent MFMA chain
    // costs ~138 cycles per link, so each accumulator sees one link per group
#pragma unroll
    for (int q = 0; q < NT; q++) {
      acc0[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ac[q].x, b0, acc0[q], 0, 0, 0);
      acc1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ac[q].y, b1, acc1[q], 0, 0, 0);
      acc2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ac[q].z, b2, acc2[q], 0, 0, 0);
      acc3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ac[q].w, b3, acc3[q], 0, 0, 0);
    }
  }
  if (st) {
    const long long s1 = (long long)__builtin_amdgcn_s_memtime();
    st[1] += s1 - s0;
    s0 = s1;
  }
#pragma unroll
  for (int q = 0; q < NT; q++) {
    const int T = T0 + q * CHOL_WAVES;
    if (T == firstTile) {
#pragma unroll
      for (int r = 0; r < 4; r++) sD[lk + 4 * r][li] = (acc0[q][r] + acc1[q][r]) + (acc2[q][r] + acc3[q][r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++)
        Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + li] = (acc0[q][r] + acc1[q][r]) + (acc2[q][r] + acc3[q][r]);
    }
  }
  if (st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[2] += (long long)__builtin_amdgcn_s_memtime() - s0;
  }
}

// MAXT = row tiles a wave may own in one block column: ceil((n32/16 + 1) / CHOL_WAVES)
template <int MAXT>
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_solve(double *Lw, double *x,
                                                             double *linv_store, int n, int n32,
                                                             int *status, int try_id, long long *tim) {
  __shared__ double sD[CB][CB + 1];              // diagonal block, then its Cholesky factor
  __shared__ double sInvD[CB];                   // 1 / diag of the factor
  __shared__ double sX[CB];
  extern __shared__ double sDyn[];  // sB [CB][ldb], then the diagonal blocks of L for the backward solve
  __shared__ double sCol[CB];
  __shared__ int sFail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int ld = n32;
  const int ldb = n32 + 1;
  double *sB = sDyn;                       // [CB][ldb] rows j..j+15 of L, columns < j
  double *sDiag = sDyn + (size_t)CB * ldb;  // [n32/CB][CB][CB+1]
  const int nTiles = n32 / 16 + 1;  // row tiles incl. the e_a tile
  if (tid == 0) sFail = 0;
  // Pull the whole matrix into this XCD's L2 first, with as many lines in flight as the
  // workgroup can keep: the producer kernel ran on all eight XCDs, so every first touch
  // would otherwise be a dependent ~2 us miss in the middle of the factorisation.
  {
    const size_t nLines = ((size_t)(n32 + 1) * n32 * sizeof(double) + 127) / 128;
    const float *base = reinterpret_cast<const float *>(Lw);
    float warm = 0.f;
    for (size_t l0 = 0; l0 < nLines; l0 += 8 * CHOL_THREADS) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const size_t l = l0 + (size_t)u * CHOL_THREADS + tid;
        v[u] = (l < nLines) ? base[l * 32] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) warm += v[u];
    }
    if (warm == 1.2345e-30f) sFail = 2;  // keeps the loads alive; never true in practice
  }
  __syncthreads();
  long long tA = 0, tB = 0, tC = 0, tBack = 0, t0 = 0, tF = 0, tS = 0;
  long long st3[3] = {0, 0, 0};
#define STAMP() (tim ? (long long)__builtin_amdgcn_s_memtime() : 0)

  for (int j = 0; j < n32; j += CB) {
    const int firstTile = j / 16;  // j is a multiple of CB
    t0 = STAMP();
    // ---- phase A: acc = C - L[rows, 0:j] L[j:j+32, 0:j]^T ---------------------------------
    for (int r = wave; r < CB; r += CHOL_WAVES) {
      const double *src = Lw + (size_t)(j + r) * ld;
      for (int c = lane; c < j; c += 64) sB[r * ldb + c] = src[c];
    }
    __syncthreads();
    {
      long long t1 = STAMP();
      tS += t1 - t0;
    }
    {
      // row tiles of this block column: firstTile .. nTiles-1, dealt to the waves round-robin
      const int T0 = firstTile + wave;
      long long *stp = (tim && tid == 0) ? st3 : nullptr;
      const int ntile = T0 < nTiles ? (nTiles - 1 - T0) / CHOL_WAVES + 1 : 0;
      switch (ntile) {
        case 1: chol_update_tiles<1>(Lw, sB, sD, ld, ldb, j, T0, firstTile, li, lk, stp); break;
        case 2: chol_update_tiles<2>(Lw, sB, sD, ld, ldb, j, T0, firstTile, li, lk, stp); break;
        case 3: chol_update_tiles<3>(Lw, sB, sD, ld, ldb, j, T0, firstTile, li, lk, stp); break;
        case 4: if (MAXT >= 4) chol_update_tiles<4>(Lw, sB, sD, ld, ldb, j, T0, firstTile, li, lk, stp); break;
        default: break;
      }
    }
    {
      long long t1 = STAMP();
      tA += t1 - t0;
      t0 = t1;
    }
    // ---- phase B: the diagonal block (now in sD) is factored by one wave -------------------
    __syncthreads();
    if (wave == 0) {
      // lane r < 32 keeps row r of the block in registers (an SSA vector); the pivot and the
      // finished column goes through LDS and is read back with uniform addresses (broadcast).
      const int r = lane & (CB - 1);
      dcb a;
#pragma unroll
      for (int c = 0; c < CB; c++) a[c] = sD[r][c];
      bool bad = false;
#pragma clang loop unroll(full)
      for (int c = 0; c < CB; c++) {
        const double d = readlane_f64(a[c], c);
        bad |= !(d > 0.0);
        double y = __builtin_amdgcn_rsq(d);            // ~ 1/sqrt(d)
        y = y * (1.5 - 0.5 * d * y * y);               // two Newton steps: full fp64
        y = y * (1.5 - 0.5 * d * y * y);
        const double l = (r == c) ? d * y : a[c] * y;  // L[c][c] = sqrt(d), L[r][c] = a/sqrt(d)
        a[c] = l;
        if (lane == c) sInvD[c] = y;                   // 1 / L[c][c]
        if (c + 1 < CB) {
          sCol[r] = l;                                 // finished column -> LDS, read back uniformly
          __builtin_amdgcn_wave_barrier();
#pragma clang loop unroll(full)
          for (int cc = c + 1; cc < CB; cc++) a[cc] -= l * sCol[cc];
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (lane < CB) {
#pragma unroll
        for (int c = 0; c < CB; c++) sD[r][c] = (c <= r) ? a[c] : 0.0;
        if (bad || !isfinite(a[r])) sFail = 1;
      }
      {
        long long t1 = STAMP();
        tF += t1 - t0;
      }
    }
    __syncthreads();
    {
      long long t1 = STAMP();
      tB += t1 - t0;
      t0 = t1;
    }
    if (sFail == 1) break;
    for (int t = tid; t < CB * CB; t += CHOL_THREADS) {
      const int r = t / CB, c = t % CB;
      Lw[(size_t)(j + r) * ld + j + c] = sD[r][c];
      sDiag[(size_t)(j / CB) * CB * (CB + 1) + r * (CB + 1) + c] = sD[r][c];
    }
    // ---- phase C: X L_dd^T = C for the rows below the diagonal block, one thread per row:
    // forward substitution with the row in registers, L_dd from LDS (uniform addresses
    // broadcast).  The e_a row is one of the rows.
    // (16 nTiles <= CHOL_THREADS + CB is checked at launch: one row per thread, no loop, so the
    // block's entries are not hoisted into registers all at once)
    if (const int R = j + CB + tid; R < 16 * nTiles) {
      double *row = Lw + (size_t)R * ld + j;
      dcb xr;
#pragma unroll
      for (int c = 0; c < CB; c++) xr[c] = row[c];
#pragma clang loop unroll(full)
      for (int c = 0; c < CB; c++) {
        double v = xr[c];
#pragma clang loop unroll(full)
        for (int k = 0; k < c; k++) v -= xr[k] * sD[c][k];
        xr[c] = v * sInvD[c];
      }
#pragma unroll
      for (int c = 0; c < CB; c++) row[c] = xr[c];
    }
    __syncthreads();
    {
      long long t1 = STAMP();
      tC += t1 - t0;
      t0 = t1;
    }
  }

  if (sFail == 1) {
    if (tid == 0) status[1] = try_id;
    for (int t = tid; t < n; t += CHOL_THREADS) x[t] = 0.0;
    return;
  }

  // ---- backward solve  L^T x = y  (y = L^-1 e_a now sits in row n32) ----------------------
  // per block J (descending): one wave solves L_dd^T x_J = y_J by substitution (L_dd from LDS),
  // then all threads apply y[c] -= sum_r L[j+r][c] x_J[r] for c < j.  The 16 L values each
  // thread needs for that update do not depend on x, so they are fetched one block ahead.
  double *y = Lw + (size_t)n32 * ld;
  double lnext[CB];
#pragma unroll
  for (int r = 0; r < CB; r++) lnext[r] = (tid < n32 - CB) ? Lw[(size_t)(n32 - CB + r) * ld + tid] : 0.0;
  for (int j = n32 - CB; j >= 0; j -= CB) {
    double lcur[CB];
#pragma unroll
    for (int r = 0; r < CB; r++) lcur[r] = lnext[r];
    if (j >= CB) {
#pragma unroll
      for (int r = 0; r < CB; r++) lnext[r] = (tid < j - CB) ? Lw[(size_t)(j - CB + r) * ld + tid] : 0.0;
    }
    if (wave == 0) {
      const int r = lane & (CB - 1);
      const double *Ld = sDiag + (size_t)(j / CB) * CB * (CB + 1);
      dcb lcol;
#pragma unroll
      for (int k = 0; k < CB; k++) lcol[k] = Ld[k * (CB + 1) + r];  // L_dd[k][r]
      double z = y[j + r];
#pragma clang loop unroll(full)
      for (int k = CB - 1; k >= 0; k--) {
        const double xk = readlane_f64(z, k) / readlane_f64(lcol[k], k);
        if (r == k) z = xk;
        if (r < k) z -= lcol[k] * xk;
      }
      if (lane < CB) {
        sX[r] = z;
        if (j + r < n) x[j + r] = z;
      }
    }
    __syncthreads();
    if (tid < j) {  // CHOL_THREADS >= n32 is checked at launch
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < CB; r++) acc += lcur[r] * sX[r];
      y[tid] -= acc;
    }
    __syncthreads();
  }
  if (tim && tid == 0) {
    tBack = STAMP() - t0;
    tim[0] = tA;
    tim[1] = tB;
    tim[2] = tC;
    tim[3] = tBack;
    tim[4] = tF;
    tim[5] = tS;
    tim[6] = st3[0] + (st3[1] << 0) * 0;
    tim[6] = st3[0];
    tim[7] = st3[1];
  }
  int bad = 0;
  for (int t = tid; t < n; t += CHOL_THREADS)
    if (!isfinite(x[t])) bad = 1;
  if (bad) status[1] = try_id;
}

int launch_chol_solve(psba_ctx *h) {
  // default: the panel chain on the whole chip (kernels_chol_graph.hip); PSBA_CHOL_SINGLE=1
  // selects this file's single-workgroup kernel
  if (!getenv("PSBA_CHOL_SINGLE")) return launch_chol_graph(h);
  const Dims &d = h->d;
  const int n32 = h->n32;
  const int nTiles = n32 / 16 + 1;
  const size_t lds = sizeof(double) * ((size_t)CB * (n32 + 1) + (size_t)(n32 / CB) * CB * (CB + 1));
  if (nTiles > 4 * CHOL_WAVES || 16 * nTiles > CHOL_THREADS + CB || n32 > CHOL_THREADS ||
      lds > 163840 - 20 * 1024)
    return fail(h, PSBA_E_INVALID, "dense solve supports 6*nCams <= 480 for now (got %d)", d.nA);
  if (!h->chol_attr_set) {
    const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_chol_solve<3>, attr, 163840 - 20 * 1024));
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_chol_solve<4>, attr, 163840 - 20 * 1024));
    h->chol_attr_set = true;
  }
  {
    ProfScope ps(h, PSBA_K_CHOLESKY);
    if (nTiles <= 3 * CHOL_WAVES)
      hipLaunchKernelGGL(k_chol_solve<3>, dim3(1), dim3(CHOL_THREADS), lds, h->stream, h->red, h->dp,
                         h->chol_ws, d.nA, n32, h->status, h->try_id, h->chol_tim);
    else
      hipLaunchKernelGGL(k_chol_solve<4>, dim3(1), dim3(CHOL_THREADS), lds, h->stream, h->red, h->dp,
                         h->chol_ws, d.nA, n32, h->status, h->try_id, h->chol_tim);
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
