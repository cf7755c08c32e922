// kernels_chol_graph.hip -- the dense reduced-camera solve as a right-looking blocked Cholesky
// spread over the whole chip: a short chain of small kernels per 32-column panel, captured
// once into a hipGraph and replayed per damping try.
//
// Same job and same data as kernels_chol.hip (which stays as the single-workgroup fallback):
// S dpa = ea on the padded reduce buffer Lw[(n32+16)][n32], e_a riding along as row n32 so that
// the forward solve is free; replaces SPDinv + matVec_mul (reference PSBA/cl_spdinv.cpp:18-204,
// CL_files/SPD_inv.cl:20-411, PSBA/cl_linearalg.cpp:19).  The reference chains ~nA
// device-enqueued launches of 3x3 blocks; here a panel is three steps:
//   diag    one workgroup factors the 32x32 diagonal block (two 16-step factorizations with rows
//           in registers + a 16x16 update) and inverts the factor -- fused into the tail of the
//           previous update;
//   trsm    one wave per 16-row tile below: X = C L_dd^-T as a 16x32x32 MFMA product;
//   update  one wave per 16x16 tile of the trailing matrix: C -= X_r X_c^T with
//           v_mfma_f64_16x16x4_f64 (K = 32), operands fetched as 64-byte pieces per lane.
// A single workgroup (kernels_chol.hip) spends its time in ~100 dependent barrier phases and on
// one CU's L2 port; here the bulk work runs on all CUs and only the diagonal factor is serial.
// Kernel boundaries inside a graph cost ~1.5 us each.
#include <cstdlib>

#include "psba_internal.h"

namespace psba {

constexpr int GB = 32;  // panel width
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d16 __attribute__((ext_vector_type(16)));  // SSA vector: never demoted to scratch

__device__ __forceinline__ double readlane_f64g(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

// In-place Cholesky of the 16x16 block at sD[o..o+15][o..o+15] by one wave (lanes 0..15 hold
// the rows in registers); sInv[o+c] = 1/L[c][c].  Returns true on a non-positive / non-finite
// pivot.  The dependent chain per column is kept to rsq + two Newton steps + one v_readlane:
//  * the finished column c is broadcast through LDS (sCol, double-buffered) and consumed one
//    step later; its reads are issued at the top of the next step, ahead of the rsq chain;
//  * the next pivot needs column c only through the next row's own entry (l * l in the diagonal
//    lane), so no cross-lane traffic sits between two pivots except the readlane of the pivot.
// Invariant at the top of step c: a[c] holds columns < c-1 for every lane, column c-1 only in
// the diagonal lane (r == c), and d = the finished pivot a[c][c] is known to all lanes.
__device__ __forceinline__ bool factor16(double (*sD)[GB + 1], double (*sCol)[16], double *sInv, int o,
                                         int lane) {
  const int r = lane & 15;
  d16 a;
#pragma unroll
  for (int c = 0; c < 16; c++) a[c] = sD[o + r][o + c];
  bool bad = false;
  double d = readlane_f64g(a[0], 0);
  double lprev = 0.0;
#pragma clang loop unroll(full)
  for (int c = 0; c < 16; c++) {
    d16 v;  // column c-1 of L as broadcast by its owners (entries c..15 used)
    if (c > 0) {
#pragma clang loop unroll(full)
      for (int cc = c; cc < 16; cc++) v[cc] = sCol[(c - 1) & 1][cc];
    }
    bad |= !(d > 0.0);
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    if (c > 0 && r != c) a[c] -= lprev * v[c];
    const double l = (r == c) ? d * y : a[c] * y;  // L[c][c] = sqrt(d), L[r][c] = a / sqrt(d)
    a[c] = l;
    if (c + 1 < 16) {
      if (c > 0) a[c + 1] -= lprev * v[c + 1];
      if (r == c + 1) a[c + 1] -= l * l;
      d = readlane_f64g(a[c + 1], c + 1);
      sCol[c & 1][r] = l;
      if (c > 0) {
#pragma clang loop unroll(full)
        for (int cc = c + 2; cc < 16; cc++) a[cc] -= lprev * v[cc];
      }
      lprev = l;
    }
    if (lane == c) sInv[o + c] = y;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; c++) sD[o + r][o + c] = (c <= r) ? a[c] : 0.0;
    if (!isfinite(a[r])) bad = true;
  }
  return bad;
}

// inverse of the 16x16 lower-triangular block at sD[o..][o..] into sLi[o..][o..]: lane c < 16
// computes column c of the inverse, column-oriented: x_r = v_r / L[r][r], then
// v_r' -= L[r'][r] x_r for the rows below -- independent updates, so the dependent chain per row
// is one multiply and one fma; the L column of the next row is read from LDS one row ahead.
__device__ __forceinline__ void invert16(double (*sD)[GB + 1], double (*sLi)[GB + 1], const double *sInv,
                                         int o, int lane) {
  if (lane >= 16) return;
  const int c = lane;
  d16 v, x, lc, ln;
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] = (r == c) ? 1.0 : 0.0;
#pragma unroll
  for (int rp = 1; rp < 16; rp++) lc[rp] = sD[o + rp][o];
#pragma clang loop unroll(full)
  for (int r = 0; r < 16; r++) {
    if (r + 1 < 16) {
#pragma clang loop unroll(full)
      for (int rp = r + 2; rp < 16; rp++) ln[rp] = sD[o + rp][o + r + 1];
    }
    x[r] = v[r] * sInv[o + r];  // zero for r < c: v stays zero above the diagonal
#pragma clang loop unroll(full)
    for (int rp = r + 1; rp < 16; rp++) v[rp] -= lc[rp] * x[r];
    lc = ln;
  }
#pragma unroll
  for (int r = 0; r < 16; r++) sLi[o + r][o + c] = x[r];
}

// Factor the 32x32 block held in sD (lower triangle valid) and invert the factor into sLi, with
// >= 192 threads of one workgroup: factor D11; [L21 = D21 L11^-T  ||  inv(L11)];
// D22 -= L21 L21^T; factor D22; inv(L22); Li21 = -inv(L22) L21 inv(L11).  Uniform control flow.
__device__ __forceinline__ void factor32(double (*sD)[GB + 1], double (*sLi)[GB + 1],
                                         double (*sCol)[16], double *sInv, int *sFail, int tid,
                                         long long *tim = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  int ts = 1;
#define FSTAMP() do { if (tim && tid == 0) tim[ts++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  FSTAMP();
  if (wave == 0) {
    if (factor16(sD, sCol, sInv, 0, lane)) *sFail = 1;
  }
  __syncthreads();
  FSTAMP();
  if (wave == 1 && lane < 16) {  // row 16 + lane of L21: x L11^T = d21, column-oriented
    const int r = 16 + lane;
    d16 x, lc, ln;
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = sD[r][c];
#pragma unroll
    for (int cp = 1; cp < 16; cp++) lc[cp] = sD[cp][0];
#pragma clang loop unroll(full)
    for (int c = 0; c < 16; c++) {
      if (c + 1 < 16) {
#pragma clang loop unroll(full)
        for (int cp = c + 2; cp < 16; cp++) ln[cp] = sD[cp][c + 1];
      }
      x[c] = x[c] * sInv[c];
#pragma clang loop unroll(full)
      for (int cp = c + 1; cp < 16; cp++) x[cp] -= lc[cp] * x[c];
      lc = ln;
    }
#pragma unroll
    for (int c = 0; c < 16; c++) sD[r][c] = x[c];
  }
  if (wave == 2) invert16(sD, sLi, sInv, 0, lane);
  __syncthreads();
  FSTAMP();
  for (int t = tid; t < 256; t += blockDim.x) {  // D22 -= L21 L21^T (lower part)
    const int r = t >> 4, c = t & 15;
    if (c <= r) {
      double v = sD[16 + r][16 + c];
#pragma unroll
      for (int k = 0; k < 16; k++) v -= sD[16 + r][k] * sD[16 + c][k];
      sD[16 + r][16 + c] = v;
    }
  }
  __syncthreads();
  FSTAMP();
  if (wave == 0) {
    if (factor16(sD, sCol, sInv, 16, lane)) *sFail = 1;
  }
  __syncthreads();
  FSTAMP();
  if (wave == 0) invert16(sD, sLi, sInv, 16, lane);
  // T = L21 inv(L11) into the (unused) upper-right quadrant of sLi
  for (int t = tid; t < 256; t += blockDim.x) {
    const int r = t >> 4, c = t & 15;
    double v = 0.0;
#pragma unroll
    for (int m = 0; m < 16; m++) v += sD[16 + r][m] * sLi[m][c];  // inv(L11) is lower: zeros above
    sLi[r][16 + c] = v;
  }
  __syncthreads();
  FSTAMP();
  double li21[(256 + 191) / 192];
  int q = 0;
  for (int t = tid; t < 256; t += blockDim.x, q++) {
    const int r = t >> 4, c = t & 15;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) v -= sLi[16 + r][16 + k] * sLi[k][16 + c];
    li21[q] = v;
  }
  __syncthreads();
  FSTAMP();
  q = 0;
  for (int t = tid; t < 256; t += blockDim.x, q++) {
    const int r = t >> 4, c = t & 15;
    sLi[16 + r][c] = li21[q];
    sLi[r][16 + c] = 0.0;
  }
  __syncthreads();
  FSTAMP();
}

// diag: factor the block at (j, j) in place and store the inverse of its factor.  Launched
// alone only for the first panel.
__global__ __launch_bounds__(256) void k_cholg_diag(double *Lw, int ld, int j, double *linv, int *status,
                                                    long long *tim) {
  __shared__ double sD[GB][GB + 1], sLi[GB][GB + 1];
  __shared__ double sCol[2][16], sInv[GB];
  __shared__ int sFail;
  const int tid = threadIdx.x;
  if (tid == 0) sFail = 0;
  if (tim && tid == 0) tim[0] = (long long)__builtin_amdgcn_s_memtime();
  for (int t = tid; t < GB * GB; t += 256) sD[t / GB][t % GB] = Lw[(size_t)(j + t / GB) * ld + j + t % GB];
  __syncthreads();
  factor32(sD, sLi, sCol, sInv, &sFail, tid, tim);
  double *li = linv + (size_t)(j / GB) * GB * GB;
  for (int t = tid; t < GB * GB; t += 256) {
    const int r = t / GB, c = t % GB;
    Lw[(size_t)(j + r) * ld + j + c] = (c <= r) ? sD[r][c] : 0.0;
    li[t] = (c <= r) ? sLi[r][c] : 0.0;
  }
  if (tid == 0 && sFail) status[1] = status[3];  // status[3] = this try's stamp
  if (tim && tid == 0) tim[15] = (long long)__builtin_amdgcn_s_memtime();
}

// trsm: X = C L_dd^-T for the 16-row tiles below the panel's diagonal block (incl. the e_a tile),
// one wave per tile: a 16x32x32 product with the stored inverse, 16 MFMAs.  k-slot pairing: MFMA
// step t (0..7) pairs lane slot lk with k = 8 lk + t, so each lane fetches its operand values as
// one 64-byte piece of its row (of C, and of L_dd^-1 whose rows are the columns of L_dd^-T).
__global__ __launch_bounds__(256) void k_cholg_trsm(double *Lw, int ld, int j, int nT, const double *linv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int T = (j + GB) / 16 + blockIdx.x * 4 + wave;
  if (T >= nT) return;
  const double *Li = linv + (size_t)(j / GB) * GB * GB;
  const double4 *ap = reinterpret_cast<const double4 *>(Lw + (size_t)(16 * T + li) * ld + j + 8 * lk);
  const double4 *b0p = reinterpret_cast<const double4 *>(Li + (size_t)li * GB + 8 * lk);
  const double4 *b1p = reinterpret_cast<const double4 *>(Li + (size_t)(16 + li) * GB + 8 * lk);
  const double4 a0 = ap[0], a1 = ap[1], p0 = b0p[0], p1 = b0p[1], q0 = b1p[0], q1 = b1p[1];
  d4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0}, y0 = {0, 0, 0, 0}, y1 = {0, 0, 0, 0};
  x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, p0.x, x0, 0, 0, 0);
  y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, q0.x, y0, 0, 0, 0);
  x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, p0.y, x1, 0, 0, 0);
  y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, q0.y, y1, 0, 0, 0);
  x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.z, p0.z, x0, 0, 0, 0);
  y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.z, q0.z, y0, 0, 0, 0);
  x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.w, p0.w, x1, 0, 0, 0);
  y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.w, q0.w, y1, 0, 0, 0);
  x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, p1.x, x0, 0, 0, 0);
  y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, q1.x, y0, 0, 0, 0);
  x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, p1.y, x1, 0, 0, 0);
  y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, q1.y, y1, 0, 0, 0);
  x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.z, p1.z, x0, 0, 0, 0);
  y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.z, q1.z, y0, 0, 0, 0);
  x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.w, p1.w, x1, 0, 0, 0);
  y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.w, q1.w, y1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + li] = x0[r] + x1[r];
    Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + 16 + li] = y0[r] + y1[r];
  }
}

// one 16x16 tile of the trailing update: C[TR][TC] -= X[TR rows][j..j+31] X[TC rows][j..j+31]^T.
// k-slot pairing: MFMA step t (0..7) pairs lane slot lk with column j + 8 lk + t, so each lane
// fetches its eight operand values as one 64-byte piece of its row.
__device__ __forceinline__ d4 update_tile(const double *Lw, int ld, int j, int TR, int TC, int li,
                                          int lk) {
  const double4 *ap = reinterpret_cast<const double4 *>(Lw + (size_t)(16 * TR + li) * ld + j + 8 * lk);
  const double4 *bp = reinterpret_cast<const double4 *>(Lw + (size_t)(16 * TC + li) * ld + j + 8 * lk);
  const double4 a0 = ap[0], a1 = ap[1], b0 = bp[0], b1 = bp[1];
  d4 c0, c1 = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; r++) c0[r] = Lw[(size_t)(16 * TR + lk + 4 * r) * ld + 16 * TC + li];
  c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0.x, b0.x, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0.y, b0.y, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0.z, b0.z, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0.w, b0.w, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1.x, b1.x, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1.y, b1.y, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1.z, b1.z, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1.w, b1.w, c1, 0, 0, 0);
  return c0 + c1;
}

// update: workgroup 0 owns the three tiles of the next diagonal block and factors it once they
// are updated (the "diag" step of the next panel); every other workgroup owns four tiles
// (one per wave) of the rest of the lower trailing triangle + the e_a tile row.
__global__ __launch_bounds__(256) void k_cholg_update(double *Lw, int ld, int j, int nT, double *linv,
                                                      int *status) {
  __shared__ double sD[GB][GB + 1], sLi[GB][GB + 1];
  __shared__ double sCol[2][16], sInv[GB];
  __shared__ int sFail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int T0 = (j + GB) / 16;  // first trailing tile row / column
  if (blockIdx.x == 0) {
    if (tid == 0) sFail = 0;
    if (wave < 3) {
      const int TR = T0 + (wave > 0), TC = T0 + (wave > 1);
      const d4 c = update_tile(Lw, ld, j, TR, TC, li, lk);
#pragma unroll
      for (int r = 0; r < 4; r++) sD[16 * (TR - T0) + lk + 4 * r][16 * (TC - T0) + li] = c[r];
    }
    __syncthreads();
    factor32(sD, sLi, sCol, sInv, &sFail, tid);
    const int jn = j + GB;
    double *li = linv + (size_t)(jn / GB) * GB * GB;
    for (int t = tid; t < GB * GB; t += 256) {
      const int r = t / GB, c = t % GB;
      Lw[(size_t)(jn + r) * ld + jn + c] = (c <= r) ? sD[r][c] : 0.0;
      li[t] = (c <= r) ? sLi[r][c] : 0.0;
    }
    if (tid == 0 && sFail) status[1] = status[3];
    return;
  }
  // tiles: the lower triangle of the M x M trailing tile grid (row-major: local row m, index
  // m (m + 1) / 2 + local column; the first three indices (0,0), (1,0), (1,1) belong to
  // workgroup 0), followed by the M tiles of the e_a tile row (tile row nT - 1)
  const long long idx = (long long)(blockIdx.x - 1) * 4 + wave + 3;
  const long long M = (nT - 1) - T0;
  const long long ntri = M * (M + 1) / 2;
  if (idx >= ntri + M) return;
  int TR, TC;
  if (idx < ntri) {
    int m = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
    while ((long long)(m + 1) * (m + 2) / 2 <= idx) m++;
    while ((long long)m * (m + 1) / 2 > idx) m--;
    TR = T0 + m;
    TC = T0 + (int)(idx - (long long)m * (m + 1) / 2);
  } else {
    TR = nT - 1;
    TC = T0 + (int)(idx - ntri);
  }
  const d4 c = update_tile(Lw, ld, j, TR, TC, li, lk);
#pragma unroll
  for (int r = 0; r < 4; r++) Lw[(size_t)(16 * TR + lk + 4 * r) * ld + 16 * TC + li] = c[r];
}

// backward solve  L^T x = y  (y = L^-1 e_a sits in row n32), one workgroup, blocks of 32:
// x_J = L_dd^-T y_J is a 32x32 mat-vec with the stored inverse (no dependent chain), then all
// threads apply y[c] -= sum_r L[j+r][c] x_J[r]; the L values of that update do not depend on x
// and are fetched before the mat-vec so that their latency overlaps it.
__global__ __launch_bounds__(512) void k_cholg_backward(double *Lw, int ld, int n, int n32, double *x,
                                                       const double *linv, int *status) {
  __shared__ double sX[GB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  double *y = Lw + (size_t)n32 * ld;
  // operands of block J-1 (its rows of L for the y update, its inverse for the mat-vec) are
  // fetched while block J is processed: none of them depends on x
  double lnext[GB], inext[GB / 2];
  const int c0 = lane & (GB - 1), half = lane >> 5;
  {
    const int j = n32 - GB;
#pragma unroll
    for (int r = 0; r < GB; r++) lnext[r] = (tid < j) ? Lw[(size_t)(j + r) * ld + tid] : 0.0;
    const double *Li = linv + (size_t)(j / GB) * GB * GB;
#pragma unroll
    for (int rr = 0; rr < GB / 2; rr++) inext[rr] = (wave == 0) ? Li[(2 * rr + half) * GB + c0] : 0.0;
  }
  for (int j = n32 - GB; j >= 0; j -= GB) {
    double lcur[GB], icur[GB / 2];
#pragma unroll
    for (int r = 0; r < GB; r++) lcur[r] = lnext[r];
#pragma unroll
    for (int rr = 0; rr < GB / 2; rr++) icur[rr] = inext[rr];
    if (j >= GB) {
      const int jn = j - GB;
#pragma unroll
      for (int r = 0; r < GB; r++) lnext[r] = (tid < jn) ? Lw[(size_t)(jn + r) * ld + tid] : 0.0;
      const double *Li = linv + (size_t)(jn / GB) * GB * GB;
#pragma unroll
      for (int rr = 0; rr < GB / 2; rr++) inext[rr] = (wave == 0) ? Li[(2 * rr + half) * GB + c0] : 0.0;
    }
    if (wave == 0) {
      double acc = 0.0;
#pragma unroll
      for (int rr = 0; rr < GB / 2; rr++)
        acc += icur[rr] * y[j + 2 * rr + half];  // (L_dd^-T y)[c] = sum_r Linv[r][c] y[r]
      acc += __shfl_xor(acc, 32, 64);
      if (lane < GB) {
        sX[c0] = acc;
        if (j + c0 < n) x[j + c0] = acc;
      }
    }
    __syncthreads();
    if (tid < j) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < GB; r++) acc += lcur[r] * sX[r];
      y[tid] -= acc;
    }
    for (int c = tid + nthr; c < j; c += nthr) {  // only when n32 > blockDim.x
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < GB; r++) acc += Lw[(size_t)(j + r) * ld + c] * sX[r];
      y[c] -= acc;
    }
    __syncthreads();
  }
  int bad = 0;
  for (int t = tid; t < n; t += nthr)
    if (!isfinite(x[t])) bad = 1;
  if (bad) status[1] = status[3];
}

static void enqueue_chain(psba_ctx *h, hipStream_t s) {
  const int n32 = h->n32, ld = h->n32, nT = n32 / 16 + 1;  // tile rows incl. the e_a tile
  double *Lw = h->red, *linv = h->chol_ws;
  hipLaunchKernelGGL(k_cholg_diag, dim3(1), dim3(256), 0, s, Lw, ld, 0, linv, h->status, h->chol_tim);
  for (int j = 0; j < n32; j += GB) {
    const int tilesBelow = nT - (j + GB) / 16;  // 16-row tiles below the panel incl. the e_a tile
    hipLaunchKernelGGL(k_cholg_trsm, dim3((tilesBelow + 3) / 4), dim3(256), 0, s, Lw, ld, j, nT, linv);
    if (j + GB < n32) {
      const long long M = (nT - 1) - (j + GB) / 16;
      const long long tiles = M * (M + 1) / 2 + M - 3;
      const int grid = 1 + (int)((tiles + 3) / 4);
      hipLaunchKernelGGL(k_cholg_update, dim3(grid), dim3(256), 0, s, Lw, ld, j, nT, linv, h->status);
    }
  }
  int thr = (n32 + 63) / 64 * 64;
  if (thr > 512) thr = 512;
  hipLaunchKernelGGL(k_cholg_backward, dim3(1), dim3(thr), 0, s, Lw, ld, h->d.nA, n32, h->dp, linv,
                     h->status);
}

int launch_chol_graph(psba_ctx *h) {
  if (!h->chol_graph || h->chol_graph_n32 != h->n32 || h->chol_graph_red != h->red) {
    if (h->chol_graph) {
      (void)hipGraphExecDestroy(h->chol_graph);
      h->chol_graph = nullptr;
    }
    hipGraph_t g = nullptr;
    PSBA_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    enqueue_chain(h, h->stream);
    PSBA_HIP(h, hipStreamEndCapture(h->stream, &g));
    hipError_t e = hipGraphInstantiate(&h->chol_graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
      h->chol_graph = nullptr;
      return fail(h, PSBA_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    h->chol_graph_n32 = h->n32;
    h->chol_graph_red = h->red;
  }
  {
    ProfScope ps(h, PSBA_K_CHOLESKY);
    PSBA_HIP(h, hipGraphLaunch(h->chol_graph, h->stream));
  }
  return PSBA_OK;
}

}  // namespace psba
