// kernels_chol_graph.hip -- the dense reduced-camera solve as a right-looking blocked Cholesky
// spread over the whole chip: a short chain of small kernels per 32-column panel, captured
// once into a hipGraph and replayed per damping try.
//
// Same job and same data as kernels_chol.hip (which stays as the single-workgroup fallback):
// S dpa = ea on the padded reduce buffer Lw[(n32+16)][n32], e_a riding along as row n32 so that
// the forward solve is free; the factor is collected in a second buffer of the same shape (Lx).
// The backward solve rides along too: n32 rows holding the identity below the e_a tile go
// through the same trsm / update steps and end up as L^-T (row i = e_i^T L^-T), so dpa is one
// parallel mat-vec L^-T y at the end instead of a strictly sequential substitution that a
// single CU has to feed from HBM (measured ~20 us of the old 120).  The identity rows cost
// no memory traffic before they are touched: a tile row e is generated in registers until the
// panel that contains column block e, and the tiles left of the diagonal stay zero; replaces SPDinv + matVec_mul (reference PSBA/cl_spdinv.cpp:18-204,
// CL_files/SPD_inv.cl:20-411, PSBA/cl_linearalg.cpp:19).  The reference chains ~nA
// device-enqueued launches of 3x3 blocks; here a panel is three steps (trsm and update fused
// into one kernel, k_cholg_panel, for matrices of the size bundle adjustment usually has):
//   diag    one wave factors the 32x32 diagonal block (4-column panels with rows in registers,
//           rank-4 MFMA updates of register-resident tiles) while a second wave inverts the
//           factor in its wake -- fused into the tail of the previous update;
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
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) double lds_double;
typedef double d16 __attribute__((ext_vector_type(16)));  // SSA vector: never demoted to scratch

__device__ __forceinline__ double readlane_f64g(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}

// ---- the 32x32 diagonal block: four waves of one workgroup, no workgroup barrier inside ----
// The serial part of the whole solve is the chain of nA pivots, so everything that is not a
// pivot step is moved off the wave that runs the chain:
//   wave 0  pivot wave: panels of four columns with lane = row and the four columns in
//           registers, eliminated without square roots (A'[i][j] = A[i][j] - A[i][k] A[j][k] / d_k;
//           the Cholesky factor is these columns scaled by 1 / sqrt(d_k), which nobody on the
//           chain needs).  Per pivot the dependent chain is readlane(pivot) -> rcp + one cubic
//           correction (3 fma) -> column / d -> next pivot (1 fma, in its own lane).  The panel's
//           rank-4 update of the NEXT panel's columns is applied here in the row layout
//           (16 fma with readlane scalars);
//   wave 2  tile wave: keeps the trailing matrix as three 16x16 tiles in the MFMA accumulator
//           layout, applies each published panel as one v_mfma_f64_16x16x4_f64 per tile and
//           stages the columns of panel p+2 (updates <= p applied) for the pivot wave;
//   wave 1  inverse of the upper-left 16x16 of the eliminated columns, column c per lane, rows
//           following the panels as they are published; then T = D21 inv(D11) by MFMA;
//   wave 3  inverse of the lower-right 16x16 the same way, then inv21 = -inv(D22) T by MFMA.
// The square roots are taken once at the end, 32 lanes in parallel: L^-1 = diag(sqrt d) D^-1.
// Progress is published through LDS words (one writer each): sFlag[0] = panels in sD,
// sFlag[1] = panels staged by the tile wave (+2), sFlag[2] = tile wave has loaded its tiles,
// sFlag[3] = T is in LDS.  A wave's LDS operations complete in order, so data written before a
// flag is visible to whoever sees the flag.  No wave exits early, so every wait ends.
struct Factor32Lds {
  // in: the block (lower triangle valid); out: the columns of its LDL^T-style elimination,
  // D[r][c] = L[r][c] sqrt(d_c) with d_c = D[c][c] the pivots (zeros above the diagonal)
  double D[GB][GB + 1];
  double Li[GB][GB + 1];  // out: inverse of D as a matrix (lower triangle valid)
  double next[2][GB][4];  // staged panels
  double rinv[GB];        // 1 / d_c
  double sq[GB];          // sqrt(d_c): L = D diag(1 / sq), L^-1 = diag(sq) Li
  int flag[4];
  int fail;
};

__device__ __forceinline__ void f32_wait(int *flag, int v) {
  // a tight poll: each of the four waves has a SIMD to itself, and the LDS round trip paces it
  while (__hip_atomic_load((lds_int *)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void f32_post(int *flag, int v, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store((lds_int *)flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 1 / d to full precision: v_rcp_f64 is good to ~2^-24, one cubic step (e + e^2) finishes it
__device__ __forceinline__ double recip_cubic(double d) {
  const double r0 = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, r0, 1.0);
  const double p = __builtin_fma(e, e, e);
  return __builtin_fma(r0, p, r0);
}

// The pivot wave issues in order, one instruction every few cycles whether or not it is on the
// dependent chain, so its instruction count is kept down as well: the 16 scalars of the row
// update come as broadcast LDS reads of the panel just published (not 32 readlanes), the
// columns are published unmasked (what lands above the diagonal only ever reaches dead entries:
// the consumers read the lower triangle, and as MFMA operands those values only touch rows /
// columns that are already final), and 1/d, d are stored once per panel.
__device__ __forceinline__ bool f32_pivot_wave(Factor32Lds &s, int lane, long long *tim = nullptr) {
  const int row = lane & 31;
  bool bad = false;
  d4 a, tp = {0, 0, 0, 0};  // this panel's columns; the previous panel's divided by their pivots
#pragma clang loop unroll(full)
  for (int q = 0; q < 8; q++) {
    const int j0 = 4 * q;
    double sc[4][4];  // the previous panel's entries in this panel's four pivot rows
    if (q > 0) {
#pragma unroll
      for (int k2 = 0; k2 < 4; k2++)
#pragma unroll
        for (int k = 0; k < 4; k++) sc[k2][k] = s.D[j0 + k2][j0 - 4 + k];
    }
    if (q < 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = s.D[row][j0 + k];
    } else {
      // flag and data are read in one batch (volatile: issued in this order, and a wave's LDS
      // reads complete in order), so a ready flag costs one LDS round trip, not two
      int f;
      do {
        f = *(volatile lds_int *)&s.flag[1];
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = *(volatile lds_double *)&s.next[q & 1][row][k];
      } while (f < q - 1);
    }
    if (q > 0) {  // rank-4 update by the previous panel, in the row layout
#pragma clang loop unroll(full)
      for (int k2 = 0; k2 < 4; k2++)
#pragma clang loop unroll(full)
        for (int k = 0; k < 4; k++) a[k2] -= tp[k] * sc[k2][k];
    }
    d4 t, rr, dd;
    double d = readlane_f64g(a[0], j0);
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; k++) {
      bad |= !(d > 0.0);  // also NaN; an infinite pivot ends in a non-finite solution, caught there
      const double r = recip_cubic(d);
      t[k] = a[k] * r;
      rr[k] = r;
      dd[k] = d;  // the square root is taken at the end, off the chain
      if (k < 3) {
        // the next pivot, in its own lane: no broadcast between two pivots but the pivot itself
        d = readlane_f64g(__builtin_fma(-t[k], a[k], a[k + 1]), j0 + k + 1);
#pragma clang loop unroll(full)
        for (int k2 = k + 1; k2 < 4; k2++) a[k2] -= t[k] * readlane_f64g(a[k], j0 + k2);
      }
    }
    if (q == 0) f32_wait(&s.flag[2], 1);  // the tile wave reads the original block first
#pragma unroll
    for (int k = 0; k < 4; k++) s.D[row][j0 + k] = a[k];  // lanes 32..63 repeat lanes 0..31
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        s.rinv[j0 + k] = rr[k];
        s.sq[j0 + k] = dd[k];
      }
    }
    f32_post(&s.flag[0], q + 1, lane);
    tp = t;
    if (tim && lane == 0) tim[5 + q] = (long long)__builtin_amdgcn_s_memtime();
  }
  return bad;
}

__device__ __forceinline__ void f32_tile_wave(Factor32Lds &s, int lane) {
  const int col = lane & 15, rc = lane >> 4;
  d4 T00, T10, T11;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    T00[r] = s.D[rc + 4 * r][col];
    T10[r] = s.D[16 + rc + 4 * r][col];
    T11[r] = s.D[16 + rc + 4 * r][16 + col];
  }
  asm volatile("" ::"v"(T00), "v"(T10), "v"(T11));  // the loads are complete before the post
  f32_post(&s.flag[2], 1, lane);
#pragma clang loop unroll(full)
  for (int p = 0; p < 6; p++) {  // panel p+2 <= 7 is the last one to stage
    const int j0 = 4 * p;
    f32_wait(&s.flag[0], p + 1);
    // rank-4 update A -= (D / d) D^T: one operand scaled by 1 / d_k, k = this lane's k slot
    const double a1 = s.D[16 + col][j0 + rc], nr = -s.rinv[j0 + rc];
    if (p < 2) {  // columns < 16 are staged for panels 2, 3 only
      const double a0 = s.D[col][j0 + rc];
      T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0 * nr, a0, T00, 0, 0, 0);
      T10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1 * nr, a0, T10, 0, 0, 0);
    }
    T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1 * nr, a1, T11, 0, 0, 0);
    const int j2 = j0 + 8;  // first column of panel p + 2
    if (j2 < 16) {
      if (col >= j2 && col < j2 + 4) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          s.next[p & 1][rc + 4 * r][col - j2] = T00[r];
          s.next[p & 1][16 + rc + 4 * r][col - j2] = T10[r];
        }
      }
    } else if (col >= j2 - 16 && col < j2 - 12) {
#pragma unroll
      for (int r = 0; r < 4; r++) s.next[p & 1][16 + rc + 4 * r][col - (j2 - 16)] = T11[r];
    }
    f32_post(&s.flag[1], p + 1, lane);
  }
}

// rows [o, o+16) of the inverse of the diagonal 16x16 block of D at (o, o): lane c owns column o + c
template <int O>
__device__ __forceinline__ d16 f32_inverse16(Factor32Lds &s, int lane) {
  const int c = lane & 15;
  d16 x;
#pragma clang loop unroll(full)
  for (int p = 0; p < 4; p++) {
    f32_wait(&s.flag[0], O / 4 + p + 1);
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; k++) {
      const int r = 4 * p + k;
      double v = (r == c) ? 1.0 : 0.0;
#pragma clang loop unroll(full)
      for (int m = 0; m < r; m++) v -= s.D[O + r][O + m] * x[m];
      const double xr = v * s.rinv[O + r];  // 1 / D[r][r]
      x[r] = xr;
      // pin the row here: otherwise the compiler sinks the arithmetic below the last wait and
      // keeps every L value read so far in registers
      asm volatile("" ::"v"(xr));
    }
  }
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 16; r++) s.Li[O + r][O + c] = x[r];
  }
  return x;
}

// 16x16x16 product of two LDS-resident blocks in the MFMA operand layouts: A[i][k] at
// pa[i * lda + k], B[k][j] at pb[k * ldb + j]
__device__ __forceinline__ d4 f32_mm16(const double *pa, int lda, const double *pb, int ldb, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; t++)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[li * lda + 4 * t + lk], pb[(4 * t + lk) * ldb + li], acc, 0, 0, 0);
  return acc;
}

// Factor the 32x32 block held in s.D (lower triangle valid) and invert the factor into s.Li.
// The caller has zeroed s.flag[] / s.fail and synchronised; needs waves 0..3 of the workgroup.
__device__ __forceinline__ void factor32(Factor32Lds &s, int tid, long long *tim = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, rc = lane >> 4;
  if (wave == 0) {
    if (f32_pivot_wave(s, lane, tim)) s.fail = 1;
    if (tim && lane == 0) tim[3] = (long long)__builtin_amdgcn_s_memtime();
  } else if (wave == 2) {
    f32_tile_wave(s, lane);
  } else if (wave == 1) {
    f32_inverse16<0>(s, lane);
    // T = L21 inv(L11) into the (otherwise unused) upper-right quadrant of Li
    f32_wait(&s.flag[0], 4);
    const d4 t = f32_mm16(&s.D[16][0], GB + 1, &s.Li[0][0], GB + 1, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) s.Li[rc + 4 * r][16 + col] = t[r];
    f32_post(&s.flag[3], 1, lane);
  } else if (wave == 3) {
    f32_inverse16<16>(s, lane);
    f32_wait(&s.flag[3], 1);
    const d4 t = f32_mm16(&s.Li[16][16], GB + 1, &s.Li[0][16], GB + 1, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      s.Li[16 + rc + 4 * r][col] = -t[r];
      s.Li[rc + 4 * r][16 + col] = 0.0;  // T is consumed: Li is read as a full 32x32 by the trsm
    }
    if (tim && lane == 0) tim[4] = (long long)__builtin_amdgcn_s_memtime();
  }
  __syncthreads();
  if (tid < GB) {  // sq held the pivots so far
    const double d = s.sq[tid];
    s.sq[tid] = d * rsqrt_nr(d);
  }
  __syncthreads();
}

// entry (r, c) of the Cholesky factor / of its inverse from what factor32 leaves in LDS
__device__ __forceinline__ double f32_L(const Factor32Lds &s, int r, int c) {
  return (c <= r) ? s.D[r][c] * (s.sq[c] * s.rinv[c]) : 0.0;  // 1 / sqrt(d) = sqrt(d) / d
}
__device__ __forceinline__ double f32_Linv(const Factor32Lds &s, int r, int c) {
  return (c <= r) ? s.Li[r][c] * s.sq[r] : 0.0;
}

// diag: factor the block at (j, j) in place and store the inverse of its factor.  Launched
// alone only for the first panel.
__global__ __launch_bounds__(256) void k_cholg_diag(const double *Lw, double *Lx, int ld, int j, double *linv,
                                                    int *status, long long *tim) {
  __shared__ Factor32Lds s;
  const int tid = threadIdx.x;
  if (tid < 4) s.flag[tid] = 0;
  if (tid == 4) s.fail = 0;
  if (tim && tid == 0) tim[0] = (long long)__builtin_amdgcn_s_memtime();
  for (int t = tid; t < GB * GB; t += 256) s.D[t / GB][t % GB] = Lw[(size_t)(j + t / GB) * ld + j + t % GB];
  __syncthreads();
  if (tim && tid == 0) tim[1] = (long long)__builtin_amdgcn_s_memtime();
  factor32(s, tid, tim);
  if (tim && tid == 0) tim[2] = (long long)__builtin_amdgcn_s_memtime();
  double *li = linv + (size_t)(j / GB) * GB * GB;
  for (int t = tid; t < GB * GB; t += 256) {
    const int r = t / GB, c = t % GB;
    Lx[(size_t)(j + r) * ld + j + c] = f32_L(s, r, c);
    li[t] = f32_Linv(s, r, c);
  }
  if (tid == 0 && s.fail) status[1] = status[3];  // status[3] = this try's stamp
  if (tim && tid == 0) tim[15] = (long long)__builtin_amdgcn_s_memtime();
}

// X = C L_dd^-T for one 16-row tile T below the diagonal block of the panel at column j (nT and
// above: the identity rows, see the top of the file), by one wave: a 16x32x32 product with the stored inverse, 16 MFMAs, result in the accumulator layout
// (xl: columns 0..15, xr: 16..31).  k-slot pairing: MFMA step t (0..7) pairs lane slot lk with
// k = 8 lk + t, so each lane fetches its operand values as one 64-byte piece of its row (of C,
// and of L_dd^-1 whose rows are the columns of L_dd^-T).
__device__ __forceinline__ void trsm_tile(const double *Lw, int ld, int j, int T, int nT, const double *Li,
                                          int li, int lk, d4 &xl, d4 &xr) {
  const double4 *b0p = reinterpret_cast<const double4 *>(Li + (size_t)li * GB + 8 * lk);
  const double4 *b1p = reinterpret_cast<const double4 *>(Li + (size_t)(16 + li) * GB + 8 * lk);
  const double4 p0 = b0p[0], p1 = b0p[1], q0 = b1p[0], q1 = b1p[1];
  double4 a0, a1;
  const int e = T - nT;  // >= 0: identity tile row e (rows 16 e .. of the identity)
  if (e >= j / 16) {
    // not touched by any panel yet: its piece of this panel is [I 0] or [0 I] -- generate it
    const int k1 = 16 * (e - j / 16) + li - 8 * lk;  // position of the 1 in this lane's piece
    a0 = make_double4(k1 == 0, k1 == 1, k1 == 2, k1 == 3);
    a1 = make_double4(k1 == 4, k1 == 5, k1 == 6, k1 == 7);
  } else {
    const double4 *ap = reinterpret_cast<const double4 *>(Lw + (size_t)(16 * T + li) * ld + j + 8 * lk);
    a0 = ap[0];
    a1 = ap[1];
  }
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
  xl = x0 + x1;
  xr = y0 + y1;
}

__device__ __forceinline__ void store_x_tile(double *Lx, int ld, int j, int T, int li, int lk, const d4 &xl,
                                             const d4 &xr) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    Lx[(size_t)(16 * T + lk + 4 * r) * ld + j + li] = xl[r];
    Lx[(size_t)(16 * T + lk + 4 * r) * ld + j + 16 + li] = xr[r];
  }
}

// trsm as a kernel of its own: one wave per 16-row tile below the panel's diagonal block (incl.
// the e_a tile), result into the factor buffer Lx.  Used for the last panel, and for every
// panel when the matrix is too large for the fused panel kernel to pay (see enqueue_chain).
__global__ __launch_bounds__(256) void k_cholg_trsm(const double *Lw, double *Lx, int ld, int j, int nT,
                                                    int nTall, const double *linv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int T = (j + GB) / 16 + blockIdx.x * 4 + wave;
  if (T >= nTall) return;  // nTall = nT, or nT + n32 / 16 when the identity rows ride along
  d4 xl, xr;
  trsm_tile(Lw, ld, j, T, nT, linv + (size_t)(j / GB) * GB * GB, li, lk, xl, xr);
  store_x_tile(Lx, ld, j, T, li, lk, xl, xr);
}

// eight operand values (k = 8 lk .. 8 lk + 7 of row li) as MFMA steps
struct Row8 {
  double v[8];
};
__device__ __forceinline__ Row8 load_row8(const double *p) {
  const double4 a = reinterpret_cast<const double4 *>(p)[0], b = reinterpret_cast<const double4 *>(p)[1];
  return {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ d4 update_mfma(d4 c0, const Row8 &a, const Row8 &b) {
  d4 c1 = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.v[t], b.v[t], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.v[t + 1], b.v[t + 1], c1, 0, 0, 0);
  }
  return c0 + c1;
}
__device__ __forceinline__ d4 load_c_tile(const double *Lw, int ld, int TR, int TC, int li, int lk) {
  d4 c;
#pragma unroll
  for (int r = 0; r < 4; r++) c[r] = Lw[(size_t)(16 * TR + lk + 4 * r) * ld + 16 * TC + li];
  return c;
}
__device__ __forceinline__ void store_c_tile(double *Lw, int ld, int TR, int TC, int li, int lk, const d4 &c) {
#pragma unroll
  for (int r = 0; r < 4; r++) Lw[(size_t)(16 * TR + lk + 4 * r) * ld + 16 * TC + li] = c[r];
}
// tile index -> (TR, TC): the lower triangle of the M x M trailing tile grid (row-major: local
// row m, index m (m + 1) / 2 + local column; the first three indices (0,0), (1,0), (1,1) belong
// to workgroup 0), followed by the M tiles of the e_a tile row (tile row nT - 1) and, when the
// identity rows ride along (nId > 0: the identity tile rows 0 .. nId - 1 this panel reaches),
// their M tiles each (tile row nT + e).  fresh: an identity tile not touched before, i.e. zero.
__device__ __forceinline__ bool tile_of_index(long long idx, int T0, int nT, int nId, int &TR, int &TC,
                                              bool &fresh) {
  const long long M = (nT - 1) - T0;
  const long long ntri = M * (M + 1) / 2;
  fresh = false;
  if (idx >= ntri + M + nId * M) return false;
  if (idx < ntri) {
    int m = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
    while ((long long)(m + 1) * (m + 2) / 2 <= idx) m++;
    while ((long long)m * (m + 1) / 2 > idx) m--;
    TR = T0 + m;
    TC = T0 + (int)(idx - (long long)m * (m + 1) / 2);
  } else if (idx < ntri + M) {
    TR = nT - 1;
    TC = T0 + (int)(idx - ntri);
  } else {
    const long long k = idx - ntri - M;
    const int e = (int)(k / M);
    TR = nT + e;
    TC = T0 + (int)(k % M);
    fresh = e >= T0 - 2;  // the two tile rows of this panel's own column block
  }
  return true;
}

// one 16x16 tile of the trailing update: C[TR][TC] -= X[TR rows][j..j+31] X[TC rows][j..j+31]^T,
// X read from the factor buffer.  k-slot pairing: MFMA step t (0..7) pairs lane slot lk with
// column j + 8 lk + t, so each lane fetches its eight operand values as one 64-byte piece of its row.
__device__ __forceinline__ d4 update_tile(const double *Lw, const double *Lx, int ld, int j, int TR, int TC,
                                          int li, int lk) {
  const Row8 a = load_row8(Lx + (size_t)(16 * TR + li) * ld + j + 8 * lk);
  const Row8 b = load_row8(Lx + (size_t)(16 * TC + li) * ld + j + 8 * lk);
  return update_mfma(load_c_tile(Lw, ld, TR, TC, li, lk), a, b);
}

// workgroup 0's tail: factor the next diagonal block (in s.D) and store factor + inverse
__device__ __forceinline__ void factor_next(Factor32Lds &s, double *Lx, int ld, int jn, double *linv,
                                            int *status, int tid) {
  factor32(s, tid);
  double *lio = linv + (size_t)(jn / GB) * GB * GB;
  for (int t = tid; t < GB * GB; t += 256) {
    const int r = t / GB, c = t % GB;
    Lx[(size_t)(jn + r) * ld + jn + c] = f32_L(s, r, c);
    lio[t] = f32_Linv(s, r, c);
  }
  if (tid == 0 && s.fail) status[1] = status[3];
}

// update (unfused chain): workgroup 0 owns the three tiles of the next diagonal block and factors
// it once they are updated (the "diag" step of the next panel); every other workgroup owns four
// tiles (one per wave) of the rest of the lower trailing triangle + the e_a tile row.
__global__ __launch_bounds__(256) void k_cholg_update(double *Lw, double *Lx, int ld, int j, int nT,
                                                      double *linv, int *status) {
  __shared__ Factor32Lds s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int T0 = (j + GB) / 16;  // first trailing tile row / column
  if (blockIdx.x == 0) {
    if (tid < 4) s.flag[tid] = 0;
    if (tid == 4) s.fail = 0;
    if (wave < 3) {
      const int TR = T0 + (wave > 0), TC = T0 + (wave > 1);
      const d4 c = update_tile(Lw, Lx, ld, j, TR, TC, li, lk);
#pragma unroll
      for (int r = 0; r < 4; r++) s.D[16 * (TR - T0) + lk + 4 * r][16 * (TC - T0) + li] = c[r];
    }
    __syncthreads();
    factor_next(s, Lx, ld, j + GB, linv, status, tid);
    return;
  }
  int TR, TC;
  bool fresh;
  if (!tile_of_index((long long)(blockIdx.x - 1) * 4 + wave + 3, T0, nT, 0, TR, TC, fresh)) return;
  store_c_tile(Lw, ld, TR, TC, li, lk, update_tile(Lw, Lx, ld, j, TR, TC, li, lk));
}

// panel (fused chain): trsm + update of the panel at column j in ONE kernel.  Every wave
// computes the two 16-row pieces of X = C L_dd^-T its tile needs itself (32 MFMAs instead of
// reading them: the panel is small and there are more idle SIMDs than tiles), turns them into
// MFMA operands through a wave-private LDS scratch and updates its tile (8 MFMAs).  The waves of
// the first trailing tile column also store their X piece into the factor buffer.  Workgroup 0
// does the same for the next diagonal block and then factors it.  Against trsm and update as
// two kernels this saves a kernel boundary and a global round trip of X per panel; the redundant
// MFMA work runs in the shadow of workgroup 0's serial factorization.
constexpr int XS = 34;  // row stride of the X scratch: 16-byte aligned rows, conflict-free pieces
__global__ __launch_bounds__(256) void k_cholg_panel(double *Lw, double *Lx, int ld, int j, int nT,
                                                     double *linv, int *status) {
  __shared__ Factor32Lds s;
  __shared__ double sX[4][2][16][XS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int T0 = (j + GB) / 16;  // first trailing tile row / column
  const double *Li = linv + (size_t)(j / GB) * GB * GB;
  if (blockIdx.x == 0) {
    if (tid < 4) s.flag[tid] = 0;
    if (tid == 4) s.fail = 0;
    // the serial part starts here, so its loads are issued together up front (they are first
    // touches of what other CUs wrote in the previous kernel: ~1 us each if chained): the three
    // tiles of the next diagonal block, and the operands of the four 16x16 pieces of X (two tile
    // rows x two column halves, one per wave: 8 MFMAs each)
    const int tr = (wave > 0), tc = (wave > 1);
    d4 c = {0, 0, 0, 0};
    if (wave < 3) c = load_c_tile(Lw, ld, T0 + tr, T0 + tc, li, lk);
    {
      const int xr = wave >> 1, half = wave & 1;
      const Row8 a = load_row8(Lw + (size_t)(16 * (T0 + xr) + li) * ld + j + 8 * lk);
      const Row8 b = load_row8(Li + (size_t)(16 * half + li) * GB + 8 * lk);
      d4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
        x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t], b.v[t], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t + 1], b.v[t + 1], x1, 0, 0, 0);
      }
      const d4 x = x0 + x1;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        sX[0][xr][lk + 4 * r][16 * half + li] = x[r];
        Lx[(size_t)(16 * (T0 + xr) + lk + 4 * r) * ld + j + 16 * half + li] = x[r];
      }
    }
    __syncthreads();
    if (wave < 3) {
      const Row8 a = load_row8(&sX[0][tr][li][8 * lk]), b = load_row8(&sX[0][tc][li][8 * lk]);
      c = update_mfma(c, a, b);
#pragma unroll
      for (int r = 0; r < 4; r++) s.D[16 * tr + lk + 4 * r][16 * tc + li] = c[r];
    }
    __syncthreads();
    factor_next(s, Lx, ld, j + GB, linv, status, tid);
    return;
  }
  int TR, TC;
  bool fresh;
  if (!tile_of_index((long long)(blockIdx.x - 1) * 4 + wave + 3, T0, nT, T0, TR, TC, fresh)) return;
  d4 c = {0, 0, 0, 0};
  if (!fresh) c = load_c_tile(Lw, ld, TR, TC, li, lk);  // in flight during the trsm
  d4 xl, xr;
  trsm_tile(Lw, ld, j, TR, nT, Li, li, lk, xl, xr);
  if (TC == T0) store_x_tile(Lx, ld, j, TR, li, lk, xl, xr);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    sX[wave][0][lk + 4 * r][li] = xl[r];
    sX[wave][0][lk + 4 * r][16 + li] = xr[r];
  }
  if (TC != TR) {
    trsm_tile(Lw, ld, j, TC, nT, Li, li, lk, xl, xr);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sX[wave][1][lk + 4 * r][li] = xl[r];
      sX[wave][1][lk + 4 * r][16 + li] = xr[r];
    }
  }
  const Row8 a = load_row8(&sX[wave][0][li][8 * lk]);
  const Row8 b = load_row8(&sX[wave][TC != TR][li][8 * lk]);
  store_c_tile(Lw, ld, TR, TC, li, lk, update_mfma(c, a, b));
}

// backward solve  L^T x = y  (y = L^-1 e_a sits in row n32), one workgroup, blocks of 32:
// x_J = L_dd^-T y_J is a 32x32 mat-vec with the stored inverse (no dependent chain), then all
// threads apply y[c] -= sum_r L[j+r][c] x_J[r]; the L values of that update do not depend on x
// and are fetched before the mat-vec so that their latency overlaps it.
__global__ __launch_bounds__(512) void k_cholg_backward(double *Lw /* the factor buffer */, int ld, int n, int n32, double *x,
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

// dpa = L^-T y: the identity rows of the factor buffer hold L^-T (row i = e_i^T L^-T, upper
// triangular) for all panels but the last, y = L^-1 e_a sits in row n32.  One wave per row, all
// loads of a lane issued at once (n32 <= 640 on this path: at most ten 64-column strides).
// The last panel's trsm is folded in instead of being a kernel of its own: with D the last
// diagonal block, R_i the last 32 columns of identity row i in the working buffer and r those of
// the e_a row, the missing part of the sum is (R_i D^-T)(D^-1 r) = R_i w with
// w = D^-T D^-1 r -- two 32x32 mat-vecs every workgroup does for itself from LDS.
__global__ __launch_bounds__(256) void k_cholg_solve(const double *Lw, const double *Lx, int ld, int n, int n32,
                                                    double *x, const double *linv, int *status) {
  __shared__ double sLi[GB][GB + 1], sV[GB], sW[GB];
  const int tid = threadIdx.x, lane = tid & 63, i = blockIdx.x * 4 + (tid >> 6);
  const int jl = n32 - GB;  // first column of the last panel
  const double *Li = linv + (size_t)(jl / GB) * GB * GB;
  for (int t = tid; t < GB * GB; t += 256) sLi[t / GB][t % GB] = Li[t];
  const double rk = tid < GB ? Lw[(size_t)n32 * ld + jl + tid] : 0.0;
  // the main part of the row's sum is in flight while w is formed
  const bool row = i < n;
  const double *y = Lx + (size_t)n32 * ld;
  const double *z = Lx + (size_t)(n32 + 16 + (row ? i : 0)) * ld;
  double zv[10], yv[10];
#pragma unroll
  for (int m = 0; m < 10; m++) {
    const int c = lane + 64 * m;
    const bool on = row && c < jl && c >= (i & ~15);  // left of the diagonal tile: zeros, never written
    zv[m] = on ? z[c] : 0.0;
    yv[m] = on ? y[c] : 0.0;
  }
  double rt = 0.0;  // R_i[k]: generated for the rows of the last panel's own column block
  if (row && lane < GB) rt = i >= jl ? (double)(lane == i - jl) : Lw[(size_t)(n32 + 16 + i) * ld + jl + lane];
  if (tid < GB) sV[tid] = rk;
  __syncthreads();
  double v = 0.0;
  if (tid < GB) {  // D^-1 r, four independent chains
    double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < GB; k++) a4[k & 3] += sLi[tid][k] * sV[k];
    v = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  __syncthreads();
  if (tid < GB) sV[tid] = v;
  __syncthreads();
  if (tid < GB) {
    double a4[4] = {0.0, 0.0, 0.0, 0.0};  // D^-T (D^-1 r)
#pragma unroll
    for (int c = 0; c < GB; c++) a4[c & 3] += sLi[c][tid] * sV[c];
    sW[tid] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  __syncthreads();
  if (!row) return;
  double acc = lane < GB ? rt * sW[lane] : 0.0;
#pragma unroll
  for (int m = 0; m < 10; m++) acc += zv[m] * yv[m];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    x[i] = acc;
    if (!isfinite(acc)) status[1] = status[3];
  }
}

static void enqueue_chain(psba_ctx *h, hipStream_t s) {
  const int n32 = h->n32, ld = h->n32, nT = n32 / 16 + 1;  // tile rows incl. the e_a tile
  double *Lw = h->red, *Lx = h->chol_L, *linv = h->chol_ws;
  // fused panel kernel (with the identity rows riding along) while a panel's tiles (one wave
  // each, 5x the MFMA work) still fit on the chip's 1024 SIMDs at once; beyond that the
  // redundant work is no longer free: two kernels per panel and a sequential backward solve
  const long long M0 = (nT - 1) - GB / 16;
  const bool fused = !getenv("PSBA_CHOL_UNFUSED") && M0 * (M0 + 1) / 2 + M0 <= 640;
  hipLaunchKernelGGL(k_cholg_diag, dim3(1), dim3(256), 0, s, Lw, Lx, ld, 0, linv, h->status, h->chol_tim);
  for (int j = 0; j < n32; j += GB) {
    const bool last = j + GB >= n32;
    const int T0 = (j + GB) / 16;
    const long long M = (nT - 1) - T0;
    if (last && fused) break;  // the last panel's trsm is part of k_cholg_solve
    if (last || !fused) {
      // 16-row tiles below the panel incl. the e_a tile
      const int nTall = nT;
      hipLaunchKernelGGL(k_cholg_trsm, dim3((nTall - T0 + 3) / 4), dim3(256), 0, s, Lw, Lx, ld, j, nT, nTall, linv);
      if (!last) {
        const int grid = 1 + (int)((M * (M + 1) / 2 + M - 3 + 3) / 4);
        hipLaunchKernelGGL(k_cholg_update, dim3(grid), dim3(256), 0, s, Lw, Lx, ld, j, nT, linv, h->status);
      }
    } else {
      const int grid = 1 + (int)((M * (M + 1) / 2 + M + (long long)T0 * M - 3 + 3) / 4);
      hipLaunchKernelGGL(k_cholg_panel, dim3(grid), dim3(256), 0, s, Lw, Lx, ld, j, nT, linv, h->status);
    }
  }
  if (fused) {
    hipLaunchKernelGGL(k_cholg_solve, dim3((h->d.nA + 3) / 4), dim3(256), 0, s, Lw, Lx, ld, h->d.nA, n32, h->dp,
                       linv, h->status);
  } else {
    int thr = (n32 + 63) / 64 * 64;
    if (thr > 512) thr = 512;
    hipLaunchKernelGGL(k_cholg_backward, dim3(1), dim3(thr), 0, s, Lx, ld, h->d.nA, n32, h->dp, linv,
                       h->status);
  }
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
