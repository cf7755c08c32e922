// chol_factor32.h -- factorization + inversion of one 32x32 diagonal block of the reduced camera
// system by four cooperating waves of a workgroup (device code, included by the panel chain in
// kernels_chol_graph.hip and by the S-reduce kernel in kernels_schur.hip, which factors the first
// block while the rest of S is still being summed).  Replaces, for one block, the reference's
// cholesky + trigMat_inv kernels (CL_files/SPD_inv.cl:20-270, PSBA/cl_spdinv.cpp:18-204).
#pragma once
#include <hip/hip_runtime.h>

namespace psba {

// F32_EXP: bit mask of timing-only experiments (scripts/ubench_f32.hip defines it; the product never does):
// 1 the pivot wave does not wait for staged panels, 2 it skips the previous panel's rank-4 update,
// 4 it does not publish 1/d and d.  Wrong numbers, right costs.
#ifndef F32_EXP
#define F32_EXP 0
#endif


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
//   wave 1  inverse of the 32x32 triangular factor, column c per lane pair (lanes c and c + 32 share a
//           column and split every row's sum), rows following the panels as they are published: what a
//           row takes from earlier panels' columns is summed before its own panel arrives, so the wave
//           ends a handful of dependent operations behind the pivot wave;
//   wave 3  square roots of the pivots and the failure flag, once the last panel is out.
// The square roots are taken once at the end, 32 lanes in parallel: L^-1 = diag(sqrt d) D^-1.
// Progress is published through LDS words (one writer each): sFlag[0] = panels in sD,
// sFlag[1] = panels staged by the tile wave (+2), sFlag[2] = tile wave has loaded its tiles.
// A wave's LDS operations complete in order, so data written before a
// flag is visible to whoever sees the flag.  No wave exits early, so every wait ends.
// (Until round 4 two waves inverted the two 16x16 diagonal blocks and combined them by two chained MFMA
// products after the last panel: 1100 cycles behind the pivot wave; scripts/ubench_f32.hip.)
struct Factor32Lds {
  // in: the block (lower triangle valid); out: the columns of its LDL^T-style elimination,
  // D[r][c] = L[r][c] sqrt(d_c) with d_c = D[c][c] the pivots (zeros above the diagonal)
  double D[GB][GB + 1];
  double Li[GB][GB + 1];  // out: inverse of D as a matrix (lower triangle valid)
  double next[2][GB][4];  // staged panels
  alignas(32) double rinv[GB];  // 1 / d_c (written four at a time)
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
  // no release fence: everything the flag announces went to the LDS from this wave before it, and
  // the LDS executes a wave's operations in order -- a fence here is an s_waitcnt lgkmcnt(0), i.e. a
  // full LDS round trip of the serial wave per panel.  (The compiler barriers keep the order.)
  asm volatile("" ::: "memory");
  if (lane == 0) *(volatile lds_int *)flag = v;
  asm volatile("" ::: "memory");
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
template <bool TIMED>
__device__ __forceinline__ void f32_pivot_wave(Factor32Lds &s, int lane, long long *tim) {
  const int row = lane & 31;
  d4 a, tp = {0, 0, 0, 0};  // this panel's columns; the previous panel's divided by their pivots
#pragma clang loop unroll(full)
  for (int q = 0; q < 8; q++) {
    const int j0 = 4 * q;
    double sc[4][4];  // the previous panel's entries in this panel's four pivot rows
    if (q > 0 && !(F32_EXP & 2)) {
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
      } while (f < q - 1 && !(F32_EXP & 1));
    }
    if (q > 0 && !(F32_EXP & 2)) {  // rank-4 update by the previous panel, in the row layout
#pragma clang loop unroll(full)
      for (int k2 = 0; k2 < 4; k2++)
#pragma clang loop unroll(full)
        for (int k = 0; k < 4; k++) a[k2] -= tp[k] * sc[k2][k];
    }
    d4 t, rr;
    double d = readlane_f64g(a[0], j0);
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; k++) {
      // (a pivot that is not positive -- or NaN -- shows in its reciprocal: wave 3 looks at those at the end)
      // 1 / d = r0 (1 + p), p = e + e^2, e = 1 - d r0 (v_rcp_f64 is good to ~2^-24; the cubic step finishes it).
      // Only p sits on the way to the next pivot: the column divided by the seed (u), its product with the
      // column (v) and a[k+1] - u a[k] (m) are formed beside e and p, and the next pivot is m - v p in its own
      // lane -- rcp + three dependent operations from pivot to pivot instead of rcp + five
      const double r0 = __builtin_amdgcn_rcp(d);
      const double e = __builtin_fma(-d, r0, 1.0);
      const double u = a[k] * r0;
      const double pp = __builtin_fma(e, e, e);
      if (k < 3) {
        const double v = u * a[k];
        const double m = __builtin_fma(-u, a[k], a[k + 1]);
        d = readlane_f64g(__builtin_fma(-v, pp, m), j0 + k + 1);
      }
      t[k] = __builtin_fma(u, pp, u);    // a[k] / d
      rr[k] = __builtin_fma(r0, pp, r0);  // 1 / d
      if (k < 3) {
#pragma clang loop unroll(full)
        for (int k2 = k + 1; k2 < 4; k2++) a[k2] -= t[k] * readlane_f64g(a[k], j0 + k2);
      }
    }
    // the tile wave reads the original block first -- but only its columns >= 8 ever leave the
    // tiles again (panels 0 and 1 are read here, from s.D), so the first two panels need not wait
    if (q == 2) f32_wait(&s.flag[2], 1);
#pragma unroll
    for (int k = 0; k < 4; k++) s.D[row][j0 + k] = a[k];  // lanes 32..63 repeat lanes 0..31
    // lane 0: the reciprocals, then the flag (in this order: a wave's LDS operations execute in order; the
    // pivots themselves are the diagonal of what was just published)
    asm volatile("" ::: "memory");
    if (lane == 0) {
      if (!(F32_EXP & 4)) *reinterpret_cast<d4 *>(&s.rinv[j0]) = rr;
      *(volatile lds_int *)&s.flag[0] = q + 1;
    }
    asm volatile("" ::: "memory");
    tp = t;
    if (TIMED && lane == 0) tim[5 + q] = (long long)__builtin_amdgcn_s_memtime();
  }
  // (measured and dropped, scripts/ubench_f32.hip: the next panel's staged columns fetched one to three pivots
  // ahead and its first pivot formed from the lane's own values before the general update -- the LDS round
  // trip of the sixteen scalars stays on the way to the second pivot, and the extra instructions cost more
  // than the first pivot gains: 810 -> 930 cycles per panel)
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
    // rank-4 update A -= (D / d) D^T: one operand scaled by 1 / d_k, k = this lane's k slot.
    // Flag and operands are read in one batch (volatile: issued in this order; a wave's LDS
    // reads complete in order), so a published panel costs one LDS round trip, not two
    double a0 = 0.0, a1, nr;
    int f;
    do {
      f = *(volatile lds_int *)&s.flag[0];
      a1 = *(volatile lds_double *)&s.D[16 + col][j0 + rc];
      nr = -*(volatile lds_double *)&s.rinv[j0 + rc];
      if (p < 2) a0 = *(volatile lds_double *)&s.D[col][j0 + rc];
    } while (f < p + 1);
    if (p < 2) {  // columns < 16 are staged for panels 2, 3 only
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

// Inverse of the eliminated columns D as a lower-triangular matrix X (X D = I), one wave, on the matrix
// pipe, four rows per published panel.  With P the panel's rows and the rows above it done,
//     X[P, :] = inv(D_PP) (E_P - ACC[P, :]),     ACC[r, :] = sum over earlier panels Q of D[r, Q] X[Q, :].
// ACC is kept as three 16x16 accumulator tiles (rows 0-15 x columns 0-15, rows 16-31 x both column halves)
// and grows by one v_mfma_f64_16x16x4_f64 per tile and panel (A = the panel's four columns of D, B = the
// four new rows of X: the accumulator layout of rows 4p .. 4p+3 -- register p mod 4 of lane group k -- IS
// the B layout, so nothing is transposed).  inv(D_PP) of the 4x4 triangle is formed entry by entry in the
// lane that holds it as an A operand: D_PP = (I + N) diag(d) with N strictly lower and N^4 = 0, so
// inv(D_PP) = diag(1/d) (I - N + N^2 - N^3): at most four products per entry.  After the last panel the wave
// is one batch of LDS reads, ~10 dependent operations and one MFMA behind the pivot wave (the two-wave
// scheme it replaces finished with two chained 16x16x16 products: 1100 cycles; scripts/ubench_f32.hip).
// Entries of D above the diagonal are never-initialised LDS (see f32_pivot_wave): they are selected away,
// never multiplied by zero, and as MFMA operands they only reach rows of ACC that are no longer read.
__device__ __forceinline__ void f32_inverse_mfma(Factor32Lds &s, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  const int i = li & 3, k = lk;                       // entry (i, k) of the 4x4 operand (lanes li >= 4: zero rows)
  const int j1 = k + 1 < 3 ? k + 1 : 3, j2 = k + 2 < 3 ? k + 2 : 3;
  const bool lower = li < 4 && i > k, two = li < 4 && i - k >= 2, three = li < 4 && i - k == 3;
  d4 acc00 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const d4 zero = {0, 0, 0, 0};
#pragma clang loop unroll(full)
  for (int p = 0; p < 8; p++) {
    const int j0 = 4 * p;
    f32_wait(&s.flag[0], p + 1);
    const double dik = s.D[j0 + i][j0 + k], dij1 = s.D[j0 + i][j0 + j1], dj1k = s.D[j0 + j1][j0 + k];
    const double dij2 = s.D[j0 + i][j0 + j2], dj2k = s.D[j0 + j2][j0 + k], dj2j1 = s.D[j0 + j2][j0 + j1];
    const double rk = s.rinv[j0 + k], rj1 = s.rinv[j0 + j1], rj2 = s.rinv[j0 + j2], ri = s.rinv[j0 + i];
    const double a0 = p < 4 ? s.D[li][j0 + lk] : 0.0, a1 = s.D[16 + li][j0 + lk];  // the panel's columns, both row blocks
    // N = D_PP diag(1/d) - I;  (I - N + N^2 - N^3)[i][k] for i > k
    const double nik = lower ? dik * rk : 0.0, nj1k = dj1k * rk;
    const double nij1 = two ? dij1 * rj1 : 0.0, nij2 = three ? dij2 * rj2 : 0.0;
    const double nj2k = dj2k * rk, nj2j1 = dj2j1 * rj1;
    double m = -nik + nij1 * nj1k + nij2 * (nj2k - nj2j1 * nj1k);
    m = (li < 4 && i == k) ? 1.0 : m;
    const double A = ri * m;  // inv(D_PP)[i][k] (zero above the diagonal and in the padding rows)
    // right-hand sides: delta - ACC in the rows of this panel (register p mod 4), per column half
    const double dl = (j0 + lk == li) ? 1.0 : 0.0, dr = (j0 + lk == 16 + li) ? 1.0 : 0.0;
    const double rhs0 = p < 4 ? dl - acc00[p & 3] : -acc10[p & 3];
    const d4 x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A, rhs0, zero, 0, 0, 0);
    d4 x1 = zero;
    if (p >= 4) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A, dr - acc11[p & 3], zero, 0, 0, 0);
    s.Li[j0 + lk][li] = x0[0];
    s.Li[j0 + lk][16 + li] = x1[0];  // zero for the rows of the upper block: Li is read as a full 32x32
    if (p < 7) {
      if (p < 4) acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x0[0], acc00, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x0[0], acc10, 0, 0, 0);
      if (p >= 4) acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x1[0], acc11, 0, 0, 0);
    }
  }
}

// Factor the 32x32 block held in s.D (lower triangle valid) and invert the factor into s.Li.
// The caller has zeroed s.flag[] / s.fail and synchronised; needs waves 0..3 of the workgroup.
template <bool TIMED = false>
__device__ __forceinline__ void factor32(Factor32Lds &s, int tid, long long *tim = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    f32_pivot_wave<TIMED>(s, lane, tim);
    if (TIMED && lane == 0) tim[3] = (long long)__builtin_amdgcn_s_memtime();
  } else if (wave == 2) {
    f32_tile_wave(s, lane);
  } else if (wave == 1) {
    f32_inverse_mfma(s, lane);
    if (TIMED && lane == 0) tim[4] = (long long)__builtin_amdgcn_s_memtime();
  } else if (wave == 3) {
    // the square roots of the pivots (the diagonal of the published columns) and the verdict: a pivot
    // that was not positive (or not a number) leaves a reciprocal that is not a positive finite number
    f32_wait(&s.flag[0], 8);
    if (lane < GB) {
      const double d = s.D[lane][lane], r = s.rinv[lane];
      s.sq[lane] = d * rsqrt_nr(d);
      if (!(r > 0.0) || !(r < __builtin_huge_val())) s.fail = 1;
    }
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

}  // namespace psba
