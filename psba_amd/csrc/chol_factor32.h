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
// (4 was: it does not publish 1/d and d -- they are no longer published at all).  Wrong numbers, right costs.
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
  // out: Z = inverse of the UNIT lower factor T + I (T below the diagonal): D^-1 = diag(1 / d) Z, L^-1 = diag(sqrt d / d) Z
  double Li[GB][GB + 1];
  // the columns divided by their pivots, T[r][c] = D[r][c] / d_c (1 on the diagonal, never-initialised above): what
  // the tile wave and the inverse wave multiply with -- nobody but the pivot wave ever needs a reciprocal
  double T[GB][GB + 1];
  double next[2][GB][4];  // staged panels
  double rsq[GB];         // 1 / sqrt(d_c): L = D diag(rsq), L^-1 = diag(rsq) Li   (the pivots d_c are the diagonal of D)
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
    d4 t;
    double d = readlane_f64g(a[0], j0);
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; k++) {
      // (a pivot that is not positive -- or NaN -- stays on the diagonal: wave 3 looks at those at the end)
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
    // the columns divided by their pivots: what the other waves multiply with, so that nobody else needs 1 / d
#pragma unroll
    for (int k = 0; k < 4; k++) s.T[row][j0 + k] = t[k];
    f32_post(&s.flag[0], q + 1, lane);
    tp = t;
    if (TIMED && lane == 0) tim[5 + q] = (long long)__builtin_amdgcn_s_memtime();
  }
  // Measured on the way here (scripts/ubench_f32.hip, cycles per 4-column panel of this wave; round 3's code: 880):
  //  * the pivot-to-pivot chain above instead of reciprocal -> column / d -> next pivot: 880 -> 810;
  //  * 1 / d and d published by lane 0 (two 16-byte stores under an exec mask): ~100 of those 810; not publishing
  //    them and letting the readers form 1 / d: 715, but the inverse wave (four reciprocals per panel) then falls
  //    behind by 80 per panel; publishing the divided columns instead (this version): 790, nobody needs 1 / d;
  //  * the same with the next panel's operands fetched between the two stores, or every column stored as soon as
  //    it is final: 780 / 800, and a longer tail; the next panel's staged columns fetched one to three pivots ahead
  //    and its first pivot formed from the lane's own values: 930.
  // What is left is ~25 instructions per pivot issued by one wave: 4 x 130 cycles, + 170 for the rank-4 update
  // with its LDS round trip, + 85 for the stores.
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
    double a0 = 0.0, t0 = 0.0, a1, t1;
    int f;
    do {
      f = *(volatile lds_int *)&s.flag[0];
      a1 = *(volatile lds_double *)&s.D[16 + col][j0 + rc];
      t1 = *(volatile lds_double *)&s.T[16 + col][j0 + rc];
      if (p < 2) {
        a0 = *(volatile lds_double *)&s.D[col][j0 + rc];
        t0 = *(volatile lds_double *)&s.T[col][j0 + rc];
      }
    } while (f < p + 1);
    if (p < 2) {  // columns < 16 are staged for panels 2, 3 only
      T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-t0, a0, T00, 0, 0, 0);
      T10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-t1, a0, T10, 0, 0, 0);
    }
    T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-t1, a1, T11, 0, 0, 0);
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

// Inverse of the unit lower factor, Z = (I + T)^-1 (T strictly lower: the eliminated columns divided by their
// pivots), one wave, on the matrix pipe, four rows per published panel.  With P the panel's rows,
//     Z[P, :] = inv(I + T_PP) (E_P - ACC[P, :]),     ACC[r, :] = sum over earlier panels Q of T[r, Q] Z[Q, :].
// ACC is kept as three 16x16 accumulator tiles (rows 0-15 x columns 0-15, rows 16-31 x both column halves)
// and grows by one v_mfma_f64_16x16x4_f64 per tile and panel (A = the panel's four columns of T, B = the
// four new rows of Z: the accumulator layout of rows 4p .. 4p+3 -- register p mod 4 of lane group k -- IS
// the B layout, so nothing is transposed).  inv(I + N) of the 4x4 block (N = T_PP, N^4 = 0) is
// I - N + N^2 - N^3, formed entry by entry in the lane that holds it as an A operand: at most four products.
// Per panel the wave's dependent work is one batch of LDS reads, ~6 operations and two MFMAs; no reciprocal
// anywhere (D^-1 = diag(1 / d) Z is never needed: L^-1 = diag(1 / sqrt d) Z, scaled when it is stored).
// (History, scripts/ubench_f32.hip: two waves inverting the two 16x16 diagonal blocks of D row by row and
// combining them with two chained 16x16x16 products finished 1100 cycles behind the pivot wave; this wave
// working on D with the pivots' reciprocals read from LDS 520; with the reciprocals formed here it fell
// behind by 80 cycles per panel.)
// Entries of T above the diagonal are never-initialised LDS (see f32_pivot_wave): they are selected away,
// never multiplied by zero, and as MFMA operands they only reach rows of ACC that are no longer read.
template <bool TIMED = false>
__device__ __forceinline__ void f32_inverse_mfma(Factor32Lds &s, int lane, long long *tim = nullptr) {
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
    if (TIMED && lane == 0) tim[16 + 2 * p] = (long long)__builtin_amdgcn_s_memtime();
    const double tik = s.T[j0 + i][j0 + k], tij1 = s.T[j0 + i][j0 + j1], tj1k = s.T[j0 + j1][j0 + k];
    const double tij2 = s.T[j0 + i][j0 + j2], tj2k = s.T[j0 + j2][j0 + k], tj2j1 = s.T[j0 + j2][j0 + j1];
    const double a0 = p < 4 ? s.T[li][j0 + lk] : 0.0, a1 = s.T[16 + li][j0 + lk];  // the panel's columns, both row blocks
    // (I - N + N^2 - N^3)[i][k] for i > k
    const double nik = lower ? tik : 0.0, nij1 = two ? tij1 : 0.0, nij2 = three ? tij2 : 0.0;
    double m = -nik + nij1 * tj1k + nij2 * (tj2k - tj2j1 * tj1k);
    m = (li < 4 && i == k) ? 1.0 : m;  // (zero above the diagonal and in the padding rows)
    // right-hand sides: delta - ACC in the rows of this panel (register p mod 4), per column half
    const double dl = (j0 + lk == li) ? 1.0 : 0.0, dr = (j0 + lk == 16 + li) ? 1.0 : 0.0;
    const double rhs0 = p < 4 ? dl - acc00[p & 3] : -acc10[p & 3];
    const d4 z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, rhs0, zero, 0, 0, 0);
    d4 z1 = zero;
    if (p >= 4) z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, dr - acc11[p & 3], zero, 0, 0, 0);
    if (p < 7) {
      if (p < 4) acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, z0[0], acc00, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, z0[0], acc10, 0, 0, 0);
      if (p >= 4) acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, z1[0], acc11, 0, 0, 0);
    }
    s.Li[j0 + lk][li] = z0[0];
    s.Li[j0 + lk][16 + li] = z1[0];  // zero for the rows of the upper block: Li is read as a full 32x32
    if (TIMED && lane == 0) tim[17 + 2 * p] = (long long)__builtin_amdgcn_s_memtime();
  }
}

// Factor the 32x32 block held in s.D (lower triangle valid) and invert the factor into s.Li.
// The caller has zeroed s.flag[] / s.fail and synchronised; needs waves 0..3 of the workgroup.
struct F32NoBackground {
  __device__ __forceinline__ void operator()(int) const {}
};
// BG: what the waves beyond the fourth do meanwhile (k_cholg_tail: the block updates that are not on the way to
// the next diagonal block); called with wave - 4, must not touch s
template <bool TIMED = false, class BG = F32NoBackground>
__device__ __forceinline__ void factor32(Factor32Lds &s, int tid, long long *tim = nullptr, const BG &bg = BG()) {
  const int lane = tid & 63, wave = tid >> 6;
  if (wave >= 4) {
    bg(wave - 4);
  } else if (wave == 0) {
    f32_pivot_wave<TIMED>(s, lane, tim);
    if (TIMED && lane == 0) tim[3] = (long long)__builtin_amdgcn_s_memtime();
  } else if (wave == 2) {
    f32_tile_wave(s, lane);
  } else if (wave == 1) {
    f32_inverse_mfma<TIMED>(s, lane, tim);
    if (TIMED && lane == 0) tim[4] = (long long)__builtin_amdgcn_s_memtime();
  } else if (wave == 3) {
    // the square roots of the pivots (the diagonal of the published columns) and the verdict: every pivot
    // positive and finite (a non-positive or NaN pivot stays where it is: the pivot wave only divides by it)
    f32_wait(&s.flag[0], 8);
    if (lane < GB) {
      const double d = s.D[lane][lane], y = rsqrt_nr(d);
      s.rsq[lane] = y;
      if (!(d > 0.0) || !(d < __builtin_huge_val())) s.fail = 1;
    }
  }
  __syncthreads();
}

// entry (r, c) of the Cholesky factor / of its inverse from what factor32 leaves in LDS
__device__ __forceinline__ double f32_L(const Factor32Lds &s, int r, int c) {
  return (c <= r) ? s.D[r][c] * s.rsq[c] : 0.0;
}
__device__ __forceinline__ double f32_Linv(const Factor32Lds &s, int r, int c) {
  return (c <= r) ? s.Li[r][c] * s.rsq[r] : 0.0;
}

}  // namespace psba
