// chol_factor32.h -- factorization + inversion of one 32x32 diagonal block of the reduced camera
// system by four cooperating waves of a workgroup (device code, included by the panel chain in
// kernels_chol_graph.hip and by the S-reduce kernel in kernels_schur.hip, which factors the first
// block while the rest of S is still being summed).  Replaces, for one block, the reference's
// cholesky + trigMat_inv kernels (CL_files/SPD_inv.cl:20-270, PSBA/cl_spdinv.cpp:18-204).
#pragma once
#include <hip/hip_runtime.h>

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
    // the tile wave reads the original block first -- but only its columns >= 8 ever leave the
    // tiles again (panels 0 and 1 are read here, from s.D), so the first two panels need not wait
    if (q == 2) f32_wait(&s.flag[2], 1);
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
      // four partial sums: the wave runs alone on its SIMD, so a single chain of up to 15
      // dependent fma would cost ~16 cycles a link, most of it in the tail behind the pivot wave
      double sm[4] = {0.0, 0.0, 0.0, 0.0};
#pragma clang loop unroll(full)
      for (int m = 0; m < r; m++) sm[m & 3] += s.D[O + r][O + m] * x[m];
      const double v = ((r == c) ? 1.0 : 0.0) - ((sm[0] + sm[1]) + (sm[2] + sm[3]));
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

}  // namespace psba
