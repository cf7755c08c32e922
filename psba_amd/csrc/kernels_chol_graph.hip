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

#include "chol_factor32.h"
#include "psba_internal.h"

namespace psba {

// diag: factor the block at (j, j) in place and store the inverse of its factor.  Launched
// alone only for the first panel.
// a 32x32 block (row stride `rs` doubles) into LDS with 256 threads: the four loads of a thread
// first, then the four stores.  Written as `for (t = tid; t < 1024; t += 256) dst[..] = src[..]`
// the loop is not unrolled and every load is waited for before the next is issued (4 dependent
// global latencies instead of one).
template <class Arr>
__device__ __forceinline__ void stage_block32(Arr &dst, const double *src, size_t rs, int tid) {
  const int r = tid / GB, c = tid % GB;
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = src[(size_t)(r + 8 * q) * rs + c];
#pragma unroll
  for (int q = 0; q < 4; q++) dst[r + 8 * q][c] = v[q];
}

__global__ __launch_bounds__(256) void k_cholg_diag(const double *Lw, double *Lx, int ld, int j, double *linv,
                                                    int *status, long long *tim) {
  __shared__ Factor32Lds s;
  const int tid = threadIdx.x;
  if (tid < 4) s.flag[tid] = 0;
  if (tid == 4) s.fail = 0;
  if (tim && tid == 0) tim[0] = (long long)__builtin_amdgcn_s_memtime();
  stage_block32(s.D, Lw + (size_t)j * ld + j, (size_t)ld, tid);
  __syncthreads();
  if (tim && tid == 0) tim[1] = (long long)__builtin_amdgcn_s_memtime();
  if (tim)
    factor32<true>(s, tid, tim);
  else
    factor32<false>(s, tid);
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
// store_L = false: the fused chain with identity rows -- nobody ever reads the diagonal block of the factor there
// (the panels and the solve work with its inverse), so its 1024 stores stay off workgroup 0's path
__device__ __forceinline__ void factor_next_block(Factor32Lds &s, double *Lx, int ld, int jn, double *linv,
                                            int *status, int tid, bool store_L = true) {
  factor32(s, tid);
  double *lio = linv + (size_t)(jn / GB) * GB * GB;
  for (int t = tid; t < GB * GB; t += 256) {
    const int r = t / GB, c = t % GB;
    if (store_L) Lx[(size_t)(jn + r) * ld + jn + c] = f32_L(s, r, c);
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
    factor_next_block(s, Lx, ld, j + GB, linv, status, tid);
    return;
  }
  int TR, TC;
  bool fresh;
  if (!tile_of_index((long long)(blockIdx.x - 1) * 4 + wave + 3, T0, nT, 0, TR, TC, fresh)) return;
  store_c_tile(Lw, ld, TR, TC, li, lk, update_tile(Lw, Lx, ld, j, TR, TC, li, lk));
}

// ---- two-level blocking for large matrices -------------------------------------------------
// With 32-column panels every panel reads and writes the whole trailing triangle once: 8 n^3 / 96
// bytes in all (144 GB at n = 12 000), which bounds the factorization by HBM at ~16 TFLOP/s whatever
// the MFMA code does (measured 10.5).  Large matrices therefore go in super-panels of NB = 256
// columns: inside a super-panel the 32-column steps update only the super-panel's own remaining
// columns (k_cholg_update_cols), and the rest of the trailing triangle is updated once per
// super-panel with K = NB (k_cholg_update_wide: one wave per 32x32 block, 2x2 MFMA tiles) --
// an eighth of the traffic and eight times the flops per byte.

// one 16x16 tile: C[TR][TC] -= X[TR rows][J..J+KW) X[TC rows][J..J+KW)^T, X from the factor buffer
__device__ __forceinline__ d4 update_tile_k(const double *Lw, const double *Lx, int ld, int J, int KW, int TR,
                                            int TC, int li, int lk) {
  d4 c = load_c_tile(Lw, ld, TR, TC, li, lk);
  for (int kk = 0; kk < KW; kk += GB) {
    const Row8 a = load_row8(Lx + (size_t)(16 * TR + li) * ld + J + kk + 8 * lk);
    const Row8 b = load_row8(Lx + (size_t)(16 * TC + li) * ld + J + kk + 8 * lk);
    c = update_mfma(c, a, b);
  }
  return c;
}

// the 32-column step inside a super-panel: tiles (TR, TC) with TC in [T0, TE), TR in [TC, nT)
// (the e_a tile row nT - 1 included), K = 32 from the panel at column j.  Workgroup 0 owns the
// three tiles of the next diagonal block and factors it (as in k_cholg_update).
__global__ __launch_bounds__(256) void k_cholg_update_cols(double *Lw, double *Lx, int ld, int j, int nT, int TE,
                                                           double *linv, int *status) {
  __shared__ Factor32Lds s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int T0 = (j + GB) / 16;
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
    factor_next_block(s, Lx, ld, j + GB, linv, status, tid);
    return;
  }
  long long idx = (long long)(blockIdx.x - 1) * 4 + wave;
  int TC = T0;
  while (TC < TE && idx >= nT - TC) {
    idx -= nT - TC;
    TC++;
  }
  if (TC >= TE) return;
  const int TR = TC + (int)idx;
  if (TR <= T0 + 1 && TC <= T0 + 1) return;  // workgroup 0's three tiles (and the unused (T0, T0+1))
  store_c_tile(Lw, ld, TR, TC, li, lk, update_tile(Lw, Lx, ld, j, TR, TC, li, lk));
}

// the super-panel's update of everything to its right: K = KW columns of X starting at J, tile
// columns >= Tw.  Waves own 32x32 blocks (macro tiles, 2x2 MFMA tiles: each operand piece is used
// twice) of the lower triangle in units of 32 rows, then single tiles of the e_a tile row.
// Workgroup 0: the next diagonal block, then its factorization.
// ncolb > 0: only the first ncolb macro columns (the look-ahead chain's near update, see k_cholg_update_wide4's part 1,
// in 32x32 pieces: a quarter of the latency of a 64x64 block -- taken when all its waves fit on the chip at once)
__global__ __launch_bounds__(256) void k_cholg_update_wide(double *Lw, double *Lx, int ld, int J, int KW, int nT,
                                                           int Tw, double *linv, int *status, int ncolb) {
  __shared__ Factor32Lds s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  if (blockIdx.x == 0) {
    if (tid < 4) s.flag[tid] = 0;
    if (tid == 4) s.fail = 0;
    if (wave < 3) {
      const int TR = Tw + (wave > 0), TC = Tw + (wave > 1);
      const d4 c = update_tile_k(Lw, Lx, ld, J, KW, TR, TC, li, lk);
#pragma unroll
      for (int r = 0; r < 4; r++) s.D[16 * (TR - Tw) + lk + 4 * r][16 * (TC - Tw) + li] = c[r];
    }
    __syncthreads();
    factor_next_block(s, Lx, ld, 16 * Tw, linv, status, tid);
    return;
  }
  const long long MR = (nT - 1 - Tw) / 2;  // macro rows: 32-row blocks below / right of the super-panel
  const long long ntri = ncolb > 0 ? MR * ncolb : MR * (MR + 1) / 2;
  long long idx = (long long)(blockIdx.x - 1) * 4 + wave + 1;  // macro index 0 = workgroup 0's block
  if (idx >= ntri) {
    // e_a tile row
    const long long e = idx - ntri;
    if (e >= nT - 1 - Tw || (ncolb > 0 && e >= 2 * ncolb)) return;
    const int TC = Tw + (int)e;
    store_c_tile(Lw, ld, nT - 1, TC, li, lk, update_tile_k(Lw, Lx, ld, J, KW, nT - 1, TC, li, lk));
    return;
  }
  int m, mc;
  if (ncolb > 0) {
    m = (int)(idx / ncolb);
    mc = (int)(idx % ncolb);
    if (mc > m) return;
  } else {
    m = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
    while ((long long)(m + 1) * (m + 2) / 2 <= idx) m++;
    while ((long long)m * (m + 1) / 2 > idx) m--;
    mc = (int)(idx - (long long)m * (m + 1) / 2);
  }
  const int TR0 = Tw + 2 * m, TC0 = Tw + 2 * mc;
  const bool dg = m == mc;  // diagonal macro tile: its upper-right 16x16 tile is not part of the triangle
  d4 c00 = load_c_tile(Lw, ld, TR0, TC0, li, lk), c10 = load_c_tile(Lw, ld, TR0 + 1, TC0, li, lk);
  d4 c11 = load_c_tile(Lw, ld, TR0 + 1, TC0 + 1, li, lk);
  d4 c01 = {0, 0, 0, 0};
  if (!dg) c01 = load_c_tile(Lw, ld, TR0, TC0 + 1, li, lk);
  for (int kk = 0; kk < KW; kk += GB) {
    const size_t col = (size_t)J + kk + 8 * lk;
    const Row8 a0 = load_row8(Lx + (size_t)(16 * TR0 + li) * ld + col);
    const Row8 a1 = load_row8(Lx + (size_t)(16 * (TR0 + 1) + li) * ld + col);
    const Row8 b0 = load_row8(Lx + (size_t)(16 * TC0 + li) * ld + col);
    const Row8 b1 = load_row8(Lx + (size_t)(16 * (TC0 + 1) + li) * ld + col);
    c00 = update_mfma(c00, a0, b0);
    c10 = update_mfma(c10, a1, b0);
    c11 = update_mfma(c11, a1, b1);
    if (!dg) c01 = update_mfma(c01, a0, b1);
  }
  store_c_tile(Lw, ld, TR0, TC0, li, lk, c00);
  store_c_tile(Lw, ld, TR0 + 1, TC0, li, lk, c10);
  store_c_tile(Lw, ld, TR0 + 1, TC0 + 1, li, lk, c11);
  if (!dg) store_c_tile(Lw, ld, TR0, TC0 + 1, li, lk, c01);
}

// The same update with a 64x64 block (4x4 MFMA tiles) per wave.  With 2x2 tiles a wave fetches
// 256 B of operands per lane for 32 MFMAs and reads its C tiles before it starts; here every
// 64-byte operand piece feeds four MFMA chains, the next 32-column step's pieces are fetched while
// the current 128 MFMAs run, and the accumulators start at zero so that the C tiles (the HBM
// traffic of the update: the whole trailing triangle once per super-panel) are fetched during the
// last step instead of in front of the first.  FULL: all sixteen tiles are inside the triangle
// (no per-tile tests in the MFMA sequence); otherwise tiles above the diagonal, beyond the last
// tile row (ragged edge) or owned by workgroup 0 are skipped.
template <bool FULL>
__device__ __forceinline__ void wide4_block(double *Lw, const double *Lx, int ld, int J, int KW, int Tw, int last,
                                            int TR0, int TC0, int li, int lk) {
  bool on[4][4];
  d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int TR = TR0 + a, TC = TC0 + b;
      on[a][b] = FULL || (TR <= last && TC <= TR && !(TR <= Tw + 1 && TC <= Tw + 1));
      acc[a][b] = d4{0, 0, 0, 0};
    }
  // rows beyond the last tile row are read from row `last` instead (their products are not used)
  const double *pa[4], *pb[4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int TR = TR0 + a <= last ? TR0 + a : last, TC = TC0 + a <= last ? TC0 + a : last;
    pa[a] = Lx + (size_t)(16 * TR + li) * ld + J + 8 * lk;
    pb[a] = Lx + (size_t)(16 * TC + li) * ld + J + 8 * lk;
  }
  // two operand buffers used in turn (no register copies between steps: a copy makes the compiler
  // wait for the fetch in the middle of the MFMA sequence it is meant to hide behind)
  Row8 a0[4], b0[4], a1[4], b1[4];
#define WIDE4_LOAD(A, B, off)                          \
  _Pragma("unroll") for (int a = 0; a < 4; a++) {      \
    A[a] = load_row8(pa[a] + (off));                   \
    B[a] = load_row8(pb[a] + (off));                   \
  }
#define WIDE4_MFMAS(A, B)                                                                         \
  _Pragma("unroll") for (int t = 0; t < 8; t++) _Pragma("unroll") for (int a = 0; a < 4; a++)     \
      _Pragma("unroll") for (int b = 0; b < 4; b++) if (on[a][b]) acc[a][b] =                     \
          __builtin_amdgcn_mfma_f64_16x16x4f64(A[a].v[t], B[b].v[t], acc[a][b], 0, 0, 0)
  // the block's FIRST 128 MFMAs tile by tile instead of step by step: the first chain needs only the first two of
  // the sixteen operand pieces, so the matrix pipe starts while the rest of the first fetch is still on its way
  // (~1.5 us of the ~2 us a block otherwise waits before its first MFMA, of ~55 us per block)
#define WIDE4_MFMAS_FIRST(A, B)                                                                   \
  _Pragma("unroll") for (int a = 0; a < 4; a++) _Pragma("unroll") for (int b = 0; b < 4; b++)     \
      _Pragma("unroll") for (int t = 0; t < 8; t++) if (on[a][b]) acc[a][b] =                     \
          __builtin_amdgcn_mfma_f64_16x16x4f64(A[a].v[t], B[b].v[t], acc[a][b], 0, 0, 0)
  // KW is a multiple of 2 GB (the host takes the 2x2-tile kernel otherwise): an even number of 32-column steps
  // (the scheduling barriers keep each fetch where it is written: hoisted further up, three operand
  // sets are live at once and the kernel spills)
#define WIDE4_FENCE() __builtin_amdgcn_sched_barrier(0)
  WIDE4_LOAD(a0, b0, 0);
  WIDE4_FENCE();
  WIDE4_LOAD(a1, b1, GB);
  WIDE4_FENCE();
  WIDE4_MFMAS_FIRST(a0, b0);
  // a1 / b1 hold step kk / GB (odd), a0 / b0 are free
  int kk = GB;
  for (; kk + 2 * GB < KW; kk += 2 * GB) {
    WIDE4_FENCE();
    WIDE4_LOAD(a0, b0, kk + GB);
    WIDE4_FENCE();
    WIDE4_MFMAS(a1, b1);
    WIDE4_FENCE();
    WIDE4_LOAD(a1, b1, kk + 2 * GB);
    WIDE4_FENCE();
    WIDE4_MFMAS(a0, b0);
  }
  WIDE4_FENCE();
  d4 c[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (on[a][b]) c[a][b] = load_c_tile(Lw, ld, TR0 + a, TC0 + b, li, lk);
  WIDE4_FENCE();
  WIDE4_MFMAS(a1, b1);
#undef WIDE4_FENCE
#undef WIDE4_MFMAS
#undef WIDE4_MFMAS_FIRST
#undef WIDE4_LOAD
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (on[a][b]) store_c_tile(Lw, ld, TR0 + a, TC0 + b, li, lk, c[a][b] - acc[a][b]);
}

// Workgroup 0: the next diagonal block + its factorization and the e_a tile row's first tiles are as
// in k_cholg_update_wide; every other wave owns one 64x64 block of the lower triangle of the
// trailing square, or one tile of the e_a tile row.
// nranks > 0: the factorization is sharded (0 = the replicated chain; 1 = a one-rank exchange, the test hook) -- this rank updates only the 64-column blocks it owns
// (absolute block B = column / 64, owner B % nranks; the e_a tiles of those columns with them); the
// owners send a super-panel's columns to everybody before it is factored (launch_chol_graph), and the
// next diagonal block is factored after that exchange by k_cholg_diag (workgroup 0's look-ahead would
// read columns this rank may not own).
// part (round 4, the look-ahead of launch_chol_graph): 0 = the whole trailing triangle; 1 = only its first ncolb
// 64-column blocks (the columns of the NEXT super-panel, all rows below) + workgroup 0's diagonal block and its
// factorization; 2 = called with Tw moved past those columns: everything to their right, no workgroup 0 -- so
// that the next super-panel's 32-column steps can run beside the bulk of this update on a second stream.
__global__ __launch_bounds__(256) void k_cholg_update_wide4(double *Lw, double *Lx, int ld, int J, int KW, int nT,
                                                            int Tw, double *linv, int *status, int nranks, int rank,
                                                            int part, int ncolb) {
  __shared__ Factor32Lds s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  if (blockIdx.x == 0) {
    if (nranks > 0 || part == 2) return;
    if (tid < 4) s.flag[tid] = 0;
    if (tid == 4) s.fail = 0;
    if (wave < 3) {
      const int TR = Tw + (wave > 0), TC = Tw + (wave > 1);
      const d4 c = update_tile_k(Lw, Lx, ld, J, KW, TR, TC, li, lk);
#pragma unroll
      for (int r = 0; r < 4; r++) s.D[16 * (TR - Tw) + lk + 4 * r][16 * (TC - Tw) + li] = c[r];
    }
    __syncthreads();
    factor_next_block(s, Lx, ld, 16 * Tw, linv, status, tid);
    return;
  }
  // (grouping the blocks in 8x8 patches per XCD took the L2 misses from 41 % to 23 % of the requests
  // and the kernel from 276 to 292 us: the operand re-reads are served by the Infinity Cache fast
  // enough, the uneven patch counts per XCD cost more -- row-major block order it is)
  const int nTl = nT - 1 - Tw;                 // tile rows of the trailing square (without the e_a row)
  const long long MR = (nTl + 3) / 4;          // 64-row block rows
  const long long ntri = part == 1 ? MR * ncolb : MR * (MR + 1) / 2;
  const long long idx = (long long)(blockIdx.x - 1) * 4 + wave;
  if (idx >= ntri) {
    const long long e = idx - ntri;  // e_a tile row
    if (e >= nTl || (part == 1 && e >= 4 * ncolb)) return;
    if (nranks > 0 && (Tw + (int)e) / 4 % nranks != rank) return;
    const int TC = Tw + (int)e;
    store_c_tile(Lw, ld, nT - 1, TC, li, lk, update_tile_k(Lw, Lx, ld, J, KW, nT - 1, TC, li, lk));
    return;
  }
  int m, mc;
  if (part == 1) {  // the first ncolb block columns, all block rows at or below the diagonal
    m = __builtin_amdgcn_readfirstlane((int)(idx / ncolb));
    mc = __builtin_amdgcn_readfirstlane((int)(idx % ncolb));
    if (mc > m) return;
  } else {
    // the blocks below the diagonal first, row by row; the diagonal blocks (10 of 16 tiles: shorter) last, so
    // that the launch's last, partly filled round of waves is made of the short ones
    const long long nfull = MR * (MR - 1) / 2;
    if (idx < nfull) {
      m = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
      while ((long long)(m + 1) * (m + 2) / 2 <= idx) m++;
      while ((long long)m * (m + 1) / 2 > idx) m--;
      mc = __builtin_amdgcn_readfirstlane((int)(idx - (long long)m * (m + 1) / 2));
      m = __builtin_amdgcn_readfirstlane(m + 1);
    } else {
      m = mc = __builtin_amdgcn_readfirstlane((int)(idx - nfull));
    }
  }

  if (nranks > 0 && (Tw / 4 + mc) % nranks != rank) return;
  const int TR0 = Tw + 4 * m, TC0 = Tw + 4 * mc;
  const int last = nT - 2;  // last tile row / column of the square
  // (sharded, or the far part of a split update: nobody else forms the first diagonal block's tiles)
  const int Tx = (nranks > 0 || part == 2) ? -4 : Tw;
  if (mc < m && TR0 + 3 <= last && (m > 0))
    wide4_block<true>(Lw, Lx, ld, J, KW, Tx, last, TR0, TC0, li, lk);
  else
    wide4_block<false>(Lw, Lx, ld, J, KW, Tx, last, TR0, TC0, li, lk);
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
// nId: identity tile rows that ride along (T0 in the fused chain, whose solve is a product with the
// transformed identity rows; 0 in the mid-size chain, which solves backward through the factor).
// TE >= 0: the 32-column step inside a super-panel of the two-level chain -- only the tiles of the
// tile columns [T0, TE), all rows below (as k_cholg_update_cols, whose trsm kernel this saves).
// factor_next = 0: the launch in front of k_cholg_tail -- workgroup 0 updates the three tiles of the next diagonal
// block like any other tile and does not factor it (the tail kernel factors the whole trailing triangle)
__global__ __launch_bounds__(256) void k_cholg_panel(double *Lw, double *Lx, int ld, int j, int nT, int nId, int TE,
                                                     double *linv, int *status, int factor_next) {
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
        // (with the identity rows riding along only the e_a row's and their X are ever read back: k_cholg_solve)
        if (nId == 0) Lx[(size_t)(16 * (T0 + xr) + lk + 4 * r) * ld + j + 16 * half + li] = x[r];
      }
    }
    __syncthreads();
    if (wave < 3) {
      const Row8 a = load_row8(&sX[0][tr][li][8 * lk]), b = load_row8(&sX[0][tc][li][8 * lk]);
      c = update_mfma(c, a, b);
      if (!factor_next) {
        store_c_tile(Lw, ld, T0 + tr, T0 + tc, li, lk, c);
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) s.D[16 * tr + lk + 4 * r][16 * tc + li] = c[r];
      }
    }
    if (!factor_next) return;  // (uniform)
    __syncthreads();
    factor_next_block(s, Lx, ld, j + GB, linv, status, tid, nId == 0);
    return;
  }
  int TR, TC;
  bool fresh = false;
  if (TE >= 0) {
    long long idx = (long long)(blockIdx.x - 1) * 4 + wave;
    TC = T0;
    while (TC < TE && idx >= nT - TC) {
      idx -= nT - TC;
      TC++;
    }
    if (TC >= TE) return;
    TR = TC + (int)idx;
    if (TR <= T0 + 1 && TC <= T0 + 1) return;  // workgroup 0's three tiles (and the unused (T0, T0+1))
  } else if (!tile_of_index((long long)(blockIdx.x - 1) * 4 + wave + 3, T0, nT, nId, TR, TC, fresh)) {
    return;
  }
  d4 c = {0, 0, 0, 0};
  if (!fresh) c = load_c_tile(Lw, ld, TR, TC, li, lk);  // in flight during the trsm
  d4 xl, xr;
  trsm_tile(Lw, ld, j, TR, nT, Li, li, lk, xl, xr);
  if (TC == T0 && (nId == 0 || TR >= nT - 1)) store_x_tile(Lx, ld, j, TR, li, lk, xl, xr);
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

// ---- the rows below a super-panel, once (round 4) ---------------------------------------------------------
// Inside a super-panel the 32-column steps used to work on ALL rows below: every step read and wrote the
// super-panel's remaining columns of the whole trailing matrix (~37 MB per step at n = 12 000: 13.8 GB per
// factorization, the in-super-panel steps were HBM traffic, not latency -- profiles/r04_cfg5).  Now the
// steps factor only the super-panel's own NB x NB diagonal block, and this kernel does the rest in one pass: one
// wave per 16-row tile row below, X = C L_D^-T by block forward substitution over the nb = NB / 32 column blocks,
//     X_k = (C_k - sum_{k' < k} X_k' L_kk'^T) L_kk^-T,
// all of it wave-local (the finished pieces stay in registers as MFMA operands; L_D is 1.2 MB, cache-resident),
// C read once, X written once.  Same flops as before, a sixth of the traffic, one launch instead of nb.
// NBK = NB / 32 column blocks (a template parameter: the pieces live in registers, every index a constant)
template <int NBK>
__global__ __launch_bounds__(256) void k_cholg_trsm_block(const double *Lw, double *Lx, int ld, int J, int T_first,
                                                          int nTall, const double *linv) {
  __shared__ double sX[4][16][XS];  // a wave's scratch: accumulator layout -> operand layout
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int T = T_first + blockIdx.x * 4 + wave;
  if (T >= nTall) return;
  Row8 X[NBK];  // the finished pieces as A operands (row li, columns 8 lk .. 8 lk + 7 of the piece)
#pragma unroll
  for (int k = 0; k < NBK; k++) {
    const int j = J + GB * k;
    // the piece of C, in the accumulator layout (two 16x16 halves)
    d4 cl, cr;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      cl[r] = Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + li];
      cr[r] = Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + 16 + li];
    }
    // minus the finished pieces times the blocks of L_D to the left of the diagonal
#pragma unroll
    for (int k2 = 0; k2 < k; k2++) {
      const double *Lkk = Lx + (size_t)j * ld + J + GB * k2;  // L_D block (k, k2): rows j .., columns J + 32 k2 ..
      const Row8 b0 = load_row8(Lkk + (size_t)li * ld + 8 * lk), b1 = load_row8(Lkk + (size_t)(16 + li) * ld + 8 * lk);
      cl = update_mfma(cl, X[k2], b0);
      cr = update_mfma(cr, X[k2], b1);
    }
    // through the scratch into the operand layout, times L_kk^-T
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sX[wave][lk + 4 * r][li] = cl[r];
      sX[wave][lk + 4 * r][16 + li] = cr[r];
    }
    const Row8 a = load_row8(&sX[wave][li][8 * lk]);
    const double *Li = linv + (size_t)(j / GB) * GB * GB;
    const Row8 p = load_row8(Li + (size_t)li * GB + 8 * lk), q = load_row8(Li + (size_t)(16 + li) * GB + 8 * lk);
    d4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0}, y0 = {0, 0, 0, 0}, y1 = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t], p.v[t], x0, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t], q.v[t], y0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t + 1], p.v[t + 1], x1, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t + 1], q.v[t + 1], y1, 0, 0, 0);
    }
    const d4 xl = x0 + x1, xr = y0 + y1;
    store_x_tile(Lx, ld, j, T, li, lk, xl, xr);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sX[wave][lk + 4 * r][li] = xl[r];
      sX[wave][lk + 4 * r][16 + li] = xr[r];
    }
    X[k] = load_row8(&sX[wave][li][8 * lk]);
  }
}

// The same block solve with FOUR waves per 16-row tile row.  One wave per tile row is bound by its own SIMD: 16 x NB^2
// flops = 1248 MFMAs of 64 cycles at NB = 384, ~46 us at best and ~80 measured, however few rows there are -- and in
// the look-ahead chain this kernel sits between a super-panel's steps and its update.  Here the wave k mod 4 owns
// column block k; right-looking inside the workgroup: the owner finishes X_k = C_k L_kk^-T and publishes it in LDS
// (operand layout), every wave subtracts X_k L_k'k^T from the blocks k' > k it owns -- the owner of k + 1 first, which
// then finishes and publishes X_k+1 before it turns to its other blocks.  Per column block the critical path is
// 16 + 16 MFMAs and two LDS transposes; the other 1200 MFMAs run beside it on the other three SIMDs.
template <int NBK>
__global__ __launch_bounds__(256, NBK <= 12 ? 2 : 1) void k_cholg_trsm_block4(const double *Lw, double *Lx, int ld, int J, int T_first,
                                                           int nTall, const double *linv) {
  __shared__ double sA[4][16][XS];  // a wave's scratch: accumulator layout -> operand layout
  __shared__ double sX[2][16][XS];  // the published X_k (operand layout), by parity of k
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int T = T_first + blockIdx.x;
  if (T >= nTall) return;
  constexpr int NQ = (NBK + 3) / 4;
  d4 cl[NQ], cr[NQ];  // the pieces of C of the blocks 4 q + wave, accumulator layout (two 16x16 halves)
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int k = 4 * q + wave, j = J + GB * k;
    if (k < NBK) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        cl[q][r] = Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + li];
        cr[q][r] = Lw[(size_t)(16 * T + lk + 4 * r) * ld + j + 16 + li];
      }
    }
  }
  // finish block k (mine, fully updated): X_k = C_k L_kk^-T, stored and published.  p, q8: the rows of L_kk^-1
  auto finish = [&](int q, int k, const Row8 &p, const Row8 &q8) {
    const int j = J + GB * k;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sA[wave][lk + 4 * r][li] = cl[q][r];
      sA[wave][lk + 4 * r][16 + li] = cr[q][r];
    }
    const Row8 a = load_row8(&sA[wave][li][8 * lk]);
    d4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0}, y0 = {0, 0, 0, 0}, y1 = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t], p.v[t], x0, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t], q8.v[t], y0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t + 1], p.v[t + 1], x1, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[t + 1], q8.v[t + 1], y1, 0, 0, 0);
    }
    const d4 xl = x0 + x1, xr = y0 + y1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sX[k & 1][lk + 4 * r][li] = xl[r];
      sX[k & 1][lk + 4 * r][16 + li] = xr[r];
    }
    store_x_tile(Lx, ld, j, T, li, lk, xl, xr);
  };
  auto linv_rows = [&](int k, Row8 &p, Row8 &q8) {
    const double *Li = linv + (size_t)((J + GB * k) / GB) * GB * GB;
    p = load_row8(Li + (size_t)li * GB + 8 * lk);
    q8 = load_row8(Li + (size_t)(16 + li) * GB + 8 * lk);
  };
  // the operands of step k's updates -- the blocks L_D(k2, k) of my blocks k2 > k -- depend on nothing computed
  // here: they are fetched a step ahead (behind the previous step's MFMAs, in front of the barrier), and so are
  // the rows of L_k+1,k+1^-1 for the wave that finishes block k + 1.  Fetched where they are used, every one of the
  // twelve steps waited for two round trips to the L2 (~35 us per launch; ~18 this way)
  Row8 B0[NQ], B1[NQ], P, Q;
  auto fetch_step = [&](int k) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int k2 = 4 * q + wave;
      if (k2 > k && k2 < NBK) {
        const double *Lkk = Lx + (size_t)(J + GB * k2) * ld + J + GB * k;  // L_D block (k2, k)
        B0[q] = load_row8(Lkk + (size_t)li * ld + 8 * lk);
        B1[q] = load_row8(Lkk + (size_t)(16 + li) * ld + 8 * lk);
      }
    }
    if (k + 1 < NBK && ((k + 1) & 3) == wave) linv_rows(k + 1, P, Q);
  };
  if (wave == 0) {
    linv_rows(0, P, Q);
    finish(0, 0, P, Q);
  }
  fetch_step(0);
#pragma unroll
  for (int k = 0; k < NBK; k++) {
    __syncthreads();  // X_k is published (and everybody is done with the buffer X_k+1 will take)
    if (k + 1 >= NBK) break;
    const Row8 X = load_row8(&sX[k & 1][li][8 * lk]);
    // my blocks behind k, nearest first; the owner of k + 1 publishes X_k+1 as soon as its block is complete
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int k2 = 4 * q + wave;
      if (k2 > k && k2 < NBK) {
        cl[q] = update_mfma(cl[q], X, B0[q]);
        cr[q] = update_mfma(cr[q], X, B1[q]);
        if (k2 == k + 1) finish(q, k2, P, Q);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (hoisted above the MFMAs, two operand sets are live at once: 270 registers)
    if (k + 2 < NBK) fetch_step(k + 1);
  }
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

// backward solve for large matrices: the same block recurrence as k_cholg_backward, but one kernel
// per 32-column block J (from the last to the first) spread over the chip instead of one workgroup
// walking the whole factor (which reads 8 n^2 / 2 bytes through a single CU: 52 ms at n = 12 000).
// Every workgroup forms x_J = L_JJ^-T y_J for itself (a 32x32 mat-vec with the stored inverse, from
// LDS), then each thread applies y[c] -= sum_r L[J+r][c] x_J[r] to one column c < J; workgroup 0
// stores x_J.  y_J is only read here and y[c < J] only written, so nothing races inside a launch.
__global__ __launch_bounds__(256) void k_cholg_back_panel(double *Lw /* the factor buffer */, int ld, int n, int n32,
                                                          int j, double *x, const double *linv, int *status) {
  __shared__ double sLi[GB][GB + 1];
  __shared__ double sY[GB], sX[GB];
  const int tid = threadIdx.x;
  double *y = Lw + (size_t)n32 * ld;
  const double *Li = linv + (size_t)(j / GB) * GB * GB;
  stage_block32(sLi, Li, GB, tid);
  if (tid < GB) sY[tid] = y[j + tid];
  __syncthreads();
  if (tid < GB) {
    double a4[4] = {0.0, 0.0, 0.0, 0.0};  // (L_dd^-T y)[c] = sum_r Linv[r][c] y[r]
#pragma unroll
    for (int r = 0; r < GB; r++) a4[r & 3] += sLi[r][tid] * sY[r];
    const double v = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    sX[tid] = v;
    if (blockIdx.x == 0) {
      if (j + tid < n) x[j + tid] = v;
      if (!isfinite(v)) status[1] = status[3];
    }
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + tid;
  if (c >= j) return;
  const double *L = Lw + (size_t)j * ld + c;
  double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < GB; r++) a4[r & 3] += L[(size_t)r * ld] * sX[r];
  y[c] -= (a4[0] + a4[1]) + (a4[2] + a4[3]);
}

// The same with up to four 32-column blocks per launch (columns [j0, j0 + 32 nb), last block first):
// every workgroup walks the small triangular part for itself -- x_b = L_bb^-T y_b, then
// y_b' -= L[b][b']^T x_b for the blocks b' < b of the launch, kept in LDS, never written back -- and
// then applies all 32 nb rows at once to its columns c < j0.  A launch costs ~5 us whatever it
// does, so four blocks per launch take the backward solve from 4.6 to ~2 us per block.
constexpr int BACK_NB = 4;
__global__ __launch_bounds__(256) void k_cholg_back_multi(double *Lw /* the factor buffer */, int ld, int n, int n32,
                                                          int j0, int nb, double *x, const double *linv, int *status) {
  // everything the serial part reads is fetched up front (the addresses depend on nothing computed
  // here): the nb inverse diagonal blocks and the nb (nb - 1) / 2 blocks L[b][b'] between them
  __shared__ double sLi[BACK_NB][GB][GB + 1];
  __shared__ double sL[BACK_NB * (BACK_NB - 1) / 2][GB][GB + 1];
  __shared__ double sY[BACK_NB * GB], sX[BACK_NB * GB], sQ[8][GB], sQ2[2][128];
  const int tid = threadIdx.x;
  double *y = Lw + (size_t)n32 * ld;
  // (all loads first, into registers, then the LDS stores: written as load-store loops the compiler
  // waits for every load before the next one -- 13 us of the kernel's 17)
  double vLi[BACK_NB][4], vL[BACK_NB * (BACK_NB - 1) / 2][4];
  const int tr = tid / GB, tc = tid % GB;  // element (tr + 8 q, tc) of a 32x32 block, q = 0..3
#pragma unroll
  for (int b = 0; b < BACK_NB; b++) {
    const double *Li = linv + (size_t)(j0 / GB + (b < nb ? b : 0)) * GB * GB;
#pragma unroll
    for (int q = 0; q < 4; q++) vLi[b][q] = Li[tid + 256 * q];
#pragma unroll
    for (int bp = 0; bp < b; bp++) {
      const double *L = Lw + (size_t)(j0 + (b < nb ? b : 0) * GB) * ld + j0 + bp * GB;
#pragma unroll
      for (int q = 0; q < 4; q++) vL[b * (b - 1) / 2 + bp][q] = L[(size_t)(tr + 8 * q) * ld + tc];
    }
  }
  const double vy = tid < nb * GB ? y[j0 + tid] : 0.0;
  // 64 columns c < j0 per workgroup, wave w applies block w's 32 rows to them: all of a thread's
  // loads are in flight during the serial part (their addresses depend on nothing computed here)
  __shared__ double sP[BACK_NB][64];
  const int w = tid >> 6, lane = tid & 63;
  const int c = blockIdx.x * 64 + lane;
  const bool mine = c < j0 && w < nb;
  const double *L = Lw + (size_t)(j0 + w * GB) * ld + (mine ? c : 0);
  double pre[GB];
#pragma unroll
  for (int r = 0; r < GB; r++) pre[r] = mine ? L[(size_t)r * ld] : 0.0;
  if (tid < nb * GB) sY[tid] = vy;
#pragma unroll
  for (int b = 0; b < BACK_NB; b++) {
#pragma unroll
    for (int q = 0; q < 4; q++) sLi[b][tr + 8 * q][tc] = vLi[b][q];
#pragma unroll
    for (int bp = 0; bp < b; bp++)
#pragma unroll
      for (int q = 0; q < 4; q++) sL[b * (b - 1) / 2 + bp][tr + 8 * q][tc] = vL[b * (b - 1) / 2 + bp][q];
  }
  __syncthreads();
  for (int b = nb - 1; b >= 0; b--) {
    {  // (L_bb^-T y_b)[c] = sum_r Linv[r][c] y_b[r]: eight threads per column, four rows each
      const int cc = tid & 31, pp = tid >> 5;
      const double q0 = sLi[b][4 * pp][cc] * sY[b * GB + 4 * pp], q1 = sLi[b][4 * pp + 1][cc] * sY[b * GB + 4 * pp + 1];
      const double q2 = sLi[b][4 * pp + 2][cc] * sY[b * GB + 4 * pp + 2], q3 = sLi[b][4 * pp + 3][cc] * sY[b * GB + 4 * pp + 3];
      sQ[pp][cc] = (q0 + q1) + (q2 + q3);
    }
    __syncthreads();
    if (tid < GB) {
      const double v = ((sQ[0][tid] + sQ[1][tid]) + (sQ[2][tid] + sQ[3][tid])) + ((sQ[4][tid] + sQ[5][tid]) + (sQ[6][tid] + sQ[7][tid]));
      sX[b * GB + tid] = v;
      if (blockIdx.x == 0) {
        if (j0 + b * GB + tid < n) x[j0 + b * GB + tid] = v;
        if (!isfinite(v)) status[1] = status[3];
      }
    }
    __syncthreads();
    {  // the launch's own blocks to the left: y_b'[c] -= sum_r L[b][b'][r][c] x_b[r], two threads per entry
      const int o = tid & 127, half = tid >> 7;
      if (o < b * GB) {
        const int bp = o / GB, cc = o % GB;
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < GB / 2; r++)
          a4[r & 3] += sL[b * (b - 1) / 2 + bp][GB / 2 * half + r][cc] * sX[b * GB + GB / 2 * half + r];
        sQ2[half][o] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      }
    }
    __syncthreads();
    if (tid < b * GB) sY[tid] -= sQ2[0][tid] + sQ2[1][tid];
    __syncthreads();
  }
  double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < GB; r++) a4[r & 3] += pre[r] * sX[(w < nb ? w : 0) * GB + r];
  sP[w][lane] = mine ? (a4[0] + a4[1]) + (a4[2] + a4[3]) : 0.0;
  __syncthreads();
  if (w == 0 && c < j0) y[c] -= (sP[0][lane] + sP[1][lane]) + (sP[2][lane] + sP[3][lane]);
}

// dpa = L^-T y: the identity rows of the factor buffer hold L^-T (row i = e_i^T L^-T, upper
// triangular) for all panels but the last, y = L^-1 e_a sits in row n32.  One wave per row, all
// loads of a lane issued at once (n32 <= 640 on this path: at most ten 64-column strides).
// The last panel's trsm is folded in instead of being a kernel of its own: with D the last
// diagonal block, R_i the last 32 columns of identity row i in the working buffer and r those of
// the e_a row, the missing part of the sum is (R_i D^-T)(D^-1 r) = R_i w with
// w = D^-T D^-1 r -- two 32x32 mat-vecs every workgroup does for itself from LDS.
template <int NZ>
__global__ __launch_bounds__(256) void k_cholg_solve(const double *Lw, const double *Lx, int ld, int n, int n32,
                                                    double *x, const double *linv, int *status) {
  __shared__ double sLi[GB][GB + 1], sV[GB], sW[GB];
  const int tid = threadIdx.x, lane = tid & 63, i = blockIdx.x * 4 + (tid >> 6);
  const int jl = n32 - GB;  // first column of the last panel
  const double *Li = linv + (size_t)(jl / GB) * GB * GB;
  stage_block32(sLi, Li, GB, tid);
  const double rk = tid < GB ? Lw[(size_t)n32 * ld + jl + tid] : 0.0;
  // the main part of the row's sum is in flight while w is formed
  const bool row = i < n;
  const double *y = Lx + (size_t)n32 * ld;
  const double *z = Lx + (size_t)(n32 + 16 + (row ? i : 0)) * ld;
  double zv[NZ], yv[NZ];
#pragma unroll
  for (int m = 0; m < NZ; m++) {
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
  for (int m = 0; m < NZ; m++) acc += zv[m] * yv[m];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    x[i] = acc;
    if (!isfinite(acc)) status[1] = status[3];
  }
}

#ifdef PSBA_BUILD_EXPERIMENTS
// ---- the last blocks of the fused chain in ONE kernel (round 4; an experiment that did not pay: built, parity-green,
// PSBA_CHOL_TAIL=n in an experiments build; venice chain 64.0 us without, 65.0 / 63.4 / 64.0 us with n = 2 / 3 / 4.
// In-kernel stamps at n = 4, cycles: staging of S' and of the rows 5600, four factorizations 6400 / 11 000 / 6700 /
// 6600 -- the "background" block updates share the CU's one LDS pipe with the factorization's latency-critical
// waves and stretch it --, the panels and diagonal updates between them 3000 / 2400 / 1600 + 1100 each, the
// backward substitution by one wave 6700: 25.6 us for what three panel launches and the solve kernel do in 24.6.
// Concentrating work on one CU is what the panel launches avoid) -------------------------------------
// A panel launch costs ~3.5 us besides the 32 pivots of its diagonal block (kernel boundary 1.9, first loads of
// what other CUs wrote 1.2, stores 0.5), and towards the end of the chain it has next to nothing to do for the
// rest of the chip.  With the factorization complete up to column jl, the solution only needs the trailing
// Schur complement S' (columns >= jl, all earlier panels applied), the same columns r' of the e_a row and R_i of
// the identity rows:      x_i = sum_{c < jl} z_i[c] y[c]  +  R_i S'^-1 r'
// (k_cholg_solve is the case of one trailing block).  Here the tail is nt <= 4 blocks: every workgroup stages the
// lower block triangle of S' (at most 10 blocks of 32x32) in LDS and factors it by itself -- factor32 per diagonal
// block on waves 0-3, the panel below it and the next diagonal block's update by all eight waves between two
// factorizations (MFMA from LDS), the remaining block updates by waves 4-7 in the shadow of the next
// factorization --, solves S' w = r' block by block with the stored inverses, and then forms its eight rows of
// x (one wave each).  All workgroups do the same factorization: nobody waits for anybody, and nt - 1 panel
// launches and the solve kernel become one launch.
constexpr int TAIL_MAX = 4;
struct TailLds {
  Factor32Lds s;
  double B[TAIL_MAX * (TAIL_MAX + 1) / 2][GB][GB + 1];  // block (r, c), c <= r, at r (r + 1) / 2 + c: S', then its factor
  double Li[TAIL_MAX][GB][GB + 1];                      // inverses of the diagonal blocks of the factor
  double r[TAIL_MAX * GB], q[16][GB];                   // r' -> y' -> w; partial sums of the mat-vecs
};
__device__ __forceinline__ int tail_idx(int r, int c) { return r * (r + 1) / 2 + c; }

// one 16x16 tile of A M^T for two 32x32 blocks in LDS: sum_k A[16 ti + i][k] M[16 tj + j][k]
__device__ __forceinline__ d4 tail_abt(const double (*A)[GB + 1], const double (*M)[GB + 1], int ti, int tj, int li, int lk) {
  d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * ti + li][4 * t + lk], M[16 * tj + li][4 * t + lk], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * ti + li][4 * t + 4 + lk], M[16 * tj + li][4 * t + 4 + lk], c1, 0, 0, 0);
  }
  return c0 + c1;
}

// a 32x32 mat-vec by ONE wave from LDS (no barrier): out[r] = sum_k M[r][k] v[k] (TRANS: sum_k M[k][r] v[k]);
// lane = (row, half): 16 products each, the halves added with one lane swap.  Every lane returns out[lane & 31].
template <bool TRANS>
__device__ __forceinline__ double tail_matvec(const double (*M)[GB + 1], const double *v, int lane) {
  const int r = lane & 31, h = lane >> 5;
  double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 16; k++) a4[k & 3] += (TRANS ? M[16 * h + k][r] : M[r][16 * h + k]) * v[16 * h + k];
  const double a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  return a + __shfl_xor(a, 32, 64);
}

// what the waves beyond the fourth do beside the factorization of diagonal block b + 1 (b = the column whose
// panel has just been formed): waves 5 and 7 (the SIMDs of the inverse wave and of the nearly idle fourth wave --
// a busy wave beside the pivot wave or the tile wave stretched the factorization from 6450 to 9150 cycles) take
// the tiles of the blocks (r, c), b + 1 <= c <= r < nt, (r, c) != (b + 1, b + 1); wave 6 carries the right-hand
// side along: y'_b = L_bb^-1 r'_b, then r'_c -= L_cb y'_b for the blocks below.
struct TailBackground {
  TailLds *t;
  int b, nt, lane;
  __device__ __forceinline__ void operator()(int w) const {
    if (b < 0) return;
    if (w == 2) {
      const double yb = tail_matvec<false>(t->Li[b], t->r + GB * b, lane);
      if (lane < GB) t->r[GB * b + lane] = yb;  // (a wave's LDS operations execute in order: the reads above are done)
      for (int c = b + 1; c < nt; c++) {
        const double u = tail_matvec<false>(t->B[tail_idx(c, b)], t->r + GB * b, lane);
        if (lane < GB) t->r[GB * c + lane] -= u;
      }
      return;
    }
    if (w != 1 && w != 3) return;
    const int me = w >> 1, li = lane & 15, lk = lane >> 4;
    int task = 0;
    for (int r = b + 2; r < nt; r++)
      for (int c = b + 1; c <= r; c++)
        for (int tt = 0; tt < 4; tt++) {
          const int ti = tt >> 1, tj = tt & 1;
          if (r == c && tj > ti) continue;  // upper tile of a diagonal block
          if ((task++ & 1) != me) continue;
          const d4 u = tail_abt(t->B[tail_idx(r, b)], t->B[tail_idx(c, b)], ti, tj, li, lk);
          double(*dst)[GB + 1] = t->B[tail_idx(r, c)];
#pragma unroll
          for (int e = 0; e < 4; e++) dst[16 * ti + lk + 4 * e][16 * tj + li] -= u[e];
        }
  }
};

template <int NZ>
__global__ __launch_bounds__(512) void k_cholg_tail(const double *Lw, const double *Lx, int ld, int n, int n32, int nt,
                                                    double *x, int *status, long long *tim) {
#define TAIL_STAMP(k) \
  if (tim && blockIdx.x == 0 && threadIdx.x == 0) tim[k] = (long long)__builtin_amdgcn_s_memtime()
  TAIL_STAMP(0);
  extern __shared__ __align__(32) unsigned char tail_smem[];
  TailLds &T = *reinterpret_cast<TailLds *>(tail_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = blockIdx.x * 8 + wave;
  const int li = lane & 15, lk = lane >> 4;
  const int jl = n32 - GB * nt, nw = GB * nt;  // first column and width of the tail
  const int rr = tid / GB, cc = tid % GB;     // element (rr, cc) and (rr + 16, cc) of a 32x32 block
  // ---- everything this workgroup reads from memory is requested up front ----
  double sv[TAIL_MAX * (TAIL_MAX + 1) / 2][2];  // S' (lower block triangle), two elements per thread and block
#pragma unroll
  for (int r = 0; r < TAIL_MAX; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) {
      const double *src = Lw + (size_t)(jl + GB * (r < nt ? r : 0) + rr) * ld + jl + GB * (r < nt ? c : 0) + cc;
      sv[r * (r + 1) / 2 + c][0] = src[0];
      sv[r * (r + 1) / 2 + c][1] = src[(size_t)16 * ld];
    }
  const double rk = tid < nw ? Lw[(size_t)n32 * ld + jl + tid] : 0.0;
  // this wave's row of the solution: the part left of the tail, and R_i (generated for the rows of the tail)
  const bool row = i < n;
  const double *y = Lx + (size_t)n32 * ld;
  const double *z = Lx + (size_t)(n32 + 16 + (row ? i : 0)) * ld;
  double zv[NZ], yv[NZ], rt[2];
#pragma unroll
  for (int m = 0; m < NZ; m++) {
    const int c = lane + 64 * m;
    const bool on = row && c < jl && c >= (i & ~15);  // left of the diagonal tile: zeros, never written
    zv[m] = on ? z[c] : 0.0;
    yv[m] = on ? y[c] : 0.0;
  }
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const int k = lane + 64 * m;
    rt[m] = 0.0;
    if (row && k < nw) rt[m] = i >= jl ? (double)(k == i - jl) : Lw[(size_t)(n32 + 16 + i) * ld + jl + k];
  }
#pragma unroll
  for (int r = 0; r < TAIL_MAX; r++)
#pragma unroll
    for (int c = 0; c <= r; c++)
      if (r < nt) {
        double(*dst)[GB + 1] = (r == 0 && c == 0) ? T.s.D : T.B[r * (r + 1) / 2 + c];  // the first diagonal block: where it is factored
        dst[rr][cc] = sv[r * (r + 1) / 2 + c][0];
        dst[rr + 16][cc] = sv[r * (r + 1) / 2 + c][1];
      }
  if (tid < nw) T.r[tid] = rk;
  if (tid == 0) T.s.fail = 0;
  if (tid < 4) T.s.flag[tid] = 0;
  __syncthreads();
  TAIL_STAMP(1);
  // ---- the factorization of S', block column by block column (the diagonal block of the turn is in T.s.D) ----
  for (int b = 0; b < nt; b++) {
    const TailBackground bg = {&T, b - 1, nt, lane};
    TAIL_STAMP(2 + 4 * b);
    factor32<false, TailBackground>(T.s, tid, nullptr, bg);  // (ends with a barrier: the background work is in)
    TAIL_STAMP(3 + 4 * b);
    // L_bb^-1 for the solve and for wave 6; the panel below, L_rb = A_rb L_bb^-T, held in registers until
    // everybody has read its operands (it is formed in place)
    T.Li[b][rr][cc] = f32_Linv(T.s, rr, cc);
    T.Li[b][rr + 16][cc] = f32_Linv(T.s, rr + 16, cc);
    const int ntask = 4 * (nt - 1 - b);
    d4 res[2];
#pragma unroll
    for (int round = 0; round < 2; round++) {
      const int task = wave + 8 * round;
      if (task < ntask) {
        // (the inverse straight from what factor32 left: Z scaled by 1 / sqrt(d) of its row)
        const double(*A)[GB + 1] = T.B[tail_idx(b + 1 + task / 4, b)];
        const int ti = (task >> 1) & 1, tj = task & 1;
        const double sc = T.s.rsq[16 * tj + li];
        d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * ti + li][4 * t + lk], T.s.Li[16 * tj + li][4 * t + lk] * sc, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * ti + li][4 * t + 4 + lk], T.s.Li[16 * tj + li][4 * t + 4 + lk] * sc, c1, 0, 0, 0);
        }
        res[round] = c0 + c1;
      }
    }
    if (b + 1 == nt) break;
    __syncthreads();
#pragma unroll
    for (int round = 0; round < 2; round++) {
      const int task = wave + 8 * round;
      if (task < ntask) {
        double(*dst)[GB + 1] = T.B[tail_idx(b + 1 + task / 4, b)];
        const int ti = (task >> 1) & 1, tj = task & 1;
#pragma unroll
        for (int e = 0; e < 4; e++) dst[16 * ti + lk + 4 * e][16 * tj + li] = res[round][e];
      }
    }
    if (tid < 4) T.s.flag[tid] = 0;
    __syncthreads();
    TAIL_STAMP(4 + 4 * b);
    // the next diagonal block, now, straight into the factorization's buffer: A - L L^T (three tiles); every
    // other update runs beside its factorization
    if (wave < 3) {
      const int ti = wave > 0, tj = wave > 1;
      const d4 u = tail_abt(T.B[tail_idx(b + 1, b)], T.B[tail_idx(b + 1, b)], ti, tj, li, lk);
      const double(*src)[GB + 1] = T.B[tail_idx(b + 1, b + 1)];
#pragma unroll
      for (int e = 0; e < 4; e++) T.s.D[16 * ti + lk + 4 * e][16 * tj + li] = src[16 * ti + lk + 4 * e][16 * tj + li] - u[e];
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0 && blockIdx.x == 0 && T.s.fail) status[1] = status[3];
  TAIL_STAMP(18);
  // ---- S' w = r'.  Forward: wave 6 has carried r' along (y'_b for b < nt - 1 is in place, r'_{nt-1} has every
  // update); the last block and the whole backward substitution by wave 0 alone -- wave-synchronous 32x32
  // mat-vecs from LDS, no barrier (eight rounds of two barriers of 512 threads took 12 000 cycles) ----
  if (wave == 0) {
    const int bl = nt - 1;
    const double yl = tail_matvec<false>(T.Li[bl], T.r + GB * bl, lane);
    if (lane < GB) T.r[GB * bl + lane] = yl;
    for (int b = nt - 1; b >= 0; b--) {
      // w_b = L_bb^-T (y'_b - sum_{c > b} L_cb^T w_c)
      double u = T.r[GB * b + (lane & 31)];
      for (int c = b + 1; c < nt; c++) u -= tail_matvec<true>(T.B[tail_idx(c, b)], T.r + GB * c, lane);
      if (lane < GB) T.r[GB * b + lane] = u;
      const double wv = tail_matvec<true>(T.Li[b], T.r + GB * b, lane);
      if (lane < GB) T.r[GB * b + lane] = wv;
    }
  }
  __syncthreads();
  TAIL_STAMP(19);
  if (!row) return;
  double acc = 0.0;
#pragma unroll
  for (int m = 0; m < 2; m++)
    if (lane + 64 * m < nw) acc += rt[m] * T.r[lane + 64 * m];
#pragma unroll
  for (int m = 0; m < NZ; m++) acc += zv[m] * yv[m];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    x[i] = acc;
    if (!isfinite(acc)) status[1] = status[3];
  }
  TAIL_STAMP(20);
#undef TAIL_STAMP
}

#endif  // PSBA_BUILD_EXPERIMENTS

// the shape of the chain for a matrix size (see enqueue_chain for the measurements behind the limits)
struct ChainShape {
  bool fused, blocked, fused2;
  int NB;
  long long cols_fused_max;
};
static ChainShape chain_shape(const psba_ctx *h) {
  const int n32 = h->n32, nT = n32 / 16 + 1;
  ChainShape c;
  const long long M0 = (nT - 1) - GB / 16;
  long long fused_max = 2200;  // i.e. every n32 <= 1024, the size k_cholg_solve<16> holds; against the mid-size chain 100 cameras 0.331 -> 0.295 ms per LM iteration, 110: 199 -> 169 us per solve, 130: 244 -> 217, 150: 290 -> 265
  if (const char *e = getenv("PSBA_CHOL_FUSED_MAX")) fused_max = atoll(e);
  c.fused = !getenv("PSBA_CHOL_UNFUSED") && M0 * (M0 + 1) / 2 + M0 <= fused_max && n32 <= 1024;
  c.NB = n32 >= 8192 ? 384 : 256;  // super-panel width (multiple of 64).  With the 4x4-tile update: n = 12 000 20.3 / 17.6 / 17.1 / 17.1 ms at 128 / 256 / 384 / 512, n = 6000 4.73 / 4.55 / 4.57 ms per LM iteration at 128 / 256 / 384, n = 3600 2.20 / 2.15 / 2.24 (PSBA_CHOL_NB: development knob)
  if (const char *e = getenv("PSBA_CHOL_NB")) c.NB = atoi(e) >= 64 ? atoi(e) / 32 * 32 : c.NB;
  // (measured, ms per LM iteration flat / two-level with the fused in-block steps: n = 1542 0.867 /
  // 0.884, 2040 1.154 / 1.103, 2400 1.385 / 1.287, 2700 1.636 / 1.469)
  c.blocked = !c.fused && (n32 >= 1792 || getenv("PSBA_CHOL_BLOCKED")) && !getenv("PSBA_CHOL_FLAT");
  // mid sizes: the fused panel kernel without the identity rows (one kernel per panel instead of
  // trsm + update: the redundant X pieces cost less than the second launch while the panel has few
  // tiles; per solve n = 600: 218 against 267 us, 780: 292 / 356, 1200: 502 / 564, 1542: 763 / 759,
  // 2040: 1241 / 1070), backward solve through the factor
  long long f2max = 3600;
  if (const char *e = getenv("PSBA_CHOL_FUSED2_MAX")) f2max = atoll(e);
  c.fused2 = !c.fused && !c.blocked && M0 * (M0 + 1) / 2 <= f2max;
  c.cols_fused_max = 3000;  // sweep 0 / 3000 / 6000 / 9000 tiles: n = 6000 4.77 / 4.56 / 4.72 / 4.69 ms per LM iteration, n = 12 000 17.85 / 17.63 / 18.19 / 18.80 ms per solve
  if (const char *e = getenv("PSBA_CHOL_COLS_FUSED_MAX")) c.cols_fused_max = atoll(e);
  return c;
}

// a pause on a stream: one wave sleeps until `ticks` of the 100 MHz wall clock have passed (look-ahead chain: the far
// update starts a few microseconds after the near one, see enqueue_superpanel)
__global__ void k_cholg_pause(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  for (int i = 0; i < 4096 && (long long)wall_clock64() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(8);
}

// one super-panel [J, J + NB) of the two-level chain: its 32-column steps, which update only the
// super-panel's remaining columns (all rows below), then ONE K = NB update of everything to its
// right.  nranks > 0: that update only for this rank's 64-column blocks (k_cholg_update_wide4); 0: replicated.
// look = true: the K = NB update split in two (see k_cholg_update_wide4): the next super-panel's columns on
// stream s, then -- on the side stream, between the events the caller manages -- everything to their right
static void enqueue_superpanel(psba_ctx *h, hipStream_t s, const ChainShape &c, int J, int nranks, int rank, bool look = false,
                               hipStream_t side = nullptr, hipEvent_t ev_steps = nullptr, hipEvent_t ev_far_prev = nullptr,
                               hipEvent_t ev_far = nullptr) {
  const int n32 = h->n32, ld = h->n32, nT = n32 / 16 + 1;
  double *Lw = h->red, *Lx = h->chol_L, *linv = h->chol_ws;
  const int NB = c.NB;
  const int JE = J + NB < n32 ? J + NB : n32;  // end column of this super-panel
  // round 4: the steps factor the super-panel's own diagonal block only (tile rows < JE / 16), the rows below take
  // ONE block triangular solve (k_cholg_trsm_block).  PSBA_CHOL_STEPS_ALL_ROWS=1: every step on all rows, as before.
  // With the one-wave block solve it paid from ~8000 columns on (n = 12 000: 16.91 -> 16.53 ms per factorization;
  // all rows / diagonal only: n = 2040 835 / 1038 us, 2400 1027 / 1282, 3600 1794 / 2088, 6000 4276 / 4506); with
  // four waves per tile row (k_cholg_trsm_block4) from ~4400: n = 3600 1760 / 1774 us, 4800 2712 / 2628, 6000
  // 3781 / 3413, 7200 5326 / 4766, 8040 6636 / 6025.  PSBA_CHOL_STEPS_DIAG_ONLY=1 forces it (tests).
  const bool steps_all_rows = getenv("PSBA_CHOL_STEPS_ALL_ROWS") != nullptr;
  const bool steps_diag_forced = getenv("PSBA_CHOL_STEPS_DIAG_ONLY") != nullptr;
  const int nbk = (JE - J) / GB;
  const bool diag_only = !steps_all_rows && (n32 >= 4400 || steps_diag_forced) && JE < n32 &&
                         (nbk == 4 || nbk == 8 || nbk == 12 || nbk == 16);
  const int nTs = diag_only ? JE / 16 : nT;  // tile rows the steps see
  for (int j = J; j < JE; j += GB) {
    const int T0 = (j + GB) / 16, TE = JE / 16;
    if (diag_only && j + GB >= JE) break;  // nothing of the diagonal block is left below its last 32 columns
    long long tiles = 0;  // tiles of the super-panel's remaining columns, all rows the steps see
    for (int TC = T0; TC < TE; TC++) tiles += nTs - TC;
    if (j + GB < JE && tiles <= c.cols_fused_max) {
      // trsm + update in one kernel (every wave forms the X pieces of its tile itself: 40 MFMAs
      // per tile instead of 8, which pays while the step has few tiles -- per LM iteration
      // n = 3600: 2.17 against 2.36 ms, 6000: 4.72 / 4.78, 12 000 (all steps): 19.0 / 17.9)
      hipLaunchKernelGGL(k_cholg_panel, dim3(1 + (unsigned)((tiles + 3) / 4)), dim3(256), 0, s, Lw, Lx, ld, j, nTs, 0,
                         TE, linv, h->status, 1);
      continue;
    }
    hipLaunchKernelGGL(k_cholg_trsm, dim3((nTs - T0 + 3) / 4), dim3(256), 0, s, Lw, Lx, ld, j, nTs, nTs, linv);
    if (j + GB < JE)
      hipLaunchKernelGGL(k_cholg_update_cols, dim3(1 + (unsigned)((tiles + 3) / 4)), dim3(256), 0, s, Lw, Lx, ld, j,
                         nTs, TE, linv, h->status);
  }
  if (diag_only && !getenv("PSBA_CHOL_TRSM_WAVE")) {  // the rows below the super-panel (and the e_a row): X = C L_D^-T in one pass
    const dim3 g(nT - JE / 16), b(256);  // four waves per tile row (PSBA_CHOL_TRSM_WAVE=1: one wave, the first form)
    if (nbk == 4) hipLaunchKernelGGL(k_cholg_trsm_block4<4>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
    if (nbk == 8) hipLaunchKernelGGL(k_cholg_trsm_block4<8>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
    if (nbk == 12) hipLaunchKernelGGL(k_cholg_trsm_block4<12>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
    if (nbk == 16) hipLaunchKernelGGL(k_cholg_trsm_block4<16>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
  } else if (diag_only) {
    const dim3 g((nT - JE / 16 + 3) / 4), b(256);
    if (nbk == 4) hipLaunchKernelGGL(k_cholg_trsm_block<4>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
    if (nbk == 8) hipLaunchKernelGGL(k_cholg_trsm_block<8>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
    if (nbk == 12) hipLaunchKernelGGL(k_cholg_trsm_block<12>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
    if (nbk == 16) hipLaunchKernelGGL(k_cholg_trsm_block<16>, g, b, 0, s, Lw, Lx, ld, J, JE / 16, nT, linv);
  }
  if (JE < n32) {
    const int Tw = JE / 16;
    if ((getenv("PSBA_CHOL_WIDE2") || (JE - J) % (2 * GB)) && nranks == 0) {  // the 2x2-tile kernel (any width; for comparison)
      const long long MR = (nT - 1 - Tw) / 2;
      const long long work = MR * (MR + 1) / 2 - 1 + (nT - 1 - Tw);  // macro tiles but the first, e_a tiles
      hipLaunchKernelGGL(k_cholg_update_wide, dim3(1 + (unsigned)((work + 3) / 4)), dim3(256), 0, s, Lw, Lx, ld, J,
                         JE - J, nT, Tw, linv, h->status, 0);
    } else if (look) {
      const int ncolb = NB / 64;
      const long long MR = (nT - 1 - Tw + 3) / 4;
      (void)hipEventRecord(ev_steps, s);
      if (ev_far_prev) (void)hipStreamWaitEvent(s, ev_far_prev, 0);  // its columns carry the previous update
      const long long nnear = MR * ncolb + (4 * ncolb < nT - 1 - Tw ? 4 * ncolb : nT - 1 - Tw);
      // ... in 32x32 pieces once all of them fit on the chip's 1024 SIMDs together (the last quarter of the matrix: one
      // round of 64x64 blocks is 57 us of latency in front of the next super-panel's steps, one of 32x32 pieces ~18)
      const long long MRh = (nT - 1 - Tw) / 2, nnear2 = MRh * (2 * ncolb) - 1 + (4 * ncolb < nT - 1 - Tw ? 4 * ncolb : nT - 1 - Tw);
      long long fine_max = 1000;
      if (const char *e = getenv("PSBA_CHOL_NEAR_FINE_MAX")) fine_max = atoll(e);
      if (nnear2 <= fine_max)
        hipLaunchKernelGGL(k_cholg_update_wide, dim3(1 + (unsigned)((nnear2 + 3) / 4)), dim3(256), 0, s, Lw, Lx, ld, J, JE - J,
                           nT, Tw, linv, h->status, 2 * ncolb);
      else
      hipLaunchKernelGGL(k_cholg_update_wide4, dim3(1 + (unsigned)((nnear + 3) / 4)), dim3(256), 0, s, Lw, Lx, ld, J, JE - J,
                         nT, Tw, linv, h->status, 0, 0, 1, ncolb);
      (void)hipStreamWaitEvent(side, ev_steps, 0);
      // both updates become runnable the moment the previous far update ends; started together, the far update's
      // workgroups take CUs before all of the near update's (a tenth as many) are placed, and the near update -- which
      // the next super-panel's steps wait for -- runs two rounds of 64x64 blocks instead of one (kernel trace: 118
      // against 57 us).  A head start of a few microseconds lets the dispatcher place the near update first.
      const int pause_ticks = getenv("PSBA_CHOL_FAR_PAUSE") ? atoi(getenv("PSBA_CHOL_FAR_PAUSE")) : 300;  // 3 us (sweep 0 / 1 / 100 / 300 / 600 / 1000 / 2000 ticks: 15.45 / 15.17 / 15.03 / 14.95-15.02 / 14.95 / 15.00 / 15.00 ms at n = 12 000)
      if (pause_ticks > 0 && ev_far_prev) hipLaunchKernelGGL(k_cholg_pause, dim3(1), dim3(64), 0, side, (long long)pause_ticks);
      const int Tw2 = Tw + 4 * ncolb;
      if (Tw2 < nT - 1) {
        const long long MR2 = (nT - 1 - Tw2 + 3) / 4;
        const unsigned grid = 1 + (unsigned)((MR2 * (MR2 + 1) / 2 + (nT - 1 - Tw2) + 3) / 4);
        hipLaunchKernelGGL(k_cholg_update_wide4, dim3(grid), dim3(256), 0, side, Lw, Lx, ld, J, JE - J, nT, Tw2, linv,
                           h->status, 0, 0, 2, ncolb);
      }
      (void)hipEventRecord(ev_far, side);
    } else {
      const long long MR = (nT - 1 - Tw + 3) / 4;
      const unsigned grid = 1 + (unsigned)((MR * (MR + 1) / 2 + (nT - 1 - Tw) + 3) / 4);  // diag; 64x64 blocks, e_a tiles
      hipLaunchKernelGGL(k_cholg_update_wide4, dim3(grid), dim3(256), 0, s, Lw, Lx, ld, J, JE - J, nT, Tw, linv,
                         h->status, nranks, rank, 0, 0);
    }
  }
}

// the backward solve of the unfused chains
static void enqueue_backward(psba_ctx *h, hipStream_t s) {
  const int n32 = h->n32, ld = h->n32;
  double *Lx = h->chol_L, *linv = h->chol_ws;
  if (!getenv("PSBA_CHOL_BACK_ONE_WG")) {
    // one small kernel per block, all CUs (see k_cholg_back_panel); against one workgroup walking
    // the factor (k_cholg_backward, kept for comparison): n = 780 356 / 385 us per solve, 1542
    // 760 / 1071, 2040 1071 / 1695
    if (getenv("PSBA_CHOL_BACK_SINGLE")) {  // one block per launch (kept for comparison)
      for (int j = n32 - GB; j >= 0; j -= GB)
        hipLaunchKernelGGL(k_cholg_back_panel, dim3(j / 256 + 1), dim3(256), 0, s, Lx, ld, h->d.nA, n32, j, h->dp,
                           linv, h->status);
    } else {
      for (int R = n32 / GB; R > 0;) {
        const int nb = R < BACK_NB ? R : BACK_NB, j0 = (R - nb) * GB;
        hipLaunchKernelGGL(k_cholg_back_multi, dim3(j0 / 64 + 1), dim3(256), 0, s, Lx, ld, h->d.nA, n32, j0, nb,
                           h->dp, linv, h->status);
        R -= nb;
      }
    }
  } else {
    int thr = (n32 + 63) / 64 * 64;
    if (thr > 512) thr = 512;
    hipLaunchKernelGGL(k_cholg_backward, dim3(1), dim3(thr), 0, s, Lx, ld, h->d.nA, n32, h->dp, linv, h->status);
  }
}

// ---- the sharded factorization's column exchange ----
// the 64-column block mc of the square to the right of the super-panel that ends at column JE: rows
// from its own diagonal to the e_a row (row n32), packed row by row
__global__ __launch_bounds__(256) void k_cholg_cols(double *Lw, int ld, int row0, int nrows, int col0, int ncols, double *buf,
                                                    int set) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)nrows * ncols) return;
  const int r = (int)(t / ncols), cc = (int)(t % ncols);
  double *at = Lw + (size_t)(row0 + r) * ld + col0 + cc;
  if (set)
    *at = buf[t];
  else
    buf[t] = *at;
}

static void enqueue_chain(psba_ctx *h, hipStream_t s, bool skip_diag) {
  const int n32 = h->n32, ld = h->n32, nT = n32 / 16 + 1;  // tile rows incl. the e_a tile
  double *Lw = h->red, *Lx = h->chol_L, *linv = h->chol_ws;
  // fused panel kernel (with the identity rows riding along) while a panel's tiles (one wave
  // each, 5x the MFMA work) still fit on the chip's 1024 SIMDs at once; beyond that the
  // redundant work is no longer free: two kernels per panel and a sequential backward solve
  const ChainShape c = chain_shape(h);
  const bool fused = c.fused, blocked = c.blocked, fused2 = c.fused2;
  // (experiments build, PSBA_CHOL_TAIL=1..4: the last `tail` blocks of the fused chain factored and solved by
  // k_cholg_tail instead of by panel launches + k_cholg_solve; measured, no gain -- see k_cholg_tail)
  int tail = 0;
#ifdef PSBA_BUILD_EXPERIMENTS
  if (const char *e = getenv("PSBA_CHOL_TAIL")) tail = fused && atoi(e) >= 0 && atoi(e) <= TAIL_MAX && atoi(e) <= n32 / GB ? atoi(e) : 0;
#endif
  const int jl = n32 - GB * tail;  // the panel launches stop here
  if (!skip_diag && !(tail > 0 && jl == 0))  // else the S-reduce kernel has factored the first diagonal block already
    hipLaunchKernelGGL(k_cholg_diag, dim3(1), dim3(256), 0, s, Lw, Lx, ld, 0, linv, h->status, h->chol_tim);
  // Look-ahead (round 4; large matrices, single rank): the K = NB update of a super-panel is split -- the next
  // super-panel's columns first, on this stream; everything to their right on a side stream, beside the next
  // super-panel's ~12 small dependent launches (5 of the 17.6 ms of a 12 000 x 12 000 factorization were those
  // launches with the chip otherwise idle).  A far update follows the previous one on its stream and the
  // 32-column steps whose panel it applies (event); the near update waits for the previous far update, which
  // wrote its columns.  PSBA_CHOL_LOOKAHEAD=0 / 1 forces it off / on (default: on; n32 >= 6000 until the end of round 4).
  // (with the pause, the fine near update and the fence-free events it pays wherever the blocked chain runs: per
  // factorization off / on, n = 1860 777 / 731 us, 2520 1070 / 1032, 3000 1333 / 1282, 4200 2204 / 2106, 5100 2938 /
  // 2596, 6000 3949 / 3410, 7200 5515 / 4808)
  bool look = blocked && c.NB % 64 == 0 && !getenv("PSBA_CHOL_WIDE2");
  if (const char *e = getenv("PSBA_CHOL_LOOKAHEAD")) look = blocked && atoi(e) != 0 && c.NB % 64 == 0 && !getenv("PSBA_CHOL_WIDE2");
  if (look) {
    if (!h->chol_side) {
      // the side stream at the LOWEST priority: the next super-panel's small dependent launches on the main stream
      // must get a CU the moment one of the far update's workgroups retires -- at equal priority they waited ~60 us
      // each for a slot (kernel trace) and the main stream, not the far update, set the pace
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (hipStreamCreateWithPriority(&h->chol_side, hipStreamNonBlocking, getenv("PSBA_CHOL_SIDE_PRIO_DEFAULT") ? 0 : least) != hipSuccess)
        look = false;
    }
    const size_t need = 2 * (size_t)(n32 / c.NB + 2);
    while (look && h->chol_events.size() < need) {
      hipEvent_t e;
      // a device-scope release when the event is recorded (hipEventReleaseToDevice): producer and consumer are kernels
      // of this device.  A default event writes the L2 back at system scope on every record (~6 us in front of each
      // near update with a factor's worth of dirty lines, kernel trace).  hipEventDisableSystemFence, no fence at
      // all, was 0.15 ms faster per 12 000 x 12 000 factorization and is documented for timing events only: one LM run
      // in about forty ended on a different cost with it -- not used.  PSBA_CHOL_EVENT_SYSFENCE=1: default events.
      const unsigned flags = getenv("PSBA_CHOL_EVENT_SYSFENCE") ? hipEventDisableTiming : hipEventDisableTiming | hipEventReleaseToDevice;
      if (hipEventCreateWithFlags(&e, flags) != hipSuccess) {
        look = false;
        break;
      }
      h->chol_events.push_back(e);
    }
  }
  {
    hipEvent_t far_prev = nullptr;
    int k = 0;
    for (int J = 0; blocked && J < n32; J += c.NB, k++) {
      if (!look) {
        enqueue_superpanel(h, s, c, J, 0, 0);
        continue;
      }
      const bool has_update = J + c.NB < n32;
      enqueue_superpanel(h, s, c, J, 0, 0, true, h->chol_side, h->chol_events[2 * k], far_prev, h->chol_events[2 * k + 1]);
      if (has_update) far_prev = h->chol_events[2 * k + 1];
    }
    if (far_prev) (void)hipStreamWaitEvent(s, far_prev, 0);  // the backward solve (and whoever comes next) behind the last far update
  }
  for (int j = 0; !blocked && j < n32; j += GB) {
    const bool last = j + GB >= n32;
    const int T0 = (j + GB) / 16;
    const long long M = (nT - 1) - T0;
    if (tail > 0 && j >= jl) break;  // the rest is k_cholg_tail's
    if (last && fused) break;  // the last panel's trsm is part of k_cholg_solve
    if (last || !(fused || fused2)) {
      // 16-row tiles below the panel incl. the e_a tile
      const int nTall = nT;
      hipLaunchKernelGGL(k_cholg_trsm, dim3((nTall - T0 + 3) / 4), dim3(256), 0, s, Lw, Lx, ld, j, nT, nTall, linv);
      if (!last) {
        const int grid = 1 + (int)((M * (M + 1) / 2 + M - 3 + 3) / 4);
        hipLaunchKernelGGL(k_cholg_update, dim3(grid), dim3(256), 0, s, Lw, Lx, ld, j, nT, linv, h->status);
      }
    } else {
      const int nId = fused ? T0 : 0;
      const int grid = 1 + (int)((M * (M + 1) / 2 + M + (long long)nId * M - 3 + 3) / 4);
      hipLaunchKernelGGL(k_cholg_panel, dim3(grid), dim3(256), 0, s, Lw, Lx, ld, j, nT, nId, -1, linv, h->status,
                         tail > 0 && j + GB >= jl ? 0 : 1);
    }
  }
#ifdef PSBA_BUILD_EXPERIMENTS
  if (fused && tail > 0) {
    const size_t lds = sizeof(TailLds);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void *)k_cholg_tail<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)k_cholg_tail<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    if (n32 <= 640)
      hipLaunchKernelGGL(k_cholg_tail<10>, dim3((h->d.nA + 7) / 8), dim3(512), lds, s, Lw, Lx, ld, h->d.nA, n32, tail, h->dp,
                         h->status, h->chol_tim);
    else
      hipLaunchKernelGGL(k_cholg_tail<16>, dim3((h->d.nA + 7) / 8), dim3(512), lds, s, Lw, Lx, ld, h->d.nA, n32, tail, h->dp,
                         h->status, h->chol_tim);
  } else
#endif
  if (fused) {
    if (n32 <= 640)
      hipLaunchKernelGGL(k_cholg_solve<10>, dim3((h->d.nA + 3) / 4), dim3(256), 0, s, Lw, Lx, ld, h->d.nA, n32, h->dp,
                         linv, h->status);
    else
      hipLaunchKernelGGL(k_cholg_solve<16>, dim3((h->d.nA + 3) / 4), dim3(256), 0, s, Lw, Lx, ld, h->d.nA, n32, h->dp,
                         linv, h->status);
  } else {
    enqueue_backward(h, s);
  }
}

// ---- the sharded factorization (large matrices, two-level chain) ----
// Every rank holds the complete S (all-reduce) and factors the super-panels itself; what is shared is
// the bulk of the flops, the K = NB update of everything to the right of a super-panel: a rank
// updates the 64-column blocks it owns (absolute block B, owner B % nranks), and before a super-panel
// is factored the owners of its blocks send them -- rows from the block's diagonal to the e_a row --
// to everybody.  Per super-panel of NB columns at column JE that is NB / 64 messages of
// 8 x 64 x (n32 + 1 - 64 B) bytes (cfg5, NB = 384: six blocks, at most 6.1 MB each; 0.58 GB in all,
// the lower triangle once); the 32-column steps inside a super-panel stay replicated.
// The pieces are exposed one by one (psba_chol_dist_*) so that a host with its own transport -- or
// a test with several handles on one GPU -- can drive them; with a communicator launch_chol_graph
// runs the same sequence with ncclBroadcast.
int chol_dist_shape(psba_ctx *h, int *NB, int *blocked) {
  const ChainShape c = chain_shape(h);
  *NB = c.NB;
  *blocked = c.blocked && c.NB % 64 == 0;
  return PSBA_OK;
}

int chol_dist_begin(psba_ctx *h) {  // the first diagonal block
  if (!h->diag_done)
    hipLaunchKernelGGL(k_cholg_diag, dim3(1), dim3(256), 0, h->stream, h->red, h->chol_L, h->n32, 0, h->chol_ws, h->status,
                       h->chol_tim);
  h->diag_done = false;
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int chol_dist_superpanel(psba_ctx *h, int J) {  // its 32-column steps and this rank's share of the update
  const ChainShape c = chain_shape(h);
  if (J > 0)  // (its columns have just been completed by the exchange)
    hipLaunchKernelGGL(k_cholg_diag, dim3(1), dim3(256), 0, h->stream, h->red, h->chol_L, h->n32, J, h->chol_ws, h->status,
                       h->chol_tim);
  enqueue_superpanel(h, h->stream, c, J, h->nranks, h->rank);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

// block B (columns [64 B, 64 B + 64)): rows 64 B .. n32 (the e_a row) to / from a packed device buffer
int chol_dist_block(psba_ctx *h, int B, double *buf_dev, int set) {
  const int row0 = 64 * B, nrows = h->n32 + 1 - row0, ncols = h->n32 - row0 < 64 ? h->n32 - row0 : 64;
  const long long n = (long long)nrows * ncols;
  hipLaunchKernelGGL(k_cholg_cols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->red, h->n32, row0, nrows,
                     64 * B, ncols, buf_dev, set);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int chol_dist_finish(psba_ctx *h) {
  enqueue_backward(h, h->stream);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

// the column exchange in front of the super-panel that begins at column JE: which 64-column blocks travel, who
// owns each (the rank that updated it: B mod nranks), the slot of the exchange buffer it is packed into and its
// length in doubles (rows from the block's own diagonal to the e_a row, by min(64, n32 - 64 B) columns -- the last
// block is half a block when n32 mod 64 = 32).  One function for the RCCL path below and for the host test that
// replays it for 2, 3 and 8 ranks (tests/test_host.py): out[k] = {B, owner, slot, doubles}; returns the count.
int chol_dist_exchange_plan(int n32, int NB, int nranks, int JE, long long (*out)[4], int cap) {
  if (JE >= n32 || NB <= 0 || nranks <= 0) return 0;
  const int B0 = JE / 64, B1 = ((JE + NB < n32 ? JE + NB : n32) + 63) / 64;
  int k = 0;
  for (int B = B0; B < B1; B++, k++) {
    if (k >= cap) return -1;
    const long long cols = n32 - 64 * B < 64 ? n32 - 64 * B : 64;
    out[k][0] = B;
    out[k][1] = B % nranks;
    out[k][2] = B - B0;
    out[k][3] = (long long)(n32 + 1 - 64 * B) * cols;
  }
  return k;
}

static int chol_dist_comm(psba_ctx *h) {
  const ChainShape c = chain_shape(h);
  const int n32 = h->n32;
  const size_t per = (size_t)(n32 + 1) * 64;  // a slot of the exchange buffer
  if (!h->dist_buf) {
    if (hipMalloc((void **)&h->dist_buf, sizeof(double) * per * (size_t)(c.NB / 64 + 1)) != hipSuccess)
      return fail(h, PSBA_E_HIP, "no memory for the column exchange buffers");
  }
  int rc = chol_dist_begin(h);
  for (int J = 0; J < n32 && rc == PSBA_OK; J += c.NB) {
    rc = chol_dist_superpanel(h, J);
    const int JE = J + c.NB;
    if (JE >= n32 || rc != PSBA_OK) break;
    long long plan[64][4];
    const int nb = chol_dist_exchange_plan(n32, c.NB, h->nranks, JE, plan, 64);
    if (nb < 0 || nb > c.NB / 64 + 1) return fail(h, PSBA_E_INVALID, "column exchange: %d blocks for a super-panel of %d columns", nb, c.NB);
    for (int k = 0; k < nb && rc == PSBA_OK; k++)
      if (plan[k][1] == h->rank) rc = chol_dist_block(h, (int)plan[k][0], h->dist_buf + (size_t)plan[k][2] * per, 0);
    if (ncclGroupStart() != ncclSuccess) return fail(h, PSBA_E_RCCL, "ncclGroupStart failed");
    bool sent = true;
    for (int k = 0; k < nb && sent; k++) {
      double *b = h->dist_buf + (size_t)plan[k][2] * per;
      sent = ncclBroadcast(b, b, (size_t)plan[k][3], ncclDouble, (int)plan[k][1], h->comm, h->stream) == ncclSuccess;
    }
    if (ncclGroupEnd() != ncclSuccess || !sent)  // (the group is closed before any error return)
      return fail(h, PSBA_E_RCCL, "ncclBroadcast failed in the column exchange");
    for (int k = 0; k < nb && rc == PSBA_OK; k++)
      if (plan[k][1] != h->rank) rc = chol_dist_block(h, (int)plan[k][0], h->dist_buf + (size_t)plan[k][2] * per, 1);
  }
  if (rc == PSBA_OK) rc = chol_dist_finish(h);
  return rc;
}

int launch_chol_graph(psba_ctx *h) {
  // (PSBA_CHOL_DIST_FORCE=1: test hook -- a one-rank communicator takes the exchange path too, so that its
  // pack / broadcast / unpack sequence runs on hardware that has a single GPU)
  if (h->comm && (h->nranks > 1 || getenv("PSBA_CHOL_DIST_FORCE")) && !getenv("PSBA_CHOL_REPLICATED")) {
    int NB, blocked;
    chol_dist_shape(h, &NB, &blocked);
    if (blocked) {
      ProfScope ps(h, PSBA_K_CHOLESKY);
      return chol_dist_comm(h);
    }
  }
  const int v = h->diag_done ? 1 : 0;
  h->diag_done = false;
  // The chain is enqueued directly.  Replaying it as a captured hipGraph (PSBA_CHOL_GRAPH=1; the
  // default until late in round 2) is slower at every size on this runtime: per LM iteration with
  // graph / without, 52 cameras 0.222 / 0.197 ms (the chain itself 84 / 75 us), 100: 0.298 / 0.279,
  // 257: 0.862 / 0.812, 340: 1.099 / 1.067 -- the host enqueues the 10 ... 150 small kernels faster
  // than the GPU retires them, and a graph launch costs ~8 us before its first node starts.  (Large
  // matrices never used one: rocprofv3's kernel tracing has been seen to crash inside the launch of
  // a graph with ~1200 kernel nodes.)
  if (h->n32 > 2048 || !getenv("PSBA_CHOL_GRAPH") || getenv("PSBA_CHOL_NO_GRAPH")) {
    ProfScope ps(h, PSBA_K_CHOLESKY);
    enqueue_chain(h, h->stream, v == 1);
    PSBA_HIP(h, hipGetLastError());
    return PSBA_OK;
  }
  if (!h->chol_graph[v] || h->chol_graph_n32[v] != h->n32 || h->chol_graph_red[v] != h->red) {
    if (h->chol_graph[v]) {
      (void)hipGraphExecDestroy(h->chol_graph[v]);
      h->chol_graph[v] = nullptr;
    }
    hipGraph_t g = nullptr;
    PSBA_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    enqueue_chain(h, h->stream, v == 1);
    PSBA_HIP(h, hipStreamEndCapture(h->stream, &g));
    hipError_t e = hipGraphInstantiate(&h->chol_graph[v], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
      h->chol_graph[v] = nullptr;
      return fail(h, PSBA_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    h->chol_graph_n32[v] = h->n32;
    h->chol_graph_red[v] = h->red;
  }
  {
    ProfScope ps(h, PSBA_K_CHOLESKY);
    PSBA_HIP(h, hipGraphLaunch(h->chol_graph[v], h->stream));
  }
  return PSBA_OK;
}

}  // namespace psba
