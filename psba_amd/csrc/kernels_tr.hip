// kernels_tr.hip -- the extra operators of the trust-region caller.
//
//   k_jmul      J x per observation (reference CL_files/compute_Jmultiply.cl:6-52, wrapper
//               PSBA/sba_func.cpp:19-75).  The reference fills a dense nP x nC x 2 grid whose
//               unobserved entries stay zero and only ever takes dot products of it
//               (PSBA/trust_region.cpp:125-126,166-176,209-211); here the result is per
//               observation (2 nO values, the same non-zeros in the same order) and the three
//               dot products of two vectors are formed on the fly.  A_ij / B_ij are recomputed
//               from the parameters (the Jacobian is never stored on this path).
//   k_pack_g    g = [g_a ; g_b] as one vector (the host output of compute_g, sba_func.cpp:536-617)
//   k_newp      compute_newp (CL_files/compute_newp.cl:6-26): proposed = current + dp
//   k_cholmod   the modified Cholesky behind the lambda estimate (PSBA/cl_cholmod.cpp:25-201,
//               CL_files/cholmod_blk.cl:87-846) as ONE workgroup walking the block columns --
//               the reference chains ~nA device-enqueued launches of 3x3 work-groups.  It only
//               runs when S is not positive definite at lambda = 0 (trust_region.cpp:341-363).
//   k_cholmod_grid  the same walk by a cooperative grid (one thread per row of the block column,
//               grid-wide barriers where the one workgroup has its own) for large S: every sum is
//               formed by one thread in the same order as in k_cholmod, so the factor is the same
//               bit for bit.
#include <hip/hip_cooperative_groups.h>

#include "camera_model.h"
#include "psba_internal.h"

namespace psba {

struct JmulArgs {
  const double *camconst, *cams, *pts, *impts;
  const int *iidx, *jidx;
  const double *x1, *x2;  // device vectors [nT]; x2 may equal x1
  double *out1;           // [2 nO] J x1 or nullptr
  double *dots;           // [3] accumulators (zeroed before the launch): x1.x1, x1.x2, x2.x2 in J-norm
  int nO, nA;
};

__global__ __launch_bounds__(256) void k_jmul(JmulArgs p) {
  __shared__ double sRed[3][4];
  double d11 = 0.0, d12 = 0.0, d22 = 0.0;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < p.nO; a += gridDim.x * blockDim.x) {
    const int i = p.iidx[a], j = p.jidx[a];
    double cc[9], cam[6], M[3], e[2], A[12], B[6];
#pragma unroll
    for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * j + k];
#pragma unroll
    for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * j + k];
#pragma unroll
    for (int k = 0; k < 3; k++) M[k] = p.pts[3 * (size_t)i + k];
    const double2 m = reinterpret_cast<const double2 *>(p.impts)[a];
    linearize_obs(cc, cc + 5, cam, M, m.x, m.y, e, A, B);
    double r1[2], r2[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {  // compute_Jmultiply.cl:32-46: row k of A_ij, then of B_ij
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        s1 += A[6 * k + c] * p.x1[6 * j + c];
        s2 += A[6 * k + c] * p.x2[6 * j + c];
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        s1 += B[3 * k + c] * p.x1[p.nA + 3 * (size_t)i + c];
        s2 += B[3 * k + c] * p.x2[p.nA + 3 * (size_t)i + c];
      }
      r1[k] = s1;
      r2[k] = s2;
    }
    if (p.out1) {
      p.out1[2 * (size_t)a] = r1[0];
      p.out1[2 * (size_t)a + 1] = r1[1];
    }
    d11 += r1[0] * r1[0] + r1[1] * r1[1];
    d12 += r1[0] * r2[0] + r1[1] * r2[1];
    d22 += r2[0] * r2[0] + r2[1] * r2[1];
  }
  double v[3] = {d11, d12, d22};
#pragma unroll
  for (int q = 0; q < 3; q++) {
    double t = v[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) sRed[q][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    atomicAdd(&p.dots[threadIdx.x], sRed[threadIdx.x][0] + sRed[threadIdx.x][1] + sRed[threadIdx.x][2] + sRed[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void k_pack_g(const double *ga, const double *PV, int nA, int nP, double *g) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (size_t)nA)
    g[t] = ga[t];
  else if (t < (size_t)nA + 3 * (size_t)nP) {
    const size_t u = t - nA;
    g[t] = PV[9 * (u / 3) + 6 + u % 3];
  }
}

__global__ __launch_bounds__(256) void k_newp(const double *cams, const double *pts, const double *dp, int nA, int nB,
                                              double *newcams, double *newpts) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (size_t)nA)
    newcams[t] = cams[t] + dp[t];
  else if (t < (size_t)nA + nB)
    newpts[t - nA] = pts[t - nA] + dp[t];
}

// ---- modified Cholesky -------------------------------------------------------------------
// M: n x n working copy of S (row stride ld), overwritten by L; aux: 5 n doubles of scratch
// (C_ij of the current column | backup of the current block column (3 n) | original diagonal);
// out[0] = lambda = |sum_i E_i| / n, out[1] = delta, out[2] = beta, out[3] = block columns
// that took the one-column route.
constexpr int CM_THREADS = 1024;

__device__ __forceinline__ double wg_max(double v, double *sRed) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = 0.0;
  for (int w = 0; w < CM_THREADS / 64; w++) m = fmax(m, sRed[w]);
  return m;
}
__device__ __forceinline__ double wg_sum(double v, double *sRed) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < CM_THREADS / 64; w++) s += sRed[w];
  return s;
}

__global__ __launch_bounds__(CM_THREADS) void k_cholmod(double *M, int ld, int n, double *aux, double *out) {
  __shared__ double sRed[CM_THREADS / 64];
  __shared__ double sT[9], sL[6];
  __shared__ int sFail, sOver;
  const int tid = threadIdx.x;
  double *C = aux, *bak = aux + n, *dg = aux + 4 * (size_t)n;
  // delta, beta (cl_cholmod.cpp:109-168, cholmod_blk.cl:796-825)
  double xi = 0.0, gamma = 0.0;
  for (int r = tid; r < n; r += CM_THREADS) {
    const double *row = M + (size_t)r * ld;
    for (int c = 0; c < n; c++) {
      const double v = fabs(row[c]);
      if (c == r)
        gamma = fmax(gamma, v);
      else
        xi = fmax(xi, v);
    }
    dg[r] = row[r];
  }
  xi = wg_max(xi, sRed);
  gamma = wg_max(gamma, sRed);
  const double delta = 1e-15 * fmax(xi + gamma, 1.0);
  const double beta = sqrt(fmax(fmax(gamma, 1e-15), xi / sqrt((double)n * n - 1.0)));
  int single = 0;
  for (int J = 0; J + 3 <= n; J += 3) {
    // backup of the block column, T_JJ = A_JJ - sum_k L_Jk L_Jk^T (cholmod_blk.cl:104-131)
    for (int r = J + tid; r < n; r += CM_THREADS)
#pragma unroll
      for (int c = 0; c < 3; c++) bak[3 * (size_t)r + c] = M[(size_t)r * ld + J + c];
    if (tid < 9) {
      const int u = tid / 3, v = tid % 3;
      double t = M[(size_t)(J + u) * ld + J + v];
      for (int k = 0; k < J; k++) t -= M[(size_t)(J + u) * ld + k] * M[(size_t)(J + v) * ld + k];
      sT[tid] = t;
    }
    if (tid == 0) sOver = 0;
    __syncthreads();
    if (tid == 0) {  // L_JJ L_JJ^T = T_JJ (:134-205)
      int fail = 0;
      double l00 = sT[0], l10 = 0, l11 = 0, l20 = 0, l21 = 0, l22 = 0;
      if (!isfinite(l00) || l00 <= 0) fail = 1;
      if (!fail) {
        l00 = sqrt(l00);
        l10 = sT[3] / l00;
        l20 = sT[6] / l00;
        l11 = sT[4] - l10 * l10;
        if (!isfinite(l11) || l11 <= 0) fail = 1;
      }
      if (!fail) {
        l11 = sqrt(l11);
        l21 = (sT[7] - l20 * l10) / l11;
        l22 = sT[8] - l20 * l20 - l21 * l21;
        if (!isfinite(l22) || l22 <= 0 || !isfinite(l10) || !isfinite(l20) || !isfinite(l21)) fail = 1;
      }
      if (!fail) {
        l22 = sqrt(l22);
        sL[0] = l00; sL[1] = l10; sL[2] = l11; sL[3] = l20; sL[4] = l21; sL[5] = l22;
      }
      sFail = fail;
    }
    __syncthreads();
    if (!sFail) {
      // L_iJ = T_iJ L_JJ^-T for the rows below; an entry above beta (compared without fabs,
      // :352-354) sends the block column to the one-column route
      const double l00 = sL[0], l10 = sL[1], l11 = sL[2], l20 = sL[3], l21 = sL[4], l22 = sL[5];
      int over = 0;
      for (int i = J + 3 + tid; i < n; i += CM_THREADS) {
        double t0 = M[(size_t)i * ld + J], t1 = M[(size_t)i * ld + J + 1], t2 = M[(size_t)i * ld + J + 2];
        const double *Li = M + (size_t)i * ld, *L0 = M + (size_t)J * ld, *L1 = L0 + ld, *L2 = L1 + ld;
        for (int k = 0; k < J; k++) {
          const double lik = Li[k];
          t0 -= lik * L0[k];
          t1 -= lik * L1[k];
          t2 -= lik * L2[k];
        }
        const double x0 = t0 / l00;
        const double x1 = (t1 - x0 * l10) / l11;
        const double x2 = (t2 - x0 * l20 - x1 * l21) / l22;
        M[(size_t)i * ld + J] = x0;
        M[(size_t)i * ld + J + 1] = x1;
        M[(size_t)i * ld + J + 2] = x2;
        if (x0 > beta || x1 > beta || x2 > beta) over = 1;
      }
      if (over) sOver = 1;
      __syncthreads();
      if (!sOver) {
        if (tid == 0) {
          double *d0 = M + (size_t)J * ld + J;
          d0[0] = l00; d0[1] = 0; d0[2] = 0;
          d0[ld] = l10; d0[ld + 1] = l11; d0[ld + 2] = 0;
          d0[2 * (size_t)ld] = l20; d0[2 * (size_t)ld + 1] = l21; d0[2 * (size_t)ld + 2] = l22;
        }
        for (int i = J + 3 + tid; i < n; i += CM_THREADS)
#pragma unroll
          for (int v = 0; v < 3; v++) M[(size_t)(J + v) * ld + i] = 0.0;
        __syncthreads();
        continue;
      }
    }
    // restore the block column and take its columns one at a time (:240-262, :436-462,
    // steps :560-706)
    single++;
    __syncthreads();
    for (int r = J + tid; r < n; r += CM_THREADS)
#pragma unroll
      for (int c = 0; c < 3; c++) M[(size_t)r * ld + J + c] = bak[3 * (size_t)r + c];
    __syncthreads();
    for (int c = 0; c < 3; c++) {
      const int j = J + c;
      if (tid == 0) {
        double d = M[(size_t)j * ld + j];
        for (int k = 0; k < j; k++) d -= M[(size_t)j * ld + k] * M[(size_t)j * ld + k];
        d = fmax(fabs(d), delta);
        sL[0] = sqrt(d);
        sOver = 0;
      }
      __syncthreads();
      double ljj = sL[0];
      int over = 0;
      double theta = 0.0;
      for (int i = j + 1 + tid; i < n; i += CM_THREADS) {
        double cij = M[(size_t)i * ld + j];
        for (int k = 0; k < j; k++) cij -= M[(size_t)i * ld + k] * M[(size_t)j * ld + k];
        C[i] = cij;
        const double l = cij / ljj;
        M[(size_t)i * ld + j] = l;
        M[(size_t)j * ld + i] = 0.0;
        if (l > beta) over = 1;  // :641 compares without fabs
        theta = fmax(theta, fabs(cij));
      }
      if (over) sOver = 1;
      theta = wg_max(theta, sRed);  // (also the barrier behind the sOver store)
      if (sOver) {
        ljj = theta / beta;  // :673-674
        for (int i = j + 1 + tid; i < n; i += CM_THREADS) M[(size_t)i * ld + j] = C[i] / ljj;
      }
      if (tid == 0) M[(size_t)j * ld + j] = ljj;
      __syncthreads();
    }
  }
  // E_i = sum_k L_ik^2 - A_ii (kern_cholmod_E, :830-846); lambda = |sum E| / n (trust_region.cpp:355-362)
  double e = 0.0;
  for (int i = tid; i < n; i += CM_THREADS) {
    double s = 0.0;
    for (int k = 0; k <= i; k++) s += M[(size_t)i * ld + k] * M[(size_t)i * ld + k];
    e += s - dg[i];
  }
  e = wg_sum(e, sRed);
  if (tid == 0) {
    out[0] = fabs(e) / n;
    out[1] = delta;
    out[2] = beta;
    out[3] = (double)single;
  }
}

// The same algorithm on a cooperative grid.  What the one workgroup shares through LDS is here
// either recomputed by every thread from a nine-double block in global memory (the 3 x 3 factor:
// same arithmetic, same result everywhere, no broadcast) or kept in per-step slots that are zeroed
// once before the launch and never reset (the "above beta" flags and the column maxima: a slot
// per block column / column, so no workgroup can read a slot that another already reuses).
// gs: [0..9) spare | ov[nJ] ints | ov1[n] ints | th[n] u64 | part[gridDim.x] doubles (host layout below).
constexpr int CMG_THREADS = 64;  // one wave per workgroup: up to 256 workgroups, one per CU
constexpr int CMG_TILE = 1024;    // entries of the three L_J rows per LDS tile (24 KiB)

__global__ __launch_bounds__(CMG_THREADS) void k_cholmod_grid(double *M, int ld, int n, double *aux, double *out,
                                                              double *gs) {
  namespace cg = cooperative_groups;
  cg::grid_group grid = cg::this_grid();
  __shared__ double sRed[CMG_THREADS / 64];
  __shared__ double sLt[3][CMG_TILE];
  const int tid = threadIdx.x;
  const int gtid = blockIdx.x * CMG_THREADS + tid, GT = gridDim.x * CMG_THREADS;
  const int nJ = n / 3;
  double *C = aux, *bak = aux + n, *dg = aux + 4 * (size_t)n;
  int *ov = reinterpret_cast<int *>(gs + 10);                       // [nJ + 1]
  int *ov1 = ov + (nJ + 2);                                          // [n]
  unsigned long long *th = reinterpret_cast<unsigned long long *>(gs + 10 + (nJ + 2 + n + 1) / 2 + 1);  // [n + 2]
  double *part = reinterpret_cast<double *>(th + n + 2);            // [gridDim.x]
  auto wave_max = [](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
  };
  auto bits = [](double v) { return (unsigned long long)__double_as_longlong(v); };
  auto peek = [](unsigned long long *p) { return __longlong_as_double((long long)*(volatile unsigned long long *)p); };
  // delta, beta (cl_cholmod.cpp:109-168, cholmod_blk.cl:796-825); th[n], th[n + 1] hold xi, gamma
  {
    double xi = 0.0, gamma = 0.0;
    for (int r = gtid; r < n; r += GT) {
      const double *row = M + (size_t)r * ld;
      for (int c = 0; c < n; c++) {
        const double v = fabs(row[c]);
        if (c == r)
          gamma = fmax(gamma, v);
        else
          xi = fmax(xi, v);
      }
      dg[r] = row[r];
    }
    xi = wave_max(xi);
    gamma = wave_max(gamma);
    if ((tid & 63) == 0) {
      atomicMax(th + n, bits(xi));
      atomicMax(th + n + 1, bits(gamma));
    }
  }
  grid.sync();
  const double xi = peek(th + n), gamma = peek(th + n + 1);
  const double delta = 1e-15 * fmax(xi + gamma, 1.0);
  const double beta = sqrt(fmax(fmax(gamma, 1e-15), xi / sqrt((double)n * n - 1.0)));
  int single = 0;
  for (int J = 0; J + 3 <= n; J += 3) {
    // T_iJ = A_iJ - sum_k L_ik L_Jk^T for every row from J on -- the rows J .. J + 2 give T_JJ (cholmod_blk.cl:104-131),
    // so no workgroup forms it alone while the others wait.  One thread per row, k ascending (the order of
    // k_cholmod); the three rows of L_J come through the LDS in tiles, the thread's own row streams from memory
    // two entries per load.  The column's original goes to bak, T into its place.
    for (int base = J; base < n; base += GT) {
      if (base + (int)blockIdx.x * CMG_THREADS >= n) break;  // (uniform over the workgroup)
      const int i = base + gtid;
      const bool live = i < n;
      const double *Li = M + (size_t)(live ? i : J) * ld;
      double t0 = Li[J], t1 = Li[J + 1], t2 = Li[J + 2];
      if (live) {
        bak[3 * (size_t)i] = t0;
        bak[3 * (size_t)i + 1] = t1;
        bak[3 * (size_t)i + 2] = t2;
      }
      for (int k0 = 0; k0 < J; k0 += CMG_TILE) {
        const int len = J - k0 < CMG_TILE ? J - k0 : CMG_TILE;
        __syncthreads();
        for (int q = tid; q < 3 * len; q += CMG_THREADS) {
          const int v = q / len, kk = q - v * len;
          sLt[v][kk] = M[(size_t)(J + v) * ld + k0 + kk];
        }
        __syncthreads();
        if (live) {
          const double *row = Li + k0;
          int kk = 0;
          // four cache lines of the row per turn, all loads issued before the first product: a lane has one
          // line in flight otherwise and the step is a chain of memory latencies (2.1 s at n = 12 000; 1.4 s
          // so; with the next four lines requested before this turn's products -- two register buffers -- 2.9 s)
          for (; kk + 64 <= len; kk += 64) {
            double2 l[32];
#pragma unroll
            for (int u = 0; u < 32; u++) l[u] = *reinterpret_cast<const double2 *>(row + kk + 2 * u);
#pragma unroll
            for (int u = 0; u < 32; u++) {
              t0 -= l[u].x * sLt[0][kk + 2 * u];
              t1 -= l[u].x * sLt[1][kk + 2 * u];
              t2 -= l[u].x * sLt[2][kk + 2 * u];
              t0 -= l[u].y * sLt[0][kk + 2 * u + 1];
              t1 -= l[u].y * sLt[1][kk + 2 * u + 1];
              t2 -= l[u].y * sLt[2][kk + 2 * u + 1];
            }
          }
          for (; kk + 2 <= len; kk += 2) {
            const double2 l = *reinterpret_cast<const double2 *>(row + kk);
            t0 -= l.x * sLt[0][kk];
            t1 -= l.x * sLt[1][kk];
            t2 -= l.x * sLt[2][kk];
            t0 -= l.y * sLt[0][kk + 1];
            t1 -= l.y * sLt[1][kk + 1];
            t2 -= l.y * sLt[2][kk + 1];
          }
          if (kk < len) {
            const double lik = row[kk];
            t0 -= lik * sLt[0][kk];
            t1 -= lik * sLt[1][kk];
            t2 -= lik * sLt[2][kk];
          }
        }
      }
      if (live) {
        double *o = M + (size_t)i * ld + J;
        o[0] = t0;
        o[1] = t1;
        o[2] = t2;
      }
    }
    grid.sync();
    // L_JJ L_JJ^T = T_JJ (:134-205), by every thread
    int fail = 0;
    const volatile double *T = M + (size_t)J * ld + J;
    double l00 = T[0], l10 = 0, l11 = 0, l20 = 0, l21 = 0, l22 = 0;
    {
      const double t3 = T[ld], t4 = T[ld + 1], t6 = T[2 * (size_t)ld], t7 = T[2 * (size_t)ld + 1], t8 = T[2 * (size_t)ld + 2];
      if (!isfinite(l00) || l00 <= 0) fail = 1;
      if (!fail) {
        l00 = sqrt(l00);
        l10 = t3 / l00;
        l20 = t6 / l00;
        l11 = t4 - l10 * l10;
        if (!isfinite(l11) || l11 <= 0) fail = 1;
      }
      if (!fail) {
        l11 = sqrt(l11);
        l21 = (t7 - l20 * l10) / l11;
        l22 = t8 - l20 * l20 - l21 * l21;
        if (!isfinite(l22) || l22 <= 0 || !isfinite(l10) || !isfinite(l20) || !isfinite(l21)) fail = 1;
      }
      if (!fail) l22 = sqrt(l22);
    }
    if (!fail) {  // (uniform over the grid: every thread factored the same nine numbers)
      int over = 0;
      for (int i = J + 3 + gtid; i < n; i += GT) {
        double *o = M + (size_t)i * ld + J;
        const double t0 = o[0], t1 = o[1], t2 = o[2];
        const double x0 = t0 / l00;
        const double x1 = (t1 - x0 * l10) / l11;
        const double x2 = (t2 - x0 * l20 - x1 * l21) / l22;
        o[0] = x0;
        o[1] = x1;
        o[2] = x2;
        if (x0 > beta || x1 > beta || x2 > beta) over = 1;
      }
      if (over) atomicOr(ov + J / 3, 1);
      grid.sync();
      if (!*(volatile int *)(ov + J / 3)) {
        if (gtid == 0) {
          double *d0 = M + (size_t)J * ld + J;
          d0[0] = l00; d0[1] = 0; d0[2] = 0;
          d0[ld] = l10; d0[ld + 1] = l11; d0[ld + 2] = 0;
          d0[2 * (size_t)ld] = l20; d0[2 * (size_t)ld + 1] = l21; d0[2 * (size_t)ld + 2] = l22;
        }
        for (int i = J + 3 + gtid; i < n; i += GT)
#pragma unroll
          for (int v = 0; v < 3; v++) M[(size_t)(J + v) * ld + i] = 0.0;
        continue;  // (what this step wrote last is read by no later step, only by the sums of E behind a barrier)
      }
    }
    // restore the block column and take its columns one at a time (:240-262, :436-462, steps :560-706)
    single++;
    for (int r = J + gtid; r < n; r += GT)
#pragma unroll
      for (int c = 0; c < 3; c++) M[(size_t)r * ld + J + c] = bak[3 * (size_t)r + c];
    grid.sync();
    for (int c = 0; c < 3; c++) {
      const int j = J + c;
      double d = M[(size_t)j * ld + j];  // (by every thread: the same sum everywhere)
      for (int k = 0; k < j; k++) d -= M[(size_t)j * ld + k] * M[(size_t)j * ld + k];
      d = fmax(fabs(d), delta);
      double ljj = sqrt(d);
      int over = 0;
      double theta = 0.0;
      for (int i = j + 1 + gtid; i < n; i += GT) {
        double cij = M[(size_t)i * ld + j];
        for (int k = 0; k < j; k++) cij -= M[(size_t)i * ld + k] * M[(size_t)j * ld + k];
        C[i] = cij;
        const double l = cij / ljj;
        M[(size_t)i * ld + j] = l;
        M[(size_t)j * ld + i] = 0.0;
        if (l > beta) over = 1;  // :641 compares without fabs
        theta = fmax(theta, fabs(cij));
      }
      if (over) atomicOr(ov1 + j, 1);
      theta = wave_max(theta);
      if ((tid & 63) == 0 && theta > 0.0) atomicMax(th + j, bits(theta));
      grid.sync();
      if (*(volatile int *)(ov1 + j)) {
        ljj = peek(th + j) / beta;  // :673-674
        for (int i = j + 1 + gtid; i < n; i += GT) M[(size_t)i * ld + j] = C[i] / ljj;
      }
      if (gtid == 0) M[(size_t)j * ld + j] = ljj;
      grid.sync();
    }
  }
  grid.sync();
  // E_i = sum_k L_ik^2 - A_ii (kern_cholmod_E, :830-846); lambda = |sum E| / n (trust_region.cpp:355-362)
  double e = 0.0;
  for (int i = gtid; i < n; i += GT) {
    double s = 0.0;
    for (int k = 0; k <= i; k++) s += M[(size_t)i * ld + k] * M[(size_t)i * ld + k];
    e += s - dg[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
  if ((tid & 63) == 0) sRed[tid >> 6] = e;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < CMG_THREADS / 64; w++) s += sRed[w];
    part[blockIdx.x] = s;
  }
  grid.sync();
  if (gtid == 0) {
    double s = 0.0;
    for (unsigned b = 0; b < gridDim.x; b++) s += ((volatile double *)part)[b];
    out[0] = fabs(s) / n;
    out[1] = delta;
    out[2] = beta;
    out[3] = (double)single;
  }
}

// ---- launchers ------------------------------------------------------------------------------
int launch_jmul(psba_ctx *h, const double *x1_dev, const double *x2_dev, double *out1_dev, double *dots_dev) {
  JmulArgs a;
  a.camconst = h->camconst;
  a.cams = h->cams[h->cur];
  a.pts = h->pts[h->cur];
  a.impts = h->impts;
  a.iidx = h->iidx;
  a.jidx = h->jidx;
  a.x1 = x1_dev;
  a.x2 = x2_dev ? x2_dev : x1_dev;
  a.out1 = out1_dev;
  a.dots = dots_dev;
  a.nO = h->d.nO;
  a.nA = h->d.nA;
  PSBA_HIP(h, hipMemsetAsync(dots_dev, 0, 3 * sizeof(double), h->stream));
  int grid = (h->d.nO + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(k_jmul, dim3(grid), dim3(256), 0, h->stream, a);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_pack_g(psba_ctx *h, double *g_dev) {
  const size_t n = (size_t)h->d.nT;
  hipLaunchKernelGGL(k_pack_g, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->ga, h->PV, h->d.nA, h->d.nP,
                     g_dev);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_newp(psba_ctx *h, const double *dp_dev) {
  const size_t n = (size_t)h->d.nT;
  hipLaunchKernelGGL(k_newp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->cams[h->cur], h->pts[h->cur],
                     dp_dev, h->d.nA, h->d.nB, h->cams[1 - h->cur], h->pts[1 - h->cur]);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

// S (at the damping it was assembled with) sits in h->red; the modified Cholesky runs on a copy in
// the factor buffer (chol_L: (2 n32 + 16) x n32 doubles, free between solves), scratch behind it
int launch_cholmod(psba_ctx *h, double *out4_dev) {
  const int n = h->d.nA, ld = h->n32;
  double *M = h->chol_L, *aux = h->chol_L + (size_t)(h->n32 + 1) * ld;  // 5 n <= (n32 + 15) * n32 doubles of room
  if ((size_t)5 * n > (size_t)(h->n32 + 15) * ld) return fail(h, PSBA_E_INVALID, "cholmod scratch too small");
  PSBA_HIP(h, hipMemcpyAsync(M, h->red, sizeof(double) * (size_t)n * ld, hipMemcpyDeviceToDevice, h->stream));
  // the cooperative grid from 50 cameras on (one workgroup against the grid, assembly included: 4.1 / 2.0 ms at
  // n = 324, 16.7 / 5.1 at 600, 98 / 15.9 at 1200, 323 / 35 at 1800, minutes / 1.4 s at 12 000; not measured
  // below); PSBA_CHOLMOD_GRID=1 / 0 forces either (the tests run one matrix through both and compare)
  const char *force = getenv("PSBA_CHOLMOD_GRID");
  const bool use_grid = force ? atoi(force) != 0 : n >= 300;
  if (use_grid) {
    int nwg = (n + CMG_THREADS - 1) / CMG_THREADS;
    if (nwg > 256) nwg = 256;  // one wave per CU at most: co-resident by a wide margin
    const int nJ = n / 3;
    double *gs = aux + 5 * (size_t)n;
    const size_t gs_doubles = 10 + (size_t)(nJ + 2 + n + 1) / 2 + 1 + (n + 2) + nwg;
    if ((size_t)5 * n + gs_doubles > (size_t)(h->n32 + 15) * ld) return fail(h, PSBA_E_INVALID, "cholmod scratch too small");
    PSBA_HIP(h, hipMemsetAsync(gs, 0, sizeof(double) * gs_doubles, h->stream));
    int ld_ = ld, n_ = n;
    void *args[] = {&M, &ld_, &n_, &aux, &out4_dev, &gs};
    PSBA_HIP(h, hipLaunchCooperativeKernel((const void *)k_cholmod_grid, dim3(nwg), dim3(CMG_THREADS), args, 0, h->stream));
    return PSBA_OK;
  }
  hipLaunchKernelGGL(k_cholmod, dim3(1), dim3(CM_THREADS), 0, h->stream, M, ld, n, aux, out4_dev);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
