// kernels_pcg.hip -- block-sparse reduced camera system and its iterative solve.
//
// The reference forms a DENSE S for every problem (CL_files/compute_S.cl:6-78: nA x nA work-items)
// and inverts it (PSBA/cl_spdinv.cpp:18-40).  S_jk is non-zero only where cameras j and k see a
// common point; on real bundle-adjustment data with many cameras that is a small part of the
// nC (nC + 1) / 2 blocks.  With psba_set_solver(PSBA_SOLVER_PCG) only those blocks exist:
//   * storage: the lower block triangle as a list of 6x6 blocks (bs_jk[b] = (j, k), k <= j, values
//     bs_val[b][36] row-major), every diagonal block present; K2's owner route (one lane per block
//     segment, sums in registers) writes straight into it -- no dense S is ever allocated;
//   * solve: conjugate gradients on S x = e_a, preconditioned with the inverses of the diagonal
//     6x6 blocks (block Jacobi), all vectors and scalars on the device; the host reads the residual
//     norm every few iterations only.  A non-positive curvature p.Sp or a diagonal block that is not
//     positive definite reports PSBA_NOT_SPD like the dense factorization does, so the LM loop's
//     damping protocol is unchanged.
#include <cstdlib>

#include "psba_internal.h"

namespace psba {

// slots of the PCG scalar block (doubles, device)
enum { PC_RZ = 0, PC_PQ = 1, PC_RZ_NEW = 2, PC_RR = 3, PC_BB = 4, PC_FAIL = 5, PC_N = 8 };

// diagonal blocks += U_j + mu I (rank 0 adds mu), e_a += g_a, the try's accumulators of K3 zeroed
__global__ __launch_bounds__(256) void k_bsr_finalize(double *val, const int *diag_slot, double *ea, const double *U,
                                                      const double *ga, double mu_add, int nC, double *scal, int *status,
                                                      int try_id) {
  if (blockIdx.x == 0 && threadIdx.x < 4 * SC_NPART) scal[SC_PART + threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 64) status[3] = try_id;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 36 * nC) {
    const int j = t / 36, rc = t % 36;
    double v = val[(size_t)36 * diag_slot[j] + rc] + U[t];
    if (rc % 7 == 0) v += mu_add;
    val[(size_t)36 * diag_slot[j] + rc] = v;
  } else if (t < 42 * nC) {
    const int e = t - 36 * nC;
    ea[e] += ga[e];
  }
}

// inverse of every diagonal block (Cholesky of the 6x6, then L^-T L^-1), one thread per camera.
// The owner route fills only the lower triangle of a diagonal block reliably symmetric; use it.
__global__ __launch_bounds__(64) void k_bsr_diag_inverse(const double *val, const int *diag_slot, double *minv, int nC,
                                                         double *pc, int *status, int try_id) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nC) return;
  const double *a = val + (size_t)36 * diag_slot[j];
  double L[6][6], X[6][6];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    double d = a[7 * c];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < c) d -= L[c][k] * L[c][k];
    if (!(d > 0.0)) bad = true;
    const double s = sqrt(d), is = 1.0 / s;
    L[c][c] = s;
#pragma unroll
    for (int r = 0; r < 6; r++)
      if (r > c) {
        double v = a[6 * r + c];
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (k < c) v -= L[r][k] * L[c][k];
        L[r][c] = v * is;
      }
  }
  // X = L^-1 (lower), then M = X^T X
#pragma unroll
  for (int c = 0; c < 6; c++) {
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r < c) {
        X[r][c] = 0.0;
      } else if (r == c) {
        X[r][c] = 1.0 / L[r][r];
      } else {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (k >= c && k < r) v -= L[r][k] * X[k][c];
        X[r][c] = v / L[r][r];
      }
    }
  }
  double *m = minv + (size_t)36 * j;
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < 6; c++) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++)
        if (k >= r && k >= c) v += X[k][r] * X[k][c];
      m[6 * r + c] = v;
    }
  if (bad) {
    status[1] = try_id;
    pc[PC_FAIL] = 1.0;
  }
}

// y += S x for the blocks of the lower block triangle: block (j, k) gives y_j += B x_k and, off the
// diagonal, y_k += B^T x_j.  Six lanes per block (one output row each), fp64 atomics into y (zeroed
// before).  A diagonal block is read through its lower triangle (see k_bsr_diag_inverse).
__global__ __launch_bounds__(256) void k_bsr_spmv(const double *val, const int2 *jk, long long nb, const double *x,
                                                  double *y) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long b = t / 6;
  const int r = (int)(t % 6);
  if (b >= nb) return;
  const int2 q = jk[b];
  const double *B = val + 36 * (size_t)b;
  const double *xk = x + 6 * (size_t)q.y, *xj = x + 6 * (size_t)q.x;
  double s = 0.0, tt = 0.0;
  if (q.x == q.y) {
#pragma unroll
    for (int c = 0; c < 6; c++) s += (c <= r ? B[6 * r + c] : B[6 * c + r]) * xk[c];
    atomicAdd(&y[6 * (size_t)q.x + r], s);
    return;
  }
#pragma unroll
  for (int c = 0; c < 6; c++) {
    s += B[6 * r + c] * xk[c];
    tt += B[6 * c + r] * xj[c];
  }
  atomicAdd(&y[6 * (size_t)q.x + r], s);
  atomicAdd(&y[6 * (size_t)q.y + r], tt);
}

__device__ __forceinline__ void block_sum_to(double v, double *dst) {
  __shared__ double sRed[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, sRed[0] + sRed[1] + sRed[2] + sRed[3]);
}

// start: x = 0, r = b, z = M^-1 r, p = z, rz = r.z, bb = b.b; one thread per unknown
__global__ __launch_bounds__(256) void k_pcg_start(const double *b, const double *minv, double *x, double *r, double *z,
                                                   double *p, double *q, int n, double *pc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  double rz = 0.0, bb = 0.0;
  if (t < n) {
    const int j = t / 6, rr = t % 6;
    const double *m = minv + 36 * (size_t)j + 6 * rr;
    const double *bj = b + 6 * (size_t)j;
    double zz = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) zz += m[c] * bj[c];
    const double bt = b[t];
    x[t] = 0.0;
    r[t] = bt;
    z[t] = zz;
    p[t] = zz;
    q[t] = 0.0;
    rz = bt * zz;
    bb = bt * bt;
  }
  block_sum_to(rz, pc + PC_RZ);
  __syncthreads();
  block_sum_to(bb, pc + PC_BB);
}

__global__ __launch_bounds__(256) void k_pcg_dot(const double *a, const double *b, int n, double *dst) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  block_sum_to(t < n ? a[t] * b[t] : 0.0, dst);
}

// x += alpha p, r -= alpha q with alpha = rz / pq; rr = r.r; (q is zeroed for the next product)
__global__ __launch_bounds__(256) void k_pcg_step1(double *x, double *r, const double *p, double *q, int n, double *pc,
                                                   int *status, int try_id) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const double pq = pc[PC_PQ], rz = pc[PC_RZ];
  if (!(pq > 0.0)) {  // not positive definite (or a break-down): the LM loop raises mu
    if (t == 0) {
      status[1] = try_id;
      pc[PC_FAIL] = 1.0;
    }
    return;
  }
  const double alpha = rz / pq;
  double rr = 0.0;
  if (t < n) {
    x[t] += alpha * p[t];
    const double rt = r[t] - alpha * q[t];
    r[t] = rt;
    q[t] = 0.0;
    rr = rt * rt;
  }
  block_sum_to(rr, pc + PC_RR);
}

// z = M^-1 r, rz_new = r.z
__global__ __launch_bounds__(256) void k_pcg_precond(const double *r, const double *minv, double *z, int n, double *pc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  double rz = 0.0;
  if (t < n) {
    const int j = t / 6, rr = t % 6;
    const double *m = minv + 36 * (size_t)j + 6 * rr;
    const double *rj = r + 6 * (size_t)j;
    double zz = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) zz += m[c] * rj[c];
    z[t] = zz;
    rz = r[t] * zz;
  }
  block_sum_to(rz, pc + PC_RZ_NEW);
}

// p = z + beta p with beta = rz_new / rz; then the scalars roll over (by block 0, after everybody has read them:
// the roll-over is a kernel of its own below)
__global__ __launch_bounds__(256) void k_pcg_step2(const double *z, double *p, int n, const double *pc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const double beta = pc[PC_RZ_NEW] / pc[PC_RZ];
  p[t] = z[t] + beta * p[t];
}

__global__ void k_pcg_roll(double *pc) {
  pc[PC_RZ] = pc[PC_RZ_NEW];
  pc[PC_RZ_NEW] = 0.0;
  pc[PC_PQ] = 0.0;
  pc[PC_RR] = 0.0;
}

int launch_bsr_finalize(psba_ctx *h, double mu) {
  const int nC = h->d.nC;
  hipLaunchKernelGGL(k_bsr_finalize, dim3((42 * nC + 255) / 256), dim3(256), 0, h->stream, h->bs_val, h->bs_diag, h->bs_ea, h->U,
                     h->ga, h->rank == 0 ? mu : 0.0, nC, h->scal, h->status, h->try_id);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

// S x = e_a by preconditioned conjugate gradients; dpa into dp[0 .. nA)
int launch_pcg_solve(psba_ctx *h) {
  const int n = h->d.nA, nC = h->d.nC, g = (n + 255) / 256;
  hipStream_t s = h->stream;
  double *x = h->dp, *r = h->pcg_vec, *z = r + n, *p = z + n, *q = p + n, *pc = h->pcg_scal;
  ProfScope ps(h, PSBA_K_CHOLESKY);
  PSBA_HIP(h, hipMemsetAsync(pc, 0, sizeof(double) * PC_N, s));
  hipLaunchKernelGGL(k_bsr_diag_inverse, dim3((nC + 63) / 64), dim3(64), 0, s, h->bs_val, h->bs_diag, h->pcg_minv, nC, pc,
                     h->status, h->try_id);
  hipLaunchKernelGGL(k_pcg_start, dim3(g), dim3(256), 0, s, h->bs_ea, h->pcg_minv, x, r, z, p, q, n, pc);
  const unsigned gb = (unsigned)((6 * h->bs_nblk + 255) / 256);
  double hs[PC_N];
  h->pcg_iters = 0;
  h->pcg_relres = 1.0;
  const double tol2 = h->pcg_tol * h->pcg_tol;
  for (int it = 0; it < h->pcg_maxit;) {
    const int burst = it + 8 < h->pcg_maxit ? 8 : h->pcg_maxit - it;
    for (int k = 0; k < burst; k++, it++) {
      hipLaunchKernelGGL(k_bsr_spmv, dim3(gb), dim3(256), 0, s, h->bs_val, h->bs_jk, h->bs_nblk, p, q);
      hipLaunchKernelGGL(k_pcg_dot, dim3(g), dim3(256), 0, s, p, q, n, pc + PC_PQ);
      hipLaunchKernelGGL(k_pcg_step1, dim3(g), dim3(256), 0, s, x, r, p, q, n, pc, h->status, h->try_id);
      if (k == burst - 1)  // the scalars of this burst's last iteration, before they roll over
        PSBA_HIP(h, hipMemcpyAsync(h->pcg_host, pc, sizeof(double) * PC_N, hipMemcpyDeviceToHost, s));
      hipLaunchKernelGGL(k_pcg_precond, dim3(g), dim3(256), 0, s, r, h->pcg_minv, z, n, pc);
      hipLaunchKernelGGL(k_pcg_step2, dim3(g), dim3(256), 0, s, z, p, n, pc);
      hipLaunchKernelGGL(k_pcg_roll, dim3(1), dim3(1), 0, s, pc);
    }
    PSBA_HIP(h, hipGetLastError());
    PSBA_HIP(h, hipStreamSynchronize(s));
    for (int k = 0; k < PC_N; k++) hs[k] = h->pcg_host[k];
    h->pcg_iters = it;
    if (hs[PC_FAIL] != 0.0) break;
    h->pcg_relres = hs[PC_BB] > 0.0 ? sqrt(hs[PC_RR] / hs[PC_BB]) : 0.0;
    if (!(hs[PC_RR] > tol2 * hs[PC_BB])) break;
  }
  return PSBA_OK;
}

}  // namespace psba
