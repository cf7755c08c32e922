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
#include <vector>

#include "psba_internal.h"

namespace psba {

// slots of the PCG scalar block (doubles, device)
// p.Sp, r.r and r.z of iteration `it` live in slot it % 4 (r.z of the next iteration accumulates in slot
// (it + 1) % 4); the last kernel of an iteration zeroes the slots two iterations ahead, so no slot is
// ever zeroed while a kernel in flight reads or adds to it, and nothing has to be rolled over
// (p.Sp is added up by thousands of workgroups: it goes into 64 partial sums per slot, PC_PQ + 64 slot + k,
// because same-address atomics serialise at ~13 ns each -- 1800 of them were 24 us per iteration)
// PC_DONE / PC_RRFIN: iterations done and r.r when the residual test first held (written once, by one
// thread; the decision itself is taken by every thread from sums that are final before the kernel
// starts, so all workgroups of a launch agree without a hand-shake)
enum { PC_BB = 0, PC_FAIL = 1, PC_DONE = 2, PC_RRFIN = 3, PC_RR = 4, PC_RZ = 8, PC_N = 16, PC_PQ = 16, PC_NPQ = 64,
       PC_ALL = 16 + 4 * 64 };

// has the solve converged before iteration `it`?  r.z of this iteration and r.r of the previous one
// are complete when iteration `it`'s kernels run.  r.z == 0 means r == 0 (M is positive definite):
// e_a == 0 or an exact solution -- a solved system, not a break-down.  Once true it stays true: the
// kernels then leave x, r, p and the sums alone, so the later slots stay zero.
__device__ __forceinline__ bool pcg_converged_before(const double *pc, int it, double tol2) {
  const double rz = pc[PC_RZ + (it & 3)];
  if (rz == 0.0) return true;
  return it > 0 && pc[PC_RR + ((it - 1) & 3)] <= tol2 * pc[PC_BB];
}

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

__device__ __forceinline__ void block_sum_to(double v, double *dst) {
  __shared__ double sRed[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, sRed[0] + sRed[1] + sRed[2] + sRed[3]);
  __syncthreads();
}

// q = S p and p.q with it, without atomics into q: one wave per block row j walks the row of the full
// symmetric pattern (bs_rowent: a stored block of the lower triangle serves its row as stored and its
// column's row transposed; a diagonal block is read through its lower triangle, see
// k_bsr_diag_inverse).  Lane = 6 g + r: eight entries at a time, one output row each; the eight partial
// sums of a row are added up by shuffles.  (The first version walked the stored blocks, six lanes each,
// with two fp64 atomics into q per lane: 2.2 M scattered atomics per product at 186 k blocks.)
__global__ __launch_bounds__(256) void k_pcg_spmv(const double *val, const int *rowptr, const int2 *rowent, int nC,
                                                  const double *p, double *q, double *pc, int it) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  double acc = 0.0;
  const int g = lane / 6, r = lane % 6;
  if (j < nC && g < 8) {
    const int e1 = rowptr[j + 1];
    for (int e = rowptr[j] + g; e < e1; e += 8) {
      const int2 en = rowent[e];
      const int how = en.y >> 28, o = en.y & 0x0fffffff;
      const double *B = val + 36 * (size_t)en.x;
      const double *xo = p + 6 * (size_t)o;
      double s = 0.0;
      if (how == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) s += B[6 * r + c] * xo[c];
      } else if (how == 1) {
#pragma unroll
        for (int c = 0; c < 6; c++) s += B[6 * c + r] * xo[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; c++) s += (c <= r ? B[6 * r + c] : B[6 * c + r]) * xo[c];
      }
      acc += s;
    }
  }
  acc += __shfl_down(acc, 24, 64);
  acc += __shfl_down(acc, 12, 64);
  acc += __shfl_down(acc, 6, 64);
  double pq = 0.0;
  if (j < nC && lane < 6) {
    q[6 * (size_t)j + lane] = acc;
    pq = p[6 * (size_t)j + lane] * acc;
  }
  block_sum_to(pq, pc + PC_PQ + PC_NPQ * (it & 3) + (blockIdx.x & (PC_NPQ - 1)));
}

// Gershgorin margins of the stored S: d[6 j + r] = S_rr - sum_{c != r} |S_rc| over camera j's block row (the same
// walk as k_pcg_spmv).  min d > 0 proves S positive definite; -min d is a shift that makes S + lambda I diagonally
// dominant -- the damping estimate of the trust-region loop in the block-sparse mode, where the reference's modified
// Cholesky (PSBA/cl_cholmod.cpp:25-107, a dense factorization) has nothing to factor.
__global__ __launch_bounds__(256) void k_bsr_gershgorin(const double *val, const int *rowptr, const int2 *rowent, int nC,
                                                        double *d) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  double off = 0.0, diag = 0.0;
  const int g = lane / 6, r = lane % 6;
  if (j < nC && g < 8) {
    const int e1 = rowptr[j + 1];
    for (int e = rowptr[j] + g; e < e1; e += 8) {
      const int2 en = rowent[e];
      const int how = en.y >> 28;
      const double *B = val + 36 * (size_t)en.x;
      if (how == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) off += fabs(B[6 * r + c]);
      } else if (how == 1) {
#pragma unroll
        for (int c = 0; c < 6; c++) off += fabs(B[6 * c + r]);
      } else {  // the diagonal block, read through its lower triangle
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const double v = c <= r ? B[6 * r + c] : B[6 * c + r];
          if (c == r) diag += v; else off += fabs(v);
        }
      }
    }
  }
  off += __shfl_down(off, 24, 64);
  off += __shfl_down(off, 12, 64);
  off += __shfl_down(off, 6, 64);
  diag += __shfl_down(diag, 24, 64);
  diag += __shfl_down(diag, 12, 64);
  diag += __shfl_down(diag, 6, 64);
  if (j < nC && lane < 6) d[6 * (size_t)j + lane] = diag - off;
}

// start: x = 0, r = b, z = M^-1 r, p = z, q = 0, r.z into slot 0, b.b; one thread per unknown
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
  block_sum_to(bb, pc + PC_BB);
}

// x += alpha p, r -= alpha q with alpha = r.z / p.Sp; then z = M^-1 r with the camera's 6x6 block: a
// thread forms the new residual of all six rows of its camera itself (they are neighbours' rows, read
// from r and q before anybody overwrites them: r and z are written to, r_new is not read back), so the
// update and the preconditioner are one pass; r.r and the next r.z with it
__global__ __launch_bounds__(256) void k_pcg_update(double *x, const double *r, double *rn, const double *p, const double *q,
                                                    const double *minv, double *z, int n, double *pc, int it, int *status,
                                                    int try_id, double tol2) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (pcg_converged_before(pc, it, tol2)) {  // the rest of a burst after convergence: nothing to do
    if (t == 0 && pc[PC_DONE] == 0.0) {
      pc[PC_RRFIN] = it > 0 ? pc[PC_RR + ((it - 1) & 3)] : 0.0;
      pc[PC_DONE] = (double)(it + 1);  // it iterations were carried out (+1: zero means "not yet")
    }
    return;
  }
  __shared__ double sPq;
  if (threadIdx.x < 64) {  // the 64 partial sums of p.Sp
    double v = pc[PC_PQ + PC_NPQ * (it & 3) + threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0) sPq = v;
  }
  __syncthreads();
  const double pq = sPq, rz = pc[PC_RZ + (it & 3)];
  if (!(pq > 0.0)) {  // not positive definite (or a break-down): the LM loop raises mu
    if (t == 0) {
      status[1] = try_id;
      pc[PC_FAIL] = 1.0;
    }
    return;
  }
  const double alpha = rz / pq;
  double rr = 0.0, rzn = 0.0;
  if (t < n) {
    const int j = t / 6, row = t % 6;
    const double *m = minv + 36 * (size_t)j + 6 * row;
    double zz = 0.0, own = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const double rc = r[6 * (size_t)j + c] - alpha * q[6 * (size_t)j + c];
      zz += m[c] * rc;
      own = c == row ? rc : own;
    }
    x[t] += alpha * p[t];
    rn[t] = own;
    z[t] = zz;
    rr = own * own;
    rzn = own * zz;
  }
  block_sum_to(rr, pc + PC_RR + (it & 3));
  block_sum_to(rzn, pc + PC_RZ + ((it + 1) & 3));
}

// p = z + beta p with beta = r.z_new / r.z; the accumulators of the iteration after next zeroed
__global__ __launch_bounds__(256) void k_pcg_direction(const double *z, double *p, int n, double *pc, int it) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < PC_NPQ) pc[PC_PQ + PC_NPQ * ((it + 2) & 3) + t] = 0.0;
  if (t == 0) {
    pc[PC_RR + ((it + 2) & 3)] = 0.0;
    pc[PC_RZ + ((it + 3) & 3)] = 0.0;
  }
  if (t >= n) return;
  const double rz = pc[PC_RZ + (it & 3)];
  if (rz == 0.0) return;  // converged before this iteration (0 / 0 otherwise): p stays as it is
  const double beta = pc[PC_RZ + ((it + 1) & 3)] / rz;
  p[t] = z[t] + beta * p[t];
}

int launch_bsr_finalize(psba_ctx *h, double mu) {
  const int nC = h->d.nC;
  hipLaunchKernelGGL(k_bsr_finalize, dim3((42 * nC + 255) / 256), dim3(256), 0, h->stream, h->bs_val, h->bs_diag, h->bs_ea, h->U,
                     h->ga, h->rank == 0 ? mu : 0.0, nC, h->scal, h->status, h->try_id);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

// lambda = max(0, -min_i d_i) of the stored S (k_bsr_gershgorin), a small positive value when S is diagonally
// dominant although the solve failed (rounding); info3 = { min d, max S_ii, 0 }
int launch_bsr_gershgorin(psba_ctx *h, double *lambda, double *info3) {
  const int n = h->d.nA, nC = h->d.nC;
  double *d = h->pcg_vec + 3 * (size_t)n;  // (the solve's q: free between solves)
  hipLaunchKernelGGL(k_bsr_gershgorin, dim3((nC + 3) / 4), dim3(256), 0, h->stream, h->bs_val, h->bs_rowptr, h->bs_rowent, nC, d);
  PSBA_HIP(h, hipGetLastError());
  std::vector<double> hd((size_t)n);
  PSBA_HIP(h, hipMemcpyAsync(hd.data(), d, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  double dmin = hd[0];
  for (int i = 1; i < n; i++) dmin = hd[i] < dmin ? hd[i] : dmin;
  // max S_ii >= max (d_i): a scale for the fallback
  double dmax = 0.0;
  for (int i = 0; i < n; i++) dmax = hd[i] > dmax ? hd[i] : dmax;
  double lam = dmin < 0.0 ? -dmin : 0.0;
  if (!(lam > 0.0)) lam = dmax > 0.0 ? 1e-6 * dmax : 1.0;
  if (!(lam == lam)) lam = 1.0;  // (NaN in S: the loop's own guard ends it)
  *lambda = lam;
  if (info3) {
    info3[0] = dmin;
    info3[1] = dmax;
    info3[2] = 0.0;
  }
  return PSBA_OK;
}

// S x = e_a by preconditioned conjugate gradients; dpa into dp[0 .. nA)
int launch_pcg_solve(psba_ctx *h) {
  const int n = h->d.nA, nC = h->d.nC, g = (n + 255) / 256;
  hipStream_t s = h->stream;
  // r is kept in two buffers used in turn: an iteration's update reads the old residual of a whole
  // camera (six rows) while other threads write the new one
  double *x = h->dp, *r0 = h->pcg_vec, *z = r0 + n, *p = z + n, *q = p + n, *r1 = q + n, *pc = h->pcg_scal;
  ProfScope ps(h, PSBA_K_CHOLESKY);
  PSBA_HIP(h, hipMemsetAsync(pc, 0, sizeof(double) * PC_ALL, s));
  hipLaunchKernelGGL(k_bsr_diag_inverse, dim3((nC + 63) / 64), dim3(64), 0, s, h->bs_val, h->bs_diag, h->pcg_minv, nC, pc,
                     h->status, h->try_id);
  hipLaunchKernelGGL(k_pcg_start, dim3(g), dim3(256), 0, s, h->bs_ea, h->pcg_minv, x, r0, z, p, q, n, pc);
  double hs[PC_N];
  h->pcg_iters = 0;
  h->pcg_relres = 1.0;
  const double tol2 = h->pcg_tol * h->pcg_tol;
  bool converged = false, failed = false;
  for (int it = 0; it < h->pcg_maxit;) {
    const int burst = it + 8 < h->pcg_maxit ? 8 : h->pcg_maxit - it;
    int last = it;
    for (int k = 0; k < burst; k++, it++) {
      double *rc = (it & 1) ? r1 : r0, *rn = (it & 1) ? r0 : r1;
      hipLaunchKernelGGL(k_pcg_spmv, dim3((nC + 3) / 4), dim3(256), 0, s, h->bs_val, h->bs_rowptr, h->bs_rowent, nC, p, q, pc, it);
      hipLaunchKernelGGL(k_pcg_update, dim3(g), dim3(256), 0, s, x, rc, rn, p, q, h->pcg_minv, z, n, pc, it, h->status,
                         h->try_id, tol2);
      if (k == burst - 1) {  // the scalars of this burst's last iteration
        PSBA_HIP(h, hipMemcpyAsync(h->pcg_host, pc, sizeof(double) * PC_N, hipMemcpyDeviceToHost, s));
        last = it;
      }
      hipLaunchKernelGGL(k_pcg_direction, dim3(g), dim3(256), 0, s, z, p, n, pc, it);
    }
    PSBA_HIP(h, hipGetLastError());
    PSBA_HIP(h, hipStreamSynchronize(s));
    for (int k = 0; k < PC_N; k++) hs[k] = h->pcg_host[k];
    h->pcg_iters = it;
    if (hs[PC_FAIL] != 0.0) {
      failed = true;
      break;
    }
    double rr = hs[PC_RR + (last & 3)];
    if (hs[PC_DONE] != 0.0) {  // the test held inside the burst: the device stopped iterating there
      h->pcg_iters = (int)hs[PC_DONE] - 1;
      rr = hs[PC_RRFIN];
      converged = true;
    } else if (rr <= tol2 * hs[PC_BB]) {  // ... or in its last iteration
      converged = true;
    }
    h->pcg_relres = hs[PC_BB] > 0.0 ? sqrt(rr / hs[PC_BB]) : 0.0;
    if (converged || !(rr == rr)) break;
  }
  // max_iter iterations without reaching tol: the step is used as it is (an inexact Newton step; the gain
  // ratio judges it), but the caller is told -- psba_schur_solve returns PSBA_PCG_MAXIT
  h->pcg_exhausted = !converged && !failed;
  return PSBA_OK;
}

}  // namespace psba
