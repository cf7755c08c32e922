// kernels_freek.hip -- free intrinsics: the same path with an 11-parameter camera block
// (fu, u0, v0, ar, s | local rotation | translation), the layout the reference's driver reads (PSBA/main.cpp:73,
// 140-149: origin_cnp = 11) before it strips the intrinsics and optimises six (CL_files/PSBA.cl:5-7: cnp 6).  The
// reference never implemented this: there is no reference arithmetic to match -- PARITY UNPINNED; the
// oracle's twin (oracle/psba_oracle.c, orc_fk_*) is checked by finite differences and against a dense solve of the
// full normal equations (tests/test_freek.py).
//
// Smallest useful size: ONE plain route, correct for any camera count, not tuned -- thread per observation, the
// camera sums and the blocks of S through global fp64 atomics (the first-generation assembly of the six-parameter
// path), then the generic dense factorization (kernels_chol*.hip works on any nA).  Everything behind
// psba_set_camera_model(h, PSBA_CAMERA_FREE_K): the fused verbs and psba_levmar; the sba_func.h mirror, the
// trust-region operators, the block-sparse solver and rank layouts stay six-parameter only.
#include "camera_model.h"
#include "psba_internal.h"
#include "schur_common.h"

namespace psba {

constexpr int FK_W = 3 * FK_CNP;                          // doubles of a W block (11 x 3)
constexpr int FK_ACC = FK_CNP * (FK_CNP + 1) / 2 + FK_CNP;  // per camera: upper triangle of A^T A (66) | A^T e (11)

struct FkArgs {
  const double *camconst, *cams, *pts, *impts;
  const int *iidx, *jidx, *ptr;
  double *W, *PV, *camacc;
  double coeff;
  int nO;
  const double *pub_src;  // the look-ahead linearization carries K3's scalar block to the host (see k_linearize)
  double *pub_dst;
  double pub_stamp;
};

__device__ __forceinline__ void fk_load(const FkArgs &p, int a, int &i, int &j, double *cam, double *q0, double *M, double2 &m) {
  i = p.iidx[a];
  j = p.jidx[a];
#pragma unroll
  for (int k = 0; k < FK_CNP; k++) cam[k] = p.cams[FK_CNP * (size_t)j + k];
#pragma unroll
  for (int k = 0; k < 4; k++) q0[k] = p.camconst[9 * (size_t)j + 5 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) M[k] = p.pts[3 * (size_t)i + k];
  m = reinterpret_cast<const double2 *>(p.impts)[a];
}

// residual, Jacobian, W; raw sums of U_j, g_a,j (camacc) and of V_i, g_b,i (PV) by global atomics
__global__ __launch_bounds__(256) void k_fk_linearize(FkArgs p) {
  if (p.pub_dst && blockIdx.x == 0) {
    if (threadIdx.x < NSCAL) p.pub_dst[threadIdx.x] = p.pub_src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) p.pub_dst[NSCAL] = p.pub_stamp;
  }
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < p.nO; a += gridDim.x * blockDim.x) {
    int i, j;
    double cam[FK_CNP], q0[4], M[3], e[2], A[2 * FK_CNP], B[6];
    double2 m;
    fk_load(p, a, i, j, cam, q0, M, m);
    linearize_obs_freek(cam, q0, M, m.x, m.y, e, A, B);
    double *w = p.W + FK_W * (size_t)a;
#pragma unroll
    for (int r = 0; r < FK_CNP; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) w[3 * r + c] = p.coeff * (A[r] * B[c] + A[FK_CNP + r] * B[3 + c]);
    double *acc = p.camacc + FK_ACC * (size_t)j;
    int q = 0;
#pragma unroll
    for (int r = 0; r < FK_CNP; r++)
#pragma unroll
      for (int c = r; c < FK_CNP; c++) atomicAdd(&acc[q++], A[r] * A[c] + A[FK_CNP + r] * A[FK_CNP + c]);
#pragma unroll
    for (int r = 0; r < FK_CNP; r++) atomicAdd(&acc[q + r], A[r] * e[0] + A[FK_CNP + r] * e[1]);
    double *pv = p.PV + 9 * (size_t)i;
    q = 0;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = r; c < 3; c++) atomicAdd(&pv[q++], B[r] * B[c] + B[3 + r] * B[3 + c]);
#pragma unroll
    for (int r = 0; r < 3; r++) atomicAdd(&pv[6 + r], B[r] * e[0] + B[3 + r] * e[1]);
  }
}

// expands the camera sums (U_j full 11 x 11 scaled by coeff, g_a by coeff_g) and scales the point sums
__global__ __launch_bounds__(256) void k_fk_finish_sums(const double *camacc, int nC, int nP, double coeff, double coeff_g,
                                                        double *U, double *ga, double *PV) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nU = (size_t)nC * FK_CNP * FK_CNP, nG = (size_t)nC * FK_CNP, nV = (size_t)nP * 9;
  if (t < nU) {
    const int j = (int)(t / (FK_CNP * FK_CNP)), e = (int)(t % (FK_CNP * FK_CNP)), r = e / FK_CNP, c = e % FK_CNP;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    U[t] = coeff * camacc[(size_t)j * FK_ACC + lo * FK_CNP - lo * (lo - 1) / 2 + (hi - lo)];
  } else if (t < nU + nG) {
    const size_t u = t - nU;
    ga[u] = coeff_g * camacc[(u / FK_CNP) * FK_ACC + FK_CNP * (FK_CNP + 1) / 2 + u % FK_CNP];
  } else if (t < nU + nG + nV) {
    const size_t u = t - nU - nG;
    PV[u] *= (u % 9 < 6) ? coeff : coeff_g;
  }
}

__global__ __launch_bounds__(256) void k_fk_residual(FkArgs p, double *cost) {
  __shared__ double sRed[4];
  double sum = 0.0;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < p.nO; a += gridDim.x * blockDim.x) {
    int i, j;
    double cam[FK_CNP], q0[4], M[3], e0, e1;
    double2 m;
    fk_load(p, a, i, j, cam, q0, M, m);
    residual_obs(cam, q0, cam + 5, M, m.x, m.y, e0, e1);
    sum += e0 * e0 + e1 * e1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(cost, sRed[0] + sRed[1] + sRed[2] + sRed[3]);
}

__global__ __launch_bounds__(256) void k_fk_max_diag(const double *U, const double *PV, int nC, int nP, double *out) {
  __shared__ double sRed[4];
  double m = 0.0;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsize = gridDim.x * blockDim.x;
  for (int t = gtid; t < FK_CNP * nC; t += gsize) m = fmax(m, U[(size_t)FK_CNP * FK_CNP * (t / FK_CNP) + (FK_CNP + 1) * (t % FK_CNP)]);
  for (int i = gtid; i < nP; i += gsize) {
    const double *v = PV + 9 * (size_t)i;
    m = fmax(m, fmax(v[0], fmax(v[3], v[5])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmax(fmax(sRed[0], sRed[1]), fmax(sRed[2], sRed[3]));
    atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(m));
  }
}

// S (lower block triangle) -= Y_a W_b^T, e_a -= Y_a g_b,i; thread per observation a, partners b <= a of its point
__global__ __launch_bounds__(256) void k_fk_schur(const double *W, const double *PV, const int *iidx, const int *jidx,
                                                  const int *ptr, double *S, double *ea, int ld, double mu, int nO,
                                                  int *status, int try_id) {
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < nO; a += gridDim.x * blockDim.x) {
    const int i = iidx[a], ja = jidx[a];
    const double *pv = PV + 9 * (size_t)i;
    double v[6], vi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pv[k];
    v[0] += mu;
    v[3] += mu;
    v[5] += mu;
    if (sym3_inverse(v, vi)) status[0] = try_id;
    const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
    const double *w = W + FK_W * (size_t)a;
    double Y[FK_W];
#pragma unroll
    for (int r = 0; r < FK_CNP; r++) {
      const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
      Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
      Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
      Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      atomicAdd(&ea[FK_CNP * ja + r], -(Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2));
    }
    for (int b = ptr[i]; b <= a; b++) {
      double *Sblk = S + (size_t)(FK_CNP * ja) * ld + FK_CNP * jidx[b];
      const double *wb = W + FK_W * (size_t)b;
      for (int c = 0; c < FK_CNP; c++) {
        const double w0 = wb[3 * c], w1 = wb[3 * c + 1], w2 = wb[3 * c + 2];
#pragma unroll
        for (int r = 0; r < FK_CNP; r++)
          atomicAdd(&Sblk[(size_t)r * ld + c], -(Y[3 * r] * w0 + Y[3 * r + 1] * w1 + Y[3 * r + 2] * w2));
      }
    }
  }
}

// S += blockdiag(U) + mu I on the lower block triangle, mirrored to the upper; e_a += g_a; identity padding;
// the accumulators of the try's back-substitution zeroed, the try stamp set (as k_schur_finalize does)
__global__ __launch_bounds__(256) void k_fk_finalize(double *S, double *ea, const double *U, const double *ga, double mu,
                                                     int nA, int n32, double *scal, int *status, int try_id) {
  if (blockIdx.x == 0 && threadIdx.x < 4 * SC_NPART) scal[SC_PART + threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 64) status[3] = try_id;
  const size_t n2 = (size_t)nA * nA;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (size_t)gridDim.x * blockDim.x;
  for (size_t t = gtid; t < n2; t += gsize) {
    const int r = (int)(t / nA), c = (int)(t % nA);
    const int kb = r / FK_CNP, lb = c / FK_CNP;
    const size_t at = (size_t)r * n32 + c;
    if (lb > kb) {
      S[at] = S[(size_t)c * n32 + r];
    } else if (lb == kb) {
      double v = S[at] + U[(size_t)FK_CNP * FK_CNP * kb + FK_CNP * (r - FK_CNP * kb) + (c - FK_CNP * lb)];
      if (r == c) v += mu;
      S[at] = v;
    }
  }
  for (size_t t = gtid; t < (size_t)nA; t += gsize) ea[t] += ga[t];
  write_padding(S, nA, n32, 1.0, gtid, gsize);
}

// back-substitution, thread per point: e_b,i = g_b,i - sum_j W_ij^T dpa_j, dpb_i = V*_i^-1 e_b,i, proposed point;
// then the residuals of its observations at the proposal; the four sums of the try as k_backsub forms them
struct FkBackArgs {
  const double *W, *PV, *camconst, *cams, *pts, *impts, *ga;
  const int *jidx, *ptr;
  double *dp, *newcams, *newpts, *scal;
  const int *status;
  double mu;
  int nC, nA, nP;
};
__global__ __launch_bounds__(256) void k_fk_backsub_cams(FkBackArgs p) {  // launched first: proposal cams + camera terms
  __shared__ double sRed[3][4];
  double s_dp = 0.0, s_den = 0.0, s_np = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    p.scal[SC_STATUS_V] = (p.status[0] == p.status[3]) ? 1.0 : 0.0;
    p.scal[SC_STATUS_SPD] = (p.status[1] == p.status[3]) ? 1.0 : 0.0;
  }
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < p.nA; t += gridDim.x * blockDim.x) {
    const double d = p.dp[t], c = p.cams[t] + d;
    p.newcams[t] = c;
    s_dp += d * d;
    s_den += d * (p.mu * d + p.ga[t]);
    s_np += c * c;
  }
  double v3[3] = {s_dp, s_den, s_np};
#pragma unroll
  for (int q = 0; q < 3; q++) {
    double v = v3[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sRed[q][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double *dst = p.scal + SC_PART + 4 * (blockIdx.x % SC_NPART);
    atomicAdd(&dst[0], sRed[0][0] + sRed[0][1] + sRed[0][2] + sRed[0][3]);
    atomicAdd(&dst[1], sRed[1][0] + sRed[1][1] + sRed[1][2] + sRed[1][3]);
    atomicAdd(&dst[3], sRed[2][0] + sRed[2][1] + sRed[2][2] + sRed[2][3]);
  }
}
__global__ __launch_bounds__(256) void k_fk_backsub_pts(FkBackArgs p) {
  __shared__ double sRed[4][4];
  double s_dp = 0.0, s_den = 0.0, s_cost = 0.0, s_np = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.nP; i += gridDim.x * blockDim.x) {
    const double *pv = p.PV + 9 * (size_t)i;
    const int o0 = p.ptr[i], o1 = p.ptr[i + 1];
    double e0 = pv[6], e1 = pv[7], e2 = pv[8];
    for (int a = o0; a < o1; a++) {
      const double *w = p.W + FK_W * (size_t)a;
      const double *da = p.dp + FK_CNP * (size_t)p.jidx[a];
#pragma unroll
      for (int k = 0; k < FK_CNP; k++) {
        e0 -= w[3 * k] * da[k];
        e1 -= w[3 * k + 1] * da[k];
        e2 -= w[3 * k + 2] * da[k];
      }
    }
    double v[6], vi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pv[k];
    v[0] += p.mu;
    v[3] += p.mu;
    v[5] += p.mu;
    sym3_inverse(v, vi);
    const double d[3] = {vi[0] * e0 + vi[1] * e1 + vi[2] * e2, vi[1] * e0 + vi[3] * e1 + vi[4] * e2,
                         vi[2] * e0 + vi[4] * e1 + vi[5] * e2};
    double n3[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      n3[q] = p.pts[3 * (size_t)i + q] + d[q];
      p.dp[p.nA + 3 * (size_t)i + q] = d[q];
      p.newpts[3 * (size_t)i + q] = n3[q];
      s_dp += d[q] * d[q];
      s_den += d[q] * (p.mu * d[q] + pv[6 + q]);
      s_np += n3[q] * n3[q];
    }
    for (int a = o0; a < o1; a++) {  // (newcams: written by k_fk_backsub_cams, launched before this kernel)
      const int j = p.jidx[a];
      double cam[FK_CNP], q0[4], r0, r1;
#pragma unroll
      for (int k = 0; k < FK_CNP; k++) cam[k] = p.newcams[FK_CNP * (size_t)j + k];
#pragma unroll
      for (int k = 0; k < 4; k++) q0[k] = p.camconst[9 * (size_t)j + 5 + k];
      const double2 m = reinterpret_cast<const double2 *>(p.impts)[a];
      residual_obs(cam, q0, cam + 5, n3, m.x, m.y, r0, r1);
      s_cost += r0 * r0 + r1 * r1;
    }
  }
  double v4[4] = {s_dp, s_den, s_cost, s_np};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double v = v4[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sRed[q][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double v = sRed[threadIdx.x][0] + sRed[threadIdx.x][1] + sRed[threadIdx.x][2] + sRed[threadIdx.x][3];
    atomicAdd(&p.scal[SC_PART + 4 * (blockIdx.x % SC_NPART) + threadIdx.x], v);
  }
}

static FkArgs fk_args(psba_ctx *h, int set) {
  FkArgs a;
  a.camconst = h->camconst;
  a.cams = h->cams[set];
  a.pts = h->pts[set];
  a.impts = h->impts;
  a.iidx = h->iidx;
  a.jidx = h->jidx;
  a.ptr = h->ptr;
  a.W = nullptr;
  a.PV = nullptr;
  a.camacc = h->camacc;
  a.coeff = h->coeff;
  a.nO = h->d.nO;
  a.pub_src = h->scal;
  a.pub_dst = nullptr;
  a.pub_stamp = 0.0;
  return a;
}
static int fk_grid(long long n) {
  const long long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int launch_linearize_fk(psba_ctx *h, bool ahead, bool publish) {
  const Dims &d = h->d;
  const int set = ahead ? 1 - h->cur : h->cur;
  FkArgs a = fk_args(h, set);
  a.W = ahead ? h->W_alt : h->W;
  a.PV = ahead ? h->PV_alt : h->PV;
  a.pub_dst = publish ? h->h_scal_dev : nullptr;
  a.pub_stamp = h->pub_seq;
  h->coeff_w = h->coeff;
  double *Uo = ahead ? h->U_alt : h->U, *gao = ahead ? h->ga_alt : h->ga;
  ProfScope ps(h, PSBA_K_LINEARIZE);
  PSBA_HIP(h, hipMemsetAsync(h->camacc, 0, sizeof(double) * FK_ACC * (size_t)d.nC, h->stream));
  PSBA_HIP(h, hipMemsetAsync(a.PV, 0, sizeof(double) * 9 * (size_t)d.nP, h->stream));
  hipLaunchKernelGGL(k_fk_linearize, dim3(fk_grid(d.nO)), dim3(256), 0, h->stream, a);
  const long long nfin = (long long)d.nC * FK_CNP * FK_CNP + (long long)d.nC * FK_CNP + (long long)d.nP * 9;
  hipLaunchKernelGGL(k_fk_finish_sums, dim3((unsigned)((nfin + 255) / 256)), dim3(256), 0, h->stream, h->camacc, d.nC, d.nP,
                     h->coeff, h->coeff_g, Uo, gao, a.PV);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_residual_fk(psba_ctx *h, int which) {
  const int set = which == PSBA_PARAMS_NEW ? 1 - h->cur : h->cur;
  FkArgs a = fk_args(h, set);
  PSBA_HIP(h, hipMemsetAsync(h->scal + SC_COST, 0, sizeof(double), h->stream));
  int grid = fk_grid(h->d.nO);
  if (grid > 256) grid = 256;
  ProfScope ps(h, PSBA_K_RESIDUAL);
  hipLaunchKernelGGL(k_fk_residual, dim3(grid), dim3(256), 0, h->stream, a, h->scal + SC_COST);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_max_diag_fk(psba_ctx *h) {
  PSBA_HIP(h, hipMemsetAsync(h->scal + SC_MAXDIAG, 0, sizeof(double), h->stream));
  int grid = fk_grid(h->d.nP);
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(k_fk_max_diag, dim3(grid), dim3(256), 0, h->stream, h->U, h->PV, h->d.nC, h->d.nP,
                     h->scal + SC_MAXDIAG);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_schur_fk(psba_ctx *h, double mu) {
  const Dims &d = h->d;
  h->try_id++;  // (the status words are generation stamps, as in launch_schur)
  h->diag_done = false;
  double *S = h->red, *ea = h->red + (size_t)h->n32 * h->n32;
  PSBA_HIP(h, hipMemsetAsync(h->red, 0, sizeof(double) * (size_t)(h->n32 + 1) * h->n32, h->stream));
  {
    ProfScope ps(h, PSBA_K_SCHUR);
    hipLaunchKernelGGL(k_fk_schur, dim3(fk_grid(d.nO)), dim3(256), 0, h->stream, h->W, h->PV, h->iidx, h->jidx, h->ptr, S, ea,
                       h->n32, mu, d.nO, h->status, h->try_id);
    const size_t n2 = (size_t)d.nA * d.nA;
    const int fgrid = (int)((n2 + 255) / 256 > 4096 ? 4096 : (n2 + 255) / 256);
    hipLaunchKernelGGL(k_fk_finalize, dim3(fgrid), dim3(256), 0, h->stream, S, ea, h->U, h->ga, mu, d.nA, h->n32, h->scal,
                       h->status, h->try_id);
  }
  h->packed_pending = false;
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_backsub_fk(psba_ctx *h, double mu) {
  const Dims &d = h->d;
  FkBackArgs a;
  a.W = h->W;
  a.PV = h->PV;
  a.camconst = h->camconst;
  a.cams = h->cams[h->cur];
  a.pts = h->pts[h->cur];
  a.impts = h->impts;
  a.ga = h->ga;
  a.jidx = h->jidx;
  a.ptr = h->ptr;
  a.dp = h->dp;
  a.newcams = h->cams[1 - h->cur];
  a.newpts = h->pts[1 - h->cur];
  a.scal = h->scal;
  a.status = h->status;
  a.mu = mu;
  a.nC = d.nC;
  a.nA = d.nA;
  a.nP = d.nP;
  ProfScope ps(h, PSBA_K_BACKSUB);
  hipLaunchKernelGGL(k_fk_backsub_cams, dim3(fk_grid(d.nA) > 16 ? 16 : fk_grid(d.nA)), dim3(256), 0, h->stream, a);
  hipLaunchKernelGGL(k_fk_backsub_pts, dim3(fk_grid(d.nP)), dim3(256), 0, h->stream, a);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
