// kernels_backsub.hip -- K3: point back-substitution, proposal and its cost in one pass.
//
// Replaces kern_compute_eb, kern_compute_dpb, kern_compute_newp, the second
// kern_compute_exQT of a damping try and the host-side compute_L2_sq / compute_rho sums
// (reference CL_files/compute_eb.cl:6-41, compute_dpb.cl:6-35, compute_newp.cl:6-26,
// compute_exQT.cl:18-71; PSBA/sba_func.cpp:1001-1213; PSBA/levmar.cpp:151-195,271-280;
// PSBA/misc.cpp:151-157).  The reference copies dp (nT doubles) and ex (2 nO doubles) to the
// host on every try; here only four scalars leave the device.
//
// Per tile of whole points: thread per observation forms W_ij^T dpa_j; thread per point
// finishes e_b,i, applies V*_i^-1, writes dpb_i and the proposed point; thread per
// observation then evaluates the residual at the proposal.
#include <cstdlib>

#include "camera_model.h"
#include "psba_internal.h"

namespace psba {

struct BackArgs {
  const double *W, *PV, *camconst, *cams, *pts, *impts, *ga;
  const int *iidx, *jidx, *ptr;
  const int4 *tile_desc;
  double *dp;          // [nA] dpa (in) | [nB] dpb (out)
  double *newcams, *newpts;
  double *scal;        // the four sums accumulate in scal[SC_PART ...]
  double *dbg_eb;
  const int *status;   // [0] singular-V stamp, [1] not-SPD stamp, [3] this try's stamp
  double mu;
  double coeff;        // W_ij = coeff A_ij^T B_ij (the coefficient of the linearization W came from)
  int nC, nA, nTiles;
  int cam_terms;       // 1 on the rank that owns the camera terms of the scalar sums
  int mode;            // development ablation (PSBA_BACK_MODE): 1 no residual pass, 2 no W^T dpa pass
};

// RECOMP: W_ij^T dpa_j = coeff B_ij^T (A_ij dpa_j) from the Jacobian blocks recomputed at the current
// parameters (~300 flop per observation) instead of from the stored W (144 bytes per observation:
// 50 of the kernel's 66 MB at venice size); the camera constants and parameters loaded for it are
// the ones the residual pass needs anyway.  (PSBA_BACK_READ_W=1: the W-reading form.)
template <bool DUMP, bool RECOMP>
__global__ __launch_bounds__(TILE_OBS) void k_backsub(BackArgs p) {
  __shared__ double sT[TILE_OBS][3];   // W_a^T dpa_j per observation
  __shared__ double sNP[TILE_OBS][3];  // proposed point per point of the tile
  __shared__ double sRed[4][4];
  const int tid = threadIdx.x;
  double s_dp = 0.0, s_den = 0.0, s_cost = 0.0, s_np = 0.0;

  // camera part (once): proposal cams + dpa and the camera terms of the scalar sums.
  // g_a here is this rank's partial; the sum over ranks of dpa.g_a is the full term.
  if (blockIdx.x == 0) {
    // the try's status as summable flags right behind the partial sums, so that
    // one all-reduce (sum) over ranks covers scalars and status
    if (tid == 0) {
      p.scal[SC_STATUS_V] = (p.status[0] == p.status[3]) ? 1.0 : 0.0;
      p.scal[SC_STATUS_SPD] = (p.status[1] == p.status[3]) ? 1.0 : 0.0;
    }
    for (int t = tid; t < p.nA; t += TILE_OBS) {
      const double d = p.dp[t], c = p.cams[t] + d;
      p.newcams[t] = c;
      s_den += d * p.ga[t];
      if (p.cam_terms) {
        s_dp += d * d;
        s_den += p.mu * d * d;
        s_np += c * c;
      }
    }
  }

  // the dependent index loads of a tile (descriptor -> observation indices -> point CSR) are
  // issued one tile ahead
  int tile = blockIdx.x;
  int4 dsc = tile < p.nTiles ? p.tile_desc[tile] : make_int4(0, 0, 0, 0);
  int i = 0, j = 0;
  if (dsc.z + tid < dsc.w) {
    i = p.iidx[dsc.z + tid];
    j = p.jidx[dsc.z + tid];
  }
  for (; tile < p.nTiles; tile += gridDim.x) {
    const int p0 = dsc.x, p1 = dsc.y, o0 = dsc.z, o1 = dsc.w;
    const int a = o0 + tid;
    const int tn = tile + gridDim.x;
    const int4 dn = tn < p.nTiles ? p.tile_desc[tn] : make_int4(0, 0, 0, 0);
    int pb0 = 0, pb1 = 0;  // CSR bounds of this thread's point
    if (p0 + tid < p1) {
      pb0 = p.ptr[p0 + tid] - o0;
      pb1 = p.ptr[p0 + tid + 1] - o0;
    }
    __syncthreads();
    double cc[9], cam[6], da[6];
    if (RECOMP && a < o1) {
#pragma unroll
      for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * j + k];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        cam[k] = p.cams[6 * j + k];
        da[k] = p.dp[6 * j + k];
      }
    }
    if (a < o1 && p.mode != 2) {
      double t0 = 0.0, t1 = 0.0, t2 = 0.0;
      if (RECOMP) {
        double M[3], e[2], A[12], B[6];
#pragma unroll
        for (int k = 0; k < 3; k++) M[k] = p.pts[3 * (size_t)i + k];
        linearize_obs(cc, cc + 5, cam, M, 0.0, 0.0, e, A, B);
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          s0 += A[k] * da[k];
          s1 += A[6 + k] * da[k];
        }
        t0 = p.coeff * (B[0] * s0 + B[3] * s1);
        t1 = p.coeff * (B[1] * s0 + B[4] * s1);
        t2 = p.coeff * (B[2] * s0 + B[5] * s1);
      } else {
        const double *w = p.W + 18 * (size_t)a;
        const double *dw = p.dp + 6 * j;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const double dk = dw[k];
          t0 += w[3 * k] * dk;
          t1 += w[3 * k + 1] * dk;
          t2 += w[3 * k + 2] * dk;
        }
      }
      sT[tid][0] = t0;
      sT[tid][1] = t1;
      sT[tid][2] = t2;
    }
    __syncthreads();
    if (p0 + tid < p1) {
      const int ip = p0 + tid;
      const double *pv = p.PV + 9 * (size_t)ip;
      const int b0 = pb0, b1 = pb1;
      const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
      double e0 = 0.0, e1 = 0.0, e2 = 0.0;
      for (int b = b0; b < b1; b++) {
        e0 += sT[b][0];
        e1 += sT[b][1];
        e2 += sT[b][2];
      }
      e0 = g0 - e0;
      e1 = g1 - e1;
      e2 = g2 - e2;
      if (DUMP) {
        p.dbg_eb[3 * (size_t)ip] = e0;
        p.dbg_eb[3 * (size_t)ip + 1] = e1;
        p.dbg_eb[3 * (size_t)ip + 2] = e2;
      }
      double v[6], vi[6];
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] = pv[k];
      v[0] += p.mu;
      v[3] += p.mu;
      v[5] += p.mu;
      sym3_inverse(v, vi);
      const double d0 = vi[0] * e0 + vi[1] * e1 + vi[2] * e2;
      const double d1 = vi[1] * e0 + vi[3] * e1 + vi[4] * e2;
      const double d2 = vi[2] * e0 + vi[4] * e1 + vi[5] * e2;
      double *dpb = p.dp + p.nA + 3 * (size_t)ip;
      dpb[0] = d0;
      dpb[1] = d1;
      dpb[2] = d2;
      const double *M = p.pts + 3 * (size_t)ip;
      const double n0 = M[0] + d0, n1 = M[1] + d1, n2 = M[2] + d2;
      double *np = p.newpts + 3 * (size_t)ip;
      np[0] = n0;
      np[1] = n1;
      np[2] = n2;
      sNP[tid][0] = n0;
      sNP[tid][1] = n1;
      sNP[tid][2] = n2;
      s_dp += d0 * d0 + d1 * d1 + d2 * d2;
      s_den += d0 * (p.mu * d0 + g0) + d1 * (p.mu * d1 + g1) + d2 * (p.mu * d2 + g2);
      s_np += n0 * n0 + n1 * n1 + n2 * n2;
    }
    __syncthreads();
    if (a < o1 && p.mode != 1) {
      double e0, e1;
      if (!RECOMP) {
#pragma unroll
        for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * j + k];
#pragma unroll
        for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * j + k] + p.dp[6 * j + k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; k++) cam[k] += da[k];
      }
      const double2 m = reinterpret_cast<const double2 *>(p.impts)[a];
      residual_obs(cc, cc + 5, cam, sNP[i - p0], m.x, m.y, e0, e1);
      s_cost += e0 * e0 + e1 * e1;
    }
    dsc = dn;
    i = j = 0;
    if (dsc.z + tid < dsc.w) {
      i = p.iidx[dsc.z + tid];
      j = p.jidx[dsc.z + tid];
    }
  }
  // workgroup reduction of the four sums, one atomic each
  double v4[4] = {s_dp, s_den, s_cost, s_np};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double v = v4[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) sRed[q][tid >> 6] = v;
  }
  __syncthreads();
  // one wave-instruction with four lanes: same-address atomics serialise at ~13 ns per
  // request, so the requests are spread over SC_NPART partial sets (summed on the host)
  if (tid < 4) {
    const double v = sRed[tid][0] + sRed[tid][1] + sRed[tid][2] + sRed[tid][3];
    atomicAdd(&p.scal[SC_PART + 4 * (blockIdx.x % SC_NPART) + tid], v);
  }
}

// A point seen by more than TILE_OBS cameras (compute_eb.cl:28-37 loops over all cameras): one
// workgroup walks its observations twice, TILE_OBS at a time -- sum of W_ij^T dpa_j (from the
// stored W), then dpb_i and the proposed point by one thread, then the residuals at the proposal.
template <bool DUMP>
__global__ __launch_bounds__(TILE_OBS) void k_backsub_long(BackArgs p, const int *long_pts) {
  __shared__ double sRed[TILE_OBS / 64][4];
  __shared__ double sNP[3];
  const int tid = threadIdx.x, i = long_pts[blockIdx.x];
  const int o0 = p.ptr[i], o1 = p.ptr[i + 1];
  double t3[3] = {0.0, 0.0, 0.0};
  for (int a = o0 + tid; a < o1; a += TILE_OBS) {
    const double *w = p.W + 18 * (size_t)a;
    const double *dw = p.dp + 6 * (size_t)p.jidx[a];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double dk = dw[k];
      t3[0] += w[3 * k] * dk;
      t3[1] += w[3 * k + 1] * dk;
      t3[2] += w[3 * k + 2] * dk;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    double v = t3[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) sRed[tid >> 6][q] = v;
  }
  __syncthreads();
  double s_dp = 0.0, s_den = 0.0, s_cost = 0.0, s_np = 0.0;
  if (tid == 0) {
    const double *pv = p.PV + 9 * (size_t)i;
    double e[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      double v = 0.0;
      for (int w2 = 0; w2 < TILE_OBS / 64; w2++) v += sRed[w2][q];
      e[q] = pv[6 + q] - v;
      if (DUMP) p.dbg_eb[3 * (size_t)i + q] = e[q];
    }
    double v[6], vi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pv[k];
    v[0] += p.mu;
    v[3] += p.mu;
    v[5] += p.mu;
    sym3_inverse(v, vi);
    const double d[3] = {vi[0] * e[0] + vi[1] * e[1] + vi[2] * e[2], vi[1] * e[0] + vi[3] * e[1] + vi[4] * e[2],
                         vi[2] * e[0] + vi[4] * e[1] + vi[5] * e[2]};
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const double n = p.pts[3 * (size_t)i + q] + d[q];
      p.dp[p.nA + 3 * (size_t)i + q] = d[q];
      p.newpts[3 * (size_t)i + q] = n;
      sNP[q] = n;
      s_dp += d[q] * d[q];
      s_den += d[q] * (p.mu * d[q] + pv[6 + q]);
      s_np += n * n;
    }
  }
  __syncthreads();
  for (int a = o0 + tid; a < o1; a += TILE_OBS) {
    const int j = p.jidx[a];
    double cc[9], cam[6], e0, e1;
#pragma unroll
    for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * (size_t)j + k];
#pragma unroll
    for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * (size_t)j + k] + p.dp[6 * (size_t)j + k];
    const double2 m = reinterpret_cast<const double2 *>(p.impts)[a];
    residual_obs(cc, cc + 5, cam, sNP, m.x, m.y, e0, e1);
    s_cost += e0 * e0 + e1 * e1;
  }
  __syncthreads();
  {
    double v = s_cost;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) sRed[tid >> 6][3] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double c = 0.0;
    for (int w2 = 0; w2 < TILE_OBS / 64; w2++) c += sRed[w2][3];
    double *dst = p.scal + SC_PART + 4 * (blockIdx.x % SC_NPART);
    atomicAdd(&dst[0], s_dp);
    atomicAdd(&dst[1], s_den);
    atomicAdd(&dst[2], c);
    atomicAdd(&dst[3], s_np);
  }
}

// the try's scalar block into pinned host memory: one small kernel instead of the runtime's
// 768-byte device-to-host copy (3.9 us on the stream in front of the linearization queued ahead)
__global__ __launch_bounds__(128) void k_publish_scal(const double *scal, double *host) {
  if (threadIdx.x < NSCAL) host[threadIdx.x] = scal[threadIdx.x];
}

int launch_publish_scal(psba_ctx *h, hipStream_t s) {
  hipLaunchKernelGGL(k_publish_scal, dim3(1), dim3(128), 0, s, h->scal, h->h_scal_dev);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_backsub(psba_ctx *h, double mu, bool dump) {
  if (h->cnp != 6) return dump ? fail(h, PSBA_E_STATE, "the sba_func.h mirror is six-parameter only") : launch_backsub_fk(h, mu);
  const Dims &d = h->d;
  BackArgs a;
  a.W = h->W;
  a.PV = h->PV;
  a.camconst = h->camconst;
  a.cams = h->cams[h->cur];
  a.pts = h->pts[h->cur];
  a.impts = h->impts;
  a.ga = h->ga;
  a.iidx = h->iidx;
  a.jidx = h->jidx;
  a.ptr = h->ptr;
  a.tile_desc = h->tile_desc;
  a.dp = h->dp;
  a.newcams = h->cams[1 - h->cur];
  a.newpts = h->pts[1 - h->cur];
  a.scal = h->scal;
  a.dbg_eb = h->dbg_eb;
  a.status = h->status;
  a.mu = mu;
  a.coeff = h->coeff_w;
  a.nC = d.nC;
  a.nA = d.nA;
  a.nTiles = d.nTiles;
  a.cam_terms = h->rank == 0 ? 1 : 0;
  {
    const char *m = getenv("PSBA_BACK_MODE");
    a.mode = m ? atoi(m) : 0;
  }
  // the fused path's accumulators are zeroed by this try's k_schur_reduce; the sba_func.h mirror
  // may run K3 more than once per assembly, so it zeroes them here
  if (dump) PSBA_HIP(h, hipMemsetAsync(h->scal + SC_PART, 0, 4 * SC_NPART * sizeof(double), h->stream));
  // persistent workgroups, three per CU (venice-shaped, 1350 tiles, K3 in us by grid: 512 18.7, 640 17.7,
  // 768 16.6, 896 18.8, 1024 18.2, 1350 17.8)
  int grid = d.nTiles < 768 ? d.nTiles : 768;
  if (const char *e = getenv("PSBA_BACK_GRID")) grid = atoi(e) > 0 && atoi(e) < grid ? atoi(e) : grid;
  {
    ProfScope ps(h, PSBA_K_BACKSUB);
    const bool read_w = getenv("PSBA_BACK_READ_W") != nullptr;
    if (dump && read_w)
      hipLaunchKernelGGL((k_backsub<true, false>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
    else if (dump)
      hipLaunchKernelGGL((k_backsub<true, true>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
    else if (read_w)
      hipLaunchKernelGGL((k_backsub<false, false>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
    else
      hipLaunchKernelGGL((k_backsub<false, true>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
    if (h->nLong) {  // points seen by more cameras than a tile holds
      if (dump)
        hipLaunchKernelGGL(k_backsub_long<true>, dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts);
      else
        hipLaunchKernelGGL(k_backsub_long<false>, dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts);
    }
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
