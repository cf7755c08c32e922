// kernels_linearize.hip -- K1: residual + Jacobian + U/V/W/g assembly in one pass (fp64).
//
// Replaces kern_compute_exQT, kern_compute_jacobiQT, kern_compute_U, kern_compute_V,
// kern_compute_Wblks, kern_compute_g (reference CL_files/compute_exQT.cl:18-71,
// compute_jacobiQT.cl:7-141, compute_U.cl:5-35, compute_V.cl:6-38, compute_Wblks.cl:7-34,
// compute_g.cl:6-60) and their wrappers PSBA/sba_func.cpp:81-617.
//
// Layout: observations are point-major; a *tile* is a run of whole points with at most
// TILE_OBS observations, one thread per observation.  A/B/e live in registers; B and e go
// through LDS once so that one thread per point sums V_i and g_b,i in camera order (the
// reference's summation order); the 27 per-camera sums (sym U_j, g_a,j) are accumulated in
// LDS per workgroup and written once per workgroup as a partial slab that k_cam_reduce
// sums in slab order.  Nothing per-observation except W is written to HBM.
#include <cstdlib>

#include "camera_model.h"
#include "psba_internal.h"

namespace psba {

struct LinArgs {
  const double *camconst, *cams, *pts, *impts;
  const int *iidx, *jidx, *ptr;
  const int4 *tile_desc;
  double *W, *PV, *campart;
  double *camacc;  // GACC: [nC][27] global accumulators (zeroed before the launch)
  double *dbg_ex, *dbg_JA, *dbg_JB;
  double coeff, coeff_g;
  int nC, nTiles;
  int mode;  // development ablation (PSBA_LIN_MODE): 1 no camera atomics, 2 no W store, 3 no per-point sums
  // the linearization queued ahead of the host's verdict carries the previous kernel's (K3's) scalar
  // block to the host: workgroup 0 copies pub_src[0 .. NSCAL) to pinned host memory, then the stamp
  const double *pub_src;
  double *pub_dst;  // nullptr: nothing to publish
  double pub_stamp;
};

// GACC: many cameras -- the 27 sums per camera do not fit the LDS.  They are then formed by a
// camera-major pass of their own (k_cam_sums) and this kernel leaves them out.
template <bool DUMP, bool GACC>
__global__ __launch_bounds__(TILE_OBS) void k_linearize(LinArgs p) {
  __shared__ double sBE[TILE_OBS][9];  // B(6) | e(2) per observation of the tile (+1: odd row stride)
  __shared__ double sW[(TILE_OBS / 2) * 19];  // W blocks of half a tile (staged in two halves: LDS for three workgroups per CU)
  extern __shared__ double sAcc[];     // [nC][27]
  const int tid = threadIdx.x;
  if (p.pub_dst && blockIdx.x == 0) {
    // (a kernel of its own for these 832 bytes sat 4 us on the stream between K3 and this kernel)
    if (tid < NSCAL) p.pub_dst[tid] = p.pub_src[tid];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) p.pub_dst[NSCAL] = p.pub_stamp;
  }
  const int nAcc = GACC ? 0 : p.nC * CAM_ACC;
  for (int t = tid; t < nAcc; t += TILE_OBS) sAcc[t] = 0.0;
  __syncthreads();

  // a workgroup has only a few tiles and few neighbours on its CU (LDS), so the dependent index
  // loads of a tile (descriptor -> observation indices -> point CSR) are issued one tile ahead
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
    // CSR bounds of the (point, component) tasks of this thread (see below)
    constexpr int NTASK = (9 * TILE_PTS + TILE_OBS - 1) / TILE_OBS;  // rounds of (point, component) tasks
    int pb[NTASK][2];
#pragma unroll
    for (int u = 0; u < NTASK; u++) {
      const int ip = (tid + u * TILE_OBS) / 9;
      pb[u][0] = ip < p1 - p0 ? p.ptr[p0 + ip] - o0 : 0;
      pb[u][1] = ip < p1 - p0 ? p.ptr[p0 + ip + 1] - o0 : 0;
    }
    double Wv[18];
    if (a < o1) {
      double cc[9], cam[6], M[3], e[2], A[12], B[6];
#pragma unroll
      for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * j + k];
#pragma unroll
      for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * j + k];
#pragma unroll
      for (int k = 0; k < 3; k++) M[k] = p.pts[3 * i + k];
      const double2 m = reinterpret_cast<const double2 *>(p.impts)[a];
      linearize_obs(cc, cc + 5, cam, M, m.x, m.y, e, A, B);
      if (DUMP) {
        p.dbg_ex[2 * a] = e[0];
        p.dbg_ex[2 * a + 1] = e[1];
#pragma unroll
        for (int k = 0; k < 12; k++) p.dbg_JA[12 * (size_t)a + k] = A[k];
#pragma unroll
        for (int k = 0; k < 6; k++) p.dbg_JB[6 * (size_t)a + k] = B[k];
      }
      // W_ij = coeff * A^T B, 6x3 row-major: staged in LDS (row stride 19 doubles: odd, so the
      // 64 lanes of a store hit distinct banks) and written to HBM as contiguous runs after the
      // barrier, instead of 144-byte pieces at a 144-byte stride per lane -- the first half of
      // the tile's observations now, the second half after the first is flushed
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Wv[3 * r + c] = p.coeff * (A[r] * B[c] + A[6 + r] * B[3 + c]);
      if (tid < TILE_OBS / 2) {
        double *w = sW + 19 * tid;
#pragma unroll
        for (int k = 0; k < 18; k++) w[k] = Wv[k];
      }
#pragma unroll
      for (int k = 0; k < 6; k++) sBE[tid][k] = B[k];
      sBE[tid][6] = e[0];
      sBE[tid][7] = e[1];
      // camera sums: upper triangle of A^T A, then A^T e
      double *acc = sAcc + CAM_ACC * (size_t)j;
      int k = 0;
      if (p.mode != 1 && !GACC) {
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) atomicAdd(&acc[k++], A[r] * A[c] + A[6 + r] * A[6 + c]);
#pragma unroll
      for (int r = 0; r < 6; r++) atomicAdd(&acc[21 + r], A[r] * e[0] + A[6 + r] * e[1]);
      }
    }
    __syncthreads();
    if (p.mode != 2) {
      double *dst = p.W + 18 * (size_t)o0;
      const int n = 18 * (o1 - o0 < TILE_OBS / 2 ? o1 - o0 : TILE_OBS / 2);
      for (int t = tid; t < n; t += TILE_OBS) dst[t] = sW[19 * (t / 18) + t % 18];
    }
    // V_i (sym6) and g_b,i: one thread per (point, component), each summing over the point's
    // observations in camera-ascending order (the reference's order, so V is reproducible bit
    // for bit); nine lanes per point instead of one keeps the lanes busy and the stores coalesced
#pragma unroll
    for (int u = 0; u < NTASK; u++) {
      const int t = tid + u * TILE_OBS;
      if (t >= 9 * (p1 - p0) || p.mode == 3) continue;
      const int comp = t % 9;
      const int b0 = pb[u][0], b1 = pb[u][1];
      // component -> the two products it sums: B[r] * B[c] + B[3+r] * B[3+c]   (V, r <= c)
      //                                    or  B[r] * e0   + B[3+r] * e1      (g_b)
      int r, c1, c2;
      if (comp < 6) {
        r = comp < 3 ? 0 : (comp < 5 ? 1 : 2);
        const int c = comp < 3 ? comp : (comp < 5 ? comp - 2 : 2);
        c1 = c;
        c2 = 3 + c;
      } else {
        r = comp - 6;
        c1 = 6;
        c2 = 7;
      }
      double acc = 0.0;
      for (int b = b0; b < b1; b++) {
        const double *B = sBE[b];
        acc += B[r] * B[c1] + B[3 + r] * B[c2];
      }
      p.PV[9 * (size_t)p0 + t] = (comp < 6 ? p.coeff : p.coeff_g) * acc;
    }
    if (o1 - o0 > TILE_OBS / 2) {  // (tile-uniform) second half of the W blocks through the same buffer
      __syncthreads();
      if (tid >= TILE_OBS / 2 && a < o1) {
        double *w = sW + 19 * (tid - TILE_OBS / 2);
#pragma unroll
        for (int k = 0; k < 18; k++) w[k] = Wv[k];
      }
      __syncthreads();
      if (p.mode != 2) {
        double *dst = p.W + 18 * (size_t)(o0 + TILE_OBS / 2);
        const int n = 18 * (o1 - o0 - TILE_OBS / 2);
        for (int t = tid; t < n; t += TILE_OBS) dst[t] = sW[19 * (t / 18) + t % 18];
      }
    }
    dsc = dn;
    i = j = 0;
    if (dsc.z + tid < dsc.w) {
      i = p.iidx[dsc.z + tid];
      j = p.jidx[dsc.z + tid];
    }
    __syncthreads();
  }
  if (GACC) return;
  double *slab = p.campart + (size_t)blockIdx.x * nAcc;
  for (int t = tid; t < nAcc; t += TILE_OBS) slab[t] = sAcc[t];
}

// ---- round 4: the same pass with fewer LDS round trips per tile -----------------------------------------
// k_linearize above spends a tile in five dependent phases (in-kernel stamps): parameter gathers
// 4900-6900 cycles, Jacobian 1000, staging + camera atomics 4000-6200, first half of W 1900-3100, per-point
// sums through a second LDS pass 4500-5000, second half of W 2100-2700.  Here
//  * V_i and g_b,i are summed ACROSS THE LANES of a point (observations are point-major: a point's observations
//    are adjacent lanes): a segmented Hillis-Steele scan with ds_bpermute shuffles, log2(longest track in the
//    wave) steps of nine values -- no B / e staging, no second pass, no loop whose length is the longest track.
//    A point that straddles two waves is finished through 40 doubles of LDS behind the tile's one barrier.
//    (The sums are no longer taken in the reference's camera order, compute_V.cl:20-34: a fixed tree instead;
//    parity is held to 1e-11, not bit for bit -- as it already is for U and g_a.)
//  * the LDS that B / e needed holds the second half of W: one staging, one barrier, one flush per tile;
//  * the NEXT tile's indices and the eighteen parameter doubles they lead to are fetched as soon as this tile's
//    Jacobian is done, so their latency runs under the camera atomics, the scan and the W flush.
template <bool DUMP, bool GACC>
__global__ __launch_bounds__(TILE_OBS) void k_linearize2(LinArgs p) {
  __shared__ double sW[TILE_OBS * 19];          // W blocks of the tile (row stride 19 doubles: odd, conflict-free stores)
  __shared__ double sCarV[TILE_OBS / 64][9];    // a wave's last lane: its scan values (the tail of its last segment)
  __shared__ int sCarP[TILE_OBS / 64][3];       // point of the wave's first lane, of its last lane, whole wave one point?
  extern __shared__ double sAcc[];              // [nC][27]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.pub_dst && blockIdx.x == 0) {
    if (tid < NSCAL) p.pub_dst[tid] = p.pub_src[tid];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) p.pub_dst[NSCAL] = p.pub_stamp;
  }
  const int nAcc = GACC ? 0 : p.nC * CAM_ACC;
  for (int t = tid; t < nAcc; t += TILE_OBS) sAcc[t] = 0.0;
  __syncthreads();

  int tile = blockIdx.x;
  int4 dsc = tile < p.nTiles ? p.tile_desc[tile] : make_int4(0, 0, 0, 0);
  int i = 0, j = 0;
  double cc[9], cam[6], M[3];
  double2 m = make_double2(0.0, 0.0);
  // prologue: the first tile's operands
  if (dsc.z + tid < dsc.w) {
    i = p.iidx[dsc.z + tid];
    j = p.jidx[dsc.z + tid];
#pragma unroll
    for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * j + k];
#pragma unroll
    for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * j + k];
#pragma unroll
    for (int k = 0; k < 3; k++) M[k] = p.pts[3 * (size_t)i + k];
    m = reinterpret_cast<const double2 *>(p.impts)[dsc.z + tid];
  }
  for (; tile < p.nTiles; tile += gridDim.x) {
    const int o0 = dsc.z, o1 = dsc.w;
    const int a = o0 + tid;
    const bool act = a < o1;
    const int tn = tile + gridDim.x;
    const int4 dn = tn < p.nTiles ? p.tile_desc[tn] : make_int4(0, 0, 0, 0);
    double v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = 0.0;
    const int pi = act ? i : -1 - tid;  // inactive lanes: ids of their own, never merged
    const int jc = j;
    if (act) {
      double e[2], A[12], B[6];
      linearize_obs(cc, cc + 5, cam, M, m.x, m.y, e, A, B);
      if (DUMP) {
        p.dbg_ex[2 * (size_t)a] = e[0];
        p.dbg_ex[2 * (size_t)a + 1] = e[1];
#pragma unroll
        for (int k = 0; k < 12; k++) p.dbg_JA[12 * (size_t)a + k] = A[k];
#pragma unroll
        for (int k = 0; k < 6; k++) p.dbg_JB[6 * (size_t)a + k] = B[k];
      }
      // W_ij = coeff * A^T B, 6x3 row-major, staged (row stride 19) and flushed as contiguous runs below
      double *w = sW + 19 * tid;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) w[3 * r + c] = p.coeff * (A[r] * B[c] + A[6 + r] * B[3 + c]);
      // this observation's terms of V_i (upper triangle) and g_b,i
      int q = 0;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = r; c < 3; c++) v[q++] = B[r] * B[c] + B[3 + r] * B[3 + c];
#pragma unroll
      for (int r = 0; r < 3; r++) v[6 + r] = B[r] * e[0] + B[3 + r] * e[1];
      // camera sums: upper triangle of A^T A, then A^T e
      if (p.mode != 1 && !GACC) {
        double *acc = sAcc + CAM_ACC * (size_t)jc;
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = r; c < 6; c++) atomicAdd(&acc[k++], A[r] * A[c] + A[6 + r] * A[6 + c]);
#pragma unroll
        for (int r = 0; r < 6; r++) atomicAdd(&acc[21 + r], A[r] * e[0] + A[6 + r] * e[1]);
      }
    }
    // the next tile's operands: in flight during everything below
    i = j = 0;
    if (dn.z + tid < dn.w) {
      i = p.iidx[dn.z + tid];
      j = p.jidx[dn.z + tid];
#pragma unroll
      for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * j + k];
#pragma unroll
      for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * j + k];
#pragma unroll
      for (int k = 0; k < 3; k++) M[k] = p.pts[3 * (size_t)i + k];
      m = reinterpret_cast<const double2 *>(p.impts)[dn.z + tid];
    }
    // segmented inclusive scan over the lanes of a point
    if (p.mode != 3) {
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int pu = __shfl_up(pi, d, 64);
        const bool ok = lane >= d && pu == pi;
        if (!__any(ok)) break;  // no segment of this wave reaches d lanes back: none reaches 2 d either
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const double vu = __shfl_up(v[k], d, 64);
          v[k] += ok ? vu : 0.0;
        }
      }
    }
    const int pn = __shfl_down(pi, 1, 64);  // (lane 63 gets its own value back: handled below)
    const int p_first = __shfl(pi, 0, 64), p_last = __shfl(pi, 63, 64);
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 9; k++) sCarV[wave][k] = v[k];
      sCarP[wave][0] = p_first;
      sCarP[wave][1] = p_last;
      sCarP[wave][2] = p_first == p_last;
    }
    __syncthreads();
    // W out: contiguous runs from the staged rows
    if (p.mode != 2) {
      double *dst = p.W + 18 * (size_t)o0;
      const int n = 18 * (o1 - o0);
      for (int t = tid; t < n; t += TILE_OBS) dst[t] = sW[19 * (t / 18) + t % 18];
    }
    // the last lane of a point's run stores V_i | g_b,i; a run that began in earlier waves collects their tails
    if (act && p.mode != 3) {
      const bool cont = lane == 63 && wave + 1 < TILE_OBS / 64 && sCarP[wave + 1][0] == pi;  // goes on in the next wave
      const bool tail = lane == 63 ? !cont : pn != pi;
      if (tail) {
        if (pi == p_first) {  // this wave's first run: may have begun before lane 0
          for (int w2 = wave - 1; w2 >= 0 && sCarP[w2][1] == pi; w2--) {
#pragma unroll
            for (int k = 0; k < 9; k++) v[k] += sCarV[w2][k];
            if (!sCarP[w2][2]) break;  // that wave held the beginning of the run
          }
        }
        double *pv = p.PV + 9 * (size_t)pi;
#pragma unroll
        for (int k = 0; k < 9; k++) pv[k] = (k < 6 ? p.coeff : p.coeff_g) * v[k];
      }
    }
    dsc = dn;
    __syncthreads();  // sW and the carry rows are free again
  }
  if (GACC) return;
  double *slab = p.campart + (size_t)blockIdx.x * nAcc;
  for (int t = tid; t < nAcc; t += TILE_OBS) slab[t] = sAcc[t];
}

// A point seen by more than TILE_OBS cameras (no limit in the reference: compute_V.cl:6-38,
// compute_g.cl:43-58 loop over all cameras): one workgroup walks its observations, TILE_OBS at a
// time.  W as in k_linearize; V_i and g_b,i summed over the workgroup; the camera sums go with global
// fp64 atomics into one extra slab of the per-workgroup camera sums (zeroed before the launch) that
// k_cam_reduce adds like any other -- with GACC the camera-major pass covers these observations anyway.
template <bool DUMP, bool GACC>
__global__ __launch_bounds__(TILE_OBS) void k_linearize_long(LinArgs p, const int *long_pts, double *cam_slab) {
  __shared__ double sRed[TILE_OBS / 64][9];
  const int tid = threadIdx.x, i = long_pts[blockIdx.x];
  const int o0 = p.ptr[i], o1 = p.ptr[i + 1];
  double M[3], acc[9];
#pragma unroll
  for (int k = 0; k < 3; k++) M[k] = p.pts[3 * (size_t)i + k];
#pragma unroll
  for (int k = 0; k < 9; k++) acc[k] = 0.0;
  for (int a = o0 + tid; a < o1; a += TILE_OBS) {
    const int j = p.jidx[a];
    double cc[9], cam[6], e[2], A[12], B[6];
#pragma unroll
    for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * (size_t)j + k];
#pragma unroll
    for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * (size_t)j + k];
    const double2 m = reinterpret_cast<const double2 *>(p.impts)[a];
    linearize_obs(cc, cc + 5, cam, M, m.x, m.y, e, A, B);
    if (DUMP) {
      p.dbg_ex[2 * (size_t)a] = e[0];
      p.dbg_ex[2 * (size_t)a + 1] = e[1];
#pragma unroll
      for (int k = 0; k < 12; k++) p.dbg_JA[12 * (size_t)a + k] = A[k];
#pragma unroll
      for (int k = 0; k < 6; k++) p.dbg_JB[6 * (size_t)a + k] = B[k];
    }
    double *w = p.W + 18 * (size_t)a;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) w[3 * r + c] = p.coeff * (A[r] * B[c] + A[6 + r] * B[3 + c]);
    int k = 0;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = r; c < 3; c++) acc[k++] += B[r] * B[c] + B[3 + r] * B[3 + c];
#pragma unroll
    for (int r = 0; r < 3; r++) acc[6 + r] += B[r] * e[0] + B[3 + r] * e[1];
    if (!GACC) {
      double *cs = cam_slab + CAM_ACC * (size_t)j;
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) atomicAdd(&cs[q++], A[r] * A[c] + A[6 + r] * A[6 + c]);
#pragma unroll
      for (int r = 0; r < 6; r++) atomicAdd(&cs[21 + r], A[r] * e[0] + A[6 + r] * e[1]);
    }
  }
#pragma unroll
  for (int k = 0; k < 9; k++) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) sRed[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid < 9) {
    double v = 0.0;
    for (int q = 0; q < TILE_OBS / 64; q++) v += sRed[q][tid];
    p.PV[9 * (size_t)i + tid] = (tid < 6 ? p.coeff : p.coeff_g) * v;
  }
}

// GACC: U_j = sum_i A_ij^T A_ij and g_a,j = sum_i A_ij^T e_ij (compute_U.cl:22-29, compute_g.cl:29-41)
// by a camera-major pass: one thread per (camera, segment of at most 256 of its observations, points
// ascending), the Jacobian block recomputed from the parameters, 27 sums in registers, one set of
// fp64 atomic adds per segment.  (Through per-observation atomics from the point-major kernel the
// same sums took 25 ms at 20 M observations; the reference scans all points per output scalar.)
__global__ __launch_bounds__(256) void k_cam_sums(LinArgs p, const int *cam_obs, const int4 *units, int nUnits) {
  // one wave per unit (a segment of at most 256 observations of one camera): the lanes stride
  // through the segment, the 27 sums are folded across the wave, lane 0 adds them to the camera's
  // totals.  (One thread per unit walked its 256 gathers one after the other: 380 us for 218 k
  // observations over 600 cameras, 1450 threads on the whole chip.)
  const int lane = threadIdx.x & 63;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= nUnits) return;
  const int4 un = units[u];
  const int j = un.x;
  double cc[9], cam[6], acc[CAM_ACC];
#pragma unroll
  for (int k = 0; k < 9; k++) cc[k] = p.camconst[9 * (size_t)j + k];
#pragma unroll
  for (int k = 0; k < 6; k++) cam[k] = p.cams[6 * (size_t)j + k];
#pragma unroll
  for (int k = 0; k < CAM_ACC; k++) acc[k] = 0.0;
  for (int t = un.y + lane; t < un.z; t += 64) {
    const int a = cam_obs[t];
    const int i = p.iidx[a];
    double M[3], e[2], A[12], B[6];
#pragma unroll
    for (int k = 0; k < 3; k++) M[k] = p.pts[3 * (size_t)i + k];
    const double2 m = reinterpret_cast<const double2 *>(p.impts)[a];
    linearize_obs(cc, cc + 5, cam, M, m.x, m.y, e, A, B);
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = r; c < 6; c++) acc[k++] += A[r] * A[c] + A[6 + r] * A[6 + c];
#pragma unroll
    for (int r = 0; r < 6; r++) acc[21 + r] += A[r] * e[0] + A[6 + r] * e[1];
  }
#pragma unroll
  for (int k = 0; k < CAM_ACC; k++) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc[k] += __shfl_down(acc[k], d, 64);
  }
  if (lane) return;
  double *dst = p.camacc + CAM_ACC * (size_t)j;
#pragma unroll
  for (int k = 0; k < CAM_ACC; k++) atomicAdd(&dst[k], acc[k]);
}

// GACC: expands the packed upper triangle of U_j to the full 6x6 and scales (U by coeff, g_a by coeff_g)
__global__ __launch_bounds__(256) void k_cam_finalize(const double *camacc, int nC, double coeff, double coeff_g,
                                                      double *U, double *ga) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 42 * nC) return;
  const int j = t / 42, e = t % 42;
  const double *src = camacc + (size_t)j * CAM_ACC;
  if (e < 36) {
    const int r = e / 6, c = e % 6;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    U[36 * (size_t)j + e] = coeff * src[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
  } else {
    ga[6 * (size_t)j + e - 36] = coeff_g * src[21 + e - 36];
  }
}

// one workgroup (1024 threads = 32 slab sequences x 32 entries) per camera: sums the partial
// slabs in a fixed order, expands sym U_j to the full 6x6 and scales (U by coeff, g_a by coeff_g).
__global__ __launch_bounds__(1024) void k_cam_reduce(const double *campart, int nPart, int nC,
                                                     double coeff, double coeff_g, double *U,
                                                     double *ga) {
  __shared__ double sPart[32][33];
  const int j = blockIdx.x, e = threadIdx.x & 31, s = threadIdx.x >> 5;
  double acc = 0.0;
  if (e < CAM_ACC) {
    const double *src = campart + (size_t)j * CAM_ACC + e;
    const size_t stride = (size_t)nC * CAM_ACC;
    int q = s;
    for (; q + 96 < nPart; q += 128) {  // four independent loads in flight, summed in order
      const double x0 = src[(size_t)q * stride], x1 = src[(size_t)(q + 32) * stride];
      const double x2 = src[(size_t)(q + 64) * stride], x3 = src[(size_t)(q + 96) * stride];
      acc += x0;
      acc += x1;
      acc += x2;
      acc += x3;
    }
    for (; q < nPart; q += 32) acc += src[(size_t)q * stride];
  }
  sPart[s][e] = acc;
  __syncthreads();
  if (s == 0 && e < CAM_ACC) {
    double t = sPart[0][e];
    for (int q = 1; q < 32; q++) t += sPart[q][e];
    sPart[0][e] = t;
  }
  __syncthreads();
  if (threadIdx.x < 36) {
    const int r = threadIdx.x / 6, c = threadIdx.x % 6;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int k = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);  // index in the packed upper triangle
    U[36 * j + threadIdx.x] = coeff * sPart[0][k];
  } else if (threadIdx.x < 42) {
    ga[6 * j + threadIdx.x - 36] = coeff_g * sPart[0][21 + threadIdx.x - 36];
  }
}

// ---- residual: kern_compute_exQT + the host compute_L2_sq (PSBA/misc.cpp:151-157) -------
__global__ __launch_bounds__(256) void k_residual(const double *camconst, const double *cams,
                                                  const double *pts, const double *impts,
                                                  const int *iidx, const int *jidx, int nO,
                                                  double *ex_out, double *cost) {
  __shared__ double sRed[4];
  double sum = 0.0;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < nO; a += gridDim.x * blockDim.x) {
    const int i = iidx[a], j = jidx[a];
    double cc[9], cam[6], M[3], e0, e1;
#pragma unroll
    for (int k = 0; k < 9; k++) cc[k] = camconst[9 * j + k];
#pragma unroll
    for (int k = 0; k < 6; k++) cam[k] = cams[6 * j + k];
#pragma unroll
    for (int k = 0; k < 3; k++) M[k] = pts[3 * i + k];
    const double2 m = reinterpret_cast<const double2 *>(impts)[a];
    residual_obs(cc, cc + 5, cam, M, m.x, m.y, e0, e1);
    if (ex_out) {
      ex_out[2 * a] = e0;
      ex_out[2 * a + 1] = e1;
    }
    sum += e0 * e0 + e1 * e1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(cost, sRed[0] + sRed[1] + sRed[2] + sRed[3]);
}

// ---- maxElmOfUV (PSBA/sba_func.cpp:422-444) --------------------------------------------
// diagonals of U and V are sums of squares (times a positive coeff in every caller that asks
// for the maximum), so the maximum is taken on the bit patterns with an integer atomicMax.
__global__ __launch_bounds__(256) void k_max_diag(const double *U, const double *PV, int nC,
                                                  int nP, double *out) {
  __shared__ double sRed[4];
  double m = 0.0;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsize = gridDim.x * blockDim.x;
  for (int t = gtid; t < 6 * nC; t += gsize) m = fmax(m, U[36 * (t / 6) + 7 * (t % 6)]);
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
    atomicMax(reinterpret_cast<unsigned long long *>(out),
              (unsigned long long)__double_as_longlong(m));
  }
}

int launch_linearize(psba_ctx *h, bool dump, bool ahead, bool publish) {
  if (h->cnp != 6) return dump ? fail(h, PSBA_E_STATE, "the sba_func.h mirror is six-parameter only") : launch_linearize_fk(h, ahead, publish);
  const Dims &d = h->d;
  LinArgs a;
  a.camconst = h->camconst;
  // ahead: the proposed parameters, into the alternate set of outputs
  const int set = ahead ? 1 - h->cur : h->cur;
  a.cams = h->cams[set];
  a.pts = h->pts[set];
  a.impts = h->impts;
  a.iidx = h->iidx;
  a.jidx = h->jidx;
  a.ptr = h->ptr;
  a.tile_desc = h->tile_desc;
  a.W = ahead ? h->W_alt : h->W;
  a.PV = ahead ? h->PV_alt : h->PV;
  a.campart = h->campart;
  a.camacc = h->camacc;
  a.dbg_ex = h->dbg_ex;
  a.dbg_JA = h->dbg_JA;
  a.dbg_JB = h->dbg_JB;
  a.coeff = h->coeff;
  h->coeff_w = h->coeff;  // K3 forms W^T dpa from the Jacobian blocks: it needs W's coefficient
  a.coeff_g = h->coeff_g;
  a.nC = d.nC;
  a.nTiles = d.nTiles;
  a.pub_src = h->scal;
  a.pub_dst = publish ? h->h_scal_dev : nullptr;
  a.pub_stamp = h->pub_seq;
  {
    const char *m = getenv("PSBA_LIN_MODE");
    a.mode = m ? atoi(m) : 0;
  }
  const bool v1 = getenv("PSBA_LIN_V1") != nullptr;  // round 1-3's kernel (per-point sums through a second LDS pass): cross-check
  const size_t lds = h->cam_global ? 0 : sizeof(double) * CAM_ACC * (size_t)d.nC;
  // static LDS of the kernel is ~54 KiB: beyond 64 KiB in all, the dynamic part needs the attribute
  if (!h->lin_attr_set && lds > 8 * 1024) {
    const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_linearize<true, false>, attr, 100 * 1024));
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_linearize<false, false>, attr, 100 * 1024));
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_linearize2<true, false>, attr, 100 * 1024));
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_linearize2<false, false>, attr, 100 * 1024));
    h->lin_attr_set = true;
  }
  double *Uo = ahead ? h->U_alt : h->U, *gao = ahead ? h->ga_alt : h->ga;
  {
    ProfScope ps(h, PSBA_K_LINEARIZE);
    if (h->cam_global) {
      PSBA_HIP(h, hipMemsetAsync(h->camacc, 0, sizeof(double) * CAM_ACC * (size_t)d.nC, h->stream));
      const int grid = d.nTiles < 2048 ? d.nTiles : 2048;
      if (v1 && dump)
        hipLaunchKernelGGL((k_linearize<true, true>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
      else if (v1)
        hipLaunchKernelGGL((k_linearize<false, true>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
      else if (dump)
        hipLaunchKernelGGL((k_linearize2<true, true>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
      else
        hipLaunchKernelGGL((k_linearize2<false, true>), dim3(grid), dim3(TILE_OBS), 0, h->stream, a);
      if (h->nLong) {
        if (dump)
          hipLaunchKernelGGL((k_linearize_long<true, true>), dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts, (double *)nullptr);
        else
          hipLaunchKernelGGL((k_linearize_long<false, true>), dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts, (double *)nullptr);
      }
      hipLaunchKernelGGL(k_cam_sums, dim3((h->nCamUnits + 3) / 4), dim3(256), 0, h->stream, a, h->cam_obs,
                         h->cam_units, h->nCamUnits);
      hipLaunchKernelGGL(k_cam_finalize, dim3((42 * d.nC + 255) / 256), dim3(256), 0, h->stream, h->camacc, d.nC,
                         h->coeff, h->coeff_g, Uo, gao);
    } else {
      if (v1 && dump)
        hipLaunchKernelGGL((k_linearize<true, false>), dim3(h->nPart), dim3(TILE_OBS), lds, h->stream, a);
      else if (v1)
        hipLaunchKernelGGL((k_linearize<false, false>), dim3(h->nPart), dim3(TILE_OBS), lds, h->stream, a);
      else if (dump)
        hipLaunchKernelGGL((k_linearize2<true, false>), dim3(h->nPart), dim3(TILE_OBS), lds, h->stream, a);
      else
        hipLaunchKernelGGL((k_linearize2<false, false>), dim3(h->nPart), dim3(TILE_OBS), lds, h->stream, a);
      int nslab = h->nPart;
      if (h->nLong) {  // their camera sums: one more slab
        double *slab = h->campart + (size_t)h->nPart * d.nC * CAM_ACC;
        PSBA_HIP(h, hipMemsetAsync(slab, 0, sizeof(double) * CAM_ACC * (size_t)d.nC, h->stream));
        if (dump)
          hipLaunchKernelGGL((k_linearize_long<true, false>), dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts, slab);
        else
          hipLaunchKernelGGL((k_linearize_long<false, false>), dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts, slab);
        nslab++;
      }
      hipLaunchKernelGGL(k_cam_reduce, dim3(d.nC), dim3(1024), 0, h->stream, h->campart, nslab, d.nC, h->coeff,
                         h->coeff_g, Uo, gao);
    }
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_residual(psba_ctx *h, int which, double *ex_out_dev) {
  if (h->cnp != 6) return ex_out_dev ? fail(h, PSBA_E_STATE, "the sba_func.h mirror is six-parameter only") : launch_residual_fk(h, which);
  const Dims &d = h->d;
  const int set = which == PSBA_PARAMS_NEW ? 1 - h->cur : h->cur;
  PSBA_HIP(h, hipMemsetAsync(h->scal + SC_COST, 0, sizeof(double), h->stream));
  int grid = (d.nO + 255) / 256;
  if (grid > 256) grid = 256;  // one atomic request per workgroup; same-address atomics serialise
  {
    ProfScope ps(h, PSBA_K_RESIDUAL);
    hipLaunchKernelGGL(k_residual, dim3(grid), dim3(256), 0, h->stream, h->camconst, h->cams[set],
                       h->pts[set], h->impts, h->iidx, h->jidx, d.nO, ex_out_dev,
                       h->scal + SC_COST);
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_max_diag(psba_ctx *h) {
  if (h->cnp != 6) return launch_max_diag_fk(h);
  PSBA_HIP(h, hipMemsetAsync(h->scal + SC_MAXDIAG, 0, sizeof(double), h->stream));
  int grid = (h->d.nP + 255) / 256;
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(k_max_diag, dim3(grid), dim3(256), 0, h->stream, h->U, h->PV, h->d.nC, h->d.nP,
                     h->scal + SC_MAXDIAG);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
