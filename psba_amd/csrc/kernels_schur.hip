// kernels_schur.hip -- K2: damping + V^-1 + Y = W V^-1 + S = U* - Y W^T + e_a = g_a - Y g_b.
//
// Replaces kern_update_UV, kern_compute_Vinv, kern_compute_Yblks, kern_compute_S,
// kern_compute_ea and kern_restore_UVdiag (reference CL_files/update_UV.cl:5-31,
// compute_Vinv.cl:6-90, compute_Yblks.cl:6-39, compute_S.cl:6-78, compute_ea.cl:6-37,
// restore_UVdiag.cl:2-25) and their wrappers PSBA/sba_func.cpp:624-995.
//
// mu is added in registers (U, V are never modified, so nothing is restored); V^-1 and Y
// are never written to HBM.  Point-major: a tile of whole points is staged in LDS (its W
// blocks, V^-1, g_b), each thread owns one observation a, forms Y_a in registers and emits
// the products Y_a W_b^T for every observation b <= a of the same point, i.e. the lower
// block triangle of S (cameras ascend inside a point).  k_schur_finalize adds U + mu I and
// g_a and mirrors the upper block triangle.
#include "camera_model.h"
#include <cstdlib>

#include "psba_internal.h"

namespace psba {

struct SchurArgs {
  const double *W, *PV;
  const int *iidx, *jidx, *ptr, *tile_pt;
  double *S, *ea;          // accumulators (zeroed before launch): -sum Y W^T, -sum Y g_b
  int *status;
  double *dbg_Y, *dbg_Vinv;
  double mu;
  int nC, nA, nTiles;
  int ld;                  // row stride of S (n32)
  int try_id;
};

// v1: global fp64 atomics straight into S (lower block triangle).
template <bool DUMP>
__global__ __launch_bounds__(TILE_OBS) void k_schur_atomic(SchurArgs p) {
  __shared__ double sW[TILE_OBS * 18];
  __shared__ double sVi[TILE_OBS][9];  // V^-1 sym6 | g_b
  __shared__ int sJ[TILE_OBS];
  extern __shared__ double sEa[];      // [nA]
  const int tid = threadIdx.x;
  for (int t = tid; t < p.nA; t += TILE_OBS) sEa[t] = 0.0;

  for (int tile = blockIdx.x; tile < p.nTiles; tile += gridDim.x) {
    const int p0 = p.tile_pt[tile], p1 = p.tile_pt[tile + 1];
    const int o0 = p.ptr[p0], o1 = p.ptr[p1];
    const int nobs = o1 - o0;
    __syncthreads();
    // coalesced copy of the tile's W blocks
    {
      const double2 *src = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)o0);
      double2 *dst = reinterpret_cast<double2 *>(sW);
      for (int t = tid; t < nobs * 9; t += TILE_OBS) dst[t] = src[t];
    }
    if (tid < nobs) sJ[tid] = p.jidx[o0 + tid];
    if (p0 + tid < p1) {
      const double *pv = p.PV + 9 * (size_t)(p0 + tid);
      double v[6], vi[6];
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] = pv[k];
      v[0] += p.mu;
      v[3] += p.mu;
      v[5] += p.mu;
      if (sym3_inverse(v, vi)) p.status[0] = p.try_id;
#pragma unroll
      for (int k = 0; k < 6; k++) sVi[tid][k] = vi[k];
#pragma unroll
      for (int k = 0; k < 3; k++) sVi[tid][6 + k] = pv[6 + k];
      if (DUMP) {
        double *o = p.dbg_Vinv + 9 * (size_t)(p0 + tid);
        o[0] = vi[0]; o[1] = vi[1]; o[2] = vi[2];
        o[3] = vi[1]; o[4] = vi[3]; o[5] = vi[4];
        o[6] = vi[2]; o[7] = vi[4]; o[8] = vi[5];
      }
    }
    __syncthreads();
    if (tid < nobs) {
      const int a = o0 + tid;
      const int i = p.iidx[a];
      const int ja = sJ[tid];
      const double *vi = sVi[i - p0];
      const double i00 = vi[0], i01 = vi[1], i02 = vi[2], i11 = vi[3], i12 = vi[4], i22 = vi[5];
      const double g0 = vi[6], g1 = vi[7], g2 = vi[8];
      double Y[18];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const double w0 = sW[18 * tid + 3 * r], w1 = sW[18 * tid + 3 * r + 1],
                     w2 = sW[18 * tid + 3 * r + 2];
        Y[3 * r] = w0 * i00 + w1 * i01 + w2 * i02;
        Y[3 * r + 1] = w0 * i01 + w1 * i11 + w2 * i12;
        Y[3 * r + 2] = w0 * i02 + w1 * i12 + w2 * i22;
        atomicAdd(&sEa[6 * ja + r], -(Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2));
      }
      if (DUMP) {
#pragma unroll
        for (int k = 0; k < 18; k++) p.dbg_Y[18 * (size_t)a + k] = Y[k];
      }
      const int b0 = p.ptr[i] - o0;
      for (int b = b0; b <= tid; b++) {
        const int jb = sJ[b];
        double *Sblk = p.S + (size_t)(6 * ja) * p.ld + 6 * jb;
        const double *wb = sW + 18 * b;
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const double w0 = wb[3 * c], w1 = wb[3 * c + 1], w2 = wb[3 * c + 2];
#pragma unroll
          for (int r = 0; r < 6; r++)
            atomicAdd(&Sblk[(size_t)r * p.ld + c],
                      -(Y[3 * r] * w0 + Y[3 * r + 1] * w1 + Y[3 * r + 2] * w2));
        }
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < p.nA; t += TILE_OBS)
    if (sEa[t] != 0.0) atomicAdd(&p.ea[t], sEa[t]);
}

// v3: the lower block triangle of S is split into camera-row groups whose packed size fits in
// LDS; workgroup (g, chunk) walks the observations of its point chunk whose camera lies in
// group g (a host-built compacted list, so every lane has work), one observation a per
// thread: V*^-1 and Y_a in registers, then for every observation b <= a of the same point
// the 6x6 product Y_a W_b^T is added into the LDS partition with ds_add_f64.  Blocks are
// laid out with a stride of 37 doubles so that the 64 lanes of one atomic instruction (same
// entry of 64 different blocks) fall into different banks (a stride of 36 is a 4-way
// conflict: measured 32 vs 8 cycles per wave-instruction).  The partition is written once,
// as plain stores, into the chunk's slab; k_schur_reduce sums the slabs in chunk order.
// The G workgroups of one chunk are mapped to the same XCD (blockIdx % 8) so that the
// re-reads of the chunk's W blocks are L2 hits.
constexpr int SCHUR_THREADS = 1024;
constexpr int BLK_STRIDE = 37;

struct SchurLdsArgs {
  const double *W, *PV;
  const int *iidx, *jidx, *ptr, *gobs, *gstart, *chunk_obs0;
  double *slab;
  int *status;
  double *dbg_Y, *dbg_Vinv;
  double mu;
  int nC, nA, nGroups, nChunks, try_id;
  unsigned long long slabStride;  // packedN + nA
  unsigned long long packedN;
  int glo[MAX_GROUPS + 1];
};

__device__ __forceinline__ int tri(int j) { return j * (j + 1) / 2; }

// MODE is development instrumentation (ablation timing, PSBA_SCHUR_MODE): 0 = full kernel;
// 1 = products without the LDS atomics; 2 = no product loop; 4 = zero + flush only.
template <bool DUMP, int MODE>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_lds(SchurLdsArgs p) {
  extern __shared__ double sPart[];  // [nblk][37] rows [lo,hi) of the block triangle, then e_a rows
  const int tid = threadIdx.x;
  int g, chunk;
  if ((p.nChunks & 7) == 0) {
    chunk = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * p.nGroups));
    g = (blockIdx.x >> 3) % p.nGroups;
  } else {
    g = blockIdx.x % p.nGroups;
    chunk = blockIdx.x / p.nGroups;
  }
  const int lo = p.glo[g], hi = p.glo[g + 1];
  const int Tlo = tri(lo);
  const int nblk = tri(hi) - Tlo;
  double *sEa = sPart + BLK_STRIDE * nblk;
  const int nEa = 6 * (hi - lo);
  for (int t = tid; t < BLK_STRIDE * nblk + nEa; t += SCHUR_THREADS) sPart[t] = 0.0;
  __syncthreads();

  const int s0 = p.gstart[chunk * p.nGroups + g];
  const int s1 = (MODE == 4) ? s0 : p.gstart[chunk * p.nGroups + g + 1];
  double keep = 0.0;
  const int obs0 = p.chunk_obs0[chunk];
  for (int t = s0 + tid; t < s1; t += SCHUR_THREADS) {
    const unsigned item = (unsigned)p.gobs[t];
    const int a = obs0 + (int)(item >> 12);
    const int kfirst = (int)((item >> 4) & 255u), kcount = (int)(item & 15u);
    const int i = p.iidx[a], ja = p.jidx[a];
    const double *pv = p.PV + 9 * (size_t)i;
    double v[6], vi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pv[k];
    const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
    v[0] += p.mu;
    v[3] += p.mu;
    v[5] += p.mu;
    if (sym3_inverse(v, vi)) p.status[0] = p.try_id;
    if (DUMP) {
      double *o = p.dbg_Vinv + 9 * (size_t)i;
      o[0] = vi[0]; o[1] = vi[1]; o[2] = vi[2];
      o[3] = vi[1]; o[4] = vi[3]; o[5] = vi[4];
      o[6] = vi[2]; o[7] = vi[4]; o[8] = vi[5];
    }
    double Y[18];
    {
      const double2 *wa = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)a);
      double w[18];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double2 q = wa[k];
        w[2 * k] = q.x;
        w[2 * k + 1] = q.y;
      }
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
        Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
        if (kfirst == 0)  // the first run of an observation also carries its e_a term
          atomicAdd(&sEa[6 * (ja - lo) + r],
                    -(Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2));
      }
    }
    if (DUMP && kfirst == 0) {
#pragma unroll
      for (int k = 0; k < 18; k++) p.dbg_Y[18 * (size_t)a + k] = Y[k];
    }
    if (MODE == 2) {
      keep += Y[0] + Y[17];
      continue;
    }
    double *rowbase = sPart + BLK_STRIDE * (tri(ja) - Tlo);
    const int bfirst = p.ptr[i] + kfirst;
    for (int b = bfirst; b < bfirst + kcount; b++) {
      double *blk = rowbase + BLK_STRIDE * p.jidx[b];
      const double2 *wb2 = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)b);
      double wb[18];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double2 q = wb2[k];
        wb[2 * k] = q.x;
        wb[2 * k + 1] = q.y;
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const double w0 = wb[3 * c], w1 = wb[3 * c + 1], w2 = wb[3 * c + 2];
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const double val = -(Y[3 * r] * w0 + Y[3 * r + 1] * w1 + Y[3 * r + 2] * w2);
          if (MODE == 1)
            keep += val;
          else
            atomicAdd(&blk[6 * r + c], val);
        }
      }
    }
  }
  if (MODE != 0 && keep == 12345.678) sPart[0] = keep;
  __syncthreads();
  double *slab = p.slab + (size_t)chunk * p.slabStride;
  for (int t = tid; t < 36 * nblk; t += SCHUR_THREADS)
    slab[36 * (size_t)Tlo + t] = sPart[BLK_STRIDE * (t / 36) + t % 36];
  for (int t = tid; t < nEa; t += SCHUR_THREADS) slab[p.packedN + 6 * lo + t] = sEa[t];
}

// writes the padding of the reduce buffer: identity (pad_one = 1 on rank 0, else 0, so that
// the all-reduce over ranks yields exactly one) on the padded diagonal, zeros elsewhere in
// the padded rows / columns and in the padded part of the e_a row.
__device__ __forceinline__ void write_padding(double *S, int nA, int n32, double pad_one,
                                              size_t gtid, size_t gsize) {
  const int np = n32 - nA;
  if (np == 0) return;
  // padded columns of rows [0, n32] (incl. the e_a row), then padded rows' columns [0, nA)
  const size_t nColPad = (size_t)(n32 + 1) * np, nRowPad = (size_t)np * nA;
  for (size_t t = gtid; t < nColPad + nRowPad; t += gsize) {
    int r, c;
    if (t < nColPad) {
      r = (int)(t / np);
      c = nA + (int)(t % np);
    } else {
      const size_t u = t - nColPad;
      r = nA + (int)(u / nA);
      c = (int)(u % nA);
    }
    S[(size_t)r * n32 + c] = (r == c) ? pad_one : 0.0;
  }
}

// sums the chunk slabs (fixed order: four interleaved chunk sequences, then their sum), adds
// blockdiag(U) + mu_add I and g_a, and writes the padded row-major S (both block triangles)
// and the e_a row.  64 outputs x 4 chunk sequences per workgroup.
__global__ __launch_bounds__(256) void k_schur_reduce(const double *slab, int nChunks,
                                                      unsigned long long slabStride,
                                                      unsigned long long packedN, const double *U,
                                                      const double *ga, double mu_add, int nA,
                                                      int n32, double pad_one, double *S,
                                                      double *ea, double *scal, int *status,
                                                      int try_id) {
  __shared__ double sAcc[4][64];
  // the accumulators of this try's K3 (||dp||^2, gain denominator, new cost, ||p+dp||^2), and
  // the try stamp the (graph-replayed, hence argument-frozen) Cholesky kernels write on failure
  if (blockIdx.x == 0 && threadIdx.x < 4) scal[SC_DP_L2 + threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 4) status[3] = try_id;
  write_padding(S, nA, n32, pad_one, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                (size_t)gridDim.x * blockDim.x);
  const unsigned long long total = packedN + nA;
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const unsigned long long e = (unsigned long long)blockIdx.x * 64 + o;
  double acc = 0.0;
  if (e < total) {
    int c = q;
    for (; c + 12 < nChunks; c += 16) {
      const double x0 = slab[(size_t)c * slabStride + e];
      const double x1 = slab[(size_t)(c + 4) * slabStride + e];
      const double x2 = slab[(size_t)(c + 8) * slabStride + e];
      const double x3 = slab[(size_t)(c + 12) * slabStride + e];
      acc += x0;
      acc += x1;
      acc += x2;
      acc += x3;
    }
    for (; c < nChunks; c += 4) acc += slab[(size_t)c * slabStride + e];
  }
  sAcc[q][o] = acc;
  __syncthreads();
  if (q != 0 || e >= total) return;
  acc = ((sAcc[0][o] + sAcc[1][o]) + sAcc[2][o]) + sAcc[3][o];
  if (e >= packedN) {
    const int t = (int)(e - packedN);
    ea[t] = ga[t] + acc;
    return;
  }
  const int blk = (int)(e / 36), rc = (int)(e % 36);
  int j = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
  while (tri(j + 1) <= blk) j++;
  while (tri(j) > blk) j--;
  const int jb = blk - tri(j);
  const int r = rc / 6, c = rc % 6;
  if (j == jb) {
    acc += U[36 * j + rc];
    if (r == c) acc += mu_add;
  } else {
    S[(size_t)(6 * jb + c) * n32 + 6 * j + r] = acc;
  }
  S[(size_t)(6 * j + r) * n32 + 6 * jb + c] = acc;
}

// (v1 path) S += blockdiag(U) + mu_add I on the lower block triangle, mirror to the upper
// block triangle, ea += g_a.  mu_add is mu on rank 0 and 0 elsewhere so that the all-reduce
// of the per-rank contributions adds mu exactly once.
__global__ __launch_bounds__(256) void k_schur_finalize(double *S, double *ea, const double *U,
                                                        const double *ga, double mu_add, int nA,
                                                        int n32, double pad_one, double *scal,
                                                        int *status, int try_id) {
  if (blockIdx.x == 0 && threadIdx.x < 4) scal[SC_DP_L2 + threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 4) status[3] = try_id;
  const size_t n2 = (size_t)nA * nA;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t gsize = (size_t)gridDim.x * blockDim.x;
  for (size_t t = gtid; t < n2; t += gsize) {
    const int r = (int)(t / nA), c = (int)(t % nA);
    const int kb = r / 6, lb = c / 6;
    const size_t at = (size_t)r * n32 + c;
    if (lb > kb) {
      S[at] = S[(size_t)c * n32 + r];
    } else if (lb == kb) {
      double v = S[at] + U[36 * kb + 6 * (r - 6 * kb) + (c - 6 * lb)];
      if (r == c) v += mu_add;
      S[at] = v;
    }
  }
  for (size_t t = gtid; t < (size_t)nA; t += gsize) ea[t] += ga[t];
  write_padding(S, nA, n32, pad_one, gtid, gsize);
}

static int launch_schur_lds(psba_ctx *h, double mu, bool dump) {
  const Dims &d = h->d;
  SchurLdsArgs a;
  a.W = h->W;
  a.PV = h->PV;
  a.iidx = h->iidx;
  a.jidx = h->jidx;
  a.ptr = h->ptr;
  a.gobs = h->gobs;
  a.gstart = h->gstart;
  a.chunk_obs0 = h->chunk_obs0;
  a.slab = h->slab;
  a.status = h->status;
  a.dbg_Y = h->dbg_Y;
  a.dbg_Vinv = h->dbg_Vinv;
  a.mu = mu;
  a.nC = d.nC;
  a.nA = d.nA;
  a.nGroups = h->nGroups;
  a.nChunks = h->nChunks;
  a.try_id = h->try_id;
  a.packedN = h->packedN;
  a.slabStride = h->packedN + d.nA;
  size_t worst = 0;
  for (int g = 0; g <= h->nGroups; g++) a.glo[g] = h->glo[g];
  for (int g = 0; g < h->nGroups; g++) {
    const size_t lo = h->glo[g], hi = h->glo[g + 1];
    const size_t n = BLK_STRIDE * (hi * (hi + 1) / 2 - lo * (lo + 1) / 2) + 6 * (hi - lo);
    if (n > worst) worst = n;
  }
  const size_t lds = sizeof(double) * worst;
  const int grid = h->nGroups * h->nChunks;
  const double mu_add = h->rank == 0 ? mu : 0.0;
  const size_t total = h->packedN + d.nA;
  const int rgrid = (int)((total + 63) / 64);
  {
    ProfScope ps(h, PSBA_K_SCHUR);
    const char *m = getenv("PSBA_SCHUR_MODE");
    const int mode = m ? atoi(m) : 0;
    const dim3 G(grid), B(SCHUR_THREADS);
    if (dump)
      hipLaunchKernelGGL((k_schur_lds<true, 0>), G, B, lds, h->stream, a);
    else if (mode == 1)
      hipLaunchKernelGGL((k_schur_lds<false, 1>), G, B, lds, h->stream, a);
    else if (mode == 2)
      hipLaunchKernelGGL((k_schur_lds<false, 2>), G, B, lds, h->stream, a);
    else if (mode == 4)
      hipLaunchKernelGGL((k_schur_lds<false, 4>), G, B, lds, h->stream, a);
    else
      hipLaunchKernelGGL((k_schur_lds<false, 0>), G, B, lds, h->stream, a);
  }
  {
    ProfScope ps(h, PSBA_K_SCHUR_REDUCE);
    hipLaunchKernelGGL(k_schur_reduce, dim3(rgrid), dim3(256), 0, h->stream, h->slab, h->nChunks,
                       a.slabStride, a.packedN, h->U, h->ga, mu_add, d.nA, h->n32,
                       h->rank == 0 ? 1.0 : 0.0, h->red, h->red + (size_t)h->n32 * h->n32, h->scal, h->status,
                       h->try_id);
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

static bool g_lds_attr_set = false;

int launch_schur(psba_ctx *h, double mu, bool dump) {
  h->try_id++;  // status words are generation stamps: nothing to zero
  if (h->nGroups > 0 && !getenv("PSBA_SCHUR_ATOMIC")) {
    if (!g_lds_attr_set) {
      const int dyn = 163840 - 256;  // allow the full 160 KiB of LDS for the partition
      const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds<true, 0>, attr, dyn));
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds<false, 0>, attr, dyn));
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds<false, 1>, attr, dyn));
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds<false, 2>, attr, dyn));
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds<false, 4>, attr, dyn));
      g_lds_attr_set = true;
    }
    return launch_schur_lds(h, mu, dump);
  }
  const Dims &d = h->d;
  SchurArgs a;
  a.W = h->W;
  a.PV = h->PV;
  a.iidx = h->iidx;
  a.jidx = h->jidx;
  a.ptr = h->ptr;
  a.tile_pt = h->tile_pt;
  a.S = h->red;
  a.ea = h->red + (size_t)h->n32 * h->n32;
  a.ld = h->n32;
  a.status = h->status;
  a.dbg_Y = h->dbg_Y;
  a.dbg_Vinv = h->dbg_Vinv;
  a.mu = mu;
  a.nC = d.nC;
  a.nA = d.nA;
  a.nTiles = d.nTiles;
  a.try_id = h->try_id;
  PSBA_HIP(h, hipMemsetAsync(h->red, 0, sizeof(double) * (size_t)(h->n32 + 1) * h->n32, h->stream));
  int grid = d.nTiles < 2048 ? d.nTiles : 2048;
  const size_t lds = sizeof(double) * (size_t)d.nA;
  {
    ProfScope ps(h, PSBA_K_SCHUR);
    if (dump)
      hipLaunchKernelGGL(k_schur_atomic<true>, dim3(grid), dim3(TILE_OBS), lds, h->stream, a);
    else
      hipLaunchKernelGGL(k_schur_atomic<false>, dim3(grid), dim3(TILE_OBS), lds, h->stream, a);
  }
  const double mu_add = h->rank == 0 ? mu : 0.0;
  size_t n2 = (size_t)d.nA * d.nA;
  int fgrid = (int)((n2 + 255) / 256);
  if (fgrid > 1024) fgrid = 1024;
  hipLaunchKernelGGL(k_schur_finalize, dim3(fgrid), dim3(256), 0, h->stream, a.S, a.ea, h->U, h->ga,
                     mu_add, d.nA, h->n32, h->rank == 0 ? 1.0 : 0.0, h->scal, h->status, h->try_id);
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
