// kernels_schur.hip -- K2: damping + V^-1 + Y = W V^-1 + S = U* - Y W^T + e_a = g_a - Y g_b.
//
// Replaces kern_update_UV, kern_compute_Vinv, kern_compute_Yblks, kern_compute_S,
// kern_compute_ea and kern_restore_UVdiag (reference CL_files/update_UV.cl:5-31,
// compute_Vinv.cl:6-90, compute_Yblks.cl:6-39, compute_S.cl:6-78, compute_ea.cl:6-37,
// restore_UVdiag.cl:2-25) and their wrappers PSBA/sba_func.cpp:624-995.
//
// mu is added in registers (U, V are never modified, so nothing is restored); V^-1 and Y
// are never written to HBM.  The kernels, in launch order:
//   k_schur_lds     one product Y_a W_b^T (b <= a, same point) per thread from the static
//                   schedule of schur_plan.cpp, accumulated in LDS partitions of the lower block
//                   triangle of S (cameras ascend inside a point), flushed as one slab per
//                   workgroup;
//   k_schur_reduce  sums the slabs, folds in U + mu I and g_a, and writes the padded S (both
//                   block triangles) and the e_a row -- or, with a communicator, the packed
//                   [tril(S) | e_a] that is all-reduced and then scattered by k_schur_expand;
//                   one extra workgroup factors the first 32x32 diagonal block (chol_factor32.h);
//   k_schur_atomic + k_schur_finalize   the first-generation path (global fp64 atomics straight
//                   into S), kept as the fallback for camera counts the LDS schedule cannot hold.
#include "camera_model.h"
#include <cstdlib>

#include "schur_common.h"
#include "schur_lds_args.h"

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
    if (nobs > TILE_OBS) continue;  // a point seen by more cameras than a tile holds: k_schur_long's
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

// (v1 path) a point seen by more than TILE_OBS cameras (no such limit in compute_S.cl:39-52): one
// workgroup, every thread the products Y_a W_b^T, b <= a, of its observations a, straight into S
// and e_a with global fp64 atomics.
template <bool DUMP>
__global__ __launch_bounds__(TILE_OBS) void k_schur_long(SchurArgs p, const int *long_pts) {
  const int tid = threadIdx.x, i = long_pts[blockIdx.x];
  const int o0 = p.ptr[i], o1 = p.ptr[i + 1];
  const double *pv = p.PV + 9 * (size_t)i;
  double v[6], vi[6];
#pragma unroll
  for (int k = 0; k < 6; k++) v[k] = pv[k];
  v[0] += p.mu;
  v[3] += p.mu;
  v[5] += p.mu;
  if (sym3_inverse(v, vi)) p.status[0] = p.try_id;
  const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
  if (DUMP && tid == 0) {
    double *o = p.dbg_Vinv + 9 * (size_t)i;
    o[0] = vi[0]; o[1] = vi[1]; o[2] = vi[2];
    o[3] = vi[1]; o[4] = vi[3]; o[5] = vi[4];
    o[6] = vi[2]; o[7] = vi[4]; o[8] = vi[5];
  }
  for (int a = o0 + tid; a < o1; a += TILE_OBS) {
    const int ja = p.jidx[a];
    const double *w = p.W + 18 * (size_t)a;
    double Y[18];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
      Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
      Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
      Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      atomicAdd(&p.ea[6 * ja + r], -(Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2));
    }
    if (DUMP) {
#pragma unroll
      for (int k = 0; k < 18; k++) p.dbg_Y[18 * (size_t)a + k] = Y[k];
    }
    for (int b = o0; b <= a; b++) {
      double *Sblk = p.S + (size_t)(6 * ja) * p.ld + 6 * p.jidx[b];
      const double *wb = p.W + 18 * (size_t)b;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const double w0 = wb[3 * c], w1 = wb[3 * c + 1], w2 = wb[3 * c + 2];
#pragma unroll
        for (int r = 0; r < 6; r++)
          atomicAdd(&Sblk[(size_t)r * p.ld + c], -(Y[3 * r] * w0 + Y[3 * r + 1] * w1 + Y[3 * r + 2] * w2));
      }
    }
  }
}

// v4: the lower block triangle of S is split into groups of blocks (whole camera rows, or ranges
// of the canonical block order for many cameras) whose 6x6 accumulators fit in LDS, and a static schedule built at upload time (schur_plan.cpp) gives every workgroup an
// equally long list of self-contained 64-bit work items, one per product Y_a W_b^T (b <= a,
// same point), point-major so that neighbouring lanes read the same W rows:
//   bits 0..23 a - obs0   24..45 i - pt0   46..53 a - b   54..63 block position in the partition
// One item per thread: all loads of the product are issued at once, V*^-1 and Y_a are formed
// in registers and the 6x6 product is added into the LDS partition with ds_add_f64.
//  * blocks sit at a stride of 37 doubles and the schedule deals items into rows of 16 lanes
//    whose block positions differ mod 16, so the 16 lanes that go through the LDS together hit
//    16 different bank pairs (measured 8 ticks per wave-instruction against 25 unscheduled);
//  * the self-product of an observation (a == b) is symmetric, so six of its upper-triangle
//    slots carry the e_a term  -Y_a g_b,i  instead: no separate e_a accumulators or atomics;
//  * the partition is written once, as plain stores, into the workgroup's slab;
//    k_schur_reduce sums the slabs of a group in a fixed order;
//  * workgroups that work on the same stretch of points (for different groups of blocks)
//    are mapped to the same XCD (blockIdx % 8) so that the re-reads of W are L2 hits.
// (SCHUR_THREADS, BLK_STRIDE and SchurLdsArgs: schur_lds_args.h)


// the a-side of a product: V*^-1 = (V_i + mu I)^-1, Y_a = -W_a V*^-1 (Y carries the sign of the product, so that
// neither the 36 values nor the e_a terms need a sign flip of their own) and e = Y_a g_b,i
template <bool DUMP>
__device__ __forceinline__ void schur_a_side(const SchurLdsArgs &p, int a, int i, double (&v)[6], double g0, double g1,
                                             double g2, const double (&w)[18], double (&Y)[18], double (&e)[6],
                                             bool dump_y) {
  double vi[6];
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
#pragma unroll
  for (int k = 0; k < 6; k++) vi[k] = -vi[k];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
    Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
    Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
    Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
    e[r] = Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2;
  }
  if (DUMP && dump_y) {
#pragma unroll
    for (int k = 0; k < 18; k++) p.dbg_Y[18 * (size_t)a + k] = -Y[k];
  }
}

// blk += Y W_b^T (6x6, ds_add_f64); the self-product's e_a terms ride in redundant upper-triangle slots (EA_SLOT)
__device__ __forceinline__ void schur_block_add(double *blk, const double (&Y)[18], const double (&wb)[18], bool self,
                                                const double (&e)[6]) {
  // a row of the block at a time: its six values are independent chains of three operations,
  // formed side by side (one after the other, every operation would wait for the one before)
#pragma unroll
  for (int r = 0; r < 6; r++) {
    double val[6];
#pragma unroll
    for (int c = 0; c < 6; c++) val[c] = Y[3 * r] * wb[3 * c];
#pragma unroll
    for (int c = 0; c < 6; c++) val[c] = fma(Y[3 * r + 1], wb[3 * c + 1], val[c]);
#pragma unroll
    for (int c = 0; c < 6; c++) val[c] = fma(Y[3 * r + 2], wb[3 * c + 2], val[c]);
#pragma unroll
    for (int c = 0; c < 6; c++) {
      if (r == 0 && c >= 1) val[c] = self ? e[c - 1] : val[c];
      if (r == 1 && c == 2) val[c] = self ? e[5] : val[c];
    }
#pragma unroll
    for (int c = 0; c < 6; c++) atomicAdd(&blk[6 * r + c], val[c]);
  }
}

// one item = one product Y_a W_b^T of one point; see the schedule above
template <bool DUMP>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_lds(SchurLdsArgs p) {
  extern __shared__ double sPart[];  // [nblk][37]
  const int tid = threadIdx.x;
  int w = blockIdx.x;
  if ((p.nWg & 7) == 0) w = (blockIdx.x & 7) * (p.nWg >> 3) + (blockIdx.x >> 3);
  const SchurWg wg = p.wg[w];
  for (int t = tid; t < BLK_STRIDE * wg.nblk; t += SCHUR_THREADS) sPart[t] = 0.0;
  __syncthreads();

#ifdef PSBA_BUILD_EXPERIMENTS
  // ---- pair items (round 4; measured slower, DESIGN 5d: experiments build, PSBA_SCHUR_PAIRS=1):
  // one observation a, its partners a - boff and a - boff + 1 (adjacent rows of W): the a-side (W_a, V*^-1, Y_a,
  // the e_a terms) is loaded and formed once for two products ----
  {
    const long long s1 = wg.itemD;
    unsigned long long item_next = (wg.item0 + tid < s1) ? p.items[wg.item0 + tid] : SCHUR_NULL_ITEM;
    for (long long t = wg.item0 + tid; t < s1; t += SCHUR_THREADS) {
      const unsigned long long item = item_next;
      if (t + SCHUR_THREADS < s1) item_next = p.items[t + SCHUR_THREADS];
      if (item == SCHUR_NULL_ITEM) continue;
      const int a = wg.obs0 + (int)(item & ((1u << PAIR_OBS_BITS) - 1));
      const int i = wg.pt0 + (int)((item >> PAIR_OBS_BITS) & ((1u << PAIR_PT_BITS) - 1));
      const int boff = (int)((item >> (PAIR_OBS_BITS + PAIR_PT_BITS)) & ((1u << ITEM_BOFF_BITS) - 1));
      const int pos = (int)((item >> (PAIR_OBS_BITS + PAIR_PT_BITS + ITEM_BOFF_BITS)) & ((1u << ITEM_POS_BITS) - 1));
      const int pos2 = (int)((item >> (PAIR_OBS_BITS + PAIR_PT_BITS + ITEM_BOFF_BITS + ITEM_POS_BITS)) & ((1u << ITEM_POS_BITS) - 1));
      const double *pv = p.PV + 9 * (size_t)i;
      const double2 *wa = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)a);
      const double2 *wb2 = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)(a - boff));
      double v[6], w[18], wb[18];
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] = pv[k];
      const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double2 q = wa[k];
        w[2 * k] = q.x;
        w[2 * k + 1] = q.y;
      }
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double2 q = wb2[k];
        wb[2 * k] = q.x;
        wb[2 * k + 1] = q.y;
      }
      double Y[18], e[6];
      schur_a_side<DUMP>(p, a, i, v, g0, g1, g2, w, Y, e, boff == 1);
      schur_block_add(sPart + BLK_STRIDE * pos, Y, wb, false, e);
      // the second partner's row takes the registers of the first (36 VGPRs more would spill at 4 waves per SIMD);
      // the compiler must not hoist these loads above the first product
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double2 q = wb2[9 + k];
        wb[2 * k] = q.x;
        wb[2 * k + 1] = q.y;
      }
      schur_block_add(sPart + BLK_STRIDE * pos2, Y, wb, boff == 1, e);
    }
  }
#endif
  const long long s1 = wg.item1;
  // the next item word is fetched a turn ahead: its latency would otherwise sit in front of the
  // record loads of every turn (a wave has only about six turns)
  unsigned long long item_next = (wg.itemD + tid < s1) ? p.items[wg.itemD + tid] : SCHUR_NULL_ITEM;
  for (long long t = wg.itemD + tid; t < s1; t += SCHUR_THREADS) {
    const unsigned long long item = item_next;
    if (t + SCHUR_THREADS < s1) item_next = p.items[t + SCHUR_THREADS];
    if (item == SCHUR_NULL_ITEM) continue;
    const int a = wg.obs0 + (int)(item & ((1u << ITEM_OBS_BITS) - 1));
    const int i = wg.pt0 + (int)((item >> ITEM_OBS_BITS) & ((1u << ITEM_PT_BITS) - 1));
    const int boff = (int)((item >> (ITEM_OBS_BITS + ITEM_PT_BITS)) & ((1u << ITEM_BOFF_BITS) - 1));
    const int pos = (int)(item >> (ITEM_OBS_BITS + ITEM_PT_BITS + ITEM_BOFF_BITS));
    // every address is known now: issue all loads of the product together
    const double *pv = p.PV + 9 * (size_t)i;
    const double2 *wa = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)a);
    const double2 *wb2 = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)(a - boff));
    double v[6], w[18], wb[18];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pv[k];
    const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double2 q = wa[k];
      w[2 * k] = q.x;
      w[2 * k + 1] = q.y;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double2 q = wb2[k];
      wb[2 * k] = q.x;
      wb[2 * k + 1] = q.y;
    }
    const bool self = boff == 0;
    double Y[18], e[6];
    schur_a_side<DUMP>(p, a, i, v, g0, g1, g2, w, Y, e, self);
    schur_block_add(sPart + BLK_STRIDE * pos, Y, wb, self, e);
  }
  __syncthreads();
  // two doubles per thread and step: 36 is even, so a pair never straddles a block, and the
  // slab offsets are multiples of 36 * 16 doubles: 16-byte stores
  double2 *slab = reinterpret_cast<double2 *>(p.slab + wg.slab_off);
  for (int t = tid; t < 18 * wg.nblk; t += SCHUR_THREADS) {
    const double *src = sPart + BLK_STRIDE * (t / 18) + 2 * (t % 18);
    slab[t] = make_double2(src[0], src[1]);
  }
  if (p.diag0 && tid < 21 * 36) {
    const int b = tid / 36, rc = tid % 36;
    if (p.diag_grp[b] == wg.group) {
      const double v = sPart[BLK_STRIDE * p.diag_pos[b] + rc];
      if (v != 0.0) atomicAdd(&p.diag0[tid], v);
    }
  }
}

// The same assembly for items in the RUNS layout (schur_plan.cpp): [turn][RUN_THREADS], a thread's consecutive
// items grouped into runs of one block position.  A run is summed in 36 registers and reaches the LDS once, at
// its end -- on clustered tracks (neighbouring points sharing camera sets, i.e. real reconstructions) that is one
// set of atomics per up to RUN_MAX products instead of one per product on ONE address.  512 threads: the 36
// accumulators take the kernel to ~190 registers, two waves per SIMD.
template <bool DUMP>
__global__ __launch_bounds__(RUN_THREADS) void k_schur_lds_runs(SchurLdsArgs p) {
  extern __shared__ double sPart[];  // [nblk][37]
  const int tid = threadIdx.x;
  int w = blockIdx.x;
  if ((p.nWg & 7) == 0) w = (blockIdx.x & 7) * (p.nWg >> 3) + (blockIdx.x >> 3);
  const SchurWg wg = p.wg[w];
  for (int t = tid; t < BLK_STRIDE * wg.nblk; t += RUN_THREADS) sPart[t] = 0.0;
  __syncthreads();
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; k++) acc[k] = 0.0;
  int cur = -1;
  unsigned long long item_next = (wg.item0 + tid < wg.item1) ? p.items[wg.item0 + tid] : SCHUR_NULL_ITEM;
  for (long long t = wg.item0 + tid; t < wg.item1; t += RUN_THREADS) {
    const unsigned long long item = item_next;
    if (t + RUN_THREADS < wg.item1) item_next = p.items[t + RUN_THREADS];
    if (item == SCHUR_NULL_ITEM) continue;  // (only behind a thread's last item)
    const int a = wg.obs0 + (int)(item & ((1u << ITEM_OBS_BITS) - 1));
    const int i = wg.pt0 + (int)((item >> ITEM_OBS_BITS) & ((1u << ITEM_PT_BITS) - 1));
    const int boff = (int)((item >> (ITEM_OBS_BITS + ITEM_PT_BITS)) & ((1u << ITEM_BOFF_BITS) - 1));
    const int pos = (int)(item >> (ITEM_OBS_BITS + ITEM_PT_BITS + ITEM_BOFF_BITS));
    const double *pv = p.PV + 9 * (size_t)i;
    const double2 *wa = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)a);
    const double2 *wb2 = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)(a - boff));
    double v[6], vi[6], wv[18], wb[18];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pv[k];
    const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double2 q = wa[k];
      wv[2 * k] = q.x;
      wv[2 * k + 1] = q.y;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double2 q = wb2[k];
      wb[2 * k] = q.x;
      wb[2 * k + 1] = q.y;
    }
    if (pos != cur) {  // the run of the previous block is over: its sums into the partition
      if (cur >= 0) {
        double *blk = sPart + BLK_STRIDE * cur;
#pragma unroll
        for (int k = 0; k < 36; k++) atomicAdd(&blk[k], acc[k]);
      }
#pragma unroll
      for (int k = 0; k < 36; k++) acc[k] = 0.0;
      cur = pos;
    }
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
    const bool self = boff == 0;
#pragma unroll
    for (int k = 0; k < 6; k++) vi[k] = -vi[k];  // Y carries the sign of the product
    double Y[18], e[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double w0 = wv[3 * r], w1 = wv[3 * r + 1], w2 = wv[3 * r + 2];
      Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
      Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
      Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      e[r] = Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2;
    }
    if (DUMP && self) {
#pragma unroll
      for (int k = 0; k < 18; k++) p.dbg_Y[18 * (size_t)a + k] = -Y[k];
    }
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double val = fma(Y[3 * r + 2], wb[3 * c + 2], fma(Y[3 * r + 1], wb[3 * c + 1], Y[3 * r] * wb[3 * c]));
        // the self-product's e_a terms ride in redundant upper-triangle slots (EA_SLOT)
        if (r == 0 && c >= 1) val = self ? e[c - 1] : val;
        if (r == 1 && c == 2) val = self ? e[5] : val;
        acc[6 * r + c] += val;
      }
  }
  if (cur >= 0) {
    double *blk = sPart + BLK_STRIDE * cur;
#pragma unroll
    for (int k = 0; k < 36; k++) atomicAdd(&blk[k], acc[k]);
  }
  __syncthreads();
  double2 *slab = reinterpret_cast<double2 *>(p.slab + wg.slab_off);
  for (int t = tid; t < 18 * wg.nblk; t += RUN_THREADS) {
    const double *src = sPart + BLK_STRIDE * (t / 18) + 2 * (t % 18);
    slab[t] = make_double2(src[0], src[1]);
  }
  if (p.diag0) {
    for (int t = tid; t < 21 * 36; t += RUN_THREADS) {
      const int b = t / 36, rc = t % 36;
      if (p.diag_grp[b] == wg.group) {
        const double v = sPart[BLK_STRIDE * p.diag_pos[b] + rc];
        if (v != 0.0) atomicAdd(&p.diag0[t], v);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_schur_reduce(SchurReduceArgs p) {
  __shared__ double2 sAcc[8][32];
  __shared__ Factor32Lds sF;
  if ((int)blockIdx.x == p.diag_wg) {
    reduce_first_diag_block(p, sF);
    return;
  }
  // the accumulators of this try's K3 (||dp||^2, gain denominator, new cost, ||p+dp||^2), and
  // the try stamp the (graph-replayed, hence argument-frozen) Cholesky kernels write on failure
  if (blockIdx.x == 0 && threadIdx.x < 4 * SC_NPART) p.scal[SC_PART + threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 64) p.status[3] = p.try_id;
  if (!p.packed)
    write_padding(p.S, p.nA, p.n32, p.pad_one, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                  (size_t)(gridDim.x - (p.diag_wg >= 0 ? 1 : 0)) * blockDim.x);
  // a chunk is 32 consecutive pairs of doubles (512 bytes of every slab of its group: 36 is even,
  // so a pair never straddles a block, and a partition is a multiple of 16 blocks = 9 x 32 pairs,
  // so a chunk never straddles groups).  A workgroup takes `chunks` (1, 2, 4 or 8) chunks with
  // 8 / chunks interleaved slab sequences each: one chunk with eight sequences where a group has
  // many slabs, eight chunks with one where every group has a single slab (many cameras: the kernel
  // then is a permutation, and 512-byte workgroups were ~300 k launches of almost idle threads).
  // 16-byte loads, four of them in flight per thread.
  const int o = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int L = 8 / p.chunks, slot = q / L, sub = q % L;
  const long long chunk = (long long)blockIdx.x * p.chunks + slot;
  double2 acc = make_double2(0.0, 0.0);
  if (chunk < p.nchunks) {
    const ReduceGroup gr = p.grp16[chunk / 9];  // 16 positions = 9 chunks; one 32-byte load, no look-up chain
    const int n = gr.nwg;
    const size_t stride = (size_t)18 * gr.nblk;  // in pairs
    const double2 *s = reinterpret_cast<const double2 *>(p.slab + gr.slab) + (chunk * 32 + o - 18LL * gr.pos0);
    int k = sub;
    for (; k + 3 * L < n; k += 4 * L) {
      const double2 x0 = s[(size_t)k * stride];
      const double2 x1 = s[(size_t)(k + L) * stride];
      const double2 x2 = s[(size_t)(k + 2 * L) * stride];
      const double2 x3 = s[(size_t)(k + 3 * L) * stride];
      acc.x += x0.x; acc.y += x0.y;
      acc.x += x1.x; acc.y += x1.y;
      acc.x += x2.x; acc.y += x2.y;
      acc.x += x3.x; acc.y += x3.y;
    }
    for (; k < n; k += L) {
      const double2 x = s[(size_t)k * stride];
      acc.x += x.x; acc.y += x.y;
    }
  }
  if (L == 1) {  // every slot has summed its chunk alone: no exchange
    if (chunk < p.nchunks) {
      reduce_scatter(p, 2 * (chunk * 32 + o), acc.x);
      reduce_scatter(p, 2 * (chunk * 32 + o) + 1, acc.y);
    }
    return;
  }
  sAcc[q][o] = acc;
  __syncthreads();
  if ((int)threadIdx.x >= 64 * p.chunks) return;
  // one thread per double from here
  const int fslot = threadIdx.x >> 6, po = (threadIdx.x & 63) >> 1, hi = threadIdx.x & 1;
  const long long fchunk = (long long)blockIdx.x * p.chunks + fslot;
  if (fchunk >= p.nchunks) return;
  double sum = 0.0;
  for (int t = 0; t < L; t++) sum += hi ? sAcc[fslot * L + t][po].y : sAcc[fslot * L + t][po].x;
  reduce_scatter(p, 2 * (fchunk * 32 + po) + hi, sum);
}

// after the all-reduce of the packed sums: scatter them into the padded row-major S (both block
// triangles) and the e_a row, and write the identity padding.  One extra workgroup (diag_wg)
// picks the first 32x32 diagonal block out of the packed sums and factors it -- as on a single
// rank, the first step of the Cholesky chain runs beside a kernel that is needed anyway.
struct SchurExpandArgs {
  const double *packed;
  double *S, *ea, *Lx, *linv;
  int *status;
  int total, nA, n32, try_id, diag_wg;
  int diag_slot[21];  // 36 * (tri(j) + k): first double of block (j, k), j <= 5, in the packed sums
};

__global__ __launch_bounds__(256) void k_schur_expand(SchurExpandArgs p) {
  __shared__ Factor32Lds sF;
  if ((int)blockIdx.x == p.diag_wg) {
    const int tid = threadIdx.x, nC = p.nA / 6;
    if (tid < 4) sF.flag[tid] = 0;
    if (tid == 4) sF.fail = 0;
    for (int t = tid; t < GB * GB; t += 256) sF.D[t / GB][t % GB] = (t / GB == t % GB) ? 1.0 : 0.0;  // padding
    __syncthreads();
    for (int e = tid; e < 21 * 36; e += 256) {
      const int blk = e / 36, rc = e % 36, r = rc / 6, c = rc % 6;
      const int j = blk < 1 ? 0 : blk < 3 ? 1 : blk < 6 ? 2 : blk < 10 ? 3 : blk < 15 ? 4 : 5, k = blk - tri(j);
      const int row = 6 * j + r, col = 6 * k + c;
      if (j >= nC || row >= GB || col >= GB || (j == k && c > r)) continue;
      sF.D[row][col] = p.packed[p.diag_slot[blk] + rc];  // U, mu already folded in
    }
    __syncthreads();
    factor32(sF, tid);
    for (int t = tid; t < GB * GB; t += 256) {
      const int r = t / GB, c = t % GB;
      p.Lx[(size_t)r * p.n32 + c] = f32_L(sF, r, c);
      p.linv[t] = f32_Linv(sF, r, c);
    }
    if (tid == 0 && sF.fail) p.status[1] = p.try_id;
    return;
  }
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  write_padding(p.S, p.nA, p.n32, 1.0, gtid, (size_t)(gridDim.x - (p.diag_wg >= 0 ? 1 : 0)) * blockDim.x);
  const int e = (int)gtid;
  if (e >= p.total) return;
  const int blk = e / 36;
  int j = (int)((sqrt(8.0 * (double)blk + 1.0) - 1.0) * 0.5);
  while ((j + 1) * (j + 2) / 2 <= blk) j++;
  while (j * (j + 1) / 2 > blk) j--;
  const int jb = blk - j * (j + 1) / 2, rc = e % 36, r = rc / 6, c = rc % 6;
  const double v = p.packed[e];
  if (j == jb && c > r) {
#pragma unroll
    for (int t = 0; t < 6; t++)
      if (rc == EA_SLOT[t]) p.ea[6 * j + t] = v;
    return;
  }
  p.S[(size_t)(6 * jb + c) * p.n32 + 6 * j + r] = v;
  p.S[(size_t)(6 * j + r) * p.n32 + 6 * jb + c] = v;
}

int launch_schur_expand(psba_ctx *h) {
  SchurExpandArgs a;
  a.packed = h->redp;
  a.S = h->red;
  a.ea = h->red + (size_t)h->n32 * h->n32;
  a.Lx = h->chol_L;
  a.linv = h->chol_ws;
  a.status = h->status;
  a.total = (int)h->packed_doubles;
  a.nA = h->d.nA;
  a.n32 = h->n32;
  a.try_id = h->try_id;
  const int grid = (a.total + 255) / 256;
  const bool fuse = !h->diag_done && !getenv("PSBA_CHOL_SEPARATE_DIAG");
  a.diag_wg = fuse ? grid : -1;
  for (int j = 0, b = 0; j < 6; j++)
    for (int k = 0; k <= j; k++, b++) a.diag_slot[b] = j < h->d.nC ? 36 * (j * (j + 1) / 2 + k) : 0;
  hipLaunchKernelGGL(k_schur_expand, dim3(grid + (fuse ? 1 : 0)), dim3(256), 0, h->stream, a);
  PSBA_HIP(h, hipGetLastError());
  if (fuse) h->diag_done = true;
  return PSBA_OK;
}

// (v1 path) S += blockdiag(U) + mu_add I on the lower block triangle, mirror to the upper
// block triangle, ea += g_a.  mu_add is mu on rank 0 and 0 elsewhere so that the all-reduce
// of the per-rank contributions adds mu exactly once.
__global__ __launch_bounds__(256) void k_schur_finalize(double *S, double *ea, const double *U,
                                                        const double *ga, double mu_add, int nA,
                                                        int n32, double pad_one, double *scal,
                                                        int *status, int try_id) {
  if (blockIdx.x == 0 && threadIdx.x < 4 * SC_NPART) scal[SC_PART + threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 64) status[3] = try_id;
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

// ---- owner route: many cameras -----------------------------------------------------------
// One thread per unit (block (j, k) of the lower block triangle x a segment of its product list,
// see OwnerPlanHost in psba_internal.h).  The thread walks its products (a, b): V*^-1 and
// Y_a = W_a V*^-1 in registers, acc -= Y_a W_b^T; a diagonal block's unit also sums
// e_a -= Y_a g_b,i.  No scatter: the 36 sums stay in registers and leave once -- plain stores when
// the unit is the block's only one, fp64 atomic adds into the zeroed S otherwise.  The reference
// gathers the same products per output scalar through comm3DIdx (CL_files/compute_S.cl:24-52).
struct SchurOwnerArgs {
  const double *W, *PV;
  const int *iidx;
  const int2 *prod;
  const OwnerWave *waves;
  const OwnerUnit *units;
  double *S, *ea;
  int *status;
  double mu;
  int nWaves, ld, try_id;
  int bsr;  // 1: S is the block-sparse list (36 doubles per block, unit.slot), e_a a vector of its own
};

__global__ __launch_bounds__(256) void k_schur_owner(SchurOwnerArgs p) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= p.nWaves) return;
  const OwnerWave wv = p.waves[w];
  const OwnerUnit un = p.units[(size_t)w * 64 + lane];
  const bool diag = un.j == un.k;
  double acc[36], ea[6];
#pragma unroll
  for (int t = 0; t < 36; t++) acc[t] = 0.0;
#pragma unroll
  for (int t = 0; t < 6; t++) ea[t] = 0.0;
  const int2 *pr = p.prod + (size_t)wv.row0 * 64 + lane;
  int2 ab = wv.len > 0 ? pr[0] : make_int2(-1, -1);
  for (int t = 0; t < wv.len; t++) {
    const int2 nxt = t + 1 < wv.len ? pr[(size_t)(t + 1) * 64] : make_int2(-1, -1);
    if (ab.x >= 0) {
      const int i = p.iidx[ab.x];
      const double *pv = p.PV + 9 * (size_t)i;
      const double2 *wa = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)ab.x);
      const double2 *wb2 = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)ab.y);
      double v[6], vi[6], wr[18], wb[18];
#pragma unroll
      for (int q = 0; q < 6; q++) v[q] = pv[q];
      const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
#pragma unroll
      for (int q = 0; q < 9; q++) {
        const double2 x = wa[q], y = wb2[q];
        wr[2 * q] = x.x;
        wr[2 * q + 1] = x.y;
        wb[2 * q] = y.x;
        wb[2 * q + 1] = y.y;
      }
      v[0] += p.mu;
      v[3] += p.mu;
      v[5] += p.mu;
      if (sym3_inverse(v, vi)) p.status[0] = p.try_id;
      double Y[18];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const double w0 = wr[3 * r], w1 = wr[3 * r + 1], w2 = wr[3 * r + 2];
        Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      }
      if (diag) {
#pragma unroll
        for (int r = 0; r < 6; r++) ea[r] -= Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2;
      }
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++)
          acc[6 * r + c] -= Y[3 * r] * wb[3 * c] + Y[3 * r + 1] * wb[3 * c + 1] + Y[3 * r + 2] * wb[3 * c + 2];
    }
    ab = nxt;
  }
  if (un.slot < 0) return;  // idle lane of the last wave
  const int ldb = p.bsr ? 6 : p.ld;
  double *Sb = p.bsr ? p.S + 36 * (size_t)un.slot : p.S + (size_t)(6 * un.j) * p.ld + 6 * un.k;
  if (un.multi) {
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) atomicAdd(&Sb[(size_t)r * ldb + c], acc[6 * r + c]);
  } else {
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) Sb[(size_t)r * ldb + c] = acc[6 * r + c];
  }
  if (diag) {
#pragma unroll
    for (int r = 0; r < 6; r++) atomicAdd(&p.ea[6 * un.j + r], ea[r]);
  }
}

static int launch_schur_lds(psba_ctx *h, double mu, bool dump) {
  const Dims &d = h->d;
  SchurLdsArgs a;
  a.W = h->W;
  a.PV = h->PV;
  a.wg = h->wg;
  a.items = h->items;
  a.slab = h->slab;
  a.status = h->status;
  a.dbg_Y = h->dbg_Y;
  a.dbg_Vinv = h->dbg_Vinv;
  a.mu = mu;
  a.nWg = h->nWg;
  a.try_id = h->try_id;
  // the first diagonal block can be factored beside the S-reduce only if S is complete on this rank
  // (PSBA_SCHUR_NO_FLUSH_DIAG: test hook -- a single-rank communicator then takes the multi-rank
  // route, where k_schur_expand factors the block after the all-reduce)
  const bool fuse_diag = h->nranks == 1 && h->diag0 && !getenv("PSBA_CHOL_SEPARATE_DIAG") &&
                         !getenv("PSBA_SCHUR_NO_FLUSH_DIAG");
  a.diag0 = fuse_diag ? h->diag0 : nullptr;
  for (int b = 0; b < 21; b++) {
    a.diag_grp[b] = h->h_diaggrp[b];
    a.diag_pos[b] = h->h_diagpos[b];
  }
  SchurReduceArgs r;
  r.slab = h->slab;
  r.U = h->U;
  r.ga = h->ga;
  r.posblock = h->posblock;
  r.S = h->red;
  r.ea = h->red + (size_t)h->n32 * h->n32;
  r.scal = h->scal;
  r.status = h->status;
  r.mu_add = h->rank == 0 ? mu : 0.0;
  r.pad_one = h->rank == 0 ? 1.0 : 0.0;
  r.nA = d.nA;
  r.n32 = h->n32;
  r.nGroups = h->nGroups;
  r.try_id = h->try_id;
  r.diag0 = h->diag0;
  // (PSBA_SCHUR_PACKED: test hook -- a handle with a rank layout but no communicator takes the
  // packed route too, and psba_get/set_reduce_buffer then move the packed sums)
  r.packed = (h->comm || (h->nranks > 1 && getenv("PSBA_SCHUR_PACKED"))) ? h->redp : nullptr;
  h->packed_pending = r.packed != nullptr;
  r.Lx = h->chol_L;
  r.linv = h->chol_ws;
  r.ring_copies = 0;
  r.ring_stride = 0;
  int worst = 0;
  long long npos = 0;
  for (int g = 0; g < h->nGroups; g++) {
    npos += h->gnblk[g];
    if (h->gnblk[g] > worst) worst = h->gnblk[g];
  }
  const size_t lds = sizeof(double) * BLK_STRIDE * (size_t)worst;
  int most = 1;
  for (int g = 0; g < h->nGroups; g++) most = h->gnwg[g] > most ? h->gnwg[g] : most;
  r.grp16 = h->gtab;
  // up to four slabs: a thread sums them alone (8 chunks, no exchange); else 4 chunks x 2 sequences
  // (venice-shaped, ~85 slabs per group: 10.2 / 9.0 / 8.8 / 10.8 us with 1 / 2 / 4 / 8 chunks)
  r.chunks = most > 4 ? 4 : 8;
  if (const char *e = getenv("PSBA_REDUCE_CHUNKS")) r.chunks = atoi(e) == 2 || atoi(e) == 4 || atoi(e) == 8 ? atoi(e) : 1;  // development knob
  r.nchunks = 36 * npos / 64;  // partitions are multiples of 16 blocks = 9 x 64 doubles
  const int rgrid = (int)((r.nchunks + r.chunks - 1) / r.chunks);
  r.diag_wg = fuse_diag ? rgrid : -1;
  // timing classes: PSBA_K_SCHUR alone = one span over both kernels (S exists only after the
  // reduce: that pair is the graded kernel); with PSBA_K_SCHUR_REDUCE also on, each kernel by itself
  const bool pair = (h->prof & (1u << PSBA_K_SCHUR)) && !(h->prof & (1u << PSBA_K_SCHUR_REDUCE));
  {
    ProfScope pp(h, pair ? PSBA_K_SCHUR : -1);
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR);
      const char *m = getenv("PSBA_SCHUR_MODE");
      const int mode = m ? atoi(m) : 0;
      const dim3 G(h->nWg), B(SCHUR_THREADS);
      bool launched = false;
      if (h->schur_runs) {  // clustered tracks: items in the runs layout
        const dim3 Br(RUN_THREADS);
        if (dump)
          hipLaunchKernelGGL((k_schur_lds_runs<true>), G, Br, lds, h->stream, a);
        else
          hipLaunchKernelGGL((k_schur_lds_runs<false>), G, Br, lds, h->stream, a);
        launched = true;
      }
#ifdef PSBA_BUILD_EXPERIMENTS
      if (!launched && !dump && mode > 0 && !h->schur_pairs) launched = launch_schur_lds_mode(mode, G, B, lds, h->stream, a);  // kernels_schur_modes.hip
#endif
      (void)mode;
      if (launched) {
      } else if (dump)
        hipLaunchKernelGGL((k_schur_lds<true>), G, B, lds, h->stream, a);
      else
        hipLaunchKernelGGL((k_schur_lds<false>), G, B, lds, h->stream, a);
    }
    PSBA_HIP(h, hipGetLastError());
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR_REDUCE);
      hipLaunchKernelGGL(k_schur_reduce, dim3(rgrid + (r.diag_wg >= 0 ? 1 : 0)), dim3(256), 0, h->stream, r);
    }
  }
  h->diag_done = r.diag_wg >= 0;
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

// block-sparse mode (psba_set_solver PSBA_SOLVER_PCG): the owner route writes the blocks that exist,
// nothing dense is allocated; k_bsr_finalize adds U + mu I and g_a
static int launch_schur_sparse(psba_ctx *h, double mu) {
  PSBA_HIP(h, hipMemsetAsync(h->bs_val, 0, sizeof(double) * ((size_t)36 * h->bs_nblk + h->d.nA), h->stream));
  SchurOwnerArgs o;
  o.W = h->W;
  o.PV = h->PV;
  o.iidx = h->iidx;
  o.prod = h->own_prod;
  o.waves = h->own_waves;
  o.units = h->own_units;
  o.S = h->bs_val;
  o.ea = h->bs_ea;
  o.status = h->status;
  o.mu = mu;
  o.nWaves = h->own_nwaves;
  o.ld = 6;
  o.try_id = h->try_id;
  o.bsr = 1;
  const bool pair = (h->prof & (1u << PSBA_K_SCHUR)) && !(h->prof & (1u << PSBA_K_SCHUR_REDUCE));
  {
    ProfScope pp(h, pair ? PSBA_K_SCHUR : -1);
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR);
      hipLaunchKernelGGL(k_schur_owner, dim3((h->own_nwaves + 3) / 4), dim3(256), 0, h->stream, o);
    }
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR_REDUCE);
      const int rc = launch_bsr_finalize(h, mu);
      if (rc != PSBA_OK) return rc;
    }
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

int launch_schur(psba_ctx *h, double mu, bool dump) {
  if (h->cnp != 6) return dump ? fail(h, PSBA_E_STATE, "the sba_func.h mirror is six-parameter only") : launch_schur_fk(h, mu);
  h->try_id++;
  h->diag_done = false;
  h->packed_pending = false;  // status words are generation stamps: nothing to zero
  if (h->solver == PSBA_SOLVER_PCG) {
    if (dump) return fail(h, PSBA_E_INVALID, "the sba_func.h mirror verbs need the dense S: PSBA_SOLVER_DENSE");
    return launch_schur_sparse(h, mu);
  }
#ifdef PSBA_BUILD_EXPERIMENTS
  if (h->ring_nWg > 0 && !getenv("PSBA_SCHUR_ATOMIC")) return launch_schur_ring(h, mu, dump);
#endif
  if (h->nGroups > 0 && !getenv("PSBA_SCHUR_ATOMIC")) {
    if (!h->lds_attr_set) {
      const int dyn = 163840 - 256;  // allow the full 160 KiB of LDS for the partition
      const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds<true>, attr, dyn));
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds<false>, attr, dyn));
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds_runs<true>, attr, dyn));
      PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_lds_runs<false>, attr, dyn));
      h->lds_attr_set = true;
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
  a.nTiles = d.nTilesAll;  // (this kernel walks the contiguous partition and skips the long points' tiles itself)
  a.try_id = h->try_id;
  PSBA_HIP(h, hipMemsetAsync(h->red, 0, sizeof(double) * (size_t)(h->n32 + 1) * h->n32, h->stream));
  const bool owner = h->own_nwaves > 0 && !dump && !getenv("PSBA_SCHUR_ATOMIC");
  int grid = d.nTilesAll < 2048 ? d.nTilesAll : 2048;
  const size_t lds = sizeof(double) * (size_t)d.nA;  // e_a accumulators of a workgroup of the global-atomic kernel
  // (the owner route needs no dynamic LDS: the limit below is the global-atomic kernel's alone)
  if (!owner && lds > 100 * 1024)
    return fail(h, PSBA_E_INVALID, "global-atomic S assembly: 6 nCams = %d needs %zu B of LDS for e_a (> 100 KiB)",
                d.nA, lds);
  if (!owner && !h->atomic_attr_set && lds > 8 * 1024) {
    const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_atomic<true>, attr, 100 * 1024));
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_atomic<false>, attr, 100 * 1024));
    h->atomic_attr_set = true;
  }
  const bool pair = (h->prof & (1u << PSBA_K_SCHUR)) && !(h->prof & (1u << PSBA_K_SCHUR_REDUCE));
  const double mu_add = h->rank == 0 ? mu : 0.0;
  size_t n2 = (size_t)d.nA * d.nA;
  size_t fg = (n2 + 255) / 256;
  const int fgrid = fg > 4096 ? 4096 : (int)fg;
  {
    ProfScope pp(h, pair ? PSBA_K_SCHUR : -1);
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR);
      if (owner) {
        SchurOwnerArgs o;
        o.W = h->W;
        o.PV = h->PV;
        o.iidx = h->iidx;
        o.prod = h->own_prod;
        o.waves = h->own_waves;
        o.units = h->own_units;
        o.S = a.S;
        o.ea = a.ea;
        o.status = h->status;
        o.mu = mu;
        o.nWaves = h->own_nwaves;
        o.ld = h->n32;
        o.try_id = h->try_id;
        o.bsr = 0;
        hipLaunchKernelGGL(k_schur_owner, dim3((h->own_nwaves + 3) / 4), dim3(256), 0, h->stream, o);
      } else {
        if (dump)
          hipLaunchKernelGGL(k_schur_atomic<true>, dim3(grid), dim3(TILE_OBS), lds, h->stream, a);
        else
          hipLaunchKernelGGL(k_schur_atomic<false>, dim3(grid), dim3(TILE_OBS), lds, h->stream, a);
        if (h->nLong) {
          if (dump)
            hipLaunchKernelGGL(k_schur_long<true>, dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts);
          else
            hipLaunchKernelGGL(k_schur_long<false>, dim3(h->nLong), dim3(TILE_OBS), 0, h->stream, a, h->long_pts);
        }
      }
    }
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR_REDUCE);
      hipLaunchKernelGGL(k_schur_finalize, dim3(fgrid), dim3(256), 0, h->stream, a.S, a.ea, h->U, h->ga,
                         mu_add, d.nA, h->n32, h->rank == 0 ? 1.0 : 0.0, h->scal, h->status, h->try_id);
    }
  }
  PSBA_HIP(h, hipGetLastError());
  return PSBA_OK;
}

}  // namespace psba
