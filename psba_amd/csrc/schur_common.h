// schur_common.h -- device helpers shared by K2's routes (kernels_schur.hip, kernels_schur_ring.hip):
// the layout of the summed slabs -> padded S / e_a (or the packed triangle), the identity padding,
// and the first 32x32 diagonal block's factorization beside the reduce.
#pragma once
#include "chol_factor32.h"
#include "psba_internal.h"

namespace psba {

// slots (6 r + c, c > r) of a diagonal block that carry e_a[0..5]
static __device__ __constant__ const int EA_SLOT[6] = {1, 2, 3, 4, 5, 8};

__device__ __forceinline__ int tri(int j) { return j * (j + 1) / 2; }

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

struct SchurReduceArgs {
  const double *slab, *U, *ga;
  const int *posblock;  // per group, per position: (j << 16) | k of the block there, -1 = padding
  double *S, *ea, *scal;
  int *status;
  double mu_add, pad_one;
  int nA, n32, nGroups, try_id;
  const ReduceGroup *grp16;  // per run of 16 positions (= 9 chunks): its group's slabs, partition size, first position, slab offset
  int chunks;         // 512-byte chunks per workgroup: 1, 2, 4 or 8 (x 8 / chunks slab sequences)
  long long nchunks;  // 18 * positions / 32
  // single rank: workgroup diag_wg (one past the regular ones, -1: off) takes the first 32x32
  // diagonal block of S as summed into diag0 by the K2 workgroups, factors it (the first step
  // of the Cholesky chain: otherwise a kernel of its own with one CU busy and 255 idle) and
  // clears diag0 for the next try
  int diag_wg;
  double *diag0, *Lx, *linv;
  // with a communicator: the sums (U, mu on rank 0 and g_a folded in) go to `packed` (canonical
  // block order) instead of into S -- the lower block triangle and e_a only, half the bytes of the
  // square -- and k_schur_expand scatters them after the all-reduce
  double *packed;
  // ring route: `slab` holds ring_copies copies of the packed triangle (canonical order, stride
  // ring_stride doubles); posblock is the canonical table.  0: the LDS-partition route's slabs
  int ring_copies;
  size_t ring_stride;
};

// sums the slabs of each group of blocks (fixed order: eight interleaved slab sequences, then
// their sum), adds blockdiag(U) + mu_add I and g_a, and writes the padded row-major S (both
// block triangles) and the e_a row.  Workgroups walk the slabs in storage order (32
// consecutive pairs of doubles x 8 slab sequences each; a group's partition is a multiple of 576
// doubles, so a workgroup never straddles groups) and scatter the few results.
__device__ __forceinline__ void reduce_first_diag_block(const SchurReduceArgs &p, Factor32Lds &s) {
  const int tid = threadIdx.x, nC = p.nA / 6;
  if (tid < 4) s.flag[tid] = 0;
  if (tid == 4) s.fail = 0;
  for (int t = tid; t < GB * GB; t += 256) s.D[t / GB][t % GB] = (t / GB == t % GB) ? 1.0 : 0.0;  // padding
  __syncthreads();
  for (int e = tid; e < 21 * 36; e += 256) {
    const int blk = e / 36, rc = e % 36, r = rc / 6, c = rc % 6;
    const int j = blk < 1 ? 0 : blk < 3 ? 1 : blk < 6 ? 2 : blk < 10 ? 3 : blk < 15 ? 4 : 5, k = blk - tri(j);
    const int row = 6 * j + r, col = 6 * k + c;
    double sum = 0.0;
    if (p.ring_copies > 0) {  // canonical order: the blocks (j, k), j <= 5, are the first 21
      if (j < nC) {  // eight loads in flight: one workgroup, every round trip counts
        const double *src = p.slab + e;
        int c2 = 0;
        for (; c2 + 7 < p.ring_copies; c2 += 8) {
          double x[8];
#pragma unroll
          for (int q = 0; q < 8; q++) x[q] = src[(size_t)(c2 + q) * p.ring_stride];
#pragma unroll
          for (int q = 0; q < 8; q++) sum += x[q];
        }
        for (; c2 < p.ring_copies; c2++) sum += src[(size_t)c2 * p.ring_stride];
      }
    } else {
      sum = p.diag0[e];
      p.diag0[e] = 0.0;
    }
    if (j >= nC || row >= GB || col >= GB || (j == k && c > r)) continue;
    double acc = sum;
    if (j == k) {
      acc += p.U[36 * j + rc];
      if (r == c) acc += p.mu_add;
    }
    s.D[row][col] = acc;
  }
  __syncthreads();
  factor32(s, tid);
  for (int t = tid; t < GB * GB; t += 256) {
    const int r = t / GB, c = t % GB;
    p.Lx[(size_t)r * p.n32 + c] = f32_L(s, r, c);
    p.linv[t] = f32_Linv(s, r, c);
  }
  if (tid == 0 && s.fail) p.status[1] = p.try_id;
}

// where slot e = 36 * (global position) + rc of the summed slabs goes: adds blockdiag(U) + mu_add I and
// g_a and writes S (both block triangles) / e_a, or the packed buffer
__device__ __forceinline__ void reduce_scatter(const SchurReduceArgs &p, long long e, double sum) {
  const int jk = p.posblock[e / 36];
  const int j = jk >> 16, jb = jk & 0xFFFF, rc = (int)(e % 36), r = rc / 6, c = rc % 6;
  int ea_slot = -1;  // a diagonal block holds its lower triangle; six upper slots carry e_a, the rest is unused
  if (jk >= 0 && j == jb && c > r) {
#pragma unroll
    for (int t = 0; t < 6; t++)
      if (rc == EA_SLOT[t]) ea_slot = t;
  }
  const bool unused = jk < 0 || (j == jb && c > r && ea_slot < 0);
  if (!unused) {
    if (ea_slot >= 0) {
      sum += p.ga[6 * j + ea_slot];
    } else if (j == jb) {
      sum += p.U[36 * j + rc];
      if (r == c) sum += p.mu_add;
    }
  }
  if (p.packed) {
    // canonical order (block tri(j) + k of the lower block triangle), NOT slab order: where a
    // block sits inside its partition is chosen per rank from that rank's own traffic counts, so
    // slab positions do not line up between ranks
    if (jk >= 0) p.packed[(size_t)36 * (tri(j) + jb) + rc] = unused ? 0.0 : sum;
    return;
  }
  if (unused) return;
  if (ea_slot >= 0) {
    p.ea[6 * j + ea_slot] = sum;
    return;
  }
  p.S[(size_t)(6 * jb + c) * p.n32 + 6 * j + r] = sum;
  p.S[(size_t)(6 * j + r) * p.n32 + 6 * jb + c] = sum;
}


}  // namespace psba
