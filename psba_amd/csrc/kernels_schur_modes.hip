// kernels_schur_modes.hip -- the instrumented fork of k_schur_lds (kernels_schur.hip): the ablation variants
// behind the "three walls" measurements of DESIGN 5c (PSBA_SCHUR_MODE=n).  Timing only -- most modes give wrong
// sums.  Compiled only with PSBA_BUILD_EXPERIMENTS=1 (psba_amd/build.py); the product kernel carries none of this.
#include <cstdlib>

#include "camera_model.h"
#include "schur_lds_args.h"

namespace psba {

// MODE is development instrumentation (ablation timing, PSBA_SCHUR_MODE): 0 = full kernel;
// 1 = products without the LDS atomics; 2 = no product loop; 3 = no W_b loads (wrong
// results); 4 = zero + flush only; 5 = every row of 16 lanes on 16 distinct bank pairs by
// construction (wrong results: what the bank conflicts of the real schedule cost); 6 = the 36
// atomics without the 108 fp64 operations that form the values; 7 = ds_add_u64 on the values' bit
// patterns; 8 = 6 with ds_add_u64; 9 = all loads and arithmetic, no LDS atomics; 10 / 11 = all loads,
// V*^-1 only, 36 f64 / u64 atomics of loaded values; 12 / 13 = no record loads, arithmetic + f64 / u64
// atomics; 14 = one a-side and two partners per turn (all of 7..14 give wrong sums).
template <bool DUMP, int MODE>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_lds_modes(SchurLdsArgs p) {
  extern __shared__ double sPart[];  // [nblk][37]
  const int tid = threadIdx.x;
  int w = blockIdx.x;
  if ((p.nWg & 7) == 0) w = (blockIdx.x & 7) * (p.nWg >> 3) + (blockIdx.x >> 3);
  const SchurWg wg = p.wg[w];
  for (int t = tid; t < BLK_STRIDE * wg.nblk; t += SCHUR_THREADS) sPart[t] = 0.0;
  __syncthreads();

  const long long s1 = (MODE == 4) ? wg.item0 : wg.item1;
  double keep = 0.0;
  // the next item word is fetched a turn ahead: its latency would otherwise sit in front of the
  // record loads of every turn (a wave has only about six turns)
  unsigned long long item_next = (wg.item0 + tid < s1) ? p.items[wg.item0 + tid] : SCHUR_NULL_ITEM;
  constexpr int STEP = (MODE == 14 ? 2 : 1) * SCHUR_THREADS;
  for (long long t = wg.item0 + tid; t < s1; t += STEP) {
    const unsigned long long item = item_next;
    if (t + STEP < s1) item_next = p.items[t + STEP];
    // MODE 14 (timing only, wrong sums): what an item with one a-side and two partners would cost --
    // the turn's second product takes its partner and its block from the item 1024 further on and
    // reuses this item's W_a, V*^-1, Y, e
    unsigned long long item2 = SCHUR_NULL_ITEM;
    if (MODE == 14 && t + SCHUR_THREADS < s1) item2 = p.items[t + SCHUR_THREADS];
    if (item == SCHUR_NULL_ITEM) continue;
    const int a = wg.obs0 + (int)(item & ((1u << ITEM_OBS_BITS) - 1));
    const int i = wg.pt0 + (int)((item >> ITEM_OBS_BITS) & ((1u << ITEM_PT_BITS) - 1));
    const int boff = (int)((item >> (ITEM_OBS_BITS + ITEM_PT_BITS)) & ((1u << ITEM_BOFF_BITS) - 1));
    int pos = (int)(item >> (ITEM_OBS_BITS + ITEM_PT_BITS + ITEM_BOFF_BITS));
    if (MODE == 5) pos = (tid & 15) + 16 * ((pos >> 4) % (wg.nblk >> 4));
    // every address is known now: issue all loads of the product together
    const double *pv = p.PV + 9 * (size_t)i;
    const double2 *wa = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)a);
    const double2 *wb2 = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)(a - boff));
    double v[6], vi[6], w[18], wb[18];
    double g0, g1, g2;
    if (MODE == 12 || MODE == 13) {
      // no record loads at all: operands made up from the item word (arithmetic and atomics only)
      const double base = 1.0 + 1e-3 * (double)(item & 1023);
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] = (k == 0 || k == 3 || k == 5) ? 4.0 + base : 0.25 * base;
      g0 = base; g1 = base + 1.0; g2 = base + 2.0;
#pragma unroll
      for (int k = 0; k < 18; k++) {
        w[k] = base + k;
        wb[k] = base - k;
      }
    } else {
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = pv[k];
    g0 = pv[6], g1 = pv[7], g2 = pv[8];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const double2 q = wa[k];
      w[2 * k] = q.x;
      w[2 * k + 1] = q.y;
    }
    }
    if (MODE == 12 || MODE == 13) {
    } else if (MODE == 3) {  // products without the W_b loads
#pragma unroll
      for (int k = 0; k < 18; k++) wb[k] = w[k] + 1.0;
    } else if (MODE != 2) {
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double2 q = wb2[k];
        wb[2 * k] = q.x;
        wb[2 * k + 1] = q.y;
      }
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
    // Y carries the sign of the product: Y = -W_a V*^-1, so that neither the 36 values nor the
    // e_a terms need a sign flip of their own
#pragma unroll
    for (int k = 0; k < 6; k++) vi[k] = -vi[k];
    double Y[18], e[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
      Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
      Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
      Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      e[r] = Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2;
    }
    if (DUMP && self) {
#pragma unroll
      for (int k = 0; k < 18; k++) p.dbg_Y[18 * (size_t)a + k] = -Y[k];
    }
    if (MODE == 2) {
      keep += Y[0] + Y[17] + e[0] + e[5];
      continue;
    }
    double *blk = sPart + BLK_STRIDE * pos;
    if (MODE == 6) {
#pragma unroll
      for (int rc = 0; rc < 36; rc++) atomicAdd(&blk[rc], wb[rc % 18]);
      continue;
    }
    if (MODE == 10 || MODE == 11) {
#pragma unroll
      for (int k = 0; k < 18; k++) asm volatile("" ::"v"(w[k]));
      asm volatile("" ::"v"(g0), "v"(g1), "v"(g2));
#pragma unroll
      for (int rc = 0; rc < 36; rc++) {
        if (MODE == 10)
          atomicAdd(&blk[rc], wb[rc % 18]);
        else
          atomicAdd(reinterpret_cast<unsigned long long *>(&blk[rc]), (unsigned long long)__double_as_longlong(wb[rc % 18]));
      }
      continue;
    }
    if (MODE == 8) {
#pragma unroll
      for (int rc = 0; rc < 36; rc++)
        atomicAdd(reinterpret_cast<unsigned long long *>(&blk[rc]), (unsigned long long)__double_as_longlong(wb[rc % 18]));
      continue;
    }
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
        // the self-product's e_a terms ride in redundant upper-triangle slots (EA_SLOT)
        if (r == 0 && c >= 1) val[c] = self ? e[c - 1] : val[c];
        if (r == 1 && c == 2) val[c] = self ? e[5] : val[c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        if (MODE == 1)
          keep += val[c];
        else if (MODE == 9)
          asm volatile("" ::"v"(val[c]));
        else if (MODE == 7 || MODE == 13)
          atomicAdd(reinterpret_cast<unsigned long long *>(&blk[6 * r + c]), (unsigned long long)__double_as_longlong(val[c]));
        else
          atomicAdd(&blk[6 * r + c], val[c]);
      }
    }
    if (MODE == 14 && item2 != SCHUR_NULL_ITEM) {
      const int boffB = (int)((item2 >> (ITEM_OBS_BITS + ITEM_PT_BITS)) & ((1u << ITEM_BOFF_BITS) - 1));
      const int posB = (int)(item2 >> (ITEM_OBS_BITS + ITEM_PT_BITS + ITEM_BOFF_BITS));
      const int bB = a - boffB > wg.obs0 ? a - boffB : wg.obs0;
      const double2 *wq = reinterpret_cast<const double2 *>(p.W + 18 * (size_t)bB);
      double wc[18];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const double2 q = wq[k];
        wc[2 * k] = q.x;
        wc[2 * k + 1] = q.y;
      }
      double *blkB = sPart + BLK_STRIDE * posB;
      const bool selfB = boffB == 0;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        double val[6];
#pragma unroll
        for (int c = 0; c < 6; c++) val[c] = Y[3 * r] * wc[3 * c];
#pragma unroll
        for (int c = 0; c < 6; c++) val[c] = fma(Y[3 * r + 1], wc[3 * c + 1], val[c]);
#pragma unroll
        for (int c = 0; c < 6; c++) val[c] = fma(Y[3 * r + 2], wc[3 * c + 2], val[c]);
#pragma unroll
        for (int c = 0; c < 6; c++) {
          if (r == 0 && c >= 1) val[c] = selfB ? e[c - 1] : val[c];
          if (r == 1 && c == 2) val[c] = selfB ? e[5] : val[c];
        }
#pragma unroll
        for (int c = 0; c < 6; c++) atomicAdd(&blkB[6 * r + c], val[c]);
      }
    }
  }
  if (MODE != 0 && keep == 12345.678) sPart[0] = keep;
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


// launches mode 1..14 of the fork (mode 0 = the product kernel, not here); false: no such mode
bool launch_schur_lds_mode(int mode, dim3 G, dim3 B, size_t lds, hipStream_t s, const SchurLdsArgs &a) {
  static bool attr_set = false;
  if (!attr_set) {  // every instantiation may use the full 160 KiB of LDS (modes 7-14 had no attribute)
    const int dyn = 163840 - 256;
    const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
#define PSBA_MODE_ATTR(M) (void)hipFuncSetAttribute((const void *)k_schur_lds_modes<false, M>, attr, dyn);
    PSBA_MODE_ATTR(1) PSBA_MODE_ATTR(2) PSBA_MODE_ATTR(3) PSBA_MODE_ATTR(4) PSBA_MODE_ATTR(5) PSBA_MODE_ATTR(6) PSBA_MODE_ATTR(7)
    PSBA_MODE_ATTR(8) PSBA_MODE_ATTR(9) PSBA_MODE_ATTR(10) PSBA_MODE_ATTR(11) PSBA_MODE_ATTR(12) PSBA_MODE_ATTR(13) PSBA_MODE_ATTR(14)
#undef PSBA_MODE_ATTR
    attr_set = true;
  }
  switch (mode) {
#define PSBA_MODE_CASE(M) case M: hipLaunchKernelGGL((k_schur_lds_modes<false, M>), G, B, lds, s, a); return true;
    PSBA_MODE_CASE(1) PSBA_MODE_CASE(2) PSBA_MODE_CASE(3) PSBA_MODE_CASE(4) PSBA_MODE_CASE(5) PSBA_MODE_CASE(6) PSBA_MODE_CASE(7)
    PSBA_MODE_CASE(8) PSBA_MODE_CASE(9) PSBA_MODE_CASE(10) PSBA_MODE_CASE(11) PSBA_MODE_CASE(12) PSBA_MODE_CASE(13) PSBA_MODE_CASE(14)
#undef PSBA_MODE_CASE
  }
  return false;
}

}  // namespace psba
