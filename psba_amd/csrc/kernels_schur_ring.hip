// kernels_schur_ring.hip -- K2, ring route (few cameras): damping + V^-1 + Y = W V^-1 + S = U* - Y W^T
// + e_a = g_a - Y g_b as a GATHER into register-resident 6x6 blocks.
//
// Replaces the same reference kernels as kernels_schur.hip (CL_files/update_UV.cl:5-31,
// compute_Vinv.cl:6-90, compute_Yblks.cl:6-39, compute_S.cl:6-78, compute_ea.cl:6-37,
// restore_UVdiag.cl:2-25; wrappers PSBA/sba_func.cpp:624-995).  The schedule is built once per
// problem by schur_ring_plan.cpp; this file replays it:
//   k_schur_vinv  (V_i + mu I)^-1 and (V_i + mu I)^-1 g_b,i, once per point and try;
//   k_schur_ring  one workgroup per (stretch of the point sequence, range of the canonical block
//                 order): 4 consumer waves whose lanes own one 6x6 block each (sums in registers,
//                 108 fma per product, both operands read from LDS with ds_read_b128), 4 mover
//                 waves that stream the W records the range needs into LDS with LDS-DMA (runs of
//                 up to 7 records of one point, contiguous in W and in LDS: one base address per
//                 instruction) and 4 prepper waves that form Y_a = W_a V*^-1 once per (observation,
//                 workgroup) and add the e_a terms -Y_a g_b,i.  Two workgroup barriers per step;
//                 the lists say what every lane does in every step, so no lane ever looks for work.
//   k_schur_sum   adds the nS copies of the packed triangle, folds in U + mu I and g_a and writes
//                 the padded S (both block triangles), e_a, the identity padding (or the packed
//                 buffer for the all-reduce); one extra workgroup factors the first 32x32
//                 diagonal block, as k_schur_reduce does on the LDS-partition route.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "camera_model.h"
#include "schur_common.h"

namespace psba {

constexpr int RING_THREADS = RING_LANES + 64 * (RING_MOVERS + RING_PREPPERS);  // 768
constexpr int RING_STEPWIN = 256;  // step descriptors kept in LDS at a time
constexpr int RING_PART_STRIDE = 37;
static_assert(RING_PAGE * 9 <= 64, "one LDS-DMA instruction (64 lanes x 16 B) moves a run of records");
static_assert(RING_MAXOPS == 16, "a mover lane u < 16 holds the u-th page load of its wave");
static_assert(RING_LANES * RING_PART_STRIDE * 8 <= RING_SLOTS * 144, "the flush area reuses the ring");

struct RingArgs {
  const double *W, *pvi;
  const RingWg *wg;
  const RingStep *steps;
  const unsigned *entries;
  const int *ops;
  const RingJob *jobs;
  const int *bl0;
  const int *canon;
  double *slab;
  int *status;
  double *dbg_Y;
  unsigned long long stride;  // doubles per copy of the packed triangle
  int try_id;
  long long *tim;  // development instrumentation (PSBA_RING_TIMING): s_memtime stamps of workgroups 0 and 100
  int mode;  // development instrumentation (PSBA_RING_MODE): bit 0 consumers skip the products, bit 1 producers skip the DMA,
             // bit 2 producers skip the wait for their loads, bit 3 producers skip the Y preparation (wrong results)
};

// LDS: ring | Y ring | e_a accumulators of the rows of this range | window of step descriptors
__host__ __device__ constexpr size_t ring_lds_bytes(int nrows_max) {
  return (size_t)RING_SLOTS * 144 + (size_t)nrows_max * 48 +
         (size_t)(RING_STEPWIN + 5) * sizeof(RingStep);
}

// One LDS-DMA instruction: lane l moves 16 bytes from its own global address to LDS byte
// lds_wave_base + 16 l.  Written as inline assembly on purpose: the compiler orders every later
// LDS access behind a global_load_lds it knows of with s_waitcnt vmcnt(0) (it cannot tell which
// LDS bytes the DMA writes), which would make the producers wait for the pages they have just
// asked for before preparing the previous step's.  What must be ordered is ordered by hand: the
// one s_waitcnt vmcnt per step in the producers' loop plus the workgroup barrier.
// Operations the compiler does not know of can only make its own vmcnt waits stricter (memory
// operations retire in issue order), never weaker.  M0 is written here and nowhere else in this
// kernel (no other instruction of it reads M0 on gfx950).
__device__ __forceinline__ void dma16(const double2 *src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_wave_base) : "memory");
}

// what a mover / prepper lane holds for one step (four such sets rotate, indexed by step % 4)
struct MoverRegs {
  int2 op;       // lane u < 16: the u-th page load of this wave in the step: first observation, first slot | records << 16
};
struct PrepRegs {
  int4 job;      // this lane's job of the step: observation, point, W slot | Y slot << 16, e_a row
  double pv[9];  // (V_i + mu I)^-1 (sym6) | (V_i + mu I)^-1 g_b,i of that job's point
};

// V*^-1 = (V_i + mu I)^-1 (closed form, reference CL_files/compute_Vinv.cl:29,76-86) and V*^-1 g_b,i,
// once per point and try; mu is added here, so V is never modified (update_UV.cl / restore_UVdiag.cl)
template <bool DUMP>
__global__ __launch_bounds__(256) void k_schur_vinv(const double *PV, double *pvi, double *dbg_Vinv, int *status, double mu,
                                                   int nP, int try_id) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nP) return;
  const double *pv = PV + 9 * (size_t)i;
  double v[6], vi[6];
#pragma unroll
  for (int k = 0; k < 6; k++) v[k] = pv[k];
  const double g0 = pv[6], g1 = pv[7], g2 = pv[8];
  v[0] += mu;
  v[3] += mu;
  v[5] += mu;
  if (sym3_inverse(v, vi)) status[0] = try_id;
  double *o = pvi + 9 * (size_t)i;
#pragma unroll
  for (int k = 0; k < 6; k++) o[k] = vi[k];
  o[6] = vi[0] * g0 + vi[1] * g1 + vi[2] * g2;
  o[7] = vi[1] * g0 + vi[3] * g1 + vi[4] * g2;
  o[8] = vi[2] * g0 + vi[4] * g1 + vi[5] * g2;
  if (DUMP) {
    double *d = dbg_Vinv + 9 * (size_t)i;
    d[0] = vi[0]; d[1] = vi[1]; d[2] = vi[2];
    d[3] = vi[1]; d[4] = vi[3]; d[5] = vi[4];
    d[6] = vi[2]; d[7] = vi[4]; d[8] = vi[5];
  }
}

// The workgroup barrier of the step loop.  __syncthreads() is a workgroup-scope fence as well, for
// which the compiler drains EVERY outstanding memory operation of the wave (s_waitcnt vmcnt(0)) --
// here that would wait at each barrier for the loads that were issued several steps ahead precisely
// so that nobody waits for them.  What the barrier must order is LDS traffic only: Y slots and e_a
// written by the preppers (ds_write / ds_add), LDS reads of slots about to be overwritten: lgkmcnt(0).
// (The LDS-DMA writes are ordered by the movers' s_waitcnt vmcnt, see there.)
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool DUMP>
__global__ __launch_bounds__(RING_THREADS) void k_schur_ring(RingArgs p, int nrows_max) {
  extern __shared__ double2 smem[];
  double2 *ring = smem;
  double *sEa = reinterpret_cast<double *>(smem + (size_t)RING_SLOTS * 9);
  RingStep *sSteps = reinterpret_cast<RingStep *>(sEa + (size_t)nrows_max * 6);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const RingWg wg = p.wg[blockIdx.x];
  double2 *yring = smem + (size_t)wg.nwslots * 9;
  for (int t = tid; t < wg.nrows * 6; t += RING_THREADS) sEa[t] = 0.0;
  const int nsteps = wg.nsteps;  // a multiple of 4
  const RingStep *gsteps = p.steps + wg.step0;
  long long *tim = nullptr;  // (development: stamps of one lane per role of workgroups 0 and 100)
  if (p.tim && lane == 0 && (wave == 0 || wave == 4 || wave == 8) && (blockIdx.x == 0 || blockIdx.x == 100))
    tim = p.tim + (blockIdx.x ? 4096 : 0);
  auto stamp = [&](int t, int k) {
    if (tim && t < 256) tim[16 * t + k] = __builtin_amdgcn_s_memtime();
  };
  // Every step has two workgroup barriers.  MID: the movers have seen the page loads of the previous
  // step land (their s_waitcnt), so the preppers may read those W records.  END: the products of the
  // step are done (their slots may be reused by the next step's loads and Y's), and the Y's prepared
  // in this step are written.  All twelve waves execute the same sequence of barriers.
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; k++) acc[k] = 0.0;

  if (wave < RING_LANES / 64) {
    // ---------------- consumers: lanes own one 6x6 block each ----------------
    // a lane's entries are read a group of four steps ahead, at the start of the group before (a step
    // is shorter than a trip to HBM: by the time a group's entries are moved into place, their loads
    // are four steps old); the list is padded, so the loads need no bounds
    __builtin_amdgcn_s_setprio(2);
    const unsigned *ent = p.entries + wg.ent0 + tid;
    unsigned q0 = ent[0], q1 = ent[(size_t)1 * RING_LANES], q2 = ent[(size_t)2 * RING_LANES], q3 = ent[(size_t)3 * RING_LANES];
    auto product = [&](unsigned e) {
      if (e != RING_NULL_ENTRY && !(p.mode & 1)) {
        const double2 *py = yring + (size_t)(e >> 16) * 9;
        const double2 *pw = ring + (size_t)(e & 0xFFFFu) * 9;
        double y[18], w[18];
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const double2 v = py[k];
          y[2 * k] = v.x;
          y[2 * k + 1] = v.y;
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const double2 v = pw[k];
          w[2 * k] = v.x;
          w[2 * k + 1] = v.y;
        }
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
          for (int r = 0; r < 6; r++) {
            double a = acc[6 * r + c];
            a = __builtin_fma(y[3 * r], w[3 * c], a);
            a = __builtin_fma(y[3 * r + 1], w[3 * c + 1], a);
            a = __builtin_fma(y[3 * r + 2], w[3 * c + 2], a);
            acc[6 * r + c] = a;
          }
      }
    };
    auto cstep = [&](int t, unsigned e) {
      ring_barrier();  // MID
      stamp(t, 0);
      product(e);
      stamp(t, 1);
      ring_barrier();  // END
    };
    for (int t = 0; t < nsteps; t += 4) {
      if ((t & (RING_STEPWIN - 1)) == 0) ring_barrier();  // (the movers reload the window of step descriptors)
      const unsigned *nx = ent + (size_t)(t + 4) * RING_LANES;
      const unsigned n0 = nx[0], n1 = nx[(size_t)1 * RING_LANES], n2 = nx[(size_t)2 * RING_LANES], n3 = nx[(size_t)3 * RING_LANES];
      cstep(t, q0);
      cstep(t + 1, q1);
      cstep(t + 2, q2);
      cstep(t + 3, q3);
      q0 = n0;
      q1 = n1;
      q2 = n2;
      q3 = n3;
    }
  } else if (wave < RING_LANES / 64 + RING_MOVERS) {
    // ---------------- movers: stream W records into the ring with LDS-DMA ----------------
    // A mover's only memory operations are its page loads (LDS-DMA) and ONE list load per step
    // (the page loads of the step three ahead, issued after the step's page loads).  Memory operations
    // retire in issue order behind one counter, so at the start of a step "all page loads of the
    // previous step have landed" is exactly s_waitcnt vmcnt(1).
    const int mv = __builtin_amdgcn_readfirstlane(wave) - RING_LANES / 64;
    const double2 *W2 = reinterpret_cast<const double2 *>(p.W);
    const int2 *ops2 = reinterpret_cast<const int2 *>(p.ops) + wg.op0;
    MoverRegs R0, R1, R2, R3;
    auto my_ops = [&](const RingStep &st, int &first, int &count) {
      const int ob = __builtin_amdgcn_readfirstlane(st.op_begin), oe = __builtin_amdgcn_readfirstlane(st.op_end);
      const int nops = oe - ob, n4 = (nops + RING_MOVERS - 1) / RING_MOVERS;
      first = ob + mv * n4;
      count = nops - mv * n4;
      count = count < 0 ? 0 : count > n4 ? n4 : count;
    };
    auto load_ops = [&](const RingStep &st, MoverRegs &R) {  // unconditional (the list is padded)
      int first, count;
      my_ops(st, first, count);
      R.op = ops2[first + (lane & (RING_MAXOPS - 1))];
    };
    auto issue = [&](const RingStep &st, const MoverRegs &R) {
      int first, count;
      my_ops(st, first, count);
      asm volatile("" ::"v"(R.op.x), "v"(R.op.y));  // (one wait for the list load here)
      for (int u = 0; u < count; u++) {
        // a run of n <= 7 records of one point: W + 144 a0 + 16 lane -> LDS byte 144 slot + 16 lane, lanes < 9 n
        const int a0 = __builtin_amdgcn_readlane(R.op.x, u), sn = __builtin_amdgcn_readlane(R.op.y, u);
        const int n = sn >> 16, slot = (p.mode & 16) ? 0 : (sn & 0xFFFF);
        // (the dynamic LDS starts at byte 0 of the workgroup's allocation: no static LDS in this kernel)
        if (lane < 9 * n && !(p.mode & 2)) dma16(W2 + (size_t)a0 * 9 + lane, (unsigned)slot * 144u);
      }
    };
    // step t: C = the set of step t; P = the set of step t - 1, which takes the op words of step t + 3
    auto step = [&](int t, MoverRegs &P, MoverRegs &C) {
      const int tw = t & (RING_STEPWIN - 1);
      stamp(t, 2);
      if (p.tim)  // (the stamps are stores: with them in the stream the count below would not hold)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (!(p.mode & 4))
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      stamp(t, 3);
      ring_barrier();  // MID
      if (!(p.mode & 32)) issue(sSteps[tw], C);
      asm volatile("" ::: "memory");  // the list load below must be issued after the page loads
      stamp(t, 4);
      load_ops(sSteps[tw + 3], P);
      stamp(t, 5);
      ring_barrier();  // END
    };
    for (int t = 0; t < nsteps; t += 4) {
      if ((t & (RING_STEPWIN - 1)) == 0) {
        // the window holds the steps [t, t + RING_STEPWIN + 4); entry RING_STEPWIN + 4 is step t - 1
        for (int q = tid - RING_LANES; q < RING_STEPWIN + 4; q += 64 * RING_MOVERS) {
          RingStep z{0, 0, 0, 0};
          if (t + q < nsteps) z = gsteps[t + q];
          sSteps[q] = z;
        }
        if (tid == RING_LANES) {
          RingStep z{0, 0, 0, 0};
          if (t > 0) z = gsteps[t - 1];
          sSteps[RING_STEPWIN + 4] = z;
        }
        ring_barrier();
        if (t == 0) {
          load_ops(sSteps[0], R0);
          load_ops(sSteps[1], R1);
          load_ops(sSteps[2], R2);
        }
      }
      step(t, R3, R0);
      step(t + 1, R0, R1);
      step(t + 2, R1, R2);
      step(t + 3, R2, R3);
    }
  } else {
    // ---------------- preppers: Y_a = W_a V*^-1, e_a -= W_a (V*^-1 g_b,i) ----------------
    // Two lanes per job (three rows of Y each), 32 jobs per wave and step.  Their global reads (job
    // descriptions three steps ahead, the point's V*^-1 | V*^-1 g_b one step ahead) are ordinary
    // loads into four register sets that take turns.
    const int pv_ = __builtin_amdgcn_readfirstlane(wave) - RING_LANES / 64 - RING_MOVERS;
    const int4 *jobs4 = reinterpret_cast<const int4 *>(p.jobs) + wg.job0;
    const int jq = lane >> 1, half = lane & 1;
    PrepRegs R0, R1, R2, R3;
    auto my_job = [&](const RingStep &st) { return st.job_begin + pv_ * RING_PREP_JOBS + jq; };
    auto load_job = [&](const RingStep &st, PrepRegs &R) { R.job = jobs4[my_job(st)]; };  // unconditional (padded list)
    auto load_pv = [&](PrepRegs &R) {
      const double *pv = p.pvi + 9 * (size_t)R.job.y;
#pragma unroll
      for (int k = 0; k < 9; k++) R.pv[k] = pv[k];
    };
    auto prep = [&](const RingStep &st, const PrepRegs &R) {
      if (my_job(st) >= st.job_end) return;
      // rows 3 half .. 3 half + 2 of W_a: nine consecutive doubles
      const double *pwa = reinterpret_cast<const double *>(ring) + (size_t)(R.job.z & 0xFFFF) * 18 + 9 * half;
      double *py = reinterpret_cast<double *>(yring) + (size_t)(R.job.z >> 16) * 18 + 9 * half;
      double w[9], yv[9];
#pragma unroll
      for (int k = 0; k < 9; k++) w[k] = pwa[k];
      const double *vi = R.pv;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
        yv[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        yv[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        yv[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      }
#pragma unroll
      for (int k = 0; k < 9; k++) py[k] = yv[k];
      if (R.job.w >= 0) {
#pragma unroll
        for (int r = 0; r < 3; r++)
          atomicAdd(&sEa[6 * R.job.w + 3 * half + r], -(w[3 * r] * vi[6] + w[3 * r + 1] * vi[7] + w[3 * r + 2] * vi[8]));
      }
      if (DUMP) {
#pragma unroll
        for (int k = 0; k < 9; k++) p.dbg_Y[18 * (size_t)R.job.x + 9 * half + k] = yv[k];
      }
    };
    // step t: N = the set of step t + 1 (V | g of its jobs is read now), P = of step t - 1 (its jobs are
    // prepared now -- their W records were loaded in step t - 1 and have landed: MID -- then the set
    // takes the job descriptions of step t + 3)
    auto step = [&](int t, PrepRegs &P, PrepRegs &N) {
      const int tw = t & (RING_STEPWIN - 1);
      if (!(p.mode & 64)) load_pv(N);
      ring_barrier();  // MID
      stamp(t, 6);
      if (t > 0 && !(p.mode & 8)) prep(sSteps[tw == 0 ? RING_STEPWIN + 4 : tw - 1], P);
      stamp(t, 7);
      load_job(sSteps[tw + 3], P);
      ring_barrier();  // END
    };
    for (int t = 0; t < nsteps; t += 4) {
      if ((t & (RING_STEPWIN - 1)) == 0) {
        ring_barrier();  // (the movers have reloaded the window)
        if (t == 0) {
          load_job(sSteps[0], R0);
          load_job(sSteps[1], R1);
          load_job(sSteps[2], R2);
          load_pv(R0);
        }
      }
      step(t, R3, R1);
      step(t + 1, R0, R2);
      step(t + 2, R1, R3);
      step(t + 3, R2, R0);
    }
  }

  // ---- flush: the lanes of a block add up through LDS (the ring is free now); -sum to the copy ----
  if (wave < RING_LANES / 64) {
    double *part = reinterpret_cast<double *>(ring) + (size_t)tid * RING_PART_STRIDE;
#pragma unroll
    for (int k = 0; k < 36; k++) part[k] = acc[k];
  }
  __syncthreads();
  {
    const double *part = reinterpret_cast<const double *>(ring);
    const int *bl0 = p.bl0 + wg.bl0;
    double *dst = p.slab + (size_t)wg.copy * p.stride + (size_t)36 * wg.blk0;
    for (int e = tid; e < 36 * wg.nblk; e += RING_THREADS) {
      const int lb = e / 36, rc = e % 36;
      const int l0 = bl0[lb], l1 = bl0[lb + 1];
      double sum = 0.0;
      for (int l = l0; l < l1; l++) sum += part[(size_t)l * RING_PART_STRIDE + rc];
      double v = -sum;
      const int jk = p.canon[wg.blk0 + lb];
      if ((jk >> 16) == (jk & 0xFFFF)) {  // a diagonal block: six upper-triangle slots carry e_a
#pragma unroll
        for (int q = 0; q < 6; q++)
          if (rc == EA_SLOT[q]) v = sEa[6 * ((jk >> 16) - wg.row0) + q];
      }
      dst[e] = v;
    }
  }
}

// adds the copies of the packed triangle; see k_schur_reduce for what happens to the sums
__global__ __launch_bounds__(256) void k_schur_sum(SchurReduceArgs p, long long total) {
  __shared__ Factor32Lds sF;
  if ((int)blockIdx.x == p.diag_wg) {
    reduce_first_diag_block(p, sF);
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x < 4 * SC_NPART) p.scal[SC_PART + threadIdx.x] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 64) p.status[3] = p.try_id;
  if (!p.packed)
    write_padding(p.S, p.nA, p.n32, p.pad_one, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                  (size_t)(gridDim.x - (p.diag_wg >= 0 ? 1 : 0)) * blockDim.x);
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  double sum = 0.0;
  const double *s = p.slab + e;
  int c = 0;
  for (; c + 3 < p.ring_copies; c += 4) {
    const double x0 = s[(size_t)c * p.ring_stride], x1 = s[(size_t)(c + 1) * p.ring_stride];
    const double x2 = s[(size_t)(c + 2) * p.ring_stride], x3 = s[(size_t)(c + 3) * p.ring_stride];
    sum += x0;
    sum += x1;
    sum += x2;
    sum += x3;
  }
  for (; c < p.ring_copies; c++) sum += s[(size_t)c * p.ring_stride];
  reduce_scatter(p, e, sum);
}

int launch_schur_ring(psba_ctx *h, double mu, bool dump) {
  const Dims &d = h->d;
  if (!h->ring_attr_set) {
    const auto attr = hipFuncAttributeMaxDynamicSharedMemorySize;
    const int dyn = (int)ring_lds_bytes(128);
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_ring<false>, attr, dyn));
    PSBA_HIP(h, hipFuncSetAttribute((const void *)k_schur_ring<true>, attr, dyn));
    h->ring_attr_set = true;
  }
  if (getenv("PSBA_RING_TIMING") && !h->chol_tim_ring) {
    PSBA_HIP(h, hipMalloc(&h->chol_tim_ring, sizeof(long long) * 8192));
    PSBA_HIP(h, hipMemset(h->chol_tim_ring, 0, sizeof(long long) * 8192));
  }
  RingArgs a;
  a.W = h->W;
  a.pvi = h->ring_pvi;
  a.wg = h->ring_wg;
  a.steps = h->ring_steps;
  a.entries = h->ring_entries;
  a.ops = h->ring_ops;
  a.jobs = h->ring_jobs;
  a.bl0 = h->ring_bl0;
  a.canon = h->ring_canon;
  a.slab = h->ring_slab;
  a.status = h->status;
  a.dbg_Y = h->dbg_Y;
  a.stride = h->packedN;
  a.try_id = h->try_id;
  a.tim = getenv("PSBA_RING_TIMING") ? h->chol_tim_ring : nullptr;
  a.mode = getenv("PSBA_RING_MODE") ? atoi(getenv("PSBA_RING_MODE")) : 0;
  const int nrows_max = d.nC < 128 ? d.nC : 128;
  const size_t lds = ring_lds_bytes(nrows_max);

  SchurReduceArgs r{};
  r.slab = h->ring_slab;
  r.U = h->U;
  r.ga = h->ga;
  r.posblock = h->ring_canon;
  r.S = h->red;
  r.ea = h->red + (size_t)h->n32 * h->n32;
  r.scal = h->scal;
  r.status = h->status;
  r.mu_add = h->rank == 0 ? mu : 0.0;
  r.pad_one = h->rank == 0 ? 1.0 : 0.0;
  r.nA = d.nA;
  r.n32 = h->n32;
  r.nGroups = 0;
  r.try_id = h->try_id;
  r.grp16 = nullptr;
  r.chunks = 0;
  r.nchunks = 0;
  r.diag0 = nullptr;
  r.Lx = h->chol_L;
  r.linv = h->chol_ws;
  r.packed = (h->comm || (h->nranks > 1 && getenv("PSBA_SCHUR_PACKED"))) ? h->redp : nullptr;
  h->packed_pending = r.packed != nullptr;
  r.ring_copies = h->ring_nS;
  r.ring_stride = h->packedN;
  const long long total = (long long)h->packedN;
  const int rgrid = (int)((total + 255) / 256);
  const bool fuse_diag = h->nranks == 1 && !getenv("PSBA_CHOL_SEPARATE_DIAG") && !getenv("PSBA_SCHUR_NO_FLUSH_DIAG");
  r.diag_wg = fuse_diag ? rgrid : -1;

  const bool pair = (h->prof & (1u << PSBA_K_SCHUR)) && !(h->prof & (1u << PSBA_K_SCHUR_REDUCE));
  {
    ProfScope pp(h, pair ? PSBA_K_SCHUR : -1);
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR);
      const dim3 vg((d.nP + 255) / 256);
      if (dump)
        hipLaunchKernelGGL(k_schur_vinv<true>, vg, dim3(256), 0, h->stream, h->PV, h->ring_pvi, h->dbg_Vinv, h->status, mu, d.nP, h->try_id);
      else
        hipLaunchKernelGGL(k_schur_vinv<false>, vg, dim3(256), 0, h->stream, h->PV, h->ring_pvi, h->dbg_Vinv, h->status, mu, d.nP, h->try_id);
      if (dump)
        hipLaunchKernelGGL(k_schur_ring<true>, dim3(h->ring_nWg), dim3(RING_THREADS), lds, h->stream, a, nrows_max);
      else
        hipLaunchKernelGGL(k_schur_ring<false>, dim3(h->ring_nWg), dim3(RING_THREADS), lds, h->stream, a, nrows_max);
    }
    PSBA_HIP(h, hipGetLastError());
    {
      ProfScope ps(h, pair ? -1 : PSBA_K_SCHUR_REDUCE);
      hipLaunchKernelGGL(k_schur_sum, dim3(rgrid + (r.diag_wg >= 0 ? 1 : 0)), dim3(256), 0, h->stream, r, total);
    }
  }
  h->diag_done = r.diag_wg >= 0;
  PSBA_HIP(h, hipGetLastError());
  if (a.tim && getenv("PSBA_RING_TIMING_DUMP")) {  // (16 stamps per step: see the kernel)  // development: per-step stamps of two workgroups, cycles since the workgroup's first
    std::vector<long long> t(8192);  // 2 x 256 steps x 16
    PSBA_HIP(h, hipMemcpyAsync(t.data(), a.tim, sizeof(long long) * 8192, hipMemcpyDeviceToHost, h->stream));
    PSBA_HIP(h, hipStreamSynchronize(h->stream));
    for (int w = 0; w < 2; w++) {
      const long long *q = t.data() + 4096 * w;
      fprintf(stderr, "[ring timing] workgroup %d: step | consumer after MID, products done | mover start, waited, issued, list load | prepper after MID, prepared\n", w ? 100 : 0);
      for (int s2 = 0; s2 < 256 && q[16 * s2]; s2++) {
        fprintf(stderr, "  %3d |", s2);
        for (int k = 0; k < 8; k++) fprintf(stderr, " %7lld%s", q[16 * s2 + k] - q[0], k == 1 || k == 5 ? " |" : "");
        fprintf(stderr, "\n");
      }
    }
  }
  return PSBA_OK;
}

}  // namespace psba
