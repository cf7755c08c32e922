// psba_internal.h -- the handle behind include/psba_hip.h and the kernel launch prototypes.
// Takes the place of PSBA_struct (reference PSBA/cl_psba.h:9-89) and the 31 cl_mem
// allocations of setup_cl (PSBA/cl_psba.cpp:40-84).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>
#include <vector>

#include "../../include/psba_hip.h"

namespace psba {

constexpr int TILE_OBS = 256;   // observations per point-aligned tile (= threads per workgroup)
constexpr int TILE_PTS = 128;   // points per tile (bounds the per-point LDS rows)
constexpr int MAX_GROUPS = 128; // row-aligned groups of the LDS-resident S partition (K2) are tried up to this count, then ranges of blocks
constexpr int CAM_ACC = 27;     // per-camera accumulators: 21 (sym U) + 6 (g_a)
constexpr int NSCAL = 104;      // device scalar block (doubles)
constexpr int SC_NPART = 16;    // K3's four sums arrive in 16 partial sets (same-address atomics serialise)

// slots of the device scalar block
enum {
  SC_COST = 0,      // ||e||^2 of psba_residual
  SC_DP_L2 = 1,
  SC_GAIN_DEN = 2,
  SC_NEW_COST = 3,
  SC_NEWP_L2 = 4,
  SC_MAXDIAG = 7,
  // [8..9] alias the status words (ints)
  SC_PART = 16,      // K3: [SC_NPART][4] partial (dp_l2, gain_den, new_cost, newp_l2), summed on the host
  SC_STATUS_V = 80,  // K3: 1.0 when some V_i was singular in this try (summable over ranks)
  SC_STATUS_SPD = 81,  // K3: 1.0 when the Cholesky of this try failed
  SC_TR_DOTS = 84,     // trust-region operators: (Jx1.Jx1, Jx1.Jx2, Jx2.Jx2)
  SC_CHOLMOD = 88,     // modified Cholesky: lambda, delta, beta, block columns on the one-column route
  SC_SUMS = 96,        // psba_allreduce_scalars: up to 8 host scalars summed over the ranks
};

struct Dims {
  int nC = 0, nP = 0, nO = 0;
  int nA = 0, nB = 0, nT = 0;
  int nTiles = 0;     // tiles of the tile kernels (K1, K3): the long points' tiles are not among their descriptors
  int nTilesAll = 0;  // tiles of the contiguous partition tile_pt (incl. one per long point)
  int maxTrack = 0;
};

// one workgroup of k_schur_lds: a range of work items of one group of blocks
// item = (a - obs0) | (i - pt0) << 24 | (a - b) << 46 | position << 54: observation (24 bits), point
// (22 bits), distance to the partner observation of the same point (8 bits: a point has at most
// TILE_OBS = 256 observations), block position in the partition (10 bits)
constexpr int ITEM_OBS_BITS = 24, ITEM_PT_BITS = 22, ITEM_BOFF_BITS = 8, ITEM_POS_BITS = 10;
struct SchurWg {
  int group, nblk;      // group of blocks; blocks in its LDS partition
  int obs0, pt0;        // the items' observation / point numbers are relative to these
  long long item0, item1;
  unsigned long long slab_off;  // where this workgroup's partition goes in psba_ctx::slab
  long long itemD;      // [item0, itemD): pair items (one a-side, two partners, see PAIR_*_BITS); [itemD, item1): one product each
};
// a pair item: one observation a and its partners a - boff and a - boff + 1 of the same point (W_a, V*^-1, Y_a and
// the e_a terms are formed once for two products): a - obs0, i - pt0, boff >= 1, the two block positions
constexpr int PAIR_OBS_BITS = 18, PAIR_PT_BITS = 16;  // (+ ITEM_BOFF_BITS + 2 * ITEM_POS_BITS = 62 bits)
constexpr unsigned long long SCHUR_NULL_ITEM = ~0ull;
// what k_schur_reduce needs to know about the group a run of 16 partition positions belongs to
struct ReduceGroup {
  int nwg, nblk;            // slabs of the group, blocks in its partition
  long long pos0;           // first (global) position of the group
  unsigned long long slab;  // first double of the group's slabs
  unsigned long long pad;
};

// K2's route for many cameras (no LDS partition can hold a useful part of S): one thread per
// unit = (block of the lower block triangle, a segment of that block's product list); the
// thread accumulates its 6x6 (and, for a diagonal block, e_a) in registers.  Units are sorted
// by length; the 64 units of a wave read their products from an ELL-style table
// prod[(row0 + t) * 64 + lane] (-1 padded), so the list loads are coalesced.
struct OwnerWave {
  long long row0;  // first ELL row of this wave
  int len;         // rows (= longest unit of the wave)
  int pad;
};
struct OwnerUnit {
  int j, k;        // block (j, k), k <= j
  int multi;       // 1: the block has several units -> atomic adds into the zeroed S; 0: plain stores
  int slot;        // index of the block in the block-sparse list (psba_set_solver PSBA_SOLVER_PCG); -1: idle lane of the last wave
};
struct OwnerPlanHost {
  std::vector<int2> prod;          // (a, b) observation indices; a = -1: padding
  std::vector<OwnerWave> waves;
  std::vector<OwnerUnit> units;    // 64 per wave
  std::vector<int2> blocks;        // the non-empty blocks (j, k), k <= j, plus every diagonal block: canonical order
  std::vector<int> diag_slot;      // [nC] index of block (j, j) in `blocks`
  long long products = 0;
};
#ifdef PSBA_BUILD_EXPERIMENTS
// ---- K2 ring route (few cameras), an experiment of round 3 that lost: schur_ring_plan.cpp /
// kernels_schur_ring.hip, compiled only with PSBA_BUILD_EXPERIMENTS=1 (psba_amd/build.py) ----
constexpr int RING_LANES = 256;      // consumer lanes of a workgroup (4 waves); every lane owns one 6x6 block
constexpr int RING_MOVERS = 4;       // waves that stream W records into the LDS ring with LDS-DMA
constexpr int RING_PREPPERS = 4;     // waves that form Y_a = W_a V*^-1 (two lanes per job)
constexpr int RING_PREP_JOBS = 32;   // jobs per prepper wave and step
constexpr int RING_PAGE = 7;         // W records per page load at most: one LDS-DMA instruction moves up to 63 x 16 B, contiguous in W and in LDS
constexpr int RING_SLOTS = 992;      // 144-byte LDS slots in all: W records first, Y behind them
constexpr int RING_MAXOPS = 16;      // page loads per mover wave and step at most
constexpr unsigned RING_NULL_ENTRY = 0xFFFFFFFFu;
struct RingStep { int op_begin, op_end, job_begin, job_end; };  // the page loads and jobs of a step (relative to the workgroup's lists)
struct RingJob {
  int obs, point;   // the a-side observation and its point
  int slots;        // slot of W_a | Y slot << 16
  int earow;        // row (relative to the workgroup's first) whose e_a this job feeds, -1: not here
};
struct RingWg {
  int blk0, nblk;   // blocks [blk0, blk0 + nblk) of the canonical order tri(j) + k
  int row0, nrows;  // camera rows those blocks lie in
  int copy, nsteps; // which copy of the packed triangle the sums go to; steps (even)
  int nwslots, pad; // W slots of this workgroup's LDS; the Y slots follow them
  long long step0, ent0, op0, job0, lane0, bl0;  // first entry of this workgroup in the plan's lists
};
struct RingPlanHost {
  int nR = 0, nS = 0, nWg = 0, lat = 2;
  std::vector<int> rb;              // block range r = [rb[r], rb[r+1])
  std::vector<RingWg> wgs;          // range-major: index r * nS + s
  std::vector<RingStep> steps;
  std::vector<unsigned> entries;    // [step][RING_LANES]: partner slot | Y slot << 16, or RING_NULL_ENTRY
  std::vector<int> ops;             // [op][2]: first observation of the run of records, first slot | records << 16
  std::vector<RingJob> jobs;
  std::vector<int> lane_blk;        // [wg][RING_LANES] lane -> local block (-1: idle)
  std::vector<int> blk_lane0;       // [wg][nblk + 1] first lane of each local block
  long long products = 0, slots = 0, loaded_recs = 0;
};
#endif  // PSBA_BUILD_EXPERIMENTS
struct SchurPlanHost {
  std::vector<unsigned long long> items;
  std::vector<SchurWg> wgs;
  std::vector<int> blockpos;   // block (j,k) -> position in its group's partition
  std::vector<int> posblock;   // inverse, groups concatenated: (j << 16) | k, -1 = padding
  size_t slab_doubles = 0;
  long long real_items = 0;
  // runs layout (clustered tracks, see build_schur_plan): a workgroup's items as [turn][RUN_THREADS], every
  // thread's consecutive items grouped into runs of one block position; tasks = number of such runs
  bool runs = false;
  long long tasks = 0;
  long long pair_items = 0;  // rows layout: items that carry two products of one observation (SchurWg::itemD)
};
constexpr int RUN_THREADS = 512;  // threads of a workgroup of k_schur_lds_runs: 192 VGPRs with the 36 accumulators, two waves per SIMD (768 threads = 168 VGPRs spill 27 registers: 85 against 59 us)
constexpr int RUN_MAX = 32;       // products a lane sums in registers before it touches the LDS at the latest
}  // namespace psba

struct psba_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  psba::Dims d;
  // multi-GPU
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;

  // ---- device memory (all owned here) ----
  double *camconst = nullptr;   // [nC][9]  K5 | q0(4)               (Kparas_buffer, initcams_buffer)
  double *cams[2] = {nullptr, nullptr};  // [nC][6] cur / proposed   (cams_buffer, newCams_buffer)
  double *pts[2] = {nullptr, nullptr};   // [nP][3] cur / proposed   (pts3D_buffer, newPts3D_buffer)
  double *params0 = nullptr;    // [nT] the parameters as uploaded (psba_reset_params)
  double *impts = nullptr;      // [nO][2]                           (impts_buffer)
  int *iidx = nullptr;          // [nO] point of each observation    (iidx_buffer)
  int *jidx = nullptr;          // [nO] camera of each observation   (jidx_buffer)
  int *ptr = nullptr;           // [nP+1] point CSR over observations (replaces blkIdx_buffer)
  int *tile_pt = nullptr;       // [nTiles+1] first point of each tile
  int *long_pts = nullptr;      // [nLong] points seen by more than TILE_OBS cameras (tiles of their own, handled by the *_long kernels)
  int nLong = 0;
  int4 *tile_desc = nullptr;    // [nTiles] (first point, end point, first observation, end observation): one load instead of a chain
  double *W = nullptr;          // [nO][18] W_ij = coeff A^T B        (W_buffer)
  double *PV = nullptr;         // [nP][9]  V_i sym6 | g_b,i          (V_buffer + g_buffer tail)
  double *U = nullptr;          // [nC][36]                           (U_buffer)
  double *ga = nullptr;         // [nA]                               (g_buffer head)
  // second set of linearization outputs, written by psba_linearize_ahead for the proposed
  // parameters while the host decides about the step; psba_accept swaps the sets
  double *W_alt = nullptr, *PV_alt = nullptr, *U_alt = nullptr, *ga_alt = nullptr;
  bool ahead = false;           // the alternate set holds the linearization at the proposed parameters
  bool lin_is_ahead = false;    // the current set was computed ahead: the next psba_linearize is a no-op
  double *h_scal_dev = nullptr;     // device address of the pinned host block h_scal
  hipEvent_t scal_event = nullptr;  // recorded behind the scalar copy of psba_backsub_async
  hipStream_t stream2 = nullptr;    // with a communicator: the try's scalar all-reduce + copy run here
  hipEvent_t k3_event = nullptr;    // K3 done (main stream) -> side stream
  hipStream_t chol_side = nullptr;      // lowest-priority stream of the far updates (look-ahead of the blocked chain)
  std::vector<hipEvent_t> chol_events;  // the look-ahead of the blocked Cholesky chain (two per super-panel)
  bool scal_side = false;           // side-stream work the main stream has not been ordered behind yet
  double *campart = nullptr;    // [nPart][nC][27] per-workgroup camera partial sums
  int nPart = 0;
  // many cameras (the 27 per-camera accumulators no longer fit a workgroup's LDS): K1 adds its
  // camera sums with global fp64 atomics into camacc [nC][27] (zeroed per launch) instead
  bool cam_global = false;
  double *camacc = nullptr;
  // ... and those sums are formed by a camera-major pass: thread = (camera, segment of its
  // observations), 27 sums in registers, one set of atomic adds per segment
  int *cam_obs = nullptr;       // [nO] observation indices sorted by camera (stable: points ascending)
  int4 *cam_units = nullptr;    // [nCamUnits] (camera, first, end in cam_obs, 0)
  int nCamUnits = 0;
  // padded reduce buffer Lw[(n32+16)][n32], n32 = nA rounded up to 32: rows < nA = S (row stride
  // n32), rows nA..n32-1 identity padding, row n32 = ea, rows above zero (S_buffer, eab_buffer)
  double *red = nullptr;
  int n32 = 0;
  // K2 (schur) static schedule, built once per problem by schur_plan.cpp
  int nGroups = 0;              // groups of blocks (one LDS partition each); 0: the owner route
  int nWg = 0;                  // workgroups of k_schur_lds
  std::vector<int> gblk0;       // group g owns the blocks [gblk0[g], gblk0[g+1]) of the canonical order tri(j) + k
  std::vector<int> gnwg;        // workgroups (= slabs) of group g
  std::vector<int> gnblk;       // blocks in group g's LDS partition (padded to 16)
  std::vector<size_t> gslab;    // first double of group g's slabs
  psba::ReduceGroup *gtab = nullptr;  // device, for k_schur_reduce: one entry per run of 16 partition positions
  psba::SchurWg *wg = nullptr;  // [nWg] ordered by position in the point sequence
  unsigned long long *items = nullptr;   // work items, one per product Y_a W_b^T (or null)
  // single rank: the K2 workgroups add their copies of the 21 blocks of the first 32x32 diagonal
  // block into diag0 (global atomics) while flushing, and one extra workgroup of the S-reduce
  // kernel factors that block -- the first step of the Cholesky chain off the critical path
  double *redp = nullptr;       // with a communicator: the packed sums (slab order, lower block triangle + e_a) that are all-reduced
  size_t packed_doubles = 0;
  bool packed_pending = false;  // this try's sums sit in redp, not yet in red
  double *diag0 = nullptr;      // [21 * 36], zero between tries
  int h_diagpos[21] = {0};      // partition positions of the blocks (j, k), j <= 5
  int h_diaggrp[21] = {0};      // and their groups (-1: no such camera)
  bool diag_done = false;       // this try's S-reduce kernel has factored the first diagonal block
  int *posblock = nullptr;      // per group, per partition position: (j << 16) | k of the block there, -1 = padding
  double *slab = nullptr;       // per workgroup: its group's partition, 36 doubles per position
  size_t packedN = 0;           // 36 * nC (nC+1) / 2 doubles: packed lower block triangle of S
  int ring_nWg = 0, ring_nS = 0;  // (workgroups of the experimental ring route; 0 in the product build)
#ifdef PSBA_BUILD_EXPERIMENTS
  // K2 ring route (few cameras): see RingPlanHost
  psba::RingWg *ring_wg = nullptr;
  psba::RingStep *ring_steps = nullptr;
  unsigned *ring_entries = nullptr;
  int *ring_ops = nullptr;
  psba::RingJob *ring_jobs = nullptr;
  int *ring_bl0 = nullptr;
  int *ring_canon = nullptr;    // [tri(nC)] (j << 16) | k of every block of the canonical order
  double *ring_slab = nullptr;  // [ring_nS][packedN] copies of the packed triangle
  double *ring_pvi = nullptr;   // [nP][9] (V_i + mu I)^-1 (sym6) | (V_i + mu I)^-1 g_b,i of the current try (k_schur_vinv)
  long long ring_products = 0, ring_slots = 0;
  size_t ring_loaded_recs = 0;
  bool ring_attr_set = false;
  long long *chol_tim_ring = nullptr;  // dev instrumentation (PSBA_RING_TIMING): per-step s_memtime stamps of two workgroups
#endif
  // block-sparse S + preconditioned CG (psba_set_solver, kernels_pcg.hip)
  int solver = 0;               // PSBA_SOLVER_*
  bool schur_runs = false;  // K2's items are in the runs layout (k_schur_lds_runs)
  bool schur_pairs = false; // K2's item lists begin with pair items (SchurWg::itemD)
  int cnp = 6;  // parameters per camera: 6 (fixed intrinsics, the reference's kernels) or 11 (psba_set_camera_model: free intrinsics)
  double pcg_tol = 1e-10;
  int pcg_maxit = 500, pcg_iters = 0;
  double pcg_relres = 0.0;
  bool pcg_exhausted = false;  // the last solve used up max_iter without reaching tol
  long long bs_nblk = 0;
  double *bs_val = nullptr;     // [bs_nblk][36] | e_a [nA] right behind (one all-reduce)
  double *bs_ea = nullptr;      // = bs_val + 36 bs_nblk
  int2 *bs_jk = nullptr;        // [bs_nblk] (j, k)
  int *bs_diag = nullptr;       // [nC] index of block (j, j)
  // the full symmetric pattern by block row, for the product S p without atomics: row j's entries
  // [bs_rowptr[j], bs_rowptr[j + 1]) = (slot of the stored block, other camera | how to read it << 28:
  // 0 as stored (j is the block's row), 1 transposed (j is its column), 2 the diagonal block)
  int *bs_rowptr = nullptr;
  int2 *bs_rowent = nullptr;
  double *pcg_vec = nullptr;    // r | z | p | q, nA each
  double *pcg_minv = nullptr;   // [nC][36] inverses of the diagonal blocks
  double *pcg_scal = nullptr, *pcg_host = nullptr;  // device scalars and their pinned mirror
  std::vector<unsigned char> bs_pattern;  // sharded points without a communicator: the union of all ranks' blocks (psba_set_sparse_pattern)
  // K2 owner route (many cameras): see OwnerPlanHost
  int2 *own_prod = nullptr;
  psba::OwnerWave *own_waves = nullptr;
  psba::OwnerUnit *own_units = nullptr;
  int own_nwaves = 0;
  long long own_products = 0;
  double *dp = nullptr;         // [nT] dpa | dpb                     (dp_buffer)
  double *trv[2] = {nullptr, nullptr};  // [nT] each: vectors of the trust-region operators (allocated on first use)
  double *jmul_out = nullptr;   // [2 nO] J x of psba_compute_Jmultiply (allocated on first use)
  hipGraphExec_t chol_graph[2] = {nullptr, nullptr};  // captured panel chain of kernels_chol_graph.hip, with / without its first step
  int chol_graph_n32[2] = {0, 0};
  double *chol_graph_red[2] = {nullptr, nullptr};
  long long *chol_tim = nullptr; // dev instrumentation: per-phase s_memtime ticks of the last solve (PSBA_CHOL_TIMING)
  double *chol_L = nullptr;     // [(2 n32+16)][n32] panel chain: the Cholesky factor | forward-solved e_a row | L^-T
  double *dist_buf = nullptr;   // sharded factorization: packed column blocks of one super-panel (allocated on first use)
  double *chol_ws = nullptr;    // [ceil(nA/32)][32*32] inverses of the diagonal blocks of L (diagBlkAux_buffer)
  double *scal = nullptr;       // [NSCAL]
  // [4] generation stamps, never zeroed: [0] == try_id <=> some V_i singular in this try,
  // [1] == try_id <=> the Cholesky of this try failed; [3] = try_id as seen by the graph-replayed
  // Cholesky kernels (written by the K2 reduce kernel).  Lives in scal[8..9]
  int *status = nullptr;
  int try_id = 0;
  double *h_scal = nullptr;     // pinned mirror of scal
  int *h_status = nullptr;      // pinned mirror of status
  // debug dumps for the sba_func.h mirror (allocated on first use)
  double *dbg_ex = nullptr, *dbg_JA = nullptr, *dbg_JB = nullptr, *dbg_Y = nullptr,
         *dbg_Vinv = nullptr, *dbg_eb = nullptr;

  // kernels whose dynamic LDS exceeds 64 KiB need hipFuncSetAttribute once per device: kept per
  // handle (one handle = one device), not per process
  bool lds_attr_set = false, atomic_attr_set = false, chol_attr_set = false, lin_attr_set = false;

  // ---- state ----
  bool uploaded = false, linearized = false, assembled = false, solved = false, backsubbed = false;
  bool backsub_pending = false;  // psba_backsub_async issued, psba_backsub_wait not yet
  // the try's scalars travel to the host with the linearization queued ahead (its workgroup 0 copies
  // them and then writes a stamp the host polls) instead of through a kernel of their own between K3
  // and that linearization: publish_deferred = K3 queued, nothing published yet; publish_in_k1 = the
  // queued K1 carries them, the host waits for pub_seq in h_scal[NSCAL]
  bool publish_deferred = false, publish_in_k1 = false;
  double pub_seq = 0.0;
  int cur = 0;                  // index of the current parameter set in cams[]/pts[]
  double coeff = 1.0, coeff_g = 1.0, mu = 0.0;
  double coeff_w = 1.0;         // the coefficient the stored W was formed with (last K1 launch)
  bool mu_applied = false;      // update_UV called (fine-grained mirror only)

  // ---- profiling ----
  unsigned prof = 0;            // bit k set: time kernel class k with HIP events
  struct Span { hipEvent_t a, b; int kind; };
  std::vector<Span> spans;
  size_t spans_used = 0;
  double prof_ms[PSBA_K_COUNT] = {0};
  int prof_n[PSBA_K_COUNT] = {0};
};

namespace psba {

// error helpers (psba_api.cpp)
int fail(psba_ctx *h, int code, const char *fmt, ...);
#define PSBA_HIP(h, call)                                                              \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess)                                                             \
      return psba::fail((h), PSBA_E_HIP, "%s failed: %s (%s:%d)", #call,               \
                        hipGetErrorString(e__), __FILE__, __LINE__);                   \
  } while (0)

struct ProfScope {
  psba_ctx *h;
  int idx = -1;
  ProfScope(psba_ctx *h, int kind);
  ~ProfScope();
};

// ---- kernel launchers (one per .hip file) ----
// kernels_linearize.hip
int launch_linearize(psba_ctx *h, bool dump, bool ahead = false, bool publish = false);
int launch_residual(psba_ctx *h, int which, double *ex_out_dev);
int launch_max_diag(psba_ctx *h);
// kernels_schur.hip
int build_schur_plan(psba_ctx *h, int nCams, int nPts, int nObs, const int *iidx, const int *jidx,
                     const int *ptr, SchurPlanHost &out);
int build_owner_plan(int nCams, int nObs, const int *iidx, const int *jidx, const int *ptr, OwnerPlanHost &out,
                     const unsigned char *pattern = nullptr);
int sparse_pattern(int nCams, int nObs, const int *iidx, const int *jidx, const int *ptr, unsigned char *flags);
#ifdef PSBA_BUILD_EXPERIMENTS
int build_ring_plan(int nCams, int nPts, int nObs, const int *iidx, const int *jidx, const int *ptr, RingPlanHost &out,
                    bool force = false);
int launch_schur_ring(psba_ctx *h, double mu, bool dump);
struct SchurLdsArgs;
bool launch_schur_lds_mode(int mode, dim3 G, dim3 B, size_t lds, hipStream_t s, const SchurLdsArgs &a);  // kernels_schur_modes.hip
#endif
int launch_schur(psba_ctx *h, double mu, bool dump);
int launch_schur_expand(psba_ctx *h);
// kernels_chol.hip
int launch_chol_solve(psba_ctx *h);
// kernels_freek.hip: the 11-parameter camera block (free intrinsics), one plain route
int launch_linearize_fk(psba_ctx *h, bool ahead, bool publish);
int launch_residual_fk(psba_ctx *h, int which);
int launch_max_diag_fk(psba_ctx *h);
int launch_schur_fk(psba_ctx *h, double mu);
int launch_backsub_fk(psba_ctx *h, double mu);
// kernels_pcg.hip
int launch_bsr_finalize(psba_ctx *h, double mu);
int launch_pcg_solve(psba_ctx *h);
int launch_bsr_gershgorin(psba_ctx *h, double *lambda, double *info3);
// kernels_chol_graph.hip
int launch_chol_graph(psba_ctx *h);
int chol_dist_shape(psba_ctx *h, int *NB, int *blocked);
int chol_dist_exchange_plan(int n32, int NB, int nranks, int JE, long long (*out)[4], int cap);
int chol_dist_begin(psba_ctx *h);
int chol_dist_superpanel(psba_ctx *h, int J);
int chol_dist_block(psba_ctx *h, int B, double *buf_dev, int set);
int chol_dist_finish(psba_ctx *h);
// kernels_backsub.hip
int launch_backsub(psba_ctx *h, double mu, bool dump);
int launch_publish_scal(psba_ctx *h, hipStream_t s);
// kernels_tr.hip
int launch_jmul(psba_ctx *h, const double *x1_dev, const double *x2_dev, double *out1_dev, double *dots_dev);
int launch_pack_g(psba_ctx *h, double *g_dev);
int launch_newp(psba_ctx *h, const double *dp_dev);
int launch_cholmod(psba_ctx *h, double *out4_dev);

}  // namespace psba
