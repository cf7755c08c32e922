// psba_api.cpp -- the thin C-ABI HIP host layer (include/psba_hip.h).
// Takes the place of PSBA/cl_psba.cpp (runtime, buffers, uploads), PSBA/sba_func.cpp (one
// host wrapper per kernel), PSBA/cl_spdinv.cpp + PSBA/cl_linearalg.cpp (dense solve) and
// PSBA/misc.cpp:178-217 (index generation) of the reference.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <new>
#include <vector>

#include "psba_internal.h"

using namespace psba;

static thread_local std::string g_create_err;

namespace psba {

int fail(psba_ctx *h, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h)
    h->err = buf;
  else
    g_create_err = buf;
  return code;
}

ProfScope::ProfScope(psba_ctx *hh, int kind) : h(hh) {
  if (kind < 0 || !(h->prof & (1u << kind))) return;
  if (h->spans_used == h->spans.size()) {
    psba_ctx::Span s;
    s.kind = kind;
    if (hipEventCreateWithFlags(&s.a, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&s.b, hipEventDisableSystemFence) != hipSuccess)
      return;
    h->spans.push_back(s);
  }
  idx = (int)h->spans_used++;
  h->spans[idx].kind = kind;
  (void)hipEventRecord(h->spans[idx].a, h->stream);
}
ProfScope::~ProfScope() {
  if (idx >= 0) (void)hipEventRecord(h->spans[idx].b, h->stream);
}

}  // namespace psba

// every entry point makes the handle's device current: a host may keep handles on several GPUs
// in one process, or call from a thread whose current device is another one
// With a communicator the try's scalar all-reduce and its copy to the host run on a side stream
// (psba_backsub_async), beside the linearization that is queued ahead; whatever else is queued on
// the main stream before psba_backsub_wait has returned must come after them.
#define CHECK_H_NOJOIN(h)                 \
  do {                                    \
    if (!(h)) return PSBA_E_INVALID;      \
    (void)hipSetDevice((h)->device);      \
  } while (0)
#define CHECK_H(h)                                                         \
  do {                                                                     \
    CHECK_H_NOJOIN(h);                                                     \
    if ((h)->scal_side) {                                                  \
      (void)hipStreamWaitEvent((h)->stream, (h)->scal_event, 0);           \
      (h)->scal_side = false;                                              \
    }                                                                      \
  } while (0)
#define NEED(h, cond, what)                                             \
  do {                                                                  \
    if (!(cond)) return fail((h), PSBA_E_STATE, "%s: %s", __func__, what); \
  } while (0)
#define RCCL(h, call)                                                                   \
  do {                                                                                  \
    ncclResult_t r__ = (call);                                                          \
    if (r__ != ncclSuccess)                                                             \
      return fail((h), PSBA_E_RCCL, "%s failed: %s", #call, ncclGetErrorString(r__));   \
  } while (0)
#define TRY(expr)                 \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ < 0) return rc__;    \
  } while (0)

template <typename T>
static int dev_alloc(psba_ctx *h, T **p, size_t n) {
  PSBA_HIP(h, hipMalloc((void **)p, sizeof(T) * (n ? n : 1)));
  // PSBA_DEBUG_POISON=1 (tests): fresh allocations filled with 0xFF bytes (NaN as doubles, -1 as ints), so that
  // anything read before it is written shows in the results instead of depending on what the allocator recycled
  static const bool poison = getenv("PSBA_DEBUG_POISON") != nullptr;
  if (poison) {
    PSBA_HIP(h, hipMemsetAsync(*p, 0xFF, sizeof(T) * (n ? n : 1), h->stream));
    PSBA_HIP(h, hipStreamSynchronize(h->stream));
  }
  return PSBA_OK;
}
template <typename T>
static void dev_free(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

static void free_problem_buffers(psba_ctx *h) {
  for (int v = 0; v < 2; v++)
    if (h->chol_graph[v]) {
      (void)hipGraphExecDestroy(h->chol_graph[v]);
      h->chol_graph[v] = nullptr;
    }
  dev_free(h->camconst);
  dev_free(h->cams[0]);
  dev_free(h->cams[1]);
  dev_free(h->pts[0]);
  dev_free(h->pts[1]);
  dev_free(h->impts);
  dev_free(h->iidx);
  dev_free(h->jidx);
  dev_free(h->ptr);
  dev_free(h->params0);
  dev_free(h->tile_pt);
  dev_free(h->tile_desc);
  dev_free(h->long_pts);
  h->nLong = 0;
  dev_free(h->W);
  dev_free(h->W_alt);
  dev_free(h->PV_alt);
  dev_free(h->U_alt);
  dev_free(h->ga_alt);
  dev_free(h->PV);
  dev_free(h->U);
  dev_free(h->ga);
  dev_free(h->campart);
  dev_free(h->camacc);
  dev_free(h->cam_obs);
  dev_free(h->cam_units);
  h->nCamUnits = 0;
  dev_free(h->red);
  dev_free(h->items);
  dev_free(h->wg);
  dev_free(h->posblock);
  dev_free(h->gtab);
  dev_free(h->diag0);
  dev_free(h->redp);
  dev_free(h->slab);
#ifdef PSBA_BUILD_EXPERIMENTS
  dev_free(h->ring_wg);
  dev_free(h->ring_steps);
  dev_free(h->ring_entries);
  dev_free(h->ring_ops);
  dev_free(h->ring_jobs);
  dev_free(h->ring_bl0);
  dev_free(h->ring_canon);
  dev_free(h->ring_slab);
  dev_free(h->ring_pvi);
#endif
  h->ring_nWg = h->ring_nS = 0;
  dev_free(h->bs_val);
  h->bs_ea = nullptr;
  dev_free(h->bs_jk);
  dev_free(h->bs_diag);
  dev_free(h->bs_rowptr);
  dev_free(h->bs_rowent);
  dev_free(h->pcg_vec);
  dev_free(h->pcg_minv);
  dev_free(h->pcg_scal);
  h->bs_nblk = 0;
  dev_free(h->own_prod);
  dev_free(h->own_waves);
  dev_free(h->own_units);
  h->own_nwaves = 0;
  dev_free(h->dp);
  dev_free(h->trv[0]);
  dev_free(h->trv[1]);
  dev_free(h->jmul_out);
  dev_free(h->chol_ws);
  dev_free(h->dist_buf);
  dev_free(h->chol_L);
  dev_free(h->dbg_ex);
  dev_free(h->dbg_JA);
  dev_free(h->dbg_JB);
  dev_free(h->dbg_Y);
  dev_free(h->dbg_Vinv);
  dev_free(h->dbg_eb);
  h->uploaded = h->linearized = h->assembled = h->solved = h->backsubbed = false;
}

extern "C" {

const char *psba_version(void) { return "psba_hip 0.1 (gfx950, fp64)"; }

const char *psba_last_error(psba_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

// the handle's stream at the HIGHEST priority: the library's only concurrent work is the far updates of the blocked
// Cholesky chain (lowest priority, kernels_chol_graph.hip), and the main stream's small dependent launches must win a
// freed CU against them (PSBA_STREAM_PRIO_DEFAULT=1: default priority)
static hipError_t create_main_stream(hipStream_t *s) {
  int least = 0, greatest = 0;
  if (getenv("PSBA_STREAM_PRIO_DEFAULT") || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess)
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

int psba_create(int device, psba_handle *out) {
  if (!out) return PSBA_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, PSBA_E_HIP, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev)
    return fail(nullptr, PSBA_E_INVALID, "device %d out of range (%d devices)", device, ndev);
  psba_ctx *h = new psba_ctx();
  h->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess ||
      (e = create_main_stream(&h->stream)) != hipSuccess ||
      (e = hipMalloc((void **)&h->scal, sizeof(double) * NSCAL)) != hipSuccess ||
      (e = hipHostMalloc((void **)&h->h_scal, sizeof(double) * (NSCAL + 8))) != hipSuccess ||  // (+ the publish stamp)
      (e = hipEventCreateWithFlags(&h->scal_event, hipEventDisableTiming)) != hipSuccess) {
    int rc = fail(nullptr, PSBA_E_HIP, "psba_create: %s", hipGetErrorString(e));
    delete h;
    return rc;
  }
  if (hipHostGetDevicePointer((void **)&h->h_scal_dev, h->h_scal, 0) != hipSuccess) h->h_scal_dev = nullptr;
  // the status stamps live in the tail of the scalar block so that one copy fetches both
  h->status = reinterpret_cast<int *>(h->scal + 8);
  h->h_status = reinterpret_cast<int *>(h->h_scal + 8);
  // (on the handle's stream and waited for: hipMemset on the null stream is asynchronous to the host for device
  // memory and a non-blocking stream does not wait for the null stream -- a late memset would wipe the status
  // stamps and the partial sums of a try already running)
  (void)hipMemsetAsync(h->scal, 0, sizeof(double) * NSCAL, h->stream);
  (void)hipStreamSynchronize(h->stream);
  h->h_scal[NSCAL] = 0.0;  // the publish stamp (psba_backsub_wait)
  *out = h;
  return PSBA_OK;
}

int psba_destroy(psba_handle h) {
  CHECK_H(h);
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  if (h->comm) ncclCommDestroy(h->comm);
  if (h->k3_event) (void)hipEventDestroy(h->k3_event);
  for (hipEvent_t e : h->chol_events) (void)hipEventDestroy(e);
  if (h->chol_side) {
    (void)hipStreamSynchronize(h->chol_side);
    (void)hipStreamDestroy(h->chol_side);
  }
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  free_problem_buffers(h);
  dev_free(h->scal);
  if (h->h_scal) (void)hipHostFree(h->h_scal);
  if (h->pcg_host) (void)hipHostFree(h->pcg_host);
  if (h->scal_event) (void)hipEventDestroy(h->scal_event);
  for (auto &s : h->spans) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PSBA_OK;
}

int psba_schur_path(psba_handle h, int *path) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  if (path) *path = h->solver == PSBA_SOLVER_PCG ? 4 : getenv("PSBA_SCHUR_ATOMIC") ? 2 : h->ring_nWg > 0 ? 3 : h->nGroups > 0 ? 0 : 1;
  return PSBA_OK;
}

int psba_set_camera_model(psba_handle h, int model) {
  CHECK_H(h);
  if (model != PSBA_CAMERA_FIXED_K && model != PSBA_CAMERA_FREE_K) return fail(h, PSBA_E_INVALID, "unknown camera model %d", model);
  NEED(h, !h->uploaded, "psba_set_camera_model before psba_upload_problem (every buffer depends on the camera block)");
  h->cnp = model == PSBA_CAMERA_FREE_K ? 11 : 6;
  return PSBA_OK;
}

int psba_camera_block(psba_handle h, int *cnp) {
  CHECK_H(h);
  if (cnp) *cnp = h->cnp;
  return PSBA_OK;
}

int psba_get_dims(psba_handle h, int *nCams, int *n3Dpts, int *n2Dprojs) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  if (nCams) *nCams = h->d.nC;
  if (n3Dpts) *n3Dpts = h->d.nP;
  if (n2Dprojs) *n2Dprojs = h->d.nO;
  return PSBA_OK;
}

int psba_upload_problem(psba_handle h, int nCams, int n3Dpts, int n2Dprojs, const double *Kparas,
                        const double *impts, const double *initrot, const double *camsEx,
                        const double *pts3D, const int *iidx, const int *jidx) {
  CHECK_H(h);
  if (nCams <= 0 || n3Dpts <= 0 || n2Dprojs <= 0 || !Kparas || !impts || !initrot || !camsEx ||
      !pts3D || !iidx || !jidx)
    return fail(h, PSBA_E_INVALID, "psba_upload_problem: null pointer or non-positive size");
  PSBA_HIP(h, hipSetDevice(h->device));
  // ---- index build: point CSR + point-aligned tiles (replaces generate_idxs) ----
  std::vector<int> ptr((size_t)n3Dpts + 1, 0);
  for (int a = 0; a < n2Dprojs; a++) {
    const int i = iidx[a], j = jidx[a];
    if (i < 0 || i >= n3Dpts || j < 0 || j >= nCams)
      return fail(h, PSBA_E_INVALID, "observation %d: point %d / camera %d out of range", a, i, j);
    if (a > 0 && (i < iidx[a - 1] || (i == iidx[a - 1] && j <= jidx[a - 1])))
      return fail(h, PSBA_E_INVALID,
                  "observations must be sorted point-major with ascending, distinct cameras "
                  "inside a point (violated at observation %d)", a);
    ptr[(size_t)i + 1]++;
  }
  int maxTrack = 0;
  for (int i = 0; i < n3Dpts; i++) {
    if (ptr[(size_t)i + 1] > maxTrack) maxTrack = ptr[(size_t)i + 1];
    ptr[(size_t)i + 1] += ptr[i];
  }
  // point-aligned tiles of at most TILE_OBS observations / TILE_PTS points.  A point seen by more
  // cameras than a tile holds (the reference has no such limit: CL_files/compute_V.cl:6-38 loops
  // over all cameras) is a tile of its own that the tile kernels skip; one workgroup per such point
  // walks its observations in the *_long kernels instead.
  std::vector<int> tile_pt, long_pts;
  tile_pt.push_back(0);
  {
    int p0 = 0;
    while (p0 < n3Dpts) {
      int p1 = p0;
      if (ptr[(size_t)p0 + 1] - ptr[p0] > TILE_OBS) {
        long_pts.push_back(p0);
        p1 = p0 + 1;
      } else {
        while (p1 < n3Dpts && (ptr[(size_t)p1 + 1] - ptr[p0]) <= TILE_OBS && (p1 - p0) < TILE_PTS) p1++;
      }
      tile_pt.push_back(p1);
      p0 = p1;
    }
  }
  free_problem_buffers(h);
  Dims d;
  d.nC = nCams;
  d.nP = n3Dpts;
  d.nO = n2Dprojs;
  const int cnp = h->cnp;  // 6, or 11 with free intrinsics (kernels_freek.hip)
  if (cnp != 6 && (h->solver == PSBA_SOLVER_PCG || h->nranks > 1 || h->comm))
    return fail(h, PSBA_E_INVALID, "free intrinsics: dense solver, single rank only");
  d.nA = cnp * nCams;
  d.nB = 3 * n3Dpts;
  d.nT = d.nA + d.nB;
  d.nTilesAll = (int)tile_pt.size() - 1;
  d.nTiles = d.nTilesAll - (int)long_pts.size();  // the tile kernels never see a long point's tile
  if (d.nTiles < 1) d.nTiles = 1;                 // (only long points: one empty tile keeps the grids non-empty)
  d.maxTrack = maxTrack;
  // K1 keeps 27 accumulators per camera in LDS while they fit beside its tile buffers (nC <= 455);
  // beyond that its camera sums go to global memory with fp64 atomics
  // K1's 27 sums per camera: LDS accumulators per workgroup while they leave room for three
  // workgroups per CU (up to ~220 cameras: 31 ... 36 us for K1 at 218 k observations), the
  // camera-major pass beyond (~40 us flat; the LDS form takes 55 us at 257 cameras and 64 at 455,
  // where it ends: PSBA_LIN_LDS_ACC=1 keeps it up to there)
  const size_t cam_lds_max = getenv("PSBA_LIN_LDS_ACC") ? 96 * 1024 : 48 * 1024;
  h->cam_global = cnp == 6 && ((size_t)CAM_ACC * nCams * sizeof(double) > cam_lds_max || getenv("PSBA_LIN_GLOBAL_ACC"));
  h->d = d;
  h->nPart = d.nTiles < 768 ? d.nTiles : 768;  // persistent workgroups: three per CU fit since W is staged in halves
  if (const char *e = getenv("PSBA_LIN_GRID")) h->nPart = atoi(e) > 0 && atoi(e) < d.nTiles ? atoi(e) : d.nTiles;
  h->cur = 0;

  std::vector<double> cc((size_t)nCams * 9);
  for (int j = 0; j < nCams; j++) {
    for (int k = 0; k < 5; k++) cc[(size_t)9 * j + k] = Kparas[5 * j + k];
    for (int k = 0; k < 4; k++) cc[(size_t)9 * j + 5 + k] = initrot[4 * j + k];
  }
  TRY(dev_alloc(h, &h->camconst, cc.size()));
  TRY(dev_alloc(h, &h->cams[0], (size_t)d.nA));
  TRY(dev_alloc(h, &h->cams[1], (size_t)d.nA));
  TRY(dev_alloc(h, &h->pts[0], (size_t)d.nB));
  TRY(dev_alloc(h, &h->pts[1], (size_t)d.nB));
  TRY(dev_alloc(h, &h->params0, (size_t)d.nT));
  TRY(dev_alloc(h, &h->impts, (size_t)2 * d.nO));
  TRY(dev_alloc(h, &h->iidx, (size_t)d.nO));
  TRY(dev_alloc(h, &h->jidx, (size_t)d.nO));
  TRY(dev_alloc(h, &h->ptr, (size_t)d.nP + 1));
  TRY(dev_alloc(h, &h->tile_pt, tile_pt.size()));
  TRY(dev_alloc(h, &h->tile_desc, (size_t)d.nTiles));
  h->nLong = (int)long_pts.size();
  if (h->nLong) {
    TRY(dev_alloc(h, &h->long_pts, long_pts.size()));
    PSBA_HIP(h, hipMemcpy(h->long_pts, long_pts.data(), sizeof(int) * long_pts.size(), hipMemcpyHostToDevice));
  }
  TRY(dev_alloc(h, &h->W, (size_t)3 * cnp * d.nO));
  TRY(dev_alloc(h, &h->W_alt, (size_t)3 * cnp * d.nO));
  TRY(dev_alloc(h, &h->PV, (size_t)9 * d.nP));
  TRY(dev_alloc(h, &h->PV_alt, (size_t)9 * d.nP));
  // (a point without observations is never written by K1, whose stores come from the last lane of a point's run
  // of observations: its V_i and g_b,i are the zeros put here)
  PSBA_HIP(h, hipMemsetAsync(h->PV, 0, sizeof(double) * 9 * (size_t)d.nP, h->stream));
  PSBA_HIP(h, hipMemsetAsync(h->PV_alt, 0, sizeof(double) * 9 * (size_t)d.nP, h->stream));
  TRY(dev_alloc(h, &h->U, (size_t)cnp * cnp * d.nC));
  TRY(dev_alloc(h, &h->U_alt, (size_t)cnp * cnp * d.nC));
  TRY(dev_alloc(h, &h->ga, (size_t)d.nA));
  TRY(dev_alloc(h, &h->ga_alt, (size_t)d.nA));
  h->ahead = h->lin_is_ahead = false;
  TRY(dev_alloc(h, &h->campart, (h->cam_global || cnp != 6) ? 1 : (size_t)(h->nPart + 1) * d.nC * CAM_ACC));  // (+1: the long points' slab)
  if (cnp != 6) TRY(dev_alloc(h, &h->camacc, (size_t)d.nC * (cnp * (cnp + 1) / 2 + cnp)));  // kernels_freek.hip: 66 + 11 sums per camera
  if (h->cam_global) {
    TRY(dev_alloc(h, &h->camacc, (size_t)d.nC * CAM_ACC));
    // camera-major index of the observations, cut into segments of at most 256
    std::vector<int> cptr((size_t)nCams + 1, 0), cobs((size_t)n2Dprojs);
    for (int a = 0; a < n2Dprojs; a++) cptr[(size_t)jidx[a] + 1]++;
    for (int j = 0; j < nCams; j++) cptr[(size_t)j + 1] += cptr[j];
    {
      std::vector<int> at(cptr.begin(), cptr.end() - 1);
      for (int a = 0; a < n2Dprojs; a++) cobs[(size_t)at[jidx[a]]++] = a;
    }
    std::vector<int4> units;
    const int LSEG = 256;
    for (int j = 0; j < nCams; j++)
      for (int f = cptr[j]; f < cptr[(size_t)j + 1]; f += LSEG)
        units.push_back(make_int4(j, f, std::min(f + LSEG, cptr[(size_t)j + 1]), 0));
    h->nCamUnits = (int)units.size();
    TRY(dev_alloc(h, &h->cam_obs, cobs.size()));
    TRY(dev_alloc(h, &h->cam_units, units.size()));
    PSBA_HIP(h, hipMemcpy(h->cam_obs, cobs.data(), sizeof(int) * cobs.size(), hipMemcpyHostToDevice));
    PSBA_HIP(h, hipMemcpy(h->cam_units, units.data(), sizeof(int4) * units.size(), hipMemcpyHostToDevice));
  }
  h->n32 = (d.nA + 31) / 32 * 32;
  // rows [0, n32 + 16) are the reduce buffer proper; n32 more rows below it are the working
  // space of the identity rows the panel chain carries along (kernels_chol_graph.hip)
  const bool sparse = h->solver == PSBA_SOLVER_PCG;  // no dense S, no factor: only the blocks that exist (below)
  const size_t dense = sparse ? 1 : (size_t)(2 * h->n32 + 16) * h->n32;
  TRY(dev_alloc(h, &h->red, dense));
  PSBA_HIP(h, hipMemsetAsync(h->red, 0, sizeof(double) * dense, h->stream));
  TRY(dev_alloc(h, &h->dp, (size_t)(d.nT > 36 * d.nC ? d.nT : 36 * d.nC)));
  TRY(dev_alloc(h, &h->chol_ws, (size_t)((d.nA + 31) / 32) * 1024));
  TRY(dev_alloc(h, &h->chol_L, dense));
  PSBA_HIP(h, hipMemsetAsync(h->chol_L, 0, sizeof(double) * dense, h->stream));
  if (getenv("PSBA_CHOL_TIMING") && !h->chol_tim) TRY(dev_alloc(h, &h->chol_tim, 32));  // [0..15] diag kernel + pivot wave, [16..31] inverse wave
  // ---- K2's static schedule (groups of blocks, workgroups, conflict-free item rows) ----
  if (sparse) {
    // block-sparse S: the owner route's product lists give the blocks that exist
    OwnerPlanHost op;
    // sharded points: every rank must hold the same block list, the union of what the ranks' points
    // produce -- one byte per block of the lower triangle, combined with a max all-reduce over the
    // communicator, or handed in by a host that has its own transport (psba_set_sparse_pattern)
    std::vector<unsigned char> pat;
    // (PSBA_SPARSE_PATTERN_FORCE=1: test hook -- a one-rank communicator exchanges its pattern too)
    if (h->nranks > 1 || (h->comm && getenv("PSBA_SPARSE_PATTERN_FORCE"))) {
      const size_t nBlk = (size_t)nCams * (nCams + 1) / 2;
      if (h->comm) {
        pat.resize(nBlk);
        TRY(sparse_pattern(nCams, n2Dprojs, iidx, jidx, ptr.data(), pat.data()));
        unsigned char *dev = nullptr;
        PSBA_HIP(h, hipMalloc((void **)&dev, nBlk));
        hipError_t e1 = hipMemcpyAsync(dev, pat.data(), nBlk, hipMemcpyHostToDevice, h->stream);
        ncclResult_t e2 = e1 == hipSuccess ? ncclAllReduce(dev, dev, nBlk, ncclUint8, ncclMax, h->comm, h->stream) : ncclSuccess;
        if (e1 == hipSuccess && e2 == ncclSuccess) e1 = hipMemcpyAsync(pat.data(), dev, nBlk, hipMemcpyDeviceToHost, h->stream);
        if (e1 == hipSuccess && e2 == ncclSuccess) e1 = hipStreamSynchronize(h->stream);
        (void)hipFree(dev);
        if (e2 != ncclSuccess) return fail(h, PSBA_E_RCCL, "all-reduce of the block pattern: %s", ncclGetErrorString(e2));
        PSBA_HIP(h, e1);
      } else {
        if (h->bs_pattern.size() != nBlk)
          return fail(h, PSBA_E_STATE, "PSBA_SOLVER_PCG with a rank layout and no communicator: psba_set_sparse_pattern "
                                       "(the union of psba_sparse_pattern over the ranks) before psba_upload_problem");
        pat = h->bs_pattern;
      }
    }
    TRY(build_owner_plan(nCams, n2Dprojs, iidx, jidx, ptr.data(), op, pat.empty() ? nullptr : pat.data()));
    h->nGroups = 0;
    h->packedN = 36 * (size_t)nCams * (nCams + 1) / 2;
    TRY(dev_alloc(h, &h->own_prod, op.prod.size()));
    TRY(dev_alloc(h, &h->own_waves, op.waves.size()));
    TRY(dev_alloc(h, &h->own_units, op.units.size()));
    PSBA_HIP(h, hipMemcpy(h->own_prod, op.prod.data(), sizeof(int2) * op.prod.size(), hipMemcpyHostToDevice));
    PSBA_HIP(h, hipMemcpy(h->own_waves, op.waves.data(), sizeof(OwnerWave) * op.waves.size(), hipMemcpyHostToDevice));
    PSBA_HIP(h, hipMemcpy(h->own_units, op.units.data(), sizeof(OwnerUnit) * op.units.size(), hipMemcpyHostToDevice));
    h->own_nwaves = (int)op.waves.size();
    h->own_products = op.products;
    h->bs_nblk = (long long)op.blocks.size();
    TRY(dev_alloc(h, &h->bs_val, (size_t)36 * op.blocks.size() + (size_t)d.nA));
    h->bs_ea = h->bs_val + (size_t)36 * op.blocks.size();
    TRY(dev_alloc(h, &h->bs_jk, op.blocks.size()));
    TRY(dev_alloc(h, &h->bs_diag, op.diag_slot.size()));
    PSBA_HIP(h, hipMemcpy(h->bs_jk, op.blocks.data(), sizeof(int2) * op.blocks.size(), hipMemcpyHostToDevice));
    PSBA_HIP(h, hipMemcpy(h->bs_diag, op.diag_slot.data(), sizeof(int) * op.diag_slot.size(), hipMemcpyHostToDevice));
    {  // the symmetric pattern by block row (kernels_pcg.hip, k_pcg_spmv)
      std::vector<int> rp((size_t)nCams + 1, 0);
      for (const int2 &b : op.blocks) {
        rp[(size_t)b.x + 1]++;
        if (b.x != b.y) rp[(size_t)b.y + 1]++;
      }
      for (int j = 0; j < nCams; j++) rp[(size_t)j + 1] += rp[(size_t)j];
      std::vector<int2> ent((size_t)rp[(size_t)nCams]);
      std::vector<int> at(rp.begin(), rp.end() - 1);
      for (size_t sl = 0; sl < op.blocks.size(); sl++) {
        const int2 b = op.blocks[sl];
        if (b.x == b.y) {
          ent[(size_t)at[(size_t)b.x]++] = make_int2((int)sl, b.x | (2 << 28));
        } else {
          ent[(size_t)at[(size_t)b.x]++] = make_int2((int)sl, b.y);
          ent[(size_t)at[(size_t)b.y]++] = make_int2((int)sl, b.x | (1 << 28));
        }
      }
      TRY(dev_alloc(h, &h->bs_rowptr, rp.size()));
      TRY(dev_alloc(h, &h->bs_rowent, ent.size()));
      PSBA_HIP(h, hipMemcpy(h->bs_rowptr, rp.data(), sizeof(int) * rp.size(), hipMemcpyHostToDevice));
      PSBA_HIP(h, hipMemcpy(h->bs_rowent, ent.data(), sizeof(int2) * ent.size(), hipMemcpyHostToDevice));
    }
    TRY(dev_alloc(h, &h->pcg_vec, (size_t)5 * d.nA));  // r, z, p, q, and r's second buffer
    TRY(dev_alloc(h, &h->pcg_minv, (size_t)36 * d.nC));
    TRY(dev_alloc(h, &h->pcg_scal, (size_t)(16 + 4 * 64)));  // scalars + the partial sums of p.Sp (kernels_pcg.hip)
    if (!h->pcg_host && hipHostMalloc((void **)&h->pcg_host, sizeof(double) * 16) != hipSuccess)
      return fail(h, PSBA_E_NOMEM, "no pinned memory for the PCG scalars");
    if (getenv("PSBA_SCHUR_PLAN_INFO"))
      fprintf(stderr, "[psba] block-sparse S: %lld of %lld blocks of the lower block triangle (%.1f %%), %lld products\n",
              h->bs_nblk, (long long)nCams * (nCams + 1) / 2, 100.0 * (double)h->bs_nblk / ((double)nCams * (nCams + 1) / 2),
              op.products);
  }
#ifdef PSBA_BUILD_EXPERIMENTS
  if (!sparse) {
    RingPlanHost rp;
    TRY(build_ring_plan(nCams, n3Dpts, n2Dprojs, iidx, jidx, ptr.data(), rp));
    if (rp.nWg > 0) {
      h->packedN = 36 * (size_t)nCams * (nCams + 1) / 2;
      h->nGroups = 0;
      h->ring_nWg = rp.nWg;
      h->ring_nS = rp.nS;
      h->ring_products = rp.products;
      h->ring_slots = rp.slots;
      h->ring_loaded_recs = (size_t)rp.loaded_recs;
      std::vector<int> canon;
      for (int j = 0; j < nCams; j++)
        for (int k = 0; k <= j; k++) canon.push_back((j << 16) | k);
      auto up = [&](auto **dst, const auto &v) -> int {
        TRY(dev_alloc(h, dst, v.size() ? v.size() : 1));
        if (v.size()) PSBA_HIP(h, hipMemcpy(*dst, v.data(), sizeof(v[0]) * v.size(), hipMemcpyHostToDevice));
        return PSBA_OK;
      };
      TRY(up(&h->ring_wg, rp.wgs));
      TRY(up(&h->ring_steps, rp.steps));
      TRY(up(&h->ring_entries, rp.entries));
      TRY(up(&h->ring_ops, rp.ops));
      TRY(up(&h->ring_jobs, rp.jobs));
      TRY(up(&h->ring_bl0, rp.blk_lane0));
      TRY(up(&h->ring_canon, canon));
      TRY(dev_alloc(h, &h->ring_slab, (size_t)rp.nS * h->packedN));
      TRY(dev_alloc(h, &h->ring_pvi, (size_t)9 * n3Dpts + 2));
      h->packed_doubles = h->packedN;
      TRY(dev_alloc(h, &h->redp, h->packed_doubles));
      if (getenv("PSBA_SCHUR_PLAN_INFO")) {
        long long steps_max = 0;
        for (const auto &w : rp.wgs) steps_max = w.nsteps > steps_max ? w.nsteps : steps_max;
        fprintf(stderr, "[psba] K2 ring route: %d block ranges x %d stretches, %lld products in %lld lane-steps (fill %.3f), "
                        "%lld record loads (%.2f per observation), %zu jobs, steps <= %lld, lists %.1f MB, copies %.1f MB\n",
                rp.nR, rp.nS, rp.products, rp.slots, rp.slots ? (double)rp.products / (double)rp.slots : 1.0,
                rp.loaded_recs, (double)rp.loaded_recs / n2Dprojs, rp.jobs.size(), steps_max,
                1e-6 * (4.0 * rp.entries.size() + 4.0 * rp.ops.size() + 16.0 * rp.jobs.size() + 16.0 * rp.steps.size()),
                8e-6 * (double)rp.nS * (double)h->packedN);
      }
    }
  }
#endif
  if (cnp != 6) {
    h->nGroups = 0;  // the free-intrinsics route needs no schedule (global atomics straight into S)
  } else if (!sparse && !h->ring_nWg) {
    SchurPlanHost plan;
    TRY(build_schur_plan(h, nCams, n3Dpts, n2Dprojs, iidx, jidx, ptr.data(), plan));
    h->schur_runs = h->nGroups > 0 && plan.runs;
    h->schur_pairs = h->nGroups > 0 && plan.pair_items > 0;
    if (h->nGroups) {
      TRY(dev_alloc(h, &h->items, plan.items.size() ? plan.items.size() : 1));
      TRY(dev_alloc(h, &h->wg, plan.wgs.size()));
      TRY(dev_alloc(h, &h->posblock, plan.posblock.size()));
      TRY(dev_alloc(h, &h->slab, plan.slab_doubles));
      PSBA_HIP(h, hipMemcpy(h->items, plan.items.data(), sizeof(unsigned long long) * plan.items.size(), hipMemcpyHostToDevice));
      PSBA_HIP(h, hipMemcpy(h->wg, plan.wgs.data(), sizeof(SchurWg) * plan.wgs.size(), hipMemcpyHostToDevice));
      PSBA_HIP(h, hipMemcpy(h->posblock, plan.posblock.data(), sizeof(int) * plan.posblock.size(), hipMemcpyHostToDevice));
      for (int j = 0, b = 0; j < 6; j++)
        for (int k = 0; k <= j; k++, b++) {
          h->h_diagpos[b] = j < nCams ? plan.blockpos[(size_t)b] : 0;
          int g = -1;
          if (j < nCams)
            for (g = 0; b >= h->gblk0[g + 1];) g++;
          h->h_diaggrp[b] = g;
        }
      {
        std::vector<ReduceGroup> tab;
        long long pos0 = 0;
        for (int g = 0; g < h->nGroups; g++) {
          tab.insert(tab.end(), (size_t)h->gnblk[g] / 16, ReduceGroup{h->gnwg[g], h->gnblk[g], pos0, h->gslab[g], 0});
          pos0 += h->gnblk[g];
        }
        TRY(dev_alloc(h, &h->gtab, tab.size()));
        PSBA_HIP(h, hipMemcpy(h->gtab, tab.data(), sizeof(ReduceGroup) * tab.size(), hipMemcpyHostToDevice));
      }
      h->packed_doubles = h->packedN;  // 36 doubles per block of the lower block triangle, canonical order
      TRY(dev_alloc(h, &h->redp, h->packed_doubles));
      TRY(dev_alloc(h, &h->diag0, (size_t)21 * 36));
      PSBA_HIP(h, hipMemsetAsync(h->diag0, 0, sizeof(double) * 21 * 36, h->stream));
      if (getenv("PSBA_SCHUR_PLAN_INFO"))
        fprintf(stderr, "[psba] K2 plan%s: %d groups, %d workgroups, %lld products in %zu item slots (fill %.3f), slabs %.1f MB\n",
                plan.runs ? " (runs layout)" : "", h->nGroups, h->nWg, plan.real_items, plan.items.size(),
                plan.items.size() ? (double)plan.real_items / (double)plan.items.size() : 1.0,
                8e-6 * (double)plan.slab_doubles);
    } else {
      // many cameras: the owner route (one thread per block segment, sums in registers)
      OwnerPlanHost op;
      TRY(build_owner_plan(nCams, n2Dprojs, iidx, jidx, ptr.data(), op));
      TRY(dev_alloc(h, &h->own_prod, op.prod.size()));
      TRY(dev_alloc(h, &h->own_waves, op.waves.size()));
      TRY(dev_alloc(h, &h->own_units, op.units.size()));
      PSBA_HIP(h, hipMemcpy(h->own_prod, op.prod.data(), sizeof(int2) * op.prod.size(), hipMemcpyHostToDevice));
      PSBA_HIP(h, hipMemcpy(h->own_waves, op.waves.data(), sizeof(OwnerWave) * op.waves.size(), hipMemcpyHostToDevice));
      PSBA_HIP(h, hipMemcpy(h->own_units, op.units.data(), sizeof(OwnerUnit) * op.units.size(), hipMemcpyHostToDevice));
      h->own_nwaves = (int)op.waves.size();
      h->own_products = op.products;
      if (getenv("PSBA_SCHUR_PLAN_INFO"))
        // no silent cliff: say which route K2 takes (psba_schur_path returns the same)
        fprintf(stderr, "[psba] K2 owner route (%d cameras): %lld products in %zu ELL slots (fill %.3f), %d waves\n",
                nCams, op.products, op.prod.size(), (double)op.products / (double)op.prod.size(), h->own_nwaves);
    }
  }
  auto H2D = [&](void *dst, const void *src, size_t bytes) {
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream);
  };
  PSBA_HIP(h, H2D(h->camconst, cc.data(), sizeof(double) * cc.size()));
  std::vector<double> cam11;  // free intrinsics: the camera block is (K | local rotation | translation)
  if (cnp != 6) {
    cam11.resize((size_t)d.nA);
    for (int j = 0; j < nCams; j++) {
      for (int k = 0; k < 5; k++) cam11[(size_t)cnp * j + k] = Kparas[5 * j + k];
      for (int k = 0; k < 6; k++) cam11[(size_t)cnp * j + 5 + k] = camsEx[6 * j + k];
    }
    camsEx = cam11.data();
  }
  PSBA_HIP(h, H2D(h->cams[0], camsEx, sizeof(double) * d.nA));
  PSBA_HIP(h, H2D(h->pts[0], pts3D, sizeof(double) * d.nB));
  PSBA_HIP(h, H2D(h->params0, camsEx, sizeof(double) * d.nA));
  PSBA_HIP(h, H2D(h->params0 + d.nA, pts3D, sizeof(double) * d.nB));
  PSBA_HIP(h, H2D(h->impts, impts, sizeof(double) * 2 * (size_t)d.nO));
  PSBA_HIP(h, H2D(h->iidx, iidx, sizeof(int) * (size_t)d.nO));
  PSBA_HIP(h, H2D(h->jidx, jidx, sizeof(int) * (size_t)d.nO));
  PSBA_HIP(h, H2D(h->ptr, ptr.data(), sizeof(int) * ptr.size()));
  PSBA_HIP(h, H2D(h->tile_pt, tile_pt.data(), sizeof(int) * tile_pt.size()));
  std::vector<int4> tile_desc;
  for (int t = 0; t < d.nTilesAll; t++)
    if (ptr[tile_pt[t + 1]] - ptr[tile_pt[t]] <= TILE_OBS)
      tile_desc.push_back(make_int4(tile_pt[t], tile_pt[t + 1], ptr[tile_pt[t]], ptr[tile_pt[t + 1]]));
  if (tile_desc.empty()) tile_desc.push_back(make_int4(0, 0, 0, 0));  // (only long points: an empty tile keeps the grids non-empty)
  PSBA_HIP(h, hipMemcpy(h->tile_desc, tile_desc.data(), sizeof(int4) * tile_desc.size(), hipMemcpyHostToDevice));
  PSBA_HIP(h, hipMemsetAsync(h->dp, 0, sizeof(double) * d.nT, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));  // host vectors go out of scope
  // ... and the blocking copies above went through the null stream, which this handle's non-blocking stream does
  // not wait for: everything on the device is done before the first kernel can be queued
  PSBA_HIP(h, hipDeviceSynchronize());
  h->uploaded = true;
  return PSBA_OK;
}

int psba_set_params(psba_handle h, const double *camsEx, const double *pts3D) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  PSBA_HIP(h, hipMemcpyAsync(h->cams[h->cur], camsEx, sizeof(double) * h->d.nA,
                             hipMemcpyHostToDevice, h->stream));
  PSBA_HIP(h, hipMemcpyAsync(h->pts[h->cur], pts3D, sizeof(double) * h->d.nB,
                             hipMemcpyHostToDevice, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  h->linearized = h->assembled = h->solved = h->backsubbed = false;
  h->ahead = h->lin_is_ahead = h->backsub_pending = h->publish_deferred = h->publish_in_k1 = false;
  return PSBA_OK;
}

int psba_reset_params(psba_handle h) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  PSBA_HIP(h, hipMemcpyAsync(h->cams[h->cur], h->params0, sizeof(double) * h->d.nA, hipMemcpyDeviceToDevice,
                             h->stream));
  PSBA_HIP(h, hipMemcpyAsync(h->pts[h->cur], h->params0 + h->d.nA, sizeof(double) * h->d.nB,
                             hipMemcpyDeviceToDevice, h->stream));
  h->linearized = h->assembled = h->solved = h->backsubbed = false;
  h->ahead = h->lin_is_ahead = h->backsub_pending = h->publish_deferred = h->publish_in_k1 = false;
  return PSBA_OK;
}

int psba_get_params(psba_handle h, int which, double *camsEx, double *pts3D) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  const int set = which == PSBA_PARAMS_NEW ? 1 - h->cur : h->cur;
  if (camsEx)
    PSBA_HIP(h, hipMemcpyAsync(camsEx, h->cams[set], sizeof(double) * h->d.nA,
                               hipMemcpyDeviceToHost, h->stream));
  if (pts3D)
    PSBA_HIP(h, hipMemcpyAsync(pts3D, h->pts[set], sizeof(double) * h->d.nB,
                               hipMemcpyDeviceToHost, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  return PSBA_OK;
}

// ---- fused verbs ------------------------------------------------------------------------

static int fetch_scalars(psba_ctx *h) {
  PSBA_HIP(h, hipMemcpyAsync(h->h_scal, h->scal, sizeof(double) * NSCAL, hipMemcpyDeviceToHost,
                             h->stream));  // scalars and the status stamps (tail of the block)
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  return PSBA_OK;
}

static int enqueue_residual(psba_ctx *h, int which) {
  TRY(launch_residual(h, which, nullptr));
  if (h->comm)
    RCCL(h, ncclAllReduce(h->scal + SC_COST, h->scal + SC_COST, 1, ncclDouble, ncclSum, h->comm,
                          h->stream));
  return PSBA_OK;
}

int psba_residual(psba_handle h, int which, double *cost) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  TRY(enqueue_residual(h, which));
  TRY(fetch_scalars(h));
  if (cost) *cost = h->h_scal[SC_COST];
  return PSBA_OK;
}

int psba_linearize(psba_handle h, double coeff, double coeff_g) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  // nothing to do when psba_linearize_ahead already produced exactly this linearization
  if (!(h->lin_is_ahead && coeff == h->coeff && coeff_g == h->coeff_g)) {
    h->coeff = coeff;
    h->coeff_g = coeff_g;
    TRY(launch_linearize(h, false));
  }
  h->lin_is_ahead = false;
  h->ahead = false;
  h->linearized = true;
  h->assembled = h->solved = h->backsubbed = false;
  return PSBA_OK;
}

static int enqueue_max_diag(psba_ctx *h) {
  if (h->comm) {
    // diag(U) is a sum over all points: reduce the per-rank U first (scratch copy in dp)
    double *tmp = h->dp;  // dp is free between linearize and the first solve
    PSBA_HIP(h, hipMemcpyAsync(tmp, h->U, sizeof(double) * 36 * h->d.nC, hipMemcpyDeviceToDevice,
                               h->stream));
    RCCL(h, ncclAllReduce(tmp, tmp, (size_t)36 * h->d.nC, ncclDouble, ncclSum, h->comm, h->stream));
    double *keep = h->U;
    h->U = tmp;
    int rc = launch_max_diag(h);
    h->U = keep;
    TRY(rc);
    RCCL(h, ncclAllReduce(h->scal + SC_MAXDIAG, h->scal + SC_MAXDIAG, 1, ncclDouble, ncclMax,
                          h->comm, h->stream));
  } else {
    TRY(launch_max_diag(h));
  }
  return PSBA_OK;
}

int psba_begin(psba_handle h, double coeff, double coeff_g, double *cost, double *max_diag) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  TRY(enqueue_residual(h, PSBA_PARAMS_CUR));
  h->coeff = coeff;
  h->coeff_g = coeff_g;
  TRY(launch_linearize(h, false));
  h->linearized = true;
  h->assembled = h->solved = h->backsubbed = false;
  h->ahead = false;
  TRY(enqueue_max_diag(h));
  TRY(fetch_scalars(h));
  if (cost) *cost = h->h_scal[SC_COST];
  if (max_diag) *max_diag = h->h_scal[SC_MAXDIAG];
  h->lin_is_ahead = true;  // the next psba_linearize with these coefficients has nothing left to do
  return PSBA_OK;
}

int psba_max_diag(psba_handle h, double *out) {
  CHECK_H(h);
  NEED(h, h->linearized, "psba_linearize first");
  TRY(enqueue_max_diag(h));
  TRY(fetch_scalars(h));
  if (out) *out = h->h_scal[SC_MAXDIAG];
  return PSBA_OK;
}

int psba_schur_assemble(psba_handle h, double mu) {
  CHECK_H(h);
  NEED(h, h->linearized, "psba_linearize first");
  h->mu = mu;
  TRY(launch_schur(h, mu, false));
  h->assembled = true;
  h->solved = h->backsubbed = false;
  return PSBA_OK;
}

// the per-try collective: [tril(S) || e_a] in slab order when the LDS schedule produced it
// (about half the bytes of the padded square), then scattered into the padded buffer; the whole
// padded square for the global-atomic fallback kernel
static int allreduce_schur(psba_ctx *h) {
  if (h->solver == PSBA_SOLVER_PCG) {  // the blocks that exist and e_a behind them: one collective
    if (h->comm)
      RCCL(h, ncclAllReduce(h->bs_val, h->bs_val, (size_t)36 * h->bs_nblk + h->d.nA, ncclDouble, ncclSum, h->comm, h->stream));
    return PSBA_OK;
  }
  if (h->packed_pending) {
    if (h->comm)
      RCCL(h, ncclAllReduce(h->redp, h->redp, h->packed_doubles, ncclDouble, ncclSum, h->comm, h->stream));
    TRY(launch_schur_expand(h));
    h->packed_pending = false;
  } else if (h->comm) {
    const size_t n = (size_t)(h->n32 + 1) * h->n32;  // S rows, padding rows and the ea row
    RCCL(h, ncclAllReduce(h->red, h->red, n, ncclDouble, ncclSum, h->comm, h->stream));
  }
  return PSBA_OK;
}

int psba_schur_reduce(psba_handle h) {
  CHECK_H(h);
  NEED(h, h->assembled, "psba_schur_assemble first");
  if (!h->comm) return PSBA_OK;
  ProfScope ps(h, PSBA_K_ALLREDUCE);
  TRY(allreduce_schur(h));
  return PSBA_OK;
}

int psba_schur_solve(psba_handle h) {
  CHECK_H(h);
  NEED(h, h->assembled, "psba_schur_assemble first");
  if (h->packed_pending) TRY(allreduce_schur(h));  // psba_schur_reduce was skipped
  if (h->solver == PSBA_SOLVER_PCG) {
    TRY(launch_pcg_solve(h));
    h->assembled = false;
    h->solved = true;
    return h->pcg_exhausted ? PSBA_PCG_MAXIT : PSBA_OK;
  }
  TRY(launch_chol_solve(h));
  if (h->chol_tim) {
    long long t[32];
    PSBA_HIP(h, hipMemcpyAsync(t, h->chol_tim, sizeof t, hipMemcpyDeviceToHost, h->stream));
    PSBA_HIP(h, hipStreamSynchronize(h->stream));
    fprintf(stderr, "chol stamps (cycles since the first):");
    for (int k = 1; k < 32; k++) fprintf(stderr, " %lld", t[k] ? t[k] - t[0] : 0);
    fprintf(stderr, "\n");
  }
  h->assembled = false;  // S is overwritten by its factor
  h->solved = true;
  return PSBA_OK;
}

// the try's scalar block to the host behind everything queued on s, and the event the host waits for
static int publish_scalars(psba_ctx *h, hipStream_t s) {
  if (h->h_scal_dev && !getenv("PSBA_SCAL_MEMCPY"))
    TRY(launch_publish_scal(h, s));
  else
    PSBA_HIP(h, hipMemcpyAsync(h->h_scal, h->scal, sizeof(double) * NSCAL, hipMemcpyDeviceToHost, s));
  PSBA_HIP(h, hipEventRecord(h->scal_event, s));
  return PSBA_OK;
}

int psba_backsub_async(psba_handle h, double mu) {
  CHECK_H(h);
  NEED(h, h->solved, "psba_schur_solve first");
  TRY(launch_backsub(h, mu, false));
  hipStream_t s = h->stream;
  if (h->comm) {
    // four sums (as partial sets) + two status flags in one collective -- on the side stream, so
    // that the linearization the loop queues next (psba_linearize_ahead) runs beside the collective
    // and the copy instead of behind them (a small all-reduce is ~20-30 us of latency, K1 33 us)
    // (one rank: the collective is a 5 us copy and the two event hops cost more than they hide --
    // 0.221 against 0.213 ms per LM iteration; PSBA_COMM_SIDE_STREAM=1 forces the side stream there,
    // which is how the tests cover it; N > 1 has not been measured
    // and without a communicator a side stream for the scalar copy alone cost 18 us per iteration
    // (0.203 against 0.185 ms: the second stream slows the graph launches of the main one), so the
    // side stream is opt-in until an N > 1 run says otherwise)
    if (h->stream2 && getenv("PSBA_COMM_SIDE_STREAM")) {
      PSBA_HIP(h, hipEventRecord(h->k3_event, h->stream));
      PSBA_HIP(h, hipStreamWaitEvent(h->stream2, h->k3_event, 0));
      s = h->stream2;
      h->scal_side = true;
    }
    RCCL(h, ncclAllReduce(h->scal + SC_PART, h->scal + SC_PART, 4 * SC_NPART + 2, ncclDouble, ncclSum, h->comm, s));
  }
  h->publish_in_k1 = false;
  // single rank: the scalars are not sent yet -- if psba_linearize_ahead comes next, its kernel carries
  // them (no kernel of their own between K3 and the linearization); psba_backsub_wait sends them
  // itself otherwise
  h->publish_deferred = !h->comm && h->h_scal_dev && !getenv("PSBA_SCAL_MEMCPY") && !getenv("PSBA_SCAL_KERNEL");
  if (!h->publish_deferred) TRY(publish_scalars(h, s));
  h->solved = false;  // the try's accumulators are consumed; a new try starts at psba_schur_assemble
  h->backsub_pending = true;
  h->ahead = false;
  return PSBA_OK;
}

int psba_linearize_ahead(psba_handle h) {
  CHECK_H_NOJOIN(h);  // K1 reads and writes nothing the scalar collective touches
  NEED(h, h->backsub_pending || h->backsubbed, "psba_backsub_async / psba_backsub first");
  const bool carry = h->backsub_pending && h->publish_deferred;
  if (carry) h->pub_seq += 1.0;
  TRY(launch_linearize(h, false, true, carry));
  if (carry) {
    h->publish_deferred = false;
    h->publish_in_k1 = true;
  }
  h->ahead = true;
  return PSBA_OK;
}

int psba_backsub_wait(psba_handle h, psba_try_scalars *out) {
  CHECK_H_NOJOIN(h);
  NEED(h, h->backsub_pending, "psba_backsub_async first");
  if (h->publish_deferred) {  // no linearization was queued ahead: send the scalars now
    TRY(publish_scalars(h, h->stream));
    h->publish_deferred = false;
  }
  if (h->publish_in_k1) {
    // workgroup 0 of the linearization queued ahead writes the block and then the stamp: poll the
    // stamp in pinned memory (a try is ~180 us of GPU work); should it not arrive, the stream's end
    // tells why
    volatile double *stamp = h->h_scal + NSCAL;
    long long spin = 0;
    while (*stamp != h->pub_seq && ++spin < 200000000LL) {
      if ((spin & 0xfffff) == 0 && hipStreamQuery(h->stream) != hipErrorNotReady) break;
    }
    if (*stamp != h->pub_seq) {
      PSBA_HIP(h, hipStreamSynchronize(h->stream));
      if (*stamp != h->pub_seq) return fail(h, PSBA_E_HIP, "the try's scalars did not reach the host");
    }
    __sync_synchronize();
    h->publish_in_k1 = false;
  } else {
    // poll the event for a while before handing the thread to the runtime's wait (0.9 us per LM
    // iteration on the venice-shaped problem)
    for (int spin = 0; spin < 20000 && hipEventQuery(h->scal_event) == hipErrorNotReady; spin++) {
    }
    PSBA_HIP(h, hipEventSynchronize(h->scal_event));
  }
  h->scal_side = false;  // the host has seen the side stream's work complete
  h->backsub_pending = false;
  // the four sums arrive as SC_NPART partial sets: add them up in a fixed order
  for (int q = 0; q < 4; q++) {
    double v = 0.0;
    for (int s = 0; s < SC_NPART; s++) v += h->h_scal[SC_PART + 4 * s + q];
    h->h_scal[SC_DP_L2 + q] = v;
  }
  // K3 publishes the status as flags (summed over ranks): > 0 <=> flagged on some rank
  const int st0 = h->h_scal[SC_STATUS_V] > 0.0 ? h->try_id : 0;
  const int st1 = h->h_scal[SC_STATUS_SPD] > 0.0 ? h->try_id : 0;
  h->backsubbed = true;
  if (out) {
    out->status = (st1 == h->try_id ? PSBA_NOT_SPD : 0) | (st0 == h->try_id ? PSBA_SINGULAR_V : 0);
    out->dp_l2 = h->h_scal[SC_DP_L2];
    out->gain_den = h->h_scal[SC_GAIN_DEN];
    out->new_cost = h->h_scal[SC_NEW_COST];
    out->newp_l2 = h->h_scal[SC_NEWP_L2];
  }
  return PSBA_OK;
}

int psba_backsub(psba_handle h, double mu, psba_try_scalars *out) {
  TRY(psba_backsub_async(h, mu));
  return psba_backsub_wait(h, out);
}

int psba_accept(psba_handle h) {
  CHECK_H(h);
  NEED(h, h->backsubbed, "psba_backsub first");
  h->cur = 1 - h->cur;
  h->linearized = h->assembled = h->solved = h->backsubbed = false;
  if (h->ahead) {  // the linearization at the (now current) proposed parameters exists already
    std::swap(h->W, h->W_alt);
    std::swap(h->PV, h->PV_alt);
    std::swap(h->U, h->U_alt);
    std::swap(h->ga, h->ga_alt);
  }
  // only a linearization that was computed ahead for THIS proposal spares the next psba_linearize
  // (an earlier accept's flag must not survive: psba_set_step -> psba_accept -> psba_linearize)
  h->lin_is_ahead = h->ahead;
  h->ahead = false;
  return PSBA_OK;
}

// ---- operators of the trust-region caller ---------------------------------

static int d2h(psba_ctx *h, void *dst, const void *src, size_t bytes);

// ---- the sharded dense factorization, piece by piece (kernels_chol_graph.hip: chol_dist_*) ----
int psba_chol_dist_shape(psba_handle h, int *n32, int *NB, int *sharded) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  int nb = 0, bl = 0;
  TRY(chol_dist_shape(h, &nb, &bl));
  if (n32) *n32 = h->n32;
  if (NB) *NB = nb;
  if (sharded) *sharded = bl;
  return PSBA_OK;
}
int psba_chol_dist_exchange_plan(int n32, int NB, int nranks, int JE, long long *out4, int cap) {
  if (!out4 || cap <= 0) return PSBA_E_INVALID;
  return psba::chol_dist_exchange_plan(n32, NB, nranks, JE, reinterpret_cast<long long(*)[4]>(out4), cap);
}
int psba_chol_dist_begin(psba_handle h) {
  CHECK_H(h);
  NEED(h, h->assembled, "psba_schur_assemble (and the reduction over ranks) first");
  if (h->packed_pending) {
    TRY(launch_schur_expand(h));
    h->packed_pending = false;
  }
  return chol_dist_begin(h);
}
int psba_chol_dist_superpanel(psba_handle h, int J) {
  CHECK_H(h);
  NEED(h, h->assembled, "psba_schur_assemble first");
  if (J < 0 || J >= h->n32) return fail(h, PSBA_E_INVALID, "super-panel column %d out of range", J);
  return chol_dist_superpanel(h, J);
}
int psba_chol_dist_block(psba_handle h, int B, int set, double *buf, long long *n_doubles) {
  CHECK_H(h);
  NEED(h, h->assembled, "psba_schur_assemble first");
  if (B < 0 || 64 * B >= h->n32) return fail(h, PSBA_E_INVALID, "column block %d out of range", B);
  const int ncols = h->n32 - 64 * B < 64 ? h->n32 - 64 * B : 64;
  const long long n = (long long)(h->n32 + 1 - 64 * B) * ncols;
  if (n_doubles) *n_doubles = n;
  if (!buf) return PSBA_OK;
  if (!h->dist_buf) TRY(dev_alloc(h, &h->dist_buf, (size_t)(h->n32 + 1) * 64 * 8));
  if (set) {
    PSBA_HIP(h, hipMemcpyAsync(h->dist_buf, buf, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    return chol_dist_block(h, B, h->dist_buf, 1);
  }
  TRY(chol_dist_block(h, B, h->dist_buf, 0));
  return d2h(h, buf, h->dist_buf, sizeof(double) * (size_t)n);
}
int psba_chol_dist_finish(psba_handle h) {
  CHECK_H(h);
  NEED(h, h->assembled, "psba_schur_assemble first");
  TRY(chol_dist_finish(h));
  h->assembled = false;
  h->solved = true;
  return PSBA_OK;
}

static int ensure_trv(psba_ctx *h) {
  for (int k = 0; k < 2; k++)
    if (!h->trv[k]) TRY(dev_alloc(h, &h->trv[k], (size_t)h->d.nT));
  return PSBA_OK;
}

int psba_jmul_dots(psba_handle h, const double *x1, const double *x2, double dots[3]) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  if (!x1 || !dots) return fail(h, PSBA_E_INVALID, "psba_jmul_dots: null pointer");
  TRY(ensure_trv(h));
  const size_t bytes = sizeof(double) * (size_t)h->d.nT;
  PSBA_HIP(h, hipMemcpyAsync(h->trv[0], x1, bytes, hipMemcpyHostToDevice, h->stream));
  if (x2 && x2 != x1) PSBA_HIP(h, hipMemcpyAsync(h->trv[1], x2, bytes, hipMemcpyHostToDevice, h->stream));
  TRY(launch_jmul(h, h->trv[0], (x2 && x2 != x1) ? h->trv[1] : h->trv[0], nullptr, h->scal + SC_TR_DOTS));
  // the dot products run over observations: with points sharded over ranks, the sum of the ranks'
  // (a rank layout without a communicator returns this rank's part)
  if (h->comm)
    RCCL(h, ncclAllReduce(h->scal + SC_TR_DOTS, h->scal + SC_TR_DOTS, 3, ncclDouble, ncclSum, h->comm, h->stream));
  TRY(fetch_scalars(h));
  for (int k = 0; k < 3; k++) dots[k] = h->h_scal[SC_TR_DOTS + k];
  return PSBA_OK;
}

int psba_set_solver(psba_handle h, int solver, double tol, int max_iter) {
  CHECK_H(h);
  if (solver != PSBA_SOLVER_DENSE && solver != PSBA_SOLVER_PCG) return fail(h, PSBA_E_INVALID, "unknown solver %d", solver);
  NEED(h, !h->uploaded || solver == h->solver, "psba_set_solver before psba_upload_problem (the buffers depend on it)");
  h->solver = solver;
  if (tol > 0) h->pcg_tol = tol;
  if (max_iter > 0) h->pcg_maxit = max_iter;
  return PSBA_OK;
}

int psba_pcg_info(psba_handle h, int *iters, double *relres, long long *blocks, long long *dense_blocks) {
  CHECK_H(h);
  NEED(h, h->uploaded && h->solver == PSBA_SOLVER_PCG, "PSBA_SOLVER_PCG and an uploaded problem");
  if (iters) *iters = h->pcg_iters;
  if (relres) *relres = h->pcg_relres;
  if (blocks) *blocks = h->bs_nblk;
  if (dense_blocks) *dense_blocks = (long long)h->d.nC * (h->d.nC + 1) / 2;
  return PSBA_OK;
}

// the block-sparse S as assembled (after psba_schur_assemble / psba_schur_reduce): jk[2 blocks], val[36 blocks], ea[nA]
int psba_get_sparse_S(psba_handle h, int *jk, double *val, double *ea) {
  CHECK_H(h);
  NEED(h, h->assembled && h->solver == PSBA_SOLVER_PCG, "psba_schur_assemble with PSBA_SOLVER_PCG first");
  if (jk) TRY(d2h(h, jk, h->bs_jk, sizeof(int2) * (size_t)h->bs_nblk));
  if (val) TRY(d2h(h, val, h->bs_val, sizeof(double) * 36 * (size_t)h->bs_nblk));
  if (ea) TRY(d2h(h, ea, h->bs_ea, sizeof(double) * (size_t)h->d.nA));
  return PSBA_OK;
}

// the counterpart for a host that sums the ranks' blocks itself (between psba_schur_assemble and psba_schur_solve)
int psba_set_sparse_S(psba_handle h, const double *val, const double *ea) {
  CHECK_H(h);
  NEED(h, h->assembled && h->solver == PSBA_SOLVER_PCG, "psba_schur_assemble with PSBA_SOLVER_PCG first");
  if (val) PSBA_HIP(h, hipMemcpyAsync(h->bs_val, val, sizeof(double) * 36 * (size_t)h->bs_nblk, hipMemcpyHostToDevice, h->stream));
  if (ea) PSBA_HIP(h, hipMemcpyAsync(h->bs_ea, ea, sizeof(double) * (size_t)h->d.nA, hipMemcpyHostToDevice, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  h->solved = h->backsubbed = false;
  return PSBA_OK;
}

// host only: which blocks of the lower block triangle a (shard of a) problem produces
int psba_sparse_pattern(int nCams, int n3Dpts, int n2Dprojs, const int *iidx, const int *jidx, unsigned char *flags) {
  if (nCams <= 0 || n3Dpts <= 0 || n2Dprojs <= 0 || !iidx || !jidx || !flags) return PSBA_E_INVALID;
  std::vector<int> ptr((size_t)n3Dpts + 1, 0);
  for (int a = 0; a < n2Dprojs; a++) {
    if (iidx[a] < 0 || iidx[a] >= n3Dpts || jidx[a] < 0 || jidx[a] >= nCams) return PSBA_E_INVALID;
    // point-major with cameras ascending inside a point, as psba_upload_problem requires (sparse_pattern's
    // b <= a loop relies on it: unordered cameras would flag the wrong blocks silently)
    if (a && (iidx[a] < iidx[a - 1] || (iidx[a] == iidx[a - 1] && jidx[a] <= jidx[a - 1]))) return PSBA_E_INVALID;
    ptr[(size_t)iidx[a] + 1]++;
  }
  for (int i = 0; i < n3Dpts; i++) ptr[(size_t)i + 1] += ptr[(size_t)i];
  return psba::sparse_pattern(nCams, n2Dprojs, iidx, jidx, ptr.data(), flags);
}

int psba_set_sparse_pattern(psba_handle h, const unsigned char *flags, long long n) {
  CHECK_H(h);
  if (!flags || n <= 0) return fail(h, PSBA_E_INVALID, "psba_set_sparse_pattern: null pattern");
  NEED(h, !h->uploaded, "psba_set_sparse_pattern before psba_upload_problem (the block list depends on it)");
  h->bs_pattern.assign(flags, flags + n);
  return PSBA_OK;
}

int psba_allreduce_scalars(psba_handle h, double *v, int n) {
  CHECK_H(h);
  if (!v || n < 0 || n > 8) return fail(h, PSBA_E_INVALID, "psba_allreduce_scalars: at most 8 values");
  if (!h->comm || n == 0) return PSBA_OK;  // one rank (or the caller sums): nothing to add
  double *tmp = h->scal + SC_SUMS;
  PSBA_HIP(h, hipMemcpyAsync(tmp, v, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  RCCL(h, ncclAllReduce(tmp, tmp, (size_t)n, ncclDouble, ncclSum, h->comm, h->stream));
  return d2h(h, v, tmp, sizeof(double) * (size_t)n);
}

int psba_compute_Jmultiply(psba_handle h, const double *x, double *Jmul) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  if (!x) return fail(h, PSBA_E_INVALID, "psba_compute_Jmultiply: null pointer");
  TRY(ensure_trv(h));
  if (!h->jmul_out) TRY(dev_alloc(h, &h->jmul_out, (size_t)2 * h->d.nO));
  PSBA_HIP(h, hipMemcpyAsync(h->trv[0], x, sizeof(double) * (size_t)h->d.nT, hipMemcpyHostToDevice, h->stream));
  TRY(launch_jmul(h, h->trv[0], h->trv[0], h->jmul_out, h->scal + SC_TR_DOTS));
  return d2h(h, Jmul, h->jmul_out, sizeof(double) * 2 * (size_t)h->d.nO);
}

int psba_get_gradient(psba_handle h, double *g) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->linearized, "psba_linearize first");
  if (!g) return PSBA_E_INVALID;
  TRY(ensure_trv(h));
  TRY(launch_pack_g(h, h->trv[0]));
  // g_a is a sum over observations: every rank holds its points' part (g_b is rank-local by nature)
  if (h->comm) RCCL(h, ncclAllReduce(h->trv[0], h->trv[0], (size_t)h->d.nA, ncclDouble, ncclSum, h->comm, h->stream));
  return d2h(h, g, h->trv[0], sizeof(double) * (size_t)h->d.nT);
}

int psba_get_dp(psba_handle h, double *dp) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  return d2h(h, dp, h->dp, sizeof(double) * (size_t)h->d.nT);
}

int psba_set_step(psba_handle h, const double *dp) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  if (!dp) return PSBA_E_INVALID;
  PSBA_HIP(h, hipMemcpyAsync(h->dp, dp, sizeof(double) * (size_t)h->d.nT, hipMemcpyHostToDevice, h->stream));
  TRY(launch_newp(h, h->dp));
  h->ahead = false;  // a linearization computed ahead belongs to another proposal
  h->backsubbed = true;  // a proposal exists: psba_accept may take it
  return PSBA_OK;
}

int psba_cholmod_lambda(psba_handle h, int reassemble, double *lambda, double *info3) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  if (h->nranks > 1 && !h->comm)
    return fail(h, PSBA_E_INVALID, "psba_cholmod_lambda on a rank layout needs the communicator (S must be complete)");
  if (h->solver == PSBA_SOLVER_PCG) {
    // block-sparse mode: no dense S to factor -- the damping estimate is the Gershgorin shift of the stored blocks
    // (kernels_pcg.hip; no reference counterpart: the reference has no sparse mode)
    NEED(h, h->linearized, "psba_linearize first");
    if (reassemble || !h->assembled) {
      TRY(launch_schur(h, 0.0, false));
      TRY(allreduce_schur(h));  // every rank then looks at the same complete blocks
    }
    double lam = 0.0;
    TRY(launch_bsr_gershgorin(h, &lam, info3));
    if (lambda) *lambda = lam;
    h->assembled = h->solved = false;
    return PSBA_OK;
  }
  if (reassemble) {
    // S at lambda = 0 again (the failed factorization worked in place), then the modified
    // Cholesky on a copy of it (trust_region.cpp:341-363)
    NEED(h, h->linearized, "psba_linearize first");
    TRY(launch_schur(h, 0.0, false));
    // with a communicator (even of one rank) the sums sit in the packed buffer until the all-reduce
    // and k_schur_expand have run: the modified Cholesky must not factor a stale square
    if (h->packed_pending || h->comm) TRY(allreduce_schur(h));  // every rank then factors the same complete S
  }
  h->diag_done = false;  // the first diagonal block's factor in chol_L is about to be overwritten
  TRY(launch_cholmod(h, h->scal + SC_CHOLMOD));
  TRY(fetch_scalars(h));
  if (lambda) *lambda = h->h_scal[SC_CHOLMOD];
  if (info3)
    for (int k = 0; k < 3; k++) info3[k] = h->h_scal[SC_CHOLMOD + 1 + k];
  h->assembled = h->solved = false;
  return PSBA_OK;
}

// ---- sba_func.h mirror ------------------------------------------------------------------

static int ensure_dbg(psba_ctx *h) {
  const Dims &d = h->d;
  if (!h->dbg_ex) TRY(dev_alloc(h, &h->dbg_ex, (size_t)2 * d.nO));
  if (!h->dbg_JA) TRY(dev_alloc(h, &h->dbg_JA, (size_t)12 * d.nO));
  if (!h->dbg_JB) TRY(dev_alloc(h, &h->dbg_JB, (size_t)6 * d.nO));
  if (!h->dbg_Y) TRY(dev_alloc(h, &h->dbg_Y, (size_t)18 * d.nO));
  if (!h->dbg_Vinv) TRY(dev_alloc(h, &h->dbg_Vinv, (size_t)9 * d.nP));
  if (!h->dbg_eb) TRY(dev_alloc(h, &h->dbg_eb, (size_t)3 * d.nP));
  return PSBA_OK;
}

static int d2h(psba_ctx *h, void *dst, const void *src, size_t bytes) {
  if (!dst) return PSBA_OK;
  PSBA_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  return PSBA_OK;
}

static int relinearize_dump(psba_ctx *h) {
  TRY(ensure_dbg(h));
  TRY(launch_linearize(h, true));
  h->linearized = true;
  h->assembled = h->solved = h->backsubbed = false;
  return PSBA_OK;
}

static int reassemble_dump(psba_ctx *h) {
  NEED(h, h->linearized, "linearise first (compute_jacobiQT / compute_U / ...)");
  TRY(ensure_dbg(h));
  TRY(launch_schur(h, h->mu_applied ? h->mu : 0.0, true));
  h->assembled = true;
  h->solved = h->backsubbed = false;
  if (h->comm) TRY(allreduce_schur(h));
  return PSBA_OK;
}

int psba_compute_exQT(psba_handle h, int which, double *ex) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  TRY(ensure_dbg(h));
  TRY(launch_residual(h, which, h->dbg_ex));
  return d2h(h, ex, h->dbg_ex, sizeof(double) * 2 * (size_t)h->d.nO);
}

int psba_compute_jacobiQT(psba_handle h, double *jac_A, double *jac_B) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  TRY(relinearize_dump(h));
  TRY(d2h(h, jac_A, h->dbg_JA, sizeof(double) * 12 * (size_t)h->d.nO));
  return d2h(h, jac_B, h->dbg_JB, sizeof(double) * 6 * (size_t)h->d.nO);
}

int psba_compute_U(psba_handle h, double coeff, double *out) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  h->coeff = coeff;
  TRY(relinearize_dump(h));
  return d2h(h, out, h->U, sizeof(double) * 36 * (size_t)h->d.nC);
}

static int download_V(psba_ctx *h, double *out, double mu) {
  if (!out) return PSBA_OK;
  std::vector<double> pv((size_t)9 * h->d.nP);
  TRY(d2h(h, pv.data(), h->PV, sizeof(double) * pv.size()));
  for (int i = 0; i < h->d.nP; i++) {
    const double *s = &pv[(size_t)9 * i];
    double *o = out + (size_t)9 * i;
    o[0] = s[0] + mu; o[1] = s[1]; o[2] = s[2];
    o[3] = s[1]; o[4] = s[3] + mu; o[5] = s[4];
    o[6] = s[2]; o[7] = s[4]; o[8] = s[5] + mu;
  }
  return PSBA_OK;
}

int psba_compute_V(psba_handle h, double coeff, double *out) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  h->coeff = coeff;
  TRY(relinearize_dump(h));
  return download_V(h, out, 0.0);
}

int psba_maxElmOfUV(psba_handle h, double *out) { return psba_max_diag(h, out); }

int psba_update_UV(psba_handle h, double mu, double *U, double *V) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->linearized, "linearise first");
  h->mu = mu;
  h->mu_applied = true;
  if (U) {
    TRY(d2h(h, U, h->U, sizeof(double) * 36 * (size_t)h->d.nC));
    for (int t = 0; t < h->d.nA; t++) U[36 * (t / 6) + 7 * (t % 6)] += mu;
  }
  return download_V(h, V, mu);
}

int psba_restore_UVdiag(psba_handle h) {
  CHECK_H(h);
  h->mu_applied = false;
  return PSBA_OK;
}

int psba_compute_Vinv(psba_handle h, double *Vinv) {
  CHECK_H(h);
  TRY(reassemble_dump(h));
  TRY(d2h(h, Vinv, h->dbg_Vinv, sizeof(double) * 9 * (size_t)h->d.nP));
  TRY(fetch_scalars(h));
  return h->h_status[0] == h->try_id ? PSBA_SINGULAR_V : PSBA_OK;
}

int psba_compute_Wblks(psba_handle h, double coeff, double *Wblks) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  h->coeff = coeff;
  TRY(relinearize_dump(h));
  return d2h(h, Wblks, h->W, sizeof(double) * 18 * (size_t)h->d.nO);
}

int psba_compute_Yblks(psba_handle h, double *Yblks) {
  CHECK_H(h);
  TRY(reassemble_dump(h));
  return d2h(h, Yblks, h->dbg_Y, sizeof(double) * 18 * (size_t)h->d.nO);
}

int psba_compute_S(psba_handle h, double *S) {
  CHECK_H(h);
  TRY(reassemble_dump(h));
  if (!S) return PSBA_OK;
  PSBA_HIP(h, hipMemcpy2DAsync(S, sizeof(double) * h->d.nA, h->red, sizeof(double) * h->n32,
                               sizeof(double) * h->d.nA, h->d.nA, hipMemcpyDeviceToHost, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  return PSBA_OK;
}

int psba_compute_g(psba_handle h, double coeff, double *g) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->uploaded, "no problem uploaded");
  h->coeff_g = coeff;
  TRY(relinearize_dump(h));
  if (!g) return PSBA_OK;
  TRY(d2h(h, g, h->ga, sizeof(double) * (size_t)h->d.nA));
  std::vector<double> pv((size_t)9 * h->d.nP);
  TRY(d2h(h, pv.data(), h->PV, sizeof(double) * pv.size()));
  for (int i = 0; i < h->d.nP; i++)
    for (int k = 0; k < 3; k++) g[h->d.nA + 3 * (size_t)i + k] = pv[(size_t)9 * i + 6 + k];
  return PSBA_OK;
}

int psba_compute_ea(psba_handle h, double *ea) {
  CHECK_H(h);
  TRY(reassemble_dump(h));
  return d2h(h, ea, h->red + (size_t)h->n32 * h->n32, sizeof(double) * (size_t)h->d.nA);
}

int psba_SPDinv_matVec(psba_handle h, double *dpa) {
  CHECK_H(h);
  NEED(h, h->assembled, "compute_S / compute_ea first");
  TRY(launch_chol_solve(h));
  h->assembled = false;
  h->solved = true;
  TRY(fetch_scalars(h));
  TRY(d2h(h, dpa, h->dp, sizeof(double) * (size_t)h->d.nA));
  return h->h_status[1] == h->try_id ? PSBA_NOT_SPD : PSBA_OK;
}

static int backsub_dump(psba_ctx *h) {
  NEED(h, h->solved, "SPDinv_matVec first");
  TRY(ensure_dbg(h));
  TRY(launch_backsub(h, h->mu_applied ? h->mu : 0.0, true));
  h->backsubbed = true;
  return PSBA_OK;
}

int psba_compute_eb(psba_handle h, double *eb) {
  CHECK_H(h);
  TRY(backsub_dump(h));
  return d2h(h, eb, h->dbg_eb, sizeof(double) * 3 * (size_t)h->d.nP);
}

int psba_compute_dpb(psba_handle h, double *dp) {
  CHECK_H(h);
  TRY(backsub_dump(h));
  return d2h(h, dp, h->dp, sizeof(double) * (size_t)h->d.nT);
}

int psba_compute_newp(psba_handle h, double *new_p) {
  CHECK_H(h);
  NEED(h, h->backsubbed, "compute_dpb first");
  if (!new_p) return PSBA_OK;
  TRY(d2h(h, new_p, h->cams[1 - h->cur], sizeof(double) * (size_t)h->d.nA));
  return d2h(h, new_p + h->d.nA, h->pts[1 - h->cur], sizeof(double) * (size_t)h->d.nB);
}

int psba_update_p(psba_handle h, double *p) {
  CHECK_H(h);
  TRY(psba_accept(h));
  if (!p) return PSBA_OK;
  TRY(d2h(h, p, h->cams[h->cur], sizeof(double) * (size_t)h->d.nA));
  return d2h(h, p + h->d.nA, h->pts[h->cur], sizeof(double) * (size_t)h->d.nB);
}

// ---- multi-GPU --------------------------------------------------------------------------

int psba_partition_points(int n3Dpts, const int *iidx, int n2Dprojs, int nranks, int *pt_begin) {
  if (n3Dpts <= 0 || nranks <= 0 || !iidx || !pt_begin || n2Dprojs < 0) return PSBA_E_INVALID;
  // contiguous point ranges; rank r ends at the first point whose cumulative observation
  // count reaches (r+1)/nranks of the total
  std::vector<long long> cum((size_t)n3Dpts + 1, 0);
  for (int a = 0; a < n2Dprojs; a++) {
    if (iidx[a] < 0 || iidx[a] >= n3Dpts) return PSBA_E_INVALID;
    cum[(size_t)iidx[a] + 1]++;
  }
  for (int i = 0; i < n3Dpts; i++) cum[(size_t)i + 1] += cum[i];
  pt_begin[0] = 0;
  int p = 0;
  for (int r = 1; r < nranks; r++) {
    const long long target = (long long)n2Dprojs * r / nranks;
    while (p < n3Dpts && cum[p] < target) p++;
    // keep at least one point per rank when there are enough points
    int lo = pt_begin[r - 1] + 1, hi = n3Dpts - (nranks - r);
    if (n3Dpts >= nranks) {
      if (p < lo) p = lo;
      if (p > hi) p = hi;
    }
    pt_begin[r] = p;
  }
  pt_begin[nranks] = n3Dpts;
  return PSBA_OK;
}

int psba_comm_unique_id(void *id128) {
  if (!id128) return PSBA_E_INVALID;
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return PSBA_E_RCCL;
  memcpy(id128, &id, sizeof id);
  return PSBA_OK;
}

int psba_comm_init(psba_handle h, int nranks, int rank, const void *id128) {
  CHECK_H(h);
  if (nranks < 1 || rank < 0 || rank >= nranks || !id128)
    return fail(h, PSBA_E_INVALID, "psba_comm_init: bad rank %d / %d", rank, nranks);
  PSBA_HIP(h, hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  RCCL(h, ncclCommInitRank(&h->comm, nranks, id, rank));
  if (!h->stream2) {
    PSBA_HIP(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    PSBA_HIP(h, hipEventCreateWithFlags(&h->k3_event, hipEventDisableTiming));
  }
  h->nranks = nranks;
  h->rank = rank;
  return PSBA_OK;
}

int psba_set_rank_layout(psba_handle h, int nranks, int rank) {
  CHECK_H(h);
  NEED(h, h->cnp == 6 || nranks <= 1, "free intrinsics: single rank only");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, PSBA_E_INVALID, "bad rank %d / %d", rank, nranks);
  if (h->comm) return fail(h, PSBA_E_STATE, "a communicator is attached: its layout is fixed");
  if (h->solver == PSBA_SOLVER_PCG && h->uploaded && nranks != h->nranks)
    return fail(h, PSBA_E_STATE, "PSBA_SOLVER_PCG: the rank layout is part of the block list, set it before psba_upload_problem");
  h->nranks = nranks;
  h->rank = rank;
  return PSBA_OK;
}

// (with the PSBA_SCHUR_PACKED test hook the buffer is the packed [tril(S) | e_a] sums a
// communicator would all-reduce: 36 doubles per block of the lower block triangle)
static bool packed_hook(psba_ctx *h) {
  return !h->comm && h->nranks > 1 && (h->nGroups > 0 || h->ring_nWg > 0) && getenv("PSBA_SCHUR_PACKED");
}

int psba_reduce_buffer_size(psba_handle h, long long *n_doubles) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  if (n_doubles) *n_doubles = packed_hook(h) ? (long long)h->packed_doubles : (long long)(h->n32 + 1) * h->n32;
  return PSBA_OK;
}

int psba_get_reduce_buffer(psba_handle h, double *out) {
  CHECK_H(h);
  NEED(h, h->assembled, "psba_schur_assemble first");
  NEED(h, !h->comm, "the reduce-buffer verbs are for handles without a communicator");
  NEED(h, h->solver != PSBA_SOLVER_PCG, "no dense reduce buffer with PSBA_SOLVER_PCG (psba_get_sparse_S)");
  if (packed_hook(h)) return d2h(h, out, h->redp, sizeof(double) * h->packed_doubles);
  return d2h(h, out, h->red, sizeof(double) * (size_t)(h->n32 + 1) * h->n32);
}

int psba_set_reduce_buffer(psba_handle h, const double *in) {
  CHECK_H(h);
  NEED(h, h->cnp == 6, "six-parameter camera blocks only (free intrinsics: the fused verbs and psba_levmar)");
  NEED(h, h->assembled, "psba_schur_assemble first");
  NEED(h, !h->comm, "the reduce-buffer verbs are for handles without a communicator");
  if (!in) return fail(h, PSBA_E_INVALID, "null buffer");
  if (packed_hook(h)) {  // the summed packed sums: psba_schur_solve scatters them (k_schur_expand)
    PSBA_HIP(h, hipMemcpyAsync(h->redp, in, sizeof(double) * h->packed_doubles, hipMemcpyHostToDevice, h->stream));
    PSBA_HIP(h, hipStreamSynchronize(h->stream));
    h->packed_pending = true;
    h->diag_done = false;
    return PSBA_OK;
  }
  PSBA_HIP(h, hipMemcpyAsync(h->red, in, sizeof(double) * (size_t)(h->n32 + 1) * h->n32,
                             hipMemcpyHostToDevice, h->stream));
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  h->diag_done = false;  // S changed under the factor of its first diagonal block
  return PSBA_OK;
}

int psba_comm_rank(psba_handle h, int *nranks, int *rank) {
  CHECK_H(h);
  if (nranks) *nranks = h->nranks;
  if (rank) *rank = h->rank;
  return PSBA_OK;
}

// ---- measurement ------------------------------------------------------------------------

static int prof_flush(psba_ctx *h) {
  PSBA_HIP(h, hipStreamSynchronize(h->stream));
  for (size_t k = 0; k < h->spans_used; k++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->spans[k].a, h->spans[k].b) == hipSuccess) {
      h->prof_ms[h->spans[k].kind] += ms;
      h->prof_n[h->spans[k].kind]++;
    }
  }
  h->spans_used = 0;
  return PSBA_OK;
}

int psba_profile_enable(psba_handle h, int on) {
  CHECK_H(h);
  TRY(prof_flush(h));
  h->prof = on < 0 ? ~0u : (unsigned)on;  // negative: every class; otherwise a bit mask of classes
  // event pairs are created here, not in front of the first profiled launches (hipEventCreate is
  // not cheap, and those launches may be the ones being timed)
  while (h->prof && h->spans.size() < 512) {
    psba_ctx::Span s;
    s.kind = 0;
    // timing events only: without the system-scope fence a default event carries (it flushes
    // caches between the kernels it sits between, and costs several microseconds itself)
    if (hipEventCreateWithFlags(&s.a, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&s.b, hipEventDisableSystemFence) != hipSuccess)
      break;
    h->spans.push_back(s);
  }
  return PSBA_OK;
}

int psba_profile_reset(psba_handle h) {
  CHECK_H(h);
  TRY(prof_flush(h));
  for (int k = 0; k < PSBA_K_COUNT; k++) {
    h->prof_ms[k] = 0;
    h->prof_n[k] = 0;
  }
  return PSBA_OK;
}

int psba_profile_get(psba_handle h, int kernel, double *total_ms, int *launches) {
  CHECK_H(h);
  if (kernel < 0 || kernel >= PSBA_K_COUNT) return fail(h, PSBA_E_INVALID, "bad kernel class");
  TRY(prof_flush(h));
  if (total_ms) *total_ms = h->prof_ms[kernel];
  if (launches) *launches = h->prof_n[kernel];
  return PSBA_OK;
}

int psba_algorithmic_bytes(psba_handle h, int kernel, double *bytes) {
  CHECK_H(h);
  NEED(h, h->uploaded, "no problem uploaded");
  const double nO = h->d.nO, nP = h->d.nP, nC = h->d.nC, nA = h->d.nA;
  double b = 0;
  switch (kernel) {
    case PSBA_K_LINEARIZE:
      b = nO * (16 + 8 + 144) + nP * (24 + 72 + 24) + nC * (120 + 288 + 48);
      break;
    case PSBA_K_SCHUR:  // SURVEY 8(d) K2, S counted once as the full square
      b = nO * (144 + 8) + nP * (72 + 24) + nC * (288 + 48) + 8 * nA * nA + 8 * nA;
      break;
    case PSBA_K_BACKSUB:  // SURVEY 8(d) K3 incl. the fused new-cost pass
      b = nO * 152 + nP * (72 + 24 + 24 + 24) + nA * 8 + nO * 24 + 8;
      break;
    case PSBA_K_CHOLESKY:
      b = 8 * nA * nA;
      break;
    case PSBA_K_RESIDUAL:
      b = nO * 24 + nP * 24 + nC * 120 + 8;
      break;
    default:
      return fail(h, PSBA_E_INVALID, "no byte model for kernel class %d", kernel);
  }
  if (bytes) *bytes = b;
  return PSBA_OK;
}

// ---- test hook: K2's static schedule without a device ----
struct psba_schur_plan {
  psba_ctx ctx;  // only the plan fields are used; no HIP call is made through it
  psba::SchurPlanHost plan;
  long long nBlocks = 0;
};

psba_schur_plan_t psba_schur_plan_create(int nCams, int n3Dpts, int n2Dprojs, const int *iidx,
                                         const int *jidx) {
  if (nCams <= 0 || n3Dpts <= 0 || n2Dprojs <= 0 || !iidx || !jidx) return nullptr;
  std::vector<int> ptr((size_t)n3Dpts + 1, 0);
  for (int a = 0; a < n2Dprojs; a++) {
    if (iidx[a] < 0 || iidx[a] >= n3Dpts || jidx[a] < 0 || jidx[a] >= nCams) return nullptr;
    // point-major, cameras ascending inside a point, as psba_upload_problem requires
    if (a && (iidx[a] < iidx[a - 1] || (iidx[a] == iidx[a - 1] && jidx[a] <= jidx[a - 1]))) return nullptr;
    ptr[(size_t)iidx[a] + 1]++;
  }
  for (int i = 0; i < n3Dpts; i++) ptr[(size_t)i + 1] += ptr[i];
  psba_schur_plan *p = new (std::nothrow) psba_schur_plan;
  if (!p) return nullptr;
  p->nBlocks = (long long)nCams * (nCams + 1) / 2;
  if (psba::build_schur_plan(&p->ctx, nCams, n3Dpts, n2Dprojs, iidx, jidx, ptr.data(), p->plan) != PSBA_OK) {
    delete p;
    return nullptr;
  }
  return p;
}

int psba_schur_plan_info(psba_schur_plan_t p, long long info[8]) {
  if (!p || !info) return PSBA_E_INVALID;
  info[0] = p->ctx.nGroups;
  info[1] = p->ctx.nGroups ? p->ctx.nWg : 0;
  info[2] = (long long)p->plan.items.size();
  info[3] = p->plan.real_items;
  info[4] = (long long)p->plan.slab_doubles;
  info[5] = p->nBlocks;
  info[6] = p->plan.runs ? p->plan.tasks : 0;  // > 0: the runs layout ([turn][512] per workgroup), number of runs
  info[7] = p->plan.pair_items;                // items that carry two products of one observation
  return PSBA_OK;
}

int psba_schur_plan_copy(psba_schur_plan_t p, unsigned long long *items, long long *wg, int *blockpos,
                         int *glo) {
  if (!p) return PSBA_E_INVALID;
  if (items) std::copy(p->plan.items.begin(), p->plan.items.end(), items);
  if (wg)
    for (size_t k = 0; k < p->plan.wgs.size(); k++) {
      const psba::SchurWg &w = p->plan.wgs[k];
      long long *o = wg + 8 * k;
      o[0] = w.group; o[1] = w.nblk; o[2] = w.obs0; o[3] = w.pt0;
      o[4] = w.item0; o[5] = w.item1; o[6] = (long long)w.slab_off; o[7] = w.itemD;
    }
  if (blockpos) std::copy(p->plan.blockpos.begin(), p->plan.blockpos.end(), blockpos);
  if (glo)  // block ranges: group g owns the blocks [glo[g], glo[g + 1]) of the canonical order tri(j) + k
    for (int g = 0; g <= p->ctx.nGroups; g++) glo[g] = p->ctx.gblk0[g];
  return PSBA_OK;
}

void psba_schur_plan_destroy(psba_schur_plan_t p) { delete p; }

// ---- test hook: the owner route's product lists (many cameras, long tracks, block-sparse S), host only ----
struct psba_owner_plan {
  psba::OwnerPlanHost plan;
};

psba_owner_plan_t psba_owner_plan_create(int nCams, int n3Dpts, int n2Dprojs, const int *iidx, const int *jidx,
                                         const unsigned char *pattern) {
  if (nCams <= 0 || n3Dpts <= 0 || n2Dprojs <= 0 || !iidx || !jidx) return nullptr;
  std::vector<int> ptr((size_t)n3Dpts + 1, 0);
  for (int a = 0; a < n2Dprojs; a++) {
    if (iidx[a] < 0 || iidx[a] >= n3Dpts || jidx[a] < 0 || jidx[a] >= nCams) return nullptr;
    // point-major, cameras ascending inside a point, as psba_upload_problem requires
    if (a && (iidx[a] < iidx[a - 1] || (iidx[a] == iidx[a - 1] && jidx[a] <= jidx[a - 1]))) return nullptr;
    ptr[(size_t)iidx[a] + 1]++;
  }
  for (int i = 0; i < n3Dpts; i++) ptr[(size_t)i + 1] += ptr[i];
  psba_owner_plan *p = new (std::nothrow) psba_owner_plan;
  if (!p) return nullptr;
  if (psba::build_owner_plan(nCams, n2Dprojs, iidx, jidx, ptr.data(), p->plan, pattern) != PSBA_OK) {
    delete p;
    return nullptr;
  }
  return p;
}

int psba_owner_plan_info(psba_owner_plan_t p, long long info[4]) {
  if (!p || !info) return PSBA_E_INVALID;
  info[0] = (long long)p->plan.waves.size();
  info[1] = (long long)p->plan.prod.size() / 64;  // ELL rows
  info[2] = p->plan.products;
  info[3] = (long long)p->plan.blocks.size();
  return PSBA_OK;
}

int psba_owner_plan_copy(psba_owner_plan_t p, long long *waves, int *units, int *prod, int *blocks, int *diag_slot) {
  if (!p) return PSBA_E_INVALID;
  if (waves)
    for (size_t w = 0; w < p->plan.waves.size(); w++) {
      waves[2 * w] = p->plan.waves[w].row0;
      waves[2 * w + 1] = p->plan.waves[w].len;
    }
  if (units)
    for (size_t u = 0; u < p->plan.units.size(); u++) {
      units[4 * u] = p->plan.units[u].j;
      units[4 * u + 1] = p->plan.units[u].k;
      units[4 * u + 2] = p->plan.units[u].multi;
      units[4 * u + 3] = p->plan.units[u].slot;
    }
  if (prod)
    for (size_t t = 0; t < p->plan.prod.size(); t++) {
      prod[2 * t] = p->plan.prod[t].x;
      prod[2 * t + 1] = p->plan.prod[t].y;
    }
  if (blocks)
    for (size_t b = 0; b < p->plan.blocks.size(); b++) {
      blocks[2 * b] = p->plan.blocks[b].x;
      blocks[2 * b + 1] = p->plan.blocks[b].y;
    }
  if (diag_slot) std::copy(p->plan.diag_slot.begin(), p->plan.diag_slot.end(), diag_slot);
  return PSBA_OK;
}

void psba_owner_plan_destroy(psba_owner_plan_t p) { delete p; }

#ifdef PSBA_BUILD_EXPERIMENTS
// ---- test hook: the ring route's schedule (schur_ring_plan.cpp), host only ----
struct psba_ring_plan {
  psba::RingPlanHost plan;
};

psba_ring_plan_t psba_ring_plan_create(int nCams, int n3Dpts, int n2Dprojs, const int *iidx, const int *jidx) {
  if (nCams <= 0 || n3Dpts <= 0 || n2Dprojs <= 0 || !iidx || !jidx) return nullptr;
  std::vector<int> ptr((size_t)n3Dpts + 1, 0);
  for (int a = 0; a < n2Dprojs; a++) {
    if (iidx[a] < 0 || iidx[a] >= n3Dpts || jidx[a] < 0 || jidx[a] >= nCams) return nullptr;
    if (a && (iidx[a] < iidx[a - 1] || (iidx[a] == iidx[a - 1] && jidx[a] <= jidx[a - 1]))) return nullptr;
    ptr[(size_t)iidx[a] + 1]++;
  }
  for (int i = 0; i < n3Dpts; i++) ptr[(size_t)i + 1] += ptr[i];
  psba_ring_plan *p = new (std::nothrow) psba_ring_plan;
  if (!p) return nullptr;
  if (psba::build_ring_plan(nCams, n3Dpts, n2Dprojs, iidx, jidx, ptr.data(), p->plan, true) != PSBA_OK) {
    delete p;
    return nullptr;
  }
  return p;
}

int psba_ring_plan_info(psba_ring_plan_t p, long long info[16]) {
  if (!p || !info) return PSBA_E_INVALID;
  const psba::RingPlanHost &q = p->plan;
  info[0] = q.nR;
  info[1] = q.nS;
  info[2] = q.nWg;
  info[3] = (long long)q.steps.size();
  info[4] = (long long)q.entries.size();
  info[5] = (long long)q.ops.size() / 2;
  info[6] = (long long)q.jobs.size();  // (each list ends in a few entries of slack the kernel may read but never uses)
  info[7] = q.products;
  info[8] = q.slots;
  info[9] = psba::RING_LANES;
  info[10] = psba::RING_SLOTS;
  info[11] = psba::RING_PAGE;
  info[12] = q.lat;
  info[13] = (long long)q.lane_blk.size();
  info[14] = (long long)q.blk_lane0.size();
  info[15] = q.loaded_recs;
  return PSBA_OK;
}

int psba_ring_plan_copy(psba_ring_plan_t p, long long *wg, int *steps, unsigned *entries, int *ops, int *jobs,
                        int *lane_blk, int *blk_lane0, int *rb) {
  if (!p) return PSBA_E_INVALID;
  const psba::RingPlanHost &q = p->plan;
  if (wg)
    for (size_t k = 0; k < q.wgs.size(); k++) {
      const psba::RingWg &w = q.wgs[k];
      long long *o = wg + 13 * k;
      o[12] = w.nwslots;
      o[0] = w.blk0; o[1] = w.nblk; o[2] = w.row0; o[3] = w.nrows; o[4] = w.copy; o[5] = w.nsteps;
      o[6] = w.step0; o[7] = w.ent0; o[8] = w.op0; o[9] = w.job0; o[10] = w.lane0; o[11] = w.bl0;
    }
  if (steps)
    for (size_t k = 0; k < q.steps.size(); k++) {
      steps[4 * k] = q.steps[k].op_begin; steps[4 * k + 1] = q.steps[k].op_end;
      steps[4 * k + 2] = q.steps[k].job_begin; steps[4 * k + 3] = q.steps[k].job_end;
    }
  if (entries) std::copy(q.entries.begin(), q.entries.end(), entries);
  if (ops) std::copy(q.ops.begin(), q.ops.end(), ops);
  if (jobs)
    for (size_t k = 0; k < q.jobs.size(); k++) {
      jobs[4 * k] = q.jobs[k].obs; jobs[4 * k + 1] = q.jobs[k].point;
      jobs[4 * k + 2] = q.jobs[k].slots; jobs[4 * k + 3] = q.jobs[k].earow;
    }
  if (lane_blk) std::copy(q.lane_blk.begin(), q.lane_blk.end(), lane_blk);
  if (blk_lane0) std::copy(q.blk_lane0.begin(), q.blk_lane0.end(), blk_lane0);
  if (rb) std::copy(q.rb.begin(), q.rb.end(), rb);
  return PSBA_OK;
}

void psba_ring_plan_destroy(psba_ring_plan_t p) { delete p; }
#endif  // PSBA_BUILD_EXPERIMENTS

}  // extern "C"
