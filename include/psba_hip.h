/*
 * psba_hip.h -- C ABI of the MI355X-native Schur-complement bundle-adjustment
 * normal-equations path (drop-in for the operator layer of eglrp/PSBA).
 *
 * What it replaces (paths relative to the reference checkout):
 *   PSBA/sba_func.h:10-138      the per-kernel host wrappers called by levmar()/trust_region()
 *   PSBA/cl_spdinv.h:7-18       SPDinv / cholesky / trigMat_inv / trigMat_mul
 *   PSBA/cl_linearalg.h:8-9     matVec_mul
 *   PSBA/cl_psba.h:9-126        PSBA_struct, setup_cl, fill_initBuffer2, fill_idxBuffer,
 *                               release_buffer
 *   PSBA/misc.cpp:178-217       generate_idxs (dense blk_idx/comm3DIdx tables -> CSR inside)
 *   PSBA/levmar.cpp:45-256      levmar() (restated over this ABI as psba_levmar)
 *   PSBA/readparams.cpp:444-519 readInitialSBAEstimate (restated as psba_read_problem)
 *
 * Conventions are the reference's: cnp = 6 (local-quaternion vector part 3 + translation 3),
 * pnp = 3, mnp = 2 (CL_files/PSBA.cl:5-7); all arithmetic fp64; K[5] = (fu,u0,v0,ar,s) per
 * camera, held fixed; initrot = unit quaternion, scalar first; observations sorted
 * point-major, camera ascending inside a point (PSBA/misc.cpp:189-216).
 *
 * Plain pointers and sizes only.  Host arrays are owned by the caller; everything on the
 * device is owned by the opaque handle.  One handle = one GPU = one host thread.  Calls
 * are asynchronous on the handle's stream and synchronise only where a value is returned
 * to the host.  No call exits the process: every entry point returns an int status
 * (0 ok, > 0 numerical, < 0 API / usage error) and psba_last_error() gives the text.
 */
#ifndef PSBA_HIP_H
#define PSBA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psba_ctx *psba_handle;

/* ---- status codes ------------------------------------------------------------------- */
#define PSBA_OK 0
#define PSBA_NOT_SPD 1     /* Cholesky met a non-positive / non-finite pivot (SPDinv ret==1.0,
                              PSBA/cl_spdinv.cpp:85-102, CL_files/SPD_inv.cl:35-38,66) */
#define PSBA_SINGULAR_V 2  /* some |det V_i| < 1e-16 (compute_Vinv ret==1.0,
                              CL_files/compute_Vinv.cl:31-32) */
#define PSBA_PCG_MAXIT 4   /* psba_schur_solve under PSBA_SOLVER_PCG: max_iter iterations without reaching
                              tol; dpa holds the last iterate (an inexact step), psba_pcg_info the residual */
#define PSBA_E_INVALID (-1)
#define PSBA_E_HIP (-2)
#define PSBA_E_RCCL (-3)
#define PSBA_E_IO (-4)
#define PSBA_E_NOMEM (-5)
#define PSBA_E_STATE (-6)  /* verb called before the state it needs exists */

/* iteration flags of the optimiser loops, PSBA/psba.h:12-18 (same numbering) */
#define PSBA_ITER_TURN_TO_LM 1
#define PSBA_ITER_TURN_TO_TR 2
#define PSBA_ITER_CONTINUE 3
#define PSBA_ITER_ERR 4
#define PSBA_ITER_DP_NO_CHANGE 5
#define PSBA_ITER_ERR_SMALL_ENOUGH 6
#define PSBA_ITER_PASS 7

/* which parameter set a verb evaluates: cams_buffer/pts3D_buffer or
 * newCams_buffer/newPts3D_buffer (PSBA/sba_func.h:15-16, PSBA/levmar.cpp:93,188-189) */
#define PSBA_PARAMS_CUR 0
#define PSBA_PARAMS_NEW 1

/* ---- lifecycle: setup_cl / release_buffer (PSBA/cl_psba.h:93-94) --------------------- */
int psba_create(int device, psba_handle *out);
int psba_destroy(psba_handle h);
/* text of the last error on this handle (h may be NULL: last error of a failed create) */
const char *psba_last_error(psba_handle h);
/* version string of the library and the code-object architecture it was built for */
const char *psba_version(void);

/* ---- problem upload: fill_initBuffer2 + generate_idxs + fill_idxBuffer ----------------
 * (PSBA/cl_psba.h:114-126, PSBA/misc.cpp:178-217, call order PSBA/main.cpp:177-189).
 * Kparas[nCams*5], impts[n2Dprojs*2], initrot[nCams*4], camsEx[nCams*6], pts3D[n3Dpts*3],
 * iidx/jidx[n2Dprojs] (point / camera of each observation, point-major sorted).
 * The dense blk_idx / comm3DIdx tables of the reference are not needed: a point CSR and
 * point-aligned observation tiles are built inside. */
int psba_upload_problem(psba_handle h, int nCams, int n3Dpts, int n2Dprojs, const double *Kparas,
                        const double *impts, const double *initrot, const double *camsEx,
                        const double *pts3D, const int *iidx, const int *jidx);
/* overwrite the current parameters (cams[nCams*6], pts[n3Dpts*3]) */
int psba_set_params(psba_handle h, const double *camsEx, const double *pts3D);
/* back to the parameters given to psba_upload_problem (device-side copy, no synchronisation):
 * what a driver that reruns the loop from the same start, like bench.py, would otherwise do
 * with psba_set_params */
int psba_reset_params(psba_handle h);
/* read back the current (PSBA_PARAMS_CUR) or proposed (PSBA_PARAMS_NEW) parameters */
int psba_get_params(psba_handle h, int which, double *camsEx, double *pts3D);
int psba_get_dims(psba_handle h, int *nCams, int *n3Dpts, int *n2Dprojs);

/* ---- free intrinsics ---------------------------------------------------------------
 * The reference's driver reads 11 parameters per camera -- (fu, u0, v0, ar, s), quaternion -> local rotation,
 * translation (PSBA/main.cpp:73,140-149; data/54camsvarK.txt) -- and then strips the intrinsics: its kernels
 * optimise six (CL_files/PSBA.cl:5-7).  PSBA_CAMERA_FREE_K optimises all eleven: the camera block becomes
 * (fu, u0, v0, ar, s | v0, v1, v2 | t0, t1, t2), nA = 11 nCams, dp = [11 per camera ; 3 per point].
 * psba_upload_problem takes the same arrays (Kparas and camsEx are joined inside); psba_get_params /
 * psba_set_params then move 11 doubles per camera.  One plain route (kernels_freek.hip: global-atomic assembly,
 * the generic dense factorization), single rank, dense solver; the fused verbs and psba_levmar work, the
 * sba_func.h mirror, the trust-region operators and psba_solve return PSBA_E_STATE.  The reference has no
 * arithmetic for this (it never implemented it): PARITY UNPINNED -- checked against the oracle's twin, finite
 * differences and a dense solve of the full normal equations.  Before psba_upload_problem. */
#define PSBA_CAMERA_FIXED_K 0
#define PSBA_CAMERA_FREE_K 1
int psba_set_camera_model(psba_handle h, int model);
int psba_camera_block(psba_handle h, int *cnp); /* 6 or 11 */
/* which S-assembly route the uploaded problem takes: 0 = LDS-resident partitions of the block
 * triangle with the static schedule (fewer than 2048 cameras, up to 8 GB of partial-sum slabs on
 * this rank), 1 = the owner route for larger problems (one thread per block segment,
 * products sorted by camera pair, sums in registers; PSBA_SCHUR_OWNER=1 forces it), 2 = global
 * fp64 atomics straight into S (the
 * first-generation kernel, kept for cross-checks: PSBA_SCHUR_ATOMIC=1), 3 = the ring route (opt-in
 * experiment, PSBA_SCHUR_RING=1), 4 = block-sparse S (PSBA_SOLVER_PCG).  The reference has one
 * route for every size (CL_files/compute_S.cl:6-78). */
int psba_schur_path(psba_handle h, int *path);

/* ---- fused verbs (what the optimiser loops call) --------------------------------------
 * One damping try = psba_schur_assemble -> psba_schur_reduce -> psba_schur_solve ->
 * psba_backsub; nothing is copied to the host in between. */

/* ||e||^2 at a parameter set: compute_exQT + compute_L2_sq
 * (PSBA/sba_func.h:10-19, PSBA/levmar.cpp:93-94,188-193, PSBA/misc.cpp:151-157).
 * With a communicator attached the value is summed over all ranks. */
int psba_residual(psba_handle h, int which, double *cost);
/* compute_jacobiQT + compute_U + compute_V + compute_Wblks + compute_g in one kernel
 * (PSBA/sba_func.h:26-56,74-81,103-109; PSBA/levmar.cpp:103-108).  coeff scales U,V,W
 * (1 in LM, 2 in TR), coeff_g scales g (1 in LM, -2 in TR; PSBA/trust_region.cpp:122,133-137).
 * A/B/e are never written to memory. */
int psba_linearize(psba_handle h, double coeff, double coeff_g);
/* maxElmOfUV (PSBA/sba_func.h:58): max over diag(U), diag(V); global over ranks */
int psba_max_diag(psba_handle h, double *out);
/* The loop's prologue in one call and one synchronisation (PSBA/levmar.cpp:93-120): the cost at
 * the current parameters, their linearization (as psba_linearize) and its largest diagonal
 * entry.  The psba_linearize that follows with the same coefficients has nothing left to do. */
int psba_begin(psba_handle h, double coeff, double coeff_g, double *cost, double *max_diag);
/* update_UV + compute_Vinv + compute_Yblks + compute_S + compute_ea
 * (PSBA/sba_func.h:60-68,84-98,114-119; PSBA/levmar.cpp:126-131).  mu is applied on the
 * fly, U/V are not modified, so there is no restore_UVdiag.  Leaves this rank's
 * contribution to [S | ea] in the reduce buffer. */
int psba_schur_assemble(psba_handle h, double mu);
/* sum [S | ea] over ranks with one RCCL all-reduce on the handle's stream; no-op without a
 * communicator.  (No counterpart in the reference: it is single-device.) */
int psba_schur_reduce(psba_handle h);
/* SPDinv + matVec_mul as one Cholesky factorisation and two triangular solves
 * (PSBA/cl_spdinv.h:7-8, PSBA/cl_linearalg.h:8-9, PSBA/levmar.cpp:134-140).  The status
 * (PSBA_OK / PSBA_NOT_SPD) stays on the device until psba_backsub returns it. */
int psba_schur_solve(psba_handle h);

typedef struct {
  int status;       /* PSBA_OK, PSBA_NOT_SPD, PSBA_SINGULAR_V (bit-or) */
  double dp_l2;     /* ||dp||^2                         PSBA/levmar.cpp:157 */
  double gain_den;  /* sum dp*(mu*dp+g)                 PSBA/levmar.cpp:271-280 */
  double new_cost;  /* ||e(p+dp)||^2                    PSBA/levmar.cpp:188-193 */
  double newp_l2;   /* ||p+dp||^2                       PSBA/levmar.cpp:212 */
} psba_try_scalars;

/* compute_eb + compute_dpb + compute_newp + compute_exQT(new) + the host reductions the
 * loop needs, in one kernel (PSBA/sba_func.h:124-137, PSBA/levmar.cpp:151-195).
 * Synchronises and fills *out (all-reduced over ranks). */
int psba_backsub(psba_handle h, double mu, psba_try_scalars *out);
/* The same in two halves, so that the host's decision (PSBA/levmar.cpp:169-223) no longer idles
 * the GPU: psba_backsub_async queues the kernel and the copy of its scalars and returns;
 * psba_linearize_ahead then queues psba_linearize's work for the PROPOSED parameters into a
 * second set of linearization buffers (the current linearization stays valid); psba_backsub_wait
 * waits for the scalars only.  If the step is accepted, psba_accept makes the proposed
 * parameters AND the linearization computed ahead current, and the next psba_linearize (same
 * coefficients) has nothing left to do; if it is rejected the work done ahead is dropped. */
int psba_backsub_async(psba_handle h, double mu);
int psba_linearize_ahead(psba_handle h);
int psba_backsub_wait(psba_handle h, psba_try_scalars *out);
/* update_p (PSBA/sba_func.h:138): the proposed parameters become current (pointer swap) */
int psba_accept(psba_handle h);

/* ---- 1:1 mirror of sba_func.h for per-kernel parity tests -----------------------------
 * Same names and argument meaning as the reference; the dimension arguments and
 * PSBA_structPtr collapse into the handle.  A NULL host pointer means "stay on device"
 * exactly as in the reference.  Each verb leaves the device state as the reference's
 * wrapper of the same name does. */
int psba_compute_exQT(psba_handle h, int which, double *ex);             /* sba_func.h:10-19 */
int psba_compute_jacobiQT(psba_handle h, double *jac_A, double *jac_B);  /* sba_func.h:26-32 */
int psba_compute_U(psba_handle h, double coeff, double *out);            /* sba_func.h:38-44 */
int psba_compute_V(psba_handle h, double coeff, double *out);            /* sba_func.h:50-56 */
int psba_maxElmOfUV(psba_handle h, double *out);                         /* sba_func.h:58 */
int psba_update_UV(psba_handle h, double mu, double *U, double *V);      /* sba_func.h:60-61 */
int psba_restore_UVdiag(psba_handle h);                                  /* sba_func.cpp:694 */
int psba_compute_Vinv(psba_handle h, double *Vinv /* n3Dpts*9, full symmetric */); /* :63-68 */
int psba_compute_Wblks(psba_handle h, double coeff, double *Wblks);      /* sba_func.h:74-81 */
int psba_compute_Yblks(psba_handle h, double *Yblks);                    /* sba_func.h:84-90 */
int psba_compute_S(psba_handle h, double *S /* (6 nCams)^2 row-major */); /* sba_func.h:93-98 */
int psba_compute_g(psba_handle h, double coeff, double *g);              /* sba_func.h:103-109 */
int psba_compute_ea(psba_handle h, double *ea);                          /* sba_func.h:114-119 */
/* SPDinv + matVec_mul: returns PSBA_OK / PSBA_NOT_SPD like SPDinv's 0.0 / 1.0; dpa[6 nCams] */
int psba_SPDinv_matVec(psba_handle h, double *dpa);          /* cl_spdinv.h:7, cl_linearalg.h:8 */
int psba_compute_eb(psba_handle h, double *eb /* 3 n3Dpts */);           /* sba_func.h:124-129 */
int psba_compute_dpb(psba_handle h, double *dp /* whole dp, nT */);      /* sba_func.h:131-135 */
int psba_compute_newp(psba_handle h, double *new_p /* nT */);            /* sba_func.h:137 */
int psba_update_p(psba_handle h, double *p /* nT */);                    /* sba_func.h:138 */

/* ---- the Levenberg-Marquardt caller, PSBA/levmar.cpp:45-256 --------------------------- */
typedef struct {
  int max_iter;     /* literal 50 in the reference (levmar.cpp:100) */
  int tr_handoff;   /* 1: return PSBA_ITER_TURN_TO_TR after 5 consecutive |rho-1|<0.2
                       (levmar.cpp:215-219); 0: stay in LM */
  int verbose;      /* print the reference's per-try line (levmar.cpp:197) */
  int log_cap;      /* rows available in log (5 doubles each), 0 = none */
  int start_itno;   /* the reference shares itno between LM and TR (main.cpp:193-208) */
  double init_mu;   /* mu_0 = init_mu * max diag(U, V); 0 = the reference's PSBA_INIT_MU 1e-3 (psba.h:6,
                       levmar.cpp:114-116), a compile-time constant there */
} psba_lm_options;

typedef struct {
  int flag;         /* PSBA_ITER_* */
  int iters;        /* value of itno at exit */
  int tries;        /* damping tries executed */
  double init_err, final_err, mu0, mu_final;
  int n_log;
  double seconds;   /* wall time inside the loop */
  int pcg_unconverged; /* PSBA_SOLVER_PCG: solves of this call that returned PSBA_PCG_MAXIT */
} psba_lm_result;

void psba_lm_default_options(psba_lm_options *o);
/* log rows: (itno, new ||e||^2, rho, mu, accepted{1,0,-1=solve failed}) per damping try */
int psba_levmar(psba_handle h, const psba_lm_options *opts, psba_lm_result *res, double *log);

/* ---- the trust-region caller, PSBA/trust_region.cpp:49-595, and its extra operators -------
 * (B = 2 J^T J, g = -2 J^T e through psba_linearize(h, 2, -2); with a communicator the points may be
 * sharded: psba_jmul_dots sums its dot products over the ranks, psba_get_gradient returns the
 * complete g_a, the modified Cholesky runs replicated on the all-reduced S). */
/* compute_Jmultiply (PSBA/sba_func.cpp:19-75, CL_files/compute_Jmultiply.cl:6-52): J x for a host
 * vector x[nT].  Jmul has 2 values per OBSERVATION (2 * n2Dprojs, observation order) -- the
 * non-zeros, in the same order, of the reference's dense nP x nC x 2 grid. */
int psba_compute_Jmultiply(psba_handle h, const double *x, double *Jmul);
/* what the loop needs of it: dots = (Jx1.Jx1, Jx1.Jx2, Jx2.Jx2); x2 NULL = x1
 * (trust_region.cpp:125-126,166-176,209-211 take these dot products on the host) */
int psba_jmul_dots(psba_handle h, const double *x1, const double *x2, double dots[3]);
/* v[0..n) (n <= 8) summed over the ranks of the communicator, in place; no communicator: untouched.
 * What a host loop over sharded points needs for its inner products: the point parts of two
 * vectors are rank-local, so a dot product is (camera part) + sum over ranks of (point part). */
int psba_allreduce_scalars(psba_handle h, double *v, int n);
/* g = [g_a ; g_b] of the last linearization (the host output of compute_g, sba_func.h:103-109) */
int psba_get_gradient(psba_handle h, double *g);
/* dp = [dpa ; dpb] of the last solve + back-substitution (host output of compute_dpb) */
int psba_get_dp(psba_handle h, double *dp);
/* dp_buffer <- dp, then compute_newp: proposed parameters = current + dp
 * (trust_region.cpp:183-187); psba_residual(PSBA_PARAMS_NEW) evaluates them, psba_accept takes them */
int psba_set_step(psba_handle h, const double *dp);
/* the lambda estimate taken when S is not positive definite at lambda = 0
 * (trust_region.cpp:341-363): S is assembled again without damping, factored by the modified
 * Cholesky of PSBA/cl_cholmod.cpp:25-201 / CL_files/cholmod_blk.cl:87-846, and
 * lambda = |sum_i E_i| / (6 nCams) with E = diag(L L^T) - diag(S).  info3 (may be NULL) =
 * (delta, beta, number of block columns that took the one-column route).  reassemble = 0 skips
 * the assembly and factors what the reduce buffer holds (psba_schur_assemble or
 * psba_set_reduce_buffer before it): the hook the tests use to factor a chosen matrix.
 * PSBA_SOLVER_PCG (no dense S, and the reference has no sparse mode): lambda is the Gershgorin shift of the
 * stored blocks instead, max(0, -min_i (S_ii - sum_{c != i} |S_ic|)) -- the smallest shift that makes S + lambda I
 * diagonally dominant -- or 1e-6 max_i of those margins when S is diagonally dominant already;
 * info3 = (min margin, max margin, 0). */
int psba_cholmod_lambda(psba_handle h, int reassemble, double *lambda, double *info3);

typedef struct {
  int max_iter;     /* literal 50, shared with levmar() through itno (trust_region.cpp:112) */
  int start_itno;
  int verbose;      /* print the reference's per-step line (trust_region.cpp:250) */
  int log_cap;      /* rows available in log (6 doubles each), 0 = none */
  double init_lambda; /* damping the loop starts with; 0 = the reference's (trust_region.cpp:95-96), where the
                         first factorization of the gauge-free S fails and the modified Cholesky picks lambda */
} psba_tr_options;

typedef struct {
  int flag;         /* PSBA_ITER_* */
  int iters;        /* value of itno at exit */
  int tries;        /* steps evaluated */
  int chol_fail;    /* factorizations of S that failed (each followed by a larger lambda) */
  double init_err, final_err, lambda, delta;
  int n_log;
  double seconds;
} psba_tr_result;

void psba_tr_default_options(psba_tr_options *o);
/* log rows: (itno, ||e(p + step)||^2, rho, delta after the update, lambda, accepted{1,0}) per step */
int psba_trust_region(psba_handle h, const psba_tr_options *opts, psba_tr_result *res, double *log);

/* the driver's alternation, PSBA/main.cpp:193-208: levmar() until ITER_TURN_TO_TR, trust_region()
 * until ITER_TURN_TO_LM, max_iter iterations in total */
typedef struct {
  int flag, iters, lm_calls, tr_calls;
  double init_err, final_err, seconds;
} psba_solve_result;
int psba_solve(psba_handle h, int max_iter, int verbose, psba_solve_result *res);

/* ---- multi-GPU: 3-D points sharded over ranks, one process per GPU -------------------- */
/* contiguous point ranges balanced on observation count; pt_begin[nranks+1] */
int psba_partition_points(int n3Dpts, const int *iidx, int n2Dprojs, int nranks, int *pt_begin);
/* 128-byte RCCL unique id, created on rank 0 and handed to every rank by the launcher */
int psba_comm_unique_id(void *id128);
int psba_comm_init(psba_handle h, int nranks, int rank, const void *id128);
int psba_comm_rank(psba_handle h, int *nranks, int *rank);
/* Bring-your-own reduction (MPI, host staging, tests): declare the rank layout without an RCCL
 * communicator, then sum the reduce buffer yourself between psba_schur_assemble and
 * psba_schur_solve.  The buffer is the padded [S | ea] block: psba_reduce_buffer_size doubles.
 * With a layout but no communicator psba_backsub / psba_residual / psba_max_diag return this
 * rank's partial sums (camera terms counted on rank 0 only), to be summed by the caller. */
int psba_set_rank_layout(psba_handle h, int nranks, int rank);
int psba_reduce_buffer_size(psba_handle h, long long *n_doubles);
int psba_get_reduce_buffer(psba_handle h, double *out);
int psba_set_reduce_buffer(psba_handle h, const double *in);

/* ---- on-disk format: readInitialSBAEstimate (PSBA/readparams.cpp:444-519) -------------
 * Reads an sba-format cams file (7 columns q,t with fixedK[5] replicated, or 12 columns
 * K5,q,t) and pts file; applies quat2vec (PSBA/misc.cpp:21-49), zeroes the local rotation
 * and splits K from the extrinsics (PSBA/main.cpp:131-149).  Arrays are malloc'ed by the
 * library and released with psba_free_problem. */
typedef struct {
  int nCams, n3Dpts, n2Dprojs;
  double *Kparas, *impts, *initrot, *camsEx, *pts3D;
  int *iidx, *jidx;
} psba_problem;
int psba_read_problem(const char *cams_file, const char *pts_file, const double *fixedK,
                      psba_problem *out);
void psba_free_problem(psba_problem *p);
/* The writer the reference declares and keeps commented out (printSBAMotionData /
 * printSBAStructureData / printSBAData, PSBA/readparams.h:13-25; output filter vec2quat,
 * PSBA/misc.cpp:60-85): cams file with one line per camera -- K5 when with_K != 0, then the full
 * quaternion q_l(v) (x) initrot and t -- and pts file "X Y Z nframes {frame x y}...".  The files
 * are valid input of psba_read_problem (which then returns initrot = that quaternion, v = 0). */
int psba_write_problem(const char *cams_file, const char *pts_file, int nCams, int n3Dpts, int n2Dprojs,
                       const double *Kparas, const double *initrot, const double *camsEx, const double *pts3D,
                       const double *impts, const int *iidx, const int *jidx, int with_K);
/* Bundle-Adjustment-in-the-Large text file -> sba cams / pts files in the reference's camera
 * model (how its data/Trafalgar-* files relate to the public BAL sets): camera turned to look
 * down +z, image y negated, K = (f,0,0,1,0); the radial terms k1, k2 are dropped (the reference
 * has no distortion) and their largest magnitude is returned in *max_abs_k (may be NULL). */
int psba_convert_bal(const char *bal_file, const char *cams_out, const char *pts_out, double *max_abs_k);

/* ---- measurement ----------------------------------------------------------------------
 * HIP-event timing of the kernels launched by the fused verbs, on the handle's stream. */
#define PSBA_K_LINEARIZE 0
#define PSBA_K_SCHUR 1
#define PSBA_K_CHOLESKY 2
#define PSBA_K_BACKSUB 3
#define PSBA_K_RESIDUAL 4
#define PSBA_K_ALLREDUCE 5
#define PSBA_K_SCHUR_REDUCE 6 /* slab sum + U + mu I + mirror that finishes S after PSBA_K_SCHUR */
#define PSBA_K_COUNT 7
/* on < 0: time every kernel class; on > 0: bit mask (1 << PSBA_K_*) of the classes to time;
 * 0: off.  Every timed launch costs two event records on the stream (~3 us). */
int psba_profile_enable(psba_handle h, int on);
int psba_profile_reset(psba_handle h);
/* total_ms and launch count per kernel class since the last reset (synchronises) */
int psba_profile_get(psba_handle h, int kernel, double *total_ms, int *launches);
/* algorithmic bytes of one launch of a kernel class for the uploaded problem
 * (SURVEY.md section 8(d) formulas; stated in DESIGN.md) */
int psba_algorithmic_bytes(psba_handle h, int kernel, double *bytes);

/* ---- block-sparse S and an iterative solve -----------------------------------------
 * The reference forms the dense nA x nA S for every problem (CL_files/compute_S.cl:6-78) and inverts it
 * (PSBA/cl_spdinv.cpp:18-40).  With PSBA_SOLVER_PCG only the 6x6 blocks of camera pairs that see a
 * common point exist (the lower block triangle as a list; no dense buffer is allocated), and
 * psba_schur_solve runs conjugate gradients with a block-Jacobi preconditioner until
 * ||S x - e_a|| <= tol ||e_a|| or max_iter iterations; a diagonal block that is not positive definite or
 * a direction of non-positive curvature reports PSBA_NOT_SPD, so psba_levmar works unchanged; a solve
 * that uses up max_iter returns PSBA_PCG_MAXIT (> 0: the step is usable, the LM gain ratio judges it, and
 * psba_levmar counts such solves in psba_lm_result.pcg_unconverged); e_a == 0 or an exactly solved system
 * is a converged solve with x as it stands, not a break-down.  The
 * sba_func.h mirror verbs and the trust-region operators need the dense S and refuse this mode.
 * Sharded points: every rank must hold the same block list, the union of the blocks its ranks' points
 * produce.  With a communicator psba_upload_problem forms it by itself (one max all-reduce of a byte per
 * block); a host with its own transport computes psba_sparse_pattern on every rank (host only, no
 * handle), ORs the flags, and hands them to psba_set_sparse_pattern before psba_upload_problem; between
 * psba_schur_assemble and psba_schur_solve it then sums psba_get_sparse_S over the ranks and returns the
 * sums with psba_set_sparse_S (the conjugate gradients themselves run replicated on every rank).
 * psba_set_solver: before psba_upload_problem; tol <= 0 / max_iter <= 0 keep 1e-10 / 500. */
#define PSBA_SOLVER_DENSE 0
#define PSBA_SOLVER_PCG 1
int psba_set_solver(psba_handle h, int solver, double tol, int max_iter);
/* iterations and ||r|| / ||e_a|| of the last solve; blocks stored / blocks of the dense lower triangle */
int psba_pcg_info(psba_handle h, int *iters, double *relres, long long *blocks, long long *dense_blocks);
/* the assembled blocks: jk[2 blocks] = (j, k), k <= j; val[36 blocks] row-major; ea[nA]; any pointer may be NULL */
int psba_get_sparse_S(psba_handle h, int *jk, double *val, double *ea);
int psba_set_sparse_S(psba_handle h, const double *val, const double *ea);
/* flags[nCams (nCams + 1) / 2]: 1 where block tri(j) + k (k <= j) has a product, i.e. cameras j and k see a common point */
int psba_sparse_pattern(int nCams, int n3Dpts, int n2Dprojs, const int *iidx, const int *jidx, unsigned char *flags);
int psba_set_sparse_pattern(psba_handle h, const unsigned char *flags, long long n);

/* ---- the dense factorization sharded over ranks (large matrices: the two-level chain) --------------
 * The reference factors S on one device (PSBA/cl_spdinv.cpp:18-40, CL_files/SPD_inv.cl:165-179).
 * Here every rank holds the complete S after the all-reduce and factors the super-panels (NB columns)
 * itself; the K = NB update of everything to their right -- nearly all of the n^3 / 3 flops -- is
 * shared: a rank updates the 64-column blocks B (columns [64 B, 64 B + 64)) with B % nranks == rank,
 * and before a super-panel is factored the owners of its blocks send them (rows 64 B .. n32, the
 * e_a row included, packed row by row) to everybody.  With a communicator psba_schur_solve runs this
 * by itself (ncclBroadcast; PSBA_CHOL_REPLICATED=1 keeps the factorization replicated); the pieces
 * below let a host with its own transport -- or a test with several handles -- drive it:
 *   psba_chol_dist_shape     n32, NB, and whether the matrix is large enough for the sharded chain
 *   psba_chol_dist_begin     the first diagonal block
 *   psba_chol_dist_superpanel(J)  the 32-column steps of the super-panel at column J (its columns
 *                            must be complete on this rank) and this rank's share of the update
 *   psba_chol_dist_block(B, set, buf, &n)  get (set = 0) / set the packed block B; n = its doubles
 *   psba_chol_dist_finish    the backward solve: dpa, as after psba_schur_solve */
int psba_chol_dist_shape(psba_handle h, int *n32, int *NB, int *sharded);
/* host only: the column exchange in front of the super-panel at column JE, as the RCCL path performs it --
 * out4[k] = {block B, owner rank (B mod nranks), slot of the exchange buffer, doubles}; returns the number of
 * blocks (0: none, the last super-panel), < 0 if cap is too small.  The owner of block B packs rows 64 B .. n32
 * (the e_a row) of its min(64, n32 - 64 B) columns, everybody else unpacks them. */
int psba_chol_dist_exchange_plan(int n32, int NB, int nranks, int JE, long long *out4, int cap);
int psba_chol_dist_begin(psba_handle h);
int psba_chol_dist_superpanel(psba_handle h, int J);
int psba_chol_dist_block(psba_handle h, int B, int set, double *buf, long long *n_doubles);
int psba_chol_dist_finish(psba_handle h);

/* ---- test hook: the static schedule of the S-assembly kernel, built on the host only ----
 * (no device needed).  The reference decides the same placement per launch through its
 * blkIdx_buffer look-ups (CL_files/compute_S.cl:13-22); here it is data that can be checked.
 * info[0..5] = groups, workgroups, item slots, products, slab doubles, blocks nC(nC+1)/2.
 * psba_schur_plan_copy: items[info[2]]; wg[info[1]][8] = group, blocks in partition, obs0, pt0,
 * first item, end item, slab offset, end of the pair items (the list begins with items that carry
 * two products of one observation a -- partners a - boff and a - boff + 1 -- as 18 bits a - obs0,
 * 16 bits i - pt0, 8 bits boff, 10 + 10 bits block positions; the single-product items follow:
 * 24 / 22 / 8 / 10 bits); blockpos[info[5]]; glo[groups+1]: group g owns the blocks
 * [glo[g], glo[g+1]) of the canonical order j (j + 1) / 2 + k.  Any pointer may be NULL. */
typedef struct psba_schur_plan *psba_schur_plan_t;
psba_schur_plan_t psba_schur_plan_create(int nCams, int n3Dpts, int n2Dprojs, const int *iidx,
                                         const int *jidx);
int psba_schur_plan_info(psba_schur_plan_t p, long long info[8]); /* info[6] > 0: the runs layout (items as [turn][512] per workgroup, a thread's consecutive items grouped into runs of one position), number of runs; info[7] = pair items */
int psba_schur_plan_copy(psba_schur_plan_t p, unsigned long long *items, long long *wg,
                         int *blockpos, int *glo);
void psba_schur_plan_destroy(psba_schur_plan_t p);

/* ---- test hook: the product lists of the owner route (K2 for >= 2048 cameras, for points seen by more
 * cameras than a tile holds, and for the block-sparse S), built on the host only.  Every product
 * Y_a W_b^T belongs to the block (camera of a, camera of b) of the lower block triangle; a thread owns a
 * unit = (block, segment of its products), 64 units to a wave, a wave's products in ELL rows.
 * info[0..3] = waves, ELL rows, products, blocks in the block list.
 * psba_owner_plan_copy: waves[waves][2] = first ELL row, rows; units[64 waves][4] = j, k, multi (the block
 * has several units), slot in the block list (-1: idle lane); prod[64 rows][2] = observations (a, b), a = -1
 * padding; blocks[blocks][2] = (j, k) in canonical order; diag_slot[nCams].  pattern (may be NULL): the
 * union pattern of psba_sparse_pattern -- blocks flagged there exist even without a product here. */
typedef struct psba_owner_plan *psba_owner_plan_t;
psba_owner_plan_t psba_owner_plan_create(int nCams, int n3Dpts, int n2Dprojs, const int *iidx, const int *jidx,
                                         const unsigned char *pattern);
int psba_owner_plan_info(psba_owner_plan_t p, long long info[4]);
int psba_owner_plan_copy(psba_owner_plan_t p, long long *waves, int *units, int *prod, int *blocks, int *diag_slot);
void psba_owner_plan_destroy(psba_owner_plan_t p);

#ifdef PSBA_BUILD_EXPERIMENTS
/* (Only in a library built with PSBA_BUILD_EXPERIMENTS=1: round 3's ring route is an experiment that lost,
 * DESIGN 5c, and is not part of the product library.)
 * ---- test hook: the schedule of the S-assembly kernel's ring route (few cameras), host only ----
 * The lower block triangle of S (reference CL_files/compute_S.cl:6-78) is cut into nR ranges of the
 * canonical block order and the point sequence into nS stretches; a workgroup per (range, stretch)
 * replays per-step lists: which runs of W records (up to 7, contiguous in W) are loaded into
 * which slots of its LDS, which observations get their Y = W V^-1 formed (jobs), and which
 * product every consumer lane takes.
 * info[0..15] = nR, nS, workgroups (0: the route does not apply), steps, lane entries, page
 * loads, jobs, products, lane-steps, lanes per workgroup, LDS slots, records per page load at most, steps between a
 * load and its first use, lane_blk entries, blk_lane0 entries, record loads.
 * psba_ring_plan_copy: wg[workgroups][13] = first block, blocks, first row, rows, copy, steps,
 * the first index of the workgroup in steps / entries / ops / jobs / lane_blk / blk_lane0, and its
 * W slots (Y slots are numbered from 0 and live behind them);
 * steps[.][4] = page loads [begin, end) and jobs [begin, end) of the step, relative to the
 * workgroup; entries[step][lanes] = partner W slot | Y slot << 16 (0xFFFFFFFF: idle);
 * ops[.][2] = first observation of a page load, first W slot | records << 16;
 * jobs[.][4] = observation, point, W slot | Y slot << 16, e_a row relative to the first row or -1;
 * lane_blk[wg][lanes] = block (relative) a lane owns; blk_lane0[.] per workgroup blocks + 1 first
 * lanes; rb[nR + 1] block range bounds.  Any pointer may be NULL. */
typedef struct psba_ring_plan *psba_ring_plan_t;
psba_ring_plan_t psba_ring_plan_create(int nCams, int n3Dpts, int n2Dprojs, const int *iidx,
                                       const int *jidx);
int psba_ring_plan_info(psba_ring_plan_t p, long long info[16]);
int psba_ring_plan_copy(psba_ring_plan_t p, long long *wg, int *steps, unsigned *entries, int *ops,
                        int *jobs, int *lane_blk, int *blk_lane0, int *rb);
void psba_ring_plan_destroy(psba_ring_plan_t p);
#endif /* PSBA_BUILD_EXPERIMENTS */

#ifdef __cplusplus
}
#endif
#endif
