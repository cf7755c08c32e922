/*
 * psba_oracle.h -- CPU restatement of the eglrp/PSBA Schur-complement bundle-adjustment
 * normal-equations path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this library.  The product path
 * (psba_amd/, include/psba_hip.h) never links or calls it.
 *
 * Parity pin: the reference cannot be built in this image (its host code needs the
 * Windows SDK / OpenCL headers and its kernels need an OpenCL-2.0 compiler; writing
 * stand-ins for those is not allowed), and the reference ships no tests.  This
 * restatement is therefore pinned against the golden scalars of SURVEY.md section 8(c)
 * (tests/golden/survey_8c.json), which the survey obtained by executing the reference's
 * own kernel sources serially.  See tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Conventions (same as the reference): cnp=6, pnp=3, mnp=2 (CL_files/PSBA.cl:5-7), fp64,
 * observations sorted point-major (i ascending, then j ascending; PSBA/misc.cpp:189-216).
 */
#ifndef PSBA_ORACLE_H
#define PSBA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* status codes of the LM loop, PSBA/psba.h:12-18 */
#define ORC_ITER_TURN_TO_LM 1
#define ORC_ITER_TURN_TO_TR 2
#define ORC_ITER_CONTINUE 3
#define ORC_ITER_ERR 4
#define ORC_ITER_DP_NO_CHANGE 5
#define ORC_ITER_ERR_SMALL_ENOUGH 6

/* CL_files/compute_exQT.cl:18-71 ; PSBA/levmar_func_cpu.cpp:82-140,185-223 */
void orc_compute_exQT(int nO, const double *K, const double *impts, const double *initrot,
                      const double *cams, const double *pts, const int *iidx, const int *jidx,
                      double *ex);

/* PSBA/levmar_func_cpu.cpp:82-140 with :147-178 and :185-223: the reference's CPU twin of the
 * residual in its own arithmetic (eight-multiplication quaternion product, expanded sandwich
 * rotation, reciprocal); K is ONE intrinsics vector shared by all cameras. */
void orc_compute_proj_err_twin(int nO, int nC, const double *K, const double *impts,
                               const double *initrot, const double *cams, const double *pts,
                               const int *iidx, const int *jidx, double *ex);

/* CL_files/compute_jacobiQT.cl:7-141 ; PSBA/levmar_func_cpu.cpp:28-49,227-455.
 * Own analytic derivation from the projection model, not the
 * Maple-generated expression list. */
void orc_compute_jacobiQT(int nO, const double *K, const double *initrot, const double *cams,
                          const double *pts, const int *iidx, const int *jidx, double *JA,
                          double *JB);

/* CL_files/compute_U.cl:5-35 */
void orc_compute_U(int nC, int nO, const double *JA, const int *jidx, double coeff, double *U,
                   double *UVdiag);
/* CL_files/compute_V.cl:6-38 */
void orc_compute_V(int nC, int nP, int nO, const double *JB, const int *iidx, double coeff,
                   double *V, double *UVdiag);
/* CL_files/compute_Wblks.cl:7-34 */
void orc_compute_Wblks(int nO, const double *JA, const double *JB, double coeff, double *W);
/* CL_files/compute_g.cl:6-60 */
void orc_compute_g(int nC, int nP, int nO, double coeff, const double *JA, const double *JB,
                   const int *iidx, const int *jidx, const double *ex, double *g);
/* PSBA/sba_func.cpp:422-444 */
double orc_maxElmOfUV(int nT, const double *UVdiag);
/* CL_files/update_UV.cl:5-31 */
void orc_update_UV(int nC, int nP, double *U, double *V, double mu);
/* CL_files/restore_UVdiag.cl:2-25 */
void orc_restore_UVdiag(int nC, int nP, double *U, double *V, const double *UVdiag);
/* CL_files/compute_Vinv.cl:6-90 (main branch :76-86).  Vinv is written as a full symmetric
 * 3x3 per point into its own array instead of the lower triangle of V.  Returns 1.0 when
 * any |T| < 1e-16 (the reference's ret flag), else 0.0. */
double orc_compute_Vinv(int nP, const double *V, double *Vinv);
/* CL_files/compute_Yblks.cl:6-39 */
void orc_compute_Yblks(int nO, const int *iidx, const double *W, const double *Vinv, double *Y);
/* CL_files/compute_S.cl:6-78 (point-major accumulation; per-entry summation order is
 * i ascending, the same order comm3DIdx yields, PSBA/misc.cpp:189-216) */
void orc_compute_S(int nC, int nP, int nO, const int *iidx, const int *jidx, const double *U,
                   const double *Y, const double *W, double *S);
/* CL_files/compute_ea.cl:6-37 */
void orc_compute_ea(int nC, int nP, int nO, const int *iidx, const int *jidx, const double *Y,
                    const double *g, double *eab);
/* PSBA/cl_spdinv.cpp:18-40 + PSBA/cl_linearalg.cpp:19 as one Cholesky solve S*dpa = ea.
 * S is overwritten by its factor.  Returns 0.0 ok, 1.0 not SPD / non-finite
 * (CL_files/SPD_inv.cl:35-38,66). */
double orc_chol_solve(int n, double *S, const double *ea, double *dpa);
/* CL_files/compute_eb.cl:6-41 */
void orc_compute_eb(int nC, int nP, int nO, const int *iidx, const int *jidx, const double *W,
                    const double *dp, const double *g, double *eab);
/* CL_files/compute_dpb.cl:6-35 */
void orc_compute_dpb(int nC, int nP, const double *Vinv, const double *eab, double *dp);
/* CL_files/compute_newp.cl:6-26 */
void orc_compute_newp(int nA, int nB, const double *cams, const double *pts, const double *dp,
                      double *newcams, double *newpts);
/* PSBA/misc.cpp:151-157 */
double orc_L2_sq(int n, const double *x);

typedef struct {
  int max_iter;     /* PSBA/levmar.cpp:100 literal 50 */
  int tr_handoff;   /* 1: return ITER_TURN_TO_TR after 5 good iterations (levmar.cpp:215-219) */
  int verbose;
  int log_cap;      /* capacity (rows) of log, 0 = no log */
  int start_itno;   /* itno is a global shared by levmar() and trust_region() (PSBA/main.cpp:193-208) */
  double init_mu;   /* 0 = PSBA_INIT_MU 1e-3 (PSBA/psba.h:6, levmar.cpp:114-116); a test knob otherwise */
} orc_lm_opts;

typedef struct {
  int flag;         /* ITER_* */
  int iters;        /* outer iterations executed (itno) */
  int tries;        /* damping tries */
  double init_err;  /* ||e||^2 at entry */
  double final_err; /* ||e||^2 at exit */
  double mu0;       /* initial damping */
  int n_log;        /* rows written to log */
  double t_linearize, t_schur, t_solve, t_backsub, t_cost; /* seconds */
} orc_lm_result;

/* PSBA/levmar.cpp:45-256.  cams/pts are updated in place.  log rows are
 * (itno, new_err, rho, mu, accepted) one per completed damping try. */
/* threads the library runs its loops on: 1, or omp_get_max_threads() in the OpenMP build
 * (libpsba_oracle_omp.so, bench.py's all-core CPU baseline only) */
int orc_threads(void);

int orc_levmar(int nC, int nP, int nO, const double *K, const double *impts,
               const double *initrot, double *cams, double *pts, const int *iidx,
               const int *jidx, const orc_lm_opts *opts, orc_lm_result *res, double *log);

/* ---- the trust-region caller and its extra operators -------------------- */

/* CL_files/compute_Jmultiply.cl:6-52 (J x), stored per observation: out[2 a + k] instead of
 * the reference's dense nP x nC x 2 grid whose unobserved entries are zero; every use is a dot
 * product, which the zeros do not change and whose order of summation (point-major, cameras
 * ascending) is the observation order. */
void orc_compute_Jmultiply(int nC, int nO, const double *JA, const double *JB, const int *iidx,
                           const int *jidx, const double *x, double *out);

/* PSBA/cl_cholmod.cpp:109-168 + CL_files/cholmod_blk.cl:796-825: delta and beta of the modified
 * Cholesky from the largest |off-diagonal| and |diagonal| entries of the n x n matrix. */
void orc_get_delta_beta(int n, const double *A, double *delta, double *beta);
/* PSBA/cl_cholmod.cpp:25-101,176-201 + CL_files/cholmod_blk.cl:87-846: modified Cholesky of the
 * symmetric n x n matrix A (n a multiple of 3), block columns of three: a plain Cholesky of the
 * block column while its pivots stay positive and its entries below beta, otherwise the three
 * columns one by one with d_j = max(|c_jj|, delta), raised to (theta_j / beta)^2 when an entry
 * would exceed beta.  L (n x n, lower) and E[i] = sum_k L[i][k]^2 - A[i][i] are returned. */
void orc_cholmod(int n, const double *A, double *L, double *E);

typedef struct {
  int max_iter;     /* literal 50, shared with levmar() through itno (trust_region.cpp:112) */
  int start_itno;
  int verbose;
  int log_cap;      /* rows of 6 doubles */
  double init_lambda; /* 0 = the reference's start (PSBA/trust_region.cpp:95-96); a test knob otherwise */
} orc_tr_opts;

typedef struct {
  int flag;         /* ITER_* */
  int iters;        /* itno at exit */
  int tries;        /* steps evaluated (inner loop passes) */
  int chol_fail;    /* failed factorizations of S (each followed by a larger lambda) */
  double init_err, final_err;
  double lambda, delta; /* at exit */
  int n_log;
} orc_tr_result;

/* PSBA/trust_region.cpp:49-288 (trust_region), :292-405 (compute_PB), :520-595 (compute_p_2).
 * cams/pts are updated in place.  log rows: (itno, ||e(p + step)||^2, rho, delta after the
 * update, lambda, accepted). */
int orc_trust_region(int nC, int nP, int nO, const double *K, const double *impts,
                     const double *initrot, double *cams, double *pts, const int *iidx,
                     const int *jidx, const orc_tr_opts *opts, orc_tr_result *res, double *log);

#ifdef __cplusplus
}
#endif
#endif
