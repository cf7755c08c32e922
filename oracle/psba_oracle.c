/*
 * psba_oracle.c -- CPU restatement (plain C, fp64, single thread) of the eglrp/PSBA
 * Schur-complement bundle-adjustment normal-equations path and its Levenberg-Marquardt
 * caller.  TEST INFRASTRUCTURE ONLY -- see psba_oracle.h for the rules and the parity pin.
 *
 * Build with -O2 -ffp-contract=off (oracle/Makefile) so that sums are evaluated in the
 * written order without fused multiply-adds.
 */
#include "psba_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CNP 6
#define PNP 3
#define MNP 2

/* ORC_OMP (oracle/Makefile: libpsba_oracle_omp.so) adds OpenMP to the loops for the all-core CPU
 * baseline of bench.py.  The serial library, which is what every test and pin uses, is compiled
 * without it and is unchanged: its sums run in the reference's order.  The OpenMP build keeps
 * that order inside a point and gives up only the order of the per-camera / per-block sums. */
#ifdef ORC_OMP
#include <omp.h>
#define ORC_DO_PRAGMA(x) _Pragma(#x)
#define ORC_PAR_FOR ORC_DO_PRAGMA(omp parallel for schedule(static))
#define ORC_PAR_FOR_RED(...) ORC_DO_PRAGMA(omp parallel for schedule(static) reduction(__VA_ARGS__))
#else
#define ORC_PAR_FOR
#define ORC_PAR_FOR_RED(...)
#endif

int orc_threads(void) {
#ifdef ORC_OMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) {
    fprintf(stderr, "psba_oracle: out of memory (%zu bytes)\n", n);
    abort();
  }
  return p;
}

/* ---- camera model --------------------------------------------------------------------
 * q_l = (sqrt(1-|v|^2), v); q = q_l (x) q0 (Hamilton, local rotation on the left),
 * CL_files/compute_exQT.cl:36-49.  q = (q[0]; q[1..3]) scalar first. */
#ifdef ORC_OMP
/* first observation of every point (observations are point-major) */
static int *point_starts(int nP, int nO, const int *iidx) {
  int *ps = (int *)xmalloc(sizeof(int) * ((size_t)nP + 1));
  int a = 0;
  for (int i = 0; i <= nP; i++) {
    while (a < nO && iidx[a] < i) a++;
    ps[i] = a;
  }
  return ps;
}
#endif

static void compose_quat(const double *q0, const double *v, double *q) {
  double si = sqrt(1 - v[0] * v[0] - v[1] * v[1] - v[2] * v[2]);
  double s0 = q0[0], a1 = q0[1], a2 = q0[2], a3 = q0[3];
  q[0] = si * s0 - (a1 * v[0] + a2 * v[1] + a3 * v[2]);
  q[1] = s0 * v[0] + si * a1 + a3 * v[1] - a2 * v[2];
  q[2] = s0 * v[1] + si * a2 + a1 * v[2] - a3 * v[0];
  q[3] = s0 * v[2] + si * a3 + a2 * v[0] - a1 * v[1];
}

static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* rotate M by the quaternion sandwich q (0,M) q*, the form the reference expands
 * (CL_files/compute_exQT.cl:59-65): with u = vec(q), w = s M + u x M,
 * R M = (u.M) u + s w + u x w. */
static void quat_rotate(const double *q, const double *M, double *out) {
  const double s = q[0];
  const double *u = q + 1;
  double uxM[3], w[3], uxw[3];
  double udM = u[0] * M[0] + u[1] * M[1] + u[2] * M[2];
  cross3(u, M, uxM);
  for (int c = 0; c < 3; c++) w[c] = s * M[c] + uxM[c];
  cross3(u, w, uxw);
  for (int c = 0; c < 3; c++) out[c] = udM * u[c] + s * w[c] + uxw[c];
}

/* x = (fu Px + s Py + u0 Pz)/Pz ; y = (fu ar Py + v0 Pz)/Pz, K = (fu,u0,v0,ar,s)
 * CL_files/compute_exQT.cl:66-69. */
static void project(const double *K, const double *P, double *xy) {
  double inv = 1 / P[2];
  xy[0] = (K[0] * P[0] + K[4] * P[1] + K[1] * P[2]) * inv;
  xy[1] = (K[0] * K[3] * P[1] + K[2] * P[2]) * inv;
}

void orc_compute_exQT(int nO, const double *K, const double *impts, const double *initrot,
                      const double *cams, const double *pts, const int *iidx, const int *jidx,
                      double *ex) {
  ORC_PAR_FOR
  for (int idx = 0; idx < nO; idx++) {
    int i = iidx[idx], j = jidx[idx];
    double q[4], P[3], xy[2];
    compose_quat(initrot + 4 * j, cams + 6 * j, q);
    quat_rotate(q, pts + 3 * i, P);
    for (int c = 0; c < 3; c++) P[c] += cams[6 * j + 3 + c];
    project(K + 5 * j, P, xy);
    ex[2 * idx] = impts[2 * idx] - xy[0];
    ex[2 * idx + 1] = impts[2 * idx + 1] - xy[1];
  }
}

/* The reference's host-side twin of the residual, PSBA/levmar_func_cpu.cpp:82-140 (compute_proj_err)
 * with its helpers :147-178 (quatMultFast) and :185-223 (calcImgProjFullR).  Same mathematics as
 * orc_compute_exQT, different arithmetic: the quaternion product is the eight-multiplication
 * scheme (products of sums and differences of the components), the rotation is the explicit
 * sandwich q (0,M) q* expanded component by component, the division is a reciprocal followed
 * by products, and ONE intrinsics vector K[0..4] serves every camera (:104-113 passes the base
 * pointer).  The golden value of SURVEY.md 8(c) / Appendix A.5 for this path differs from the
 * kernel path in the 14th digit, which is what tests/test_oracle_golden.py pins. */
static void quat_mul_8(const double *a, const double *b, double *p) {
  /* levmar_func_cpu.cpp:152-177 */
  double s1 = (a[0] + a[1]) * (b[0] + b[1]);
  double s2 = (a[3] - a[2]) * (b[2] - b[3]);
  double s3 = (a[1] - a[0]) * (b[2] + b[3]);
  double s4 = (a[2] + a[3]) * (b[1] - b[0]);
  double s5 = (a[1] + a[3]) * (b[1] + b[2]);
  double s6 = (a[1] - a[3]) * (b[1] - b[2]);
  double s7 = (a[0] + a[2]) * (b[0] - b[3]);
  double s8 = (a[0] - a[2]) * (b[0] + b[3]);
  double h = 0.5 * (s5 - s6 + s7 + s8);
  p[0] = s2 + h - s5;
  p[1] = s1 - h - s6;
  p[2] = -s3 + h - s8;
  p[3] = -s4 + h - s7;
}

void orc_compute_proj_err_twin(int nO, int nC, const double *K, const double *impts,
                               const double *initrot, const double *cams, const double *pts,
                               const int *iidx, const int *jidx, double *ex) {
  (void)nC;
  for (int idx = 0; idx < nO; idx++) {
    int i = iidx[idx], j = jidx[idx];
    const double *v = cams + 6 * j, *t = v + 3, *M = pts + 3 * i;
    double ql[4], q[4];
    ql[0] = sqrt(1 - v[0] * v[0] - v[1] * v[1] - v[2] * v[2]); /* :107 */
    ql[1] = v[0];
    ql[2] = v[1];
    ql[3] = v[2];
    quat_mul_8(ql, initrot + 4 * j, q); /* :111 */
    /* q (0,M): scalar part, then vector part (:204-208) */
    double ps = -M[0] * q[1] - q[2] * M[1] - q[3] * M[2];
    double px = q[0] * M[0] + q[2] * M[2] - q[3] * M[1];
    double py = M[1] * q[0] + q[3] * M[0] - M[2] * q[1];
    double pz = q[0] * M[2] + M[1] * q[1] - q[2] * M[0];
    /* (q (0,M)) q*, vector part, plus t (:209-213) */
    double X = -q[1] * ps + q[0] * px - py * q[3] + q[2] * pz + t[0];
    double Y = -q[2] * ps + q[0] * py - pz * q[1] + q[3] * px + t[1];
    double Z = -ps * q[3] + q[0] * pz - q[2] * px + q[1] * py + t[2];
    double inv = 1 / Z;
    double x = (K[0] * X + K[4] * Y + K[1] * Z) * inv;
    double y = (K[0] * K[3] * Y + K[2] * Z) * inv;
    ex[2 * idx] = impts[2 * idx] - x;         /* :114-115 */
    ex[2 * idx + 1] = impts[2 * idx + 1] - y;
  }
}

void orc_compute_jacobiQT(int nO, const double *K, const double *initrot, const double *cams,
                          const double *pts, const int *iidx, const int *jidx, double *JA,
                          double *JB) {
  ORC_PAR_FOR
  for (int idx = 0; idx < nO; idx++) {
    int i = iidx[idx], j = jidx[idx];
    const double *Kj = K + 5 * j, *q0 = initrot + 4 * j, *v = cams + 6 * j, *M = pts + 3 * i;
    double q[4], P[3], xy[2], D[2][3];
    compose_quat(q0, v, q);
    quat_rotate(q, M, P);
    for (int c = 0; c < 3; c++) P[c] += v[3 + c];
    project(Kj, P, xy);
    double inv = 1 / P[2];
    /* D = d(x,y)/dP */
    D[0][0] = Kj[0] * inv;
    D[0][1] = Kj[4] * inv;
    D[0][2] = (Kj[1] - xy[0]) * inv;
    D[1][0] = 0.0;
    D[1][1] = Kj[0] * Kj[3] * inv;
    D[1][2] = (Kj[2] - xy[1]) * inv;

    double *A = JA + 12 * idx, *B = JB + 6 * idx;
    const double s = q[0];
    const double *u = q + 1;

    /* B = D * dP/dM ; P is linear in M, column m of dP/dM is the rotation of e_m */
    for (int m = 0; m < 3; m++) {
      double e[3] = {0, 0, 0}, col[3];
      e[m] = 1.0;
      quat_rotate(q, e, col);
      B[m] = D[0][0] * col[0] + D[0][1] * col[1] + D[0][2] * col[2];
      B[3 + m] = D[1][0] * col[0] + D[1][1] * col[1] + D[1][2] * col[2];
    }

    /* A[:,0..2] = D * dP/dv_k.  dq_l/dv_k = (-v_k/s_l, e_k); dq = dq_l (x) q0. */
    double si = sqrt(1 - v[0] * v[0] - v[1] * v[1] - v[2] * v[2]);
    double uxM[3], w[3];
    double udM = u[0] * M[0] + u[1] * M[1] + u[2] * M[2];
    cross3(u, M, uxM);
    for (int c = 0; c < 3; c++) w[c] = s * M[c] + uxM[c];
    for (int k = 0; k < 3; k++) {
      double dsl = -v[k] / si;
      double e[3] = {0, 0, 0}, exv0[3], du[3], duxM[3], dw[3], duxw[3], uxdw[3], dP[3];
      e[k] = 1.0;
      cross3(e, q0 + 1, exv0);
      double ds = dsl * q0[0] - q0[1 + k];
      for (int c = 0; c < 3; c++) du[c] = q0[0] * e[c] + dsl * q0[1 + c] + exv0[c];
      cross3(du, M, duxM);
      for (int c = 0; c < 3; c++) dw[c] = ds * M[c] + duxM[c];
      cross3(du, w, duxw);
      cross3(u, dw, uxdw);
      double dudM = du[0] * M[0] + du[1] * M[1] + du[2] * M[2];
      for (int c = 0; c < 3; c++)
        dP[c] = dudM * u[c] + udM * du[c] + ds * w[c] + s * dw[c] + duxw[c] + uxdw[c];
      A[k] = D[0][0] * dP[0] + D[0][1] * dP[1] + D[0][2] * dP[2];
      A[6 + k] = D[1][0] * dP[0] + D[1][1] * dP[1] + D[1][2] * dP[2];
    }
    /* A[:,3..5] = D (translation) */
    for (int c = 0; c < 3; c++) {
      A[3 + c] = D[0][c];
      A[9 + c] = D[1][c];
    }
  }
}

void orc_compute_U(int nC, int nO, const double *JA, const int *jidx, double coeff, double *U,
                   double *UVdiag) {
  memset(U, 0, sizeof(double) * 36 * (size_t)nC);
  /* observations are visited in point-ascending order, so each U_j(r,c) is summed over i
   * ascending exactly as compute_U.cl:22-29 does */
  ORC_PAR_FOR_RED(+ : U[0 : 36 * nC])
  for (int idx = 0; idx < nO; idx++) {
    const double *A = JA + 12 * idx;
    double *Uj = U + 36 * jidx[idx];
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) Uj[6 * r + c] = Uj[6 * r + c] + A[r] * A[c] + A[6 + r] * A[6 + c];
  }
  for (int j = 0; j < nC; j++)
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        double sum = coeff * U[36 * j + 6 * r + c];
        U[36 * j + 6 * r + c] = sum;
        if (r == c) UVdiag[6 * j + r] = sum;
      }
}

void orc_compute_V(int nC, int nP, int nO, const double *JB, const int *iidx, double coeff,
                   double *V, double *UVdiag) {
  memset(V, 0, sizeof(double) * 9 * (size_t)nP);
#ifdef ORC_OMP
  int *ps = point_starts(nP, nO, iidx);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nP; i++)
    for (int idx = ps[i]; idx < ps[i + 1]; idx++) {
      const double *B = JB + 6 * idx;
      double *Vi = V + 9 * i;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Vi[3 * r + c] = Vi[3 * r + c] + B[r] * B[c] + B[3 + r] * B[3 + c];
    }
  free(ps);
#else
  for (int idx = 0; idx < nO; idx++) {
    const double *B = JB + 6 * idx;
    double *Vi = V + 9 * iidx[idx];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Vi[3 * r + c] = Vi[3 * r + c] + B[r] * B[c] + B[3 + r] * B[3 + c];
  }
#endif
  for (int i = 0; i < nP; i++)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double sum = coeff * V[9 * i + 3 * r + c];
        V[9 * i + 3 * r + c] = sum;
        if (r == c) UVdiag[6 * nC + 3 * i + r] = sum;
      }
}

void orc_compute_Wblks(int nO, const double *JA, const double *JB, double coeff, double *W) {
  ORC_PAR_FOR
  for (int idx = 0; idx < nO; idx++) {
    const double *A = JA + 12 * idx, *B = JB + 6 * idx;
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 3; c++) W[18 * idx + 3 * r + c] = coeff * (A[r] * B[c] + A[6 + r] * B[3 + c]);
  }
}

void orc_compute_g(int nC, int nP, int nO, double coeff, const double *JA, const double *JB,
                   const int *iidx, const int *jidx, const double *ex, double *g) {
  int nA = 6 * nC, nT = 6 * nC + 3 * nP;
  memset(g, 0, sizeof(double) * (size_t)nT);
#ifdef ORC_OMP
#pragma omp parallel for schedule(static) reduction(+ : g[0 : nA])
  for (int idx = 0; idx < nO; idx++) {
    const double *A = JA + 12 * idx;
    double e0 = ex[2 * idx], e1 = ex[2 * idx + 1];
    double *ga = g + 6 * jidx[idx];
    for (int k = 0; k < 6; k++) ga[k] = ga[k] + A[k] * e0 + A[6 + k] * e1;
  }
  int *ps = point_starts(nP, nO, iidx);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nP; i++)
    for (int idx = ps[i]; idx < ps[i + 1]; idx++) {
      const double *B = JB + 6 * idx;
      double e0 = ex[2 * idx], e1 = ex[2 * idx + 1];
      double *gb = g + nA + 3 * i;
      for (int k = 0; k < 3; k++) gb[k] = gb[k] + B[k] * e0 + B[3 + k] * e1;
    }
  free(ps);
#else
  for (int idx = 0; idx < nO; idx++) {
    const double *A = JA + 12 * idx, *B = JB + 6 * idx;
    double e0 = ex[2 * idx], e1 = ex[2 * idx + 1];
    double *ga = g + 6 * jidx[idx], *gb = g + nA + 3 * iidx[idx];
    for (int k = 0; k < 6; k++) ga[k] = ga[k] + A[k] * e0 + A[6 + k] * e1;
    for (int k = 0; k < 3; k++) gb[k] = gb[k] + B[k] * e0 + B[3 + k] * e1;
  }
#endif
  for (int t = 0; t < nT; t++) g[t] = coeff * g[t];
}

double orc_maxElmOfUV(int nT, const double *UVdiag) {
  double m = UVdiag[0];
  for (int t = 1; t < nT; t++)
    if (UVdiag[t] > m) m = UVdiag[t];
  return m;
}

void orc_update_UV(int nC, int nP, double *U, double *V, double mu) {
  for (int j = 0; j < nC; j++)
    for (int r = 0; r < 6; r++) U[36 * j + 7 * r] = U[36 * j + 7 * r] + mu;
  for (int i = 0; i < nP; i++)
    for (int r = 0; r < 3; r++) V[9 * i + 4 * r] = V[9 * i + 4 * r] + mu;
}

void orc_restore_UVdiag(int nC, int nP, double *U, double *V, const double *UVdiag) {
  for (int j = 0; j < nC; j++)
    for (int r = 0; r < 6; r++) U[36 * j + 7 * r] = UVdiag[6 * j + r];
  for (int i = 0; i < nP; i++)
    for (int r = 0; r < 3; r++) V[9 * i + 4 * r] = UVdiag[6 * nC + 3 * i + r];
}

double orc_compute_Vinv(int nP, const double *V, double *Vinv) {
  double ret = 0.0;
  ORC_PAR_FOR_RED(max : ret)
  for (int i = 0; i < nP; i++) {
    const double *a = V + 9 * i;
    double a11 = a[0], a12 = a[1], a13 = a[2], a22 = a[4], a23 = a[5], a33 = a[8];
    /* T = -det, compute_Vinv.cl:29 */
    double T = (a33 * a12 * a12 - 2 * a12 * a13 * a23 + a22 * a13 * a13 + a11 * a23 * a23 -
                a11 * a22 * a33);
    if (fabs(T) < 1e-16) ret = 1.0;
    double *o = Vinv + 9 * i;
    o[0] = -(-a23 * a23 + a22 * a33) / T;
    o[3] = -(a13 * a23 - a12 * a33) / T;
    o[4] = -(-a13 * a13 + a11 * a33) / T;
    o[6] = -(a12 * a23 - a13 * a22) / T;
    o[7] = -(a12 * a13 - a11 * a23) / T;
    o[8] = -(-a12 * a12 + a11 * a22) / T;
    o[1] = o[3];
    o[2] = o[6];
    o[5] = o[7];
  }
  return ret;
}

void orc_compute_Yblks(int nO, const int *iidx, const double *W, const double *Vinv, double *Y) {
  ORC_PAR_FOR
  for (int idx = 0; idx < nO; idx++) {
    const double *Vi = Vinv + 9 * iidx[idx];
    for (int r = 0; r < 6; r++) {
      const double *w = W + 18 * idx + 3 * r;
      for (int c = 0; c < 3; c++) Y[18 * idx + 3 * r + c] = w[0] * Vi[c] + w[1] * Vi[3 + c] + w[2] * Vi[6 + c];
    }
  }
}

void orc_compute_S(int nC, int nP, int nO, const int *iidx, const int *jidx, const double *U,
                   const double *Y, const double *W, double *S) {
  int nA = 6 * nC;
  (void)nP;
  memset(S, 0, sizeof(double) * (size_t)nA * nA);
#ifdef ORC_OMP
  {
    int *ps = point_starts(nP, nO, iidx);
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : S[0 : nA * nA])
    for (int i = 0; i < nP; i++)
      for (int a = ps[i]; a < ps[i + 1]; a++)
        for (int b = ps[i]; b < ps[i + 1]; b++) {
          int k = jidx[a], l = jidx[b];
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) {
              const double *y = Y + 18 * a + 3 * r, *w = W + 18 * b + 3 * c;
              S[(size_t)(6 * k + r) * nA + 6 * l + c] += y[0] * w[0] + y[1] * w[1] + y[2] * w[2];
            }
        }
    free(ps);
  }
  int a0 = nO;
#else
  int a0 = 0;
#endif
  while (a0 < nO) {
    int a1 = a0;
    while (a1 < nO && iidx[a1] == iidx[a0]) a1++;
    for (int a = a0; a < a1; a++)
      for (int b = a0; b < a1; b++) {
        int k = jidx[a], l = jidx[b];
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) {
            const double *y = Y + 18 * a + 3 * r, *w = W + 18 * b + 3 * c;
            S[(size_t)(6 * k + r) * nA + 6 * l + c] += y[0] * w[0] + y[1] * w[1] + y[2] * w[2];
          }
      }
    a0 = a1;
  }
  for (int k = 0; k < nC; k++)
    for (int l = 0; l < nC; l++)
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
          size_t at = (size_t)(6 * k + r) * nA + 6 * l + c;
          S[at] = (k == l) ? U[36 * k + 6 * r + c] - S[at] : -S[at];
        }
}

void orc_compute_ea(int nC, int nP, int nO, const int *iidx, const int *jidx, const double *Y,
                    const double *g, double *eab) {
  int nA = 6 * nC;
  (void)nP;
  double *sum = (double *)xmalloc(sizeof(double) * (size_t)nA);
  memset(sum, 0, sizeof(double) * (size_t)nA);
  ORC_PAR_FOR_RED(+ : sum[0 : nA])
  for (int idx = 0; idx < nO; idx++) {
    const double *gb = g + nA + 3 * iidx[idx];
    for (int r = 0; r < 6; r++) {
      const double *y = Y + 18 * idx + 3 * r;
      sum[6 * jidx[idx] + r] = sum[6 * jidx[idx] + r] + y[0] * gb[0] + y[1] * gb[1] + y[2] * gb[2];
    }
  }
  for (int t = 0; t < nA; t++) eab[t] = g[t] - sum[t];
  free(sum);
}

double orc_chol_solve(int n, double *S, const double *ea, double *dpa) {
  /* in-place lower Cholesky, row-major */
  for (int j = 0; j < n; j++) {
    double d = S[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= S[(size_t)j * n + k] * S[(size_t)j * n + k];
    double l = sqrt(d);
    if (!(d > 0.0) || !isfinite(l)) return 1.0;
    S[(size_t)j * n + j] = l;
    int bad = 0;
    ORC_PAR_FOR_RED(| : bad)
    for (int i = j + 1; i < n; i++) {
      double t = S[(size_t)i * n + j];
      for (int k = 0; k < j; k++) t -= S[(size_t)i * n + k] * S[(size_t)j * n + k];
      t = t / l;
      if (!isfinite(t)) bad |= 1;
      S[(size_t)i * n + j] = t;
    }
    if (bad) return 1.0;
  }
  for (int i = 0; i < n; i++) {
    double t = ea[i];
    for (int k = 0; k < i; k++) t -= S[(size_t)i * n + k] * dpa[k];
    dpa[i] = t / S[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = dpa[i];
    for (int k = i + 1; k < n; k++) t -= S[(size_t)k * n + i] * dpa[k];
    dpa[i] = t / S[(size_t)i * n + i];
  }
  for (int i = 0; i < n; i++)
    if (!isfinite(dpa[i])) return 1.0;
  return 0.0;
}

void orc_compute_eb(int nC, int nP, int nO, const int *iidx, const int *jidx, const double *W,
                    const double *dp, const double *g, double *eab) {
  int nA = 6 * nC;
  double *sum = (double *)xmalloc(sizeof(double) * 3 * (size_t)nP);
  memset(sum, 0, sizeof(double) * 3 * (size_t)nP);
#ifdef ORC_OMP
  int *ps = point_starts(nP, nO, iidx);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nP; i++)
    for (int idx = ps[i]; idx < ps[i + 1]; idx++) {
      const double *w = W + 18 * idx, *dpa = dp + 6 * jidx[idx];
      double *s = sum + 3 * i;
      for (int c = 0; c < 3; c++)
        for (int k = 0; k < 6; k++) s[c] = s[c] + w[3 * k + c] * dpa[k];
    }
  free(ps);
#else
  for (int idx = 0; idx < nO; idx++) {
    const double *w = W + 18 * idx, *dpa = dp + 6 * jidx[idx];
    double *s = sum + 3 * iidx[idx];
    for (int c = 0; c < 3; c++)
      for (int k = 0; k < 6; k++) s[c] = s[c] + w[3 * k + c] * dpa[k];
  }
#endif
  for (int t = 0; t < 3 * nP; t++) eab[nA + t] = g[nA + t] - sum[t];
  free(sum);
}

void orc_compute_dpb(int nC, int nP, const double *Vinv, const double *eab, double *dp) {
  int nA = 6 * nC;
  ORC_PAR_FOR
  for (int i = 0; i < nP; i++) {
    const double *Vi = Vinv + 9 * i, *eb = eab + nA + 3 * i;
    for (int r = 0; r < 3; r++) dp[nA + 3 * i + r] = Vi[3 * r] * eb[0] + Vi[3 * r + 1] * eb[1] + Vi[3 * r + 2] * eb[2];
  }
}

void orc_compute_newp(int nA, int nB, const double *cams, const double *pts, const double *dp,
                      double *newcams, double *newpts) {
  for (int t = 0; t < nA; t++) newcams[t] = cams[t] + dp[t];
  ORC_PAR_FOR
  for (int t = 0; t < nB; t++) newpts[t] = pts[t] + dp[nA + t];
}

double orc_L2_sq(int n, const double *x) {
  double sum = 0;
  ORC_PAR_FOR_RED(+ : sum)
  for (int i = 0; i < n; i++) sum += x[i] * x[i];
  return sum;
}

int orc_levmar(int nC, int nP, int nO, const double *K, const double *impts,
               const double *initrot, double *cams, double *pts, const int *iidx,
               const int *jidx, const orc_lm_opts *opts, orc_lm_result *res, double *log) {
  const int nA = 6 * nC, nB = 3 * nP, nT = nA + nB;
  const double STOP = 1e-12, EPS_SQ = 1e-12 * 1e-12; /* PSBA/psba.h:7-10 */
  double *ex = xmalloc(sizeof(double) * 2 * (size_t)nO);
  double *JA = xmalloc(sizeof(double) * 12 * (size_t)nO);
  double *JB = xmalloc(sizeof(double) * 6 * (size_t)nO);
  double *W = xmalloc(sizeof(double) * 18 * (size_t)nO);
  double *Y = xmalloc(sizeof(double) * 18 * (size_t)nO);
  double *U = xmalloc(sizeof(double) * 36 * (size_t)nC);
  double *V = xmalloc(sizeof(double) * 9 * (size_t)nP);
  double *Vinv = xmalloc(sizeof(double) * 9 * (size_t)nP);
  double *UVdiag = xmalloc(sizeof(double) * (size_t)nT);
  double *S = xmalloc(sizeof(double) * (size_t)nA * nA);
  double *g = xmalloc(sizeof(double) * (size_t)nT);
  double *eab = xmalloc(sizeof(double) * (size_t)nT);
  double *dp = xmalloc(sizeof(double) * (size_t)nT);
  double *newcams = xmalloc(sizeof(double) * (size_t)nA);
  double *newpts = xmalloc(sizeof(double) * (size_t)nB);

  memset(res, 0, sizeof(*res));
  double mu = 0, rho, p_L2 = 0, dp_L2, ex_L2, new_ex_L2;
  int nu = 2, first = 1, gooditer = 0, tries = 0, itno = opts->start_itno, nlog = 0;
  int flag = ORC_ITER_CONTINUE;
  double t0;

  t0 = now_s();
  orc_compute_exQT(nO, K, impts, initrot, cams, pts, iidx, jidx, ex); /* levmar.cpp:93-95 */
  ex_L2 = orc_L2_sq(2 * nO, ex);
  res->t_cost += now_s() - t0;
  res->init_err = ex_L2;

  for (; itno < opts->max_iter && flag == ORC_ITER_CONTINUE; itno++) { /* levmar.cpp:100 */
    t0 = now_s();
    orc_compute_jacobiQT(nO, K, initrot, cams, pts, iidx, jidx, JA, JB); /* :103 */
    orc_compute_U(nC, nO, JA, jidx, 1.0, U, UVdiag);                     /* :104 */
    orc_compute_V(nC, nP, nO, JB, iidx, 1.0, V, UVdiag);                 /* :105 */
    orc_compute_Wblks(nO, JA, JB, 1.0, W);                               /* :106 */
    /* ex still holds the residual at the current parameters: the try loop is only left
     * with an accepted step (whose residual was the last one evaluated) or by stopping. */
    orc_compute_g(nC, nP, nO, 1.0, JA, JB, iidx, jidx, ex, g); /* :108 */
    res->t_linearize += now_s() - t0;

    if (first) { /* :114-120 */
      mu = (opts->init_mu != 0.0 ? opts->init_mu : 1e-3) * orc_maxElmOfUV(nT, UVdiag);
      res->mu0 = mu;
      first = 0;
      p_L2 = 1e+3;
      nu = 2;
    }
    while (1) {
      tries++;
      t0 = now_s();
      orc_update_UV(nC, nP, U, V, mu);                          /* :126 */
      orc_compute_Vinv(nP, V, Vinv);                            /* :127 (return ignored) */
      orc_compute_Yblks(nO, iidx, W, Vinv, Y);                  /* :128 */
      orc_compute_S(nC, nP, nO, iidx, jidx, U, Y, W, S);        /* :130 */
      orc_compute_ea(nC, nP, nO, iidx, jidx, Y, g, eab);        /* :131 */
      res->t_schur += now_s() - t0;
      t0 = now_s();
      double ret = orc_chol_solve(nA, S, eab, dp);              /* :134-140 */
      res->t_solve += now_s() - t0;
      if (ret == 0.0) {
        t0 = now_s();
        orc_compute_eb(nC, nP, nO, iidx, jidx, W, dp, g, eab);  /* :151 */
        orc_compute_dpb(nC, nP, Vinv, eab, dp);                 /* :155 */
        dp_L2 = orc_L2_sq(nT, dp);                              /* :157 */
        res->t_backsub += now_s() - t0;
        if (dp_L2 < p_L2 * STOP * STOP) { /* :169-173 */
          flag = ORC_ITER_DP_NO_CHANGE;
          break;
        }
        if (dp_L2 >= (p_L2 + STOP) / EPS_SQ) { /* :174-179 */
          flag = ORC_ITER_ERR;
          break;
        }
        orc_restore_UVdiag(nC, nP, U, V, UVdiag);               /* :182 */
        orc_compute_newp(nA, nB, cams, pts, dp, newcams, newpts); /* :185 */
        t0 = now_s();
        orc_compute_exQT(nO, K, impts, initrot, newcams, newpts, iidx, jidx, ex); /* :188 */
        new_ex_L2 = orc_L2_sq(2 * nO, ex);                      /* :193 */
        res->t_cost += now_s() - t0;
        double sum = 0; /* compute_rho, :271-280 */
        for (int t = 0; t < nT; t++) sum += dp[t] * (mu * dp[t] + g[t]);
        rho = (ex_L2 - new_ex_L2) / sum;
        if (opts->verbose)
          printf("itno=%d\t\tErr=%.15E\t\trho=%f\t\tmu=%f\n", itno, new_ex_L2, rho, mu);
        if (log && nlog < opts->log_cap) {
          double *row = log + 5 * nlog++;
          row[0] = itno; row[1] = new_ex_L2; row[2] = rho; row[3] = mu; row[4] = rho > 0;
        }
        if (rho > 0) { /* :200-223 */
          double tmp = 2 * rho - 1;
          tmp = 1.0 - tmp * tmp * tmp;
          mu = mu * ((tmp >= (1.0 / 3.0)) ? tmp : (1.0 / 3.0));
          nu = 2;
          memcpy(cams, newcams, sizeof(double) * (size_t)nA); /* update_p :211 */
          memcpy(pts, newpts, sizeof(double) * (size_t)nB);
          p_L2 = 0; /* :212, one running sum over cams then points */
          for (int t = 0; t < nA; t++) p_L2 += cams[t] * cams[t];
          for (int t = 0; t < nB; t++) p_L2 += pts[t] * pts[t];
          ex_L2 = new_ex_L2;
          if (fabs(rho - 1) < (1.0 / 5.0)) {
            gooditer++;
            if (opts->tr_handoff && gooditer >= 5) {
              flag = ORC_ITER_TURN_TO_TR;
              break;
            }
          } else
            gooditer = 0;
          break;
        }
      } else { /* :227-233 */
        gooditer = 0;
        orc_restore_UVdiag(nC, nP, U, V, UVdiag);
        if (log && nlog < opts->log_cap) {
          double *row = log + 5 * nlog++;
          row[0] = itno; row[1] = NAN; row[2] = NAN; row[3] = mu; row[4] = -1;
        }
      }
      mu *= nu; /* :237-244 */
      double nu2 = 2.0 * nu;
      if (nu2 <= nu || nu2 > 1e9) {
        flag = ORC_ITER_ERR;
        break;
      }
      nu = (int)nu2;
    }
    if (ex_L2 <= STOP) flag = ORC_ITER_ERR_SMALL_ENOUGH; /* :247-248 */
  }

  res->flag = flag;
  res->iters = itno;
  res->tries = tries;
  res->final_err = ex_L2;
  res->n_log = nlog;
  free(ex); free(JA); free(JB); free(W); free(Y); free(U); free(V); free(Vinv); free(UVdiag);
  free(S); free(g); free(eab); free(dp); free(newcams); free(newpts);
  return flag;
}


/* ======================================================================================== */
/* trust-region caller                                                         */

void orc_compute_Jmultiply(int nC, int nO, const double *JA, const double *JB, const int *iidx,
                           const int *jidx, const double *x, double *out) {
  const int nA = 6 * nC;
  ORC_PAR_FOR
  for (int a = 0; a < nO; a++) {
    const double *xa = x + 6 * jidx[a], *xb = x + nA + 3 * iidx[a];
    for (int k = 0; k < 2; k++) { /* compute_Jmultiply.cl:32-46: row k of A_ij, then of B_ij */
      const double *A = JA + 12 * (size_t)a + 6 * k, *B = JB + 6 * (size_t)a + 3 * k;
      double sum = 0;
      for (int c = 0; c < 6; c++) sum += A[c] * xa[c];
      for (int c = 0; c < 3; c++) sum += B[c] * xb[c];
      out[2 * (size_t)a + k] = sum;
    }
  }
}

static double dot_n(int n, const double *a, const double *b) { /* PSBA/misc.cpp dotProduct */
  double s = 0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

void orc_get_delta_beta(int n, const double *A, double *delta, double *beta) {
  double xi = 0, gamma = 0; /* largest |off-diagonal|, largest |diagonal| */
  for (int r = 0; r < n; r++) {
    for (int c = 0; c < n; c++) {
      double v = fabs(A[(size_t)r * n + c]);
      if (r == c) {
        if (v > gamma) gamma = v;
      } else if (v > xi)
        xi = v;
    }
  }
  *delta = 1e-15 * fmax(xi + gamma, 1); /* cl_cholmod.cpp:163 */
  double b = fmax(gamma, 1e-15);
  b = fmax(b, xi / sqrt((double)n * n - 1));
  *beta = sqrt(b);
}

/* one column of the modified factorization (cholmod_blk.cl:560-706: step1..step4) */
static void cholmod_single_column(int n, double *M, int j, double beta, double delta, double *C) {
  double d = M[(size_t)j * n + j];
  for (int k = 0; k < j; k++) d -= M[(size_t)j * n + k] * M[(size_t)j * n + k];
  d = fmax(fabs(d), delta);
  double ljj = sqrt(d);
  M[(size_t)j * n + j] = ljj;
  int over = 0;
  for (int i = j + 1; i < n; i++) {
    double c = M[(size_t)i * n + j];
    for (int k = 0; k < j; k++) c -= M[(size_t)i * n + k] * M[(size_t)j * n + k];
    C[i] = c;
    M[(size_t)i * n + j] = c / ljj;
    M[(size_t)j * n + i] = 0;
    if (M[(size_t)i * n + j] > beta) over = 1; /* :641 compares without fabs */
  }
  if (over) {
    double theta = 0;
    for (int i = j + 1; i < n; i++) theta = fmax(theta, fabs(C[i]));
    ljj = theta / beta; /* :673-674 */
    M[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; i++) M[(size_t)i * n + j] = C[i] / ljj;
  }
}

void orc_cholmod(int n, const double *A, double *L, double *E) {
  double delta, beta;
  orc_get_delta_beta(n, A, &delta, &beta);
  memcpy(L, A, sizeof(double) * (size_t)n * n);
  double *C = (double *)xmalloc(sizeof(double) * (size_t)n);
  double *bak = (double *)xmalloc(sizeof(double) * 3 * (size_t)n);
  for (int J = 0; J + 3 <= n; J += 3) {
    /* block column J as a plain 3-column Cholesky step (cholmod_blk.cl:87-268, :276-420) */
    for (int r = J; r < n; r++)
      for (int c = 0; c < 3; c++) bak[3 * (size_t)r + c] = L[(size_t)r * n + J + c];
    int fail = 0;
    double T[3][3];
    for (int u = 0; u < 3; u++)
      for (int v = 0; v < 3; v++) {
        double t = L[(size_t)(J + u) * n + J + v];
        for (int k = 0; k < J; k++) t -= L[(size_t)(J + u) * n + k] * L[(size_t)(J + v) * n + k];
        T[u][v] = t;
      }
    double l00 = T[0][0], l10 = 0, l11 = 0, l20 = 0, l21 = 0, l22 = 0;
    if (!isfinite(l00) || l00 <= 0) fail = 1;
    if (!fail) {
      l00 = sqrt(l00);
      l10 = T[1][0] / l00;
      l20 = T[2][0] / l00;
      l11 = T[1][1] - l10 * l10;
      if (!isfinite(l11) || l11 <= 0) fail = 1;
    }
    if (!fail) {
      l11 = sqrt(l11);
      l21 = (T[2][1] - l20 * l10) / l11;
      l22 = T[2][2] - l20 * l20 - l21 * l21;
      if (!isfinite(l22) || l22 <= 0 || !isfinite(l10) || !isfinite(l20) || !isfinite(l21)) fail = 1;
    }
    if (!fail) {
      l22 = sqrt(l22);
      /* L_iJ = T_iJ L_JJ^-T for the rows below; an entry above beta sends the block column to the
       * one-column route (:352-354, compared without fabs) */
      L[(size_t)J * n + J] = l00;
      L[(size_t)(J + 1) * n + J] = l10; L[(size_t)(J + 1) * n + J + 1] = l11;
      L[(size_t)(J + 2) * n + J] = l20; L[(size_t)(J + 2) * n + J + 1] = l21; L[(size_t)(J + 2) * n + J + 2] = l22;
      L[(size_t)J * n + J + 1] = L[(size_t)J * n + J + 2] = L[(size_t)(J + 1) * n + J + 2] = 0;
      for (int i = J + 3; i < n && !fail; i++) {
        double t[3];
        for (int v = 0; v < 3; v++) {
          double x = L[(size_t)i * n + J + v];
          for (int k = 0; k < J; k++) x -= L[(size_t)i * n + k] * L[(size_t)(J + v) * n + k];
          t[v] = x;
        }
        double x0 = t[0] / l00;
        double x1 = (t[1] - x0 * l10) / l11;
        double x2 = (t[2] - x0 * l20 - x1 * l21) / l22;
        L[(size_t)i * n + J] = x0; L[(size_t)i * n + J + 1] = x1; L[(size_t)i * n + J + 2] = x2;
        for (int v = 0; v < 3; v++) L[(size_t)(J + v) * n + i] = 0;
        if (x0 > beta || x1 > beta || x2 > beta) fail = 1;
      }
    }
    if (fail) { /* restore the block column and take its columns one at a time (:240-262, :436-462) */
      for (int r = J; r < n; r++)
        for (int c = 0; c < 3; c++) L[(size_t)r * n + J + c] = bak[3 * (size_t)r + c];
      for (int c = 0; c < 3; c++) cholmod_single_column(n, L, J + c, beta, delta, C);
    }
  }
  for (int i = 0; i < n; i++) { /* kern_cholmod_E, :830-846 */
    double s = 0;
    for (int k = 0; k <= i; k++) s += L[(size_t)i * n + k] * L[(size_t)i * n + k];
    E[i] = s - A[(size_t)i * n + i];
  }
  free(C);
  free(bak);
}

/* the step inside the trust region: the minimiser of the model over span{P_U, P_B} when it is
 * short enough, else the dog-leg path (trust_region.cpp:520-595, compute_p_2) */
static double tr_step(int n, double uBu, double uBb, double bBb, double delta, const double *PU,
                      const double *PB, double *p, const double *g) {
  const double ug = dot_n(n, PU, g), bg = dot_n(n, PB, g);
  const double det = -uBb * uBb + bBb * uBu;
  const double eta1 = (bg * uBb) / det - (bBb * ug) / det;
  const double eta2 = (ug * uBb) / det - (bg * uBu) / det;
  double nrm = 0;
  for (int i = 0; i < n; i++) {
    p[i] = eta1 * PU[i] + eta2 * PB[i];
    nrm += p[i] * p[i];
  }
  nrm = sqrt(nrm);
  if (!(nrm > delta)) return nrm;
  double nu = 0, nb = 0;
  for (int i = 0; i < n; i++) {
    nu += PU[i] * PU[i];
    nb += PB[i] * PB[i];
  }
  nu = sqrt(nu);
  nb = sqrt(nb);
  if (nu > delta) {
    for (int i = 0; i < n; i++) p[i] = delta * PU[i] / nu;
    return delta;
  }
  if (nb <= delta) { /* (the reference returns sqrt(nrm + nb^2) here, a value only printed) */
    for (int i = 0; i < n; i++) p[i] = PB[i];
    return nb;
  }
  double a = 0, b = 0, c = 0;
  for (int i = 0; i < n; i++) {
    const double Ai = PB[i] - PU[i], Bi = 2 * PU[i] - PB[i];
    a += Ai * Ai;
    b += Ai * Bi;
    c += Bi * Bi;
  }
  b = 2 * b;
  c = c - delta * delta;
  double disc = b * b - 4 * a * c;
  if (fabs(disc) < 1e-12) disc = 0;
  const double tau = (-b + sqrt(disc)) / (2 * a);
  for (int i = 0; i < n; i++) p[i] = PU[i] + (tau - 1) * (PB[i] - PU[i]);
  return delta;
}

int orc_trust_region(int nC, int nP, int nO, const double *K, const double *impts,
                     const double *initrot, double *cams, double *pts, const int *iidx,
                     const int *jidx, const orc_tr_opts *opts, orc_tr_result *res, double *log) {
  const int nA = 6 * nC, nB = 3 * nP, nT = nA + nB;
  const double EPS2 = 1e-12, MAX_DELTA = 10000; /* psba.h:9, trust_region.cpp:18 */
  double *ex = xmalloc(sizeof(double) * 2 * (size_t)nO);
  double *JA = xmalloc(sizeof(double) * 12 * (size_t)nO);
  double *JB = xmalloc(sizeof(double) * 6 * (size_t)nO);
  double *W = xmalloc(sizeof(double) * 18 * (size_t)nO);
  double *Y = xmalloc(sizeof(double) * 18 * (size_t)nO);
  double *U = xmalloc(sizeof(double) * 36 * (size_t)nC);
  double *V = xmalloc(sizeof(double) * 9 * (size_t)nP);
  double *Vinv = xmalloc(sizeof(double) * 9 * (size_t)nP);
  double *UVdiag = xmalloc(sizeof(double) * (size_t)nT);
  double *S = xmalloc(sizeof(double) * (size_t)nA * nA);
  double *Sf = xmalloc(sizeof(double) * (size_t)nA * nA);
  double *g = xmalloc(sizeof(double) * (size_t)nT);
  double *eab = xmalloc(sizeof(double) * (size_t)nT);
  double *PU = xmalloc(sizeof(double) * (size_t)nT);
  double *PB = xmalloc(sizeof(double) * (size_t)nT);
  double *P = xmalloc(sizeof(double) * (size_t)nT);
  double *Jx1 = xmalloc(sizeof(double) * 2 * (size_t)nO);
  double *Jx2 = xmalloc(sizeof(double) * 2 * (size_t)nO);
  double *newcams = xmalloc(sizeof(double) * (size_t)nA);
  double *newpts = xmalloc(sizeof(double) * (size_t)nB);
  double *exn = xmalloc(sizeof(double) * 2 * (size_t)nO);
  memset(res, 0, sizeof(*res));
  double dk = 1, lambda = opts->init_lambda, origin_lambda = 0; /* :95-96 */
  int nu = 2, notgood = 0, good_iters = 0, itno = opts->start_itno, nlog = 0, tries = 0;
  int flag = ORC_ITER_CONTINUE;

  orc_compute_exQT(nO, K, impts, initrot, cams, pts, iidx, jidx, ex); /* :106-107 */
  double ex_L2 = orc_L2_sq(2 * nO, ex);
  res->init_err = ex_L2;
  double final_err = ex_L2;

  for (; itno < opts->max_iter; itno++) { /* :112 */
    orc_compute_jacobiQT(nO, K, initrot, cams, pts, iidx, jidx, JA, JB); /* :117 */
    orc_compute_g(nC, nP, nO, -2.0, JA, JB, iidx, jidx, ex, g);          /* :122, g = grad ||e||^2 */
    orc_compute_Jmultiply(nC, nO, JA, JB, iidx, jidx, g, Jx1);           /* :125 */
    const double gtBg = 2 * dot_n(2 * nO, Jx1, Jx1), gtg = dot_n(nT, g, g);
    for (int i = 0; i < nT; i++) PU[i] = -(g[i] * gtg) / gtBg;           /* :128-130 Cauchy step */
    orc_compute_U(nC, nO, JA, jidx, 2.0, U, UVdiag);                     /* :133-137 */
    orc_compute_V(nC, nP, nO, JB, iidx, 2.0, V, UVdiag);
    orc_compute_Wblks(nO, JA, JB, 2.0, W);
    int solved = 0;
    while (!solved) { /* :141-163 around compute_PB (:292-405) */
      orc_update_UV(nC, nP, U, V, lambda);
      orc_compute_Vinv(nP, V, Vinv);
      orc_compute_Yblks(nO, iidx, W, Vinv, Y);
      orc_compute_S(nC, nP, nO, iidx, jidx, U, Y, W, S);
      memcpy(Sf, S, sizeof(double) * (size_t)nA * nA); /* :332-334 backup, the factorization overwrites */
      orc_compute_ea(nC, nP, nO, iidx, jidx, Y, g, eab);
      double ret = orc_chol_solve(nA, Sf, eab, P);
      if (ret != 0.0) {
        res->chol_fail++;
        if (lambda == 0.0) { /* :341-363: lambda from the diagonal correction of a modified Cholesky */
          double *E = xmalloc(sizeof(double) * (size_t)nA);
          orc_cholmod(nA, S, Sf, E);
          double sum = 0;
          for (int i = 0; i < nA; i++) sum += E[i];
          lambda = fabs(sum) / nA;
          free(E);
        } else {
          lambda = 2 * lambda; /* :365-368 */
        }
        if (origin_lambda != 0.0) { /* :144-155 */
          if (nu > 4) {
            flag = ORC_ITER_TURN_TO_LM;
            final_err = ex_L2;
            goto done;
          }
          lambda = lambda * nu;
          nu = nu * 2;
        }
        orc_restore_UVdiag(nC, nP, U, V, UVdiag);
        if (!isfinite(lambda) || res->chol_fail > 200) { /* (guard: the reference would spin) */
          flag = ORC_ITER_ERR;
          goto done;
        }
        continue;
      }
      orc_compute_eb(nC, nP, nO, iidx, jidx, W, P, g, eab); /* :385-391 */
      orc_compute_dpb(nC, nP, Vinv, eab, P);
      for (int i = 0; i < nT; i++) PB[i] = -P[i];            /* :394-395 */
      orc_restore_UVdiag(nC, nP, U, V, UVdiag);              /* (U, V reused by the next lambda) */
      solved = 1;
      nu = 2;
      origin_lambda = lambda; /* :158-161 */
    }
    orc_compute_Jmultiply(nC, nO, JA, JB, iidx, jidx, PU, Jx1); /* :166-176 */
    orc_compute_Jmultiply(nC, nO, JA, JB, iidx, jidx, PB, Jx2);
    const double uBu = 2 * dot_n(2 * nO, Jx1, Jx1), uBb = 2 * dot_n(2 * nO, Jx1, Jx2),
                 bBb = 2 * dot_n(2 * nO, Jx2, Jx2);
    flag = ORC_ITER_CONTINUE;
    while (flag == ORC_ITER_CONTINUE) { /* :180-277 */
      tries++;
      const double p_norm = tr_step(nT, uBu, uBb, bBb, dk, PU, PB, P, g);
      orc_compute_newp(nA, nB, cams, pts, P, newcams, newpts);
      orc_compute_exQT(nO, K, impts, initrot, newcams, newpts, iidx, jidx, exn);
      const double act = orc_L2_sq(2 * nO, exn);
      if (fabs((ex_L2 - act) / ex_L2) < EPS2) { /* :197-202 */
        flag = ORC_ITER_DP_NO_CHANGE;
        break;
      }
      orc_compute_Jmultiply(nC, nO, JA, JB, iidx, jidx, P, Jx1); /* :209-213 */
      const double Jx_norm = 2 * orc_L2_sq(2 * nO, Jx1);
      const double pred = dot_n(nT, g, P) + ex_L2 + Jx_norm / 2;
      const double rho = (ex_L2 - act) / (ex_L2 - pred); /* :221-222 */
      int accepted = 0;
      if (rho < 0.25 || act > ex_L2) {
        dk = dk / 4;
      } else if (rho >= 0.75 && act < ex_L2) {
        accepted = 1;
        dk = fmin(2 * dk, MAX_DELTA);
      } else if (rho >= 0.25 && rho < 0.75 && act < ex_L2) {
        accepted = 1;
      } else if (isnan(rho)) {
        flag = ORC_ITER_TURN_TO_LM;
        final_err = ex_L2;
        goto done;
      }
      if (accepted) { /* update_p, :233-234,:244-245 */
        flag = 7; /* ITER_PASS */
        memcpy(cams, newcams, sizeof(double) * (size_t)nA);
        memcpy(pts, newpts, sizeof(double) * (size_t)nB);
        memcpy(ex, exn, sizeof(double) * 2 * (size_t)nO); /* residual at the new current parameters */
        final_err = act;
      }
      if (opts->verbose)
        printf("itno=%d\tErr:%.15E\tDelta=%f\tRho=%f\tnorm_p=%f\tLambda=%E\n", itno, act, dk, rho, p_norm, lambda);
      if (log && nlog < opts->log_cap) {
        double *row = log + 6 * nlog++;
        row[0] = itno; row[1] = act; row[2] = rho; row[3] = dk; row[4] = lambda; row[5] = accepted;
      }
      if (fabs((act - ex_L2) / ex_L2) <= EPS2) { /* :252-255 */
        flag = ORC_ITER_ERR_SMALL_ENOUGH;
        break;
      }
      if (rho < 0.25) { /* :257-264 */
        if (++notgood >= 5) {
          flag = ORC_ITER_TURN_TO_LM;
          break;
        }
      } else
        notgood = 0;
      if (rho > 0.75 && act < ex_L2) { /* :266-271 */
        if (++good_iters >= 10) {
          lambda = 0.0;
          origin_lambda = 0.0;
          good_iters = 0;
        }
      } else
        good_iters = 0;
      if (rho > 0.25 && act < ex_L2) ex_L2 = act; /* :272-275 */
    }
    if (flag != 7) break; /* :278-279 */
  }
done:
  res->flag = flag;
  res->iters = itno;
  res->tries = tries;
  res->final_err = final_err;
  res->lambda = lambda;
  res->delta = dk;
  res->n_log = nlog;
  free(ex); free(JA); free(JB); free(W); free(Y); free(U); free(V); free(Vinv); free(UVdiag); free(S); free(Sf);
  free(g); free(eab); free(PU); free(PB); free(P); free(Jx1); free(Jx2); free(newcams); free(newpts); free(exn);
  return flag;
}

/* ---- free intrinsics: camera block (fu, u0, v0, ar, s | v | t), 11 parameters -----------------
 * The reference reads this layout (PSBA/main.cpp:73,140-149; data/54camsvarK.txt) and never optimises the first
 * five (CL_files/PSBA.cl:5-7): there is no reference arithmetic, so this twin is PARITY UNPINNED.  It is written
 * differently from the HIP route on purpose -- the Jacobian of the intrinsics by its closed form here, checked by
 * central differences in tests/test_freek.py; the normal equations as ONE dense (11 nC + 3 nP)^2 matrix J^T J + mu I
 * solved by a plain Cholesky, no Schur complement -- so that the GPU's block elimination has an independent judge. */
#define FK 11
static void fk_project(const double *cam, const double *q0, const double *M, double *xy, double *P) {
  double q[4];
  compose_quat(q0, cam + 5, q);
  quat_rotate(q, M, P);
  for (int c = 0; c < 3; c++) P[c] += cam[8 + c];
  project(cam, P, xy);
}

/* ex[2 nO] = measured - projected */
void orc_fk_exQT(int nO, const double *impts, const double *initrot, const double *cams11, const double *pts,
                 const int *iidx, const int *jidx, double *ex) {
  for (int a = 0; a < nO; a++) {
    double xy[2], P[3];
    fk_project(cams11 + FK * jidx[a], initrot + 4 * jidx[a], pts + 3 * iidx[a], xy, P);
    ex[2 * a] = impts[2 * a] - xy[0];
    ex[2 * a + 1] = impts[2 * a + 1] - xy[1];
  }
}

/* JA[22 nO] (2 x 11 row-major), JB[6 nO]: the six extrinsic columns and B from the six-parameter Jacobian with
 * this camera's K, the five intrinsic columns in closed form */
void orc_fk_jacobi(int nO, const double *initrot, const double *cams11, const double *pts, const int *iidx,
                   const int *jidx, double *JA, double *JB) {
  for (int a = 0; a < nO; a++) {
    const int i = iidx[a], j = jidx[a], one = 0;
    const double *cam = cams11 + FK * j;
    double A6[12], xy[2], P[3];
    orc_compute_jacobiQT(1, cam, initrot + 4 * j, cam + 5, pts + 3 * i, &one, &one, A6, JB + 6 * a);
    fk_project(cam, initrot + 4 * j, pts + 3 * i, xy, P);
    const double xn = P[0] / P[2], yn = P[1] / P[2];
    double *A = JA + 2 * FK * a;
    A[0] = xn; A[1] = 1; A[2] = 0; A[3] = 0; A[4] = yn;
    A[FK + 0] = cam[3] * yn; A[FK + 1] = 0; A[FK + 2] = 1; A[FK + 3] = cam[0] * yn; A[FK + 4] = 0;
    for (int k = 0; k < 6; k++) {
      A[5 + k] = A6[k];
      A[FK + 5 + k] = A6[6 + k];
    }
  }
}

/* N = J^T J (dense, nT x nT, nT = 11 nC + 3 nP) and g = J^T e; returns ||e||^2 */
double orc_fk_normal(int nC, int nP, int nO, const double *impts, const double *initrot, const double *cams11,
                     const double *pts, const int *iidx, const int *jidx, double *N, double *g) {
  const int nA = FK * nC, nT = nA + 3 * nP;
  double *JA = (double *)xmalloc(sizeof(double) * 2 * FK * (size_t)nO), *JB = (double *)xmalloc(sizeof(double) * 6 * (size_t)nO);
  double *ex = (double *)xmalloc(sizeof(double) * 2 * (size_t)nO);
  orc_fk_jacobi(nO, initrot, cams11, pts, iidx, jidx, JA, JB);
  orc_fk_exQT(nO, impts, initrot, cams11, pts, iidx, jidx, ex);
  memset(N, 0, sizeof(double) * (size_t)nT * nT);
  memset(g, 0, sizeof(double) * (size_t)nT);
  double cost = 0;
  for (int a = 0; a < nO; a++) {
    int col[FK + 3];
    double row[2][FK + 3];
    for (int k = 0; k < FK; k++) {
      col[k] = FK * jidx[a] + k;
      row[0][k] = JA[2 * FK * a + k];
      row[1][k] = JA[2 * FK * a + FK + k];
    }
    for (int k = 0; k < 3; k++) {
      col[FK + k] = nA + 3 * iidx[a] + k;
      row[0][FK + k] = JB[6 * a + k];
      row[1][FK + k] = JB[6 * a + 3 + k];
    }
    for (int r = 0; r < FK + 3; r++) {
      g[col[r]] += row[0][r] * ex[2 * a] + row[1][r] * ex[2 * a + 1];
      for (int c = 0; c < FK + 3; c++) N[(size_t)col[r] * nT + col[c]] += row[0][r] * row[0][c] + row[1][r] * row[1][c];
    }
    cost += ex[2 * a] * ex[2 * a] + ex[2 * a + 1] * ex[2 * a + 1];
  }
  free(JA); free(JB); free(ex);
  return cost;
}

/* levmar() (PSBA/levmar.cpp:45-256: same damping schedule, gain ratio and stop tests) on the 11-parameter blocks,
 * every step from the DENSE damped normal equations.  Small problems only (nT^3 / 3 flops per try). */
int orc_fk_levmar(int nC, int nP, int nO, const double *impts, const double *initrot, double *cams11, double *pts,
                  const int *iidx, const int *jidx, const orc_lm_opts *opts, orc_lm_result *res, double *log) {
  const int nA = FK * nC, nB = 3 * nP, nT = nA + nB;
  const double STOP = 1e-12, EPS_SQ = 1e-24;
  double *N = (double *)xmalloc(sizeof(double) * (size_t)nT * nT), *Nd = (double *)xmalloc(sizeof(double) * (size_t)nT * nT);
  double *g = (double *)xmalloc(sizeof(double) * nT), *dp = (double *)xmalloc(sizeof(double) * nT);
  double *nc = (double *)xmalloc(sizeof(double) * nA), *np_ = (double *)xmalloc(sizeof(double) * nB);
  double *ex = (double *)xmalloc(sizeof(double) * 2 * (size_t)nO);
  double mu = 0, p_L2 = 1e3, ex_L2 = 0;
  int nu = 2, tries = 0, nlog = 0, flag = 3, itno = 0;
  memset(res, 0, sizeof *res);
  for (; itno < opts->max_iter && flag == 3; itno++) {
    ex_L2 = orc_fk_normal(nC, nP, nO, impts, initrot, cams11, pts, iidx, jidx, N, g);
    if (itno == 0) {
      res->init_err = ex_L2;
      double mx = 0;
      for (int t = 0; t < nT; t++) mx = N[(size_t)t * nT + t] > mx ? N[(size_t)t * nT + t] : mx;
      mu = (opts->init_mu != 0.0 ? opts->init_mu : 1e-3) * mx;
      res->mu0 = mu;
    }
    while (1) {
      tries++;
      memcpy(Nd, N, sizeof(double) * (size_t)nT * nT);
      for (int t = 0; t < nT; t++) Nd[(size_t)t * nT + t] += mu;
      const double bad = orc_chol_solve(nT, Nd, g, dp);
      if (bad == 0.0) {
        double dp_L2 = 0, den = 0, newp = 0;
        for (int t = 0; t < nT; t++) {
          dp_L2 += dp[t] * dp[t];
          den += dp[t] * (mu * dp[t] + g[t]);
        }
        if (dp_L2 < p_L2 * STOP * STOP) { flag = 5; break; }
        if (dp_L2 >= (p_L2 + STOP) / EPS_SQ) { flag = 4; break; }
        for (int t = 0; t < nA; t++) { nc[t] = cams11[t] + dp[t]; newp += nc[t] * nc[t]; }
        for (int t = 0; t < nB; t++) { np_[t] = pts[t] + dp[nA + t]; newp += np_[t] * np_[t]; }
        orc_fk_exQT(nO, impts, initrot, nc, np_, iidx, jidx, ex);
        double new_L2 = 0;
        for (int t = 0; t < 2 * nO; t++) new_L2 += ex[t] * ex[t];
        const double rho = (ex_L2 - new_L2) / den;
        if (log && nlog < opts->log_cap) {
          double *row = log + 5 * nlog++;
          row[0] = itno; row[1] = new_L2; row[2] = rho; row[3] = mu; row[4] = rho > 0;
        }
        if (rho > 0) {
          double tmp = 2 * rho - 1;
          tmp = 1.0 - tmp * tmp * tmp;
          mu *= tmp >= 1.0 / 3.0 ? tmp : 1.0 / 3.0;
          nu = 2;
          memcpy(cams11, nc, sizeof(double) * nA);
          memcpy(pts, np_, sizeof(double) * nB);
          p_L2 = newp;
          ex_L2 = new_L2;
          break;
        }
      }
      mu *= nu;
      if (2.0 * nu > 1e9) { flag = 4; break; }
      nu *= 2;
    }
    if (ex_L2 <= STOP) flag = 6;
  }
  res->flag = flag; res->iters = itno; res->tries = tries; res->final_err = ex_L2; res->n_log = nlog;
  free(N); free(Nd); free(g); free(dp); free(nc); free(np_); free(ex);
  return flag;
}
