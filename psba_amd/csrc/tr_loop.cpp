// tr_loop.cpp -- the dog-leg trust-region caller and the driver's LM <-> TR alternation,
// restated over the C ABI.
// Follows reference PSBA/trust_region.cpp:49-288 (trust_region), :292-405 (compute_PB) and
// :520-595 (compute_p_2) step for step, and PSBA/main.cpp:193-208 for the alternation with the
// shared iteration counter.  As in the reference the step vectors (g, P_U, P_B, P) live on the
// host and the loop does its O(nT) vector arithmetic there (dotProduct, the dog-leg blend); what
// runs on the device is what the reference runs on the device: the linearization with coeff 2 /
// -2, J x products (psba_jmul_dots: the three dot products at once instead of copying the dense
// nP x nC x 2 vectors back), the damped Schur solve, the modified-Cholesky lambda estimate, the
// cost at the proposal.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>

#include "psba_internal.h"

extern "C" int psba_jmul_dots(psba_handle h, const double *x1, const double *x2, double dots[3]);
extern "C" int psba_get_gradient(psba_handle h, double *g);
extern "C" int psba_get_dp(psba_handle h, double *dp);
extern "C" int psba_set_step(psba_handle h, const double *dp);
extern "C" int psba_cholmod_lambda(psba_handle h, int reassemble, double *lambda, double *info3);
extern "C" int psba_allreduce_scalars(psba_handle h, double *v, int n);

namespace {

// Inner products of parameter vectors [cameras ; points] (PSBA/misc.cpp dotProduct).  With the points
// sharded over ranks the camera parts are the same on every rank and the point parts add up: K products
// at once, one small collective (none on a single rank).
template <int K>
int dots(psba_handle h, size_t nA, size_t nT, const double *const (&x)[K], const double *const (&y)[K], double (&out)[K]) {
  double pts[K];
  for (int k = 0; k < K; k++) {
    double c = 0, q = 0;
    for (size_t i = 0; i < nA; i++) c += x[k][i] * y[k][i];
    for (size_t i = nA; i < nT; i++) q += x[k][i] * y[k][i];
    out[k] = c;
    pts[k] = q;
  }
  const int rc = psba_allreduce_scalars(h, pts, K);
  for (int k = 0; k < K; k++) out[k] += pts[k];
  return rc;
}

// The step inside the trust region (what the reference's compute_p_2 returns, PSBA/trust_region.cpp:
// 520-595), written from the mathematics.  Model: m(p) = g^T p + p^T B p / 2 with B = 2 J^T J.
// Every candidate is a combination p = alpha P_U + beta P_B of the Cauchy step P_U and the
// Gauss-Newton step P_B, so five inner products decide everything:
//  1. the minimiser of m over span{P_U, P_B}: the 2x2 system  [uBu uBb; uBb bBb] (a, b)^T = -(u.g, b.g)^T;
//     taken when it lies inside the region (|p|^2 = a^2 u.u + 2 a b u.b + b^2 b.b);
//  2. else the dog-leg path: the Cauchy direction cut at the boundary when P_U alone leaves the
//     region, P_B when it is inside, else the point of the segment P_U -> P_B on the boundary:
//     |P_U + s (P_B - P_U)|^2 = delta^2, the root s in [0, 1] of  (d.d) s^2 + 2 (u.d) s + (u.u - delta^2) = 0,
//     d = P_B - P_U.
// Returns |p|; p = alpha P_U + beta P_B.
// dd = d.d and ud = u.d are summed over d itself (as the reference does, trust_region.cpp:574-586): formed from
// u.u, u.b and b.b they lose their digits to cancellation when P_U is close to P_B, and dd divides the root.
double tr_step(double uBu, double uBb, double bBb, double delta, double uu, double ub, double bb, double ug, double bg,
               double dd, double ud, double &alpha, double &beta) {
  double len;
  {
    const double det = uBu * bBb - uBb * uBb;  // Cramer on the 2x2 Gram system of B
    alpha = (uBb * bg - bBb * ug) / det;
    beta = (uBb * ug - uBu * bg) / det;
    len = std::sqrt(alpha * alpha * uu + 2 * alpha * beta * ub + beta * beta * bb);
  }
  if (len > delta) {  // (a NaN minimiser -- singular Gram matrix -- is taken as is, as the reference's comparison does)
    const double lu = std::sqrt(uu), lb = std::sqrt(bb);
    if (lu > delta) {
      alpha = delta / lu;
      beta = 0;
      len = delta;
    } else if (lb <= delta) {  // (the reference returns sqrt(|p|^2 + |P_B|^2) here: a value it only prints)
      alpha = 0;
      beta = 1;
      len = lb;
    } else {
      const double c = uu - delta * delta;
      double disc = ud * ud - dd * c;
      if (std::fabs(disc) < 0.25e-12) disc = 0;  // the reference zeroes |b^2 - 4ac| < 1e-12 (b = 2 ud); below that, NaN as there
      const double sgm = (-ud + std::sqrt(disc)) / dd;
      alpha = 1 - sgm;
      beta = sgm;
      len = delta;
    }
  }
  return len;
}

}  // namespace

extern "C" {

void psba_tr_default_options(psba_tr_options *o) {
  if (!o) return;
  o->max_iter = 50;  // trust_region.cpp:112
  o->start_itno = 0;
  o->verbose = 0;
  o->log_cap = 0;
  o->init_lambda = 0.0;
}

int psba_trust_region(psba_handle h, const psba_tr_options *opts, psba_tr_result *res, double *log) {
  if (!h || !opts || !res) return PSBA_E_INVALID;
  if (h->cnp != 6) return PSBA_E_STATE;  // free intrinsics: psba_levmar only (include/psba_hip.h)
  const double EPS2 = 1e-12, MAX_DELTA = 10000;  // psba.h:9, trust_region.cpp:18
  *res = psba_tr_result();
  auto t_begin = std::chrono::steady_clock::now();
  int rc;
#define TR_TRY(x)          \
  do {                     \
    rc = (x);              \
    if (rc < 0) return rc; \
  } while (0)
  int nC = 0, nP = 0, nO = 0;
  TR_TRY(psba_get_dims(h, &nC, &nP, &nO));
  const size_t nA = (size_t)6 * nC, nT = nA + (size_t)3 * nP;
  std::vector<double> g(nT), PU(nT), PB(nT), P(nT);
  double dk = 1, lambda = opts->init_lambda, origin_lambda = 0;  // :95-96
  int nu = 2, notgood = 0, good_iters = 0, itno = opts->start_itno, nlog = 0, tries = 0, chol_fail = 0;
  int flag = PSBA_ITER_CONTINUE;
  double ex_L2 = 0;
  TR_TRY(psba_residual(h, PSBA_PARAMS_CUR, &ex_L2));  // :106-107
  res->init_err = ex_L2;
  double final_err = ex_L2;
  bool stop = false;

  for (; itno < opts->max_iter && !stop; itno++) {  // :112
    TR_TRY(psba_linearize(h, 2.0, -2.0));            // :117-122,133-137: B = 2 J^T J, g = grad ||e||^2
    TR_TRY(psba_get_gradient(h, g.data()));
    double d3[3];
    TR_TRY(psba_jmul_dots(h, g.data(), nullptr, d3));  // :125
    double gtg1[1];
    {
      const double *const xs[1] = {g.data()};
      TR_TRY(dots<1>(h, nA, nT, xs, xs, gtg1));
    }
    const double gtBg = 2 * d3[0], gtg = gtg1[0];
    for (size_t i = 0; i < nT; i++) PU[i] = -(g[i] * gtg) / gtBg;  // :128-130 Cauchy step
    // block-sparse mode: the conjugate gradients do not stop at the singular S of lambda = 0 the way a factorization
    // does (the gauge directions of an unanchored problem: a consistent semidefinite system) -- they crawl, and the
    // step they return leans along those directions as the rounding of their sums happens to fall: the same problem
    // then takes 11 or 50 iterations, run by run.  The sparse mode therefore never solves undamped: it starts where
    // the reference ends up after its failed factorization (trust_region.cpp:341-363), at a damping at rounding level
    // of the diagonal.
    if (h->solver == PSBA_SOLVER_PCG && lambda == 0.0) {
      double md = 0;
      TR_TRY(psba_max_diag(h, &md));
      lambda = 1e-8 * md;
    }
    bool solved = false;
    while (!solved) {  // :141-163 around compute_PB (:292-405)
      psba_try_scalars sc;
      TR_TRY(psba_schur_assemble(h, lambda));
      TR_TRY(psba_schur_reduce(h));
      TR_TRY(psba_schur_solve(h));
      TR_TRY(psba_backsub(h, lambda, &sc));
      if (sc.status & PSBA_NOT_SPD) {
        chol_fail++;
        if (lambda == 0.0) {  // :341-363
          TR_TRY(psba_cholmod_lambda(h, 1, &lambda, nullptr));
        } else {
          lambda = 2 * lambda;  // :365-368
        }
        if (origin_lambda != 0.0) {  // :144-155
          if (nu > 4) {
            flag = PSBA_ITER_TURN_TO_LM;
            final_err = ex_L2;
            stop = true;
            break;
          }
          lambda = lambda * nu;
          nu = nu * 2;
        }
        if (!std::isfinite(lambda) || chol_fail > 200) {  // (guard: the reference would spin)
          flag = PSBA_ITER_ERR;
          stop = true;
          break;
        }
        continue;
      }
      TR_TRY(psba_get_dp(h, P.data()));
      for (size_t i = 0; i < nT; i++) PB[i] = -P[i];  // :394-395
      solved = true;
      nu = 2;
      origin_lambda = lambda;  // :158-161
    }
    if (stop) break;
    TR_TRY(psba_jmul_dots(h, PU.data(), PB.data(), d3));  // :166-176
    const double uBu = 2 * d3[0], uBb = 2 * d3[1], bBb = 2 * d3[2];
    double ip[7];  // u.u, u.b, b.b, u.g, b.g, d.d, u.d with d = P_B - P_U
    {
      for (size_t i = 0; i < nT; i++) P[i] = PB[i] - PU[i];  // (P is free here: it is rebuilt for every try below)
      const double *const xs[7] = {PU.data(), PU.data(), PB.data(), PU.data(), PB.data(), P.data(), PU.data()};
      const double *const ys[7] = {PU.data(), PB.data(), PB.data(), g.data(), g.data(), P.data(), P.data()};
      TR_TRY(dots<7>(h, nA, nT, xs, ys, ip));
    }
    flag = PSBA_ITER_CONTINUE;
    while (flag == PSBA_ITER_CONTINUE) {  // :180-277
      tries++;
      double alpha, beta;
      const double p_norm = tr_step(uBu, uBb, bBb, dk, ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6], alpha, beta);
      for (size_t i = 0; i < nT; i++) P[i] = alpha * PU[i] + beta * PB[i];
      TR_TRY(psba_set_step(h, P.data()));  // dp_buffer <- P, compute_newp (:183-187)
      double act = 0;
      TR_TRY(psba_residual(h, PSBA_PARAMS_NEW, &act));  // :192-194
      if (std::fabs((ex_L2 - act) / ex_L2) < EPS2) {    // :197-202
        flag = PSBA_ITER_DP_NO_CHANGE;
        break;
      }
      TR_TRY(psba_jmul_dots(h, P.data(), nullptr, d3));  // :209-213
      const double Jx_norm = 2 * d3[0];
      const double pred = alpha * ip[3] + beta * ip[4] + ex_L2 + Jx_norm / 2;  // g.p = alpha u.g + beta b.g
      const double rho = (ex_L2 - act) / (ex_L2 - pred);  // :221-222
      bool accepted = false;
      if (rho < 0.25 || act > ex_L2) {
        dk = dk / 4;
      } else if (rho >= 0.75 && act < ex_L2) {
        accepted = true;
        dk = std::fmin(2 * dk, MAX_DELTA);
      } else if (rho >= 0.25 && rho < 0.75 && act < ex_L2) {
        accepted = true;
      } else if (std::isnan(rho)) {
        flag = PSBA_ITER_TURN_TO_LM;
        final_err = ex_L2;
        stop = true;
        break;
      }
      if (accepted) {  // update_p, :233-234,:244-245
        flag = PSBA_ITER_PASS;
        TR_TRY(psba_accept(h));
        final_err = act;
      }
      if (opts->verbose)
        printf("itno=%d\tErr:%.15E\tDelta=%f\tRho=%f\tnorm_p=%f\tLambda=%E\n", itno, act, dk, rho, p_norm, lambda);  // :250
      if (log && nlog < opts->log_cap) {
        double *row = log + 6 * nlog++;
        row[0] = itno; row[1] = act; row[2] = rho; row[3] = dk; row[4] = lambda; row[5] = accepted ? 1 : 0;
      }
      if (std::fabs((act - ex_L2) / ex_L2) <= EPS2) {  // :252-255
        flag = PSBA_ITER_ERR_SMALL_ENOUGH;
        break;
      }
      if (rho < 0.25) {  // :257-264
        if (++notgood >= 5) {
          flag = PSBA_ITER_TURN_TO_LM;
          break;
        }
      } else {
        notgood = 0;
      }
      if (rho > 0.75 && act < ex_L2) {  // :266-271
        if (++good_iters >= 10) {
          lambda = 0.0;
          origin_lambda = 0.0;
          good_iters = 0;
        }
      } else {
        good_iters = 0;
      }
      if (rho > 0.25 && act < ex_L2) ex_L2 = act;  // :272-275
    }
    if (stop || flag != PSBA_ITER_PASS) break;  // :278-279
  }
#undef TR_TRY
  if (hipStreamSynchronize(h->stream) != hipSuccess) return PSBA_E_HIP;
  res->flag = flag;
  res->iters = itno;
  res->tries = tries;
  res->chol_fail = chol_fail;
  res->final_err = final_err;
  res->lambda = lambda;
  res->delta = dk;
  res->n_log = nlog;
  res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  return flag;
}

// PSBA/main.cpp:193-208: levmar() until it hands over (ITER_TURN_TO_TR), trust_region() until it
// hands back (ITER_TURN_TO_LM); itno is shared and capped at max_iter in total.
int psba_solve(psba_handle h, int max_iter, int verbose, psba_solve_result *res) {
  if (!h || !res) return PSBA_E_INVALID;
  *res = psba_solve_result();
  auto t_begin = std::chrono::steady_clock::now();
  int itno = 0, flag = PSBA_ITER_CONTINUE;
  bool first = true;
  while (true) {
    psba_lm_options lo;
    psba_lm_default_options(&lo);
    lo.max_iter = max_iter;
    lo.verbose = verbose;
    lo.start_itno = itno;
    psba_lm_result lr;
    int rc = psba_levmar(h, &lo, &lr, nullptr);
    if (rc < 0) return rc;
    res->lm_calls++;
    if (first) res->init_err = lr.init_err;
    first = false;
    res->final_err = lr.final_err;
    itno = lr.iters;
    flag = lr.flag;
    if (flag != PSBA_ITER_TURN_TO_TR) break;
    psba_tr_options to;
    psba_tr_default_options(&to);
    to.max_iter = max_iter;
    to.verbose = verbose;
    to.start_itno = itno;
    psba_tr_result tr;
    rc = psba_trust_region(h, &to, &tr, nullptr);
    if (rc < 0) return rc;
    res->tr_calls++;
    res->final_err = tr.final_err;
    itno = tr.iters;
    flag = tr.flag;
    if (flag != PSBA_ITER_TURN_TO_LM) break;
  }
  res->flag = flag;
  res->iters = itno;
  res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  return flag;
}

}  // extern "C"
