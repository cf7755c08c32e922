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

namespace {

double dot_n(size_t n, const double *a, const double *b) {  // PSBA/misc.cpp dotProduct
  double s = 0;
  for (size_t i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

// trust_region.cpp:520-595 (compute_p_2): the minimiser of the quadratic model over
// span{P_U, P_B} when it lies inside the region, else the dog-leg path.
double tr_step(size_t n, double uBu, double uBb, double bBb, double delta, const double *PU, const double *PB,
               double *p, const double *g) {
  const double ug = dot_n(n, PU, g), bg = dot_n(n, PB, g);
  const double det = -uBb * uBb + bBb * uBu;
  const double eta1 = (bg * uBb) / det - (bBb * ug) / det;
  const double eta2 = (ug * uBb) / det - (bg * uBu) / det;
  double nrm = 0;
  for (size_t i = 0; i < n; i++) {
    p[i] = eta1 * PU[i] + eta2 * PB[i];
    nrm += p[i] * p[i];
  }
  nrm = std::sqrt(nrm);
  if (!(nrm > delta)) return nrm;
  double nu = 0, nb = 0;
  for (size_t i = 0; i < n; i++) {
    nu += PU[i] * PU[i];
    nb += PB[i] * PB[i];
  }
  nu = std::sqrt(nu);
  nb = std::sqrt(nb);
  if (nu > delta) {
    for (size_t i = 0; i < n; i++) p[i] = delta * PU[i] / nu;
    return delta;
  }
  if (nb <= delta) {  // (the reference returns sqrt(nrm + nb^2) here: a value it only prints)
    for (size_t i = 0; i < n; i++) p[i] = PB[i];
    return nb;
  }
  double a = 0, b = 0, c = 0;
  for (size_t i = 0; i < n; i++) {
    const double Ai = PB[i] - PU[i], Bi = 2 * PU[i] - PB[i];
    a += Ai * Ai;
    b += Ai * Bi;
    c += Bi * Bi;
  }
  b = 2 * b;
  c = c - delta * delta;
  double disc = b * b - 4 * a * c;
  if (std::fabs(disc) < 1e-12) disc = 0;
  const double tau = (-b + std::sqrt(disc)) / (2 * a);
  for (size_t i = 0; i < n; i++) p[i] = PU[i] + (tau - 1) * (PB[i] - PU[i]);
  return delta;
}

}  // namespace

extern "C" {

void psba_tr_default_options(psba_tr_options *o) {
  if (!o) return;
  o->max_iter = 50;  // trust_region.cpp:112
  o->start_itno = 0;
  o->verbose = 0;
  o->log_cap = 0;
}

int psba_trust_region(psba_handle h, const psba_tr_options *opts, psba_tr_result *res, double *log) {
  if (!h || !opts || !res) return PSBA_E_INVALID;
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
  const size_t nT = (size_t)6 * nC + (size_t)3 * nP;
  std::vector<double> g(nT), PU(nT), PB(nT), P(nT);
  double dk = 1, lambda = 0, origin_lambda = 0;  // :95-96
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
    const double gtBg = 2 * d3[0], gtg = dot_n(nT, g.data(), g.data());
    for (size_t i = 0; i < nT; i++) PU[i] = -(g[i] * gtg) / gtBg;  // :128-130 Cauchy step
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
    flag = PSBA_ITER_CONTINUE;
    while (flag == PSBA_ITER_CONTINUE) {  // :180-277
      tries++;
      const double p_norm = tr_step(nT, uBu, uBb, bBb, dk, PU.data(), PB.data(), P.data(), g.data());
      TR_TRY(psba_set_step(h, P.data()));  // dp_buffer <- P, compute_newp (:183-187)
      double act = 0;
      TR_TRY(psba_residual(h, PSBA_PARAMS_NEW, &act));  // :192-194
      if (std::fabs((ex_L2 - act) / ex_L2) < EPS2) {    // :197-202
        flag = PSBA_ITER_DP_NO_CHANGE;
        break;
      }
      TR_TRY(psba_jmul_dots(h, P.data(), nullptr, d3));  // :209-213
      const double Jx_norm = 2 * d3[0];
      const double pred = dot_n(nT, g.data(), P.data()) + ex_L2 + Jx_norm / 2;
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
