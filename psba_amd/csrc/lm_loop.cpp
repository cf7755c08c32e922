// lm_loop.cpp -- the Levenberg-Marquardt caller restated over the C ABI.
// Follows reference PSBA/levmar.cpp:45-256 step for step (Nielsen mu/nu update, the same
// stop tests and ITER_* flags); what changes is only what crosses the device boundary: per
// damping try one struct of scalars instead of dp (nT doubles), ex (2 nO doubles) and three
// status words, and per outer iteration nothing instead of g (nT doubles).
#include <chrono>
#include <cmath>
#include <cstdio>

#include "psba_internal.h"

extern "C" {

void psba_lm_default_options(psba_lm_options *o) {
  if (!o) return;
  o->max_iter = 50;   // levmar.cpp:100
  o->tr_handoff = 1;  // levmar.cpp:215-219
  o->verbose = 0;
  o->log_cap = 0;
  o->start_itno = 0;
  o->init_mu = 0.0;
}

int psba_levmar(psba_handle h, const psba_lm_options *opts, psba_lm_result *res, double *log) {
  if (!h || !opts || !res) return PSBA_E_INVALID;
  const double STOP = 1e-12;           // PSBA_STOP_THRESH, psba.h:7
  const double EPS_SQ = 1e-12 * 1e-12; // PSBA_EPSILON_SQ, psba.h:10
  const double tau = opts->init_mu != 0.0 ? opts->init_mu : 1e-3;  // PSBA_INIT_MU, psba.h:6
  *res = psba_lm_result();
  auto t_begin = std::chrono::steady_clock::now();
  int rc;
#define LM_TRY(x)          \
  do {                     \
    rc = (x);              \
    if (rc < 0) return rc; \
  } while (0)

  double ex_L2 = 0, new_ex_L2 = 0, p_L2 = 0, dp_L2 = 0, mu = 0, rho = 0;
  int nu = 2, gooditer = 0, tries = 0, nlog = 0;
  bool first = true;
  int flag = PSBA_ITER_CONTINUE;
  int itno = opts->start_itno;

  // levmar.cpp:93-95 and, of the first iteration, :103-108 and :114-120 -- one synchronisation
  double maxdiag0 = 0;
  LM_TRY(psba_begin(h, 1.0, 1.0, &ex_L2, &maxdiag0));
  res->init_err = ex_L2;

  for (; itno < opts->max_iter && flag == PSBA_ITER_CONTINUE; itno++) {  // :100
    LM_TRY(psba_linearize(h, 1.0, 1.0));                                  // :103-108
    if (first) {                                                          // :114-120
      mu = tau * maxdiag0;
      res->mu0 = mu;
      first = false;
      p_L2 = 1e+3;
      nu = 2;
    }
    while (true) {  // :124
      tries++;
      psba_try_scalars sc;
      LM_TRY(psba_schur_assemble(h, mu));  // :126-131
      LM_TRY(psba_schur_reduce(h));
      LM_TRY(psba_schur_solve(h));         // :134-140
      if (rc == PSBA_PCG_MAXIT) {          // iterative solve only: an inexact step; rho judges it
        res->pcg_unconverged++;
        if (opts->verbose) {
          int it = 0;
          double rr = 0;
          psba_pcg_info(h, &it, &rr, nullptr, nullptr);
          printf("itno=%d\t\tpcg: %d iterations without reaching the tolerance, relres=%.3e\n", itno, it, rr);
        }
      }
      // :151-157,185-193 -- and, while the host looks at the scalars, the next iteration's
      // linearization at the proposed parameters (dropped if the step is rejected)
      LM_TRY(psba_backsub_async(h, mu));
      // (queued before the try's status is known: after a failed factorization K3 has written
      // non-finite proposed parameters and this linearization of them is ~35 us of wasted GPU time
      // per failed try -- its results are dropped with the rejected step, nothing else depends on
      // them, and skipping the launch would need the status on the host first, i.e. the very
      // round trip the look-ahead exists to hide)
      if (itno + 1 < opts->max_iter) LM_TRY(psba_linearize_ahead(h));  // no iteration left to use it otherwise
      LM_TRY(psba_backsub_wait(h, &sc));
      if (!(sc.status & PSBA_NOT_SPD)) {
        dp_L2 = sc.dp_l2;
        if (dp_L2 < p_L2 * STOP * STOP) {  // :169-173
          flag = PSBA_ITER_DP_NO_CHANGE;
          break;
        }
        if (dp_L2 >= (p_L2 + STOP) / EPS_SQ) {  // :174-179
          flag = PSBA_ITER_ERR;
          break;
        }
        new_ex_L2 = sc.new_cost;
        rho = (ex_L2 - new_ex_L2) / sc.gain_den;  // :195, 271-280
        if (opts->verbose)
          printf("itno=%d\t\tErr=%.15E\t\trho=%f\t\tmu=%f\n", itno, new_ex_L2, rho, mu);  // :197
        if (log && nlog < opts->log_cap) {
          double *row = log + 5 * nlog++;
          row[0] = itno; row[1] = new_ex_L2; row[2] = rho; row[3] = mu; row[4] = rho > 0 ? 1 : 0;
        }
        if (rho > 0) {  // :200-223
          double tmp = 2 * rho - 1;
          tmp = 1.0 - tmp * tmp * tmp;
          mu = mu * ((tmp >= (1.0 / 3.0)) ? tmp : (1.0 / 3.0));
          nu = 2;
          LM_TRY(psba_accept(h));  // update_p :211
          p_L2 = sc.newp_l2;       // :212
          ex_L2 = new_ex_L2;
          if (std::fabs(rho - 1) < (1.0 / 5.0)) {
            gooditer++;
            if (opts->tr_handoff && gooditer >= 5) {
              flag = PSBA_ITER_TURN_TO_TR;
              break;
            }
          } else {
            gooditer = 0;
          }
          break;
        }
      } else {  // :227-233
        gooditer = 0;
        if (log && nlog < opts->log_cap) {
          double *row = log + 5 * nlog++;
          row[0] = itno; row[1] = NAN; row[2] = NAN; row[3] = mu; row[4] = -1;
        }
      }
      mu *= nu;  // :237-244
      const double nu2 = 2.0 * nu;
      if (nu2 <= nu || nu2 > 1e9) {
        flag = PSBA_ITER_ERR;
        break;
      }
      nu = (int)nu2;
    }
    if (ex_L2 <= STOP) flag = PSBA_ITER_ERR_SMALL_ENOUGH;  // :247-248
  }
#undef LM_TRY
  // nothing of this call is left on the device when it returns (a linearization computed ahead
  // for a step that was then not taken may still be running)
  if (hipStreamSynchronize(h->stream) != hipSuccess) return PSBA_E_HIP;
  res->flag = flag;
  res->iters = itno;
  res->tries = tries;
  res->final_err = ex_L2;
  res->mu_final = mu;
  res->n_log = nlog;
  res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  return flag;
}

}  // extern "C"
