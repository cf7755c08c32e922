"""Development helper (GPU box): the dense factorization at mid sizes (blocked chain with the look-ahead) with the
32-column steps on all rows / the diagonal-only steps + block solve, look-ahead on / off: HIP-event time of the Cholesky class.
usage: chol_mid_sizes.py n_cams [n_cams ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psba_amd
from psba_amd import capi, synth

for n_cams in [int(a) for a in sys.argv[1:]]:
    prob = synth.make_problem(n_cams=n_cams, n_pts=40000, mean_track=5.0, seed=7)
    for label, env in (("default", {}), ("all rows", {"PSBA_CHOL_STEPS_ALL_ROWS": "1"}), ("diagonal only", {"PSBA_CHOL_STEPS_DIAG_ONLY": "1"}),
                       ("look-ahead on", {"PSBA_CHOL_LOOKAHEAD": "1"}), ("look-ahead off", {"PSBA_CHOL_LOOKAHEAD": "0"})):
        for k in ("PSBA_CHOL_STEPS_DIAG_ONLY", "PSBA_CHOL_STEPS_ALL_ROWS", "PSBA_CHOL_LOOKAHEAD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        h = psba_amd.Psba(0)
        h.upload_problem(prob)
        h.levmar(max_iter=2, tr_handoff=False, log_cap=0)
        h.reset_params()
        h.profile_enable(True)
        h.profile_reset()
        res, _ = h.levmar(max_iter=4, tr_handoff=False, log_cap=0)
        ms, n = h.profile_get(capi.K_CHOLESKY)
        h.profile_enable(False)
        h.reset_params()
        import time
        t0 = time.perf_counter()
        res2, _ = h.levmar(max_iter=6, tr_handoff=False, log_cap=0)
        per = 1e3 * (time.perf_counter() - t0) / max(res2.iters, 1)
        print(f"{n_cams} cameras (n = {6 * n_cams}) {label:14s}: cholesky {1e3 * ms / max(n, 1):9.1f} us  ({n} solves, final cost {res.final_err:.8g}); {per:.3f} ms per LM iteration", flush=True)
        h.close()
