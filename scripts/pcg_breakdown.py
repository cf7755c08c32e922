"""Development helper: per-kernel-class time of an LM iteration with the block-sparse solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth
names = [("linearize", capi.K_LINEARIZE), ("schur", capi.K_SCHUR), ("reduce", capi.K_SCHUR_REDUCE),
         ("solve", capi.K_CHOLESKY), ("backsub", capi.K_BACKSUB)]
prob = synth.make_problem(n_cams=2000, n_pts=80000, mean_track=5.0, seed=2011, window=40)
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-10
for solver in (1, 0):
    h = psba_amd.Psba(0)
    if solver:
        h.set_solver(1, tol=tol, max_iter=2000)
    h.upload_problem(prob)
    h.levmar(max_iter=3, tr_handoff=False)
    h.upload_problem(prob)
    res0, _ = h.levmar(max_iter=8, tr_handoff=False)
    h.upload_problem(prob)
    h.profile_enable(True)
    h.profile_reset()
    res, _ = h.levmar(max_iter=8, tr_handoff=False)
    parts = []
    for nm, k in names:
        ms, n = h.profile_get(k)
        parts.append(f"{nm} {1e3 * ms / max(n, 1):8.1f} us x{n}")
    it = h.pcg_info()[0] if solver else 0
    print(f"solver={solver} tol={tol:g} last solve {it} iterations, final cost {res.final_err:.9e} tries={res.tries} unprofiled {1e3 * res0.seconds / max(res0.iters, 1):8.3f} ms/iter, profiled {1e3 * res.seconds / max(res.iters, 1):8.3f}  " + "  ".join(parts), flush=True)
    h.close()
