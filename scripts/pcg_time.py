"""Development helper: dense factorization against block-sparse S + PCG on banded problems
(a point's cameras within a window of neighbours), per solve and per LM iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth

for n_cams, window, n_pts in [(1000, 40, 40000), (2000, 40, 80000), (2000, 100, 80000)]:
    prob = synth.make_problem(n_cams=n_cams, n_pts=n_pts, mean_track=5.0, seed=11 + n_cams, window=window)
    for solver in (0, 1):
        h = psba_amd.Psba(0)
        if solver:
            h.set_solver(1, tol=1e-10, max_iter=2000)
        h.upload_problem(prob)
        h.linearize(1.0, 1.0)
        mu = 1e-3 * h.max_diag()
        h.profile_enable(True)
        for _ in range(2):
            h.schur_assemble(mu); h.schur_reduce(); h.schur_solve()
        h.profile_reset()
        for _ in range(5):
            h.schur_assemble(mu); h.schur_reduce(); h.schur_solve()
        out = []
        for name, k in (("schur", capi.K_SCHUR), ("reduce", capi.K_SCHUR_REDUCE), ("solve", capi.K_CHOLESKY)):
            ms, n = h.profile_get(k)
            out.append(f"{name} {1e3 * ms / max(n, 1):9.1f} us")
        info = ""
        if solver:
            it, rel, nb, nd = h.pcg_info()
            info = f"  iterations {it}, relres {rel:.1e}, blocks {nb} of {nd} ({100.0 * nb / nd:.1f} %)"
        t0 = time.perf_counter()
        res, _ = h.levmar(max_iter=5)
        dt = time.perf_counter() - t0
        print(f"{n_cams} cameras, window {window}, {prob['nO']} obs, "
              f"{'PCG  ' if solver else 'dense'}: " + ", ".join(out) + f", LM {1e3 * dt / max(res.iters, 1):8.2f} ms/iter" + info, flush=True)
        h.close()
