"""Development helper: per-kernel-class time of one LM iteration over camera counts (HIP-event
profile classes of the library), to see which kernel dominates where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth

names = [("linearize", capi.K_LINEARIZE), ("schur", capi.K_SCHUR), ("reduce", capi.K_SCHUR_REDUCE),
         ("cholesky", capi.K_CHOLESKY), ("backsub", capi.K_BACKSUB)]
for n_cams in [int(c) for c in sys.argv[1:]] or [52, 130, 257, 400, 600, 1000]:
    prob = synth.make_problem(n_cams=n_cams, n_pts=40000, mean_track=5.42, seed=5)
    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    h.levmar(max_iter=3, tr_handoff=False)
    h.upload_problem(prob)
    h.profile_enable(True)
    h.profile_reset()
    res, _ = h.levmar(max_iter=6, tr_handoff=False)
    parts = []
    for nm, k in names:
        ms, n = h.profile_get(k)
        parts.append(f"{nm} {1e3 * ms / max(n, 1):8.1f} us x{n}")
    print(f"nC={n_cams:5d} n={6 * n_cams:5d} tries={res.tries} {1e3 * res.seconds / max(res.iters, 1):8.3f} ms/iter (profiled)  " + "  ".join(parts), flush=True)
    h.close()
