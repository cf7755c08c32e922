"""Development helper: K2 (assemble + reduce/finalize) time of the LDS-partition route against the
owner route over camera counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth

import sys as _s
for n_cams, n_pts in ([(int(c), 40000) for c in _s.argv[1:]] or [(52, 64053), (96, 40000), (130, 40000), (160, 40000), (200, 40000)]):
    prob = synth.make_problem(n_cams=n_cams, n_pts=n_pts, mean_track=5.42, seed=5)
    for owner in (0, 1):
        if owner:
            os.environ["PSBA_SCHUR_OWNER"] = "1"
        else:
            os.environ.pop("PSBA_SCHUR_OWNER", None)
        h = psba_amd.Psba(0)
        h.upload_problem(prob)
        h.linearize(1.0, 1.0)
        mu = 1e-3 * h.max_diag()
        h.profile_enable(True)
        for _ in range(3):
            h.schur_assemble(mu)
        h.profile_reset()
        for _ in range(10):
            h.schur_assemble(mu)
        ms, n = h.profile_get(capi.K_SCHUR)
        ms2, n2 = h.profile_get(capi.K_SCHUR_REDUCE)
        print(f"nC={n_cams:4d} nO={prob['nO']:7d} path={h.schur_path()} assemble {1e3 * ms / n:8.1f} us  reduce/finalize {1e3 * ms2 / n2:7.1f} us", flush=True)
        h.close()
