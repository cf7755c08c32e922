import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psba_amd
from psba_amd import synth
from oracle_lib import Oracle
prob = synth.venice_shaped()
o = Oracle(prob)
o.levmar(max_iter=10, tr_handoff=False)
h = psba_amd.Psba(0); h.upload_problem(prob); h.set_params(o.cams, o.pts)
lin = o.linearize()
for mu in [1e3, 10.0, 1.88, 0.5, 1e-2]:
    sch = o.schur(lin, mu)
    h.linearize(1.0, 1.0); h.update_UV(mu)
    S = h.compute_S(); ea = h.compute_ea()
    dS = np.abs(S - sch["S"]).max() / np.abs(sch["S"]).max()
    w = np.linalg.eigvalsh(sch["S"]); wg = np.linalg.eigvalsh(S)
    rc, dpa = h.SPDinv_matVec()
    ret, dp, _ = o.solve(lin, sch)
    want = np.linalg.solve(sch["S"], sch["eab"][:o.nA])
    print(f"mu {mu:8.2e}: S rel diff {dS:.2e}; eig min cpu {w[0]:.3e} gpu {wg[0]:.3e} max {w[-1]:.3e}; gpu rc {rc} cpu ret {ret}; "
          f"dpa gpu-vs-lapack {np.abs(dpa-want).max()/np.abs(want).max():.2e} cpu-vs-lapack {np.abs(dp[:o.nA]-want).max()/np.abs(want).max():.2e}")
    h.restore_UVdiag()
