"""Development helper (GPU box): the blocked chain with the look-ahead, the same factorization again and again --
dpa of every repeat against the first one's and against LAPACK on the same S.  usage: chol_race_check.py n_cams repeats"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import psba_amd
from psba_amd import synth

n_cams, reps = int(sys.argv[1]), int(sys.argv[2])
prob = synth.make_problem(n_cams=n_cams, n_pts=40000, mean_track=5.0, seed=7)
h = psba_amd.Psba(0)
h.upload_problem(prob)
h.linearize(1.0, 1.0)
mu = 1e-3 * h.max_diag()
nA = 6 * n_cams
n32 = (nA + 31) // 32 * 32
ref = None
bad = 0
for r in range(reps):
    h.linearize(1.0, 1.0)
    h.schur_assemble(mu)
    if ref is None:
        M = h.get_reduce_buffer().reshape(n32 + 1, n32)
        S = np.tril(M[:nA, :nA]); S = S + S.T - np.diag(np.diag(S))
        ref = np.linalg.solve(S, M[n32, :nA])
    h.schur_reduce(); h.schur_solve()
    sc = h.backsub(mu)
    dpa = h.get_dp()[:nA]
    err = np.abs(dpa - ref).max() / np.abs(ref).max()
    if err > 1e-8 or sc.status != 0:
        bad += 1
        print(f"repeat {r}: status {sc.status} rel err {err:.3e}", flush=True)
print(f"{n_cams} cameras: {reps} repeats, {bad} bad", flush=True)
