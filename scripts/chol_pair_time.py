"""Development helper: time of the Cholesky chain (schur_solve) on venice-shaped, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth
prob = synth.venice_shaped()
h = psba_amd.Psba(0)
h.upload_problem(prob)
h.linearize(1.0, 1.0)
mu = 1e-3 * h.max_diag()
h.profile_enable(True)
for rep in range(3):
    for _ in range(3):
        h.schur_assemble(mu); h.schur_reduce(); h.schur_solve()
    h.profile_reset()
    for _ in range(20):
        h.schur_assemble(mu); h.schur_reduce(); h.schur_solve()
    ms, n = h.profile_get(capi.K_CHOLESKY)
    print(f"cholesky chain {1e3 * ms / n:8.2f} us", flush=True)
