"""Development helper: times K1 (linearize) and K3 (backsub) under their ablation modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth

prob = synth.venice_shaped()
h = psba_amd.Psba(0)
h.upload_problem(prob)
h.profile_enable(True)
for var, kern, modes in (("PSBA_LIN_MODE", capi.K_LINEARIZE, "0123"), ("PSBA_BACK_MODE", capi.K_BACKSUB, "012")):
    for m in modes:
        os.environ[var] = m
        h.linearize(1.0, 1.0)
        mu = 1e-3 * h.max_diag()
        for rep in range(3):
            h.profile_reset()
            for _ in range(10):
                if kern == capi.K_LINEARIZE:
                    h.linearize(1.0, 1.0)
                else:
                    h.schur_assemble(mu); h.schur_reduce(); h.schur_solve(); h.backsub(mu)
            ms, n = h.profile_get(kern)
        print(f"{var}={m}: {1e3 * ms / n:8.1f} us", flush=True)
    os.environ[var] = "0"
