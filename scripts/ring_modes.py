"""Development helper: times K2's ring route (k_schur_ring + k_schur_sum) under its ablation modes
(PSBA_RING_MODE bits: 1 no products, 2 no DMA, 4 no wait for loads, 8 no Y preparation)."""
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
modes = sys.argv[1:] or ["0", "1", "2", "3", "6", "8", "15"]
for rep in range(2):
    for m in modes:
        os.environ["PSBA_RING_MODE"] = m
        for _ in range(3):
            h.schur_assemble(mu)
        h.profile_reset()
        for _ in range(20):
            h.schur_assemble(mu)
        ms, n = h.profile_get(capi.K_SCHUR)
        ms2, n2 = h.profile_get(capi.K_SCHUR_REDUCE)
        print(f"mode {m:>2}: ring {1e3 * ms / n:8.1f} us   sum {1e3 * ms2 / n2:6.1f} us", flush=True)
