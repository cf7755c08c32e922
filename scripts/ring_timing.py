"""Development helper: per-step s_memtime stamps of two workgroups of k_schur_ring (PSBA_RING_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PSBA_RING_TIMING"] = "1"
import psba_amd
from psba_amd import synth
prob = synth.venice_shaped()
h = psba_amd.Psba(0)
h.upload_problem(prob)
h.linearize(1.0, 1.0)
mu = 1e-3 * h.max_diag()
for _ in range(5):
    h.schur_assemble(mu)
os.environ["PSBA_RING_TIMING_DUMP"] = "1"
h.schur_assemble(mu)
