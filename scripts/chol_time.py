import os, sys
os.environ["PSBA_CHOL_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import synth
prob = synth.venice_shaped(n_pts=8000)
h = psba_amd.Psba(0); h.upload_problem(prob); h.linearize(1.0, 1.0); mu = 1e-3 * h.max_diag()
for _ in range(3):
    h.schur_assemble(mu); h.schur_solve()
