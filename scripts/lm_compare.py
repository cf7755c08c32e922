"""Development helper: per-try LM log of the GPU path next to the CPU oracle's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psba_amd
from psba_amd import synth
from oracle_lib import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prob = synth.venice_shaped()
h = psba_amd.Psba(0); h.upload_problem(prob)
res, log = h.levmar(max_iter=n, tr_handoff=False)
ores, olog = Oracle(prob).levmar(max_iter=n, tr_handoff=False)
for a, b in zip(log, olog):
    print(f"it {int(a[0]):2d} gpu cost {a[1]:.15e} rho {a[2]:.12f} mu {a[3]:.9e} | cpu cost {b[1]:.15e} rho {b[2]:.12f} mu {b[3]:.9e} | rel {abs(a[1]-b[1])/b[1]:.2e}")
