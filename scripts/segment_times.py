"""Development helper: wall time of many consecutive psba_levmar calls of 10 iterations each
(restarting from the uploaded parameters): shows the clock step-up stall(s) of a fresh process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, psba_amd
from psba_amd import capi, synth
prob = synth.venice_shaped()
h = psba_amd.Psba(0); h.upload_problem(prob)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ts = []
t00 = time.perf_counter()
for rep in range(n):
    t0 = time.perf_counter(); h.reset_params()
    res, _ = h.levmar(max_iter=10, tr_handoff=False, log_cap=0); t1 = time.perf_counter()
    ts.append((1e3 * (t0 - t00), 1e6 * (t1 - t0)))
slow = [(round(a, 1), round(b)) for a, b in ts if b > 2050]
print("segments:", n, " median us:", round(float(np.median([b for _, b in ts]))), " slow ones (start ms, us):", slow)
