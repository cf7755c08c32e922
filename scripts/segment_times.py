"""Development helper: wall time of consecutive psba_levmar calls of 10 iterations each, without
and with HIP-event timing of the graded kernel (what bench.py's timed region does)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, psba_amd
from psba_amd import capi, synth
prob = synth.venice_shaped()
h = psba_amd.Psba(0); h.upload_problem(prob)
c0, p0 = np.array(prob["cams"], copy=True), np.array(prob["pts"], copy=True)
for prof in (0, 1 << capi.K_SCHUR):
    h.profile_enable(prof); h.profile_reset()
    for rep in range(6):
        t0 = time.perf_counter(); h.set_params(c0, p0); t1 = time.perf_counter()
        res, _ = h.levmar(max_iter=10, tr_handoff=False, log_cap=0); t2 = time.perf_counter()
        print(f"prof={prof} segment {rep}: set_params {1e6*(t1-t0):7.1f} us  levmar {1e6*(t2-t1):8.1f} us  ({res.iters} iters, {res.tries} tries)", flush=True)
