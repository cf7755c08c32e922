"""Development helper (GPU box): a handle's FIRST LM run after handles of other sizes have come and gone -- the final
cost must not depend on what ran before.  usage: lm_first_run_check.py repeats"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psba_amd
from psba_amd import synth
from sba_text import KK

DATA = os.path.join(ROOT, "tests", "golden", "data")
prob54 = psba_amd.read_problem(os.path.join(DATA, "54cams.txt"), os.path.join(DATA, "54pts.txt"), KK)
rng = np.random.default_rng(3)
c = collections.Counter()
for r in range(int(sys.argv[1])):
    n_cams = int(rng.choice([5, 12, 40, 130, 300]))
    other = synth.make_problem(n_cams=n_cams, n_pts=int(rng.integers(300, 3000)), mean_track=min(4.0, n_cams), seed=int(rng.integers(1 << 30)))
    a = psba_amd.Psba(0)
    a.upload_problem(other)
    a.levmar(max_iter=int(rng.integers(1, 4)), tr_handoff=False)
    a.close()
    h = psba_amd.Psba(0)
    h.upload_problem(prob54)
    res, log = h.levmar(max_iter=6, tr_handoff=False, log_cap=32)
    key = (res.iters, res.tries, f"{res.final_err:.10g}")
    if key not in c:
        print("run", r, "after", n_cams, "cameras:", key, flush=True)
        print(np.array2string(np.asarray(log)[: res.tries + 1], precision=6, max_line_width=200), flush=True)
    c[key] += 1
    h.close()
print(c)
