"""Development helper (GPU box): the same short LM run on fresh handles, again and again -- the final costs must agree
to rounding.  usage: lm_repeat_check.py n_cams repeats [profile]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psba_amd
from psba_amd import synth

n_cams, reps = int(sys.argv[1]), int(sys.argv[2])
prof = len(sys.argv) > 3
prob = synth.make_problem(n_cams=n_cams, n_pts=40000, mean_track=5.0, seed=7)
costs = []
for r in range(reps):
    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    h.levmar(max_iter=2, tr_handoff=False, log_cap=0)
    h.reset_params()
    if prof:
        h.profile_enable(True)
        h.profile_reset()
    res, log = h.levmar(max_iter=4, tr_handoff=False, log_cap=16)
    costs.append(res.final_err)
    print(f"run {r}: iters {res.iters} tries {res.tries} final {res.final_err:.10g}", flush=True)
    h.close()
print("distinct:", sorted(set(f"{c:.8g}" for c in costs)))
