"""Development helper (GPU box): the sequence of tests/test_gpu_trust_region.py on ONE re-used handle, again and
again -- LM until the hand-over, then the trust-region loop from a given damping, on 7cams / 54cams / trafalgar21 in
turn.  Every repetition's LM and TR logs must equal the first one's to rounding; a repetition that differs is printed
in full next to the first (the try whose gain ratio or status differs is the thing to look at).
usage: tr_repeat_check.py repeats"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import psba_amd
from sba_text import read_problem

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
with open(os.path.join(ROOT, "tests", "golden", "survey_8c.json")) as f:
    golden = json.load(f)
DATA = os.path.join(ROOT, "tests", "golden", "data")
names = ["7cams", "54cams", "trafalgar21"]
probs = {n: read_problem(os.path.join(DATA, golden["problems"][n]["cams"]), os.path.join(DATA, golden["problems"][n]["pts"]))
         for n in names}
h = psba_amd.Psba(0)
first, odd = {}, 0
np.set_printoptions(linewidth=200, precision=12)
for r in range(reps):
    for n in names:
        h.upload_problem(probs[n])
        h.linearize(2.0, -2.0)
        lam = 1e-6 * h.max_diag()
        h.upload_problem(probs[n])
        res, lmlog = h.levmar(max_iter=50, tr_handoff=True)
        tr, trlog = h.trust_region(start_itno=res.iters, init_lambda=lam)
        key = (res.iters, res.tries)
        if n not in first:
            first[n] = (key, lmlog, trlog)
            print(f"{n}: LM iters {res.iters} tries {res.tries} final {res.final_err:.12g}; TR iters {tr.iters} final {tr.final_err:.12g}", flush=True)
            continue
        k0, lm0, tr0 = first[n]
        same = key == k0 and lmlog.shape == lm0.shape and trlog.shape == tr0.shape and \
            np.allclose(np.nan_to_num(lmlog), np.nan_to_num(lm0), rtol=1e-6, atol=0) and \
            np.allclose(np.nan_to_num(trlog[:6]), np.nan_to_num(tr0[:6]), rtol=1e-6, atol=0)
        if not same:
            odd += 1
            print(f"ODD repetition {r} {n}: {key} against {k0}")
            print("LM log (itno, cost, rho, mu, accepted) of this run:\n", lmlog)
            print("LM log of the first run:\n", lm0)
            print("TR log of this run:\n", trlog[:8])
            print("TR log of the first run:\n", tr0[:8], flush=True)
    if r % 20 == 19:
        print(f"{r + 1} repetitions, {odd} odd", flush=True)
print(f"done: {reps} repetitions, {odd} odd")
h.close()
