"""Development helper (GPU box): K2 (assembly + reduce, one HIP-event span) at several camera counts under
PSBA_SCHUR_WINDOW = 2 / 3.  usage: k2_window_sizes.py n_cams [n_cams ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psba_amd
from psba_amd import capi, synth

for n_cams in [int(a) for a in sys.argv[1:]]:
    prob = synth.make_problem(n_cams=n_cams, n_pts=60000, mean_track=5.4, seed=11)
    for w in ("2", "3", "2", "3"):
        os.environ["PSBA_SCHUR_WINDOW"] = w
        h = psba_amd.Psba(0)
        h.upload_problem(prob)
        h.levmar(max_iter=3, tr_handoff=False, log_cap=0)
        h.reset_params()
        h.profile_enable(1 << capi.K_SCHUR)
        h.profile_reset()
        res, _ = h.levmar(max_iter=8, tr_handoff=False, log_cap=0)
        ms, n = h.profile_get(capi.K_SCHUR)
        print(f"{n_cams} cameras window {w}: pair {1e3 * ms / max(n, 1):9.2f} us ({n} launches), final cost {res.final_err:.8g}", flush=True)
        h.close()
