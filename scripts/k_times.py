"""Development helper: per-kernel-class times of N LM iterations on the venice-shaped set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth
prob = synth.venice_shaped()
h = psba_amd.Psba(0); h.upload_problem(prob)
h.levmar(max_iter=3, tr_handoff=False, log_cap=0); h.set_params(prob["cams"], prob["pts"])
h.profile_enable(True); h.profile_reset()
for rep in range(3):
    h.set_params(prob["cams"], prob["pts"]); h.levmar(max_iter=10, tr_handoff=False, log_cap=0)
print({n: round(1e3 * h.profile_get(k)[0] / max(h.profile_get(k)[1], 1), 2) for k, n in enumerate(capi.KERNEL_NAMES)})
