"""Development helper: K1 (linearize + cam_reduce, HIP events) under the ablation modes of PSBA_LIN_MODE
(1 no camera atomics, 2 no W store, 3 no per-point sums; timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth

prob = synth.venice_shaped()
h = psba_amd.Psba(0)
h.upload_problem(prob)
h.profile_enable(True)
for mode in sys.argv[1:] or ["0", "1", "2", "3"]:
    os.environ["PSBA_LIN_MODE"] = mode
    for _ in range(5):
        h.linearize(1.0, 1.0)
    h.profile_reset()
    for _ in range(30):
        h.linearize(1.0, 1.0)
    ms, n = h.profile_get(capi.K_LINEARIZE)
    print(f"PSBA_LIN_MODE={mode}: linearize + cam_reduce {1e3 * ms / n:8.1f} us", flush=True)
h.close()
