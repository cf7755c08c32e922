"""Development helper: K1 (linearize + cam_reduce) time over the number of persistent workgroups
(PSBA_LIN_GRID, read at upload)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth

prob = synth.venice_shaped()
for grid in sys.argv[1:] or ["256", "512", "768", "1024", "1536"]:
    os.environ["PSBA_LIN_GRID"] = grid
    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    h.profile_enable(True)
    for _ in range(5):
        h.linearize(1.0, 1.0)
    h.profile_reset()
    for _ in range(20):
        h.linearize(1.0, 1.0)
    ms, n = h.profile_get(capi.K_LINEARIZE)
    print(f"PSBA_LIN_GRID={grid}: linearize + cam_reduce {1e3 * ms / n:8.1f} us", flush=True)
    h.close()
