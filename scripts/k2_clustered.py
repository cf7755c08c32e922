"""Development helper: the graded K2 pair (HIP-event span over k_schur_lds + k_schur_reduce) on venice-shaped with
clustered tracks (runs of `cluster` consecutive points sharing one camera set), against the uniform draw."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import capi, synth
for cluster in [int(x) for x in (sys.argv[1:] or ["1", "4", "16", "64"])]:
    prob = synth.venice_shaped(cluster=cluster)
    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    h.linearize(1.0, 1.0)
    mu = 1e-3 * h.max_diag()
    h.profile_enable(1 << capi.K_SCHUR)
    for _ in range(5):
        h.schur_assemble(mu)
    h.profile_reset()
    for _ in range(30):
        h.schur_assemble(mu)
    ms, n = h.profile_get(capi.K_SCHUR)
    b = h.algorithmic_bytes(capi.K_SCHUR)
    print(f"cluster {cluster:3d}: {prob['nO']} observations, pair {1e3 * ms / n:7.2f} us, {b / (ms / n * 1e-3) / 1e9 / 8000:.3f} of the HBM roofline", flush=True)
    h.close()
