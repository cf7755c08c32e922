"""Development helper: LM iterations on the venice-shaped problem with a one-rank communicator (the
RCCL path: packed all-reduce, expand, scalar all-reduce), main stream only vs side stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import synth
prob = synth.venice_shaped()
for rep in range(2):
    for one in (True, False):
        if one:
            os.environ.pop("PSBA_COMM_SIDE_STREAM", None)
        else:
            os.environ["PSBA_COMM_SIDE_STREAM"] = "1"
        h = psba_amd.Psba(0)
        h.comm_init(1, 0, psba_amd.Psba.comm_unique_id())
        h.upload_problem(prob)
        h.levmar(max_iter=20, tr_handoff=False)
        ts = []
        for _ in range(5):
            h.reset_params()
            t0 = time.perf_counter()
            res, _ = h.levmar(max_iter=40, tr_handoff=False)
            ts.append((time.perf_counter() - t0) / res.iters)
        print(f"one_stream={one}: {1e3 * min(ts):.4f} ms/iter (min of 5 x 40 iterations), final {res.final_err:.9e}", flush=True)
        h.close()
