"""Modified Cholesky (psba_cholmod_lambda) at cfg5's size: 2000 cameras, nA = 12 000, S at lambda = 0 of a
synthetic problem (few points: only the size of S matters), on the cooperative grid.  The one-workgroup kernel
needs minutes there and is not run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import synth
n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
prob = synth.make_problem(n_cams, 40 * n_cams, 8.0, seed=7)
h = psba_amd.Psba(0); h.upload_problem(prob); h.linearize(2.0, -2.0)
for rep in range(2):
    t0 = time.perf_counter()
    lam, info = h.cholmod_lambda()
    print("n =", 6 * n_cams, "lambda", lam, "delta/beta/one-column block columns", list(info),
          "%.1f ms (assembly included)" % (1e3 * (time.perf_counter() - t0)), flush=True)
