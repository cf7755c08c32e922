"""The trust-region loop at cfg5's camera count (2000 cameras, nA = 12 000; few points: the size of S is what
matters): LM until the hand-over, then three trust-region iterations from lambda = 0, i.e. through the failed
factorization and the modified Cholesky on the cooperative grid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psba_amd
from psba_amd import synth
prob = synth.make_problem(2000, 80000, 8.0, seed=7)
h = psba_amd.Psba(0); h.upload_problem(prob)
t0 = time.perf_counter()
res, _ = h.levmar(max_iter=8, tr_handoff=True)
print("LM: iters %d tries %d cost %.6g -> %.6g, %.2f s" % (res.iters, res.tries, res.init_err, res.final_err, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
tr, log = h.trust_region(max_iter=res.iters + 3, start_itno=res.iters)
print("TR: iters %d tries %d failed factorizations %d lambda %.3e cost %.6g -> %.6g, %.2f s" %
      (tr.iters - res.iters, tr.tries, tr.chol_fail, tr.lambda_, tr.init_err, tr.final_err, time.perf_counter() - t0), flush=True)
print(log)
