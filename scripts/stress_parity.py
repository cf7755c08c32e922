"""Development helper (GPU box): randomized shapes through the fused verbs against the oracle's S / e_a
and LAPACK on the oracle's S -- camera counts across every K2 grouping and every Cholesky chain.
usage: stress_parity.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psba_amd
from psba_amd import synth
from oracle_lib import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
h = psba_amd.Psba(0)
n = 0
worst = {"S": 0.0, "ea": 0.0, "dpa": 0.0, "dp": 0.0}
while time.time() - t0 < budget:
    n_cams = int(rng.choice([rng.integers(3, 20), rng.integers(20, 100), rng.integers(100, 260), rng.integers(260, 520), rng.integers(520, 760)]))
    n_pts = int(rng.integers(max(40, n_cams), 4000))
    track = float(rng.uniform(2.5, min(8.0, n_cams)))
    prob = synth.make_problem(n_cams=n_cams, n_pts=n_pts, mean_track=track, seed=int(rng.integers(1 << 30)))
    o = Oracle(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    os.environ.pop("PSBA_SCHUR_SPLIT", None)
    os.environ.pop("PSBA_SCHUR_BLOCK_GROUPS", None)
    os.environ.pop("PSBA_SCHUR_RUNS", None)
    if rng.random() < 0.3:  # round 4: the runs layout of K2's items, forced (row-aligned groups only)
        os.environ["PSBA_SCHUR_RUNS"] = "1"
    if rng.random() < 0.3:  # block-range groups (also where row-aligned ones would do), several slabs per group
        os.environ["PSBA_SCHUR_BLOCK_GROUPS"] = "1"
        os.environ["PSBA_SCHUR_SPLIT"] = str(int(rng.integers(1, 4)))
    h.upload_problem(prob)
    h.linearize(1.0, 1.0)
    h.schur_assemble(mu)
    n32 = (o.nA + 31) // 32 * 32
    M = h.get_reduce_buffer().reshape(n32 + 1, n32)
    S, ea = M[: o.nA, : o.nA], M[n32, : o.nA]
    e = {"S": np.abs(S - sch["S"]).max() / np.abs(sch["S"]).max(), "ea": np.abs(ea - sch["eab"][: o.nA]).max() / np.abs(sch["eab"][: o.nA]).max()}
    h.schur_reduce(); h.schur_solve()
    sc = h.backsub(mu)
    dp = h.get_dp()
    ref = np.linalg.solve(sch["S"], sch["eab"][: o.nA])
    e["dpa"] = np.abs(dp[: o.nA] - ref).max() / np.abs(ref).max()
    _, dpo, _ = o.solve(lin, sch) if o.nA <= 1600 else (None, None, None)
    if dpo is not None:
        e["dp"] = np.abs(dp - dpo).max() / np.abs(dpo).max()
    if n % 4 == 0 and n_cams <= 200:  # the LM loop, four iterations, against the oracle's trajectory
        h.upload_problem(prob)
        res, log = h.levmar(max_iter=4, tr_handoff=False)
        ores, olog = Oracle(prob).levmar(max_iter=4, tr_handoff=False)
        acc, oacc = log[log[:, 4] > 0], olog[olog[:, 4] > 0]
        m = min(len(acc), len(oacc))
        e["lm"] = float(np.abs(acc[:m, 1] - oacc[:m, 1]).max() / ores.init_err) if m else 0.0
        if len(acc) != len(oacc) or res.tries != ores.tries:
            e["lm"] = max(e["lm"], 1.0)
        worst["lm"] = max(worst.get("lm", 0.0), e["lm"])
    bad = e.get("lm", 0.0) > 1e-8 or e["S"] > 1e-11 or e["ea"] > 1e-10 or e["dpa"] > 1e-7 or e.get("dp", 0) > 1e-7 or (sc.status != 0)
    for k, v in e.items():
        worst[k] = max(worst.get(k, 0.0), v)
    n += 1
    if bad or n % 10 == 0:
        print(f"{'BAD ' if bad else ''}case {n}: nC={n_cams} nP={n_pts} nO={prob['nO']} path={h.schur_path()} status={sc.status} " + " ".join(f"{k}={v:.2e}" for k, v in e.items()), flush=True)
    if bad:
        sys.exit(1)
print(f"{n} cases ok in {time.time() - t0:.0f} s; worst " + " ".join(f"{k}={v:.2e}" for k, v in worst.items()))
