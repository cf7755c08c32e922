"""The N>1 path on CPU: world_size-2 gloo processes, each owning a contiguous range of 3-D
points (psba_partition_points through the product's C ABI), the per-rank math done by the CPU
oracle.  Checks the sharding rules the HIP library uses (psba_api.cpp / kernels_*.hip):
  * [S | ea] summed over ranks equals the single-rank [S | ea] when mu*I (and nothing else) is
    added on rank 0 only and U_j, g_a are per-rank partial sums;
  * max diag needs diag(U) summed over ranks before the max;
  * the try scalars (||dp||^2, gain denominator, new cost, ||p+dp||^2) are sums of rank-local
    point terms plus camera terms counted once (rank 0), with dpa . g_a_local summed over ranks;
  * an LM loop driven by those reduced scalars takes the same decisions on every rank and ends at
    the single-rank cost.
"""
import os
import socket

import numpy as np
import pytest

from oracle_lib import Oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _allreduce(dist, torch, arr, op="sum"):
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return t.numpy()


def _worker(rank, world, port, prob_name, out_q):
    import torch
    import torch.distributed as dist
    from conftest import ROOT  # noqa: F401
    import json
    from psba_amd import capi
    from sba_text import read_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_8c.json")))["problems"][prob_name]
    data = os.path.join(ROOT, "tests", "golden", "data")
    full = read_problem(os.path.join(data, g["cams"]), os.path.join(data, g["pts"]))
    shard = capi.shard_problem(full, world, rank)
    o = Oracle(shard)
    nA = o.nA
    res = {}

    def linearize():
        lin = o.linearize()
        # rank-local partial U, g_a ; V, g_b, W are point-local
        Ud = np.array([lin["U"][36 * (t // 6) + 7 * (t % 6)] for t in range(nA)])
        Ud = _allreduce(dist, torch, Ud)
        vmax = lin["UVdiag"][nA:].max()
        lin["maxdiag_global"] = float(_allreduce(dist, torch, np.array([max(Ud.max(), vmax)]), "max")[0])
        return lin

    def try_step(lin, mu):
        sch = o.schur(lin, mu)  # adds mu to the rank-local U and V
        S = sch["S"].copy()
        if rank != 0:  # mu*I on the camera block is added exactly once
            S[np.arange(nA), np.arange(nA)] -= mu
        red = _allreduce(dist, torch, np.r_[S.reshape(-1), sch["eab"][:nA]])
        S, ea = red[: nA * nA].reshape(nA, nA), red[nA * nA:]
        sch2 = dict(sch, S=S, eab=np.r_[ea, sch["eab"][nA:]])
        ret, dp, eab = o.solve(lin, sch2)  # replicated Cholesky, rank-local back-substitution
        dpa, dpb = dp[:nA], dp[nA:]
        newc, newp = o.cams + dpa, o.pts + dpb
        ex = o.exQT(cams=newc, pts=newp)
        cam_terms = 1.0 if rank == 0 else 0.0
        sc = np.array([
            dpb @ dpb + cam_terms * (dpa @ dpa),
            dpb @ (mu * dpb + lin["g"][nA:]) + dpa @ lin["g"][:nA] + cam_terms * mu * (dpa @ dpa),
            ex @ ex,
            newp @ newp + cam_terms * (newc @ newc),
        ])
        return ret, dp, S, ea, _allreduce(dist, torch, sc), newc, newp

    lin = linearize()
    mu0 = 1e-3 * lin["maxdiag_global"]
    ret, dp, S, ea, sc, newc, newp = try_step(lin, mu0)
    res.update(mu0=mu0, S=S, ea=ea, dpa=dp[:nA], sc=sc, ret=ret)

    # LM loop on reduced scalars (psba_amd/csrc/lm_loop.cpp restated), 8 iterations
    ex = o.exQT()
    ex_L2 = float(_allreduce(dist, torch, np.array([ex @ ex]))[0])
    mu, nu, p_L2 = mu0, 2, 1e3
    costs = []
    for itno in range(8):
        if itno > 0:
            lin = linearize()
        while True:
            ret, dp, S_, ea_, sc, newc, newp = try_step(lin, mu)
            if ret == 0.0:
                rho = (ex_L2 - sc[2]) / sc[1]
                if rho > 0:
                    tmp = 1.0 - (2 * rho - 1) ** 3
                    mu *= max(tmp, 1.0 / 3.0)
                    nu = 2
                    o.cams[:], o.pts[:] = newc, newp
                    p_L2, ex_L2 = sc[3], sc[2]
                    break
            mu *= nu
            nu *= 2
        costs.append(ex_L2)
    res["costs"] = costs
    if rank == 0:
        out_q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("prob_name", ["7cams"])
def test_two_rank_sharding_matches_single_rank(prob_name, golden, problems):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, prob_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-rank oracle
    o = Oracle(problems[prob_name])
    lin = o.linearize()
    mu0 = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu0)
    ret, dp, _ = o.solve(lin, sch)
    nA = o.nA
    assert abs(res["mu0"] - mu0) <= 1e-13 * mu0
    assert np.abs(res["S"] - sch["S"]).max() <= 1e-12 * np.abs(sch["S"]).max()
    assert np.abs(res["ea"] - sch["eab"][:nA]).max() <= 1e-11 * np.abs(sch["eab"][:nA]).max()
    assert np.abs(res["dpa"] - dp[:nA]).max() <= 1e-9 * np.abs(dp[:nA]).max()
    newp = np.r_[o.cams, o.pts] + dp
    ex = o.exQT(cams=newp[:nA], pts=newp[nA:])
    want = np.array([dp @ dp, dp @ (mu0 * dp + lin["g"]), ex @ ex, newp @ newp])
    assert np.all(np.abs(res["sc"] - want) <= 1e-9 * np.abs(want))
    # LM trajectory vs the goldens (first five) and vs the single-rank oracle (all eight)
    g = golden["problems"][prob_name]
    for k in range(5):
        assert abs(res["costs"][k] - g["err_after_itno"][k]) <= 1e-9 * g["err_after_itno"][k]
    ores, olog = Oracle(problems[prob_name]).levmar(max_iter=8)
    acc = olog[olog[:, 4] > 0][:, 1]
    assert np.all(np.abs(np.array(res["costs"]) - acc[:8]) <= 1e-9 * acc[:8])
