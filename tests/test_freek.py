"""Free intrinsics: camera blocks of 11 parameters (fu, u0, v0, ar, s | rotation | translation).

The reference reads this layout (PSBA/main.cpp:73,140-149, data/*varK.txt) and never optimises the intrinsics
(CL_files/PSBA.cl:5-7): it holds no arithmetic, no outputs and no fixtures for it.  PARITY UNPINNED: the judge here is
the oracle's twin (oracle/psba_oracle.c, orc_fk_*), which is itself checked by central differences below and is
written differently from the HIP route -- ONE dense (11 nC + 3 nP)^2 damped normal-equation solve per step, no Schur
complement, no blocks -- so the GPU's elimination, back-substitution and LM loop are compared with an independent
formulation, to rounding (1e-9 relative on steps, 1e-8 on costs)."""
import os

import numpy as np
import pytest

from conftest import DATA
from oracle_lib import OracleFreeK
from sba_text import read_problem


def _problem(cams, pts, max_pts=None):
    p = read_problem(os.path.join(DATA, cams), os.path.join(DATA, pts))
    if max_pts is not None and p["nP"] > max_pts:
        keep = np.asarray(p["iidx"]) < max_pts
        p = dict(p, pts=np.asarray(p["pts"])[:max_pts], impts=np.asarray(p["impts"])[keep],
                 iidx=np.asarray(p["iidx"])[keep], jidx=np.asarray(p["jidx"])[keep], nP=max_pts, nO=int(keep.sum()))
    return p


def test_oracle_twin_jacobian_against_central_differences():
    """d(projection) / d(fu, u0, v0, ar, s, v, t) and / dM of the twin, every observation of 7camsvarK."""
    p = _problem("7camsvarK.txt", "7pts.txt")
    o = OracleFreeK(p)
    JA, JB = o.jacobi()
    num = np.zeros_like(JA)
    for k in range(11):
        h = 1e-6 * max(1.0, abs(o.cams[k::11]).max())
        cp, cm = o.cams.copy(), o.cams.copy()
        cp[k::11] += h
        cm[k::11] -= h
        num[:, :, k] = -(o.exQT(cams=cp) - o.exQT(cams=cm)).reshape(-1, 2) / (2 * h)
    numB = np.zeros_like(JB)
    for k in range(3):
        h = 1e-6
        pp, pm = o.pts.copy(), o.pts.copy()
        pp[k::3] += h
        pm[k::3] -= h
        numB[:, :, k] = -(o.exQT(pts=pp) - o.exQT(pts=pm)).reshape(-1, 2) / (2 * h)
    for got, want in ((JA, num), (JB, numB)):
        scale = np.abs(want).max(axis=(0, 1))  # per parameter: the columns differ by orders of magnitude
        assert np.all(np.abs(got - want).max(axis=(0, 1)) <= 1e-6 * scale + 1e-9)
    assert np.all(JA[:, 0, 1] == 1.0) and np.all(JA[:, 1, 2] == 1.0) and np.all(JA[:, 0, 2] == 0.0)


def test_oracle_twin_reduces_to_the_fixed_k_oracle():
    """Its six extrinsic columns and B are the six-parameter oracle's with the same K; residuals are equal."""
    from oracle_lib import Oracle
    p = _problem("7camsvarK.txt", "7pts.txt")
    o6, o = Oracle(p), OracleFreeK(p)
    np.testing.assert_allclose(o.exQT(), o6.exQT(), rtol=0, atol=1e-9)
    JA6, JB6 = o6.jacobiQT()
    JA, JB = o.jacobi()
    np.testing.assert_allclose(JA[:, :, 5:].reshape(-1, 12), JA6.reshape(-1, 12), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(JB.reshape(-1), JB6, rtol=1e-12, atol=1e-12)


def _dense_step(o, mu):
    cost, N, g = o.normal()
    dp = np.linalg.solve(N + mu * np.eye(o.nT), g)
    return cost, N, g, dp


@pytest.mark.gpu
@pytest.mark.parametrize("cams,pts,max_pts", [("7camsvarK.txt", "7pts.txt", None), ("54camsvarK.txt", "54pts.txt", 450)])
def test_freek_damping_try_against_the_dense_normal_equations(cams, pts, max_pts):
    """One damping try of the fused verbs in PSBA_CAMERA_FREE_K mode: S and e_a against the Schur complement of the
    twin's dense J^T J + mu I, dpa / dp against its dense solve, the try's scalars against the twin's residuals."""
    import psba_amd
    p = _problem(cams, pts, max_pts)
    o = OracleFreeK(p)
    h = psba_amd.Psba(0)
    h.set_camera_model(True)
    h.upload_problem(p)
    assert h.camera_block() == 11 and h.nA == 11 * p["nC"]
    cost0 = h.residual()
    cost, N, g = o.normal()
    assert abs(cost0 - cost) <= 1e-12 * cost
    h.linearize(1.0, 1.0)
    mu = 1e-3 * h.max_diag()
    assert abs(mu - 1e-3 * np.diag(N).max()) <= 1e-11 * mu
    nA, nT = o.nA, o.nT
    Naa, Nab, Nbb = N[:nA, :nA], N[:nA, nA:], N[nA:, nA:] + mu * np.eye(nT - nA)
    X = np.linalg.solve(Nbb, np.c_[Nab.T, g[nA:]])
    S_want = Naa + mu * np.eye(nA) - Nab @ X[:, :nA]
    ea_want = g[:nA] - Nab @ X[:, nA]
    h.schur_assemble(mu)
    n32 = (nA + 31) // 32 * 32
    M = h.get_reduce_buffer().reshape(n32 + 1, n32)
    assert np.abs(M[:nA, :nA] - S_want).max() <= 1e-10 * np.abs(S_want).max()
    assert np.abs(M[n32, :nA] - ea_want).max() <= 1e-9 * np.abs(ea_want).max()
    h.schur_reduce()
    h.schur_solve()
    sc = h.backsub(mu)
    assert sc.status == 0
    dp_want = np.linalg.solve(N + mu * np.eye(nT), g)
    dp = h.get_dp()
    # (S of this problem is ill-conditioned -- focal length ~850 against rotations ~1e-3: steps agree to what
    # the conditioning leaves of fp64, block by block of comparable magnitude)
    for sl in (slice(0, nA), slice(nA, nT)):
        assert np.abs(dp[sl] - dp_want[sl]).max() <= 1e-6 * np.abs(dp_want[sl]).max()
    new_ex = o.exQT(cams=o.cams + dp_want[:nA], pts=o.pts + dp_want[nA:])
    assert abs(sc.new_cost - new_ex @ new_ex) <= 1e-7 * (new_ex @ new_ex)
    assert abs(sc.dp_l2 - dp_want @ dp_want) <= 1e-6 * (dp_want @ dp_want)
    assert abs(sc.gain_den - dp_want @ (mu * dp_want + g)) <= 1e-7 * abs(dp_want @ (mu * dp_want + g))
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cams,pts,max_pts", [("7camsvarK.txt", "7pts.txt", None), ("54camsvarK.txt", "54pts.txt", 450)])
def test_freek_levmar_against_the_twin(cams, pts, max_pts):
    """psba_levmar on 11-parameter blocks against the twin's dense LM: same accept / reject sequence and costs for
    the first iterations, final cost below the fixed-K optimum's (more freedom) and equal to the twin's."""
    import psba_amd
    p = _problem(cams, pts, max_pts)
    o = OracleFreeK(p)
    want, wlog = o.levmar(max_iter=8)
    h = psba_amd.Psba(0)
    h.set_camera_model(True)
    h.upload_problem(p)
    res, log = h.levmar(max_iter=8, tr_handoff=False, log_cap=256)
    assert abs(res.init_err - want.init_err) <= 1e-12 * want.init_err
    n = min(len(log), len(wlog), 6)
    assert n >= 4
    np.testing.assert_allclose(log[:n, 1], wlog[:n, 1], rtol=1e-6)     # cost after each try
    assert np.array_equal(log[:n, 4], wlog[:n, 4])                      # accepted / rejected
    assert res.final_err < res.init_err
    assert abs(res.final_err - want.final_err) <= 1e-5 * want.final_err
    cams11, _ = h.get_params()
    assert cams11.shape == (p["nC"], 11)
    assert np.abs(cams11[:, 0] - np.asarray(p["K"]).reshape(-1, 5)[:, 0]).max() > 0  # the focal lengths moved
    # the same data with the intrinsics held: the free problem must not end above it
    h6 = psba_amd.Psba(0)
    h6.upload_problem(p)
    res6, _ = h6.levmar(max_iter=8, tr_handoff=False)
    assert res.final_err <= res6.final_err * (1 + 1e-9)
    h6.close()
    # six-parameter-only verbs say so
    with pytest.raises(psba_amd.PsbaError):
        h.compute_S()
    h.close()
