"""The oracle's trust-region twin (oracle/psba_oracle.c: orc_trust_region, orc_cholmod,
orc_compute_Jmultiply; reference PSBA/trust_region.cpp, PSBA/cl_cholmod.cpp,
CL_files/cholmod_blk.cl, CL_files/compute_Jmultiply.cl).  The reference holds no vectors for
this path and SURVEY 8(c) gives none beyond "the reference would return ITER_TURN_TO_TR after
itno = 4": parity of the TR path is UNPINNED beyond that; these tests check the mathematics the
twin restates (CPU only)."""
import numpy as np
import pytest

from oracle_lib import Oracle, cholmod, solve_like_main, trust_region


def test_jmultiply_is_the_jacobian_product(problems):
    import ctypes as C
    from oracle_lib import _jmul
    o = Oracle(problems["7cams"])
    lin = o.linearize()
    rng = np.random.default_rng(2)
    x = rng.normal(size=o.nT)
    out = np.empty(2 * o.nO)
    _jmul(o.nC, o.nO, lin["JA"], lin["JB"], o.iidx, o.jidx, x, out)
    JA, JB = lin["JA"].reshape(-1, 2, 6), lin["JB"].reshape(-1, 2, 3)
    want = np.einsum("akc,ac->ak", JA, x[: o.nA].reshape(-1, 6)[o.jidx]) + \
        np.einsum("akc,ac->ak", JB, x[o.nA:].reshape(-1, 3)[o.iidx])
    np.testing.assert_allclose(out.reshape(-1, 2), want, rtol=1e-12, atol=1e-9)
    # g^T B g = 2 ||J g||^2 with B = 2 J^T J (trust_region.cpp:125-126)
    g = -2 * lin["g"]
    _jmul(o.nC, o.nO, lin["JA"], lin["JB"], o.iidx, o.jidx, g, out)
    J = np.zeros((2 * o.nO, o.nT))
    for a in range(o.nO):
        J[2 * a: 2 * a + 2, 6 * o.jidx[a]: 6 * o.jidx[a] + 6] = JA[a]
        J[2 * a: 2 * a + 2, o.nA + 3 * o.iidx[a]: o.nA + 3 * o.iidx[a] + 3] = JB[a]
    assert abs(2 * out @ out - g @ (2 * J.T @ J) @ g) <= 1e-10 * abs(2 * out @ out)


@pytest.mark.parametrize("shift", [0.0, -3.0, -40.0])
def test_modified_cholesky_properties(shift):
    """L L^T = A + diag(E): off-diagonals untouched, E >= 0 up to delta, L L^T positive definite;
    for a positive definite A it is the plain Cholesky factor (E ~ 0)."""
    rng = np.random.default_rng(int(-shift) + 1)
    n = 30
    B = rng.normal(size=(n, n))
    A = B @ B.T + shift * np.eye(n)
    L, E, delta, beta = cholmod(A)
    assert np.all(np.triu(L, 1) == 0)
    R = L @ L.T - A
    scale = np.abs(A).max()
    assert np.abs(R - np.diag(np.diag(R))).max() <= 1e-12 * scale
    np.testing.assert_allclose(np.diag(R), E, rtol=0, atol=1e-12 * scale)
    assert np.linalg.eigvalsh(L @ L.T).min() > 0
    xi = np.abs(A - np.diag(np.diag(A))).max()
    gamma = np.abs(np.diag(A)).max()
    assert abs(delta - 1e-15 * max(xi + gamma, 1)) <= 1e-30
    assert abs(beta - np.sqrt(max(gamma, 1e-15, xi / np.sqrt(n * n - 1)))) <= 1e-12 * beta
    if shift == 0.0:
        np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-9, atol=1e-12)
        assert np.abs(E).max() <= 1e-10 * scale
    else:
        assert E.sum() > 0


@pytest.mark.parametrize("name", ["7cams", "54cams", "trafalgar21"])
def test_lm_tr_alternation_like_main(name, golden, problems):
    """PSBA/main.cpp:193-208 on the bundled problems: LM hands over after itno = 4,
    TR continues from there, the cost never increases, and the alternation ends at (about) the
    cost LM alone reaches."""
    g = golden["problems"][name]
    o = Oracle(problems[name])
    seq = solve_like_main(o)
    assert seq[0][0] == "lm" and seq[0][1].flag == 2 and seq[0][1].iters == 5
    assert abs(seq[0][1].final_err - g["err_after_itno"][4]) <= 1e-10 * g["err_after_itno"][4]
    assert seq[1][0] == "tr" and seq[1][1].init_err == seq[0][1].final_err
    costs = [seq[0][1].init_err] + [r.final_err for _, r in seq]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))
    assert seq[-1][1].iters <= 50
    ex = o.exQT()
    assert abs(ex @ ex - seq[-1][1].final_err) <= 1e-9 * seq[-1][1].final_err
    assert seq[-1][1].final_err <= 1.05 * g["final_err"]


def test_trust_region_step_log(problems):
    o = Oracle(problems["54cams"])
    res, _ = o.levmar(max_iter=50, tr_handoff=True)
    tr, log = trust_region(o, start_itno=res.iters)
    assert tr.n_log == len(log) and tr.tries in (len(log), len(log) + 1)  # a step that ends on ITER_DP_NO_CHANGE is not logged
    acc = log[log[:, 5] > 0]
    assert len(acc) >= 1 and np.all(np.diff(np.r_[tr.init_err, acc[:, 1]]) < 0)
    # rejected steps shrink the region by 4, well predicted ones double it (trust_region.cpp:223-238)
    assert np.all((log[:, 2] >= 0.25) | (log[:, 5] == 0))
