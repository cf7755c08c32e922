"""The trust-region caller and its operators on the GPU against the oracle's twin
(tests/test_trust_region_oracle.py says what pins the twin: nothing beyond the hand-off point).
Tolerances: J x and g 1e-11 of the largest magnitude; lambda of the modified Cholesky 1e-6
relative (S at lambda = 0 is singular up to rounding, the correction is a sum of quantities at
rounding level times max|S|... see the test); TR costs 1e-8 relative over the accepted steps."""
import numpy as np
import pytest

from oracle_lib import Oracle, cholmod, solve_like_main, trust_region
from test_gpu_parity import close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import psba_amd
    h = psba_amd.Psba(0)
    yield h
    h.close()


@pytest.mark.parametrize("name", ["7cams", "trafalgar21"])
def test_jmultiply_and_gradient(name, problems, gpu):
    import ctypes as C
    from oracle_lib import _jmul
    prob = problems[name]
    o = Oracle(prob)
    gpu.upload_problem(prob)
    gpu.linearize(2.0, -2.0)
    lin = o.linearize(2.0, -2.0)
    g = gpu.get_gradient()
    close(g, lin["g"], 1e-11, "g (coeff -2)")
    rng = np.random.default_rng(4)
    x = rng.normal(size=o.nT)
    want = np.empty(2 * o.nO)
    _jmul(o.nC, o.nO, lin["JA"], lin["JB"], o.iidx, o.jidx, x, want)
    close(gpu.compute_Jmultiply(x), want, 1e-11, "J x")
    y = rng.normal(size=o.nT)
    wy = np.empty(2 * o.nO)
    _jmul(o.nC, o.nO, lin["JA"], lin["JB"], o.iidx, o.jidx, y, wy)
    d = gpu.jmul_dots(x, y)
    np.testing.assert_allclose(d, [want @ want, want @ wy, wy @ wy], rtol=1e-11)
    np.testing.assert_allclose(gpu.jmul_dots(x)[0], want @ want, rtol=1e-11)


@pytest.mark.parametrize("shift", [-3.0, -40.0, 0.0])
def test_modified_cholesky_on_a_chosen_matrix(shift, problems, gpu):
    """k_cholmod against the oracle's twin on the same matrix (put into the reduce buffer with
    psba_set_reduce_buffer): an indefinite B B^T + shift I whose pivots stay away from zero, so the
    correction is not decided by rounding noise.  7cams supplies the size (42 x 42, padded to 64)."""
    prob = problems["7cams"]
    gpu.upload_problem(prob)
    n, n32 = 42, 64
    rng = np.random.default_rng(int(-shift) + 11)
    B = rng.normal(size=(n, n))
    A = B @ B.T + shift * np.eye(n)
    buf = np.zeros((n32 + 1, n32))
    buf[:n, :n] = A
    buf[n:n32, n:] = np.eye(n32 - n)
    gpu.linearize(1.0, 1.0)
    gpu.schur_assemble(1.0)  # any assembly: makes the reduce buffer the current state
    gpu.set_reduce_buffer(buf.reshape(-1))
    lam, info = gpu.cholmod_lambda(reassemble=False)
    L, E, delta, beta = cholmod(A)
    assert abs(info[0] - delta) <= 1e-12 * delta and abs(info[1] - beta) <= 1e-12 * beta
    want = abs(E.sum()) / n
    assert abs(lam - want) <= 1e-9 * want + 1e-13 * np.abs(A).max()  # shift 0: E is rounding noise on both sides
    if shift < 0:
        assert lam > 0


@pytest.mark.parametrize("shift", [-3.0, -400.0, 0.0])
def test_modified_cholesky_on_the_cooperative_grid(shift, problems, gpu, monkeypatch):
    """k_cholmod_grid (the route of large S, forced here with PSBA_CHOLMOD_GRID=1) against the one-workgroup
    kernel and the oracle's twin on one 324 x 324 matrix (54cams supplies the size): every sum is formed by one
    thread in the same order on both routes, so delta, beta and the count of block columns on the one-column
    route are equal and lambda differs by the summation order of the E_i only."""
    prob = problems["54cams"]
    gpu.upload_problem(prob)
    n = 6 * int(prob["nC"])
    n32 = gpu.chol_dist_shape()[0]
    rng = np.random.default_rng(int(-shift) + 5)
    B = rng.normal(size=(n, n))
    A = B @ B.T + shift * np.eye(n)
    buf = np.zeros((n32 + 1, n32))
    buf[:n, :n] = A
    buf[n:n32, n:] = np.eye(n32 - n)
    got = {}
    for route in ("0", "1"):
        monkeypatch.setenv("PSBA_CHOLMOD_GRID", route)
        gpu.linearize(1.0, 1.0)
        gpu.schur_assemble(1.0)
        gpu.set_reduce_buffer(buf.reshape(-1))
        got[route] = gpu.cholmod_lambda(reassemble=False)
    (lam0, info0), (lam1, info1) = got["0"], got["1"]
    assert info0[0] == info1[0] and info0[1] == info1[1] and info0[2] == info1[2]
    L, E, delta, beta = cholmod(A)
    want = abs(E.sum()) / n
    scale = 1e-9 * want + 1e-13 * np.abs(A).max()
    assert abs(lam1 - lam0) <= 1e-3 * scale + 1e-12 * abs(lam0)
    assert abs(lam1 - want) <= scale
    if shift < 0:
        assert lam1 > 0 and info1[2] > 0  # the one-column route was taken


def test_modified_cholesky_lambda_makes_S_positive_definite(problems, gpu):
    """S at lambda = 0 of 7cams is not positive definite (no gauge is fixed): the reference then
    estimates lambda from a modified Cholesky (trust_region.cpp:341-363).  The singular directions
    have pivots at rounding level, so the value itself is decided by rounding noise (it is not
    compared); what must hold is that the damped S then factors."""
    prob = problems["7cams"]
    gpu.upload_problem(prob)
    gpu.linearize(2.0, -2.0)
    gpu.schur_assemble(0.0); gpu.schur_reduce(); gpu.schur_solve()
    assert gpu.backsub(0.0).status & 1
    lam, info = gpu.cholmod_lambda()
    assert np.isfinite(lam) and lam > 0
    for _ in range(60):  # compute_PB doubles lambda until the factorization goes through (:365-368)
        gpu.schur_assemble(lam); gpu.schur_reduce(); gpu.schur_solve()
        if not (gpu.backsub(lam).status & 1):
            break
        lam *= 2
    else:
        raise AssertionError("S never became positive definite")


@pytest.mark.parametrize("name", ["7cams", "54cams", "trafalgar21"])
def test_trust_region_against_the_oracle(name, problems, gpu):
    """levmar() until it hands over, then trust_region(): step-by-step costs against the oracle."""
    prob = problems[name]
    o = Oracle(prob)
    ores, _ = o.levmar(max_iter=50, tr_handoff=True)
    otr, olog = trust_region(o, start_itno=ores.iters)
    gpu.upload_problem(prob)
    res, _ = gpu.levmar(max_iter=50, tr_handoff=True)
    assert res.flag == 2 and res.iters == ores.iters
    tr, log = gpu.trust_region(start_itno=res.iters)
    assert abs(tr.init_err - otr.init_err) <= 1e-10 * otr.init_err
    acc, oacc = log[log[:, 5] > 0], olog[olog[:, 5] > 0]
    n = min(len(acc), len(oacc), 4)
    assert n >= 1
    # with a failed factorization at lambda = 0 the damping comes from the modified Cholesky of a
    # singular S, whose value rounding noise decides (see the test above): the paths then agree to
    # a few digits only and meet again at the end
    tight = tr.chol_fail == 0 and otr.chol_fail == 0
    np.testing.assert_allclose(acc[:n, 1], oacc[:n, 1], rtol=1e-7 if tight else 2e-2)
    assert tr.final_err < tr.init_err
    assert abs(tr.final_err - otr.final_err) <= 1e-4 * otr.final_err
    cams, pts = gpu.get_params()
    ex = Oracle(prob).exQT(cams=cams, pts=pts)
    assert abs(ex @ ex - tr.final_err) <= 1e-9 * tr.final_err


@pytest.mark.parametrize("name", ["7cams", "54cams", "trafalgar21"])
def test_trust_region_steps_at_a_given_damping(name, problems, gpu):
    """The loop as the reference runs it starts at lambda = 0, where S (no gauge fixed) is singular
    to rounding: the first factorization fails on all three data sets (chol_fail = 1 in the test
    above) and the damping that follows is whatever the modified Cholesky makes of rounding noise --
    a lambda some 1e-17 of S's scale, with which the Gauss-Newton step is decided by the gauge
    directions' noise; hence the loose per-step tolerance there.  Started instead from a GIVEN,
    well-conditioned damping (1e-6 max diag B), both sides factor S + lambda I at once, take the
    same lambda sequence, and the accepted steps must agree closely.  Parity of the trust-region path
    remains unpinned (the reference holds no vectors for it); this holds the HIP path to the oracle's
    twin."""
    prob = problems[name]
    o = Oracle(prob)
    ores, _ = o.levmar(max_iter=50, tr_handoff=True)
    lam = 1e-6 * o.linearize(2.0, -2.0)["maxdiag"]
    otr, olog = trust_region(o, start_itno=ores.iters, init_lambda=lam)
    gpu.upload_problem(prob)
    res, _ = gpu.levmar(max_iter=50, tr_handoff=True)
    tr, log = gpu.trust_region(start_itno=res.iters, init_lambda=lam)
    acc, oacc = log[log[:, 5] > 0], olog[olog[:, 5] > 0]
    # (shown when the test fails: the try whose gain ratio or status differs is the thing to look at)
    print("LM here / oracle: iters", res.iters, ores.iters, "tries", res.tries, ores.tries, "final", res.final_err, ores.final_err)
    print("TR log here (itno, cost, rho, delta, lambda, accepted):\n", log[:10], "\nTR log of the oracle:\n", olog[:10])
    # the steps before the loop resets lambda to 0 (ten good steps in a row, trust_region.cpp:266-271;
    # the factorization after that fails again and the paths part as in the test above)
    n = min(len(acc), len(oacc), 6, int(np.argmax(oacc[:, 4] != lam)) if (oacc[:, 4] != lam).any() else len(oacc))
    assert n >= 3
    np.testing.assert_allclose(acc[:n, 1], oacc[:n, 1], rtol=1e-7)
    assert np.all(acc[:n, 4] == lam)  # the branch: every factorization at the given damping succeeded
    assert abs(tr.final_err - otr.final_err) <= 1e-4 * otr.final_err


def test_trust_region_with_a_single_rank_communicator(problems):
    """The sharded form of the loop -- inner products as (camera part) + sum over ranks of (point
    part), J x dot products and g_a summed over the ranks, the modified Cholesky replicated on the
    all-reduced S -- must give the plain loop's results when the communicator has one rank (RCCL with
    more ranks has not run on this builder's single GPU: unmeasured on hardware)."""
    import psba_amd
    prob = problems["54cams"]
    ref = psba_amd.Psba(0)
    ref.upload_problem(prob)
    want = ref.solve(max_iter=30)
    ref.close()
    h = psba_amd.Psba(0)
    h.comm_init(1, 0, psba_amd.Psba.comm_unique_id())
    h.upload_problem(prob)
    got = h.solve(max_iter=30)
    h.close()
    assert (got.lm_calls, got.tr_calls, got.iters) == (want.lm_calls, want.tr_calls, want.iters)
    assert abs(got.final_err - want.final_err) <= 1e-9 * want.final_err


@pytest.mark.parametrize("name", ["7cams", "54cams"])
def test_solve_alternates_like_main(name, golden, problems, gpu):
    """psba_solve = PSBA/main.cpp:193-208; final cost against the oracle's alternation and
    against the LM-only golden."""
    prob = problems[name]
    o = Oracle(prob)
    seq = solve_like_main(o)
    gpu.upload_problem(prob)
    res = gpu.solve(max_iter=50)
    assert res.lm_calls >= 1 and res.tr_calls >= 1 and res.iters <= 50
    assert abs(res.final_err - seq[-1][1].final_err) <= 1e-4 * seq[-1][1].final_err
    assert res.final_err <= 1.05 * golden["problems"][name]["final_err"]
    cams, pts = gpu.get_params()
    ex = Oracle(prob).exQT(cams=cams, pts=pts)
    assert abs(ex @ ex - res.final_err) <= 1e-9 * res.final_err


def test_sparse_mode_damping_estimate(problems):
    """PSBA_SOLVER_PCG: psba_cholmod_lambda has no dense S to factor; it returns the Gershgorin shift of the
    stored blocks (round 4).  (The gauge-singular S of lambda = 0 does not stop the conjugate gradients -- they
    solve a consistent semidefinite system -- so the indefinite S here is made with a negative damping.)  The
    shift equals the one computed from the blocks on the host, and S + lambda I solves."""
    import psba_amd
    prob = problems["54cams"]
    h = psba_amd.Psba(0)
    h.set_solver(1, 1e-12, 4000)
    h.upload_problem(prob)
    h.linearize(2.0, -2.0)
    c = 0.05 * h.max_diag()
    h.schur_assemble(-c); h.schur_reduce(); h.schur_solve()
    assert h.backsub(-c).status & 1
    h.linearize(2.0, -2.0)
    h.schur_assemble(-c); h.schur_reduce()
    jk, val, _ = h.get_sparse_S()
    nA = 6 * prob["nC"]
    S = np.zeros((nA, nA))
    for (j, k), B in zip(jk, val):
        S[6 * j:6 * j + 6, 6 * k:6 * k + 6] = B
        if j != k:
            S[6 * k:6 * k + 6, 6 * j:6 * j + 6] = B.T
    S = np.tril(S) + np.tril(S, -1).T  # (a diagonal block is read through its lower triangle)
    assert np.linalg.eigvalsh(S).min() < 0
    margin = np.diag(S) - (np.abs(S).sum(1) - np.abs(np.diag(S)))
    lam, info = h.cholmod_lambda(reassemble=False)  # the blocks just assembled
    assert abs(info[0] - margin.min()) <= 1e-12 * np.abs(S).max()
    assert lam > 0 and abs(lam + margin.min()) <= 1e-12 * np.abs(S).max()
    assert np.linalg.eigvalsh(S + lam * np.eye(nA)).min() > 0
    h.linearize(2.0, -2.0)
    h.schur_assemble(lam - c); h.schur_reduce(); h.schur_solve()
    assert h.backsub(lam - c).status == 0
    h.close()


@pytest.mark.parametrize("name", ["7cams", "54cams"])
def test_solve_alternates_in_sparse_mode(name, golden, problems):
    """psba_solve (LM <-> TR, PSBA/main.cpp:193-208) on the block-sparse S + conjugate gradients: both loops run and
    the run ends where the dense mode's does.  The sparse trust-region loop never solves undamped (the conjugate
    gradients crawl on the gauge-singular S of lambda = 0 and the step then depends on the rounding of their sums:
    11 to 50 iterations for the same problem, run by run, before this was so); it starts at 1e-8 of the largest
    diagonal entry, where the reference ends up after its failed factorization."""
    import psba_amd
    prob = problems[name]
    ref = psba_amd.Psba(0)
    ref.upload_problem(prob)
    want = ref.solve(max_iter=50)
    ref.close()
    h = psba_amd.Psba(0)
    h.set_solver(1, 1e-12, 4000)
    h.upload_problem(prob)
    res = h.solve(max_iter=50)
    assert res.lm_calls >= 1 and res.tr_calls >= 1 and res.iters <= 50
    if name in golden["problems"]:
        assert res.final_err <= 1.05 * golden["problems"][name]["final_err"]
    assert abs(res.final_err - want.final_err) <= 2e-2 * want.final_err
    cams, pts = h.get_params()
    ex = Oracle(prob).exQT(cams=cams, pts=pts)
    assert abs(ex @ ex - res.final_err) <= 1e-9 * res.final_err
    h.close()
