"""HIP path vs the CPU oracle, through the C ABI (include/psba_hip.h).  Needs an MI355X.

Tolerances (fp64 everywhere).  The GPU uses fused multiply-adds and its own summation
orders, the oracle neither, so results agree to rounding, not bitwise:
  * per-observation quantities (ex, JA, JB, W, Y): 1e-11 of the largest magnitude of the array
  * reductions (U, V, g, S, ea): 1e-11 of the largest magnitude (SURVEY 8c suggests 1e-10/1e-11)
  * dp at iteration 0 (cond(S) ~ 1e3): 1e-9 relative
  * LM cost trajectory for itno <= 4 vs goldens: 1e-9 relative; final cost 1e-6 relative (the
    north-star bar) and in practice ~1e-9.
"""
import numpy as np
import pytest

from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
NAMES = ["7cams", "54cams", "trafalgar21"]


def close(got, want, tol, what=""):
    got, want = np.asarray(got), np.asarray(want)
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= tol * scale, f"{what}: max|diff|={err:.3e} scale={scale:.3e} rel={err / scale:.3e}"


@pytest.fixture(scope="module")
def gpu():
    import psba_amd
    h = psba_amd.Psba(0)
    yield h
    h.close()


@pytest.mark.parametrize("name", NAMES)
def test_per_kernel_parity(name, problems, gpu):
    """Walks one LM damping try with the sba_func.h mirror, comparing every intermediate."""
    prob = problems[name]
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize()
    # compute_exQT, compute_jacobiQT
    close(gpu.compute_exQT(), lin["ex"], 1e-11, "ex")
    JA, JB = gpu.compute_jacobiQT()
    close(JA, lin["JA"], 1e-11, "JA")
    close(JB, lin["JB"], 1e-11, "JB")
    assert np.all(JA.reshape(-1, 12)[:, 9] == 0.0)  # the reference writes an exact 0 (jacobiQT.cl:114)
    # compute_U / V / Wblks / g
    close(gpu.compute_U(1.0), lin["U"], 1e-11, "U")
    close(gpu.compute_V(1.0), lin["V"], 1e-11, "V")
    close(gpu.compute_Wblks(1.0), lin["W"], 1e-11, "W")
    close(gpu.compute_g(1.0), lin["g"], 1e-11, "g")
    mx = gpu.maxElmOfUV()
    assert abs(mx - lin["maxdiag"]) <= 1e-12 * lin["maxdiag"]
    # damping try
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    Us, Vs = gpu.update_UV(mu)
    close(Us, sch["Ustar"], 1e-11, "U*")
    close(Vs, sch["Vstar"], 1e-11, "V*")
    rc, Vinv = gpu.compute_Vinv()
    assert rc == 0
    close(Vinv, sch["Vinv"], 1e-10, "Vinv")
    close(gpu.compute_Yblks(), sch["Y"], 1e-10, "Y")
    S = gpu.compute_S()
    close(S, sch["S"], 1e-11, "S")
    assert np.abs(S - S.T).max() <= 1e-14 * np.abs(S).max()
    close(gpu.compute_ea(), sch["eab"][: o.nA], 1e-10, "ea")
    ret, dp, eab = o.solve(lin, sch)
    assert ret == 0.0
    rc, dpa = gpu.SPDinv_matVec()
    assert rc == 0
    close(dpa, dp[: o.nA], 1e-9, "dpa")
    close(gpu.compute_eb(), eab[o.nA:], 1e-9, "eb")
    close(gpu.compute_dpb(), dp, 1e-9, "dp")
    newp = gpu.compute_newp()
    close(newp, np.r_[o.cams, o.pts] + dp, 1e-12, "newp")
    gpu.restore_UVdiag()
    # cost at the proposal, then accept (update_p)
    new_cost = gpu.residual(1)
    ex_new = o.exQT(cams=newp[: o.nA], pts=newp[o.nA:])
    assert abs(new_cost - ex_new @ ex_new) <= 1e-9 * (ex_new @ ex_new)
    p = gpu.update_p()
    assert np.array_equal(p, newp)


@pytest.mark.parametrize("coeff,coeff_g", [(2.0, -2.0)])
def test_trust_region_coefficients(coeff, coeff_g, problems, gpu):
    """U,V,W scale with coeff and g with coeff_g (trust_region.cpp:122,133-137)."""
    prob = problems["7cams"]
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize(coeff, coeff_g)
    close(gpu.compute_U(coeff), lin["U"], 1e-11, "U")
    close(gpu.compute_V(coeff), lin["V"], 1e-11, "V")
    close(gpu.compute_Wblks(coeff), lin["W"], 1e-11, "W")
    close(gpu.compute_g(coeff_g), lin["g"], 1e-11, "g")


@pytest.mark.parametrize("name", NAMES)
def test_fused_try_scalars(name, problems, gpu):
    """The fused verbs of one damping try return the scalars the LM loop needs."""
    prob = problems[name]
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    _, dp, _ = o.solve(lin, sch)
    cost0 = gpu.residual(0)
    assert abs(cost0 - lin["ex"] @ lin["ex"]) <= 1e-12 * cost0
    gpu.linearize(1.0, 1.0)
    assert abs(gpu.max_diag() - lin["maxdiag"]) <= 1e-12 * lin["maxdiag"]
    gpu.schur_assemble(mu)
    gpu.schur_reduce()
    gpu.schur_solve()
    sc = gpu.backsub(mu)
    assert sc.status == 0
    newp = np.r_[o.cams, o.pts] + dp
    ex_new = o.exQT(cams=newp[: o.nA], pts=newp[o.nA:])
    for got, want in [(sc.dp_l2, dp @ dp), (sc.gain_den, dp @ (mu * dp + lin["g"])),
                      (sc.new_cost, ex_new @ ex_new), (sc.newp_l2, newp @ newp)]:
        assert abs(got - want) <= 1e-8 * abs(want), (got, want)


@pytest.mark.parametrize("name", NAMES)
def test_levmar_matches_goldens_and_oracle(name, golden, problems, gpu):
    g, prob = golden["problems"][name], problems[name]
    gpu.upload_problem(prob)
    res, log = gpu.levmar(max_iter=50, tr_handoff=False)
    acc = log[log[:, 4] > 0]
    assert abs(res.init_err - g["init_err"]) <= 1e-12 * g["init_err"]
    assert abs(res.mu0 - g["mu0"]) <= 1e-12 * g["mu0"]
    for k, want in enumerate(g["err_after_itno"]):
        assert abs(acc[k, 1] - want) <= 1e-9 * want, (k, acc[k, 1], want)
    # final cost: north-star bar 1e-6 rel of the CPU path; stall-tail noise is ~1e-12
    assert abs(res.final_err - g["final_err"]) <= 1e-6 * g["final_err"]
    ores, _ = Oracle(prob).levmar(max_iter=50, tr_handoff=False)
    assert abs(res.final_err - ores.final_err) <= 1e-6 * ores.final_err
    assert res.flag == g["flag"]
    # final parameters reproduce the final cost
    cams, pts = gpu.get_params()
    ex = Oracle(prob).exQT(cams=cams, pts=pts)
    assert abs(ex @ ex - res.final_err) <= 1e-9 * res.final_err


def test_levmar_tr_handoff(problems, gpu):
    gpu.upload_problem(problems["54cams"])
    res, _ = gpu.levmar(max_iter=50, tr_handoff=True)
    assert res.flag == 2 and res.iters == 5


def test_not_spd_is_reported_and_lm_recovers(problems, gpu):
    """mu = -huge makes S indefinite: the solve must flag it (SPDinv ret = 1.0), not crash."""
    prob = problems["7cams"]
    gpu.upload_problem(prob)
    gpu.linearize(1.0, 1.0)
    gpu.schur_assemble(-1e30)
    gpu.schur_reduce()
    gpu.schur_solve()
    sc = gpu.backsub(-1e30)
    assert sc.status & 1
    # and a sane mu afterwards works
    mu = 1e-3 * gpu.max_diag()
    gpu.schur_assemble(mu); gpu.schur_reduce(); gpu.schur_solve()
    assert gpu.backsub(mu).status == 0


def test_ragged_tracks_and_tile_edges(gpu):
    """Synthetic problem with track lengths 1..nC, more than one tile, a point seen by every
    camera and points seen by one camera only."""
    import psba_amd.synth as synth
    prob = synth.make_problem(n_cams=12, n_pts=700, mean_track=4.0, seed=3, min_track=1, max_track=12)
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize()
    close(gpu.compute_U(1.0), lin["U"], 1e-11, "U")
    close(gpu.compute_V(1.0), lin["V"], 1e-11, "V")
    close(gpu.compute_g(1.0), lin["g"], 1e-11, "g")
    mu = 1e-3 * lin["maxdiag"]
    gpu.update_UV(mu)
    sch = o.schur(lin, mu)
    close(gpu.compute_S(), sch["S"], 1e-11, "S")
    close(gpu.compute_ea(), sch["eab"][: o.nA], 1e-10, "ea")
    _, dp, _ = o.solve(lin, sch)
    gpu.SPDinv_matVec()
    close(gpu.compute_dpb(), dp, 1e-8, "dp")


def test_upload_rejects_unsorted_observations(problems, gpu):
    import psba_amd
    prob = dict(problems["7cams"])
    prob["iidx"] = prob["iidx"][::-1].copy()
    with pytest.raises(psba_amd.PsbaError):
        gpu.upload_problem(prob)


def test_verbs_before_upload_fail_cleanly():
    import psba_amd
    h = psba_amd.Psba(0)
    with pytest.raises(psba_amd.PsbaError):
        h.residual(0)
    h.close()


def test_venice_sized_parity_away_from_the_initial_point(gpu):
    """Full-size check (52 cameras, 64k points, 347k observations, several camera-row groups
    and 80 point chunks in K2) at parameters with non-zero local rotations: S, ea, dp and the
    LM trajectory against the oracle."""
    import psba_amd.synth as synth
    prob = synth.venice_shaped()
    o = Oracle(prob)
    o.levmar(max_iter=3, tr_handoff=False)  # move away from v = 0
    gpu.upload_problem(prob)
    gpu.set_params(o.cams, o.pts)
    lin = o.linearize()
    for mu in (1e-3 * lin["maxdiag"], 1.0):
        sch = o.schur(lin, mu)
        gpu.linearize(1.0, 1.0)
        gpu.update_UV(mu)
        close(gpu.compute_S(), sch["S"], 1e-11, "S")
        close(gpu.compute_ea(), sch["eab"][: o.nA], 1e-9, "ea")
        ret, dp, _ = o.solve(lin, sch)
        rc, dpa = gpu.SPDinv_matVec()
        assert rc == 0 and ret == 0.0
        # cond(S) grows as mu falls: compare through the residual of the linear system instead
        r = sch["S"] @ dpa - sch["eab"][: o.nA]
        assert np.abs(r).max() <= 1e-9 * np.abs(sch["eab"][: o.nA]).max()
        gpu.restore_UVdiag()
    gpu.upload_problem(prob)
    res, log = gpu.levmar(max_iter=12, tr_handoff=False)
    ores, olog = Oracle(prob).levmar(max_iter=12, tr_handoff=False)
    assert res.tries == ores.tries
    assert abs(res.final_err - ores.final_err) <= 1e-9 * ores.final_err


@pytest.mark.parametrize("first_block", ["flush", "expand", "kernel", "expand+side-stream"])
def test_single_rank_communicator_runs_the_rccl_path(problems, golden, monkeypatch, first_block):
    """psba_comm_init with one rank: the all-reduce of the packed [tril(S) | ea] sums (then
    scattered into the padded buffer), of the try scalars and of the status flags goes through
    RCCL; results must not change.  The first diagonal block of S is factored beside the S-reduce
    kernel (one rank), beside the scatter kernel after the all-reduce (the route several ranks
    take), or by a kernel of its own.  side-stream: the scalar all-reduce and its copy on the
    handle's second stream, beside the linearization queued ahead (what N > 1 does by default)."""
    import psba_amd
    if first_block.startswith("expand"):
        monkeypatch.setenv("PSBA_SCHUR_NO_FLUSH_DIAG", "1")
    if first_block.endswith("side-stream"):
        monkeypatch.setenv("PSBA_COMM_SIDE_STREAM", "1")
    if first_block == "kernel":
        monkeypatch.setenv("PSBA_CHOL_SEPARATE_DIAG", "1")
    h = psba_amd.Psba(0)
    h.comm_init(1, 0, psba_amd.Psba.comm_unique_id())
    prob = problems["54cams"]
    h.upload_problem(prob)
    res, log = h.levmar(max_iter=6, tr_handoff=False)
    acc = log[log[:, 4] > 0]
    g = golden["problems"]["54cams"]
    for k, want in enumerate(g["err_after_itno"]):
        assert abs(acc[k, 1] - want) <= 1e-9 * want
    assert abs(res.mu0 - g["mu0"]) <= 1e-12 * g["mu0"]
    h.close()


def test_points_seen_by_one_camera_and_empty_tail(gpu):
    """Degenerate tracks: points with a single observation (V_i rank 2: only mu makes it
    invertible) must not break the path."""
    import psba_amd.synth as synth
    prob = synth.make_problem(n_cams=9, n_pts=300, mean_track=1.6, seed=11, min_track=1, max_track=9)
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    gpu.linearize(1.0, 1.0)
    gpu.update_UV(mu)
    sch = o.schur(lin, mu)
    close(gpu.compute_S(), sch["S"], 1e-11, "S")
    _, dp, _ = o.solve(lin, sch)
    gpu.SPDinv_matVec()
    close(gpu.compute_dpb(), dp, 1e-8, "dp")


def test_two_rank_layout_emulated_on_one_gpu(problems):
    """The rank-dependent kernel logic (mu*I, identity padding and camera terms on rank 0 only,
    per-rank partial U / g_a folded into the reduce buffer) with the all-reduce done by hand:
    two handles on one GPU, each owning a point shard (psba_set_rank_layout +
    psba_get/set_reduce_buffer), must reproduce the single-handle try."""
    import psba_amd
    from psba_amd import capi
    prob = problems["54cams"]
    o = Oracle(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    _, dp, _ = o.solve(lin, sch)
    ref = psba_amd.Psba(0)
    ref.upload_problem(prob)
    ref.linearize(1.0, 1.0)
    ref.schur_assemble(mu); ref.schur_reduce(); ref.schur_solve()
    want = ref.backsub(mu)
    hs = []
    for r in range(2):
        h = psba_amd.Psba(0)
        h.set_rank_layout(2, r)
        h.upload_problem(capi.shard_problem(prob, 2, r))
        h.linearize(1.0, 1.0)
        h.schur_assemble(mu)
        hs.append(h)
    total = hs[0].get_reduce_buffer() + hs[1].get_reduce_buffer()
    n32 = (o.nA + 31) // 32 * 32
    M = total.reshape(n32 + 1, n32)
    close(M[: o.nA, : o.nA], sch["S"], 1e-11, "S summed over ranks")
    close(M[n32, : o.nA], sch["eab"][: o.nA], 1e-10, "ea summed over ranks")
    assert np.array_equal(M[o.nA: n32, o.nA:], np.eye(n32 - o.nA))  # identity padding exactly once
    got = np.zeros(4)
    for h in hs:
        h.set_reduce_buffer(total)
        h.schur_solve()
        sc = h.backsub(mu)
        assert sc.status == 0
        got += [sc.dp_l2, sc.gain_den, sc.new_cost, sc.newp_l2]
    for g, w in zip(got, [want.dp_l2, want.gain_den, want.new_cost, want.newp_l2]):
        assert abs(g - w) <= 1e-9 * abs(w), (g, w)
    # proposed cameras are replicated, proposed points are the shards
    c0, p0 = hs[0].get_params(1)
    c1, p1 = hs[1].get_params(1)
    assert np.array_equal(c0, c1)
    newp = np.r_[o.cams, o.pts] + dp
    close(np.r_[c0.reshape(-1), p0.reshape(-1), p1.reshape(-1)], newp, 1e-9, "proposal")
    for h in hs + [ref]:
        h.close()


def test_global_atomic_fallback_path(problems, monkeypatch):
    """The K2 fallback used when S's block triangle cannot be split into <= 32 LDS-sized groups
    (very many cameras) -- global fp64 atomics -- stays correct (forced with PSBA_SCHUR_ATOMIC)."""
    import psba_amd
    monkeypatch.setenv("PSBA_SCHUR_ATOMIC", "1")
    prob = problems["54cams"]
    o = Oracle(prob)
    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    h.linearize(1.0, 1.0)
    h.update_UV(mu)
    close(h.compute_S(), sch["S"], 1e-11, "S")
    close(h.compute_ea(), sch["eab"][: o.nA], 1e-10, "ea")
    _, dp, _ = o.solve(lin, sch)
    h.SPDinv_matVec()
    close(h.compute_dpb(), dp, 1e-9, "dp")
    h.close()


@pytest.mark.parametrize("n_cams", [96, 106, 170, 200])
def test_many_cameras(gpu, n_cams):
    """96, 106 and 170 cameras (nA = 576 / 636 / 1020; n32 = 1024 is the largest matrix on the fused
    identity-row chain, k_cholg_solve<10> up to n32 = 640 and <16> beyond) and 200 (nA = 1200: the
    fused panel kernel without identity rows + the multi-block backward solve)."""
    import psba_amd.synth as synth
    prob = synth.make_problem(n_cams=n_cams, n_pts=3000, mean_track=5.0, seed=5)
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    gpu.linearize(1.0, 1.0)
    gpu.update_UV(mu)
    close(gpu.compute_S(), sch["S"], 1e-11, "S")
    ret, dp, _ = o.solve(lin, sch)
    rc, dpa = gpu.SPDinv_matVec()
    assert rc == 0 and ret == 0.0
    close(dpa, dp[: o.nA], 1e-8, "dpa")
    close(gpu.compute_dpb(), dp, 1e-8, "dp")
    gpu.restore_UVdiag()
    gpu.upload_problem(prob)
    res, _ = gpu.levmar(max_iter=6, tr_handoff=False)
    ores, _ = Oracle(prob).levmar(max_iter=6, tr_handoff=False)
    assert abs(res.final_err - ores.final_err) <= 1e-9 * ores.final_err


def test_single_workgroup_cholesky_fallback(problems, monkeypatch):
    """PSBA_CHOL_SINGLE=1 selects the one-workgroup kernel of kernels_chol.hip; it must agree."""
    import psba_amd
    monkeypatch.setenv("PSBA_CHOL_SINGLE", "1")
    prob = problems["54cams"]
    o = Oracle(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    _, dp, _ = o.solve(lin, sch)
    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    h.linearize(1.0, 1.0)
    h.update_UV(mu)
    h.compute_S()
    rc, dpa = h.SPDinv_matVec()
    assert rc == 0
    close(dpa, dp[: o.nA], 1e-9, "dpa")
    h.close()


def test_unfused_panel_chain(problems, monkeypatch):
    """PSBA_CHOL_UNFUSED=1 selects trsm and update as two kernels per panel (the path matrices
    too large for the fused panel kernel take); it must agree with the oracle as well."""
    import psba_amd
    monkeypatch.setenv("PSBA_CHOL_UNFUSED", "1")
    prob = problems["54cams"]
    o = Oracle(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    _, dp, _ = o.solve(lin, sch)
    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    h.linearize(1.0, 1.0)
    h.update_UV(mu)
    h.compute_S()
    rc, dpa = h.SPDinv_matVec()
    assert rc == 0
    close(dpa, dp[: o.nA], 1e-9, "dpa")
    h.close()


def test_linearize_ahead_equals_plain_verbs(problems):
    """psba_backsub_async + psba_linearize_ahead + psba_backsub_wait (the host decides while the
    GPU already linearizes at the proposed parameters) must give the same iterates as the plain
    verb sequence, through accepted and rejected steps alike."""
    import psba_amd
    prob = problems["54cams"]

    def run(ahead):
        h = psba_amd.Psba(0)
        h.upload_problem(prob)
        cost = h.residual()
        h.linearize(1.0, 1.0)
        mu = 1e-3 * h.max_diag()
        costs, nu = [], 2
        for it in range(6):
            h.linearize(1.0, 1.0)
            # the second try of iteration 2 is forced to fail (huge negative damping is not SPD)
            for attempt in range(3):
                m = -1e30 if (it == 2 and attempt == 0) else mu
                h.schur_assemble(m); h.schur_reduce(); h.schur_solve()
                if ahead:
                    h.backsub_async(m); h.linearize_ahead(); sc = h.backsub_wait()
                else:
                    sc = h.backsub(m)
                if not (sc.status & 1) and cost - sc.new_cost > 0:
                    h.accept(); cost = sc.new_cost; mu *= 0.5
                    break
                mu *= nu
            costs.append(cost)
        cams, pts = h.get_params()
        h.close()
        return np.array(costs), cams, pts

    c0, cams0, pts0 = run(False)
    c1, cams1, pts1 = run(True)
    assert c0[-1] < c0[0]
    # W / PV are written by plain stores in both runs and U / g_a by LDS atomics whose order is
    # not fixed: agreement to rounding, not bit for bit
    np.testing.assert_allclose(c1, c0, rtol=1e-12)
    np.testing.assert_allclose(cams1, cams0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(pts1, pts0, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("bad", ["huge", "overflow", "nan", "inf", "rot>1"])
def test_garbage_step_through_the_lookahead_chain(problems, bad):
    """VERDICT r3 / ADVICE r1: the loop queues K1 at the proposed parameters before it knows whether the
    solve failed, so K3 -> look-ahead K1 -> residual must survive ANY dpa.  A garbage step (huge, overflowing,
    NaN, inf, a local rotation with |v| > 1 whose sqrt(1 - |v|^2) is NaN) is injected with psba_set_step
    between the solve and the back-substitution.  Expected: no memory fault (every address in K1 / K3 /
    k_residual comes from the static index arrays, never from a value), scalars returned (non-finite or
    huge), and after the rejected step the handle continues exactly like one that never saw it."""
    import psba_amd
    prob = problems["54cams"]
    want_h = psba_amd.Psba(0)
    want_h.upload_problem(prob)
    want, _ = want_h.levmar(max_iter=6, tr_handoff=False)
    want_h.close()

    h = psba_amd.Psba(0)
    h.upload_problem(prob)
    cost0 = h.residual()
    h.linearize(1.0, 1.0)
    mu = 1e-3 * h.max_diag()
    h.schur_assemble(mu); h.schur_reduce(); h.schur_solve()
    dp = h.get_dp()
    nA = h.nA
    if bad == "huge":
        dp[:] = 1e12 * np.sign(dp + 1e-300)
    elif bad == "overflow":
        dp[:] = 1e300
    elif bad == "nan":
        dp[:nA:7] = np.nan
    elif bad == "inf":
        dp[1:nA:5] = np.inf
        dp[2:nA:5] = -np.inf
    else:
        dp[:nA].reshape(-1, 6)[:, :3] = 0.9  # |v|^2 = 2.43 > 1 after the update
    h.set_step(dp)
    h.backsub_async(mu); h.linearize_ahead(); sc = h.backsub_wait()
    new_cost = h.residual(1)
    assert not (sc.status & 1)            # the factorization itself was fine
    assert not (new_cost < cost0)         # NaN or larger: never an improvement
    assert not (sc.new_cost < cost0)
    # the step is rejected: nothing of it may survive.  The plain loop from here equals a fresh handle's.
    assert abs(h.residual(0) - cost0) <= 1e-13 * cost0   # (workgroup sums arrive in any order: rounding, not bits)
    res, _ = h.levmar(max_iter=6, tr_handoff=False)
    same = res.iters == want.iters and abs(res.final_err - want.final_err) <= 1e-12 * want.final_err
    if not same:
        # DESIGN 5d, "LM runs that ended on another cost": about one fresh handle's run in a few hundred takes one
        # damping try differently (open).  The reference run above may be that one: a second fresh handle decides.
        import warnings
        again_h = psba_amd.Psba(0)
        again_h.upload_problem(prob)
        again, _ = again_h.levmar(max_iter=6, tr_handoff=False)
        again_h.close()
        warnings.warn(f"two fresh handles disagree: {want.final_err!r} / {again.final_err!r} (after the garbage step: {res.final_err!r})")
        same = res.iters == again.iters and abs(res.final_err - again.final_err) <= 1e-12 * again.final_err
    assert same
    h.close()


@pytest.mark.parametrize("n_cams", [3, 5, 6, 11])
def test_few_cameras(gpu, n_cams):
    """Edge sizes of the panel chain: a single 32-column panel (<= 5 cameras: only the first
    diagonal factor, the last-panel trsm and the final mat-vec run), a matrix that just spills
    into a second panel, and one that ends in a mostly padded panel."""
    import psba_amd.synth as synth
    prob = synth.make_problem(n_cams=n_cams, n_pts=400, mean_track=min(3.0, n_cams), seed=11 + n_cams,
                              max_track=n_cams)
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    gpu.linearize(1.0, 1.0)
    gpu.update_UV(mu)
    close(gpu.compute_S(), sch["S"], 1e-11, "S")
    ret, dp, _ = o.solve(lin, sch)
    rc, dpa = gpu.SPDinv_matVec()
    assert rc == 0 and ret == 0.0
    close(dpa, dp[: o.nA], 1e-8, "dpa")
    close(gpu.compute_dpb(), dp, 1e-8, "dp")


@pytest.mark.parametrize("n_cams,n_pts", [(52, 20000), (96, 6000), (200, 6000)])
def test_solve_residual_property(gpu, n_cams, n_pts):
    """Size-independent check of the dense solve on matrices the oracle would take long to factor:
    with S and e_a as the GPU assembled them, ||S dpa - e_a|| / (||S|| ||dpa|| + ||e_a||) is at
    rounding level -- for the fused panel chain with its L^-T mat-vec (52, 96 cameras) and for the
    two-kernel panels with the sequential backward solve (200)."""
    import psba_amd.synth as synth
    prob = synth.make_problem(n_cams=n_cams, n_pts=n_pts, mean_track=5.0, seed=21 + n_cams)
    gpu.upload_problem(prob)
    gpu.linearize(1.0, 1.0)
    mu = 1e-3 * gpu.max_diag()
    gpu.update_UV(mu)
    S = gpu.compute_S()
    ea = gpu.compute_ea()
    rc, dpa = gpu.SPDinv_matVec()
    assert rc == 0
    assert np.abs(S - S.T).max() <= 1e-14 * np.abs(S).max()
    r = S @ dpa - ea
    scale = np.linalg.norm(S, 2) * np.linalg.norm(dpa) + np.linalg.norm(ea)
    assert np.linalg.norm(r) <= 1e-13 * scale, np.linalg.norm(r) / scale
    # and against LAPACK on the same S
    ref = np.linalg.solve(S, ea)
    np.testing.assert_allclose(dpa, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    gpu.restore_UVdiag()


def test_begin_equals_residual_linearize_max_diag(gpu, problems):
    """psba_begin (one synchronisation) against the three verbs it stands for."""
    prob = problems["54cams"]
    gpu.upload_problem(prob)
    cost = gpu.residual()
    gpu.linearize(1.0, 1.0)
    md = gpu.max_diag()
    gpu.upload_problem(prob)
    c2, m2 = gpu.begin(1.0, 1.0)
    assert abs(c2 - cost) <= 1e-13 * cost  # one atomic per workgroup: summation order is not fixed
    assert abs(m2 - md) <= 1e-13 * md  # U's LDS atomics: summation order is not fixed
    gpu.linearize(1.0, 1.0)            # nothing to do: the linearization is already there
    o = Oracle(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    gpu.update_UV(mu)
    close(gpu.compute_S(), o.schur(lin, mu)["S"], 1e-11, "S")


def test_reset_params_restores_the_uploaded_parameters(gpu, problems):
    prob = problems["7cams"]
    gpu.upload_problem(prob)
    res1, _ = gpu.levmar(max_iter=4, tr_handoff=False)
    c, p = gpu.get_params()
    assert np.abs(c - np.asarray(prob["cams"]).reshape(c.shape)).max() > 0
    gpu.reset_params()
    c, p = gpu.get_params()
    np.testing.assert_array_equal(c, np.asarray(prob["cams"]).reshape(c.shape))
    np.testing.assert_array_equal(p, np.asarray(prob["pts"]).reshape(p.shape))
    res2, _ = gpu.levmar(max_iter=4, tr_handoff=False)
    assert abs(res2.final_err - res1.final_err) <= 1e-12 * res1.final_err


@pytest.mark.parametrize("n_cams,n_pts,mean_track,min_track,max_track", [
    (8, 500, 7.5, 2, 8),      # every point seen by almost every camera
    (52, 800, 30.0, 20, 52),  # long tracks: up to 1378 products per point
    (17, 1200, 2.2, 1, 5),    # short tracks, single observations
    (64, 600, 10.0, 3, 64),
    (33, 5000, 3.0, 2, 12),
    (120, 1500, 6.0, 2, 30),  # two-kernel panels, 25+ camera-row groups
])
def test_problem_shape_sweep(gpu, n_cams, n_pts, mean_track, min_track, max_track):
    """The whole path against the oracle over problem shapes that stress different parts of the
    static K2 schedule (track length distribution, number of groups) and of the panel chain."""
    import psba_amd.synth as synth
    prob = synth.make_problem(n_cams=n_cams, n_pts=n_pts, mean_track=mean_track, seed=100 + n_cams,
                              min_track=min_track, max_track=max_track)
    o = Oracle(prob)
    gpu.upload_problem(prob)
    lin = o.linearize()
    mu = 1e-3 * lin["maxdiag"]
    sch = o.schur(lin, mu)
    gpu.linearize(1.0, 1.0)
    gpu.update_UV(mu)
    close(gpu.compute_S(), sch["S"], 1e-11, "S")
    close(gpu.compute_ea(), sch["eab"][: o.nA], 1e-10, "ea")
    ret, dp, _ = o.solve(lin, sch)
    rc, dpa = gpu.SPDinv_matVec()
    assert rc == 0 and ret == 0.0
    close(dpa, dp[: o.nA], 1e-8, "dpa")
    close(gpu.compute_dpb(), dp, 1e-8, "dp")
    gpu.restore_UVdiag()
    gpu.upload_problem(prob)
    res, _ = gpu.levmar(max_iter=5, tr_handoff=False)
    ores, _ = Oracle(prob).levmar(max_iter=5, tr_handoff=False)
    assert abs(res.final_err - ores.final_err) <= 1e-9 * ores.final_err
