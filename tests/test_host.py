"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the
header declares, the product's sba reader matches an independent reader, point partitioning."""
import os
import re

import numpy as np
import pytest

from conftest import DATA, ROOT


def test_library_exports_every_declared_symbol():
    from psba_amd import capi
    hdr = open(os.path.join(ROOT, "include", "psba_hip.h")).read()
    hdr = re.sub(r"#ifdef PSBA_BUILD_EXPERIMENTS.*?#endif /\* PSBA_BUILD_EXPERIMENTS \*/", "", hdr, flags=re.S)  # not in the product library
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(psba_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(capi.lib, name), f"{name} declared in psba_hip.h but not exported"
    bound = {s[0] for s in capi.SIGNATURES}
    assert declared == bound, (declared ^ bound)


def test_no_gpu_means_loud_failure_not_fallback():
    import subprocess, sys
    code = ("import psba_amd, sys\n"
            "try:\n    psba_amd.Psba(0)\n    print('HANDLE')\n"
            "except psba_amd.PsbaError as e:\n    print('RAISED', e.code)\n")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert "RAISED -2" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("cams,pts", [("7cams.txt", "7pts.txt"), ("7camsvarK.txt", "7pts.txt"),
                                      ("9cams.txt", "9pts.txt"), ("54camsvarK.txt", "54pts.txt"),
                                      ("3cams.txt", "3pts.txt"), ("5cams.txt", "5pts.txt"),
                                      ("Trafalgar-21-11315-cams.txt", "Trafalgar-21-11315-pts.txt")])
def test_reader_matches_independent_reader(cams, pts):
    import psba_amd
    from sba_text import KK, read_problem
    a = psba_amd.read_problem(os.path.join(DATA, cams), os.path.join(DATA, pts), KK)
    b = read_problem(os.path.join(DATA, cams), os.path.join(DATA, pts))
    assert (a["nC"], a["nP"], a["nO"]) == (b["nC"], b["nP"], b["nO"])
    for k in ("K", "initrot", "cams", "pts", "impts", "iidx", "jidx"):
        assert np.array_equal(np.asarray(a[k]).reshape(-1), np.asarray(b[k]).reshape(-1)), k
    # local rotations are zeroed, quaternions are unit with non-negative scalar part
    assert np.all(a["cams"][:, :3] == 0)
    assert np.allclose((a["initrot"] ** 2).sum(1), 1.0, atol=1e-12) and np.all(a["initrot"][:, 0] >= 0)


def test_varK_file_reproduces_fixed_K_problem():
    import psba_amd
    from sba_text import KK
    a = psba_amd.read_problem(os.path.join(DATA, "7cams.txt"), os.path.join(DATA, "7pts.txt"), KK)
    b = psba_amd.read_problem(os.path.join(DATA, "7camsvarK.txt"), os.path.join(DATA, "7pts.txt"))
    for k in ("K", "initrot", "cams", "pts", "impts"):
        np.testing.assert_allclose(a[k], b[k], rtol=0, atol=0)


def test_reader_errors():
    import psba_amd
    with pytest.raises(psba_amd.PsbaError):
        psba_amd.read_problem("/nonexistent/cams.txt", "/nonexistent/pts.txt")
    with pytest.raises(psba_amd.PsbaError):  # 7-column cams without a fixed K
        psba_amd.read_problem(os.path.join(DATA, "7cams.txt"), os.path.join(DATA, "7pts.txt"))


@pytest.mark.parametrize("nranks", [1, 2, 3, 4, 8])
def test_partition_points_is_contiguous_and_balanced(nranks, problems):
    import psba_amd
    prob = problems["trafalgar21"]
    b = psba_amd.partition_points(prob["nP"], prob["iidx"], nranks)
    assert b[0] == 0 and b[-1] == prob["nP"] and np.all(np.diff(b) > 0)
    counts = np.array([np.sum((prob["iidx"] >= b[r]) & (prob["iidx"] < b[r + 1])) for r in range(nranks)])
    assert counts.sum() == prob["nO"]
    assert counts.max() - counts.min() <= 2 * np.bincount(prob["iidx"]).max() + 1


def test_partition_more_ranks_than_points():
    import psba_amd
    b = psba_amd.partition_points(2, np.array([0, 0, 1], dtype=np.int32), 4)
    assert b[0] == 0 and b[-1] == 2 and np.all(np.diff(b) >= 0)


def _check_plan(nC, nP, iidx, jidx):
    """Every product (a, b <= a, same point) appears exactly once, in the workgroup list of its
    block's group, at its block's position; in a row of 16 item slots no LDS bank pair is
    hit more than twice; workgroup lists are equally long within a group."""
    from psba_amd import capi
    plan = capi.schur_plan(nC, nP, iidx, jidx)
    assert plan["groups"] >= 1
    items, wg, pos, glo = plan["items"], plan["wg"], plan["blockpos"], plan["glo"]
    ptr = np.searchsorted(iidx, np.arange(nP + 1))
    tri = lambda j: j * (j + 1) // 2
    # block positions: a bijection into the padded partition of the block's group
    for g in range(plan["groups"]):
        b0, b1 = int(glo[g]), int(glo[g + 1])
        ps = pos[b0:b1]
        assert len(set(ps.tolist())) == b1 - b0 and ps.min() >= 0
        nblk = {int(w[1]) for w in wg if w[0] == g}
        assert len(nblk) == 1 and ps.max() < nblk.pop() <= (b1 - b0 + 15) // 16 * 16
    seen = set()
    slabs = set()
    runs = plan["run_tasks"] > 0
    n_runs = 0
    n_pairs = 0
    products_of = {}
    for w in wg:
        g, nblk, obs0, pt0, s0, s1, slab, sD = (int(x) for x in w)
        assert s0 % 16 == 0 and s1 % 16 == 0 and sD % 16 == 0 and s0 <= sD <= s1 and (g, slab) not in slabs
        slabs.add((g, slab))
        # pair items first: one observation a, partners a - boff and a - boff + 1, two block positions
        it = items[s0:sD]
        live = it != np.uint64(0xFFFFFFFFFFFFFFFF)
        assert not runs or sD == s0
        a2 = obs0 + (it & np.uint64(0x3FFFF)).astype(np.int64)                      # 18 bits
        i2 = pt0 + ((it >> np.uint64(18)) & np.uint64(0xFFFF)).astype(np.int64)     # 16 bits
        boff2 = ((it >> np.uint64(34)) & np.uint64(0xFF)).astype(np.int64)
        p2a = ((it >> np.uint64(42)) & np.uint64(0x3FF)).astype(np.int64)
        p2b = ((it >> np.uint64(52)) & np.uint64(0x3FF)).astype(np.int64)
        assert ((it >> np.uint64(62))[live] == 0).all() and (boff2[live] >= 1).all()
        for r in range(0, len(it), 16):
            for pp in (p2a, p2b):  # each of the two phases of atomics on its own
                q = np.unique(pp[r:r + 16][live[r:r + 16]]) % 16
                assert np.bincount(q, minlength=16).max() <= 2
        n_pairs += int(live.sum())
        pa, pi, pboff, ppos = (np.concatenate([a2[live], a2[live]]), np.concatenate([i2[live], i2[live]]),
                               np.concatenate([boff2[live], boff2[live] - 1]), np.concatenate([p2a[live], p2b[live]]))
        it = items[sD:s1]
        live = it != np.uint64(0xFFFFFFFFFFFFFFFF)
        a = obs0 + (it & np.uint64(0xFFFFFF)).astype(np.int64)          # 24 bits
        i = pt0 + ((it >> np.uint64(24)) & np.uint64(0x3FFFFF)).astype(np.int64)  # 22 bits
        boff = ((it >> np.uint64(46)) & np.uint64(0xFF)).astype(np.int64)
        p = ((it >> np.uint64(54)) & np.uint64(0x3FF)).astype(np.int64)
        if runs:
            # the runs layout: [turn][512]; a thread's live items come first, grouped into runs of one position of
            # at most 32 products, each run in point order, and a position never comes back in a later run of the
            # same thread sooner than ... (it may: a block with more than 32 products has several runs)
            assert (s1 - s0) % 512 == 0
            T = (s1 - s0) // 512
            L, P, I = live.reshape(T, 512), p.reshape(T, 512), i.reshape(T, 512)
            assert not (L[1:] & ~L[:-1]).any()          # no live item behind a null one
            for thr in range(512):
                n = int(L[:, thr].sum())
                pp, ii = P[:n, thr], I[:n, thr]
                cut = np.flatnonzero(np.diff(pp) != 0) + 1
                for seg in np.split(np.arange(n), cut):
                    assert len(seg) <= 32 and (np.diff(ii[seg]) >= 0).all()
                n_runs += len(cut) + (1 if n else 0)
        else:
            for r in range(0, len(it), 16):
                q = np.unique(p[r:r + 16][live[r:r + 16]]) % 16  # lanes on one address serialise wherever they sit
                assert np.bincount(q, minlength=16).max() <= 2  # a bank pair: at most two addresses per row
        a, i, boff, p = (np.concatenate([a[live], pa]), np.concatenate([i[live], pi]), np.concatenate([boff[live], pboff]),
                         np.concatenate([p[live], ppos]))
        products_of[(g, slab)] = len(a)
        b = a - boff
        assert (iidx[a] == i).all() and (iidx[b] == i).all() and (b >= ptr[i]).all()
        ja, jb = jidx[a].astype(np.int64), jidx[b].astype(np.int64)
        blk = ja * (ja + 1) // 2 + jb
        assert ((blk >= glo[g]) & (blk < glo[g + 1])).all()
        assert (pos[blk] == p).all()
        keys = set(zip(a.tolist(), b.tolist()))
        assert len(keys) == len(a) and not (keys & seen)
        seen |= keys
    want = int(((np.arange(len(iidx)) - ptr[iidx]) + 1).sum())
    assert len(seen) == want == plan["products"]
    if runs:  # (two runs of one position that meet in a thread's list merge: never more runs than the plan says)
        assert 0 < n_runs <= plan["run_tasks"]
    assert n_pairs == plan["pair_items"]
    # balance: product counts of the workgroups of one group differ by at most one
    for g in range(plan["groups"]):
        n = [v for (gg, _), v in products_of.items() if gg == g]
        assert max(n) - min(n) <= 1
    return plan


def test_schur_plan_pair_items(problems, monkeypatch):
    """Round 4's pair items (experiments build only): every product once, each of the two phases of atomics with at
    most two addresses per bank pair and row, most products carried by pairs."""
    from psba_amd import capi, synth
    if not capi.HAS_EXPERIMENTS:
        pytest.skip("experiment build only (PSBA_BUILD_EXPERIMENTS=1)")
    monkeypatch.setenv("PSBA_SCHUR_PAIRS", "1")
    pr = synth.venice_shaped(n_pts=12000)
    plan = _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))
    assert 2 * plan["pair_items"] > 0.8 * plan["products"]
    pr = problems["54cams"]
    _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))


def test_schur_plan_small(problems):
    for name in ("7cams", "54cams"):
        pr = problems[name]
        _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))


def test_schur_plan_venice_shaped():
    from psba_amd import synth
    pr = synth.venice_shaped()
    plan = _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))
    assert plan["run_tasks"] == 0  # the uniform draw of SURVEY 8(d) has no runs: the row layout
    # the bank-pair schedule should cost little padding on a realistic problem
    assert plan["products"] / len(plan["items"]) > 0.75
    assert len(plan["wg"]) == 256


@pytest.mark.parametrize("cluster", [16, 64])
def test_schur_plan_clustered_tracks_take_the_runs_layout(cluster):
    """Neighbouring points that share camera sets (what real reconstructions look like in file order): a block's
    products come in runs, and the plan hands every thread whole runs (summed in registers, one set of LDS atomics
    per run) instead of dealing single products into bank-pair rows."""
    from psba_amd import synth
    pr = synth.venice_shaped(cluster=cluster)
    plan = _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))
    assert plan["run_tasks"] > 0
    assert plan["products"] / plan["run_tasks"] >= 0.5 * min(cluster, 6)   # runs about as long as the clusters (capped per workgroup: 6200 products, 512 threads)
    assert plan["products"] / len(plan["items"]) > 0.75                     # [turn][512] padding


@pytest.mark.parametrize("n_cams", [300, 700])
def test_schur_plan_many_cameras(n_cams):
    """Beyond the row-aligned groups (more than 128 of them, or a camera row longer than an LDS
    partition) the groups are consecutive ranges of the canonical block order, one workgroup each."""
    from psba_amd import synth
    pr = synth.make_problem(n_cams=n_cams, n_pts=3000, mean_track=5.0, seed=11)
    plan = _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))
    assert len(plan["wg"]) >= plan["groups"] and (plan["groups"] > 128 or n_cams < 500)
    glo = plan["glo"]
    assert glo[0] == 0 and glo[-1] == n_cams * (n_cams + 1) // 2 and (np.diff(glo) > 0).all()
    assert np.diff(glo).max() * 37 * 8 <= 160 * 1024 - 256


def _replay_ring_plan(nC, nP, iidx, jidx, max_wgs=None):
    """Replays the ring route's schedule (schur_ring_plan.cpp) the way k_schur_ring does and checks
    what the kernel relies on: a lane only ever gets products of the block it owns; both operands
    of a product are in the slots its entry names -- the partner's W record loaded at least `lat`
    steps earlier and not overwritten since, Y_a prepared (one step after its W_a was loaded) and
    still there; every product (a, b <= a, same point) is taken exactly once by the workgroup of
    its block range and point stretch; every observation feeds e_a exactly once."""
    from psba_amd import capi
    plan = capi.ring_plan(nC, nP, iidx, jidx)
    assert plan is not None
    wg, steps, ent, ops, jobs = plan["wg"], plan["steps"], plan["entries"], plan["ops"], plan["jobs"]
    lanes, lat, page, nslots = plan["lanes"], plan["lat"], plan["page"], plan["slots"]
    ptr = np.searchsorted(iidx, np.arange(nP + 1))
    rb = plan["rb"]
    assert rb[0] == 0 and rb[-1] == nC * (nC + 1) // 2 and (np.diff(rb) > 0).all()
    seen, ea_seen = set(), set()
    full = max_wgs is None
    pick = range(len(wg)) if full else np.linspace(0, len(wg) - 1, max_wgs).astype(int)
    for w in pick:
        blk0, nblk, row0, nrows, copy, nsteps, step0, ent0, op0, job0, lane0, bl0, nw = (int(x) for x in wg[w])
        assert nsteps % 4 == 0 and 0 < nw < nslots and 0 <= copy < plan["nS"]
        lane_blk = plan["lane_blk"][lane0:lane0 + lanes]
        first = plan["blk_lane0"][bl0:bl0 + nblk + 1]
        assert first[0] == 0 and first[-1] <= lanes and (np.diff(first) >= 1).all()
        for b in range(nblk):
            assert (lane_blk[first[b]:first[b + 1]] == b).all()
        wslot = {}   # slot -> (observation, step of its load)
        yslot = {}   # Y slot -> (observation, step in which it is written)
        for t in range(nsteps):
            ob, oe, jb, je = (int(x) for x in steps[step0 + t])
            assert oe - ob <= 4 * 16 and je - jb <= 4 * 32
            for a0, sn in ops[op0 + ob: op0 + oe]:
                s0, n = int(sn) & 0xFFFF, int(sn) >> 16
                assert 1 <= n <= page and s0 + n <= nw
                assert iidx[a0] == iidx[a0 + n - 1]  # a run of records of one point: contiguous in W
                for r in range(n):
                    wslot[s0 + r] = (int(a0) + r, t)
            e = ent[ent0 + t * lanes: ent0 + (t + 1) * lanes]
            for l in np.nonzero(e != 0xFFFFFFFF)[0]:
                ws, ys = int(e[l]) & 0xFFFF, int(e[l]) >> 16
                b, tl = wslot[ws]
                a, tp = yslot[ys]
                assert tl <= t - lat and tp <= t - 1 and b >= 0
                assert iidx[a] == iidx[b] and b <= a
                blk = int(jidx[a]) * (int(jidx[a]) + 1) // 2 + int(jidx[b])
                assert blk0 <= blk < blk0 + nblk and lane_blk[l] == blk - blk0
                assert (a, b) not in seen
                seen.add((a, b))
            for q in jobs[job0 + jb: job0 + je]:
                a, pt, sl, earow = (int(x) for x in q)
                assert iidx[a] == pt and wslot[sl & 0xFFFF] == (a, t)
                assert nw + (sl >> 16) < nslots
                yslot[sl >> 16] = (a, t + 1)
                if earow >= 0:
                    assert earow == jidx[a] - row0 < nrows and a not in ea_seen
                    ea_seen.add(a)
                else:
                    d = int(jidx[a]) * (int(jidx[a]) + 3) // 2
                    assert not (blk0 <= d < blk0 + nblk)
    if full:
        want = int(((np.arange(len(iidx)) - ptr[iidx]) + 1).sum())
        assert len(seen) == want == plan["products"]
        assert len(ea_seen) == len(iidx)
    return plan


def _needs_experiments():
    from psba_amd import capi
    if not capi.HAS_EXPERIMENTS:
        pytest.skip("the ring route is an experiment (round 3, rejected): PSBA_BUILD_EXPERIMENTS=1 python psba_amd/build.py, "
                    "then PSBA_LIB=psba_amd/libpsba_hip_exp.so")


def test_ring_plan_small(problems):
    _needs_experiments()
    for name in ("7cams", "54cams"):
        pr = problems[name]
        _replay_ring_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))


def test_ring_plan_venice_shaped():
    _needs_experiments()
    from psba_amd import synth
    pr = synth.venice_shaped()
    plan = _replay_ring_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]), max_wgs=6)
    assert len(plan["wg"]) == 256
    # the schedule should keep most lanes busy on a realistic problem
    assert plan["products"] / plan["lane_steps"] > 0.6


def test_writer_round_trip(tmp_path):
    """psba_write_problem (the writer the reference declares and comments out,
    PSBA/readparams.h:13-25) -> psba_read_problem: the same problem, with the local rotation
    folded into the quaternion; residuals (oracle) agree to rounding."""
    import psba_amd
    from oracle_lib import Oracle
    from sba_text import KK
    prob = psba_amd.read_problem(os.path.join(DATA, "9cams.txt"), os.path.join(DATA, "9pts.txt"), KK)
    rng = np.random.default_rng(3)
    cams = prob["cams"].copy()
    cams[:, :3] = rng.normal(0, 2e-2, (prob["nC"], 3))  # non-zero local rotations, as after an LM run
    cams[:, 3:] += rng.normal(0, 1e-2, (prob["nC"], 3))
    pts = prob["pts"] + rng.normal(0, 1e-2, prob["pts"].shape)
    for with_K in (True, False):
        fc, fp = str(tmp_path / f"c{with_K}.txt"), str(tmp_path / f"p{with_K}.txt")
        psba_amd.write_problem(fc, fp, prob, cams=cams, pts=pts, with_K=with_K)
        assert len(open(fc).readline().split()) == (12 if with_K else 7)
        back = psba_amd.read_problem(fc, fp, KK)
        assert (back["nC"], back["nP"], back["nO"]) == (prob["nC"], prob["nP"], prob["nO"])
        assert np.array_equal(back["iidx"], prob["iidx"]) and np.array_equal(back["jidx"], prob["jidx"])
        assert np.array_equal(back["impts"], prob["impts"]) and np.array_equal(back["pts"], pts)
        assert np.array_equal(back["K"], prob["K"]) and np.all(back["cams"][:, :3] == 0)
        assert np.array_equal(back["cams"][:, 3:], cams[:, 3:])
        o = Oracle(prob)
        e0 = o.exQT(cams=cams, pts=pts)
        e1 = Oracle(back).exQT()
        np.testing.assert_allclose(e1, e0, rtol=1e-11, atol=1e-9)


def test_reader_skips_covariance_columns(tmp_path):
    """pts files may carry a full 2x2 or an upper-triangular covariance after every x y
    (reference PSBA/readparams.cpp:272-283: detected from the first line; PSBA/main.cpp:112: read,
    then ignored).  None of the bundled files has one, so the fixture is made here from 7pts.txt."""
    import psba_amd
    from sba_text import KK
    base = psba_amd.read_problem(os.path.join(DATA, "7cams.txt"), os.path.join(DATA, "7pts.txt"), KK)
    for ncov in (4, 3):
        out = tmp_path / f"pts_cov{ncov}.txt"
        with open(os.path.join(DATA, "7pts.txt")) as f, open(out, "w") as g:
            for line in f:
                t = line.split()
                if not t or t[0].startswith("#"):
                    g.write(line)
                    continue
                n = int(t[3])
                rec = t[:4]
                for k in range(n):
                    rec += t[4 + 3 * k: 7 + 3 * k] + (["1.5", "0.25", "0.25", "2.5"] if ncov == 4 else ["1.5", "0.25", "2.5"])
                g.write(" ".join(rec) + "\n")
        got = psba_amd.read_problem(os.path.join(DATA, "7cams.txt"), str(out), KK)
        for k in ("K", "initrot", "cams", "pts", "impts", "iidx", "jidx"):
            assert np.array_equal(np.asarray(got[k]), np.asarray(base[k])), k


def test_bal_converter(tmp_path):
    """psba_convert_bal: a BAL text file (camera looking down -z, p = -P/P.z, radial k1 k2) becomes
    sba files in the reference's model (camera looking down +z, image y negated, K = (f,0,0,1,0)).
    The BAL file is synthesised here with k1 = k2 = 0; the BAL-model residuals computed with numpy
    must equal the reference-model residuals (oracle) of the converted problem up to the y sign."""
    import psba_amd
    from oracle_lib import Oracle
    rng = np.random.default_rng(5)
    nC, nP = 6, 40
    r = rng.normal(0, 0.3, (nC, 3))
    t = rng.normal(0, 0.5, (nC, 3)) + [0, 0, -6.0]  # scene in front of a -z looking camera
    f = rng.uniform(500, 1500, nC)
    X = rng.normal(0, 1.0, (nP, 3))
    obs = []
    for i in range(nP):
        for j in sorted(rng.choice(nC, size=rng.integers(2, nC + 1), replace=False)):
            obs.append((j, i, rng.normal(0, 100.0), rng.normal(0, 100.0)))
    order = rng.permutation(len(obs))  # BAL lists observations in any order
    bal = tmp_path / "problem.txt"
    with open(bal, "w") as g:
        g.write(f"{nC} {nP} {len(obs)}\n")
        for k in order:
            g.write("%d %d %.17g %.17g\n" % obs[k])
        for j in range(nC):
            for v in list(r[j]) + list(t[j]) + [f[j], 0.0, 0.0]:
                g.write("%.17g\n" % v)
        for v in X.reshape(-1):
            g.write("%.17g\n" % v)
    fc, fp = str(tmp_path / "cams.txt"), str(tmp_path / "pts.txt")
    assert psba_amd.convert_bal(str(bal), fc, fp) == 0.0
    prob = psba_amd.read_problem(fc, fp)
    assert (prob["nC"], prob["nP"], prob["nO"]) == (nC, nP, len(obs))
    assert np.array_equal(prob["K"][:, 0], f) and np.all(prob["K"][:, 1:] == [0, 0, 1, 0])
    ex = Oracle(prob).exQT().reshape(-1, 2)

    def rodrigues(rv, x):
        th = np.linalg.norm(rv)
        k = rv / th
        return x * np.cos(th) + np.cross(k, x) * np.sin(th) + k * (k @ x) * (1 - np.cos(th))

    for a in range(prob["nO"]):
        i, j = prob["iidx"][a], prob["jidx"][a]
        P = rodrigues(r[j], X[i]) + t[j]
        p = -P[:2] / P[2] * f[j]
        x, y = [(o[2], o[3]) for o in obs if o[0] == j and o[1] == i][0]
        want = np.array([x - p[0], -(y - p[1])])  # y axis flipped
        np.testing.assert_allclose(ex[a], want, rtol=0, atol=1e-9)


def test_sparse_pattern_is_the_covisibility_of_camera_pairs():
    """psba_sparse_pattern (host only): block tri(j) + k of the lower block triangle is flagged exactly when
    cameras j and k see a common point (every camera with an observation flags its diagonal block); the
    patterns of point shards OR to the pattern of the whole problem -- what the block-sparse route under
    a rank layout relies on."""
    from psba_amd import capi, synth
    prob = synth.make_problem(n_cams=30, n_pts=200, mean_track=3.5, seed=5, window=8)
    nC = prob["nC"]
    want = np.zeros((nC, nC), dtype=bool)
    ii, jj = prob["iidx"], prob["jidx"]
    for i in range(prob["nP"]):
        cams = jj[ii == i]
        want[np.ix_(cams, cams)] = True
    flags = capi.sparse_pattern(prob)
    for j in range(nC):
        for k in range(j + 1):
            assert bool(flags[j * (j + 1) // 2 + k]) == bool(want[j, k]), (j, k)
    parts = [capi.sparse_pattern(capi.shard_problem(prob, 3, r)) for r in range(3)]
    assert np.array_equal(np.maximum.reduce(parts), flags)


@pytest.mark.parametrize("n_cams,window", [(40, None), (300, 25)])
def test_owner_plan_lists_every_product_once(n_cams, window):
    """The owner route's static plan (schur_plan.cpp build_owner_plan; K2 for >= 2048 cameras, long tracks
    and the block-sparse S), replayed on the host: every product (a, b) of a point's observations, b <= a,
    sits in exactly one unit of the block (camera of a, camera of b); a block with several units is
    marked for atomic adds; the block list is the co-visible camera pairs plus every diagonal block in
    canonical order, and with a union pattern handed in (sharded points) the flagged blocks exist too."""
    from psba_amd import capi, synth
    prob = synth.make_problem(n_cams=n_cams, n_pts=500, mean_track=4.0, seed=31 + n_cams, window=window)
    ii, jj = prob["iidx"], prob["jidx"]
    pl = capi.owner_plan(prob["nC"], prob["nP"], ii, jj)
    want = {}
    start = 0
    for a in range(prob["nO"]):
        if a and ii[a] != ii[a - 1]:
            start = a
        for b in range(start, a + 1):
            want[(a, b)] = (int(jj[a]), int(jj[b]))
    assert pl["products"] == len(want)
    blocks = [tuple(x) for x in pl["blocks"]]
    assert blocks == sorted(blocks)  # canonical order: by j, then k
    present = set(want.values()) | {(j, j) for j in range(n_cams)}
    assert set(blocks) == present
    assert all(blocks[pl["diag_slot"][j]] == (j, j) for j in range(n_cams))
    seen = {}
    units_of_block = {}
    for w, (row0, ln) in enumerate(pl["waves"]):
        for lane in range(64):
            j, k, multi, slot = pl["units"][64 * w + lane]
            if slot < 0:
                continue
            assert blocks[slot] == (j, k)
            units_of_block.setdefault((j, k), []).append(multi)
            for t in range(ln):
                a, b = pl["prod"][row0 + t, lane]
                if a < 0:
                    continue
                assert want[(int(a), int(b))] == (j, k)
                assert (int(a), int(b)) not in seen
                seen[(int(a), int(b))] = 1
    assert len(seen) == len(want)
    for blk, ms in units_of_block.items():
        assert all(m == (1 if len(ms) > 1 else 0) for m in ms), blk
    # sharded points: a pattern with extra blocks puts them into the list (they stay empty here)
    pat = capi.sparse_pattern(prob).copy()
    extra = [(j, k) for j in range(n_cams) for k in range(j) if (j, k) not in present][:5]
    for j, k in extra:
        pat[j * (j + 1) // 2 + k] = 1
    pl2 = capi.owner_plan(prob["nC"], prob["nP"], ii, jj, pattern=pat)
    assert set(tuple(x) for x in pl2["blocks"]) == present | set(extra)


def _run_bench(extra_env, *argv, timeout=240):
    import subprocess, sys
    env = dict(os.environ, PSBA_BENCH_STUB="1", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` (no launcher, no WORLD_SIZE) must start: the parent
    spawns N fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- before importing
    the library or touching a GPU -- waits, and relays rank 0's single JSON line.  The GPU work is replaced
    by the stub (PSBA_BENCH_STUB=1: rendezvous, barrier and max-over-ranks still run, over gloo)."""
    import json
    out = _run_bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout          # exactly one line on stdout, and it is JSON
    d = json.loads(lines[0])
    assert d["stub"] is True and d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "strong"             # the default: ONE problem split over the ranks (configs[3])
    seen = sorted(tuple(x) for x in d["ranks_seen"])
    assert [s[0] for s in seen] == [0, 1] and [s[1] for s in seen] == [0, 1]  # distinct RANK / LOCAL_RANK
    assert seen[0][2] != seen[1][2]                                           # two processes
    assert abs(d["max_over_ranks"] - 0.002) < 1e-12                           # the MAX over ranks reached rank 0
    assert "rank 1 says hello" in out.stderr and "rank 1 says hello" not in out.stdout
    assert "stray line" in out.stderr           # anything else rank 0 printed went to stderr


def test_bench_parent_fails_when_a_rank_fails():
    out = _run_bench({"PSBA_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2")
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.strip()]  # no JSON line from a failed job
    out = _run_bench({"PSBA_BENCH_STUB_FAIL_RANK": "0"}, "--gpus", "2")
    assert out.returncode != 0


def test_bench_parent_does_not_load_the_gpu_library():
    """The spawning parent must not have initialised a GPU runtime: bench.py imports psba_amd (which dlopens
    the HIP library) only in load_library(), which the spawn path never reaches."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    top = src.split("def load_library")[0]
    assert "import psba_amd" not in top and "import torch" not in top
    main_src = src.split("def main():")[1]
    assert main_src.index("spawn_ranks(") < main_src.index("load_library()")


def test_bench_collective_bytes():
    """0.41 MB all-reduce at 52 cameras, 0.58 GB at cfg5 (packed lower block triangle)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.collective_bytes(52, 1, False)["allreduce_S_ea"] == 0
    v = b.collective_bytes(52, 4, False)
    assert v["allreduce_S_ea"] == 1378 * 288 and v["broadcast_factor_columns"] == 0
    c = b.collective_bytes(2000, 8, True)
    assert abs(c["allreduce_S_ea"] / 1e9 - 0.576) < 0.001
    assert abs(c["broadcast_factor_columns"] / 1e9 - 0.576) < 0.001


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
@pytest.mark.parametrize("n32,NB", [(2048, 256), (2080, 256), (12000, 384), (4128, 256), (1824, 256)])
def test_sharded_factorization_exchange_plan(n32, NB, nranks):
    """the RCCL column exchange of the sharded dense factorization (kernels_chol_graph.hip,
    chol_dist_comm) has only ever run with one rank, where every block's owner is rank 0 and the unpack branch
    never executes.  The owner / slot / length arithmetic it uses is one host function; here it is replayed for
    2, 3 and 8 ranks (and n32 mod 64 = 32, where the last block is half a block) against the sequence the
    three-handle GPU test performs by hand: every column >= the first super-panel's end travels exactly once, from
    the rank that updated it (block B mod nranks), in front of the super-panel that needs it, packed rows from the
    block's diagonal to the e_a row."""
    from psba_amd import capi
    assert n32 % 32 == 0
    seen = {}
    for J in range(0, n32, NB):
        JE = J + NB
        plan = capi.chol_dist_exchange_plan(n32, NB, nranks, JE)
        want = [] if JE >= n32 else list(range(JE // 64, (min(JE + NB, n32) + 63) // 64))
        assert [b for b, _, _, _ in plan] == want
        for k, (B, owner, slot, doubles) in enumerate(plan):
            assert owner == B % nranks and 0 <= owner < nranks
            assert slot == k and slot <= NB // 64            # the exchange buffer has NB / 64 + 1 slots
            cols = min(64, n32 - 64 * B)
            assert cols in (32, 64) and doubles == (n32 + 1 - 64 * B) * cols
            assert doubles <= (n32 + 1) * 64                 # fits a slot
            assert B not in seen
            seen[B] = JE
    # every 64-column block right of the first super-panel is exchanged exactly once, before the super-panel that holds it
    first = NB // 64 if NB % 64 == 0 else None
    if first is not None and n32 > NB:
        assert sorted(seen) == list(range(NB // 64, (n32 + 63) // 64))
        for B, JE in seen.items():
            assert JE <= 64 * B + 63 and 64 * B < JE + NB
