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
    camera-row group, at its block's position; in a row of 16 item slots no LDS bank pair is
    hit more than twice; workgroup lists are equally long within a group."""
    from psba_amd import capi
    plan = capi.schur_plan(nC, nP, iidx, jidx)
    assert plan["groups"] >= 1
    items, wg, pos, glo = plan["items"], plan["wg"], plan["blockpos"], plan["glo"]
    ptr = np.searchsorted(iidx, np.arange(nP + 1))
    tri = lambda j: j * (j + 1) // 2
    # block positions: a bijection into the padded partition of the block's group
    for g in range(plan["groups"]):
        b0, b1 = tri(int(glo[g])), tri(int(glo[g + 1]))
        ps = pos[b0:b1]
        assert len(set(ps.tolist())) == b1 - b0 and ps.min() >= 0
        nblk = {int(w[1]) for w in wg if w[0] == g}
        assert len(nblk) == 1 and ps.max() < nblk.pop() <= (b1 - b0 + 15) // 16 * 16
    seen = set()
    slabs = set()
    for w in wg:
        g, nblk, obs0, pt0, s0, s1, slab = (int(x) for x in w)
        assert s0 % 16 == 0 and s1 % 16 == 0 and (g, slab) not in slabs
        slabs.add((g, slab))
        it = items[s0:s1]
        live = it != np.uint64(0xFFFFFFFFFFFFFFFF)
        a = obs0 + (it & np.uint64(0x3FFFF)).astype(np.int64)
        i = pt0 + ((it >> np.uint64(18)) & np.uint64(0xFFFF)).astype(np.int64)
        boff = ((it >> np.uint64(34)) & np.uint64(0x7FF)).astype(np.int64)
        p = ((it >> np.uint64(45)) & np.uint64(0x3FF)).astype(np.int64)
        for r in range(0, len(it), 16):
            q = p[r:r + 16][live[r:r + 16]] % 16
            assert np.bincount(q, minlength=16).max() <= 2  # a bank pair at most twice per row
        a, i, boff, p = a[live], i[live], boff[live], p[live]
        b = a - boff
        assert (iidx[a] == i).all() and (iidx[b] == i).all() and (b >= ptr[i]).all()
        ja, jb = jidx[a].astype(np.int64), jidx[b].astype(np.int64)
        assert ((ja >= glo[g]) & (ja < glo[g + 1])).all()
        assert (pos[ja * (ja + 1) // 2 + jb] == p).all()
        keys = set(zip(a.tolist(), b.tolist()))
        assert len(keys) == len(a) and not (keys & seen)
        seen |= keys
    want = int(((np.arange(len(iidx)) - ptr[iidx]) + 1).sum())
    assert len(seen) == want == plan["products"]
    # balance: item counts of the workgroups of one group differ by at most one
    for g in range(plan["groups"]):
        n = [int(((items[int(w[4]):int(w[5])]) != np.uint64(0xFFFFFFFFFFFFFFFF)).sum()) for w in wg if w[0] == g]
        assert max(n) - min(n) <= 1
    return plan


def test_schur_plan_small(problems):
    for name in ("7cams", "54cams"):
        pr = problems[name]
        _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))


def test_schur_plan_venice_shaped():
    from psba_amd import synth
    pr = synth.venice_shaped()
    plan = _check_plan(pr["nC"], pr["nP"], np.asarray(pr["iidx"]), np.asarray(pr["jidx"]))
    # the bank-pair schedule should cost little padding on a realistic problem
    assert plan["products"] / len(plan["items"]) > 0.75
    assert len(plan["wg"]) == 256
