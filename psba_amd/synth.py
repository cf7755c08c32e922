"""Synthetic bundle-adjustment problems in the reference's parameterisation (host-side
workload generator for bench.py and tests; not part of the GPU path).

SURVEY.md section 8(d): the real Venice-52-64053 / Trafalgar-50-20431 point files are missing
from the reference checkout, so the benchmark uses *shaped* stand-ins: the named number of
cameras and points, track length k ~ min_track + Geometric with the named mean, k distinct
cameras drawn uniformly, points placed in front of their cameras, observations = projection of
the true point + N(0, 1 px), initial parameters = truth + small perturbation.  Every result made
with this generator is labelled "synthetic-shaped", never with the dataset's name alone.
"""
import os

import numpy as np

from .capi import Problem


def _quat_mul(a, b):
    """Hamilton product of scalar-first quaternions, broadcasting."""
    s = a[..., 0] * b[..., 0] - (a[..., 1:] * b[..., 1:]).sum(-1)
    v = a[..., :1] * b[..., 1:] + b[..., :1] * a[..., 1:] + np.cross(a[..., 1:], b[..., 1:])
    return np.concatenate([s[..., None], v], -1)


def _rotate(q, M):
    u, s = q[..., 1:], q[..., :1]
    w = s * M + np.cross(u, M)
    return (u * M).sum(-1, keepdims=True) * u + s * w + np.cross(u, w)


def project(K, q0, cams, M):
    """Projection model of the reference for matching rows of
    K[.,5], q0[.,4], cams[.,6], M[.,3]."""
    v = cams[..., :3]
    ql = np.concatenate([np.sqrt(1.0 - (v * v).sum(-1, keepdims=True)), v], -1)
    P = _rotate(_quat_mul(ql, q0), M) + cams[..., 3:]
    x = (K[..., 0] * P[..., 0] + K[..., 4] * P[..., 1] + K[..., 1] * P[..., 2]) / P[..., 2]
    y = (K[..., 0] * K[..., 3] * P[..., 1] + K[..., 2] * P[..., 2]) / P[..., 2]
    return np.stack([x, y], -1), P[..., 2]


def make_problem(n_cams, n_pts, mean_track, seed, min_track=2, max_track=None, noise_px=1.0,
                 cam_sigma=1e-3, pt_sigma=1e-2, focal=1000.0, shard=0, window=None):
    """Cameras on a circle of radius 10 looking at the origin; points in the unit ball.

    `shard` selects an independent stream for the points and observations while the cameras
    (and their initial perturbation) depend on `seed` only: shards 0..N-1 of one seed are the
    per-rank pieces of one N-times-larger problem over the same cameras.

    `window`: a point's cameras are drawn from `window` neighbours on the circle (around a random
    camera) instead of from all of them -- the co-visibility of a camera sequence: the reduced camera
    matrix S then has non-zero blocks only within that band (and its wrap-around corner)."""
    rng = np.random.default_rng([seed, 1 + shard])
    rng_cam = np.random.default_rng([seed, 0])
    max_track = n_cams if max_track is None else min(max_track, n_cams)
    if window is not None:
        max_track = min(max_track, int(window))
    # cameras: centre c_j on a circle, rotation taking the world z-axis towards the origin
    ang = 2 * np.pi * np.arange(n_cams) / n_cams
    C = np.stack([10 * np.cos(ang), 10 * np.sin(ang), 0.3 * np.sin(3 * ang)], -1)
    zc = -C / np.linalg.norm(C, axis=1, keepdims=True)
    up = np.tile([0.0, 0.0, 1.0], (n_cams, 1))
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc, axis=1, keepdims=True)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc], 1)  # rows = camera axes: P = R (M - C)
    # rotation matrix -> unit quaternion (scalar first, scalar >= 0), Shepperd's method
    q = np.empty((n_cams, 4))
    for j in range(n_cams):
        m = R[j]
        cand = [1 + m[0, 0] + m[1, 1] + m[2, 2], 1 + m[0, 0] - m[1, 1] - m[2, 2],
                1 - m[0, 0] + m[1, 1] - m[2, 2], 1 - m[0, 0] - m[1, 1] + m[2, 2]]
        c = int(np.argmax(cand))
        r = np.sqrt(cand[c]) * 2
        if c == 0:
            qq = [r / 4, (m[2, 1] - m[1, 2]) / r, (m[0, 2] - m[2, 0]) / r, (m[1, 0] - m[0, 1]) / r]
        elif c == 1:
            qq = [(m[2, 1] - m[1, 2]) / r, r / 4, (m[0, 1] + m[1, 0]) / r, (m[0, 2] + m[2, 0]) / r]
        elif c == 2:
            qq = [(m[0, 2] - m[2, 0]) / r, (m[0, 1] + m[1, 0]) / r, r / 4, (m[1, 2] + m[2, 1]) / r]
        else:
            qq = [(m[1, 0] - m[0, 1]) / r, (m[0, 2] + m[2, 0]) / r, (m[1, 2] + m[2, 1]) / r, r / 4]
        qq = np.array(qq) / np.linalg.norm(qq)
        q[j] = qq if qq[0] >= 0 else -qq
    t = -np.einsum("jrc,jc->jr", R, C)
    K = np.tile([focal, 0.0, 0.0, 1.0, 0.0], (n_cams, 1))
    # points uniform in the unit ball
    M = rng.normal(size=(n_pts, 3))
    M *= (rng.random(n_pts) ** (1 / 3) / np.linalg.norm(M, axis=1))[:, None]
    # track lengths: min_track + Geometric, mean matched, capped
    extra = max(mean_track - min_track, 1e-9)
    k = min_track + rng.geometric(1.0 / (1.0 + extra), size=n_pts) - 1
    k = np.clip(k, min_track, max_track)
    iidx = np.repeat(np.arange(n_pts, dtype=np.int32), k)
    # k distinct cameras per point, ascending
    jidx = np.empty(iidx.size, dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(k)])
    for kk in np.unique(k):
        rows = np.nonzero(k == kk)[0]
        if window is not None:
            keys = rng.random((rows.size, int(window)))
            offs = np.argpartition(keys, kk - 1, axis=1)[:, :kk]
            start = rng.integers(0, n_cams, size=(rows.size, 1))
            pick = np.sort((start + offs) % n_cams, axis=1).astype(np.int32)
        elif n_cams > 256 and 4 * kk < n_cams:
            # many cameras, short tracks: draw indices and redraw the few rows with repeats (the
            # dense rows x cameras key matrix below would be rows x n_cams doubles)
            pick = np.sort(rng.integers(0, n_cams, size=(rows.size, kk)), axis=1)
            while True:
                bad = np.nonzero((np.diff(pick, axis=1) == 0).any(axis=1))[0]
                if bad.size == 0:
                    break
                pick[bad] = np.sort(rng.integers(0, n_cams, size=(bad.size, kk)), axis=1)
            pick = pick.astype(np.int32)
        else:
            # rank random keys row-wise
            keys = rng.random((rows.size, n_cams))
            pick = np.sort(np.argpartition(keys, kk - 1, axis=1)[:, :kk], axis=1).astype(np.int32)
        dest = (off[rows][:, None] + np.arange(kk)[None, :]).reshape(-1)
        jidx[dest] = pick.reshape(-1)
    true_cams = np.concatenate([np.zeros((n_cams, 3)), t], 1)
    xy, depth = project(K[jidx], q[jidx], true_cams[jidx], M[iidx])
    assert np.all(depth > 0)
    impts = xy + rng.normal(0, noise_px, xy.shape)
    cams0 = true_cams + np.concatenate([rng_cam.normal(0, cam_sigma, (n_cams, 3)),
                                        rng_cam.normal(0, cam_sigma, (n_cams, 3))], 1)
    pts0 = M + rng.normal(0, pt_sigma, M.shape)
    return Problem(K=K, initrot=q, cams=cams0, pts=pts0, impts=impts, iidx=iidx, jidx=jidx,
                   nC=n_cams, nP=n_pts, nO=int(iidx.size))


def _rotmat(q):
    """Rotation matrices [n,3,3] of unit quaternions [n,4] (scalar first)."""
    s_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - s_ * z), 2 * (x * z + s_ * y)], -1),
                     np.stack([2 * (x * y + s_ * z), 1 - 2 * (x * x + z * z), 2 * (y * z - s_ * x)], -1),
                     np.stack([2 * (x * z - s_ * y), 2 * (y * z + s_ * x), 1 - 2 * (x * x + y * y)], -1)], 1)


def shaped_from_cams(cams_file, n_pts, mean_track, seed, shard=0, min_track=2, noise_px=1.0,
                     cam_sigma=1e-3, pt_sigma=1e-2, cluster=1):
    """SURVEY 8(d)'s stand-in for a dataset whose point file is missing: the REAL cameras of a
    12-column sba cams file (K5, unit quaternion, t; data/*-cams.txt), the named number of points,
    track length k ~ min_track + Geometric with the named mean (capped at the camera count), k
    distinct cameras drawn uniformly, the point at the centroid of those cameras' centres + noise,
    pushed along their mean optical axis until it has positive depth in all k views, observation
    = the reference's projection of the true point + N(0, 1 px), initial parameters = truth +
    a small perturbation.  Cameras (and their perturbation) depend on `seed` only, points and
    observations also on `shard` (shards of one seed = pieces of one larger problem).
    `cluster` > 1: runs of that many consecutive points share ONE camera set (track length and cameras drawn once
    per run) -- what real reconstructions look like in file order, where neighbouring points come from the same
    image pairs; SURVEY 8(d)'s uniform draw (cluster = 1) is the worst case for block locality in S."""
    a = np.loadtxt(cams_file, comments="#", ndmin=2)
    if a.shape[1] != 12:
        raise ValueError(f"{cams_file}: expected 12 columns (K5 q4 t3), got {a.shape[1]}")
    n_cams = a.shape[0]
    K = a[:, :5].copy()
    q = a[:, 5:9] / np.linalg.norm(a[:, 5:9], axis=1, keepdims=True)
    q[q[:, 0] < 0] *= -1.0  # quat2vec's sign convention (PSBA/misc.cpp:38-43)
    t = a[:, 9:12].copy()
    R = _rotmat(q)
    C = -np.einsum("jrc,jr->jc", R, t)  # camera centres: P = R M + t = R (M - C)
    axis = R[:, 2, :]                     # optical axes in world coordinates
    rng = np.random.default_rng([seed, 1 + shard])
    rng_cam = np.random.default_rng([seed, 0])
    extra = max(mean_track - min_track, 1e-9)
    n_runs = (n_pts + cluster - 1) // cluster
    k_run = min_track + rng.geometric(1.0 / (1.0 + extra), size=n_runs) - 1
    k_run = np.clip(k_run, min_track, n_cams)
    run_of = np.arange(n_pts) // cluster
    k = k_run[run_of]
    iidx = np.repeat(np.arange(n_pts, dtype=np.int32), k)
    jidx = np.empty(iidx.size, dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(k)])
    for kk in np.unique(k_run):
        runs = np.nonzero(k_run == kk)[0]
        keys = rng.random((runs.size, n_cams))
        pick_run = np.sort(np.argpartition(keys, kk - 1, axis=1)[:, :kk], axis=1).astype(np.int32)
        where = np.full(n_runs, -1)
        where[runs] = np.arange(runs.size)
        rows = np.nonzero(k == kk)[0]
        pick = pick_run[where[run_of[rows]]]
        dest = (off[rows][:, None] + np.arange(kk)[None, :]).reshape(-1)
        jidx[dest] = pick.reshape(-1)
    spread = float(np.linalg.norm(C.std(axis=0)))
    depth_min = 0.25 * spread
    true_cams = np.concatenate([np.zeros((n_cams, 3)), t], 1)
    noise = rng.normal(0, 0.5 * spread, (n_pts, 3))
    M = np.zeros((n_pts, 3))
    todo = np.arange(n_pts)
    for attempt in range(40):
        # centroid and mean optical axis of each point's cameras; push the point along that axis
        # (doubling) until every view has it at positive depth
        sel = np.zeros(n_pts, dtype=bool)
        sel[todo] = True
        obs = np.nonzero(sel[iidx])[0]
        ii, jj = iidx[obs], jidx[obs]
        cen = np.zeros((n_pts, 3))
        axm = np.zeros((n_pts, 3))
        np.add.at(cen, ii, C[jj])
        np.add.at(axm, ii, axis[jj])
        cen[todo] /= k[todo][:, None]
        axm[todo] /= np.maximum(np.linalg.norm(axm[todo], axis=1, keepdims=True), 1e-12)
        push = np.full(n_pts, 2.0 * spread)
        worst = np.full(n_pts, np.inf)
        for _ in range(8):
            Mp = cen + noise + push[:, None] * axm
            _, depth = project(K[jj], q[jj], true_cams[jj], Mp[ii])
            worst[todo] = np.inf
            np.minimum.at(worst, ii, depth)
            bad = todo[worst[todo] < depth_min]
            if bad.size == 0:
                break
            push[bad] *= 2.0
        good = todo[worst[todo] >= depth_min]
        M[good] = (cen + noise + push[:, None] * axm)[good]
        todo = todo[worst[todo] < depth_min]
        if todo.size == 0:
            break
        # cameras facing away from each other: draw another camera set (same track length) for the
        # points that could not be placed; late attempts take cameras looking the same way as a
        # random anchor camera
        for i in todo:
            if attempt < 30:
                pick = np.sort(rng.choice(n_cams, size=k[i], replace=False))
            else:
                anchor = rng.integers(n_cams)
                pick = np.sort(np.argsort(-(axis @ axis[anchor]))[: k[i]])
            jidx[off[i]: off[i + 1]] = pick
    if todo.size:
        raise ValueError("could not place every point in front of its cameras")
    xy, depth = project(K[jidx], q[jidx], true_cams[jidx], M[iidx])
    assert np.all(depth > 0)
    impts = xy + rng.normal(0, noise_px, xy.shape)
    cams0 = true_cams + np.concatenate([rng_cam.normal(0, cam_sigma, (n_cams, 3)),
                                        rng_cam.normal(0, cam_sigma, (n_cams, 3))], 1)
    pts0 = M + rng.normal(0, pt_sigma, M.shape)
    return Problem(K=K, initrot=q, cams=cams0, pts=pts0, impts=impts, iidx=iidx, jidx=jidx,
                   nC=n_cams, nP=n_pts, nO=int(iidx.size))


# the reference's bundled data/*.txt, copied as fixtures (the GPU box receives only this repository)
_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")


def venice_shaped(n_pts=64053, seed=0x5BA0 + 4, shard=0, cluster=1):
    """Venice-52-64053-shaped (BASELINE configs[3]): the 52 real cameras of
    data/Venice-52-64053-cams.txt, 64053 synthetic points, mean track 5.42.  cluster: see shaped_from_cams."""
    return shaped_from_cams(os.path.join(_DATA, "Venice-52-64053-cams.txt"), n_pts, 5.42, seed, shard=shard, cluster=cluster)


def trafalgar50_shaped(seed=0x5BA0 + 3, shard=0):
    """Trafalgar-50-20431-shaped (BASELINE configs[2]): the 50 real cameras of
    data/Trafalgar-50-20431-cams.txt, 20431 synthetic points, mean track 3.62."""
    return shaped_from_cams(os.path.join(_DATA, "Trafalgar-50-20431-cams.txt"), 20431, 3.62, seed, shard=shard)


def cfg5(n_pts=2_000_000, seed=0x5BA5, shard=0, n_cams=2000):
    """BASELINE configs[4] as SURVEY 8(d) specifies it: 2000 cameras on a circle of radius 10
    looking at the origin, f = 1000, points uniform in the unit ball, each seen by exactly 10
    cameras drawn uniformly (every camera has the whole ball at positive depth), 1 px noise,
    initial parameters = truth + N(0, 1e-3) (cameras) / N(0, 1e-2) (points); seed 0x5BA5.
    n_pts scales the problem down for time-bound runs (the camera count, hence the 12 000 x 12 000
    dense S, stays)."""
    return make_problem(n_cams, n_pts, 10.0, seed, min_track=10, max_track=10, shard=shard)
