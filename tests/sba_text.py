"""Independent (test-only) numpy reader for the sba text format.

Follows SURVEY.md Appendix C and the behaviour of /root/reference PSBA/readparams.cpp:444-519 +
PSBA/misc.cpp:21-49 (quat2vec) + PSBA/main.cpp:131-149 (zero the local rotation, split K).
Used to cross-check the product's own reader (psba_amd/csrc/sba_io.cpp) and to feed the oracle.
"""
import numpy as np

KK = np.array([851.57945, 330.24755, 262.19500, 1.00169, 0.0])


def _rows(path):
    out = []
    with open(path) as f:
        for line in f:
            s = line.strip()
            if not s or s.startswith("#"):
                continue
            out.append(s.split())
    return out


def read_problem(cams_path, pts_path):
    """Returns dict(K[nC,5], initrot[nC,4], cams[nC,6], pts[nP,3], impts[nO,2], iidx, jidx)."""
    crow = _rows(cams_path)
    nC = len(crow)
    ncol = len(crow[0])
    assert ncol in (7, 12), "7 (q,t) or 12 (K5,q,t) columns expected"
    cam = np.array(crow, dtype=np.float64)
    if ncol == 7:
        K = np.tile(KK, (nC, 1))
        q = cam[:, 0:4]
        t = cam[:, 4:7]
    else:
        K = cam[:, 0:5].copy()
        q = cam[:, 5:9]
        t = cam[:, 9:12]
    # quat2vec: normalise, make the scalar part non-negative, keep the vector part
    mag = np.sqrt((q * q).sum(1))
    sg = np.where(q[:, 0] >= 0.0, 1.0, -1.0)
    v = q[:, 1:4] * (sg / mag)[:, None]
    initrot = np.empty((nC, 4))
    initrot[:, 1:4] = v
    initrot[:, 0] = np.sqrt(1.0 - v[:, 0] * v[:, 0] - v[:, 1] * v[:, 1] - v[:, 2] * v[:, 2])
    cams = np.zeros((nC, 6))
    cams[:, 3:6] = t
    pts, impts, iidx, jidx = [], [], [], []
    for i, r in enumerate(_rows(pts_path)):
        pts.append([float(r[0]), float(r[1]), float(r[2])])
        nfr = int(r[3])
        per = (len(r) - 4) // nfr  # 3 without covariance, 6 tri, 7 full
        obs = []
        for f in range(nfr):
            base = 4 + per * f
            obs.append((int(r[base]), float(r[base + 1]), float(r[base + 2])))
        # the reference stores impts in file order but indexes in camera-ascending order and so
        # silently requires ascending frame ids; sort to make that explicit
        obs.sort(key=lambda o: o[0])
        for j, x, y in obs:
            iidx.append(i)
            jidx.append(j)
            impts.append([x, y])
    return dict(
        K=np.ascontiguousarray(K), initrot=initrot, cams=cams,
        pts=np.array(pts, dtype=np.float64), impts=np.array(impts, dtype=np.float64),
        iidx=np.array(iidx, dtype=np.int32), jidx=np.array(jidx, dtype=np.int32),
        nC=nC, nP=len(pts), nO=len(iidx))
