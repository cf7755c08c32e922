"""ctypes wrapper around oracle/libpsba_oracle.so (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libpsba_oracle.so")
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class LmOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tr_handoff", C.c_int), ("verbose", C.c_int), ("log_cap", C.c_int),
                ("start_itno", C.c_int), ("init_mu", C.c_double)]


class LmResult(C.Structure):
    _fields_ = [("flag", C.c_int), ("iters", C.c_int), ("tries", C.c_int), ("init_err", C.c_double),
                ("final_err", C.c_double), ("mu0", C.c_double), ("n_log", C.c_int),
                ("t_linearize", C.c_double), ("t_schur", C.c_double), ("t_solve", C.c_double),
                ("t_backsub", C.c_double), ("t_cost", C.c_double)]


def _load():
    src = os.path.join(_ROOT, "oracle", "psba_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return C.CDLL(_SO)


_lib = _load()


def _sig(name, res, *args):
    f = getattr(_lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_i, _d = C.c_int, C.c_double
_exQT = _sig("orc_compute_exQT", None, _i, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp)
_jac = _sig("orc_compute_jacobiQT", None, _i, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp)
_U = _sig("orc_compute_U", None, _i, _i, _dp, _ip, _d, _dp, _dp)
_V = _sig("orc_compute_V", None, _i, _i, _i, _dp, _ip, _d, _dp, _dp)
_W = _sig("orc_compute_Wblks", None, _i, _dp, _dp, _d, _dp)
_g = _sig("orc_compute_g", None, _i, _i, _i, _d, _dp, _dp, _ip, _ip, _dp, _dp)
_maxuv = _sig("orc_maxElmOfUV", _d, _i, _dp)
_upd = _sig("orc_update_UV", None, _i, _i, _dp, _dp, _d)
_vinv = _sig("orc_compute_Vinv", _d, _i, _dp, _dp)
_Y = _sig("orc_compute_Yblks", None, _i, _ip, _dp, _dp, _dp)
_S = _sig("orc_compute_S", None, _i, _i, _i, _ip, _ip, _dp, _dp, _dp, _dp)
_ea = _sig("orc_compute_ea", None, _i, _i, _i, _ip, _ip, _dp, _dp, _dp)
_chol = _sig("orc_chol_solve", _d, _i, _dp, _dp, _dp)
_eb = _sig("orc_compute_eb", None, _i, _i, _i, _ip, _ip, _dp, _dp, _dp, _dp)
_dpb = _sig("orc_compute_dpb", None, _i, _i, _dp, _dp, _dp)
_lm = _sig("orc_levmar", _i, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.POINTER(LmOpts),
           C.POINTER(LmResult), C.c_void_p)


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


class Oracle:
    """One problem instance; methods mirror the reference's sba_func.h verbs."""

    def __init__(self, prob):
        self.nC, self.nP, self.nO = int(prob["nC"]), int(prob["nP"]), int(prob["nO"])
        self.K = _c(prob["K"]).reshape(-1)
        self.initrot = _c(prob["initrot"]).reshape(-1)
        self.impts = _c(prob["impts"]).reshape(-1)
        self.iidx = _c(prob["iidx"], np.int32)
        self.jidx = _c(prob["jidx"], np.int32)
        self.cams = _c(prob["cams"]).reshape(-1).copy()
        self.pts = _c(prob["pts"]).reshape(-1).copy()
        self.nA, self.nB = 6 * self.nC, 3 * self.nP
        self.nT = self.nA + self.nB

    def exQT(self, cams=None, pts=None):
        ex = np.empty(2 * self.nO)
        _exQT(self.nO, self.K, self.impts, self.initrot, self.cams if cams is None else _c(cams).reshape(-1),
              self.pts if pts is None else _c(pts).reshape(-1), self.iidx, self.jidx, ex)
        return ex

    def jacobiQT(self):
        JA, JB = np.empty(12 * self.nO), np.empty(6 * self.nO)
        _jac(self.nO, self.K, self.initrot, self.cams, self.pts, self.iidx, self.jidx, JA, JB)
        return JA, JB

    def linearize(self, coeff=1.0, coeff_g=1.0):
        """jacobiQT + U + V + Wblks + g (levmar.cpp:103-108). Returns dict."""
        ex = self.exQT()
        JA, JB = self.jacobiQT()
        U, V, UVdiag = np.empty(36 * self.nC), np.empty(9 * self.nP), np.empty(self.nT)
        W, g = np.empty(18 * self.nO), np.empty(self.nT)
        _U(self.nC, self.nO, JA, self.jidx, coeff, U, UVdiag)
        _V(self.nC, self.nP, self.nO, JB, self.iidx, coeff, V, UVdiag)
        _W(self.nO, JA, JB, coeff, W)
        _g(self.nC, self.nP, self.nO, coeff_g, JA, JB, self.iidx, self.jidx, ex, g)
        return dict(ex=ex, JA=JA, JB=JB, U=U, V=V, UVdiag=UVdiag, W=W, g=g, maxdiag=_maxuv(self.nT, UVdiag))

    def schur(self, lin, mu):
        """update_UV + Vinv + Yblks + S + ea (levmar.cpp:126-131)."""
        U, V = lin["U"].copy(), lin["V"].copy()
        _upd(self.nC, self.nP, U, V, mu)
        Vinv = np.empty(9 * self.nP)
        ret = _vinv(self.nP, V, Vinv)
        Y = np.empty(18 * self.nO)
        _Y(self.nO, self.iidx, lin["W"], Vinv, Y)
        S = np.empty(self.nA * self.nA)
        _S(self.nC, self.nP, self.nO, self.iidx, self.jidx, U, Y, lin["W"], S)
        eab = np.zeros(self.nT)
        _ea(self.nC, self.nP, self.nO, self.iidx, self.jidx, Y, lin["g"], eab)
        return dict(Ustar=U, Vstar=V, Vinv=Vinv, vinv_ret=ret, Y=Y, S=S.reshape(self.nA, self.nA), eab=eab)

    def solve(self, lin, sch):
        """SPDinv + matVec_mul + eb + dpb (levmar.cpp:134-155). Returns (status, dp, eab)."""
        S = sch["S"].copy().reshape(-1)
        dp = np.zeros(self.nT)
        eab = sch["eab"].copy()
        ret = _chol(self.nA, S, eab[: self.nA].copy(), dp)
        if ret != 0.0:
            return ret, dp, eab
        _eb(self.nC, self.nP, self.nO, self.iidx, self.jidx, lin["W"], dp, lin["g"], eab)
        _dpb(self.nC, self.nP, sch["Vinv"], eab, dp)
        return ret, dp, eab

    def levmar(self, max_iter=50, tr_handoff=False, log_cap=512, verbose=False, start_itno=0, init_mu=0.0):
        opts = LmOpts(max_iter, int(tr_handoff), int(verbose), log_cap, start_itno, init_mu)
        res = LmResult()
        log = np.zeros((max(log_cap, 1), 5))
        _lm(self.nC, self.nP, self.nO, self.K, self.impts, self.initrot, self.cams, self.pts, self.iidx,
            self.jidx, C.byref(opts), C.byref(res), log.ctypes.data_as(C.c_void_p))
        return res, log[: res.n_log].copy()


class TrOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("start_itno", C.c_int), ("verbose", C.c_int), ("log_cap", C.c_int),
                ("init_lambda", C.c_double)]


class TrResult(C.Structure):
    _fields_ = [("flag", C.c_int), ("iters", C.c_int), ("tries", C.c_int), ("chol_fail", C.c_int),
                ("init_err", C.c_double), ("final_err", C.c_double), ("lambda_", C.c_double),
                ("delta", C.c_double), ("n_log", C.c_int)]


_tr = _sig("orc_trust_region", _i, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.POINTER(TrOpts),
           C.POINTER(TrResult), C.c_void_p)
_jmul = _sig("orc_compute_Jmultiply", None, _i, _i, _dp, _dp, _ip, _ip, _dp, _dp)
_cholmod = _sig("orc_cholmod", None, _i, _dp, _dp, _dp)
_delta_beta = _sig("orc_get_delta_beta", None, _i, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double))


def cholmod(A):
    """orc_cholmod: (L, E, delta, beta) of the symmetric matrix A."""
    A = _c(A)
    n = A.shape[0]
    L, E = np.empty((n, n)), np.empty(n)
    d, b = C.c_double(), C.c_double()
    _delta_beta(n, A.reshape(-1), C.byref(d), C.byref(b))
    _cholmod(n, A.reshape(-1), L.reshape(-1), E)
    return L, E, d.value, b.value


def trust_region(o, max_iter=50, start_itno=0, log_cap=512, verbose=False, init_lambda=0.0):
    """orc_trust_region on the Oracle instance o (its cams / pts are updated in place)."""
    opts = TrOpts(max_iter, start_itno, int(verbose), log_cap, init_lambda)
    res = TrResult()
    log = np.zeros((max(log_cap, 1), 6))
    _tr(o.nC, o.nP, o.nO, o.K, o.impts, o.initrot, o.cams, o.pts, o.iidx, o.jidx, C.byref(opts), C.byref(res),
        log.ctypes.data_as(C.c_void_p))
    return res, log[: res.n_log].copy()


def solve_like_main(o, max_total=50):
    """The reference driver's alternation (PSBA/main.cpp:193-208): levmar() until it hands over
    (ITER_TURN_TO_TR), trust_region() until it hands back (ITER_TURN_TO_LM), sharing itno.
    Returns the list of (which, result) in order."""
    out, itno = [], 0
    while True:
        res, _ = o.levmar(max_iter=max_total, tr_handoff=True, start_itno=itno)
        out.append(("lm", res))
        itno = res.iters
        if res.flag != 2:
            break
        res, _ = trust_region(o, max_iter=max_total, start_itno=itno)
        out.append(("tr", res))
        itno = res.iters
        if res.flag != 1:
            break
    return out


_SO_OMP = os.path.join(_ROOT, "oracle", "libpsba_oracle_omp.so")


def levmar_all_cores(prob, max_iter=10, tr_handoff=False):
    """orc_levmar of the OpenMP build of the same source (oracle/Makefile): only for bench.py's
    all-core CPU baseline -- no test or pin uses it.  Returns (result, threads used)."""
    src = os.path.join(_ROOT, "oracle", "psba_oracle.c")
    if (not os.path.exists(_SO_OMP)) or os.path.getmtime(_SO_OMP) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    # as many threads as this process may really use: the affinity mask, capped at the CPU share of
    # a one-GPU box (16) -- an OpenMP team larger than the CPUs it gets spins on its barriers
    # (seen: minutes instead of seconds), hence also the passive wait policy
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(n, 16))))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("OMP_DYNAMIC", "FALSE")
    lib = C.CDLL(_SO_OMP)
    f = lib.orc_levmar
    f.restype = _lm.restype
    f.argtypes = _lm.argtypes
    lib.orc_threads.restype = C.c_int
    o = Oracle(prob)
    opts = LmOpts(max_iter, int(tr_handoff), 0, 0, 0, 0.0)
    res = LmResult()
    log = np.zeros((1, 5))
    f(o.nC, o.nP, o.nO, o.K, o.impts, o.initrot, o.cams, o.pts, o.iidx, o.jidx, C.byref(opts), C.byref(res),
      log.ctypes.data_as(C.c_void_p))
    return res, int(lib.orc_threads())



# ---- free intrinsics: the oracle's twin on 11-parameter camera blocks (parity unpinned) ----
_fk_ex = _sig("orc_fk_exQT", None, _i, _dp, _dp, _dp, _dp, _ip, _ip, _dp)
_fk_jac = _sig("orc_fk_jacobi", None, _i, _dp, _dp, _dp, _ip, _ip, _dp, _dp)
_fk_normal = _sig("orc_fk_normal", _d, _i, _i, _i, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp)
_fk_lm = _sig("orc_fk_levmar", _i, _i, _i, _i, _dp, _dp, _dp, _dp, _ip, _ip, C.POINTER(LmOpts), C.POINTER(LmResult), C.c_void_p)


class OracleFreeK:
    """Camera block (fu, u0, v0, ar, s | v | t): cams11[nC, 11] from the problem's K and cams."""

    def __init__(self, prob):
        self.nC, self.nP, self.nO = int(prob["nC"]), int(prob["nP"]), int(prob["nO"])
        self.initrot = _c(prob["initrot"]).reshape(-1)
        self.impts = _c(prob["impts"]).reshape(-1)
        self.iidx, self.jidx = _c(prob["iidx"], np.int32), _c(prob["jidx"], np.int32)
        self.cams = np.hstack([_c(prob["K"]).reshape(self.nC, 5), _c(prob["cams"]).reshape(self.nC, 6)]).reshape(-1).copy()
        self.pts = _c(prob["pts"]).reshape(-1).copy()
        self.nA, self.nT = 11 * self.nC, 11 * self.nC + 3 * self.nP

    def exQT(self, cams=None, pts=None):
        ex = np.empty(2 * self.nO)
        _fk_ex(self.nO, self.impts, self.initrot, self.cams if cams is None else _c(cams).reshape(-1),
               self.pts if pts is None else _c(pts).reshape(-1), self.iidx, self.jidx, ex)
        return ex

    def jacobi(self):
        JA, JB = np.empty(22 * self.nO), np.empty(6 * self.nO)
        _fk_jac(self.nO, self.initrot, self.cams, self.pts, self.iidx, self.jidx, JA, JB)
        return JA.reshape(-1, 2, 11), JB.reshape(-1, 2, 3)

    def normal(self):
        """(||e||^2, J^T J dense, J^T e)"""
        N, g = np.empty(self.nT * self.nT), np.empty(self.nT)
        cost = _fk_normal(self.nC, self.nP, self.nO, self.impts, self.initrot, self.cams, self.pts, self.iidx, self.jidx, N, g)
        return cost, N.reshape(self.nT, self.nT), g

    def levmar(self, max_iter=20, log_cap=256, init_mu=0.0):
        opts = LmOpts(max_iter, 0, 0, log_cap, 0, init_mu)
        res = LmResult()
        log = np.zeros((max(log_cap, 1), 5))
        _fk_lm(self.nC, self.nP, self.nO, self.impts, self.initrot, self.cams, self.pts, self.iidx, self.jidx,
               C.byref(opts), C.byref(res), log.ctypes.data_as(C.c_void_p))
        return res, log[: res.n_log].copy()
