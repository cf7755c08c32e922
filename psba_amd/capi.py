"""ctypes binding of include/psba_hip.h (one method per entry point, same names)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PSBA_LIB: another build of the library (psba_amd/libpsba_hip_exp.so = PSBA_BUILD_EXPERIMENTS=1 python psba_amd/build.py)
lib_path = os.environ.get("PSBA_LIB") or os.path.join(_HERE, "libpsba_hip.so")

if not os.path.exists(lib_path):
    raise ImportError(
        f"{lib_path} is missing: build the HIP extension first (python -m psba_amd.build, or "
        "__graft_entry__.build()). psba_amd has no CPU fallback.")
lib = C.CDLL(lib_path)

PSBA_OK, PSBA_NOT_SPD, PSBA_SINGULAR_V = 0, 1, 2
PARAMS_CUR, PARAMS_NEW = 0, 1
ITER_TURN_TO_LM, ITER_TURN_TO_TR, ITER_CONTINUE, ITER_ERR = 1, 2, 3, 4
ITER_DP_NO_CHANGE, ITER_ERR_SMALL_ENOUGH, ITER_PASS = 5, 6, 7
K_LINEARIZE, K_SCHUR, K_CHOLESKY, K_BACKSUB, K_RESIDUAL, K_ALLREDUCE, K_SCHUR_REDUCE = range(7)
KERNEL_NAMES = ["linearize", "schur", "cholesky", "backsub", "residual", "allreduce", "schur_reduce"]

_h = C.c_void_p
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class TryScalars(C.Structure):
    _fields_ = [("status", C.c_int), ("dp_l2", C.c_double), ("gain_den", C.c_double),
                ("new_cost", C.c_double), ("newp_l2", C.c_double)]


class LmOptions(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tr_handoff", C.c_int), ("verbose", C.c_int),
                ("log_cap", C.c_int), ("start_itno", C.c_int), ("init_mu", C.c_double)]


class LmResult(C.Structure):
    _fields_ = [("flag", C.c_int), ("iters", C.c_int), ("tries", C.c_int), ("init_err", C.c_double),
                ("final_err", C.c_double), ("mu0", C.c_double), ("mu_final", C.c_double),
                ("n_log", C.c_int), ("seconds", C.c_double), ("pcg_unconverged", C.c_int)]


class TrOptions(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("start_itno", C.c_int), ("verbose", C.c_int), ("log_cap", C.c_int),
                ("init_lambda", C.c_double)]


class TrResult(C.Structure):
    _fields_ = [("flag", C.c_int), ("iters", C.c_int), ("tries", C.c_int), ("chol_fail", C.c_int),
                ("init_err", C.c_double), ("final_err", C.c_double), ("lambda_", C.c_double),
                ("delta", C.c_double), ("n_log", C.c_int), ("seconds", C.c_double)]


class SolveResult(C.Structure):
    _fields_ = [("flag", C.c_int), ("iters", C.c_int), ("lm_calls", C.c_int), ("tr_calls", C.c_int),
                ("init_err", C.c_double), ("final_err", C.c_double), ("seconds", C.c_double)]


class CProblem(C.Structure):
    _fields_ = [("nCams", C.c_int), ("n3Dpts", C.c_int), ("n2Dprojs", C.c_int), ("Kparas", _dp),
                ("impts", _dp), ("initrot", _dp), ("camsEx", _dp), ("pts3D", _dp), ("iidx", _ip),
                ("jidx", _ip)]


# every symbol include/psba_hip.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("psba_create", C.c_int, [C.c_int, C.POINTER(_h)]),
    ("psba_destroy", C.c_int, [_h]),
    ("psba_last_error", C.c_char_p, [_h]),
    ("psba_version", C.c_char_p, []),
    ("psba_upload_problem", C.c_int, [_h, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip]),
    ("psba_set_params", C.c_int, [_h, _dp, _dp]),
    ("psba_reset_params", C.c_int, [_h]),
    ("psba_get_params", C.c_int, [_h, C.c_int, _dp, _dp]),
    ("psba_get_dims", C.c_int, [_h, _ip, _ip, _ip]),
    ("psba_schur_path", C.c_int, [_h, C.POINTER(C.c_int)]),
    ("psba_residual", C.c_int, [_h, C.c_int, _dp]),
    ("psba_linearize", C.c_int, [_h, C.c_double, C.c_double]),
    ("psba_max_diag", C.c_int, [_h, _dp]),
    ("psba_begin", C.c_int, [_h, C.c_double, C.c_double, _dp, _dp]),
    ("psba_schur_assemble", C.c_int, [_h, C.c_double]),
    ("psba_schur_reduce", C.c_int, [_h]),
    ("psba_schur_solve", C.c_int, [_h]),
    ("psba_backsub", C.c_int, [_h, C.c_double, C.POINTER(TryScalars)]),
    ("psba_backsub_async", C.c_int, [_h, C.c_double]),
    ("psba_linearize_ahead", C.c_int, [_h]),
    ("psba_backsub_wait", C.c_int, [_h, C.POINTER(TryScalars)]),
    ("psba_accept", C.c_int, [_h]),
    ("psba_compute_exQT", C.c_int, [_h, C.c_int, _dp]),
    ("psba_compute_jacobiQT", C.c_int, [_h, _dp, _dp]),
    ("psba_compute_U", C.c_int, [_h, C.c_double, _dp]),
    ("psba_compute_V", C.c_int, [_h, C.c_double, _dp]),
    ("psba_maxElmOfUV", C.c_int, [_h, _dp]),
    ("psba_update_UV", C.c_int, [_h, C.c_double, _dp, _dp]),
    ("psba_restore_UVdiag", C.c_int, [_h]),
    ("psba_compute_Vinv", C.c_int, [_h, _dp]),
    ("psba_compute_Wblks", C.c_int, [_h, C.c_double, _dp]),
    ("psba_compute_Yblks", C.c_int, [_h, _dp]),
    ("psba_compute_S", C.c_int, [_h, _dp]),
    ("psba_compute_g", C.c_int, [_h, C.c_double, _dp]),
    ("psba_compute_ea", C.c_int, [_h, _dp]),
    ("psba_SPDinv_matVec", C.c_int, [_h, _dp]),
    ("psba_compute_eb", C.c_int, [_h, _dp]),
    ("psba_compute_dpb", C.c_int, [_h, _dp]),
    ("psba_compute_newp", C.c_int, [_h, _dp]),
    ("psba_update_p", C.c_int, [_h, _dp]),
    ("psba_lm_default_options", None, [C.POINTER(LmOptions)]),
    ("psba_levmar", C.c_int, [_h, C.POINTER(LmOptions), C.POINTER(LmResult), _dp]),
    ("psba_partition_points", C.c_int, [C.c_int, _ip, C.c_int, C.c_int, _ip]),
    ("psba_comm_unique_id", C.c_int, [C.c_void_p]),
    ("psba_comm_init", C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    ("psba_comm_rank", C.c_int, [_h, _ip, _ip]),
    ("psba_set_rank_layout", C.c_int, [_h, C.c_int, C.c_int]),
    ("psba_reduce_buffer_size", C.c_int, [_h, C.POINTER(C.c_longlong)]),
    ("psba_get_reduce_buffer", C.c_int, [_h, _dp]),
    ("psba_set_reduce_buffer", C.c_int, [_h, _dp]),
    ("psba_read_problem", C.c_int, [C.c_char_p, C.c_char_p, _dp, C.POINTER(CProblem)]),
    ("psba_free_problem", None, [C.POINTER(CProblem)]),
    ("psba_profile_enable", C.c_int, [_h, C.c_int]),
    ("psba_profile_reset", C.c_int, [_h]),
    ("psba_profile_get", C.c_int, [_h, C.c_int, _dp, _ip]),
    ("psba_algorithmic_bytes", C.c_int, [_h, C.c_int, _dp]),
    ("psba_compute_Jmultiply", C.c_int, [_h, _dp, _dp]),
    ("psba_jmul_dots", C.c_int, [_h, _dp, _dp, _dp]),
    ("psba_get_gradient", C.c_int, [_h, _dp]),
    ("psba_get_dp", C.c_int, [_h, _dp]),
    ("psba_set_step", C.c_int, [_h, _dp]),
    ("psba_cholmod_lambda", C.c_int, [_h, C.c_int, _dp, _dp]),
    ("psba_tr_default_options", None, [C.POINTER(TrOptions)]),
    ("psba_trust_region", C.c_int, [_h, C.POINTER(TrOptions), C.POINTER(TrResult), _dp]),
    ("psba_solve", C.c_int, [_h, C.c_int, C.c_int, C.POINTER(SolveResult)]),
    ("psba_write_problem", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int]),
    ("psba_convert_bal", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _dp]),
    ("psba_allreduce_scalars", C.c_int, [_h, _dp, C.c_int]),
    ("psba_set_solver", C.c_int, [_h, C.c_int, C.c_double, C.c_int]),
    ("psba_pcg_info", C.c_int, [_h, _ip, _dp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    ("psba_get_sparse_S", C.c_int, [_h, _ip, _dp, _dp]),
    ("psba_set_sparse_S", C.c_int, [_h, _dp, _dp]),
    ("psba_sparse_pattern", C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(C.c_ubyte)]),
    ("psba_set_sparse_pattern", C.c_int, [_h, C.POINTER(C.c_ubyte), C.c_longlong]),
    ("psba_set_camera_model", C.c_int, [_h, C.c_int]),
    ("psba_camera_block", C.c_int, [_h, _ip]),
    ("psba_chol_dist_exchange_plan", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.c_int]),
    ("psba_chol_dist_shape", C.c_int, [_h, _ip, _ip, _ip]),
    ("psba_chol_dist_begin", C.c_int, [_h]),
    ("psba_chol_dist_superpanel", C.c_int, [_h, C.c_int]),
    ("psba_chol_dist_block", C.c_int, [_h, C.c_int, C.c_int, _dp, C.POINTER(C.c_longlong)]),
    ("psba_chol_dist_finish", C.c_int, [_h]),
    ("psba_schur_plan_create", C.c_void_p, [C.c_int, C.c_int, C.c_int, _ip, _ip]),
    ("psba_schur_plan_info", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    ("psba_schur_plan_copy", C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_longlong), _ip, _ip]),
    ("psba_schur_plan_destroy", None, [C.c_void_p]),
    ("psba_owner_plan_create", C.c_void_p, [C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(C.c_ubyte)]),
    ("psba_owner_plan_info", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    ("psba_owner_plan_copy", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), _ip, _ip, _ip, _ip]),
    ("psba_owner_plan_destroy", None, [C.c_void_p]),
]
# only in a library built with PSBA_BUILD_EXPERIMENTS=1 (round 3's rejected ring route and its test hooks)
EXPERIMENT_SIGNATURES = [
    ("psba_ring_plan_create", C.c_void_p, [C.c_int, C.c_int, C.c_int, _ip, _ip]),
    ("psba_ring_plan_info", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    ("psba_ring_plan_copy", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), _ip, C.POINTER(C.c_uint), _ip, _ip, _ip, _ip, _ip]),
    ("psba_ring_plan_destroy", None, [C.c_void_p]),
]
HAS_EXPERIMENTS = hasattr(lib, "psba_ring_plan_create")
for _name, _res, _args in SIGNATURES + (EXPERIMENT_SIGNATURES if HAS_EXPERIMENTS else []):
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


class PsbaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"psba error {code}: {msg}")
        self.code = code


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


class Problem(dict):
    """K[nC,5] initrot[nC,4] cams[nC,6] pts[nP,3] impts[nO,2] iidx[nO] jidx[nO] nC nP nO."""


def sparse_pattern(prob):
    """psba_sparse_pattern (host only): one byte per block tri(j) + k of the lower block triangle."""
    nC = int(prob["nC"])
    flags = np.zeros(nC * (nC + 1) // 2, dtype=np.uint8)
    ii = np.ascontiguousarray(prob["iidx"], dtype=np.int32)
    jj = np.ascontiguousarray(prob["jidx"], dtype=np.int32)
    rc = lib.psba_sparse_pattern(nC, int(prob["nP"]), int(prob["nO"]), _i(ii), _i(jj),
                                 flags.ctypes.data_as(C.POINTER(C.c_ubyte)))
    if rc != 0:
        raise PsbaError(rc, "psba_sparse_pattern failed")
    return flags


def read_problem(cams_file, pts_file, fixedK=None):
    """psba_read_problem -> Problem (host only, works without a GPU)."""
    cp = CProblem()
    k = None if fixedK is None else _c(fixedK)
    rc = lib.psba_read_problem(os.fsencode(cams_file), os.fsencode(pts_file), _d(k), C.byref(cp))
    if rc != 0:
        raise PsbaError(rc, f"psba_read_problem({cams_file}, {pts_file}) failed")
    nC, nP, nO = cp.nCams, cp.n3Dpts, cp.n2Dprojs
    arr = lambda p, n, shape: np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(shape)
    out = Problem(K=arr(cp.Kparas, 5 * nC, (nC, 5)), initrot=arr(cp.initrot, 4 * nC, (nC, 4)),
                  cams=arr(cp.camsEx, 6 * nC, (nC, 6)), pts=arr(cp.pts3D, 3 * nP, (nP, 3)),
                  impts=arr(cp.impts, 2 * nO, (nO, 2)), iidx=arr(cp.iidx, nO, (nO,)).astype(np.int32),
                  jidx=arr(cp.jidx, nO, (nO,)).astype(np.int32), nC=nC, nP=nP, nO=nO)
    lib.psba_free_problem(C.byref(cp))
    return out


def write_problem(cams_file, pts_file, prob, cams=None, pts=None, with_K=True):
    """psba_write_problem: the problem's data with the given (default: its own) parameters."""
    c = _c(prob["cams"] if cams is None else cams)
    p = _c(prob["pts"] if pts is None else pts)
    K, rot, im = _c(prob["K"]), _c(prob["initrot"]), _c(prob["impts"])
    ii, jj = _c(prob["iidx"], np.int32), _c(prob["jidx"], np.int32)
    rc = lib.psba_write_problem(cams_file.encode(), pts_file.encode(), int(prob["nC"]), int(prob["nP"]),
                                int(prob["nO"]), _d(K), _d(rot), _d(c), _d(p), _d(im), _i(ii), _i(jj), int(with_K))
    if rc != 0:
        raise PsbaError(rc, f"psba_write_problem({cams_file}, {pts_file}) failed")


def convert_bal(bal_file, cams_out, pts_out):
    """psba_convert_bal; returns the largest |k1|, |k2| that was dropped."""
    k = C.c_double()
    rc = lib.psba_convert_bal(bal_file.encode(), cams_out.encode(), pts_out.encode(), C.byref(k))
    if rc != 0:
        raise PsbaError(rc, f"psba_convert_bal({bal_file}) failed")
    return k.value


def partition_points(n_pts, iidx, nranks):
    iidx = _c(iidx, np.int32)
    out = np.zeros(nranks + 1, dtype=np.int32)
    rc = lib.psba_partition_points(int(n_pts), _i(iidx), int(iidx.size), int(nranks), _i(out))
    if rc != 0:
        raise PsbaError(rc, "psba_partition_points failed")
    return out


def schur_plan(n_cams, n_pts, iidx, jidx):
    """The static schedule of the S-assembly kernel for a sparsity pattern (host only, no
    device): dict with groups, items (uint64), wg [nWg,8], blockpos, glo and the counters."""
    iidx = _c(iidx, np.int32)
    jidx = _c(jidx, np.int32)
    p = lib.psba_schur_plan_create(int(n_cams), int(n_pts), int(iidx.size), _i(iidx), _i(jidx))
    if not p:
        raise PsbaError(-1, "psba_schur_plan_create failed")
    try:
        info = (C.c_longlong * 8)()
        lib.psba_schur_plan_info(p, info)
        groups, nwg, nslots, products, slab, nblocks, run_tasks, pair_items = (int(x) for x in info)
        items = np.zeros(nslots, dtype=np.uint64)
        wg = np.zeros((nwg, 8), dtype=np.int64)
        blockpos = np.zeros(nblocks, dtype=np.int32)
        glo = np.zeros(groups + 1, dtype=np.int32)
        lib.psba_schur_plan_copy(p, items.ctypes.data_as(C.POINTER(C.c_ulonglong)),
                                 wg.ctypes.data_as(C.POINTER(C.c_longlong)), _i(blockpos), _i(glo))
    finally:
        lib.psba_schur_plan_destroy(p)
    return dict(groups=groups, items=items, wg=wg, blockpos=blockpos, glo=glo, products=products,
                slab_doubles=slab, run_tasks=run_tasks, pair_items=pair_items)


def owner_plan(n_cams, n_pts, iidx, jidx, pattern=None):
    """The product lists of the owner route for a sparsity pattern (host only, no device): dict of the
    arrays psba_owner_plan_copy documents."""
    iidx = _c(iidx, np.int32)
    jidx = _c(jidx, np.int32)
    pat = None if pattern is None else np.ascontiguousarray(pattern, dtype=np.uint8)
    pp = None if pat is None else pat.ctypes.data_as(C.POINTER(C.c_ubyte))
    p = lib.psba_owner_plan_create(int(n_cams), int(n_pts), int(iidx.size), _i(iidx), _i(jidx), pp)
    if not p:
        raise PsbaError(-1, "psba_owner_plan_create failed")
    try:
        info = (C.c_longlong * 4)()
        lib.psba_owner_plan_info(p, info)
        nw, rows, products, nblk = (int(x) for x in info)
        waves = np.zeros((nw, 2), dtype=np.int64)
        units = np.zeros((64 * nw, 4), dtype=np.int32)
        prod = np.zeros((64 * rows, 2), dtype=np.int32)
        blocks = np.zeros((nblk, 2), dtype=np.int32)
        diag = np.zeros(int(n_cams), dtype=np.int32)
        lib.psba_owner_plan_copy(p, waves.ctypes.data_as(C.POINTER(C.c_longlong)), _i(units), _i(prod), _i(blocks), _i(diag))
    finally:
        lib.psba_owner_plan_destroy(p)
    return dict(waves=waves, units=units, prod=prod.reshape(rows, 64, 2), blocks=blocks, diag_slot=diag, products=products)


def chol_dist_exchange_plan(n32, NB, nranks, JE):
    """[(block, owner, slot, doubles)] of the column exchange in front of the super-panel at column JE."""
    out = np.zeros((64, 4), dtype=np.int64)
    n = lib.psba_chol_dist_exchange_plan(int(n32), int(NB), int(nranks), int(JE), out.ctypes.data_as(C.POINTER(C.c_longlong)), 64)
    if n < 0:
        raise PsbaError(n, "psba_chol_dist_exchange_plan failed")
    return [tuple(int(v) for v in row) for row in out[:n]]


def ring_plan(n_cams, n_pts, iidx, jidx):
    """The schedule of the S-assembly kernel's ring route for a sparsity pattern (host only, no
    device): dict of the arrays psba_ring_plan_copy documents, or None when the route does not
    apply to the problem.  Needs a library built with PSBA_BUILD_EXPERIMENTS=1."""
    if not HAS_EXPERIMENTS:
        raise PsbaError(-6, "the ring route is an experiment: build with PSBA_BUILD_EXPERIMENTS=1")
    iidx = _c(iidx, np.int32)
    jidx = _c(jidx, np.int32)
    p = lib.psba_ring_plan_create(int(n_cams), int(n_pts), int(iidx.size), _i(iidx), _i(jidx))
    if not p:
        raise PsbaError(-1, "psba_ring_plan_create failed")
    try:
        info = (C.c_longlong * 16)()
        lib.psba_ring_plan_info(p, info)
        (nR, nS, nwg, nsteps, nent, nops, njob, products, slots, lanes, pages, yslots, lat, nlb, nbl0, loaded) = (int(x) for x in info)
        if nwg == 0:
            return None
        wg = np.zeros((nwg, 13), dtype=np.int64)
        steps = np.zeros((nsteps, 4), dtype=np.int32)
        entries = np.zeros(nent, dtype=np.uint32)
        ops = np.zeros((nops, 2), dtype=np.int32)
        jobs = np.zeros((njob, 4), dtype=np.int32)
        lane_blk = np.zeros(nlb, dtype=np.int32)
        blk_lane0 = np.zeros(nbl0, dtype=np.int32)
        rb = np.zeros(nR + 1, dtype=np.int32)
        lib.psba_ring_plan_copy(p, wg.ctypes.data_as(C.POINTER(C.c_longlong)), _i(steps),
                                entries.ctypes.data_as(C.POINTER(C.c_uint)), _i(ops), _i(jobs), _i(lane_blk),
                                _i(blk_lane0), _i(rb))
    finally:
        lib.psba_ring_plan_destroy(p)
    return dict(nR=nR, nS=nS, wg=wg, steps=steps, entries=entries, ops=ops, jobs=jobs, lane_blk=lane_blk,
                blk_lane0=blk_lane0, rb=rb, products=products, lane_steps=slots, lanes=lanes, slots=pages,
                page=yslots, lat=lat, loaded_recs=loaded)


def shard_problem(prob, nranks, rank):
    """The sub-problem rank `rank` owns: a contiguous point range and its observations;
    cameras are replicated."""
    bounds = partition_points(prob["nP"], prob["iidx"], nranks)
    p0, p1 = int(bounds[rank]), int(bounds[rank + 1])
    iidx = np.asarray(prob["iidx"])
    sel = (iidx >= p0) & (iidx < p1)
    return Problem(K=prob["K"], initrot=prob["initrot"], cams=prob["cams"], pts=prob["pts"][p0:p1],
                   impts=np.asarray(prob["impts"])[sel], iidx=(iidx[sel] - p0).astype(np.int32),
                   jidx=np.asarray(prob["jidx"])[sel].astype(np.int32), nC=prob["nC"], nP=p1 - p0,
                   nO=int(sel.sum()))


class Psba:
    """One handle = one GPU.  Methods are the C entry points without the psba_ prefix."""

    def __init__(self, device=0):
        self._h = _h()
        rc = lib.psba_create(int(device), C.byref(self._h))
        if rc != 0:
            raise PsbaError(rc, lib.psba_last_error(None).decode())
        self.nC = self.nP = self.nO = 0

    def close(self):
        if self._h:
            lib.psba_destroy(self._h)
            self._h = _h()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise PsbaError(rc, lib.psba_last_error(self._h).decode())
        return rc

    # ---- setup ----
    def set_camera_model(self, free_k):
        """PSBA_CAMERA_FREE_K: camera blocks of 11 (fu, u0, v0, ar, s | rotation | translation); before upload."""
        self._ck(lib.psba_set_camera_model(self._h, 1 if free_k else 0))

    def camera_block(self):
        n = C.c_int()
        self._ck(lib.psba_camera_block(self._h, C.byref(n)))
        return n.value

    def upload_problem(self, prob):
        self.nC, self.nP, self.nO = int(prob["nC"]), int(prob["nP"]), int(prob["nO"])
        self.cnp = self.camera_block()
        self.nA, self.nB = self.cnp * self.nC, 3 * self.nP
        self.nT = self.nA + self.nB
        a = [_c(prob[k]).reshape(-1) for k in ("K", "impts", "initrot", "cams", "pts")]
        ii, jj = _c(prob["iidx"], np.int32), _c(prob["jidx"], np.int32)
        self._ck(lib.psba_upload_problem(self._h, self.nC, self.nP, self.nO, _d(a[0]), _d(a[1]),
                                         _d(a[2]), _d(a[3]), _d(a[4]), _i(ii), _i(jj)))

    def set_params(self, cams, pts):
        c, p = _c(cams).reshape(-1), _c(pts).reshape(-1)
        self._ck(lib.psba_set_params(self._h, _d(c), _d(p)))

    def reset_params(self):
        self._ck(lib.psba_reset_params(self._h))

    def get_params(self, which=PARAMS_CUR):
        c, p = np.empty(self.nA), np.empty(self.nB)
        self._ck(lib.psba_get_params(self._h, which, _d(c), _d(p)))
        return c.reshape(self.nC, self.cnp), p.reshape(self.nP, 3)

    # ---- fused verbs ----
    def schur_path(self):
        """0: LDS-partition schedule, 1: owner route, 2: global-atomic assembly kernel, 3: ring route
        (psba_schur_path)."""
        v = C.c_int()
        self._ck(lib.psba_schur_path(self._h, C.byref(v)))
        return v.value

    def residual(self, which=PARAMS_CUR):
        v = C.c_double()
        self._ck(lib.psba_residual(self._h, which, C.byref(v)))
        return v.value

    def linearize(self, coeff=1.0, coeff_g=1.0):
        self._ck(lib.psba_linearize(self._h, coeff, coeff_g))

    def begin(self, coeff=1.0, coeff_g=1.0):
        """(cost, max diagonal) at the current parameters + their linearization, one sync."""
        c, m = C.c_double(), C.c_double()
        self._ck(lib.psba_begin(self._h, coeff, coeff_g, C.byref(c), C.byref(m)))
        return c.value, m.value

    def max_diag(self):
        v = C.c_double()
        self._ck(lib.psba_max_diag(self._h, C.byref(v)))
        return v.value

    def schur_assemble(self, mu):
        self._ck(lib.psba_schur_assemble(self._h, mu))

    def schur_reduce(self):
        self._ck(lib.psba_schur_reduce(self._h))

    def schur_solve(self):
        """PSBA_OK, or PSBA_PCG_MAXIT (4) when the iterative solve used up max_iter (the iterate is kept)."""
        return self._ck(lib.psba_schur_solve(self._h))

    def backsub(self, mu):
        s = TryScalars()
        self._ck(lib.psba_backsub(self._h, mu, C.byref(s)))
        return s

    def backsub_async(self, mu):
        self._ck(lib.psba_backsub_async(self._h, mu))

    def linearize_ahead(self):
        self._ck(lib.psba_linearize_ahead(self._h))

    def backsub_wait(self):
        s = TryScalars()
        self._ck(lib.psba_backsub_wait(self._h, C.byref(s)))
        return s

    def accept(self):
        self._ck(lib.psba_accept(self._h))

    # ---- sba_func.h mirror ----
    def _out(self, fn, n, *pre):
        out = np.empty(n)
        rc = self._ck(fn(self._h, *pre, _d(out)))
        return rc, out

    def compute_exQT(self, which=PARAMS_CUR):
        return self._out(lib.psba_compute_exQT, 2 * self.nO, which)[1]

    def compute_jacobiQT(self):
        JA, JB = np.empty(12 * self.nO), np.empty(6 * self.nO)
        self._ck(lib.psba_compute_jacobiQT(self._h, _d(JA), _d(JB)))
        return JA, JB

    def compute_U(self, coeff=1.0):
        return self._out(lib.psba_compute_U, 36 * self.nC, coeff)[1]

    def compute_V(self, coeff=1.0):
        return self._out(lib.psba_compute_V, 9 * self.nP, coeff)[1]

    def maxElmOfUV(self):
        v = C.c_double()
        self._ck(lib.psba_maxElmOfUV(self._h, C.byref(v)))
        return v.value

    def update_UV(self, mu):
        U, V = np.empty(36 * self.nC), np.empty(9 * self.nP)
        self._ck(lib.psba_update_UV(self._h, mu, _d(U), _d(V)))
        return U, V

    def restore_UVdiag(self):
        self._ck(lib.psba_restore_UVdiag(self._h))

    def compute_Vinv(self):
        return self._out(lib.psba_compute_Vinv, 9 * self.nP)

    def compute_Wblks(self, coeff=1.0):
        return self._out(lib.psba_compute_Wblks, 18 * self.nO, coeff)[1]

    def compute_Yblks(self):
        return self._out(lib.psba_compute_Yblks, 18 * self.nO)[1]

    def compute_S(self):
        return self._out(lib.psba_compute_S, self.nA * self.nA)[1].reshape(self.nA, self.nA)

    def compute_g(self, coeff=1.0):
        return self._out(lib.psba_compute_g, self.nT, coeff)[1]

    def compute_ea(self):
        return self._out(lib.psba_compute_ea, self.nA)[1]

    def SPDinv_matVec(self):
        return self._out(lib.psba_SPDinv_matVec, self.nA)

    def compute_eb(self):
        return self._out(lib.psba_compute_eb, self.nB)[1]

    def compute_dpb(self):
        return self._out(lib.psba_compute_dpb, self.nT)[1]

    def compute_newp(self):
        return self._out(lib.psba_compute_newp, self.nT)[1]

    def update_p(self):
        return self._out(lib.psba_update_p, self.nT)[1]

    # ---- LM ----
    def levmar(self, max_iter=50, tr_handoff=False, verbose=False, log_cap=512, start_itno=0, init_mu=0.0):
        opts = LmOptions(max_iter, int(tr_handoff), int(verbose), log_cap, start_itno, init_mu)
        res = LmResult()
        log = np.zeros((max(log_cap, 1), 5))
        self._ck(lib.psba_levmar(self._h, C.byref(opts), C.byref(res), _d(log)))
        return res, log[: res.n_log].copy()

    # ---- block-sparse S + PCG ----
    def set_solver(self, solver, tol=0.0, max_iter=0):
        """solver: 0 dense Cholesky, 1 block-sparse S + preconditioned CG (before upload_problem)."""
        self._ck(lib.psba_set_solver(self._h, int(solver), float(tol), int(max_iter)))

    def pcg_info(self):
        it, rr, nb, nd = C.c_int(), C.c_double(), C.c_longlong(), C.c_longlong()
        self._ck(lib.psba_pcg_info(self._h, C.byref(it), C.byref(rr), C.byref(nb), C.byref(nd)))
        return it.value, rr.value, nb.value, nd.value

    def get_sparse_S(self):
        nb = self.pcg_info()[2]
        jk = np.empty((nb, 2), dtype=np.int32)
        val = np.empty((nb, 6, 6))
        ea = np.empty(6 * self.nC)
        self._ck(lib.psba_get_sparse_S(self._h, _i(jk), _d(val), _d(ea)))
        return jk, val, ea

    def set_sparse_S(self, val, ea):
        val = np.ascontiguousarray(val, dtype=np.float64)
        ea = np.ascontiguousarray(ea, dtype=np.float64)
        self._ck(lib.psba_set_sparse_S(self._h, _d(val), _d(ea)))

    def set_sparse_pattern(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self._ck(lib.psba_set_sparse_pattern(self._h, flags.ctypes.data_as(C.POINTER(C.c_ubyte)), flags.size))

    # ---- the sharded dense factorization, piece by piece ----
    def chol_dist_shape(self):
        n32, nb, sh = C.c_int(), C.c_int(), C.c_int()
        self._ck(lib.psba_chol_dist_shape(self._h, C.byref(n32), C.byref(nb), C.byref(sh)))
        return n32.value, nb.value, bool(sh.value)

    def chol_dist_begin(self):
        self._ck(lib.psba_chol_dist_begin(self._h))

    def chol_dist_superpanel(self, J):
        self._ck(lib.psba_chol_dist_superpanel(self._h, int(J)))

    def chol_dist_get_block(self, B):
        n = C.c_longlong()
        self._ck(lib.psba_chol_dist_block(self._h, int(B), 0, None, C.byref(n)))
        buf = np.empty(n.value)
        self._ck(lib.psba_chol_dist_block(self._h, int(B), 0, _d(buf), C.byref(n)))
        return buf

    def chol_dist_set_block(self, B, buf):
        n = C.c_longlong()
        self._ck(lib.psba_chol_dist_block(self._h, int(B), 1, _d(_c(buf)), C.byref(n)))

    def chol_dist_finish(self):
        self._ck(lib.psba_chol_dist_finish(self._h))

    # ---- trust region ----
    def compute_Jmultiply(self, x):
        return self._out(lambda h, out: lib.psba_compute_Jmultiply(h, _d(_c(x)), out), 2 * self.nO)[1]

    def jmul_dots(self, x1, x2=None):
        out = np.empty(3)
        a1 = _c(x1)
        a2 = None if x2 is None else _c(x2)
        self._ck(lib.psba_jmul_dots(self._h, _d(a1), _d(a2), _d(out)))
        return out

    def get_gradient(self):
        return self._out(lib.psba_get_gradient, self.nT)[1]

    def get_dp(self):
        return self._out(lib.psba_get_dp, self.nT)[1]

    def set_step(self, dp):
        self._ck(lib.psba_set_step(self._h, _d(_c(dp))))

    def cholmod_lambda(self, reassemble=True):
        lam, info = C.c_double(), np.empty(3)
        self._ck(lib.psba_cholmod_lambda(self._h, int(reassemble), C.byref(lam), _d(info)))
        return lam.value, info

    def trust_region(self, max_iter=50, start_itno=0, verbose=False, log_cap=512, init_lambda=0.0):
        opts = TrOptions(max_iter, start_itno, int(verbose), log_cap, init_lambda)
        res = TrResult()
        log = np.zeros((max(log_cap, 1), 6))
        self._ck(lib.psba_trust_region(self._h, C.byref(opts), C.byref(res), _d(log)))
        return res, log[: res.n_log].copy()

    def solve(self, max_iter=50, verbose=False):
        res = SolveResult()
        self._ck(lib.psba_solve(self._h, max_iter, int(verbose), C.byref(res)))
        return res

    # ---- multi-GPU ----
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        rc = lib.psba_comm_unique_id(buf)
        if rc != 0:
            raise PsbaError(rc, "psba_comm_unique_id failed")
        return buf.raw

    def comm_init(self, nranks, rank, uid):
        buf = C.create_string_buffer(uid, 128)
        self._ck(lib.psba_comm_init(self._h, nranks, rank, buf))

    def set_rank_layout(self, nranks, rank):
        self._ck(lib.psba_set_rank_layout(self._h, nranks, rank))

    def reduce_buffer_size(self):
        n = C.c_longlong()
        self._ck(lib.psba_reduce_buffer_size(self._h, C.byref(n)))
        return n.value

    def get_reduce_buffer(self):
        out = np.empty(self.reduce_buffer_size())
        self._ck(lib.psba_get_reduce_buffer(self._h, _d(out)))
        return out

    def set_reduce_buffer(self, buf):
        b = _c(buf).reshape(-1)
        assert b.size == self.reduce_buffer_size()
        self._ck(lib.psba_set_reduce_buffer(self._h, _d(b)))

    # ---- measurement ----
    def profile_enable(self, on=True):
        """True: every kernel class; False: off; int > 0: bit mask of 1 << K_*."""
        self._ck(lib.psba_profile_enable(self._h, -1 if on is True else int(on)))

    def profile_reset(self):
        self._ck(lib.psba_profile_reset(self._h))

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_int()
        self._ck(lib.psba_profile_get(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def algorithmic_bytes(self, kernel):
        b = C.c_double()
        self._ck(lib.psba_algorithmic_bytes(self._h, kernel, C.byref(b)))
        return b.value
