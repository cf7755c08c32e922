"""Builds the in-tree HIP C-ABI library (psba_amd/libpsba_hip.so) for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["psba_api.cpp", "schur_plan.cpp", "lm_loop.cpp", "tr_loop.cpp", "sba_io.cpp", "kernels_linearize.hip",
           "kernels_tr.hip",
           "kernels_schur.hip", "kernels_freek.hip", "kernels_chol.hip", "kernels_pcg.hip", "kernels_chol_graph.hip", "kernels_backsub.hip"]
# PSBA_BUILD_EXPERIMENTS=1: also the rejected experiments (round 3's K2 ring route) and their test hooks
EXPERIMENTS = bool(os.environ.get("PSBA_BUILD_EXPERIMENTS"))
if EXPERIMENTS:
    SOURCES += ["kernels_schur_ring.hip", "schur_ring_plan.cpp", "kernels_schur_modes.hip"]
OUT = os.path.join(HERE, "libpsba_hip_exp.so" if EXPERIMENTS else "libpsba_hip.so")
HEADERS = ["psba_internal.h", "camera_model.h", "chol_factor32.h", "schur_common.h", "schur_lds_args.h", os.path.join("..", "..", "include", "psba_hip.h")]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-result", "-x", "hip"] + (["-DPSBA_BUILD_EXPERIMENTS"] if EXPERIMENTS else [])


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    objdir = os.path.join(HERE, "build_exp" if EXPERIMENTS else "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, rebuilt = [], False
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
    if rebuilt or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + [
            "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
