"""psba_amd -- MI355X-native Schur-complement bundle-adjustment normal-equations path.

The product is the HIP C-ABI library ``psba_amd/libpsba_hip.so`` (include/psba_hip.h).  This
package is only a ctypes binding to it for tests, bench.py and Python callers.  There is no CPU
fallback: importing :mod:`psba_amd.capi` raises if the library has not been built, and creating
a handle raises if no GPU is present.
"""
from .capi import (Psba, PsbaError, Problem, lib, lib_path, read_problem, partition_points,  # noqa: F401
                   write_problem, convert_bal)

__all__ = ["Psba", "PsbaError", "Problem", "lib", "lib_path", "read_problem", "partition_points",
           "write_problem", "convert_bal"]
