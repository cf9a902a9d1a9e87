"""ctypes binding of libccalib.so (the C ABI declared in include/ccalib.h).

There is no fallback: if the shared library is missing, or lacks a declared
symbol, load() raises.  Every compute call needs a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCALIB_LIB") or os.path.join(_HERE, "libccalib.so")   # CCALIB_LIB: a development build

# every symbol include/ccalib.h declares
SYMBOLS = [
    "ccalib_last_error", "ccalib_version", "ccalib_dist_coeffs_out",
    "ccalib_calibrate_camera", "ccalib_stereo_calibrate", "ccalib_calibrate_camera_views", "ccalib_stereo_calibrate_views",
    "ccalib_reprojection_error", "ccalib_release_cached_memory",
    "ccalib_reprojection_error_noalloc",
    "ccalib_init_camera_matrix_2d", "ccalib_write_mono_yaml", "ccalib_write_stereo_yaml",
    "ccalib_solver_create", "ccalib_solver_destroy", "ccalib_solver_set_start",
    "ccalib_rccl_unique_id", "ccalib_solver_use_rccl", "ccalib_solver_use_callback",
    "ccalib_solver_run", "ccalib_solver_get", "ccalib_solver_counters", "ccalib_solver_num_shared",
    "ccalib_solver_eval_error", "ccalib_solver_eval_jacobian", "ccalib_solver_fetch_blocks",
    "ccalib_solver_reduce", "ccalib_solver_time_kernels", "ccalib_lm_run",
    "ccalib_solver_init_intrinsics", "ccalib_solver_init_poses", "ccalib_solver_init_relative_pose",
    "ccalib_last_lm_trace", "ccalib_solver_describe", "ccalib_solver_profile_jac",
    "ccalib_solver_executed_passes",
]

OK, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_STATE, ERR_NUMERIC = range(7)


class Stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("jac_passes", C.c_int), ("err_passes", C.c_int), ("lambda_lg10", C.c_int),
                ("final_err", C.c_double), ("device_ms", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
JAC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double),
                     C.POINTER(C.c_double), C.POINTER(C.c_double))
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double),
                        C.POINTER(C.c_double))
STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
COMMIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
FIXUP_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))


class Backend(C.Structure):
    _fields_ = [("ns", C.c_int), ("jac", JAC_FN), ("reduce", REDUCE_FN), ("step_err", STEP_FN),
                ("commit", COMMIT_FN), ("fixup", FIXUP_FN)]


class CcalibError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ccalib error {code}: {msg}")
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                           "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    # an older build named by CCALIB_OLD_LIB_OK=1 + CCALIB_LIB (same-box A/B timing against an earlier commit) may lack
    # the newer entry points; the in-tree library must have them all
    older_ok = os.environ.get("CCALIB_OLD_LIB_OK") == "1" and os.environ.get("CCALIB_LIB")
    for name in SYMBOLS:
        if not hasattr(lib, name) and not older_ok:
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
    lib.ccalib_last_error.restype = C.c_char_p
    lib.ccalib_version.restype = C.c_char_p
    lib.ccalib_solver_destroy.restype = None
    if hasattr(lib, "ccalib_release_cached_memory"):
        lib.ccalib_release_cached_memory.restype = C.c_size_t
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise CcalibError(rc, load().ccalib_last_error().decode())
