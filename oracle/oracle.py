"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the calibration path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.  Parity unpinned
(see calib_oracle.h for what that means and why).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# flag values (OpenCV 4.1.0 calib3d.hpp)
USE_INTRINSIC_GUESS = 0x00001
FIX_ASPECT_RATIO = 0x00002
FIX_PRINCIPAL_POINT = 0x00004
ZERO_TANGENT_DIST = 0x00008
FIX_FOCAL_LENGTH = 0x00010
FIX_K1, FIX_K2, FIX_K3 = 0x00020, 0x00040, 0x00080
FIX_INTRINSIC = 0x00100
SAME_FOCAL_LENGTH = 0x00200
FIX_K4, FIX_K5, FIX_K6 = 0x00800, 0x01000, 0x02000
RATIONAL_MODEL = 0x04000
USE_LU = 1 << 17
TERM_COUNT, TERM_EPS = 1, 2
SOLVE_LU, SOLVE_SVD, SOLVE_CHOLESKY = 0, 1, 2


class Stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("jac_passes", C.c_int), ("err_passes", C.c_int),
                ("lambda_lg10", C.c_int), ("final_err", C.c_double)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_calibrate_camera.restype = C.c_double
        _LIB.orc_stereo_calibrate.restype = C.c_double
        _LIB.orc_reprojection_error.restype = C.c_double
        _LIB.orc_mono_blocks.restype = C.c_double
        _LIB.orc_trace_count.restype = C.c_int
        _LIB.orc_stereo_blocks.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


TRACE_FIELDS = ("iter", "lambda_lg10_first", "lambda_lg10_accepted", "trials", "err_prev", "err")


class _Trace:
    """Per-iteration record of the LM run inside one calibrate call (orc_set_trace): rows of TRACE_FIELDS."""

    def __init__(self, on, cap=1024):
        self.buf = np.zeros((cap, len(TRACE_FIELDS))) if on else None

    def __enter__(self):
        if self.buf is not None:
            lib().orc_set_trace(_p(self.buf), C.c_int(self.buf.shape[0]))
        return self

    def __exit__(self, *exc):
        if self.buf is not None:
            self.rows = self.buf[:lib().orc_trace_count()].copy()
            lib().orc_set_trace(None, C.c_int(0))
        return False


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


VARIANT_PP_HALF_SIZE, VARIANT_SAME_FOCAL_MASK = 1, 2


def set_variants(bits: int):
    """What-if switches for the two recalled details of OpenCV 4.1.0 (see calib_oracle.h); 0 = the restatement."""
    lib().orc_set_variants(C.c_int(bits))


def rodrigues_v2m(r):
    R, J = np.zeros(9), np.zeros(27)
    lib().orc_rodrigues_v2m(_p(_d(r)), _p(R), _p(J))
    return R.reshape(3, 3), J.reshape(3, 9)


def rodrigues_m2v(Rm):
    r, J = np.zeros(3), np.zeros(27)
    lib().orc_rodrigues_m2v(_p(_d(Rm).reshape(-1)), _p(r), _p(J))
    return r, J.reshape(3, 9)


def compose_rt(r1, t1, r2, t2):
    out = {k: np.zeros(9) for k in ("dr3dr1", "dr3dr2", "dt3dr2", "dt3dt1", "dt3dt2")}
    r3, t3 = np.zeros(3), np.zeros(3)
    lib().orc_compose_rt(_p(_d(r1)), _p(_d(t1)), _p(_d(r2)), _p(_d(t2)), _p(r3), _p(t3), _p(out["dr3dr1"]),
                         _p(out["dr3dr2"]), _p(out["dt3dr2"]), _p(out["dt3dt1"]), _p(out["dt3dt2"]))
    return r3, t3, {k: v.reshape(3, 3) for k, v in out.items()}


def project_points(obj, r, t, fc, k14, aspect=0.0, jac=True):
    obj = _d(obj).reshape(-1, 3)
    n = obj.shape[0]
    m = np.zeros((n, 2))
    Je = np.zeros((2 * n, 6)) if jac else None
    Ji = np.zeros((2 * n, 18)) if jac else None
    lib().orc_project_points(C.c_int(n), _p(obj), _p(_d(r)), _p(_d(t)), _p(_d(fc)), _p(_d(k14)),
                             C.c_double(aspect), _p(m), _p(Je), _p(Ji))
    return m, Je, Ji


def find_homography(src, dst):
    src, dst = _d(src).reshape(-1, 2), _d(dst).reshape(-1, 2)
    H = np.zeros(9)
    ok = lib().orc_find_homography(C.c_int(src.shape[0]), _p(src), _p(dst), _p(H))
    return H.reshape(3, 3) if ok else None


def _flatten(obj_views, img_views):
    """Accepts (P,3) shared board + (N,P,2), or lists of per-view arrays."""
    if isinstance(img_views, np.ndarray) and img_views.ndim == 3:
        N, P = img_views.shape[:2]
        obj_views = np.asarray(obj_views)
        if obj_views.ndim == 2:
            obj = np.broadcast_to(obj_views[None], (N, P, 3))
        else:
            obj = obj_views
        npts = np.full(N, P, dtype=np.int32)
        return N, npts, _d(obj).reshape(-1, 3), _d(img_views).reshape(-1, 2)
    npts = np.array([len(v) for v in img_views], dtype=np.int32)
    obj = _d(np.concatenate([np.asarray(o, dtype=np.float64).reshape(-1, 3) for o in obj_views]))
    img = _d(np.concatenate([np.asarray(o, dtype=np.float64).reshape(-1, 2) for o in img_views]))
    return len(npts), npts, obj, img


def init_camera_matrix_2d(obj_views, img_views, size, aspect=0.0):
    N, npts, obj, img = _flatten(obj_views, img_views)
    A = np.zeros(9)
    lib().orc_init_intrinsic_params_2d(C.c_int(N), _p(npts), _p(obj), _p(img), C.c_int(size[0]), C.c_int(size[1]),
                                       C.c_double(aspect), _p(A))
    return A.reshape(3, 3)


def find_extrinsic(obj, img, A, k14):
    obj, img = _d(obj).reshape(-1, 3), _d(img).reshape(-1, 2)
    r, t = np.zeros(3), np.zeros(3)
    ok = lib().orc_find_extrinsic(C.c_int(obj.shape[0]), _p(obj), _p(img), _p(_d(A).reshape(-1)), _p(_d(k14)), _p(r), _p(t))
    return (r, t) if ok else None


def calibrate_camera(obj_views, img_views, size, K=None, dist=None, ndist=5, flags=0,
                     criteria=(TERM_COUNT | TERM_EPS, 30, np.finfo(np.float64).eps), init_rt=None,
                     schur=True, solve=-1, nthreads=1, trace=False):
    """Returns dict(rms, K, dist, rvecs, tvecs, iters, jac_passes, err_passes, lambda_lg10[, trace])."""
    N, npts, obj, img = _flatten(obj_views, img_views)
    A = np.eye(3).reshape(-1) if K is None else _d(K).reshape(-1).copy()
    d = np.zeros(ndist)
    if dist is not None:
        dd = _d(dist).reshape(-1)
        d[:min(ndist, dd.size)] = dd[:ndist]
    rv, tv = np.zeros((N, 3)), np.zeros((N, 3))
    st = Stats()
    irt = None if init_rt is None else _d(init_rt).reshape(N, 6)
    with _Trace(trace) as tr:
        rms = lib().orc_calibrate_camera(C.c_int(N), _p(npts), _p(obj), _p(img), C.c_int(size[0]), C.c_int(size[1]),
                                         _p(A), _p(d), C.c_int(ndist), _p(rv), _p(tv), C.c_int(flags),
                                         C.c_int(criteria[0]), C.c_int(criteria[1]), C.c_double(criteria[2]),
                                         _p(irt), C.c_int(1 if schur else 0), C.c_int(solve), C.c_int(nthreads),
                                         C.byref(st))
    out = dict(rms=rms, K=A.reshape(3, 3), dist=d, rvecs=rv, tvecs=tv, iters=st.iters,
               jac_passes=st.jac_passes, err_passes=st.err_passes, lambda_lg10=st.lambda_lg10,
               final_err=st.final_err)
    if trace:
        out["trace"] = tr.rows
    return out


def stereo_calibrate(obj_views, img1, img2, K1, d1, K2, d2, size, flags,
                     criteria=(TERM_COUNT | TERM_EPS, 30, 1e-6), init_rt=None, init_lr=None, schur=True,
                     nthreads=1, trace=False):
    N, npts, obj, i1 = _flatten(obj_views, img1)
    _, _, _, i2 = _flatten(obj_views, img2)
    A1, A2 = _d(K1).reshape(-1).copy(), _d(K2).reshape(-1).copy()
    D1, D2 = np.zeros(14), np.zeros(14)
    if d1 is not None:
        D1[:np.size(d1)] = np.ravel(d1)
    if d2 is not None:
        D2[:np.size(d2)] = np.ravel(d2)
    R, T, E, F = np.zeros(9), np.zeros(3), np.zeros(9), np.zeros(9)
    rt = np.zeros((N, 6))
    st = Stats()
    irt = None if init_rt is None else _d(init_rt).reshape(N, 6)
    ilr = None if init_lr is None else _d(init_lr).reshape(6)
    with _Trace(trace) as tr:
        rms = lib().orc_stereo_calibrate(C.c_int(N), _p(npts), _p(obj), _p(i1), _p(i2), _p(A1), _p(D1), _p(A2), _p(D2),
                                         C.c_int(size[0]), C.c_int(size[1]), _p(R), _p(T), _p(E), _p(F), C.c_int(flags),
                                         C.c_int(criteria[0]), C.c_int(criteria[1]), C.c_double(criteria[2]),
                                         _p(irt), _p(ilr), _p(rt), C.c_int(1 if schur else 0), C.c_int(nthreads),
                                         C.byref(st))
    out = dict(rms=rms, K1=A1.reshape(3, 3), d1=D1, K2=A2.reshape(3, 3), d2=D2, R=R.reshape(3, 3), T=T,
               E=E.reshape(3, 3), F=F.reshape(3, 3), rt=rt, iters=st.iters, jac_passes=st.jac_passes,
               err_passes=st.err_passes, lambda_lg10=st.lambda_lg10, final_err=st.final_err)
    if trace:
        out["trace"] = tr.rows
    return out


def reprojection_error(obj_views, img_views, K, dist, rvecs, tvecs):
    N, npts, obj, img = _flatten(obj_views, img_views)
    d = _d(dist).reshape(-1)
    pv = np.zeros(N)
    rms = lib().orc_reprojection_error(C.c_int(N), _p(npts), _p(obj), _p(img), _p(_d(K).reshape(-1)), _p(d),
                                       C.c_int(d.size), _p(_d(rvecs)), _p(_d(tvecs)), _p(pv))
    return rms, pv


def mono_blocks(obj_views, img_views, intr18, aspect, rt, free_idx):
    """Per-view normal-equation pieces at fixed parameters (see calib_oracle.h)."""
    N, npts, obj, img = _flatten(obj_views, img_views)
    fi = np.ascontiguousarray(free_idx, dtype=np.int32)
    c = fi.size
    U, gc = np.zeros((c, c)), np.zeros(c)
    V, W, ge = np.zeros((N, 6, 6)), np.zeros((N, c, 6)), np.zeros((N, 6))
    err = lib().orc_mono_blocks(C.c_int(N), _p(npts), _p(obj), _p(img), _p(_d(intr18)), C.c_double(aspect),
                                _p(_d(rt).reshape(N, 6)), C.c_int(c), _p(fi), _p(U), _p(gc), _p(V), _p(W), _p(ge))
    return dict(err=err, U=U, gc=gc, V=V, W=W, ge=ge)


def stereo_blocks(obj_views, img1, img2, K1, d1, K2, d2, flags, rt, lr):
    """The stereo normal equations of one Jacobian pass at fixed parameters (see calib_oracle.h): dict(err, U 42x42
    (upper block triangle), gs 42, V Nx6x6, W Nx42x6, ge Nx6); shared order [om_LR T_LR | intr cam 0 | intr cam 1]."""
    N, npts, obj, i1 = _flatten(obj_views, img1)
    _, _, _, i2 = _flatten(obj_views, img2)
    D1, D2 = np.zeros(14), np.zeros(14)
    D1[:np.size(d1)] = np.ravel(d1)
    D2[:np.size(d2)] = np.ravel(d2)
    U, gs = np.zeros((42, 42)), np.zeros(42)
    V, W, ge = np.zeros((N, 6, 6)), np.zeros((N, 42, 6)), np.zeros((N, 6))
    err = lib().orc_stereo_blocks(C.c_int(N), _p(npts), _p(obj), _p(i1), _p(i2), _p(_d(K1).reshape(-1)), _p(D1),
                                  _p(_d(K2).reshape(-1)), _p(D2), C.c_int(flags), _p(_d(rt).reshape(N, 6)),
                                  _p(_d(lr).reshape(6)), _p(U), _p(gs), _p(V), _p(W), _p(ge))
    return dict(err=err, U=U, gs=gs, V=V, W=W, ge=ge)
