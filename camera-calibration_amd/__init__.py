"""MI355X-native calibration LM core: host mirror of the reference's call surface.

The compute lives in csrc/ (hand-written HIP for gfx950 behind the C ABI of
include/ccalib.h).  This package is the thin Python host side used by tests and
bench.py; the function names and argument meaning follow the OpenCV calls the
reference makes:

    calibrate_camera   cv::calibrateCamera / calibrateCameraRO(iFixedPoint=-1)
                       (/root/reference/monocular_circle.cpp:93-94,
                        /root/reference/monocular_chessboard.cpp:315-317)
    stereo_calibrate   cv::stereoCalibrate
                       (/root/reference/binocular_chessboard.cpp:152-162)
    reprojection_error cv::projectPoints + RMS (/root/reference/monocular_circle.cpp:5-29)
    init_camera_matrix_2d  cv::initCameraMatrix2D (/root/reference/binocular_chessboard.cpp:147-148)

It never falls back to a CPU implementation: without libccalib.so or without a
GPU every compute call raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi, synth  # noqa: F401
from .capi import CcalibError  # noqa: F401

# flag values (OpenCV 4.1.0 calib3d.hpp)
CALIB_USE_INTRINSIC_GUESS = 0x00001
CALIB_FIX_ASPECT_RATIO = 0x00002
CALIB_FIX_PRINCIPAL_POINT = 0x00004
CALIB_ZERO_TANGENT_DIST = 0x00008
CALIB_FIX_FOCAL_LENGTH = 0x00010
CALIB_FIX_K1, CALIB_FIX_K2, CALIB_FIX_K3 = 0x00020, 0x00040, 0x00080
CALIB_FIX_INTRINSIC = 0x00100
CALIB_SAME_FOCAL_LENGTH = 0x00200
CALIB_FIX_K4, CALIB_FIX_K5, CALIB_FIX_K6 = 0x00800, 0x01000, 0x02000
CALIB_RATIONAL_MODEL = 0x04000
CALIB_THIN_PRISM_MODEL = 0x08000
CALIB_FIX_S1_S2_S3_S4 = 0x10000
CALIB_USE_LU = 1 << 17
CALIB_TILTED_MODEL = 0x40000
CALIB_FIX_TAUX_TAUY = 0x80000
CALIB_FIX_TANGENT_DIST = 0x200000
TERM_COUNT, TERM_EPS = 1, 2
DBL_EPSILON = float(np.finfo(np.float64).eps)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _flatten(obj_views, img_views):
    """(P,3) shared board + (N,P,2), or per-view lists -> (N, npts, obj float32 (T,3), img float32 (T,2))."""
    if isinstance(img_views, np.ndarray) and img_views.ndim == 3:
        N, P = img_views.shape[:2]
        obj_views = np.asarray(obj_views, dtype=np.float32)
        obj = np.broadcast_to(obj_views[None], (N, P, 3)) if obj_views.ndim == 2 else obj_views
        return (N, np.full(N, P, dtype=np.int32), np.ascontiguousarray(obj, dtype=np.float32).reshape(-1, 3),
                np.ascontiguousarray(img_views, dtype=np.float32).reshape(-1, 2))
    npts = np.array([len(v) for v in img_views], dtype=np.int32)
    obj = np.ascontiguousarray(np.concatenate([np.asarray(o, dtype=np.float32).reshape(-1, 3) for o in obj_views]))
    img = np.ascontiguousarray(np.concatenate([np.asarray(o, dtype=np.float32).reshape(-1, 2) for o in img_views]))
    if len(obj) != len(img):
        raise ValueError("object and image point counts differ")
    return len(npts), npts, obj, img


def _view_ptrs(views, N, width, npts=None):
    """One pointer per view (what the ccalib_*_views entry points take) without gathering anything: returns
    (uint64 array of N addresses, points per view, the arrays those addresses point into -- keep them alive)."""
    if isinstance(views, np.ndarray) and views.ndim == 3:
        a = np.ascontiguousarray(views, dtype=np.float32)
        if a.shape[0] != N or a.shape[2] != width:
            raise ValueError("expected %d views of %d-vectors, got %r" % (N, width, a.shape))
        step = a.shape[1] * width * 4
        return a.ctypes.data + np.arange(N, dtype=np.uint64) * np.uint64(step), np.full(N, a.shape[1], dtype=np.int32), a
    if isinstance(views, np.ndarray) and views.ndim == 2:  # one board seen in every view
        a = np.ascontiguousarray(views, dtype=np.float32)
        if a.shape[1] != width:
            raise ValueError("expected %d-vectors, got %r" % (width, a.shape))
        return np.full(N, a.ctypes.data, dtype=np.uint64), np.full(N, a.shape[0], dtype=np.int32), a
    arrs = [np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1, width)) for v in views]
    if len(arrs) != N:
        raise ValueError("expected %d views, got %d" % (N, len(arrs)))
    return (np.array([a.ctypes.data for a in arrs], dtype=np.uint64), np.array([len(a) for a in arrs], dtype=np.int32), arrs)


def _views(obj_views, *img_sets):
    """Object points + one or two sets of image points as per-view pointer arrays, checked for matching counts."""
    N = len(img_sets[0])
    optr, onp, okeep = _view_ptrs(obj_views, N, 3)
    out = []
    for img in img_sets:
        ptr, n, keep = _view_ptrs(img, N, 2)
        if not np.array_equal(n, onp):
            raise ValueError("object and image point counts differ")
        out.append((ptr, keep))
    return N, onp, (optr, okeep), out


TRACE_FIELDS = ("iter", "lambda_lg10_first", "lambda_lg10_accepted", "trials", "err_prev", "err", "step_norm2", "ms")


def release_cached_memory():
    """Returns the device blocks kept from finished calls to the driver (ccalib_release_cached_memory); bytes released."""
    return int(capi.load().ccalib_release_cached_memory())


def last_lm_trace():
    """Per-iteration records of the last LM run on this thread (ccalib_last_lm_trace): rows of TRACE_FIELDS."""
    lib = capi.load()
    n = C.c_int()
    buf = np.zeros((1000, len(TRACE_FIELDS)))
    capi.check(lib.ccalib_last_lm_trace(_p(buf), C.c_int(buf.shape[0]), C.byref(n)))
    return buf[:min(n.value, buf.shape[0])].copy()


def calibrate_camera(obj_views, img_views, size, K=None, dist=None, ndist=5, flags=0,
                     criteria=(TERM_COUNT | TERM_EPS, 30, DBL_EPSILON), init_rt=None):
    """cv::calibrateCamera.  Returns dict(rms, K, dist, rvecs, tvecs, iters, jac_passes, err_passes, ...)."""
    lib = capi.load()
    N, npts, (obj, _ko), [(img, _ki)] = _views(obj_views, img_views)
    A = np.eye(3).reshape(-1) if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(-1).copy()
    d = np.zeros(ndist)
    if dist is not None:
        dd = np.asarray(dist, dtype=np.float64).reshape(-1)
        d[:min(ndist, dd.size)] = dd[:ndist]
    rv, tv = np.zeros((N, 3)), np.zeros((N, 3))
    irt = None if init_rt is None else np.ascontiguousarray(init_rt, dtype=np.float64).reshape(N, 6)
    rms = C.c_double()
    st = capi.Stats()
    capi.check(lib.ccalib_calibrate_camera_views(
        C.c_int(N), _p(npts), _p(obj), _p(img), C.c_int(size[0]), C.c_int(size[1]), _p(A), _p(d), C.c_int(ndist),
        _p(rv), _p(tv), C.c_int(flags), C.c_int(criteria[0]), C.c_int(criteria[1]), C.c_double(criteria[2]),
        _p(irt), C.byref(rms), C.byref(st)))
    return dict(rms=rms.value, K=A.reshape(3, 3), dist=d, rvecs=rv, tvecs=tv, iters=st.iters,
                jac_passes=st.jac_passes, err_passes=st.err_passes, lambda_lg10=st.lambda_lg10,
                final_err=st.final_err, device_ms=st.device_ms)


def stereo_calibrate(obj_views, img1, img2, K1, d1, K2, d2, size, flags,
                     criteria=(TERM_COUNT | TERM_EPS, 30, 1e-6), init_rt=None, init_lr=None):
    """cv::stereoCalibrate.  Distortion vectors come back with 14 entries."""
    lib = capi.load()
    N, npts, (obj, _ko), [(i1, _k1), (i2, _k2)] = _views(obj_views, img1, img2)
    A1 = np.eye(3).reshape(-1) if K1 is None else np.ascontiguousarray(K1, dtype=np.float64).reshape(-1).copy()
    A2 = np.eye(3).reshape(-1) if K2 is None else np.ascontiguousarray(K2, dtype=np.float64).reshape(-1).copy()
    D1, D2 = np.zeros(14), np.zeros(14)
    if d1 is not None:
        D1[:np.size(d1)] = np.ravel(d1)
    if d2 is not None:
        D2[:np.size(d2)] = np.ravel(d2)
    R, T, E, F = np.zeros(9), np.zeros(3), np.zeros(9), np.zeros(9)
    rt = np.zeros((N, 6))
    irt = None if init_rt is None else np.ascontiguousarray(init_rt, dtype=np.float64).reshape(N, 6)
    ilr = None if init_lr is None else np.ascontiguousarray(init_lr, dtype=np.float64).reshape(6)
    rms = C.c_double()
    st = capi.Stats()
    capi.check(lib.ccalib_stereo_calibrate_views(
        C.c_int(N), _p(npts), _p(obj), _p(i1), _p(i2), _p(A1), _p(D1), _p(A2), _p(D2), C.c_int(size[0]),
        C.c_int(size[1]), _p(R), _p(T), _p(E), _p(F), C.c_int(flags), C.c_int(criteria[0]), C.c_int(criteria[1]),
        C.c_double(criteria[2]), _p(irt), _p(ilr), _p(rt), C.byref(rms), C.byref(st)))
    return dict(rms=rms.value, K1=A1.reshape(3, 3), d1=D1, K2=A2.reshape(3, 3), d2=D2, R=R.reshape(3, 3), T=T,
                E=E.reshape(3, 3), F=F.reshape(3, 3), rt=rt, iters=st.iters, jac_passes=st.jac_passes,
                err_passes=st.err_passes, lambda_lg10=st.lambda_lg10, final_err=st.final_err,
                device_ms=st.device_ms)


def write_mono_yaml(path, size, K, dist, rvecs, tvecs):
    """The file monocular_chessboard.cpp:321-328 writes: imageSize, M, D, R, t (cv::FileStorage YAML)."""
    lib = capi.load()
    d = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    rv = np.ascontiguousarray(rvecs, dtype=np.float64).reshape(-1, 3)
    tv = np.ascontiguousarray(tvecs, dtype=np.float64).reshape(-1, 3)
    capi.check(lib.ccalib_write_mono_yaml(str(path).encode(), C.c_int(size[0]), C.c_int(size[1]),
                                          _p(np.ascontiguousarray(K, dtype=np.float64).reshape(-1)), _p(d), C.c_int(d.size),
                                          C.c_int(rv.shape[0]), _p(rv), _p(tv)))


def write_stereo_yaml(intrinsics_path, extrinsics_path, size, K1, d1, K2, d2, R, T, E=None, F=None):
    """The two files binocular_chessboard.cpp:203-210 and :234-241 write (M1 D1 M2 D2; imageSize R T [E F])."""
    lib = capi.load()
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    a1, a2 = f(d1), f(d2)
    capi.check(lib.ccalib_write_stereo_yaml(str(intrinsics_path).encode(), str(extrinsics_path).encode(),
                                            C.c_int(size[0]), C.c_int(size[1]), _p(f(K1)), _p(a1), C.c_int(a1.size),
                                            _p(f(K2)), _p(a2), C.c_int(a2.size), _p(f(R)), _p(f(T)), _p(f(E)), _p(f(F))))


def reprojection_error(obj_views, img_views, K, dist, rvecs, tvecs, want_points=False):
    """projectPoints over all views + the RMS the reference recomputes from it.
    Returns (total_rms, per_view_rms[, projected (T,2) float32])."""
    lib = capi.load()
    N, npts, obj, img = _flatten(obj_views, img_views)
    d = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    pv = np.zeros(N)
    proj = np.zeros_like(img) if want_points else None
    tot = C.c_double()
    capi.check(lib.ccalib_reprojection_error(
        C.c_int(N), _p(npts), _p(obj), _p(img), _p(np.ascontiguousarray(K, dtype=np.float64).reshape(-1)), _p(d),
        C.c_int(d.size), _p(np.ascontiguousarray(rvecs, dtype=np.float64)),
        _p(np.ascontiguousarray(tvecs, dtype=np.float64)), _p(proj), _p(pv), C.byref(tot)))
    return (tot.value, pv, proj) if want_points else (tot.value, pv)


def init_camera_matrix_2d(obj_views, img_views, size, aspect=0.0):
    lib = capi.load()
    if isinstance(obj_views, np.ndarray) and obj_views.ndim == 2 and isinstance(img_views, np.ndarray) and img_views.ndim == 3:
        sol = Solver(obj_views, img_views)  # one board: nothing to replicate on the host
        try:
            return sol.init_intrinsics(size, aspect=aspect)
        finally:
            sol.close()
    N, npts, obj, img = _flatten(obj_views, img_views)
    A = np.zeros(9)
    capi.check(lib.ccalib_init_camera_matrix_2d(C.c_int(N), _p(npts), _p(obj), _p(img), C.c_int(size[0]),
                                                C.c_int(size[1]), C.c_double(aspect), _p(A)))
    return A.reshape(3, 3)


class Solver:
    """Device-resident problem (ccalib_solver_*): what bench.py times and multi-GPU runs shard."""

    def __init__(self, obj, img1, img2=None, device=0, shared_board=True):
        lib = capi.load()
        self._lib = lib
        shared = bool(shared_board and isinstance(obj, np.ndarray) and np.asarray(obj).ndim == 2)
        if shared and isinstance(img1, np.ndarray) and img1.ndim == 3:  # nothing to gather
            N, npts = img1.shape[0], np.full(img1.shape[0], img1.shape[1], dtype=np.int32)
            i1 = np.ascontiguousarray(img1, dtype=np.float32).reshape(-1, 2)
            if len(obj) != img1.shape[1]:
                raise ValueError("object and image point counts differ")
            o = None
        else:
            N, npts, o, i1 = _flatten(obj, img1)
        self.nviews, self.npts, self.total = N, npts, int(npts.sum())
        self.ncams = 1 if img2 is None else 2
        i2 = None
        if img2 is not None:
            i2 = (np.ascontiguousarray(img2, dtype=np.float32).reshape(-1, 2) if isinstance(img2, np.ndarray) and img2.ndim == 3
                  else np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1, 2) for v in img2])))
            if len(i2) != len(i1):
                raise ValueError("the two cameras' point counts differ")
        o_up = np.ascontiguousarray(np.asarray(obj, dtype=np.float32).reshape(-1, 3)) if shared else o
        self._h = C.c_void_p()
        capi.check(lib.ccalib_solver_create(C.byref(self._h), C.c_int(device), C.c_int(self.ncams), C.c_int(N),
                                            _p(npts), _p(o_up), C.c_int(1 if shared else 0), _p(i1), _p(i2)))
        self._cb = None

    def close(self):
        if self._h:
            self._lib.ccalib_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ns(self):
        n = C.c_int()
        capi.check(self._lib.ccalib_solver_num_shared(self._h, C.byref(n)))
        return n.value

    def set_start(self, intr, imask, rt, aspect=None, same_focal=False, lr=None, fix_intrinsic=False):
        intr = np.ascontiguousarray(intr, dtype=np.float64).reshape(self.ncams, 18)
        imask = np.ascontiguousarray(imask, dtype=np.uint8).reshape(self.ncams, 18)
        asp = np.zeros(self.ncams) if aspect is None else np.ascontiguousarray(aspect, dtype=np.float64).reshape(self.ncams)
        rt = None if rt is None else np.ascontiguousarray(rt, dtype=np.float64).reshape(self.nviews, 6)
        lrp = None if lr is None else np.ascontiguousarray(lr, dtype=np.float64).reshape(6)
        capi.check(self._lib.ccalib_solver_set_start(self._h, _p(intr), _p(imask), _p(asp),
                                                     C.c_int((1 if same_focal else 0) | (2 if fix_intrinsic else 0)),
                                                     _p(rt), _p(lrp)))

    def init_intrinsics(self, size, cam=0, aspect=0.0):
        """cv::initCameraMatrix2D on the device for camera `cam` (summed over ranks in a sharded run)."""
        A = np.zeros(9)
        capi.check(self._lib.ccalib_solver_init_intrinsics(self._h, C.c_int(cam), C.c_int(size[0]), C.c_int(size[1]),
                                                           C.c_double(aspect), _p(A)))
        return A.reshape(3, 3)

    def init_poses(self, K, dist=None, cam=0, want=True):
        """cvFindExtrinsicCameraParams2 for every view on the device; camera 0's poses stay there as the
        starting poses (set_start(rt=None)).  Returns (nviews, 6) r|t when want."""
        d = np.zeros(14)
        if dist is not None:
            d[:np.size(dist)] = np.ravel(dist)
        rt = np.zeros((self.nviews, 6)) if want else None
        capi.check(self._lib.ccalib_solver_init_poses(self._h, C.c_int(cam),
                                                      _p(np.ascontiguousarray(K, dtype=np.float64).reshape(-1)), _p(d), _p(rt)))
        return rt

    def init_relative_pose(self):
        lr = np.zeros(6)
        capi.check(self._lib.ccalib_solver_init_relative_pose(self._h, _p(lr)))
        return lr

    def use_rccl(self, nranks, rank, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        capi.check(self._lib.ccalib_solver_use_rccl(self._h, C.c_int(nranks), C.c_int(rank), buf))

    def use_callback(self, fn):
        """fn(numpy float64 array) must sum the array over all ranks in place."""
        def tramp(_user, ptr, n):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(n,)))
                return 0
            except Exception:  # noqa: BLE001 - reported through the status code
                return 1
        self._cb = capi.ALLREDUCE_FN(tramp)
        capi.check(self._lib.ccalib_solver_use_callback(self._h, self._cb, None))

    def run(self, criteria=(TERM_COUNT | TERM_EPS, 30, DBL_EPSILON), use_lu=False):
        st = capi.Stats()
        capi.check(self._lib.ccalib_solver_run(self._h, C.c_int(criteria[0]), C.c_int(criteria[1]),
                                               C.c_double(criteria[2]), C.c_int(0 if use_lu else 1), C.byref(st)))
        return dict(iters=st.iters, jac_passes=st.jac_passes, err_passes=st.err_passes,
                    lambda_lg10=st.lambda_lg10, final_err=st.final_err, device_ms=st.device_ms)

    def get(self):
        intr = np.zeros((self.ncams, 18))
        rt = np.zeros((self.nviews, 6))
        lr = np.zeros(6)
        err = C.c_double()
        capi.check(self._lib.ccalib_solver_get(self._h, _p(intr), _p(rt), _p(lr), C.byref(err)))
        return dict(intr=intr, rt=rt, lr=lr, err=err.value)

    def counters(self):
        """dict(ahead, used, dropped): iterations the GPU ran ahead of the host state machine (see ccalib.h)."""
        c = (C.c_longlong * 3)()
        capi.check(self._lib.ccalib_solver_counters(self._h, c))
        return dict(ahead=c[0], used=c[1], dropped=c[2])

    def executed_passes(self):
        """(Jacobian passes, residual-only passes) actually enqueued since the solver was created."""
        c = (C.c_longlong * 2)()
        capi.check(self._lib.ccalib_solver_executed_passes(self._h, c))
        return int(c[0]), int(c[1])

    def describe(self):
        """How the Jacobian pass runs: dict(kernel ('vector' | 'fused16' | 'fused12'), cams_per_launch, grid, ncols, ncams)."""
        o = (C.c_int * 8)()
        capi.check(self._lib.ccalib_solver_describe(self._h, o))
        return dict(kernel=("vector", "fused16", "fused12")[o[0]], cams_per_launch=o[1], grid=o[2], ncols=o[3], ncams=o[4],
                    sums_quad=bool(o[5]))

    def profile_jac(self, on):
        """Switches in-loop timing of the Jacobian launches on/off; returns (mean ms, launches) recorded so far."""
        ms, n = C.c_double(), C.c_int()
        capi.check(self._lib.ccalib_solver_profile_jac(self._h, C.c_int(1 if on else 0), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def eval_error(self):
        e = C.c_double()
        capi.check(self._lib.ccalib_solver_eval_error(self._h, C.byref(e)))
        return e.value

    def eval_jacobian(self):
        ns = self.ns
        U, gs = np.zeros((ns, ns)), np.zeros(ns)
        e = C.c_double()
        capi.check(self._lib.ccalib_solver_eval_jacobian(self._h, _p(U), _p(gs), C.byref(e)))
        return U, gs, e.value

    def fetch_blocks(self):
        """Per-view V, W, ge of the last Jacobian pass (W rows: 18 intrinsics, or the 42 stereo shared parameters)."""
        nrow = 18 if self.ncams == 1 else 42
        V, W, ge = np.zeros((self.nviews, 6, 6)), np.zeros((self.nviews, nrow, 6)), np.zeros((self.nviews, 6))
        capi.check(self._lib.ccalib_solver_fetch_blocks(self._h, _p(V), _p(W), _p(ge)))
        return V, W, ge

    def reduce(self, lam, nfree):
        S, b = np.zeros((nfree, nfree)), np.zeros(nfree)
        capi.check(self._lib.ccalib_solver_reduce(self._h, C.c_double(lam), _p(S), _p(b)))
        return S, b

    def time_kernels(self, reps=10):
        """Mean launch duration in ms (HIP events): dict(jac_shared, jac_view, err) for camera 0."""
        ms = (C.c_double * 3)()
        capi.check(self._lib.ccalib_solver_time_kernels(self._h, C.c_int(reps), ms))
        return dict(jac_shared=ms[0], jac_view=ms[1], err=ms[2])


def mono_start(cam_or_K, dist5=None, flags=0):
    """(intr18, mask18, aspect) for calibrateCamera-style flags, given a starting K/dist."""
    K = np.asarray(cam_or_K, dtype=np.float64)
    intr = np.zeros(18)
    intr[:4] = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    if dist5 is not None:
        d = np.ravel(dist5)
        intr[4:4 + d.size] = d
    mask = np.ones(18, dtype=np.uint8)
    f = flags
    if f & CALIB_FIX_ASPECT_RATIO:
        mask[0] = 0
    if f & CALIB_FIX_FOCAL_LENGTH:
        mask[0] = mask[1] = 0
    if f & CALIB_FIX_PRINCIPAL_POINT:
        mask[2] = mask[3] = 0
    if f & (CALIB_ZERO_TANGENT_DIST | CALIB_FIX_TANGENT_DIST):
        mask[6] = mask[7] = 0
        if f & CALIB_ZERO_TANGENT_DIST:
            intr[6] = intr[7] = 0
    if f & CALIB_FIX_K1:
        mask[4] = 0
    if f & CALIB_FIX_K2:
        mask[5] = 0
    if f & CALIB_FIX_K3:
        mask[8] = 0
    rat = bool(f & CALIB_RATIONAL_MODEL)
    mask[9] = rat and not (f & CALIB_FIX_K4)
    mask[10] = rat and not (f & CALIB_FIX_K5)
    mask[11] = rat and not (f & CALIB_FIX_K6)
    mask[12:] = 0
    aspect = K[0, 0] / K[1, 1] if f & CALIB_FIX_ASPECT_RATIO else 0.0
    return intr, mask, aspect
