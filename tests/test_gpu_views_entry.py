"""ccalib_calibrate_camera_views / ccalib_stereo_calibrate_views: one pointer per view (the memory a
vector<vector<Point3f>> holds, monocular_chessboard.cpp:249, :295-305) must give exactly what the flat entry points
give on the same numbers -- shared board, N separate copies of it, ragged views, the Z-clearing path, bad arguments."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SIZE = (640, 480)


def _flat_mono(cc, npts, obj, img, ndist=5, flags=0):
    capi = cc.capi
    lib = capi.load()
    N = len(npts)
    A, d, rv, tv = np.eye(3).reshape(-1), np.zeros(ndist), np.zeros((N, 3)), np.zeros((N, 3))
    rms, st = C.c_double(), capi.Stats()
    capi.check(lib.ccalib_calibrate_camera(
        C.c_int(N), cc._p(npts), cc._p(obj), cc._p(img), C.c_int(SIZE[0]), C.c_int(SIZE[1]), cc._p(A), cc._p(d), C.c_int(ndist),
        cc._p(rv), cc._p(tv), C.c_int(flags), C.c_int(3), C.c_int(30), C.c_double(cc.DBL_EPSILON), None, C.byref(rms),
        C.byref(st)))
    return dict(rms=rms.value, K=A.reshape(3, 3), dist=d, rvecs=rv, tvecs=tv, iters=st.iters)


def _same(a, b):
    assert a["iters"] == b["iters"]
    for k in ("rms", "K", "dist", "rvecs", "tvecs"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_mono_views_equal_flat_shared_board_and_separate_copies(pkg):
    cc, capi, synth = pkg, pkg.capi, pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 3000, 0xC0FFEE + 41)
    N, npts, o, img = cc._flatten(obj, ms.img)
    want = _flat_mono(cc, npts, o, img)
    _same(cc.calibrate_camera(obj, ms.img, SIZE), want)                       # every view points at one board
    copies = [obj.copy() for _ in range(N)]                                   # every view has its own copy
    _same(cc.calibrate_camera(copies, [ms.img[i] for i in range(N)], SIZE), want)


def test_mono_views_ragged_and_distinct_boards(pkg):
    cc, capi, synth = pkg, pkg.capi, pkg.synth
    rng = np.random.default_rng(7)
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 400, 0xC0FFEE + 42)
    keep = [np.sort(rng.choice(len(obj), size=int(rng.integers(20, len(obj) + 1)), replace=False)) for _ in range(400)]
    ov = [obj[k] for k in keep]
    iv = [ms.img[i][k] for i, k in enumerate(keep)]
    npts = np.array([len(k) for k in keep], dtype=np.int32)
    want = _flat_mono(cc, npts, np.ascontiguousarray(np.concatenate(ov)), np.ascontiguousarray(np.concatenate(iv)))
    _same(cc.calibrate_camera(ov, iv, SIZE), want)


def test_mono_views_clear_a_negligible_z(pkg):
    cc, capi, synth = pkg, pkg.capi, pkg.synth
    """Z below OpenCV's 1e-5 planarity threshold is cleared before use, in either shape of the input."""
    obj = synth.chessboard(9, 6, 25.0).copy()
    ms = synth.make_views(obj, 300, 0xC0FFEE + 43)
    tilted = obj.copy()
    tilted[:, 2] = 3e-6 * np.sin(np.arange(len(obj)))
    clean = cc.calibrate_camera(obj, ms.img, SIZE)
    _same(cc.calibrate_camera(tilted, ms.img, SIZE), clean)
    N, npts, o, img = cc._flatten(tilted, ms.img)
    _same(_flat_mono(cc, npts, o, img), clean)
    _same(cc.calibrate_camera([tilted.copy() for _ in range(N)], list(ms.img), SIZE), clean)


def test_mono_views_reject_nonplanar_and_null_views(pkg):
    cc, capi, synth = pkg, pkg.capi, pkg.synth
    obj = synth.chessboard(9, 6, 25.0).copy()
    ms = synth.make_views(obj, 50, 0xC0FFEE + 44)
    bent = obj.copy()
    bent[::2, 2] = 5.0
    with pytest.raises(cc.CcalibError, match="non-planar"):
        cc.calibrate_camera(bent, ms.img, SIZE)
    lib = capi.load()
    N, npts, o, img = cc._flatten(obj, ms.img)
    optr = np.full(N, o.ctypes.data, dtype=np.uint64)
    iptr = img.ctypes.data + np.arange(N, dtype=np.uint64) * np.uint64(8 * len(obj))
    iptr[17] = 0
    A, d, rv, tv = np.eye(3).reshape(-1), np.zeros(5), np.zeros((N, 3)), np.zeros((N, 3))
    rms = C.c_double()
    rc = lib.ccalib_calibrate_camera_views(
        C.c_int(N), cc._p(npts), cc._p(optr), cc._p(iptr), C.c_int(640), C.c_int(480), cc._p(A), cc._p(d), C.c_int(5),
        cc._p(rv), cc._p(tv), C.c_int(0), C.c_int(3), C.c_int(30), C.c_double(1e-9), None, C.byref(rms), None)
    assert rc == capi.ERR_ARG and b"view 17" in lib.ccalib_last_error()


def test_stereo_views_equal_flat(pkg):
    cc, capi, synth = pkg, pkg.capi, pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 1500, 0xC0FFEE + 45)
    flags = (cc.CALIB_FIX_ASPECT_RATIO | cc.CALIB_ZERO_TANGENT_DIST | cc.CALIB_USE_INTRINSIC_GUESS | cc.CALIB_SAME_FOCAL_LENGTH |
             cc.CALIB_RATIONAL_MODEL | cc.CALIB_FIX_K3 | cc.CALIB_FIX_K4 | cc.CALIB_FIX_K5)
    K1 = cc.init_camera_matrix_2d(obj, st.img1, SIZE)
    K2 = cc.init_camera_matrix_2d(obj, st.img2, SIZE)
    got = cc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, SIZE, flags, criteria=(3, 100, 1e-5))
    lib = capi.load()
    N, npts, o, i1 = cc._flatten(obj, st.img1)
    i2 = cc._flatten(obj, st.img2)[3]
    A1, A2, D1, D2 = K1.reshape(-1).copy(), K2.reshape(-1).copy(), np.zeros(14), np.zeros(14)
    R, T, E, F, rt = np.zeros(9), np.zeros(3), np.zeros(9), np.zeros(9), np.zeros((N, 6))
    rms, s = C.c_double(), capi.Stats()
    capi.check(lib.ccalib_stereo_calibrate(
        C.c_int(N), cc._p(npts), cc._p(o), cc._p(i1), cc._p(i2), cc._p(A1), cc._p(D1), cc._p(A2), cc._p(D2), C.c_int(640),
        C.c_int(480), cc._p(R), cc._p(T), cc._p(E), cc._p(F), C.c_int(flags), C.c_int(3), C.c_int(100), C.c_double(1e-5), None,
        None, cc._p(rt), C.byref(rms), C.byref(s)))
    assert s.iters == got["iters"] and rms.value == got["rms"]
    for a, b in ((A1, got["K1"]), (A2, got["K2"]), (D1, got["d1"]), (D2, got["d2"]), (R, got["R"]), (T, got["T"]), (E, got["E"]),
                 (F, got["F"]), (rt, got["rt"])):
        assert np.array_equal(np.ravel(a), np.ravel(b))


def test_blocks_of_finished_calls_are_reused_and_released(pkg):
    """A finished call leaves its device blocks in the library's pool; the next call of the same shape takes them (same
    results, to the bit), and ccalib_release_cached_memory hands everything back."""
    cc, synth = pkg, pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 2000, 0xC0FFEE + 46)
    cc.release_cached_memory()
    first = cc.calibrate_camera(obj, ms.img, SIZE)
    second = cc.calibrate_camera(obj, ms.img, SIZE)   # runs in the first call's blocks, whatever they still hold
    _same(second, first)
    assert cc.release_cached_memory() > 0
    assert cc.release_cached_memory() == 0
    _same(cc.calibrate_camera(obj, ms.img, SIZE), first)


def test_doomed_trial_beside_its_retry_changes_nothing(pkg, monkeypatch):
    """Small stereo problems run a trial that is expected to be rejected on a second stream beside the retry queued for
    it: same LM record and same results as with one stream."""
    cc, synth = pkg, pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 3000, 0xC0FFEE + 47)
    flags = (cc.CALIB_FIX_ASPECT_RATIO | cc.CALIB_ZERO_TANGENT_DIST | cc.CALIB_USE_INTRINSIC_GUESS | cc.CALIB_SAME_FOCAL_LENGTH |
             cc.CALIB_RATIONAL_MODEL | cc.CALIB_FIX_K3 | cc.CALIB_FIX_K4 | cc.CALIB_FIX_K5)
    K1 = cc.init_camera_matrix_2d(obj, st.img1, SIZE)
    K2 = cc.init_camera_matrix_2d(obj, st.img2, SIZE)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CCALIB_AUX_STREAM", mode)
        got = cc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, SIZE, flags, criteria=(3, 60, 1e-7))
        runs[mode] = (got, cc.last_lm_trace())
    a, b = runs["1"], runs["0"]
    assert np.array_equal(a[1][:, :7], b[1][:, :7])
    assert a[1][:, 3].max() >= 2, "no iteration needed a retry: the second stream was never used"
    for k in ("rms", "K1", "K2", "d1", "d2", "R", "T", "rt"):
        assert np.array_equal(np.asarray(a[0][k]), np.asarray(b[0][k])), k


def test_sums_sent_by_every_workgroup_itself_change_nothing(pkg, monkeypatch):
    """Passes with fewer than 256 sums reach the host as {sum, sum XOR pass number} pairs written by each column-sum
    workgroup; the flag paths (one burst from the last workgroup or a publishing kernel) must hold the same bits.  Two
    calls in a row per mode, so the second one reads pinned blocks that still hold the first one's pairs; the stereo
    run mixes both ways inside one call (499 sums of its Jacobian pass with a flag, residual passes without)."""
    cc, synth = pkg, pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 1500, 0xC0FFEE + 51)
    st = synth.make_stereo(obj, 1200, 0xC0FFEE + 52)
    K1 = cc.init_camera_matrix_2d(obj, st.img1, SIZE)
    K2 = cc.init_camera_matrix_2d(obj, st.img2, SIZE)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CCALIB_DIRECT_SUMS", mode)
        mono = [cc.calibrate_camera(obj, ms.img, SIZE) for _ in range(2)]
        mono_trace = cc.last_lm_trace()
        stereo = cc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, SIZE, cc.CALIB_USE_INTRINSIC_GUESS,
                                     criteria=(3, 40, 1e-9))
        runs[mode] = (mono, mono_trace, stereo, cc.last_lm_trace())
    a, b = runs["1"], runs["0"]
    _same(a[0][1], a[0][0])
    _same(a[0][0], b[0][0])
    assert np.array_equal(a[1][:, :7], b[1][:, :7]) and np.array_equal(a[3][:, :7], b[3][:, :7])
    for k in ("rms", "K1", "K2", "d1", "d2", "R", "T", "rt"):
        assert np.array_equal(np.asarray(a[2][k]), np.asarray(b[2][k])), k


def test_poses_come_back_whole_when_they_take_several_staging_rounds(pkg):
    """400 000 views: the refined poses (19 MB) leave the device in two rounds through the pinned staging buffers and are
    split into rvecs / tvecs on the way; the solver-level download of the same run must hold the same numbers."""
    cc, synth = pkg, pkg.synth
    obj = synth.chessboard(4, 3, 30.0)
    ms = synth.make_views(obj, 40000, 0xC0FFEE + 49)
    img = np.ascontiguousarray(np.tile(ms.img, (10, 1, 1)))
    N = img.shape[0]
    got = cc.calibrate_camera(obj, img, SIZE, criteria=(3, 3, 1e-12))
    assert got["iters"] == 3 and np.isfinite(got["rvecs"]).all() and np.isfinite(got["tvecs"]).all()
    # the ten copies of every view end at the same pose, and no row was left unwritten or shifted
    rv, tv = got["rvecs"].reshape(10, -1, 3), got["tvecs"].reshape(10, -1, 3)
    assert np.array_equal(rv[0], rv[9]) and np.array_equal(tv[0], tv[9]) and np.array_equal(rv[3], rv[6])
    assert np.abs(tv[0][:, 2]).min() > 1.0
    # the same problem through the solver level (one hipMemcpy-shaped download of r|t rows)
    s = cc.Solver(obj, img)
    K0 = s.init_intrinsics(SIZE)
    s.init_poses(K0, None, want=False)
    intr, mask, _ = cc.mono_start(K0, np.zeros(5), 0)
    s.set_start(intr, mask, None)
    s.run(criteria=(3, 3, 1e-12))
    out = s.get()
    s.close()
    assert np.array_equal(out["rt"][:, :3], got["rvecs"]) and np.array_equal(out["rt"][:, 3:], got["tvecs"])
    assert N == 400000
