"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Tolerances are BASELINE.json's: RMS, intrinsics and distortion within 1e-6
relative, extrinsics within 1e-6 rad / unit, iteration count +-1.  The oracle
restates OpenCV 4.1.0 and is itself unpinned against a real OpenCV (none is
installed); see tests/test_oracle.py for what pins it instead.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float64).eps)
REL = 1e-6


def _close_params(got, ref, pose_tol=1e-6):
    assert abs(got["rms"] - ref["rms"]) <= REL * ref["rms"], (got["rms"], ref["rms"])
    assert np.allclose(got["K"], ref["K"], rtol=REL, atol=1e-9)
    assert np.allclose(got["dist"], ref["dist"], rtol=REL, atol=1e-9)
    assert np.abs(got["rvecs"] - ref["rvecs"]).max() < pose_tol
    assert np.abs(got["tvecs"] - ref["tvecs"]).max() < pose_tol * max(1.0, np.abs(ref["tvecs"]).max())


def _dist_tol(ref, alt):
    """Element-wise relative tolerance for a distortion vector: BASELINE's 1e-6, unless the minimum itself does not
    determine the coefficients that well on this input -- measured by the oracle against itself (`ref` from its
    dense solver, `alt` from its Schur solver): then four times that spread, which must stay below 1e-5."""
    nz = ref != 0
    spread = float((np.abs(alt - ref)[nz] / np.abs(ref[nz])).max()) if nz.any() else 0.0
    tol = max(REL, 4 * spread)
    assert tol <= 1e-5, tol
    return tol


def _start(pkg, synth, ms, flags=0, jitter=1.0):
    cam = ms.cam
    K = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.98, cam.cy - 2], [0, 0, 1.0]])
    d = cam.dist(5) * 0.8
    intr, mask, aspect = pkg.mono_start(K, d, flags)
    rt = synth.perturbed_poses(ms.rvec, ms.tvec, 99, rot_sigma=0.01 * jitter, trans_sigma=3.0 * jitter)
    return intr, mask, aspect, rt


@pytest.mark.parametrize("flags_name", ["zero", "aspect", "rational6"])
def test_jacobian_blocks_match_oracle(pkg, orc, synth, flags_name):
    flags = {"zero": 0, "aspect": pkg.CALIB_FIX_ASPECT_RATIO,
             "rational6": (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_RATIONAL_MODEL |
                           pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 | pkg.CALIB_FIX_K5)}[flags_name]
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 37, 7)
    intr, mask, aspect, rt = _start(pkg, synth, ms, flags)
    if flags_name == "rational6":
        intr[11] = 0.03  # k6 away from zero so the rational branch is exercised
    if aspect:
        intr[0] = intr[1] * aspect
    s = pkg.Solver(obj, ms.img)
    s.set_start(intr, mask, rt, aspect=[aspect])
    U, gs, err = s.eval_jacobian()
    V, W, ge = s.fetch_blocks()
    free = np.flatnonzero(mask)
    ref = orc.mono_blocks(obj, ms.img, intr, aspect, rt, free)
    assert abs(err - ref["err"]) < 1e-11 * ref["err"]
    assert abs(s.eval_error() - ref["err"]) < 1e-11 * ref["err"]
    Uf = np.triu(U[np.ix_(free, free)])
    scale = np.sqrt(np.outer(np.diag(ref["U"]), np.diag(ref["U"])))
    assert np.abs(Uf - np.triu(ref["U"])).max() / 1.0 <= 1e-10 * scale.max()
    assert np.all(np.abs(Uf - np.triu(ref["U"])) <= 1e-10 * scale + 1e-30)
    assert np.allclose(gs[free], ref["gc"], rtol=1e-9, atol=1e-9 * np.abs(ref["gc"]).max())
    assert np.allclose(V, ref["V"], rtol=1e-10, atol=1e-10 * np.abs(ref["V"]).max())
    assert np.allclose(ge, ref["ge"], rtol=1e-9, atol=1e-9 * np.abs(ref["ge"]).max())
    assert np.allclose(W[:, free, :], ref["W"], rtol=1e-9, atol=1e-10 * np.abs(ref["W"]).max())
    # Schur sums for one damping value, against the same algebra in numpy on the oracle's blocks
    lam = 1e-3
    Sref = np.zeros((len(free), len(free)))
    bref = np.zeros(len(free))
    for i in range(ms.img.shape[0]):
        Vd = ref["V"][i] + lam * np.diag(np.diag(ref["V"][i]))
        Y = np.linalg.solve(Vd, ref["W"][i].T)
        Sref += ref["W"][i] @ Y
        bref += Y.T @ ref["ge"][i]
    S, b = s.reduce(lam, len(free)) if True else (None, None)
    # reduce() reports every kernel column; the free ones are a subset in the same order
    s.close()
    kcols = {"zero": [0, 1, 2, 3, 4, 5, 6, 7, 8], "aspect": [1, 2, 3, 4, 5, 6, 7, 8],
             "rational6": [1, 2, 3, 4, 5, 11]}[flags_name]
    sel = [kcols.index(f) for f in free]
    assert np.allclose(S[np.ix_(sel, sel)], Sref, rtol=1e-8, atol=1e-9 * np.abs(Sref).max())
    assert np.allclose(b[sel], bref, rtol=1e-8, atol=1e-9 * np.abs(bref).max())


def test_config1_mono_chessboard_20_views(pkg, orc, synth):
    """BASELINE config 1: 9x6 board, 20 views, flags 0 -- full call incl. initialisation."""
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 20, 0xC0FFEE + 1)
    crit = (3, 30, 1e-9)
    got = pkg.calibrate_camera(obj, ms.img, (640, 480), criteria=crit)
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), criteria=crit, schur=False)  # dense, OpenCV-shaped
    _close_params(got, ref)
    assert abs(got["iters"] - ref["iters"]) <= 1
    # default criteria (30, DBL_EPSILON): the stop lands on round-off, results must still agree
    got = pkg.calibrate_camera(obj, ms.img, (640, 480))
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), schur=False)
    _close_params(got, ref)
    # the identity the reference prints (monocular_circle.cpp:96-103)
    rms2, per_view = pkg.reprojection_error(obj, ms.img, got["K"], got["dist"], got["rvecs"], got["tvecs"])
    assert abs(rms2 - got["rms"]) < 1e-5 * got["rms"]   # projectPoints rounds to float32


def test_reference_mono_chessboard_flags(pkg, orc, synth):
    """monocular_chessboard.cpp:226-227,315-317: 8x5 board at 20 mm, FIX_ASPECT_RATIO | USE_LU."""
    obj = synth.chessboard(8, 5, 20.0)
    ms = synth.make_views(obj, 25, 31, depth=(300.0, 900.0))
    flags = pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_USE_LU
    crit = (3, 30, 1e-9)
    got = pkg.calibrate_camera(obj, ms.img, (640, 480), flags=flags, criteria=crit)
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), flags=flags, criteria=crit, schur=False)
    _close_params(got, ref)
    assert abs(got["iters"] - ref["iters"]) <= 1
    assert got["K"][0, 0] == got["K"][1, 1]


def test_config2_circle_grid_1k_views(pkg, orc, synth):
    """BASELINE config 2: asymmetric circle grid, 88 points, 1000 noisy views, dist zeros(8,1)-style call."""
    obj = synth.asymmetric_circles(8, 11, 12.5)
    ms = synth.make_views(obj, 1000, 0xC0FFEE + 2)
    crit = (3, 30, 1e-9)
    got = pkg.calibrate_camera(obj, ms.img, (640, 480), criteria=crit)
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), criteria=crit, schur=True, nthreads=8)
    _close_params(got, ref)
    assert abs(got["iters"] - ref["iters"]) <= 1


def test_ragged_views_and_distinct_boards(pkg, orc, synth):
    rng = np.random.default_rng(5)
    base = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(base, 40, 17)
    objs, imgs = [], []
    for i in range(40):
        keep = np.sort(rng.choice(54, size=int(rng.integers(12, 55)), replace=False))
        objs.append(base[keep])
        imgs.append(ms.img[i][keep])
    crit = (3, 30, 1e-9)
    got = pkg.calibrate_camera(objs, imgs, (640, 480), criteria=crit)
    ref = orc.calibrate_camera(objs, imgs, (640, 480), criteria=crit, schur=False)
    _close_params(got, ref)
    assert abs(got["iters"] - ref["iters"]) <= 1


def test_intrinsic_guess_and_fixed_subsets(pkg, orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 30, 23)
    K0 = np.array([[780.0, 0, 322.0], [0, 790.0, 238.0], [0, 0, 1]])
    crit = (3, 40, 1e-9)
    for flags, nd in ((pkg.CALIB_USE_INTRINSIC_GUESS | pkg.CALIB_FIX_PRINCIPAL_POINT | pkg.CALIB_FIX_K3, 5),
                      (pkg.CALIB_USE_INTRINSIC_GUESS | pkg.CALIB_ZERO_TANGENT_DIST, 4),
                      (pkg.CALIB_USE_INTRINSIC_GUESS | pkg.CALIB_RATIONAL_MODEL, 8)):
        got = pkg.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(nd), ndist=nd, flags=flags, criteria=crit)
        ref = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(nd), ndist=nd, flags=flags,
                                   criteria=crit, schur=False)
        assert abs(got["rms"] - ref["rms"]) <= 1e-6 * ref["rms"]
        assert np.allclose(got["K"], ref["K"], rtol=1e-6)     # also for the nearly degenerate 8-coefficient model
        assert np.abs(got["rvecs"] - ref["rvecs"]).max() < 1e-6


@pytest.mark.parametrize("seed", range(8))
def test_random_flag_subsets_match_oracle(pkg, orc, synth, seed):
    """Flag words drawn at random from the CALIB_* bits the library accepts (every mask the flag -> mask mapping of
    cvCalibrateCamera2Internal can produce for the 12-coefficient model), with an intrinsic guess."""
    rng = np.random.default_rng(1000 + seed)
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 18, 50 + seed)
    cam = ms.cam
    optional = [pkg.CALIB_FIX_ASPECT_RATIO, pkg.CALIB_FIX_PRINCIPAL_POINT, pkg.CALIB_ZERO_TANGENT_DIST, pkg.CALIB_FIX_K1,
                pkg.CALIB_FIX_K2, pkg.CALIB_FIX_K3, pkg.CALIB_FIX_FOCAL_LENGTH, pkg.CALIB_FIX_TANGENT_DIST, pkg.CALIB_USE_LU]
    flags = pkg.CALIB_USE_INTRINSIC_GUESS
    for f in optional:
        if rng.random() < 0.35:
            flags |= f
    nd = 5
    if rng.random() < 0.4:   # rational model with a random subset of k4..k6 held
        flags |= pkg.CALIB_RATIONAL_MODEL
        nd = 8
        for f in (pkg.CALIB_FIX_K4, pkg.CALIB_FIX_K5, pkg.CALIB_FIX_K6):
            if rng.random() < 0.6:
                flags |= f
    K0 = np.array([[cam.fx * 1.02, 0, cam.cx - 2], [0, cam.fy * 1.02, cam.cy + 3], [0, 0, 1.0]])
    d0 = np.zeros(nd)
    d0[:5] = cam.dist(5) * 0.9
    crit = (3, 25, 1e-8)
    got = pkg.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=d0, ndist=nd, flags=flags, criteria=crit)
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=d0, ndist=nd, flags=flags, criteria=crit, schur=False)
    free_rational = nd == 8 and not all(flags & f for f in (pkg.CALIB_FIX_K4, pkg.CALIB_FIX_K5, pkg.CALIB_FIX_K6))
    tol = 1e-6
    if free_rational:
        # numerator and denominator polynomials nearly cancel on this data: the minimum is flat along one direction
        # and two correct solvers end at visibly different points of it.  The yardstick is the oracle itself --
        # its dense and its Schur solver on the same input -- and the HIP path may differ from the dense one by no
        # more than a few times what the oracle's two solvers differ by.
        alt = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=d0, ndist=nd, flags=flags, criteria=crit, schur=True)
        spread = max(np.abs(alt["K"] - ref["K"]).max() / np.abs(ref["K"]).max(), np.abs(alt["rvecs"] - ref["rvecs"]).max() / 10)
        tol = max(1e-6, 4 * spread)
        assert tol <= 2e-5, tol
    assert abs(got["rms"] - ref["rms"]) <= 1e-6 * ref["rms"], hex(flags)
    assert abs(got["iters"] - ref["iters"]) <= 1, hex(flags)
    assert np.allclose(got["K"], ref["K"], rtol=tol, atol=1e-9), hex(flags)
    assert np.abs(got["rvecs"] - ref["rvecs"]).max() < 10 * tol, hex(flags)
    if not free_rational:
        alt = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=d0, ndist=nd, flags=flags, criteria=crit, schur=True)
        assert np.allclose(got["dist"], ref["dist"], rtol=_dist_tol(ref["dist"], alt["dist"]), atol=1e-9), hex(flags)
    # held parameters come back exactly as given
    if flags & pkg.CALIB_FIX_PRINCIPAL_POINT:
        assert got["K"][0, 2] == K0[0, 2] and got["K"][1, 2] == K0[1, 2]
    if flags & pkg.CALIB_FIX_FOCAL_LENGTH:
        assert got["K"][1, 1] == K0[1, 1]
    if flags & pkg.CALIB_FIX_K2:
        assert got["dist"][1] == d0[1]
    if flags & pkg.CALIB_ZERO_TANGENT_DIST:
        assert got["dist"][2] == 0 and got["dist"][3] == 0


def test_config3_stereo_reference_flags(pkg, orc, synth):
    """BASELINE config 3 at reduced N: the flag set of binocular_chessboard.cpp:156-161, criteria (100, 1e-5)."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 200, 0xC0FFEE + 3)
    K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
    flags = (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_USE_INTRINSIC_GUESS |
             pkg.CALIB_SAME_FOCAL_LENGTH | pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 |
             pkg.CALIB_FIX_K5)
    crit = (3, 100, 1e-5)
    got = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit)
    ref = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit, schur=True)
    assert abs(got["rms"] - ref["rms"]) <= REL * ref["rms"]
    assert abs(got["iters"] - ref["iters"]) <= 1
    for key in ("K1", "K2", "R", "T", "E", "F"):
        assert np.allclose(got[key], ref[key], rtol=REL, atol=1e-8 * np.abs(ref[key]).max()), key
    assert np.allclose(got["d1"], ref["d1"], rtol=REL, atol=1e-9)
    assert np.allclose(got["d2"], ref["d2"], rtol=REL, atol=1e-9)
    assert np.abs(got["rt"] - ref["rt"])[:, :3].max() < 1e-6


def test_stereo_full_distortion_small_dense(pkg, orc, synth):
    """The all-free model (config 5's flag set) at a size the dense OpenCV-shaped oracle can solve."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 24, 77)
    K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
    crit = (3, 100, 1e-7)
    got = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), pkg.CALIB_USE_INTRINSIC_GUESS,
                               criteria=crit)
    ref = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), orc.USE_INTRINSIC_GUESS,
                               criteria=crit, schur=False)
    assert abs(got["rms"] - ref["rms"]) <= REL * ref["rms"]
    assert abs(got["iters"] - ref["iters"]) <= 1
    for key in ("K1", "K2", "R", "T", "E", "F"):
        assert np.allclose(got[key], ref[key], rtol=REL, atol=1e-8 * np.abs(ref[key]).max()), key
    alt = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), orc.USE_INTRINSIC_GUESS,
                               criteria=crit, schur=True)
    for key in ("d1", "d2"):
        assert np.allclose(got[key][:5], ref[key][:5], rtol=_dist_tol(ref[key][:5], alt[key][:5]), atol=1e-9), key


def test_stereo_fix_intrinsic_default_flag(pkg, orc, synth):
    """cv::stereoCalibrate's default flag word, CALIB_FIX_INTRINSIC: only om_LR, T_LR and the poses move, and the
    convergence norm runs over those alone (the intrinsics are not in the parameter vector)."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 60, 0xC0FFEE + 6)
    m1 = pkg.calibrate_camera(obj, st.img1, (640, 480))
    m2 = pkg.calibrate_camera(obj, st.img2, (640, 480))
    for crit in ((3, 100, 1e-6), (3, 100, 1e-9)):
        got = pkg.stereo_calibrate(obj, st.img1, st.img2, m1["K"], m1["dist"], m2["K"], m2["dist"], (640, 480),
                                   pkg.CALIB_FIX_INTRINSIC, criteria=crit)
        ref = orc.stereo_calibrate(obj, st.img1, st.img2, m1["K"], m1["dist"], m2["K"], m2["dist"], (640, 480),
                                   pkg.CALIB_FIX_INTRINSIC, criteria=crit, schur=False)
        assert abs(got["rms"] - ref["rms"]) <= REL * ref["rms"]
        assert abs(got["iters"] - ref["iters"]) <= 1
        assert np.array_equal(got["K1"], m1["K"]) and np.array_equal(got["d2"][:5], m2["dist"])   # untouched
        for key in ("R", "T", "E", "F"):
            assert np.allclose(got[key], ref[key], rtol=REL, atol=1e-8 * np.abs(ref[key]).max()), key
        assert np.abs(got["rt"] - ref["rt"])[:, :3].max() < 1e-6
    assert np.allclose(got["T"], [-60.0, 0.5, 1.0], atol=0.5)


@pytest.mark.parametrize("case", ["aspect8", "rational12", "ragged"])
def test_stereo_other_column_sets_and_ragged_views(pkg, orc, synth, case):
    """stereoCalibrate over the remaining built-in column sets (8 columns per camera with a fixed aspect ratio, all
    twelve with the rational model: the two-kernel vector path) and over views that each see their own subset of
    the board (per-view object points, different point counts)."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 22, 4100)
    K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
    flags = pkg.CALIB_USE_INTRINSIC_GUESS
    objs, i1, i2 = obj, st.img1, st.img2
    crit = (3, 60, 1e-7)
    if case == "aspect8":
        flags |= pkg.CALIB_FIX_ASPECT_RATIO
    elif case == "rational12":
        flags |= pkg.CALIB_RATIONAL_MODEL
        crit = (3, 25, 1e-7)
    else:
        rng = np.random.default_rng(8)
        keep = [np.sort(rng.choice(54, size=int(rng.integers(10, 55)), replace=False)) for _ in range(22)]
        objs = [obj[k] for k in keep]
        i1 = [st.img1[i][k] for i, k in enumerate(keep)]
        i2 = [st.img2[i][k] for i, k in enumerate(keep)]
    got = pkg.stereo_calibrate(objs, i1, i2, K1, None, K2, None, (640, 480), flags, criteria=crit)
    ref = orc.stereo_calibrate(objs, i1, i2, K1, None, K2, None, (640, 480), flags, criteria=crit, schur=False)
    tol = REL
    if case == "rational12":
        # The 12-coefficient model is nearly degenerate on this data and the run stops at its iteration limit, in
        # the middle of a walk along a flat valley: two correct solvers are then visibly apart.  The yardstick is
        # the oracle's own dense-vs-Schur spread on this input; the HIP path may be a few times that from the dense
        # result, and the yardstick itself must stay small.
        alt = orc.stereo_calibrate(objs, i1, i2, K1, None, K2, None, (640, 480), flags, criteria=crit, schur=True)
        spread = max([abs(alt["rms"] - ref["rms"]) / ref["rms"]] +
                     [np.abs(alt[k] - ref[k]).max() / np.abs(ref[k]).max() for k in ("K1", "K2", "R", "T")])
        tol = max(REL, 4 * spread)
        assert tol <= 1e-5, tol
    assert abs(got["rms"] - ref["rms"]) <= tol * ref["rms"]
    assert abs(got["iters"] - ref["iters"]) <= 1
    for key in ("K1", "K2", "R", "T"):
        assert np.allclose(got[key], ref[key], rtol=tol, atol=tol * np.abs(ref[key]).max()), key
    assert np.abs(got["rt"] - ref["rt"])[:, :3].max() < 2 * tol
    if case == "aspect8":
        assert got["K1"][0, 0] / got["K1"][1, 1] == pytest.approx(K1[0, 0] / K1[1, 1], rel=1e-12)


@pytest.mark.parametrize("case", ["reference6", "aspect8", "plain9", "rational_aspect11", "rational12"])
def test_stereo_sum_kernels_agree_for_every_column_set(pkg, synth, monkeypatch, case):
    """The stereo Schur sums have two kernels (four lanes per view for small problems, one lane per view in three
    passes for large ones; size picks one) and the column sums two ways to reach the host (from the last workgroup
    or from a publishing kernel; column count picks one).  The same six LM iterations through each: the two sum
    kernels add in different orders, so they agree to round-off, not bitwise; the two ways of publishing carry
    the same numbers, so they agree bitwise.  (The parity tests against the oracle run whichever the size picks.)"""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 150, 977)     # 150 views: two full 64-view workgroups and a ragged third
    K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
    flags = pkg.CALIB_USE_INTRINSIC_GUESS
    if case == "reference6":
        flags |= (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_SAME_FOCAL_LENGTH |
                  pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 | pkg.CALIB_FIX_K5)
    elif case == "aspect8":
        flags |= pkg.CALIB_FIX_ASPECT_RATIO
    elif case == "rational_aspect11":
        flags |= pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_ASPECT_RATIO
    elif case == "rational12":
        flags |= pkg.CALIB_RATIONAL_MODEL
    crit = (1, 6, 0.0)
    outs = {}
    for quad, split in (("1", "0"), ("0", "0"), ("0", "1"), ("1", "1")):
        monkeypatch.setenv("CCALIB_SUMS_QUAD", quad)
        monkeypatch.setenv("CCALIB_SPLIT_PUBLISH", split)
        outs[quad, split] = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit)
    for quad in "01":
        a, b = outs[quad, "0"], outs[quad, "1"]
        assert a["rms"] == b["rms"] and a["iters"] == b["iters"]
        for key in ("K1", "K2", "d1", "d2", "R", "T", "rt"):
            assert np.array_equal(a[key], b[key]), (quad, key)
    a, b = outs["1", "0"], outs["0", "0"]
    assert a["iters"] == b["iters"]
    assert abs(a["rms"] - b["rms"]) <= 1e-10 * b["rms"]
    for key in ("K1", "K2", "R", "T"):
        assert np.abs(a[key] - b[key]).max() <= 1e-8 * np.abs(b[key]).max(), key
    assert np.abs(a["rt"] - b["rt"]).max() < 1e-7


def test_retry_launched_behind_a_doomed_trial_changes_nothing(pkg, synth, monkeypatch):
    """With the reference's stereo flag set every first trial of an iteration is rejected and its retry accepted.  The
    solver then puts the retry's kernels into the stream behind the first trial's (it works the retry's step out
    with the driver's own function while the GPU is busy) and the driver's call is answered from them only when its
    step is bit for bit the prepared one: a run with the mechanism equals a run without it bit for bit, and the
    counters show that it engaged."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 3000, 0xC0FFEE + 3)
    K = [pkg.init_camera_matrix_2d(obj, im, (640, 480)) for im in (st.img1, st.img2)]
    intr = np.zeros((2, 18))
    for c in range(2):
        intr[c, :4] = [(K[0][0, 0] + K[1][0, 0]) / 2, (K[0][1, 1] + K[1][1, 1]) / 2,
                       (K[0][0, 2] + K[1][0, 2]) / 2, (K[0][1, 2] + K[1][1, 2]) / 2]
    mask = np.zeros((2, 18), dtype=np.uint8)
    mask[:, [1, 2, 3, 4, 5, 11]] = 1           # fy cx cy k1 k2 k6: the reference's stereo flag set
    aspect = [intr[0, 0] / intr[0, 1], intr[1, 0] / intr[1, 1]]
    rt0 = synth.perturbed_poses(st.rvec, st.tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    lr0 = np.r_[st.r_lr, st.t_lr] + np.array([1e-3, -1e-3, 5e-4, 0.5, -0.2, 0.3])
    outs = []
    for off in ("1", "0"):
        monkeypatch.setenv("CCALIB_NO_RETRY_AHEAD", off)
        s = pkg.Solver(obj, st.img1, st.img2)
        s.set_start(intr, mask, rt0, aspect=aspect, same_focal=True, lr=lr0)
        stt = s.run(criteria=(3, 25, 1e-12))
        outs.append((stt, s.get(), s.counters()))
        s.close()
    (st0, o0, c0), (st1, o1, c1) = outs
    assert st0["iters"] == st1["iters"] and st0["err_passes"] == st1["err_passes"]
    assert st0["err_passes"] >= st0["iters"] + 5                   # there were rejected trials to run ahead of
    assert o0["err"] == o1["err"]
    for key in ("intr", "rt", "lr"):
        assert np.array_equal(o0[key], o1[key]), key
    assert c1["used"] >= c0["used"] + 5, (c0, c1, st1)


def test_config5_shape_17x12_board_stereo_and_mono(pkg, orc, synth):
    """BASELINE config 5's shape at a size the dense oracle solves: 17x12 board (204 points per view, so
    every view spans four 64-point chunks of the fused kernel), both cameras with k1-k3, p1, p2 free."""
    obj = synth.chessboard(17, 12, 10.0)
    st = synth.make_stereo(obj, 16, 0xC0FFEE + 5)
    K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
    assert np.allclose(K1, orc.init_camera_matrix_2d(obj, st.img1, (640, 480)), rtol=1e-8)
    crit = (3, 60, 1e-7)
    got = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), pkg.CALIB_USE_INTRINSIC_GUESS,
                               criteria=crit)
    ref = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), orc.USE_INTRINSIC_GUESS,
                               criteria=crit, schur=False)
    assert abs(got["rms"] - ref["rms"]) <= REL * ref["rms"]
    assert abs(got["iters"] - ref["iters"]) <= 1
    for key in ("K1", "K2", "R", "T", "E", "F"):
        assert np.allclose(got[key], ref[key], rtol=REL, atol=1e-8 * np.abs(ref[key]).max()), key
    alt = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), orc.USE_INTRINSIC_GUESS,
                               criteria=crit, schur=True)
    for key in ("d1", "d2"):
        assert np.allclose(got[key][:5], ref[key][:5], rtol=_dist_tol(ref[key][:5], alt[key][:5]), atol=1e-9), key
    assert np.abs(got["rt"] - ref["rt"])[:, :3].max() < 1e-6
    # one camera of it through calibrateCamera (initialisation included)
    crit = (3, 30, 1e-9)
    gm = pkg.calibrate_camera(obj, st.img1, (640, 480), criteria=crit)
    rm = orc.calibrate_camera(obj, st.img1, (640, 480), criteria=crit, schur=False)
    _close_params(gm, rm)
    assert abs(gm["iters"] - rm["iters"]) <= 1


def test_extreme_shapes_one_view_and_a_1200_point_board(pkg, orc, synth):
    """Smallest and largest shapes the kernels take: a single view (one lane of one wavefront carries the whole
    elimination), three views (less than one tile of four), and a 40x30 board whose 1200 points per view exceed
    what the shared-board LDS copy holds (the board is then replicated per view and walked in 19 chunks)."""
    obj = synth.chessboard(9, 6, 25.0)
    crit = (3, 30, 1e-9)
    for n in (1, 3):
        ms = synth.make_views(obj, n, 900 + n)
        K0 = ms.cam.K() * np.array([[1.01, 1, 1], [1, 1.01, 1], [1, 1, 1]])
        flags = pkg.CALIB_USE_INTRINSIC_GUESS | pkg.CALIB_FIX_K3 | pkg.CALIB_ZERO_TANGENT_DIST
        got = pkg.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(5), flags=flags, criteria=crit)
        ref = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(5), flags=flags, criteria=crit, schur=False)
        assert abs(got["rms"] - ref["rms"]) <= 1e-6 * ref["rms"]
        assert np.allclose(got["K"], ref["K"], rtol=1e-6)
        assert abs(got["iters"] - ref["iters"]) <= 1
    big = synth.chessboard(40, 30, 4.0)
    ms = synth.make_views(big, 5, 77)
    got = pkg.calibrate_camera(big, ms.img, (640, 480), criteria=crit)
    ref = orc.calibrate_camera(big, ms.img, (640, 480), criteria=crit, schur=False)
    _close_params(got, ref)
    assert abs(got["iters"] - ref["iters"]) <= 1


def test_reprojection_error_matches_oracle(pkg, orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 50, 3)
    cam = ms.cam
    rms, pv, proj = pkg.reprojection_error(obj, ms.img, cam.K(), cam.dist(5), ms.rvec, ms.tvec, want_points=True)
    want = synth.project(obj, ms.rvec, ms.tvec, cam).reshape(-1, 2)
    assert np.abs(proj - want).max() < 1e-4            # float32 output
    rms_ref, pv_ref = orc.reprojection_error(obj, ms.img, cam.K(), cam.dist(5), ms.rvec, ms.tvec)
    assert abs(rms - rms_ref) < 1e-4 * rms_ref
    assert np.allclose(pv, pv_ref, rtol=1e-3)


def test_full_size_100k_views_properties(pkg, synth):
    """BASELINE config 4 on one GPU, checked through size-independent properties."""
    obj = synth.chessboard(9, 6, 25.0)
    N = 100_000
    ms = synth.make_views(obj, N, 0xC0FFEE + 4)
    intr, mask, aspect, rt = _start(pkg, synth, ms)
    runs = []
    for _ in range(2):
        s = pkg.Solver(obj, ms.img)
        s.set_start(intr, mask, rt)
        st = s.run(criteria=(3, 30, 1e-10))
        out = s.get()
        runs.append((st, out))
        e2 = s.eval_error()
        assert abs(e2 - out["err"]) <= 1e-12 * out["err"]       # returned error == error re-evaluated
        s.close()
    (st, out), (st_b, out_b) = runs
    # fixed-order reductions: two runs are bitwise identical
    assert np.array_equal(out["intr"], out_b["intr"]) and np.array_equal(out["rt"], out_b["rt"])
    assert st["iters"] == st_b["iters"]
    rms = np.sqrt(out["err"] / (N * 54))
    assert 0.40 < rms < 0.44                                    # 0.3 px noise per coordinate
    cam = ms.cam
    assert np.allclose(out["intr"][0, :4], [cam.fx, cam.fy, cam.cx, cam.cy], rtol=2e-4)
    assert np.allclose(out["intr"][0, 4:9], cam.dist(5), atol=1e-2)   # k2, k3 are weakly determined
    # the RMS recomputed by re-projection agrees (monocular_circle.cpp:5-29)
    K = np.array([[out["intr"][0, 0], 0, out["intr"][0, 2]], [0, out["intr"][0, 1], out["intr"][0, 3]], [0, 0, 1]])
    rms2, _ = pkg.reprojection_error(obj, ms.img, K, out["intr"][0, 4:9], out["rt"][:, :3], out["rt"][:, 3:])
    assert abs(rms2 - rms) < 1e-5 * rms


def test_config4_full_size_100k_views_against_schur_oracle(pkg, orc, synth):
    """BASELINE configs[3] at its stated 100 000 views, the whole calibrateCamera call (initialisation included on both
    sides) against the Schur oracle on the host's cores: results to BASELINE's tolerances, the iteration count to +-1,
    and the LM record decision for decision."""
    import os
    from tests.trace_util import assert_same_decisions
    obj = synth.chessboard(9, 6, 25.0)
    N = 100_000
    ms = synth.make_views(obj, N, 0xC0FFEE + 4)
    crit = (3, 30, 1e-9)
    got = pkg.calibrate_camera(obj, ms.img, (640, 480), criteria=crit)
    tr = pkg.last_lm_trace()
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), criteria=crit, schur=True, nthreads=min(16, os.cpu_count() or 1),
                               trace=True)
    _close_params(got, ref)
    assert abs(got["iters"] - ref["iters"]) <= 1
    assert assert_same_decisions(tr, ref["trace"], "configs[3]") >= 5


def test_one_and_three_tile_workgroups_agree_bitwise(pkg, synth, monkeypatch):
    """The fused Jacobian kernel runs with one or three tiles per workgroup depending on what fits a CU; the
    assignment of views to wavefronts and the order of every sum are the same, so the results must be too."""
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 20_003, 31337)
    intr, mask, aspect, rt = _start(pkg, synth, ms)
    outs = []
    for tpb in ("1", "3"):
        monkeypatch.setenv("CCALIB_FUSED_TPB", tpb)
        s = pkg.Solver(obj, ms.img)
        s.set_start(intr, mask, rt)
        st = s.run(criteria=(3, 12, 1e-12))
        outs.append((st["iters"], s.get()))
        s.close()
    (it1, o1), (it3, o3) = outs
    assert it1 == it3
    assert np.array_equal(o1["intr"], o3["intr"]) and np.array_equal(o1["rt"], o3["rt"]) and o1["err"] == o3["err"]


def test_running_ahead_of_the_host_changes_nothing(pkg, synth, monkeypatch):
    """The GPU runs one LM iteration ahead of the host state machine (the device predicts the host's next step);
    the host adopts that work only when its own step is bit for bit the predicted one, so a run with the mechanism
    must equal a run without it bit for bit -- and the mechanism must actually engage."""
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 30_011, 2024)
    intr, mask, aspect, rt = _start(pkg, synth, ms)
    outs = []
    for on in ("0", "1"):
        monkeypatch.setenv("CCALIB_CHAIN", on)
        s = pkg.Solver(obj, ms.img)
        s.set_start(intr, mask, rt)
        st = s.run(criteria=(3, 25, 1e-13))
        outs.append((st, s.get(), s.counters()))
        s.close()
    (st0, o0, c0), (st1, o1, c1) = outs
    assert c0["ahead"] == 0
    assert c1["used"] >= st1["iters"] - 3, (c1, st1)          # all but the first steps and the ones after a rejection
    assert st0["iters"] == st1["iters"] and st0["err_passes"] == st1["err_passes"]
    assert np.array_equal(o0["intr"], o1["intr"]) and np.array_equal(o0["rt"], o1["rt"]) and o0["err"] == o1["err"]


def test_noise_free_recovery_10k_views(pkg, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 10_000, 4242, noise=0.0)
    intr, mask, aspect, rt = _start(pkg, synth, ms)
    s = pkg.Solver(obj, ms.img)
    s.set_start(intr, mask, rt)
    s.run(criteria=(3, 50, 1e-12))
    out = s.get()
    s.close()
    cam = ms.cam
    assert np.sqrt(out["err"] / (10_000 * 54)) < 5e-5           # float32 rounding of the image points
    assert np.allclose(out["intr"][0, :4], [cam.fx, cam.fy, cam.cx, cam.cy], rtol=1e-6)
    assert np.abs(out["rt"][:, :3] - ms.rvec).max() < 1e-4
    assert np.abs(out["rt"][:, 3:] - ms.tvec).max() < 5e-2


def test_bad_arguments_raise(pkg, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 4, 1)
    with pytest.raises(pkg.CcalibError):
        pkg.calibrate_camera(obj, ms.img, (640, 480), ndist=6)
    with pytest.raises(pkg.CcalibError):
        pkg.calibrate_camera(obj, ms.img, (640, 480), flags=pkg.CALIB_TILTED_MODEL, ndist=14)
    with pytest.raises(pkg.CcalibError):
        pkg.calibrate_camera(obj[:3], ms.img[:, :3], (640, 480))
    bent = obj.copy()
    bent[:, 2] = np.linspace(0, 30, len(obj))
    with pytest.raises(pkg.CcalibError):   # non-planar rig without an intrinsic guess
        pkg.calibrate_camera(bent, ms.img, (640, 480))


def _run_shard_worker(mode, port):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port),
                           os.path.join(root, "tests", "gpu_shard_worker.py"), mode],
                          env=env, capture_output=True, text=True, timeout=280)


def test_two_ranks_shard_views_callback_allreduce():
    """N > 1 through the real kernels: two processes (sharing the box's one GPU) own half the views each."""
    out = _run_shard_worker("callback", 29621)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "GPU_SHARDED_OK callback" in out.stdout
    assert "GPU_SHARDED_INIT_OK callback" in out.stdout           # device initialisation summed over ranks
    assert "GPU_SHARDED_STEREO_OK callback" in out.stdout         # stereoCalibrate sharded over pairs


def test_two_ranks_rccl_allreduce_one_gpu_per_rank():
    """The same worker with in-stream RCCL all-reduces over a two-rank communicator, one GPU per rank.  Runs wherever
    two GPUs are visible (the launcher starts before anything here touches a GPU); skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one per rank): RCCL refuses two ranks on one device")
    out = _run_shard_worker("rccl", 29623)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for tag in ("GPU_SHARDED_OK rccl", "GPU_SHARDED_INIT_OK rccl", "GPU_SHARDED_STEREO_OK rccl"):
        assert tag in out.stdout


def test_bench_two_ranks_gloo_rehearsal():
    """bench.py --gpus 2 end to end as the driver launches it (torch.distributed.run, one JSON line from rank 0),
    rehearsed on whatever GPUs are present: with CCALIB_BENCH_BACKEND=gloo the ranks share GPUs and every sum goes
    through the host callback."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CCALIB_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29625", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "1", "--views", "3000", "--no-cpu"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["scaling"] == "weak"
    assert r["config"]["views_per_gpu"] == 3000 and r["config"]["allreduce"] == "torch"
    assert 0.3 < r["rms_px"] < 0.6 and r["value"] > 0


def test_rccl_allreduce_path(pkg, synth):
    """In-stream RCCL all-reduce: a one-rank communicator on this box (multi-rank needs one GPU per rank)."""
    import ctypes as C
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 300, 5)
    intr, mask, aspect, rt = _start(pkg, synth, ms)
    res = []
    for use in (False, True):
        s = pkg.Solver(obj, ms.img)
        if use:
            buf = C.create_string_buffer(128)
            pkg.capi.check(pkg.capi.load().ccalib_rccl_unique_id(buf))
            s.use_rccl(1, 0, buf.raw)
        s.set_start(intr, mask, rt)
        st = s.run(criteria=(3, 30, 1e-10))
        res.append((st, s.get()))
        s.close()
    assert res[0][0]["iters"] == res[1][0]["iters"]
    assert np.array_equal(res[0][1]["intr"], res[1][1]["intr"])


def test_cpp_dropin_header_end_to_end():
    """The C++ header with OpenCV's call shapes (include/ccalib.hpp), driven like the reference's programs:
    calibrateCamera -> reprojection RMS -> initCameraMatrix2D -> stereoCalibrate on noise-free synthetic
    views, so the known cameras must come back."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([os.path.join(root, "tests", "cpp", "dropin_check")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = {ln.split()[0]: [float(x) for x in ln.split()[1:]] for ln in out.stdout.splitlines() if ln.strip()}
    assert r["mono_rms"][0] < 2e-4                       # float32 image points: ~1e-5 px quantisation
    assert np.allclose(r["mono_K"], [812.5, 806.25, 322.0, 238.5], rtol=1e-4)
    assert np.allclose(r["mono_D"], [-0.21, 0.09, 0.0007, -0.0004, -0.015], atol=2e-3)
    # the re-check rounds projections to float32 as cv::projectPoints into vector<Point2f> does, so on
    # noise-free input the two figures only agree to the float32 spacing of a pixel coordinate (~3e-5)
    assert abs(r["reproj_rms"][0] - r["mono_rms"][0]) < 3e-5
    assert r["stereo_rms"][0] < 2e-4
    assert np.allclose(r["stereo_T"], [-120.0, 1.5, 2.5], atol=2e-2)
    assert np.allclose(r["stereo_K2"], [798.0, 801.5, 316.5, 244.0], rtol=1e-4)
    assert r["bad_input_throws"][0] == 1
