"""The CPU oracle checked by means independent of itself.

OpenCV (whose algorithm the oracle restates) is not installed and the reference
holds no fixtures for this path, so parity with OpenCV is UNPINNED.  What these
tests pin instead: analytic Jacobians against finite differences, the dense
(OpenCV-shaped) system against the Schur-eliminated one, the minimum against a
generic least-squares solver, exact recovery on noise-free data, and the
identity the reference itself prints (monocular_circle.cpp:96-103): the RMS
returned by calibrateCamera equals the RMS recomputed by re-projection.
"""
import numpy as np
import pytest

EPS = float(np.finfo(np.float64).eps)


def _fd(f, x, h=1e-6):
    x = np.asarray(x, dtype=np.float64)
    f0 = np.asarray(f(x))
    J = np.zeros((f0.size, x.size))
    for i in range(x.size):
        d = np.zeros_like(x)
        d[i] = h * max(1.0, abs(x[i]))
        J[:, i] = (np.asarray(f(x + d)) - np.asarray(f(x - d))).ravel() / (2 * d[i])
    return J


def test_rodrigues_roundtrip_and_jacobians(orc):
    rng = np.random.default_rng(1)
    for _ in range(20):
        r = rng.normal(size=3) * rng.uniform(0.01, 1.5)
        R, J = orc.rodrigues_v2m(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        Jfd = _fd(lambda v: orc.rodrigues_v2m(v)[0].ravel(), r)   # (9,3) = dR/dr
        assert np.allclose(J.T, Jfd, atol=1e-8)
        r2, Jinv = orc.rodrigues_m2v(R)
        assert np.allclose(r2, r, atol=1e-12)
        # d r / d R restricted to rotations: chain through R(r) must give identity
        assert np.allclose(Jinv @ J.T, np.eye(3), atol=1e-8)
    R0, J0 = orc.rodrigues_v2m(np.zeros(3))
    assert np.array_equal(R0, np.eye(3))
    assert np.allclose(J0.T, _fd(lambda v: orc.rodrigues_v2m(v)[0].ravel(), np.array([1e-3, 0, 0]) * 0 + 1e-9), atol=1e-6)


def test_compose_rt_jacobians(orc):
    rng = np.random.default_rng(2)
    for _ in range(10):
        r1, r2 = rng.normal(size=3) * 0.4, rng.normal(size=3) * 0.05
        t1, t2 = rng.normal(size=3) * 300, rng.normal(size=3) * 60
        r3, t3, d = orc.compose_rt(r1, t1, r2, t2)
        R1, R2, R3 = (orc.rodrigues_v2m(v)[0] for v in (r1, r2, r3))
        assert np.allclose(R3, R2 @ R1, atol=1e-13)
        assert np.allclose(t3, R2 @ t1 + t2, atol=1e-10)
        assert np.allclose(d["dr3dr1"], _fd(lambda v: orc.compose_rt(v, t1, r2, t2)[0], r1), atol=1e-7)
        assert np.allclose(d["dr3dr2"], _fd(lambda v: orc.compose_rt(r1, t1, v, t2)[0], r2), atol=1e-7)
        assert np.allclose(d["dt3dr2"], _fd(lambda v: orc.compose_rt(r1, t1, v, t2)[1], r2), atol=1e-5)
        assert np.allclose(d["dt3dt1"], R2, atol=1e-14)
        assert np.array_equal(d["dt3dt2"], np.eye(3))


@pytest.mark.parametrize("aspect", [0.0, 1.03])
def test_project_points_jacobians(orc, aspect):
    rng = np.random.default_rng(3)
    obj = rng.uniform(-100, 100, size=(7, 3))
    obj[:, 2] *= 0.2
    r = np.array([0.2, -0.3, 0.1])
    t = np.array([10.0, -20.0, 600.0])
    fc = np.array([800.0, 790.0, 320.0, 240.0])
    k = np.array([-0.28, 0.09, 5e-4, -3e-4, -0.01, 0.02, -0.01, 0.005, 1e-3, -2e-3, 5e-4, 1e-3, 0, 0])
    m, Je, Ji = orc.project_points(obj, r, t, fc, k, aspect)

    def proj(r_=r, t_=t, fc_=fc, k_=k):
        return orc.project_points(obj, r_, t_, fc_, k_, aspect, jac=False)[0].ravel()

    assert np.allclose(Je[:, :3], _fd(lambda v: proj(r_=v), r), rtol=1e-6, atol=1e-5)
    assert np.allclose(Je[:, 3:], _fd(lambda v: proj(t_=v), t), rtol=1e-6, atol=1e-6)
    Jk = _fd(lambda v: proj(k_=np.concatenate([v, [0, 0]])), k[:12])
    assert np.allclose(Ji[:, 4:16], Jk, rtol=1e-6, atol=1e-4)
    Jf = _fd(lambda v: proj(fc_=v), fc)
    assert np.allclose(Ji[:, 2:4], Jf[:, 2:4], atol=1e-6)
    if aspect == 0.0:
        assert np.allclose(Ji[:, 0:2], Jf[:, 0:2], atol=1e-6)
    else:
        # fx is tied to fy: d/dfy picks up aspect * d/dfx, and the fx column is zero
        fx_eff = fc.copy()
        fx_eff[0] = fc[1] * aspect
        Jt = _fd(lambda v: proj(fc_=np.array([v[0] * aspect, v[0], fc[2], fc[3]])), np.array([fc[1]]))
        assert np.allclose(Ji[:, 1], Jt[:, 0], atol=1e-6)
        assert np.all(Ji[:, 0] == 0)


def test_homography_and_pose_init(orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    cam = synth.Camera(k1=0, k2=0, p1=0, p2=0, k3=0)
    ms = synth.make_views(obj, 5, 11, cam=cam, noise=0.0)
    for i in range(5):
        H = orc.find_homography(obj[:, :2], ms.img[i])
        p = (H @ np.c_[obj[:, :2], np.ones(len(obj))].T).T
        p = p[:, :2] / p[:, 2:]
        assert np.abs(p - ms.img[i]).max() < 1e-3
        r, t = orc.find_extrinsic(obj, ms.img[i], cam.K(), np.zeros(14))
        assert np.abs(r - ms.rvec[i]).max() < 1e-5
        assert np.abs(t - ms.tvec[i]).max() < 1e-2
    K0 = orc.init_camera_matrix_2d(obj, ms.img, (640, 480))
    assert abs(K0[0, 0] - 800) < 8 and abs(K0[1, 1] - 800) < 8
    assert K0[0, 2] == 319.5 and K0[1, 2] == 239.5


def test_noise_free_recovery(orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 20, 0xC0FFEE + 1, noise=0.0)
    r = orc.calibrate_camera(obj, ms.img, (640, 480), schur=False)
    assert r["rms"] < 5e-5          # float32 rounding of the image points is ~3e-5 px
    cam = ms.cam
    assert np.allclose([r["K"][0, 0], r["K"][1, 1], r["K"][0, 2], r["K"][1, 2]],
                       [cam.fx, cam.fy, cam.cx, cam.cy], rtol=2e-6)
    assert np.allclose(r["dist"], cam.dist(5), atol=2e-5)
    assert np.abs(r["rvecs"] - ms.rvec).max() < 1e-5
    assert np.abs(r["tvecs"] - ms.tvec).max() < 1e-2


@pytest.mark.parametrize("flags_name", ["zero", "aspect_lu"])
def test_dense_equals_schur_mono(orc, synth, flags_name):
    flags = 0 if flags_name == "zero" else (orc.FIX_ASPECT_RATIO | orc.USE_LU)
    obj = synth.chessboard(9, 6, 25.0) if flags_name == "zero" else synth.chessboard(8, 5, 20.0)
    ms = synth.make_views(obj, 20, 0xC0FFEE + 1, noise=0.3)
    a = orc.calibrate_camera(obj, ms.img, (640, 480), flags=flags, schur=False)
    b = orc.calibrate_camera(obj, ms.img, (640, 480), flags=flags, schur=True)
    assert abs(a["rms"] - b["rms"]) < 1e-10 * a["rms"]
    assert np.allclose(a["K"], b["K"], rtol=1e-8)
    assert np.allclose(a["dist"], b["dist"], rtol=1e-7, atol=1e-10)
    assert np.abs(a["rvecs"] - b["rvecs"]).max() < 1e-8
    assert np.abs(a["tvecs"] - b["tvecs"]).max() < 1e-6
    assert abs(a["iters"] - b["iters"]) <= 2  # eps = DBL_EPSILON: the stop lands on round-off
    # the identity the reference prints: returned RMS == RMS recomputed by projectPoints
    rms2, per_view = orc.reprojection_error(obj, ms.img, a["K"], a["dist"], a["rvecs"], a["tvecs"])
    assert abs(rms2 - a["rms"]) < 1e-12
    assert per_view.shape == (20,)


def test_solvers_agree(orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 8, 5, noise=0.3)
    res = [orc.calibrate_camera(obj, ms.img, (640, 480), schur=False, solve=s,
                                criteria=(3, 30, 1e-9)) for s in (orc.SOLVE_LU, orc.SOLVE_SVD, orc.SOLVE_CHOLESKY)]
    for r in res[1:]:
        assert abs(r["rms"] - res[0]["rms"]) < 1e-10
        assert np.allclose(r["K"], res[0]["K"], rtol=1e-7)
        assert r["iters"] == res[0]["iters"]


def test_same_minimum_as_generic_least_squares(orc, synth):
    scipy_opt = pytest.importorskip("scipy.optimize")
    obj = synth.chessboard(9, 6, 25.0)
    N = 6
    ms = synth.make_views(obj, N, 21, noise=0.3)
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), schur=False)
    img = ms.img.astype(np.float64)

    def resid(p):
        cam = synth.Camera(fx=p[0], fy=p[1], cx=p[2], cy=p[3], k1=p[4], k2=p[5], p1=p[6], p2=p[7], k3=p[8])
        rt = p[9:].reshape(N, 6)
        return (synth.project(obj, rt[:, :3], rt[:, 3:], cam) - img).ravel()

    x0 = np.concatenate([[780, 780, 319.5, 239.5, 0, 0, 0, 0, 0],
                         np.c_[ms.rvec, ms.tvec].ravel()])
    sol = scipy_opt.least_squares(resid, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=4000)
    rms = np.sqrt(np.sum(sol.fun ** 2) / (N * len(obj)))
    assert abs(rms - ref["rms"]) < 1e-8
    assert np.allclose(sol.x[:4], [ref["K"][0, 0], ref["K"][1, 1], ref["K"][0, 2], ref["K"][1, 2]], rtol=1e-5)
    assert np.allclose(sol.x[4:9], ref["dist"], rtol=1e-3, atol=1e-5)


def test_dense_equals_schur_stereo(orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 10, 0xC0FFEE + 3, noise=0.3)
    K1 = orc.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = orc.init_camera_matrix_2d(obj, st.img2, (640, 480))
    # the reference's flag set (binocular_chessboard.cpp:156-161) and the all-free model
    ref_flags = (orc.FIX_ASPECT_RATIO | orc.ZERO_TANGENT_DIST | orc.USE_INTRINSIC_GUESS | orc.SAME_FOCAL_LENGTH |
                 orc.RATIONAL_MODEL | orc.FIX_K3 | orc.FIX_K4 | orc.FIX_K5)
    for flags, crit in ((ref_flags, (3, 30, 1e-5)), (orc.USE_INTRINSIC_GUESS, (3, 100, 1e-6))):
        a = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit, schur=False)
        b = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit, schur=True)
        assert abs(a["rms"] - b["rms"]) < 1e-9 * a["rms"]
        assert a["iters"] == b["iters"]
        for key in ("K1", "K2", "R", "T", "E", "F"):
            assert np.allclose(a[key], b[key], rtol=1e-7, atol=1e-9), key
        assert np.allclose(a["d1"], b["d1"], rtol=1e-6, atol=1e-9)
    # the all-free model recovers the rig
    assert np.abs(b["T"] - st.t_lr).max() < 1.0
    assert np.abs(synth.rotmat_to_rvec(b["R"])[0] - st.r_lr).max() < 5e-3
    # E = [T]x R and F ~ K2^-T E K1^-1 scaled so F[2,2] = 1
    T = b["T"]
    Tx = np.array([[0, -T[2], T[1]], [T[2], 0, -T[0]], [-T[1], T[0], 0]])
    assert np.allclose(b["E"], Tx @ b["R"], atol=1e-9)
    F = np.linalg.inv(b["K2"]).T @ b["E"] @ np.linalg.inv(b["K1"])
    assert np.allclose(b["F"], F / F[2, 2], rtol=1e-8, atol=1e-12)


def test_schur_scales_and_threads_agree(orc, synth):
    obj = synth.asymmetric_circles(8, 11, 12.5)
    ms = synth.make_views(obj, 300, 0xC0FFEE + 2, noise=0.3)
    a = orc.calibrate_camera(obj, ms.img, (640, 480), schur=True, nthreads=1)
    b = orc.calibrate_camera(obj, ms.img, (640, 480), schur=True, nthreads=4)
    assert abs(a["rms"] - b["rms"]) < 1e-10
    assert np.allclose(a["K"], b["K"], rtol=1e-8)
    assert 0.25 < a["rms"] < 0.45   # 0.3 px noise per coordinate -> ~0.42 px RMS per point


def test_iteration_count_at_machine_epsilon_is_not_reproducible(orc, synth):
    """With eps = DBL_EPSILON (cv::calibrateCamera's default criteria) a run ends when a trial at the noise floor is
    rejected all the way up the damping ladder, i.e. on comparisons of residual sums that differ in their last bits.
    The oracle's own variants -- dense or Schur elimination, one or several threads: the same algorithm, different
    summation orders -- already disagree on the iteration count by several iterations while agreeing on the result
    to round-off.  This is the yardstick for the +-6 the GPU tests allow at such criteria (and +-1 elsewhere).
    (On the 64-view input of the smoke check the same three give 25, 19 and 18 iterations; that takes two minutes.)"""
    obj = synth.chessboard(9, 6, 25.0)
    variants = ((True, 1), (False, 1), (True, 3))   # (Schur elimination?, threads)
    spreads = []
    for nviews, seed in ((16, 7), (20, 11)):
        ms = synth.make_views(obj, nviews, seed)
        tight = (3, 30, float(np.finfo(np.float64).eps))
        runs = [orc.calibrate_camera(obj, ms.img, (ms.width, ms.height), flags=0, criteria=tight, schur=sc, nthreads=nt)
                for sc, nt in variants]
        iters = [r["iters"] for r in runs]
        print(f"[oracle] {nviews} views: iterations at eps = DBL_EPSILON {iters}")
        spreads.append(max(iters) - min(iters))
        for r in runs[1:]:
            assert abs(r["rms"] - runs[0]["rms"]) <= 1e-12 * runs[0]["rms"]
            assert np.allclose(r["K"], runs[0]["K"], rtol=1e-8)
        # with a criterion that means something the same three agree to +-1
        loose = (3, 30, 1e-9)
        iters = [orc.calibrate_camera(obj, ms.img, (ms.width, ms.height), flags=0, criteria=loose, schur=sc, nthreads=nt)["iters"]
                 for sc, nt in variants]
        assert max(iters) - min(iters) <= 1, iters
    assert max(spreads) >= 3, spreads
