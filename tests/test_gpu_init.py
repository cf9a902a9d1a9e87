"""GPU tests of the starting values computed on the device (ccalib_solver_init_*), through the C ABI,
against the CPU oracle's restatement of cvInitIntrinsicParams2D, cvFindExtrinsicCameraParams2 and the
relative-pose median of cvStereoCalibrateImpl (reached inside the calls at monocular_circle.cpp:93-94
and binocular_chessboard.cpp:147-148, :152-162).

The device solves its small symmetric systems by Cholesky factorisation and finds the DLT null vector
by inverse iteration where OpenCV (and the oracle) use SVD / Jacobi: the results are the same minimisers,
equal to round-off of those solvers, not bit for bit.  Tolerances: 1e-8 relative on the intrinsic seed,
1e-6 rad / 1e-6 relative on the poses (the refinement stops at a FLT_EPSILON step, so two runs of it only
agree to about that times the conditioning of a view).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_poses(orc, objs, imgs, K, k14):
    out = []
    for o, m in zip(objs, imgs):
        r, t = orc.find_extrinsic(np.asarray(o, dtype=np.float32), np.asarray(m, dtype=np.float32), K, k14)
        out.append(np.r_[r, t])
    return np.array(out)


def _assert_poses(got, ref):
    assert np.abs(got[:, :3] - ref[:, :3]).max() < 1e-6
    assert np.abs(got[:, 3:] - ref[:, 3:]).max() < 1e-6 * np.abs(ref[:, 3:]).max()


def test_init_camera_matrix_matches_oracle(pkg, orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 12, 9)
    K = pkg.init_camera_matrix_2d(obj, ms.img, (640, 480))
    Kref = orc.init_camera_matrix_2d(obj, ms.img, (640, 480))
    assert np.allclose(K, Kref, rtol=1e-8)
    assert K[0, 2] == 319.5 and K[1, 2] == 239.5 and K[0, 1] == 0 and K[2, 2] == 1
    K2 = pkg.init_camera_matrix_2d(obj, ms.img, (640, 480), aspect=1.0)
    assert K2[0, 0] == K2[1, 1]
    assert np.allclose(K2, orc.init_camera_matrix_2d(obj, ms.img, (640, 480), aspect=1.0), rtol=1e-8)
    with pytest.raises(pkg.CcalibError):
        pkg.init_camera_matrix_2d(obj[:3], ms.img[:, :3], (640, 480))
    # circle grid (BASELINE config 2's board) at a size that spans several workgroups, solver-level entry
    obj2 = synth.asymmetric_circles(8, 11, 12.5)
    ms2 = synth.make_views(obj2, 300, 0xC0FFEE + 2)
    s = pkg.Solver(obj2, ms2.img)
    Ka = s.init_intrinsics((640, 480))
    s.close()
    assert np.allclose(Ka, orc.init_camera_matrix_2d(obj2, ms2.img, (640, 480)), rtol=1e-8)


@pytest.mark.parametrize("model", ["plain5", "rational8", "no_distortion"])
def test_init_poses_shared_board_match_oracle(pkg, orc, synth, model):
    obj = synth.chessboard(9, 6, 25.0)
    N = 200                                   # four workgroups of 64 lanes, the last one part-filled
    ms = synth.make_views(obj, N, 0xC0FFEE + 11)
    cam = ms.cam
    K = np.array([[cam.fx * 0.98, 0, cam.cx + 2], [0, cam.fy * 1.01, cam.cy - 1], [0, 0, 1.0]])
    k14 = np.zeros(14)
    if model == "plain5":
        k14[:5] = cam.dist(5) * 0.9
    elif model == "rational8":
        k14[:5] = cam.dist(5) * 0.9
        k14[5:8] = [0.02, -0.01, 0.005]
    s = pkg.Solver(obj, ms.img)
    got = s.init_poses(K, k14)
    s.close()
    ref = _oracle_poses(orc, [obj] * N, ms.img, K, k14)
    _assert_poses(got, ref)
    # and they are what they should be: near the poses the views were rendered from (K is 1-2 % off)
    if model == "plain5":
        assert np.abs(got[:, :3] - ms.rvec).max() < 0.3


def test_init_poses_ragged_views_and_tilted_boards(pkg, orc, synth):
    """Per-view boards: different point subsets per view, and boards that do not lie in Z = 0, so the
    principal-axes frame of cvFindExtrinsicCameraParams2 is a real rotation."""
    rng = np.random.default_rng(12)
    base = synth.chessboard(9, 6, 25.0)
    N = 70
    ms = synth.make_views(base, N, 21)
    cam = ms.cam
    tilt = orc.rodrigues_v2m(np.array([0.3, -0.2, 0.1]))[0]
    objs, imgs, rts = [], [], []
    for i in range(N):
        keep = np.sort(rng.choice(54, size=int(rng.integers(8, 55)), replace=False))
        o = base[keep]
        if i % 2:                             # the same physical points expressed in a tilted, shifted frame
            o = (o - [10.0, 5.0, 0.0]) @ tilt.T + [3.0, -2.0, 40.0]
        objs.append(o.astype(np.float32))
        imgs.append(ms.img[i][keep])
    K = cam.K()
    k14 = np.zeros(14)
    k14[:5] = cam.dist(5)
    s = pkg.Solver(objs, imgs, shared_board=False)
    got = s.init_poses(K, k14)
    s.close()
    ref = _oracle_poses(orc, objs, imgs, K, k14)
    _assert_poses(got, ref)


def test_init_minimum_four_points_per_view(pkg, orc, synth):
    base = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(base, 5, 3, noise=0.0)
    keep = [0, 8, 45, 53]                     # the four corners: DLT only, no homography polish
    objs = [base[keep]] * 5
    imgs = [m[keep] for m in ms.img]
    cam = ms.cam
    k14 = np.zeros(14)
    s = pkg.Solver(objs, imgs, shared_board=False)
    got = s.init_poses(cam.K(), k14)
    s.close()
    ref = _oracle_poses(orc, objs, imgs, cam.K(), k14)
    _assert_poses(got, ref)


def test_stereo_relative_pose_median(pkg, orc, synth):
    obj = synth.chessboard(9, 6, 25.0)
    N = 151
    st = synth.make_stereo(obj, N, 0xC0FFEE + 3)
    K1 = orc.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = orc.init_camera_matrix_2d(obj, st.img2, (640, 480))
    k14 = np.zeros(14)
    s = pkg.Solver(obj, st.img1, st.img2)
    with pytest.raises(pkg.CcalibError):
        s.init_relative_pose()                # poses first
    p0 = s.init_poses(K1, k14, cam=0)
    p1 = s.init_poses(K2, k14, cam=1)
    lr = s.init_relative_pose()
    s.close()
    r0 = _oracle_poses(orc, [obj] * N, st.img1, K1, k14)
    r1 = _oracle_poses(orc, [obj] * N, st.img2, K2, k14)
    _assert_poses(p0, r0)
    _assert_poses(p1, r1)
    rel = np.zeros((N, 6))
    for i in range(N):
        R0, R1 = orc.rodrigues_v2m(r0[i, :3])[0], orc.rodrigues_v2m(r1[i, :3])[0]
        Rr = R1 @ R0.T
        rel[i, :3] = orc.rodrigues_m2v(Rr)[0]
        rel[i, 3:] = r1[i, 3:] - Rr @ r0[i, 3:]
    want = np.median(rel, axis=0)
    assert np.abs(lr[:3] - want[:3]).max() < 1e-6
    assert np.abs(lr[3:] - want[3:]).max() < 1e-6 * np.abs(want[3:]).max()


def test_non_planar_rig_is_refused(pkg, synth):
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 4, 1)
    bent = obj.copy()
    bent[:, 2] = (np.arange(len(obj)) % 7) * 9.0
    s = pkg.Solver(bent, ms.img)
    with pytest.raises(pkg.CcalibError) as ei:
        s.init_poses(ms.cam.K(), np.zeros(14))
    s.close()
    assert ei.value.code == pkg.capi.ERR_UNSUPPORTED
    with pytest.raises(pkg.CcalibError):      # and through the whole call, with an intrinsic guess
        pkg.calibrate_camera(bent, ms.img, (640, 480), K=ms.cam.K(), flags=pkg.CALIB_USE_INTRINSIC_GUESS)


def test_device_resident_start_equals_uploaded_start(pkg, synth):
    """set_start(rt=None) after init_poses steps from exactly the poses init_poses returned."""
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 500, 77)
    cam = ms.cam
    intr, mask, _ = pkg.mono_start(cam.K(), cam.dist(5), 0)
    outs = []
    for resident in (True, False):
        s = pkg.Solver(obj, ms.img)
        rt = s.init_poses(cam.K(), np.r_[cam.dist(5), np.zeros(9)])
        s.set_start(intr, mask, None if resident else rt)
        s.run(criteria=(3, 10, 0.0))
        outs.append(s.get())
        s.close()
    assert np.array_equal(outs[0]["rt"], outs[1]["rt"]) and np.array_equal(outs[0]["intr"], outs[1]["intr"])
    s = pkg.Solver(obj, ms.img)
    with pytest.raises(pkg.CcalibError):
        s.set_start(intr, mask, None)         # nothing on the device yet
    s.close()


def test_full_size_100k_views_whole_call(pkg, synth):
    """BASELINE config 4 through the whole call from host buffers: initialisation on the device, LM loop,
    download.  Checked through properties: the known camera comes back and the returned RMS equals the RMS
    recomputed by re-projection (monocular_circle.cpp:5-29)."""
    obj = synth.chessboard(9, 6, 25.0)
    N = 100_000
    ms = synth.make_views(obj, N, 0xC0FFEE + 4)
    got = pkg.calibrate_camera(obj, ms.img, (640, 480))
    cam = ms.cam
    assert 0.40 < got["rms"] < 0.44
    assert np.allclose([got["K"][0, 0], got["K"][1, 1], got["K"][0, 2], got["K"][1, 2]],
                       [cam.fx, cam.fy, cam.cx, cam.cy], rtol=2e-4)
    dev = np.abs(got["rvecs"] - ms.rvec).max(axis=1)
    assert np.median(dev) < 0.01 and np.mean(dev > 0.05) < 1e-3   # a planar view now and then keeps its mirror pose
    rms2, _ = pkg.reprojection_error(obj, ms.img, got["K"], got["dist"], got["rvecs"], got["tvecs"])
    assert abs(rms2 - got["rms"]) < 1e-5 * got["rms"]


@pytest.mark.gpu
def test_parking_point_of_the_refinement_changes_nothing(pkg, synth, monkeypatch):
    """The pose refinement leaves views that are still running after `cut` iterations to a second launch.  Both
    launches give a view four lanes and the same arithmetic, so where a view is parked must not show in its pose."""
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 5000, 0xC0FFEE + 12)
    cam = ms.cam
    K = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 1.02, cam.cy - 2], [0, 0, 1.0]])
    k14 = np.r_[cam.dist(5) * 0.8, np.zeros(9)]
    poses = {}
    for cut in ("1", "6", "20"):
        monkeypatch.setenv("CCALIB_INIT_CUT", cut)
        s = pkg.Solver(obj, ms.img)
        poses[cut] = s.init_poses(K, k14)
        s.close()
    assert np.array_equal(poses["1"], poses["6"]) and np.array_equal(poses["6"], poses["20"])
