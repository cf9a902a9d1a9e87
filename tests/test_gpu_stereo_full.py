"""stereoCalibrate at the sizes BASELINE.json names.

configs[2] (10 000 pairs of a 9x6 board, the flag set and criteria of binocular_chessboard.cpp:152-162) runs whole
against the Schur oracle; one GPU's share of configs[4] (125 000 pairs of a 17x12 board, k1-k3, p1, p2 free in both
cameras) is beyond the oracle's reach in test time and is checked through size-independent properties.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-6


def test_config3_stereo_10k_pairs_against_schur_oracle(pkg, orc, synth):
    """BASELINE configs[2] at its stated size, initialisation included on both sides."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 10_000, 0xC0FFEE + 3)
    K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
    flags = (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_USE_INTRINSIC_GUESS |
             pkg.CALIB_SAME_FOCAL_LENGTH | pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 |
             pkg.CALIB_FIX_K5)
    crit = (3, 100, 1e-5)
    got = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit)
    ref = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit, schur=True,
                               nthreads=min(16, os.cpu_count() or 1))
    assert abs(got["rms"] - ref["rms"]) <= REL * ref["rms"]
    assert abs(got["iters"] - ref["iters"]) <= 1
    for key in ("K1", "K2", "R", "T", "E", "F"):
        assert np.allclose(got[key], ref[key], rtol=REL, atol=1e-8 * np.abs(ref[key]).max()), key
    assert np.allclose(got["d1"], ref["d1"], rtol=REL, atol=1e-9)
    assert np.allclose(got["d2"], ref["d2"], rtol=REL, atol=1e-9)
    assert np.abs(got["rt"] - ref["rt"])[:, :3].max() < 1e-6
    assert np.abs(got["rt"] - ref["rt"])[:, 3:].max() < 1e-6 * np.abs(ref["rt"][:, 3:]).max()


def test_config5_one_gpu_share_125k_pairs_properties(pkg, synth):
    """One GPU's share of BASELINE configs[4]: 125 000 pairs, 17x12 board (204 points), both cameras with
    fx fy cx cy k1 k2 p1 p2 k3 free.  Properties: two runs agree bit for bit, the returned error is the error
    re-evaluated at the returned parameters, and the rig and both cameras come back."""
    obj = synth.chessboard(17, 12, 10.0)
    N, chunk = 125_000, 25_000
    parts = [synth.make_stereo(obj, chunk, 0xC0FFEE + 5, first_view=k) for k in range(0, N, chunk)]
    img1 = np.concatenate([p.img1 for p in parts])
    img2 = np.concatenate([p.img2 for p in parts])
    rvec = np.concatenate([p.rvec for p in parts])
    tvec = np.concatenate([p.tvec for p in parts])
    st = parts[0]
    del parts
    intr = np.zeros((2, 18))
    for c, cam in enumerate((st.cam1, st.cam2)):
        intr[c, :4] = [cam.fx * 0.98, cam.fy * 0.985, cam.cx + 2.5, cam.cy - 2.0]
        intr[c, 4:9] = cam.dist(5) * 0.8
    mask = np.zeros((2, 18), dtype=np.uint8)
    mask[:, :9] = 1
    rt0 = synth.perturbed_poses(rvec, tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    lr0 = np.r_[st.r_lr, st.t_lr] + np.array([1e-3, -1e-3, 5e-4, 0.5, -0.2, 0.3])
    runs = []
    for _ in range(2):
        s = pkg.Solver(obj, img1, img2)
        s.set_start(intr, mask, rt0, lr=lr0)
        stt = s.run(criteria=(3, 40, 1e-9))
        out = s.get()
        e2 = s.eval_error()
        assert abs(e2 - out["err"]) <= 1e-12 * out["err"]
        s.close()
        runs.append((stt, out))
    (sa, oa), (sb, ob) = runs
    assert sa["iters"] == sb["iters"] and oa["err"] == ob["err"]
    assert np.array_equal(oa["intr"], ob["intr"]) and np.array_equal(oa["lr"], ob["lr"]) and np.array_equal(oa["rt"], ob["rt"])
    rms = np.sqrt(oa["err"] / (2.0 * N * 204))
    assert 0.41 < rms < 0.44                                  # 0.3 px noise per coordinate -> 0.424 px
    r_true, t_true = np.array(st.r_lr), np.array(st.t_lr)
    assert np.abs(oa["lr"][:3] - r_true).max() < 1e-4       # trades against the principal points at this level
    assert np.abs(oa["lr"][3:] - t_true).max() < 5e-2
    for c, cam in enumerate((st.cam1, st.cam2)):
        assert np.allclose(oa["intr"][c, :4], [cam.fx, cam.fy, cam.cx, cam.cy], rtol=3e-4)
        assert np.allclose(oa["intr"][c, 4:9], cam.dist(5), atol=5e-3)
    assert np.abs(oa["rt"][:, :3] - rvec).max() < 6e-2      # a single noisy view determines its pose this well
    assert np.median(np.abs(oa["rt"][:, :3] - rvec)) < 1e-3


def test_a_view_with_unusable_blocks_counts_for_nothing_in_both_sum_kernels(pkg, synth, monkeypatch):
    """One pair out of 33 000 starts from a pose that is not a number: its blocks hold NaN, its 6 x 6 factorisation
    fails, and it must drop out of the Schur sums -- 0 * NaN must not reach them -- in the one-lane-per-view kernel
    (which large problems run) exactly as in the four-lanes-per-view kernel."""
    obj = synth.chessboard(9, 6, 25.0)
    N = 33_000
    st = synth.make_stereo(obj, N, 0xC0FFEE + 3)
    img2 = st.img2
    intr = np.zeros((2, 18))
    for c, cam in enumerate((st.cam1, st.cam2)):
        intr[c, :4] = [cam.fx * 0.98, cam.fy * 0.985, cam.cx + 2.5, cam.cy - 2.0]
        intr[c, 4:9] = cam.dist(5) * 0.8
    mask = np.zeros((2, 18), dtype=np.uint8)
    mask[:, :9] = 1
    rt0 = synth.perturbed_poses(st.rvec, st.tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    lr0 = np.r_[st.r_lr, st.t_lr] + np.array([1e-3, -1e-3, 5e-4, 0.5, -0.2, 0.3])
    rt0[12_345, 4] = np.nan
    sums = {}
    for quad in ("0", "1"):
        monkeypatch.setenv("CCALIB_SUMS_QUAD", quad)
        s = pkg.Solver(obj, st.img1, img2)
        s.set_start(intr, mask, rt0, lr=lr0)
        s.eval_jacobian()
        sums[quad] = s.reduce(1e-3, 24)
        s.close()
    (S0, b0), (S1, b1) = sums["0"], sums["1"]
    assert np.isfinite(S0).all() and np.isfinite(b0).all() and np.isfinite(S1).all() and np.isfinite(b1).all()
    assert np.allclose(S0, S1, rtol=1e-9, atol=1e-10 * np.abs(S1).max()) and np.allclose(b0, b1, rtol=1e-9, atol=1e-10 * np.abs(b1).max())
    # and they are the sums over the other 32 999 pairs: the same problem without that pair gives them to round-off
    keep = np.arange(N) != 12_345
    monkeypatch.setenv("CCALIB_SUMS_QUAD", "0")
    s = pkg.Solver(obj, st.img1[keep], img2[keep])
    s.set_start(intr, mask, rt0[keep], lr=lr0)
    s.eval_jacobian()
    S2, b2 = s.reduce(1e-3, 24)
    s.close()
    assert np.allclose(S0, S2, rtol=1e-9, atol=1e-10 * np.abs(S2).max()) and np.allclose(b0, b2, rtol=1e-9, atol=1e-10 * np.abs(b2).max())


def test_config5_shape_10k_pairs_against_schur_oracle(pkg, orc, synth):
    """The shape of BASELINE configs[4] -- 17x12 board, 204 points per view (four chunks of 56 in the fused kernel), nine
    free intrinsics per camera -- at a size the Schur oracle finishes in seconds: whole stereoCalibrate calls,
    initialisation included on both sides."""
    obj = synth.chessboard(17, 12, 10.0)
    st = synth.make_stereo(obj, 10_000, 0xC0FFEE + 6)
    K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
    flags = pkg.CALIB_USE_INTRINSIC_GUESS
    crit = (3, 40, 1e-8)
    got = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit)
    ref = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=crit, schur=True,
                               nthreads=min(16, os.cpu_count() or 1))
    assert abs(got["rms"] - ref["rms"]) <= REL * ref["rms"]
    assert abs(got["iters"] - ref["iters"]) <= 1
    for key in ("K1", "K2", "R", "T", "E", "F"):
        assert np.allclose(got[key], ref[key], rtol=REL, atol=1e-8 * np.abs(ref[key]).max()), key
    assert np.allclose(got["d1"], ref["d1"], rtol=REL, atol=1e-9)
    assert np.allclose(got["d2"], ref["d2"], rtol=REL, atol=1e-9)
    assert np.abs(got["rt"] - ref["rt"])[:, :3].max() < 1e-6
    assert np.abs(got["rt"] - ref["rt"])[:, 3:].max() < 1e-6 * np.abs(ref["rt"][:, 3:]).max()
