"""Golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the dense, OpenCV-shaped
oracle).  The reference ships no test data and OpenCV is not installed, so the files pin the oracle's answers,
not a real OpenCV's ("parity unpinned"): the CPU tests here show the oracle still reproduces them (both of its
solvers), the GPU tests hold the HIP path, through the C ABI, to the same files at BASELINE.json's tolerances
(1e-6 relative on RMS / intrinsics / distortion, 1e-6 on poses, iteration count +-1)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MONO = ["mono_9x6_20views", "mono_8x5_aspect_lu", "mono_circles_30views", "mono_circles_4x11_ref"]
STEREO = ["stereo_9x6_16pairs", "stereo_circles_4x11_ref"]
REL = 1e-6


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def _mono_args(g):
    crit = (int(g["criteria"][0]), int(g["criteria"][1]), float(g["criteria"][2]))
    kw = dict(flags=int(g["flags"]), criteria=crit)
    if "ndist" in g.files:
        kw["ndist"] = int(g["ndist"])
    return tuple(int(x) for x in g["size"]), kw


def _check_mono(got, g, pose_tol):
    assert abs(got["rms"] - float(g["rms"])) <= REL * float(g["rms"])
    assert np.allclose(got["K"], g["K"], rtol=REL, atol=1e-9)
    # distortion: 1e-6 relative, element by element -- unless the minimum itself does not determine the coefficients
    # that well: the file holds the oracle's own dense-vs-Schur spread on this input, and where four times that
    # exceeds 1e-6 (the 88-point circle grid: 1.2e-6, on p2 = 2.5e-4) it is the bound.  The absolute floor of 1e-9
    # (a millionth of a pixel at this focal length) is for coefficients that happen to come out near zero.
    dtol = max(REL, 4 * float(g["dist_spread"]))
    assert dtol <= 5e-6, dtol
    assert np.allclose(got["dist"], g["dist"], rtol=dtol, atol=1e-9)
    assert np.abs(got["rvecs"] - g["rvecs"]).max() < pose_tol
    assert np.abs(got["tvecs"] - g["tvecs"]).max() < pose_tol * max(1.0, np.abs(g["tvecs"]).max())
    # +-1 where the criteria stop the run on a meaningful step size.  With eps = DBL_EPSILON (cv::calibrateCamera's
    # default) the stop lands on round-off -- the last iterations move nothing -- and the count depends on the order
    # of the sums (the oracle's own what-if variants move it by 4 on this input, tests/test_oracle_variants.py).
    eps = float(g["criteria"][2])
    assert abs(got["iters"] - int(g["iters"])) <= (1 if eps > 1e-12 else 6)


@pytest.mark.parametrize("name", MONO)
def test_oracle_reproduces_golden_mono(orc, name):
    g = _load(name)
    size, kw = _mono_args(g)
    dense = orc.calibrate_camera(g["obj"], g["img"], size, schur=False, **kw)
    assert dense["rms"] == float(g["rms"]) and np.array_equal(dense["K"], g["K"])   # same code, same machine arithmetic
    schur = orc.calibrate_camera(g["obj"], g["img"], size, schur=True, **kw)
    _check_mono(schur, g, 1e-7)


@pytest.mark.parametrize("name", STEREO)
def test_oracle_reproduces_golden_stereo(orc, name):
    g = _load(name)
    size = tuple(int(x) for x in g["size"])
    crit = (int(g["criteria"][0]), int(g["criteria"][1]), float(g["criteria"][2]))
    for schur in (False, True):
        r = orc.stereo_calibrate(g["obj"], g["img1"], g["img2"], g["K1_init"], None, g["K2_init"], None, size, int(g["flags"]),
                                 criteria=crit, schur=schur)
        assert abs(r["rms"] - float(g["rms"])) <= 1e-9 * float(g["rms"])
        for key in ("K1", "K2", "R", "T", "E", "F"):
            assert np.allclose(r[key], g[key], rtol=1e-7, atol=1e-9 * np.abs(g[key]).max()), key


@pytest.mark.gpu
@pytest.mark.parametrize("name", MONO)
def test_hip_matches_golden_mono(pkg, name):
    g = _load(name)
    size, kw = _mono_args(g)
    got = pkg.calibrate_camera(g["obj"], g["img"], size, **kw)
    _check_mono(got, g, 1e-6)
    # views handed over one by one (vector<vector<Point>> of the reference) give the same answer as one array
    views = [v for v in g["img"]]
    again = pkg.calibrate_camera([g["obj"]] * len(views), views, size, **kw)
    assert again["rms"] == got["rms"] and np.array_equal(again["K"], got["K"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", STEREO)
def test_hip_matches_golden_stereo(pkg, name):
    g = _load(name)
    size = tuple(int(x) for x in g["size"])
    crit = (int(g["criteria"][0]), int(g["criteria"][1]), float(g["criteria"][2]))
    K1 = pkg.init_camera_matrix_2d(g["obj"], g["img1"], size)
    K2 = pkg.init_camera_matrix_2d(g["obj"], g["img2"], size)
    assert np.allclose(K1, g["K1_init"], rtol=1e-8) and np.allclose(K2, g["K2_init"], rtol=1e-8)
    got = pkg.stereo_calibrate(g["obj"], g["img1"], g["img2"], g["K1_init"], None, g["K2_init"], None, size, int(g["flags"]),
                               criteria=crit)
    assert abs(got["rms"] - float(g["rms"])) <= REL * float(g["rms"])
    # +-1 where the criteria stop the run on a meaningful step size.  With eps = DBL_EPSILON (cv::calibrateCamera's
    # default) the stop lands on round-off -- the last iterations move nothing -- and the count depends on the order
    # of the sums (the oracle's own what-if variants move it by 4 on this input, tests/test_oracle_variants.py).
    eps = float(g["criteria"][2])
    assert abs(got["iters"] - int(g["iters"])) <= (1 if eps > 1e-12 else 6)
    for key in ("K1", "K2", "R", "T", "E", "F"):
        assert np.allclose(got[key], g[key], rtol=REL, atol=1e-8 * np.abs(g[key]).max()), key
    assert np.allclose(got["d1"], g["d1"], rtol=REL, atol=1e-9) and np.allclose(got["d2"], g["d2"], rtol=REL, atol=1e-9)
    assert np.abs(got["rt"] - g["rt"])[:, :3].max() < 1e-6
