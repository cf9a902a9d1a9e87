#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle's dense, OpenCV-shaped solver.

    python tests/golden/make_golden.py

The reference (rxjf/Camera-Calibration) ships no test data and its arithmetic lives in OpenCV 4.1.0, which is
not installed here, so these files are NOT outputs of the reference: they freeze the oracle's answers on fixed
inputs (inputs stored as the float32 arrays a caller would pass, outputs as float64), so that a change in either
the oracle or the HIP path shows up as a diff against a committed file.  Cases:
  mono_9x6_20views      BASELINE config 1: calibrateCamera(flags = 0), default criteria
  mono_8x5_aspect_lu    the call of monocular_chessboard.cpp:315-317 (8x5 board at 20 mm, FIX_ASPECT_RATIO | USE_LU)
  mono_circles_30views  BASELINE config 2's board (4x11 asymmetric grid) at 30 views, dist = zeros(8,1)-style call
  stereo_9x6_16pairs    the flag set of binocular_chessboard.cpp:156-161, criteria (100, 1e-5)
  mono_circles_4x11_ref    the literal call of monocular_circle.cpp:83-94: 4x11 asymmetric grid (44 points) at spacing 2,
                           cameraMatrix = eye, distCoeffs = zeros(8,1), flags = 0, default criteria
  stereo_circles_4x11_ref  the literal call of binocular_circle.cpp:139-163: the same grid at spacing 20, initCameraMatrix2D
                           per camera, the reference's stereo flag set, criteria (100, 1e-5)
The monocular files hold `dist_spread`: the largest element-wise relative difference of the distortion vector between
the oracle's dense and Schur solvers on that input (the yardstick for how tightly the minimum determines it).
Every file also holds `trace`: the oracle's per-iteration LM record (iteration, damping exponent first / accepted, trial
steps, residual sum before / after), which the GPU tests compare the HIP path's own record with.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def dist_spread(orc, r, obj, img, size, **kw):
    """Largest element-wise relative difference of the distortion vector between the oracle's dense solver (result r)
    and its Schur solver on one and on eight threads: how well the minimum determines these coefficients at all."""
    worst = 0.0
    for nth in (1, 8):
        s = orc.calibrate_camera(obj, img, size, schur=True, nthreads=nth, **kw)
        nz = r["dist"] != 0
        worst = max(worst, float((np.abs(s["dist"] - r["dist"])[nz] / np.abs(r["dist"][nz])).max()))
    return worst


def main():
    pkg, orc = ge.load_package(), ge.load_oracle()
    synth = pkg.synth
    size = (640, 480)

    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 20, 0xC0FFEE + 1)
    r = orc.calibrate_camera(obj, ms.img, size, schur=False, trace=True)
    np.savez_compressed(os.path.join(HERE, "mono_9x6_20views.npz"), obj=obj.astype(np.float32), img=ms.img, size=size,
                        flags=0, criteria=(3, 30, np.finfo(np.float64).eps), rms=r["rms"], K=r["K"], dist=r["dist"],
                        rvecs=r["rvecs"], tvecs=r["tvecs"], iters=r["iters"], trace=r["trace"],
                        dist_spread=dist_spread(orc, r, obj, ms.img, size))

    obj = synth.chessboard(8, 5, 20.0)
    ms = synth.make_views(obj, 25, 31, depth=(300.0, 900.0))
    flags = pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_USE_LU
    r = orc.calibrate_camera(obj, ms.img, size, flags=flags, criteria=(3, 30, 1e-9), schur=False, trace=True)
    np.savez_compressed(os.path.join(HERE, "mono_8x5_aspect_lu.npz"), obj=obj.astype(np.float32), img=ms.img, size=size,
                        flags=flags, criteria=(3, 30, 1e-9), rms=r["rms"], K=r["K"], dist=r["dist"], rvecs=r["rvecs"],
                        tvecs=r["tvecs"], iters=r["iters"], trace=r["trace"],
                        dist_spread=dist_spread(orc, r, obj, ms.img, size, flags=flags, criteria=(3, 30, 1e-9)))

    obj = synth.asymmetric_circles(8, 11, 12.5)
    ms = synth.make_views(obj, 30, 0xC0FFEE + 2)
    r = orc.calibrate_camera(obj, ms.img, size, ndist=8, criteria=(3, 30, 1e-9), schur=False, trace=True)
    np.savez_compressed(os.path.join(HERE, "mono_circles_30views.npz"), obj=obj.astype(np.float32), img=ms.img, size=size,
                        flags=0, ndist=8, criteria=(3, 30, 1e-9), rms=r["rms"], K=r["K"], dist=r["dist"],
                        rvecs=r["rvecs"], tvecs=r["tvecs"], iters=r["iters"], trace=r["trace"],
                        dist_spread=dist_spread(orc, r, obj, ms.img, size, ndist=8, criteria=(3, 30, 1e-9)))

    # the reference's literal circle grid: Size(4, 11), squareSize = 2 (monocular_circle.cpp:49-50, :35-37)
    obj = synth.asymmetric_circles(4, 11, 2.0)
    ms = synth.make_views(obj, 24, 0xC0FFEE + 11, depth=(40.0, 110.0))
    crit = (3, 30, np.finfo(np.float64).eps)
    r = orc.calibrate_camera(obj, ms.img, size, ndist=8, flags=0, criteria=crit, schur=False, trace=True)
    np.savez_compressed(os.path.join(HERE, "mono_circles_4x11_ref.npz"), obj=obj.astype(np.float32), img=ms.img, size=size,
                        flags=0, ndist=8, criteria=crit, rms=r["rms"], K=r["K"], dist=r["dist"],
                        rvecs=r["rvecs"], tvecs=r["tvecs"], iters=r["iters"], trace=r["trace"],
                        dist_spread=dist_spread(orc, r, obj, ms.img, size, ndist=8, flags=0, criteria=crit))

    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 16, 0xC0FFEE + 3)
    K1 = orc.init_camera_matrix_2d(obj, st.img1, size)
    K2 = orc.init_camera_matrix_2d(obj, st.img2, size)
    flags = (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_USE_INTRINSIC_GUESS |
             pkg.CALIB_SAME_FOCAL_LENGTH | pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 |
             pkg.CALIB_FIX_K5)
    r = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, size, flags, criteria=(3, 100, 1e-5), schur=False,
                             trace=True)
    np.savez_compressed(os.path.join(HERE, "stereo_9x6_16pairs.npz"), obj=obj.astype(np.float32), img1=st.img1, img2=st.img2,
                        size=size, flags=flags, criteria=(3, 100, 1e-5), K1_init=K1, K2_init=K2, rms=r["rms"], K1=r["K1"],
                        d1=r["d1"], K2=r["K2"], d2=r["d2"], R=r["R"], T=r["T"], E=r["E"], F=r["F"], rt=r["rt"],
                        iters=r["iters"], trace=r["trace"])

    # the reference's literal stereo circle grid: Size(4, 11), squareSize = 20 (binocular_circle.cpp:52, :407-408, :19-26)
    obj = synth.asymmetric_circles(4, 11, 20.0)
    st = synth.make_stereo(obj, 18, 0xC0FFEE + 12)
    K1 = orc.init_camera_matrix_2d(obj, st.img1, size)
    K2 = orc.init_camera_matrix_2d(obj, st.img2, size)
    r = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, size, flags, criteria=(3, 100, 1e-5), schur=False,
                             trace=True)
    np.savez_compressed(os.path.join(HERE, "stereo_circles_4x11_ref.npz"), obj=obj.astype(np.float32), img1=st.img1,
                        img2=st.img2, size=size, flags=flags, criteria=(3, 100, 1e-5), K1_init=K1, K2_init=K2, rms=r["rms"],
                        K1=r["K1"], d1=r["d1"], K2=r["K2"], d2=r["d2"], R=r["R"], T=r["T"], E=r["E"], F=r["F"], rt=r["rt"],
                        iters=r["iters"], trace=r["trace"])
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
