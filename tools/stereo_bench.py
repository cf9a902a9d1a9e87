#!/usr/bin/env python3
"""Times the stereoCalibrate LM loop on one GPU:
   python tools/stereo_bench.py [pairs] [iters]      prints LM ms/iteration and point pairs/s
   BOARD=9x6 (default; with the reference's flag set: BASELINE config 3's shape) or e.g. BOARD=17x12 FULL=1
   (all of k1-k3, p1, p2 free in both cameras: config 5's shape, 125 000 pairs being one GPU's share)"""
import json
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    pkg = ge.load_package()
    synth = pkg.synth
    bw, bh = (int(x) for x in os.environ.get("BOARD", "9x6").split("x"))
    obj = synth.chessboard(bw, bh, 25.0 if bw * bh <= 64 else 10.0)
    st = synth.make_stereo(obj, pairs, 0xC0FFEE + 3)
    K1 = pkg.init_camera_matrix_2d(obj, st.img1[:200], (640, 480))
    K2 = pkg.init_camera_matrix_2d(obj, st.img2[:200], (640, 480))
    flags = (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_USE_INTRINSIC_GUESS |
             pkg.CALIB_SAME_FOCAL_LENGTH | pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 |
             pkg.CALIB_FIX_K5)
    if os.environ.get("FULL"):
        flags = pkg.CALIB_USE_INTRINSIC_GUESS
    rt0 = synth.perturbed_poses(st.rvec, st.tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    lr0 = np.r_[st.r_lr, st.t_lr] + np.array([1e-3, -1e-3, 5e-4, 0.5, -0.2, 0.3])
    for rep in range(2):   # first call warms code objects
        got = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags,
                                   criteria=(3, iters, 0.0), init_rt=rt0, init_lr=lr0)
    ms = got["device_ms"] / got["iters"]
    print(json.dumps({"board": f"{bw}x{bh}", "pairs": pairs, "iters": got["iters"], "jac_passes": got["jac_passes"], "err_passes": got["err_passes"],
                      "ms_per_iter": round(ms, 4), "point_pairs_per_s": pairs * len(obj) * got["jac_passes"] / (got["device_ms"] * 1e-3),
                      "rms": got["rms"]}))


if __name__ == "__main__":
    main()
