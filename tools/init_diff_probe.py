#!/usr/bin/env python3
"""How far the device's starting values (cv::initCameraMatrix2D + cvFindExtrinsicCameraParams2 per view) are from the
oracle's on one golden input, and what that does to the residual sum at the start of the LM loop.
    python tools/init_diff_probe.py mono_circles_4x11_ref"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (one HIP runtime per process: before the library)
import __graft_entry__ as ge  # noqa: E402

pkg, orc = ge.load_package(), ge.load_oracle()
name = sys.argv[1] if len(sys.argv) > 1 else "mono_circles_4x11_ref"
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
obj, img = g["obj"], g["img"]
size = tuple(int(x) for x in g["size"])
K_o = orc.init_camera_matrix_2d(obj, img, size)
s = pkg.Solver(obj, img)
K_d = s.init_intrinsics(size)
rt_d = s.init_poses(K_d)
s.close()
rt_o = np.array([np.r_[orc.find_extrinsic(obj, v, K_o, np.zeros(14))] for v in img])
print("K seed rel diff", np.abs(K_d - K_o).max() / np.abs(K_o).max())
print("pose diff: rvec", np.abs(rt_d[:, :3] - rt_o[:, :3]).max(), "tvec rel", np.abs(rt_d[:, 3:] - rt_o[:, 3:]).max() / np.abs(rt_o[:, 3:]).max())
for label, rt in (("device start", rt_d), ("oracle start", rt_o)):
    r = orc.calibrate_camera(obj, img, size, K=K_o, dist=np.zeros(8), ndist=8, flags=orc.USE_INTRINSIC_GUESS, criteria=(3, 2, 0.0),
                             init_rt=rt, schur=False, trace=True)
    print(label, ["%.13e" % v for v in r["trace"][:, 4]], ["%.13e" % v for v in r["trace"][:, 5]])
