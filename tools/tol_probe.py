#!/usr/bin/env python3
"""Prints how far the HIP path is from the oracle on the inputs whose parity tolerances are wider than 1e-6
(the rational-model cases and the one- and three-view cases of tests/test_gpu_parity.py), next to the oracle's own
dense-vs-Schur spread on the same inputs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package(); orc = ge.load_oracle(); synth = pkg.synth
obj = synth.chessboard(9, 6, 25.0)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())

ms = synth.make_views(obj, 30, 23)
K0 = np.array([[780.0, 0, 322.0], [0, 790.0, 238.0], [0, 0, 1]])
crit = (3, 40, 1e-9)
fl = pkg.CALIB_USE_INTRINSIC_GUESS | pkg.CALIB_RATIONAL_MODEL
g = pkg.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(8), ndist=8, flags=fl, criteria=crit)
a = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(8), ndist=8, flags=fl, criteria=crit, schur=False)
b = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(8), ndist=8, flags=fl, criteria=crit, schur=True)
print("mono rational8: hip-vs-dense K", rel(g["K"], a["K"]), "rvec", np.abs(g["rvecs"] - a["rvecs"]).max(), "rms", abs(g["rms"] - a["rms"]) / a["rms"],
      "| oracle dense-vs-schur K", rel(b["K"], a["K"]), "iters", g["iters"], a["iters"], b["iters"])
for it in (10, 20, 40):
    c = (3, it, 0.0)
    g = pkg.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(8), ndist=8, flags=fl, criteria=c)
    a = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(8), ndist=8, flags=fl, criteria=c, schur=False)
    b = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(8), ndist=8, flags=fl, criteria=c, schur=True)
    print(f"  after exactly {it} iterations: hip-vs-dense K {rel(g['K'], a['K']):.2e} dist {np.abs(g['dist'] - a['dist']).max():.2e} | dense-vs-schur K {rel(b['K'], a['K']):.2e} dist {np.abs(b['dist'] - a['dist']).max():.2e}")

st = synth.make_stereo(obj, 22, 4100)
K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480)); K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
crit = (3, 25, 1e-7)
g = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), fl, criteria=crit)
a = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), fl, criteria=crit, schur=False)
b = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), fl, criteria=crit, schur=True)
for k in ("K1", "K2", "R", "T"):
    print("stereo rational12", k, "hip-vs-dense", rel(g[k], a[k]), "| dense-vs-schur", rel(b[k], a[k]))
print("stereo rational12 rt", np.abs(g["rt"] - a["rt"])[:, :3].max(), "|", np.abs(b["rt"] - a["rt"])[:, :3].max(), "iters", g["iters"], a["iters"], b["iters"])

crit = (3, 30, 1e-9)
for n in (1, 3):
    ms = synth.make_views(obj, n, 900 + n)
    K0 = ms.cam.K() * np.array([[1.01, 1, 1], [1, 1.01, 1], [1, 1, 1]])
    f2 = pkg.CALIB_USE_INTRINSIC_GUESS | pkg.CALIB_FIX_K3 | pkg.CALIB_ZERO_TANGENT_DIST
    g = pkg.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(5), flags=f2, criteria=crit)
    a = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(5), flags=f2, criteria=crit, schur=False)
    b = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=np.zeros(5), flags=f2, criteria=crit, schur=True)
    print("views", n, "hip-vs-dense K", rel(g["K"], a["K"]), "| dense-vs-schur", rel(b["K"], a["K"]), "iters", g["iters"], a["iters"], b["iters"])
