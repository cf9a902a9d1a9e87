#!/usr/bin/env python3
"""Times the whole calibrateCamera call (pose initialisation + LM + download) from host buffers:
   python tools/wholecall_bench.py [views]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    """Three shapes of the same call, best of five each: the flat C entry point, the per-view-pointer entry point over
    N separate copies of the board (objectPoints.resize(n, objectPoints[0]), monocular_chessboard.cpp:249) and the
    Python mirror."""
    views = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    pkg = ge.load_package()
    synth = pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, views, 0xC0FFEE + 4)
    import ctypes as C
    lib = pkg.capi.load()
    N, npts, o, img = pkg._flatten(obj, ms.img)        # the caller's buffers: vector<vector<Point3f/2f>> flattened
    boards = o.reshape(N, -1, 3)                       # N copies of the board, each view pointing at its own
    optr = boards.ctypes.data + np.arange(N, dtype=np.uint64) * np.uint64(boards[0].nbytes)
    iptr = img.ctypes.data + np.arange(N, dtype=np.uint64) * np.uint64(8 * int(npts[0]))
    out = {"views": views}

    def run(name, call):
        best = None
        for rep in range(5):
            A, d, rv, tv = np.eye(3).reshape(-1), np.zeros(5), np.zeros((N, 3)), np.zeros((N, 3))
            rms, st = C.c_double(), pkg.capi.Stats()
            t0 = time.perf_counter()
            got = call(A, d, rv, tv, rms, st)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[name] = dict(whole_call_ms=round(best * 1e3, 3), **got)

    def flat(A, d, rv, tv, rms, st):
        pkg.capi.check(lib.ccalib_calibrate_camera(
            C.c_int(N), pkg._p(npts), pkg._p(o), pkg._p(img), C.c_int(640), C.c_int(480), pkg._p(A), pkg._p(d), C.c_int(5),
            pkg._p(rv), pkg._p(tv), C.c_int(0), C.c_int(3), C.c_int(30), C.c_double(2.2e-16), None, C.byref(rms), C.byref(st)))
        return dict(lm_loop_ms=round(st.device_ms, 3), iters=st.iters, rms=rms.value)

    def ptrs(A, d, rv, tv, rms, st):
        pkg.capi.check(lib.ccalib_calibrate_camera_views(
            C.c_int(N), pkg._p(npts), pkg._p(optr), pkg._p(iptr), C.c_int(640), C.c_int(480), pkg._p(A), pkg._p(d), C.c_int(5),
            pkg._p(rv), pkg._p(tv), C.c_int(0), C.c_int(3), C.c_int(30), C.c_double(2.2e-16), None, C.byref(rms), C.byref(st)))
        return dict(lm_loop_ms=round(st.device_ms, 3), iters=st.iters, rms=rms.value)

    def mirror(A, d, rv, tv, rms, st):
        got = pkg.calibrate_camera(obj, ms.img, (640, 480))
        return dict(lm_loop_ms=round(got["device_ms"], 3), iters=got["iters"], rms=got["rms"])

    run("flat", flat)
    run("views", ptrs)
    run("python", mirror)
    print(json.dumps(out))


def stereo():
    """python tools/wholecall_bench.py stereo [pairs]: cv::stereoCalibrate's whole call with the reference's flag
    set (BASELINE config 3: 10 000 pairs), initCameraMatrix2D for both cameras included."""
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    pkg = ge.load_package()
    synth = pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, pairs, 0xC0FFEE + 3)
    flags = (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_USE_INTRINSIC_GUESS |
             pkg.CALIB_SAME_FOCAL_LENGTH | pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 |
             pkg.CALIB_FIX_K5)
    for rep in range(3):
        t0 = time.perf_counter()
        K1 = pkg.init_camera_matrix_2d(obj, st.img1, (640, 480))
        K2 = pkg.init_camera_matrix_2d(obj, st.img2, (640, 480))
        t1 = time.perf_counter()
        got = pkg.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, (640, 480), flags, criteria=(3, 100, 1e-5))
        t2 = time.perf_counter()
    print(json.dumps({"pairs": pairs, "init_camera_matrix_s": round(t1 - t0, 4), "stereo_calibrate_s": round(t2 - t1, 4),
                      "lm_loop_ms": round(got["device_ms"], 3), "iters": got["iters"], "rms": got["rms"]}))


if __name__ == "__main__":
    stereo() if len(sys.argv) > 1 and sys.argv[1] == "stereo" else main()
