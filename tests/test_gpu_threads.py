"""The reference calls calibrateCamera / stereoCalibrate as plain synchronous functions (monocular_circle.cpp:93-94,
binocular_chessboard.cpp:152-162); cv:: functions may be called from several host threads at once.  The library keeps
process-wide state (cached device and pinned blocks, the pinned staging buffers, pass numbers): calls from concurrent
threads must return exactly what the same calls return one after the other."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SIZE = (640, 480)


def test_concurrent_calls_from_four_threads_match_sequential_ones(pkg):
    cc, synth = pkg, pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    jobs = []
    for t in range(4):
        ms = synth.make_views(obj, 150 + 170 * t, 0xC0FFEE + 60 + t)
        st = synth.make_stereo(obj, 120 + 90 * t, 0xC0FFEE + 70 + t)
        jobs.append((ms, st))

    def work(ms, st):
        mono = cc.calibrate_camera(obj, ms.img, SIZE)
        K1 = cc.init_camera_matrix_2d(obj, st.img1, SIZE)
        K2 = cc.init_camera_matrix_2d(obj, st.img2, SIZE)
        stereo = cc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, SIZE, cc.CALIB_USE_INTRINSIC_GUESS,
                                     criteria=(3, 25, 1e-9))
        reproj = cc.reprojection_error(obj, ms.img, mono["K"], mono["dist"], mono["rvecs"], mono["tvecs"])
        return mono, stereo, reproj

    want = [work(*j) for j in jobs]
    for rnd in range(3):
        got, errs = [None] * 4, []

        def run(i):
            try:
                got[i] = work(*jobs[i])
            except Exception as e:  # noqa: BLE001  (reported below, on the main thread)
                errs.append((i, repr(e)))

        th = [threading.Thread(target=run, args=(i,)) for i in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(4):
            assert got[i][0]["iters"] == want[i][0]["iters"]
            for k in ("rms", "K", "dist", "rvecs", "tvecs"):
                assert np.array_equal(np.asarray(got[i][0][k]), np.asarray(want[i][0][k])), (rnd, i, k)
            for k in ("rms", "K1", "K2", "d1", "d2", "R", "T", "rt"):
                assert np.array_equal(np.asarray(got[i][1][k]), np.asarray(want[i][1][k])), (rnd, i, k)
            for a, b in zip(got[i][2], want[i][2]):   # the RMS re-check of monocular_circle.cpp:5-29
                assert np.array_equal(np.asarray(a), np.asarray(b)), (rnd, i)
