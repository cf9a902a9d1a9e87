"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host code only; no GPU involved).

`make -C oracle sancheck` builds a stand-alone driver over the oracle's sources with -fsanitize=address,undefined
and -fno-sanitize-recover, so any out-of-bounds access, use after free, leak or undefined arithmetic in the paths
the parity tests rely on ends the run non-zero.  Its results must also equal the ordinary build's.
"""
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_runs_clean_under_asan_ubsan(orc, synth, tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sancheck"], stdout=subprocess.DEVNULL)
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 12, 0xC0FFEE + 3)
    path = tmp_path / "views.bin"
    with open(path, "wb") as f:
        np.array([12, 54], dtype=np.int32).tofile(f)
        obj.astype(np.float64).tofile(f)
        st.img1.astype(np.float64).tofile(f)
        st.img2.astype(np.float64).tofile(f)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="2")
    out = subprocess.run([os.path.join(ROOT, "oracle", "_san", "orc_san"), str(path)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    got = {}
    for ln in out.stdout.splitlines():
        m = re.match(r"(mono|mono aspect\+lu|stereo) schur=(\d) rms (\S+) iters (\d+)", ln)
        if m:
            got[(m.group(1), int(m.group(2)))] = (float(m.group(3)), int(m.group(4)))
    crit = (3, 30, 1e-9)
    for schur in (0, 1):
        ref = orc.calibrate_camera(obj, st.img1, (640, 480), criteria=crit, schur=bool(schur), nthreads=2)
        assert abs(got[("mono", schur)][0] - ref["rms"]) <= 1e-12 * ref["rms"]
        assert got[("mono", schur)][1] == ref["iters"]
    assert len(got) == 6
    # the dense and the Schur paths of the sanitized build agree with each other as the ordinary build's do
    assert abs(got[("stereo", 0)][0] - got[("stereo", 1)][0]) <= 1e-9 * got[("stereo", 0)][0]
