"""include/ccalib_cv.hpp (the adaptor with OpenCV's types a maintainer of the reference would include) and
include/ccalib.hpp go through a compiler: built with -Wall -Wextra against stand-in <opencv2/*.hpp> headers
(tests/stub_opencv -- NOT OpenCV, a compile check only: it pins nothing about OpenCV's behaviour), linked against
libccalib.so so every C-ABI name the adaptor uses resolves, and run for the parts that need no GPU (distortion-vector
lengths, flag values, host-side argument checks)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cv_adaptor_compiles_links_and_checks(tmp_path):
    lib = os.path.join(ROOT, "camera-calibration_amd")
    assert os.path.exists(os.path.join(lib, "libccalib.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "cv_adaptor_check")
    for std in ("c++11", "c++17"):
        cmd = ["g++", f"-std={std}", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "stub_opencv"),
               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cv_adaptor_check.cpp"), "-o", exe,
               "-L", lib, "-lccalib", f"-Wl,-rpath,{lib}"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_adaptor_without_opencv_headers_compiles_to_nothing(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "ccalib_cv.hpp"\n#ifdef CCALIB_CV_AVAILABLE\n#error "found OpenCV headers where none are"\n#endif\n'
                   'int main() { return 0; }\n')
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
