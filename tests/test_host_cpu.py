"""CPU-only tests of the host side: the C ABI surface, the host initialisation routines and the
LM driver (single rank and sharded over two gloo ranks).  No kernel is launched here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.capi.load()
    header = open(os.path.join(ROOT, "include", "ccalib.h")).read()
    declared = set(re.findall(r"\b(ccalib_[a-z0-9_]+)\s*\(", header))
    declared -= {"ccalib_allreduce_fn"}
    assert declared, "no declarations found"
    assert declared == set(pkg.capi.SYMBOLS), (declared ^ set(pkg.capi.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ccalib_version().startswith(b"ccalib")


def test_compute_fails_loudly_without_a_gpu(pkg, synth):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 4, 1)
    with pytest.raises(pkg.CcalibError) as ei:
        pkg.calibrate_camera(obj, ms.img, (640, 480))
    assert ei.value.code == pkg.capi.ERR_NO_DEVICE
    with pytest.raises(pkg.CcalibError):
        pkg.Solver(obj, ms.img)


def test_cpp_dropin_header_builds_and_reports_missing_gpu():
    """include/ccalib.hpp (the cv::-shaped C++ surface) links against the C ABI with plain g++; without a
    GPU the first compute call throws ccalib::Exception carrying CCALIB_ERR_NO_DEVICE."""
    import torch
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_check")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, out.stdout + out.stderr
    assert out.stdout.startswith("exception 3 ")


def test_initialisation_has_no_cpu_path(pkg, synth):
    """cv::initCameraMatrix2D goes to the device like everything else: without a GPU it fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 12, 9)
    with pytest.raises(pkg.CcalibError) as ei:
        pkg.init_camera_matrix_2d(obj, ms.img, (640, 480))
    assert ei.value.code == pkg.capi.ERR_NO_DEVICE
    with pytest.raises(pkg.CcalibError):      # argument checks come first
        pkg.init_camera_matrix_2d(obj[:3], ms.img[:, :3], (640, 480))


@pytest.mark.parametrize("flags_name", ["zero", "aspect_lu"])
def test_lm_driver_matches_oracle_state_machine(pkg, orc, synth, flags_name):
    """The product's C++ driver over a CPU backend against the oracle's own (restated CvLevMarq) loop."""
    from tests.cpu_backend import OracleBackend, run_lm
    flags = 0 if flags_name == "zero" else (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_USE_LU)
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 16, 0xC0FFEE + 1)
    cam = ms.cam
    K0 = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.97, cam.cy - 2], [0, 0, 1.0]])
    rt0 = synth.perturbed_poses(ms.rvec, ms.tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    intr, mask, aspect = pkg.mono_start(K0, cam.dist(5) * 0.8, flags)
    crit = (3, 30, 1e-9)
    be = OracleBackend(pkg, orc, obj, ms.img, rt0, mask, aspect)
    got_intr, got_rt, st = run_lm(pkg, be, intr, crit, use_lu=bool(flags & pkg.CALIB_USE_LU))
    ref = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=cam.dist(5) * 0.8, ndist=5,
                               flags=flags | orc.USE_INTRINSIC_GUESS, criteria=crit, init_rt=rt0, schur=False)
    assert st["iters"] == ref["iters"] and st["jac_passes"] == ref["jac_passes"]
    # near convergence the accept/reject test compares sums that agree to round-off, so the number of
    # rejected trial steps (extra residual passes) is not reproducible between two summation orders
    assert st["err_passes"] >= st["iters"] and abs(st["err_passes"] - ref["err_passes"]) <= 16
    assert abs(st["final_err"] - ref["final_err"]) <= 1e-9 * ref["final_err"]
    assert np.allclose(got_intr[:4], [ref["K"][0, 0], ref["K"][1, 1], ref["K"][0, 2], ref["K"][1, 2]], rtol=1e-8)
    assert np.allclose(got_intr[4:9], ref["dist"], rtol=1e-6, atol=1e-10)
    assert np.abs(got_rt[:, :3] - ref["rvecs"]).max() < 1e-8
    assert be.calls["commit"] == st["iters"]


def test_lm_trace_one_json_line_per_iteration(pkg, orc, synth, tmp_path, monkeypatch):
    """CCALIB_TRACE=<file>: the driver appends one JSON record per LM iteration (damping exponent tried
    first and accepted, number of trial steps, residual sums before and after)."""
    import json
    from tests.cpu_backend import OracleBackend, run_lm
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 8, 3)
    cam = ms.cam
    K0 = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.97, cam.cy - 2], [0, 0, 1.0]])
    intr, mask, aspect = pkg.mono_start(K0, cam.dist(5) * 0.8, 0)
    rt0 = synth.perturbed_poses(ms.rvec, ms.tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    path = tmp_path / "trace.jsonl"
    monkeypatch.setenv("CCALIB_TRACE", str(path))
    _, _, st = run_lm(pkg, OracleBackend(pkg, orc, obj, ms.img, rt0, mask, aspect), intr, (3, 6, 0.0))
    recs = [json.loads(ln) for ln in path.read_text().splitlines()]
    assert [r["iter"] for r in recs] == list(range(st["iters"])) and st["iters"] == 6
    assert sum(r["trials"] for r in recs) == st["err_passes"]
    assert all(r["err"] <= r["err_prev"] for r in recs) and recs[-1]["err"] == st["final_err"]
    assert recs[0]["lambda_lg10_first"] == -3


def test_sharded_lm_over_two_gloo_ranks():
    """world_size 2 on CPU: each rank owns half the views, every sum goes through gloo all_reduce."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613",
                          os.path.join(ROOT, "tests", "gloo_worker.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_OK" in out.stdout


def _load_opencv_yaml(path):
    """Reads cv::FileStorage YAML 1.0 (the '%YAML:1.0' directive and '!!opencv-matrix' nodes) with PyYAML."""
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    def matrix(loader, node):
        m = loader.construct_mapping(node, deep=True)
        assert m["dt"] == "d"
        return np.array(m["data"], dtype=np.float64).reshape(m["rows"], m["cols"])

    Loader.add_constructor("tag:yaml.org,2002:opencv-matrix", matrix)
    text = open(path).read()
    assert text.startswith("%YAML:1.0\n---\n")
    return yaml.load(text.split("---\n", 1)[1], Loader=Loader)


def test_yaml_result_files_round_trip(pkg, tmp_path):
    """The files the reference writes after calibrating (monocular_chessboard.cpp:321-328,
    binocular_chessboard.cpp:203-210 and :234-241): same keys, same node types, values bit-exact."""
    rng = np.random.default_rng(5)
    K = np.array([[812.123456789, 0, 322.5], [0, 806.25, 238.0], [0, 0, 1.0]])
    D = rng.normal(size=5) * 0.1
    rv, tv = rng.normal(size=(7, 3)), rng.normal(size=(7, 3)) * 100
    p = tmp_path / "trinsics.yml"
    pkg.write_mono_yaml(p, (640, 480), K, D, rv, tv)
    y = _load_opencv_yaml(p)
    assert list(y) == ["imageSize", "M", "D", "R", "t"] and y["imageSize"] == [640, 480]
    assert np.array_equal(y["M"], K) and np.array_equal(y["D"].ravel(), D) and y["D"].shape == (5, 1)
    assert len(y["R"]) == 7 and all(m.shape == (3, 1) for m in y["R"])
    assert np.array_equal(np.hstack(y["R"]).T, rv) and np.array_equal(np.hstack(y["t"]).T, tv)
    pi, pe = tmp_path / "intrinsics.yml", tmp_path / "extrinsics.yml"
    D14 = np.r_[D, rng.normal(size=3) * 0.01, np.zeros(6)]
    R, T = np.eye(3) + 1e-3 * rng.normal(size=(3, 3)), np.array([-60.0, 0.5, 1.0])
    pkg.write_stereo_yaml(pi, pe, (640, 480), K, D14, K * 1.01, D14 * 0.5, R, T, E=R * 2, F=R * 3)
    yi, ye = _load_opencv_yaml(pi), _load_opencv_yaml(pe)
    assert list(yi) == ["M1", "D1", "M2", "D2"] and yi["D1"].shape == (1, 14)
    assert np.array_equal(yi["M2"], K * 1.01) and np.array_equal(yi["D2"].ravel(), D14 * 0.5)
    assert list(ye) == ["imageSize", "R", "T", "E", "F"] and np.array_equal(ye["R"], R) and np.array_equal(ye["T"].ravel(), T)
    with pytest.raises(pkg.CcalibError):
        pkg.write_mono_yaml(tmp_path / "no_such_dir" / "x.yml", (1, 1), K, D, rv, tv)
