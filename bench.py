#!/usr/bin/env python3
"""Benchmark of the calibration LM loop on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload: BASELINE.json configs[3] -- monocular 9x6 chessboard, 100 000 synthetic
640x480 views PER GPU (weak scaling: N GPUs solve ONE calibration over N x 100k
views sharded by view; the shared normal-equation block and three scalars are
all-reduced every pass), 9 free intrinsics (flags = 0), FP64.
A step is one LM iteration: Jacobian pass (residuals + analytic Jacobians + block
accumulation + per-view 6x6 Schur elimination), host solve of the reduced 9x9
system, back-substitution and the residual pass at the trial point (repeated if
the step is rejected).  Inputs are resident in HBM before the timed region.
value = residual+Jacobian point evaluations per second over all ranks
      = (points over all ranks) x (Jacobian passes in the timed region) / time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

# Algorithmic FP64 flops per point (FMA = 2), instrumented by oracle/flopcount.py on the
# zero-exploiting, factored formulation; and algorithmic HBM bytes per point for P = 54, c = 9.
FLOPS_JAC_PASS = 556       # whole Jacobian pass in one sweep: projection + Jacobians + U, gc, V, ge, W updates
FLOPS_JAC_VIEW = 428       # the per-view-block part alone (the two-kernel path for rational models)
FLOPS_RESIDUAL = 54
# The fused kernel does not exploit the zeros: it multiplies the dense 16-column row matrix on the
# FP64 matrix pipe (upper-triangle 4x4 blocks: ceil(2P/8)*5 instructions of 4x4x4_4b per view) after
# ~156 vector FP64 instructions per point.  Executed flops are reported beside the algorithmic ones.
MFMA_FLOPS = 2 * 4 * (4 * 4 * 4)     # v_mfma_f64_4x4x4_4b_f64: four 4x4x4 blocks
BYTES_JAC_VIEW = (8 * 54 + 104 + 8 * (21 + 6 + 6 * 9)) / 54.0   # image points + pose record in, blocks out, per point
PEAK_FP64_TFLOPS = 78.6    # MI355X dense FP64, matrix and vector alike (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
PEAK_HBM_GBPS = 8000.0
PROFILE_PMC = os.path.join(ROOT, "profiles", "r03_bench_pmc.json")   # rocprofv3 FETCH_SIZE / WRITE_SIZE passes
PROFILE_STEREO_PMC = os.path.join(ROOT, "profiles", "r03_stereo_pmc.json")
REPEATS = 20               # timed runs of the same K steps in front of the reported one: spread of ms_per_step, and the clock
# stereoCalibrate with the reference's flag set: per camera fy cx cy k1 k2 k6 free, fx tied, rational model
# (oracle/flopcount.py, stereo_reference_flags_6col_per_camera); a point pair is one point in each camera
FLOPS_STEREO_PER_CAMERA_POINT = 473


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=600,
                    help="untimed LM iterations, taken in runs of --steps iterations from the same start point, i.e. the very "
                         "iterations the timed run executes.  The default keeps the GPU busy for about 0.15 s: the card takes "
                         "about 0.1 s of load to reach its sustained clock (measured: 0.174 ms per step right after 20 "
                         "warm-up steps, falling to 0.155 over the next ten runs)")
    ap.add_argument("--views", type=int, default=100_000, help="views per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-views", type=int, default=4000, help="views in the bounded CPU sample")
    ap.add_argument("--allreduce", choices=["rccl", "torch"], default="rccl")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --views per GPU (default); strong: --views in total, split over the GPUs")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CCALIB_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        # (ranks then share GPUs and every sum goes through a host callback)
        backend = os.environ.get("CCALIB_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group(backend=backend)
            args.allreduce = "torch"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the library has no CPU path)")
    torch.cuda.set_device(local_rank)

    pkg = ge.load_package()
    synth = pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    V = args.views if args.scaling == "weak" else args.views // world   # views this rank owns
    seed = 0xC0FFEE + 4
    ms = synth.make_views(obj, V, seed, first_view=rank * V)
    cam = ms.cam
    # start: intrinsics a few percent off, poses perturbed (pose initialisation is not part of the timed loop)
    K0 = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.98, cam.cy - 2], [0, 0, 1.0]])
    intr, mask, _ = pkg.mono_start(K0, cam.dist(5) * 0.8, 0)
    rt0 = synth.perturbed_poses(ms.rvec, ms.tvec, seed, first_view=rank * V, rot_sigma=0.01, trans_sigma=3.0)

    solver = pkg.Solver(obj, ms.img, device=local_rank)
    mode = "single"
    if world > 1:
        mode = args.allreduce
        if mode == "rccl":
            try:
                ids = [None]
                if rank == 0:
                    import ctypes as C
                    buf = C.create_string_buffer(128)
                    pkg.capi.check(pkg.capi.load().ccalib_rccl_unique_id(buf))
                    ids = [buf.raw]
                dist.broadcast_object_list(ids, src=0)
                solver.use_rccl(world, rank, ids[0])
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] in-stream RCCL unavailable ({e}); using torch.distributed", file=sys.stderr)
                mode = "torch"
            # the ranks must agree: one rank on the callback path while the others wait in ncclAllReduce hangs the job
            ok = torch.tensor([1 if mode == "rccl" else 0], dtype=torch.int32,
                              device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and mode == "rccl":
                solver.close()
                solver = pkg.Solver(obj, ms.img, device=local_rank)
                mode = "torch"
        if mode == "torch":
            gloo = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None

            def allreduce(buf):
                t = torch.from_numpy(buf)
                dist.all_reduce(t, group=gloo)
            solver.use_callback(allreduce)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run():
        """Exactly K LM iterations from the start point, bracketed by barrier + synchronize; max over ranks."""
        solver.set_start(intr, mask, rt0)   # uploads the starting poses; untimed
        barrier()
        t0 = time.perf_counter()
        st = solver.run(criteria=(3, args.steps, 0.0))   # eps = 0: exactly K iterations
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return st, dt

    # warm-up: W untimed LM iterations from the start point
    for w0 in range(0, args.warmup, max(1, args.steps)):
        solver.set_start(intr, mask, rt0)
        solver.run(criteria=(3, min(args.steps, args.warmup - w0), 0.0))
    # The same K steps, REPEATS + 1 times, every run bracketed and timed the same way.  THE timed region -- what `value`
    # and `ms_per_step` are computed from -- is the LAST of them: a run is a few milliseconds of GPU work and the card
    # needs about 0.1 s of load to reach its sustained clock, so the first run after a short warm-up measures the clock
    # ramp (0.174 ms per step against 0.155 ten runs later).  All runs are listed in order in `ms_per_step_series`.
    spread = []
    for _ in range(REPEATS):
        spread.append(timed_run()[1] / args.steps * 1e3)
    exec0 = solver.executed_passes()
    st, dt = timed_run()
    exec1 = solver.executed_passes()
    spread.append(dt / args.steps * 1e3)
    out = solver.get()
    # and once more with a HIP event before and after every per-point Jacobian launch: the kernel's duration where
    # it runs, between the other kernels of an iteration (the events cost time, so this is not the timed run)
    solver.profile_jac(True)
    timed_run()
    inloop_ms, inloop_n = solver.profile_jac(False)
    points_local = V * obj.shape[0]
    points_total = points_local * world
    rms = float(np.sqrt(out["err"] / points_total))
    value = points_total * st["jac_passes"] / dt

    result = None
    if rank == 0:
        # dominant kernel alone, HIP events on the solver's stream
        solver.set_start(out["intr"], mask, out["rt"])
        km = solver.time_kernels(reps=50)
        fused = km["jac_shared"] == 0.0      # the fused kernel covers the whole Jacobian pass
        # priced on the in-loop launches when the pass is one kernel (the events bracket exactly that kernel)
        t_jac = (inloop_ms if fused and inloop_n > 0 else km["jac_view"]) * 1e-3
        flops_alg = FLOPS_JAC_PASS if fused else FLOPS_JAC_VIEW
        achieved_tf = flops_alg * points_local / t_jac / 1e12
        P = int(obj.shape[0])
        # five block instructions per eight rows of A (ten of the sixteen 4x4 blocks of the symmetric product)
        executed = (-(-2 * P // 8) * 5 * MFMA_FLOPS + 2 * 156 * P) / P if fused else None
        traffic = None
        traffic_source = None
        try:   # per-launch HBM bytes from the committed counter passes (FETCH_SIZE doubled, as gfx950 needs):
            # counters cannot be read inside this run, so the figure is only given for the configuration it was
            # taken on, and labelled as what it is
            if not (V == 100_000 and obj.shape[0] == 54):
                raise ValueError("profile taken at 100k views of a 9x6 board")
            pm = json.load(open(PROFILE_PMC))
            for name, c in pm.items():
                if name.startswith("k_jac_fused" if fused else "k_jac_view"):
                    traffic = int((2 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024)
                    traffic_source = ("committed rocprofv3 --pmc passes (%s), same kernel and configuration; "
                                      "not measured in this run" % os.path.relpath(PROFILE_PMC, ROOT))
        except (OSError, KeyError, ValueError):
            pass
        roofline = {
            "bound": "mfma", "kernel": "k_jac_fused" if fused else "k_jac_view", "achieved": round(achieved_tf, 3),
            "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved_tf / PEAK_FP64_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "launch_ms": round(t_jac * 1e3, 4),
            "launch_ms_source": ("HIP events around each of the %d launches inside a run of the timed region's %d steps "
                                 "(a separate, identical run: the events cost time)" % (inloop_n, args.steps)) if fused and inloop_n > 0
                                else "HIP events around 50 back-to-back launches",
            "launch_ms_back_to_back": round(km["jac_view"], 4),
            "algorithmic_flops_per_point": flops_alg,
            "executed_flops_per_point": round(executed, 1) if executed else None,
            "executed_tflops": round(executed * points_local / t_jac / 1e12, 2) if executed else None,
            "algorithmic_bytes_per_launch": int(BYTES_JAC_VIEW * points_local),
            "hbm_achieved_GBps": round(BYTES_JAC_VIEW * points_local / t_jac / 1e9, 1),
            "hbm_frac": round(BYTES_JAC_VIEW * points_local / t_jac / 1e9 / PEAK_HBM_GBPS, 4),
            "other_kernels_ms": {"k_jac_shared": round(km["jac_shared"], 4), "k_err_wave": round(km["err"], 4)},
            "note": "FP64 path: the normal-equation update of a view is one A^T A product on the FP64 matrix pipe "
                    "(v_mfma_f64_4x4x4_4b, upper-triangle blocks only), whose dense peak equals the FP64 vector peak on this part; achieved counts "
                    "the zero-exploiting algorithmic flops (oracle/flopcount.py), executed_* what the kernel issues",
        }
        # the same pass on the vector ALU alone (two kernels, zero-exploiting accumulation), for comparison
        valu = None
        try:
            os.environ["CCALIB_NO_FUSED"] = "1"
            s2 = pkg.Solver(obj, ms.img, device=local_rank)
            s2.set_start(out["intr"], mask, out["rt"])
            k2 = s2.time_kernels(reps=20)
            s2.close()
            t2 = (k2["jac_shared"] + k2["jac_view"]) * 1e-3
            valu = {"k_jac_shared_ms": round(k2["jac_shared"], 4), "k_jac_view_ms": round(k2["jac_view"], 4),
                    "algorithmic_tflops": round((FLOPS_JAC_VIEW + 190) * points_local / t2 / 1e12, 2),
                    "note": "the Jacobian pass without the matrix pipe (CCALIB_NO_FUSED=1): FP64 vector ALU only"}
        except Exception as e:  # noqa: BLE001
            valu = {"error": str(e)}
        finally:
            os.environ.pop("CCALIB_NO_FUSED", None)
        roofline["valu_only_path"] = valu
        cpu = None
        whole = None
        stereo = None
        other = None
        if not args.no_cpu and world == 1:   # the CPU leg and the whole-call leg are timed at N = 1 only
            cpu = cpu_baseline(ge, synth, obj, args.cpu_views, seed, pkg)
            cpu["opencv"] = opencv_check(pkg, synth)
            whole = whole_call(pkg, obj, ms)
            stereo = stereo_block(pkg, synth, local_rank, args.steps, args.warmup)
            other = {"configs[1]": config1_block(pkg, synth, local_rank, args.steps, args.warmup),
                     "configs[4] one-GPU share": config4_share_block(pkg, synth, local_rank, args.warmup)}
        result = {
            "metric": "residual+Jacobian point evaluations/s in the calibrateCamera LM loop (%s, 9x6 board)"
                      % (f"{args.views // 1000}k views/GPU" if args.scaling == "weak" else f"{args.views // 1000}k views in total"),
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "ms_per_step_min": min(spread), "ms_per_step_median": float(np.median(spread)),
            "ms_per_step_max": max(spread), "ms_per_step_runs": len(spread), "ms_per_step_series": [round(x, 4) for x in spread],
            "timed_region": "run %d of %d identical runs of %d steps (the last); %d untimed steps precede it: %d warm-up + the "
                            "earlier runs" % (len(spread), len(spread), args.steps, args.warmup + (len(spread) - 1) * args.steps,
                                              args.warmup),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[3]: monocular 9x6 chessboard, %d synthetic 640x480 views %s, "
                                   "flags=0 (9 free intrinsics), sharded by view"
                                   % (args.views, "per GPU" if args.scaling == "weak" else "split over the GPUs"),
                       "views_per_gpu": V,
                       "points_per_view": int(obj.shape[0]), "allreduce": mode},
            "lm_iters_per_sec": args.steps / dt,
            # passes over all points that actually ran in the timed region (a residual pass the state machine asks for at
            # a point whose Jacobian pass was run ahead is answered from that pass and does not run)
            "projected_points_per_sec": points_total * ((exec1[0] - exec0[0]) + (exec1[1] - exec0[1])) / dt,
            "executed_passes": {"jacobian": exec1[0] - exec0[0], "residual_only": exec1[1] - exec0[1]},
            "jac_passes": st["jac_passes"], "err_passes": st["err_passes"], "rms_px": rms,
            "roofline": roofline, "cpu_baseline": cpu, "whole_call": whole, "stereo": stereo, "other_configs": other,
        }
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def _stereo_start(pkg, synth, obj, st):
    """Start of the stereo LM loop for the reference's flag set: averaged intrinsic guesses, perturbed poses."""
    K = [pkg.init_camera_matrix_2d(obj, im[:2000], (640, 480)) for im in (st.img1, st.img2)]
    intr = np.zeros((2, 18))
    for c in range(2):      # CALIB_SAME_FOCAL_LENGTH averages the two guesses, as cvStereoCalibrateImpl does
        intr[c, :4] = [(K[0][0, 0] + K[1][0, 0]) / 2, (K[0][1, 1] + K[1][1, 1]) / 2,
                       (K[0][0, 2] + K[1][0, 2]) / 2, (K[0][1, 2] + K[1][1, 2]) / 2]
    mask = np.zeros((2, 18), dtype=np.uint8)
    mask[:, [1, 2, 3, 4, 5, 11]] = 1
    aspect = [intr[0, 0] / intr[0, 1], intr[1, 0] / intr[1, 1]]
    rt0 = synth.perturbed_poses(st.rvec, st.tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    lr0 = np.r_[st.r_lr, st.t_lr] + np.array([1e-3, -1e-3, 5e-4, 0.5, -0.2, 0.3])
    return dict(intr=intr, imask=mask, rt=rt0, aspect=aspect, same_focal=True, lr=lr0)


def _timed_steps(s, start, steps, warmup, repeats=REPEATS):
    """K LM iterations from `start`, timed `repeats` + 1 times (the last run is the reported one), then one with in-loop events."""
    for w0 in range(0, warmup, max(1, steps)):
        s.set_start(**start)
        s.run(criteria=(3, min(steps, warmup - w0), 0.0))
    ms, stt = [], None
    for r in range(1 + repeats):   # the reported run is the last one (the clock: see the main block)
        last = r == repeats
        s.set_start(**start)
        if last:
            e0 = s.executed_passes()
        t0 = time.perf_counter()
        got = s.run(criteria=(3, steps, 0.0))
        ms.append((time.perf_counter() - t0) / steps * 1e3)
        if last:
            stt, e1 = got, s.executed_passes()
    s.profile_jac(True)
    s.set_start(**start)
    s.run(criteria=(3, steps, 0.0))
    inloop_ms, inloop_n = s.profile_jac(False)
    return dict(stats=stt, ms=ms, inloop_ms=inloop_ms, inloop_n=inloop_n, executed=(e1[0] - e0[0], e1[1] - e0[1]))


def stereo_block(pkg, synth, dev, steps, warmup):
    """The stereoCalibrate LM loop (binocular_chessboard.cpp:152-162: its flag set, 9x6 board) on one GPU with the
    pairs resident in HBM: BASELINE configs[2] (10 000 pairs) and, because that size is bound by launch and host
    latencies rather than by the kernels, the same problem at 100 000 pairs.  A step is one LM iteration (both
    cameras' Jacobian pass, the per-view rewrite and elimination, the 18x18 reduced solve, back-substitution,
    and the residual passes of rejected trial steps)."""
    obj = synth.chessboard(9, 6, 25.0)
    P = int(obj.shape[0])
    out = {}
    try:
        pmc = json.load(open(PROFILE_STEREO_PMC))
    except (OSError, ValueError):
        pmc = {}
    for label, pairs in (("configs[2]: 10k pairs", 10_000), ("100k pairs", 100_000)):
        st = synth.make_stereo(obj, pairs, 0xC0FFEE + 3)
        s = pkg.Solver(obj, st.img1, st.img2, device=dev)
        start = _stereo_start(pkg, synth, obj, st)
        r = _timed_steps(s, start, steps, warmup)
        res = s.get()
        how = s.describe()
        km = s.time_kernels(reps=50)      # the Jacobian launch alone, back to back, HIP events on the solver's stream
        s.close()
        stt, dt_ms = r["stats"], r["ms"][-1]
        cams = how["cams_per_launch"]
        t_jac = (r["inloop_ms"] if r["inloop_n"] > 0 else km["jac_view"]) * 1e-3
        tf = FLOPS_STEREO_PER_CAMERA_POINT * cams * pairs * P / t_jac / 1e12
        traffic = traffic_source = None
        for name, c in pmc.get(str(pairs), {}).items():   # counters of the committed passes, same kernel and size
            if name.startswith("k_jac_fused12" if how["kernel"] == "fused12" else "k_jac_fused<"):
                traffic = int((2 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024)
                traffic_source = ("committed rocprofv3 --pmc passes (%s), same kernel and size; not measured in this run"
                                  % os.path.relpath(PROFILE_STEREO_PMC, ROOT))
        bytes_alg = cams * pairs * (8 * P + 104 + 8 * (27 + 6 * how["ncols"]))
        out[label] = {
            "pairs": pairs, "points_per_view": P, "steps": steps, "ms_per_step": round(dt_ms, 4),
            "ms_per_step_min": round(min(r["ms"]), 4), "ms_per_step_median": round(float(np.median(r["ms"])), 4),
            "ms_per_step_max": round(max(r["ms"]), 4),
            "lm_iters_per_sec": round(1e3 / dt_ms, 1),
            "point_pairs_per_sec": pairs * P * stt["jac_passes"] / (dt_ms * steps * 1e-3),
            "jac_passes": stt["jac_passes"], "err_passes": stt["err_passes"],
            "executed_passes": {"jacobian": r["executed"][0], "residual_only": r["executed"][1]},
            "rms_px": float(np.sqrt(res["err"] / (2.0 * pairs * P))),
            "roofline": {"bound": "mfma",
                         "kernel": "k_jac_fused12 (12-column tile, both cameras in one launch)" if how["kernel"] == "fused12"
                                   else "k_jac_fused (16-column tile, one launch per camera)",
                         "launch_ms": round(t_jac * 1e3, 4),
                         "launch_ms_source": "HIP events around each of the %d launches inside a run of %d steps" % (r["inloop_n"], steps)
                                             if r["inloop_n"] > 0 else "HIP events around 50 back-to-back launches",
                         "launch_ms_back_to_back": round(km["jac_view"], 4), "cameras_per_launch": cams,
                         "achieved": round(tf, 3), "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP64_TFLOPS, 4),
                         "algorithmic_flops_per_camera_point": FLOPS_STEREO_PER_CAMERA_POINT,
                         "algorithmic_bytes_per_launch": int(bytes_alg), "traffic": traffic, "traffic_source": traffic_source},
        }
    out["note"] = ("stereoCalibrate LM loop, flags FIX_ASPECT_RATIO+ZERO_TANGENT_DIST+USE_INTRINSIC_GUESS+SAME_FOCAL_LENGTH+"
                   "RATIONAL_MODEL+FIX_K3+FIX_K4+FIX_K5, one GPU, pairs resident in HBM; a point pair is one board point "
                   "seen by both cameras")
    return out


def config1_block(pkg, synth, dev, steps, warmup):
    """BASELINE configs[1]: 1000 noisy views of an asymmetric circle grid through calibrateCamera(flags = 0) -- the
    88-point 11x8 grid the config names and the reference's literal 4x11 grid of 44 points at spacing 2
    (monocular_circle.cpp:49-50, :35-37, :83-94).  LM loop from HBM (ms per iteration) and the whole call from host
    buffers at cv::calibrateCamera's default criteria."""
    out = {}
    for label, obj, depth in (("11x8 grid, 88 points", synth.asymmetric_circles(8, 11, 12.5), (400.0, 1200.0)),
                              ("reference 4x11 grid, 44 points", synth.asymmetric_circles(4, 11, 2.0), (40.0, 110.0))):
        ms = synth.make_views(obj, 1000, 0xC0FFEE + 2, depth=depth)
        cam = ms.cam
        K0 = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.98, cam.cy - 2], [0, 0, 1.0]])
        intr, mask, _ = pkg.mono_start(K0, cam.dist(5) * 0.8, 0)
        rt0 = synth.perturbed_poses(ms.rvec, ms.tvec, 2, rot_sigma=0.01, trans_sigma=0.003 * depth[1])
        s = pkg.Solver(obj, ms.img, device=dev)
        r = _timed_steps(s, dict(intr=intr, imask=mask, rt=rt0), steps, warmup)
        how = s.describe()
        s.close()
        pkg.calibrate_camera(obj, ms.img[:100], (640, 480), ndist=8)
        t0 = time.perf_counter()
        got = pkg.calibrate_camera(obj, ms.img, (640, 480), ndist=8)     # zeros(8,1), flags 0, default criteria
        whole = (time.perf_counter() - t0) * 1e3
        P = int(obj.shape[0])
        out[label] = {"views": 1000, "points_per_view": P, "steps": steps, "ms_per_step": round(r["ms"][-1], 4),
                      "ms_per_step_min": round(min(r["ms"]), 4), "ms_per_step_median": round(float(np.median(r["ms"])), 4),
                      "ms_per_step_max": round(max(r["ms"]), 4),
                      "points_per_sec": 1000 * P * r["stats"]["jac_passes"] / (r["ms"][-1] * steps * 1e-3),
                      "jacobian_kernel": how["kernel"], "jacobian_launch_ms_in_loop": round(r["inloop_ms"], 4),
                      "whole_call_ms": round(whole, 2), "whole_call_lm_loop_ms": round(got["device_ms"], 2),
                      "whole_call_iters": got["iters"], "rms_px": got["rms"]}
    out["note"] = ("1000 views is far too little work to fill the GPU (250 tiles of four views for 1024 SIMDs): these lines are "
                   "bound by launch and host latencies, not by the kernels")
    return out


def config4_share_block(pkg, synth, dev, warmup, steps=10):
    """One GPU's share of BASELINE configs[4]: 125 000 stereo pairs of a 17x12 board (204 points, so every view spans four
    chunks of the fused kernel), fx fy cx cy k1 k2 p1 p2 k3 free in both cameras."""
    obj = synth.chessboard(17, 12, 10.0)
    P, N, distinct = int(obj.shape[0]), 125_000, 25_000
    st = synth.make_stereo(obj, distinct, 0xC0FFEE + 5)
    rep = N // distinct
    img1, img2 = np.tile(st.img1, (rep, 1, 1)), np.tile(st.img2, (rep, 1, 1))
    rvec, tvec = np.tile(st.rvec, (rep, 1)), np.tile(st.tvec, (rep, 1))
    intr = np.zeros((2, 18))
    for c, cam in enumerate((st.cam1, st.cam2)):
        intr[c, :4] = [cam.fx * 0.98, cam.fy * 0.985, cam.cx + 2.5, cam.cy - 2.0]
        intr[c, 4:9] = cam.dist(5) * 0.8
    mask = np.zeros((2, 18), dtype=np.uint8)
    mask[:, :9] = 1
    rt0 = synth.perturbed_poses(rvec, tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    lr0 = np.r_[st.r_lr, st.t_lr] + np.array([1e-3, -1e-3, 5e-4, 0.5, -0.2, 0.3])
    s = pkg.Solver(obj, img1, img2, device=dev)
    r = _timed_steps(s, dict(intr=intr, imask=mask, rt=rt0, lr=lr0), steps, min(warmup, 2), repeats=3)
    res = s.get()
    how = s.describe()
    km = s.time_kernels(reps=20)
    s.close()
    t_jac = (r["inloop_ms"] if r["inloop_n"] > 0 else km["jac_view"]) * 1e-3
    tf = FLOPS_JAC_PASS * how["cams_per_launch"] * N * P / t_jac / 1e12
    return {"pairs": N, "points_per_view": P, "steps": steps, "ms_per_step": round(r["ms"][-1], 4),
            "ms_per_step_min": round(min(r["ms"]), 4), "ms_per_step_max": round(max(r["ms"]), 4),
            "point_pairs_per_sec": N * P * r["stats"]["jac_passes"] / (r["ms"][-1] * steps * 1e-3),
            "jac_passes": r["stats"]["jac_passes"], "err_passes": r["stats"]["err_passes"],
            "rms_px": float(np.sqrt(res["err"] / (2.0 * N * P))),
            "roofline": {"bound": "mfma", "kernel": "k_jac_fused (16-column tile, views in four chunks of 56 points, one launch per camera)",
                         "launch_ms": round(t_jac * 1e3, 4), "launch_ms_back_to_back": round(km["jac_view"], 4),
                         "cameras_per_launch": how["cams_per_launch"], "achieved": round(tf, 3), "peak": PEAK_FP64_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(tf / PEAK_FP64_TFLOPS, 4), "algorithmic_flops_per_camera_point": FLOPS_JAC_PASS,
                         "traffic": None},
            "data": "synthetic: %d distinct pairs, each taken %d times" % (distinct, rep)}


def whole_call(pkg, obj, ms):
    """cv::calibrateCamera's whole call from host buffers on this rank's views (not `value`: it includes the
    upload over PCIe, the initialisation kernels and the pose download; default criteria = 30 iterations)."""
    pkg.calibrate_camera(obj, ms.img[:2000], (640, 480))     # code objects warm
    t0 = time.perf_counter()
    got = pkg.calibrate_camera(obj, ms.img, (640, 480))
    dt = time.perf_counter() - t0
    pts = ms.img.shape[0] * ms.img.shape[1]
    return {"ms": round(dt * 1e3, 2), "lm_loop_ms": round(got["device_ms"], 2), "iters": got["iters"], "rms_px": got["rms"],
            "points_per_s": pts * got["jac_passes"] / dt,
            "note": "host buffers in, results out (ccalib_calibrate_camera_views: one pointer per view, nothing gathered on "
                    "the host): argument checks, upload through pinned staging, device initialisation, LM loop, download; "
                    "device blocks come from the library's pool (the warm-up call above left them there)"}


def opencv_check(pkg, synth):
    """If a real OpenCV is importable on this host, run cv2.calibrateCamera on BASELINE config 1 and report the
    deltas; otherwise say so (parity is then against the restated oracle only)."""
    try:
        import cv2
    except Exception as e:  # noqa: BLE001
        return {"available": False, "note": f"import cv2 failed ({type(e).__name__}): no OpenCV on this host, "
                                            f"results are checked against the restated oracle only"}
    try:
        obj = synth.chessboard(9, 6, 25.0)
        ms = synth.make_views(obj, 20, 0xC0FFEE + 1)
        objs = [obj.astype(np.float32)] * 20
        imgs = [m.astype(np.float32) for m in ms.img]
        rms, K, d, rv, tv = cv2.calibrateCamera(objs, imgs, (640, 480), None, None)
        got = pkg.calibrate_camera(obj, ms.img, (640, 480))
        return {"available": True, "version": cv2.__version__, "config": "configs[0]: 20 views, 9x6, flags 0",
                "rms_opencv": float(rms), "rms_ours": got["rms"], "rms_rel_delta": abs(got["rms"] - rms) / rms,
                "K_rel_delta": float(np.abs(got["K"] - K).max() / np.abs(K).max()),
                "dist_abs_delta": float(np.abs(got["dist"] - np.ravel(d)[:5]).max()),
                "rvec_abs_delta": float(np.abs(got["rvecs"] - np.array(rv).reshape(-1, 3)).max())}
    except Exception as e:  # noqa: BLE001
        return {"available": True, "error": f"{type(e).__name__}: {e}"}


def cpu_baseline(ge, synth, obj, nviews, seed, pkg):
    """The CPU oracle (Schur variant of the restated OpenCV algorithm) timed on this host, rank 0 only,
    on a bounded sample of the same workload: the first `nviews` views, 5 LM iterations."""
    orc = ge.load_oracle()
    ms = synth.make_views(obj, nviews, seed)
    cam = ms.cam
    K0 = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.98, cam.cy - 2], [0, 0, 1.0]])
    rt0 = synth.perturbed_poses(ms.rvec, ms.tvec, seed, rot_sigma=0.01, trans_sigma=3.0)
    ncores = min(16, os.cpu_count() or 1)
    res = {}
    for label, nth in (("1", 1), ("all", ncores)):
        t0 = time.perf_counter()
        r = orc.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=cam.dist(5) * 0.8, ndist=5,
                                 flags=orc.USE_INTRINSIC_GUESS, criteria=(3, 5, 0.0), init_rt=rt0, schur=True,
                                 nthreads=nth)
        dt = time.perf_counter() - t0
        res[label] = nviews * obj.shape[0] * r["jac_passes"] / dt
    # the same sample, start and iteration count through the HIP path: RMS and intrinsics against the oracle's
    g = pkg.calibrate_camera(obj, ms.img, (640, 480), K=K0, dist=cam.dist(5) * 0.8, ndist=5,
                             flags=pkg.CALIB_USE_INTRINSIC_GUESS, criteria=(3, 5, 0.0), init_rt=rt0)
    # The dense restatement (the full (18+6N)^2 normal matrix solved whole, as OpenCV 4.1.0 does), one core, at the
    # sizes it can reach: all of configs[0], and a 64-view sample of configs[1] (3 iterations; the oracle's Jacobi
    # SVD of the 402x402 system dominates -- cv::solve(DECOMP_SVD) is the reference's default there)
    dense = {}
    m0 = synth.make_views(obj, 20, 0xC0FFEE + 1)
    t0 = time.perf_counter()
    r0 = orc.calibrate_camera(obj, m0.img, (640, 480), schur=False)
    dt0 = time.perf_counter() - t0
    dense["configs[0]"] = {"points_per_s": 20 * obj.shape[0] * r0["jac_passes"] / dt0, "ms": round(dt0 * 1e3, 1),
                           "iters": r0["iters"], "unknowns": 18 + 6 * 20, "solve": "SVD"}
    grid = synth.asymmetric_circles(8, 11, 12.5)
    m1 = synth.make_views(grid, 100, 0xC0FFEE + 2)
    for name, nv, fl in (("configs[1] sample, SVD", 64, 0), ("configs[1] sample, LU", 100, orc.USE_LU)):
        t0 = time.perf_counter()
        r1 = orc.calibrate_camera(grid, m1.img[:nv], (640, 480), criteria=(3, 3, 0.0), schur=False, flags=fl)
        dt1 = time.perf_counter() - t0
        dense[name] = {"points_per_s": nv * grid.shape[0] * r1["jac_passes"] / dt1, "ms": round(dt1 * 1e3, 1),
                       "views": nv, "iters": r1["iters"], "unknowns": 18 + 6 * nv}
    return {"value": res["all"], "unit": "points/s", "cores": ncores, "kind": "port",
            "single_core_value": res["1"], "dense_one_core": dense,
            "rms_oracle": float(r["rms"]), "rms_hip": float(g["rms"]),
            "rms_rel_delta": abs(float(g["rms"]) - float(r["rms"])) / float(r["rms"]),
            "K_rel_delta": float(np.abs(g["K"] - r["K"]).max() / np.abs(r["K"]).max()),
            "sample": f"first {nviews} views of the same set, 5 LM iterations, Schur-eliminated restatement of the "
                      f"OpenCV 4.1.0 algorithm (oracle/), OpenMP over views; OpenCV itself is not installed and its "
                      f"dense (18+6N)^2 solve does not scale to this size"}


if __name__ == "__main__":
    main()
