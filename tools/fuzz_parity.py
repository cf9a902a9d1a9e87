#!/usr/bin/env python3
"""Randomised comparison of the HIP path with the CPU oracle (Schur form) over shapes no fixed test names: board
sizes from 3x3 to 14x10 (more than 64 points: several chunks per view), 3 to 600 views, ragged views (points
dropped at random), cameras, noise levels, flag words, with and without an intrinsic guess; mono and stereo.
   python tools/fuzz_parity.py [cases] [seed] [out.json]
   python tools/fuzz_parity.py only <case> [seed]      one case again, with the LM records of both sides
Prints one line per case that leaves the tolerances of the parity tests (RMS, K 1e-6 relative; poses 1e-6; the
iteration count +-1) and a summary of the worst deviations.  The oracle is the checker here, never the product."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    only = None
    if len(sys.argv) > 2 and sys.argv[1] == "only":
        only = int(sys.argv[2])
        sys.argv[1:3] = [str(only + 1)]
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    summary = run(cases, seed, only)
    print(json.dumps({k: v for k, v in summary.items() if k != "outliers"}))
    if out_path:
        json.dump(summary, open(out_path, "w"), indent=1)


def run(cases, seed, only=None, quiet=False):
    """Cases 0..cases-1 of stream `seed` (or just case `only`); returns the summary dict."""
    cc, orc = ge.load_package(), ge.load_oracle()
    synth = cc.synth
    worst = {"mono": {}, "stereo": {}}
    bad, skipped, done = [], 0, {"mono": 0, "stereo": 0}
    iters_at_eps = {}
    unexplained = 0
    t_start = time.time()
    for case in range(0 if only is None else only, cases):
        rng = np.random.default_rng([seed, case])   # every case has its own stream: any one can be run again alone
        stereo = rng.random() < 0.35
        cols, rows = int(rng.integers(3, 15)), int(rng.integers(3, 11))
        if stereo and cols * rows > 64:
            cols, rows = 9, 6 + int(rng.integers(0, 2))
        pitch = float(rng.uniform(8.0, 30.0))
        obj = synth.chessboard(cols, rows, pitch) if rng.random() < 0.8 else synth.asymmetric_circles(min(cols, 6), rows, pitch)
        n = int(np.exp(rng.uniform(np.log(3), np.log(300 if stereo else 600))))
        noise = float(rng.choice([0.0, 0.1, 0.3, 0.8]))
        cam = synth.Camera(fx=float(rng.uniform(600, 1100)), fy=0.0, cx=float(rng.uniform(300, 340)), cy=float(rng.uniform(225, 255)),
                           k1=float(rng.uniform(-0.3, 0.1)), k2=float(rng.uniform(-0.1, 0.15)), p1=float(rng.uniform(-1e-3, 1e-3)),
                           p2=float(rng.uniform(-1e-3, 1e-3)), k3=float(rng.uniform(-0.02, 0.02)))
        cam.fy = cam.fx * float(rng.uniform(0.97, 1.03))
        extent = pitch * max(cols, rows)
        depth = (max(250.0, 1.6 * extent * cam.fx / 480.0), max(250.0, 1.6 * extent * cam.fx / 480.0) * 2.5)
        size = (640, 480)
        flags = 0
        guess = rng.random() < 0.5
        for f in (cc.CALIB_ZERO_TANGENT_DIST, cc.CALIB_FIX_K3, cc.CALIB_FIX_K2, cc.CALIB_FIX_PRINCIPAL_POINT, cc.CALIB_USE_LU,
                  cc.CALIB_FIX_ASPECT_RATIO):
            if rng.random() < 0.25:
                flags |= f
        nd = 5
        if rng.random() < 0.2:
            flags |= cc.CALIB_RATIONAL_MODEL | cc.CALIB_FIX_K4 | cc.CALIB_FIX_K5
            nd = 8
            if rng.random() < 0.5:
                flags |= cc.CALIB_FIX_K6
        if flags & (cc.CALIB_FIX_PRINCIPAL_POINT | cc.CALIB_FIX_ASPECT_RATIO):
            guess = True
        crit = (3, 30, 2.220446049250313e-16) if rng.random() < 0.4 else (3, int(rng.integers(8, 40)), float(10.0 ** rng.uniform(-10, -6)))
        try:
            if not stereo:
                ms = synth.make_views(obj, n, 5000 + case, cam=cam, noise=noise, depth=depth)
                imgs = ms.img
                objs = obj
                if rng.random() < 0.3:
                    # ragged: every view keeps a random subset of at least 8 points and 60 % of the board (fewer can leave
                    # seven points on one line and one beside it: no homography, and any two implementations part ways)
                    P = obj.shape[0]
                    lo = min(P, max(8, (6 * P + 9) // 10))
                    keep = [np.sort(rng.choice(P, size=int(rng.integers(lo, P + 1)), replace=False)) for _ in range(n)]
                    objs = [np.ascontiguousarray(obj[k]) for k in keep]
                    imgs = [np.ascontiguousarray(ms.img[i][k]) for i, k in enumerate(keep)]
                kw = {}
                if guess:
                    flags |= cc.CALIB_USE_INTRINSIC_GUESS
                    kw = dict(K=np.array([[cam.fx * 1.03, 0, cam.cx - 3], [0, cam.fy * 1.03, cam.cy + 2], [0, 0, 1.0]]),
                              dist=np.concatenate([cam.dist(5) * 0.8, np.zeros(nd - 5)]))
                t0 = time.time()
                got = cc.calibrate_camera(objs, imgs, size, ndist=nd, flags=flags, criteria=crit, **kw)
                got_trace = cc.last_lm_trace()
                ref = orc.calibrate_camera(objs, imgs, size, ndist=nd, flags=flags, criteria=crit, schur=True, nthreads=8, **kw)
                dev = dict(rms=abs(got["rms"] - ref["rms"]) / ref["rms"], K=rel(got["K"], ref["K"]),
                           dist=float(np.abs(got["dist"] - ref["dist"]).max()), rvec=float(np.abs(got["rvecs"] - ref["rvecs"]).max()),
                           tvec=rel(got["tvecs"], ref["tvecs"]), iters=abs(got["iters"] - ref["iters"]))
                kind = "mono"
                again = lambda **o: orc.calibrate_camera(objs, imgs, size, ndist=nd, flags=flags, criteria=crit, **o, **kw)
                keys = ("K", "rvecs")
            else:
                st = synth.make_stereo(obj, n, 7000 + case, noise=noise, cam1=cam,
                                       cam2=synth.Camera(fx=cam.fx * 1.01, fy=cam.fy * 1.01, cx=cam.cx + 2, cy=cam.cy - 1, k1=cam.k1,
                                                         k2=cam.k2, p1=cam.p1, p2=cam.p2, k3=cam.k3),
                                       depth=(depth[0] * 1.2, depth[1] * 1.2))
                K1 = cc.init_camera_matrix_2d(obj, st.img1, size)
                K2 = cc.init_camera_matrix_2d(obj, st.img2, size)
                sflags = cc.CALIB_USE_INTRINSIC_GUESS | (flags & (cc.CALIB_ZERO_TANGENT_DIST | cc.CALIB_FIX_K3 | cc.CALIB_FIX_ASPECT_RATIO))
                if rng.random() < 0.4:   # the reference's set (binocular_chessboard.cpp:156-161)
                    sflags = (cc.CALIB_FIX_ASPECT_RATIO | cc.CALIB_ZERO_TANGENT_DIST | cc.CALIB_USE_INTRINSIC_GUESS |
                              cc.CALIB_SAME_FOCAL_LENGTH | cc.CALIB_RATIONAL_MODEL | cc.CALIB_FIX_K3 | cc.CALIB_FIX_K4 | cc.CALIB_FIX_K5)
                scrit = (3, int(rng.integers(10, 100)), float(10.0 ** rng.uniform(-8, -5)))
                got = cc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, size, sflags, criteria=scrit)
                got_trace = cc.last_lm_trace()
                ref = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, size, sflags, criteria=scrit, schur=True, nthreads=8)
                dev = dict(rms=abs(got["rms"] - ref["rms"]) / ref["rms"], K=max(rel(got["K1"], ref["K1"]), rel(got["K2"], ref["K2"])),
                           dist=float(max(np.abs(got["d1"] - ref["d1"]).max(), np.abs(got["d2"] - ref["d2"]).max())),
                           rvec=float(np.abs(got["rt"] - ref["rt"])[:, :3].max()), tvec=max(rel(got["T"], ref["T"]), rel(got["R"], ref["R"])),
                           iters=abs(got["iters"] - ref["iters"]))
                kind, flags, crit = "stereo", sflags, scrit
                again = lambda **o: orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, size, sflags, criteria=scrit, **o)
                keys = ("K1", "rt")
        except (RuntimeError, cc.CcalibError) as e:
            skipped += 1
            print("case %d skipped: %s" % (case, str(e)[:120]), flush=True)
            continue
        done[kind] += 1
        if only is not None:
            ref_trace = again(schur=True, nthreads=8, trace=True)["trace"]
            np.set_printoptions(linewidth=200, precision=12)
            print(dict(kind=kind, board=[cols, rows], views=n, noise=noise, flags=hex(int(flags)), guess=bool(guess), crit=crit))
            print("dev", dev)
            if kind == "mono":
                # Both sides again from ONE starting point (the oracle's intrinsic seed and planar poses): what is left
                # of the deviation then is the LM loop's, the rest came from the two initialisations
                per_obj = objs if isinstance(objs, list) else [obj] * n
                per_img = imgs if isinstance(imgs, list) else [imgs[v] for v in range(n)]
                K0 = kw["K"] if guess else orc.init_camera_matrix_2d(objs, imgs, size)
                d0 = np.zeros(14)
                if guess:
                    d0[:nd] = kw["dist"]
                rt0 = np.zeros((n, 6))
                for v in range(n):
                    o = orc.find_extrinsic(per_obj[v], per_img[v], K0, d0)
                    rt0[v] = np.concatenate(o) if o is not None else 0.
                if not guess:
                    Kh = cc.init_camera_matrix_2d(objs, imgs, size)
                    print("intrinsic seed: HIP fx fy", Kh[0, 0], Kh[1, 1], " oracle", K0[0, 0], K0[1, 1], " relative", rel(Kh, K0))
                if True:
                    sol = cc.Solver(objs, imgs)
                    rth = sol.init_poses(K0, d0)
                    sol.close()
                    dv = np.maximum(np.abs(rth[:, :3] - rt0[:, :3]).max(1),
                                    np.abs(rth[:, 3:] - rt0[:, 3:]).max(1) / np.maximum(1., np.abs(rt0[:, 3:]).max(1)))
                    w = int(np.argmax(dv))
                    print("planar poses from the same intrinsics: largest difference %.3g (view %d), median %.3g" % (dv[w], w, np.median(dv)))
                    print("   HIP   ", rth[w], "\n   oracle", rt0[w])
                    if dv[w] > 1e-3:
                        print("   view %d: %d points" % (w, len(per_img[w])))
                        print("   object points", np.asarray(per_obj[w]).tolist())
                        print("   image points", np.asarray(per_img[w]).tolist())
                        print("   K", np.asarray(K0).tolist(), "dist", d0.tolist())
                f2 = flags | cc.CALIB_USE_INTRINSIC_GUESS
                g2 = cc.calibrate_camera(objs, imgs, size, K=K0, dist=d0[:nd], ndist=nd, flags=f2, criteria=crit, init_rt=rt0)
                r2 = orc.calibrate_camera(objs, imgs, size, K=K0, dist=d0[:nd], ndist=nd, flags=f2, criteria=crit, init_rt=rt0,
                                          schur=True, nthreads=8)
                print("from one starting point: rms", abs(g2["rms"] - r2["rms"]) / r2["rms"], "K", rel(g2["K"], r2["K"]), "rvec",
                      float(np.abs(g2["rvecs"] - r2["rvecs"]).max()), "iters", g2["iters"], r2["iters"])
            if kind == "stereo":
                # the same for a rig: both sides from the device's planar poses and median relative pose
                sol = cc.Solver(obj, st.img1, st.img2)
                rt0 = sol.init_poses(K1, None, cam=0)
                sol.init_poses(K2, None, cam=1, want=False)
                lr0 = sol.init_relative_pose()
                sol.close()
                g2 = cc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, size, sflags, criteria=scrit, init_rt=rt0, init_lr=lr0)
                r2 = orc.stereo_calibrate(obj, st.img1, st.img2, K1, None, K2, None, size, sflags, criteria=scrit, init_rt=rt0,
                                          init_lr=lr0, schur=True, nthreads=8)
                print("from one starting point: rms", abs(g2["rms"] - r2["rms"]) / r2["rms"], "K", max(rel(g2["K1"], r2["K1"]), rel(g2["K2"], r2["K2"])),
                      "rvec", float(np.abs(g2["rt"] - r2["rt"])[:, :3].max()), "T", rel(g2["T"], r2["T"]), "iters", g2["iters"], r2["iters"])
            print("iteration, first exponent, accepted exponent, trials, sum before, sum after:  HIP | oracle")
            for i in range(max(len(got_trace), len(ref_trace))):
                a = got_trace[i, :6] if i < len(got_trace) else None
                b = ref_trace[i, :6] if i < len(ref_trace) else None
                print(i, a, "|", b)
        for k, v in dev.items():
            if v > worst[kind].get(k, (-1,))[0]:
                worst[kind][k] = (float(v), case)
        # with eps = DBL_EPSILON a run ends on its iteration limit or on an exactly repeated parameter vector: the count is
        # not an invariant of the algorithm there (the oracle's own dense and Schur forms differ in it), the values are
        at_eps = crit[2] < 1e-15
        if at_eps:
            iters_at_eps[int(dev["iters"])] = iters_at_eps.get(int(dev["iters"]), 0) + 1
        out = dev["rms"] > 1e-6 or dev["K"] > 1e-6 or dev["rvec"] > 1e-6 or (dev["iters"] > 1 and not at_eps)
        if out:
            rec = dict(case=case, kind=kind, board=[cols, rows], points=int(obj.shape[0]), views=n, noise=noise, flags=hex(int(flags)),
                       guess=bool(guess), crit=[int(crit[1]), float(crit[2])], iters=[int(got["iters"]), int(ref["iters"])],
                       rms_px=float(ref["rms"]), dev={k: float(v) for k, v in dev.items()})
            # the yardstick: how far the oracle moves when only its own summation order (one thread instead of eight) or
            # its own solver (the dense system instead of the Schur form; up to 40 views) changes
            alt = again(schur=True, nthreads=1)
            rec["oracle_1_thread_vs_8"] = dict(rms=abs(alt["rms"] - ref["rms"]) / ref["rms"], K=rel(alt[keys[0]], ref[keys[0]]),
                                               poses=float(np.abs(alt[keys[1]] - ref[keys[1]]).max()), iters=abs(alt["iters"] - ref["iters"]))
            if n <= 40:   # (the dense system is (6 n + 18)^2, solved by SVD every iteration)
                alt = again(schur=False)
                rec["oracle_dense_vs_schur"] = dict(rms=abs(alt["rms"] - ref["rms"]) / ref["rms"], K=rel(alt[keys[0]], ref[keys[0]]),
                                                    poses=float(np.abs(alt[keys[1]] - ref[keys[1]]).max()), iters=abs(alt["iters"] - ref["iters"]))
            yard = [rec[k] for k in ("oracle_1_thread_vs_8", "oracle_dense_vs_schur") if k in rec]
            rec["explained_by_oracle_spread"] = bool(any(
                y["rms"] * 10 >= dev["rms"] and y["K"] * 10 >= dev["K"] and y["poses"] * 10 >= dev["rvec"] and
                (y["iters"] >= 1 or dev["iters"] <= 1 or at_eps) for y in yard))
            unexplained += not rec["explained_by_oracle_spread"]
            bad.append(rec)
            if not quiet:
                print(json.dumps(rec), flush=True)
        if case % 250 == 249 and not quiet:
            print("... %d cases, %.0f s" % (case + 1, time.time() - t_start), flush=True)
    summary = dict(cases=cases, seed=seed, done=done, skipped=skipped, outside_tolerance=len(bad), not_explained_by_oracle_spread=unexplained,
                   worst=worst, iteration_count_difference_at_dbl_epsilon=dict(sorted(iters_at_eps.items())), outliers=bad)
    return summary


if __name__ == "__main__":
    main()
