#!/usr/bin/env python3
"""Times the per-point kernels alone (HIP events) on one GPU: python tools/kernel_bench.py [views] [reps]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    views = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    pkg = ge.load_package(); synth = pkg.synth
    bw, bh = (int(x) for x in os.environ.get('BOARD', '9x6').split('x'))
    obj = synth.chessboard(bw, bh, 25.0)
    ms = synth.make_views(obj, views, 0xC0FFEE + 4)
    cam = ms.cam
    intr, mask, _ = pkg.mono_start(cam.K(), cam.dist(5), 0)
    s = pkg.Solver(obj, ms.img)
    s.set_start(intr, mask, np.c_[ms.rvec, ms.tvec])
    km = s.time_kernels(reps=reps)
    pts = views * len(obj)
    print(json.dumps({"G": os.environ.get("CCALIB_G"), "grid": os.environ.get("CCALIB_GRID"), **{k: round(v, 4) for k, v in km.items()},
                      "jac_pts_per_s": pts / ((km["jac_view"] + km["jac_shared"]) * 1e-3)}))
    s.close()

if __name__ == "__main__":
    main()
