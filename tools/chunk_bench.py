#!/usr/bin/env python3
"""The fused Jacobian kernel on boards of more than 64 points (several chunks per view), back to back:
   python tools/chunk_bench.py  ->  one JSON line per board: ms per launch, HIP events on the solver's stream."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    pkg = ge.load_package()
    synth = pkg.synth
    for cols, rows, pitch, views, rep in ((17, 12, 10.0, 10_000, 5), (11, 8, 20.0, 20_000, 5)):
        obj = synth.chessboard(cols, rows, pitch)
        ms = synth.make_views(obj, views, 0xC0FFEE + 9)
        img = np.ascontiguousarray(np.tile(ms.img, (rep, 1, 1)))
        rt = np.tile(np.concatenate([ms.rvec, ms.tvec], axis=1), (rep, 1))
        s = pkg.Solver(obj, img)
        intr, mask, _ = pkg.mono_start(ms.cam.K(), ms.cam.dist(5), 0)
        s.set_start(intr, mask, rt)
        s.time_kernels(reps=20)
        km = [s.time_kernels(reps=40)["jac_view"] for _ in range(3)]
        how = s.describe()
        s.close()
        print(json.dumps({"board": [cols, rows], "points": int(obj.shape[0]), "views": views * rep, "kernel": how["kernel"],
                          "jac_ms": [round(x, 4) for x in km]}))


if __name__ == "__main__":
    main()
