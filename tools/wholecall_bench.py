#!/usr/bin/env python3
"""Times the whole calibrateCamera call (pose initialisation + LM + download) from host buffers:
   python tools/wholecall_bench.py [views]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    views = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    pkg = ge.load_package()
    synth = pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, views, 0xC0FFEE + 4)
    for rep in range(2):
        t0 = time.perf_counter()
        got = pkg.calibrate_camera(obj, ms.img, (640, 480), criteria=(3, 30, 2.2e-16))
        dt = time.perf_counter() - t0
    print(json.dumps({"views": views, "whole_call_s": round(dt, 4), "lm_loop_ms": round(got["device_ms"], 3), "iters": got["iters"],
                      "rms": got["rms"]}))


if __name__ == "__main__":
    main()
