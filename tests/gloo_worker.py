"""One rank of the CPU sharding test (launched by tests/test_host_cpu.py through torch.distributed.run).

Each rank regenerates ITS OWN views from the shard-reproducible generator, runs the product's LM
driver over a CPU backend whose sums go through gloo all_reduce, and rank 0 checks the result
against the oracle solving the whole problem in one process.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from tests.cpu_backend import OracleBackend, run_lm  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg, orc = ge.load_package(), ge.load_oracle()
    synth = pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    per_rank, seed = 9, 77
    ms = synth.make_views(obj, per_rank, seed, first_view=rank * per_rank)
    cam = ms.cam
    K0 = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.98, cam.cy - 2], [0, 0, 1.0]])
    rt0 = synth.perturbed_poses(ms.rvec, ms.tvec, seed, first_view=rank * per_rank, rot_sigma=0.01, trans_sigma=3.0)
    intr, mask, aspect = pkg.mono_start(K0, cam.dist(5) * 0.8, 0)

    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t)

    be = OracleBackend(pkg, orc, obj, ms.img, rt0, mask, aspect, allreduce)
    crit = (3, 30, 1e-9)
    got_intr, got_rt, st = run_lm(pkg, be, intr, crit)
    # every rank must have walked the same state machine
    chk = torch.tensor([st["iters"], st["jac_passes"], st["err_passes"]], dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi), (lo, hi)
    if rank == 0:
        whole = synth.make_views(obj, per_rank * world, seed)
        rt_all = synth.perturbed_poses(whole.rvec, whole.tvec, seed, rot_sigma=0.01, trans_sigma=3.0)
        assert np.array_equal(whole.img[:per_rank], ms.img)            # shard-reproducible generator
        ref = orc.calibrate_camera(obj, whole.img, (640, 480), K=K0, dist=cam.dist(5) * 0.8, ndist=5,
                                   flags=orc.USE_INTRINSIC_GUESS, criteria=crit, init_rt=rt_all, schur=False)
        assert st["iters"] == ref["iters"], (st, ref["iters"])
        assert abs(st["final_err"] - ref["final_err"]) <= 1e-9 * ref["final_err"]
        assert np.allclose(got_intr[:4], [ref["K"][0, 0], ref["K"][1, 1], ref["K"][0, 2], ref["K"][1, 2]], rtol=1e-8)
        assert np.abs(got_rt[:, :3] - ref["rvecs"][:per_rank]).max() < 1e-8
        print("SHARDED_OK", st)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
