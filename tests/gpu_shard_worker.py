"""One rank of the GPU sharding test: both ranks use GPU 0 (the test box has one GPU), each owns half of
the views, and every reduction of the LM loop is summed over ranks -- through a gloo callback
(mode "callback") or in-stream RCCL (mode "rccl", needs one GPU per rank)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    mode = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = ge.load_package()
    synth = pkg.synth
    obj = synth.chessboard(9, 6, 25.0)
    per_rank, seed = 500, 123
    ms = synth.make_views(obj, per_rank, seed, first_view=rank * per_rank)
    cam = ms.cam
    K0 = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.98, cam.cy - 2], [0, 0, 1.0]])
    intr, mask, _ = pkg.mono_start(K0, cam.dist(5) * 0.8, 0)
    rt0 = synth.perturbed_poses(ms.rvec, ms.tvec, seed, first_view=rank * per_rank, rot_sigma=0.01, trans_sigma=3.0)
    dev = 0 if mode == "callback" else rank % max(1, torch.cuda.device_count())
    s = pkg.Solver(obj, ms.img, device=dev)
    if mode == "callback":
        s.use_callback(lambda a: dist.all_reduce(torch.from_numpy(a)))
    else:
        ids = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            pkg.capi.check(pkg.capi.load().ccalib_rccl_unique_id(buf))
            ids = [buf.raw]
        dist.broadcast_object_list(ids, src=0)
        s.use_rccl(world, rank, ids[0])
    s.set_start(intr, mask, rt0)
    crit = (3, 30, 1e-8)
    st = s.run(criteria=crit)
    out = s.get()
    s.close()
    if rank == 0:
        whole = synth.make_views(obj, per_rank * world, seed)
        rt_all = synth.perturbed_poses(whole.rvec, whole.tvec, seed, rot_sigma=0.01, trans_sigma=3.0)
        one = pkg.Solver(obj, whole.img, device=0)
        one.set_start(intr, mask, rt_all)
        st1 = one.run(criteria=crit)
        ref = one.get()
        one.close()
        assert abs(st["iters"] - st1["iters"]) <= 1, (st, st1)   # the sums are taken in a different order
        assert abs(out["err"] - ref["err"]) <= 1e-9 * ref["err"]
        assert np.allclose(out["intr"], ref["intr"], rtol=1e-7, atol=1e-12)
        assert np.abs(out["rt"] - ref["rt"][:per_rank]).max() < 1e-6
        print("GPU_SHARDED_OK", mode, st)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
