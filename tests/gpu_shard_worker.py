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

    def rccl_id():
        ids = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            pkg.capi.check(pkg.capi.load().ccalib_rccl_unique_id(buf))
            ids = [buf.raw]
        dist.broadcast_object_list(ids, src=0)
        return ids[0]

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
        s.use_rccl(world, rank, rccl_id())
    s.set_start(intr, mask, rt0)
    crit = (3, 30, 1e-8)
    st = s.run(criteria=crit)
    out = s.get()
    s.close()
    whole = None
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

    # starting values on the device, sharded: the intrinsic seed's normal equations are summed over ranks, so
    # every rank gets the K a single solver over all views computes; poses are per view and need no exchange
    s = pkg.Solver(obj, ms.img, device=dev)
    if mode == "callback":
        s.use_callback(lambda a: dist.all_reduce(torch.from_numpy(a)))
    else:
        s.use_rccl(world, rank, rccl_id())
    Ksh = s.init_intrinsics((640, 480))
    rt_sh = s.init_poses(Ksh, None)
    s.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, Ksh)
    assert all(np.array_equal(g, gathered[0]) for g in gathered)      # same host arithmetic on the same sums
    if rank == 0:
        one = pkg.Solver(obj, whole.img, device=0)
        Kone = one.init_intrinsics((640, 480))
        rt_one = one.init_poses(Kone, None)
        one.close()
        assert np.allclose(Ksh, Kone, rtol=1e-12)
        assert np.abs(rt_sh - rt_one[:per_rank]).max() < 1e-6 * max(1.0, np.abs(rt_one).max())
        print("GPU_SHARDED_INIT_OK", mode)

    # stereo, sharded: each rank owns half of the pairs; the relative-pose seed is the exact median over the pairs
    # of all ranks (bisection with all-reduced counts), so every rank holds the single-solver seed bit for bit
    npairs = 120
    flags = (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_SAME_FOCAL_LENGTH |
             pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 | pkg.CALIB_FIX_K5)
    sw = synth.make_stereo(obj, npairs * world, 321)
    lo, hi = rank * npairs, (rank + 1) * npairs
    K1 = np.array([[790.0, 0, 319.5], [0, 790.0, 239.5], [0, 0, 1.0]])
    intr2 = np.zeros((2, 18))
    intr2[:, :4] = [790.0, 790.0, 319.5, 239.5]
    mask2 = np.zeros((2, 18), dtype=np.uint8)
    mask2[:, [1, 2, 3, 4, 5, 11]] = 1                                # fy cx cy k1 k2 k6, fx tied: the reference's flags
    def stereo_run(img1, img2, shard):
        so = pkg.Solver(obj, img1, img2, device=dev if shard else 0)
        if shard and mode == "callback":
            so.use_callback(lambda a: dist.all_reduce(torch.from_numpy(a)))
        elif shard:
            so.use_rccl(world, rank, rccl_id())
        so.init_poses(K1, None, cam=0, want=False)
        so.init_poses(K1, None, cam=1, want=False)
        lr = so.init_relative_pose()
        so.set_start(intr2, mask2, None, aspect=[1.0, 1.0], same_focal=True, lr=lr)
        sst = so.run(criteria=(3, 40, 1e-7))
        res = so.get()
        so.close()
        return sst, res, lr
    st_s, out_s, lr_s = stereo_run(sw.img1[lo:hi], sw.img2[lo:hi], True)
    seeds = [None] * world
    dist.all_gather_object(seeds, lr_s)
    assert all(np.array_equal(g, seeds[0]) for g in seeds)            # every rank computed the same bits
    if rank == 0:
        st_1, ref_s, lr_1 = stereo_run(sw.img1, sw.img2, False)        # one solver over all pairs, its own seed
        assert np.array_equal(lr_s, lr_1), (lr_s, lr_1)               # sharded seed == single-solver seed, bit for bit
        assert abs(st_s["iters"] - st_1["iters"]) <= 1, (st_s, st_1)
        assert abs(out_s["err"] - ref_s["err"]) <= 1e-8 * ref_s["err"]
        assert np.allclose(out_s["intr"], ref_s["intr"], rtol=1e-6, atol=1e-10)
        assert np.allclose(out_s["lr"], ref_s["lr"], rtol=1e-6, atol=1e-8)
        assert np.abs(out_s["rt"] - ref_s["rt"][:npairs]).max() < 1e-5
        print("GPU_SHARDED_STEREO_OK", mode, st_s)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
