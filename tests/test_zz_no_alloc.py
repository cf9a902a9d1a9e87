"""ccalib_reprojection_error_noalloc (the entry point that allocates nothing) against ccalib_reprojection_error.

Both run the same projection kernel with the same launch geometry, so projected pixels and per-view RMS must agree
bit for bit; the global RMS is summed in a different order (tree on the device, sequential on the host) and is
compared to 1e-13 relative.  Two calls in a row share the GPU's static scratch array, the second with a smaller
input, so stale contents of the first call would show.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ragged(synth, nviews, seed, lo=8, hi=55):
    rng = np.random.default_rng(seed)
    base = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(base, nviews, seed)
    objs, imgs = [], []
    for i in range(nviews):
        keep = np.sort(rng.choice(54, size=int(rng.integers(lo, hi)), replace=False))
        objs.append(base[keep])
        imgs.append(ms.img[i][keep])
    return ms, objs, imgs


def _call_noalloc(pkg, torch, objs, imgs, K, dist, rvecs, tvecs, max_npts=None, capacity=None, npts_override=None,
                  stream=None, want=("proj", "pv", "tot")):
    lib = pkg.capi.load()
    npts = np.array([len(v) for v in imgs], dtype=np.int32)
    total = int(npts.sum())
    obj = np.concatenate(objs).astype(np.float32)
    img = np.concatenate(imgs).astype(np.float32)
    dev = torch.device("cuda", 0)
    d_npts = torch.from_numpy(npts if npts_override is None else npts_override).to(dev)
    d_obj, d_img = torch.from_numpy(obj).to(dev), torch.from_numpy(img).to(dev)
    d_rv = torch.from_numpy(np.ascontiguousarray(rvecs, dtype=np.float64)).to(dev)
    d_tv = torch.from_numpy(np.ascontiguousarray(tvecs, dtype=np.float64)).to(dev)
    d_proj = torch.full((total, 2), -7.0, dtype=torch.float32, device=dev)
    d_pv = torch.zeros(len(npts), dtype=torch.float64, device=dev)
    d_tot = torch.zeros(1, dtype=torch.float64, device=dev)
    Kh = np.ascontiguousarray(K, dtype=np.float64).reshape(-1)
    dh = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    torch.cuda.synchronize()
    st = stream.cuda_stream if stream is not None else None
    rc = lib.ccalib_reprojection_error_noalloc(
        C.c_int(len(npts)), C.c_void_p(d_npts.data_ptr()), C.c_int(int(npts.max()) if max_npts is None else max_npts),
        C.c_int(total if capacity is None else capacity), C.c_void_p(d_obj.data_ptr()), C.c_void_p(d_img.data_ptr()),
        Kh.ctypes.data_as(C.c_void_p), dh.ctypes.data_as(C.c_void_p), C.c_int(dh.size), C.c_void_p(d_rv.data_ptr()),
        C.c_void_p(d_tv.data_ptr()), C.c_void_p(d_proj.data_ptr() if "proj" in want else None),
        C.c_void_p(d_pv.data_ptr() if "pv" in want else None), C.c_void_p(d_tot.data_ptr() if "tot" in want else None),
        C.c_void_p(st))
    pkg.capi.check(rc)
    torch.cuda.synchronize()
    return float(d_tot.cpu()[0]), d_pv.cpu().numpy(), d_proj.cpu().numpy()


def test_noalloc_matches_allocating_entry_point_twice(pkg, synth):
    import torch
    for nviews, seed, dist in ((700, 11, None), (90, 12, "rational")):     # second call: smaller input, same scratch
        ms, objs, imgs = _ragged(synth, nviews, seed)
        cam = ms.cam
        d = cam.dist(5) if dist is None else np.concatenate([cam.dist(5), [0.01, -0.02, 0.005]])
        rms, pv, proj = pkg.reprojection_error(objs, imgs, cam.K(), d, ms.rvec, ms.tvec, want_points=True)
        rms2, pv2, proj2 = _call_noalloc(pkg, torch, objs, imgs, cam.K(), d, ms.rvec, ms.tvec)
        assert np.array_equal(proj, proj2)                       # same kernel, same geometry: bit for bit
        assert np.array_equal(pv, pv2)
        assert abs(rms - rms2) <= 1e-13 * rms
    # on a side stream, per-view output only needed: same numbers again
    s = torch.cuda.Stream()
    rms3, pv3, proj3 = _call_noalloc(pkg, torch, objs, imgs, cam.K(), d, ms.rvec, ms.tvec, stream=s)
    assert np.array_equal(pv3, pv2) and np.array_equal(proj3, proj2) and rms3 == rms2


def test_noalloc_rejects_what_it_cannot_hold(pkg, synth):
    import torch
    ms, objs, imgs = _ragged(synth, 40, 5)
    cam = ms.cam
    args = (objs, imgs, cam.K(), cam.dist(5), ms.rvec, ms.tvec)
    # a point count above max_npts is found on the device: NaN out, the first pixel marked, the other projections untouched
    rms, pv, proj = _call_noalloc(pkg, torch, *args, max_npts=20)
    assert np.isnan(rms) and np.isnan(pv).all() and np.isnan(proj[0]).all() and (proj[1:] == -7.0).all()
    # a total beyond the stated capacity likewise
    rms, pv, proj = _call_noalloc(pkg, torch, *args, capacity=100)
    assert np.isnan(rms) and np.isnan(pv).all() and np.isnan(proj[0]).all() and (proj[1:] == -7.0).all()
    # a caller that asks for projections only still learns of it: the first pixel reads NaN
    _, _, proj = _call_noalloc(pkg, torch, *args, capacity=100, want=("proj",))
    assert np.isnan(proj[0]).all() and (proj[1:] == -7.0).all()
    _, _, proj = _call_noalloc(pkg, torch, *args, want=("proj",))
    _, _, proj_want = pkg.reprojection_error(*args, want_points=True)
    assert np.array_equal(proj, proj_want)
    # and the next good call is not affected by the flag the bad one left in the scratch array
    rms, pv, proj = _call_noalloc(pkg, torch, *args)
    want, pv_want = pkg.reprojection_error(*args)
    assert np.array_equal(pv, pv_want) and abs(rms - want) <= 1e-13 * want
    # more views than the static buffer holds: refused on the host with a clear message
    lib = pkg.capi.load()
    one = C.c_void_p(8)
    K = np.eye(3)
    rc = lib.ccalib_reprojection_error_noalloc(C.c_int(65537), one, C.c_int(54), C.c_int(54 * 65537), one, one,
                                               K.ctypes.data_as(C.c_void_p), None, C.c_int(0), one, one, one, one, one, None)
    assert rc == pkg.capi.ERR_ARG
    assert b"65536" in lib.ccalib_last_error()


def test_noalloc_under_graph_capture(pkg, synth):
    """What the entry point is for: after one ordinary call on the device (which fetches the scratch array's address),
    a call made while a stream is capturing records three kernel launches and nothing else; replaying the graph gives
    the eager call's numbers, and again after the inputs behind the device pointers have changed."""
    import torch
    lib = pkg.capi.load()
    ms, objs, imgs = _ragged(synth, 300, 21)
    cam = ms.cam
    want_rms, want_pv, want_proj = _call_noalloc(pkg, torch, objs, imgs, cam.K(), cam.dist(5), ms.rvec, ms.tvec)   # eager, warms up
    npts = np.array([len(v) for v in imgs], dtype=np.int32)
    total = int(npts.sum())
    dev = torch.device("cuda", 0)
    d_npts = torch.from_numpy(npts).to(dev)
    d_obj = torch.from_numpy(np.concatenate(objs).astype(np.float32)).to(dev)
    d_img = torch.from_numpy(np.concatenate(imgs).astype(np.float32)).to(dev)
    d_rv, d_tv = torch.from_numpy(ms.rvec.copy()).to(dev), torch.from_numpy(ms.tvec.copy()).to(dev)
    d_proj = torch.zeros((total, 2), dtype=torch.float32, device=dev)
    d_pv = torch.zeros(len(npts), dtype=torch.float64, device=dev)
    d_tot = torch.zeros(1, dtype=torch.float64, device=dev)
    Kh = np.ascontiguousarray(cam.K(), dtype=np.float64).reshape(-1)
    dh = np.ascontiguousarray(cam.dist(5), dtype=np.float64)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        rc = lib.ccalib_reprojection_error_noalloc(
            C.c_int(len(npts)), C.c_void_p(d_npts.data_ptr()), C.c_int(int(npts.max())), C.c_int(total),
            C.c_void_p(d_obj.data_ptr()), C.c_void_p(d_img.data_ptr()), Kh.ctypes.data_as(C.c_void_p),
            dh.ctypes.data_as(C.c_void_p), C.c_int(dh.size), C.c_void_p(d_rv.data_ptr()), C.c_void_p(d_tv.data_ptr()),
            C.c_void_p(d_proj.data_ptr()), C.c_void_p(d_pv.data_ptr()), C.c_void_p(d_tot.data_ptr()),
            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    pkg.capi.check(rc)
    assert float(d_tot.cpu()[0]) == 0.0                  # captured, not executed
    g.replay()
    torch.cuda.synchronize()
    assert float(d_tot.cpu()[0]) == want_rms and np.array_equal(d_pv.cpu().numpy(), want_pv)
    assert np.array_equal(d_proj.cpu().numpy(), want_proj)
    # new poses behind the same pointers: the replay reads them
    tv2 = ms.tvec.copy()
    tv2[:, 2] *= 1.01
    d_tv.copy_(torch.from_numpy(tv2))
    g.replay()
    torch.cuda.synchronize()
    rms2, pv2, proj2 = _call_noalloc(pkg, torch, objs, imgs, cam.K(), cam.dist(5), ms.rvec, tv2)
    assert float(d_tot.cpu()[0]) == rms2 and np.array_equal(d_pv.cpu().numpy(), pv2) and np.array_equal(d_proj.cpu().numpy(), proj2)
