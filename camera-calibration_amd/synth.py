"""Synthetic calibration data: planar boards seen by a distorted pinhole camera.

Deterministic and shard-reproducible: every random draw is a pure function of
(seed, view index, draw index) through a counter-based 64-bit mixer, so any
rank can regenerate exactly its own views.

Shapes follow the reference's call sites:
  * chessboard corners (k*s, j*s, 0), j over rows, k over columns
    (/root/reference/monocular_chessboard.cpp:298-300,
     /root/reference/binocular_chessboard.cpp:140-142);
  * asymmetric circle grid ((2*j + i%2)*s, i*s, 0)
    (/root/reference/monocular_circle.cpp:35-37).
Image points are returned as float32 (the reference hands Point2f to OpenCV).
"""
from __future__ import annotations

import dataclasses

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def u01(seed: int, view: np.ndarray, draw) -> np.ndarray:
    """Uniform [0,1) doubles keyed by (seed, view, draw)."""
    with np.errstate(over="ignore"):
        v = np.asarray(view, dtype=np.uint64)
        d = np.asarray(draw, dtype=np.uint64)
        x = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + v * np.uint64(0x9E3779B97F4A7C15)
        x = _mix(x) + d * np.uint64(0xD1B54A32D192ED03)
        x = _mix(x)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, view: np.ndarray, draw) -> np.ndarray:
    """Standard normals by Box-Muller; consumes draws 2*draw and 2*draw+1."""
    d = np.asarray(draw, dtype=np.uint64)
    a = u01(seed, view, 2 * d)
    b = u01(seed, view, 2 * d + 1)
    return np.sqrt(-2.0 * np.log(1.0 - a)) * np.cos(2.0 * np.pi * b)


@dataclasses.dataclass
class Camera:
    fx: float = 800.0
    fy: float = 800.0
    cx: float = 319.5
    cy: float = 239.5
    k1: float = -0.28
    k2: float = 0.09
    p1: float = 5e-4
    p2: float = -3e-4
    k3: float = -0.01
    k4: float = 0.0
    k5: float = 0.0
    k6: float = 0.0

    def K(self) -> np.ndarray:
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1.0]])

    def dist(self, n: int = 5) -> np.ndarray:
        d = np.array([self.k1, self.k2, self.p1, self.p2, self.k3, self.k4, self.k5, self.k6,
                      0, 0, 0, 0, 0, 0], dtype=np.float64)
        return d[:n].copy()


def chessboard(cols: int, rows: int, pitch: float) -> np.ndarray:
    j, k = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    return np.stack([k * pitch, j * pitch, np.zeros_like(k, dtype=np.float64)], -1).reshape(-1, 3).astype(np.float32)


def asymmetric_circles(width: int, height: int, spacing: float) -> np.ndarray:
    i, j = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    return np.stack([(2 * j + i % 2) * spacing, i * spacing, np.zeros_like(i, dtype=np.float64)], -1
                    ).reshape(-1, 3).astype(np.float32)


def rodrigues(r: np.ndarray) -> np.ndarray:
    """(N,3) rotation vectors -> (N,3,3) matrices."""
    r = np.asarray(r, dtype=np.float64).reshape(-1, 3)
    th = np.linalg.norm(r, axis=1)
    small = th < 1e-300
    ax = r / np.where(small, 1.0, th)[:, None]
    c, s = np.cos(th), np.sin(th)
    K = np.zeros((r.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -ax[:, 2], ax[:, 1]
    K[:, 1, 0], K[:, 1, 2] = ax[:, 2], -ax[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -ax[:, 1], ax[:, 0]
    R = c[:, None, None] * np.eye(3) + (1 - c)[:, None, None] * ax[:, :, None] * ax[:, None, :] + s[:, None, None] * K
    R[small] = np.eye(3)
    return R


def project(obj: np.ndarray, r: np.ndarray, t: np.ndarray, cam: Camera) -> np.ndarray:
    """obj (P,3), r,t (N,3) -> (N,P,2) pixel coordinates (float64)."""
    R = rodrigues(r)
    Xc = np.einsum("nij,pj->npi", R, obj.astype(np.float64)) + np.asarray(t, dtype=np.float64)[:, None, :]
    x = Xc[..., 0] / Xc[..., 2]
    y = Xc[..., 1] / Xc[..., 2]
    r2 = x * x + y * y
    r4, r6 = r2 * r2, r2 * r2 * r2
    cd = (1 + cam.k1 * r2 + cam.k2 * r4 + cam.k3 * r6) / (1 + cam.k4 * r2 + cam.k5 * r4 + cam.k6 * r6)
    xd = x * cd + 2 * cam.p1 * x * y + cam.p2 * (r2 + 2 * x * x)
    yd = y * cd + cam.p1 * (r2 + 2 * y * y) + 2 * cam.p2 * x * y
    return np.stack([cam.fx * xd + cam.cx, cam.fy * yd + cam.cy], -1)


def compose(r1, t1, r2, t2):
    """(r2,t2) o (r1,t1) for batches: R3 = R2 R1, t3 = R2 t1 + t2 -> (R3 matrices, t3)."""
    R1, R2 = rodrigues(r1), rodrigues(r2)
    R3 = R2 @ R1
    t3 = np.einsum("nij,nj->ni", R2, np.asarray(t1, dtype=np.float64).reshape(-1, 3)) + np.asarray(t2).reshape(-1, 3)
    return R3, t3


def rotmat_to_rvec(R: np.ndarray) -> np.ndarray:
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    w = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], -1)
    s = 0.5 * np.linalg.norm(w, axis=1)
    c = np.clip(0.5 * (np.trace(R, axis1=1, axis2=2) - 1), -1, 1)
    th = np.arctan2(s, c)
    scale = np.where(s > 1e-12, th / np.where(s > 1e-12, 2 * s, 1.0), 0.5)
    return w * scale[:, None]


@dataclasses.dataclass
class MonoSet:
    obj: np.ndarray        # (P,3) float32, one board shared by every view
    img: np.ndarray        # (N,P,2) float32
    rvec: np.ndarray       # (N,3) ground-truth poses
    tvec: np.ndarray       # (N,3)
    cam: Camera
    width: int = 640
    height: int = 480


def make_views(obj: np.ndarray, nviews: int, seed: int, cam: Camera | None = None, noise: float = 0.3,
               first_view: int = 0, width: int = 640, height: int = 480, margin: float = 4.0,
               max_angle_deg: float = 35.0, depth=(400.0, 1200.0)) -> MonoSet:
    """Random board poses, all corners inside the image (rejection sampled per view)."""
    cam = cam or Camera()
    view = np.arange(first_view, first_view + nviews, dtype=np.uint64)
    rvec, tvec = _views_for(obj, view, seed, cam, width, height, depth, margin, max_angle_deg)
    px = project(obj, rvec, tvec, cam)
    if noise > 0:
        P = obj.shape[0]
        pidx = np.arange(2 * P, dtype=np.uint64)[None, :]
        g = normal(seed, view[:, None], np.uint64(100000) + pidx).reshape(nviews, P, 2)
        px = px + noise * g
    return MonoSet(obj=obj, img=px.astype(np.float32), rvec=rvec, tvec=tvec, cam=cam, width=width, height=height)


@dataclasses.dataclass
class StereoSet:
    obj: np.ndarray
    img1: np.ndarray
    img2: np.ndarray
    rvec: np.ndarray      # camera-0 poses
    tvec: np.ndarray
    r_lr: np.ndarray
    t_lr: np.ndarray
    cam1: Camera
    cam2: Camera
    width: int = 640
    height: int = 480


def make_stereo(obj: np.ndarray, nviews: int, seed: int, noise: float = 0.3, first_view: int = 0,
                cam1: Camera | None = None, cam2: Camera | None = None,
                r_lr=(0.01, -0.02, 0.005), t_lr=(-60.0, 0.5, 1.0), width: int = 640, height: int = 480,
                depth=(500.0, 1200.0)) -> StereoSet:
    """Stereo pairs: camera-1 pose = (r_lr,t_lr) o camera-0 pose.  Views where the board leaves
    either image are re-drawn (both cameras must see every corner)."""
    cam1 = cam1 or Camera()
    cam2 = cam2 or Camera(fx=805.0, fy=805.0)
    r_lr = np.asarray(r_lr, dtype=np.float64)
    t_lr = np.asarray(t_lr, dtype=np.float64)
    got_r = np.zeros((nviews, 3))
    got_t = np.zeros((nviews, 3))
    todo = np.arange(nviews)
    salt = 0
    while todo.size:
        # keyed by global view id, so each view keeps its own stream whatever the shard
        v = (first_view + todo).astype(np.uint64)
        ms = _views_for(obj, v, seed + 7919 * salt, cam1, width, height, depth)
        R3, t3 = compose(ms[0], ms[1], np.broadcast_to(r_lr, ms[0].shape), np.broadcast_to(t_lr, ms[1].shape))
        r3 = rotmat_to_rvec(R3)
        px2 = project(obj, r3, t3, cam2)
        ok = ((px2[..., 0] >= 4) & (px2[..., 0] <= width - 5) & (px2[..., 1] >= 4) & (px2[..., 1] <= height - 5)).all(1)
        got_r[todo[ok]] = ms[0][ok]
        got_t[todo[ok]] = ms[1][ok]
        todo = todo[~ok]
        salt += 1
        if salt > 200:
            raise RuntimeError("stereo placement failed")
    R3, t3 = compose(got_r, got_t, np.broadcast_to(r_lr, got_r.shape), np.broadcast_to(t_lr, got_t.shape))
    r3 = rotmat_to_rvec(R3)
    px1 = project(obj, got_r, got_t, cam1)
    px2 = project(obj, r3, t3, cam2)
    if noise > 0:
        P = obj.shape[0]
        view = np.arange(first_view, first_view + nviews, dtype=np.uint64)
        pidx = np.arange(2 * P, dtype=np.uint64)[None, :]
        px1 = px1 + noise * normal(seed, view[:, None], np.uint64(100000) + pidx).reshape(nviews, P, 2)
        px2 = px2 + noise * normal(seed, view[:, None], np.uint64(200000) + pidx).reshape(nviews, P, 2)
    return StereoSet(obj=obj, img1=px1.astype(np.float32), img2=px2.astype(np.float32), rvec=got_r, tvec=got_t,
                     r_lr=r_lr, t_lr=t_lr, cam1=cam1, cam2=cam2, width=width, height=height)


def _views_for(obj, view_ids, seed, cam, width, height, depth, margin=4.0, max_angle_deg=35.0):
    """make_views' pose sampler for an explicit list of global view ids; returns (rvec, tvec)."""
    n = view_ids.shape[0]
    centre = 0.5 * (obj.max(0) + obj.min(0)).astype(np.float64)
    rvec = np.zeros((n, 3))
    tvec = np.zeros((n, 3))
    todo = np.arange(n)
    for attempt in range(400):
        if todo.size == 0:
            break
        v = view_ids[todo]
        base = attempt * 8
        z = 2 * u01(seed, v, base + 0) - 1
        phi = 2 * np.pi * u01(seed, v, base + 1)
        s = np.sqrt(1 - z * z)
        axis = np.stack([s * np.cos(phi), s * np.sin(phi), z], -1)
        ang = np.deg2rad(max_angle_deg) * u01(seed, v, base + 2)
        r = axis * ang[:, None]
        d = depth[0] + (depth[1] - depth[0]) * u01(seed, v, base + 3)
        uc = width * (0.15 + 0.7 * u01(seed, v, base + 4))
        vc = height * (0.15 + 0.7 * u01(seed, v, base + 5))
        pc = np.stack([(uc - cam.cx) / cam.fx * d, (vc - cam.cy) / cam.fy * d, d], -1)
        R = rodrigues(r)
        t = pc - np.einsum("nij,j->ni", R, centre)
        px = project(obj, r, t, cam)
        Xc = np.einsum("nij,pj->npi", R, obj.astype(np.float64)) + t[:, None, :]
        ok = ((px[..., 0] >= margin) & (px[..., 0] <= width - 1 - margin) & (px[..., 1] >= margin) &
              (px[..., 1] <= height - 1 - margin) & (Xc[..., 2] > min(50.0, 0.125 * depth[0]))).all(1)
        rvec[todo[ok]] = r[ok]
        tvec[todo[ok]] = t[ok]
        todo = todo[~ok]
    if todo.size:
        raise RuntimeError(f"{todo.size} views could not be placed inside the image")
    return rvec, tvec


def perturbed_poses(rvec: np.ndarray, tvec: np.ndarray, seed: int, first_view: int = 0,
                    rot_sigma: float = 0.02, trans_sigma: float = 5.0) -> np.ndarray:
    """(N,6) starting poses = ground truth + Gaussian perturbation (for runs that skip pose init)."""
    n = rvec.shape[0]
    view = np.arange(first_view, first_view + n, dtype=np.uint64)[:, None]
    k = np.arange(6, dtype=np.uint64)[None, :]
    g = normal(seed, view, np.uint64(900000) + k)
    rt = np.concatenate([rvec, tvec], 1)
    rt[:, :3] += rot_sigma * g[:, :3]
    rt[:, 3:] += trans_sigma * g[:, 3:]
    return rt
