"""Instrumented FP64 operation counts for one point of the calibration passes.

Test/measurement infrastructure (lives with the oracle; never imported by the product).
Every arithmetic operation on a Counted value adds to a global tally:
add/sub/mul = 1 flop, divide = 1, sqrt = 1 (FMA therefore counts 2, as the roofline
convention asks).  Multiplications by the structural constants 0 and 1 are not performed
(the zero-exploiting count), so the resulting fraction of peak is not flattered.

Two formulations are counted:
  * "opencv": the per-column chain rule exactly as cvProjectPoints2Internal writes it,
    followed by the three small GEMMs per view (the CPU oracle's formulation);
  * "factored": the kernels' formulation -- one symmetric 2x2 distortion Jacobian and one
    2x3 projection Jacobian per point, 3 FMAs per pose column.
Run:  python oracle/flopcount.py
"""
from __future__ import annotations

import json
import math

FLOPS = 0


class C:
    """A double whose arithmetic is tallied.  zero / one mark structural constants."""
    __slots__ = ("v", "zero", "one")

    def __init__(self, v, zero=False, one=False):
        self.v, self.zero, self.one = float(v), zero, one

    @staticmethod
    def lift(x):
        return x if isinstance(x, C) else C(x, zero=(x == 0), one=(x == 1))

    def __add__(self, o):
        global FLOPS
        o = C.lift(o)
        if self.zero:
            return o
        if o.zero:
            return self
        FLOPS += 1
        return C(self.v + o.v)

    __radd__ = __add__

    def __sub__(self, o):
        global FLOPS
        o = C.lift(o)
        if o.zero:
            return self
        FLOPS += 1
        return C(self.v - o.v)

    def __rsub__(self, o):
        return C.lift(o).__sub__(self)

    def __neg__(self):
        return C(-self.v, zero=self.zero)   # sign flips fold into the consuming operation

    def __mul__(self, o):
        global FLOPS
        o = C.lift(o)
        if self.zero or o.zero:
            return C(0.0, zero=True)
        if self.one:
            return o
        if o.one:
            return self
        FLOPS += 1
        return C(self.v * o.v)

    __rmul__ = __mul__

    def __truediv__(self, o):
        global FLOPS
        o = C.lift(o)
        FLOPS += 1
        return C(self.v / o.v)

    def __rtruediv__(self, o):
        return C.lift(o).__truediv__(self)


def project(X, Y, R, t, fx, fy, cx, cy, k, u_obs, v_obs):
    """Shared by both formulations: camera-frame point, distortion, residual (planar board, Z = 0)."""
    xc = R[0] * X + R[1] * Y + t[0]
    yc = R[3] * X + R[4] * Y + t[1]
    zc = R[6] * X + R[7] * Y + t[2]
    z = 1.0 / zc
    x, y = xc * z, yc * z
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    a1 = 2 * x * y
    a2 = r2 + 2 * x * x
    a3 = r2 + 2 * y * y
    cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6
    xd = x * cdist + k[2] * a1 + k[3] * a2
    yd = y * cdist + k[2] * a3 + k[3] * a1
    ex = xd * fx + cx - u_obs
    ey = yd * fy + cy - v_obs
    return dict(x=x, y=y, z=z, r2=r2, r4=r4, r6=r6, a1=a1, a2=a2, a3=a3, cdist=cdist, xd=xd, yd=yd, ex=ex, ey=ey)


def intrinsic_columns(p, fx, fy):
    """Ji for the 9 free columns fx fy cx cy k1 k2 p1 p2 k3 (5-coefficient model)."""
    Z, O = C(0, zero=True), C(1, one=True)
    qx, qy = fx * p["x"], fy * p["y"]
    jx = [p["xd"], Z, O, Z, qx * p["r2"], qx * p["r4"], fx * p["a1"], fx * p["a2"], qx * p["r6"]]
    jy = [Z, p["yd"], Z, O, qy * p["r2"], qy * p["r4"], fy * p["a3"], fy * p["a1"], qy * p["r6"]]
    return jx, jy


def pose_columns_factored(p, X, Y, dR, fx, fy, k):
    x, y, z, r2, r4 = p["x"], p["y"], p["z"], p["r2"], p["r4"]
    gp = k[0] + 2 * k[1] * r2 + 3 * k[4] * r4
    g = p["cdist"]
    D00 = g + 2 * x * x * gp + 2 * k[2] * y + 6 * k[3] * x
    D01 = p["a1"] * gp + 2 * k[2] * x + 2 * k[3] * y
    D11 = g + 2 * y * y * gp + 6 * k[2] * y + 2 * k[3] * x
    sx, sy = fx * z, fy * z
    Tx = [sx * D00, sx * D01, None]
    Ty = [sy * D01, sy * D11, None]
    Tx[2] = -(Tx[0] * x + Tx[1] * y)
    Ty[2] = -(Ty[0] * x + Ty[1] * y)
    jex, jey = [None] * 6, [None] * 6
    for j in range(3):
        d = dR[j]
        dX, dY, dZ = d[0] * X + d[1] * Y, d[3] * X + d[4] * Y, d[6] * X + d[7] * Y
        jex[j] = Tx[0] * dX + Tx[1] * dY + Tx[2] * dZ
        jey[j] = Ty[0] * dX + Ty[1] * dY + Ty[2] * dZ
        jex[3 + j], jey[3 + j] = Tx[j], Ty[j]
    return jex, jey


def pose_columns_opencv(p, X, Y, dR, fx, fy, k):
    x, y, z, r2, r4, cdist = p["x"], p["y"], p["z"], p["r2"], p["r4"], p["cdist"]
    jex, jey = [None] * 6, [None] * 6
    Z = C(0, zero=True)
    dxdt = [z, Z, -(x * z)]
    dydt = [Z, z, -(y * z)]

    def chain(dx, dy):
        dr2 = 2 * x * dx + 2 * y * dy
        dcd = k[0] * dr2 + 2 * k[1] * r2 * dr2 + 3 * k[4] * r4 * dr2
        da1 = 2 * (x * dy + y * dx)
        dmx = dx * cdist + x * dcd + k[2] * da1 + k[3] * (dr2 + 4 * x * dx)
        dmy = dy * cdist + y * dcd + k[2] * (dr2 + 4 * y * dy) + k[3] * da1
        return fx * dmx, fy * dmy

    for j in range(3):
        jex[3 + j], jey[3 + j] = chain(dxdt[j], dydt[j])
    for j in range(3):
        d = dR[j]
        dx0, dy0, dz0 = X * d[0] + Y * d[1], X * d[3] + Y * d[4], X * d[6] + Y * d[7]
        dxdr = z * (dx0 - x * dz0)
        dydr = z * (dy0 - y * dz0)
        jex[j], jey[j] = chain(dxdr, dydr)
    return jex, jey


def accumulate_shared(jx, jy, ex, ey):
    n = len(jx)
    acc = 0
    for a in range(n):
        for b in range(a, n):
            acc = C(0, zero=True) + jx[a] * jx[b] + jy[a] * jy[b] + C(1.0)   # "+ C(1.0)": add into the running sum
        acc = C(0, zero=True) + jx[a] * ex + jy[a] * ey + C(1.0)
    acc = ex * ex + ey * ey + C(1.0)
    return acc


def accumulate_view(jx, jy, jex, jey, ex, ey):
    for a in range(6):
        for b in range(a, 6):
            _ = jex[a] * jex[b] + jey[a] * jey[b] + C(1.0)
        _ = jex[a] * ex + jey[a] * ey + C(1.0)
    for a in range(len(jx)):
        for b in range(6):
            _ = C(0, zero=True) + jx[a] * jex[b] + jy[a] * jey[b] + C(1.0)


def count():
    global FLOPS
    R = [C(v) for v in (0.9, -0.1, 0.2, 0.1, 0.95, -0.15, -0.2, 0.1, 0.97)]
    t = [C(10.0), C(-20.0), C(600.0)]
    dR = [[C(0.1 * (i + j + 1)) for i in range(9)] for j in range(3)]
    fx, fy, cx, cy = C(800.0), C(805.0), C(320.0), C(240.0)
    k = [C(-0.28), C(0.09), C(5e-4), C(-3e-4), C(-0.01)]
    X, Y, u, v = C(50.0), C(75.0), C(400.0), C(300.0)
    out = {}

    FLOPS = 0
    p = project(X, Y, R, t, fx, fy, cx, cy, k, u, v)
    _ = p["ex"] * p["ex"] + p["ey"] * p["ey"] + C(1.0)
    out["residual_pass"] = FLOPS

    FLOPS = 0
    p = project(X, Y, R, t, fx, fy, cx, cy, k, u, v)
    jx, jy = intrinsic_columns(p, fx, fy)
    accumulate_shared(jx, jy, p["ex"], p["ey"])
    out["jac_shared_kernel"] = FLOPS

    for name, fn in (("factored", pose_columns_factored), ("opencv", pose_columns_opencv)):
        FLOPS = 0
        p = project(X, Y, R, t, fx, fy, cx, cy, k, u, v)
        jx, jy = intrinsic_columns(p, fx, fy)
        jex, jey = fn(p, X, Y, dR, fx, fy, k)
        accumulate_view(jx, jy, jex, jey, p["ex"], p["ey"])
        out[f"jac_view_kernel_{name}"] = FLOPS
        FLOPS = 0
        p = project(X, Y, R, t, fx, fy, cx, cy, k, u, v)
        jx, jy = intrinsic_columns(p, fx, fy)
        jex, jey = fn(p, X, Y, dR, fx, fy, k)
        accumulate_shared(jx, jy, p["ex"], p["ey"])
        accumulate_view(jx, jy, jex, jey, p["ex"], p["ey"])
        out[f"jacobian_pass_single_sweep_{name}"] = FLOPS
    out["jacobian_pass_two_kernels_factored"] = out["jac_shared_kernel"] + out["jac_view_kernel_factored"]
    return out


if __name__ == "__main__":
    print(json.dumps(count(), indent=1))
    _ = math.pi
