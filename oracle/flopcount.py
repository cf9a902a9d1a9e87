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


def count_model(cols, aspect=False, rational=False):
    """Single-sweep Jacobian pass of one point for an arbitrary column set of the 12-coefficient model, in the
    kernels' factored formulation.  cols: names out of fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6; aspect: fx = fy * a and
    the fy column carries (xd * a, yd); rational: the denominator 1 + k4 r^2 + k5 r^4 + k6 r^6 is evaluated."""
    global FLOPS
    R = [C(v) for v in (0.9, -0.1, 0.2, 0.1, 0.95, -0.15, -0.2, 0.1, 0.97)]
    t = [C(10.0), C(-20.0), C(600.0)]
    dR = [[C(0.1 * (i + j + 1)) for i in range(9)] for j in range(3)]
    fy, cx, cy, asp = C(805.0), C(320.0), C(240.0), C(1.01)
    k = [C(-0.28), C(0.09), C(5e-4), C(-3e-4), C(-0.01), C(0.01), C(0.02), C(0.03)]   # k1 k2 p1 p2 k3 k4 k5 k6
    X, Y, u, v = C(50.0), C(75.0), C(400.0), C(300.0)
    Z, O = C(0, zero=True), C(1, one=True)
    fx = fy * asp if aspect else C(800.0)
    # intrinsic columns: charged per requested column (the products shared between them once)
    charged = {"fx": 0, "fy": 1 if aspect else 0, "cx": 0, "cy": 0, "k1": 2, "k2": 2, "p1": 2, "p2": 2, "k3": 2,
               "k4": 2, "k5": 2, "k6": 2}
    col_flops = sum(charged[c] for c in cols)
    shared = 0
    if any(c in cols for c in ("k1", "k2", "k3", "k4", "k5", "k6")):
        shared += 4 if rational else 2       # qx = fx x ic, qy = fy y ic
    if any(c in cols for c in ("k4", "k5", "k6")):
        shared += 2                          # wx = -qx g, wy = -qy g
    FLOPS = 0
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
    ic = 1.0 / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6) if rational else O
    g = cdist * ic
    xd = x * g + k[2] * a1 + k[3] * a2
    yd = y * g + k[2] * a3 + k[3] * a1
    ex = xd * fx + cx - u
    ey = yd * fy + cy - v
    projection = FLOPS + (1 if aspect else 0)
    FLOPS = 0
    gp = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * ic
    if rational:
        gp = gp - cdist * ic * ic * (k[5] + 2 * k[6] * r2 + 3 * k[7] * r4)
    D00 = g + 2 * x * x * gp + 2 * k[2] * y + 6 * k[3] * x
    D01 = a1 * gp + 2 * k[2] * x + 2 * k[3] * y
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
    pose_cols = FLOPS
    qx, qy = fx * x * ic, fy * y * ic
    wx, wy = (-(qx * g), -(qy * g)) if rational else (Z, Z)
    col = {"fx": (xd, Z), "fy": ((xd * asp) if aspect else Z, yd), "cx": (O, Z), "cy": (Z, O),
           "k1": (qx * r2, qy * r2), "k2": (qx * r4, qy * r4), "p1": (fx * a1, fy * a3), "p2": (fx * a2, fy * a1),
           "k3": (qx * r6, qy * r6), "k4": (wx * r2, wy * r2), "k5": (wx * r4, wy * r4), "k6": (wx * r6, wy * r6)}
    jx = [col[c][0] for c in cols]
    jy = [col[c][1] for c in cols]
    FLOPS = 0
    accumulate_shared(jx, jy, ex, ey)
    accumulate_view(jx, jy, jex, jey, ex, ey)
    accumulation = FLOPS
    return {"projection": projection, "intrinsic_columns": col_flops + shared, "pose_columns": pose_cols,
            "accumulation": accumulation, "total": projection + col_flops + shared + pose_cols + accumulation}


if __name__ == "__main__":
    out = count()
    out["mono_flags0_9col"] = count_model(["fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3"])
    # the reference's stereoCalibrate flag set, per camera: fy cx cy k1 k2 k6 with fx tied, rational model
    out["stereo_reference_flags_6col_per_camera"] = count_model(["fy", "cx", "cy", "k1", "k2", "k6"], aspect=True,
                                                                rational=True)
    print(json.dumps(out, indent=1))
    _ = math.pi
