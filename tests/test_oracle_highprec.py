"""The oracle's analytic Jacobians against 50-digit arithmetic, and its Rodrigues derivative against a symbolic one.

OpenCV is absent, so nothing external pins the oracle (parity unpinned).  What can be pinned is that every
derivative the oracle states IS the derivative of the function it states: here the functions are re-implemented
in mpmath at 50 digits straight from their definitions (cvProjectPoints2Internal's camera model, Rodrigues' formula
both ways, cvComposeRT's composition), differentiated by central differences with h = 1e-20 (truncation error
~1e-40), and compared with the oracle's float64 analytic values.  A sympy derivation of dR/dr from Rodrigues'
formula is evaluated at 40 digits as a second, symbolic witness.  Float64 finite differences (tests/test_oracle.py)
cannot see below ~1e-8; these can see the last few bits.
"""
import mpmath as mp
import numpy as np
import pytest
import sympy as sy

mp.mp.dps = 50
H = mp.mpf(10) ** -20


def _mpv(a):
    return [mp.mpf(float(x)) for x in np.ravel(a)]


def _v2m(r):
    """Rodrigues' formula, vector -> matrix (mp)."""
    th = mp.sqrt(r[0] ** 2 + r[1] ** 2 + r[2] ** 2)
    if th == 0:
        return mp.eye(3)
    ax = [c / th for c in r]
    c, s = mp.cos(th), mp.sin(th)
    K = mp.matrix([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    A = mp.matrix(3, 3)
    for i in range(3):
        for j in range(3):
            A[i, j] = ax[i] * ax[j]
    return c * mp.eye(3) + (1 - c) * A + s * K


def _m2v(R):
    """Matrix -> vector as cvRodrigues2's Jacobian defines it on the nine matrix entries: the skew part om gives the
    direction, the trace alone gives theta and the factor theta / (2 sin theta) (mp).  Away from theta = 0, pi."""
    om = [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]
    th = mp.acos((R[0, 0] + R[1, 1] + R[2, 2] - 1) / 2)
    return [o * th / (2 * mp.sin(th)) for o in om]


def _project(obj, r, t, fc, k, aspect):
    """cvProjectPoints2Internal's camera model (no tilt), (mp): returns the 2n pixel coordinates."""
    R = _v2m(r)
    fy = fc[1]
    fx = fy * aspect if aspect != 0 else fc[0]
    out = []
    for X in obj:
        P = [R[i, 0] * X[0] + R[i, 1] * X[1] + R[i, 2] * X[2] + t[i] for i in range(3)]
        x, y = P[0] / P[2], P[1] / P[2]
        r2 = x * x + y * y
        r4, r6 = r2 * r2, r2 * r2 * r2
        cd = (1 + k[0] * r2 + k[1] * r4 + k[4] * r6) / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6)
        a1, a2, a3 = 2 * x * y, r2 + 2 * x * x, r2 + 2 * y * y
        xd = x * cd + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4
        yd = y * cd + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4
        out += [fx * xd + fc[2], fy * yd + fc[3]]
    return out


def _cdiff(f, x, i):
    """d f / d x_i by a central difference at 50 digits."""
    xp, xm = list(x), list(x)
    xp[i] += H
    xm[i] -= H
    fp, fm = f(xp), f(xm)
    return [(a - b) / (2 * H) for a, b in zip(fp, fm)]


def _assert_close(got, want, rel, what):
    got = np.asarray(got, dtype=np.float64)
    want = np.array([[float(v) for v in row] for row in want]) if isinstance(want[0], (list, tuple)) else \
        np.array([float(v) for v in want])
    scale = np.abs(want).max() if want.size else 1.0
    assert np.abs(got - want).max() <= rel * max(scale, 1e-300), (what, np.abs(got - want).max() / scale)


@pytest.mark.parametrize("aspect", [0.0, 1.03])
def test_project_points_jacobians_50_digits(orc, aspect):
    rng = np.random.default_rng(3)
    obj = rng.uniform(-100, 100, size=(5, 3))
    obj[:, 2] *= 0.2
    r = np.array([0.2, -0.3, 0.1])
    t = np.array([10.0, -20.0, 600.0])
    fc = np.array([800.0, 790.0, 320.0, 240.0])
    k = np.array([-0.28, 0.09, 5e-4, -3e-4, -0.01, 0.02, -0.01, 0.005, 1e-3, -2e-3, 5e-4, 1e-3, 0, 0])
    m, Je, Ji = orc.project_points(obj, r, t, fc, k, aspect)
    O = [_mpv(p) for p in obj]
    rm, tm, fm, km = _mpv(r), _mpv(t), _mpv(fc), _mpv(k)
    am = mp.mpf(float(aspect))
    # the projection itself
    _assert_close(m.ravel(), _project(O, rm, tm, fm, km, am), 1e-14, "projection")
    for j in range(3):   # pose columns
        _assert_close(Je[:, j], _cdiff(lambda v: _project(O, v, tm, fm, km, am), rm, j), 2e-12, f"d/dr{j}")
        _assert_close(Je[:, 3 + j], _cdiff(lambda v: _project(O, rm, v, fm, km, am), tm, j), 2e-12, f"d/dt{j}")
    for j in range(12):  # all twelve distortion coefficients incl. rational and thin-prism terms
        _assert_close(Ji[:, 4 + j], _cdiff(lambda v: _project(O, rm, tm, fm, v, am), km, j), 2e-12, f"d/dk{j}")
    for j in (2, 3):     # principal point
        _assert_close(Ji[:, j], _cdiff(lambda v: _project(O, rm, tm, v, km, am), fm, j), 1e-13, f"d/dc{j}")
    if aspect == 0.0:
        for j in (0, 1):
            _assert_close(Ji[:, j], _cdiff(lambda v: _project(O, rm, tm, v, km, am), fm, j), 2e-13, f"d/df{j}")
    else:                # fx tied to fy: the fy column carries both rows, the fx column is zero
        _assert_close(Ji[:, 1], _cdiff(lambda v: _project(O, rm, tm, v, km, am), fm, 1), 2e-13, "d/dfy tied")
        assert np.all(Ji[:, 0] == 0)


def test_rodrigues_both_directions_50_digits(orc):
    rng = np.random.default_rng(7)
    cases = [rng.normal(size=3) * s for s in (1e-7, 1e-3, 0.3, 1.2, 2.5, 3.1)]
    for r in cases:
        R, J = orc.rodrigues_v2m(r)                       # J[i, k] = dR[k] / dr[i]
        rm = _mpv(r)
        Rm = _v2m(rm)
        _assert_close(R.ravel(), [Rm[i, j] for i in range(3) for j in range(3)], 2e-15, "R(r)")
        for i in range(3):
            want = _cdiff(lambda v: [_v2m(v)[a, b] for a in range(3) for b in range(3)], rm, i)
            _assert_close(J[i], want, 5e-13 if np.linalg.norm(r) > 1e-5 else 1e-8, f"dR/dr{i} at |r|={np.linalg.norm(r):.1e}")
        if 1e-4 < np.linalg.norm(r) < 3.0:
            # theta = acos((tr R - 1) / 2) is ill-conditioned at small angles: absolute error ~ eps / theta
            r2, _ = orc.rodrigues_m2v(R)
            assert np.abs(r2 - r).max() <= 1e-15 / min(1.0, np.linalg.norm(r)) + 1e-15
        if 0.05 < np.linalg.norm(r) < 3.0:
            # matrix -> vector on the exact rotation, and its 3x9 derivative w.r.t. the nine entries taken as
            # independent variables (how cvRodrigues2 defines it)
            r2, Ji = orc.rodrigues_m2v(R)                 # Ji[i, k] = dr[i] / dR[k]
            Rl = [Rm[a, b] for a in range(3) for b in range(3)]

            def f(v):
                M = mp.matrix(3, 3)
                for a in range(3):
                    for b in range(3):
                        M[a, b] = v[3 * a + b]
                return _m2v(M)
            for kk in range(9):
                want = _cdiff(f, Rl, kk)
                _assert_close(Ji[:, kk], want, 2e-10, f"dr/dR{kk}")
    # theta < DBL_EPSILON: R = I and the fixed skew derivative (the generators of so(3))
    R0, J0 = orc.rodrigues_v2m(np.zeros(3))
    assert np.array_equal(R0, np.eye(3))
    gen = np.zeros((3, 9))
    gen[0, 5], gen[0, 7] = -1, 1
    gen[1, 2], gen[1, 6] = 1, -1
    gen[2, 1], gen[2, 3] = -1, 1
    assert np.array_equal(J0, gen)
    # the limit of the derivative as r -> 0 is that same matrix
    tiny = _mpv([1e-30, -2e-30, 5e-31])
    for i in range(3):
        want = _cdiff(lambda v: [_v2m(v)[a, b] for a in range(3) for b in range(3)], tiny, i)
        _assert_close(gen[i], want, 1e-15, "generator limit")
    # theta close to pi: the vector comes back (up to sign of the axis at exactly pi)
    axis = np.array([0.3, -0.5, 0.8])
    axis /= np.linalg.norm(axis)
    for th in (np.pi - 1e-3, np.pi - 1e-7):
        R, _ = orc.rodrigues_v2m(axis * th)
        r2, _ = orc.rodrigues_m2v(R)
        assert min(np.abs(r2 - axis * th).max(), np.abs(r2 + axis * th).max()) < 1e-6


def test_compose_rt_jacobians_50_digits(orc):
    rng = np.random.default_rng(2)
    for _ in range(4):
        r1, r2 = rng.normal(size=3) * 0.4, rng.normal(size=3) * 0.05
        t1, t2 = rng.normal(size=3) * 300, rng.normal(size=3) * 60
        r3, t3, d = orc.compose_rt(r1, t1, r2, t2)
        a1, a2, b1, b2 = _mpv(r1), _mpv(r2), _mpv(t1), _mpv(t2)

        def comp(x1, y1, x2, y2):
            R1, R2 = _v2m(x1), _v2m(x2)
            rr = _m2v(R2 * R1)
            tt = [R2[i, 0] * y1[0] + R2[i, 1] * y1[1] + R2[i, 2] * y1[2] + y2[i] for i in range(3)]
            return rr, tt
        rr, tt = comp(a1, b1, a2, b2)
        _assert_close(r3, rr, 1e-13, "r3")
        _assert_close(t3, tt, 1e-14, "t3")
        for j in range(3):
            _assert_close(d["dr3dr1"][:, j], _cdiff(lambda v: comp(v, b1, a2, b2)[0], a1, j), 5e-12, "dr3/dr1")
            _assert_close(d["dr3dr2"][:, j], _cdiff(lambda v: comp(a1, b1, v, b2)[0], a2, j), 5e-12, "dr3/dr2")
            _assert_close(d["dt3dr2"][:, j], _cdiff(lambda v: comp(a1, b1, v, b2)[1], a2, j), 5e-12, "dt3/dr2")
            _assert_close(d["dt3dt1"][:, j], _cdiff(lambda v: comp(a1, v, a2, b2)[1], b1, j), 1e-14, "dt3/dt1")
            _assert_close(d["dt3dt2"][:, j], _cdiff(lambda v: comp(a1, b1, a2, v)[1], b2, j), 1e-14, "dt3/dt2")


def test_rodrigues_derivative_matches_symbolic_derivation(orc):
    """dR/dr derived by sympy from R = cos(t) I + (1 - cos t) a a^T + sin(t) [a]x, a = r / t, t = |r|, and
    evaluated at 40 digits: the a0..a4 coefficient form cvRodrigues2 uses must be this derivative."""
    rx, ry, rz = sy.symbols("rx ry rz", real=True)
    th = sy.sqrt(rx ** 2 + ry ** 2 + rz ** 2)
    a = sy.Matrix([rx, ry, rz]) / th
    K = sy.Matrix([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = sy.cos(th) * sy.eye(3) + (1 - sy.cos(th)) * (a * a.T) + sy.sin(th) * K
    dR = [sy.diff(R, v) for v in (rx, ry, rz)]
    rng = np.random.default_rng(11)
    for _ in range(4):
        r = rng.normal(size=3) * rng.uniform(0.05, 2.0)
        _, J = orc.rodrigues_v2m(r)
        sub = {rx: sy.Float(float(r[0]), 40), ry: sy.Float(float(r[1]), 40), rz: sy.Float(float(r[2]), 40)}
        for i in range(3):
            want = np.array([[float(dR[i][p, q].evalf(40, subs=sub)) for q in range(3)] for p in range(3)]).ravel()
            assert np.abs(J[i] - want).max() <= 5e-13 * max(1.0, np.abs(want).max())
