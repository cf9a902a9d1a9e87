"""The stereo normal equations block by block, and the rotation edge cases, on the HIP path.

cvStereoCalibrateImpl (OpenCV 4.1.0 calibration.cpp, reached from binocular_chessboard.cpp:152-162 and
binocular_circle.cpp:153-163) rewrites every Jacobian row of camera 1 from d/d(r3,t3) into d/d(r1,t1) and
d/d(om_LR,T_LR) and accumulates
    JtJ[LR,LR] += J_LR^T J_LR     JtJ[LR,e_i] = J_LR^T Je     JtJ[e_i,e_i] += Je^T Je
    JtJ[i_k,i_k] += Ji^T Ji       JtJ[e_i,i_k] += Je^T Ji     JtJ[LR,i_1] += J_LR^T Ji.
The HIP path never forms those rows: it accumulates camera 1 in the (r3,t3) basis and rewrites the accumulated blocks
once per view (V = V0 + A^T V3 A, W_LR = B^T V3 A, ...).  Converged results cannot tell a wrong-but-descending term
from a right one, so here one Jacobian pass is compared with the oracle's (orc_stereo_blocks: the oracle's own loop
stopped after its first pass) entry by entry: the 42 x 42 shared block and its gradient, every view's V, ge and 42 x 6
coupling block, the Schur sums of both sum kernels, and one whole LM iteration.

Edge cases of cvRodrigues2 / cvComposeRT that a random rig never meets: a view with rvec = 0 exactly (theta < eps:
R = I and the fixed skew Jacobian), a parallel rig (om_LR = 0), both at once (composed rotation = I: the matrix -> vector
branch for sin(theta) < 1e-5, cos > 0), and a composed rotation within 1e-6 of pi (the cos < 0 branch, where OpenCV
takes the axis from the diagonal and leaves the Jacobian zero).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REF6 = [1, 2, 3, 4, 5, 11]         # fy cx cy k1 k2 k6: the reference's stereo flag set (fx tied)
PLAIN9 = [0, 1, 2, 3, 4, 5, 6, 7, 8]


def _ref_flags(pkg):
    return (pkg.CALIB_FIX_ASPECT_RATIO | pkg.CALIB_ZERO_TANGENT_DIST | pkg.CALIB_USE_INTRINSIC_GUESS |
            pkg.CALIB_SAME_FOCAL_LENGTH | pkg.CALIB_RATIONAL_MODEL | pkg.CALIB_FIX_K3 | pkg.CALIB_FIX_K4 | pkg.CALIB_FIX_K5)


def _setup(pkg, case, st):
    """Starting intrinsics a few percent off the truth, as (intr 2x18, mask 2x18, aspect, same_focal, flags)."""
    intr = np.zeros((2, 18))
    mask = np.zeros((2, 18), dtype=np.uint8)
    if case == "reference6":
        for c in range(2):      # same focal length: both cameras start from the same fx, fy, cx, cy (OpenCV averages them)
            intr[c, :4] = [812.0, 806.0, 322.0, 237.5]
            intr[c, 4:6] = [-0.22, 0.07]
            intr[c, 11] = 0.03 + 0.01 * c   # k6 away from zero: the rational denominator is exercised
        mask[:, REF6] = 1
        aspect = [intr[0, 0] / intr[0, 1], intr[1, 0] / intr[1, 1]]
        return intr, mask, aspect, True, _ref_flags(pkg)
    for c, cam in enumerate((st.cam1, st.cam2)):
        intr[c, :4] = [cam.fx * 0.98, cam.fy * 0.985, cam.cx + 2.5, cam.cy - 2.0]
        intr[c, 4:9] = cam.dist(5) * 0.8
    mask[:, PLAIN9] = 1
    return intr, mask, None, False, pkg.CALIB_USE_INTRINSIC_GUESS


def _K(row):
    return np.array([[row[0], 0, row[2]], [0, row[1], row[3]], [0, 0, 1.0]])


def _free(mask):
    return np.r_[np.arange(6), 6 + np.flatnonzero(mask[0]), 24 + np.flatnonzero(mask[1])]


def _check_blocks(pkg, orc, obj, img1, img2, intr, mask, aspect, same_focal, flags, rt, lr, monkeypatch, tol=1e-10):
    """One Jacobian pass on the GPU against orc_stereo_blocks; returns the oracle's blocks."""
    ref = orc.stereo_blocks(obj, img1, img2, _K(intr[0]), intr[0, 4:], _K(intr[1]), intr[1, 4:], flags, rt, lr)
    free = _free(mask)
    lam = 1e-3
    Sref = np.zeros((len(free), len(free)))
    bref = np.zeros(len(free))
    for i in range(len(rt)):
        Vd = ref["V"][i] + lam * np.diag(np.diag(ref["V"][i]))
        Y = np.linalg.solve(Vd, ref["W"][i][free].T)
        Sref += ref["W"][i][free] @ Y
        bref += Y.T @ ref["ge"][i]
    for quad in ("0", "1"):          # both Schur-sum kernels (size picks one in a run)
        monkeypatch.setenv("CCALIB_SUMS_QUAD", quad)
        s = pkg.Solver(obj, img1, img2)
        s.set_start(intr, mask, rt, aspect=aspect, same_focal=same_focal, lr=lr)
        U, gs, err = s.eval_jacobian()
        V, W, ge = s.fetch_blocks()
        S, b = s.reduce(lam, len(free))
        e2 = s.eval_error()
        s.close()
        assert abs(err - ref["err"]) <= 1e-11 * ref["err"] and abs(e2 - ref["err"]) <= 1e-11 * ref["err"]
        Uf, Ur = np.triu(U[np.ix_(free, free)]), np.triu(ref["U"][np.ix_(free, free)])
        scale = np.sqrt(np.outer(np.diag(Ur), np.diag(Ur)))
        assert np.all(np.abs(Uf - Ur) <= tol * scale + 1e-30), np.argwhere(np.abs(Uf - Ur) > tol * scale + 1e-30)
        assert np.allclose(gs[free], ref["gs"][free], rtol=10 * tol, atol=10 * tol * np.abs(ref["gs"][free]).max())
        assert np.allclose(V, ref["V"], rtol=tol, atol=tol * np.abs(ref["V"]).max(axis=(1, 2), keepdims=True))
        assert np.allclose(ge, ref["ge"], rtol=10 * tol, atol=10 * tol * np.abs(ref["ge"]).max())
        Wf, Wr = W[:, free, :], ref["W"][:, free, :]
        assert np.allclose(Wf, Wr, rtol=10 * tol, atol=tol * np.abs(Wr).max(axis=(1, 2), keepdims=True))
        # what OpenCV leaves zero stays zero: om_LR/T_LR against camera 0, camera 0 against camera 1
        assert np.all(U[:6, 6:24] == 0) and np.all(U[6:24, 24:] == 0)
        assert np.allclose(S, Sref, rtol=1e-8, atol=1e-9 * np.abs(Sref).max())
        assert np.allclose(b, bref, rtol=1e-8, atol=1e-9 * np.abs(bref).max())
    return ref


def _one_iteration(pkg, orc, obj, img1, img2, intr, mask, aspect, same_focal, flags, rt, lr):
    """One whole LM iteration (Jacobian pass, Schur reduction, 18/24-unknown solve, back-substitution, trial residuals,
    retries on the same Jacobian if the error grows) from the same start: against the oracle's Schur solver, which
    solves the same reduced system, to round-off; against its dense, OpenCV-shaped solver (a Jacobi SVD of the whole
    (6(N+1)+36)-unknown system) to four times what the oracle's two solvers differ by themselves -- with k1, k2 and k6
    free that system is conditioned badly enough for the two to part at 6e-9 in the residual sum after one step."""
    s = pkg.Solver(obj, img1, img2)
    s.set_start(intr, mask, rt, aspect=aspect, same_focal=same_focal, lr=lr)
    stt = s.run(criteria=(1, 1, 0.0))
    out = s.get()
    tr = pkg.last_lm_trace()
    s.close()
    refs = [orc.stereo_calibrate(obj, img1, img2, _K(intr[0]), intr[0, 4:], _K(intr[1]), intr[1, 4:], (640, 480), flags,
                                 criteria=(1, 1, 0.0), init_rt=rt, init_lr=lr, schur=schur, trace=True) for schur in (True, False)]
    got = dict(err=out["err"], K1=_K(out["intr"][0]), K2=_K(out["intr"][1]), d1=out["intr"][0, 4:], d2=out["intr"][1, 4:],
               R=pkg.synth.rodrigues(out["lr"][:3])[0], T=out["lr"][3:], r=out["rt"][:, :3], t=out["rt"][:, 3:])

    def fields(r):
        return dict(err=r["final_err"], K1=r["K1"], K2=r["K2"], d1=r["d1"], d2=r["d2"], R=r["R"], T=r["T"], r=r["rt"][:, :3],
                    t=r["rt"][:, 3:])

    def dist(a, b):     # difference relative to the size of the quantity
        return {k: float(np.abs(np.asarray(a[k]) - np.asarray(b[k])).max() / max(1e-300, np.abs(np.asarray(b[k])).max())) for k in b}

    fs, fd = fields(refs[0]), fields(refs[1])
    spread = dist(fs, fd)
    # one step from a start several percent off: a wrong Jacobian term would show at 1e-2; two correct implementations
    # differ by the round-off of their blocks (1e-10) times the conditioning of the reduced system
    floor = dict(err=2e-9, K1=1e-8, K2=1e-8, d1=1e-6, d2=1e-6, R=1e-8, T=1e-7, r=1e-7, t=1e-7)
    worst = dist(got, fs)
    print("one LM iteration, relative distance to the Schur oracle:", {k: "%.1e" % v for k, v in worst.items()})
    for k, v in worst.items():
        assert v <= floor[k], ("vs Schur oracle", k, v)
    for k, v in dist(got, fd).items():
        assert v <= max(floor[k], 4 * spread[k]) and spread[k] < 1e-5, ("vs dense oracle", k, v, spread[k])
    for ref in refs:
        assert stt["iters"] == ref["iters"] == 1
        assert tuple(tr[0, 1:4]) == tuple(ref["trace"][0, 1:4])              # same damping, same number of trial steps
    return out


@pytest.mark.parametrize("case", ["reference6", "plain9"])
def test_stereo_blocks_match_oracle(pkg, orc, synth, monkeypatch, case):
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 150, 0xC0FFEE + 3)       # two full 64-view workgroups and a ragged third
    intr, mask, aspect, same_focal, flags = _setup(pkg, case, st)
    rt = synth.perturbed_poses(st.rvec, st.tvec, 3, rot_sigma=0.01, trans_sigma=3.0)
    lr = np.r_[st.r_lr, st.t_lr] + np.array([1e-3, -1e-3, 5e-4, 0.5, -0.2, 0.3])
    _check_blocks(pkg, orc, obj, st.img1, st.img2, intr, mask, aspect, same_focal, flags, rt, lr, monkeypatch)
    _one_iteration(pkg, orc, obj, st.img1[:20], st.img2[:20], intr, mask, aspect, same_focal, flags, rt[:20], lr)


def _rig(synth, obj, rvec, tvec, r_lr, t_lr, cam1, cam2, seed):
    """Image points of both cameras for camera-0 poses (rvec, tvec) and the relative pose, with 0.1 px of noise."""
    R3, t3 = synth.compose(rvec, tvec, np.broadcast_to(r_lr, rvec.shape), np.broadcast_to(t_lr, tvec.shape))
    r3 = synth.rotmat_to_rvec(R3)
    n, P = rvec.shape[0], obj.shape[0]
    view = np.arange(n, dtype=np.uint64)[:, None]
    pidx = np.arange(2 * P, dtype=np.uint64)[None, :]
    px1 = synth.project(obj, rvec, tvec, cam1) + 0.1 * synth.normal(seed, view, np.uint64(100000) + pidx).reshape(n, P, 2)
    px2 = synth.project(obj, r3, t3, cam2) + 0.1 * synth.normal(seed, view, np.uint64(200000) + pidx).reshape(n, P, 2)
    return px1.astype(np.float32), px2.astype(np.float32)


@pytest.mark.parametrize("case", ["reference6", "plain9"])
def test_zero_rotation_vectors_and_a_parallel_rig(pkg, orc, synth, monkeypatch, case):
    """rvec = 0 exactly in two views, om_LR = 0 exactly: cvRodrigues2's theta < eps branch on both sides of the
    composition, and for those two views a composed rotation of exactly I (matrix -> vector: sin < 1e-5, cos > 0)."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 12, 555, r_lr=(0.0, 0.0, 0.0), t_lr=(-60.0, 0.0, 0.0))
    rvec, tvec = st.rvec.copy(), st.tvec.copy()
    rvec[0] = rvec[5] = 0.0
    tvec[0], tvec[5] = [-80.0, -60.0, 600.0], [-130.0, -40.0, 800.0]          # the board faces both cameras squarely
    r_lr, t_lr = np.zeros(3), np.array([-60.0, 0.0, 0.0])
    img1, img2 = _rig(synth, obj, rvec, tvec, r_lr, t_lr, st.cam1, st.cam2, 77)
    intr, mask, aspect, same_focal, flags = _setup(pkg, case, st)
    rt = np.c_[rvec, tvec]
    rt[[1, 2, 3, 4, 6, 7, 8, 9, 10, 11]] = synth.perturbed_poses(rvec, tvec, 5, rot_sigma=0.01, trans_sigma=3.0)[[1, 2, 3, 4, 6, 7, 8, 9, 10, 11]]
    rt[0, 3:] += [1.0, -2.0, 3.0]                                             # rotation stays exactly zero
    lr = np.r_[r_lr, t_lr + [0.5, -0.2, 0.3]]
    assert np.all(rt[0, :3] == 0) and np.all(rt[5, :3] == 0) and np.all(lr[:3] == 0)
    ref = _check_blocks(pkg, orc, obj, img1, img2, intr, mask, aspect, same_focal, flags, rt, lr, monkeypatch)
    assert np.all(np.linalg.eigvalsh(ref["V"][0]) > 0)                        # the view is well posed all the same
    _one_iteration(pkg, orc, obj, img1, img2, intr, mask, aspect, same_focal, flags, rt, lr)


@pytest.mark.parametrize("off_pi", [5e-7, 1e-3])
def test_composed_rotation_next_to_pi(pkg, orc, synth, monkeypatch, off_pi):
    """Camera 1 looks at the back of the board: the relative rotation is contrived so that view 0's composed rotation is
    pi - off_pi about a skew axis.  5e-7: inside cvRodrigues2's sin(theta) < 1e-5, cos < 0 branch (axis from the
    diagonal, zero Jacobian, hence no rotation coupling of that view through camera 1).  1e-3: just outside, where
    the regular formula divides by sin(theta) -- the blocks carry factors of 1e3 to 1e6 there and are held to 1e-7."""
    obj = synth.chessboard(9, 6, 25.0)
    st = synth.make_stereo(obj, 10, 808)
    rvec, tvec = st.rvec.copy(), st.tvec.copy()
    axis = np.array([0.96, 0.2, -0.196]) / np.linalg.norm([0.96, 0.2, -0.196])
    R3_0 = synth.rodrigues(axis * (np.pi - off_pi))[0]
    R1_0 = synth.rodrigues(rvec[0])[0]
    R2 = R3_0 @ R1_0.T
    # rotation vector of R2 (a few hundredths of a radian short of pi: its sine is far above the 1e-5 of the special branch)
    w = np.array([R2[2, 1] - R2[1, 2], R2[0, 2] - R2[2, 0], R2[1, 0] - R2[0, 1]])
    ang = np.arctan2(0.5 * np.linalg.norm(w), 0.5 * (np.trace(R2) - 1))
    r_lr = w / np.linalg.norm(w) * ang
    assert 0.3 < ang < np.pi - 1e-3
    t_lr = np.array([20.0, -30.0, 1900.0])        # R2 turns the views' depth negative; this brings the board in front again
    img1, img2 = _rig(synth, obj, rvec, tvec, r_lr, t_lr, st.cam1, st.cam2, 99)
    intr, mask, aspect, same_focal, flags = _setup(pkg, "plain9", st)
    rt = np.c_[rvec, tvec]                        # view 0 starts exactly where the composition lands next to pi
    rt[1:] = synth.perturbed_poses(rvec, tvec, 9, rot_sigma=0.005, trans_sigma=2.0)[1:]
    lr = np.r_[r_lr, t_lr]
    # the oracle's own composition confirms which branch view 0 takes
    r3, _, d = orc.compose_rt(rt[0, :3], rt[0, 3:], lr[:3], lr[3:])
    assert abs(np.linalg.norm(r3) - (np.pi - off_pi)) < 1e-9
    assert (np.abs(d["dr3dr1"]).max() == 0) == (off_pi < 1e-5)
    _check_blocks(pkg, orc, obj, img1, img2, intr, mask, aspect, same_focal, flags, rt, lr, monkeypatch,
                  tol=1e-10 if off_pi < 1e-5 else 1e-7)
    if off_pi < 1e-5:
        _one_iteration(pkg, orc, obj, img1, img2, intr, mask, aspect, same_focal, flags, rt, lr)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_mono_blocks_with_a_zero_rotation_vector(pkg, orc, synth, monkeypatch, fused):
    """calibrateCamera's pass with rvec = 0 exactly in some views, through the fused matrix-pipe kernel (pose records:
    theta < eps gives the fixed skew Jacobian) and through the two vector kernels (cvRodrigues2's branch as such)."""
    monkeypatch.setenv("CCALIB_NO_FUSED", "0" if fused == "1" else "1")
    obj = synth.chessboard(9, 6, 25.0)
    ms = synth.make_views(obj, 70, 4321)
    rt = synth.perturbed_poses(ms.rvec, ms.tvec, 99, rot_sigma=0.01, trans_sigma=3.0)
    img = ms.img.copy()
    for v, t in ((0, [-100.0, -60.0, 700.0]), (64, [-90.0, -70.0, 500.0]), (69, [-110.0, -50.0, 900.0])):
        rt[v] = np.r_[0.0, 0.0, 0.0, t]
        img[v] = (synth.project(obj, np.zeros((1, 3)), np.array([t]), ms.cam)[0] + 0.2 * np.sin(np.arange(108)).reshape(54, 2)).astype(np.float32)
    cam = ms.cam
    K = np.array([[cam.fx * 0.97, 0, cam.cx + 3], [0, cam.fy * 0.98, cam.cy - 2], [0, 0, 1.0]])
    intr, mask, aspect = pkg.mono_start(K, cam.dist(5) * 0.8, 0)
    s = pkg.Solver(obj, img)
    s.set_start(intr, mask, rt)
    U, gs, err = s.eval_jacobian()
    V, W, ge = s.fetch_blocks()
    e2 = s.eval_error()
    s.close()
    free = np.flatnonzero(mask)
    ref = orc.mono_blocks(obj, img, intr, aspect, rt, free)
    assert abs(err - ref["err"]) < 1e-11 * ref["err"] and abs(e2 - ref["err"]) < 1e-11 * ref["err"]
    Uf = np.triu(U[np.ix_(free, free)])
    scale = np.sqrt(np.outer(np.diag(ref["U"]), np.diag(ref["U"])))
    assert np.all(np.abs(Uf - np.triu(ref["U"])) <= 1e-10 * scale + 1e-30)
    assert np.allclose(V, ref["V"], rtol=1e-10, atol=1e-10 * np.abs(ref["V"]).max(axis=(1, 2), keepdims=True))
    assert np.allclose(ge, ref["ge"], rtol=1e-9, atol=1e-9 * np.abs(ref["ge"]).max())
    assert np.allclose(W[:, free, :], ref["W"], rtol=1e-9, atol=1e-10 * np.abs(ref["W"]).max(axis=(1, 2), keepdims=True))
    # projectPoints with rvec = 0 (monocular_circle.cpp:18-19's call shape)
    rms, pv, proj = pkg.reprojection_error(obj, img[:1], cam.K(), cam.dist(5), np.zeros((1, 3)), rt[:1, 3:], want_points=True)
    want = synth.project(obj, np.zeros((1, 3)), rt[:1, 3:], cam).reshape(-1, 2)
    assert np.abs(proj - want).max() < 1e-4
