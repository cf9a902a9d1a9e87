"""A CPU backend for the host LM driver (ccalib_lm_run), built from the ORACLE's per-view blocks.

Test infrastructure only.  It lets the driver -- the same C++ state machine that steps the HIP
kernels -- run without a GPU, so that its damping / acceptance / convergence logic and the
sharded (multi-rank) sums can be checked on CPU with gloo.  Every rank owns a slice of the views;
each quantity the driver receives is summed over ranks by `allreduce`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class OracleBackend:
    def __init__(self, pkg, orc, obj, img, poses, mask18, aspect=0.0, allreduce=None):
        self.pkg, self.orc = pkg, orc
        self.obj, self.img = obj, img
        self.trial = np.array(poses, dtype=np.float64).reshape(-1, 6).copy()
        self.committed = self.trial.copy()
        self.mask = np.asarray(mask18, dtype=np.uint8)
        self.free = np.flatnonzero(self.mask).astype(np.int32)
        self.aspect = float(aspect)
        self.allreduce = allreduce or (lambda a: None)
        self.blocks = None
        self.calls = dict(jac=0, reduce=0, step_err=0, commit=0)
        capi = pkg.capi
        self._keep = [capi.JAC_FN(self._jac), capi.REDUCE_FN(self._reduce), capi.STEP_FN(self._step_err),
                      capi.COMMIT_FN(self._commit), capi.FIXUP_FN(self._fixup)]
        self.struct = capi.Backend(18, *self._keep)

    # -- helpers -------------------------------------------------------------------------------
    def _blocks(self, shared, poses):
        return self.orc.mono_blocks(self.obj, self.img, np.asarray(shared, dtype=np.float64), self.aspect, poses,
                                    self.free)

    def _damped_solves(self, lam):
        b = self.blocks
        for i in range(len(self.img)):
            Vd = b["V"][i] + lam * np.diag(np.diag(b["V"][i]))
            yield i, Vd

    # -- the backend operations ----------------------------------------------------------------
    def _jac(self, _u, shared, lam_hint, U, gs, err):
        self.calls["jac"] += 1
        sh = np.ctypeslib.as_array(shared, shape=(18,)).copy()
        self.blocks = self._blocks(sh, self.trial)
        packed = np.concatenate([self.blocks["U"].ravel(), self.blocks["gc"], [self.blocks["err"]]])
        self.allreduce(packed)
        nf = len(self.free)
        Uo = np.ctypeslib.as_array(U, shape=(18, 18))
        go = np.ctypeslib.as_array(gs, shape=(18,))
        Uo[:] = 0
        go[:] = 0
        Uo[np.ix_(self.free, self.free)] = np.triu(packed[:nf * nf].reshape(nf, nf))
        go[self.free] = packed[nf * nf:nf * nf + nf]
        err[0] = packed[-1]
        return 0

    def _reduce(self, _u, lam, fidx, nf, Ssum, bsum):
        self.calls["reduce"] += 1
        f = np.ctypeslib.as_array(fidx, shape=(nf,))
        assert np.array_equal(f, self.free)
        S = np.zeros((nf, nf))
        bb = np.zeros(nf)
        for i, Vd in self._damped_solves(lam):
            Y = np.linalg.solve(Vd, self.blocks["W"][i].T)
            S += self.blocks["W"][i] @ Y
            bb += Y.T @ self.blocks["ge"][i]
        packed = np.concatenate([S.ravel(), bb])
        self.allreduce(packed)
        np.ctypeslib.as_array(Ssum, shape=(nf * nf,))[:] = packed[:nf * nf]
        np.ctypeslib.as_array(bsum, shape=(nf,))[:] = packed[nf * nf:]
        return 0

    def _step_err(self, _u, lam, _lam_next, fidx, nf, dshared, shared_new, dn2, pn2, err):
        self.calls["step_err"] += 1
        d = np.ctypeslib.as_array(dshared, shape=(nf,)).copy()
        sh = np.ctypeslib.as_array(shared_new, shape=(18,)).copy()
        dn = 0.0
        for i, Vd in self._damped_solves(lam):
            de = np.linalg.solve(Vd, self.blocks["ge"][i] - self.blocks["W"][i].T @ d)
            self.trial[i] = self.committed[i] - de
            dn += float(de @ de)
        e = self._blocks(sh, self.trial)["err"]
        packed = np.array([dn, float(np.sum(self.committed ** 2)), e])
        self.allreduce(packed)
        dn2[0], pn2[0], err[0] = packed
        return 0

    def _commit(self, _u):
        self.calls["commit"] += 1
        self.committed = self.trial.copy()
        return 0

    def _fixup(self, _u, shared, prev):
        if self.aspect != 0.0:
            shared[0] = shared[1] * self.aspect
            prev[0] = prev[1] * self.aspect


def run_lm(pkg, backend, intr18, criteria, use_lu=False):
    """Runs ccalib_lm_run over `backend`; returns (intr18, poses, stats dict)."""
    lib = pkg.capi.load()
    shared = np.array(intr18, dtype=np.float64).copy()
    smask = np.ascontiguousarray(backend.mask, dtype=np.uint8)
    st = pkg.capi.Stats()
    rc = lib.ccalib_lm_run(C.byref(backend.struct), None, shared.ctypes.data_as(C.c_void_p),
                           smask.ctypes.data_as(C.c_void_p), C.c_int(criteria[0]), C.c_int(criteria[1]),
                           C.c_double(criteria[2]), C.c_int(0 if use_lu else 1), C.byref(st))
    pkg.capi.check(rc)
    return shared, backend.trial.copy(), dict(iters=st.iters, jac_passes=st.jac_passes, err_passes=st.err_passes,
                                              lambda_lg10=st.lambda_lg10, final_err=st.final_err)
