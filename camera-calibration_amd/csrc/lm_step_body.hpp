// lm_step_body.hpp -- device code of the LM step prediction (see lm_step.hpp), shared by k_lm_step and by the last
// workgroup of k_colsum.  Runs in ONE wavefront (lanes 0..63 of the calling workgroup; the caller makes sure only
// that wavefront enters), synchronising through wave-level barriers only.
//
// The arithmetic repeats ccalib_lm_run's step_and_eval (lm_driver.cpp) and chol_solve_checked (reduced_solve.hpp)
// operation for operation: every entry of the system, of L, of y and of x is the same chain of IEEE operations in
// the same order as on the host (the lanes only run different chains side by side), contraction is off, and f64
// divide and square root are correctly rounded on both sides.  The host compares the predicted step with its own
// bit for bit before it uses anything that was run ahead.
#pragma once
#include <cfloat>

#include "lm_step.hpp"

namespace ccalib {

constexpr int kLmLd = kMaxShared + 1;
constexpr int kLmRedLds = 1024;  // sums of a pass kept in LDS (larger passes are read from memory)

struct LmStepLds {
    double A[kMaxShared * kLmLd];
    double red[kLmRedLds];
    double x[kMaxShared], sh[kMaxShared];
    int u[kMaxShared * kMaxShared], s[kMaxShared * kMaxShared];
    int fidx[kMaxShared], slot[kMaxShared], g[kMaxShared], bs[kMaxShared];
};

__device__ __forceinline__ void lm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double lm_bcast(double v, int src_lane) {  // src_lane wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// `lane` = threadIdx.x of a thread of the first wavefront.  red_in: the pass's sums (device memory, complete and
// visible to this wavefront).
__device__ __forceinline__ void lm_step_body(const LmStepArgs& a, LmStepLds& w, int lane) {
#pragma clang fp contract(off)
    const LmCtrl* cin = a.cin;
#ifdef CCALIB_LM_TIMING
    long long tstamp[8];
    int nst = 0;
#define LM_STAMP() tstamp[nst++] = wall_clock64()
#else
#define LM_STAMP()
#endif
    LM_STAMP();
    // Everything the step needs is fetched with independent, coalesced loads first (sums and index tables into
    // LDS): the arithmetic then never waits for global memory, whose latency -- not the few hundred operations --
    // is what a single wavefront would otherwise spend its time on.
    const bool in_lds = a.n <= kLmRedLds;
    const volatile double* red_in = a.red;
    for (int i = lane; i < a.n; i += 64) {
        const double v = red_in[i];
        if (in_lds) w.red[i] = v;
        a.hred[i] = v;
    }
    if (a.cout) {
        const LmMaps& m = *a.maps;
        const int nf = m.nf, ns = m.ns;
        for (int i = lane; i < nf * nf; i += 64) w.u[i] = m.u_idx[i], w.s[i] = m.s_idx[i];
        if (lane < nf) w.fidx[lane] = m.fidx[lane], w.slot[lane] = m.dc_slot[lane], w.g[lane] = m.g_idx[lane], w.bs[lane] = m.bs_idx[lane];
        const double prev_err = cin ? cin->prev_err : a.prev_err;
        if (lane < ns) w.sh[lane] = cin ? cin->shared[lane] : a.shared[lane];
        lm_wave_sync();
        LM_STAMP();
        const double* r = (in_lds ? w.red : a.red) + a.base;
        double err = 0;
        for (int c = 0; c < m.nerr; c++) err += r[m.err_idx[c]];
        const bool rejected = err > prev_err;  // CvLevMarq retries while errNorm > prevErrNorm (a NaN is accepted)
        bool ok = !rejected;
        double rhs = 0, diag = 0;
        if (ok && lane < nf) {
            for (int c = 0; c <= lane; c++) {
                const int ui = w.u[lane * nf + c];
                double u = ui >= 0 ? r[ui] : 0.;
                if (c == lane) u *= 1. + a.lambda;
                diag = u - r[w.s[lane * nf + c]];
                w.A[lane * kLmLd + c] = diag;
            }
            const int gi = w.g[lane];
            rhs = (gi >= 0 ? r[gi] : 0.) - r[w.bs[lane]];
        }
        lm_wave_sync();
        LM_STAMP();
        double invd = 0, xv = 0;  // this lane's reciprocal pivot and solution entry
        if (ok) {
            double trace = 0;
            for (int i = 0; i < nf; i++) trace += fabs(w.A[i * kLmLd + i]);
            const double cut = (1e4 * 2 * DBL_EPSILON) * trace;
            for (int j = 0; j < nf; j++) {  // column j of L: lane i >= j owns entry (i, j)
                double sum = 0;
                if (lane >= j && lane < nf) {
                    sum = w.A[lane * kLmLd + j];
                    for (int k = 0; k < j; k++) sum -= w.A[lane * kLmLd + k] * w.A[j * kLmLd + k];
                }
                double piv = 0;
                if (lane == j) {
                    if (!(sum > 1e-10 * diag) || !(sum > cut))
                        piv = 0.;  // not positive enough: the host takes its eigen-decomposition path
                    else
                        piv = 1. / sqrt(sum), invd = piv;
                }
                piv = lm_bcast(piv, j);
                if (piv == 0.) {
                    ok = false;
                    break;
                }
                if (lane > j && lane < nf) w.A[lane * kLmLd + j] = sum * piv;
                lm_wave_sync();
            }
        }
        LM_STAMP();
        if (ok) {
            double sum = rhs;  // forward substitution, column by column: lane i subtracts L[i][k] y[k] for k ascending
            double y = 0;
            for (int k = 0; k < nf; k++) {
                const double yk = lm_bcast(sum * invd, k);
                if (lane == k) y = yk;
                if (lane > k && lane < nf) sum -= w.A[lane * kLmLd + k] * yk;
            }
            sum = y;  // backward: lane i subtracts L[k][i] x[k] for k descending
            for (int k = nf - 1; k >= 0; k--) {
                const double xk = lm_bcast(sum * invd, k);
                if (lane == k) xv = xk;
                if (lane < k) sum -= w.A[k * kLmLd + lane] * xk;
            }
            if (lane < nf) w.x[lane] = xv;
            // shared_next = trial point - step on the free entries, then the ties between parameters
            if (lane < nf) w.sh[w.fidx[lane]] = w.sh[w.fidx[lane]] - xv;
            lm_wave_sync();
            if (lane == 0) {
                double* sh = w.sh;
                if (m.ncams == 1) {
                    if (a.aspect[0] != 0) sh[0] = sh[1] * a.aspect[0];
                } else if (!a.fix_intrinsic) {
                    double* ip = sh + 6;
                    if (a.same_focal) ip[18] = ip[0], ip[19] = ip[1];
                    for (int c = 0; c < 2; c++)
                        if (a.aspect[c] != 0) ip[18 * c] = ip[18 * c + 1] * a.aspect[c];
                }
            }
            lm_wave_sync();
        }
        LM_STAMP();
        LmCtrl* o = a.cout;
        if (lane == 0) {
            o->go = ok ? 1 : 0;
            o->prev_err = err;
            a.hred[kPredOfs] = ok ? 1. : 0.;
        }
        if (ok) {
            if (lane < ns) {
                o->shared[lane] = w.sh[lane];
                a.hred[kPredOfs + 1 + nf + lane] = w.sh[lane];
            }
            if (lane < kCtrlCols) {  // the step in kernel-column order, zero where a column is not optimised
                double v = 0.;
                for (int f = 0; f < nf; f++)
                    if (w.slot[f] == lane) v = w.x[f];
                o->dc[lane] = v;
            }
            if (lane < nf) a.hred[kPredOfs + 1 + lane] = xv;
            if (lane < m.ncams) {
                CamParams cp = a.cp0[lane];
                if (!(m.ncams == 2 && a.fix_intrinsic)) {
                    const double* ip = w.sh + (m.ncams == 1 ? 0 : 6 + 18 * lane);
                    cp.fx = ip[0], cp.fy = ip[1], cp.cx = ip[2], cp.cy = ip[3];
                    for (int i = 0; i < 8; i++) cp.k[i] = ip[4 + i];
                }
                o->cp[lane] = cp;
            }
            if (lane < 6 && m.ncams == 2) o->lr.v[lane] = w.sh[lane];
        }
    } else if (lane == 0) {
        a.hred[kPredOfs] = 0.;
    }
    LM_STAMP();
    __threadfence_system();
    lm_wave_sync();
    LM_STAMP();
#ifdef CCALIB_LM_TIMING
    if (lane == 0) {
        for (int i = 0; i < nst; i++) a.hred[kPredOfs + 200 + i] = (double)(tstamp[i] - tstamp[0]);
        a.hred[kPredOfs + 199] = nst;
        __threadfence_system();
    }
#endif
    if (lane == 0) *(volatile unsigned long long*)a.flag = a.seq;
}

}  // namespace ccalib
