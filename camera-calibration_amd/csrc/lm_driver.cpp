// lm_driver.cpp -- host side of the LM loop: damping, the reduced shared-block solve, acceptance
// and convergence.  It follows the CvLevMarq state machine of OpenCV 4.1.0
// (modules/calib3d/src/compat_ptsetreg.cpp: updateAlt / step) as the reference drives it through
// calibrateCamera (monocular_circle.cpp:93-94) and stereoCalibrate (binocular_chessboard.cpp:152-162):
//   lambda = 10^lambdaLg10 starting at 1e-3, diag *= 1 + lambda, retry on the same Jacobian with
//   lambda * 10 when the error grows, lambda / 10 after an accepted step, stop on iteration count or
//   on |param - prevParam| / |prevParam| < eps taken over ALL parameters (views included).
// OpenCV solves the full (shared + 6N)^2 system; here each view's 6x6 block is eliminated by the
// backend (on the GPU) and only the small reduced system is solved on the host.  The two are the
// same linear system.
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ccalib.h"
#include "host_util.hpp"
#include "reduced_solve.hpp"

namespace ccalib {

// Gaussian elimination with partial pivoting (cv::solve DECOMP_LU).  A is n x n row-major.
bool solve_lu(std::vector<double> A, std::vector<double>& b, int n) {
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++)
            if (std::fabs(A[j * n + i]) > std::fabs(A[k * n + i])) k = j;
        if (std::fabs(A[k * n + i]) < DBL_EPSILON * 100) return false;
        if (k != i) {
            for (int j = i; j < n; j++) std::swap(A[i * n + j], A[k * n + j]);
            std::swap(b[i], b[k]);
        }
        double d = -1 / A[i * n + i];
        for (int j = i + 1; j < n; j++) {
            double alpha = A[j * n + i] * d;
            for (int c = i + 1; c < n; c++) A[j * n + c] += alpha * A[i * n + c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int c = i + 1; c < n; c++) s -= A[i * n + c] * b[c];
        b[i] = s / A[i * n + i];
    }
    return true;
}

// Symmetric eigen-decomposition by cyclic Jacobi rotations: w descending, rows of V = eigenvectors.
void eig_sym(std::vector<double> A, int n, std::vector<double>& w, std::vector<double>& V) {
    V.assign((size_t)n * n, 0.);
    for (int i = 0; i < n; i++) V[i * n + i] = 1.;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0) continue;
                double th = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (th >= 0 ? 1 : -1) / (std::fabs(th) + std::sqrt(th * th + 1));
                double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    double x = A[k * n + p], y = A[k * n + q];
                    A[k * n + p] = c * x - s * y, A[k * n + q] = s * x + c * y;
                }
                for (int k = 0; k < n; k++) {
                    double x = A[p * n + k], y = A[q * n + k];
                    A[p * n + k] = c * x - s * y, A[q * n + k] = s * x + c * y;
                }
                for (int k = 0; k < n; k++) {
                    double x = V[p * n + k], y = V[q * n + k];
                    V[p * n + k] = c * x - s * y, V[q * n + k] = s * x + c * y;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (w[j] > w[m]) m = j;
        if (m != i) {
            std::swap(w[i], w[m]);
            for (int k = 0; k < n; k++) std::swap(V[i * n + k], V[m * n + k]);
        }
    }
}

// cv::solve DECOMP_SVD on a symmetric matrix.
// The SVD solution of a symmetric positive definite system that is not numerically singular is its
// ordinary solution, so that case -- every reduced system of a healthy calibration -- is solved by a
// Cholesky factorisation (0.5 us for 9x9 against 12 us for the eigen decomposition, which sits on the
// critical path of every LM iteration while the GPU waits).  Like Jacobi rotations, Cholesky is accurate
// relative to the diagonally scaled matrix, which matters here: the columns of the reduced system differ by
// many orders of magnitude (focal lengths against k3).  Whenever a pivot falls below 1e-10 of its diagonal
// entry or below 1e4 times SVD's own cut-off (2 eps sum|w| = 2 eps trace) -- the matrix may then be rank
// deficient for SVD's purposes -- the eigen path decides: eigen back-substitution, dropping
// |w| <= 2 eps sum|w| as SVD back-substitution does.
void solve_sym_svd(const std::vector<double>& A, std::vector<double>& b, int n) {
    static const bool eigen_only = std::getenv("CCALIB_EIGEN_SOLVE") != nullptr;  // development: never the Cholesky shortcut
    if (n <= 64 && !eigen_only) {
        double L[64 * 64], y[64];
        if (chol_solve_checked(A.data(), b.data(), n, L, y)) return;
    }
    std::vector<double> w, V, x(n, 0.);
    eig_sym(A, n, w, V);
    double thr = 0;
    for (double v : w) thr += std::fabs(v);
    thr *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        if (std::fabs(w[i]) <= thr) continue;
        double s = 0;
        for (int k = 0; k < n; k++) s += V[i * n + k] * b[k];
        s /= w[i];
        for (int k = 0; k < n; k++) x[k] += s * V[i * n + k];
    }
    b = x;
}

// One CvLevMarq::step(): the reduced system S = U' - Ssum, b = gs - bsum over the free set (U' = U mirrored, its
// diagonal times 1 + lambda), its solution, and the trial point shared = prev - solution on the free entries.  The
// driver below and a backend that predicts the driver's next step both call this, so the two cannot differ by a bit.
void lm_trial_step(int ns, int nf, const int* fidx, const double* U, const double* gs, const double* Ssum,
                   const double* bsum, double lambda, int solve_method, const double* prev, double* shared, double* dsh) {
    std::vector<double> S((size_t)nf * nf), b(nf);
    for (int a = 0; a < nf; a++) {
        for (int c = 0; c < nf; c++) {
            int ia = fidx[a], ic = fidx[c];
            double u = ia <= ic ? U[(size_t)ia * ns + ic] : U[(size_t)ic * ns + ia];  // completeSymm
            if (a == c) u *= 1. + lambda;
            S[(size_t)a * nf + c] = u - Ssum[(size_t)a * nf + c];
        }
        b[a] = gs[fidx[a]] - bsum[a];
    }
    if (nf > 0) {
        if (solve_method == 0) {
            if (!solve_lu(S, b, nf)) std::fill(b.begin(), b.end(), 0.);  // cv::solve leaves zeros
        } else
            solve_sym_svd(S, b, nf);
    }
    for (int i = 0; i < ns; i++) shared[i] = prev[i];
    for (int a = 0; a < nf; a++) {
        dsh[a] = b[a];
        shared[fidx[a]] = prev[fidx[a]] - b[a];
    }
}

// per-iteration records of the last ccalib_lm_run on this thread (ccalib_last_lm_trace)
thread_local std::vector<double> g_lm_trace;

}  // namespace ccalib

using namespace ccalib;

extern "C" int ccalib_last_lm_trace(double* records, int cap, int* nrecords) {
    if (!nrecords || cap < 0 || (cap > 0 && !records)) return set_error(CCALIB_ERR_ARG, "ccalib_last_lm_trace: bad arguments");
    const int n = (int)(g_lm_trace.size() / CCALIB_TRACE_FIELDS);
    *nrecords = n;
    const int m = std::min(n, cap);
    if (m > 0) std::memcpy(records, g_lm_trace.data(), sizeof(double) * CCALIB_TRACE_FIELDS * m);
    return CCALIB_OK;
}

extern "C" int ccalib_lm_run(const ccalib_backend* be, void* user, double* shared, const unsigned char* smask,
                             int term_type, int max_iter, double eps, int solve_method, ccalib_stats* stats) {
    if (!be || !shared || !smask || be->ns <= 0 || !be->jac || !be->reduce || !be->step_err || !be->commit)
        return set_error(CCALIB_ERR_ARG, "ccalib_lm_run: incomplete backend");
    auto t0 = std::chrono::steady_clock::now();
    const int ns = be->ns;
    // CvLevMarq::init
    if (term_type & CCALIB_TERM_COUNT)
        max_iter = std::min(std::max(max_iter, 1), 1000);
    else
        max_iter = 30;
    if (term_type & CCALIB_TERM_EPS)
        eps = std::max(eps, 0.);
    else
        eps = DBL_EPSILON;
    int lambdaLg10 = -3, iters = 0, jac_passes = 0, err_passes = 0;
    const double LOG10 = std::log(10.);
    auto lam = [&](int lg) { return std::exp(lg * LOG10); };

    // CCALIB_TRACE=<file>: one JSON line per LM iteration (damping, residual sum, trial steps, host time)
    std::FILE* trace = nullptr;
    if (const char* tp = std::getenv("CCALIB_TRACE")) trace = std::fopen(tp, "a");
    struct TraceClose {
        std::FILE* f;
        ~TraceClose() {
            if (f) std::fclose(f);
        }
    } trace_close{trace};

    g_lm_trace.clear();
    std::vector<int> fidx;
    for (int i = 0; i < ns; i++)
        if (smask[i]) fidx.push_back(i);
    const int nf = (int)fidx.size();
    std::vector<double> prev(ns, 0.), U((size_t)ns * ns), gs(ns), Ssum((size_t)nf * nf), bsum(nf), dsh(nf);
    double errNorm = 0, prevErrNorm = DBL_MAX, dn2 = 0, pn2 = 0;
    int rc;

    // one CvLevMarq::step() + the residual pass that follows it
    auto step_and_eval = [&](double& err_out) -> int {
        double lambda = lam(lambdaLg10);
        if ((rc = be->reduce(user, lambda, fidx.data(), nf, Ssum.data(), bsum.data()))) return rc;
        lm_trial_step(ns, nf, fidx.data(), U.data(), gs.data(), Ssum.data(), bsum.data(), lambda, solve_method, prev.data(),
                      shared, dsh.data());
        if (be->fixup) be->fixup(user, shared, prev.data());
        err_passes++;
        return be->step_err(user, lambda, lam(std::max(lambdaLg10 - 1, -16)), fidx.data(), nf, dsh.data(), shared, &dn2,
                            &pn2, &err_out);
    };

    // STARTED -> CALC_J
    if (be->fixup) be->fixup(user, shared, prev.data());
    jac_passes++;
    if ((rc = be->jac(user, shared, lam(lambdaLg10), U.data(), gs.data(), &errNorm))) return rc;
    for (;;) {
        // CALC_J: prevParam = param, step
        prev.assign(shared, shared + ns);
        if ((rc = be->commit(user))) return rc;
        prevErrNorm = errNorm;
        const auto ti = std::chrono::steady_clock::now();
        const int lambda_first = lambdaLg10;
        int trials = 1;
        if ((rc = step_and_eval(errNorm))) return rc;
        // CHECK_ERR
        while (errNorm > prevErrNorm) {  // a NaN compares false and is accepted, as in CvLevMarq
            if (++lambdaLg10 > 16) break;
            trials++;
            if ((rc = step_and_eval(errNorm))) return rc;
        }
        {
            const double rec[CCALIB_TRACE_FIELDS] = {(double)iters, (double)lambda_first, (double)std::min(lambdaLg10, 16),
                                                     (double)trials, prevErrNorm, errNorm, dn2,
                                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti).count()};
            if (g_lm_trace.size() < (size_t)CCALIB_TRACE_FIELDS * 1000) g_lm_trace.insert(g_lm_trace.end(), rec, rec + CCALIB_TRACE_FIELDS);
        }
        if (trace)
            std::fprintf(trace,
                         "{\"iter\": %d, \"lambda_lg10_first\": %d, \"lambda_lg10_accepted\": %d, \"trials\": %d, "
                         "\"err_prev\": %.17g, \"err\": %.17g, \"step_norm2\": %.17g, \"ms\": %.4f}\n",
                         iters, lambda_first, std::min(lambdaLg10, 16), trials, prevErrNorm, errNorm, dn2,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti).count());
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double num = dn2, den = pn2;
        for (int i = 0; i < ns; i++) {
            num += (shared[i] - prev[i]) * (shared[i] - prev[i]);
            den += prev[i] * prev[i];
        }
        if (++iters >= max_iter || std::sqrt(num) / std::sqrt(den) < eps) break;
        if (be->fixup) be->fixup(user, shared, prev.data());
        jac_passes++;
        double unused;
        if ((rc = be->jac(user, shared, lam(lambdaLg10), U.data(), gs.data(), &unused))) return rc;
    }
    if (stats) {
        stats->iters = iters;
        stats->jac_passes = jac_passes;
        stats->err_passes = err_passes;
        stats->lambda_lg10 = lambdaLg10;
        stats->final_err = errNorm;
        stats->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return CCALIB_OK;
}
