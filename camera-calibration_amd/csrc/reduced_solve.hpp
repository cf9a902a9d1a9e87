// reduced_solve.hpp -- Cholesky solve of the reduced (shared-parameter) system in ONE fixed order of IEEE operations.
//
// The host LM driver solves the reduced system of every iteration (cv::solve in CvLevMarq::step, OpenCV 4.1.0
// compat_ptsetreg.cpp); the device predicts that solve one iteration ahead (lm_step.hip) and the host adopts the
// run-ahead only when both agree bit for bit.  So the arithmetic is pinned down here: every entry of L, y and x is
// one chain "start value, then subtract products in a stated index order, then one divide or square root", with
// floating-point contraction off.  The device version distributes the chains over lanes but keeps each chain's
// order.
#pragma once
#include <cfloat>
#include <cmath>

namespace ccalib {

// A: n x n symmetric (row-major, lower triangle read), b: right-hand side in, solution out.
// L (n*n) and y (n) are work space.  Returns false -- leaving b untouched -- when a pivot falls below 1e-10 of its
// diagonal entry or below 1e4 * (2 eps trace): the caller then takes the eigen-decomposition path.
//   d[j]    = 1 / sqrt(A[j][j] - sum_{k<j, ascending} L[j][k]^2)              (the only divisions: one per pivot)
//   L[i][j] = (A[i][j] - sum_{k<j, ascending} L[i][k] L[j][k]) * d[j]
//   y[i]    = (b[i]    - sum_{k<i, ascending} L[i][k] y[k]) * d[i]
//   x[i]    = (y[i]    - sum_{k>i, descending} L[k][i] x[k]) * d[i]
// (multiplying by the reciprocal pivot keeps the divisions off the substitutions' dependent chains, which is what
// the device version's run time consists of)
inline bool chol_solve_checked(const double* A, double* b, int n, double* L, double* y) {
#pragma clang fp contract(off)
    double trace = 0;
    for (int i = 0; i < n; i++) trace += std::fabs(A[i * n + i]);
    const double cut = (1e4 * 2 * DBL_EPSILON) * trace;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j <= i; j++) {
            double sum = A[i * n + j];
            for (int k = 0; k < j; k++) sum -= L[i * n + k] * L[j * n + k];
            if (i == j) {
                if (!(sum > 1e-10 * A[i * n + i]) || !(sum > cut)) return false;
                L[i * n + i] = 1. / std::sqrt(sum);  // the diagonal slot keeps the reciprocal pivot d[i]
            } else
                L[i * n + j] = sum * L[j * n + j];
        }
    }
    for (int i = 0; i < n; i++) {
        double sum = b[i];
        for (int k = 0; k < i; k++) sum -= L[i * n + k] * y[k];
        y[i] = sum * L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double sum = y[i];
        for (int k = n - 1; k > i; k--) sum -= L[k * n + i] * b[k];
        b[i] = sum * L[i * n + i];
    }
    return true;
}

}  // namespace ccalib
