/* orc_math.c -- dense solvers for the CPU oracle (test infrastructure only). */
#include "orc_math.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "calib_oracle.h"

int orc_solve_lu(double* A, double* b, int n) {
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++)
            if (fabs(A[(size_t)j * n + i]) > fabs(A[(size_t)k * n + i])) k = j;
        if (fabs(A[(size_t)k * n + i]) < DBL_EPSILON * 100) return 0;
        if (k != i) {
            for (int j = i; j < n; j++) {
                double t = A[(size_t)i * n + j];
                A[(size_t)i * n + j] = A[(size_t)k * n + j];
                A[(size_t)k * n + j] = t;
            }
            double t = b[i];
            b[i] = b[k];
            b[k] = t;
        }
        double d = -1 / A[(size_t)i * n + i];
        for (int j = i + 1; j < n; j++) {
            double alpha = A[(size_t)j * n + i] * d;
            if (alpha == 0) continue;
            double* rj = A + (size_t)j * n;
            const double* ri = A + (size_t)i * n;
            for (int c = i + 1; c < n; c++) rj[c] += alpha * ri[c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int c = i + 1; c < n; c++) s -= A[(size_t)i * n + c] * b[c];
        b[i] = s / A[(size_t)i * n + i];
    }
    return 1;
}

int orc_solve_chol(double* A, double* b, int n) {
    for (int i = 0; i < n; i++) {
        for (int j = 0; j <= i; j++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            if (i == j) {
                if (!(s > 0)) return 0;
                A[(size_t)i * n + i] = sqrt(s);
            } else
                A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return 1;
}

void orc_eig_sym(const double* Ain, int n, double* w, double* V) {
    double* A = (double*)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = i == j;
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
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1 : -1) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) { /* eigenvectors as rows */
                    double vpk = V[p * n + k], vqk = V[q * n + k];
                    V[p * n + k] = c * vpk - s * vqk;
                    V[q * n + k] = s * vpk + c * vqk;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    /* sort descending */
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (w[j] > w[m]) m = j;
        if (m != i) {
            double t = w[i];
            w[i] = w[m];
            w[m] = t;
            for (int k = 0; k < n; k++) {
                t = V[i * n + k];
                V[i * n + k] = V[m * n + k];
                V[m * n + k] = t;
            }
        }
    }
    free(A);
}

int orc_solve_sym_svd(const double* A, const double* b, int n, double* x) {
    double* w = (double*)malloc(sizeof(double) * n);
    double* V = (double*)malloc(sizeof(double) * n * n);
    orc_eig_sym(A, n, w, V);
    double thr = 0;
    for (int i = 0; i < n; i++) thr += fabs(w[i]);
    thr *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int i = 0; i < n; i++) {
        if (fabs(w[i]) <= thr) continue;
        double s = 0;
        for (int k = 0; k < n; k++) s += V[i * n + k] * b[k];
        s /= w[i];
        for (int k = 0; k < n; k++) x[k] += s * V[i * n + k];
    }
    free(w);
    free(V);
    return 1;
}

int orc_solve(const double* A, const double* b, int n, int method, double* x) {
    if (method == ORC_SOLVE_SVD) return orc_solve_sym_svd(A, b, n, x);
    double* Ac = (double*)malloc(sizeof(double) * (size_t)n * n);
    memcpy(Ac, A, sizeof(double) * (size_t)n * n);
    memcpy(x, b, sizeof(double) * n);
    int ok = method == ORC_SOLVE_CHOLESKY ? orc_solve_chol(Ac, x, n) : orc_solve_lu(Ac, x, n);
    if (!ok) { /* cv::solve leaves dst zero-filled when the matrix is singular */
        for (int i = 0; i < n; i++) x[i] = 0;
    }
    free(Ac);
    return ok;
}
