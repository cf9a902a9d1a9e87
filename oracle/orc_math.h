/* orc_math.h -- small dense linear algebra for the CPU oracle (test infrastructure only). */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <stddef.h>

/* C(m x n) = A(m x k) * B(k x n), row-major, C must not alias A or B. */
static inline void orc_matmul(const double* A, int m, int k, const double* B, int n, double* C) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int p = 0; p < k; p++) s += A[i * k + p] * B[p * n + j];
            C[i * n + j] = s;
        }
}

static inline void orc_transpose(const double* A, int m, int n, double* At) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) At[j * m + i] = A[i * n + j];
}

static inline double orc_det3(const double* a) {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
           a[2] * (a[3] * a[7] - a[4] * a[6]);
}

static inline int orc_inv3(const double* a, double* inv) {
    double d = orc_det3(a);
    if (d == 0) return 0;
    double id = 1. / d;
    inv[0] = (a[4] * a[8] - a[5] * a[7]) * id;
    inv[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    inv[3] = (a[5] * a[6] - a[3] * a[8]) * id;
    inv[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    inv[6] = (a[3] * a[7] - a[4] * a[6]) * id;
    inv[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return 1;
}

/* Solve A x = b in place (A n x n row-major is destroyed, b becomes x).
 * Gaussian elimination with partial pivoting.  Returns 0 if singular. */
int orc_solve_lu(double* A, double* b, int n);
/* Cholesky solve of an SPD system (A destroyed).  Returns 0 if not SPD. */
int orc_solve_chol(double* A, double* b, int n);
/* Cyclic Jacobi eigen-decomposition of a symmetric matrix.  On return w holds
 * eigenvalues in descending order and V (n x n) the eigenvectors as ROWS. */
void orc_eig_sym(const double* A, int n, double* w, double* V);
/* Least-squares / minimum-norm solve of a symmetric system through its
 * eigen-decomposition, dropping |w| <= 2*eps*sum|w| the way SVD back-substitution does. */
int orc_solve_sym_svd(const double* A, const double* b, int n, double* x);
/* dispatch on ORC_SOLVE_*; A and b are preserved, x receives the solution. */
int orc_solve(const double* A, const double* b, int n, int method, double* x);
#endif
