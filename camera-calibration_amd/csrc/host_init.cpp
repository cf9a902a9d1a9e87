// host_init.cpp -- starting values for the LM loop, computed on the host and threaded over views.
//
// These are the steps OpenCV 4.1.0 performs before its LM loop inside calibrateCamera /
// stereoCalibrate (reference call sites monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317,
// binocular_chessboard.cpp:147-148 and :152-162):
//   cvInitIntrinsicParams2D        focal lengths from the vanishing points of per-view homographies
//   cvFindExtrinsicCameraParams2   planar pose from a homography, refined by a 6-parameter LM (<= 20 iterations)
//   findHomography (method 0)      normalised DLT followed by 10 LM iterations on the transfer error
// They decide where the LM loop starts (and so its iteration count), not where it ends.
#include "host_init.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "host_util.hpp"

namespace ccalib {

namespace {

struct M3 {
    double a[9];
    double& operator()(int r, int c) { return a[r * 3 + c]; }
    double operator()(int r, int c) const { return a[r * 3 + c]; }
};
M3 mul(const M3& x, const M3& y) {
    M3 z;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) z(i, j) = x(i, 0) * y(0, j) + x(i, 1) * y(1, j) + x(i, 2) * y(2, j);
    return z;
}
double det(const M3& m) {
    return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
           m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}
bool inverse(const M3& m, M3& inv) {
    double d = det(m);
    if (d == 0) return false;
    double s = 1 / d;
    inv(0, 0) = (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) * s;
    inv(0, 1) = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * s;
    inv(0, 2) = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * s;
    inv(1, 0) = (m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) * s;
    inv(1, 1) = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * s;
    inv(1, 2) = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * s;
    inv(2, 0) = (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)) * s;
    inv(2, 1) = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * s;
    inv(2, 2) = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * s;
    return true;
}

// Transfer residuals of the 8-parameter homography h (h22 = 1) and, optionally, their Jacobian.
void homography_eval(int n, const double* M, const double* m, const double* h, std::vector<double>& r,
                     std::vector<double>* J) {
    for (int i = 0; i < n; i++) {
        double X = M[2 * i], Y = M[2 * i + 1];
        double w = h[6] * X + h[7] * Y + 1.;
        w = std::fabs(w) > DBL_EPSILON ? 1. / w : 0;
        double xi = (h[0] * X + h[1] * Y + h[2]) * w, yi = (h[3] * X + h[4] * Y + h[5]) * w;
        r[2 * i] = xi - m[2 * i];
        r[2 * i + 1] = yi - m[2 * i + 1];
        if (J) {
            double* a = J->data() + (size_t)16 * i;
            double* b = a + 8;
            a[0] = X * w, a[1] = Y * w, a[2] = w, a[3] = a[4] = a[5] = 0, a[6] = -X * w * xi, a[7] = -Y * w * xi;
            b[0] = b[1] = b[2] = 0, b[3] = X * w, b[4] = Y * w, b[5] = w, b[6] = -X * w * yi, b[7] = -Y * w * yi;
        }
    }
}

void normal_eq(int rows, int n, const std::vector<double>& J, const std::vector<double>& r, std::vector<double>& A,
               std::vector<double>& v) {
    A.assign((size_t)n * n, 0.);
    v.assign(n, 0.);
    for (int i = 0; i < rows; i++) {
        const double* ji = J.data() + (size_t)i * n;
        for (int a = 0; a < n; a++) {
            for (int b = a; b < n; b++) A[a * n + b] += ji[a] * ji[b];
            v[a] += ji[a] * r[i];
        }
    }
    for (int a = 0; a < n; a++)
        for (int b = 0; b < a; b++) A[a * n + b] = A[b * n + a];
}
double sumsq(const std::vector<double>& v) {
    double s = 0;
    for (double x : v) s += x * x;
    return s;
}
double maxabs(const std::vector<double>& v) {
    double s = 0;
    for (double x : v) s = std::max(s, std::fabs(x));
    return s;
}

// cv::LMSolver::run with maxIters = 10 and eps = FLT_EPSILON, as findHomography uses it.
void homography_polish(int n, const double* M, const double* m, double h[8]) {
    const int lx = 8, rows = 2 * n, maxIters = 10;
    std::vector<double> x(h, h + 8), xd(8), r(rows), rd(rows), J((size_t)rows * 8), A, v, d(8), D(8);
    homography_eval(n, M, m, x.data(), r, &J);
    double S = sumsq(r);
    normal_eq(rows, lx, J, r, A, v);
    for (int i = 0; i < lx; i++) D[i] = A[i * lx + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    for (int iter = 0;;) {
        std::vector<double> Ap = A;
        for (int i = 0; i < lx; i++) Ap[i * lx + i] += lambda * D[i];
        d = v;
        solve_sym_svd(Ap, d, lx);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        homography_eval(n, M, m, xd.data(), rd, nullptr);
        double Sd = sumsq(rd), dS = 0, t = 0;
        for (int i = 0; i < lx; i++) {
            double Ad = 0;
            for (int j = 0; j < lx; j++) Ad += A[i * lx + j] * d[j];
            dS += d[i] * (2 * v[i] - Ad);
            t += d[i] * v[i];
        }
        double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                std::vector<double> w, V;
                eig_sym(A, lx, w, V);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) {  // largest |diagonal| of A^-1
                    double s = 0;
                    for (int k = 0; k < lx; k++)
                        if (std::fabs(w[k]) > DBL_EPSILON * 2 * std::fabs(w[0])) s += V[k * lx + i] * V[k * lx + i] / w[k];
                    maxval = std::max(maxval, std::fabs(s));
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            x = xd;
            homography_eval(n, M, m, x.data(), r, &J);
            normal_eq(rows, lx, J, r, A, v);
        }
        iter++;
        if (!(iter < maxIters && maxabs(d) >= FLT_EPSILON && maxabs(r) >= FLT_EPSILON)) break;
    }
    std::memcpy(h, x.data(), sizeof(double) * 8);
}

// cvProjectPoints2 on the host for the pose refinement: pixels and d(pixels)/d(r,t).
void project_pose(int n, const float* obj, const double r[3], const double t[3], const double A[9], const double k[14],
                  double* m, double* Je) {
    double R[9], dR[27];
    host_rodrigues_v2m(r, R, Je ? dR : nullptr);
    const double fx = A[0], fy = A[4], cx = A[2], cy = A[5];
    for (int i = 0; i < n; i++) {
        double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        double z = zc ? 1. / zc : 1;
        double x = xc * z, y = yc * z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cd = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double ic = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        double g = cd * ic;
        m[2 * i] = fx * (x * g + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4) + cx;
        m[2 * i + 1] = fy * (y * g + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4) + cy;
        if (!Je) continue;
        // symmetric 2x2 distortion Jacobian, then the 2x3 Jacobian w.r.t. the camera-frame point
        double gp = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * ic - cd * ic * ic * (k[5] + 2 * k[6] * r2 + 3 * k[7] * r4);
        double sx = k[8] + 2 * k[9] * r2, sy = k[10] + 2 * k[11] * r2;  // thin-prism terms
        double D00 = g + 2 * x * x * gp + 2 * k[2] * y + 6 * k[3] * x + 2 * x * sx;
        double D01 = a1 * gp + 2 * k[2] * x + 2 * k[3] * y + 2 * y * sx;
        double D10 = a1 * gp + 2 * k[2] * x + 2 * k[3] * y + 2 * x * sy;
        double D11 = g + 2 * y * y * gp + 6 * k[2] * y + 2 * k[3] * x + 2 * y * sy;
        double Tx[3] = {fx * z * D00, fx * z * D01, 0}, Ty[3] = {fy * z * D10, fy * z * D11, 0};
        Tx[2] = -(Tx[0] * x + Tx[1] * y);
        Ty[2] = -(Ty[0] * x + Ty[1] * y);
        double* ju = Je + (size_t)12 * i;
        double* jv = ju + 6;
        for (int j = 0; j < 3; j++) {
            const double* d = dR + 9 * j;
            double dX = d[0] * X + d[1] * Y + d[2] * Z, dY = d[3] * X + d[4] * Y + d[5] * Z,
                   dZ = d[6] * X + d[7] * Y + d[8] * Z;
            ju[j] = Tx[0] * dX + Tx[1] * dY + Tx[2] * dZ;
            jv[j] = Ty[0] * dX + Ty[1] * dY + Ty[2] * dZ;
            ju[3 + j] = Tx[j];
            jv[3 + j] = Ty[j];
        }
    }
}

}  // namespace

void host_rodrigues_v2m(const double r[3], double R[9], double* J) {
    double x = r[0], y = r[1], z = r[2], th = std::sqrt(x * x + y * y + z * z);
    if (th < DBL_EPSILON) {
        for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0);
        if (J) {
            std::memset(J, 0, 27 * sizeof(double));
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = std::cos(th), s = std::sin(th), c1 = 1 - c, it = 1 / th;
    x *= it, y *= it, z *= it;
    const double ax[3] = {x, y, z};
    double rrt[9], sk[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) rrt[i * 3 + j] = ax[i] * ax[j];
    for (int k = 0; k < 9; k++) R[k] = c * (k % 4 == 0) + c1 * rrt[k] + s * sk[k];
    if (!J) return;
    for (int i = 0; i < 3; i++) {
        double ri = ax[i], a0 = -s * ri, a1 = (s - 2 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
        double drrt[9] = {0}, dsk[9] = {0};
        for (int k = 0; k < 3; k++) drrt[i * 3 + k] += ax[k], drrt[k * 3 + i] += ax[k];
        int p = (i + 1) % 3, q = (i + 2) % 3;  // d[r]x/dr_i: +1 at (q,p), -1 at (p,q)
        dsk[q * 3 + p] = 1, dsk[p * 3 + q] = -1;
        for (int k = 0; k < 9; k++) J[i * 9 + k] = a0 * (k % 4 == 0) + a1 * rrt[k] + a2 * drrt[k] + a3 * sk[k] + a4 * dsk[k];
    }
}

void host_rodrigues_m2v(const double Rin[9], double r[3]) {
    M3 R;
    std::memcpy(R.a, Rin, sizeof(R.a));
    // nearest rotation (the U V^T of OpenCV's SVD) by Newton's polar iteration
    for (int it = 0; it < 20; it++) {
        M3 inv;
        if (!inverse(R, inv)) break;
        double chg = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double nv = 0.5 * (R(i, j) + inv(j, i));
                chg = std::max(chg, std::fabs(nv - R(i, j)));
                R(i, j) = nv;
            }
        if (chg < 1e-16) break;
    }
    double x = R(2, 1) - R(1, 2), y = R(0, 2) - R(2, 0), z = R(1, 0) - R(0, 1);
    double s = std::sqrt((x * x + y * y + z * z) * 0.25), c = (R(0, 0) + R(1, 1) + R(2, 2) - 1) * 0.5;
    c = std::min(1., std::max(-1., c));
    double th = std::acos(c);
    if (s < 1e-5) {
        if (c > 0)
            x = y = z = 0;
        else {
            x = std::sqrt(std::max((R(0, 0) + 1) * 0.5, 0.));
            y = std::sqrt(std::max((R(1, 1) + 1) * 0.5, 0.)) * (R(0, 1) < 0 ? -1. : 1.);
            z = std::sqrt(std::max((R(2, 2) + 1) * 0.5, 0.)) * (R(0, 2) < 0 ? -1. : 1.);
            if (std::fabs(x) < std::fabs(y) && std::fabs(x) < std::fabs(z) && (R(1, 2) > 0) != (y * z > 0)) z = -z;
            double sc = th / std::sqrt(x * x + y * y + z * z);
            x *= sc, y *= sc, z *= sc;
        }
    } else {
        double sc = th / (2 * s);
        x *= sc, y *= sc, z *= sc;
    }
    r[0] = x, r[1] = y, r[2] = z;
}

bool host_find_homography(int n, const double* src, const double* dst, double H[9]) {
    if (n < 4) return false;
    std::vector<double> M(2 * n), m(2 * n);
    for (int i = 0; i < 2 * n; i++) M[i] = (double)(float)src[i], m[i] = (double)(float)dst[i];  // OpenCV works on Point2f
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 2; c++) cM[c] += M[2 * i + c], cm[c] += m[2 * i + c];
    for (int c = 0; c < 2; c++) cM[c] /= n, cm[c] /= n;
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 2; c++) sM[c] += std::fabs(M[2 * i + c] - cM[c]), sm[c] += std::fabs(m[2 * i + c] - cm[c]);
    for (int c = 0; c < 2; c++) {
        if (std::fabs(sM[c]) < DBL_EPSILON || std::fabs(sm[c]) < DBL_EPSILON) return false;
        sM[c] = n / sM[c], sm[c] = n / sm[c];
    }
    std::vector<double> LtL(81, 0.), w, V;
    for (int i = 0; i < n; i++) {
        double x = (m[2 * i] - cm[0]) * sm[0], y = (m[2 * i + 1] - cm[1]) * sm[1];
        double X = (M[2 * i] - cM[0]) * sM[0], Y = (M[2 * i + 1] - cM[1]) * sM[1];
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    eig_sym(LtL, 9, w, V);
    M3 H0, invHn = {{1 / sm[0], 0, cm[0], 0, 1 / sm[1], cm[1], 0, 0, 1}},
           Hn2 = {{sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1}};
    std::memcpy(H0.a, V.data() + 72, sizeof(H0.a));  // eigenvector of the smallest eigenvalue
    M3 Hf = mul(mul(invHn, H0), Hn2);
    for (int i = 0; i < 9; i++) H[i] = Hf.a[i] / Hf.a[8];
    if (n > 4) {
        homography_polish(n, M.data(), m.data(), H);
        H[8] = 1;
    }
    return true;
}

void host_init_intrinsics(int nviews, const int* npts, const float* obj, const float* img, int width, int height,
                          double aspect, double Aout[9]) {
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
    a[2] = !width ? 0.5 : (width - 1) * 0.5;
    a[5] = !height ? 0.5 : (height - 1) * 0.5;
    std::vector<double> rowsA((size_t)4 * nviews), rowsB((size_t)2 * nviews);
    std::vector<size_t> ofs(nviews + 1, 0);
    for (int i = 0; i < nviews; i++) ofs[i + 1] = ofs[i] + npts[i];
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < nviews; i++) {
        int ni = npts[i];
        std::vector<double> M(2 * ni), m(2 * ni);
        for (int j = 0; j < ni; j++) {
            M[2 * j] = obj[3 * (ofs[i] + j)], M[2 * j + 1] = obj[3 * (ofs[i] + j) + 1];
            m[2 * j] = img[2 * (ofs[i] + j)], m[2 * j + 1] = img[2 * (ofs[i] + j) + 1];
        }
        double H[9] = {0};
        host_find_homography(ni, M.data(), m.data(), H);
        for (int c = 0; c < 3; c++) H[c] -= H[6 + c] * a[2], H[3 + c] -= H[6 + c] * a[5];
        double h[3], v[3], d1[3], d2[3], nn[4] = {0, 0, 0, 0};
        for (int j = 0; j < 3; j++) {
            double t0 = H[j * 3], t1 = H[j * 3 + 1];
            h[j] = t0, v[j] = t1, d1[j] = (t0 + t1) * 0.5, d2[j] = (t0 - t1) * 0.5;
            nn[0] += t0 * t0, nn[1] += t1 * t1, nn[2] += d1[j] * d1[j], nn[3] += d2[j] * d2[j];
        }
        for (double& q : nn) q = 1. / std::sqrt(q);
        for (int j = 0; j < 3; j++) h[j] *= nn[0], v[j] *= nn[1], d1[j] *= nn[2], d2[j] *= nn[3];
        double* Ap = rowsA.data() + (size_t)4 * i;
        Ap[0] = h[0] * v[0], Ap[1] = h[1] * v[1], Ap[2] = d1[0] * d2[0], Ap[3] = d1[1] * d2[1];
        rowsB[2 * i] = -h[2] * v[2], rowsB[2 * i + 1] = -d1[2] * d2[2];
    }
    std::vector<double> N(4, 0.), rhs(2, 0.);
    for (int i = 0; i < 2 * nviews; i++) {
        double a0 = rowsA[2 * i], a1 = rowsA[2 * i + 1];
        N[0] += a0 * a0, N[1] += a0 * a1, N[3] += a1 * a1;
        rhs[0] += a0 * rowsB[i], rhs[1] += a1 * rowsB[i];
    }
    N[2] = N[1];
    solve_sym_svd(N, rhs, 2);
    a[0] = std::sqrt(std::fabs(1. / rhs[0]));
    a[4] = std::sqrt(std::fabs(1. / rhs[1]));
    if (aspect != 0) {
        double tf = (a[0] + a[4]) / (aspect + 1.);
        a[0] = aspect * tf, a[4] = tf;
    }
    std::memcpy(Aout, a, sizeof(a));
}

bool host_find_extrinsic(int n, const float* obj, const float* img, const double A[9], const double k[14], double r[3],
                         double t[3]) {
    // undistorted normalised image points: 5 fixed-point iterations (cvUndistortPoints)
    std::vector<double> mn(2 * n);
    for (int i = 0; i < n; i++) {
        double x0 = (img[2 * i] - A[2]) / A[0], y0 = (img[2 * i + 1] - A[5]) / A[4], x = x0, y = y0;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icd = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icd, y = (y0 - dy) * icd;
        }
        mn[2 * i] = x, mn[2 * i + 1] = y;
    }
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) Mc[c] += obj[3 * i + c];
    for (double& q : Mc) q /= n;
    std::vector<double> MM(9, 0.), W, V;
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) MM[a * 3 + b] += (obj[3 * i + a] - Mc[a]) * (obj[3 * i + b] - Mc[b]);
    eig_sym(MM, 3, W, V);
    if (!(W[2] / W[1] < 1e-3)) return false;  // only planar rigs (every board of the reference is planar)
    M3 Rt;
    std::memcpy(Rt.a, V.data(), sizeof(Rt.a));
    if (V[2] * V[2] + V[5] * V[5] < 1e-10) Rt = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    if (det(Rt) < 0)
        for (double& q : Rt.a) q = -q;
    double Tt[3];
    for (int i = 0; i < 3; i++) Tt[i] = -(Rt(i, 0) * Mc[0] + Rt(i, 1) * Mc[1] + Rt(i, 2) * Mc[2]);
    std::vector<double> Mxy(2 * n);
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 2; c++)
            Mxy[2 * i + c] = Rt(c, 0) * obj[3 * i] + Rt(c, 1) * obj[3 * i + 1] + Rt(c, 2) * obj[3 * i + 2] + Tt[c];
    M3 H, Rm;
    bool ok = host_find_homography(n, Mxy.data(), mn.data(), H.a);
    for (double q : H.a) ok = ok && std::isfinite(q);
    double p[6];
    if (ok) {
        double n1 = std::sqrt(H(0, 0) * H(0, 0) + H(1, 0) * H(1, 0) + H(2, 0) * H(2, 0));
        double n2 = std::sqrt(H(0, 1) * H(0, 1) + H(1, 1) * H(1, 1) + H(2, 1) * H(2, 1));
        double tt[3];
        for (int i = 0; i < 3; i++) {
            tt[i] = H(i, 2) * (2. / std::max(n1 + n2, DBL_EPSILON));
            H(i, 0) /= std::max(n1, DBL_EPSILON);
            H(i, 1) /= std::max(n2, DBL_EPSILON);
        }
        H(0, 2) = H(1, 0) * H(2, 1) - H(2, 0) * H(1, 1);
        H(1, 2) = H(2, 0) * H(0, 1) - H(0, 0) * H(2, 1);
        H(2, 2) = H(0, 0) * H(1, 1) - H(1, 0) * H(0, 1);
        double rr[3];
        host_rodrigues_m2v(H.a, rr);
        host_rodrigues_v2m(rr, H.a, nullptr);
        for (int i = 0; i < 3; i++) p[3 + i] = H(i, 0) * Tt[0] + H(i, 1) * Tt[1] + H(i, 2) * Tt[2] + tt[i];
        Rm = mul(H, Rt);
    } else {
        Rm = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
        p[3] = p[4] = p[5] = 0;
    }
    host_rodrigues_m2v(Rm.a, p);

    // CvLevMarq(6 parameters, explicit J and err) with criteria (COUNT+EPS, 20, FLT_EPSILON)
    const int rows = 2 * n, max_iter = 20;
    std::vector<double> J((size_t)rows * 6), err(rows), proj(rows), JtJ, Jte, prev(6);
    auto eval = [&](bool wantJ) {
        project_pose(n, obj, p, p + 3, A, k, proj.data(), wantJ ? J.data() : nullptr);
        for (int i = 0; i < rows; i++) err[i] = proj[i] - img[i];
    };
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX, errNorm = 0;
    auto step = [&]() {
        double lambda = std::exp(lambdaLg10 * std::log(10.));
        std::vector<double> Aq = JtJ, x = Jte;
        for (int i = 0; i < 6; i++) Aq[i * 6 + i] *= 1. + lambda;
        solve_sym_svd(Aq, x, 6);
        for (int i = 0; i < 6; i++) p[i] = prev[i] - x[i];
    };
    eval(true);
    for (;;) {
        normal_eq(rows, 6, J, err, JtJ, Jte);
        std::memcpy(prev.data(), p, sizeof(double) * 6);
        step();
        if (iters == 0) prevErrNorm = std::sqrt(sumsq(err));
        eval(false);
        for (;;) {
            errNorm = std::sqrt(sumsq(err));
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) {
                step();
                eval(false);
                continue;
            }
            break;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double num = 0, den = 0;
        for (int i = 0; i < 6; i++) num += (p[i] - prev[i]) * (p[i] - prev[i]), den += prev[i] * prev[i];
        if (++iters >= max_iter || std::sqrt(num) / std::sqrt(den) < FLT_EPSILON) break;
        prevErrNorm = errNorm;
        eval(true);
    }
    std::memcpy(r, p, 3 * sizeof(double));
    std::memcpy(t, p + 3, 3 * sizeof(double));
    return true;
}

void host_init_poses(int nviews, const int* npts, const float* obj, bool shared_board, const float* img,
                     const double A[9], const double k[14], double* rt) {
    std::vector<size_t> ofs(nviews + 1, 0);
    for (int i = 0; i < nviews; i++) ofs[i + 1] = ofs[i] + npts[i];
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < nviews; i++) {
        const float* o = shared_board ? obj : obj + 3 * ofs[i];
        double* q = rt + (size_t)6 * i;
        if (!host_find_extrinsic(npts[i], o, img + 2 * ofs[i], A, k, q, q + 3))
            for (int c = 0; c < 6; c++) q[c] = 0;
    }
}

}  // namespace ccalib
