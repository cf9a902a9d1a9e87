/*
 * orc_init.c -- starting values for the LM loop.  TEST INFRASTRUCTURE ONLY; parity unpinned.
 *
 * Restates, from knowledge of OpenCV 4.1.0:
 *   modules/calib3d/src/fundam.cpp       findHomography (method 0): normalised DLT +
 *                                         HomographyRefineCallback under cv::LMSolver(10)
 *   modules/calib3d/src/calibration.cpp  cvInitIntrinsicParams2D, cvFindExtrinsicCameraParams2
 *   modules/imgproc/src/undistort.cpp    cvUndistortPoints (5 fixed-point iterations)
 * Reached by the reference inside calibrateCamera (monocular_circle.cpp:93-94,
 * monocular_chessboard.cpp:315-317), through initCameraMatrix2D
 * (binocular_chessboard.cpp:147-148, binocular_circle.cpp:147-148) and inside
 * stereoCalibrate (binocular_chessboard.cpp:152-162, binocular_circle.cpp:153-163).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "calib_oracle.h"
#include "orc_levmarq.h"
#include "orc_math.h"

/* ---- homography ------------------------------------------------------------------------ */

static int homography_dlt(int n, const double* M, const double* m, double H[9]) {
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < n; i++) {
        cmx += m[2 * i], cmy += m[2 * i + 1];
        cMx += M[2 * i], cMy += M[2 * i + 1];
    }
    cmx /= n, cmy /= n, cMx /= n, cMy /= n;
    for (int i = 0; i < n; i++) {
        smx += fabs(m[2 * i] - cmx), smy += fabs(m[2 * i + 1] - cmy);
        sMx += fabs(M[2 * i] - cMx), sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON ||
        fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = n / smx, smy = n / smy, sMx = n / sMx, sMy = n / sMy;
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    double LtL[81];
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < n; i++) {
        double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    double W[9], V[81];
    orc_eig_sym(LtL, 9, W, V);
    const double* H0 = V + 72; /* eigenvector of the smallest eigenvalue */
    double Htemp[9], Hfull[9];
    orc_matmul(invHnorm, 3, 3, H0, 3, Htemp);
    orc_matmul(Htemp, 3, 3, Hnorm2, 3, Hfull);
    for (int i = 0; i < 9; i++) H[i] = Hfull[i] / Hfull[8];
    return 1;
}

static void homography_residual(int n, const double* M, const double* m, const double* h, double* err,
                                double* J) {
    for (int i = 0; i < n; i++) {
        double Mx = M[2 * i], My = M[2 * i + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        err[2 * i] = xi - m[2 * i];
        err[2 * i + 1] = yi - m[2 * i + 1];
        if (J) {
            double* jx = J + (size_t)(2 * i) * 8;
            double* jy = jx + 8;
            jx[0] = Mx * ww, jx[1] = My * ww, jx[2] = ww, jx[3] = jx[4] = jx[5] = 0;
            jx[6] = -Mx * ww * xi, jx[7] = -My * ww * xi;
            jy[0] = jy[1] = jy[2] = 0, jy[3] = Mx * ww, jy[4] = My * ww, jy[5] = ww;
            jy[6] = -Mx * ww * yi, jy[7] = -My * ww * yi;
        }
    }
}

static void jtj8(int rows, const double* J, const double* r, double* A, double* v) {
    for (int a = 0; a < 8; a++) {
        for (int b = 0; b < 8; b++) {
            double s = 0;
            for (int i = 0; i < rows; i++) s += J[i * 8 + a] * J[i * 8 + b];
            A[a * 8 + b] = s;
        }
        double s = 0;
        for (int i = 0; i < rows; i++) s += J[i * 8 + a] * r[i];
        v[a] = s;
    }
}

static double sumsq(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return s;
}
static double norminf(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s = fmax(s, fabs(v[i]));
    return s;
}

/* cv::LMSolver::run as used by findHomography (maxIters = 10, eps = FLT_EPSILON). */
static void homography_refine(int n, const double* M, const double* m, double h[8]) {
    const int lx = 8, maxIters = 10;
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    int rows = 2 * n;
    double* r = (double*)malloc(sizeof(double) * rows);
    double* rd = (double*)malloc(sizeof(double) * rows);
    double* J = (double*)malloc(sizeof(double) * rows * 8);
    double x[8], xd[8], A[64], Ap[64], v[8], d[8], D[8], temp_d[8];
    memcpy(x, h, sizeof(x));
    homography_residual(n, M, m, x, r, J);
    double S = sumsq(r, rows);
    jtj8(rows, J, r, A, v);
    for (int i = 0; i < lx; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(A));
        for (int i = 0; i < lx; i++) Ap[i * 8 + i] += lambda * D[i];
        orc_solve_sym_svd(Ap, v, lx, d);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        homography_residual(n, M, m, xd, rd, 0);
        double Sd = sumsq(rd, rows);
        for (int i = 0; i < lx; i++) {
            double s = 0;
            for (int j = 0; j < lx; j++) s += A[i * 8 + j] * d[j];
            temp_d[i] = -s + 2 * v[i];
        }
        double dS = 0;
        for (int i = 0; i < lx; i++) dS += d[i] * temp_d[i];
        double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < lx; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = fmin(fmax(nu, 2.), 10.);
            if (lambda == 0) {
                /* invert(A, Ap, DECOMP_EIG): only the largest |diagonal| of the inverse is used */
                double w[8], V[64], maxval = DBL_EPSILON;
                orc_eig_sym(A, 8, w, V);
                for (int i = 0; i < lx; i++) {
                    double s = 0;
                    for (int k = 0; k < lx; k++)
                        if (fabs(w[k]) > DBL_EPSILON * 2 * fabs(w[0])) s += V[k * 8 + i] * V[k * 8 + i] / w[k];
                    maxval = fmax(maxval, fabs(s));
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof(x));
            homography_residual(n, M, m, x, r, J);
            jtj8(rows, J, r, A, v);
        }
        iter++;
        int proceed = iter < maxIters && norminf(d, lx) >= epsx && norminf(r, rows) >= epsf;
        if (!proceed) break;
    }
    memcpy(h, x, sizeof(x));
    free(r);
    free(rd);
    free(J);
}

int orc_find_homography(int n, const double* src, const double* dst, double H[9]) {
    if (n < 4) return 0;
    /* findHomography converts both point sets to 32-bit floats first */
    double* M = (double*)malloc(sizeof(double) * 2 * n);
    double* m = (double*)malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < 2 * n; i++) {
        M[i] = (double)(float)src[i];
        m[i] = (double)(float)dst[i];
    }
    int ok = homography_dlt(n, M, m, H);
    if (ok && n > 4) {
        homography_refine(n, M, m, H); /* first 8 entries are the parameters, H[8] stays 1 */
        H[8] = 1;
    }
    free(M);
    free(m);
    return ok;
}

/* ---- intrinsics ------------------------------------------------------------------------ */

int orc_init_intrinsic_params_2d(int nviews, const int* npts, const double* obj, const double* img,
                                 int width, int height, double aspect, double Aout[9]) {
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
    a[2] = (!width) ? 0.5 : (width - 1) * 0.5;
    a[5] = (!height) ? 0.5 : (height - 1) * 0.5;
    if (orc_variants & ORC_VARIANT_PP_HALF_SIZE) { /* what-if: principal point seeded at w/2, h/2 */
        a[2] = width * 0.5;
        a[5] = height * 0.5;
    }
    double* Am = (double*)malloc(sizeof(double) * 4 * nviews);
    double* bm = (double*)malloc(sizeof(double) * 2 * nviews);
    int pos = 0;
    for (int i = 0; i < nviews; i++) {
        int ni = npts[i];
        double* Mxy = (double*)malloc(sizeof(double) * 2 * ni);
        for (int j = 0; j < ni; j++) {
            Mxy[2 * j] = obj[(size_t)(pos + j) * 3];
            Mxy[2 * j + 1] = obj[(size_t)(pos + j) * 3 + 1];
        }
        double H[9];
        orc_find_homography(ni, Mxy, img + (size_t)pos * 2, H);
        free(Mxy);
        H[0] -= H[6] * a[2], H[1] -= H[7] * a[2], H[2] -= H[8] * a[2];
        H[3] -= H[6] * a[5], H[4] -= H[7] * a[5], H[5] -= H[8] * a[5];
        double h[3], v[3], d1[3], d2[3], nn[4] = {0, 0, 0, 0};
        for (int j = 0; j < 3; j++) {
            double t0 = H[j * 3], t1 = H[j * 3 + 1];
            h[j] = t0, v[j] = t1;
            d1[j] = (t0 + t1) * 0.5;
            d2[j] = (t0 - t1) * 0.5;
            nn[0] += t0 * t0, nn[1] += t1 * t1;
            nn[2] += d1[j] * d1[j], nn[3] += d2[j] * d2[j];
        }
        for (int j = 0; j < 4; j++) nn[j] = 1. / sqrt(nn[j]);
        for (int j = 0; j < 3; j++) {
            h[j] *= nn[0], v[j] *= nn[1];
            d1[j] *= nn[2], d2[j] *= nn[3];
        }
        double* Ap = Am + i * 4;
        double* bp = bm + i * 2;
        Ap[0] = h[0] * v[0], Ap[1] = h[1] * v[1];
        Ap[2] = d1[0] * d2[0], Ap[3] = d1[1] * d2[1];
        bp[0] = -h[2] * v[2], bp[1] = -d1[2] * d2[2];
        pos += ni;
    }
    /* cvSolve(A, b, f, CV_NORMAL + CV_SVD): 2x2 normal equations */
    double N[4] = {0, 0, 0, 0}, rhs[2] = {0, 0}, f[2];
    for (int i = 0; i < 2 * nviews; i++) {
        double a0 = Am[i * 2], a1 = Am[i * 2 + 1];
        N[0] += a0 * a0, N[1] += a0 * a1, N[3] += a1 * a1;
        rhs[0] += a0 * bm[i], rhs[1] += a1 * bm[i];
    }
    N[2] = N[1];
    orc_solve_sym_svd(N, rhs, 2, f);
    a[0] = sqrt(fabs(1. / f[0]));
    a[4] = sqrt(fabs(1. / f[1]));
    if (aspect != 0) {
        double tf = (a[0] + a[4]) / (aspect + 1.);
        a[0] = aspect * tf;
        a[4] = tf;
    }
    memcpy(Aout, a, sizeof(a));
    free(Am);
    free(bm);
    return 1;
}

/* ---- per-view pose --------------------------------------------------------------------- */

static void undistort_points(int n, const double* src, const double A[9], const double k[14], double* dst) {
    double fx = A[0], fy = A[4], ifx = 1. / fx, ify = 1. / fy, cx = A[2], cy = A[5];
    for (int i = 0; i < n; i++) {
        double x = (src[2 * i] - cx) * ifx, y = (src[2 * i + 1] - cy) * ify;
        double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        dst[2 * i] = x;
        dst[2 * i + 1] = y;
    }
}

int orc_find_extrinsic(int n, const double* obj, const double* img, const double A[9],
                       const double k[14], double r[3], double t[3]) {
    const int max_iter = 20;
    double* mn = (double*)malloc(sizeof(double) * 2 * n);
    undistort_points(n, img, A, k, mn);

    double Mc[3] = {0, 0, 0}, MM[9];
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) Mc[c] += obj[i * 3 + c];
    for (int c = 0; c < 3; c++) Mc[c] /= n;
    memset(MM, 0, sizeof(MM));
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) MM[a * 3 + b] += (obj[i * 3 + a] - Mc[a]) * (obj[i * 3 + b] - Mc[b]);
    double W[3], V[9]; /* V rows = right singular vectors (CV_SVD_V_T) */
    orc_eig_sym(MM, 3, W, V);

    double R[9], rr[3], tt[3];
    if (W[2] / W[1] < 1e-3) {
        /* planar structure */
        double Rt[9], Tt[3], H[9];
        memcpy(Rt, V, sizeof(Rt));
        if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
            memset(Rt, 0, sizeof(Rt));
            Rt[0] = Rt[4] = Rt[8] = 1;
        }
        if (orc_det3(Rt) < 0)
            for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
        for (int i = 0; i < 3; i++) Tt[i] = -(Rt[i * 3] * Mc[0] + Rt[i * 3 + 1] * Mc[1] + Rt[i * 3 + 2] * Mc[2]);
        double* Mxy = (double*)malloc(sizeof(double) * 2 * n);
        for (int i = 0; i < n; i++) {
            const double* s = obj + i * 3;
            Mxy[2 * i] = Rt[0] * s[0] + Rt[1] * s[1] + Rt[2] * s[2] + Tt[0];
            Mxy[2 * i + 1] = Rt[3] * s[0] + Rt[4] * s[1] + Rt[5] * s[2] + Tt[1];
        }
        int ok = orc_find_homography(n, Mxy, mn, H);
        free(Mxy);
        for (int i = 0; i < 9; i++)
            if (!isfinite(H[i])) ok = 0;
        if (ok) {
            double h1n = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
            double h2n = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
            double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON);
            double s3 = 2. / fmax(h1n + h2n, DBL_EPSILON);
            for (int i = 0; i < 3; i++) {
                tt[i] = H[i * 3 + 2] * s3;
                H[i * 3] *= s1;
                H[i * 3 + 1] *= s2;
            }
            /* h3 = h1 x h2 */
            H[2] = H[3] * H[7] - H[6] * H[4];
            H[5] = H[6] * H[1] - H[0] * H[7];
            H[8] = H[0] * H[4] - H[3] * H[1];
            orc_rodrigues_m2v(H, rr, 0);
            orc_rodrigues_v2m(rr, H, 0);
            double t2[3];
            for (int i = 0; i < 3; i++)
                t2[i] = H[i * 3] * Tt[0] + H[i * 3 + 1] * Tt[1] + H[i * 3 + 2] * Tt[2] + tt[i];
            memcpy(tt, t2, sizeof(t2));
            orc_matmul(H, 3, 3, Rt, 3, R);
        } else {
            memset(R, 0, sizeof(R));
            R[0] = R[4] = R[8] = 1;
            tt[0] = tt[1] = tt[2] = 0;
        }
        orc_rodrigues_m2v(R, rr, 0);
    } else {
        free(mn);
        return 0; /* non-planar DLT start is not restated: the reference only uses planar boards */
    }

    /* refine with CvLevMarq(6, 2n, (EPS+ITER, 20, FLT_EPSILON), completeSymm = true) */
    orc_levmarq lm;
    orc_lm_init(&lm, 6, 2 * n, ORC_TERM_COUNT | ORC_TERM_EPS, max_iter, FLT_EPSILON, 1);
    memcpy(lm.param, rr, sizeof(rr));
    memcpy(lm.param + 3, tt, sizeof(tt));
    double fc[4] = {A[0], A[4], A[2], A[5]};
    double* proj = (double*)malloc(sizeof(double) * 2 * n);
    for (;;) {
        int wJ, wE;
        int proceed = orc_lm_update(&lm, &wJ, &wE);
        if (!proceed || !wE) break;
        orc_project_points(n, obj, lm.param, lm.param + 3, fc, k, 0., proj, wJ ? lm.J : 0, 0);
        for (int i = 0; i < 2 * n; i++) lm.err[i] = proj[i] - img[i];
    }
    memcpy(r, lm.param, 3 * sizeof(double));
    memcpy(t, lm.param + 3, 3 * sizeof(double));
    free(proj);
    orc_lm_free(&lm);
    free(mn);
    return 1;
}
