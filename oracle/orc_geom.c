/*
 * orc_geom.c -- Rodrigues, pose composition and point projection with analytic
 * Jacobians.  TEST INFRASTRUCTURE ONLY (see calib_oracle.h; parity unpinned).
 *
 * Restates, from knowledge of OpenCV 4.1.0 modules/calib3d/src/calibration.cpp:
 *   cvRodrigues2, cvComposeRT (with cvCalcMatMulDeriv), cvProjectPoints2Internal.
 * Reached by the reference through monocular_circle.cpp:18-19 (projectPoints)
 * and inside every calibrateCamera / stereoCalibrate call listed in the header.
 */
#include <float.h>
#include <math.h>
#include <string.h>

#include "calib_oracle.h"
#include "orc_math.h"

void orc_rodrigues_v2m(const double rin[3], double R[9], double J[27]) {
    double rx = rin[0], ry = rin[1], rz = rin[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        memset(R, 0, 9 * sizeof(double));
        R[0] = R[4] = R[8] = 1.0;
        if (J) {
            memset(J, 0, 27 * sizeof(double));
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c;
    double itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0,       0,  rz, 0,  0,       //
                           0,       rx, 0,  rx, ry + ry, rz, 0,  rz, 0,       //
                           0,       0,  rx, 0,  0,       ry, rx, ry, rz + rz};
        static const double d_r_x_[27] = {0, 0,  0, 0, 0, -1, 0,  1, 0,  //
                                          0, 0,  1, 0, 0, 0,  -1, 0, 0,  //
                                          0, -1, 0, 1, 0, 0,  0,  0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] +
                               a4 * d_r_x_[i * 9 + k];
        }
    }
}

/* U*Vt of the SVD of a near-rotation = its orthogonal polar factor; Newton's
 * iteration X <- (X + X^-T)/2 converges to it quadratically. */
static void polar_orthogonalise(double R[9]) {
    for (int it = 0; it < 20; it++) {
        double inv[9];
        if (!orc_inv3(R, inv)) return;
        double d = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double nv = 0.5 * (R[i * 3 + j] + inv[j * 3 + i]);
                d = fmax(d, fabs(nv - R[i * 3 + j]));
                R[i * 3 + j] = nv;
            }
        if (d < 1e-16) break;
    }
}

void orc_rodrigues_m2v(const double Rin[9], double rout[3], double J[27]) {
    double R[9];
    memcpy(R, Rin, sizeof(R));
    for (int k = 0; k < 9; k++)
        if (!(R[k] > -100 && R[k] < 100)) { /* checkRange failure */
            rout[0] = rout[1] = rout[2] = 0;
            if (J) memset(J, 0, 27 * sizeof(double));
            return;
        }
    polar_orthogonalise(R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = sqrt(fmax(t, 0.));
            t = (R[4] + 1) * 0.5;
            ry = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
        if (J) {
            memset(J, 0, 27 * sizeof(double));
            if (c > 0) {
                J[5] = J[15] = J[19] = -0.5;
                J[7] = J[11] = J[21] = 0.5;
            }
        }
    } else {
        double vth = 1 / (2 * s);
        if (J) {
            double dtheta_dtr = -1. / s;
            double dvth_dtheta = -vth * c / s;
            double d1 = 0.5 * dvth_dtheta * dtheta_dtr;
            double d2 = 0.5 * dtheta_dtr;
            double dvardR[5 * 9] = {0,  0, 0,  0,  0,  1, 0, -1, 0,   //
                                    0,  0, -1, 0,  0,  0, 1, 0,  0,   //
                                    0,  1, 0,  -1, 0,  0, 0, 0,  0,   //
                                    d1, 0, 0,  0,  d1, 0, 0, 0,  d1,  //
                                    d2, 0, 0,  0,  d2, 0, 0, 0,  d2};
            double dvar2dvar[4 * 5] = {vth, 0, 0, rx, 0, 0, vth, 0, ry, 0, 0, 0, vth, rz, 0, 0, 0, 0, 0, 1};
            double domegadvar2[3 * 4] = {theta, 0, 0, rx * vth, 0, theta, 0, ry * vth, 0, 0, theta, rz * vth};
            double t0[3 * 5];
            orc_matmul(domegadvar2, 3, 4, dvar2dvar, 5, t0);
            orc_matmul(t0, 3, 5, dvardR, 9, J);
            /* transpose every row of J (rows treated as 3x3 matrices) */
            for (int i = 0; i < 3; i++) {
                double* q = J + 9 * i;
                double t;
                t = q[1], q[1] = q[3], q[3] = t;
                t = q[2], q[2] = q[6], q[6] = t;
                t = q[5], q[5] = q[7], q[7] = t;
            }
        }
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    rout[0] = rx;
    rout[1] = ry;
    rout[2] = rz;
}

/* cvCalcMatMulDeriv for C = A(3x3) * B(3xN), N in {1,3}: dCdA is (3N)x9, dCdB is (3N)x(3N). */
static void matmul_deriv(const double* A, const double* B, int N, double* dCdA, double* dCdB) {
    int M = 3, L = 3;
    for (int i1 = 0; i1 < M; i1++)
        for (int i2 = 0; i2 < N; i2++) {
            int row = i1 * N + i2;
            if (dCdA) {
                double* d = dCdA + row * (M * L);
                for (int j = 0; j < M * L; j++) d[j] = 0;
                for (int j = 0; j < L; j++) d[i1 * L + j] = B[j * N + i2];
            }
            if (dCdB) {
                double* d = dCdB + row * (L * N);
                for (int j = 0; j < L * N; j++) d[j] = 0;
                for (int j = 0; j < L; j++) d[j * N + i2] = A[i1 * L + j];
            }
        }
}

void orc_compose_rt(const double r1[3], const double t1[3], const double r2[3], const double t2[3],
                    double r3[3], double t3[3], double dr3dr1[9], double dr3dr2[9],
                    double dt3dr2[9], double dt3dt1[9], double dt3dt2[9]) {
    double R1[9], R2[9], J1[27], J2[27]; /* J: 3x9 */
    double dR1dr1[27], dR2dr2[27];        /* 9x3 */
    orc_rodrigues_v2m(r1, R1, J1);
    orc_rodrigues_v2m(r2, R2, J2);
    orc_transpose(J1, 3, 9, dR1dr1);
    orc_transpose(J2, 3, 9, dR2dr2);

    double R3[9], dR3dR1[81], dR3dR2[81], dr3dR3[27], W1[27], W2[9];
    orc_matmul(R2, 3, 3, R1, 3, R3);
    matmul_deriv(R2, R1, 3, dR3dR2, dR3dR1);
    double r3l[3];
    orc_rodrigues_m2v(R3, r3l, dr3dR3);
    if (r3) memcpy(r3, r3l, sizeof(r3l));
    if (dr3dr1) {
        orc_matmul(dr3dR3, 3, 9, dR3dR1, 9, W1);
        orc_matmul(W1, 3, 9, dR1dr1, 3, W2);
        memcpy(dr3dr1, W2, sizeof(W2));
    }
    if (dr3dr2) {
        orc_matmul(dr3dR3, 3, 9, dR3dR2, 9, W1);
        orc_matmul(W1, 3, 9, dR2dr2, 3, W2);
        memcpy(dr3dr2, W2, sizeof(W2));
    }
    double t3l[3];
    for (int i = 0; i < 3; i++)
        t3l[i] = R2[i * 3] * t1[0] + R2[i * 3 + 1] * t1[1] + R2[i * 3 + 2] * t1[2] + t2[i];
    if (t3) memcpy(t3, t3l, sizeof(t3l));
    if (dt3dr2 || dt3dt1) {
        double dxdR2[27], dxdt1[9], W3[9];
        matmul_deriv(R2, t1, 1, dxdR2, dxdt1);
        if (dt3dr2) {
            orc_matmul(dxdR2, 3, 9, dR2dr2, 3, W3);
            memcpy(dt3dr2, W3, sizeof(W3));
        }
        if (dt3dt1) memcpy(dt3dt1, dxdt1, sizeof(dxdt1));
    }
    if (dt3dt2) {
        memset(dt3dt2, 0, 9 * sizeof(double));
        dt3dt2[0] = dt3dt2[4] = dt3dt2[8] = 1;
    }
}

void orc_project_points(int n, const double* obj, const double r[3], const double t[3],
                        const double fc[4], const double k[14], double aspect, double* m,
                        double* Je, double* Ji) {
    double R[9], dRdr[27];
    int calc = (Je != 0) || (Ji != 0);
    orc_rodrigues_v2m(r, R, calc ? dRdr : 0);
    double fx = fc[0], fy = fc[1], cx = fc[2], cy = fc[3];
    int fixedAspect = aspect != 0.0;
    if (fixedAspect) fx = fy * aspect;
    /* tilt (k[12], k[13]) is required to be zero: matTilt = I, so vecTilt = (xd0, yd0, 1),
     * invProj = 1 and dMatTilt = I exactly. */
    for (int i = 0; i < n; i++) {
        double X = obj[i * 3], Y = obj[i * 3 + 1], Z = obj[i * 3 + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4;
        double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
        if (m) {
            m[i * 2] = xd * fx + cx;
            m[i * 2 + 1] = yd * fy + cy;
        }
        if (Ji) {
            double* jx = Ji + (size_t)(2 * i) * 18;
            double* jy = jx + 18;
            memset(jx, 0, 36 * sizeof(double));
            if (fixedAspect) {
                jx[0] = 0, jx[1] = xd * aspect;
                jy[0] = 0, jy[1] = yd;
            } else {
                jx[0] = xd, jx[1] = 0;
                jy[0] = 0, jy[1] = yd;
            }
            jx[2] = 1, jx[3] = 0;
            jy[2] = 0, jy[3] = 1;
            double* kx = jx + 4;
            double* ky = jy + 4;
            kx[0] = fx * (x * icdist2 * r2), ky[0] = fy * (y * icdist2 * r2);
            kx[1] = fx * (x * icdist2 * r4), ky[1] = fy * (y * icdist2 * r4);
            kx[2] = fx * a1, ky[2] = fy * a3;
            kx[3] = fx * a2, ky[3] = fy * a1;
            kx[4] = fx * (x * icdist2 * r6), ky[4] = fy * (y * icdist2 * r6);
            kx[5] = fx * (x * cdist * (-icdist2) * icdist2 * r2);
            ky[5] = fy * (y * cdist * (-icdist2) * icdist2 * r2);
            kx[6] = fx * (x * cdist * (-icdist2) * icdist2 * r4);
            ky[6] = fy * (y * cdist * (-icdist2) * icdist2 * r4);
            kx[7] = fx * (x * cdist * (-icdist2) * icdist2 * r6);
            ky[7] = fy * (y * cdist * (-icdist2) * icdist2 * r6);
            kx[8] = fx * r2, ky[8] = 0;   /* s1 */
            kx[9] = fx * r4, ky[9] = 0;   /* s2 */
            kx[10] = 0, ky[10] = fy * r2; /* s3 */
            kx[11] = 0, ky[11] = fy * r4; /* s4 */
            /* tau columns (12, 13) left zero: always masked out here */
        }
        if (Je) {
            double* ex = Je + (size_t)(2 * i) * 6;
            double* ey = ex + 6;
            double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) {
                double dr2dt = 2 * x * dxdt[j] + 2 * y * dydt[j];
                double dcdist_dt = k[0] * dr2dt + 2 * k[1] * r2 * dr2dt + 3 * k[4] * r4 * dr2dt;
                double dicdist2_dt =
                    -icdist2 * icdist2 * (k[5] * dr2dt + 2 * k[6] * r2 * dr2dt + 3 * k[7] * r4 * dr2dt);
                double da1dt = 2 * (x * dydt[j] + y * dxdt[j]);
                double dmxdt = (dxdt[j] * cdist * icdist2 + x * dcdist_dt * icdist2 +
                                x * cdist * dicdist2_dt + k[2] * da1dt +
                                k[3] * (dr2dt + 4 * x * dxdt[j]) + k[8] * dr2dt + 2 * r2 * k[9] * dr2dt);
                double dmydt = (dydt[j] * cdist * icdist2 + y * dcdist_dt * icdist2 +
                                y * cdist * dicdist2_dt + k[2] * (dr2dt + 4 * y * dydt[j]) +
                                k[3] * da1dt + k[10] * dr2dt + 2 * r2 * k[11] * dr2dt);
                ex[3 + j] = fx * dmxdt;
                ey[3 + j] = fy * dmydt;
            }
            double dx0dr[3], dy0dr[3], dz0dr[3];
            for (int j = 0; j < 3; j++) {
                const double* d = dRdr + 9 * j;
                dx0dr[j] = X * d[0] + Y * d[1] + Z * d[2];
                dy0dr[j] = X * d[3] + Y * d[4] + Z * d[5];
                dz0dr[j] = X * d[6] + Y * d[7] + Z * d[8];
            }
            for (int j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                double dr2dr = 2 * x * dxdr + 2 * y * dydr;
                double dcdist_dr = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * dr2dr;
                double dicdist2_dr = -icdist2 * icdist2 * (k[5] + 2 * k[6] * r2 + 3 * k[7] * r4) * dr2dr;
                double da1dr = 2 * (x * dydr + y * dxdr);
                double dmxdr = (dxdr * cdist * icdist2 + x * dcdist_dr * icdist2 +
                                x * cdist * dicdist2_dr + k[2] * da1dr +
                                k[3] * (dr2dr + 4 * x * dxdr) + (k[8] + 2 * r2 * k[9]) * dr2dr);
                double dmydr = (dydr * cdist * icdist2 + y * dcdist_dr * icdist2 +
                                y * cdist * dicdist2_dr + k[2] * (dr2dr + 4 * y * dydr) +
                                k[3] * da1dr + (k[10] + 2 * r2 * k[11]) * dr2dr);
                ex[j] = fx * dmxdr;
                ey[j] = fy * dmydr;
            }
        }
    }
}
