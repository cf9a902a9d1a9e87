// host_rodrigues.cpp -- cvRodrigues2 on the host, both directions (OpenCV 4.1.0 calibration.cpp), for the
// small conversions around the device passes: R from om_LR for the R, E, F outputs of stereoCalibrate
// (reference call site binocular_chessboard.cpp:152-162).
#include "host_rodrigues.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>

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

}  // namespace ccalib
