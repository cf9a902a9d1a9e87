// Exercises the C++ drop-in header (include/ccalib.hpp) the way the reference's programs call
// OpenCV: vector<vector<Point3f>> boards, vector<vector<Point2f>> detections, then
// calibrateCamera -> computeReprojectionErrors -> stereoCalibrate
// (monocular_circle.cpp:93-94 and :5-29, binocular_chessboard.cpp:147-162).
// Prints one "key value..." line per result; exits 3 when the library reports that no GPU is usable.
#include <cmath>
#include <cstdio>
#include <vector>

#include "ccalib.hpp"

using namespace ccalib;

namespace {
struct Pose { double r[3], t[3]; };

void rot(const double r[3], double R[9]) {
    double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double c = std::cos(th), s = std::sin(th), k[3] = {0, 0, 0};
    if (th > 1e-12) for (int i = 0; i < 3; i++) k[i] = r[i] / th;
    double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = c * (i == j) + (1 - c) * k[i] * k[j] + s * K[3 * i + j];
}

Point2f project(const Point3f& X, const double R[9], const double t[3], const double K[4], const double d[5]) {
    double x = R[0] * X.x + R[1] * X.y + R[2] * X.z + t[0], y = R[3] * X.x + R[4] * X.y + R[5] * X.z + t[1],
           z = R[6] * X.x + R[7] * X.y + R[8] * X.z + t[2];
    x /= z, y /= z;
    double r2 = x * x + y * y, cd = 1 + r2 * (d[0] + r2 * (d[1] + r2 * d[4]));
    double xd = x * cd + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x), yd = y * cd + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y;
    return Point2f{(float)(K[0] * xd + K[2]), (float)(K[1] * yd + K[3])};
}
}  // namespace

int main() {
    const int nviews = 14, bw = 9, bh = 6;
    const double Kt[4] = {812.5, 806.25, 322.0, 238.5}, dt[5] = {-0.21, 0.09, 0.0007, -0.0004, -0.015};
    const double K2t[4] = {798.0, 801.5, 316.5, 244.0}, d2t[5] = {-0.18, 0.06, -0.0005, 0.0006, 0.01};
    const double rlr[3] = {0.01, -0.03, 0.005}, tlr[3] = {-120.0, 1.5, 2.5};
    std::vector<Point3f> board;
    for (int i = 0; i < bh; i++)
        for (int j = 0; j < bw; j++) board.push_back(Point3f{j * 25.f, i * 25.f, 0.f});
    std::vector<std::vector<Point3f>> obj(nviews, board);
    std::vector<std::vector<Point2f>> img1(nviews), img2(nviews);
    double Rlr[9];
    rot(rlr, Rlr);
    for (int v = 0; v < nviews; v++) {
        Pose p = {{0.35 * std::sin(1.3 * v + 0.2), 0.4 * std::cos(0.9 * v), 0.15 * std::sin(0.5 * v)},
                  {-100. + 12. * std::sin(2.1 * v), -60. + 9. * std::cos(1.7 * v), 520. + 60. * std::sin(0.8 * v + 1.)}};
        double R[9], R2[9], t2[3];
        rot(p.r, R);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) R2[3 * i + j] = Rlr[3 * i] * R[j] + Rlr[3 * i + 1] * R[3 + j] + Rlr[3 * i + 2] * R[6 + j];
            t2[i] = Rlr[3 * i] * p.t[0] + Rlr[3 * i + 1] * p.t[1] + Rlr[3 * i + 2] * p.t[2] + tlr[i];
        }
        for (const Point3f& X : board) {
            img1[v].push_back(project(X, R, p.t, Kt, dt));
            img2[v].push_back(project(X, R2, t2, K2t, d2t));
        }
    }
    try {
        Mat K, D;
        std::vector<Mat> rvecs, tvecs;
        double rms = calibrateCamera(obj, img1, Size{640, 480}, K, D, rvecs, tvecs, 0,
                                     TermCriteria(TermCriteria::COUNT + TermCriteria::EPS, 60, 1e-12));
        std::printf("mono_rms %.17g\n", rms);
        std::printf("mono_K %.17g %.17g %.17g %.17g\n", K(0, 0), K(1, 1), K(0, 2), K(1, 2));
        std::printf("mono_D %.17g %.17g %.17g %.17g %.17g\n", D.v[0], D.v[1], D.v[2], D.v[3], D.v[4]);
        std::vector<float> per_view;
        double rms2 = computeReprojectionErrors(obj, img1, rvecs, tvecs, K, D, per_view);
        std::printf("reproj_rms %.17g\n", rms2);

        Mat K1 = initCameraMatrix2D(obj, img1, Size{640, 480}, 0), K2 = initCameraMatrix2D(obj, img2, Size{640, 480}, 0);
        Mat D1, D2, R, T, E, F;
        double srms = stereoCalibrate(obj, img1, img2, K1, D1, K2, D2, Size{640, 480}, R, T, E, F,
                                      CALIB_USE_INTRINSIC_GUESS, TermCriteria(TermCriteria::COUNT + TermCriteria::EPS, 100, 1e-9));
        std::printf("stereo_rms %.17g\n", srms);
        std::printf("stereo_T %.17g %.17g %.17g\n", T.v[0], T.v[1], T.v[2]);
        std::printf("stereo_K2 %.17g %.17g %.17g %.17g\n", K2(0, 0), K2(1, 1), K2(0, 2), K2(1, 2));

        // error behaviour: mismatched sizes must throw, as cv::calibrateCamera does
        bool threw = false;
        try {
            auto bad = img1;
            bad[3].pop_back();
            calibrateCamera(obj, bad, Size{640, 480}, K, D, rvecs, tvecs);
        } catch (const Exception&) { threw = true; }
        std::printf("bad_input_throws %d\n", (int)threw);
    } catch (const Exception& e) {
        std::printf("exception %d %s\n", e.code, e.what());
        return e.code == CCALIB_ERR_NO_DEVICE ? 3 : 1;
    }
    return 0;
}
