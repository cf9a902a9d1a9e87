// ccalib.hpp -- header-only C++ drop-in over the C ABI (ccalib.h), with the call surface the
// reference uses:
//   calibrateCamera     /root/reference/monocular_circle.cpp:93-94
//   calibrateCameraRO   /root/reference/monocular_chessboard.cpp:315-317   (iFixedPoint must be -1)
//   stereoCalibrate     /root/reference/binocular_chessboard.cpp:152-162
//   initCameraMatrix2D  /root/reference/binocular_chessboard.cpp:147-148
//   projectPoints-based RMS check  /root/reference/monocular_circle.cpp:5-29
// Same argument order, flag values and TermCriteria semantics as OpenCV 4.1.0; errors are thrown
// as ccalib::Exception where OpenCV throws cv::Exception.  OpenCV itself is not needed: the POD
// types below stand in for cv::Point2f / Point3f / Size / Mat.  With OpenCV present, a maintainer
// maps them one to one.
#pragma once
#include <cfloat>
#include <stdexcept>
#include <string>
#include <vector>

#include "ccalib.h"

namespace ccalib {

struct Point2f { float x, y; };
struct Point3f { float x, y, z; };
struct Size { int width, height; };
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = 1, EPS = 2 };
    int type, maxCount;
    double epsilon;
    TermCriteria(int t = COUNT + EPS, int n = 30, double e = DBL_EPSILON) : type(t), maxCount(n), epsilon(e) {}
};
// Row-major double matrix; empty() means "not provided", as an empty cv::Mat does.
struct Mat {
    int rows = 0, cols = 0;
    std::vector<double> v;
    Mat() {}
    Mat(int r, int c, double fill = 0.) : rows(r), cols(c), v((size_t)r * c, fill) {}
    static Mat eye(int n) { Mat m(n, n); for (int i = 0; i < n; i++) m.v[(size_t)i * n + i] = 1.; return m; }
    bool empty() const { return v.empty(); }
    size_t total() const { return v.size(); }
    double& operator()(int r, int c) { return v[(size_t)r * cols + c]; }
    double operator()(int r, int c) const { return v[(size_t)r * cols + c]; }
};
enum {
    CALIB_USE_INTRINSIC_GUESS = CCALIB_USE_INTRINSIC_GUESS, CALIB_FIX_ASPECT_RATIO = CCALIB_FIX_ASPECT_RATIO,
    CALIB_FIX_PRINCIPAL_POINT = CCALIB_FIX_PRINCIPAL_POINT, CALIB_ZERO_TANGENT_DIST = CCALIB_ZERO_TANGENT_DIST,
    CALIB_FIX_FOCAL_LENGTH = CCALIB_FIX_FOCAL_LENGTH, CALIB_FIX_K1 = CCALIB_FIX_K1, CALIB_FIX_K2 = CCALIB_FIX_K2,
    CALIB_FIX_K3 = CCALIB_FIX_K3, CALIB_FIX_K4 = CCALIB_FIX_K4, CALIB_FIX_K5 = CCALIB_FIX_K5,
    CALIB_FIX_K6 = CCALIB_FIX_K6, CALIB_RATIONAL_MODEL = CCALIB_RATIONAL_MODEL,
    CALIB_FIX_TANGENT_DIST = CCALIB_FIX_TANGENT_DIST, CALIB_SAME_FOCAL_LENGTH = CCALIB_SAME_FOCAL_LENGTH,
    CALIB_FIX_INTRINSIC = CCALIB_FIX_INTRINSIC, CALIB_USE_LU = CCALIB_USE_LU
};

struct Exception : std::runtime_error {
    int code;
    Exception(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
namespace detail {
inline void check(int rc) { if (rc) throw Exception(rc, ccalib_last_error()); }
inline void flatten(const std::vector<std::vector<Point3f>>& obj, const std::vector<std::vector<Point2f>>& img,
                    std::vector<int>& npts, std::vector<float>& o, std::vector<float>& m) {
    if (obj.size() != img.size() || obj.empty()) throw Exception(CCALIB_ERR_ARG, "objectPoints / imagePoints view counts differ");
    for (size_t i = 0; i < obj.size(); i++) {
        if (obj[i].size() != img[i].size()) throw Exception(CCALIB_ERR_ARG, "objectPoints / imagePoints sizes differ");
        npts.push_back((int)obj[i].size());
        for (const Point3f& p : obj[i]) { o.push_back(p.x); o.push_back(p.y); o.push_back(p.z); }
        for (const Point2f& p : img[i]) { m.push_back(p.x); m.push_back(p.y); }
    }
}
// One pointer per view into the vectors the caller already holds (Point3f / Point2f are plain float triples / pairs):
// what the ccalib_*_views entry points take, so that nothing is gathered here.
struct ViewPointers {
    std::vector<int> npts;
    std::vector<const float*> obj, img[2];
};
inline ViewPointers view_pointers(const std::vector<std::vector<Point3f>>& obj, const std::vector<std::vector<Point2f>>& img1,
                                  const std::vector<std::vector<Point2f>>* img2 = nullptr) {
    static_assert(sizeof(Point3f) == 3 * sizeof(float) && sizeof(Point2f) == 2 * sizeof(float), "points must be packed floats");
    if (obj.size() != img1.size() || (img2 && img2->size() != obj.size()) || obj.empty())
        throw Exception(CCALIB_ERR_ARG, "objectPoints / imagePoints view counts differ");
    ViewPointers vp;
    for (size_t i = 0; i < obj.size(); i++) {
        if (obj[i].size() != img1[i].size() || (img2 && (*img2)[i].size() != obj[i].size()))
            throw Exception(CCALIB_ERR_ARG, "objectPoints / imagePoints sizes differ");
        if (obj[i].empty()) throw Exception(CCALIB_ERR_ARG, "a view without points");
        vp.npts.push_back((int)obj[i].size());
        vp.obj.push_back(&obj[i][0].x);
        vp.img[0].push_back(&img1[i][0].x);
        if (img2) vp.img[1].push_back(&(*img2)[i][0].x);
    }
    return vp;
}
// OpenCV's prepareCameraMatrix: empty -> eye(3).  (Output length of the distortion vector: ccalib_dist_coeffs_out.)
inline Mat camera_or_eye(const Mat& K) { return K.total() == 9 ? K : Mat::eye(3); }
}  // namespace detail

inline double calibrateCameraRO(const std::vector<std::vector<Point3f>>& objectPoints,
                                const std::vector<std::vector<Point2f>>& imagePoints, Size imageSize, int iFixedPoint,
                                Mat& cameraMatrix, Mat& distCoeffs, std::vector<Mat>& rvecs, std::vector<Mat>& tvecs,
                                std::vector<Point3f>& newObjPoints, int flags = 0, TermCriteria criteria = TermCriteria()) {
    if (iFixedPoint >= 0) throw Exception(CCALIB_ERR_UNSUPPORTED, "object-releasing calibration (iFixedPoint >= 0) is not built");
    detail::ViewPointers vp = detail::view_pointers(objectPoints, imagePoints);
    const std::vector<int>& npts = vp.npts;
    Mat K = detail::camera_or_eye(cameraMatrix);
    int nd = ccalib_dist_coeffs_out(flags, 0);
    std::vector<double> d(nd, 0.);
    for (size_t i = 0; i < distCoeffs.total() && i < d.size(); i++) d[i] = distCoeffs.v[i];
    int n = (int)npts.size();
    std::vector<double> rv((size_t)3 * n), tv((size_t)3 * n);
    double rms = 0;
    detail::check(ccalib_calibrate_camera_views(n, npts.data(), vp.obj.data(), vp.img[0].data(), imageSize.width, imageSize.height,
                                                K.v.data(), d.data(), nd, rv.data(), tv.data(), flags, criteria.type,
                                                criteria.maxCount, criteria.epsilon, nullptr, &rms, nullptr));
    cameraMatrix = K;
    // prepareDistCoeffs keeps the caller's orientation: a column for an n x 1 input, a row otherwise (empty included)
    distCoeffs = distCoeffs.cols == 1 ? Mat(nd, 1) : Mat(1, nd);
    distCoeffs.v = d;
    rvecs.assign(n, Mat(3, 1));
    tvecs.assign(n, Mat(3, 1));
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) rvecs[i].v[c] = rv[(size_t)3 * i + c], tvecs[i].v[c] = tv[(size_t)3 * i + c];
    if (!objectPoints.empty()) newObjPoints = objectPoints[0];
    return rms;
}

inline double calibrateCamera(const std::vector<std::vector<Point3f>>& objectPoints,
                              const std::vector<std::vector<Point2f>>& imagePoints, Size imageSize, Mat& cameraMatrix,
                              Mat& distCoeffs, std::vector<Mat>& rvecs, std::vector<Mat>& tvecs, int flags = 0,
                              TermCriteria criteria = TermCriteria()) {
    std::vector<Point3f> unused;
    return calibrateCameraRO(objectPoints, imagePoints, imageSize, -1, cameraMatrix, distCoeffs, rvecs, tvecs, unused, flags,
                             criteria);
}

inline Mat initCameraMatrix2D(const std::vector<std::vector<Point3f>>& objectPoints,
                              const std::vector<std::vector<Point2f>>& imagePoints, Size imageSize, double aspectRatio = 1.0) {
    std::vector<int> npts;
    std::vector<float> o, m;
    detail::flatten(objectPoints, imagePoints, npts, o, m);
    Mat K(3, 3);
    detail::check(ccalib_init_camera_matrix_2d((int)npts.size(), npts.data(), o.data(), m.data(), imageSize.width,
                                               imageSize.height, aspectRatio, K.v.data()));
    return K;
}

inline double stereoCalibrate(const std::vector<std::vector<Point3f>>& objectPoints,
                              const std::vector<std::vector<Point2f>>& imagePoints1,
                              const std::vector<std::vector<Point2f>>& imagePoints2, Mat& cameraMatrix1, Mat& distCoeffs1,
                              Mat& cameraMatrix2, Mat& distCoeffs2, Size imageSize, Mat& R, Mat& T, Mat& E, Mat& F,
                              int flags = CALIB_FIX_INTRINSIC,
                              TermCriteria criteria = TermCriteria(TermCriteria::COUNT + TermCriteria::EPS, 30, 1e-6)) {
    detail::ViewPointers vp = detail::view_pointers(objectPoints, imagePoints1, &imagePoints2);
    Mat K1 = detail::camera_or_eye(cameraMatrix1), K2 = detail::camera_or_eye(cameraMatrix2);
    std::vector<double> d1(14, 0.), d2(14, 0.);
    for (size_t i = 0; i < distCoeffs1.total() && i < 14; i++) d1[i] = distCoeffs1.v[i];
    for (size_t i = 0; i < distCoeffs2.total() && i < 14; i++) d2[i] = distCoeffs2.v[i];
    R = Mat(3, 3), T = Mat(3, 1), E = Mat(3, 3), F = Mat(3, 3);
    double rms = 0;
    detail::check(ccalib_stereo_calibrate_views((int)vp.npts.size(), vp.npts.data(), vp.obj.data(), vp.img[0].data(),
                                                vp.img[1].data(), K1.v.data(), d1.data(), K2.v.data(), d2.data(), imageSize.width,
                                                imageSize.height, R.v.data(), T.v.data(), E.v.data(), F.v.data(), flags,
                                                criteria.type, criteria.maxCount, criteria.epsilon, nullptr, nullptr, nullptr, &rms,
                                                nullptr));
    cameraMatrix1 = K1, cameraMatrix2 = K2;
    int nd = ccalib_dist_coeffs_out(flags, 1);
    distCoeffs1 = distCoeffs1.cols == 1 ? Mat(nd, 1) : Mat(1, nd);
    distCoeffs2 = distCoeffs2.cols == 1 ? Mat(nd, 1) : Mat(1, nd);
    distCoeffs1.v.assign(d1.begin(), d1.begin() + nd);
    distCoeffs2.v.assign(d2.begin(), d2.begin() + nd);
    return rms;
}

// The reference's computeReprojectionErrors (monocular_circle.cpp:5-29) in one call.
inline double computeReprojectionErrors(const std::vector<std::vector<Point3f>>& objectPoints,
                                        const std::vector<std::vector<Point2f>>& imagePoints, const std::vector<Mat>& rvecs,
                                        const std::vector<Mat>& tvecs, const Mat& cameraMatrix, const Mat& distCoeffs,
                                        std::vector<float>& perViewErrors) {
    std::vector<int> npts;
    std::vector<float> o, m;
    detail::flatten(objectPoints, imagePoints, npts, o, m);
    int n = (int)npts.size();
    std::vector<double> rv((size_t)3 * n), tv((size_t)3 * n), pv(n);
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) rv[(size_t)3 * i + c] = rvecs[i].v[c], tv[(size_t)3 * i + c] = tvecs[i].v[c];
    double total = 0;
    detail::check(ccalib_reprojection_error(n, npts.data(), o.data(), m.data(), cameraMatrix.v.data(), distCoeffs.v.data(),
                                            (int)distCoeffs.total(), rv.data(), tv.data(), nullptr, pv.data(), &total));
    perViewErrors.assign(pv.begin(), pv.end());
    return total;
}

// The result files of the reference (cv::FileStorage YAML): monocular_chessboard.cpp:321-328 and
// binocular_chessboard.cpp:203-210, :234-241.
inline void writeMonoYaml(const std::string& path, Size imageSize, const Mat& M, const Mat& D, const std::vector<Mat>& rvecs,
                          const std::vector<Mat>& tvecs) {
    int n = (int)rvecs.size();
    std::vector<double> rv((size_t)3 * n), tv((size_t)3 * n);
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) rv[(size_t)3 * i + c] = rvecs[i].v[c], tv[(size_t)3 * i + c] = tvecs[i].v[c];
    detail::check(ccalib_write_mono_yaml(path.c_str(), imageSize.width, imageSize.height, M.v.data(), D.v.data(),
                                         (int)D.total(), n, rv.data(), tv.data()));
}
inline void writeStereoYaml(const std::string& intrinsicsPath, const std::string& extrinsicsPath, Size imageSize,
                            const Mat& M1, const Mat& D1, const Mat& M2, const Mat& D2, const Mat& R, const Mat& T,
                            const Mat& E, const Mat& F) {
    detail::check(ccalib_write_stereo_yaml(intrinsicsPath.c_str(), extrinsicsPath.c_str(), imageSize.width, imageSize.height,
                                           M1.v.data(), D1.v.data(), (int)D1.total(), M2.v.data(), D2.v.data(),
                                           (int)D2.total(), R.v.data(), T.v.data(), E.empty() ? nullptr : E.v.data(),
                                           F.empty() ? nullptr : F.v.data()));
}

}  // namespace ccalib
