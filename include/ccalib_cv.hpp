// ccalib_cv.hpp -- the reference's five OpenCV call shapes over libccalib.so, with real OpenCV types.
//
// For a maintainer of rxjf/Camera-Calibration: add this header next to the sources, link with -lccalib and
// replace the cv:: call at each site by the ccalib_cv:: one of the same name:
//   monocular_circle.cpp:93-94        calibrateCamera
//   monocular_chessboard.cpp:315-317  calibrateCameraRO (iFixedPoint = -1)
//   binocular_chessboard.cpp:152-162, binocular_circle.cpp:153-163   stereoCalibrate
//   binocular_chessboard.cpp:147-148  initCameraMatrix2D
//   monocular_circle.cpp:5-29         computeReprojectionErrors (the projectPoints + norm loop)
// The header compiles to nothing where OpenCV's headers are not installed; include/ccalib.hpp is the same
// surface over plain POD types and needs no OpenCV.
#pragma once
#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define CCALIB_CV_AVAILABLE 1
#endif
#endif

#ifdef CCALIB_CV_AVAILABLE
#include <cfloat>
#include <opencv2/calib3d.hpp>  // the CALIB_* flags
#include <opencv2/core.hpp>
#include <vector>

#include "ccalib.h"

namespace ccalib_cv {

inline void check(int rc) {
    if (rc) CV_Error(rc == CCALIB_ERR_ARG ? cv::Error::StsBadArg : cv::Error::StsError, ccalib_last_error());
}

// vector<vector<Point3f>> / vector<vector<Point2f>> -> the flat arrays of the C ABI
inline void flatten(const std::vector<std::vector<cv::Point3f>>& obj, const std::vector<std::vector<cv::Point2f>>& img,
                    std::vector<int>& npts, std::vector<float>& o, std::vector<float>& m) {
    CV_Assert(obj.size() == img.size() && !obj.empty());
    for (size_t i = 0; i < obj.size(); i++) {
        CV_Assert(obj[i].size() == img[i].size() && !obj[i].empty());
        npts.push_back((int)obj[i].size());
        o.insert(o.end(), &obj[i][0].x, &obj[i][0].x + 3 * obj[i].size());  // Point3f is three packed floats
        m.insert(m.end(), &img[i][0].x, &img[i][0].x + 2 * img[i].size());
    }
}
// ... or one pointer per view into the vectors themselves, for the ccalib_*_views entry points: nothing is copied here
struct ViewPointers {
    std::vector<int> npts;
    std::vector<const float*> obj, img1, img2;
};
inline ViewPointers view_pointers(const std::vector<std::vector<cv::Point3f>>& obj, const std::vector<std::vector<cv::Point2f>>& img1,
                                  const std::vector<std::vector<cv::Point2f>>* img2 = nullptr) {
    static_assert(sizeof(cv::Point3f) == 3 * sizeof(float) && sizeof(cv::Point2f) == 2 * sizeof(float), "packed points");
    CV_Assert(obj.size() == img1.size() && !obj.empty() && (!img2 || img2->size() == obj.size()));
    ViewPointers vp;
    for (size_t i = 0; i < obj.size(); i++) {
        CV_Assert(obj[i].size() == img1[i].size() && !obj[i].empty() && (!img2 || (*img2)[i].size() == obj[i].size()));
        vp.npts.push_back((int)obj[i].size());
        vp.obj.push_back(&obj[i][0].x);
        vp.img1.push_back(&img1[i][0].x);
        if (img2) vp.img2.push_back(&(*img2)[i][0].x);
    }
    return vp;
}
inline cv::Mat as_f64(const cv::Mat& a, int rows, int cols, bool identity) {
    cv::Mat r = identity ? cv::Mat::eye(rows, cols, CV_64F) : cv::Mat::zeros(rows, cols, CV_64F);
    if (!a.empty()) {
        cv::Mat t;
        a.reshape(1, (int)a.total() == rows * cols ? rows : (int)a.total()).convertTo(t, CV_64F);
        int n = std::min((int)t.total(), rows * cols);
        for (int i = 0; i < n; i++) r.ptr<double>()[i] = t.ptr<double>()[i];
    }
    return r;
}
inline void split_poses(const cv::Mat& rv, const cv::Mat& tv, std::vector<cv::Mat>& rvecs, std::vector<cv::Mat>& tvecs) {
    rvecs.resize(rv.rows), tvecs.resize(tv.rows);
    for (int i = 0; i < rv.rows; i++) rvecs[i] = rv.row(i).t(), tvecs[i] = tv.row(i).t();
}

inline double calibrateCamera(const std::vector<std::vector<cv::Point3f>>& objectPoints,
                              const std::vector<std::vector<cv::Point2f>>& imagePoints, cv::Size imageSize,
                              cv::Mat& cameraMatrix, cv::Mat& distCoeffs, std::vector<cv::Mat>& rvecs,
                              std::vector<cv::Mat>& tvecs, int flags = 0,
                              cv::TermCriteria crit = cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30,
                                                                       DBL_EPSILON)) {
    const ViewPointers vp = view_pointers(objectPoints, imagePoints);
    const int n = (int)vp.npts.size(), nd = ccalib_dist_coeffs_out(flags, 0);
    cv::Mat K = as_f64(cameraMatrix, 3, 3, true), D = as_f64(distCoeffs, nd, 1, false), rv(n, 3, CV_64F), tv(n, 3, CV_64F);
    double rms = 0;
    check(ccalib_calibrate_camera_views(n, vp.npts.data(), vp.obj.data(), vp.img1.data(), imageSize.width, imageSize.height,
                                        K.ptr<double>(), D.ptr<double>(), nd, rv.ptr<double>(), tv.ptr<double>(), flags, crit.type,
                                        crit.maxCount, crit.epsilon, nullptr, &rms, nullptr));
    // prepareDistCoeffs keeps the caller's orientation: a column for an n x 1 input, a row otherwise (empty included)
    const bool column = distCoeffs.cols == 1;
    cameraMatrix = K, distCoeffs = column ? D : D.reshape(1, 1);
    split_poses(rv, tv, rvecs, tvecs);
    return rms;
}

// the reference passes iFixedPoint = -1 (monocular_chessboard.cpp:230, :310-312), for which
// calibrateCameraRO is calibrateCamera; any other value is refused
inline double calibrateCameraRO(const std::vector<std::vector<cv::Point3f>>& objectPoints,
                                const std::vector<std::vector<cv::Point2f>>& imagePoints, cv::Size imageSize,
                                int iFixedPoint, cv::Mat& cameraMatrix, cv::Mat& distCoeffs, std::vector<cv::Mat>& rvecs,
                                std::vector<cv::Mat>& tvecs, cv::Mat& newObjPoints, int flags = 0,
                                cv::TermCriteria crit = cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30,
                                                                         DBL_EPSILON)) {
    if (iFixedPoint >= 0) CV_Error(cv::Error::StsNotImplemented, "object-releasing calibration (iFixedPoint >= 0)");
    newObjPoints.release();
    return calibrateCamera(objectPoints, imagePoints, imageSize, cameraMatrix, distCoeffs, rvecs, tvecs, flags, crit);
}

inline double stereoCalibrate(const std::vector<std::vector<cv::Point3f>>& objectPoints,
                              const std::vector<std::vector<cv::Point2f>>& imagePoints1,
                              const std::vector<std::vector<cv::Point2f>>& imagePoints2, cv::Mat& cameraMatrix1,
                              cv::Mat& distCoeffs1, cv::Mat& cameraMatrix2, cv::Mat& distCoeffs2, cv::Size imageSize,
                              cv::Mat& R, cv::Mat& T, cv::Mat& E, cv::Mat& F, int flags = cv::CALIB_FIX_INTRINSIC,
                              cv::TermCriteria crit = cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30,
                                                                       1e-6)) {
    const ViewPointers vp = view_pointers(objectPoints, imagePoints1, &imagePoints2);
    cv::Mat K1 = as_f64(cameraMatrix1, 3, 3, true), K2 = as_f64(cameraMatrix2, 3, 3, true);
    cv::Mat D1 = as_f64(distCoeffs1, 14, 1, false), D2 = as_f64(distCoeffs2, 14, 1, false);
    R.create(3, 3, CV_64F), T.create(3, 1, CV_64F), E.create(3, 3, CV_64F), F.create(3, 3, CV_64F);
    double rms = 0;
    check(ccalib_stereo_calibrate_views((int)vp.npts.size(), vp.npts.data(), vp.obj.data(), vp.img1.data(), vp.img2.data(),
                                        K1.ptr<double>(), D1.ptr<double>(), K2.ptr<double>(), D2.ptr<double>(), imageSize.width,
                                        imageSize.height, R.ptr<double>(), T.ptr<double>(), E.ptr<double>(), F.ptr<double>(), flags,
                                        crit.type, crit.maxCount, crit.epsilon, nullptr, nullptr, nullptr, &rms, nullptr));
    const int nd = ccalib_dist_coeffs_out(flags, 1);
    const bool column1 = distCoeffs1.cols == 1, column2 = distCoeffs2.cols == 1;  // orientation kept, as above
    cameraMatrix1 = K1, cameraMatrix2 = K2;
    distCoeffs1 = D1.rowRange(0, nd).clone(), distCoeffs2 = D2.rowRange(0, nd).clone();
    if (!column1) distCoeffs1 = distCoeffs1.reshape(1, 1);
    if (!column2) distCoeffs2 = distCoeffs2.reshape(1, 1);
    return rms;
}

inline cv::Mat initCameraMatrix2D(const std::vector<std::vector<cv::Point3f>>& objectPoints,
                                  const std::vector<std::vector<cv::Point2f>>& imagePoints, cv::Size imageSize,
                                  double aspectRatio = 1.0) {
    std::vector<int> npts;
    std::vector<float> o, m;
    flatten(objectPoints, imagePoints, npts, o, m);
    cv::Mat K(3, 3, CV_64F);
    check(ccalib_init_camera_matrix_2d((int)npts.size(), npts.data(), o.data(), m.data(), imageSize.width, imageSize.height,
                                       aspectRatio, K.ptr<double>()));
    return K;
}

// monocular_circle.cpp:5-29: per-view RMS into perViewErrors, total RMS returned
inline double computeReprojectionErrors(const std::vector<std::vector<cv::Point3f>>& objectPoints,
                                        const std::vector<std::vector<cv::Point2f>>& imagePoints,
                                        const std::vector<cv::Mat>& rvecs, const std::vector<cv::Mat>& tvecs,
                                        const cv::Mat& cameraMatrix, const cv::Mat& distCoeffs,
                                        std::vector<float>& perViewErrors) {
    std::vector<int> npts;
    std::vector<float> o, m;
    flatten(objectPoints, imagePoints, npts, o, m);
    const int n = (int)npts.size();
    cv::Mat K = as_f64(cameraMatrix, 3, 3, true), D = as_f64(distCoeffs, (int)distCoeffs.total(), 1, false);
    cv::Mat rv(n, 3, CV_64F), tv(n, 3, CV_64F);
    for (int i = 0; i < n; i++) {
        cv::Mat r64, t64;
        rvecs[i].reshape(1, 1).convertTo(r64, CV_64F), tvecs[i].reshape(1, 1).convertTo(t64, CV_64F);
        r64.copyTo(rv.row(i)), t64.copyTo(tv.row(i));
    }
    std::vector<double> pv(n);
    double total = 0;
    check(ccalib_reprojection_error(n, npts.data(), o.data(), m.data(), K.ptr<double>(), D.ptr<double>(), (int)D.total(),
                                    rv.ptr<double>(), tv.ptr<double>(), nullptr, pv.data(), &total));
    perViewErrors.assign(pv.begin(), pv.end());
    return total;
}

}  // namespace ccalib_cv
#endif  // CCALIB_CV_AVAILABLE
