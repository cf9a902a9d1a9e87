// Instantiates every function of include/ccalib_cv.hpp (against the stand-in headers under tests/stub_opencv) so the
// compiler sees all of it, links against libccalib.so so every C-ABI name it uses resolves, and checks the pieces that
// need no GPU: the distortion-vector lengths and the argument checks that fail before any device call.
#include <cstdio>
#include <cstring>

#include "ccalib_cv.hpp"
#ifndef CCALIB_CV_AVAILABLE
#error "ccalib_cv.hpp did not find <opencv2/core.hpp>"
#endif
#include "ccalib.hpp"

int main(int argc, char**) {
    using namespace ccalib_cv;
    int bad = 0;
    // lengths OpenCV 4.1.0 returns (recalled): 5 by default; 14 with the rational model; calibrateCamera 12 with thin prism alone
    bad += ccalib_dist_coeffs_out(0, 0) != 5 || ccalib_dist_coeffs_out(0, 1) != 5;
    bad += ccalib_dist_coeffs_out(cv::CALIB_RATIONAL_MODEL, 0) != 14 || ccalib_dist_coeffs_out(cv::CALIB_RATIONAL_MODEL, 1) != 14;
    bad += ccalib_dist_coeffs_out(cv::CALIB_THIN_PRISM_MODEL, 0) != 12 || ccalib_dist_coeffs_out(cv::CALIB_THIN_PRISM_MODEL, 1) != 14;
    bad += ccalib_dist_coeffs_out(cv::CALIB_TILTED_MODEL | cv::CALIB_THIN_PRISM_MODEL, 0) != 14;
    // the flag values the adaptor forwards unchanged must be the library's
    bad += (int)cv::CALIB_RATIONAL_MODEL != (int)CCALIB_RATIONAL_MODEL || (int)cv::CALIB_FIX_INTRINSIC != (int)CCALIB_FIX_INTRINSIC ||
           (int)cv::CALIB_SAME_FOCAL_LENGTH != (int)CCALIB_SAME_FOCAL_LENGTH || (int)cv::CALIB_FIX_K3 != (int)ccalib::CALIB_FIX_K3 ||
           (int)cv::CALIB_USE_LU != (int)CCALIB_USE_LU;
    std::vector<std::vector<cv::Point3f>> obj(1, std::vector<cv::Point3f>(4));
    std::vector<std::vector<cv::Point2f>> img(1, std::vector<cv::Point2f>(4)), img_short(1, std::vector<cv::Point2f>(3));
    std::vector<int> npts;
    std::vector<float> o, m;
    flatten(obj, img, npts, o, m);
    bad += npts.size() != 1 || npts[0] != 4 || o.size() != 12 || m.size() != 8;
    try {  // mismatched point counts are refused on the host
        flatten(obj, img_short, npts, o, m);
        bad++;
    } catch (const cv::Exception&) {
    }
    cv::Mat K, D, nobj;
    std::vector<cv::Mat> rv, tv;
    try {  // object-releasing calibration is refused on the host
        calibrateCameraRO(obj, img, cv::Size(640, 480), 0, K, D, rv, tv, nobj);
        bad++;
    } catch (const cv::Exception& e) {
        bad += e.code != cv::Error::StsNotImplemented;
    }
    if (argc > 100) {  // never run here (needs a GPU): present so that every adaptor function is instantiated and linked
        cv::Mat K2, D2, R, T, E, F;
        std::vector<float> pv;
        calibrateCamera(obj, img, cv::Size(640, 480), K, D, rv, tv);
        stereoCalibrate(obj, img, img, K, D, K2, D2, cv::Size(640, 480), R, T, E, F);
        initCameraMatrix2D(obj, img, cv::Size(640, 480));
        computeReprojectionErrors(obj, img, rv, tv, K, D, pv);
    }
    std::printf("cv adaptor check: %d problem(s)\n", bad);
    return bad != 0;
}
