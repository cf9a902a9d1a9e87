// api.cpp -- whole-call entry points: the argument handling cv::calibrateCamera and
// cv::stereoCalibrate perform around their LM loops (OpenCV 4.1.0 calibration.cpp:
// cvCalibrateCamera2Internal, cvStereoCalibrateImpl and the C++ wrappers), as the reference
// calls them at monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317,
// binocular_chessboard.cpp:152-162 and binocular_circle.cpp:153-163.  The loops themselves run
// in the device solver (solver.hip) under the host LM driver (lm_driver.cpp).
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#include "ccalib.h"
#include "host_rodrigues.hpp"
#include "host_util.hpp"
#include "solver_internal.hpp"

using namespace ccalib;

namespace {

// CCALIB_TIMING=1: phase times of the whole-call entry points on stderr
struct PhaseTimer {
    bool on = std::getenv("CCALIB_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ccalib] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

struct SolverGuard {
    ccalib_solver* s = nullptr;
    ~SolverGuard() { ccalib_solver_destroy(s); }
};

bool views_share_board(int nviews, const int* npts, const float* obj) {
    for (int i = 1; i < nviews; i++)
        if (npts[i] != npts[0]) return false;
    const size_t row = (size_t)3 * npts[0];
    int differs = 0;
#pragma omp parallel for schedule(static) reduction(| : differs)
    for (int i = 1; i < nviews; i++)
        if (!differs && std::memcmp(obj, obj + row * i, sizeof(float) * row) != 0) differs = 1;
    return !differs;
}

const float* view_ptr(const HostViews& h, const std::vector<size_t>& off, int i, int fpp) {
    return h.ptrs ? h.ptrs[i] : h.flat + (size_t)fpp * off[i];
}

// the same test over either shape of the caller's object points; views that point at one array share it trivially
bool views_share_board(int nviews, const int* npts, const HostViews& obj) {
    if (obj.flat) return views_share_board(nviews, npts, obj.flat);
    for (int i = 1; i < nviews; i++)
        if (npts[i] != npts[0]) return false;
    const size_t bytes = sizeof(float) * 3 * npts[0];
    int differs = 0;
#pragma omp parallel for schedule(static) reduction(| : differs)
    for (int i = 1; i < nviews; i++)
        if (!differs && obj.ptrs[i] != obj.ptrs[0] && std::memcmp(obj.ptrs[0], obj.ptrs[i], bytes) != 0) differs = 1;
    return !differs;
}

// flag word -> 18-entry mask, exactly as cvCalibrateCamera2Internal fills CvLevMarq::mask
int intrinsic_mask(int flags, unsigned char mask[18]) {
    std::memset(mask, 1, 18);
    if (flags & CCALIB_FIX_ASPECT_RATIO) mask[0] = 0;
    if (flags & CCALIB_FIX_FOCAL_LENGTH) mask[0] = mask[1] = 0;
    if (flags & CCALIB_FIX_PRINCIPAL_POINT) mask[2] = mask[3] = 0;
    if (flags & CCALIB_ZERO_TANGENT_DIST) mask[6] = mask[7] = 0;
    if (!(flags & CCALIB_RATIONAL_MODEL)) flags |= CCALIB_FIX_K4 | CCALIB_FIX_K5 | CCALIB_FIX_K6;
    if (!(flags & CCALIB_THIN_PRISM_MODEL)) flags |= CCALIB_FIX_S1_S2_S3_S4;
    if (!(flags & CCALIB_TILTED_MODEL)) flags |= CCALIB_FIX_TAUX_TAUY;
    mask[4] = !(flags & CCALIB_FIX_K1);
    mask[5] = !(flags & CCALIB_FIX_K2);
    if (flags & CCALIB_FIX_TANGENT_DIST) mask[6] = mask[7] = 0;
    mask[8] = !(flags & CCALIB_FIX_K3);
    mask[9] = !(flags & CCALIB_FIX_K4);
    mask[10] = !(flags & CCALIB_FIX_K5);
    mask[11] = !(flags & CCALIB_FIX_K6);
    if (flags & CCALIB_FIX_S1_S2_S3_S4) mask[12] = mask[13] = mask[14] = mask[15] = 0;
    if (flags & CCALIB_FIX_TAUX_TAUY) mask[16] = mask[17] = 0;
    return flags;
}

int check_model_flags(int flags) {
    if (flags & CCALIB_TILTED_MODEL) return set_error(CCALIB_ERR_UNSUPPORTED, "CALIB_TILTED_MODEL is not built into this library");
    if (flags & CCALIB_THIN_PRISM_MODEL)
        return set_error(CCALIB_ERR_UNSUPPORTED, "CALIB_THIN_PRISM_MODEL is not built into this library");
    return CCALIB_OK;
}

}  // namespace

extern "C" {

int ccalib_init_camera_matrix_2d(int nviews, const int* npts, const float* obj_xyz, const float* img_uv, int width,
                                 int height, double aspect, double* K) {
    if (nviews < 1 || !npts || !obj_xyz || !img_uv || !K) return set_error(CCALIB_ERR_ARG, "initCameraMatrix2D: bad arguments");
    for (int i = 0; i < nviews; i++)
        if (npts[i] < 4) return set_error(CCALIB_ERR_ARG, "view %d has fewer than 4 points", i);
    bool shared = views_share_board(nviews, npts, obj_xyz);
    SolverGuard g;
    int rc = ccalib_solver_create(&g.s, 0, 1, nviews, npts, obj_xyz, shared ? 1 : 0, img_uv, nullptr);
    if (rc) return rc;
    return ccalib_solver_init_intrinsics(g.s, 0, width, height, aspect, K);
}

}  // extern "C"

namespace {

int calibrate_camera_impl(int nviews, const int* npts, HostViews objv, HostViews imgv, int width, int height, double* K,
                          double* dist, int ndist, double* rvecs, double* tvecs, int flags, int term_type, int max_iter, double eps,
                          const double* init_rt, double* rms, ccalib_stats* stats) {
    if (nviews < 1 || !npts || (!objv.flat && !objv.ptrs) || (!imgv.flat && !imgv.ptrs) || !K || !dist || !rvecs || !tvecs || !rms)
        return set_error(CCALIB_ERR_ARG, "calibrateCamera: null argument");
    if (!(ndist == 4 || ndist == 5 || ndist == 8 || ndist == 12 || ndist == 14))
        return set_error(CCALIB_ERR_ARG, "the distortion vector must have 4, 5, 8, 12 or 14 entries (got %d)", ndist);
    int rc = check_model_flags(flags);
    if (rc) return rc;
    PhaseTimer pt;
    std::vector<size_t> off((size_t)nviews + 1, 0);
    for (int i = 0; i < nviews; i++) {
        if (npts[i] < 4) return set_error(CCALIB_ERR_ARG, "view %d has %d points (at least 4 needed)", i, npts[i]);
        if (objv.ptrs && (!objv.ptrs[i] || (imgv.ptrs && !imgv.ptrs[i]))) return set_error(CCALIB_ERR_ARG, "view %d: null point array", i);
        off[i + 1] = off[i] + npts[i];
    }
    const size_t total = off[nviews];
    if (ndist < 8) {
        if (ndist < 5) flags |= CCALIB_FIX_K3;
        flags |= CCALIB_FIX_K4 | CCALIB_FIX_K5 | CCALIB_FIX_K6;
    }
    // one board seen in every view (the reference's case: objectPoints.resize(n, objectPoints[0])) is found first:
    // everything below then looks at its few dozen points instead of at every view's copy
    const bool shared = views_share_board(nviews, npts, objv);
    pt.lap("shared-board test");
    const int nobjviews = shared ? 1 : nviews;
    std::vector<float> flattened;             // only when the rig's Z has to be cleared
    std::vector<const float*> flat_ptrs;
    double A[9], k[14] = {0}, aspect = 0;
    std::memcpy(A, K, sizeof(A));
    if (flags & CCALIB_USE_INTRINSIC_GUESS) {
        if (A[0] <= 0 || A[4] <= 0) return set_error(CCALIB_ERR_ARG, "focal length (fx and fy) must be positive");
        if (A[2] < 0 || A[2] >= width || A[5] < 0 || A[5] >= height)
            return set_error(CCALIB_ERR_ARG, "principal point must be within the image");
        if (std::fabs(A[1]) > 1e-5) return set_error(CCALIB_ERR_ARG, "non-zero skew is not supported");
        if (std::fabs(A[3]) > 1e-5 || std::fabs(A[6]) > 1e-5 || std::fabs(A[7]) > 1e-5 || std::fabs(A[8] - 1) > 1e-5)
            return set_error(CCALIB_ERR_ARG, "the intrinsic matrix must have [fx 0 cx; 0 fy cy; 0 0 1] shape");
        A[1] = A[3] = A[6] = A[7] = 0, A[8] = 1;
        if (flags & CCALIB_FIX_ASPECT_RATIO) aspect = A[0] / A[4];
        for (int i = 0; i < ndist; i++) k[i] = dist[i];
    } else {
        // mean and spread of Z over all object points (a shared board counts once per view: same numbers)
        double mean = 0, sq = 0;
#pragma omp parallel for schedule(static) reduction(+ : mean) if (nobjviews > 64)
        for (int i = 0; i < nobjviews; i++) {
            const float* o = view_ptr(objv, off, i, 3);
            for (int p = 0; p < npts[i]; p++) mean += o[3 * p + 2];
        }
        const double nobjpts = shared ? (double)npts[0] : (double)total;
        mean /= nobjpts;
#pragma omp parallel for schedule(static) reduction(+ : sq) if (nobjviews > 64)
        for (int i = 0; i < nobjviews; i++) {
            const float* o = view_ptr(objv, off, i, 3);
            for (int p = 0; p < npts[i]; p++) sq += (o[3 * p + 2] - mean) * (o[3 * p + 2] - mean);
        }
        if (std::fabs(mean) > 1e-5 || std::sqrt(sq / nobjpts) > 1e-5)
            return set_error(CCALIB_ERR_ARG, "for non-planar calibration rigs the initial intrinsic matrix must be specified");
        if (sq != 0 || mean != 0) {  // OpenCV clears the Z it has just found negligible
            const size_t nflat = shared ? (size_t)npts[0] : total;
            flattened.resize(3 * nflat);
            flat_ptrs.resize(nobjviews);
            for (int i = 0; i < nobjviews; i++) {
                float* d = flattened.data() + 3 * off[i];
                std::memcpy(d, view_ptr(objv, off, i, 3), sizeof(float) * 3 * npts[i]);
                for (int p = 0; p < npts[i]; p++) d[3 * p + 2] = 0.f;
                flat_ptrs[i] = d;
            }
            if (shared) flat_ptrs.assign(nviews, flattened.data());
            objv = HostViews{nullptr, flat_ptrs.data()};
        }
        if (flags & CCALIB_FIX_ASPECT_RATIO) aspect = A[0] / A[4];
    }
    if ((flags & CCALIB_FIX_ASPECT_RATIO) && !(aspect >= 0.01 && aspect <= 100.))
        return set_error(CCALIB_ERR_ARG, "the specified aspect ratio (= cameraMatrix[0][0] / cameraMatrix[1][1]) is incorrect");
    // the points go to the device once; initialisation and the LM loop both run there
    pt.lap("argument checks");
    SolverGuard g;
    if ((rc = solver_create_views(&g.s, 0, 1, nviews, npts, objv, shared ? 1 : 0, imgv, HostViews{nullptr, nullptr}))) return rc;
    pt.lap("solver create + upload");
    if (!(flags & CCALIB_USE_INTRINSIC_GUESS) && (rc = ccalib_solver_init_intrinsics(g.s, 0, width, height, aspect, A))) return rc;
    pt.lap("intrinsic seed");

    double intr[18];
    unsigned char mask[18];
    intr[0] = A[0], intr[1] = A[4], intr[2] = A[2], intr[3] = A[5];
    std::memcpy(intr + 4, k, sizeof(k));
    if (flags & CCALIB_ZERO_TANGENT_DIST) intr[6] = intr[7] = 0;
    flags = intrinsic_mask(flags, mask);

    // starting poses: the caller's, or cvFindExtrinsicCameraParams2 per view (left on the device)
    if (!init_rt && (rc = ccalib_solver_init_poses(g.s, 0, A, k, nullptr))) return rc;
    pt.lap("pose initialisation");
    double asp[1] = {(flags & CCALIB_FIX_ASPECT_RATIO) ? aspect : 0.};
    if ((rc = ccalib_solver_set_start(g.s, intr, mask, asp, 0, init_rt, nullptr))) return rc;
    ccalib_stats st;
    int method = (flags & CCALIB_USE_LU) ? 0 : 1;
    if ((rc = ccalib_solver_run(g.s, term_type, max_iter, eps, method, &st))) return rc;
    pt.lap("set_start + LM loop");
    double err = 0;
    if ((rc = ccalib_solver_get(g.s, intr, nullptr, nullptr, &err))) return rc;
    if ((rc = solver_get_poses_split(g.s, rvecs, tvecs))) return rc;
    K[0] = intr[0], K[1] = 0, K[2] = intr[2], K[3] = 0, K[4] = intr[1], K[5] = intr[3], K[6] = 0, K[7] = 0, K[8] = 1;
    for (int i = 0; i < ndist; i++) dist[i] = intr[4 + i];
    *rms = std::sqrt(err / (double)total);
    if (stats) *stats = st;
    pt.lap("download");
    return CCALIB_OK;
}

}  // namespace

extern "C" {

int ccalib_calibrate_camera(int nviews, const int* npts, const float* obj_xyz, const float* img_uv, int width,
                            int height, double* K, double* dist, int ndist, double* rvecs, double* tvecs, int flags,
                            int term_type, int max_iter, double eps, const double* init_rt, double* rms,
                            ccalib_stats* stats) {
    return calibrate_camera_impl(nviews, npts, HostViews{obj_xyz, nullptr}, HostViews{img_uv, nullptr}, width, height, K, dist, ndist,
                                 rvecs, tvecs, flags, term_type, max_iter, eps, init_rt, rms, stats);
}

int ccalib_calibrate_camera_views(int nviews, const int* npts, const float* const* obj_views, const float* const* img_views,
                                  int width, int height, double* K, double* dist, int ndist, double* rvecs, double* tvecs,
                                  int flags, int term_type, int max_iter, double eps, const double* init_rt, double* rms,
                                  ccalib_stats* stats) {
    return calibrate_camera_impl(nviews, npts, HostViews{nullptr, obj_views}, HostViews{nullptr, img_views}, width, height, K, dist,
                                 ndist, rvecs, tvecs, flags, term_type, max_iter, eps, init_rt, rms, stats);
}

}  // extern "C"

namespace {

int stereo_calibrate_impl(int nviews, const int* npts, HostViews objv, HostViews img1v, HostViews img2v, double* K1, double* dist1,
                          double* K2, double* dist2, int width, int height, double* R, double* T, double* E, double* F, int flags,
                          int term_type, int max_iter, double eps, const double* init_rt, const double* init_lr, double* rt_out,
                          double* rms, ccalib_stats* stats) {
    auto given_views = [](const HostViews& h) { return h.flat || h.ptrs; };
    if (nviews < 1 || !npts || !given_views(objv) || !given_views(img1v) || !given_views(img2v) || !K1 || !K2 || !dist1 || !dist2 ||
        !R || !T || !rms)
        return set_error(CCALIB_ERR_ARG, "stereoCalibrate: null argument");
    int rc = check_model_flags(flags);
    if (rc) return rc;
    if ((init_rt == nullptr) != (init_lr == nullptr))
        return set_error(CCALIB_ERR_ARG, "stereoCalibrate: init_rt and init_lr must be given together or not at all");
    size_t total = 0;
    for (int i = 0; i < nviews; i++) {
        if (npts[i] < 4) return set_error(CCALIB_ERR_ARG, "view %d has %d points (at least 4 needed)", i, npts[i]);
        for (const HostViews* h : {&objv, &img1v, &img2v})
            if (h->ptrs && !h->ptrs[i]) return set_error(CCALIB_ERR_ARG, "view %d: null point array", i);
        total += npts[i];
    }
    const HostViews imgs[2] = {img1v, img2v};
    double* Kio[2] = {K1, K2};
    double* dio[2] = {dist1, dist2};
    double Ak[2][9] = {{0}}, dk[2][14] = {{0}};
    for (int c = 0; c < 2; c++) {
        if (flags & (CCALIB_FIX_INTRINSIC | CCALIB_USE_INTRINSIC_GUESS | CCALIB_FIX_ASPECT_RATIO | CCALIB_FIX_FOCAL_LENGTH))
            std::memcpy(Ak[c], Kio[c], sizeof(Ak[c]));
        if (flags & (CCALIB_FIX_INTRINSIC | CCALIB_USE_INTRINSIC_GUESS | CCALIB_FIX_K1 | CCALIB_FIX_K2 | CCALIB_FIX_K3 | CCALIB_FIX_K4 |
                     CCALIB_FIX_K5 | CCALIB_FIX_K6 | CCALIB_FIX_TANGENT_DIST))
            std::memcpy(dk[c], dio[c], sizeof(dk[c]));
        if (!(flags & (CCALIB_FIX_INTRINSIC | CCALIB_USE_INTRINSIC_GUESS))) {
            // OpenCV calibrates each camera on its own first
            std::vector<double> rv((size_t)3 * nviews), tv((size_t)3 * nviews);
            double r0;
            if ((rc = calibrate_camera_impl(nviews, npts, objv, imgs[c], width, height, Ak[c], dk[c], 14, rv.data(), tv.data(),
                                            flags, CCALIB_TERM_COUNT | CCALIB_TERM_EPS, 30, DBL_EPSILON, nullptr, &r0, nullptr)))
                return rc;
        }
    }
    if (flags & CCALIB_SAME_FOCAL_LENGTH) {
        static const int avg_idx[] = {0, 4, 2, 5};
        for (int q : avg_idx) Ak[0][q] = Ak[1][q] = (Ak[0][q] + Ak[1][q]) * 0.5;
    }
    double aspect[2] = {0, 0};
    if (flags & CCALIB_FIX_ASPECT_RATIO)
        for (int c = 0; c < 2; c++) aspect[c] = Ak[c][0] / Ak[c][4];

    unsigned char mask[36];
    {
        // cvStereoCalibrateImpl applies one flag word to both cameras
        int f = flags;
        std::memset(mask, 1, 36);
        if (!(f & CCALIB_RATIONAL_MODEL)) f |= CCALIB_FIX_K4 | CCALIB_FIX_K5 | CCALIB_FIX_K6;
        f |= CCALIB_FIX_S1_S2_S3_S4 | CCALIB_FIX_TAUX_TAUY;
        for (int c = 0; c < 2; c++) {
            unsigned char* m = mask + 18 * c;
            if (f & CCALIB_FIX_ASPECT_RATIO) m[0] = 0;
            if (f & CCALIB_FIX_FOCAL_LENGTH) m[0] = m[1] = 0;
            if (f & CCALIB_FIX_PRINCIPAL_POINT) m[2] = m[3] = 0;
            if (f & (CCALIB_ZERO_TANGENT_DIST | CCALIB_FIX_TANGENT_DIST)) m[6] = m[7] = 0;
            if (f & CCALIB_FIX_K1) m[4] = 0;
            if (f & CCALIB_FIX_K2) m[5] = 0;
            if (f & CCALIB_FIX_K3) m[8] = 0;
            if (f & CCALIB_FIX_K4) m[9] = 0;
            if (f & CCALIB_FIX_K5) m[10] = 0;
            if (f & CCALIB_FIX_K6) m[11] = 0;
            for (int q = 12; q < 18; q++) m[q] = 0;
        }
        if (flags & CCALIB_FIX_INTRINSIC) std::memset(mask, 0, 36);  // intrinsics are not parameters at all
    }
    bool shared = views_share_board(nviews, npts, objv);
    SolverGuard g;
    if ((rc = solver_create_views(&g.s, 0, 2, nviews, npts, objv, shared ? 1 : 0, img1v, img2v))) return rc;
    double lr[6];
    const bool given = init_rt && init_lr;
    if (given) {
        std::memcpy(lr, init_lr, sizeof(lr));
    } else {
        // per-view, per-camera planar poses; relative pose = per-component median over views
        for (int c = 0; c < 2; c++)
            if ((rc = ccalib_solver_init_poses(g.s, c, Ak[c], dk[c], nullptr))) return rc;
        if ((rc = ccalib_solver_init_relative_pose(g.s, lr))) return rc;
    }
    double intr[36];
    for (int c = 0; c < 2; c++) {
        if (flags & CCALIB_ZERO_TANGENT_DIST) dk[c][2] = dk[c][3] = 0;
        double* ip = intr + 18 * c;
        ip[0] = Ak[c][0], ip[1] = Ak[c][4], ip[2] = Ak[c][2], ip[3] = Ak[c][5];
        std::memcpy(ip + 4, dk[c], sizeof(dk[c]));
    }
    const bool fixed = (flags & CCALIB_FIX_INTRINSIC) != 0;
    if (fixed) aspect[0] = aspect[1] = 0;  // no ties to enforce between fixed values
    const int opts = fixed ? CCALIB_STEREO_FIX_INTRINSIC : (flags & CCALIB_SAME_FOCAL_LENGTH) ? CCALIB_STEREO_SAME_FOCAL_LENGTH : 0;
    if ((rc = ccalib_solver_set_start(g.s, intr, mask, aspect, opts, given ? init_rt : nullptr, lr)))
        return rc;
    ccalib_stats st;
    int method = (flags & CCALIB_USE_LU) ? 0 : 1;
    if ((rc = ccalib_solver_run(g.s, term_type, max_iter, eps, method, &st))) return rc;
    double err = 0;
    if ((rc = ccalib_solver_get(g.s, intr, rt_out ? rt_out : nullptr, lr, &err))) return rc;
    for (int c = 0; c < 2; c++) {
        const double* ip = intr + 18 * c;
        double* Kc = Kio[c];
        Kc[0] = ip[0], Kc[1] = 0, Kc[2] = ip[2], Kc[3] = 0, Kc[4] = ip[1], Kc[5] = ip[3], Kc[6] = 0, Kc[7] = 0, Kc[8] = 1;
        std::memcpy(dio[c], ip + 4, 14 * sizeof(double));
    }
    double Rm[9];
    host_rodrigues_v2m(lr, Rm, nullptr);
    std::memcpy(R, Rm, sizeof(Rm));
    std::memcpy(T, lr + 3, 3 * sizeof(double));
    if (E || F) {
        const double* t = lr + 3;
        double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0}, Em[9];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Em[a * 3 + b] = tx[a * 3] * Rm[b] + tx[a * 3 + 1] * Rm[3 + b] + tx[a * 3 + 2] * Rm[6 + b];
        if (E) std::memcpy(E, Em, sizeof(Em));
        if (F) {
            // F = K2^-T E K1^-1 for K = [fx 0 cx; 0 fy cy; 0 0 1], scaled so that F[2][2] = 1
            auto kinv = [](const double* Kc, double* o) {
                double fx = Kc[0], fy = Kc[4], cx = Kc[2], cy = Kc[5];
                double v[9] = {1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1};
                std::memcpy(o, v, sizeof(v));
            };
            double i1[9], i2[9], tmp[9], Fm[9];
            kinv(K1, i1), kinv(K2, i2);
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) tmp[a * 3 + b] = i2[a] * Em[b] + i2[3 + a] * Em[3 + b] + i2[6 + a] * Em[6 + b];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) Fm[a * 3 + b] = tmp[a * 3] * i1[b] + tmp[a * 3 + 1] * i1[3 + b] + tmp[a * 3 + 2] * i1[6 + b];
            double sc = std::fabs(Fm[8]) > 0 ? 1. / Fm[8] : 1;
            for (int a = 0; a < 9; a++) F[a] = Fm[a] * sc;
        }
    }
    *rms = std::sqrt(err / (2. * (double)total));
    if (stats) *stats = st;
    return CCALIB_OK;
}

}  // namespace

extern "C" {

int ccalib_stereo_calibrate(int nviews, const int* npts, const float* obj_xyz, const float* img1_uv,
                            const float* img2_uv, double* K1, double* dist1, double* K2, double* dist2, int width,
                            int height, double* R, double* T, double* E, double* F, int flags, int term_type,
                            int max_iter, double eps, const double* init_rt, const double* init_lr, double* rt_out,
                            double* rms, ccalib_stats* stats) {
    return stereo_calibrate_impl(nviews, npts, HostViews{obj_xyz, nullptr}, HostViews{img1_uv, nullptr}, HostViews{img2_uv, nullptr},
                                 K1, dist1, K2, dist2, width, height, R, T, E, F, flags, term_type, max_iter, eps, init_rt, init_lr,
                                 rt_out, rms, stats);
}

int ccalib_stereo_calibrate_views(int nviews, const int* npts, const float* const* obj_views, const float* const* img1_views,
                                  const float* const* img2_views, double* K1, double* dist1, double* K2, double* dist2, int width,
                                  int height, double* R, double* T, double* E, double* F, int flags, int term_type, int max_iter,
                                  double eps, const double* init_rt, const double* init_lr, double* rt_out, double* rms,
                                  ccalib_stats* stats) {
    return stereo_calibrate_impl(nviews, npts, HostViews{nullptr, obj_views}, HostViews{nullptr, img1_views},
                                 HostViews{nullptr, img2_views}, K1, dist1, K2, dist2, width, height, R, T, E, F, flags, term_type,
                                 max_iter, eps, init_rt, init_lr, rt_out, rms, stats);
}

}  // extern "C"
