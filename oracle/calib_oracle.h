/*
 * calib_oracle.h -- CPU restatement of the calibration hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker, not the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link, load or call it.
 *
 * PARITY UNPINNED.  The algorithm restated here lives in a third-party
 * dependency of the reference that is absent from the reference tree:
 * OpenCV 4.1.0, module calib3d (pinned by the import libraries named in
 * /root/reference/camera_calibration.vcxproj:114 and :151).  It was restated
 * from knowledge of the published algorithm
 *   modules/calib3d/src/calibration.cpp      (cvCalibrateCamera2Internal,
 *       cvStereoCalibrateImpl, cvProjectPoints2Internal, cvRodrigues2,
 *       cvComposeRT, cvInitIntrinsicParams2D, cvFindExtrinsicCameraParams2)
 *   modules/calib3d/src/compat_ptsetreg.cpp  (CvLevMarq)
 * and is anchored on the reference's call sites:
 *   /root/reference/monocular_circle.cpp:93-94     calibrateCamera(..., flags=0)
 *   /root/reference/monocular_circle.cpp:5-29      projectPoints RMS re-check
 *   /root/reference/monocular_chessboard.cpp:315-317 calibrateCameraRO(iFixedPoint=-1,
 *                                                   FIX_ASPECT_RATIO|USE_LU)
 *   /root/reference/binocular_chessboard.cpp:147-162 initCameraMatrix2D + stereoCalibrate
 *   /root/reference/binocular_circle.cpp:147-163     same, circle grid
 * The reference holds no tests, fixtures or golden vectors for this path and
 * OpenCV is not installed, so nothing external pins these results; the oracle
 * is instead cross-checked by independent means (finite differences, a
 * generic least-squares solver, dense-vs-Schur agreement) in tests/.
 *
 * All arithmetic is FP64.  Row-major matrices throughout.
 */
#ifndef CALIB_ORACLE_H
#define CALIB_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Flag bit values of OpenCV 4.1.0 calib3d.hpp (the integers the reference passes). */
enum {
    ORC_USE_INTRINSIC_GUESS = 0x00001,
    ORC_FIX_ASPECT_RATIO = 0x00002,
    ORC_FIX_PRINCIPAL_POINT = 0x00004,
    ORC_ZERO_TANGENT_DIST = 0x00008,
    ORC_FIX_FOCAL_LENGTH = 0x00010,
    ORC_FIX_K1 = 0x00020,
    ORC_FIX_K2 = 0x00040,
    ORC_FIX_K3 = 0x00080,
    ORC_FIX_INTRINSIC = 0x00100,
    ORC_SAME_FOCAL_LENGTH = 0x00200,
    ORC_FIX_K4 = 0x00800,
    ORC_FIX_K5 = 0x01000,
    ORC_FIX_K6 = 0x02000,
    ORC_RATIONAL_MODEL = 0x04000,
    ORC_THIN_PRISM_MODEL = 0x08000,
    ORC_FIX_S1_S2_S3_S4 = 0x10000,
    ORC_USE_LU = (1 << 17),
    ORC_TILTED_MODEL = 0x40000,
    ORC_FIX_TAUX_TAUY = 0x80000,
    ORC_USE_QR = 0x100000,
    ORC_FIX_TANGENT_DIST = 0x200000,
    ORC_USE_EXTRINSIC_GUESS = (1 << 22)
};

enum { ORC_TERM_COUNT = 1, ORC_TERM_EPS = 2 };
enum { ORC_NINTRINSIC = 18 };

/* solver choices for the damped normal equations */
enum { ORC_SOLVE_LU = 0, ORC_SOLVE_SVD = 1, ORC_SOLVE_CHOLESKY = 2 };

/*
 * What-if switches for the two details of OpenCV 4.1.0 that were restated from memory and could not be checked
 * against its text (all 0 = the restatement the product is tested against).  They exist so that the size of the
 * effect can be measured (tests/test_oracle_variants.py): a maintainer with a real OpenCV then knows what a
 * mismatch from either cause would look like.
 *   ORC_VARIANT_PP_HALF_SIZE     cvInitIntrinsicParams2D seeds the principal point at (w/2, h/2) instead of
 *                                ((w-1)/2, (h-1)/2)
 *   ORC_VARIANT_SAME_FOCAL_MASK  CALIB_SAME_FOCAL_LENGTH also masks camera 1's fx, fy out of the solve instead of
 *                                only overwriting them from camera 0 before every pass
 */
enum { ORC_VARIANT_PP_HALF_SIZE = 1, ORC_VARIANT_SAME_FOCAL_MASK = 2 };
extern int orc_variants;
void orc_set_variants(int bits);

/*
 * Per-iteration record of the LM run inside the next orc_calibrate_camera / orc_stereo_calibrate call: ORC_TRACE_FIELDS
 * doubles per finished iteration -- iteration number, damping exponent (lambdaLg10) of the first trial, exponent of
 * the accepted trial, number of trial steps, residual sum at the committed point, residual sum at the accepted
 * point.  buf holds cap records; NULL switches the record off.  orc_trace_count() = records written by the last run.
 * (A stereo call without an intrinsic guess calibrates each camera first; the record is that of the stereo loop.)
 */
enum { ORC_TRACE_FIELDS = 6 };
void orc_set_trace(double* buf, int cap);
int orc_trace_count(void);

/* ---- geometry ---------------------------------------------------------- */
/* cvRodrigues2, vector -> matrix.  J (may be NULL) is 3x9: J[i*9+k] = dR[k]/dr[i]. */
void orc_rodrigues_v2m(const double r[3], double R[9], double J[27]);
/* cvRodrigues2, matrix -> vector.  J (may be NULL) is 3x9: J[i*9+k] = dr[i]/dR[k]. */
void orc_rodrigues_m2v(const double Rin[9], double r[3], double J[27]);
/* cvComposeRT: (r3,t3) = (r2,t2) o (r1,t1).  Derivative outputs are 3x3, any may be NULL. */
void orc_compose_rt(const double r1[3], const double t1[3], const double r2[3], const double t2[3],
                    double r3[3], double t3[3], double dr3dr1[9], double dr3dr2[9],
                    double dt3dr2[9], double dt3dt1[9], double dt3dt2[9]);
/* cvProjectPoints2Internal.  obj: n x 3 doubles.  fc = {fx,fy,cx,cy}.  k: 14 coefficients
 * (k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 taux tauy; taux,tauy must be 0).
 * aspect != 0 selects the fixed-aspect Jacobian form.  Outputs (any may be NULL):
 * m: n x 2, Je: 2n x 6 (dp/dr | dp/dt), Ji: 2n x 18 (dpdf(2) dpdc(2) dpdk(14)). */
void orc_project_points(int n, const double* obj, const double r[3], const double t[3],
                        const double fc[4], const double k[14], double aspect, double* m,
                        double* Je, double* Ji);

/* ---- initialisation ---------------------------------------------------- */
/* cvFindHomography(method 0): normalised DLT + LM refinement.  src,dst: n x 2. */
int orc_find_homography(int n, const double* src, const double* dst, double H[9]);
/* cvInitIntrinsicParams2D.  obj: total x 3, img: total x 2, npts[nviews]. */
int orc_init_intrinsic_params_2d(int nviews, const int* npts, const double* obj, const double* img,
                                 int width, int height, double aspect, double A[9]);
/* cvFindExtrinsicCameraParams2 (no extrinsic guess). */
int orc_find_extrinsic(int n, const double* obj, const double* img, const double A[9],
                       const double k[14], double r[3], double t[3]);

/* ---- calibration ------------------------------------------------------- */
typedef struct {
    int iters;        /* accepted LM iterations (CvLevMarq::iters)          */
    int jac_passes;   /* passes that evaluated Jacobians                    */
    int err_passes;   /* passes that evaluated residuals only               */
    int lambda_lg10;  /* final damping exponent                             */
    double final_err; /* sum of squared residuals at the returned params    */
} orc_stats;

/*
 * cvCalibrateCamera2Internal with iFixedPoint = -1.
 *   obj/img: concatenated per-view points (doubles; the caller widens the floats).
 *   A (3x3) and dist (ndist in {4,5,8,12,14}) are in/out as in OpenCV.
 *   rvecs/tvecs: nviews x 3 out.  If init_rt != NULL it supplies nviews x 6
 *   starting poses and skips the per-view pose initialisation.
 *   schur = 0: dense (18+6N)^2 system exactly as OpenCV assembles it;
 *   schur = 1: the same LM state machine with the arrow system solved by
 *              per-view 6x6 elimination (scales to large N).
 *   nthreads: OpenMP threads for the schur path (dense path is serial).
 * Returns the RMS reprojection error, or a negative value on bad arguments.
 */
double orc_calibrate_camera(int nviews, const int* npts, const double* obj, const double* img,
                            int width, int height, double A[9], double* dist, int ndist,
                            double* rvecs, double* tvecs, int flags, int term_type,
                            int max_iter, double eps, const double* init_rt, int schur,
                            int solve_override, int nthreads, orc_stats* stats);

/*
 * cvStereoCalibrateImpl.  dist1/dist2 hold 14 doubles each (in/out).
 * R (3x3), T (3), E (3x3), F (3x3) out.  init_rt (nviews x 6, camera-0 poses)
 * plus init_lr (6) skip the pose initialisation when non-NULL.
 * Only USE_INTRINSIC_GUESS / FIX_INTRINSIC entry is supported (the reference's
 * call sites pass USE_INTRINSIC_GUESS).
 */
double orc_stereo_calibrate(int nviews, const int* npts, const double* obj, const double* img1,
                            const double* img2, double A1[9], double dist1[14], double A2[9],
                            double dist2[14], int width, int height, double R[9], double T[3],
                            double E[9], double F[9], int flags, int term_type, int max_iter,
                            double eps, const double* init_rt, const double* init_lr,
                            double* rt_out, int schur, int nthreads, orc_stats* stats);

/* Sum of squared reprojection residuals + optional per-view RMS, as the
 * reference recomputes it in monocular_circle.cpp:5-29. */
double orc_reprojection_error(int nviews, const int* npts, const double* obj, const double* img,
                              const double A[9], const double* dist, int ndist,
                              const double* rvecs, const double* tvecs, double* per_view_rms);

/* One evaluation of the per-view normal-equation pieces at fixed parameters
 * (used to check device kernels block by block).  free_idx lists the nfree
 * intrinsic columns kept (indices into the 18-vector).  Outputs:
 *   U  nfree x nfree (full symmetric), gc nfree, per view: V 6x6, W nfree x 6, ge 6.
 * Returns the sum of squared residuals. */
double orc_mono_blocks(int nviews, const int* npts, const double* obj, const double* img,
                       const double intr[18], double aspect, const double* rt, int nfree,
                       const int* free_idx, double* U, double* gc, double* V, double* W,
                       double* ge);

/* The same for the stereo system: one Jacobian pass of cvStereoCalibrateImpl at the given intrinsics (A, 14
 * coefficients per camera), camera-0 poses rt (nviews x 6) and relative pose lr (6), under `flags` (the ties
 * CALIB_FIX_ASPECT_RATIO and CALIB_SAME_FOCAL_LENGTH are applied before the pass, as in the loop).  The shared
 * block is ns = 42 wide: [om_LR T_LR | 18 intrinsics of camera 0 | 18 of camera 1], every intrinsic column kept.
 *   U  42 x 42, filled where OpenCV fills JtJ (diagonal blocks in full, the blocks above the diagonal), gs 42,
 *   per view: V 6 x 6 (the pose block after both cameras), W 42 x 6, ge 6.
 * Returns the sum of squared residuals over both cameras. */
double orc_stereo_blocks(int nviews, const int* npts, const double* obj, const double* img1, const double* img2,
                         const double A1[9], const double dist1[14], const double A2[9], const double dist2[14],
                         int flags, const double* rt, const double* lr, double* U, double* gs, double* V, double* W,
                         double* ge);

#ifdef __cplusplus
}
#endif
#endif
