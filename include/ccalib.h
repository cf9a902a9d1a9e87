/*
 * ccalib.h -- C ABI of the MI355X-native calibration core (libccalib.so).
 *
 * Drop-in boundary for the Levenberg-Marquardt path that the reference reaches through
 * OpenCV 4.1.0:
 *   cv::calibrateCamera     /root/reference/monocular_circle.cpp:93-94
 *   cv::calibrateCameraRO   /root/reference/monocular_chessboard.cpp:315-317 (iFixedPoint = -1)
 *   cv::stereoCalibrate     /root/reference/binocular_chessboard.cpp:152-162,
 *                           /root/reference/binocular_circle.cpp:153-163
 *   cv::projectPoints       /root/reference/monocular_circle.cpp:18-19 (RMS re-check)
 *   cv::initCameraMatrix2D  /root/reference/binocular_chessboard.cpp:147-148
 *
 * Plain pointers and sizes only; every function returns a status code and never throws.
 * ccalib_last_error() returns a thread-local message for the last non-zero status.
 * The library computes on the GPU only: there is no CPU fallback, and every compute entry
 * point fails with CCALIB_ERR_NO_DEVICE when no HIP device is usable.
 *
 * Two levels:
 *   whole-call  (ccalib_calibrate_camera, ccalib_stereo_calibrate, ...): what an FFI binding of
 *               the cv:: functions above would call;
 *   solver      (ccalib_solver_*): the device-resident problem the host LM driver steps; it is
 *               what bench.py times (inputs already in HBM) and what multi-GPU runs shard.
 */
#ifndef CCALIB_H
#define CCALIB_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status ---------------------------------------------------------------------------- */
enum {
    CCALIB_OK = 0,
    CCALIB_ERR_ARG = 1,         /* bad argument (the cases where OpenCV raises cv::Exception) */
    CCALIB_ERR_HIP = 2,         /* a HIP / RCCL call failed                                   */
    CCALIB_ERR_NO_DEVICE = 3,   /* no usable GPU                                              */
    CCALIB_ERR_UNSUPPORTED = 4, /* flag combination outside this build (tilted / thin-prism)  */
    CCALIB_ERR_STATE = 5,       /* call order violated                                        */
    CCALIB_ERR_NUMERIC = 6      /* singular reduced system                                    */
};
const char* ccalib_last_error(void);
const char* ccalib_version(void);

/* ---- flag and criteria values: the integers of OpenCV 4.1.0 calib3d.hpp ------------------ */
enum {
    CCALIB_USE_INTRINSIC_GUESS = 0x00001,
    CCALIB_FIX_ASPECT_RATIO = 0x00002,
    CCALIB_FIX_PRINCIPAL_POINT = 0x00004,
    CCALIB_ZERO_TANGENT_DIST = 0x00008,
    CCALIB_FIX_FOCAL_LENGTH = 0x00010,
    CCALIB_FIX_K1 = 0x00020,
    CCALIB_FIX_K2 = 0x00040,
    CCALIB_FIX_K3 = 0x00080,
    CCALIB_FIX_INTRINSIC = 0x00100,
    CCALIB_SAME_FOCAL_LENGTH = 0x00200,
    CCALIB_FIX_K4 = 0x00800,
    CCALIB_FIX_K5 = 0x01000,
    CCALIB_FIX_K6 = 0x02000,
    CCALIB_RATIONAL_MODEL = 0x04000,
    CCALIB_THIN_PRISM_MODEL = 0x08000,
    CCALIB_FIX_S1_S2_S3_S4 = 0x10000,
    CCALIB_USE_LU = (1 << 17),
    CCALIB_TILTED_MODEL = 0x40000,
    CCALIB_FIX_TAUX_TAUY = 0x80000,
    CCALIB_USE_QR = 0x100000,
    CCALIB_FIX_TANGENT_DIST = 0x200000,
    CCALIB_USE_EXTRINSIC_GUESS = (1 << 22)
};
enum { CCALIB_TERM_COUNT = 1, CCALIB_TERM_EPS = 2 };

/*
 * How many distortion coefficients cv::calibrateCamera / cv::calibrateCameraRO (stereo = 0) or cv::stereoCalibrate
 * (stereo = 1) of OpenCV 4.1.0 hand back for a flag word (calibration.cpp, recalled -- OpenCV is not part of the
 * reference tree): the vector is prepared with 14 entries (12 in calibrateCamera under CALIB_THIN_PRISM_MODEL without
 * CALIB_TILTED_MODEL) and cut back to 5 unless one of RATIONAL / THIN_PRISM / TILTED is set.  Both C++ adaptors
 * (ccalib.hpp, ccalib_cv.hpp) size their outputs with it.
 */
int ccalib_dist_coeffs_out(int flags, int stereo);

typedef struct {
    int iters;          /* accepted LM iterations (CvLevMarq::iters)                    */
    int jac_passes;     /* passes over all points that evaluated residuals + Jacobians  */
    int err_passes;     /* passes that evaluated residuals only                         */
    int lambda_lg10;    /* final damping exponent                                       */
    double final_err;   /* sum of squared residuals at the returned parameters          */
    double device_ms;   /* wall time spent inside the LM loop (host clock)              */
} ccalib_stats;

/* ---- whole-call level ------------------------------------------------------------------- */

/*
 * cv::calibrateCamera / cv::calibrateCameraRO(iFixedPoint = -1)
 * (monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317).
 *   npts[nviews]          points per view (>= 4 each)
 *   obj_xyz               sum(npts) x 3 floats (Point3f), view after view
 *   img_uv                sum(npts) x 2 floats (Point2f)
 *   K[9]                  in/out 3x3 row-major.  Read when USE_INTRINSIC_GUESS or FIX_ASPECT_RATIO
 *                         is set (pass identity for "empty", as OpenCV substitutes eye()).
 *   dist[ndist]           in/out, ndist in {4,5,8,12,14}; read only with USE_INTRINSIC_GUESS
 *   rvecs, tvecs          out, nviews x 3 each
 *   init_rt               optional nviews x 6 starting poses (r|t); NULL runs the per-view
 *                         pose initialisation OpenCV performs
 *   rms                   out, sqrt(sum err^2 / sum npts)
 */
int ccalib_calibrate_camera(int nviews, const int* npts, const float* obj_xyz, const float* img_uv,
                            int width, int height, double* K, double* dist, int ndist,
                            double* rvecs, double* tvecs, int flags, int term_type, int max_iter,
                            double eps, const double* init_rt, double* rms, ccalib_stats* stats);

/*
 * cv::stereoCalibrate (binocular_chessboard.cpp:152-162, binocular_circle.cpp:153-163).
 *   K1,K2 in/out 3x3; dist1,dist2 in/out 14 doubles each; R 3x3, T 3, E 3x3, F 3x3 out.
 *   init_rt (nviews x 6 camera-0 poses) and init_lr (6: om_LR|T_LR) are an optional start: both or neither
 *   (exactly one of them is CCALIB_ERR_ARG); with neither, the poses and the relative pose are initialised on the
 *   device as OpenCV does.
 *   rt_out optional nviews x 6 refined camera-0 poses.
 *   rms = sqrt(sum err^2 / (2 * sum npts)).
 */
int ccalib_stereo_calibrate(int nviews, const int* npts, const float* obj_xyz, const float* img1_uv,
                            const float* img2_uv, double* K1, double* dist1, double* K2,
                            double* dist2, int width, int height, double* R, double* T, double* E,
                            double* F, int flags, int term_type, int max_iter, double eps,
                            const double* init_rt, const double* init_lr, double* rt_out,
                            double* rms, ccalib_stats* stats);

/*
 * The same two calls over one pointer per view instead of one flat array: obj_views[i] -> npts[i] x 3 floats,
 * img_views[i] -> npts[i] x 2 floats.  This is the memory a std::vector<std::vector<cv::Point3f>> /
 * std::vector<std::vector<cv::Point2f>> already holds (monocular_chessboard.cpp:249, :295-305: every view's
 * object points are a copy of one board; binocular_chessboard.cpp:118-131), so a binding passes &views[i][0]
 * and nothing is gathered on the host: the library copies each view straight into its pinned upload buffer.
 * Entries of obj_views may all point at the same board.  Every other argument, the results and the error
 * behaviour are those of the flat calls.
 */
int ccalib_calibrate_camera_views(int nviews, const int* npts, const float* const* obj_views,
                                  const float* const* img_views, int width, int height, double* K, double* dist,
                                  int ndist, double* rvecs, double* tvecs, int flags, int term_type, int max_iter,
                                  double eps, const double* init_rt, double* rms, ccalib_stats* stats);
int ccalib_stereo_calibrate_views(int nviews, const int* npts, const float* const* obj_views,
                                  const float* const* img1_views, const float* const* img2_views, double* K1,
                                  double* dist1, double* K2, double* dist2, int width, int height, double* R,
                                  double* T, double* E, double* F, int flags, int term_type, int max_iter, double eps,
                                  const double* init_rt, const double* init_lr, double* rt_out, double* rms,
                                  ccalib_stats* stats);

/*
 * Memory kept between calls.  No counterpart in the reference: OpenCV's calls own nothing on a device.  Every
 * whole-call entry point builds a device-resident problem and tears it down again; the device blocks, the pinned
 * block and the stream of a destroyed problem are kept (per device, up to CCALIB_POOL_MB megabytes of device memory
 * in total, default 1024; CCALIB_POOL_MB=0 keeps nothing) and handed to the next one that they fit, which saves the
 * allocation and release calls (about a millisecond per call).  Two pinned 8 MB upload buffers are created on first
 * use and kept as well.  This call returns all cached blocks to the driver; the result is the number of device
 * bytes released.  A request that the driver cannot satisfy releases the cache by itself and tries again.
 */
size_t ccalib_release_cached_memory(void);

/*
 * cv::projectPoints over many views + the RMS the reference recomputes from it
 * (monocular_circle.cpp:5-29).  proj_uv (optional) receives sum(npts) x 2 floats,
 * per_view_rms (optional) nviews doubles, total_rms the global figure.
 */
int ccalib_reprojection_error(int nviews, const int* npts, const float* obj_xyz, const float* img_uv,
                              const double* K, const double* dist, int ndist, const double* rvecs,
                              const double* tvecs, float* proj_uv, double* per_view_rms,
                              double* total_rms);

/*
 * The pass of ccalib_reprojection_error as an entry point that allocates nothing.  It has no counterpart in the
 * reference: it exists for callers that cannot allocate device memory -- code under HIP graph capture, serving
 * loops, harnesses that forbid hipMalloc.  It calls no hipMalloc / hipFree / stream or event creation; its working
 * storage (view offsets, (r|t) pose rows, per-view squared errors, a validity word) is one static device array of
 * about 3.9 MB per GPU, which every GPU the library runs on holds until the process ends.
 *   Every array argument whose name starts with d_ is a DEVICE pointer on the current device; K (3x3) and
 *   dist[ndist] are host pointers, read before the call returns.
 *   d_npts[nviews]     points per view (>= 4 each), nviews <= 65536 (more is rejected with CCALIB_ERR_ARG)
 *   max_npts           an upper bound of every d_npts[i];  capacity = points the three point arrays hold
 *   d_obj_xyz, d_img_uv  capacity x 3 / capacity x 2 floats, view after view;  d_rvecs, d_tvecs  nviews x 3 doubles
 *   d_proj_uv (optional, capacity x 2 floats), d_per_view_rms (optional, nviews), d_total_rms (optional, 1)
 *   stream             a hipStream_t (NULL = the default stream); the work is enqueued there and the call returns
 *                      without waiting, so the outputs are valid once that stream reaches this point
 * A point count outside [4, max_npts] or a total beyond `capacity` is found on the device: d_per_view_rms and
 * d_total_rms then receive NaN, the first pixel of d_proj_uv receives (NaN, NaN) and the rest of d_proj_uv is left
 * untouched; nothing is read past the stated sizes.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 * calls enqueued on one stream are ordered and safe.
 * Under HIP graph capture: make one ordinary call on the device first (the first call per device asks the runtime for
 * the array's address, which may load the code object and is not a capturable operation); after that a call issues
 * only kernel launches on `stream` and can be captured.  K, dist, nviews, max_npts and capacity are kernel arguments
 * by value: they are baked into the captured graph, the arrays behind the device pointers are read at replay.
 */
int ccalib_reprojection_error_noalloc(int nviews, const int* d_npts, int max_npts, int capacity,
                                      const float* d_obj_xyz, const float* d_img_uv, const double* K,
                                      const double* dist, int ndist, const double* d_rvecs, const double* d_tvecs,
                                      float* d_proj_uv, double* d_per_view_rms, double* d_total_rms, void* stream);

/* cv::initCameraMatrix2D (binocular_chessboard.cpp:147-148).  K out, 3x3. */
int ccalib_init_camera_matrix_2d(int nviews, const int* npts, const float* obj_xyz,
                                 const float* img_uv, int width, int height, double aspect,
                                 double* K);

/*
 * The result files the reference writes with cv::FileStorage, in the same "%YAML:1.0" dialect
 * (!!opencv-matrix nodes), readable by FileStorage::READ (binocular_chessboard.cpp:285-295).
 *   mono    imageSize, M, D, R (sequence of 3x1 rvecs), t (sequence of 3x1 tvecs)
 *           (monocular_chessboard.cpp:321-328)
 *   stereo  intrinsics file M1, D1, M2, D2 (binocular_chessboard.cpp:203-210); extrinsics file
 *           imageSize, R, T and, when given, E, F (binocular_chessboard.cpp:234-241; the R1, R2, P1,
 *           P2, Q entries there come from cv::stereoRectify, which is not part of this library)
 */
int ccalib_write_mono_yaml(const char* path, int width, int height, const double* K, const double* dist,
                           int ndist, int nviews, const double* rvecs, const double* tvecs);
int ccalib_write_stereo_yaml(const char* intrinsics_path, const char* extrinsics_path, int width, int height,
                             const double* K1, const double* dist1, int ndist1, const double* K2,
                             const double* dist2, int ndist2, const double* R, const double* T,
                             const double* E, const double* F);

/* ---- solver level: device-resident problem + host LM driver ------------------------------ */
typedef struct ccalib_solver ccalib_solver;

/*
 * Creates a problem on HIP device `device` and uploads the points (float -> kept as float in HBM,
 * widened in registers).  ncams is 1 (calibrateCamera) or 2 (stereoCalibrate); img2_uv is NULL
 * for ncams == 1.  If every view has the same object points, pass shared_board = 1 and only the
 * first view's npts[0] x 3 floats in obj_xyz: the board then lives in LDS instead of HBM.
 * In a multi-GPU run every rank creates a solver over ITS OWN views.
 */
int ccalib_solver_create(ccalib_solver** out, int device, int ncams, int nviews, const int* npts,
                         const float* obj_xyz, int shared_board, const float* img1_uv,
                         const float* img2_uv);
void ccalib_solver_destroy(ccalib_solver* s);

/*
 * Sets the starting point and the free-parameter set.
 *   intr      ncams x 18 doubles: fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 taux tauy
 *   imask     ncams x 18 bytes, 1 = optimised (entries 12..17 must be 0)
 *   aspect    ncams doubles; non-zero ties fx = fy * aspect (CALIB_FIX_ASPECT_RATIO)
 *   stereo_opts  stereo only, bits: CCALIB_STEREO_SAME_FOCAL_LENGTH copies fx,fy of camera 0 into camera 1
 *             before every pass; CCALIB_STEREO_FIX_INTRINSIC (every imask entry 0) takes the intrinsics out of
 *             the parameter vector altogether, as cv::stereoCalibrate does under CALIB_FIX_INTRINSIC: the
 *             convergence norm then runs over om_LR, T_LR and the poses only
 *   rt        nviews x 6 poses of camera 0, or NULL to start from the poses ccalib_solver_init_poses
 *             left on the device;  lr  6 doubles (stereo; NULL for mono)
 */
enum { CCALIB_STEREO_SAME_FOCAL_LENGTH = 1, CCALIB_STEREO_FIX_INTRINSIC = 2 };
int ccalib_solver_set_start(ccalib_solver* s, const double* intr, const unsigned char* imask,
                            const double* aspect, int stereo_opts, const double* rt,
                            const double* lr);

/*
 * Starting values, computed on the device from the points the solver already holds: the steps OpenCV
 * performs before its LM loop inside calibrateCamera / stereoCalibrate (monocular_circle.cpp:93-94,
 * binocular_chessboard.cpp:147-148 and :152-162).  One lane per view.
 *   init_intrinsics     cvInitIntrinsicParams2D / cv::initCameraMatrix2D for camera `cam`: focal
 *                       lengths from the vanishing points of the per-view homographies, principal
 *                       point at the image centre.  K out, 3x3.  In a multi-GPU run (after use_rccl /
 *                       use_callback) the normal equations are summed over all ranks.
 *   init_poses          cvFindExtrinsicCameraParams2 for every view of camera `cam` at intrinsics K
 *                       (3x3) and distortion dist14 (14 doubles, NULL = zero): homography of the
 *                       undistorted points, pose from it, 6-parameter LM refinement.  Camera 0's poses
 *                       stay on the device as the starting poses (pass rt = NULL to set_start);
 *                       rt_out (optional) receives nviews x 6 (r|t).  Planar rigs only.
 *   init_relative_pose  stereo, after init_poses for both cameras: lr[6] = per-component median of the relative
 *                       pose (om | T) of camera 1 w.r.t. camera 0 over the views.  In a multi-GPU run (after
 *                       use_rccl / use_callback; every rank must call it) the median is the exact one over the
 *                       views of ALL ranks, found by bisection with all-reduced counts: every rank receives the
 *                       bits a single solver over all views computes.
 */
int ccalib_solver_init_intrinsics(ccalib_solver* s, int cam, int width, int height, double aspect,
                                  double* K);
int ccalib_solver_init_poses(ccalib_solver* s, int cam, const double* K, const double* dist14,
                             double* rt_out);
int ccalib_solver_init_relative_pose(ccalib_solver* s, double* lr);

/*
 * Joins a multi-GPU run: after this call every reduction the LM loop needs (the shared normal-
 * equation block, its right-hand side and three scalars per pass) is summed over all ranks.
 * Exactly one of the two forms is used:
 *   ccalib_solver_use_rccl      in-stream ncclAllReduce over a communicator built from `unique_id`
 *                               (128 bytes from ccalib_rccl_unique_id on rank 0, shared by the
 *                               caller, e.g. through torch.distributed)
 *   ccalib_solver_use_callback  host callback, called with a host buffer to be summed in place
 */
int ccalib_rccl_unique_id(char id[128]);
int ccalib_solver_use_rccl(ccalib_solver* s, int nranks, int rank, const char unique_id[128]);
typedef int (*ccalib_allreduce_fn)(void* user, double* buf, int n);
int ccalib_solver_use_callback(ccalib_solver* s, ccalib_allreduce_fn fn, void* user);

/*
 * Runs the CvLevMarq state machine (damping, reduced solve, convergence on the host; passes on
 * the GPU) until done.  solve_method: 0 LU (CALIB_USE_LU), 1 eigen/SVD (OpenCV's default).
 * May be called again to continue (bench.py runs a fixed number of iterations with eps = 0).
 */
int ccalib_solver_run(ccalib_solver* s, int term_type, int max_iter, double eps, int solve_method,
                      ccalib_stats* stats);

/*
 * How often ccalib_solver_run kept the GPU one iteration ahead of the host state machine (CvLevMarq waits for the
 * sums of every pass before the next can start; here the device predicts the host's next step -- same IEEE
 * operations in the same order -- and the kernels of that step are already queued; the host uses them only when
 * its own step is bit for bit the predicted one, so results never depend on it).  out[0] iterations enqueued
 * ahead, out[1] of them used by the driver, out[2] dropped (step rejected, convergence reached, or no match).
 * A retry launched behind a trial step that the damping history says will be rejected (its step worked out on the
 * host with the driver's own function, used only on a bitwise match) is counted the same way.
 * Since the solver was created.  The first mechanism is off unless CCALIB_CHAIN=1 is in the environment: measured on
 * MI355X it hides the host's turn-around but costs as much in device-side work, so it is kept as an option only.
 */
int ccalib_solver_counters(ccalib_solver* s, long long out[3]);
/*
 * Passes over all points that the solver actually ENQUEUED since it was created: out[0] Jacobian passes (residuals +
 * Jacobians + block accumulation), out[1] residual-only passes.  ccalib_stats counts the passes CvLevMarq asks for; a
 * residual pass asked for at a point whose Jacobian pass was run ahead is answered from that pass and never runs, and
 * a Jacobian pass run ahead of a step that is then rejected runs without being asked for.
 */
int ccalib_solver_executed_passes(ccalib_solver* s, long long out[2]);

/* Results: intr ncams x 18, rt nviews x 6, lr 6 (NULL ok), err = sum of squared residuals. */
int ccalib_solver_get(ccalib_solver* s, double* intr, double* rt, double* lr, double* err);

/*
 * Single passes, exposed so that tests can check each kernel against the oracle block by block
 * and so that bench.py can time the dominant kernel alone.
 *   eval_error     sum of squared residuals at the current parameters
 *   eval_jacobian  fills the per-view blocks in HBM; returns the shared block U (ns x ns, upper
 *                  triangle, ns = number of free shared parameters), its gradient gs and err
 *   fetch_blocks   copies per-view V (nviews x 36), W (nviews x 18 x 6), ge (nviews x 6) to host.  Stereo solver: the
 *                  view systems after cvStereoCalibrateImpl's rewrite of camera 1's rows (binocular_chessboard.cpp:
 *                  152-162), V = both cameras' pose block, W nviews x 42 x 6 with rows [om_LR T_LR | intrinsics of
 *                  camera 0 | of camera 1] (always 42 rows, also under CCALIB_STEREO_FIX_INTRINSIC)
 *   reduce         Schur sums for damping lambda: Ssum = sum_i W_i V_i'^-1 W_i^T (ns x ns),
 *                  bsum = sum_i W_i V_i'^-1 ge_i
 *   time_kernels   launches each per-point kernel of camera 0 `reps` times back to back and returns
 *                  the mean duration of one launch in ms, measured with HIP events on the
 *                  solver's stream: ms[0] shared-block Jacobian kernel, ms[1] per-view-block
 *                  Jacobian kernel, ms[2] residual kernel
 */
/*
 * How the solver runs its Jacobian pass (after set_start):  out[0] per-point kernel: 0 the two vector kernels, 1 the
 * fused matrix-pipe kernel on a 16-column tile (one launch per camera), 2 the fused kernel on a 12-column tile (every
 * camera in one launch);  out[1] cameras per launch;  out[2] workgroup slots of that launch;  out[3] intrinsic columns per
 * camera;  out[4] cameras;  out[5] 1 when the stereo Schur sums take four lanes per view;  out[6] 1 when a trial step that
 * is expected to be rejected runs on a second stream beside its retry (small problems);  out[7] how many did so far.
 * ccalib_solver_time_kernels' ms[1] is the duration of ONE such launch (all out[1] cameras).
 */
int ccalib_solver_describe(ccalib_solver* s, int out[8]);
/*
 * Durations of the per-point Jacobian launches as they occur INSIDE ccalib_solver_run (between the kernels of a real
 * iteration, not back to back): switches the recording of a HIP event before and after each such launch on or off;
 * when mean_ms / launches are given, first waits for the stream and reports the mean duration and the number of
 * launches recorded since the recording was switched on.  The events cost about a microsecond each: not for timed runs.
 */
int ccalib_solver_profile_jac(ccalib_solver* s, int on, double* mean_ms, int* launches);
int ccalib_solver_num_shared(ccalib_solver* s, int* ns);
int ccalib_solver_eval_error(ccalib_solver* s, double* err);
int ccalib_solver_eval_jacobian(ccalib_solver* s, double* U, double* gs, double* err);
int ccalib_solver_fetch_blocks(ccalib_solver* s, double* V, double* W, double* ge);
int ccalib_solver_reduce(ccalib_solver* s, double lambda, double* Ssum, double* bsum);
int ccalib_solver_time_kernels(ccalib_solver* s, int reps, double ms[3]);

/*
 * The host LM driver over a caller-supplied backend: the same state machine ccalib_solver_run
 * uses, with the device passes replaced by function pointers.  This is the seam between the
 * driver and the kernels; tests drive it with a CPU backend to check sharded runs without a GPU.
 * All backend outputs must already be summed over ranks.
 *   jac       Jacobian pass at (shared, trial poses): U (ns x ns, upper triangle), gs (ns), err.
 *             lambda_hint is the damping the next reduce() will ask for, so a backend may
 *             compute those sums in the same pass
 *   reduce    Schur sums for lambda over the free shared set `fidx` (nf indices into shared):
 *             Ssum (nf x nf, symmetric) = sum_i W_i V_i'^-1 W_i^T,  bsum (nf) = sum_i W_i V_i'^-1 ge_i
 *   step_err  trial poses = committed poses - V'^-1 (ge - W^T dshared) for every view, then the
 *             residual pass at (shared_new, trial poses); returns sum |delta_e|^2,
 *             sum |committed pose|^2 and the new sum of squared residuals.  lambda_next is the
 *             damping the driver will announce to jac() if this trial point is accepted, so a backend
 *             may already run that Jacobian pass here (it yields the residual sum too) and answer
 *             the following jac() call, made at the same point, from what it kept
 *   commit    trial poses become the committed poses
 *   fixup     called before every pass with the current and previous shared vectors so that ties
 *             between parameters (fixed aspect ratio, same focal length) can be enforced; may be NULL
 */
typedef struct {
    int ns; /* shared parameter count (18 mono, 42 stereo) */
    int (*jac)(void* user, const double* shared, double lambda_hint, double* U, double* gs, double* err);
    int (*reduce)(void* user, double lambda, const int* fidx, int nf, double* Ssum, double* bsum);
    int (*step_err)(void* user, double lambda, double lambda_next, const int* fidx, int nf, const double* dshared,
                    const double* shared_new, double* dnorm2, double* pnorm2, double* err);
    int (*commit)(void* user);
    void (*fixup)(void* user, double* shared, double* prev_shared);
} ccalib_backend;
int ccalib_lm_run(const ccalib_backend* be, void* user, double* shared, const unsigned char* smask,
                  int term_type, int max_iter, double eps, int solve_method, ccalib_stats* stats);

/*
 * Per-iteration record of the last ccalib_lm_run on the calling thread (hence of the last ccalib_solver_run,
 * ccalib_calibrate_camera or ccalib_stereo_calibrate): what CvLevMarq did, iteration by iteration, so that a run can be
 * compared with another implementation of the same state machine trial for trial.  CCALIB_TRACE_FIELDS doubles per
 * finished iteration: iteration number, damping exponent (lambdaLg10) of the first trial step, exponent of the accepted
 * one, number of trial steps, residual sum at the committed point, residual sum at the accepted point, squared norm of
 * the accepted pose update, host milliseconds.  Copies min(cap, n) records; *nrecords = n (at most 1000 are kept).
 * The same fields go to the file named by the environment variable CCALIB_TRACE, one JSON line per iteration.
 */
enum { CCALIB_TRACE_FIELDS = 8 };
int ccalib_last_lm_trace(double* records, int cap, int* nrecords);

#ifdef __cplusplus
}
#endif
#endif
