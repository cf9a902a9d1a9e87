// host_init.hpp -- starting values for the LM loop (host side, threaded over views).
#pragma once

namespace ccalib {

// cvRodrigues2 on the host (both directions, no Jacobians needed outside the pose refinement).
void host_rodrigues_v2m(const double r[3], double R[9], double* dRdr27);
void host_rodrigues_m2v(const double R[9], double r[3]);

// findHomography(method 0): normalised DLT + 10 LM iterations.  src/dst: n x 2.
bool host_find_homography(int n, const double* src, const double* dst, double H[9]);

// cvInitIntrinsicParams2D.  obj: 3 floats per point, img: 2 floats per point (view after view).
void host_init_intrinsics(int nviews, const int* npts, const float* obj, const float* img, int width, int height,
                          double aspect, double A[9]);

// cvFindExtrinsicCameraParams2 for one planar view.  Returns false for a non-planar point set.
bool host_find_extrinsic(int n, const float* obj, const float* img, const double A[9], const double k[14], double r[3],
                         double t[3]);

// All views at once (OpenMP); rt receives nviews x 6.  stride_obj = 0 for one shared board.
void host_init_poses(int nviews, const int* npts, const float* obj, bool shared_board, const float* img,
                     const double A[9], const double k[14], double* rt);

}  // namespace ccalib
