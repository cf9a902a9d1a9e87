// init_kernels.hpp -- host-visible launch interface of the initialisation kernels (see init_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.hpp"

namespace ccalib {

struct InitArgs {
    const float2* img;   // view-major image points of one camera
    const float* obj;    // object points: 3*total floats, or 3*board_np when shared_board
    const int* off;      // nviews+1 prefix offsets into img (and obj when not shared)
    int nviews;
    int shared_board;
    int board_np;
    int maxnp;           // largest point count of any view
    int PS;              // LDS row length in points (odd), set by init_geometry
    int VPB;             // views per workgroup (a few lanes each), set by init_geometry
    unsigned lds_bytes;  // dynamic LDS per workgroup, set by init_geometry
    int cut;             // pose refinement: iterations after which a view still running is parked, set by init_geometry
};

// Views whose refinement was cut short by the first pose kernel, finished by the second one.
constexpr int kParkedDoubles = 9;  // pose (6), residual norm, damping exponent, iteration count
struct Parked {
    int* count;     // number of parked views (device)
    int* view;      // nviews entries
    double* state;  // nviews x kParkedDoubles
};

// Chooses PS, VPB and lds_bytes for the problem in `a`; with_src reserves room for the homography's
// source points (the pose kernel).  Returns false when not even one view fits the LDS budget.
bool init_geometry(InitArgs& a, bool with_src);

// cvFindExtrinsicCameraParams2 for every view: pose receives nviews x 6 (r|t), status 0 = ok,
// 1 = the view's object points are not planar (pose left at zero).
void launch_init_pose(const InitArgs& a, const CamParams& cp, bool rational, double* pose, int* status, const Parked& parked,
                      hipStream_t);

// cvInitIntrinsicParams2D: per-workgroup sums of the 2-unknown normal equations (5 doubles each);
// returns the number of workgroups.
int launch_init_intrinsics(const InitArgs& a, double cx, double cy, double* part, hipStream_t);

// Relative pose (om | T) of camera 1 w.r.t. camera 0 per view, as six rows of npad doubles.
void launch_relative_pose(const double* pose0, const double* pose1, int nviews, int npad, double* out, hipStream_t);

}  // namespace ccalib
