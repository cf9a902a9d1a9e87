// solver_internal.hpp -- what the whole-call entry points (api.cpp) use of the solver beyond the C ABI.
#pragma once
#include "ccalib.h"

namespace ccalib {

// The points of all views on the host, either as one flat array (view after view) or as one pointer per view -- the
// shape of vector<vector<Point3f>> / vector<vector<Point2f>>, which the reference hands to OpenCV
// (monocular_chessboard.cpp:295-305, :249): view i's points are contiguous, the views are not.
struct HostViews {
    const float* flat;
    const float* const* ptrs;
};

// ccalib_solver_create over either shape.  With shared_board the object points are view 0's only.
int solver_create_views(ccalib_solver** out, int device, int ncams, int nviews, const int* npts, HostViews obj, int shared_board,
                        HostViews img1, HostViews img2);

// ccalib_solver_get's poses as two nviews x 3 arrays (rotation vectors, translations).
int solver_get_poses_split(ccalib_solver* s, double* rvecs, double* tvecs);

}  // namespace ccalib
