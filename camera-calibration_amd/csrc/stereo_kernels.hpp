// stereo_kernels.hpp -- per-view kernels that tie the two cameras of a stereo rig together.
#pragma once
#include <hip/hip_runtime.h>

namespace ccalib {

constexpr int kCmpRows = 27;        // per view: dr3/dr1 (9), dr3/dr2 (9), dt3/dr2 (9), SoA
constexpr int kStereoMaxCols = 30;  // 6 relative-pose columns + two cameras of up to 12

struct StereoLR {
    double v[6];  // om_LR | T_LR
};
struct StereoDelta {
    double v[kStereoMaxCols];  // step of [om_LR T_LR | camera-0 columns | camera-1 columns]
};

// pose3 = (om_LR,T_LR) o pose1 for every view; cmp (may be null) receives the derivatives, pc3 (may be null)
// pose3 as the 16-double records the fused Jacobian kernel reads.
void launch_compose(const double* pose1, const StereoLR& lr, int nviews, int npad, double* pose3, double* cmp, double* pc3,
                    hipStream_t st);
// The three per-view kernels below need the relative pose the Jacobian pass was taken at (`lr`).
// Both cameras use the same column set (nc columns each).  They return false for a column count
// that is not built in.
// Per 64-view workgroup: sums that enter the shared block through the relative pose:
// B^T V3 B (21, upper), B^T ge3 (6), W3 B (nc x 6).
bool launch_stereo_lr_sums(int nc, const double* blk1, const double* cmp, const StereoLR& lr, int npad, int nviews,
                           double* part, hipStream_t st);
// Per 64-view workgroup: upper triangle of sum W V'^-1 W^T and sum W V'^-1 ge over the columns
// [om_LR T_LR | camera 0 | camera 1].
bool launch_stereo_schur(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                         int nviews, double lambda, double* part, hipStream_t st);
bool launch_stereo_backsub(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr,
                           int npad, int nviews, double lambda, const StereoDelta& d, const double* pose_prev,
                           double* pose_cur, double* part, double* pc, hipStream_t st);  // pc: trial poses as records, or null

}  // namespace ccalib
