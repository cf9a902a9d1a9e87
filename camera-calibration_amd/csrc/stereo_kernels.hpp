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

// ---- layout of the four-lanes-per-view sums kernel (shared by the kernel and the host-side index table) ----
// A view's rows (6 relative-pose rows, the gradient, nc rows per camera) sit in "slots": slot s of quad lane a
// carries one row or none.  Slots 0,1: relative-pose rows 4s+a (row 6 = the gradient, row 7 = none); then
// ceil(nc/4) camera-0 slots and as many camera-1 slots, row 4i+a of the camera.
constexpr int kQuadChunk = 16;  // products summed over a wavefront's views per trip through LDS
__host__ __device__ constexpr int quad_sc(int nc) { return (nc + 3) / 4; }
__host__ __device__ constexpr int quad_nslot(int nc) { return 2 + 2 * quad_sc(nc); }
// index (0..kc-1, kc = the gradient) of the row in slot s of lane a among the reduced system's columns; -1: none
__host__ __device__ constexpr int quad_row(int nc, int s, int a) {
    const int sc = quad_sc(nc), kc = 6 + 2 * nc;
    if (s < 2) return 4 * s + a < 6 ? 4 * s + a : 4 * s + a == 6 ? kc : -1;
    if (s < 2 + sc) return 4 * (s - 2) + a < nc ? 6 + 4 * (s - 2) + a : -1;
    return 4 * (s - 2 - sc) + a < nc ? 6 + nc + 4 * (s - 2 - sc) + a : -1;
}
// number of relative-pose values a lane pushes (6 per relative-pose slot and per camera-1 slot)
__host__ __device__ constexpr int quad_nlr(int nc) { return 6 * (2 + quad_sc(nc)); }
// order of a lane's Gram products: t = how many places round the quad the partner lane is (0: the lane itself);
// s1 = own slot, s2 = partner slot (t = 0, 2: s1 <= s2)
__host__ __device__ constexpr int quad_gram_index(int nc, int t, int s1, int s2) {
    const int n = quad_nslot(nc), t0 = n * (n + 1) / 2;
    return t == 0 ? s1 * n - s1 * (s1 - 1) / 2 + (s2 - s1) : t == 1 ? t0 + s2 * n + s1 : t0 + n * n + s2 * (s2 + 1) / 2 + s1;
}
__host__ __device__ constexpr int quad_ngram(int nc) { return quad_nslot(nc) * (quad_nslot(nc) + 1) + quad_nslot(nc) * quad_nslot(nc); }

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
// One pass over the blocks of a Jacobian pass, four lanes per view: the Schur sums for `lambda` (layout as
// launch_stereo_view_sums) and, with part_lr != null, the relative-pose sums (layout as launch_stereo_lr_sums), one row
// of partial sums per workgroup; `rows` workgroups (stereo_sums_rows()).  `tab` = the device copy of
// stereo_sums_table(nc).
int stereo_sums_rows(int nviews);
int stereo_sums_table(int nc, short* out);  // fills out (when not null), returns the entry count; 0: nc not built in
bool launch_stereo_sums(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                        int nviews, double lambda, const short* tab, int rows, double* part, double* part_lr, hipStream_t st,
                        double lambda2 = 0., double* part2 = nullptr);
// The Schur sums with one lane per view and one row of partial sums per 64 views: upper triangle of
// sum W V'^-1 W^T and sum W V'^-1 ge over the columns [om_LR T_LR | camera 0 | camera 1]; `dst` = the device copy
// of stereo_view_table(nc).
int stereo_view_table(int nc, short* out);
bool launch_stereo_view_sums(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                             int nviews, double lambda, const short* dst, double* part, hipStream_t st, double lambda2 = 0.,
                             double* part2 = nullptr);
// (both sums launchers: with part2 the same sums for a second damping, lambda2, go to part2 -- the grid's second half,
// so that at small problems the two run side by side instead of one launch after the other)
// A second trial step computed from the same read of the blocks (another damping, another step): its camera-0 poses,
// their records (may be null) and its two step norms per 64 views go to buffers of its own.
struct StereoSecondTrial {
    double lambda;
    StereoDelta d;
    double* pose;
    double* pc;
    double* part;
};
bool launch_stereo_backsub(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr,
                           int npad, int nviews, double lambda, const StereoDelta& d, const double* pose_prev,
                           double* pose_cur, double* part, double* pc,  // pc: trial poses as records, or null
                           const StereoLR& lr_new, double* pose3, double* pc3,  // pose3 != null: camera 1's trial poses
                           hipStream_t st,                                        // for the relative pose lr_new as well
                           const StereoSecondTrial* second = nullptr,
                           double* cmp_next = nullptr);  // with pose3: the composition derivatives at the trial point too
                                                         // (what launch_compose writes to `cmp`), for the Jacobian pass there

// The assembled stereo view systems of a Jacobian pass, for inspection (tests): V nviews x 36, ge nviews x 6,
// W nviews x (6 + 2 nc) x 6 in kernel column order [om_LR T_LR | camera 0 | camera 1], all device pointers.
bool launch_stereo_fetch(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                         int nviews, double* V, double* W, double* ge, hipStream_t st);

}  // namespace ccalib
