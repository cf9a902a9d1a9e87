// kernels.hpp -- host-visible launch interface of the HIP passes (see kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.hpp"

namespace ccalib {

struct LmCtrl;      // lm_step.hpp: control block of an iteration run ahead of the host (null = by-value arguments)
struct LmStepArgs;  // lm_step.hpp

struct PassArgs {
    const float2* img;   // view-major image points of one camera
    const float* obj;    // object points: 3*total floats, or 3*board_np when shared_board
    const int* off;      // nviews+1 prefix offsets into img (and obj when not shared)
    const double* pose;  // nviews x 6 (r|t) evaluated by this pass
    int nviews;
    int shared_board;
    int board_np;
    int maxnp;           // largest point count of any view
    int G;               // lanes per view (power of two, 2..64)
    int staged;          // 1: the per-point kernels copy each round's points to LDS first; 0: they read global memory
    unsigned lds_bytes;  // dynamic LDS per workgroup, from pass_lds_bytes()
};

// Per-view normal-equation blocks in HBM: views are grouped in tiles of kTileViews (the views one
// workgroup of the Jacobian pass owns at a time); a tile holds its NB quantities quantity-major,
// so one workgroup writes one contiguous tile and a lane-per-view consumer reads, for a fixed
// quantity, runs of kTileViews consecutive doubles.
//   quantity q: 0..20 V upper triangle (tri6), 21..26 ge, 27 + 6*i + b = W[i][b]
constexpr int kTileViews = 4;
__host__ __device__ inline size_t blk_index(int NB, int q, int v) {
    return ((size_t)(v / kTileViews) * NB + q) * kTileViews + (v % kTileViews);
}

// Pose in the 13-double form the fused Jacobian kernel reads through the scalar cache
// (ax[3] c c1 s b1 b2 b3 b4 t[3], padded to 16 doubles per view).
constexpr int kPoseConstStride = 16;
void launch_pose_const(const double* pose, int nviews, double* pc, hipStream_t);

// LDS geometry of the fused Jacobian kernel: CH points per chunk, RS doubles between columns.
struct FusedGeom {
    int CH, RS;
    unsigned lds_bytes;
};
// LDS geometry of the fused Jacobian kernel for workgroups of `tiles` tiles (4 wavefronts each)
FusedGeom fused_geom(int maxnp, bool shared_board, int board_np, bool const_cols = true, int tiles = 1);
constexpr unsigned kLdsPerCU = 160 * 1024;

struct DeltaC {
    double v[I_NMODEL];  // step of the kernel's intrinsic columns (0 for columns not optimised)
};

struct ModelOps {
    unsigned mask;
    bool aspect, rational;
    int NC, NRED_U, NBLK;
    void (*jac_shared)(const PassArgs&, const CamParams&, int grid, double* part, hipStream_t);
    void (*jac_view)(const PassArgs&, const CamParams&, int grid, double* blk, int npad, hipStream_t);
    // part2 != null: the same sums for a second damping lambda2 in the same launch (second half of the grid)
    void (*schur)(const double* blk, int npad, int nviews, double lambda, double* part, hipStream_t, const LmCtrl* ctrl,
                  double lambda2, double* part2);
    void (*backsub)(const double* blk, int npad, int nviews, double lambda, const DeltaC&, const double* pose_prev,
                    double* pose_cur, double* part, double* pc, hipStream_t, const LmCtrl* ctrl);
    // whole Jacobian pass in one kernel (shared block partials: 4 * grid rows of NRED_U); null when the
    // column set does not fit the 16-wide matrix tile
    void (*jac_fused)(const PassArgs&, const CamParams&, const double* pc, int grid, double* blk, double* part,
                      hipStream_t, const LmCtrl* ctrl, int cam);
    int col(int i) const {  // i-th produced column -> intrinsic index
        int seen = 0;
        for (int q = 0; q < I_NMODEL; q++)
            if (mask >> q & 1u) {
                if (seen == i) return q;
                seen++;
            }
        return -1;
    }
};

// The fused pass on a 12-column tile (fused12.hip): column sets of at most six intrinsics with cx and cy among them,
// every camera of the solver in one launch (grid.y = camera).  All cameras share the view table, the object points and
// the column set; what differs per camera travels in Fused12Cam.
struct Fused12Cam {
    CamParams cp;
    const float2* img;   // the camera's image points
    const double* pc;    // its pose records (nviews x kPoseConstStride)
    double* blk;         // its per-view blocks
    double* part;        // its shared-block partial sums: `grid` rows of NRED_U
};
struct Fused12Cams {
    Fused12Cam c[2];
};
struct Fused12Geom {
    int CH, RS;
    unsigned lds_bytes;
};
Fused12Geom fused12_geom(int maxnp, bool shared_board, int board_np, int tiles);
bool has_jac_fused12(unsigned mask, bool aspect, bool rational);
// false: the column set is not built for this tile, or a view does not fit its LDS geometry
bool launch_jac_fused12(unsigned mask, bool aspect, bool rational, const PassArgs& a, const Fused12Cams& cams, int ncams, int grid,
                        hipStream_t st);

// Smallest built-in column set that covers `need_mask` (bits over the 12 modelled intrinsics).
const ModelOps* find_model(unsigned need_mask, bool aspect, bool need_rational);
// sum of squared residuals from pose records (pc: nviews x kPoseConstStride), one wavefront per view
void launch_err_wave(const PassArgs&, const CamParams&, bool rational, const double* pc, int grid, double* part,
                     hipStream_t);
// The same for both cameras of a stereo rig in one launch (same board layout, same distortion model).
struct ErrPair {
    PassArgs a[2];
    CamParams cp[2];
    const double* pc[2];
    double* part[2];
};
void launch_err_wave_pair(const ErrPair& e, bool rational, int grid, hipStream_t st);
// projected pixels (as float, optional) and the squared reprojection error of every view; a non-null `bad`
// points at a device word which, when non-zero, makes the kernel return at once
void launch_reproj(const PassArgs&, const CamParams&, bool rational, int grid, float2* proj, double* view_err,
                   hipStream_t, const int* bad = nullptr);
// One launch reduces the partial-result tables of a whole pass (see k_colsum).
struct ColsumSeg {
    const double* part;
    int nrows, n, out_ofs;
};
constexpr int kMaxColsumSegs = 8;
struct ColsumArgs {
    ColsumSeg seg[kMaxColsumSegs];
    int nseg;
};
// The sums go to `out` (device memory).  With `flag`, the workgroup that finishes last copies out[0..n_out) to
// host_out and raises the flag.  With `step`, it instead runs the LM step prediction on the sums (step->red == out)
// and publishes through step->hred / step->flag.
void launch_colsum(const ColsumArgs&, double* out, unsigned* counter, double* host_out, int n_out, unsigned long long* flag,
                   unsigned long long seq, hipStream_t, const LmCtrl* ctrl = nullptr, const LmStepArgs* step = nullptr,
                   double2* pairs = nullptr);  // pairs (pinned, 16 bytes per sum): every workgroup sends {sum, seq} itself
void launch_publish(const double* src, int n, double* dst, unsigned long long* flag, unsigned long long seq, hipStream_t);
int pass_block_size();
// dynamic LDS one workgroup of the per-point kernels needs
size_t pass_lds_bytes(int G, int maxnp, bool shared_board, int board_np);

}  // namespace ccalib
