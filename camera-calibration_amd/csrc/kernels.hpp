// kernels.hpp -- host-visible launch interface of the HIP passes (see kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.hpp"

namespace ccalib {

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
    unsigned lds_bytes;  // dynamic LDS per workgroup, from pass_lds_bytes()
};

struct DeltaC {
    double v[I_NMODEL];  // step of the kernel's intrinsic columns (0 for columns not optimised)
};

struct ModelOps {
    unsigned mask;
    bool aspect, rational;
    int NC, NRED_U, NBLK;
    void (*jac_shared)(const PassArgs&, const CamParams&, int grid, double* part, hipStream_t);
    void (*jac_view)(const PassArgs&, const CamParams&, int grid, double* blk, int npad, hipStream_t);
    void (*schur)(const double* blk, int npad, int nviews, double lambda, double* part, hipStream_t);
    void (*backsub)(const double* blk, int npad, int nviews, double lambda, const DeltaC&, const double* pose_prev,
                    double* pose_cur, double* part, hipStream_t);
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

// Smallest built-in column set that covers `need_mask` (bits over the 12 modelled intrinsics).
const ModelOps* find_model(unsigned need_mask, bool aspect, bool need_rational);
void launch_err(const PassArgs&, const CamParams&, bool rational, int grid, double* part, hipStream_t);
// projected pixels (as float, optional) and the squared reprojection error of every view
void launch_reproj(const PassArgs&, const CamParams&, bool rational, int grid, float2* proj, double* view_err,
                   hipStream_t);
void launch_colsum(const double* part, int nrows, int n, double* out, hipStream_t);
int pass_block_size();
// dynamic LDS one workgroup of the per-point kernels needs
size_t pass_lds_bytes(int G, int maxnp, bool shared_board, int board_np);

}  // namespace ccalib
