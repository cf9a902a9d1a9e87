// lm_step.hpp -- the device half of running one LM iteration ahead of the host.
//
// CvLevMarq (OpenCV 4.1.0 compat_ptsetreg.cpp; reached from monocular_circle.cpp:93-94 and
// binocular_chessboard.cpp:152-162) is a host state machine: after every pass over the points the host looks at
// the sums, solves the small reduced system and only then can the next pass start.  That turn-around (flag over
// PCIe, host arithmetic, a kernel launch, its dispatch) leaves the GPU idle for ~20 us of a ~190 us iteration.
// Here the GPU does not wait: the last kernel of a Jacobian pass (k_lm_step) repeats, on the device, exactly the
// arithmetic the host driver is about to do -- acceptance test, reduced system, Cholesky solve, new shared
// parameters -- and writes the result into a control block in device memory; the kernels of the NEXT iteration
// are already in the stream and take their step, camera parameters and "go" word from that block.  The host still
// runs its state machine on the published sums and stays the authority: it uses what the GPU ran ahead only when
// the step it computes itself is bit for bit the one the device predicted (same IEEE operations in the same
// order on both sides, contraction off), so results do not depend on whether the run-ahead was taken.
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "stereo_kernels.hpp"

namespace ccalib {

constexpr int kMaxShared = 42;   // 6 + 18 + 18
constexpr int kCtrlCols = 32;    // kernel columns a step can carry (>= kStereoMaxCols, >= I_NMODEL)

// What the kernels of a run-ahead iteration read instead of by-value arguments.
struct LmCtrl {
    int go;                      // 0: every kernel of the iteration returns at once
    int pad;
    double prev_err;             // residual sum at the point the iteration steps from
    double shared[kMaxShared];   // the trial point (shared parameter vector) the iteration evaluates
    double dc[kCtrlCols];        // its step in kernel-column order (DeltaC / StereoDelta layout)
    CamParams cp[2];             // the cameras at the trial point
    StereoLR lr;                 // relative pose at the trial point
};

// Where the driver's reduced system sits among the sums of a Jacobian pass (indices relative to the pass's
// base offset in the reduction buffer; -1 = structurally zero).  Built on the host, kept in device memory.
struct LmMaps {
    int nf, ns, ncams, nerr;
    int fidx[kMaxShared];               // free shared parameters, ascending (the driver's fidx)
    int dc_slot[kMaxShared];            // kernel column of each free parameter
    int g_idx[kMaxShared];              // gradient entry of each free parameter
    int bs_idx[kMaxShared];             // Schur right-hand-side entry
    int err_idx[2];                     // residual sums, one per camera
    int u_idx[kMaxShared * kMaxShared]; // nf x nf: shared-block entry (upper triangle, mirrored)
    int s_idx[kMaxShared * kMaxShared]; // nf x nf: Schur-sum entry
};

struct LmStepArgs {
    const double* red;                  // reduced sums of the pass (device memory), n of them
    int n;
    double* hred;                       // pinned host copy; the prediction block follows at kPredOfs
    unsigned long long* flag;           // pinned; receives seq when everything has landed
    unsigned long long seq;
    const LmCtrl* cin;                  // control block of the iteration this pass belongs to (null: by-value below)
    LmCtrl* cout;                       // control block of the next iteration (null: publish only)
    const LmMaps* maps;
    int base;                           // offset of the Jacobian sums inside red
    double lambda;                      // damping of the step to predict
    double prev_err;                    // cin == null: residual sum the trial is compared with (+inf: no test)
    double shared[kMaxShared];          // cin == null: the trial point
    double aspect[2];
    int same_focal, fix_intrinsic;
    CamParams cp0[2];                   // the cameras as the host holds them (fields not in the shared vector stay)
};

constexpr int kPredOfs = 3072;          // prediction block inside the pinned copy: go, delta[nf], shared_next[ns]

void launch_lm_step(const LmStepArgs& a, hipStream_t st);

}  // namespace ccalib
