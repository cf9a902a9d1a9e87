// lm_step.hip -- k_lm_step: publishes the sums of a Jacobian pass to the host and predicts, on the device, the
// step the host LM driver is about to compute from them (see lm_step.hpp, lm_step_body.hpp).  One wavefront.
// Used where the sums pass through an in-stream all-reduce first; single-GPU passes run the same code in the last
// workgroup of the column-sum kernel instead.
#include "lm_step.hpp"

#include "lm_step_body.hpp"

namespace ccalib {

namespace {

__global__ __launch_bounds__(64) void k_lm_step(LmStepArgs a) {
    __shared__ LmStepLds w;
    if (a.cin && !a.cin->go) return;  // this iteration was called off by the one before: nothing to publish
    lm_step_body(a, w, threadIdx.x);
}

}  // namespace

void launch_lm_step(const LmStepArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_lm_step, dim3(1), dim3(64), 0, st, a); }

}  // namespace ccalib
