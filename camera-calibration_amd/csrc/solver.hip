// solver.hip -- device-resident calibration problem and the HIP backend of the host LM driver.
//
// One ccalib_solver owns the views of one rank: image points (as given: view-major float2),
// object points (or one shared board), poses, and the per-view normal-equation blocks, all in
// HBM for the life of the solve.  Each LM pass is a short, fixed sequence of kernel launches on
// one stream followed by one small device-to-host copy (and, in multi-GPU runs, one all-reduce
// of the same small buffer).  The passes replace the view/point loops of
// cvCalibrateCamera2Internal and cvStereoCalibrateImpl (OpenCV 4.1.0 calibration.cpp; reference
// call sites monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317,
// binocular_chessboard.cpp:152-162, binocular_circle.cpp:153-163).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <mutex>
#include <vector>

#include <omp.h>

#include "ccalib.h"
#include "host_util.hpp"
#include "init_kernels.hpp"
#include "kernels.hpp"
#include "lm_step.hpp"
#include "solver_internal.hpp"
#include "stereo_kernels.hpp"

using namespace ccalib;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_error(CCALIB_ERR_HIP, "%s failed: %s (%s:%d)%s", #expr, hipGetErrorString(e_), \
                             __FILE__, __LINE__, hip_runtime_note());                             \
    } while (0)
#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t e_ = (expr);                                                                   \
        if (e_ != ncclSuccess)                                                                      \
            return set_error(CCALIB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(e_), \
                             __FILE__, __LINE__);                                                   \
    } while (0)

// A pass whose sums are on their way to the host: which slot, and the value its flag will take.
struct Pending {
    int slot = 0;
    unsigned long long seq = 0;
    hipStream_t stream = nullptr;  // the stream the reduction was submitted to
    // direct publication (every column-sum workgroup writes {sum, seq} to pinned memory): the ranges of sums to wait for
    bool direct = false;
    int nrange = 0;
    struct Range {
        int ofs, n;
    } range[kMaxColsumSegs];
};

struct ccalib_solver {
    int device = 0, ncams = 1, nviews = 0, total = 0, maxnp = 0;
    bool shared_board = false;
    int board_np = 0;
    int G = 4, nblk64 = 0, npad = 0;
    bool staged = true;  // per-point kernels stage a round's points in LDS (false: views too long for that)
    int grid_view = 0;  // per-view kernels: every wavefront does one round
    int grid_red = 0;   // reducing kernels: fewer workgroups, several rounds each, one reduction at the end
    int grid_fused = 0; // fused Jacobian kernel: workgroups striding over tiles of kTileViews views
    int grid_fused12 = 0;  // the same for the 12-column kernel, which takes every camera in one launch
    bool use_tile12 = true;  // CCALIB_NO_TILE12=1: column sets that fit the 12-column tile use the 16-column kernel
    int grid_err = 0;   // residual kernel: one wavefront per view, striding
    bool use_fused = true;
    unsigned lds_bytes = 0;
    hipStream_t stream = nullptr;
    // A second stream for a trial step that is expected to be rejected while the retry prepared behind it is expected to be
    // accepted: the retry and the Jacobian pass run ahead at its point go to `stream` first, the doomed trial's
    // back-substitution, residual pass and reduction run beside them here (be_step_err).  It has camera-1 pose buffers
    // and a completion counter of its own; everything else the two touch is distinct already.
    hipStream_t aux = nullptr;
    bool aux_enabled = true;  // chosen by problem size at creation; CCALIB_AUX_STREAM=0|1 decides otherwise
    double *d_pose3_alt = nullptr, *d_pc3_alt = nullptr;
    long long n_aside = 0;
    // Device memory comes from two allocations carved into pieces: `arena` (everything whose size the view table fixes:
    // points, poses, records, reduction buffers, initialisation state) and `arena2` (what depends on the column set:
    // the per-view blocks and the partial-result arenas); one pinned block holds both slots' host-visible sums and
    // flags.  A solver is created and destroyed inside every cv::calibrateCamera-shaped call, and each hipMalloc /
    // hipFree costs 0.05-0.3 ms.
    char* arena = nullptr;
    char* arena2 = nullptr;
    size_t arena_bytes = 0, arena2_bytes = 0, pinned_bytes = 0;
    char* pinned = nullptr;
    float* d_obj = nullptr;
    int* d_off = nullptr;
    float2* d_img[2] = {nullptr, nullptr};
    double *d_pose_prev = nullptr, *d_pose_cur = nullptr;  // camera-0 poses, nviews x 6
    double* d_pose3 = nullptr;                              // stereo: composed camera-1 poses
    // The per-view blocks exist twice: `cur` holds the Jacobian the driver is stepping from, the other
    // one receives the pass run ahead at a trial point (see be_step_err).
    int cur = 0;
    short* d_sums_tab = nullptr;                            // stereo: where the products of the sums kernel go
    short* d_view_tab = nullptr;                            // stereo: the same for the one-lane-per-view kernel
    bool sums_quad = false;                                 // stereo: four lanes per view in the sums kernel
    int sums_rows = 0;                                      // stereo: partial-sum rows (workgroups) of the sums kernel
    double* d_cmp[2] = {nullptr, nullptr};                  // stereo: per-view composition Jacobians (SoA), per buffer
    double* d_blk[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [camera][buffer]
    double* d_pc = nullptr;  // nviews x kPoseConstStride: poses in the form the fused Jacobian kernel reads
    double* d_pc3 = nullptr; // stereo: the same for the composed camera-1 poses (written by the composition kernel)
    // initialisation (ccalib_solver_init_*): per-view status words, per-camera "poses are on the device" marks
    int* d_status = nullptr;
    int* d_parked_view = nullptr;      // views the first pose kernel parked (count in entry nviews)
    double* d_parked_state = nullptr;
    double* d_init_part = nullptr;     // partial sums of the intrinsic initialisation (5 per workgroup)
    double* d_pose_init1 = nullptr;  // stereo: camera-1 planar poses until the relative pose has been taken
    bool poses_ready[2] = {false, false};
    bool pc_fresh = false;   // d_pc matches d_pose_cur
    bool pc3_fresh = false;  // d_pc3 matches d_pose3
    // Partial results of the kernels of one pass: a bump-allocated arena, reduced by ONE column-sum
    // launch per pass whose output goes straight to pinned host memory; the host polls a flag there.
    // Arena, reduction buffers and flag exist twice ("slots", used in turn): with an iteration run ahead the next
    // pass is enqueued before the host has read the sums of the one before.
    double* d_part[2] = {nullptr, nullptr};
    size_t part_cap = 0, part_used = 0;
    ColsumArgs segs;
    int slot = 0;                                              // slot the next pass takes
    double* d_red[2] = {nullptr, nullptr};
    double* h_red[2] = {nullptr, nullptr};                     // pinned, device-writable
    unsigned long long* h_flag[2] = {nullptr, nullptr};        // pinned
    double2* h_pair[2] = {nullptr, nullptr};                   // pinned: {sum, sequence number} per sum, written by k_colsum
    bool direct_sums = true;                                   // CCALIB_DIRECT_SUMS=0: publish through a flag
    unsigned* d_counter = nullptr;
    unsigned long long seq = 0;
    double wait_limit_s = 120.;         // longest wait for one pass's sums (CCALIB_WAIT_LIMIT_S)
    double t_wait = 0, t_enqueue = 0;  // host seconds spent polling / inside backend calls before polling
    double t_think = 0, t_launch1 = 0;  // ... between the end of a wait and the next trial's first launch / inside that launch
    std::chrono::steady_clock::time_point t_wait_end;
    std::chrono::steady_clock::time_point t_enter;
    static constexpr int kRedCap = 4096;

    // parameters
    bool started = false;
    double intr[2][18];
    unsigned char imask[2][18];
    double aspect[2] = {0, 0};
    bool same_focal = false;
    bool fix_intrinsic = false;  // stereo, CALIB_FIX_INTRINSIC: the parameter vector holds om_LR, T_LR and the poses only
    double lr[6] = {0, 0, 0, 0, 0, 0};
    StereoLR lr_jac[2];  // relative pose the per-view blocks of each buffer were computed at
    const ModelOps* ops[2] = {nullptr, nullptr};
    bool rational[2] = {false, false};
    double last_err = 0;

    // Schur sums cached by the Jacobian pass for the damping the driver announced
    bool have_cached = false;
    double cached_lambda = 0;
    std::vector<double> cachedS;  // kernel-column space, see reduce()
    // A second set for the damping one decade up, taken in the same pass when the run has settled into
    // "first trial rejected, retry accepted" (the reference's stereoCalibrate flag set does): the retry's
    // reduce() is then answered without a pass of its own.
    bool have_cached2 = false;
    double cached_lambda2 = 0;
    std::vector<double> cachedS2;
    bool spec_have2 = false;
    double spec_lambda2 = 0;
    std::vector<double> spec_S2;

    // Jacobian pass run ahead at the last trial point (valid until the next pass or step)
    bool speculate = true;
    bool spec_valid = false;
    double spec_shared[42];
    std::vector<double> spec_U, spec_gs, spec_S;
    double spec_err = 0, spec_lambda = 0;
    int steps_since_jac = 0;
    int prev_trials = 1;              // trials the iteration before this one took (1: its first trial was accepted)
    // Whether the last trial step taken at each damping level (10^-16 .. 10^16) was accepted: a step is run
    // ahead only at a level whose last trial was accepted, which follows the accept/reject pattern an LM
    // run settles into (e.g. every first trial at 1e-3 rejected, every retry at 1e-2 accepted).
    bool level_accepted[33];
    int last_level = 0;

    // One iteration run ahead of the host (lm_step.hpp): the kernels of the next trial step are enqueued before the
    // host has seen the sums of the current one and take their parameters from a control block the device writes.
    int split_publish = -1;          // CCALIB_SPLIT_PUBLISH=0/1 forces where the sums are published from (-1: by size)
    // A retry launched behind a trial that the damping history says will be rejected (be_step_err): the kernels of
    // the step the driver is expected to ask for next are already in the stream when it asks.
    struct Trial {
        Pending p;
        int n = 0;
        bool ahead = false, predicted = false;
        double lam2 = 0.;
    };
    struct Retry {
        bool inflight = false;
        double lambda = 0, lambda_next = 0;
        int nf = 0, cur = 0;
        double dsh[42], shared[42];
        Trial t;
    } retry;
    bool retry_enabled = true;            // CCALIB_NO_RETRY_AHEAD=1 turns it off
    bool dual_backsub = false;            // CCALIB_DUAL_BACKSUB=1: the retry's back-substitution rides in the first trial's kernel
    bool fold_compose = true;             // stereo: a trial followed by a run-ahead Jacobian pass composes camera 1's poses,
                                          // derivatives included, in its own kernel (CCALIB_FOLD_COMPOSE=0: separate launch)
    double* d_pose_retry = nullptr;       // trial poses of such a retry (they must not touch the first trial's)
    double* d_pc_retry = nullptr;         // ... and their records
    std::vector<double> com_U, com_gs;    // shared block and gradient of the committed Jacobian, as the driver holds them
    double com_shared[42];                // the committed point
    bool com_valid = false;
    bool chain_enabled = false;      // CCALIB_CHAIN=1 turns running ahead of the host on (measured: no gain)
    LmCtrl* d_ctrl = nullptr;        // [2], indexed by the slot of the iteration that reads it
    LmMaps* d_maps = nullptr;
    LmMaps h_maps;
    std::vector<int> Uidx, gidx;     // shared block / gradient entry -> index among a Jacobian pass's sums
    int err_idx[2] = {0, 0};
    struct Chain {
        bool inflight = false;       // enqueued and not yet claimed by the driver's next step
        bool predicts = false;       // its k_lm_step predicts the iteration after it
        double lambda = 0, lambda_next = 0;
        int blk_base = 0;            // block buffer it steps from (its Jacobian goes to the other one)
        double *pose_base = nullptr, *pose_trial = nullptr;
        Pending pend;
    } chain;
    bool pred_go = false;            // what the last Jacobian pass's k_lm_step predicted for the next step
    double pred_delta[kMaxShared], pred_shared[kMaxShared];
    int run_solve_method = 1, run_max_iter = 0, commits = 0;
    bool run_first_jac = false;
    double drv_err = 0, committed_err = 0;  // residual sums as the driver holds them: latest trial / committed point
    long long n_ahead = 0, n_ahead_used = 0, n_ahead_dropped = 0;
    long long n_jac_exec = 0, n_err_exec = 0;  // passes over all points that were actually enqueued (ccalib_solver_counters)
    int slot_after() const { return slot ^ 1; }

    // In-loop timing of the per-point Jacobian launches (ccalib_solver_profile_jac): an event before and after each
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    bool prof_on = false;

    // multi-GPU
    ncclComm_t comm = nullptr;
    ccalib_allreduce_fn cb = nullptr;
    void* cb_user = nullptr;

    int ns() const { return ncams == 1 ? 18 : fix_intrinsic ? 6 : 42; }
    int kcols() const { return ncams == 1 ? ops[0]->NC : 6 + ops[0]->NC + ops[1]->NC; }
    // kernel column -> index in the shared vector
    int kcol_to_shared(int k) const {
        if (ncams == 1) return ops[0]->col(k);
        if (k < 6) return k;
        k -= 6;
        if (k < ops[0]->NC) return 6 + ops[0]->col(k);
        return 6 + 18 + ops[1]->col(k - ops[0]->NC);
    }
};

namespace {

CamParams cam_params(const double* intr18, double aspect) {
    CamParams cp;
    cp.fx = intr18[0], cp.fy = intr18[1], cp.cx = intr18[2], cp.cy = intr18[3];
    for (int i = 0; i < 8; i++) cp.k[i] = intr18[4 + i];
    cp.aspect = aspect;
    return cp;
}

PassArgs pass_args(const ccalib_solver* s, int cam, const double* pose) {
    PassArgs a;
    a.img = s->d_img[cam];
    a.obj = s->d_obj;
    a.off = s->d_off;
    a.pose = pose;
    a.nviews = s->nviews;
    a.shared_board = s->shared_board;
    a.board_np = s->board_np;
    a.maxnp = s->maxnp;
    a.lds_bytes = s->lds_bytes;
    a.G = s->G;
    a.staged = s->staged ? 1 : 0;
    return a;
}

// Arena for the partial results of a pass.  There are two of everything a pass publishes through -- arena,
// device and pinned reduction buffers, flag -- used in turn ("slots"), because with an iteration run ahead the
// next pass is enqueued while the host has not yet read the sums of the one before.
double* part_take(ccalib_solver* s, size_t n) {
    n = (n + 31) / 32 * 32;
    if (s->part_used + n > s->part_cap) return nullptr;
    double* p = s->d_part[s->slot] + s->part_used;
    s->part_used += n;
    return p;
}
#define PART_TAKE(ptr, s, n)                                                                   \
    double* ptr = part_take(s, n);                                                             \
    if (!ptr) return set_error(CCALIB_ERR_STATE, "partial-result arena exhausted (%s:%d)", __FILE__, __LINE__)

int queue_colsum(ccalib_solver* s, const double* part, int nrows, int n, int out_ofs) {
    if (s->segs.nseg >= kMaxColsumSegs) return set_error(CCALIB_ERR_STATE, "too many reductions in one pass");
    s->segs.seg[s->segs.nseg++] = ColsumSeg{part, nrows, n, out_ofs};
    return CCALIB_OK;
}

// Reduces everything queued in this pass into n sums, over ranks too (if any), and sends them to the pinned
// buffer of the current slot.  Plain passes: the last workgroup of the column-sum kernel copies the sums to pinned
// host memory and raises the flag (RCCL runs: the sums are all-reduced in-stream and then published).  Jacobian
// passes of a run that may run ahead (`step` given): the sums stay in device memory, and k_lm_step publishes them
// together with its prediction of the next step.  `ctrl`: control block of a run-ahead iteration (its kernels
// return at once when the iteration was called off), else null.
int submit_reduction(ccalib_solver* s, int n, const LmCtrl* ctrl, LmStepArgs* step, Pending* out) {
    const int q = s->slot;
    // process-wide numbers: pinned blocks are reused by later solvers, and a number an earlier one left beside a sum
    // must never be taken for this pass's
    static std::atomic<unsigned long long> g_seq{0};
    const unsigned long long seq = s->seq = ++g_seq;
    out->direct = false, out->nrange = 0;
    // Publishing from the last workgroup of the column-sum kernel saves a kernel boundary (~3 us) but costs every
    // workgroup a device-wide fence and an atomic on one counter (serialised; the fence writes the XCD's L2 back):
    // past a few dozen columns a publishing kernel of its own is cheaper (499 columns: 25.1 us against 8.9 + 3.9; the
    // 111 columns of a monocular Jacobian pass: 0.1536 against 0.1515 ms per iteration; the 3 or 4 of a trial step
    // are better off in one kernel).
    int ncol = 0;
    for (int i = 0; i < s->segs.nseg; i++) ncol += s->segs.seg[i].n;
    const bool split = s->split_publish >= 0 ? s->split_publish != 0 : ncol >= 64;
    if (step) {
        step->red = s->d_red[q], step->n = n, step->hred = s->h_red[q], step->flag = s->h_flag[q], step->seq = seq;
        step->cin = ctrl;
        if (s->comm) {  // the sums pass through the all-reduce first: the step runs as a kernel of its own after it
            launch_colsum(s->segs, s->d_red[q], s->d_counter, nullptr, 0, nullptr, 0, s->stream, ctrl);
            NCCL_TRY(ncclAllReduce(s->d_red[q], s->d_red[q], n, ncclDouble, ncclSum, s->comm, s->stream));
            launch_lm_step(*step, s->stream);
        } else if (split) {
            launch_colsum(s->segs, s->d_red[q], s->d_counter, nullptr, 0, nullptr, 0, s->stream, ctrl);
            launch_lm_step(*step, s->stream);
        } else  // single GPU: the last workgroup of the column-sum kernel runs it
            launch_colsum(s->segs, s->d_red[q], s->d_counter, nullptr, 0, nullptr, 0, s->stream, ctrl, step);
    } else if (s->comm) {
        launch_colsum(s->segs, s->d_red[q], s->d_counter, nullptr, 0, nullptr, 0, s->stream, ctrl);
        NCCL_TRY(ncclAllReduce(s->d_red[q], s->d_red[q], n, ncclDouble, ncclSum, s->comm, s->stream));
        launch_publish(s->d_red[q], n, s->h_red[q], s->h_flag[q], seq, s->stream);
    } else if (s->direct_sums && !ctrl && ncol < 256) {  // (beyond a few hundred sums one burst from a publishing kernel is as fast)
        // one GPU, plain pass: no flag at all.  Every workgroup of the column-sum kernel sends its sum and the pass's
        // number to pinned memory in one 16-byte store; the host waits until every sum it expects carries the number.
        launch_colsum(s->segs, s->d_red[q], s->d_counter, nullptr, 0, nullptr, seq, s->stream, nullptr, nullptr, s->h_pair[q]);
        out->direct = true;
        out->nrange = s->segs.nseg;
        for (int i = 0; i < s->segs.nseg; i++) out->range[i] = Pending::Range{s->segs.seg[i].out_ofs, s->segs.seg[i].n};
    } else if (split) {
        launch_colsum(s->segs, s->d_red[q], s->d_counter, nullptr, 0, nullptr, 0, s->stream, ctrl);
        launch_publish(s->d_red[q], n, s->h_red[q], s->h_flag[q], seq, s->stream);
    } else {
        launch_colsum(s->segs, s->d_red[q], s->d_counter, s->h_red[q], n, s->h_flag[q], seq, s->stream, ctrl);
    }
    s->segs.nseg = 0;
    s->part_used = 0;
    s->slot ^= 1;
    HIP_TRY(hipGetLastError());
    out->slot = q, out->seq = seq, out->stream = s->stream;
    return CCALIB_OK;
}

// Waits for the sums of a submitted pass; they are then in s->h_red[p.slot].
int wait_reduction(ccalib_solver* s, const Pending& p, int n) {
    auto tw0 = std::chrono::steady_clock::now();
    s->t_enqueue += std::chrono::duration<double>(tw0 - s->t_enter).count();
    volatile unsigned long long* flag = s->h_flag[p.slot];
    // direct publication: sum i has arrived when the two words of its pair differ by exactly this pass's number.
    // A pair caught half-way between two passes cannot pass for new with an old sum: with y old, x ^ y == seq needs
    // x == x_old ^ seq_old ^ seq, which is not x_old (the numbers differ), and with y new it needs x == x_new.
    const volatile unsigned long long* pairs = reinterpret_cast<const volatile unsigned long long*>(s->h_pair[p.slot]);
    int ri = 0, rk = 0;  // next sum to wait for: range ri, entry rk
    auto arrived = [&]() {
        if (!p.direct) return *flag == p.seq;
        for (; ri < p.nrange; ri++, rk = 0)
            for (; rk < p.range[ri].n; rk++) {
                const int i = p.range[ri].ofs + rk;
                const unsigned long long x = pairs[2 * i], y = pairs[2 * i + 1];
                if ((x ^ y) != p.seq) return false;
                std::memcpy(&s->h_red[p.slot][i], &x, sizeof(double));
            }
        return true;
    };
    for (unsigned spins = 0; !arrived(); spins++) {
        if ((spins & 0xffff) == 0xffff) {  // a failed launch or a fault must not leave the host spinning
            hipError_t q = hipStreamQuery(p.stream ? p.stream : s->stream);
            if (q != hipSuccess && q != hipErrorNotReady)
                return set_error(CCALIB_ERR_HIP, "stream failed while waiting for a reduction: %s", hipGetErrorString(q));
            if (q == hipSuccess && !arrived()) return set_error(CCALIB_ERR_HIP, "reduction finished without publishing");
            // nor must a peer that left a sharded run (its all-reduce never completes): bounded wait
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count() > s->wait_limit_s)
                return set_error(CCALIB_ERR_HIP, "no reduction result after %.0f s (a rank of a sharded run may have stopped)",
                                 s->wait_limit_s);
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    s->t_wait_end = std::chrono::steady_clock::now();
    s->t_wait += std::chrono::duration<double>(s->t_wait_end - tw0).count();
    s->t_enter = s->t_wait_end;
    if (s->cb) {
        int rc = s->cb(s->cb_user, s->h_red[p.slot], n);
        if (rc) return set_error(CCALIB_ERR_HIP, "all-reduce callback failed (%d)", rc);
    }
    return CCALIB_OK;
}

// A plain pass: submit, wait; returns the pinned buffer holding the sums.
int finish_reduction(ccalib_solver* s, int n, const double** red) {
    Pending p;
    int rc = submit_reduction(s, n, nullptr, nullptr, &p);
    if (rc) return rc;
    if ((rc = wait_reduction(s, p, n))) return rc;
    *red = s->h_red[p.slot];
    return CCALIB_OK;
}

// Worst-case arena size of one pass (doubles).
size_t part_capacity(const ccalib_solver* s) {
    size_t n = 0;
    int kc = s->kcols();
    for (int c = 0; c < s->ncams; c++)
        n += (size_t)std::max(std::max(s->grid_fused, s->grid_fused12) * kTileViews, s->grid_red) * s->ops[c]->NRED_U + 64 +
             (size_t)std::max(s->grid_red, s->grid_err) + 64;
    const size_t srows = s->ncams == 2 ? s->sums_rows : s->nblk64;
    n += 2 * (srows * (kc * (kc + 1) / 2 + kc) + 64);  // Schur sums, for up to two dampings in one pass
    n += (size_t)s->nblk64 * 2 + 64;                   // step norms
    if (s->ncams == 2) n += srows * (27 + 6 * s->ops[1]->NC) + 64;
    return n;
}

// Residual pass of camera c at the trial poses: partial sums (s->grid_err of them) into `part`.
void enqueue_err(ccalib_solver* s, int c, const CamParams& cp, double* part) {
    PassArgs a = pass_args(s, c, c == 0 ? s->d_pose_cur : s->d_pose3);
    double* pc = c == 0 ? s->d_pc : s->d_pc3;
    bool& fresh = c == 0 ? s->pc_fresh : s->pc3_fresh;
    if (!fresh) launch_pose_const(a.pose, s->nviews, pc, s->stream);
    fresh = true;
    launch_err_wave(a, cp, s->rational[c], pc, s->grid_err, part, s->stream);
}

void current_shared(const ccalib_solver* s, double* shared) {
    if (s->ncams == 1) {
        std::memcpy(shared, s->intr[0], sizeof(double) * 18);
    } else {
        std::memcpy(shared, s->lr, sizeof(double) * 6);
        if (s->fix_intrinsic) return;
        std::memcpy(shared + 6, s->intr[0], sizeof(double) * 18);
        std::memcpy(shared + 24, s->intr[1], sizeof(double) * 18);
    }
}
void store_shared(ccalib_solver* s, const double* shared) {
    if (s->ncams == 1) {
        std::memcpy(s->intr[0], shared, sizeof(double) * 18);
    } else {
        std::memcpy(s->lr, shared, sizeof(double) * 6);
        if (s->fix_intrinsic) return;
        std::memcpy(s->intr[0], shared + 6, sizeof(double) * 18);
        std::memcpy(s->intr[1], shared + 24, sizeof(double) * 18);
    }
}

// Stereo: camera-1 poses (and, for Jacobian passes, the composition derivatives, into buffer `buf`)
// for the camera-0 poses `pose1` and the current relative pose.
int stereo_compose(ccalib_solver* s, bool with_jac, int buf, const double* pose1) {
    if (s->ncams != 2) return CCALIB_OK;
    StereoLR lr;
    for (int i = 0; i < 6; i++) lr.v[i] = s->lr[i];
    const bool want_pc3 = !with_jac || (s->ops[1]->jac_fused && s->use_fused);  // the residual kernel reads records too
    launch_compose(pose1, lr, s->nviews, s->npad, s->d_pose3, with_jac ? s->d_cmp[buf] : nullptr,
                   want_pc3 ? s->d_pc3 : nullptr, s->stream);
    s->pc3_fresh = want_pc3;
    return CCALIB_OK;
}

// ---- the backend operations ---------------------------------------------------------------------

// Launches the Schur-sum kernels of buffer `buf` for `lambda` and appends their column sums to the reduction
// at `ofs`.  Layout appended: upper triangle (kc(kc+1)/2) then right-hand side (kc), kc = kernel columns.
//   lambda2 != 0: the same sums for a second damping, appended at ofs2 (same launch, other half of the grid).
int enqueue_schur(ccalib_solver* s, int buf, double lambda, int ofs, const LmCtrl* ctrl, int lr_ofs = -1, double lambda2 = 0.,
                  int ofs2 = -1) {
    int kc = s->kcols(), nsum = kc * (kc + 1) / 2 + kc;
    if (s->ncams == 1) {
        PART_TAKE(part, s, (size_t)s->nblk64 * nsum);
        double* part2 = nullptr;
        if (lambda2 != 0.) {
            PART_TAKE(p2, s, (size_t)s->nblk64 * nsum);
            part2 = p2;
        }
        s->ops[0]->schur(s->d_blk[0][buf], s->npad, s->nviews, lambda, part, s->stream, ctrl, lambda2, part2);
        int rc = queue_colsum(s, part, s->nblk64, nsum, ofs);
        if (!rc && part2) rc = queue_colsum(s, part2, s->nblk64, nsum, ofs2);
        return rc;
    }
    // stereo: one kernel takes the Schur sums and, with lr_ofs >= 0, the sums that enter the shared block through
    // the relative pose: B^T V3 B (21), B^T ge3 (6), W3 B (NC1 x 6)
    const int nlr = 27 + 6 * s->ops[1]->NC;
    PART_TAKE(part, s, (size_t)s->sums_rows * nsum);
    double* part_lr = nullptr;
    if (lr_ofs >= 0) {
        PART_TAKE(pl, s, (size_t)s->sums_rows * nlr);
        part_lr = pl;
    }
    double* part2 = nullptr;
    if (lambda2 != 0.) {
        PART_TAKE(p2, s, (size_t)s->sums_rows * nsum);
        part2 = p2;
    }
    bool ok;
    if (s->sums_quad)
        ok = launch_stereo_sums(s->ops[0]->NC, s->d_blk[0][buf], s->d_blk[1][buf], s->d_cmp[buf], s->lr_jac[buf], s->npad,
                                s->nviews, lambda, s->d_sums_tab, s->sums_rows, part, part_lr, s->stream, lambda2, part2);
    else {
        ok = launch_stereo_view_sums(s->ops[0]->NC, s->d_blk[0][buf], s->d_blk[1][buf], s->d_cmp[buf], s->lr_jac[buf], s->npad,
                                     s->nviews, lambda, s->d_view_tab, part, s->stream, lambda2, part2);
        if (ok && part_lr)
            ok = launch_stereo_lr_sums(s->ops[1]->NC, s->d_blk[1][buf], s->d_cmp[buf], s->lr_jac[buf], s->npad, s->nviews,
                                       part_lr, s->stream);
    }
    if (!ok) return set_error(CCALIB_ERR_UNSUPPORTED, "stereo column count %d is not built in", s->ops[0]->NC);
    int rc = queue_colsum(s, part, s->sums_rows, nsum, ofs);
    if (!rc && part_lr) rc = queue_colsum(s, part_lr, s->sums_rows, nlr, lr_ofs);
    if (!rc && part2) rc = queue_colsum(s, part2, s->sums_rows, nsum, ofs2);
    return rc;
}

// Where the sums of one Jacobian pass sit in the reduction buffer, relative to the pass's base offset.
struct JacLayout {
    int cam_ofs[2] = {0, 0};
    int lr_ofs = 0, schur_ofs = 0, nsum = 0, end = 0;
};
JacLayout jac_layout(const ccalib_solver* s) {
    JacLayout L;
    int ofs = 0;
    for (int c = 0; c < s->ncams; c++) L.cam_ofs[c] = ofs, ofs += s->ops[c]->NRED_U;
    L.lr_ofs = ofs;
    if (s->ncams == 2) ofs += 27 + 6 * s->ops[1]->NC;
    L.schur_ofs = ofs;
    const int kc = s->kcols();
    L.nsum = kc * (kc + 1) / 2 + kc;
    L.end = ofs + L.nsum;
    return L;
}

// In-loop timing: an event on the solver's stream, from a pool that grows as needed (only while profiling is on).
void prof_mark(ccalib_solver* s) {
    if (!s->prof_on) return;
    if (s->prof_used == s->prof_ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        s->prof_ev.push_back(e);
    }
    (void)hipEventRecord(s->prof_ev[s->prof_used++], s->stream);
}

// Whether the Jacobian pass runs as one launch of the 12-column fused kernel over all cameras.
bool tile12_possible(const ccalib_solver* s) {
    if (!s->use_tile12 || !s->use_fused) return false;
    for (int c = 0; c < s->ncams; c++)
        if (s->ops[c] != s->ops[0] || !has_jac_fused12(s->ops[c]->mask, s->ops[c]->aspect, s->ops[c]->rational)) return false;
    return fused12_geom(s->maxnp, s->shared_board, s->board_np, 1).lds_bytes <= 64 * 1024;
}

// Enqueues a whole Jacobian pass into block buffer `buf`, with the Schur sums for `lambda_hint`; the reduced sums
// land in the reduction buffer from `ofs` on, laid out as jac_layout() says.  The pass evaluates camera 0 at the
// poses `pose` (records in d_pc must match them).  By-value parameters (the solver's current intrinsics and
// relative pose) are used unless `ctrl` names the control block of a run-ahead iteration.
// lambda_extra != 0: a second set of Schur sums, for that damping, follows the first at offset jac_layout().end.
// composed: camera 1's poses, their records and the composition derivatives in d_cmp[buf] are already in place.
int enqueue_jac(ccalib_solver* s, int buf, double lambda_hint, int ofs, const double* pose, const LmCtrl* ctrl,
                double lambda_extra = 0., bool composed = false) {
    int rc = composed ? CCALIB_OK : stereo_compose(s, true, buf, pose);
    if (rc) return rc;
    s->n_jac_exec++;
    const JacLayout L = jac_layout(s);
    for (int i = 0; i < 6; i++) s->lr_jac[buf].v[i] = s->lr[i];
    if (tile12_possible(s) && !ctrl) {
        // every camera's fused pass in one launch on the 12-column tile
        const ModelOps* m = s->ops[0];
        Fused12Cams cams;
        const int rows = s->grid_fused12;
        for (int c = 0; c < s->ncams; c++) {
            PART_TAKE(part, s, (size_t)rows * m->NRED_U);
            double* pc = c == 0 ? s->d_pc : s->d_pc3;
            bool& fresh = c == 0 ? s->pc_fresh : s->pc3_fresh;
            if (!fresh) launch_pose_const(c == 0 ? pose : s->d_pose3, s->nviews, pc, s->stream);
            fresh = true;
            cams.c[c] = Fused12Cam{cam_params(s->intr[c], s->aspect[c]), s->d_img[c], pc, s->d_blk[c][buf], part};
            if ((rc = queue_colsum(s, part, rows, m->NRED_U, ofs + L.cam_ofs[c]))) return rc;
        }
        if (s->ncams == 1) cams.c[1] = cams.c[0];
        prof_mark(s);
        if (!launch_jac_fused12(m->mask, m->aspect, m->rational, pass_args(s, 0, pose), cams, s->ncams, rows, s->stream))
            return set_error(CCALIB_ERR_STATE, "the 12-column fused pass refused a problem it was selected for");
        prof_mark(s);
    } else
    for (int c = 0; c < s->ncams; c++) {
        const ModelOps* m = s->ops[c];
        CamParams cp = cam_params(s->intr[c], s->aspect[c]);
        PassArgs a = pass_args(s, c, c == 0 ? pose : s->d_pose3);
        if (m->jac_fused && s->use_fused) {
            int rows = s->grid_fused;  // one row of partial sums per tile slot
            PART_TAKE(part, s, (size_t)rows * m->NRED_U);
            double* pc = c == 0 ? s->d_pc : s->d_pc3;
            bool& fresh = c == 0 ? s->pc_fresh : s->pc3_fresh;
            if (!fresh && !ctrl) launch_pose_const(a.pose, s->nviews, pc, s->stream);
            fresh = true;
            prof_mark(s);
            m->jac_fused(a, cp, pc, s->grid_fused, s->d_blk[c][buf], part, s->stream, ctrl, c);
            prof_mark(s);
            if ((rc = queue_colsum(s, part, rows, m->NRED_U, ofs + L.cam_ofs[c]))) return rc;
        } else {
            PART_TAKE(part, s, (size_t)s->grid_red * m->NRED_U);
            m->jac_shared(a, cp, s->grid_red, part, s->stream);
            if ((rc = queue_colsum(s, part, s->grid_red, m->NRED_U, ofs + L.cam_ofs[c]))) return rc;
            m->jac_view(a, cp, s->grid_view, s->d_blk[c][buf], s->npad, s->stream);
        }
    }
    if ((rc = enqueue_schur(s, buf, lambda_hint, ofs + L.schur_ofs, ctrl, s->ncams == 2 ? ofs + L.lr_ofs : -1, lambda_extra,
                            ofs + L.end)))
        return rc;
    if (ofs + L.end + (lambda_extra != 0. ? L.nsum : 0) > kPredOfs) return set_error(CCALIB_ERR_STATE, "reduction buffer too small");
    return CCALIB_OK;
}

// Index of every entry of the shared block U (ns x ns, upper triangle) and of its gradient among the sums of a
// Jacobian pass (relative to the pass's base offset; -1: structurally zero).  The host unpacks U with it and the
// device assembles the reduced system with it, so the two cannot disagree on where a number lives.
void shared_block_index(const ccalib_solver* s, const JacLayout& L, std::vector<int>& Uidx, std::vector<int>& gidx, int err_idx[2]) {
    const int ns = s->ns();
    Uidx.assign((size_t)ns * ns, -1);
    gidx.assign(ns, -1);
    for (int c = 0; c < s->ncams; c++) {
        const ModelOps* m = s->ops[c];
        const int r0 = L.cam_ofs[c], base = s->ncams == 1 ? 0 : 6 + 18 * c;
        err_idx[c] = r0 + m->NRED_U - 1;
        if (s->fix_intrinsic) continue;  // the intrinsic columns are not parameters: only the residual sum is kept
        int k = 0;
        for (int i = 0; i < m->NC; i++)
            for (int j = i; j < m->NC; j++) Uidx[(size_t)(base + m->col(i)) * ns + base + m->col(j)] = r0 + k++;
        for (int i = 0; i < m->NC; i++) gidx[base + m->col(i)] = r0 + k++;
    }
    if (s->ncams == 2) {
        int k = L.lr_ofs;
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++) Uidx[(size_t)i * ns + j] = k++;
        for (int i = 0; i < 6; i++) gidx[i] = k++;
        const ModelOps* m1 = s->ops[1];
        if (!s->fix_intrinsic)
            for (int i = 0; i < m1->NC; i++)
                for (int b = 0; b < 6; b++) Uidx[(size_t)b * ns + 24 + m1->col(i)] = k++;
    }
}

// Builds the index tables for the solver's current column sets and free-parameter mask: the host's (Uidx, gidx,
// err_idx) and the device's (LmMaps, uploaded).  The free set is the one ccalib_solver_run hands the driver.
int build_step_maps(ccalib_solver* s) {
    const JacLayout L = jac_layout(s);
    shared_block_index(s, L, s->Uidx, s->gidx, s->err_idx);
    LmMaps& m = s->h_maps;
    std::memset(&m, 0, sizeof(m));
    const int ns = s->ns(), kc = s->kcols();
    m.ns = ns, m.ncams = s->ncams, m.nerr = s->ncams;
    m.err_idx[0] = s->err_idx[0], m.err_idx[1] = s->err_idx[1];
    unsigned char smask[kMaxShared];
    if (s->ncams == 1)
        std::memcpy(smask, s->imask[0], 18);
    else {
        std::memset(smask, 1, 6);
        if (!s->fix_intrinsic) std::memcpy(smask + 6, s->imask[0], 18), std::memcpy(smask + 24, s->imask[1], 18);
    }
    std::vector<int> s2k(ns, -1);
    for (int k = 0; k < kc; k++)
        if (s->kcol_to_shared(k) < ns) s2k[s->kcol_to_shared(k)] = k;
    int nf = 0;
    bool ok = kc <= kCtrlCols;
    for (int i = 0; i < ns; i++)
        if (smask[i]) {
            m.fidx[nf] = i;
            m.dc_slot[nf] = s2k[i];
            ok = ok && s2k[i] >= 0;
            nf++;
        }
    if (!ok) nf = 0;  // a free parameter without a kernel column: the driver reports it; nothing is run ahead
    m.nf = nf;
    auto tri = [kc](int a, int b) { return a * kc - a * (a - 1) / 2 + (b - a); };
    for (int a = 0; a < nf; a++) {
        const int ia = m.fidx[a], ka = m.dc_slot[a];
        m.g_idx[a] = s->gidx[ia];
        m.bs_idx[a] = L.schur_ofs + kc * (kc + 1) / 2 + ka;
        for (int c = 0; c < nf; c++) {
            const int ic = m.fidx[c], kb = m.dc_slot[c];
            m.u_idx[a * nf + c] = s->Uidx[(size_t)std::min(ia, ic) * ns + std::max(ia, ic)];
            m.s_idx[a * nf + c] = L.schur_ofs + (ka <= kb ? tri(ka, kb) : tri(kb, ka));
        }
    }
    HIP_TRY(hipMemcpy(s->d_maps, &m, sizeof(m), hipMemcpyHostToDevice));
    return CCALIB_OK;
}

// Unpacks the sums of a finished pass (r = reduction buffer at the pass's base offset): shared block U (ns x ns,
// upper triangle), its gradient, the residual sum and the Schur sums.
void parse_jac(const ccalib_solver* s, const double* r, double* U, double* gs, double* err, std::vector<double>& S) {
    const int ns = s->ns();
    const JacLayout L = jac_layout(s);
    for (int i = 0; i < ns * ns; i++) U[i] = s->Uidx[i] >= 0 ? r[s->Uidx[i]] : 0.;
    for (int i = 0; i < ns; i++) gs[i] = s->gidx[i] >= 0 ? r[s->gidx[i]] : 0.;
    double e = 0;
    for (int c = 0; c < s->ncams; c++) e += r[s->err_idx[c]];
    *err = e;
    S.assign(r + L.schur_ofs, r + L.schur_ofs + L.nsum);
}

// ---- running one iteration ahead of the host (see lm_step.hpp) ----------------------------------

double next_lambda(double lambda) {  // what ccalib_lm_run announces after an accepted step: one decade down, floor 1e-16
    const int lg = (int)std::lround(std::log10(lambda));
    return std::exp(std::max(lg - 1, -16) * std::log(10.));
}
int lambda_level(double lambda) { return std::min(32, std::max(0, (int)std::lround(std::log10(lambda)) + 16)); }

// The damping whose Schur sums are worth taking along with those for `lambda`: one decade up, when the last
// trial at `lambda`'s level was rejected and the last one a decade up accepted; 0 otherwise.
double retry_lambda(const ccalib_solver* s, double lambda) {
    const int lv = lambda_level(lambda);
    if (!s->speculate || lv >= 32 || s->level_accepted[lv] || !s->level_accepted[lv + 1]) return 0.;
    return std::exp((lv + 1 - 16) * std::log(10.));  // the value ccalib_lm_run passes for that exponent
}

bool chain_possible(const ccalib_solver* s) {
    return s->chain_enabled && s->speculate && !s->cb && s->ncams == 1 && s->run_solve_method == 1 && s->d_ctrl &&
           s->ops[0]->jac_fused && s->use_fused && s->h_maps.nf > 0 && !tile12_possible(s);
}

// Arguments of the k_lm_step that ends a Jacobian pass evaluated at `shared` (by value; a run-ahead iteration
// takes its trial point from its control block instead): predict the step of damping `lambda` from it.
void fill_step(const ccalib_solver* s, LmStepArgs& a, int base, double lambda, double prev_err, const double* shared, bool predict) {
    std::memset(&a, 0, sizeof(a));
    a.maps = s->d_maps;
    a.base = base;
    a.lambda = lambda;
    a.prev_err = prev_err;
    if (shared) std::memcpy(a.shared, shared, sizeof(double) * s->ns());
    a.aspect[0] = s->aspect[0], a.aspect[1] = s->aspect[1];
    a.same_focal = s->same_focal, a.fix_intrinsic = s->fix_intrinsic;
    for (int c = 0; c < s->ncams; c++) a.cp0[c] = cam_params(s->intr[c], s->aspect[c]);
    a.cout = predict ? s->d_ctrl + s->slot_after() : nullptr;
}

// Enqueues iteration "trial step of damping `lambda` from the point the pass just submitted evaluates", on the
// assumption that that point is accepted: its kernels take step, cameras and go word from the control block
// the preceding k_lm_step writes.  `blk_base`: block buffer holding the Jacobian at that point; `pose_base` /
// `pose_trial`: committed poses then, and where the trial poses go.
int enqueue_chain(ccalib_solver* s, double lambda, int blk_base, double* pose_base, double* pose_trial) {
    ccalib_solver::Chain& ch = s->chain;
    const LmCtrl* ctrl = s->d_ctrl + s->slot;
    ch.inflight = false;
    ch.lambda = lambda, ch.lambda_next = next_lambda(lambda);
    ch.blk_base = blk_base, ch.pose_base = pose_base, ch.pose_trial = pose_trial;
    int rc;
    PART_TAKE(npart, s, (size_t)s->nblk64 * 2);
    DeltaC dc;
    for (int k = 0; k < I_NMODEL; k++) dc.v[k] = 0;
    s->ops[0]->backsub(s->d_blk[0][blk_base], s->npad, s->nviews, lambda, dc, pose_base, pose_trial, npart, s->d_pc, s->stream, ctrl);
    if ((rc = queue_colsum(s, npart, s->nblk64, 2, 0))) return rc;
    if ((rc = enqueue_jac(s, blk_base ^ 1, ch.lambda_next, 2, pose_trial, ctrl))) return rc;
    LmStepArgs st;
    // one more iteration may follow this one (the driver stops at its iteration limit)
    const bool more = s->commits + 1 < s->run_max_iter && s->level_accepted[lambda_level(ch.lambda_next)];
    fill_step(s, st, 2, ch.lambda_next, 0., nullptr, more);
    ch.predicts = more;
    if ((rc = submit_reduction(s, 2 + jac_layout(s).end, ctrl, &st, &ch.pend))) return rc;
    ch.inflight = true;
    s->n_ahead++;
    return CCALIB_OK;
}

// Reads what the k_lm_step of the pass in `slot` predicted.
void read_prediction(ccalib_solver* s, int slot, bool predicted) {
    const double* h = s->h_red[slot] + kPredOfs;
    s->pred_go = predicted && h[0] != 0.;
#ifdef CCALIB_LM_TIMING
    if (std::getenv("CCALIB_TIMING")) {
        std::fprintf(stderr, "[lm_step] ticks(10ns):");
        for (int i = 0; i < (int)h[199]; i++) std::fprintf(stderr, " %.0f", h[200 + i]);
        std::fprintf(stderr, "\n");
    }
#endif
    if (s->pred_go) {
        std::memcpy(s->pred_delta, h + 1, sizeof(double) * s->h_maps.nf);
        std::memcpy(s->pred_shared, h + 1 + s->h_maps.nf, sizeof(double) * s->ns());
    }
}

int be_jac(void* user, const double* shared, double lambda_hint, double* U, double* gs, double* err) {
    ccalib_solver* s = (ccalib_solver*)user;
    s->t_enter = std::chrono::steady_clock::now();
    (void)hipGetLastError();  // an error another HIP user of this thread left behind is not ours to report
    const int ns = s->ns();
    if (s->steps_since_jac > 0) {  // the driver moved on: that step was accepted
        s->level_accepted[s->last_level] = true;
        s->prev_trials = s->steps_since_jac;
    }
    s->steps_since_jac = 0;
    if (s->spec_valid && std::memcmp(shared, s->spec_shared, sizeof(double) * ns) == 0) {
        // this pass was run ahead by the step that produced the point: adopt its buffer and sums
        s->spec_valid = false;
        s->cur ^= 1;
        store_shared(s, shared);
        std::memcpy(U, s->spec_U.data(), sizeof(double) * ns * ns);
        std::memcpy(gs, s->spec_gs.data(), sizeof(double) * ns);
        *err = s->spec_err;
        s->last_err = s->spec_err;
        s->cachedS = s->spec_S;
        s->cached_lambda = s->spec_lambda;
        s->have_cached = true;
        s->have_cached2 = s->spec_have2;
        if (s->spec_have2) s->cachedS2 = s->spec_S2, s->cached_lambda2 = s->spec_lambda2;
        s->com_U.assign(U, U + (size_t)ns * ns), s->com_gs.assign(gs, gs + ns);
        std::memcpy(s->com_shared, shared, sizeof(double) * ns);
        s->com_valid = true;
        return CCALIB_OK;
    }
    s->spec_valid = false;
    if (s->retry.inflight) {  // the first trial was accepted after all: the retry behind it runs to its end unused
        s->retry.inflight = false;
        s->pc_fresh = s->pc3_fresh = false;  // the pose records are the retry's
        s->n_ahead_dropped++;
    }
    if (s->chain.inflight) {  // an iteration run ahead from a point the driver did not go to
        s->chain.inflight = false;
        s->pc_fresh = false;
        s->n_ahead_dropped++;
    }
    store_shared(s, shared);
    int rc;
    const double lam2 = retry_lambda(s, lambda_hint);
    if ((rc = enqueue_jac(s, s->cur, lambda_hint, 0, s->d_pose_cur, nullptr, lam2))) return rc;
    const JacLayout L0 = jac_layout(s);
    const int n = L0.end + (lam2 != 0. ? L0.nsum : 0);
    Pending p;
    const bool chain = chain_possible(s) && s->commits < s->run_max_iter && s->level_accepted[lambda_level(lambda_hint)];
    if (chain_possible(s)) {
        LmStepArgs st;
        fill_step(s, st, 0, lambda_hint, HUGE_VAL, shared, chain);
        if ((rc = submit_reduction(s, n, nullptr, &st, &p))) return rc;
        // the first step of the run, should the driver take it from here: committed poses are d_pose_cur (the
        // driver's commit swaps the two pose buffers before it steps)
        if (chain && (rc = enqueue_chain(s, lambda_hint, s->cur, s->d_pose_cur, s->d_pose_prev))) return rc;
    } else if ((rc = submit_reduction(s, n, nullptr, nullptr, &p)))
        return rc;
    if ((rc = wait_reduction(s, p, n))) return rc;
    parse_jac(s, s->h_red[p.slot], U, gs, err, s->cachedS);
    read_prediction(s, p.slot, chain);
    s->cached_lambda = lambda_hint;
    s->have_cached = true;
    s->have_cached2 = lam2 != 0.;
    if (lam2 != 0.) s->cachedS2.assign(s->h_red[p.slot] + L0.end, s->h_red[p.slot] + L0.end + L0.nsum), s->cached_lambda2 = lam2;
    s->last_err = *err;
    if (s->run_first_jac) s->drv_err = *err, s->run_first_jac = false;
    s->com_U.assign(U, U + (size_t)ns * ns), s->com_gs.assign(gs, gs + ns);
    std::memcpy(s->com_shared, shared, sizeof(double) * ns);
    s->com_valid = true;
    return CCALIB_OK;
}

// Schur sums in kernel-column order (packed upper triangle, then right-hand side) -> the driver's free set
int map_schur_sums(const ccalib_solver* s, const std::vector<double>& S, const int* fidx, int nf, double* Ssum, double* bsum) {
    const int kc = s->kcols();
    std::vector<int> s2k(s->ns(), -1);
    for (int k = 0; k < kc; k++)
        if (s->kcol_to_shared(k) < s->ns()) s2k[s->kcol_to_shared(k)] = k;
    auto tri = [kc](int a, int b) { return a * kc - a * (a - 1) / 2 + (b - a); };
    for (int a = 0; a < nf; a++) {
        int ka = s2k[fidx[a]];
        if (ka < 0) return set_error(CCALIB_ERR_STATE, "free parameter %d is not a kernel column", fidx[a]);
        for (int c = 0; c < nf; c++) {
            int kb = s2k[fidx[c]];
            if (kb < 0) return set_error(CCALIB_ERR_STATE, "free parameter %d is not a kernel column", fidx[c]);
            Ssum[(size_t)a * nf + c] = S[ka <= kb ? tri(ka, kb) : tri(kb, ka)];
        }
        bsum[a] = S[kc * (kc + 1) / 2 + ka];
    }
    return CCALIB_OK;
}

int be_reduce(void* user, double lambda, const int* fidx, int nf, double* Ssum, double* bsum) {
    ccalib_solver* s = (ccalib_solver*)user;
    s->t_enter = std::chrono::steady_clock::now();
    (void)hipGetLastError();  // an error another HIP user of this thread left behind is not ours to report
    int kc = s->kcols(), nsum = kc * (kc + 1) / 2 + kc;
    if (!(s->have_cached && s->cached_lambda == lambda) && s->have_cached2 && s->cached_lambda2 == lambda) {
        s->cachedS.swap(s->cachedS2);  // the sums taken along for the retry
        std::swap(s->cached_lambda, s->cached_lambda2);
        s->have_cached = true;
    }
    if (!(s->have_cached && s->cached_lambda == lambda)) {
        // A damping nobody prepared sums for: a retry after a rejection.  Rejections come in runs (near the noise
        // floor the damping climbs many decades inside one iteration), so the sums for the next decade are taken in
        // the same round trip: one more pass over the blocks against a whole host turn-around if they are needed.
        const int lv = lambda_level(lambda);
        const double lambda_up = s->speculate && lv < 32 ? std::exp((lv + 1 - 16) * std::log(10.)) : 0.;
        int rc = enqueue_schur(s, s->cur, lambda, 0, nullptr, -1, lambda_up, nsum);
        if (rc) return rc;
        const double* red;
        if ((rc = finish_reduction(s, lambda_up != 0. ? 2 * nsum : nsum, &red))) return rc;
        s->cachedS.assign(red, red + nsum);
        s->cached_lambda = lambda;
        s->have_cached = true;
        s->have_cached2 = lambda_up != 0.;
        if (lambda_up != 0.) s->cachedS2.assign(red + nsum, red + 2 * nsum), s->cached_lambda2 = lambda_up;
    }
    return map_schur_sums(s, s->cachedS, fidx, nf, Ssum, bsum);
}

// The trial point's residual sum comes either from the residual kernels or, when the step is
// likely to be accepted, from a whole Jacobian pass run ahead into the spare buffer: the driver's
// next call is then jac() at this same point, which be_jac answers without touching the GPU.
// A rejected step only costs the extra work; the committed Jacobian stays in buffer `cur`.

void be_fixup(void* user, double* shared, double* prev);

// Whether the nth trial of an iteration (1 = the first), at this damping, gets the Jacobian pass of its trial point
// run ahead instead of a residual pass.  A wrong "yes" wastes a Jacobian pass, a wrong "no" costs a residual pass and a
// host turn-around, so the guess follows what LM runs are seen to do -- they repeat the iteration before:
//   * first trials are accepted iteration after iteration while the damping walks down: yes for a first trial when
//     the iteration before accepted its first trial;
//   * a run settles into "first trial rejected, retry one decade up accepted" (the reference's stereo flag set):
//     yes for the nth trial when the iteration before was accepted at its nth trial and the last trial at this
//     damping level was accepted;
//   * near the noise floor the damping climbs many decades inside one iteration: those retries match neither;
//   * never in the run's last iteration, whose accepted point the driver does not ask a Jacobian for.
bool trial_runs_ahead(const ccalib_solver* s, double lambda, int nth) {
    if (!s->speculate || s->commits >= s->run_max_iter) return false;
    if (nth == 1 && s->prev_trials == 1) return true;
    return (nth == 1 || s->prev_trials == nth) && s->level_accepted[lambda_level(lambda)];
}

// A retry whose step is known before the trial ahead of it is launched (plan_retry_ahead).
struct RetryPlan {
    double lambda2 = 0;
    int nf = 0;
    double dsh[42], shared[42];
};

// Enqueues a trial step -- back-substitution into s->d_pose_cur, then either the Jacobian pass run ahead or the
// residual pass -- and submits its reduction.  Does not wait.
//   second        (stereo) the planned retry behind this trial: its back-substitution is taken along in this trial's
//                 kernel -- one read of the blocks for both -- into d_pose_retry / d_pc_retry, its step norms into
//                 the first 2 * nblk64 doubles of the other slot's arena, where the retry's own pass looks for them
//   backsub_done  this trial's back-substitution was taken along that way: nothing to launch for it
int launch_trial(ccalib_solver* s, double lambda, double lambda_next, const int* fidx, int nf, const double* dshared,
                 const double* shared_new, int nth, ccalib_solver::Trial* t, const RetryPlan* second = nullptr,
                 bool backsub_done = false) {
    int rc;
    const int cur = s->cur, ns = s->ns();
    bool stereo_composed = false;
    std::vector<double> dfull(ns, 0.);
    for (int a = 0; a < nf; a++) dfull[fidx[a]] = dshared[a];
    const bool ahead = trial_runs_ahead(s, lambda, nth);
    PART_TAKE(npart, s, (size_t)s->nblk64 * 2);
    const auto tl0 = std::chrono::steady_clock::now();
    if (backsub_done) {
        const bool want_pc = s->ops[0]->jac_fused && s->use_fused;
        s->pc_fresh = want_pc;
        s->pc3_fresh = false;
    } else if (s->ncams == 1) {
        DeltaC dc;
        for (int k = 0; k < I_NMODEL; k++) dc.v[k] = 0;
        for (int k = 0; k < s->ops[0]->NC; k++) dc.v[k] = dfull[s->ops[0]->col(k)];
        const bool want_pc = s->ops[0]->jac_fused && s->use_fused;
        s->ops[0]->backsub(s->d_blk[0][cur], s->npad, s->nviews, lambda, dc, s->d_pose_prev, s->d_pose_cur, npart,
                           want_pc ? s->d_pc : nullptr, s->stream, nullptr);
        s->pc_fresh = want_pc;
    } else {
        StereoDelta sd;
        int kc = s->kcols();
        for (int k = 0; k < kStereoMaxCols; k++)
            sd.v[k] = k < kc && s->kcol_to_shared(k) < ns ? dfull[s->kcol_to_shared(k)] : 0.;
        const bool want_pc = s->ops[0]->jac_fused && s->use_fused;
        // a residual pass follows unless the next Jacobian pass is run ahead (which composes with derivatives):
        // camera 1's trial poses then come out of this kernel too
        // ... and when it is run ahead, the composition derivatives that pass needs come out of it as well
        const bool compose_here = !ahead;
        const bool compose_jac = ahead && !second && s->fold_compose;
        const bool want_pc3 = compose_here || (s->ops[1]->jac_fused && s->use_fused);
        StereoLR lr_new;
        for (int i = 0; i < 6; i++) lr_new.v[i] = shared_new[i];
        StereoSecondTrial st2;
        if (second) {
            std::vector<double> d2(ns, 0.);
            for (int a = 0; a < second->nf; a++) d2[fidx[a]] = second->dsh[a];
            for (int k = 0; k < kStereoMaxCols; k++)
                st2.d.v[k] = k < kc && s->kcol_to_shared(k) < ns ? d2[s->kcol_to_shared(k)] : 0.;
            st2.lambda = second->lambda2;
            st2.pose = s->d_pose_retry;
            st2.pc = want_pc ? s->d_pc_retry : nullptr;
            st2.part = s->d_part[s->slot ^ 1];
        }
        if (!launch_stereo_backsub(s->ops[0]->NC, s->d_blk[0][cur], s->d_blk[1][cur], s->d_cmp[cur], s->lr_jac[cur],
                                   s->npad, s->nviews, lambda, sd, s->d_pose_prev, s->d_pose_cur, npart,
                                   want_pc ? s->d_pc : nullptr, lr_new, compose_here || compose_jac ? s->d_pose3 : nullptr,
                                   (compose_here || compose_jac) && want_pc3 ? s->d_pc3 : nullptr, s->stream,
                                   second ? &st2 : nullptr, compose_jac ? s->d_cmp[cur ^ 1] : nullptr))
            return set_error(CCALIB_ERR_UNSUPPORTED, "stereo column count %d is not built in", s->ops[0]->NC);
        s->pc_fresh = want_pc;
        s->pc3_fresh = (compose_here || compose_jac) && want_pc3;
        stereo_composed = compose_here || compose_jac;
    }
    s->t_think += std::chrono::duration<double>(tl0 - s->t_wait_end).count();
    s->t_launch1 += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
    if ((rc = queue_colsum(s, npart, s->nblk64, 2, 0))) return rc;
    store_shared(s, shared_new);
    t->ahead = ahead;
    t->predicted = false;
    t->lam2 = 0.;
    if (ahead) {
        t->lam2 = retry_lambda(s, lambda_next);
        if ((rc = enqueue_jac(s, cur ^ 1, lambda_next, 2, s->d_pose_cur, nullptr, t->lam2, stereo_composed))) return rc;
        const JacLayout L0 = jac_layout(s);
        t->n = 2 + L0.end + (t->lam2 != 0. ? L0.nsum : 0);
        if (chain_possible(s)) {
            t->predicted = s->commits < s->run_max_iter && s->level_accepted[lambda_level(lambda_next)];
            LmStepArgs st;
            fill_step(s, st, 2, lambda_next, s->committed_err, shared_new, t->predicted);
            if ((rc = submit_reduction(s, t->n, nullptr, &st, &t->p))) return rc;
            if (t->predicted && (rc = enqueue_chain(s, lambda_next, cur ^ 1, s->d_pose_cur, s->d_pose_prev))) return rc;
        } else if ((rc = submit_reduction(s, t->n, nullptr, nullptr, &t->p)))
            return rc;
    } else {
        // residual pass at the trial point
        s->n_err_exec++;
        if (!stereo_composed && (rc = stereo_compose(s, false, cur, s->d_pose_cur))) return rc;
        if (s->ncams == 2 && s->rational[0] == s->rational[1]) {
            ErrPair e;
            for (int c = 0; c < 2; c++) {
                e.a[c] = pass_args(s, c, c == 0 ? s->d_pose_cur : s->d_pose3);
                e.cp[c] = cam_params(s->intr[c], s->aspect[c]);
                double* pc = c == 0 ? s->d_pc : s->d_pc3;
                bool& fresh = c == 0 ? s->pc_fresh : s->pc3_fresh;
                if (!fresh) launch_pose_const(e.a[c].pose, s->nviews, pc, s->stream);
                fresh = true;
                e.pc[c] = pc;
                PART_TAKE(epart, s, (size_t)s->grid_err);
                e.part[c] = epart;
                if ((rc = queue_colsum(s, epart, s->grid_err, 1, 2 + c))) return rc;
            }
            launch_err_wave_pair(e, s->rational[0], s->grid_err, s->stream);
        } else
            for (int c = 0; c < s->ncams; c++) {
                CamParams cp = cam_params(s->intr[c], s->aspect[c]);
                PART_TAKE(epart, s, (size_t)s->grid_err);
                enqueue_err(s, c, cp, epart);
                if ((rc = queue_colsum(s, epart, s->grid_err, 1, 2 + c))) return rc;
            }
        t->n = 2 + s->ncams;
        if ((rc = submit_reduction(s, t->n, nullptr, nullptr, &t->p))) return rc;
    }
    return CCALIB_OK;
}

// After a trial that is expected to be rejected (it got a residual pass only): the driver will then raise the damping one decade
// and step again from the same Jacobian.  When the Schur sums for that damping are already here (they were taken
// along with the Jacobian pass), its step can be worked out now -- with the very function the driver uses -- and its
// kernels put into the stream behind the first trial's, so that the GPU does not idle through a host turn-around
// between the two.  The retry writes its poses to a buffer of its own; the driver's call is answered from it only
// when damping, step and trial point are bit for bit the ones prepared here.
// The plan: whether such a retry can be prepared now, and its step.
bool plan_retry_ahead(ccalib_solver* s, double lambda, const int* fidx, int nf, RetryPlan* plan, int* rc_out) {
    *rc_out = CCALIB_OK;
    const int lv = lambda_level(lambda), ns = s->ns();
    if (!s->retry_enabled || !s->speculate || chain_possible(s) || !s->com_valid || s->cb || lv >= 32 || nf > 42) return false;
    const double lambda2 = std::exp((lv + 1 - 16) * std::log(10.));
    if (!s->have_cached2 || s->cached_lambda2 != lambda2) return false;
    std::vector<double> Ssum((size_t)nf * nf), bsum(nf);
    if ((*rc_out = map_schur_sums(s, s->cachedS2, fidx, nf, Ssum.data(), bsum.data()))) return false;
    double prev[42];
    std::memcpy(prev, s->com_shared, sizeof(double) * ns);
    lm_trial_step(ns, nf, fidx, s->com_U.data(), s->com_gs.data(), Ssum.data(), bsum.data(), lambda2, s->run_solve_method, prev,
                  plan->shared, plan->dsh);
    be_fixup(s, plan->shared, prev);
    plan->lambda2 = lambda2;
    plan->nf = nf;
    return true;
}

// Puts the planned retry into the stream behind the trial just launched.
int launch_retry_ahead(ccalib_solver* s, double lambda, const int* fidx, int nf, const RetryPlan& plan, bool backsub_done) {
    ccalib_solver::Retry& r = s->retry;
    const int lv = lambda_level(lambda), ns = s->ns();
    std::memcpy(r.shared, plan.shared, sizeof(double) * ns);
    std::memcpy(r.dsh, plan.dsh, sizeof(double) * nf);
    // as be_step_err will see the damping history when the driver asks: this level rejected
    const bool was = s->level_accepted[lv];
    s->level_accepted[lv] = false;
    std::swap(s->d_pose_cur, s->d_pose_retry);
    std::swap(s->d_pc, s->d_pc_retry);
    int rc = launch_trial(s, plan.lambda2, lambda, fidx, nf, r.dsh, r.shared, s->steps_since_jac + 1, &r.t, nullptr, backsub_done);
    std::swap(s->d_pose_cur, s->d_pose_retry);
    std::swap(s->d_pc, s->d_pc_retry);
    s->level_accepted[lv] = was;
    if (rc) return rc;
    r.lambda = plan.lambda2, r.lambda_next = lambda, r.nf = nf, r.cur = s->cur;
    r.inflight = true;
    s->n_ahead++;
    return CCALIB_OK;
}

// When the device has already run this very step ahead of the host (same damping, and the step the driver
// computed is bit for bit the one k_lm_step predicted), its sums are simply waited for.
int be_step_err(void* user, double lambda, double lambda_next, const int* fidx, int nf, const double* dshared,
                const double* shared_new, double* dnorm2, double* pnorm2, double* err) {
    ccalib_solver* s = (ccalib_solver*)user;
    s->t_enter = std::chrono::steady_clock::now();
    (void)hipGetLastError();  // an error another HIP user of this thread left behind is not ours to report
    int rc;
    const int cur = s->cur, ns = s->ns();
    if (s->steps_since_jac > 0) s->level_accepted[s->last_level] = false;  // a retry: the step before was rejected
    s->last_level = lambda_level(lambda);
    s->steps_since_jac++;
    s->spec_valid = false;
    s->have_cached = false;
    ccalib_solver::Trial t;
    bool launched = false;

    ccalib_solver::Chain& ch = s->chain;
    if (ch.inflight) {
        const bool match = s->pred_go && ch.lambda == lambda && ch.lambda_next == lambda_next && ch.blk_base == cur &&
                           ch.pose_base == s->d_pose_prev && ch.pose_trial == s->d_pose_cur && nf == s->h_maps.nf &&
                           std::memcmp(dshared, s->pred_delta, sizeof(double) * nf) == 0 &&
                           std::memcmp(shared_new, s->pred_shared, sizeof(double) * ns) == 0;
        ch.inflight = false;
        if (match) {
            // the GPU is already on this step; queue the one after it, then wait for this one's sums
            t.p = ch.pend;
            t.predicted = ch.predicts;
            t.ahead = true;
            t.lam2 = 0.;
            t.n = 2 + jac_layout(s).end;
            store_shared(s, shared_new);
            s->pc_fresh = true;
            if (t.predicted && (rc = enqueue_chain(s, lambda_next, cur ^ 1, s->d_pose_cur, s->d_pose_prev))) return rc;
            launched = true;
            s->n_ahead_used++;
        } else {
            // called off on the device (its go word is 0: nothing ran) or not the step the driver took (it runs to
            // its end in stream order, into trial buffers that the launches below overwrite)
            s->pc_fresh = false;
            s->n_ahead_dropped++;
        }
    }
    ccalib_solver::Retry& r = s->retry;
    if (r.inflight) {
        r.inflight = false;
        const bool ahead_now = trial_runs_ahead(s, lambda, s->steps_since_jac);
        const bool match = !launched && r.lambda == lambda && r.lambda_next == lambda_next && r.nf == nf && r.cur == cur &&
                           r.t.ahead == ahead_now && (!ahead_now || r.t.lam2 == retry_lambda(s, lambda_next)) &&
                           std::memcmp(dshared, r.dsh, sizeof(double) * nf) == 0 &&
                           std::memcmp(shared_new, r.shared, sizeof(double) * ns) == 0;
        if (match) {
            std::swap(s->d_pose_cur, s->d_pose_retry);  // its poses (and their records) are the trial poses now
            std::swap(s->d_pc, s->d_pc_retry);
            store_shared(s, shared_new);
            t = r.t;
            launched = true;
            s->n_ahead_used++;
        } else {
            s->pc_fresh = s->pc3_fresh = false;  // the pose records are the retry's
            s->n_ahead_dropped++;
        }
    }
    if (!launched) {
        // a trial that gets a residual pass only is expected to be rejected: its retry is planned first, so that (stereo)
        // one kernel can do the back-substitution of both
        RetryPlan plan;
        bool planned = false;
        if (!trial_runs_ahead(s, lambda, s->steps_since_jac)) {
            planned = plan_retry_ahead(s, lambda, fidx, nf, &plan, &rc);
            if (rc) return rc;
        }
        const bool dual = planned && s->ncams == 2 && s->dual_backsub;
        const bool aside = planned && !dual && s->aux && !s->comm;
        if (aside) {
            // the retry and the pass behind it first, on the main stream: they are the longer chain, and what the next
            // iteration waits for; the trial itself beside them (a sharded run keeps one stream: its all-reduces must
            // stay in one order on every rank)
            if ((rc = launch_retry_ahead(s, lambda, fidx, nf, plan, false))) return rc;
            const bool pc3_main = s->pc3_fresh;  // the main camera-1 buffers hold the retry's composition
            std::swap(s->stream, s->aux);
            std::swap(s->d_pose3, s->d_pose3_alt);
            std::swap(s->d_pc3, s->d_pc3_alt);
            s->d_counter += 32;
            s->pc3_fresh = false;
            rc = launch_trial(s, lambda, lambda_next, fidx, nf, dshared, shared_new, s->steps_since_jac, &t, nullptr);
            s->d_counter -= 32;
            std::swap(s->d_pc3, s->d_pc3_alt);
            std::swap(s->d_pose3, s->d_pose3_alt);
            std::swap(s->stream, s->aux);
            s->pc3_fresh = pc3_main;
            if (rc) return rc;
            if (t.ahead) return set_error(CCALIB_ERR_STATE, "a trial with a planned retry was given a Jacobian pass");
            s->n_aside++;
        } else {
            if ((rc = launch_trial(s, lambda, lambda_next, fidx, nf, dshared, shared_new, s->steps_since_jac, &t,
                                   dual ? &plan : nullptr)))
                return rc;
            if (!t.ahead && planned && (rc = launch_retry_ahead(s, lambda, fidx, nf, plan, dual))) return rc;
        }
        if (s->retry.inflight) store_shared(s, shared_new);  // the solver's parameters are this trial's until the driver moves on
    }
    if ((rc = wait_reduction(s, t.p, t.n))) return rc;
    const double* red = s->h_red[t.p.slot];
    if (t.ahead) {
        s->spec_U.resize((size_t)ns * ns);
        s->spec_gs.resize(ns);
        parse_jac(s, red + 2, s->spec_U.data(), s->spec_gs.data(), &s->spec_err, s->spec_S);
        read_prediction(s, t.p.slot, t.predicted);
        s->spec_lambda = lambda_next;
        s->spec_have2 = t.lam2 != 0.;
        if (t.lam2 != 0.) {
            const JacLayout L0 = jac_layout(s);
            s->spec_S2.assign(red + 2 + L0.end, red + 2 + L0.end + L0.nsum);
            s->spec_lambda2 = t.lam2;
        }
        std::memcpy(s->spec_shared, shared_new, sizeof(double) * ns);
        s->spec_valid = true;
        *err = s->spec_err;
    } else
        *err = red[2] + (s->ncams == 2 ? red[3] : 0.);
    *dnorm2 = red[0];
    *pnorm2 = red[1];
    s->last_err = *err;
    s->drv_err = *err;
    return CCALIB_OK;
}

int be_commit(void* user) {
    ccalib_solver* s = (ccalib_solver*)user;
    std::swap(s->d_pose_prev, s->d_pose_cur);  // the next step rewrites every trial pose
    // (the cached Schur sums belong to the committed Jacobian and stay valid)
    s->committed_err = s->drv_err;             // what the driver now compares trial points with
    s->commits++;
    return CCALIB_OK;
}

void be_fixup(void* user, double* shared, double* prev) {
    ccalib_solver* s = (ccalib_solver*)user;
    if (s->ncams == 1) {
        if (s->aspect[0] != 0) {
            shared[0] = shared[1] * s->aspect[0];
            prev[0] = prev[1] * s->aspect[0];
        }
        return;
    }
    if (s->fix_intrinsic) return;
    double* ip = shared + 6;
    double* pp = prev + 6;
    if (s->same_focal) {
        ip[18] = ip[0], ip[19] = ip[1];
        pp[18] = pp[0], pp[19] = pp[1];
    }
    for (int c = 0; c < 2; c++)
        if (s->aspect[c] != 0) {
            ip[18 * c] = ip[18 * c + 1] * s->aspect[c];
            pp[18 * c] = pp[18 * c + 1] * s->aspect[c];
        }
}

// Lanes per view.  Fewer lanes per view mean more points per lane, which amortises the per-view
// work (pose set-up, cross-lane reduction, block stores); more lanes per view mean more wavefronts.
// Pick the smallest G that still gives every SIMD a few wavefronts, within the LDS budget.
int choose_G(int nviews, int maxnp, bool shared, int board_np) {
    const size_t lds_cap = 64 * 1024;
    int g_env = 0;
    if (const char* e = std::getenv("CCALIB_G")) g_env = std::atoi(e);
    if (g_env >= 2 && g_env <= 64 && (g_env & (g_env - 1)) == 0 &&
        pass_lds_bytes(g_env, maxnp, shared, board_np) <= lds_cap)
        return g_env;
    int best = 0;
    for (int g = 2; g <= 64; g <<= 1) {
        if (pass_lds_bytes(g, maxnp, shared, board_np) > lds_cap) continue;
        best = g;
        long long waves = ((long long)nviews * g + 63) / 64;
        if (waves >= 2048 || g >= maxnp) break;  // 2 wavefronts on each of the 1024 SIMDs
    }
    return best;
}

}  // namespace

extern "C" {

int ccalib_solver_create(ccalib_solver** out, int device, int ncams, int nviews, const int* npts,
                         const float* obj_xyz, int shared_board, const float* img1_uv, const float* img2_uv) {
    if (!obj_xyz || !img1_uv || (ncams == 2 && !img2_uv)) return set_error(CCALIB_ERR_ARG, "ccalib_solver_create: bad arguments");
    return solver_create_views(out, device, ncams, nviews, npts, HostViews{obj_xyz, nullptr}, shared_board,
                               HostViews{img1_uv, nullptr}, HostViews{img2_uv, nullptr});
}

}  // extern "C"

namespace ccalib {

namespace {

// Blocks a destroyed solver hands back, kept for the next one: cv::calibrateCamera-shaped callers create and destroy
// a solver per call, and hipMalloc + hipFree of its two arenas, the pinned block and the stream cost about a
// millisecond per call (hipFree also waits for the whole device).  Device blocks are kept per device up to
// CCALIB_POOL_MB megabytes in total (default 1024, 0 = keep nothing) and reused when the request fills at least a
// quarter of the block; ccalib_release_cached_memory() returns everything to the driver.
struct BlockPool {
    enum Kind { kDevice, kPinned, kStream };
    struct Block {
        int device;
        Kind kind;
        void* p;
        size_t bytes;
    };
    std::mutex mu;
    std::vector<Block> free_blocks;
    size_t cached_device_bytes = 0;
    size_t cap() {
        static const size_t c = [] {
            const char* e = std::getenv("CCALIB_POOL_MB");
            return (size_t)(e ? std::max(0, std::atoi(e)) : 1024) << 20;
        }();
        return c;
    }
    void* take(int device, Kind kind, size_t bytes, size_t* got) {
        std::lock_guard<std::mutex> lock(mu);
        int best = -1;
        for (int i = 0; i < (int)free_blocks.size(); i++) {
            const Block& b = free_blocks[i];
            if (b.device != device || b.kind != kind || b.bytes < bytes || b.bytes / 4 > bytes) continue;
            if (best < 0 || b.bytes < free_blocks[best].bytes) best = i;
        }
        if (best < 0) return nullptr;
        Block b = free_blocks[best];
        free_blocks.erase(free_blocks.begin() + best);
        if (b.kind == kDevice) cached_device_bytes -= b.bytes;
        if (got) *got = b.bytes;
        return b.p;
    }
    static void release(const Block& b) {
        if (b.kind == kDevice) (void)hipFree(b.p);
        if (b.kind == kPinned) (void)hipHostFree(b.p);
        if (b.kind == kStream) (void)hipStreamDestroy((hipStream_t)b.p);
    }
    // the caller has made sure nothing on the device still uses the block
    void give(int device, Kind kind, void* p, size_t bytes) {
        if (!p) return;
        Block b{device, kind, p, bytes};
        {
            std::lock_guard<std::mutex> lock(mu);
            const bool fits = kind != kDevice || cached_device_bytes + bytes <= cap();
            if (cap() > 0 && fits && free_blocks.size() < 64) {
                free_blocks.push_back(b);
                if (kind == kDevice) cached_device_bytes += bytes;
                return;
            }
        }
        release(b);
    }
    size_t trim() {
        std::vector<Block> all;
        {
            std::lock_guard<std::mutex> lock(mu);
            all.swap(free_blocks);
            cached_device_bytes = 0;
        }
        size_t n = 0;
        int cur = -1;
        (void)hipGetDevice(&cur);
        for (const Block& b : all) {
            (void)hipSetDevice(b.device);
            release(b);
            if (b.kind == kDevice) n += b.bytes;
        }
        if (cur >= 0) (void)hipSetDevice(cur);
        return n;
    }
};
BlockPool g_pool;

int pool_device_alloc(int device, size_t bytes, char** out, size_t* got) {
    *got = bytes;
    if (void* p = g_pool.take(device, BlockPool::kDevice, bytes, got)) {
        *out = static_cast<char*>(p);
        return CCALIB_OK;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory && g_pool.trim() > 0) {  // the cache gives way to a live request
        (void)hipGetLastError();                          // (the failed attempt is not to be reported by a later check)
        e = hipMalloc(out, bytes);
    }
    HIP_TRY(e);
    return CCALIB_OK;
}

// Hands out 256-byte aligned pieces of one allocation; with a null base it only adds the sizes up.
struct Carver {
    char* base;
    size_t used = 0;
    template <class T>
    T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += (n * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

// Pinned staging for uploads, shared by every solver of the process (created on first use, kept): two buffers, so
// that the host fills one while the copy engine drains the other.
struct Staging {
    std::mutex mu;
    float* buf[2] = {nullptr, nullptr};
    static constexpr size_t kFloats = 2u << 20;  // 8 MB each
};
Staging g_staging;
// with g_staging.mu held: the two buffers exist (portable: usable from every device of the process)
int staging_ready() {
    for (int b = 0; b < 2; b++)
        if (!g_staging.buf[b]) HIP_TRY(hipHostMalloc(&g_staging.buf[b], sizeof(float) * Staging::kFloats, hipHostMallocPortable));
    return CCALIB_OK;
}

// Copies the points of all views -- `fpp` floats per point, view i's at src.ptrs[i] or, for one flat array, at
// src.flat + fpp * off[i] -- into device memory at dst, view after view.  The host side gathers chunk k + 1 into pinned
// memory on several threads while chunk k crosses the bus.
int upload_views(ccalib_solver* s, float* dst, const HostViews& src, int fpp, const std::vector<int>& off) {
    const int nviews = (int)off.size() - 1;
    const size_t total = (size_t)fpp * off[nviews];
    std::lock_guard<std::mutex> lock(g_staging.mu);
    if (int rc = staging_ready()) return rc;
    hipEvent_t ev[2] = {nullptr, nullptr};
    struct EvGuard {
        hipEvent_t* e;
        ~EvGuard() {
            for (int b = 0; b < 2; b++)
                if (e[b]) (void)hipEventDestroy(e[b]);
        }
    } guard{ev};
    for (int b = 0; b < 2; b++) HIP_TRY(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
    const int nthreads = std::max(1, std::min(16, omp_get_max_threads()));
    int v0 = 0;
    size_t f0 = 0;  // floats already sent (flat source: position in the flat array)
    for (int k = 0; f0 < total; k++) {
        const int b = k & 1;
        if (k >= 2) HIP_TRY(hipEventSynchronize(ev[b]));
        float* stage = g_staging.buf[b];
        size_t n = 0;
        if (src.flat) {
            n = std::min(Staging::kFloats, total - f0);
            const size_t piece = (n + nthreads - 1) / nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int t = 0; t < nthreads; t++) {
                const size_t a = (size_t)t * piece, e = std::min(n, a + piece);
                if (a < e) std::memcpy(stage + a, src.flat + f0 + a, sizeof(float) * (e - a));
            }
        } else {
            int v1 = v0;
            while (v1 < nviews && (size_t)fpp * (off[v1 + 1] - off[v0]) <= Staging::kFloats) v1++;
            if (v1 == v0) return set_error(CCALIB_ERR_ARG, "view %d has more points than one upload chunk holds", v0);
            n = (size_t)fpp * (off[v1] - off[v0]);
            const int base = off[v0];
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int v = v0; v < v1; v++)
                std::memcpy(stage + (size_t)fpp * (off[v] - base), src.ptrs[v], sizeof(float) * fpp * (off[v + 1] - off[v]));
            v0 = v1;
        }
        HIP_TRY(hipMemcpyAsync(dst + f0, stage, sizeof(float) * n, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(ev[b], s->stream));
        f0 += n;
    }
    // the staging buffers go back to the pool only when the copies out of them are done
    HIP_TRY(hipStreamSynchronize(s->stream));
    return CCALIB_OK;
}

}  // namespace

int solver_create_views(ccalib_solver** out, int device, int ncams, int nviews, const int* npts, HostViews obj, int shared_board,
                        HostViews img1, HostViews img2) {
    if (!out || !npts || (!obj.flat && !obj.ptrs) || (!img1.flat && !img1.ptrs) || nviews < 1 || (ncams != 1 && ncams != 2) ||
        (ncams == 2 && !img2.flat && !img2.ptrs))
        return set_error(CCALIB_ERR_ARG, "ccalib_solver_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(CCALIB_ERR_NO_DEVICE, "no HIP device is usable (this library has no CPU path)");
    if (device < 0 || device >= ndev) return set_error(CCALIB_ERR_ARG, "device %d out of range (%d present)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    ccalib_solver* s = new ccalib_solver();
    // everything allocated so far is released if any step below fails
    struct CreateGuard {
        ccalib_solver* s;
        ~CreateGuard() {
            if (s) ccalib_solver_destroy(s);
        }
    } guard{s};
    s->device = device, s->ncams = ncams, s->nviews = nviews, s->shared_board = shared_board != 0;
    std::vector<int> off(nviews + 1, 0);
    for (int i = 0; i < nviews; i++) {
        if (npts[i] < 4) return set_error(CCALIB_ERR_ARG, "view %d has %d points (at least 4 needed)", i, npts[i]);
        if (shared_board && npts[i] != npts[0])
            return set_error(CCALIB_ERR_ARG, "shared_board needs the same point count in every view");
        if (npts[i] > INT_MAX / 4 - off[i])  // point indices (and 3 x them, in floats of a view's run) are ints
            return set_error(CCALIB_ERR_ARG, "more than %d points in all: not supported", INT_MAX / 4);
        off[i + 1] = off[i] + npts[i];
        s->maxnp = std::max(s->maxnp, npts[i]);
    }
    s->total = off[nviews];
    // a shared board too large for LDS is replicated per view instead
    const float* board = obj.flat ? obj.flat : obj.ptrs[0];
    std::vector<const float*> same_board;
    if (shared_board && npts[0] > 1024) {
        same_board.assign(nviews, board);
        obj = HostViews{nullptr, same_board.data()};
        shared_board = 0;
        s->shared_board = false;
    }
    s->board_np = shared_board ? npts[0] : 0;
    s->G = choose_G(nviews, s->maxnp, s->shared_board, s->board_np);
    if (s->G) {
        s->lds_bytes = (unsigned)pass_lds_bytes(s->G, s->maxnp, s->shared_board, s->board_np);
    } else {  // not even one view per wavefront fits the stage: all 64 lanes on a view, points straight from HBM
        s->G = 64;
        s->staged = false;
        s->lds_bytes = (unsigned)(sizeof(float) * (s->shared_board ? 3 * (size_t)s->board_np : 0));
    }
    int views_per_block = pass_block_size() / s->G;
    s->grid_view = (nviews + views_per_block - 1) / views_per_block;
    s->grid_red = std::min(s->grid_view, 512);
    if (const char* e = std::getenv("CCALIB_GRID_RED")) s->grid_red = std::max(1, std::min(s->grid_view, std::atoi(e)));
    s->grid_fused = std::min((nviews + kTileViews - 1) / kTileViews, 768);  // tiles in flight: three per CU
    if (const char* e = std::getenv("CCALIB_GRID_FUSED"))
        s->grid_fused = std::max(1, std::min((nviews + kTileViews - 1) / kTileViews, std::atoi(e)));
    s->grid_fused = (s->grid_fused + 2) / 3 * 3;  // a workgroup may own three tiles; tiles past the last view do nothing
    // all cameras in one launch: the same number of tile slots in flight, shared between them
    s->grid_fused12 = std::min((nviews + kTileViews - 1) / kTileViews, 768 / ncams);
    s->grid_fused12 = (s->grid_fused12 + 2) / 3 * 3;
    if (const char* e = std::getenv("CCALIB_NO_TILE12")) s->use_tile12 = std::atoi(e) == 0;
    if (const char* e = std::getenv("CCALIB_NO_FUSED")) s->use_fused = std::atoi(e) == 0;
    s->grid_err = std::min((nviews + 3) / 4, 2048);  // eight wavefronts per SIMD
    if (const char* e = std::getenv("CCALIB_NO_AHEAD")) s->speculate = std::atoi(e) == 0;
    if (const char* e = std::getenv("CCALIB_WAIT_LIMIT_S")) s->wait_limit_s = std::max(1., std::atof(e));
    if (fused_geom(s->maxnp, s->shared_board, s->board_np).lds_bytes > 64 * 1024) s->use_fused = false;
    s->nblk64 = (nviews + 63) / 64;
    s->npad = (nviews + 63) / 64 * 64;
    if (void* st = g_pool.take(device, BlockPool::kStream, 0, nullptr))
        s->stream = (hipStream_t)st;
    else
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    // Side by side pays while the passes leave the GPU room: 10 000 stereo pairs of 54 points go from 0.117 to 0.104 ms per
    // iteration; at 100 000 pairs (0.463 against 0.466) and 100 000 mono views every CU is busy and nothing is gained.
    s->aux_enabled = (long long)s->total * ncams <= 3000000;
    if (const char* e = std::getenv("CCALIB_AUX_STREAM")) s->aux_enabled = std::atoi(e) != 0;
    if (s->aux_enabled) {
        if (void* st = g_pool.take(device, BlockPool::kStream, 0, nullptr))
            s->aux = (hipStream_t)st;
        else
            HIP_TRY(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking));
    }
    const size_t nobj = shared_board ? (size_t)3 * s->board_np : (size_t)3 * s->total;
    const size_t N = nviews;
    // one allocation, carved: first pass adds the sizes up, second hands the pieces out
    auto layout = [&](Carver& c) {
        s->d_obj = c.take<float>(nobj);
        s->d_off = c.take<int>(N + 1);
        for (int cam = 0; cam < ncams; cam++) s->d_img[cam] = c.take<float2>(s->total);
        s->d_pose_prev = c.take<double>(6 * N);
        s->d_pose_cur = c.take<double>(6 * N);
        s->d_pose_retry = c.take<double>(6 * N);
        s->d_pc = c.take<double>(kPoseConstStride * N);
        s->d_pc_retry = c.take<double>(kPoseConstStride * N);
        if (ncams == 2) {
            s->d_pose3 = c.take<double>(6 * N);
            s->d_pc3 = c.take<double>(kPoseConstStride * N);
            s->d_pose3_alt = c.take<double>(6 * N);
            s->d_pc3_alt = c.take<double>(kPoseConstStride * N);
            for (int b = 0; b < 2; b++) s->d_cmp[b] = c.take<double>((size_t)kCmpRows * s->npad);
            s->d_pose_init1 = c.take<double>(6 * N);
            s->d_sums_tab = c.take<short>(4 * (quad_nlr(12) + quad_ngram(12)));
            s->d_view_tab = c.take<short>(stereo_view_table(12, nullptr));
        }
        for (int q = 0; q < 2; q++) s->d_red[q] = c.take<double>(ccalib_solver::kRedCap);
        s->d_ctrl = c.take<LmCtrl>(2);
        s->d_maps = c.take<LmMaps>(1);
        s->d_counter = c.take<unsigned>(64);
        // initialisation (ccalib_solver_init_*)
        s->d_status = c.take<int>(N);
        s->d_parked_view = c.take<int>(N + 1);
        s->d_parked_state = c.take<double>((size_t)kParkedDoubles * N);
        s->d_init_part = c.take<double>(5 * (N + 1));
    };
    Carver sizing{nullptr};
    layout(sizing);
    int rc = pool_device_alloc(device, sizing.used, &s->arena, &s->arena_bytes);
    if (rc) return rc;
    Carver carve{s->arena};
    layout(carve);
    HIP_TRY(hipMemcpy(s->d_off, off.data(), sizeof(int) * (nviews + 1), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(s->d_pose_prev, 0, sizeof(double) * 6 * nviews, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_ctrl, 0, sizeof(LmCtrl) * 2, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_counter, 0, sizeof(unsigned) * 64, s->stream));
    if (shared_board)
        HIP_TRY(hipMemcpy(s->d_obj, board, sizeof(float) * nobj, hipMemcpyHostToDevice));
    else if ((rc = upload_views(s, s->d_obj, obj, 3, off)))
        return rc;
    const HostViews imgs[2] = {img1, img2};
    for (int c = 0; c < ncams; c++)
        if ((rc = upload_views(s, reinterpret_cast<float*>(s->d_img[c]), imgs[c], 2, off))) return rc;
    {   // both slots' sums and flags in one pinned block; coherent (uncached on the device side): the host reads
        // these while the stream that writes them is still running
        const size_t per = sizeof(double) * ccalib_solver::kRedCap + 256 + sizeof(double2) * ccalib_solver::kRedCap;
        s->pinned = static_cast<char*>(g_pool.take(device, BlockPool::kPinned, 2 * per, nullptr));
        if (!s->pinned) {
            HIP_TRY(hipHostMalloc(&s->pinned, 2 * per, hipHostMallocCoherent));
            std::memset(s->pinned, 0, 2 * per);  // no pair of words in it may differ by a sequence number by accident
        }
        s->pinned_bytes = 2 * per;
        for (int q = 0; q < 2; q++) {
            s->h_red[q] = reinterpret_cast<double*>(s->pinned + q * per);
            s->h_flag[q] = reinterpret_cast<unsigned long long*>(s->pinned + q * per + sizeof(double) * ccalib_solver::kRedCap);
            *s->h_flag[q] = 0;
            s->h_pair[q] = reinterpret_cast<double2*>(s->pinned + q * per + sizeof(double) * ccalib_solver::kRedCap + 256);
        }
    }
    std::memset(&s->h_maps, 0, sizeof(LmMaps));
    if (const char* e = std::getenv("CCALIB_CHAIN")) s->chain_enabled = std::atoi(e) != 0;
    if (const char* e = std::getenv("CCALIB_NO_RETRY_AHEAD")) s->retry_enabled = std::atoi(e) == 0;
    if (const char* e = std::getenv("CCALIB_DUAL_BACKSUB")) s->dual_backsub = std::atoi(e) != 0;
    if (const char* e = std::getenv("CCALIB_FOLD_COMPOSE")) s->fold_compose = std::atoi(e) != 0;
    if (const char* e = std::getenv("CCALIB_DIRECT_SUMS")) s->direct_sums = std::atoi(e) != 0;
    if (const char* e = std::getenv("CCALIB_SPLIT_PUBLISH")) s->split_publish = std::atoi(e) != 0;
    s->segs.nseg = 0;
    HIP_TRY(hipStreamSynchronize(s->stream));
    guard.s = nullptr;
    *out = s;
    return CCALIB_OK;
}

// The stream has been drained: the solver's blocks go to the pool (or back to the driver when the pool is full).
void solver_release_blocks(ccalib_solver* s) {
    g_pool.give(s->device, BlockPool::kDevice, s->arena, s->arena_bytes);
    g_pool.give(s->device, BlockPool::kDevice, s->arena2, s->arena2_bytes);
    g_pool.give(s->device, BlockPool::kPinned, s->pinned, s->pinned_bytes);
    g_pool.give(s->device, BlockPool::kStream, s->stream, 0);
    g_pool.give(s->device, BlockPool::kStream, s->aux, 0);
    s->arena = s->arena2 = s->pinned = nullptr;
    s->stream = s->aux = nullptr;
}

int solver_arena2(ccalib_solver* s, size_t bytes) {
    if (bytes <= s->arena2_bytes) return CCALIB_OK;
    HIP_TRY(hipStreamSynchronize(s->stream));
    g_pool.give(s->device, BlockPool::kDevice, s->arena2, s->arena2_bytes);
    s->arena2 = nullptr, s->arena2_bytes = 0;
    return pool_device_alloc(s->device, bytes, &s->arena2, &s->arena2_bytes);
}

// Device -> host through the pinned staging buffers: the copy engine fills one while host threads empty the other
// into the caller's (pageable) memory.
int download(ccalib_solver* s, void* dst, const void* src, size_t bytes) {
    if (bytes < (1u << 20)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        return CCALIB_OK;
    }
    std::lock_guard<std::mutex> lock(g_staging.mu);
    if (int rc = staging_ready()) return rc;
    const size_t chunk = std::min<size_t>(sizeof(float) * Staging::kFloats, 2u << 20);  // small pieces: the unpacking overlaps
    const int nthreads = std::max(1, std::min(8, omp_get_max_threads()));
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    const size_t per_buf = sizeof(float) * Staging::kFloats / chunk;  // pieces one staging buffer holds
    // all pieces are queued at once into the two buffers' slots (2 * per_buf pieces in flight at most)
    size_t done = 0;
    while (done < nchunks) {
        const size_t batch = std::min(nchunks - done, 2 * per_buf);
        for (size_t k = 0; k < batch; k++) {
            const size_t off = (done + k) * chunk, n = std::min(chunk, bytes - off);
            char* stage = reinterpret_cast<char*>(g_staging.buf[k / per_buf]) + (k % per_buf) * chunk;
            HIP_TRY(hipMemcpyAsync(stage, static_cast<const char*>(src) + off, n, hipMemcpyDeviceToHost, s->stream));
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
        const size_t off0 = done * chunk, nb = std::min(batch * chunk, bytes - off0);
        // the batch lies in the two buffers piece after piece; copy it out on several threads
        const size_t piece = (nb + nthreads - 1) / nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int t = 0; t < nthreads; t++) {
            size_t a = (size_t)t * piece, e = std::min(nb, a + piece);
            while (a < e) {
                const size_t k = a / chunk, in = a % chunk, n = std::min(e - a, chunk - in);
                const char* stage = reinterpret_cast<const char*>(g_staging.buf[k / per_buf]) + (k % per_buf) * chunk + in;
                std::memcpy(static_cast<char*>(dst) + off0 + a, stage, n);
                a += n;
            }
        }
        done += batch;
    }
    return CCALIB_OK;
}

// The refined poses as cv::calibrateCamera returns them: rvecs and tvecs apart, nviews x 3 each.  The r|t records
// come through the pinned staging buffers and are taken apart on several host threads on the way out.
int solver_get_poses_split(ccalib_solver* s, double* rvecs, double* tvecs) {
    if (!s || !s->started || !rvecs || !tvecs) return set_error(CCALIB_ERR_STATE, "poses requested before set_start");
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> lock(g_staging.mu);
    if (int rc = staging_ready()) return rc;
    const size_t per_buf = sizeof(float) * Staging::kFloats / (6 * sizeof(double));  // views one buffer holds
    [[maybe_unused]] const int nthreads = std::max(1, std::min(8, omp_get_max_threads()));
    for (size_t v0 = 0; v0 < (size_t)s->nviews; v0 += 2 * per_buf) {
        const size_t nv = std::min<size_t>(2 * per_buf, s->nviews - v0);
        for (int b = 0; b < 2; b++) {
            const size_t a = std::min(nv, b * per_buf), e = std::min(nv, (b + 1) * per_buf);
            if (a < e)
                HIP_TRY(hipMemcpyAsync(g_staging.buf[b], s->d_pose_cur + 6 * (v0 + a), sizeof(double) * 6 * (e - a),
                                       hipMemcpyDeviceToHost, s->stream));
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nv > 4096)
        for (long long i = 0; i < (long long)nv; i++) {
            const double* rec = reinterpret_cast<const double*>(g_staging.buf[(size_t)i / per_buf]) + 6 * ((size_t)i % per_buf);
            double* r = rvecs + 3 * (v0 + i);
            double* t = tvecs + 3 * (v0 + i);
            r[0] = rec[0], r[1] = rec[1], r[2] = rec[2], t[0] = rec[3], t[1] = rec[4], t[2] = rec[5];
        }
    }
    return CCALIB_OK;
}

}  // namespace ccalib

extern "C" {

size_t ccalib_release_cached_memory(void) { return ccalib::g_pool.trim(); }

void ccalib_solver_destroy(ccalib_solver* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->aux) (void)hipStreamSynchronize(s->aux);
    if (s->comm) ncclCommDestroy(s->comm);
    ccalib::solver_release_blocks(s);
    for (hipEvent_t e : s->prof_ev) (void)hipEventDestroy(e);
    delete s;
}

int ccalib_solver_set_start(ccalib_solver* s, const double* intr, const unsigned char* imask, const double* aspect,
                            int stereo_opts, const double* rt, const double* lr) {
    if (!s || !intr || !imask || (s->ncams == 2 && !lr)) return set_error(CCALIB_ERR_ARG, "set_start: bad arguments");
    if (!rt && !s->poses_ready[0])
        return set_error(CCALIB_ERR_STATE, "set_start: rt is NULL and ccalib_solver_init_poses has not run for camera 0");
    HIP_TRY(hipSetDevice(s->device));
    unsigned need = 0;
    bool asp = false, rat = false;
    for (int c = 0; c < s->ncams; c++) {
        std::memcpy(s->intr[c], intr + 18 * c, sizeof(double) * 18);
        std::memcpy(s->imask[c], imask + 18 * c, 18);
        s->aspect[c] = aspect ? aspect[c] : 0.;
        for (int q = 0; q < 18; q++) {
            if (q >= I_NMODEL && (imask[18 * c + q] || intr[18 * c + q] != 0.))
                return set_error(CCALIB_ERR_UNSUPPORTED, "thin-prism / tilted coefficients are not built into this library");
            if (q < I_NMODEL && imask[18 * c + q]) need |= 1u << q;
        }
        if (c > 0 && (s->aspect[c] != 0.) != asp)
            return set_error(CCALIB_ERR_ARG, "both cameras must agree on whether the aspect ratio is fixed");
        asp = s->aspect[c] != 0.;
        rat = rat || (need & 0xE00u) || intr[18 * c + 9] != 0. || intr[18 * c + 10] != 0. || intr[18 * c + 11] != 0.;
    }
    if (asp && (need & 1u)) return set_error(CCALIB_ERR_ARG, "fx cannot be free when the aspect ratio is fixed");
    // one column set for both cameras (the flag word applies to both, as in OpenCV)
    const ModelOps* m = find_model(need, asp, rat);
    if (!m) return set_error(CCALIB_ERR_UNSUPPORTED, "no kernel column set covers mask 0x%x", need);
    for (int c = 0; c < s->ncams; c++) {
        s->rational[c] = rat;
        s->ops[c] = m;
    }
    if (s->ncams == 2) {
        const int n = stereo_sums_table(m->NC, nullptr);
        if (!n) return set_error(CCALIB_ERR_UNSUPPORTED, "stereo column count %d is not built in", m->NC);
        std::vector<short> tab(n);
        stereo_sums_table(m->NC, tab.data());
        HIP_TRY(hipMemcpy(s->d_sums_tab, tab.data(), sizeof(short) * n, hipMemcpyHostToDevice));
        {
            std::vector<short> vt(stereo_view_table(m->NC, nullptr));
            if (!stereo_view_table(m->NC, vt.data())) return set_error(CCALIB_ERR_STATE, "stereo sum table is inconsistent");
            HIP_TRY(hipMemcpy(s->d_view_tab, vt.data(), sizeof(short) * vt.size(), hipMemcpyHostToDevice));
        }
        // Few views: four lanes per view fill the chip where one lane per view would leave most of it idle.
        // Many views: one lane per view issues a quarter of the instructions.
        const char* e = std::getenv("CCALIB_SUMS_QUAD");
        s->sums_quad = e ? std::atoi(e) != 0 : s->nblk64 < 512;
        s->sums_rows = s->sums_quad ? stereo_sums_rows(s->nviews) : s->nblk64;
    }
    {   // what depends on the column set: the per-view blocks (two buffers per camera) and the partial-result arenas
        const size_t need_part = part_capacity(s);
        auto layout2 = [&](Carver& c) {
            for (int cam = 0; cam < s->ncams; cam++)
                for (int b = 0; b < 2; b++) s->d_blk[cam][b] = c.take<double>((size_t)m->NBLK * s->npad);
            for (int q = 0; q < 2; q++) s->d_part[q] = c.take<double>(need_part);
        };
        Carver sizing{nullptr};
        layout2(sizing);
        if (sizing.used > s->arena2_bytes) {
            for (int cam = 0; cam < 2; cam++) s->d_blk[cam][0] = s->d_blk[cam][1] = nullptr;
            s->d_part[0] = s->d_part[1] = nullptr;
            int rc2 = ccalib::solver_arena2(s, sizing.used);
            if (rc2) return rc2;
        }
        Carver carve{s->arena2};
        layout2(carve);
        s->part_cap = need_part;
        s->part_used = 0;
        s->segs.nseg = 0;
    }
    s->same_focal = (stereo_opts & CCALIB_STEREO_SAME_FOCAL_LENGTH) != 0;
    s->fix_intrinsic = s->ncams == 2 && (stereo_opts & CCALIB_STEREO_FIX_INTRINSIC) != 0;
    if (s->fix_intrinsic)
        for (int q = 0; q < 36; q++)
            if (imask[q]) return set_error(CCALIB_ERR_ARG, "CCALIB_STEREO_FIX_INTRINSIC needs every intrinsic masked out");
    if (lr) std::memcpy(s->lr, lr, sizeof(double) * 6);
    if (rt) HIP_TRY(hipMemcpy(s->d_pose_cur, rt, sizeof(double) * 6 * s->nviews, hipMemcpyHostToDevice));
    s->poses_ready[0] = s->poses_ready[1] = false;  // the LM loop takes the pose buffers over
    int rc = build_step_maps(s);
    if (rc) return rc;
    s->chain.inflight = false;
    s->retry.inflight = false;
    s->com_valid = false;
    s->pred_go = false;
    s->started = true;
    s->pc_fresh = s->pc3_fresh = false;
    s->have_cached = false;
    s->have_cached2 = false;
    s->spec_valid = false;
    s->steps_since_jac = 0;
    s->prev_trials = 1;
    for (bool& b : s->level_accepted) b = true;
    return CCALIB_OK;
}

// ---- initialisation on the device -------------------------------------------------------------

static InitArgs init_args(const ccalib_solver* s, int cam) {
    InitArgs a;
    a.img = s->d_img[cam];
    a.obj = s->d_obj;
    a.off = s->d_off;
    a.nviews = s->nviews;
    a.shared_board = s->shared_board;
    a.board_np = s->board_np;
    a.maxnp = s->maxnp;
    a.PS = a.VPB = 0;
    a.lds_bytes = 0;
    return a;
}

// Sums n doubles over the ranks of a multi-GPU run (no-op for a single rank).
static int allreduce_host(ccalib_solver* s, double* buf, int n) {
    if (s->comm) {
        HIP_TRY(hipMemcpyAsync(s->d_red[0], buf, sizeof(double) * n, hipMemcpyHostToDevice, s->stream));
        NCCL_TRY(ncclAllReduce(s->d_red[0], s->d_red[0], n, ncclDouble, ncclSum, s->comm, s->stream));
        HIP_TRY(hipMemcpyAsync(buf, s->d_red[0], sizeof(double) * n, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    } else if (s->cb) {
        int rc = s->cb(s->cb_user, buf, n);
        if (rc) return set_error(CCALIB_ERR_HIP, "all-reduce callback failed (%d)", rc);
    }
    return CCALIB_OK;
}

int ccalib_solver_init_intrinsics(ccalib_solver* s, int cam, int width, int height, double aspect, double* K) {
    if (!s || !K || cam < 0 || cam >= s->ncams) return set_error(CCALIB_ERR_ARG, "init_intrinsics: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    InitArgs a = init_args(s, cam);
    if (!init_geometry(a, false))
        return set_error(CCALIB_ERR_UNSUPPORTED, "a view with %d points does not fit the initialisation kernels' LDS staging", s->maxnp);
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
    A[2] = !width ? 0.5 : (width - 1) * 0.5;
    A[5] = !height ? 0.5 : (height - 1) * 0.5;
    const int grid = (s->nviews + a.VPB - 1) / a.VPB;
    if (grid > s->nviews + 1) return set_error(CCALIB_ERR_STATE, "init_intrinsics: more workgroups than views");
    double* d_part = s->d_init_part;
    launch_init_intrinsics(a, A[2], A[5], d_part, s->stream);
    std::vector<double> part((size_t)5 * grid);
    HIP_TRY(hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    double sums[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < grid; b++)
        for (int k = 0; k < 5; k++) sums[k] += part[(size_t)5 * b + k];
    int rc = allreduce_host(s, sums, 5);
    if (rc) return rc;
    // (1/fx^2, 1/fy^2) from the 2x2 normal equations, as cvInitIntrinsicParams2D's cvSolve(DECOMP_NORMAL | DECOMP_SVD)
    std::vector<double> N = {sums[0], sums[1], sums[1], sums[2]}, rhs = {sums[3], sums[4]};
    solve_sym_svd(N, rhs, 2);
    A[0] = std::sqrt(std::fabs(1. / rhs[0]));
    A[4] = std::sqrt(std::fabs(1. / rhs[1]));
    if (aspect != 0) {
        double tf = (A[0] + A[4]) / (aspect + 1.);
        A[0] = aspect * tf, A[4] = tf;
    }
    std::memcpy(K, A, sizeof(A));
    return CCALIB_OK;
}

int ccalib_solver_init_poses(ccalib_solver* s, int cam, const double* K, const double* dist14, double* rt_out) {
    if (!s || !K || cam < 0 || cam >= s->ncams) return set_error(CCALIB_ERR_ARG, "init_poses: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    double intr[18] = {K[0], K[4], K[2], K[5]};
    if (dist14)
        for (int i = 0; i < 14; i++) {
            if (i >= 8 && dist14[i] != 0.)
                return set_error(CCALIB_ERR_UNSUPPORTED, "thin-prism / tilted coefficients are not built into this library");
            intr[4 + i] = dist14[i];
        }
    CamParams cp = cam_params(intr, 0.);
    const bool rational = intr[9] != 0. || intr[10] != 0. || intr[11] != 0.;
    InitArgs a = init_args(s, cam);
    if (!init_geometry(a, true))
        return set_error(CCALIB_ERR_UNSUPPORTED, "a view with %d points does not fit the initialisation kernels' LDS staging", s->maxnp);
    double* dst = cam == 1 ? s->d_pose_init1 : s->d_pose_cur;
    Parked parked{s->d_parked_view + s->nviews, s->d_parked_view, s->d_parked_state};
    (void)hipGetLastError();  // an error another HIP user of this thread left behind is not ours to report
    launch_init_pose(a, cp, rational, dst, s->d_status, parked, s->stream);
    HIP_TRY(hipGetLastError());
    std::vector<int> status(s->nviews);
    HIP_TRY(hipMemcpyAsync(status.data(), s->d_status, sizeof(int) * s->nviews, hipMemcpyDeviceToHost, s->stream));
    if (rt_out) HIP_TRY(hipMemcpyAsync(rt_out, dst, sizeof(double) * 6 * s->nviews, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    int first_bad = -1;
    for (int i = 0; i < s->nviews && first_bad < 0; i++)
        if (status[i] != 0) first_bad = i;
    // in a sharded run the decision is taken together: a rank that left here alone would leave the others waiting
    // in the next all-reduce
    double nbad = first_bad >= 0 ? 1. : 0.;
    int rc = allreduce_host(s, &nbad, 1);
    if (rc) return rc;
    if (first_bad >= 0)
        return set_error(CCALIB_ERR_UNSUPPORTED, "view %d: pose initialisation is built for planar calibration rigs only", first_bad);
    if (nbad > 0.)
        return set_error(CCALIB_ERR_UNSUPPORTED, "pose initialisation failed on another rank (planar calibration rigs only)");
    s->poses_ready[cam] = true;
    s->pc_fresh = false;
    return CCALIB_OK;
}

int ccalib_solver_init_relative_pose(ccalib_solver* s, double* lr) {
    if (!s || !lr || s->ncams != 2) return set_error(CCALIB_ERR_ARG, "init_relative_pose: stereo solver and an output needed");
    if (!s->poses_ready[0] || !s->poses_ready[1])
        return set_error(CCALIB_ERR_STATE, "init_relative_pose: ccalib_solver_init_poses must run for both cameras first");
    HIP_TRY(hipSetDevice(s->device));
    // six rows of npad doubles: the composition-Jacobian buffer is free until the first LM pass
    double* d_rel = s->d_cmp[0];
    static_assert(kCmpRows >= 6, "relative poses are staged in the composition-Jacobian buffer");
    launch_relative_pose(s->d_pose_cur, s->d_pose_init1, s->nviews, s->npad, d_rel, s->stream);
    std::vector<double> rel((size_t)6 * s->npad);
    HIP_TRY(hipMemcpyAsync(rel.data(), d_rel, sizeof(double) * rel.size(), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int n = s->nviews;
    if (!s->comm && !s->cb) {
        for (int a = 0; a < 6; a++) {  // per-component median, as cvStereoCalibrateImpl seeds om_LR, T_LR
            double* col = rel.data() + (size_t)a * s->npad;
            std::nth_element(col, col + n / 2, col + n);
            double hi = col[n / 2];
            lr[a] = n % 2 ? hi : (*std::max_element(col, col + n / 2) + hi) * 0.5;
        }
        return CCALIB_OK;
    }
    // Sharded run: the median is taken over the views of ALL ranks, exactly.  Doubles are mapped to 64-bit keys in the
    // same order; for every component the key of the wanted order statistic is found eight bits at a time: each round
    // asks all ranks how many of their values lie at or below each of the 256 candidates that extend the bits found
    // so far (one all-reduce of 12 x 256 counts per round, eight rounds).  Every rank ends with the same bits a single
    // solver over all views computes.
    auto to_key = [](double v) {
        unsigned long long b;
        std::memcpy(&b, &v, 8);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    };
    auto from_key = [](unsigned long long k) {
        unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        double v;
        std::memcpy(&v, &b, 8);
        return v;
    };
    std::vector<std::vector<unsigned long long>> keys(6, std::vector<unsigned long long>(n));
    for (int a = 0; a < 6; a++) {
        for (int i = 0; i < n; i++) keys[a][i] = to_key(rel[(size_t)a * s->npad + i]);
        std::sort(keys[a].begin(), keys[a].end());
    }
    double cnt = (double)n;
    int rc = allreduce_host(s, &cnt, 1);
    if (rc) return rc;
    const long long N = std::llround(cnt);
    // searches 0..5: order statistic N/2 of each component; 6..11: N/2 - 1 (used when N is even)
    unsigned long long prefix[12];
    long long want[12];
    for (int j = 0; j < 12; j++) prefix[j] = 0, want[j] = j < 6 ? N / 2 : std::max(0ll, N / 2 - 1);
    static_assert(12 * 256 <= ccalib_solver::kRedCap, "the counts of one round go through the reduction buffer");
    std::vector<double> c(12 * 256);
    for (int shift = 56; shift >= 0; shift -= 8) {
        const unsigned long long low = shift ? (1ull << shift) - 1 : 0ull;
        for (int j = 0; j < 12; j++) {
            const auto& k = keys[j % 6];
            for (int b = 0; b < 256; b++) {  // values at or below the largest key that starts with prefix | b
                const unsigned long long top = prefix[j] | ((unsigned long long)b << shift) | low;
                c[j * 256 + b] = (double)(std::upper_bound(k.begin(), k.end(), top) - k.begin());
            }
        }
        if ((rc = allreduce_host(s, c.data(), 12 * 256))) return rc;
        for (int j = 0; j < 12; j++) {
            int b = 0;
            while (b < 255 && std::llround(c[j * 256 + b]) < want[j] + 1) b++;
            prefix[j] |= (unsigned long long)b << shift;
        }
    }
    for (int a = 0; a < 6; a++) {
        const double vhi = from_key(prefix[a]), vlo = from_key(prefix[6 + a]);
        lr[a] = N % 2 ? vhi : (vlo + vhi) * 0.5;
    }
    return CCALIB_OK;
}

int ccalib_rccl_unique_id(char id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    std::memcpy(id, &u, 128);
    return CCALIB_OK;
}

int ccalib_solver_use_rccl(ccalib_solver* s, int nranks, int rank, const char unique_id[128]) {
    if (!s || nranks < 1 || rank < 0 || rank >= nranks) return set_error(CCALIB_ERR_ARG, "use_rccl: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    ncclUniqueId u;
    std::memcpy(&u, unique_id, 128);
    NCCL_TRY(ncclCommInitRank(&s->comm, nranks, u, rank));
    s->cb = nullptr;
    return CCALIB_OK;
}

int ccalib_solver_use_callback(ccalib_solver* s, ccalib_allreduce_fn fn, void* user) {
    if (!s) return set_error(CCALIB_ERR_ARG, "use_callback: null solver");
    s->cb = fn;
    s->cb_user = user;
    return CCALIB_OK;
}

// The ranks of a sharded run enqueue their collectives independently, each from its own copy of the host state
// machine: they pair up only if every rank takes the same decisions about what to run ahead, with the same column
// set and the same criteria.  One small all-reduce before the loop compares the settings that decide this (sum of
// 1, w, w^2 over ranks: all w equal iff n * sum w^2 == (sum w)^2, by Cauchy-Schwarz) and refuses a mismatch instead of hanging later.
static int check_rank_agreement(ccalib_solver* s, int term_type, int max_iter, int solve_method) {
    if (!s->comm && !s->cb) return CCALIB_OK;
    // two words below 2^14: their squares, summed over up to a thousand ranks and multiplied by the rank count, stay
    // exact in a double
    const double w1 = (s->retry_enabled ? 1 : 0) + (s->speculate ? 2 : 0) + (s->chain_enabled ? 4 : 0) + (s->use_fused ? 8 : 0) +
                      16. * (s->split_publish + 1) + 64. * (s->ncams - 1) + 128. * s->kcols() + 4096. * (s->fix_intrinsic ? 1 : 0) +
                      8192. * (s->same_focal ? 1 : 0);
    const double w2 = (term_type & 3) + 4. * (solve_method & 1) + 8. * std::min(std::max(max_iter, 0), 1000);
    double buf[5] = {1., w1, w1 * w1, w2, w2 * w2};
    int rc = allreduce_host(s, buf, 5);
    if (rc) return rc;
    if (buf[0] * buf[2] != buf[1] * buf[1] || buf[0] * buf[4] != buf[3] * buf[3])
        return set_error(CCALIB_ERR_STATE, "the ranks of this sharded run do not agree on their settings (environment switches such as "
                                           "CCALIB_NO_AHEAD / CCALIB_NO_RETRY_AHEAD / CCALIB_CHAIN, the free-parameter set, the solver or the "
                                           "criteria): every rank must be configured alike, or their all-reduces do not pair up");
    return CCALIB_OK;
}

int ccalib_solver_run(ccalib_solver* s, int term_type, int max_iter, double eps, int solve_method, ccalib_stats* stats) {
    if (!s || !s->started) return set_error(CCALIB_ERR_STATE, "ccalib_solver_run before ccalib_solver_set_start");
    HIP_TRY(hipSetDevice(s->device));
    {
        int rc0 = check_rank_agreement(s, term_type, max_iter, solve_method);
        if (rc0) return rc0;
    }
    ccalib_backend be;
    be.ns = s->ns();
    be.jac = be_jac, be.reduce = be_reduce, be.step_err = be_step_err, be.commit = be_commit, be.fixup = be_fixup;
    double shared[42];
    unsigned char smask[42];
    current_shared(s, shared);
    if (s->ncams == 1)
        std::memcpy(smask, s->imask[0], 18);
    else {
        std::memset(smask, 1, 6);
        if (!s->fix_intrinsic) {
            std::memcpy(smask + 6, s->imask[0], 18);
            std::memcpy(smask + 24, s->imask[1], 18);
        }
    }
    s->t_wait = s->t_enqueue = s->t_think = s->t_launch1 = 0;
    s->t_wait_end = std::chrono::steady_clock::now();
    // what the run-ahead needs to know of the driver's plans: its solver and its iteration limit (CvLevMarq::init)
    s->run_solve_method = solve_method;
    s->run_max_iter = (term_type & CCALIB_TERM_COUNT) ? std::min(std::max(max_iter, 1), 1000) : 30;
    s->commits = 0;
    s->run_first_jac = true;
    s->chain.inflight = false;
    s->retry.inflight = false;
    s->com_valid = false;
    s->pred_go = false;
    int rc = ccalib_lm_run(&be, s, shared, smask, term_type, max_iter, eps, solve_method, stats);
    store_shared(s, shared);
    if (s->chain.inflight || s->retry.inflight) {  // the driver stopped with work still running ahead of it
        s->chain.inflight = s->retry.inflight = false;
        s->pc_fresh = s->pc3_fresh = false;
        s->n_ahead_dropped++;
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    if (std::getenv("CCALIB_TIMING") && stats)
        std::fprintf(stderr,
                     "[ccalib] run: total %.3f ms, enqueue %.3f ms, polling %.3f ms, other host %.3f ms; wait end -> first launch of "
                     "a trial %.3f ms, that launch %.3f ms (%d trials)\n",
                     stats->device_ms, s->t_enqueue * 1e3, s->t_wait * 1e3, stats->device_ms - (s->t_enqueue + s->t_wait) * 1e3,
                     s->t_think * 1e3, s->t_launch1 * 1e3, stats->err_passes);
    return rc;
}

int ccalib_solver_get(ccalib_solver* s, double* intr, double* rt, double* lr, double* err) {
    if (!s || !s->started) return set_error(CCALIB_ERR_STATE, "ccalib_solver_get before set_start");
    HIP_TRY(hipSetDevice(s->device));
    if (intr)
        for (int c = 0; c < s->ncams; c++) std::memcpy(intr + 18 * c, s->intr[c], sizeof(double) * 18);
    if (rt) {
        int rc = ccalib::download(s, rt, s->d_pose_cur, sizeof(double) * 6 * s->nviews);
        if (rc) return rc;
    }
    if (lr) std::memcpy(lr, s->lr, sizeof(double) * 6);
    if (err) *err = s->last_err;
    return CCALIB_OK;
}

int ccalib_solver_counters(ccalib_solver* s, long long out[3]) {
    if (!s || !out) return set_error(CCALIB_ERR_ARG, "counters: bad arguments");
    out[0] = s->n_ahead, out[1] = s->n_ahead_used, out[2] = s->n_ahead_dropped;
    return CCALIB_OK;
}

int ccalib_solver_executed_passes(ccalib_solver* s, long long out[2]) {
    if (!s || !out) return set_error(CCALIB_ERR_ARG, "executed_passes: bad arguments");
    out[0] = s->n_jac_exec, out[1] = s->n_err_exec;
    return CCALIB_OK;
}

int ccalib_solver_describe(ccalib_solver* s, int out[8]) {
    if (!s || !s->started || !out) return set_error(CCALIB_ERR_STATE, "describe: a started solver and an output needed");
    for (int i = 0; i < 8; i++) out[i] = 0;
    const bool fused = s->ops[0]->jac_fused && s->use_fused;
    const bool t12 = tile12_possible(s);
    out[0] = t12 ? 2 : fused ? 1 : 0;
    out[1] = t12 ? s->ncams : 1;
    out[2] = t12 ? s->grid_fused12 : fused ? s->grid_fused : s->grid_view;
    out[3] = s->ops[0]->NC;
    out[4] = s->ncams;
    out[5] = s->sums_quad ? 1 : 0;
    out[6] = s->aux ? 1 : 0;
    out[7] = (int)std::min<long long>(s->n_aside, 0x7fffffff);
    return CCALIB_OK;
}

int ccalib_solver_profile_jac(ccalib_solver* s, int on, double* mean_ms, int* launches) {
    if (!s) return set_error(CCALIB_ERR_ARG, "profile_jac: null solver");
    HIP_TRY(hipSetDevice(s->device));
    if (mean_ms || launches) {  // read what was recorded since profiling was switched on
        HIP_TRY(hipStreamSynchronize(s->stream));
        double tot = 0;
        int n = 0;
        for (size_t i = 0; i + 1 < s->prof_used; i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, s->prof_ev[i], s->prof_ev[i + 1]) == hipSuccess) tot += t, n++;
        }
        if (mean_ms) *mean_ms = n ? tot / n : 0.;
        if (launches) *launches = n;
    }
    s->prof_on = on != 0;
    s->prof_used = 0;
    return CCALIB_OK;
}

int ccalib_solver_num_shared(ccalib_solver* s, int* ns) {
    if (!s || !ns) return set_error(CCALIB_ERR_ARG, "num_shared: bad arguments");
    *ns = s->ns();
    return CCALIB_OK;
}

int ccalib_solver_eval_error(ccalib_solver* s, double* err) {
    if (!s || !s->started || !err) return set_error(CCALIB_ERR_STATE, "eval_error before set_start");
    HIP_TRY(hipSetDevice(s->device));
    (void)hipGetLastError();  // an error another HIP user of this thread left behind is not ours to report
    s->t_enter = std::chrono::steady_clock::now();
    int rc;
    s->n_err_exec++;
    if ((rc = stereo_compose(s, false, s->cur, s->d_pose_cur))) return rc;
    for (int c = 0; c < s->ncams; c++) {
        CamParams cp = cam_params(s->intr[c], s->aspect[c]);
        if (s->aspect[c] != 0) cp.fx = cp.fy * s->aspect[c];
        PART_TAKE(epart, s, (size_t)s->grid_err);
        enqueue_err(s, c, cp, epart);
        if ((rc = queue_colsum(s, epart, s->grid_err, 1, c))) return rc;
    }
    const double* red;
    if ((rc = finish_reduction(s, s->ncams, &red))) return rc;
    *err = red[0] + (s->ncams == 2 ? red[1] : 0.);
    return CCALIB_OK;
}

int ccalib_solver_eval_jacobian(ccalib_solver* s, double* U, double* gs, double* err) {
    if (!s || !s->started || !U || !gs || !err) return set_error(CCALIB_ERR_STATE, "eval_jacobian before set_start");
    HIP_TRY(hipSetDevice(s->device));
    double shared[42], prev[42] = {0};
    current_shared(s, shared);
    be_fixup(s, shared, prev);
    return be_jac(s, shared, 1e-3, U, gs, err);
}

int ccalib_solver_fetch_blocks(ccalib_solver* s, double* V, double* W, double* ge) {
    if (!s || !s->started || !V || !W || !ge) return set_error(CCALIB_ERR_STATE, "fetch_blocks: a started solver and three outputs needed");
    HIP_TRY(hipSetDevice(s->device));
    const ModelOps* m = s->ops[0];
    if (s->ncams == 2) {
        // the view systems as the stereo elimination kernels assemble them from both cameras' blocks
        const int kc = s->kcols(), ns = 42;
        const size_t n = s->nviews;
        double* d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(double) * n * (36 + 6 + (size_t)kc * 6)));
        double *dV = d, *dge = d + n * 36, *dW = d + n * 42;
        std::vector<double> hW(n * kc * 6);
        hipError_t e = hipSuccess;
        if (!launch_stereo_fetch(m->NC, s->d_blk[0][s->cur], s->d_blk[1][s->cur], s->d_cmp[s->cur], s->lr_jac[s->cur], s->npad,
                                 s->nviews, dV, dW, dge, s->stream))
            e = hipErrorInvalidValue;
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e == hipSuccess) e = hipMemcpy(V, dV, sizeof(double) * n * 36, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(ge, dge, sizeof(double) * n * 6, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hW.data(), dW, sizeof(double) * hW.size(), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        HIP_TRY(e);
        std::memset(W, 0, sizeof(double) * n * ns * 6);
        for (size_t v = 0; v < n; v++)
            for (int k = 0; k < kc; k++) {
                const int row = k < 6 ? k : k < 6 + m->NC ? 6 + m->col(k - 6) : 24 + s->ops[1]->col(k - 6 - m->NC);
                std::memcpy(W + (v * ns + row) * 6, hW.data() + (v * kc + k) * 6, 6 * sizeof(double));
            }
        return CCALIB_OK;
    }
    std::vector<double> h((size_t)m->NBLK * s->npad);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(h.data(), s->d_blk[0][s->cur], sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    const int ns = 18;
    for (int v = 0; v < s->nviews; v++) {
        int k = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++, k++) V[(size_t)v * 36 + a * 6 + b] = V[(size_t)v * 36 + b * 6 + a] = h[blk_index(m->NBLK, k, v)];
        for (int a = 0; a < 6; a++) ge[(size_t)v * 6 + a] = h[blk_index(m->NBLK, 21 + a, v)];
        for (int i = 0; i < ns * 6; i++) W[(size_t)v * ns * 6 + i] = 0;
        for (int i = 0; i < m->NC; i++)
            for (int b = 0; b < 6; b++) W[((size_t)v * ns + m->col(i)) * 6 + b] = h[blk_index(m->NBLK, 27 + 6 * i + b, v)];
    }
    return CCALIB_OK;
}

int ccalib_solver_reduce(ccalib_solver* s, double lambda, double* Ssum, double* bsum) {
    if (!s || !s->started) return set_error(CCALIB_ERR_STATE, "reduce before set_start");
    HIP_TRY(hipSetDevice(s->device));
    // free set = every kernel column, reported in shared-vector order
    int kc = s->kcols();
    std::vector<int> fidx;
    for (int i = 0; i < s->ns(); i++)
        for (int k = 0; k < kc; k++)
            if (s->kcol_to_shared(k) == i) fidx.push_back(i);
    s->have_cached = false;
    s->have_cached2 = false;
    s->spec_valid = false;
    return be_reduce(s, lambda, fidx.data(), (int)fidx.size(), Ssum, bsum);
}

int ccalib_solver_time_kernels(ccalib_solver* s, int reps, double ms[3]) {
    if (!s || !s->started || reps < 1 || !ms) return set_error(CCALIB_ERR_STATE, "time_kernels: bad state");
    HIP_TRY(hipSetDevice(s->device));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const ModelOps* m = s->ops[0];
    const bool fused = m->jac_fused && s->use_fused;
    const bool t12 = tile12_possible(s);
    s->part_used = 0;
    PART_TAKE(tpart, s, t12 ? (size_t)2 * s->grid_fused12 * m->NRED_U + s->grid_err
                            : (size_t)std::max(std::max(s->grid_red, s->grid_err), s->grid_fused * kTileViews) * m->NRED_U);
    s->part_used = 0;
    CamParams cp = cam_params(s->intr[0], s->aspect[0]);
    PassArgs a = pass_args(s, 0, s->d_pose_cur);
    if (fused) launch_pose_const(a.pose, s->nviews, s->d_pc, s->stream), s->pc_fresh = true;
    if (fused && t12 && s->ncams == 2) {  // camera 1's pose records for the merged launch
        int rc0 = stereo_compose(s, false, s->cur, s->d_pose_cur);
        if (rc0) return rc0;
    }
    for (int which = 0; which < 3; which++) {
        if (which == 0 && fused) {  // the fused kernel does the shared block as well
            ms[0] = 0.;
            continue;
        }
        auto launch = [&]() {
            if (which == 0)
                m->jac_shared(a, cp, s->grid_red, tpart, s->stream);
            else if (which == 1 && fused && t12) {
                Fused12Cams cams;
                for (int c = 0; c < 2; c++) {
                    const int cc = c < s->ncams ? c : 0;
                    cams.c[c] = Fused12Cam{cam_params(s->intr[cc], s->aspect[cc]), s->d_img[cc], cc == 0 ? s->d_pc : s->d_pc3,
                                           s->d_blk[cc][s->cur], tpart + (size_t)c * s->grid_fused12 * m->NRED_U};
                }
                launch_jac_fused12(m->mask, m->aspect, m->rational, a, cams, s->ncams, s->grid_fused12, s->stream);
            } else if (which == 1 && fused)
                m->jac_fused(a, cp, s->d_pc, s->grid_fused, s->d_blk[0][s->cur], tpart, s->stream, nullptr, 0);
            else if (which == 1)
                m->jac_view(a, cp, s->grid_view, s->d_blk[0][s->cur], s->npad, s->stream);
            else
                enqueue_err(s, 0, cp, tpart);
        };
        launch();  // warm-up
        HIP_TRY(hipEventRecord(e0, s->stream));
        for (int i = 0; i < reps; i++) launch();
        HIP_TRY(hipEventRecord(e1, s->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        ms[which] = t / reps;
    }
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
    s->have_cached = false;
    s->have_cached2 = false;
    s->spec_valid = false;
    return CCALIB_OK;
}

int ccalib_reprojection_error(int nviews, const int* npts, const float* obj_xyz, const float* img_uv, const double* K,
                              const double* dist, int ndist, const double* rvecs, const double* tvecs, float* proj_uv,
                              double* per_view_rms, double* total_rms) {
    if (nviews < 1 || !npts || !obj_xyz || !img_uv || !K || !rvecs || !tvecs || ndist < 0 || ndist > 14 || (ndist && !dist))
        return set_error(CCALIB_ERR_ARG, "reprojection_error: bad arguments");
    for (int i = 8; i < ndist; i++)
        if (dist[i] != 0.) return set_error(CCALIB_ERR_UNSUPPORTED, "thin-prism / tilted coefficients are not built into this library");
    ccalib_solver* s = nullptr;
    int rc = ccalib_solver_create(&s, 0, 1, nviews, npts, obj_xyz, 0, img_uv, nullptr);
    if (rc) return rc;
    struct Guard {
        ccalib_solver* s;
        float2* proj = nullptr;
        double* verr = nullptr;
        ~Guard() {
            (void)hipFree(proj), (void)hipFree(verr);
            ccalib_solver_destroy(s);
        }
    } g{s};
    std::vector<double> rt((size_t)6 * nviews);
    for (int i = 0; i < nviews; i++) {
        std::memcpy(&rt[6 * (size_t)i], rvecs + 3 * i, 3 * sizeof(double));
        std::memcpy(&rt[6 * (size_t)i + 3], tvecs + 3 * i, 3 * sizeof(double));
    }
    HIP_TRY(hipMemcpy(s->d_pose_cur, rt.data(), sizeof(double) * rt.size(), hipMemcpyHostToDevice));
    double intr[18] = {K[0], K[4], K[2], K[5]};
    for (int i = 0; i < ndist && i < 8; i++) intr[4 + i] = dist[i];
    CamParams cp = cam_params(intr, 0.);
    bool rational = intr[9] != 0. || intr[10] != 0. || intr[11] != 0.;
    if (proj_uv) HIP_TRY(hipMalloc(&g.proj, sizeof(float2) * s->total));
    HIP_TRY(hipMalloc(&g.verr, sizeof(double) * nviews));
    launch_reproj(pass_args(s, 0, s->d_pose_cur), cp, rational, s->grid_view, g.proj, g.verr, s->stream);
    std::vector<double> verr(nviews);
    HIP_TRY(hipMemcpyAsync(verr.data(), g.verr, sizeof(double) * nviews, hipMemcpyDeviceToHost, s->stream));
    if (proj_uv) HIP_TRY(hipMemcpyAsync(proj_uv, g.proj, sizeof(float2) * s->total, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    double tot = 0;
    for (int i = 0; i < nviews; i++) {
        tot += verr[i];
        if (per_view_rms) per_view_rms[i] = std::sqrt(verr[i] / npts[i]);
    }
    if (total_rms) *total_rms = std::sqrt(tot / s->total);
    return CCALIB_OK;
}

}  // extern "C"

// ---- reprojection error without any allocation ------------------------------------------------
// ccalib_reprojection_error_noalloc: the pass of ccalib_reprojection_error for callers that cannot allocate
// (code under HIP graph capture, serving loops).  Everything the pass needs beyond the caller's inputs and
// outputs -- the prefix offsets of the views, the poses in the (r|t) rows the kernel reads, the per-view squared
// errors, a validity word -- lives in one file-scope device array per GPU.
namespace {

constexpr int kNoAllocMaxViews = 65536;
// byte offsets inside the scratch array
constexpr size_t kNaOff = 0;                                                        // int[kNoAllocMaxViews + 1]
constexpr size_t kNaBad = (sizeof(int) * (kNoAllocMaxViews + 1) + 255) / 256 * 256; // int
constexpr size_t kNaPose = kNaBad + 256;                                            // double[kNoAllocMaxViews][6]
constexpr size_t kNaErr = kNaPose + sizeof(double) * 6 * kNoAllocMaxViews;          // double[kNoAllocMaxViews]
constexpr size_t kNoAllocScratchBytes = kNaErr + sizeof(double) * kNoAllocMaxViews; // about 3.9 MB

// Every GPU the library runs on holds its own copy of this array, at its own address, until the process ends.
// Host code never uses the name below as a pointer: scratch_for_current_device() asks the runtime for the array's
// device address once per device, and that address is what kernels, memsets and copies receive.
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};
std::atomic<bool> g_scratch_known[kMaxDevices];

int scratch_for_current_device(unsigned char** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return set_error(CCALIB_ERR_ARG, "device %d is beyond the %d this entry point tracks", dev, kMaxDevices);
    if (!g_scratch_known[dev].load(std::memory_order_acquire)) {
        unsigned char* p = nullptr;
        HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (!p) return set_error(CCALIB_ERR_HIP, "hipGetSymbolAddress returned no address for the scratch array");
        g_scratch_ptr[dev] = p;  // two threads racing here store the same address
        g_scratch_known[dev].store(true, std::memory_order_release);
    }
    *out = g_scratch_ptr[dev];
    return CCALIB_OK;
}

constexpr int kNaBlock = 1024;

// One workgroup: exclusive prefix sum of the point counts into off[0..nviews], the poses as (r|t) rows, and the
// validity word: a view with fewer than 4 or more than max_npts points, or a total beyond `capacity`, makes the
// whole table invalid (the later kernels then read nothing through it).
__global__ __launch_bounds__(kNaBlock) void k_na_prepare(const int* __restrict__ npts, int nviews, int max_npts, int capacity,
                                                         const double* __restrict__ rvecs, const double* __restrict__ tvecs,
                                                         int* off, int* bad, double* pose) {
    __shared__ long long part[kNaBlock];
    __shared__ int any_bad;
    const int t = threadIdx.x;
    if (t == 0) any_bad = 0;
    __syncthreads();
    const int per = (nviews + kNaBlock - 1) / kNaBlock;
    const int lo = min(t * per, nviews), hi = min(lo + per, nviews);
    long long sum = 0;
    bool mine_bad = false;
    for (int i = lo; i < hi; i++) {
        const int n = npts[i];
        mine_bad = mine_bad || n < 4 || n > max_npts;
        sum += n > 0 ? n : 0;
    }
    part[t] = sum;
    if (mine_bad) any_bad = 1;
    __syncthreads();
    // inclusive scan of the per-thread sums (Hillis-Steele; 10 rounds)
    for (int d = 1; d < kNaBlock; d <<= 1) {
        long long add = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    const long long total = part[kNaBlock - 1];
    const bool invalid = any_bad != 0 || total > (long long)capacity;
    long long run = t > 0 ? part[t - 1] : 0;
    for (int i = lo; i < hi; i++) {
        off[i] = invalid ? 0 : (int)run;
        run += npts[i] > 0 ? npts[i] : 0;
    }
    if (t == 0) {
        off[nviews] = invalid ? 0 : (int)total;
        *bad = invalid ? 1 : 0;
    }
    for (int i = t; i < 3 * nviews; i += kNaBlock) {
        const int v = i / 3, k = i - 3 * v;
        pose[6 * (size_t)v + k] = rvecs[i];
        pose[6 * (size_t)v + 3 + k] = tvecs[i];
    }
}

// One workgroup: per-view RMS and the global RMS = sqrt(sum of squared errors / sum of points), summed in a fixed
// order; NaN everywhere when the table was invalid.
__global__ __launch_bounds__(kNaBlock) void k_na_finish(const int* __restrict__ off, const int* __restrict__ bad,
                                                        const double* __restrict__ verr, int nviews, double* per_view_rms,
                                                        double* total_rms, float2* proj) {
    __shared__ double sm[kNaBlock];
    const int t = threadIdx.x;
    const bool invalid = *bad != 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // a caller that asked for projections only still learns that nothing was projected: the first pixel reads NaN
    if (invalid && proj && t == 0) proj[0] = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
    double s = 0;
    for (int i = t; i < nviews; i += kNaBlock) {
        const double e = invalid ? nan : verr[i];
        s += e;
        if (per_view_rms) per_view_rms[i] = invalid ? nan : sqrt(e / (double)(off[i + 1] - off[i]));
    }
    sm[t] = s;
    __syncthreads();
    for (int h = kNaBlock / 2; h > 0; h >>= 1) {
        if (t < h) sm[t] += sm[t + h];
        __syncthreads();
    }
    if (t == 0 && total_rms) *total_rms = invalid ? nan : sqrt(sm[0] / (double)off[nviews]);
}

}  // namespace

extern "C" int ccalib_reprojection_error_noalloc(int nviews, const int* d_npts, int max_npts, int capacity,
                                                 const float* d_obj_xyz, const float* d_img_uv, const double* K,
                                                 const double* dist, int ndist, const double* d_rvecs,
                                                 const double* d_tvecs, float* d_proj_uv, double* d_per_view_rms,
                                                 double* d_total_rms, void* stream) {
    if (nviews < 1 || !d_npts || !d_obj_xyz || !d_img_uv || !K || !d_rvecs || !d_tvecs || ndist < 0 || ndist > 14 ||
        (ndist && !dist) || max_npts < 4 || capacity < 4 || (!d_proj_uv && !d_per_view_rms && !d_total_rms))
        return set_error(CCALIB_ERR_ARG, "reprojection_error_noalloc: bad arguments");
    if (nviews > kNoAllocMaxViews)
        return set_error(CCALIB_ERR_ARG, "reprojection_error_noalloc: %d views exceed the %d the static scratch buffer holds",
                         nviews, kNoAllocMaxViews);
    if ((long long)max_npts > (long long)capacity)
        return set_error(CCALIB_ERR_ARG, "reprojection_error_noalloc: max_npts %d exceeds the capacity %d", max_npts, capacity);
    for (int i = 8; i < ndist; i++)
        if (dist[i] != 0.) return set_error(CCALIB_ERR_UNSUPPORTED, "thin-prism / tilted coefficients are not built into this library");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return set_error(CCALIB_ERR_NO_DEVICE, "no HIP device is usable (this library has no CPU path)");
    {   // Two HIP runtimes in one process (the caller's framework loaded after this library): the caller's device
        // pointers and stream then belong to the other copy and every launch below would fail with a bare "invalid
        // argument".  Found here instead: a device pointer this runtime does not know, while two copies are mapped.
        static thread_local int copies = -1;
        if (copies < 0) copies = hip_runtime_copies(nullptr, 0);
        if (copies > 1) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, d_img_uv) != hipSuccess) {
                (void)hipGetLastError();
                return set_error(CCALIB_ERR_STATE, "reprojection_error_noalloc: d_img_uv is not a device pointer of this library's HIP runtime%s",
                                 hip_runtime_note());
            }
        }
    }
    unsigned char* base = nullptr;
    int rc = scratch_for_current_device(&base);
    if (rc) return rc;
    int* off = reinterpret_cast<int*>(base + kNaOff);
    int* bad = reinterpret_cast<int*>(base + kNaBad);
    double* pose = reinterpret_cast<double*>(base + kNaPose);
    double* verr = reinterpret_cast<double*>(base + kNaErr);
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();  // an error another HIP user of this thread left behind is not ours to report

    // the same launch geometry ccalib_reprojection_error gets from its solver (views with their own object points)
    PassArgs a;
    a.img = reinterpret_cast<const float2*>(d_img_uv);
    a.obj = d_obj_xyz;
    a.off = off;
    a.pose = pose;
    a.nviews = nviews;
    a.shared_board = 0;
    a.board_np = 0;
    a.maxnp = max_npts;
    a.G = choose_G(nviews, max_npts, false, 0);
    a.staged = 1;
    if (a.G) {
        a.lds_bytes = (unsigned)pass_lds_bytes(a.G, max_npts, false, 0);
    } else {
        a.G = 64, a.staged = 0, a.lds_bytes = 0;
    }
    const int views_per_block = pass_block_size() / a.G;
    const int grid = (nviews + views_per_block - 1) / views_per_block;
    double intr[18] = {K[0], K[4], K[2], K[5]};
    for (int i = 0; i < ndist && i < 8; i++) intr[4 + i] = dist[i];
    const CamParams cp = cam_params(intr, 0.);
    const bool rational = intr[9] != 0. || intr[10] != 0. || intr[11] != 0.;

    hipLaunchKernelGGL(k_na_prepare, dim3(1), dim3(kNaBlock), 0, st, d_npts, nviews, max_npts, capacity, d_rvecs, d_tvecs, off,
                       bad, pose);
    launch_reproj(a, cp, rational, grid, reinterpret_cast<float2*>(d_proj_uv), verr, st, bad);
    hipLaunchKernelGGL(k_na_finish, dim3(1), dim3(kNaBlock), 0, st, off, bad, verr, nviews, d_per_view_rms, d_total_rms,
                       reinterpret_cast<float2*>(d_proj_uv));
    HIP_TRY(hipGetLastError());
    return CCALIB_OK;
}
