// kernels.hip -- the data-parallel passes of the calibration LM loop, hand-written for gfx950.
//
// Replaces the per-view / per-point loops of cvCalibrateCamera2Internal and
// cvStereoCalibrateImpl (OpenCV 4.1.0 calibration.cpp), which the reference reaches through
// monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317 and
// binocular_chessboard.cpp:152-162.
//
// Work decomposition (all FP64):
//   * k_jac_fused   the whole Jacobian pass of a camera for column sets of up to nine intrinsics:
//                   one wavefront per view, one lane per point; the normal-equation update is the
//                   product A^T A of the view's row matrix A = [Ji | Je | e], accumulated by
//                   v_mfma_f64_4x4x4_4b_f64 block by block from a per-wave LDS buffer (see the kernel's comment);
//   * the G-lane kernels: a view is owned by G consecutive lanes of one wavefront (G = 2..64, power
//     of two); lane g handles points g, g+G, ...; image points are read in the caller's own layout
//     (view-major float2), so the G lanes of a view read one contiguous run:
//       k_reproj          projected points and per-view squared error;
//       k_jac_shared      the shared (intrinsic) block JiT*Ji, JiT*e, eT*e, reduced once per workgroup;
//       k_jac_view        the per-view blocks JeT*Je, JeT*e, JiT*Je, butterfly-reduced over the G lanes
//     (the last two serve the rational column sets, whose columns do not fit one 16-wide tile);
//   * k_err_wave    residual sum: one wavefront per view, one lane per point, poses as records;
//   * k_schur / k_backsub  one lane per view: 6x6 Cholesky elimination of the view block;
//   * k_colsum      every column sum of a pass in one launch, written where the host polls.
// Reductions are fixed-order trees (no atomics on data), so results are bitwise reproducible.
#include "kernels.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_math.hpp"
#include "lm_step.hpp"
#include "lm_step_body.hpp"

namespace ccalib {

static constexpr int kBlock = 256;
static constexpr int kWaves = kBlock / 64;
#ifndef CCALIB_VIEW_MINWAVES
#define CCALIB_VIEW_MINWAVES 1  // wavefronts per SIMD the per-view-block kernel is compiled for
#endif

// Dynamic LDS of the per-point kernels (floats):
//   [shared board: 3 * board_np]  [wave 0 stage] [wave 1 stage] [wave 2 stage] [wave 3 stage]
// A wave stage holds the image points (float2) of the 64/G views the wavefront owns in the current
// round -- one contiguous run of the caller's view-major array, copied with coalesced loads -- and,
// when views have their own object points, those too (3 floats per point).
extern __shared__ float lds_dyn[];

// A view is owned by G consecutive lanes of one wavefront; a wavefront owns 64/G consecutive views
// per round and strides over rounds.
template <int G>
struct Lane {
    static constexpr int VPW = 64 / G;  // views per wavefront per round
    int lane, g;
    int v0;      // first view of this wavefront in the current round (wave-uniform)
    int v;       // this lane's view
    int stride;  // views per round over the whole grid
    float2* simg;  // this wavefront's stage
    float* sobj;
    const float* board;
    int base;      // global index of the first point of the current round (set by stage_views)
    bool staged;   // false: views too long for the stage -- points are read from global memory, which with all
                   // 64 lanes on one view (the only G such views get) is one contiguous run per load anyway
    __device__ __forceinline__ Lane(const PassArgs& a) {
        lane = threadIdx.x & 63;
        g = lane & (G - 1);
        int w = threadIdx.x >> 6;
        v0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kWaves + w) * VPW);
        v = v0 + lane / G;
        stride = gridDim.x * kWaves * VPW;
        int S = VPW * a.maxnp;
        float* p = lds_dyn + (a.shared_board ? 3 * a.board_np : 0);
        int per_wave = S * (a.shared_board ? 2 : 5);
        simg = reinterpret_cast<float2*>(p + w * per_wave);
        sobj = p + w * per_wave + 2 * S;
        board = lds_dyn;
        base = 0;
        staged = a.staged != 0;
    }
    __device__ __forceinline__ void next_round() {
        v0 += stride;
        v += stride;
    }
};

__device__ __forceinline__ void stage_board(const PassArgs& a) {
    if (a.shared_board) {
        for (int i = threadIdx.x; i < 3 * a.board_np; i += blockDim.x) lds_dyn[i] = a.obj[i];
        __syncthreads();
    }
}

// Copies the points of views [v0, v0 + VPW) into the wavefront's stage; returns the global index of
// the first staged point.  Only lanes of this wavefront touch the stage, so a wave barrier orders it.
template <int G>
__device__ __forceinline__ int stage_views(const PassArgs& a, Lane<G>& l) {
    int vA = l.v0 < a.nviews ? l.v0 : a.nviews;
    int vB = l.v0 + Lane<G>::VPW < a.nviews ? l.v0 + Lane<G>::VPW : a.nviews;
    int base = a.off[vA], n = a.off[vB] - base;
    l.base = base;
    if (!l.staged) return base;
    __builtin_amdgcn_wave_barrier();  // the previous round's reads are done before the stage is overwritten
    for (int i = l.lane; i < n; i += 64) l.simg[i] = a.img[base + i];
    if (!a.shared_board) {
        const float* src = a.obj + 3 * (size_t)base;
        for (int i = l.lane; i < 3 * n; i += 64) l.sobj[i] = src[i];
    }
    __builtin_amdgcn_wave_barrier();
    return base;
}

// What a lane needs to start on a view.
struct ViewHead {
    int first, np;
    double ps[6];
};
__device__ __forceinline__ ViewHead load_head(const PassArgs& a, int v) {
    ViewHead h;
    bool live = v < a.nviews;
    int vv = live ? v : 0;
    h.first = a.off[vv];
    h.np = live ? a.off[vv + 1] - h.first : 0;
    const double* ps = a.pose + 6 * (size_t)vv;
#pragma unroll
    for (int k = 0; k < 6; k++) h.ps[k] = ps[k];
    return h;
}

// Sums N per-lane values over the workgroup and writes them to out[0..N).  red: kWaves*N doubles.
template <int N>
__device__ __forceinline__ void block_reduce_store(double (&acc)[N], double* red, double* out) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) {
        double s = wave_sum(acc[k]);
        if (lane == 0) red[w * N + k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += kBlock) {
        double s = 0;
#pragma unroll
        for (int ww = 0; ww < kWaves; ww++) s += red[ww * N + k];
        out[k] = s;
    }
}

// Walks the points g, g+G, ... of one staged view; body(X, Y, Z, u, v) is called once per point.
// `local` is the view's first point relative to the stage.
template <int G, class Body>
__device__ __forceinline__ void for_points(const PassArgs& a, const Lane<G>& l, int local, int np, Body&& body) {
    const float* obj = a.shared_board ? l.board : l.staged ? l.sobj + 3 * local : a.obj + 3 * (size_t)(l.base + local);
    const float2* img = l.staged ? l.simg + local : a.img + (l.base + local);
    for (int p = l.g; p < np; p += G) {
        float2 m = img[p];
        float X = obj[3 * p], Y = obj[3 * p + 1], Z = obj[3 * p + 2];
        body((double)X, (double)Y, (double)Z, (double)m.x, (double)m.y);
    }
}

// ---- residual pass -------------------------------------------------------------------------
// Residual pass, one wavefront per view and one lane per point (64 points at a time), poses taken as the
// 16-double records the fused Jacobian kernel reads.  A point's arithmetic is one dependent chain, so what
// keeps the FP64 pipe issuing is the number of wavefronts per SIMD: with nothing staged in LDS but the shared
// board, eight fit (the G-lane layout below, with its per-wavefront stage, allows two).
template <bool RAT>
__device__ __forceinline__ void err_wave_body(const PassArgs& a, const CamParams& cp, const double* __restrict__ pc, double* part) {
    using M = Model<0u, false, RAT>;
    __shared__ double red[kWaves];
    stage_board(a);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nwaves = gridDim.x * kWaves;
    double acc[1] = {0.};
    int v = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + w);
    if (a.shared_board && a.board_np <= 64) {
        // Every view is the same board of at most 64 points: a lane's object point never changes and is read once.
        // A view is ~0.15 us of arithmetic and a memory round trip ~2 us, which eight wavefronts per SIMD taking one
        // view at a time do not cover (56 % of SIMD time in vector instructions, measured).  So a wavefront takes
        // four consecutive views at a time and asks for the next four while it works: one 512-byte load brings
        // their four pose records (lane l: double l%16 of record l/16), four more the image points.  The records
        // reach all lanes through a 512-byte LDS slot per wavefront (broadcast reads).  Nothing in the loop is a
        // scalar load, so no wait drains more than it needs.
        __shared__ double recs[kWaves][64];
        const int np = a.board_np;
        const bool on = lane < np;
        const float* ob = lds_dyn + 3 * (on ? lane : 0);
        const double X = ob[0], Y = ob[1], Z = ob[2];
        constexpr int GV = 4;  // views per trip
        const int ngroups = (a.nviews + GV - 1) / GV;
        struct Group {
            double rec;
            float2 m[GV];
        };
        auto fetch = [&](int g) {
            Group r;
            const int vr = min(GV * g + min(lane >> 4, GV - 1), a.nviews - 1);
            r.rec = pc[kPoseConstStride * (size_t)vr + (lane & 15)];
#pragma unroll
            for (int j = 0; j < GV; j++) {
                const int vj = min(GV * g + j, a.nviews - 1);
                r.m[j] = on ? a.img[(size_t)vj * np + lane] : make_float2(0.f, 0.f);
            }
            return r;
        };
        int g = v;  // one group per wavefront and trip
        Group h = {};
        if (g < ngroups) h = fetch(g);
        while (g < ngroups) {
            const int gn = g + nwaves;
            Group hn = {};
            if (gn < ngroups) hn = fetch(gn);
            __builtin_amdgcn_wave_barrier();
            recs[w][lane] = h.rec;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < GV; j++) {
                const double* q = recs[w] + 16 * j;
                const double ax = q[0], ay = q[1], az = q[2], c = q[3], c1 = q[4], sn = q[5], tx = q[10], ty = q[11], tz = q[12];
                if (on && GV * g + j < a.nviews) {
                    const double d = c1 * (ax * X + ay * Y + az * Z);
                    const double Xc = tx + c * X + d * ax + sn * (ay * Z - az * Y);
                    const double Yc = ty + c * Y + d * ay + sn * (az * X - ax * Z);
                    const double Zc = tz + c * Z + d * az + sn * (ax * Y - ay * X);
                    PointOut<M> o;
                    point_eval<M, false, false>(Xc, Yc, Zc, (double)h.m[j].x, (double)h.m[j].y, cp, o);
                    acc[0] += o.ex * o.ex + o.ey * o.ey;
                }
            }
            h = hn;
            g = gn;
        }
        v = a.nviews;
    }
    for (; v < a.nviews; v += nwaves) {
        // a shared board means equal point counts: no trip to the offset table before the points can be asked for
        const int first = a.shared_board ? v * a.board_np : a.off[v];
        const int np = a.shared_board ? a.board_np : a.off[v + 1] - first;
        const double* q = pc + kPoseConstStride * (size_t)v;  // wave-uniform: scalar loads
        const double ax = q[0], ay = q[1], az = q[2], c = q[3], c1 = q[4], sn = q[5], tx = q[10], ty = q[11], tz = q[12];
        for (int p0 = 0; p0 < np; p0 += 64) {
            const int p = p0 + lane;
            if (p < np) {
                const float2 m = a.img[first + p];
                const float* ob = a.shared_board ? lds_dyn + 3 * p : a.obj + 3 * (size_t)(first + p);
                const double X = ob[0], Y = ob[1], Z = ob[2];
                const double d = c1 * (ax * X + ay * Y + az * Z);
                const double Xc = tx + c * X + d * ax + sn * (ay * Z - az * Y);
                const double Yc = ty + c * Y + d * ay + sn * (az * X - ax * Z);
                const double Zc = tz + c * Z + d * az + sn * (ax * Y - ay * X);
                PointOut<M> o;
                point_eval<M, false, false>(Xc, Yc, Zc, (double)m.x, (double)m.y, cp, o);
                acc[0] += o.ex * o.ex + o.ey * o.ey;
            }
        }
    }
    block_reduce_store<1>(acc, red, part + blockIdx.x);
}

template <bool RAT>
__global__ __launch_bounds__(kBlock) void k_err_wave(PassArgs a, CamParams cp, const double* __restrict__ pc, double* part) {
    err_wave_body<RAT>(a, cp, pc, part);
}

// Both cameras of a stereo rig in one launch (blockIdx.y = camera): at ten thousand pairs a residual pass is a few
// microseconds of work, and a second launch costs as much again in the gap between the two.
template <bool RAT>
__global__ __launch_bounds__(kBlock) void k_err_wave_pair(ErrPair e) {
    if (blockIdx.y == 0)
        err_wave_body<RAT>(e.a[0], e.cp[0], e.pc[0], e.part[0]);
    else
        err_wave_body<RAT>(e.a[1], e.cp[1], e.pc[1], e.part[1]);
}

// ---- projectPoints + per-view squared error (the reference's RMS re-check, monocular_circle.cpp:5-29) ----
template <bool RAT, int G>
__global__ __launch_bounds__(kBlock) void k_reproj(PassArgs a, CamParams cp, float2* proj, double* view_err,
                                                   const int* __restrict__ bad) {
    using M = Model<0u, false, RAT>;
    // set by a preceding kernel that found the view table inconsistent: nothing may be read through it then
    if (bad && *bad) return;
    stage_board(a);
    Lane<G> l(a);
    for (; l.v0 < a.nviews; l.next_round()) {
        bool live = l.v < a.nviews;
        ViewHead h = load_head(a, l.v);
        int base = stage_views(a, l);
        double R[9], acc = 0.;
        rodrigues<false>(h.ps, R, nullptr);
        int p = l.g;
        for_points<G>(a, l, h.first - base, h.np, [&](double X, double Y, double Z, double u, double v) {
            double Xc = R[0] * X + R[1] * Y + R[2] * Z + h.ps[3];
            double Yc = R[3] * X + R[4] * Y + R[5] * Z + h.ps[4];
            double Zc = R[6] * X + R[7] * Y + R[8] * Z + h.ps[5];
            PointOut<M> o;
            point_eval<M, false, false>(Xc, Yc, Zc, 0., 0., cp, o);
            // projectPoints returns Point2f; the reference compares the rounded values
            float2 pr = make_float2((float)o.ex, (float)o.ey);
            if (proj) proj[h.first + p] = pr;
            p += G;
            double dx = u - (double)pr.x, dy = v - (double)pr.y;
            acc += dx * dx + dy * dy;
        });
#pragma unroll
        for (int msk = 1; msk < G; msk <<= 1) acc += __shfl_xor(acc, msk, 64);
        if (live && l.g == 0) view_err[l.v] = acc;
    }
}

// ---- shared (intrinsic) block ----------------------------------------------------------------
template <class M, int G>
__global__ __launch_bounds__(kBlock) void k_jac_shared(PassArgs a, CamParams cp, double* part) {
    constexpr int NC = M::NC, NU = M::NU, NR = M::NRED_U;
    __shared__ double red[kWaves * NR];
    stage_board(a);
    Lane<G> l(a);
    double acc[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) acc[k] = 0.;
    for (; l.v0 < a.nviews; l.next_round()) {
        ViewHead h = load_head(a, l.v);
        int base = stage_views(a, l);
        double R[9];
        rodrigues<false>(h.ps, R, nullptr);
        for_points<G>(a, l, h.first - base, h.np, [&](double X, double Y, double Z, double u, double v) {
            double Xc = R[0] * X + R[1] * Y + R[2] * Z + h.ps[3];
            double Yc = R[3] * X + R[4] * Y + R[5] * Z + h.ps[4];
            double Zc = R[6] * X + R[7] * Y + R[8] * Z + h.ps[5];
            PointOut<M> o;
            point_eval<M, true, false>(Xc, Yc, Zc, u, v, cp, o);
            // upper triangle of Ji^T Ji, skipping the structural zeros of the fx/fy/cx/cy columns
            static_for<0, NC>([&](auto ia) {
                constexpr int i = decltype(ia)::value;
                static_for<i, NC>([&](auto ib) {
                    constexpr int j = decltype(ib)::value;
                    constexpr int k = i * NC - i * (i - 1) / 2 + (j - i);
                    if constexpr (M::hasx(i) && M::hasx(j)) {
                        if constexpr (M::unitx(i) && M::unitx(j))
                            acc[k] += 1.;
                        else if constexpr (M::unitx(i))
                            acc[k] += o.jx[j];
                        else if constexpr (M::unitx(j))
                            acc[k] += o.jx[i];
                        else
                            acc[k] += o.jx[i] * o.jx[j];
                    }
                    if constexpr (M::hasy(i) && M::hasy(j)) {
                        if constexpr (M::unity(i) && M::unity(j))
                            acc[k] += 1.;
                        else if constexpr (M::unity(i))
                            acc[k] += o.jy[j];
                        else if constexpr (M::unity(j))
                            acc[k] += o.jy[i];
                        else
                            acc[k] += o.jy[i] * o.jy[j];
                    }
                });
                if constexpr (M::hasx(i)) {
                    if constexpr (M::unitx(i))
                        acc[NU + i] += o.ex;
                    else
                        acc[NU + i] += o.jx[i] * o.ex;
                }
                if constexpr (M::hasy(i)) {
                    if constexpr (M::unity(i))
                        acc[NU + i] += o.ey;
                    else
                        acc[NU + i] += o.jy[i] * o.ey;
                }
            });
            acc[NU + NC] += o.ex * o.ex + o.ey * o.ey;
        });
    }
    block_reduce_store<NR>(acc, red, part + (size_t)blockIdx.x * NR);
}

// ---- per-view blocks ---------------------------------------------------------------------------
// blk quantities (addressed through blk_index): 0..20 V upper triangle, 21..26 ge,
// 27 + 6*i + b  = W[i][b] (intrinsic column i, pose column b).
//
// The 21 + 6 + 6*NC accumulators fill most of the register file, so the pose is kept in its
// 13-double axis-angle form (PoseConst) and R X and the three dR/dr_j X are rebuilt per point
// (point_jacobian).
template <class M, int G>
__global__ __launch_bounds__(kBlock, CCALIB_VIEW_MINWAVES) void k_jac_view(PassArgs a, CamParams cp, double* blk, int npad) {
    constexpr int NC = M::NC, NB = M::NBLK;
    stage_board(a);
    Lane<G> l(a);
    for (; l.v0 < a.nviews; l.next_round()) {  // wave-uniform trip count (cross-lane shuffles inside)
        bool live = l.v < a.nviews;
        ViewHead h = load_head(a, l.v);
        int base = stage_views(a, l);
        const PoseConst q = pose_const(h.ps);
        double acc[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) acc[k] = 0.;
        for_points<G>(a, l, h.first - base, h.np, [&](double X, double Y, double Z, double u, double v) {
            PointOut<M> o;
            double jex[6], jey[6];
            point_jacobian<M>(q, cp, X, Y, Z, u, v, o, jex, jey);
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = i; j < 6; j++) acc[tri6(i, j)] += jex[i] * jex[j] + jey[i] * jey[j];
                acc[21 + i] += jex[i] * o.ex + jey[i] * o.ey;
            }
            static_for<0, NC>([&](auto ia) {
                constexpr int i = decltype(ia)::value;
#pragma unroll
                for (int b = 0; b < 6; b++) {
                    if constexpr (M::hasx(i)) {
                        if constexpr (M::unitx(i))
                            acc[27 + 6 * i + b] += jex[b];
                        else
                            acc[27 + 6 * i + b] += o.jx[i] * jex[b];
                    }
                    if constexpr (M::hasy(i)) {
                        if constexpr (M::unity(i))
                            acc[27 + 6 * i + b] += jey[b];
                        else
                            acc[27 + 6 * i + b] += o.jy[i] * jey[b];
                    }
                }
            });
        });
        // butterfly over the view's G lanes
#pragma unroll
        for (int m = 1; m < G; m <<= 1) {
#pragma unroll
            for (int k = 0; k < NB; k++) acc[k] += __shfl_xor(acc[k], m, 64);
        }
        if (live && l.g == 0) {
#pragma unroll
            for (int k = 0; k < NB; k++) blk[blk_index(NB, k, l.v)] = acc[k];
        }
    }
}

// ---- fused Jacobian pass on the FP64 matrix pipe ---------------------------------------------
// For column sets with NC + 7 <= 16 the whole normal-equation update of a view is one product:
// with A = [Ji | Je | e] (2P x 16: intrinsic columns, six pose columns, the residual as a column),
//   A^T A = [ U    W    gc  ]
//           [ W^T  V    ge  ]
//           [ gc^T ge^T err ]
// which the FP64 matrix pipe accumulates in 4x4 blocks, upper triangle only.  One wavefront owns one view
// at a time: lane p evaluates point p (residual, intrinsic columns, pose columns), writes its two
// rows of A into a per-wave LDS buffer stored column-major, and the wavefront then streams the
// buffer through the MFMAs.  Four wavefronts own one tile of kTileViews = 4 views and write that tile of
// blk contiguously (a workgroup holds one tile or three); the shared block is accumulated over the
// wavefront's views in registers and written once per wavefront (fixed assignment, so sums are bitwise
// reproducible).
//
// LDS (after the shared board): per wavefront 16 columns x RS doubles, RS >= 2*CH rows, spaced so that
// the operand reads of the matrix phase fall in disjoint banks (see fused_geom).

__global__ __launch_bounds__(64) void k_pose_const(const double* pose, int nviews, double* pc) {
    int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= nviews) return;
    double ps[6];
#pragma unroll
    for (int k = 0; k < 6; k++) ps[k] = pose[6 * (size_t)v + k];
    double rec[16];
    store_pose_const(ps, rec);
#pragma unroll
    for (int k = 0; k < 16; k++) pc[kPoseConstStride * (size_t)v + k] = rec[k];
}

// TPB tiles per workgroup (4 TPB wavefronts): the per-workgroup part of the LDS (board, shared columns) is
// paid once, so where three one-tile workgroups do not fit a CU one three-tile workgroup still does.
template <class M, bool SHARED, bool MULTI, int TPB>
__global__ __launch_bounds__(kBlock * TPB, TPB == 1 ? 3 : 1) void k_jac_fused(PassArgs a, CamParams cp, const int* __restrict__ off,
                                                         const double* __restrict__ pc, double* __restrict__ blk,
                                                         double* __restrict__ part, int RS, int CH,
                                                         const LmCtrl* __restrict__ ctrl, int cam) {
    constexpr int NC = M::NC, NB = M::NBLK, NU = M::NU, NR = M::NRED_U;
    if (ctrl) {  // an iteration run ahead of the host: camera parameters and the go word come from device memory
        if (!ctrl->go) return;
        cp = ctrl->cp[cam];
    }
    constexpr int CPOSE = NC, CRES = NC + 6, NCOL = NC + 7;
    static_assert(NCOL <= 16, "column set does not fit one 16x16 tile");
    static_assert(kWaves == kTileViews, "one wavefront per view of a tile");
    // the wavefront's number within the workgroup, as a scalar: tile numbers and the loop condition derived
    // from it must stay wave-uniform for the compiler, or the view loop turns into masked vector control flow
    const int lane = threadIdx.x & 63, wl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = wl & (kWaves - 1), tb = wl / kWaves;          // view of the tile, tile of the workgroup
    stage_board(a);
    // The principal-point columns are the same for every view -- (1, 0) and (0, 1) down the rows -- so
    // when the set has both they are kept once per workgroup instead of once per wavefront; with 14
    // instead of 16 private columns twelve wavefronts fit a CU.  Their rows never read as zero past
    // a view's last point, which only spoils the two sums of ones U(cx,cx), U(cy,cy): those are
    // replaced by the exact point count when the wavefront stores its shared-block sums.
    constexpr int ICX = M::index_of(I_CX), ICY = M::index_of(I_CY);
    constexpr bool CONSTC = ICX >= 0 && ICY == ICX + 1;
    constexpr int NPRIV = CONSTC ? 14 : 16;  // private columns per wavefront
    const int board_f = SHARED ? (3 * a.board_np + 3) / 4 * 4 : 0;  // floats, 16-byte granules
    double* const ldsd = reinterpret_cast<double*>(lds_dyn);  // indexed, so every access stays an LDS access
    const int cbase = board_f / 2;                             // the two shared columns (in doubles)
    const int WS = (NPRIV * RS + 31) & ~31;                          // doubles between the wavefronts' buffers
    const int jt = cbase + (CONSTC ? 2 * RS : 0) + wl * WS;         // this wavefront's buffer
    // after the column buffers: one 16-double slot per wavefront through which a view's pose record
    // reaches all lanes (a dword per lane in, broadcast reads out: no vector-ALU instruction involved,
    // which matters because nothing else issues on a SIMD while an FP64 MFMA executes)
    const int pslot = cbase + (CONSTC ? 2 * RS : 0) + TPB * kWaves * WS + wl * kPoseConstStride;
    auto slot = [&](int c) { return CONSTC && c > ICY ? c - 2 : c; };  // private column -> position in the buffer
    if constexpr (CONSTC) {
        for (int r = threadIdx.x; r < RS; r += kBlock * TPB) {
            ldsd[cbase + r] = (r & 1) ? 0. : 1.;
            ldsd[cbase + RS + r] = (r & 1) ? 1. : 0.;
        }
    }
    for (int c = NCOL; c < 16; c++)  // columns the set does not use stay zero
        for (int r = lane; r < RS; r += 64) ldsd[jt + slot(c) * RS + r] = 0.;
    __syncthreads();
    // The product is symmetric, so it is taken in 4x4 blocks with v_mfma_f64_4x4x4_4b_f64 (four independent
    // blocks per instruction; A/B lane = 16 k + 4 block + i|j, result lane = 16 i + 4 block + j) and the six
    // blocks below the diagonal are never computed.  With column groups g = 0..3 of four columns each, a lane
    // keeps three results: D = block (g, g), E = block (g, g+1 mod 4), F = block (g, g+2) for g < 2.
    const int col = lane & 15, rq = lane >> 4;
    const int ri = lane >> 4, rg = (lane >> 2) & 3, rj = lane & 3;
    int qv[3], qu[3];
    {
        const int p[3] = {4 * rg + ri, 4 * rg + ri, 4 * rg + ri};
        const int q[3] = {4 * rg + rj, 4 * ((rg + 1) & 3) + rj, 4 * ((rg + 2) & 3) + rj};
        const bool keep[3] = {ri <= rj, true, rg < 2};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int lo = p[r] < q[r] ? p[r] : q[r], hi = p[r] < q[r] ? q[r] : p[r];
            qv[r] = -1, qu[r] = -1;
            if (!keep[r]) continue;
            if (hi < NC)
                qu[r] = lo * NC - lo * (lo - 1) / 2 + (hi - lo);
            else if (hi < CRES) {
                if (lo < NC)
                    qv[r] = 27 + 6 * lo + (hi - CPOSE);
                else
                    qv[r] = tri6(lo - CPOSE, hi - CPOSE);
            } else if (hi == CRES) {
                if (lo < NC)
                    qu[r] = NU + lo;
                else if (lo < CRES)
                    qv[r] = 21 + (lo - CPOSE);
                else
                    qu[r] = NU + NC;
            }
        }
    }
    int voff[3];  // byte offset of each result inside a tile of blk
#pragma unroll
    for (int r = 0; r < 3; r++) voff[r] = qv[r] >= 0 ? (qv[r] * kTileViews + w) * (int)sizeof(double) : 0x7fffffff;
    double usum[3] = {0., 0., 0.};
    const int ntiles = (a.nviews + kTileViews - 1) / kTileViews;
    // What the wavefront needs to start on a view -- point range, this lane's image point, the pose --
    // is fetched one view ahead, between the point phase and the matrix phase of the view before,
    // so that its latency hides under the MFMAs.  The pose record (26 dwords) is fetched with ONE
    // dword load, lane k taking dword k, and handed to all lanes through the wavefront's LDS slot after
    // the matrix phase (spread_pose): a broadcast vector load of the record would cost 16 L1 cycles per
    // 16 bytes, scalar loads share their wait counter with the LDS reads of the matrix phase, and
    // v_readlane costs a vector-ALU slot per dword.
    int first = 0, np = 0;
    float2 mpt = make_float2(0.f, 0.f);
    float bX = 0.f, bY = 0.f, bZ = 0.f;  // this lane's object point of the first chunk
    unsigned qword = 0;
    auto fetch = [&](int v) {
        if constexpr (SHARED) {
            np = a.board_np, first = v * a.board_np;
        } else {
            first = off[v], np = off[v + 1] - first;
        }
        if (lane < np) {
            mpt = a.img[first + lane];
            if constexpr (!SHARED) {
                const float* ob = a.obj + 3 * (size_t)(first + lane);
                bX = ob[0], bY = ob[1], bZ = ob[2];
            }
        }
        const unsigned* qs = reinterpret_cast<const unsigned*>(pc + kPoseConstStride * (size_t)v);
        qword = qs[lane & 31];
    };
    // with a shared board lane p always works on board point p (of the first chunk)
    if constexpr (SHARED) {
        if (lane < a.board_np) bX = lds_dyn[3 * lane], bY = lds_dyn[3 * lane + 1], bZ = lds_dyn[3 * lane + 2];
    }
    int tile = blockIdx.x * TPB + tb;
    int v = __builtin_amdgcn_readfirstlane(tile * kTileViews + w);
    bool valid = tile < ntiles && v < a.nviews;  // wave-uniform; the loop holds no workgroup barrier
    if (valid) fetch(v);
    {   // Three stores that the range check drops.  Inside the loop the loads of a view are followed by
        // the three result stores of the view before; giving the first view the same number of younger
        // stores lets the wait before the point phase be "all but three" on both paths instead of "all".
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        auto none = __builtin_amdgcn_make_buffer_rsrc(blk, 0, 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < 3; r++) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, none, 0x7fff0000 + 64 * r + lane, 0, 0);
    }
    const double2* const lds2r = reinterpret_cast<const double2*>(lds_dyn);
    double2* const lds2 = reinterpret_cast<double2*>(lds_dyn);
    auto colbase = [&](int c) { return CONSTC && c == ICX ? cbase : CONSTC && c == ICY ? cbase + RS : jt + slot(c) * RS; };
    // operand addresses of the first eight rows (each later group of eight is 64 bytes further on):
    //   N  (16 bytes): column col,     rows 2rq, 2rq+1 -- the .x halves are one set of four rows, the .y the other
    //   N1 (16 bytes): column col + 4 (mod 16), same rows
    //   AR, BR (8 bytes): block (g, g+2) of the .x rows in the lanes of groups 0, 1, of the .y rows in groups 2, 3
    const int srcN = (colbase(col) >> 1) + rq, srcN1 = (colbase((col + 4) & 15) >> 1) + rq;
    const int srcA = col < 8 ? colbase(col) + 2 * rq : colbase(col - 8) + 2 * rq + 1;
    const int srcB = col < 8 ? colbase(col + 8) + 2 * rq : colbase(col) + 2 * rq + 1;
    const int dst = (jt >> 1) + lane;  // in double2 units; RS is even
    int dirty = CH;                    // rows of points [n, dirty) hold an earlier view's values
    int npsum = 0;                     // points this wavefront has seen
    // Point phase of one chunk of at most CH points: lane p < n evaluates its point and writes its two
    // rows of A; rows of an earlier, longer chunk are cleared.
    double qd[14];  // the current view's pose record, the same in every lane
    auto spread_pose = [&]() {
        unsigned* const slotw = reinterpret_cast<unsigned*>(lds_dyn) + 2 * pslot;
        if (lane < 32) slotw[lane] = qword;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 7; k++) {
            double2 t = lds2r[(pslot >> 1) + k];
            qd[2 * k] = t.x, qd[2 * k + 1] = t.y;
        }
    };
    auto point_phase = [&](int n, float2 m, float X, float Y, float Z) {
        if (lane < n) {
            PoseConst q;
            q.ax[0] = qd[0], q.ax[1] = qd[1], q.ax[2] = qd[2], q.c = qd[3], q.c1 = qd[4], q.s = qd[5];
            q.b1 = qd[6], q.b2 = qd[7], q.b3 = qd[8], q.b4 = qd[9], q.t[0] = qd[10], q.t[1] = qd[11], q.t[2] = qd[12];
            PointOut<M> o;
            double jex[6], jey[6];
            point_jacobian<M>(q, cp, (double)X, (double)Y, (double)Z, (double)m.x, (double)m.y, o, jex, jey);
            static_for<0, NC>([&](auto ic_) {
                constexpr int i = decltype(ic_)::value;
                if constexpr (!(CONSTC && (i == ICX || i == ICY)))
                    lds2[dst + ((slot(i) * RS) >> 1)] = make_double2(o.jx[i], o.jy[i]);
            });
#pragma unroll
            for (int b = 0; b < 6; b++) lds2[dst + ((slot(CPOSE + b) * RS) >> 1)] = make_double2(jex[b], jey[b]);
            lds2[dst + ((slot(CRES) * RS) >> 1)] = make_double2(o.ex, o.ey);
        }
        if (n < dirty) {  // rows past the chunk's last point must read as zero (wave-uniform branch)
            if (lane >= n && lane < dirty) {
#pragma unroll
                for (int c = 0; c < NCOL - (CONSTC ? 2 : 0); c++) lds2[dst + ((c * RS) >> 1)] = make_double2(0., 0.);
            }
        }
        dirty = n;
    };
    // Matrix phase: eight rows per group, five instructions on five accumulators (a chain of dependent
    // 4x4x4 MFMAs issues every 46 cycles, alternating chains every 19): diagonal blocks from (N, N), blocks
    // (g, g+1) from (N, N1), both once for the .x and once for the .y rows, and blocks (0,2), (1,3) of both
    // row sets in one instruction from (AR, BR).  The reads of a group are in flight while the group before
    // multiplies (they may run past the chunk's rows: what they fetch is then never used); two register sets
    // take turns and one address update serves two groups.  CH is a multiple of 8, so the rows up to the
    // next multiple of 8 exist and are zero.
    struct Acc {
        double d0, d1, e0, e1, f;
    };
    auto matrix_phase = [&](int n, Acc& acc, bool from_zero) {
        __builtin_amdgcn_wave_barrier();
        // The point phase is one long chain of dependent FP64 instructions, the matrix phase independent work
        // that fills any gap: the wavefront issues at low priority here and at high priority outside.
        __builtin_amdgcn_s_setprio(0);
        const int ngroups = (2 * n + 7) >> 3;  // >= 1
        auto mfma5 = [&](const double2& u, const double2& u1, double ar, double br) {
            acc.d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.x, u.x, acc.d0, 0, 0, 0);
            acc.e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.x, u1.x, acc.e0, 0, 0, 0);
            acc.f = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, br, acc.f, 0, 0, 0);
            acc.d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.y, u.y, acc.d1, 0, 0, 0);
            acc.e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.y, u1.y, acc.e1, 0, 0, 0);
        };
        // unrolled over the sixteen groups a chunk can have, so that every read is a base address plus an immediate
        // offset: no address arithmetic on the vector ALU inside the phase, only scalar compares and branches
        double2 pn = lds2r[srcN], pn1 = lds2r[srcN1];
        double pa = ldsd[srcA], pb = ldsd[srcB];
        if (from_zero) acc = Acc{0., 0., 0., 0., 0.};
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
            if (g >= ngroups) break;
            const double2 qn = lds2r[srcN + 4 * (g + 1)], qn1 = lds2r[srcN1 + 4 * (g + 1)];
            const double qa = ldsd[srcA + 8 * (g + 1)], qb = ldsd[srcB + 8 * (g + 1)];
            __builtin_amdgcn_sched_barrier(0);
            mfma5(pn, pn1, pa, pb);
            if (g + 1 >= ngroups) break;
            pn = lds2r[srcN + 4 * (g + 2)], pn1 = lds2r[srcN1 + 4 * (g + 2)];
            pa = ldsd[srcA + 8 * (g + 2)], pb = ldsd[srcB + 8 * (g + 2)];
            __builtin_amdgcn_sched_barrier(0);
            mfma5(qn, qn1, qa, qb);
        }
        __builtin_amdgcn_s_setprio(3);
        __builtin_amdgcn_wave_barrier();  // the reads are done before the buffer is overwritten
    };
    auto advance = [&]() {  // the next view's loads go out ahead of the matrix phase
        tile += gridDim.x * TPB;
        v = __builtin_amdgcn_readfirstlane(tile * kTileViews + w);
        valid = tile < ntiles && v < a.nviews;
        if (valid) fetch(v);
    };
    if (valid) spread_pose();
    while (valid) {
        const int tile_cur = tile, vfirst = first, vnp = np;
        npsum += vnp;
        Acc acc = {0., 0., 0., 0., 0.};
        if constexpr (!MULTI) {  // the launcher guarantees np <= CH <= 64
            point_phase(vnp, mpt, bX, bY, bZ);
            advance();
            matrix_phase(vnp, acc, true);
            if (valid) spread_pose();
        } else {
            for (int p0 = 0; p0 < vnp; p0 += CH) {
                const int n = vnp - p0 < CH ? vnp - p0 : CH;
                float2 m = mpt;
                float X = bX, Y = bY, Z = bZ;
                if (p0 > 0 && lane < n) {  // later chunks are not fetched ahead
                    m = a.img[vfirst + p0 + lane];
                    if constexpr (SHARED) {
                        X = lds_dyn[3 * (p0 + lane)], Y = lds_dyn[3 * (p0 + lane) + 1], Z = lds_dyn[3 * (p0 + lane) + 2];
                    } else {
                        const float* ob = a.obj + 3 * (size_t)(vfirst + p0 + lane);
                        X = ob[0], Y = ob[1], Z = ob[2];
                    }
                }
                point_phase(n, m, X, Y, Z);
                const bool last = p0 + CH >= vnp;
                if (last) advance();
                matrix_phase(n, acc, p0 == 0);
                if (last && valid) spread_pose();
            }
        }
        // the two row sets of D and E, and the two halves of F (lanes 8 apart in a row of 16), are summed here
        const double res[3] = {acc.d0 + acc.d1, acc.e0 + acc.e1, acc.f + dpp_move<0x128>(acc.f)};
#pragma unroll
        for (int r = 0; r < 3; r++) usum[r] += res[r];
        // Unconditional buffer stores: lanes whose result is not kept carry an offset
        // past the tile, which the range check drops.  (Stores under a branch would hide their
        // number from the wait-count bookkeeping and make the next view wait for all of them.)
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc(blk + (size_t)tile_cur * NB * kTileViews, 0,
                                                      NB * kTileViews * (int)sizeof(double), 0x00020000);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, res[r]), rsrc, voff[r], 0, 0);
        }
    }
    // The four wavefronts of a tile slot add their shared-block sums through LDS (each into its own, now idle,
    // column buffer) and the first of them writes one row of `part` per tile slot, in a fixed order.
    // a sum of ones (the principal-point columns against themselves): the point count
    const bool ones = CONSTC && ri == rj && (4 * rg + ri == ICX || 4 * rg + ri == ICY);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 3; r++) ldsd[jt + 64 * r + lane] = r == 0 && ones ? (double)npsum : usum[r];
    __syncthreads();
    if (w == 0) {
        double* po = part + (size_t)(blockIdx.x * TPB + tb) * NR;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            double s = 0;
#pragma unroll
            for (int ww = 0; ww < kWaves; ww++) s += ldsd[jt + ww * WS + 64 * r + lane];
            if (qu[r] >= 0) po[qu[r]] = s;
        }
    }
}

// ---- per-view elimination --------------------------------------------------------------------
// One lane per view, one wavefront per workgroup.  Output per workgroup: the upper triangle of
// sum W V'^-1 W^T over its 64 views (NC(NC+1)/2 values) followed by sum W V'^-1 ge (NC values).
template <int NC>
__global__ __launch_bounds__(64) void k_schur(const double* blk, int npad, int nviews, double lambda, double* part,
                                              const LmCtrl* __restrict__ ctrl, double lambda_y1, double* part_y1) {
    constexpr int NS = NC * (NC + 1) / 2 + NC, NB = 27 + 6 * NC;
    if (ctrl && !ctrl->go) return;
    // a launch for two dampings: the second half of the grid (blockIdx.y = 1) takes the other one.  Small problems
    // pay a launch and the gap in front of it (4.7 + 4.6 us at 20 views) instead of a few more workgroups.
    if (blockIdx.y) lambda = lambda_y1, part = part_y1;
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    double Vu[21], L[21], invd[6], y[6], Z[NC][6];
#pragma unroll
    for (int k = 0; k < 21; k++) Vu[k] = blk[blk_index(NB, k, vv)];
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = blk[blk_index(NB, 21 + k, vv)];
    // every load of the view goes out before the (long, serial) factorisation starts
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
        for (int b = 0; b < 6; b++) Z[i][b] = blk[blk_index(NB, 27 + 6 * i + b, vv)];
    __builtin_amdgcn_sched_barrier(0);  // ... and the scheduler is not to sink them back among it
    bool ok = chol6(Vu, lambda, L, invd) && live;
    chol6_fwd(L, invd, y);
#pragma unroll
    for (int i = 0; i < NC; i++) chol6_fwd(L, invd, Z[i]);  // Z_i = L^-1 W_i^T, so W V'^-1 W^T = Z Z^T
    __shared__ double sm[32 * 65];
    double acc[NS];
    int k = 0;
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
        for (int j = i; j < NC; j++) {
            double s = 0;
#pragma unroll
            for (int b = 0; b < 6; b++) s += Z[i][b] * Z[j][b];
            acc[k++] = ok ? s : 0.;
        }
#pragma unroll
    for (int i = 0; i < NC; i++) {
        double s = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) s += Z[i][b] * y[b];
        acc[k++] = ok ? s : 0.;
    }
    lanes_reduce_store<NS>(acc, sm, part + (size_t)blockIdx.x * NS);
}

// trial pose = committed pose - V'^-1 (ge - W^T dc).  part per workgroup: sum |delta|^2, sum |pose_prev|^2.
// With pc != null the trial pose is also written in the form the fused Jacobian kernel reads.
template <int NC>
__global__ __launch_bounds__(64) void k_backsub(const double* blk, int npad, int nviews, double lambda, DeltaC dc,
                                                const double* pose_prev, double* pose_cur, double* part, double* pc,
                                                const LmCtrl* __restrict__ ctrl) {
    constexpr int NB = 27 + 6 * NC;
    if (ctrl) {  // an iteration run ahead of the host: the step comes from device memory
        if (!ctrl->go) return;
#pragma unroll
        for (int i = 0; i < NC; i++) dc.v[i] = ctrl->dc[i];
    }
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    double Vu[21], L[21], invd[6], rhs[6];
#pragma unroll
    for (int k = 0; k < 21; k++) Vu[k] = blk[blk_index(NB, k, vv)];
#pragma unroll
    for (int k = 0; k < 6; k++) rhs[k] = blk[blk_index(NB, 21 + k, vv)];
    double Wv[NC][6];  // all of the view's loads first, one memory round trip for the kernel
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
        for (int b = 0; b < 6; b++) Wv[i][b] = blk[blk_index(NB, 27 + 6 * i + b, vv)];
    double prev_in[6];
#pragma unroll
    for (int k = 0; k < 6; k++) prev_in[k] = pose_prev[6 * (size_t)vv + k];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NC; i++) {
        double d = dc.v[i];
#pragma unroll
        for (int b = 0; b < 6; b++) rhs[b] -= Wv[i][b] * d;
    }
    bool ok = chol6(Vu, lambda, L, invd);
    chol6_fwd(L, invd, rhs);
    chol6_bwd(L, invd, rhs);
    // the wavefront's 64 poses (and pose records) are contiguous in memory: written as such
    __shared__ double sm[64 * 17];
    const int v0 = blockIdx.x * 64, nlive = min(64, nviews - v0);
    double dn = 0, pn = 0, pnew[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double pv = prev_in[k];
        double d = ok ? rhs[k] : 0.;  // a singular view block gets no update
        pnew[k] = pv - d;
        dn += d * d;
        pn += pv * pv;
    }
    wave_store_rows<6>(pnew, sm, pose_cur + 6 * (size_t)v0, nlive);
    if (pc) {
        double rec[16];
        store_pose_const(pnew, rec);
        wave_store_rows<16>(rec, sm, pc + kPoseConstStride * (size_t)v0, nlive);
    }
    dn = wave_sum(live ? dn : 0.);
    pn = wave_sum(live ? pn : 0.);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = dn;
        part[2 * blockIdx.x + 1] = pn;
    }
}

// Column sums of several partial-result tables in one launch.  Workgroup b reduces one column of
// one table: out[seg.out_ofs + k] = sum over rows of seg.part[row * seg.n + k], as a fixed-order tree; `out` is
// device memory.  With `flag` set, the last workgroup to finish copies out[0..n_out) to host_out (pinned host
// memory) in one piece and then stores `seq` to *flag (also pinned), which a polling host thread takes as "all sums
// have landed": one burst over the bus instead of one 8-byte write and one system-wide fence per workgroup.
// With `pairs` (pinned host memory, 16 bytes per sum) every workgroup sends its sum there itself, together with the
// pass's sequence number, in one 16-byte store: a host that finds the number beside a sum has the sum.  No fence, no
// counter, no last workgroup -- the host waits for all the numbers instead of for one flag.
__global__ __launch_bounds__(kBlock) void k_colsum(ColsumArgs a, double* out, unsigned* counter, double* host_out, int n_out,
                                                   volatile unsigned long long* flag, unsigned long long seq,
                                                   const LmCtrl* __restrict__ ctrl, LmStepArgs st, int has_step,
                                                   double2* __restrict__ pairs) {
    __shared__ double sm[kWaves];
    __shared__ int is_last;
    if (ctrl && !ctrl->go) return;
    int k = blockIdx.x, sidx = 0;
    while (sidx + 1 < a.nseg && k >= a.seg[sidx].n) k -= a.seg[sidx].n, sidx++;
    const ColsumSeg sg = a.seg[sidx];
    // eight loads in flight per thread (the rows are a cache line or more apart, so every load is a round trip of
    // its own, and with a few hundred workgroups in all nothing else hides it): summed in a fixed order
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r0 = threadIdx.x; r0 < sg.nrows; r0 += 8 * kBlock) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = r0 + j * kBlock;
            t[j] = r < sg.nrows ? sg.part[(size_t)r * sg.n + k] : 0.;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] += t[j];
    }
    const double ws = wave_sum(((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = sm[0];
        for (int ww = 1; ww < kWaves; ww++) tot += sm[ww];
        out[sg.out_ofs + k] = tot;
        // (the second word is the sum's bits XOR the number: a host that saw a new half beside an old one could tell)
        if (pairs) pairs[sg.out_ofs + k] = make_double2(tot, __longlong_as_double(__double_as_longlong(tot) ^ (long long)seq));
        is_last = 0;
        if (flag || has_step) {
            __threadfence();
            if (atomicAdd(counter, 1u) == gridDim.x - 1) {
                *counter = 0;
                __threadfence();
                is_last = 1;
            }
        }
    }
    if (!flag && !has_step) return;
    __syncthreads();
    if (!is_last) return;
    if (has_step) {
        // The workgroup that finishes last holds every sum of the pass: its first wavefront publishes them and
        // predicts the host's next step (lm_step_body.hpp) without another kernel boundary in between.
        __shared__ LmStepLds w;
        if (threadIdx.x < 64) lm_step_body(st, w, threadIdx.x);
    } else {
        const volatile double* src = out;  // written by other workgroups of this launch
        for (int i = threadIdx.x; i < n_out; i += kBlock) host_out[i] = src[i];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *flag = seq;
    }
}

// Copies n doubles to host-visible memory and then raises the flag (used after an in-stream all-reduce).
__global__ __launch_bounds__(kBlock) void k_publish(const double* src, int n, double* dst, volatile unsigned long long* flag,
                                                    unsigned long long seq) {
    for (int i = threadIdx.x; i < n; i += kBlock) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *flag = seq;
}

// ---- launchers -------------------------------------------------------------------------------

// lanes per view are a compile-time constant of the per-point kernels
template <class F>
static void dispatch_G(int G, F&& f) {
    switch (G) {
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        case 8: f(std::integral_constant<int, 8>{}); break;
#ifndef CCALIB_FAST_BUILD  // development builds keep compile time down
        case 16: f(std::integral_constant<int, 16>{}); break;
        case 32: f(std::integral_constant<int, 32>{}); break;
        default: f(std::integral_constant<int, 64>{}); break;
#else
        default: break;
#endif
    }
}

template <class M>
static void launch_jac_shared(const PassArgs& a, const CamParams& cp, int grid, double* part, hipStream_t st) {
    dispatch_G(a.G, [&](auto g) {
        k_jac_shared<M, decltype(g)::value><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, part);
    });
}
template <class M>
static void launch_jac_view(const PassArgs& a, const CamParams& cp, int grid, double* blk, int npad, hipStream_t st) {
    dispatch_G(a.G, [&](auto g) {
        k_jac_view<M, decltype(g)::value><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, blk, npad);
    });
}
template <class M>
static void launch_jac_fused(const PassArgs& a, const CamParams& cp, const double* pc, int grid, double* blk, double* part,
                             hipStream_t st, const LmCtrl* ctrl, int cam) {
    constexpr bool const_cols = M::index_of(I_CX) >= 0 && M::index_of(I_CY) == M::index_of(I_CX) + 1;
    FusedGeom g = fused_geom(a.maxnp, a.shared_board != 0, a.board_np, const_cols);
    // `grid` counts tiles in flight.  Three one-tile workgroups per CU if their LDS allows it; otherwise, if it
    // fits, one three-tile workgroup (above the 64 KB default limit of dynamic LDS: raised per function).
    const FusedGeom g3 = fused_geom(a.maxnp, a.shared_board != 0, a.board_np, const_cols, 3);
    bool three = 3 * ((g.lds_bytes + 511u) / 512u * 512u) > kLdsPerCU && g3.lds_bytes <= kLdsPerCU && grid % 3 == 0;
    if (const char* e = std::getenv("CCALIB_FUSED_TPB")) three = std::atoi(e) == 3 && g3.lds_bytes <= kLdsPerCU && grid % 3 == 0;
    auto go = [&](auto shared, auto multi) {
        auto f3 = k_jac_fused<M, decltype(shared)::value, decltype(multi)::value, 3>;
        if (three && hipFuncSetAttribute(reinterpret_cast<const void*>(f3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)g3.lds_bytes) == hipSuccess) {
            f3<<<dim3(grid / 3), dim3(3 * kBlock), g3.lds_bytes, st>>>(a, cp, a.off, pc, blk, part, g.RS, g.CH, ctrl, cam);
        } else {
            if (three) (void)hipGetLastError();  // the refused attribute must not surface as a launch error later
            k_jac_fused<M, decltype(shared)::value, decltype(multi)::value, 1>
                <<<dim3(grid), dim3(kBlock), g.lds_bytes, st>>>(a, cp, a.off, pc, blk, part, g.RS, g.CH, ctrl, cam);
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    const bool multi = a.maxnp > g.CH;  // views longer than one chunk of 64 points
    if (a.shared_board)
        multi ? go(T{}, T{}) : go(T{}, F{});
    else
        multi ? go(F{}, T{}) : go(F{}, F{});
}
template <int NC>
static void launch_schur(const double* blk, int npad, int nviews, double lambda, double* part, hipStream_t st,
                         const LmCtrl* ctrl, double lambda2, double* part2) {
    hipLaunchKernelGGL(k_schur<NC>, dim3((nviews + 63) / 64, part2 ? 2 : 1), dim3(64), 0, st, blk, npad, nviews, lambda, part, ctrl,
                       lambda2, part2);
}
template <int NC>
static void launch_backsub(const double* blk, int npad, int nviews, double lambda, const DeltaC& dc,
                           const double* pose_prev, double* pose_cur, double* part, double* pc, hipStream_t st,
                           const LmCtrl* ctrl) {
    hipLaunchKernelGGL(k_backsub<NC>, dim3((nviews + 63) / 64), dim3(64), 0, st, blk, npad, nviews, lambda, dc,
                       pose_prev, pose_cur, part, pc, ctrl);
}

template <class M>
constexpr auto fused_or_null() -> decltype(&launch_jac_fused<M>) {
    if constexpr (M::NC + 7 <= 16)
        return &launch_jac_fused<M>;
    else
        return nullptr;
}

#define CCALIB_MODEL_ENTRY(MASK, ASPECT, RAT)                                                         \
    {                                                                                                 \
        MASK, ASPECT, RAT, Model<MASK, ASPECT, RAT>::NC, Model<MASK, ASPECT, RAT>::NRED_U,            \
            Model<MASK, ASPECT, RAT>::NBLK, launch_jac_shared<Model<MASK, ASPECT, RAT>>,              \
            launch_jac_view<Model<MASK, ASPECT, RAT>>, launch_schur<Model<MASK, ASPECT, RAT>::NC>,    \
            launch_backsub<Model<MASK, ASPECT, RAT>::NC>, fused_or_null<Model<MASK, ASPECT, RAT>>()   \
    }

// Column sets built into this library.  The first three are the free sets of the reference's
// call sites; the last two cover every other flag combination of the 12-coefficient model (the
// host driver simply ignores the columns it does not optimise).
static const ModelOps g_models[] = {
    CCALIB_MODEL_ENTRY(0x1FFu, false, false),  // fx fy cx cy k1 k2 p1 p2 k3   (flags = 0)
#ifndef CCALIB_FAST_BUILD
    CCALIB_MODEL_ENTRY(0x1FEu, true, false),   // fy cx cy k1 k2 p1 p2 k3, fx tied (FIX_ASPECT_RATIO)
    CCALIB_MODEL_ENTRY(0x83Eu, true, true),    // fy cx cy k1 k2 k6, fx tied (the reference's stereo flags)
    CCALIB_MODEL_ENTRY(0xFFFu, false, true),   // all twelve
    CCALIB_MODEL_ENTRY(0xFFEu, true, true),    // all but fx, fx tied
#endif
};

const ModelOps* find_model(unsigned need_mask, bool aspect, bool need_rational) {
    const ModelOps* best = nullptr;
    for (const ModelOps& m : g_models) {
        if (m.aspect != aspect) continue;
        if ((m.mask & need_mask) != need_mask) continue;
        if (need_rational && !m.rational) continue;
        if (!best || m.NC < best->NC) best = &m;
    }
    return best;
}

void launch_err_wave(const PassArgs& a, const CamParams& cp, bool rational, const double* pc, int grid, double* part,
                     hipStream_t st) {
    const unsigned lds = (unsigned)(sizeof(float) * (a.shared_board ? 3 * (size_t)a.board_np : 0));
    if (rational)
        k_err_wave<true><<<dim3(grid), dim3(kBlock), lds, st>>>(a, cp, pc, part);
    else
        k_err_wave<false><<<dim3(grid), dim3(kBlock), lds, st>>>(a, cp, pc, part);
}

void launch_err_wave_pair(const ErrPair& e, bool rational, int grid, hipStream_t st) {
    const unsigned lds = (unsigned)(sizeof(float) * (e.a[0].shared_board ? 3 * (size_t)e.a[0].board_np : 0));
    if (rational)
        k_err_wave_pair<true><<<dim3(grid, 2), dim3(kBlock), lds, st>>>(e);
    else
        k_err_wave_pair<false><<<dim3(grid, 2), dim3(kBlock), lds, st>>>(e);
}

void launch_reproj(const PassArgs& a, const CamParams& cp, bool rational, int grid, float2* proj, double* view_err,
                   hipStream_t st, const int* bad) {
    dispatch_G(a.G, [&](auto g) {
        constexpr int GG = decltype(g)::value;
        if (rational)
            k_reproj<true, GG><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, proj, view_err, bad);
        else
            k_reproj<false, GG><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, proj, view_err, bad);
    });
}

void launch_colsum(const ColsumArgs& a, double* out, unsigned* counter, double* host_out, int n_out, unsigned long long* flag,
                   unsigned long long seq, hipStream_t st, const LmCtrl* ctrl, const LmStepArgs* step, double2* pairs) {
    int total = 0;
    for (int i = 0; i < a.nseg; i++) total += a.seg[i].n;
    static const LmStepArgs none = {};
    if (total > 0)
        hipLaunchKernelGGL(k_colsum, dim3(total), dim3(kBlock), 0, st, a, out, counter, host_out, n_out, flag, seq, ctrl,
                           step ? *step : none, step ? 1 : 0, pairs);
}

void launch_publish(const double* src, int n, double* dst, unsigned long long* flag, unsigned long long seq, hipStream_t st) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(kBlock), 0, st, src, n, dst, flag, seq);
}

void launch_pose_const(const double* pose, int nviews, double* pc, hipStream_t st) {
    hipLaunchKernelGGL(k_pose_const, dim3((nviews + 63) / 64), dim3(64), 0, st, pose, nviews, pc);
}

FusedGeom fused_geom(int maxnp, bool shared_board, int board_np, bool const_cols, int tiles) {
    FusedGeom g;
    g.CH = maxnp >= 64 ? 64 : (maxnp + 7) / 8 * 8;
    // longer views go in chunks; chunks of 56 points leave room for twelve wavefronts per CU instead of eight,
    // and are taken when they do not add a chunk
    if (maxnp > 64 && (maxnp + 55) / 56 <= (maxnp + 63) / 64) g.CH = 56;
    // Banks are 4 bytes wide, 64 of them for the 8- and 16-byte reads of the matrix phase, which the LDS
    // serves in lane groups of 32 and 16.  With RS = 4 (mod 8) doubles between columns and a multiple of 32
    // doubles between the wavefronts' buffers, all four operand reads of a group of rows are conflict-free
    // in every wavefront (column c, column c + 4, and the two half-and-half reads of columns c, c +- 8).
    g.RS = 2 * g.CH + 4;
    size_t board_f = shared_board ? (size_t)(3 * board_np + 3) / 4 * 4 : 0;
    // 14 private columns per wavefront + 2 shared ones when the principal point is in the set, else 16;
    // then one pose-record slot per wavefront, and room for the reads that run ahead past the last column
    const size_t ws = ((size_t)(const_cols ? 14 : 16) * g.RS + 31) / 32 * 32;
    g.lds_bytes = (unsigned)(sizeof(float) * board_f + sizeof(double) * ((const_cols ? 2 : 0) * (size_t)g.RS + tiles * kWaves * ws) +
                             sizeof(double) * tiles * kWaves * kPoseConstStride + 128);
    return g;
}

int pass_block_size() { return kBlock; }

size_t pass_lds_bytes(int G, int maxnp, bool shared_board, int board_np) {
    size_t S = (size_t)(64 / G) * maxnp;
    return sizeof(float) * ((shared_board ? 3 * (size_t)board_np : 0) + kWaves * S * (shared_board ? 2 : 5));
}

}  // namespace ccalib
