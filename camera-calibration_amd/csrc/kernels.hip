// kernels.hip -- the data-parallel passes of the calibration LM loop, hand-written for gfx950.
//
// Replaces the per-view / per-point loops of cvCalibrateCamera2Internal and
// cvStereoCalibrateImpl (OpenCV 4.1.0 calibration.cpp), which the reference reaches through
// monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317 and
// binocular_chessboard.cpp:152-162.
//
// Work decomposition (all FP64 VALU, no MFMA: the blocks are 6x6 .. 18x18 rank-2 updates):
//   * a view is owned by G consecutive lanes of one wavefront (G = 2..64, power of two);
//     lane g handles points g, g+G, ...; image points are read in the caller's own layout
//     (view-major float2), so the G lanes of a view read one contiguous run;
//   * k_err         residuals only, one accumulator per lane;
//   * k_jac_shared  the shared (intrinsic) block: JiT*Ji, JiT*e, eT*e accumulated per lane over
//                   every point the lane sees, reduced once per workgroup (wave shuffles + LDS);
//   * k_jac_view    the per-view blocks JeT*Je, JeT*e, JiT*Je accumulated per lane over the
//                   lane's points, butterfly-reduced over the view's G lanes, stored SoA;
//                   per-view constants (R, t, dR/dr) are staged in LDS, not held in VGPRs;
//   * k_schur / k_backsub  one lane per view: 6x6 Cholesky elimination of the view block.
// Reductions are fixed-order trees (no atomics), so results are bitwise reproducible.
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include "device_math.hpp"

namespace ccalib {

static constexpr int kBlock = 256;
static constexpr int kWaves = kBlock / 64;
static constexpr int kBoardMax = 1024;  // shared-board points staged in LDS (12 KiB as float)
static constexpr int kVcStride = 39;    // doubles per view of LDS-resident constants (R 9, t 3, dR 27)

struct Lane {
    int g;       // index inside the view's lane group
    int slot0;   // first view slot of this lane
    int nslot;   // stride in view slots
    int vl;      // view slot inside the workgroup
};
__device__ __forceinline__ Lane lane_map(int G) {
    Lane l;
    int tid = threadIdx.x;
    l.g = tid & (G - 1);
    l.vl = tid / G;
    int per_block = kBlock / G;
    l.slot0 = blockIdx.x * per_block + l.vl;
    l.nslot = gridDim.x * per_block;
    return l;
}

__device__ __forceinline__ void stage_board(const PassArgs& a, float* board) {
    if (a.board_in_lds) {
        for (int i = threadIdx.x; i < 3 * a.board_np; i += kBlock) board[i] = a.obj[i];
        __syncthreads();
    }
}
__device__ __forceinline__ void load_obj(const PassArgs& a, const float* board, int view_first, int p, double& X,
                                         double& Y, double& Z) {
    if (a.board_in_lds) {
        X = board[3 * p], Y = board[3 * p + 1], Z = board[3 * p + 2];
    } else {
        const float* q = a.obj + 3 * (size_t)(a.shared_board ? p : view_first + p);
        X = q[0], Y = q[1], Z = q[2];
    }
}

// Sums `n` per-lane values over the workgroup and writes them to out[0..n).  red: kWaves*n doubles.
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

// ---- residual pass -------------------------------------------------------------------------
template <bool RAT>
__global__ __launch_bounds__(kBlock) void k_err(PassArgs a, CamParams cp, double* part) {
    using M = Model<0u, false, RAT>;
    __shared__ float board[3 * kBoardMax];
    __shared__ double red[kWaves];
    stage_board(a, board);
    Lane l = lane_map(a.G);
    double acc[1] = {0.};
    for (int v = l.slot0; v < a.nviews; v += l.nslot) {
        int first = a.off[v], np = a.off[v + 1] - first;
        const double* ps = a.pose + 6 * (size_t)v;
        double r[3] = {ps[0], ps[1], ps[2]}, t[3] = {ps[3], ps[4], ps[5]}, R[9];
        rodrigues<false>(r, R, nullptr);
        for (int p = l.g; p < np; p += a.G) {
            double X, Y, Z;
            load_obj(a, board, first, p, X, Y, Z);
            float2 m = a.img[first + p];
            double Xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
            double Yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
            double Zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
            PointOut<M> o;
            point_eval<M, false, false>(Xc, Yc, Zc, (double)m.x, (double)m.y, cp, o);
            acc[0] += o.ex * o.ex + o.ey * o.ey;
        }
    }
    block_reduce_store<1>(acc, red, part + blockIdx.x);
}

// ---- projectPoints + per-view squared error (the reference's RMS re-check, monocular_circle.cpp:5-29) ----
template <bool RAT>
__global__ __launch_bounds__(kBlock) void k_reproj(PassArgs a, CamParams cp, float2* proj, double* view_err) {
    using M = Model<0u, false, RAT>;
    __shared__ float board[3 * kBoardMax];
    stage_board(a, board);
    Lane l = lane_map(a.G);
    int nround = (a.nviews + l.nslot - 1) / l.nslot;
    for (int rd = 0; rd < nround; rd++) {
        int v = l.slot0 + rd * l.nslot;
        bool live = v < a.nviews;
        double acc = 0.;
        if (live) {
            int first = a.off[v], np = a.off[v + 1] - first;
            const double* ps = a.pose + 6 * (size_t)v;
            double r[3] = {ps[0], ps[1], ps[2]}, t[3] = {ps[3], ps[4], ps[5]}, R[9];
            rodrigues<false>(r, R, nullptr);
            for (int p = l.g; p < np; p += a.G) {
                double X, Y, Z;
                load_obj(a, board, first, p, X, Y, Z);
                float2 m = a.img[first + p];
                double Xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
                double Yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
                double Zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
                PointOut<M> o;
                point_eval<M, false, false>(Xc, Yc, Zc, 0., 0., cp, o);
                // projectPoints returns Point2f; the reference compares the rounded values
                float2 pr = make_float2((float)o.ex, (float)o.ey);
                if (proj) proj[first + p] = pr;
                double dx = (double)m.x - (double)pr.x, dy = (double)m.y - (double)pr.y;
                acc += dx * dx + dy * dy;
            }
        }
        for (int msk = 1; msk < a.G; msk <<= 1) acc += __shfl_xor(acc, msk, 64);
        if (live && l.g == 0) view_err[v] = acc;
    }
}

// ---- shared (intrinsic) block ----------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(kBlock) void k_jac_shared(PassArgs a, CamParams cp, double* part) {
    constexpr int NC = M::NC, NU = M::NU, NR = M::NRED_U;
    __shared__ float board[3 * kBoardMax];
    __shared__ double red[kWaves * NR];
    stage_board(a, board);
    Lane l = lane_map(a.G);
    double acc[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) acc[k] = 0.;
    for (int v = l.slot0; v < a.nviews; v += l.nslot) {
        int first = a.off[v], np = a.off[v + 1] - first;
        const double* ps = a.pose + 6 * (size_t)v;
        double r[3] = {ps[0], ps[1], ps[2]}, t[3] = {ps[3], ps[4], ps[5]}, R[9];
        rodrigues<false>(r, R, nullptr);
        for (int p = l.g; p < np; p += a.G) {
            double X, Y, Z;
            load_obj(a, board, first, p, X, Y, Z);
            float2 m = a.img[first + p];
            double Xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
            double Yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
            double Zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
            PointOut<M> o;
            point_eval<M, true, false>(Xc, Yc, Zc, (double)m.x, (double)m.y, cp, o);
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
        }
    }
    block_reduce_store<NR>(acc, red, part + (size_t)blockIdx.x * NR);
}

// ---- per-view blocks ---------------------------------------------------------------------------
// blk layout (SoA, leading dimension npad): rows 0..20 V upper triangle, 21..26 ge,
// 27 + 6*i + b  = W[i][b] (intrinsic column i, pose column b).
template <class M>
__global__ __launch_bounds__(kBlock) void k_jac_view(PassArgs a, CamParams cp, double* blk, int npad) {
    constexpr int NC = M::NC, NB = M::NBLK;
    __shared__ float board[3 * kBoardMax];
    __shared__ double vc[(kBlock / 2) * kVcStride];
    stage_board(a, board);
    Lane l = lane_map(a.G);
    // every lane of the workgroup runs the same number of slot iterations (barriers inside)
    int nround = (a.nviews + l.nslot - 1) / l.nslot;
    for (int rd = 0; rd < nround; rd++) {
        int v = l.slot0 + rd * l.nslot;
        bool live = v < a.nviews;
        double* c = vc + l.vl * kVcStride;
        if (live && l.g == 0) {
            const double* ps = a.pose + 6 * (size_t)v;
            double r[3] = {ps[0], ps[1], ps[2]}, R[9], dR[3][9];
            rodrigues<true>(r, R, dR);
#pragma unroll
            for (int k = 0; k < 9; k++) c[k] = R[k];
            c[9] = ps[3], c[10] = ps[4], c[11] = ps[5];
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 9; k++) c[12 + 9 * j + k] = dR[j][k];
        }
        __syncthreads();
        double acc[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) acc[k] = 0.;
        int first = 0, np = 0;
        if (live) {
            first = a.off[v];
            np = a.off[v + 1] - first;
        }
        for (int p = l.g; p < np; p += a.G) {
            double X, Y, Z;
            load_obj(a, board, first, p, X, Y, Z);
            float2 m = a.img[first + p];
            double Xc = c[0] * X + c[1] * Y + c[2] * Z + c[9];
            double Yc = c[3] * X + c[4] * Y + c[5] * Z + c[10];
            double Zc = c[6] * X + c[7] * Y + c[8] * Z + c[11];
            PointOut<M> o;
            point_eval<M, true, true>(Xc, Yc, Zc, (double)m.x, (double)m.y, cp, o);
            double jex[6], jey[6];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double* d = c + 12 + 9 * j;
                double dX = d[0] * X + d[1] * Y + d[2] * Z;
                double dY = d[3] * X + d[4] * Y + d[5] * Z;
                double dZ = d[6] * X + d[7] * Y + d[8] * Z;
                jex[j] = o.Tx[0] * dX + o.Tx[1] * dY + o.Tx[2] * dZ;
                jey[j] = o.Ty[0] * dX + o.Ty[1] * dY + o.Ty[2] * dZ;
                jex[3 + j] = o.Tx[j];
                jey[3 + j] = o.Ty[j];
            }
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
        }
        // butterfly over the view's G lanes
        for (int m = 1; m < a.G; m <<= 1) {
#pragma unroll
            for (int k = 0; k < NB; k++) acc[k] += __shfl_xor(acc[k], m, 64);
        }
        if (live && l.g == 0) {
#pragma unroll
            for (int k = 0; k < NB; k++) blk[(size_t)k * npad + v] = acc[k];
        }
        __syncthreads();
    }
}

// ---- per-view elimination --------------------------------------------------------------------
// One lane per view, one wavefront per workgroup.  Output per workgroup: the upper triangle of
// sum W V'^-1 W^T over its 64 views (NC(NC+1)/2 values) followed by sum W V'^-1 ge (NC values).
template <int NC>
__global__ __launch_bounds__(64) void k_schur(const double* blk, int npad, int nviews, double lambda, double* part) {
    constexpr int NS = NC * (NC + 1) / 2 + NC;
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    double Vu[21], L[21], invd[6], y[6], Z[NC][6];
#pragma unroll
    for (int k = 0; k < 21; k++) Vu[k] = blk[(size_t)k * npad + vv];
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = blk[(size_t)(21 + k) * npad + vv];
    bool ok = chol6(Vu, lambda, L, invd) && live;
    chol6_fwd(L, invd, y);
#pragma unroll
    for (int i = 0; i < NC; i++) {
#pragma unroll
        for (int b = 0; b < 6; b++) Z[i][b] = blk[(size_t)(27 + 6 * i + b) * npad + vv];
        chol6_fwd(L, invd, Z[i]);  // Z_i = L^-1 W_i^T, so W V'^-1 W^T = Z Z^T
    }
    double* out = part + (size_t)blockIdx.x * NS;
    int k = 0;
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
        for (int j = i; j < NC; j++) {
            double s = 0;
#pragma unroll
            for (int b = 0; b < 6; b++) s += Z[i][b] * Z[j][b];
            s = wave_sum(ok ? s : 0.);
            if (threadIdx.x == 0) out[k] = s;
            k++;
        }
#pragma unroll
    for (int i = 0; i < NC; i++) {
        double s = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) s += Z[i][b] * y[b];
        s = wave_sum(ok ? s : 0.);
        if (threadIdx.x == 0) out[k] = s;
        k++;
    }
}

// trial pose = committed pose - V'^-1 (ge - W^T dc).  part per workgroup: sum |delta|^2, sum |pose_prev|^2.
template <int NC>
__global__ __launch_bounds__(64) void k_backsub(const double* blk, int npad, int nviews, double lambda, DeltaC dc,
                                                const double* pose_prev, double* pose_cur, double* part) {
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    double Vu[21], L[21], invd[6], rhs[6];
#pragma unroll
    for (int k = 0; k < 21; k++) Vu[k] = blk[(size_t)k * npad + vv];
#pragma unroll
    for (int k = 0; k < 6; k++) rhs[k] = blk[(size_t)(21 + k) * npad + vv];
#pragma unroll
    for (int i = 0; i < NC; i++) {
        double d = dc.v[i];
#pragma unroll
        for (int b = 0; b < 6; b++) rhs[b] -= blk[(size_t)(27 + 6 * i + b) * npad + vv] * d;
    }
    bool ok = chol6(Vu, lambda, L, invd);
    chol6_fwd(L, invd, rhs);
    chol6_bwd(L, invd, rhs);
    double dn = 0, pn = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double pv = pose_prev[6 * (size_t)vv + k];
        double d = ok ? rhs[k] : 0.;  // a singular view block gets no update
        if (live) pose_cur[6 * (size_t)vv + k] = pv - d;
        dn += d * d;
        pn += pv * pv;
    }
    dn = wave_sum(live ? dn : 0.);
    pn = wave_sum(live ? pn : 0.);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = dn;
        part[2 * blockIdx.x + 1] = pn;
    }
}

// out[k] = sum over rows of part[row * n + k]; one workgroup per k, fixed-order tree.
__global__ __launch_bounds__(kBlock) void k_colsum(const double* part, int nrows, int n, double* out) {
    __shared__ double sm[kBlock];
    int k = blockIdx.x;
    double s = 0;
    for (int r = threadIdx.x; r < nrows; r += kBlock) s += part[(size_t)r * n + k];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int h = kBlock / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sm[threadIdx.x] += sm[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = sm[0];
}

// ---- launchers -------------------------------------------------------------------------------

template <class M>
static void launch_jac_shared(const PassArgs& a, const CamParams& cp, int grid, double* part, hipStream_t st) {
    hipLaunchKernelGGL(k_jac_shared<M>, dim3(grid), dim3(kBlock), 0, st, a, cp, part);
}
template <class M>
static void launch_jac_view(const PassArgs& a, const CamParams& cp, int grid, double* blk, int npad, hipStream_t st) {
    hipLaunchKernelGGL(k_jac_view<M>, dim3(grid), dim3(kBlock), 0, st, a, cp, blk, npad);
}
template <int NC>
static void launch_schur(const double* blk, int npad, int nviews, double lambda, double* part, hipStream_t st) {
    hipLaunchKernelGGL(k_schur<NC>, dim3((nviews + 63) / 64), dim3(64), 0, st, blk, npad, nviews, lambda, part);
}
template <int NC>
static void launch_backsub(const double* blk, int npad, int nviews, double lambda, const DeltaC& dc,
                           const double* pose_prev, double* pose_cur, double* part, hipStream_t st) {
    hipLaunchKernelGGL(k_backsub<NC>, dim3((nviews + 63) / 64), dim3(64), 0, st, blk, npad, nviews, lambda, dc,
                       pose_prev, pose_cur, part);
}

#define CCALIB_MODEL_ENTRY(MASK, ASPECT, RAT)                                                         \
    {                                                                                                 \
        MASK, ASPECT, RAT, Model<MASK, ASPECT, RAT>::NC, Model<MASK, ASPECT, RAT>::NRED_U,            \
            Model<MASK, ASPECT, RAT>::NBLK, launch_jac_shared<Model<MASK, ASPECT, RAT>>,              \
            launch_jac_view<Model<MASK, ASPECT, RAT>>, launch_schur<Model<MASK, ASPECT, RAT>::NC>,    \
            launch_backsub<Model<MASK, ASPECT, RAT>::NC>                                              \
    }

// Column sets built into this library.  The first three are the free sets of the reference's
// call sites; the last two cover every other flag combination of the 12-coefficient model (the
// host driver simply ignores the columns it does not optimise).
static const ModelOps g_models[] = {
    CCALIB_MODEL_ENTRY(0x1FFu, false, false),  // fx fy cx cy k1 k2 p1 p2 k3   (flags = 0)
    CCALIB_MODEL_ENTRY(0x1FEu, true, false),   // fy cx cy k1 k2 p1 p2 k3, fx tied (FIX_ASPECT_RATIO)
    CCALIB_MODEL_ENTRY(0x83Eu, true, true),    // fy cx cy k1 k2 k6, fx tied (the reference's stereo flags)
    CCALIB_MODEL_ENTRY(0xFFFu, false, true),   // all twelve
    CCALIB_MODEL_ENTRY(0xFFEu, true, true),    // all but fx, fx tied
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

void launch_err(const PassArgs& a, const CamParams& cp, bool rational, int grid, double* part, hipStream_t st) {
    if (rational)
        hipLaunchKernelGGL(k_err<true>, dim3(grid), dim3(kBlock), 0, st, a, cp, part);
    else
        hipLaunchKernelGGL(k_err<false>, dim3(grid), dim3(kBlock), 0, st, a, cp, part);
}

void launch_reproj(const PassArgs& a, const CamParams& cp, bool rational, int grid, float2* proj, double* view_err,
                   hipStream_t st) {
    if (rational)
        hipLaunchKernelGGL(k_reproj<true>, dim3(grid), dim3(kBlock), 0, st, a, cp, proj, view_err);
    else
        hipLaunchKernelGGL(k_reproj<false>, dim3(grid), dim3(kBlock), 0, st, a, cp, proj, view_err);
}

void launch_colsum(const double* part, int nrows, int n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_colsum, dim3(n), dim3(kBlock), 0, st, part, nrows, n, out);
}

int pass_block_size() { return kBlock; }
int board_lds_capacity() { return kBoardMax; }

}  // namespace ccalib
