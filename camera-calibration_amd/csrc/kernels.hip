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
//                   the pose stays in a 13-double axis-angle form so that the accumulators fit;
//   * k_schur / k_backsub  one lane per view: 6x6 Cholesky elimination of the view block.
// Reductions are fixed-order trees (no atomics), so results are bitwise reproducible.
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include "device_math.hpp"

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
    }
    __device__ __forceinline__ void next_round() {
        v0 += stride;
        v += stride;
    }
};

__device__ __forceinline__ void stage_board(const PassArgs& a) {
    if (a.shared_board) {
        for (int i = threadIdx.x; i < 3 * a.board_np; i += kBlock) lds_dyn[i] = a.obj[i];
        __syncthreads();
    }
}

// Copies the points of views [v0, v0 + VPW) into the wavefront's stage; returns the global index of
// the first staged point.  Only lanes of this wavefront touch the stage, so a wave barrier orders it.
template <int G>
__device__ __forceinline__ int stage_views(const PassArgs& a, const Lane<G>& l) {
    int vA = l.v0 < a.nviews ? l.v0 : a.nviews;
    int vB = l.v0 + Lane<G>::VPW < a.nviews ? l.v0 + Lane<G>::VPW : a.nviews;
    int base = a.off[vA], n = a.off[vB] - base;
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
    const float* obj = a.shared_board ? l.board : l.sobj + 3 * local;
    const float2* img = l.simg + local;
    for (int p = l.g; p < np; p += G) {
        float2 m = img[p];
        float X = obj[3 * p], Y = obj[3 * p + 1], Z = obj[3 * p + 2];
        body((double)X, (double)Y, (double)Z, (double)m.x, (double)m.y);
    }
}

// ---- residual pass -------------------------------------------------------------------------
template <bool RAT, int G>
__global__ __launch_bounds__(kBlock) void k_err(PassArgs a, CamParams cp, double* part) {
    using M = Model<0u, false, RAT>;
    __shared__ double red[kWaves];
    stage_board(a);
    Lane<G> l(a);
    double acc[1] = {0.};
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
            point_eval<M, false, false>(Xc, Yc, Zc, u, v, cp, o);
            acc[0] += o.ex * o.ex + o.ey * o.ey;
        });
    }
    block_reduce_store<1>(acc, red, part + blockIdx.x);
}

// ---- projectPoints + per-view squared error (the reference's RMS re-check, monocular_circle.cpp:5-29) ----
template <bool RAT, int G>
__global__ __launch_bounds__(kBlock) void k_reproj(PassArgs a, CamParams cp, float2* proj, double* view_err) {
    using M = Model<0u, false, RAT>;
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
// blk layout (SoA, leading dimension npad): rows 0..20 V upper triangle, 21..26 ge,
// 27 + 6*i + b  = W[i][b] (intrinsic column i, pose column b).
//
// The 21 + 6 + 6*NC accumulators fill most of the register file, so the pose is kept in its
// 13-double axis-angle form (PoseConst) and R X and the three dR/dr_j X are rebuilt per point from
// a.X and a x X; with T the 2x3 Jacobian of the pixel w.r.t. the camera-frame point,
//   T (dR/dr_j X) = a_j (T.base) + b2 ((a.X) T_j + X_j (T.a)) + b4 (X x T)_j,
//   base = -s X + b1 (a.X) a + b3 (a x X).
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
            double d = q.ax[0] * X + q.ax[1] * Y + q.ax[2] * Z;
            double wx = q.ax[1] * Z - q.ax[2] * Y, wy = q.ax[2] * X - q.ax[0] * Z, wz = q.ax[0] * Y - q.ax[1] * X;
            double c1d = q.c1 * d;
            double Xc = q.t[0] + q.c * X + c1d * q.ax[0] + q.s * wx;
            double Yc = q.t[1] + q.c * Y + c1d * q.ax[1] + q.s * wy;
            double Zc = q.t[2] + q.c * Z + c1d * q.ax[2] + q.s * wz;
            PointOut<M> o;
            point_eval<M, true, true>(Xc, Yc, Zc, u, v, cp, o);
            double b1d = q.b1 * d, b2d = q.b2 * d;
            double base0 = b1d * q.ax[0] + q.b3 * wx - q.s * X;
            double base1 = b1d * q.ax[1] + q.b3 * wy - q.s * Y;
            double base2 = b1d * q.ax[2] + q.b3 * wz - q.s * Z;
            double jex[6], jey[6];
            {
                const double* T = o.Tx;
                double A = T[0] * base0 + T[1] * base1 + T[2] * base2;
                double B = q.b2 * (T[0] * q.ax[0] + T[1] * q.ax[1] + T[2] * q.ax[2]);
                double C0 = Y * T[2] - Z * T[1], C1 = Z * T[0] - X * T[2], C2 = X * T[1] - Y * T[0];
                jex[0] = q.ax[0] * A + b2d * T[0] + B * X + q.b4 * C0;
                jex[1] = q.ax[1] * A + b2d * T[1] + B * Y + q.b4 * C1;
                jex[2] = q.ax[2] * A + b2d * T[2] + B * Z + q.b4 * C2;
                jex[3] = T[0], jex[4] = T[1], jex[5] = T[2];
            }
            {
                const double* T = o.Ty;
                double A = T[0] * base0 + T[1] * base1 + T[2] * base2;
                double B = q.b2 * (T[0] * q.ax[0] + T[1] * q.ax[1] + T[2] * q.ax[2]);
                double C0 = Y * T[2] - Z * T[1], C1 = Z * T[0] - X * T[2], C2 = X * T[1] - Y * T[0];
                jey[0] = q.ax[0] * A + b2d * T[0] + B * X + q.b4 * C0;
                jey[1] = q.ax[1] * A + b2d * T[1] + B * Y + q.b4 * C1;
                jey[2] = q.ax[2] * A + b2d * T[2] + B * Z + q.b4 * C2;
                jey[3] = T[0], jey[4] = T[1], jey[5] = T[2];
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
        });
        // butterfly over the view's G lanes
#pragma unroll
        for (int m = 1; m < G; m <<= 1) {
#pragma unroll
            for (int k = 0; k < NB; k++) acc[k] += __shfl_xor(acc[k], m, 64);
        }
        if (live && l.g == 0) {
#pragma unroll
            for (int k = 0; k < NB; k++) blk[(size_t)k * npad + l.v] = acc[k];
        }
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

void launch_err(const PassArgs& a, const CamParams& cp, bool rational, int grid, double* part, hipStream_t st) {
    dispatch_G(a.G, [&](auto g) {
        constexpr int GG = decltype(g)::value;
        if (rational)
            k_err<true, GG><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, part);
        else
            k_err<false, GG><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, part);
    });
}

void launch_reproj(const PassArgs& a, const CamParams& cp, bool rational, int grid, float2* proj, double* view_err,
                   hipStream_t st) {
    dispatch_G(a.G, [&](auto g) {
        constexpr int GG = decltype(g)::value;
        if (rational)
            k_reproj<true, GG><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, proj, view_err);
        else
            k_reproj<false, GG><<<dim3(grid), dim3(kBlock), a.lds_bytes, st>>>(a, cp, proj, view_err);
    });
}

void launch_colsum(const double* part, int nrows, int n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_colsum, dim3(n), dim3(kBlock), 0, st, part, nrows, n, out);
}

int pass_block_size() { return kBlock; }

size_t pass_lds_bytes(int G, int maxnp, bool shared_board, int board_np) {
    size_t S = (size_t)(64 / G) * maxnp;
    return sizeof(float) * ((shared_board ? 3 * (size_t)board_np : 0) + kWaves * S * (shared_board ? 2 : 5));
}

}  // namespace ccalib
