// stereo_kernels.hip -- per-view stereo kernels (one lane per view, FP64 VALU).
//
// cvStereoCalibrateImpl (OpenCV 4.1.0 calibration.cpp; reference call sites
// binocular_chessboard.cpp:152-162, binocular_circle.cpp:153-163) poses camera 1 of view i at
// (r3,t3) = (om_LR,T_LR) o (r1,t1) through cvComposeRT and rewrites every Jacobian ROW of camera 1
// from d/d(r3,t3) into d/d(r1,t1) and d/d(om_LR,T_LR).  Both rewrites are multiplications by
// per-view constant 6x6 matrices
//     Je1 = Je3 A,   A = [dr3/dr1 0; 0 R2]           J_LR = Je3 B,   B = [dr3/dr2 0; dt3/dr2 I]
// so here camera 1 is accumulated in the (r3,t3) basis by the same per-point kernels as camera 0
// and the rewrite is applied once per view to the accumulated blocks:
//     V = V0 + A^T V3 A      ge = ge0 + A^T ge3     W_LR = B^T V3 A     W_cam1 = W3 A
//     U_LR,LR += B^T V3 B    g_LR += B^T ge3        U_LR,cam1 += (W3 B)^T
#include "stereo_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "device_math.hpp"
#include "kernels.hpp"

namespace ccalib {

namespace {

// cvRodrigues2 matrix -> vector for an orthonormal R, with its 3x9 Jacobian in closed form:
//   dr_i/dR_k = theta*vth * d(om_i)/dR_k + om_i * kappa * I_k,
// om = (R21-R12, R02-R20, R10-R01), vth = 1/(2 sin), kappa = theta*d1 + vth*d2.
// Returns the two scalars; the caller contracts them with whatever dR it has.
struct M2V {
    double r[3];     // rotation vector
    double om[3];    // unnormalised axis
    double tv, kappa;  // theta*vth and kappa (both 0 when the Jacobian is zero)
};
__device__ __forceinline__ M2V rodrigues_m2v(const double R[9]) {
    M2V o;
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    o.om[0] = rx, o.om[1] = ry, o.om[2] = rz;
    if (s < 1e-5) {
        if (c > 0) {
            o.r[0] = o.r[1] = o.r[2] = 0;
            o.tv = 0.5, o.kappa = 0;
        } else {
            double t;
            t = (R[0] + 1) * 0.5;
            double ax = sqrt(fmax(t, 0.));
            t = (R[4] + 1) * 0.5;
            double ay = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            double az = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(ax) < fabs(ay) && fabs(ax) < fabs(az) && (R[5] > 0) != (ay * az > 0)) az = -az;
            double sc = theta / sqrt(ax * ax + ay * ay + az * az);
            o.r[0] = ax * sc, o.r[1] = ay * sc, o.r[2] = az * sc;
            o.tv = 0, o.kappa = 0;
        }
        return o;
    }
    double vth = 1 / (2 * s);
    double d1 = 0.5 * vth * c / (s * s);  // 0.5 * dvth/dtheta * dtheta/dtr
    double d2 = -0.5 / s;
    o.tv = theta * vth;
    o.kappa = theta * d1 + vth * d2;
    o.r[0] = rx * o.tv, o.r[1] = ry * o.tv, o.r[2] = rz * o.tv;
    return o;
}
// dr/dparam for a parameter that moves R by the matrix D (row-major 3x3)
__device__ __forceinline__ void m2v_apply(const M2V& m, const double D[9], double out[3]) {
    double tr = D[0] + D[4] + D[8];
    out[0] = m.tv * (D[7] - D[5]) + m.om[0] * m.kappa * tr;
    out[1] = m.tv * (D[2] - D[6]) + m.om[1] * m.kappa * tr;
    out[2] = m.tv * (D[3] - D[1]) + m.om[2] * m.kappa * tr;
}

__device__ __forceinline__ void mul33(const double A[9], const double B[9], double C[9]) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// Camera 1's pose for view v: (r3, t3) = (r2, t2) o (r1, t1) into p3; with JAC the derivatives of the composition
// (dr3/dr1, dr3/dr2, dt3/dr2; 27 numbers, one column of `cmp` per number) are stored for that view as well.
template <bool JAC>
__device__ __forceinline__ void compose_view(const double r1[3], const double t1[3], const StereoLR& lr, double p3[6], double* cmp,
                                             int npad, int v) {
    double r2[3] = {lr.v[0], lr.v[1], lr.v[2]};
    double R1[9], R2[9], dR1[3][9], dR2[3][9], R3[9];
    if (JAC) {
        rodrigues<true>(r1, R1, dR1);
        rodrigues<true>(r2, R2, dR2);
    } else {
        rodrigues<false>(r1, R1, nullptr);
        rodrigues<false>(r2, R2, nullptr);
    }
    mul33(R2, R1, R3);
    M2V m = rodrigues_m2v(R3);
    p3[0] = m.r[0], p3[1] = m.r[1], p3[2] = m.r[2];
#pragma unroll
    for (int i = 0; i < 3; i++) p3[3 + i] = R2[i * 3] * t1[0] + R2[i * 3 + 1] * t1[1] + R2[i * 3 + 2] * t1[2] + lr.v[3 + i];
    if (!JAC || !cmp) return;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double D[9], col[3];
        mul33(R2, dR1[j], D);  // dR3/dr1_j
        m2v_apply(m, D, col);
#pragma unroll
        for (int i = 0; i < 3; i++) cmp[(size_t)(i * 3 + j) * npad + v] = col[i];
        mul33(dR2[j], R1, D);  // dR3/dr2_j
        m2v_apply(m, D, col);
#pragma unroll
        for (int i = 0; i < 3; i++) cmp[(size_t)(9 + i * 3 + j) * npad + v] = col[i];
#pragma unroll
        for (int i = 0; i < 3; i++)  // dt3/dr2_j = dR2_j t1
            cmp[(size_t)(18 + i * 3 + j) * npad + v] =
                dR2[j][i * 3] * t1[0] + dR2[j][i * 3 + 1] * t1[1] + dR2[j][i * 3 + 2] * t1[2];
    }
}

__global__ __launch_bounds__(64) void k_compose(const double* pose1, StereoLR lr, int nviews, int npad, double* pose3,
                                                double* cmp, double* pc3) {
    int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= nviews) return;
    const double* p = pose1 + 6 * (size_t)v;
    double r1[3] = {p[0], p[1], p[2]}, t1[3] = {p[3], p[4], p[5]}, p3[6];
    if (cmp)
        compose_view<true>(r1, t1, lr, p3, cmp, npad, v);
    else
        compose_view<false>(r1, t1, lr, p3, nullptr, npad, v);
    double* q = pose3 + 6 * (size_t)v;
#pragma unroll
    for (int k = 0; k < 6; k++) q[k] = p3[k];
    if (pc3) {  // camera 1's pose in the form the fused Jacobian kernel reads
        double rec[16];
        store_pose_const(p3, rec);
#pragma unroll
        for (int k = 0; k < 16; k++) pc3[16 * (size_t)v + k] = rec[k];
    }
}

// per-view composition matrices in registers
struct Cmp {
    double Ar[9], Brr[9], Btr[9], R2[9];
};
__device__ __forceinline__ void load_cmp(const double* cmp, int npad, int v, const StereoLR& lr, Cmp& c) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
        c.Ar[k] = cmp[(size_t)k * npad + v];
        c.Brr[k] = cmp[(size_t)(9 + k) * npad + v];
        c.Btr[k] = cmp[(size_t)(18 + k) * npad + v];
    }
    double r2[3] = {lr.v[0], lr.v[1], lr.v[2]}, Rv[9];
    rodrigues<false>(r2, Rv, nullptr);
#pragma unroll
    for (int k = 0; k < 9; k++)  // the same in every lane: kept in scalar registers
        c.R2[k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(Rv[k])),
                                   __builtin_amdgcn_readfirstlane(__double2loint(Rv[k])));
}
// row-vector (6) times A or B
__device__ __forceinline__ void row_times_A(const Cmp& c, const double w[6], double o[6]) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        o[j] = w[0] * c.Ar[j] + w[1] * c.Ar[3 + j] + w[2] * c.Ar[6 + j];
        o[3 + j] = w[3] * c.R2[j] + w[4] * c.R2[3 + j] + w[5] * c.R2[6 + j];
    }
}
__device__ __forceinline__ void row_times_B(const Cmp& c, const double w[6], double o[6]) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        o[j] = w[0] * c.Brr[j] + w[1] * c.Brr[3 + j] + w[2] * c.Brr[6 + j] + w[3] * c.Btr[j] + w[4] * c.Btr[3 + j] +
               w[5] * c.Btr[6 + j];
        o[3 + j] = w[3 + j];
    }
}
__device__ __forceinline__ void unpack_sym6(const double u[21], double F[6][6]) {
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) F[a][b] = F[b][a] = u[tri6(a, b)];
}

// rows of X^T Y for X in {A, B}: out[a][:] = sum_k X[k][a] * Y[k][:], using the block structure of X
__device__ __forceinline__ void At_times(const Cmp& c, const double Y[6][6], double out[6][6]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            out[a][j] = c.Ar[a] * Y[0][j] + c.Ar[3 + a] * Y[1][j] + c.Ar[6 + a] * Y[2][j];
            out[3 + a][j] = c.R2[a] * Y[3][j] + c.R2[3 + a] * Y[4][j] + c.R2[6 + a] * Y[5][j];
        }
}
__device__ __forceinline__ void Bt_times(const Cmp& c, const double Y[6][6], double out[6][6]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            out[a][j] = c.Brr[a] * Y[0][j] + c.Brr[3 + a] * Y[1][j] + c.Brr[6 + a] * Y[2][j] + c.Btr[a] * Y[3][j] +
                        c.Btr[3 + a] * Y[4][j] + c.Btr[6 + a] * Y[5][j];
            out[3 + a][j] = Y[3 + a][j];
        }
}

template <int NC1>
__global__ __launch_bounds__(64) void k_stereo_lr_sums(const double* blk1, const double* cmp, StereoLR lr, int npad,
                                                       int nviews, double* part) {
    constexpr int NOUT = 27 + 6 * NC1;
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    Cmp c;
    load_cmp(cmp, npad, vv, lr, c);
    double u[21], V3[6][6], ge3[6];
#pragma unroll
    for (int k = 0; k < 21; k++) u[k] = blk1[blk_index((27 + 6 * NC1), k, vv)];
#pragma unroll
    for (int k = 0; k < 6; k++) ge3[k] = blk1[blk_index((27 + 6 * NC1), 21 + k, vv)];
    unpack_sym6(u, V3);
    double N[6][6], BtN[6][6];
#pragma unroll
    for (int a = 0; a < 6; a++) row_times_B(c, V3[a], N[a]);  // V3 B
    Bt_times(c, N, BtN);
    __shared__ double sm[32 * 65];
    LaneReducer red(sm, part + (size_t)blockIdx.x * NOUT);
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) red.push(live ? BtN[a][b] : 0.);
    double g[6];
    // B^T ge3
#pragma unroll
    for (int a = 0; a < 3; a++) {
        g[a] = c.Brr[a] * ge3[0] + c.Brr[3 + a] * ge3[1] + c.Brr[6 + a] * ge3[2] + c.Btr[a] * ge3[3] +
               c.Btr[3 + a] * ge3[4] + c.Btr[6 + a] * ge3[5];
        g[3 + a] = ge3[3 + a];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) red.push(live ? g[a] : 0.);
#pragma unroll
    for (int i = 0; i < NC1; i++) {
        double w[6], wb[6];
#pragma unroll
        for (int b = 0; b < 6; b++) w[b] = blk1[blk_index((27 + 6 * NC1), 27 + 6 * i + b, vv)];
        row_times_B(c, w, wb);
#pragma unroll
        for (int b = 0; b < 6; b++) red.push(live ? wb[b] : 0.);
    }
    red.flush();
}

// Assembles the stereo view block: V (packed upper, damped later), ge, and M = V3 A (kept for W_LR).
struct ViewSys {
    double Vu[21], ge[6];
    double M[6][6];
};
template <int NC0, int NC1>
__device__ __forceinline__ void assemble_view(const double* blk0, const double* blk1, int npad, int vv, const Cmp& c,
                                              ViewSys& s) {
    double u[21], V3[6][6], ge3[6], AtM[6][6];
#pragma unroll
    for (int k = 0; k < 21; k++) u[k] = blk1[blk_index((27 + 6 * NC1), k, vv)];
#pragma unroll
    for (int k = 0; k < 6; k++) ge3[k] = blk1[blk_index((27 + 6 * NC1), 21 + k, vv)];
    unpack_sym6(u, V3);
#pragma unroll
    for (int a = 0; a < 6; a++) row_times_A(c, V3[a], s.M[a]);
    At_times(c, s.M, AtM);
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) s.Vu[tri6(a, b)] = blk0[blk_index((27 + 6 * NC0), tri6(a, b), vv)] + AtM[a][b];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        s.ge[a] = blk0[blk_index((27 + 6 * NC0), 21 + a, vv)] + c.Ar[a] * ge3[0] + c.Ar[3 + a] * ge3[1] + c.Ar[6 + a] * ge3[2];
        s.ge[3 + a] =
            blk0[blk_index((27 + 6 * NC0), 24 + a, vv)] + c.R2[a] * ge3[3] + c.R2[3 + a] * ge3[4] + c.R2[6 + a] * ge3[5];
    }
}

// ---- Schur sums and relative-pose sums, four lanes per view ----------------------------------------
// The rows of the view's coupling block W (6 relative-pose rows, NC rows per camera) and its gradient ge are dealt
// out to the four lanes of a quad: every lane builds V = V0 + A^T V3 A and its Cholesky factor for itself, prepares
// and forward-solves only its own rows, and the products of the Gram matrix Z Z^T are taken between a lane's rows
// and the rows of the lane 0, 1 or 2 places further round the quad (DPP quad_perm).  A wavefront carries 16 views;
// per-lane products go through LDS transposed, 16 at a time, where lane 4p+a adds up the 16 views' contributions of
// product p in quad position a, in a fixed order, and keeps the sum over the wavefront's views in a register.
//
// Every row is prepared by one formula,  w = raw0 + x A,  x = c V3 + e:
//   relative-pose row r:  c = column r of B           (x A = B^T V3 A row r,  x B = B^T V3 B row r)
//   gradient row:         e = ge3, raw0 = ge0         (x A + raw0 = ge,       x B = B^T ge3)
//   camera-0 row i:       raw0 = W0[i]
//   camera-1 row i:       e = W3[i]                   (x A = W3 A row i,      x B = W3 B row i)
// The gradient is carried as one more row (index kc), so Z[i].y needs no code of its own.
template <int CTRL>
__device__ __forceinline__ double quad_take(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

constexpr int kStageRow = 68;  // doubles between products in the LDS stage: quads of one half-wavefront on distinct banks

template <int NC, bool WITH_LR>
__global__ __launch_bounds__(64) void k_stereo_sums(const double* __restrict__ blk0, const double* __restrict__ blk1,
                                                    const double* __restrict__ cmp, StereoLR lr, int npad, int nviews,
                                                    int ngroups, double lambda_y0, const short* __restrict__ tab,
                                                    double* __restrict__ part_y0, double* __restrict__ part_lr, double lambda_y1,
                                                    double* __restrict__ part_y1) {
    // gridDim.y = 2: the same sums for a second damping, side by side with the first (the relative-pose sums do
    // not depend on the damping: the second half pushes zeros in their place and stores none of them)
    const bool second = blockIdx.y != 0;
    const double lambda = second ? lambda_y1 : lambda_y0;
    double* __restrict__ part = second ? part_y1 : part_y0;
    const bool with_lr = WITH_LR && !second;
    constexpr int NB = 27 + 6 * NC, SC = quad_sc(NC), NS = quad_nslot(NC), KC = 6 + 2 * NC;
    constexpr int LRV = quad_nlr(NC), P0 = WITH_LR ? LRV : 0, P = P0 + quad_ngram(NC), NCH = (P + kQuadChunk - 1) / kQuadChunk;
    constexpr int NSUM = KC * (KC + 1) / 2 + KC, NOUT = 27 + 6 * NC;
    const int lane = threadIdx.x, a = lane & 3;
    __shared__ double stage[kQuadChunk * kStageRow];
    double acc[NCH];
#pragma unroll
    for (int k = 0; k < NCH; k++) acc[k] = 0.;
    double R2[9];
    {
        double r2[3] = {lr.v[0], lr.v[1], lr.v[2]};
        rodrigues<false>(r2, R2, nullptr);
    }
    // parks product K of this lane; after the last product of a chunk the chunk is summed over the 16 views
    auto push = [&](auto kk, double val) {
        constexpr int K = decltype(kk)::value, slot = K % kQuadChunk, ch = K / kQuadChunk;
        stage[slot * kStageRow + lane] = val;
        if constexpr (slot == kQuadChunk - 1 || K == P - 1) {
            __builtin_amdgcn_wave_barrier();
            const double* col = stage + (lane >> 2) * kStageRow + a;
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int q = 0; q < 16; q += 4) s0 += col[4 * q], s1 += col[4 * q + 4], s2 += col[4 * q + 8], s3 += col[4 * q + 12];
            if ((lane >> 2) <= slot) acc[ch] += (s0 + s1) + (s2 + s3);
            __builtin_amdgcn_wave_barrier();
        }
    };
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int v = g * 16 + (lane >> 2);
        const bool live = v < nviews;
        const int vv = live ? v : nviews - 1;
        double Ar[9], V3[6][6], Vu[21];
        {
            double u[21];
#pragma unroll
            for (int k = 0; k < 21; k++) u[k] = blk1[blk_index(NB, k, vv)];
            unpack_sym6(u, V3);
        }
#pragma unroll
        for (int k = 0; k < 9; k++) Ar[k] = cmp[(size_t)k * npad + vv];
        // V = V0 + A^T V3 A, A = diag(Ar, R2), block by block
        {
            double Pm[3][3], Q[3][3], T[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    Pm[i][j] = V3[i][0] * Ar[j] + V3[i][1] * Ar[3 + j] + V3[i][2] * Ar[6 + j];
                    Q[i][j] = V3[i][3] * R2[j] + V3[i][4] * R2[3 + j] + V3[i][5] * R2[6 + j];
                    T[i][j] = V3[3 + i][3] * R2[j] + V3[3 + i][4] * R2[3 + j] + V3[3 + i][5] * R2[6 + j];
                }
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if (q >= p) {
                        Vu[tri6(p, q)] = blk0[blk_index(NB, tri6(p, q), vv)] + (Ar[p] * Pm[0][q] + Ar[3 + p] * Pm[1][q] + Ar[6 + p] * Pm[2][q]);
                        Vu[tri6(3 + p, 3 + q)] =
                            blk0[blk_index(NB, tri6(3 + p, 3 + q), vv)] + (R2[p] * T[0][q] + R2[3 + p] * T[1][q] + R2[6 + p] * T[2][q]);
                    }
                    Vu[tri6(p, 3 + q)] = blk0[blk_index(NB, tri6(p, 3 + q), vv)] + (Ar[p] * Q[0][q] + Ar[3 + p] * Q[1][q] + Ar[6 + p] * Q[2][q]);
                }
        }
        double L[21], invd[6];
        const bool use = chol6(Vu, lambda, L, invd) && live;
        __builtin_amdgcn_sched_barrier(0);
        double Z[NS][6];
        // x -> x A (+ raw0), forward-solved into Z[s]; x B pushed when the relative-pose sums are wanted
        auto finish_row = [&](auto ss, const double x[6], const double raw0[6], auto lr_slot) {
            constexpr int S = decltype(ss)::value, LS = decltype(lr_slot)::value;
            double w[6];
            __builtin_amdgcn_sched_barrier(0);  // keeps the next slots' loads from piling up ahead of this one
#pragma unroll
            for (int j = 0; j < 3; j++) {
                w[j] = raw0[j] + (x[0] * Ar[j] + x[1] * Ar[3 + j] + x[2] * Ar[6 + j]);
                w[3 + j] = raw0[3 + j] + (x[3] * R2[j] + x[4] * R2[3 + j] + x[5] * R2[6 + j]);
            }
            chol6_fwd(L, invd, w);
#pragma unroll
            for (int b = 0; b < 6; b++) Z[S][b] = use ? w[b] : 0.;
            if constexpr (WITH_LR && LS >= 0) {
                double Brr[9], Btr[9];
#pragma unroll
                for (int k = 0; k < 9; k++) Brr[k] = Btr[k] = 0.;
                if (with_lr) {
#pragma unroll
                    for (int k = 0; k < 9; k++) Brr[k] = cmp[(size_t)(9 + k) * npad + vv], Btr[k] = cmp[(size_t)(18 + k) * npad + vv];
                }
                static_for<0, 6>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    double o = j < 3 ? x[0] * Brr[j] + x[1] * Brr[3 + j] + x[2] * Brr[6 + j] + x[3] * Btr[j] + x[4] * Btr[3 + j] + x[5] * Btr[6 + j]
                                     : x[j];
                    push(std::integral_constant<int, 6 * LS + j>{}, live ? o : 0.);
                });
            }
        };
        const double zero6[6] = {0, 0, 0, 0, 0, 0};
        {   // slot 0: relative-pose rows 0..3
            const int rc = a < 3 ? a : 2;
            double c[6], x[6];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double br = cmp[(size_t)(9 + 3 * k + rc) * npad + vv], bt = cmp[(size_t)(18 + 3 * k + rc) * npad + vv];
                c[k] = a < 3 ? br : 0.;
                c[3 + k] = a < 3 ? bt : (k == 0 ? 1. : 0.);
            }
#pragma unroll
            for (int j = 0; j < 6; j++)
                x[j] = c[0] * V3[0][j] + c[1] * V3[1][j] + c[2] * V3[2][j] + c[3] * V3[3][j] + c[4] * V3[4][j] + c[5] * V3[5][j];
            finish_row(std::integral_constant<int, 0>{}, x, zero6, std::integral_constant<int, 0>{});
        }
        {   // slot 1: relative-pose rows 4, 5, the gradient, nothing
            double x[6], raw0[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                double g3 = blk1[blk_index(NB, 21 + j, vv)], g0 = blk0[blk_index(NB, 21 + j, vv)];
                x[j] = a == 0 ? V3[4][j] : a == 1 ? V3[5][j] : a == 2 ? g3 : 0.;
                raw0[j] = a == 2 ? g0 : 0.;
            }
            finish_row(std::integral_constant<int, 1>{}, x, raw0, std::integral_constant<int, 1>{});
        }
        static_for<0, SC>([&](auto ii) {  // camera-0 rows
            constexpr int I = decltype(ii)::value;
            const int i = 4 * I + a, ic = i < NC ? i : NC - 1;
            double raw0[6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                double wv = blk0[blk_index(NB, 27 + 6 * ic + b, vv)];
                raw0[b] = i < NC ? wv : 0.;
            }
            finish_row(std::integral_constant<int, 2 + I>{}, zero6, raw0, std::integral_constant<int, -1>{});
        });
        static_for<0, SC>([&](auto ii) {  // camera-1 rows
            constexpr int I = decltype(ii)::value;
            const int i = 4 * I + a, ic = i < NC ? i : NC - 1;
            double x[6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                double wv = blk1[blk_index(NB, 27 + 6 * ic + b, vv)];
                x[b] = i < NC ? wv : 0.;
            }
            finish_row(std::integral_constant<int, 2 + SC + I>{}, x, zero6, std::integral_constant<int, 2 + I>{});
        });
        // Gram products: own rows, then against the lane one place on, then two places on
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NS>([&](auto s1) {
            constexpr int S1 = decltype(s1)::value;
            static_for<S1, NS>([&](auto s2) {
                constexpr int S2 = decltype(s2)::value;
                double d = 0;
#pragma unroll
                for (int b = 0; b < 6; b++) d += Z[S1][b] * Z[S2][b];
                push(std::integral_constant<int, P0 + quad_gram_index(NC, 0, S1, S2)>{}, d);
            });
        });
        static_for<0, NS>([&](auto s2) {
            constexpr int S2 = decltype(s2)::value;
            double o[6];
#pragma unroll
            for (int b = 0; b < 6; b++) o[b] = quad_take<0x39>(Z[S2][b]);
            static_for<0, NS>([&](auto s1) {
                constexpr int S1 = decltype(s1)::value;
                double d = 0;
#pragma unroll
                for (int b = 0; b < 6; b++) d += Z[S1][b] * o[b];
                push(std::integral_constant<int, P0 + quad_gram_index(NC, 1, S1, S2)>{}, d);
            });
        });
        static_for<0, NS>([&](auto s2) {
            constexpr int S2 = decltype(s2)::value;
            double o[6];
#pragma unroll
            for (int b = 0; b < 6; b++) o[b] = quad_take<0x4E>(Z[S2][b]);
            static_for<0, S2 + 1>([&](auto s1) {
                constexpr int S1 = decltype(s1)::value;
                double d = 0;
#pragma unroll
                for (int b = 0; b < 6; b++) d += Z[S1][b] * o[b];
                push(std::integral_constant<int, P0 + quad_gram_index(NC, 2, S1, S2)>{}, d);
            });
        });
    }
    // lane 4p+a holds, per chunk, the wavefront's sum of product p in quad position a; the table says where it goes
    const short* t = tab + (WITH_LR ? 0 : 4 * LRV);
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const int p = ch * kQuadChunk + (lane >> 2);
        if (p < P) {
            const int e = t[4 * p + a];
            if (e >= 0)
                part[(size_t)blockIdx.x * NSUM + e] = acc[ch];
            else if (e <= -2 && with_lr)
                part_lr[(size_t)blockIdx.x * NOUT + (-2 - e)] = acc[ch];
        }
    }
}

// ---- Schur sums, one lane per view, in three passes over the Gram matrix ---------------------------------
// Holding all of Z = W L^-T (kc x 6 per view) at once takes more registers than two wavefronts per SIMD leave, and
// with one wavefront per SIMD nothing hides the latency of the FP64 chains.  So Z is held a part at a time:
//   pass A: relative-pose rows and camera-0 rows     -> blocks (LR,LR) (LR,c0) (c0,c0) and their right-hand sides
//   pass B: camera-1 rows replace camera-0 rows      -> blocks (LR,c1) (c1,c1) and theirs
//   pass C: camera-0 rows again, one at a time       -> block (c0,c1)
// Products leave in that order; `dst` (stereo_view_table) says where each belongs in the packed sums.
// (The relative-pose sums stay in k_stereo_lr_sums: taken along here they cost 15-19 us against 12.5 us there,
// because at two wavefronts per SIMD this kernel's arithmetic is not hidden behind its memory traffic.)
struct TabReducer {  // LaneReducer with all 64 lanes adding and a destination table
    double* sm;
    const short* dst;
    double* out;
    int k = 0, base = 0;
    __device__ __forceinline__ TabReducer(double* sm_, const short* dst_, double* o) : sm(sm_), dst(dst_), out(o) {}
    __device__ __forceinline__ void push(double v) {
        sm[k * 65 + (threadIdx.x & 63)] = v;
        if (++k == 32) flush();
    }
    __device__ __forceinline__ void flush() {
        if (k == 0) return;
        const int lane = threadIdx.x & 63;
        __builtin_amdgcn_wave_barrier();
        const double* row = sm + (lane & 31) * 65 + (lane >> 5) * 32;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int j = 0; j < 32; j += 4) s0 += row[j], s1 += row[j + 1], s2 += row[j + 2], s3 += row[j + 3];
        double s = (s0 + s1) + (s2 + s3);
        s += __shfl_xor(s, 32);
        if (lane < k) out[dst[base + lane]] = s;
        __builtin_amdgcn_wave_barrier();
        base += k;
        k = 0;
    }
};

template <int NC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_stereo_view_sums(
    const double* __restrict__ blk0, const double* __restrict__ blk1, const double* __restrict__ cmp, StereoLR lr, int npad,
    int nviews, double lambda_y0, const short* __restrict__ dst, double* __restrict__ part_y0, double lambda_y1,
    double* __restrict__ part_y1) {
    constexpr int NB = 27 + 6 * NC, KC = 6 + 2 * NC, NSUM = KC * (KC + 1) / 2 + KC;
    const double lambda = blockIdx.y ? lambda_y1 : lambda_y0;  // gridDim.y = 2: a second damping side by side
    double* __restrict__ part = blockIdx.y ? part_y1 : part_y0;
    const int v = blockIdx.x * 64 + threadIdx.x;
    const bool live = v < nviews;
    const int vv = live ? v : nviews - 1;
    __shared__ double sm[32 * 65];
    TabReducer red(sm, dst, part + (size_t)blockIdx.x * NSUM);
    double Ar[9], R2[9];
    {
        double r2[3] = {lr.v[0], lr.v[1], lr.v[2]}, Rv[9];
        rodrigues<false>(r2, Rv, nullptr);
#pragma unroll
        for (int k = 0; k < 9; k++)  // the same in every lane: kept in scalar registers
            R2[k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(Rv[k])),
                                     __builtin_amdgcn_readfirstlane(__double2loint(Rv[k])));
    }
#pragma unroll
    for (int k = 0; k < 9; k++) Ar[k] = cmp[(size_t)k * npad + vv];
    auto times_A = [&](const double w[6], double o[6]) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            o[j] = w[0] * Ar[j] + w[1] * Ar[3 + j] + w[2] * Ar[6 + j];
            o[3 + j] = w[3] * R2[j] + w[4] * R2[3 + j] + w[5] * R2[6 + j];
        }
    };
    double ZL[6][6], Vu[21], y[6];
#pragma unroll
    for (int k = 0; k < 21; k++) Vu[k] = blk0[blk_index(NB, k, vv)];
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = blk0[blk_index(NB, 21 + k, vv)];
    {
        double V3[6][6], ge3[6], M[6][6], Brr[9], Btr[9];
        {
            double u[21];
#pragma unroll
            for (int k = 0; k < 21; k++) u[k] = blk1[blk_index(NB, k, vv)];
            unpack_sym6(u, V3);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) ge3[k] = blk1[blk_index(NB, 21 + k, vv)];
#pragma unroll
        for (int k = 0; k < 9; k++) Brr[k] = cmp[(size_t)(9 + k) * npad + vv], Btr[k] = cmp[(size_t)(18 + k) * npad + vv];
#pragma unroll
        for (int r = 0; r < 6; r++) times_A(V3[r], M[r]);
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 6; q++) {
                if (q >= p) Vu[tri6(p, q)] += Ar[p] * M[0][q] + Ar[3 + p] * M[1][q] + Ar[6 + p] * M[2][q];
                if (q >= 3 + p) Vu[tri6(3 + p, q)] += R2[p] * M[3][q] + R2[3 + p] * M[4][q] + R2[6 + p] * M[5][q];
                ZL[p][q] = Brr[p] * M[0][q] + Brr[3 + p] * M[1][q] + Brr[6 + p] * M[2][q] + Btr[p] * M[3][q] + Btr[3 + p] * M[4][q] +
                           Btr[6 + p] * M[5][q];
                ZL[3 + p][q] = M[3 + p][q];
            }
        double g[6];
        times_A(ge3, g);
#pragma unroll
        for (int k = 0; k < 6; k++) y[k] += g[k];
    }
    double L[21], invd[6];
    const bool use = chol6(Vu, lambda, L, invd) && live;
    if (!use) {  // a view that does not count: with a zero factor every forward solve, hence every product, is zero
#pragma unroll
        for (int k = 0; k < 21; k++) L[k] = 0.;
#pragma unroll
        for (int k = 0; k < 6; k++) invd[k] = 0.;
    }
    // ... unless the view's blocks hold Inf or NaN (the usual reason a factorisation fails): 0 * NaN is NaN.  Then the
    // forward-solved rows are cleared outright, as the four-lane kernel does.  The test is wave-uniform and false in
    // any healthy run, so the selects below cost a scalar branch.
    const bool any_out = __builtin_amdgcn_ballot_w64(!use && live) != 0;
    auto clear_row = [&](double z[6]) {
        if (any_out) {
#pragma unroll
            for (int b = 0; b < 6; b++) z[b] = use ? z[b] : 0.;
        }
    };
    chol6_fwd(L, invd, y);
    clear_row(y);
#pragma unroll
    for (int r = 0; r < 6; r++) {
        chol6_fwd(L, invd, ZL[r]);
        clear_row(ZL[r]);
    }
    auto dot = [&](const double p[6], const double q[6]) {
        double d = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) d += p[b] * q[b];
        return d;
    };
    auto load_c0 = [&](int i, double z[6]) {
#pragma unroll
        for (int b = 0; b < 6; b++) z[b] = blk0[blk_index(NB, 27 + 6 * i + b, vv)];
        chol6_fwd(L, invd, z);
        clear_row(z);
    };
    {   // pass A
        double Z0[NC][6];
#pragma unroll
        for (int i = 0; i < NC; i++) load_c0(i, Z0[i]);
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p; q < 6; q++) red.push(dot(ZL[p], ZL[q]));
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int i = 0; i < NC; i++) red.push(dot(ZL[p], Z0[i]));
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
            for (int j = i; j < NC; j++) red.push(dot(Z0[i], Z0[j]));
#pragma unroll
        for (int p = 0; p < 6; p++) red.push(dot(ZL[p], y));
#pragma unroll
        for (int i = 0; i < NC; i++) red.push(dot(Z0[i], y));
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // passes B and C
        double Z1[NC][6];
#pragma unroll
        for (int i = 0; i < NC; i++) {
            double w[6];
#pragma unroll
            for (int b = 0; b < 6; b++) w[b] = blk1[blk_index(NB, 27 + 6 * i + b, vv)];
            times_A(w, Z1[i]);
            chol6_fwd(L, invd, Z1[i]);
            clear_row(Z1[i]);
        }
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int i = 0; i < NC; i++) red.push(dot(ZL[p], Z1[i]));
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
            for (int j = i; j < NC; j++) red.push(dot(Z1[i], Z1[j]));
#pragma unroll
        for (int i = 0; i < NC; i++) red.push(dot(Z1[i], y));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NC; i++) {
            double z[6];
            load_c0(i, z);
#pragma unroll
            for (int j = 0; j < NC; j++) red.push(dot(z, Z1[j]));
        }
    }
    red.flush();
}

// DUAL: a second trial step from the same blocks in the same pass -- damping lambda2, step d2 -- whose poses go to
// pose_cur2 (records to pc2, norms to part2).  The blocks are what the kernel waits for (1 224 bytes per pair at six
// columns); the second factorisation and solve ride along for a few hundred instructions.
// CMPJ: the trial point is where the next Jacobian pass is taken (run ahead of the accept decision): camera 1's pose comes
// with the composition derivatives, into cmp_next -- what a k_compose launch in front of that pass would do.
template <int NC0, int NC1, bool DUAL, bool CMPJ>
__global__ __launch_bounds__(64) void k_stereo_backsub(const double* blk0, const double* blk1, const double* cmp,
                                                       StereoLR lr, int npad, int nviews, double lambda, StereoDelta d,
                                                       const double* pose_prev, double* pose_cur, double* part, double* pc,
                                                       StereoLR lr_new, double* pose3, double* pc3, double lambda2,
                                                       StereoDelta d2, double* pose_cur2, double* part2, double* pc2,
                                                       double* cmp_next) {
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    Cmp c;
    load_cmp(cmp, npad, vv, lr, c);
    ViewSys s;
    assemble_view<NC0, NC1>(blk0, blk1, npad, vv, c, s);
    double rhs[6], rhs2[6];
#pragma unroll
    for (int k = 0; k < 6; k++) rhs[k] = rhs2[k] = s.ge[k];
    {
        double WLR[6][6];
        Bt_times(c, s.M, WLR);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) {
                rhs[b] -= WLR[a][b] * d.v[a];
                if constexpr (DUAL) rhs2[b] -= WLR[a][b] * d2.v[a];
            }
    }
#pragma unroll
    for (int i = 0; i < NC0; i++)
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const double wv = blk0[blk_index((27 + 6 * NC0), 27 + 6 * i + b, vv)];
            rhs[b] -= wv * d.v[6 + i];
            if constexpr (DUAL) rhs2[b] -= wv * d2.v[6 + i];
        }
    {
        double acc[6] = {0, 0, 0, 0, 0, 0}, acc2[6] = {0, 0, 0, 0, 0, 0}, wa[6];  // (sum_i d_i W3_i) A
#pragma unroll
        for (int i = 0; i < NC1; i++)
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const double wv = blk1[blk_index((27 + 6 * NC1), 27 + 6 * i + b, vv)];
                acc[b] += wv * d.v[6 + NC0 + i];
                if constexpr (DUAL) acc2[b] += wv * d2.v[6 + NC0 + i];
            }
        row_times_A(c, acc, wa);
#pragma unroll
        for (int b = 0; b < 6; b++) rhs[b] -= wa[b];
        if constexpr (DUAL) {
            row_times_A(c, acc2, wa);
#pragma unroll
            for (int b = 0; b < 6; b++) rhs2[b] -= wa[b];
        }
    }
    __shared__ double sm[64 * 17];
    const int v0 = blockIdx.x * 64, nlive = min(64, nviews - v0);
    double prev[6];
#pragma unroll
    for (int k = 0; k < 6; k++) prev[k] = pose_prev[6 * (size_t)vv + k];
    double pnew[6];
    // one trial step: factorise V' for `lam`, solve, store the trial pose (and its record) and the step norms
    auto trial = [&](double lam, double (&r)[6], double* pose_out, double* pc_out, double* part_out) {
        double L[21], invd[6];
        bool ok = chol6(s.Vu, lam, L, invd);
        chol6_fwd(L, invd, r);
        chol6_bwd(L, invd, r);
        double dn = 0, pn = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double dd = ok ? r[k] : 0.;  // a singular view block gets no update
            pnew[k] = prev[k] - dd;
            dn += dd * dd;
            pn += prev[k] * prev[k];
        }
        wave_store_rows<6>(pnew, sm, pose_out + 6 * (size_t)v0, nlive);
        if (pc_out) {
            double rec[16];
            store_pose_const(pnew, rec);
            wave_store_rows<16>(rec, sm, pc_out + 16 * (size_t)v0, nlive);
        }
        dn = wave_sum(live ? dn : 0.);
        pn = wave_sum(live ? pn : 0.);
        if (threadIdx.x == 0) {
            part_out[2 * blockIdx.x] = dn;
            part_out[2 * blockIdx.x + 1] = pn;
        }
    };
    if constexpr (DUAL) trial(lambda2, rhs2, pose_cur2, pc2, part2);
    trial(lambda, rhs, pose_cur, pc, part);   // last: pnew is this trial's pose below
    if (pose3) {  // camera 1's trial pose: (r3,t3) = lr_new o (r1,t1)
        double p3[6];
        compose_view<CMPJ>(pnew, pnew + 3, lr_new, p3, live ? cmp_next : nullptr, npad, vv);
        wave_store_rows<6>(p3, sm, pose3 + 6 * (size_t)v0, nlive);
        if (pc3) {
            double rec[16];
            store_pose_const(p3, rec);
            wave_store_rows<16>(rec, sm, pc3 + 16 * (size_t)v0, nlive);
        }
    }
}

// The stereo view system as the elimination kernels assemble it, written out for inspection: per view
// V = V0 + A^T V3 A (36, full), ge = ge0 + A^T ge3 (6), and the coupling rows W (6 + 2 NC rows of 6, kernel column order:
// W_LR = B^T V3 A, camera 0's W0, camera 1's W3 A).  Same device functions as k_stereo_backsub.
template <int NC>
__global__ __launch_bounds__(64) void k_stereo_fetch(const double* blk0, const double* blk1, const double* cmp, StereoLR lr,
                                                     int npad, int nviews, double* V, double* W, double* ge) {
    constexpr int NB = 27 + 6 * NC, KC = 6 + 2 * NC;
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= nviews) return;
    Cmp c;
    load_cmp(cmp, npad, v, lr, c);
    ViewSys s;
    assemble_view<NC, NC>(blk0, blk1, npad, v, c, s);
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) V[(size_t)v * 36 + a * 6 + b] = V[(size_t)v * 36 + b * 6 + a] = s.Vu[tri6(a, b)];
    for (int a = 0; a < 6; a++) ge[(size_t)v * 6 + a] = s.ge[a];
    double WLR[6][6];
    Bt_times(c, s.M, WLR);
    double* Wv = W + (size_t)v * KC * 6;
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) Wv[a * 6 + b] = WLR[a][b];
    for (int i = 0; i < NC; i++) {
        double w3[6], wa[6];
        for (int b = 0; b < 6; b++) {
            Wv[(6 + i) * 6 + b] = blk0[blk_index(NB, 27 + 6 * i + b, v)];
            w3[b] = blk1[blk_index(NB, 27 + 6 * i + b, v)];
        }
        row_times_A(c, w3, wa);
        for (int b = 0; b < 6; b++) Wv[(6 + NC + i) * 6 + b] = wa[b];
    }
}

}  // namespace

void launch_compose(const double* pose1, const StereoLR& lr, int nviews, int npad, double* pose3, double* cmp, double* pc3,
                    hipStream_t st) {
    hipLaunchKernelGGL(k_compose, dim3((nviews + 63) / 64), dim3(64), 0, st, pose1, lr, nviews, npad, pose3, cmp, pc3);
}

template <int NC>
static void lr_sums_t(dim3 grid, hipStream_t st, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                      int nviews, double* part) {
    hipLaunchKernelGGL(k_stereo_lr_sums<NC>, grid, dim3(64), 0, st, blk1, cmp, lr, npad, nviews, part);
}
template <int NC>
static void backsub_t(dim3 grid, hipStream_t st, const double* blk0, const double* blk1, const double* cmp,
                      const StereoLR& lr, int npad, int nviews, double lambda, const StereoDelta& d,
                      const double* pose_prev, double* pose_cur, double* part, double* pc, const StereoLR& lr_new, double* pose3,
                      double* pc3, const StereoSecondTrial* t2, double* cmp_next) {
    if (t2)
        hipLaunchKernelGGL((k_stereo_backsub<NC, NC, true, false>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, lambda,
                           d, pose_prev, pose_cur, part, pc, lr_new, pose3, pc3, t2->lambda, t2->d, t2->pose, t2->part, t2->pc,
                           nullptr);
    else if (cmp_next)
        hipLaunchKernelGGL((k_stereo_backsub<NC, NC, false, true>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, lambda,
                           d, pose_prev, pose_cur, part, pc, lr_new, pose3, pc3, 0., d, nullptr, nullptr, nullptr, cmp_next);
    else
        hipLaunchKernelGGL((k_stereo_backsub<NC, NC, false, false>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews,
                           lambda, d, pose_prev, pose_cur, part, pc, lr_new, pose3, pc3, 0., d, nullptr, nullptr, nullptr, nullptr);
}

// column counts of the built-in column sets (kernels.hip)
#define CCALIB_FOR_NC(nc, F, ...)            \
    switch (nc) {                            \
        case 6: F<6>(__VA_ARGS__); break;    \
        case 8: F<8>(__VA_ARGS__); break;    \
        case 9: F<9>(__VA_ARGS__); break;    \
        case 11: F<11>(__VA_ARGS__); break;  \
        case 12: F<12>(__VA_ARGS__); break;  \
        default: return false;               \
    }

template <int NC>
static void fetch_t(dim3 grid, hipStream_t st, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr,
                    int npad, int nviews, double* V, double* W, double* ge) {
    hipLaunchKernelGGL(k_stereo_fetch<NC>, grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, V, W, ge);
}

bool launch_stereo_fetch(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                         int nviews, double* V, double* W, double* ge, hipStream_t st) {
    dim3 grid((nviews + 63) / 64);
    CCALIB_FOR_NC(nc, fetch_t, grid, st, blk0, blk1, cmp, lr, npad, nviews, V, W, ge);
    return true;
}

bool launch_stereo_lr_sums(int nc, const double* blk1, const double* cmp, const StereoLR& lr, int npad, int nviews,
                           double* part, hipStream_t st) {
    dim3 grid((nviews + 63) / 64);
    CCALIB_FOR_NC(nc, lr_sums_t, grid, st, blk1, cmp, lr, npad, nviews, part);
    return true;
}

int stereo_sums_rows(int nviews) {
    static const int cap = [] {
        const char* e = std::getenv("CCALIB_SUMS_GRID");
        int v = e ? std::atoi(e) : 0;
        return v > 0 ? v : 2048;
    }();
    return std::max(1, std::min((nviews + 15) / 16, cap));
}

int stereo_sums_table(int nc, short* out) {
    if (nc != 6 && nc != 8 && nc != 9 && nc != 11 && nc != 12) return 0;
    const int kc = 6 + 2 * nc, ns = quad_nslot(nc), sc = quad_sc(nc), nlr = quad_nlr(nc), n = nlr + quad_ngram(nc);
    if (!out) return 4 * n;
    auto tri = [kc](int i, int j) { return i * kc - i * (i - 1) / 2 + (j - i); };
    for (int i = 0; i < 4 * n; i++) out[i] = -1;
    for (int a = 0; a < 4; a++) {
        // relative-pose values: slot ls (0, 1: relative-pose slots; 2+i: camera-1 slot i), component j
        for (int ls = 0; ls < 2 + sc; ls++)
            for (int j = 0; j < 6; j++) {
                int e = -1;
                if (ls < 2) {
                    const int r = 4 * ls + a;
                    if (r < 6 && j >= r) e = r * 6 - r * (r - 1) / 2 + (j - r);
                    if (r == 6) e = 21 + j;
                } else if (4 * (ls - 2) + a < nc)
                    e = 27 + 6 * (4 * (ls - 2) + a) + j;
                if (e >= 0) out[4 * (6 * ls + j) + a] = (short)(-2 - e);
            }
        for (int t = 0; t < 3; t++)
            for (int s1 = 0; s1 < ns; s1++)
                for (int s2 = (t == 1 ? 0 : s1); s2 < ns; s2++) {
                    const int r1 = quad_row(nc, s1, a), r2 = quad_row(nc, s2, (a + t) & 3);
                    if (r1 < 0 || r2 < 0 || (r1 == kc && r2 == kc)) continue;
                    if (t == 2 && s1 == s2 && a >= 2) continue;  // the lane two places on takes the same product
                    const int lo = std::min(r1, r2), hi = std::max(r1, r2);
                    const int e = hi == kc ? kc * (kc + 1) / 2 + lo : tri(lo, hi);
                    out[4 * (nlr + quad_gram_index(nc, t, s1, s2)) + a] = (short)e;
                }
    }
    // every sum must be produced exactly once
    std::vector<int> hits_s(kc * (kc + 1) / 2 + kc, 0), hits_l(27 + 6 * nc, 0);
    for (int i = 0; i < 4 * n; i++) {
        if (out[i] >= 0) hits_s[out[i]]++;
        if (out[i] <= -2) hits_l[-2 - out[i]]++;
    }
    for (int h : hits_s)
        if (h != 1) return 0;
    for (int h : hits_l)
        if (h != 1) return 0;
    return 4 * n;
}

// destinations of k_stereo_view_sums' products, in the order it pushes them (index into the packed upper triangle
// + right-hand side)
int stereo_view_table(int nc, short* out) {
    const int kc = 6 + 2 * nc, nsum = kc * (kc + 1) / 2 + kc;
    if (!out) return nsum;
    auto tri = [kc](int i, int j) { return i * kc - i * (i - 1) / 2 + (j - i); };
    const int rhs = kc * (kc + 1) / 2;
    int n = 0;
    for (int p = 0; p < 6; p++)
        for (int q = p; q < 6; q++) out[n++] = (short)tri(p, q);
    for (int p = 0; p < 6; p++)
        for (int i = 0; i < nc; i++) out[n++] = (short)tri(p, 6 + i);
    for (int i = 0; i < nc; i++)
        for (int j = i; j < nc; j++) out[n++] = (short)tri(6 + i, 6 + j);
    for (int p = 0; p < 6 + nc; p++) out[n++] = (short)(rhs + p);
    for (int p = 0; p < 6; p++)
        for (int i = 0; i < nc; i++) out[n++] = (short)tri(p, 6 + nc + i);
    for (int i = 0; i < nc; i++)
        for (int j = i; j < nc; j++) out[n++] = (short)tri(6 + nc + i, 6 + nc + j);
    for (int i = 0; i < nc; i++) out[n++] = (short)(rhs + 6 + nc + i);
    for (int i = 0; i < nc; i++)
        for (int j = 0; j < nc; j++) out[n++] = (short)tri(6 + i, 6 + nc + j);
    std::vector<int> hits(nsum, 0);
    for (int i = 0; i < n; i++) hits[out[i]]++;
    for (int h : hits)
        if (h != 1) return 0;
    return n == nsum ? n : 0;
}

template <int NC>
static void view_sums_t(dim3 grid, hipStream_t st, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr,
                        int npad, int nviews, double lambda, const short* dst, double* part, double lambda2, double* part2) {
    hipLaunchKernelGGL((k_stereo_view_sums<NC>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, lambda, dst, part, lambda2,
                       part2);
}

bool launch_stereo_view_sums(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                             int nviews, double lambda, const short* dst, double* part, hipStream_t st, double lambda2,
                             double* part2) {
    dim3 grid((nviews + 63) / 64, part2 ? 2 : 1);
    CCALIB_FOR_NC(nc, view_sums_t, grid, st, blk0, blk1, cmp, lr, npad, nviews, lambda, dst, part, lambda2, part2);
    return true;
}

template <int NC>
static void sums_t(dim3 grid, hipStream_t st, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr,
                   int npad, int nviews, int ngroups, double lambda, const short* tab, double* part, double* part_lr, double lambda2,
                   double* part2) {
    if (part_lr)
        hipLaunchKernelGGL((k_stereo_sums<NC, true>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, ngroups, lambda,
                           tab, part, part_lr, lambda2, part2);
    else
        hipLaunchKernelGGL((k_stereo_sums<NC, false>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, ngroups, lambda,
                           tab, part, part_lr, lambda2, part2);
}

bool launch_stereo_sums(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                        int nviews, double lambda, const short* tab, int rows, double* part, double* part_lr, hipStream_t st,
                        double lambda2, double* part2) {
    dim3 grid(rows, part2 ? 2 : 1);
    const int ngroups = (nviews + 15) / 16;
    CCALIB_FOR_NC(nc, sums_t, grid, st, blk0, blk1, cmp, lr, npad, nviews, ngroups, lambda, tab, part, part_lr, lambda2, part2);
    return true;
}

bool launch_stereo_backsub(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr,
                           int npad, int nviews, double lambda, const StereoDelta& d, const double* pose_prev,
                           double* pose_cur, double* part, double* pc, const StereoLR& lr_new, double* pose3, double* pc3,
                           hipStream_t st, const StereoSecondTrial* second, double* cmp_next) {
    if (cmp_next && (second || !pose3)) return false;  // the derivatives come with camera 1's pose, for one trial
    dim3 grid((nviews + 63) / 64);
    CCALIB_FOR_NC(nc, backsub_t, grid, st, blk0, blk1, cmp, lr, npad, nviews, lambda, d, pose_prev, pose_cur, part, pc, lr_new,
                  pose3, pc3, second, cmp_next);
    return true;
}

}  // namespace ccalib
