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

__global__ __launch_bounds__(64) void k_compose(const double* pose1, StereoLR lr, int nviews, int npad, double* pose3,
                                                double* cmp, double* pc3) {
    int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= nviews) return;
    const double* p = pose1 + 6 * (size_t)v;
    double r1[3] = {p[0], p[1], p[2]}, t1[3] = {p[3], p[4], p[5]};
    double r2[3] = {lr.v[0], lr.v[1], lr.v[2]};
    double R1[9], R2[9], dR1[3][9], dR2[3][9], R3[9];
    bool jac = cmp != nullptr;
    if (jac) {
        rodrigues<true>(r1, R1, dR1);
        rodrigues<true>(r2, R2, dR2);
    } else {
        rodrigues<false>(r1, R1, nullptr);
        rodrigues<false>(r2, R2, nullptr);
    }
    mul33(R2, R1, R3);
    M2V m = rodrigues_m2v(R3);
    double* q = pose3 + 6 * (size_t)v;
    q[0] = m.r[0], q[1] = m.r[1], q[2] = m.r[2];
#pragma unroll
    for (int i = 0; i < 3; i++) q[3 + i] = R2[i * 3] * t1[0] + R2[i * 3 + 1] * t1[1] + R2[i * 3 + 2] * t1[2] + lr.v[3 + i];
    if (pc3) {  // camera 1's pose in the form the fused Jacobian kernel reads
        double p3[6] = {q[0], q[1], q[2], q[3], q[4], q[5]}, rec[16];
        store_pose_const(p3, rec);
#pragma unroll
        for (int k = 0; k < 16; k++) pc3[16 * (size_t)v + k] = rec[k];
    }
    if (!jac) return;
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
    double r2[3] = {lr.v[0], lr.v[1], lr.v[2]};
    rodrigues<false>(r2, c.R2, nullptr);
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

template <int NC0, int NC1>
__global__ __launch_bounds__(64) void k_stereo_schur(const double* blk0, const double* blk1, const double* cmp,
                                                     StereoLR lr, int npad, int nviews, double lambda, double* part) {
    constexpr int KC = 6 + NC0 + NC1;
    constexpr int NS = KC * (KC + 1) / 2 + KC;
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    Cmp c;
    load_cmp(cmp, npad, vv, lr, c);
    ViewSys s;
    assemble_view<NC0, NC1>(blk0, blk1, npad, vv, c, s);
    double L[21], invd[6], y[6], Z[KC][6];
    bool ok = chol6(s.Vu, lambda, L, invd) && live;
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = s.ge[k];
    chol6_fwd(L, invd, y);
    {
        double WLR[6][6];
        Bt_times(c, s.M, WLR);  // B^T V3 A
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b < 6; b++) Z[a][b] = WLR[a][b];
            chol6_fwd(L, invd, Z[a]);
        }
    }
#pragma unroll
    for (int i = 0; i < NC0; i++) {
#pragma unroll
        for (int b = 0; b < 6; b++) Z[6 + i][b] = blk0[blk_index((27 + 6 * NC0), 27 + 6 * i + b, vv)];
        chol6_fwd(L, invd, Z[6 + i]);
    }
#pragma unroll
    for (int i = 0; i < NC1; i++) {
        double w[6];
#pragma unroll
        for (int b = 0; b < 6; b++) w[b] = blk1[blk_index((27 + 6 * NC1), 27 + 6 * i + b, vv)];
        row_times_A(c, w, Z[6 + NC0 + i]);
        chol6_fwd(L, invd, Z[6 + NC0 + i]);
    }
    __shared__ double sm[32 * 65];
    LaneReducer red(sm, part + (size_t)blockIdx.x * NS);
#pragma unroll
    for (int i = 0; i < KC; i++)
#pragma unroll
        for (int j = i; j < KC; j++) {
            double d = 0;
#pragma unroll
            for (int b = 0; b < 6; b++) d += Z[i][b] * Z[j][b];
            red.push(ok ? d : 0.);
        }
#pragma unroll
    for (int i = 0; i < KC; i++) {
        double d = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) d += Z[i][b] * y[b];
        red.push(ok ? d : 0.);
    }
    red.flush();
}

template <int NC0, int NC1>
__global__ __launch_bounds__(64) void k_stereo_backsub(const double* blk0, const double* blk1, const double* cmp,
                                                       StereoLR lr, int npad, int nviews, double lambda, StereoDelta d,
                                                       const double* pose_prev, double* pose_cur, double* part, double* pc) {
    int v = blockIdx.x * 64 + threadIdx.x;
    bool live = v < nviews;
    int vv = live ? v : nviews - 1;
    Cmp c;
    load_cmp(cmp, npad, vv, lr, c);
    ViewSys s;
    assemble_view<NC0, NC1>(blk0, blk1, npad, vv, c, s);
    double rhs[6];
#pragma unroll
    for (int k = 0; k < 6; k++) rhs[k] = s.ge[k];
    {
        double WLR[6][6];
        Bt_times(c, s.M, WLR);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) rhs[b] -= WLR[a][b] * d.v[a];
    }
#pragma unroll
    for (int i = 0; i < NC0; i++)
#pragma unroll
        for (int b = 0; b < 6; b++) rhs[b] -= blk0[blk_index((27 + 6 * NC0), 27 + 6 * i + b, vv)] * d.v[6 + i];
    {
        double acc[6] = {0, 0, 0, 0, 0, 0}, wa[6];  // (sum_i d_i W3_i) A
#pragma unroll
        for (int i = 0; i < NC1; i++)
#pragma unroll
            for (int b = 0; b < 6; b++) acc[b] += blk1[blk_index((27 + 6 * NC1), 27 + 6 * i + b, vv)] * d.v[6 + NC0 + i];
        row_times_A(c, acc, wa);
#pragma unroll
        for (int b = 0; b < 6; b++) rhs[b] -= wa[b];
    }
    double L[21], invd[6];
    bool ok = chol6(s.Vu, lambda, L, invd);
    chol6_fwd(L, invd, rhs);
    chol6_bwd(L, invd, rhs);
    double dn = 0, pn = 0, pnew[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double pv = pose_prev[6 * (size_t)vv + k];
        double dd = ok ? rhs[k] : 0.;
        pnew[k] = pv - dd;
        dn += dd * dd;
        pn += pv * pv;
    }
    __shared__ double sm[64 * 17];
    const int v0 = blockIdx.x * 64, nlive = min(64, nviews - v0);
    wave_store_rows<6>(pnew, sm, pose_cur + 6 * (size_t)v0, nlive);
    if (pc) {
        double rec[16];
        store_pose_const(pnew, rec);
        wave_store_rows<16>(rec, sm, pc + 16 * (size_t)v0, nlive);
    }
    dn = wave_sum(live ? dn : 0.);
    pn = wave_sum(live ? pn : 0.);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = dn;
        part[2 * blockIdx.x + 1] = pn;
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
static void schur_t(dim3 grid, hipStream_t st, const double* blk0, const double* blk1, const double* cmp,
                    const StereoLR& lr, int npad, int nviews, double lambda, double* part) {
    hipLaunchKernelGGL((k_stereo_schur<NC, NC>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, lambda, part);
}
template <int NC>
static void backsub_t(dim3 grid, hipStream_t st, const double* blk0, const double* blk1, const double* cmp,
                      const StereoLR& lr, int npad, int nviews, double lambda, const StereoDelta& d,
                      const double* pose_prev, double* pose_cur, double* part, double* pc) {
    hipLaunchKernelGGL((k_stereo_backsub<NC, NC>), grid, dim3(64), 0, st, blk0, blk1, cmp, lr, npad, nviews, lambda, d,
                       pose_prev, pose_cur, part, pc);
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

bool launch_stereo_lr_sums(int nc, const double* blk1, const double* cmp, const StereoLR& lr, int npad, int nviews,
                           double* part, hipStream_t st) {
    dim3 grid((nviews + 63) / 64);
    CCALIB_FOR_NC(nc, lr_sums_t, grid, st, blk1, cmp, lr, npad, nviews, part);
    return true;
}

bool launch_stereo_schur(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr, int npad,
                         int nviews, double lambda, double* part, hipStream_t st) {
    dim3 grid((nviews + 63) / 64);
    CCALIB_FOR_NC(nc, schur_t, grid, st, blk0, blk1, cmp, lr, npad, nviews, lambda, part);
    return true;
}

bool launch_stereo_backsub(int nc, const double* blk0, const double* blk1, const double* cmp, const StereoLR& lr,
                           int npad, int nviews, double lambda, const StereoDelta& d, const double* pose_prev,
                           double* pose_cur, double* part, double* pc, hipStream_t st) {
    dim3 grid((nviews + 63) / 64);
    CCALIB_FOR_NC(nc, backsub_t, grid, st, blk0, blk1, cmp, lr, npad, nviews, lambda, d, pose_prev, pose_cur, part, pc);
    return true;
}

}  // namespace ccalib
