// device_math.hpp -- per-point camera model and small per-view algebra for gfx950 (FP64 VALU).
//
// What is computed follows cvProjectPoints2Internal / cvRodrigues2 of OpenCV 4.1.0
// (modules/calib3d/src/calibration.cpp), the functions the reference reaches through
// monocular_circle.cpp:18-19 and inside every calibrateCamera / stereoCalibrate call.
// How it is computed does not: the chain rule is factored per point into a symmetric 2x2
// distortion Jacobian D and a 2x3 projection Jacobian T, so each pose column costs 3 FMAs.
#pragma once
#include <hip/hip_runtime.h>

namespace ccalib {

// Intrinsic columns, in the order of OpenCV's 18-vector (only the first 12 are modelled).
enum : int { I_FX = 0, I_FY, I_CX, I_CY, I_K1, I_K2, I_P1, I_P2, I_K3, I_K4, I_K5, I_K6, I_NMODEL = 12 };

struct CamParams {     // by-value kernel argument: lands in SGPRs
    double fx, fy, cx, cy;
    double k[8];       // k1 k2 p1 p2 k3 k4 k5 k6
    double aspect;     // != 0: fx = fy * aspect and the fy column carries (xd*aspect, yd)
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(static_cast<F&&>(f));
    }
}

// Compile-time description of which intrinsic columns a kernel produces.
//   MASK   bit q set = column q (enum above) is produced
//   ASPECT fixed aspect ratio (bit I_FX must be clear)
//   RAT    the rational denominator 1 + k4 r^2 + k5 r^4 + k6 r^6 is evaluated
template <unsigned MASK, bool ASPECT, bool RAT>
struct Model {
    static constexpr unsigned mask = MASK;
    static constexpr bool aspect = ASPECT;
    static constexpr bool rational = RAT;
    static constexpr int popcnt(unsigned m) { return m ? (int)(m & 1u) + popcnt(m >> 1) : 0; }
    static constexpr int NC = popcnt(MASK);
    // i-th produced column -> intrinsic index
    static constexpr int col(int i) {
        int seen = 0;
        for (int q = 0; q < I_NMODEL; q++)
            if (MASK >> q & 1u) {
                if (seen == i) return q;
                seen++;
            }
        return -1;
    }
    // position of intrinsic q among the produced columns, -1 when it is not produced
    static constexpr int index_of(int q) {
        if (!(MASK >> q & 1u)) return -1;
        int seen = 0;
        for (int p = 0; p < q; p++) seen += (int)(MASK >> p & 1u);
        return seen;
    }
    // does the column touch the u (x) / v (y) residual row, and is its entry the constant 1?
    static constexpr bool hasx(int i) {
        int q = col(i);
        return q == I_FX || q == I_CX || q >= I_K1 || (q == I_FY && ASPECT);
    }
    static constexpr bool hasy(int i) {
        int q = col(i);
        return q == I_FY || q == I_CY || q >= I_K1;
    }
    static constexpr bool unitx(int i) { return col(i) == I_CX; }
    static constexpr bool unity(int i) { return col(i) == I_CY; }
    static constexpr int NU = NC * (NC + 1) / 2;   // upper triangle of the shared block
    static constexpr int NRED_U = NU + NC + 1;      // + gradient + sum of squared residuals
    static constexpr int NBLK = 21 + 6 + NC * 6;    // per-view: V (upper), ge, W
    static_assert(!(ASPECT && (MASK & 1u)), "fixed aspect ratio drops the fx column");
};

struct Rt {  // pose as rotation matrix + translation
    double R[9], t[3];
};

// cvRodrigues2 vector -> matrix, optionally with dR/dr as three 3x3 matrices (dR[j] = dR/dr_j).
template <bool WITH_J>
__device__ __forceinline__ void rodrigues(const double r[3], double R[9], double (*dR)[9]) {
    double rx = r[0], ry = r[1], rz = r[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < 2.220446049250313e-16) {
        R[0] = 1, R[1] = 0, R[2] = 0, R[3] = 0, R[4] = 1, R[5] = 0, R[6] = 0, R[7] = 0, R[8] = 1;
        if constexpr (WITH_J) {
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 9; k++) dR[j][k] = 0;
            dR[0][5] = -1, dR[0][7] = 1;
            dR[1][2] = 1, dR[1][6] = -1;
            dR[2][1] = -1, dR[2][3] = 1;
        }
        return;
    }
    double s, c;
    sincos(theta, &s, &c);
    double c1 = 1. - c, it = 1. / theta;
    rx *= it, ry *= it, rz *= it;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double rxm[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c1 * rrt[k] + s * rxm[k];
    R[0] += c, R[4] += c, R[8] += c;
    if constexpr (WITH_J) {
        const double ax[3] = {rx, ry, rz};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double ri = ax[i];
            double a0 = -s * ri, a1 = (s - 2 * c1 * it) * ri, a2 = c1 * it;
            double a3 = (c - s * it) * ri;
            // d(rr^T)/d(axis_i): row i and column i hold the axis, doubled on the diagonal
            double drrt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < 3; k++) {
                drrt[i * 3 + k] += ax[k];
                drrt[k * 3 + i] += ax[k];
            }
            for (int k = 0; k < 9; k++) dR[i][k] = a1 * rrt[k] + a2 * drrt[k] + a3 * rxm[k];
            dR[i][0] += a0, dR[i][4] += a0, dR[i][8] += a0;
        }
        // + a4 * d[r]x/dr_i
        double a4 = s * it;
        dR[0][5] -= a4, dR[0][7] += a4;
        dR[1][2] += a4, dR[1][6] -= a4;
        dR[2][1] -= a4, dR[2][3] += a4;
    }
}

// A pose in the form the per-view-block kernel keeps in registers: unit axis, the three scalars
// of R = c I + c1 a a^T + s [a]x, the four scalars of cvRodrigues2's Jacobian
//   dR/dr_i = r_i (-s I + b1 a a^T + b3 [a]x) + b2 (e_i a^T + a e_i^T) + b4 [e_i]x,
//   b1 = s - 2 c1/theta, b2 = c1/theta, b3 = c - s/theta, b4 = s/theta,
// and the translation: 13 doubles instead of the 39 of (R, t, dR/dr).
struct PoseConst {
    double ax[3];
    double c, c1, s, b1, b2, b3, b4;
    double t[3];
};
__device__ __forceinline__ PoseConst pose_const(const double ps[6]) {
    PoseConst q;
    double rx = ps[0], ry = ps[1], rz = ps[2];
    q.t[0] = ps[3], q.t[1] = ps[4], q.t[2] = ps[5];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < 2.220446049250313e-16) {  // R = I, dR/dr_i = [e_i]x
        q.ax[0] = q.ax[1] = q.ax[2] = 0;
        q.c = 1, q.c1 = 0, q.s = 0, q.b1 = 0, q.b2 = 0, q.b3 = 0, q.b4 = 1;
        return q;
    }
    double s, c;
    sincos(theta, &s, &c);
    double it = 1. / theta, c1 = 1. - c;
    q.ax[0] = rx * it, q.ax[1] = ry * it, q.ax[2] = rz * it;
    q.c = c, q.c1 = c1, q.s = s;
    q.b1 = s - 2 * c1 * it;
    q.b2 = c1 * it;
    q.b3 = c - s * it;
    q.b4 = s * it;
    return q;
}

// A pose written in that form, padded to 16 doubles: the record the fused Jacobian kernel reads.
__device__ __forceinline__ void store_pose_const(const double ps[6], double (&o)[16]) {
    PoseConst q = pose_const(ps);
    o[0] = q.ax[0], o[1] = q.ax[1], o[2] = q.ax[2], o[3] = q.c, o[4] = q.c1, o[5] = q.s, o[6] = q.b1, o[7] = q.b2;
    o[8] = q.b3, o[9] = q.b4, o[10] = q.t[0], o[11] = q.t[1], o[12] = q.t[2], o[13] = 0, o[14] = 0, o[15] = 0;
}

// 1 / d to within an ulp: the hardware estimate and two Newton steps.  The correctly rounded quotient the
// compiler emits for 1. / d is a chain of eleven dependent instructions (scale, estimate, four refinements,
// fix-up) in the middle of every point's arithmetic; d here is a depth of some hundred millimetres, far from
// the ranges that chain exists for.
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.), r, r);
    r = fma(fma(-d, r, 1.), r, r);
    return r;
}

// Everything a pass needs from one point.
template <class M>
struct PointOut {
    double ex, ey;          // residual (projected - observed)
    double jx[M::NC > 0 ? M::NC : 1];  // d u / d intrinsic column (entries for !hasx columns unused)
    double jy[M::NC > 0 ? M::NC : 1];  // d v / d intrinsic column
    double Tx[3], Ty[3];    // d u / d (camera-frame point), d v / d (camera-frame point)
};

// Projection with distortion; fills residuals and, as requested, intrinsic columns and the 2x3
// Jacobian with respect to the camera-frame point.
template <class M, bool WANT_JI, bool WANT_T>
__device__ __forceinline__ void point_eval(double Xc, double Yc, double Zc, double u_obs, double v_obs,
                                           const CamParams& cp, PointOut<M>& o) {
    const double fy = cp.fy;
    const double fx = M::aspect ? fy * cp.aspect : cp.fx;
    double z = Zc != 0. ? fast_rcp(Zc) : 1.;
    double x = Xc * z, y = Yc * z;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double k1 = cp.k[0], k2 = cp.k[1], p1 = cp.k[2], p2 = cp.k[3], k3 = cp.k[4];
    double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
    double ic = 1.;
    if constexpr (M::rational) ic = 1. / (1 + cp.k[5] * r2 + cp.k[6] * r4 + cp.k[7] * r6);
    double g = cdist * ic;
    double xd = x * g + p1 * a1 + p2 * a2;
    double yd = y * g + p1 * a3 + p2 * a1;
    o.ex = xd * fx + cp.cx - u_obs;
    o.ey = yd * fy + cp.cy - v_obs;
    if constexpr (WANT_JI) {
        double qx = fx * x * ic, qy = fy * y * ic;
        double wx = 0, wy = 0;
        if constexpr (M::rational) wx = -qx * g, wy = -qy * g;
        static_for<0, M::NC>([&](auto ic_) {
            constexpr int i = decltype(ic_)::value;
            constexpr int q = M::col(i);
            if constexpr (q == I_FX) o.jx[i] = xd, o.jy[i] = 0;
            if constexpr (q == I_FY) o.jx[i] = M::aspect ? xd * cp.aspect : 0., o.jy[i] = yd;
            if constexpr (q == I_CX) o.jx[i] = 1, o.jy[i] = 0;
            if constexpr (q == I_CY) o.jx[i] = 0, o.jy[i] = 1;
            if constexpr (q == I_K1) o.jx[i] = qx * r2, o.jy[i] = qy * r2;
            if constexpr (q == I_K2) o.jx[i] = qx * r4, o.jy[i] = qy * r4;
            if constexpr (q == I_P1) o.jx[i] = fx * a1, o.jy[i] = fy * a3;
            if constexpr (q == I_P2) o.jx[i] = fx * a2, o.jy[i] = fy * a1;
            if constexpr (q == I_K3) o.jx[i] = qx * r6, o.jy[i] = qy * r6;
            if constexpr (q == I_K4) o.jx[i] = wx * r2, o.jy[i] = wy * r2;
            if constexpr (q == I_K5) o.jx[i] = wx * r4, o.jy[i] = wy * r4;
            if constexpr (q == I_K6) o.jx[i] = wx * r6, o.jy[i] = wy * r6;
        });
    }
    if constexpr (WANT_T) {
        // g' = d(cdist*ic)/d(r^2); D = d(xd,yd)/d(x,y) is symmetric
        double cd1 = k1 + 2 * k2 * r2 + 3 * k3 * r4;
        double gp = cd1 * ic;
        if constexpr (M::rational) gp -= cdist * ic * ic * (cp.k[5] + 2 * cp.k[6] * r2 + 3 * cp.k[7] * r4);
        double D00 = g + 2 * x * x * gp + 2 * p1 * y + 6 * p2 * x;
        double D01 = a1 * gp + 2 * p1 * x + 2 * p2 * y;
        double D11 = g + 2 * y * y * gp + 6 * p1 * y + 2 * p2 * x;
        double sx = fx * z, sy = fy * z;
        o.Tx[0] = sx * D00, o.Tx[1] = sx * D01;
        o.Ty[0] = sy * D01, o.Ty[1] = sy * D11;
        o.Tx[2] = -(o.Tx[0] * x + o.Tx[1] * y);
        o.Ty[2] = -(o.Ty[0] * x + o.Ty[1] * y);
    }
}

// One point of one view: residual, intrinsic columns (o) and the two rows of the 2x6 pose
// Jacobian (jex, jey: d u, d v / d (r0 r1 r2 t0 t1 t2)), from the 13-double pose form.  With T the
// 2x3 Jacobian of the pixel w.r.t. the camera-frame point,
//   T (dR/dr_j X) = a_j (T.base) + b2 ((a.X) T_j + X_j (T.a)) + b4 (X x T)_j,
//   base = -s X + b1 (a.X) a + b3 (a x X).
template <class M>
__device__ __forceinline__ void point_jacobian(const PoseConst& q, const CamParams& cp, double X, double Y, double Z,
                                               double u, double v, PointOut<M>& o, double jex[6], double jey[6]) {
    double d = q.ax[0] * X + q.ax[1] * Y + q.ax[2] * Z;
    double wx = q.ax[1] * Z - q.ax[2] * Y, wy = q.ax[2] * X - q.ax[0] * Z, wz = q.ax[0] * Y - q.ax[1] * X;
    double c1d = q.c1 * d;
    double Xc = q.t[0] + q.c * X + c1d * q.ax[0] + q.s * wx;
    double Yc = q.t[1] + q.c * Y + c1d * q.ax[1] + q.s * wy;
    double Zc = q.t[2] + q.c * Z + c1d * q.ax[2] + q.s * wz;
    point_eval<M, true, true>(Xc, Yc, Zc, u, v, cp, o);
    double b1d = q.b1 * d, b2d = q.b2 * d;
    double base0 = b1d * q.ax[0] + q.b3 * wx - q.s * X;
    double base1 = b1d * q.ax[1] + q.b3 * wy - q.s * Y;
    double base2 = b1d * q.ax[2] + q.b3 * wz - q.s * Z;
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
}

// Sum over the 64 lanes of a wavefront, returned in every lane.  Within rows of 16 lanes the butterfly
// runs on DPP moves (quad permutes, then the half-row and row mirrors), the four row sums are read out
// with v_readlane: no LDS crossbar traffic and nothing to wait for.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror: every lane of a row now holds the row's sum
    int lo = __double2loint(v), hi = __double2hiint(v);
    double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// Sums N per-lane values over the 64 lanes of a single-wavefront workgroup and stores them to out[0..N):
// the values go through LDS transposed, 32 at a time, and lane k adds up the 64 contributions of value k
// in a fixed order.  sm: 32 * 65 doubles (the odd row length keeps the column reads off one bank).
template <int N>
__device__ __forceinline__ void lanes_reduce_store(const double (&acc)[N], double* sm, double* out) {
    const int lane = threadIdx.x & 63;
    static_for<0, (N + 31) / 32>([&](auto cc) {
        constexpr int c0 = decltype(cc)::value * 32;
        constexpr int n = N - c0 < 32 ? N - c0 : 32;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < n; k++) sm[k * 65 + lane] = acc[c0 + k];
        __builtin_amdgcn_wave_barrier();
        if (lane < n) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                s0 += sm[lane * 65 + j], s1 += sm[lane * 65 + j + 1];
                s2 += sm[lane * 65 + j + 2], s3 += sm[lane * 65 + j + 3];
            }
            out[c0 + lane] = (s0 + s1) + (s2 + s3);
        }
    });
}

// The same reduction for values produced one at a time (so that they need not all be live at once):
// push() parks the value in LDS, and every 32 values lanes 0..31 add up one value's 64 contributions
// each, in a fixed order, and store them to consecutive slots of out.  All indices are compile-time
// constants once the calling loops are unrolled.  sm: 32 * 65 doubles.
struct LaneReducer {
    double* sm;
    double* out;
    int k = 0, base = 0;
    __device__ __forceinline__ LaneReducer(double* sm_, double* out_) : sm(sm_), out(out_) {}
    __device__ __forceinline__ void push(double v) {
        sm[k * 65 + (threadIdx.x & 63)] = v;
        if (++k == 32) flush();
    }
    __device__ __forceinline__ void flush() {
        if (k == 0) return;
        const int lane = threadIdx.x & 63;
        __builtin_amdgcn_wave_barrier();
        if (lane < k) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                s0 += sm[lane * 65 + j], s1 += sm[lane * 65 + j + 1];
                s2 += sm[lane * 65 + j + 2], s3 += sm[lane * 65 + j + 3];
            }
            out[base + lane] = (s0 + s1) + (s2 + s3);
        }
        __builtin_amdgcn_wave_barrier();
        base += k;
        k = 0;
    }
};

// Stores one row of NV doubles per lane to a row-major global array (row r of the wavefront at out + r * NV, rows
// 0..nrows-1 exist) through LDS, so that the wavefront writes the rows' contiguous bytes with full-width stores
// instead of 64 lanes each touching its own cache line NV times.  sm: 64 * (NV + 1) doubles, one wavefront.
template <int NV>
__device__ __forceinline__ void wave_store_rows(const double (&vals)[NV], double* sm, double* out, int nrows) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NV; k++) sm[lane * (NV + 1) + k] = vals[k];
    __builtin_amdgcn_wave_barrier();
    const int total = nrows * NV;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int i = j * 64 + lane;
        if (i < total) out[i] = sm[(i / NV) * (NV + 1) + i % NV];
    }
    __builtin_amdgcn_wave_barrier();
}

// In-place lower Cholesky of a symmetric 6x6 given by its upper triangle packed row-major
// (21 values: (0,0) (0,1) .. (0,5) (1,1) ..).  On return L[i][j] (j <= i) and the reciprocal
// diagonal.  Returns false when the matrix is not positive definite.
__device__ __forceinline__ constexpr int tri6(int a, int b) {  // a <= b
    return a * 6 - a * (a - 1) / 2 + (b - a);
}
__device__ __forceinline__ bool chol6(const double Vu[21], double lambda, double L[21], double invd[6]) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = Vu[tri6(j, i)];
            if (i == j) s *= 1. + lambda;
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[tri6(k, i)] * L[tri6(k, j)];
            if (i == j) {
                ok = ok && (s > 0.);
                double d = sqrt(s);
                L[tri6(i, i)] = d;
                invd[i] = 1. / d;
            } else
                L[tri6(j, i)] = s * invd[j];  // stores L[i][j] at the (j,i) slot
        }
    }
    return ok;
}
// forward substitution  L y = b  (in place)
__device__ __forceinline__ void chol6_fwd(const double L[21], const double invd[6], double b[6]) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[tri6(k, i)] * b[k];
        b[i] = s * invd[i];
    }
}
// backward substitution  L^T x = y  (in place)
__device__ __forceinline__ void chol6_bwd(const double L[21], const double invd[6], double b[6]) {
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= L[tri6(i, k)] * b[k];
        b[i] = s * invd[i];
    }
}

}  // namespace ccalib
