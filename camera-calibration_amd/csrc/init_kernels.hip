// init_kernels.hip -- starting values of the LM loop, computed on the GPU (gfx950, FP64 VALU).
//
// These are the steps OpenCV 4.1.0 performs before its LM loop inside calibrateCamera /
// stereoCalibrate (reference call sites monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317,
// binocular_chessboard.cpp:147-148 and :152-162):
//   cvInitIntrinsicParams2D        focal lengths from the vanishing points of per-view homographies
//   cvFindExtrinsicCameraParams2   planar pose from a homography of the undistorted points, refined by
//                                  a 6-parameter CvLevMarq (<= 20 iterations, eps = FLT_EPSILON)
//   findHomography (method 0)      normalised DLT followed by <= 10 cv::LMSolver iterations
//   stereo                         per-view relative pose of the two cameras (its per-component median
//                                  over views, taken on the host, seeds om_LR, T_LR)
// At the sizes the kernels are built for (1e5..1e6 views) this work, done per view on the host, costs
// seventy times the LM loop itself.
//
// Work decomposition: every step above is a short dependent chain over a view's few dozen points with
// 6x6 .. 9x9 dense algebra in between, so ONE LANE owns one view from start to finish and a wavefront
// owns up to 64 views.  The points of a wavefront's views are one contiguous run of the caller's
// view-major array: they are copied to LDS with coalesced loads (rows of odd length, so that the 64
// lanes of a `point p of my view` read hit distinct banks) and are then read only from there.  All
// small matrices live in registers (fully unrolled packed-triangle Cholesky).
//
// Two deliberate differences from OpenCV's arithmetic, both at round-off level for the well-posed
// systems a calibration view produces: symmetric positive definite systems are solved by Cholesky
// factorisation instead of SVD / eigen decomposition, and the null vector of the 9x9 DLT matrix comes
// from shifted inverse iteration instead of a Jacobi eigen solver.
#include "init_kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_math.hpp"

namespace ccalib {

namespace {

constexpr double kDblEps = 2.220446049250313e-16;
constexpr double kFltEps = 1.1920928955078125e-07;

extern __shared__ float lds_init[];

template <int N>
__device__ __forceinline__ constexpr int tri(int i, int j) {  // i <= j: packed upper triangle, row-major
    return i * N - i * (i - 1) / 2 + (j - i);
}

// In-place Cholesky A = L L^T of a symmetric matrix given by its packed upper triangle: on return slot
// (j, i), j < i, holds L[i][j], the diagonal holds L[i][i] and invd its reciprocal.  A pivot that is not
// above `floor_` is replaced by it, which keeps a (numerically) semi-definite matrix usable.
template <int N>
__device__ __forceinline__ void chol_inplace(double (&a)[N * (N + 1) / 2], double (&invd)[N], double floor_) {
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = a[tri<N>(j, i)];
#pragma unroll
            for (int k = 0; k < j; k++) s -= a[tri<N>(k, i)] * a[tri<N>(k, j)];
            if (i == j) {
                s = s > floor_ ? s : floor_;
                double d = sqrt(s);
                a[tri<N>(i, i)] = d;
                invd[i] = 1. / d;
            } else
                a[tri<N>(j, i)] = s * invd[j];
        }
    }
}
template <int N>
__device__ __forceinline__ void chol_fwd(const double (&L)[N * (N + 1) / 2], const double (&invd)[N], double (&b)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[tri<N>(k, i)] * b[k];
        b[i] = s * invd[i];
    }
}
template <int N>
__device__ __forceinline__ void chol_bwd(const double (&L)[N * (N + 1) / 2], const double (&invd)[N], double (&b)[N]) {
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= L[tri<N>(i, k)] * b[k];
        b[i] = s * invd[i];
    }
}

struct Mat3 {
    double a[9];
};
__device__ __forceinline__ Mat3 mul3(const Mat3& x, const Mat3& y) {
    Mat3 z;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) z.a[i * 3 + j] = x.a[i * 3] * y.a[j] + x.a[i * 3 + 1] * y.a[3 + j] + x.a[i * 3 + 2] * y.a[6 + j];
    return z;
}
__device__ __forceinline__ double det3(const Mat3& m) {
    return m.a[0] * (m.a[4] * m.a[8] - m.a[5] * m.a[7]) - m.a[1] * (m.a[3] * m.a[8] - m.a[5] * m.a[6]) +
           m.a[2] * (m.a[3] * m.a[7] - m.a[4] * m.a[6]);
}

// cvRodrigues2 matrix -> vector.  OpenCV first replaces R by the U V^T of its SVD; the same nearest
// rotation is reached here by Newton's iteration for the polar factor, R <- (R + R^-T) / 2.
__device__ void rodrigues_m2v_polar(Mat3 R, double r[3]) {
    for (int it = 0; it < 20; it++) {
        double d = det3(R);
        if (d == 0) break;
        double s = 1. / d;
        Mat3 inv;  // R^-1
        inv.a[0] = (R.a[4] * R.a[8] - R.a[5] * R.a[7]) * s;
        inv.a[1] = (R.a[2] * R.a[7] - R.a[1] * R.a[8]) * s;
        inv.a[2] = (R.a[1] * R.a[5] - R.a[2] * R.a[4]) * s;
        inv.a[3] = (R.a[5] * R.a[6] - R.a[3] * R.a[8]) * s;
        inv.a[4] = (R.a[0] * R.a[8] - R.a[2] * R.a[6]) * s;
        inv.a[5] = (R.a[2] * R.a[3] - R.a[0] * R.a[5]) * s;
        inv.a[6] = (R.a[3] * R.a[7] - R.a[4] * R.a[6]) * s;
        inv.a[7] = (R.a[1] * R.a[6] - R.a[0] * R.a[7]) * s;
        inv.a[8] = (R.a[0] * R.a[4] - R.a[1] * R.a[3]) * s;
        double chg = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double nv = 0.5 * (R.a[i * 3 + j] + inv.a[j * 3 + i]);
                chg = fmax(chg, fabs(nv - R.a[i * 3 + j]));
                R.a[i * 3 + j] = nv;
            }
        if (chg < 1e-16) break;
    }
    double x = R.a[7] - R.a[5], y = R.a[2] - R.a[6], z = R.a[3] - R.a[1];
    double s = sqrt((x * x + y * y + z * z) * 0.25), c = (R.a[0] + R.a[4] + R.a[8] - 1) * 0.5;
    c = fmin(1., fmax(-1., c));
    double th = acos(c);
    if (s < 1e-5) {
        if (c > 0)
            x = y = z = 0;
        else {
            x = sqrt(fmax((R.a[0] + 1) * 0.5, 0.));
            y = sqrt(fmax((R.a[4] + 1) * 0.5, 0.)) * (R.a[1] < 0 ? -1. : 1.);
            z = sqrt(fmax((R.a[8] + 1) * 0.5, 0.)) * (R.a[2] < 0 ? -1. : 1.);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R.a[5] > 0) != (y * z > 0)) z = -z;
            double sc = th / sqrt(x * x + y * y + z * z);
            x *= sc, y *= sc, z *= sc;
        }
    } else {
        double sc = th / (2 * s);
        x *= sc, y *= sc, z *= sc;
    }
    r[0] = x, r[1] = y, r[2] = z;
}

// One Jacobi rotation of the symmetric 3x3 m (full storage) in the (P, Q) plane; V accumulates the
// eigenvectors as ROWS.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot3(double (&m)[9], double (&V)[9]) {
    constexpr int K = 3 - P - Q;
    double apq = m[P * 3 + Q];
    if (fabs(apq) <= 1e-300) return;
    double theta = (m[Q * 3 + Q] - m[P * 3 + P]) / (2 * apq);
    double t = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1));
    double c = 1. / sqrt(t * t + 1), s = t * c;
    double app = m[P * 3 + P] - t * apq, aqq = m[Q * 3 + Q] + t * apq;
    double akp = c * m[K * 3 + P] - s * m[K * 3 + Q], akq = s * m[K * 3 + P] + c * m[K * 3 + Q];
    m[P * 3 + P] = app, m[Q * 3 + Q] = aqq, m[P * 3 + Q] = m[Q * 3 + P] = 0;
    m[K * 3 + P] = m[P * 3 + K] = akp, m[K * 3 + Q] = m[Q * 3 + K] = akq;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double vp = V[P * 3 + k], vq = V[Q * 3 + k];
        V[P * 3 + k] = c * vp - s * vq, V[Q * 3 + k] = s * vp + c * vq;
    }
}

// Sum over the L consecutive lanes that share a view, returned in all of them: the butterfly adds the same two
// numbers in both lanes of a pair, so every lane of the group ends up with the same bits and takes the same branches.
template <int L>
__device__ __forceinline__ double group_sum(double v) {
    static_assert(L == 1 || L == 2 || L == 4 || L == 8 || L == 16 || L == 64, "lanes per view");
    if (L == 64) return wave_sum(v);
    if (L >= 2) v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    if (L >= 4) v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    if (L >= 8) v += dpp_move<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of each half row
    if (L >= 16) v += dpp_move<0x140>(v);  // row_mirror: lane i <-> 15 - i of each row
    return v;
}

// The same for the maximum.
template <int L>
__device__ __forceinline__ double group_max(double v) {
    if (L == 64) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
        return v;
    }
    if (L >= 2) v = fmax(v, dpp_move<0xB1>(v));
    if (L >= 4) v = fmax(v, dpp_move<0x4E>(v));
    if (L >= 8) v = fmax(v, dpp_move<0x141>(v));
    if (L >= 16) v = fmax(v, dpp_move<0x140>(v));
    return v;
}

// The frame cvFindExtrinsicCameraParams2 moves a planar rig into: rows of Rt are the principal axes of
// the object points (largest spread first), Tt = -Rt * centroid.  Returns false for a non-planar rig.
// (L lanes share the view, as in refine_pose: lane `sub` takes points sub, sub + L, ...; every sum over points is
// completed over the group, so all L lanes hold the same numbers.  The same holds for the homography functions below.)
template <int L, class Obj>
__device__ bool board_frame(int n, Obj obj, Mat3& Rt, double Tt[3], int sub) {
    double Mc[3] = {0, 0, 0};
    for (int p = sub; p < n; p += L) {
        float3 o = obj(p);
        Mc[0] += o.x, Mc[1] += o.y, Mc[2] += o.z;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) Mc[k] = group_sum<L>(Mc[k]) / n;
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = sub; p < n; p += L) {
        float3 o = obj(p);
        double d[3] = {o.x - Mc[0], o.y - Mc[1], o.z - Mc[2]};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = a; b < 3; b++) m[a * 3 + b] += d[a] * d[b];
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = a; b < 3; b++) m[a * 3 + b] = group_sum<L>(m[a * 3 + b]);
    m[3] = m[1], m[6] = m[2], m[7] = m[5];
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 8; sweep++) {
        jacobi_rot3<0, 1>(m, V);
        jacobi_rot3<0, 2>(m, V);
        jacobi_rot3<1, 2>(m, V);
    }
    double w[3] = {m[0], m[4], m[8]};
    // sort descending (rows of V follow)
    auto swp = [&](int i, int j) {
        if (w[i] < w[j]) {
            double t = w[i];
            w[i] = w[j], w[j] = t;
#pragma unroll
            for (int k = 0; k < 3; k++) t = V[i * 3 + k], V[i * 3 + k] = V[j * 3 + k], V[j * 3 + k] = t;
        }
    };
    swp(0, 1), swp(0, 2), swp(1, 2);
    if (!(w[2] / w[1] < 1e-3)) return false;
#pragma unroll
    for (int k = 0; k < 9; k++) Rt.a[k] = V[k];
    if (V[2] * V[2] + V[5] * V[5] < 1e-10) Rt = Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    if (det3(Rt) < 0)
#pragma unroll
        for (int k = 0; k < 9; k++) Rt.a[k] = -Rt.a[k];
#pragma unroll
    for (int i = 0; i < 3; i++) Tt[i] = -(Rt.a[i * 3] * Mc[0] + Rt.a[i * 3 + 1] * Mc[1] + Rt.a[i * 3 + 2] * Mc[2]);
    return true;
}

// ---- findHomography(method 0) ----------------------------------------------------------------
// Transfer residuals of the 8-parameter homography h (h22 = 1): sum of squares, and with WANT_J the
// normal equations A = J^T J (packed upper 8x8), v = J^T r and max |r|.
template <bool WANT_J, int L, class Src, class Dst>
__device__ __forceinline__ double homography_eval(int n, Src src, Dst dst, const double (&h)[8], double (&A)[36],
                                                  double (&v)[8], double& rmax, int sub) {
    double S = 0;
    if constexpr (WANT_J) {
#pragma unroll
        for (int k = 0; k < 36; k++) A[k] = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = 0;
        rmax = 0;
    }
    for (int p = sub; p < n; p += L) {
        float2 M = src(p), m = dst(p);
        double X = M.x, Y = M.y;
        double w = h[6] * X + h[7] * Y + 1.;
        w = fabs(w) > kDblEps ? 1. / w : 0;
        double xi = (h[0] * X + h[1] * Y + h[2]) * w, yi = (h[3] * X + h[4] * Y + h[5]) * w;
        double rx = xi - (double)m.x, ry = yi - (double)m.y;
        S += rx * rx + ry * ry;
        if constexpr (WANT_J) {
            rmax = fmax(rmax, fmax(fabs(rx), fabs(ry)));
            // rows (Xw, Yw, w, 0, 0, 0, -Xw xi, -Yw xi) and (0, 0, 0, Xw, Yw, w, -Xw yi, -Yw yi)
            double g[3] = {X * w, Y * w, w};
            double ax[2] = {-g[0] * xi, -g[1] * xi}, ay[2] = {-g[0] * yi, -g[1] * yi};
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = a; b < 3; b++) {
                    double gg = g[a] * g[b];
                    A[tri<8>(a, b)] += gg;
                    A[tri<8>(3 + a, 3 + b)] += gg;
                }
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    A[tri<8>(a, 6 + b)] += g[a] * ax[b];
                    A[tri<8>(3 + a, 6 + b)] += g[a] * ay[b];
                }
                v[a] += g[a] * rx;
                v[3 + a] += g[a] * ry;
            }
#pragma unroll
            for (int a = 0; a < 2; a++) {
#pragma unroll
                for (int b = a; b < 2; b++) A[tri<8>(6 + a, 6 + b)] += ax[a] * ax[b] + ay[a] * ay[b];
                v[6 + a] += ax[a] * rx + ay[a] * ry;
            }
        }
    }
    if constexpr (WANT_J) {
#pragma unroll
        for (int k = 0; k < 36; k++) A[k] = group_sum<L>(A[k]);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = group_sum<L>(v[k]);
        rmax = group_max<L>(rmax);
    }
    return group_sum<L>(S);
}

// cv::LMSolver::run(maxIters = 10, eps = FLT_EPSILON) on the transfer error, as findHomography polishes
// its DLT estimate.
template <int LN, class Src, class Dst>
__device__ void homography_polish(int n, Src src, Dst dst, double (&x)[8], int sub) {
    double A[36], v[8], D[8], xd[8], d[8], L[36], invd[8], rmax = 0, dummy = 0;
    double S = homography_eval<true, LN>(n, src, dst, x, A, v, rmax, sub);
#pragma unroll
    for (int i = 0; i < 8; i++) D[i] = A[tri<8>(i, i)];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    for (int iter = 0;;) {
#pragma unroll
        for (int k = 0; k < 36; k++) L[k] = A[k];
        double mx = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            L[tri<8>(i, i)] += lambda * D[i];
            mx = fmax(mx, L[tri<8>(i, i)]);
            d[i] = v[i];
        }
        chol_inplace<8>(L, invd, mx * 1e-30);
        chol_fwd<8>(L, invd, d);
        chol_bwd<8>(L, invd, d);
#pragma unroll
        for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
        double Sd = homography_eval<false, LN>(n, src, dst, xd, L, invd, dummy, sub);
        double dS = 0, t = 0, dmax = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            double Ad = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) Ad += A[i <= j ? tri<8>(i, j) : tri<8>(j, i)] * d[j];
            dS += d[i] * (2 * v[i] - Ad);
            t += d[i] * v[i];
            dmax = fmax(dmax, fabs(d[i]));
        }
        double R = (S - Sd) / (fabs(dS) > kDblEps ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double nu = (Sd - S) / (fabs(t) > kDblEps ? t : 1) + 2;
            nu = fmin(fmax(nu, 2.), 10.);
            if (lambda == 0) {
                // largest |diagonal| of A^-1 = L^-T L^-1: column i of L^-1 by forward substitution
#pragma unroll
                for (int k = 0; k < 36; k++) L[k] = A[k];
                double m2 = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) m2 = fmax(m2, A[tri<8>(i, i)]);
                chol_inplace<8>(L, invd, m2 * 1e-30);
                double maxval = kDblEps;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    double e[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) e[k] = k == i ? 1. : 0.;
                    chol_fwd<8>(L, invd, e);
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) s += e[k] * e[k];  // |L^-1 e_i|^2 = (A^-1)_ii
                    maxval = fmax(maxval, fabs(s));
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = xd[i];
            homography_eval<true, LN>(n, src, dst, x, A, v, rmax, sub);
        }
        iter++;
        if (!(iter < 10 && dmax >= kFltEps && rmax >= kFltEps)) break;
    }
}

template <int L, class Src, class Dst>
__device__ bool find_homography(int n, Src src, Dst dst, Mat3& H, int sub) {
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int p = sub; p < n; p += L) {
        float2 M = src(p), m = dst(p);
        cM[0] += M.x, cM[1] += M.y, cm[0] += m.x, cm[1] += m.y;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) cM[k] = group_sum<L>(cM[k]) / n, cm[k] = group_sum<L>(cm[k]) / n;
    for (int p = sub; p < n; p += L) {
        float2 M = src(p), m = dst(p);
        sM[0] += fabs(M.x - cM[0]), sM[1] += fabs(M.y - cM[1]);
        sm[0] += fabs(m.x - cm[0]), sm[1] += fabs(m.y - cm[1]);
    }
#pragma unroll
    for (int k = 0; k < 2; k++) sM[k] = group_sum<L>(sM[k]), sm[k] = group_sum<L>(sm[k]);
    if (fabs(sM[0]) < kDblEps || fabs(sM[1]) < kDblEps || fabs(sm[0]) < kDblEps || fabs(sm[1]) < kDblEps) return false;
    sM[0] = n / sM[0], sM[1] = n / sM[1], sm[0] = n / sm[0], sm[1] = n / sm[1];
    // L^T L of the 2n x 9 DLT matrix with rows (X, Y, 1, 0, 0, 0, -xX, -xY, -x), (0, 0, 0, X, Y, 1, -yX, -yY, -y):
    // four symmetric 3x3 blocks  [ A 0 Bx ; 0 A By ; Bx By C ]  over q = (XX, XY, X, YY, Y, 1)
    double qa[6] = {0, 0, 0, 0, 0, 0}, qx[6] = {0, 0, 0, 0, 0, 0}, qy[6] = {0, 0, 0, 0, 0, 0}, qc[6] = {0, 0, 0, 0, 0, 0};
    for (int p = sub; p < n; p += L) {
        float2 M = src(p), m = dst(p);
        double X = (M.x - cM[0]) * sM[0], Y = (M.y - cM[1]) * sM[1];
        double x = (m.x - cm[0]) * sm[0], y = (m.y - cm[1]) * sm[1];
        double q[6] = {X * X, X * Y, X, Y * Y, Y, 1.};
        double rr = x * x + y * y;
#pragma unroll
        for (int k = 0; k < 6; k++) qa[k] += q[k], qx[k] -= x * q[k], qy[k] -= y * q[k], qc[k] += rr * q[k];
    }
#pragma unroll
    for (int k = 0; k < 6; k++)
        qa[k] = group_sum<L>(qa[k]), qx[k] = group_sum<L>(qx[k]), qy[k] = group_sum<L>(qy[k]), qc[k] = group_sum<L>(qc[k]);
    double a[45], invd[9], tr = 0;
    auto s3 = [](int i, int j) { return i <= j ? i * 3 - i * (i - 1) / 2 + (j - i) : j * 3 - j * (j - 1) / 2 + (i - j); };
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = i; j < 9; j++) {
            int bi = i / 3, bj = j / 3, k = s3(i % 3, j % 3);
            double val = 0;
            if (bi == bj) val = bi < 2 ? qa[k] : qc[k];
            else if (bj == 2) val = bi == 0 ? qx[k] : qy[k];
            a[tri<9>(i, j)] = val;
        }
#pragma unroll
    for (int i = 0; i < 9; i++) tr += a[tri<9>(i, i)];
    // null vector by inverse iteration on A + shift I (same eigenvectors, positive definite)
    const double shift = tr * 1e-12;
#pragma unroll
    for (int i = 0; i < 9; i++) a[tri<9>(i, i)] += shift;
    chol_inplace<9>(a, invd, tr * 1e-16);
    double b[9] = {0.31, -0.52, 0.73, 0.44, -0.65, 0.86, -0.17, 0.28, 0.99};
    for (int it = 0; it < 8; it++) {
        chol_fwd<9>(a, invd, b);
        chol_bwd<9>(a, invd, b);
        double nn = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nn += b[i] * b[i];
        nn = 1. / sqrt(nn);
#pragma unroll
        for (int i = 0; i < 9; i++) b[i] *= nn;
    }
    Mat3 H0, invHn = {{1 / sm[0], 0, cm[0], 0, 1 / sm[1], cm[1], 0, 0, 1}},
             Hn2 = {{sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1}};
#pragma unroll
    for (int k = 0; k < 9; k++) H0.a[k] = b[k];
    Mat3 Hf = mul3(mul3(invHn, H0), Hn2);
    double sc = 1. / Hf.a[8];
#pragma unroll
    for (int k = 0; k < 9; k++) H.a[k] = Hf.a[k] * sc;
    if (n > 4) {
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = H.a[k];
        homography_polish<L>(n, src, dst, x, sub);
#pragma unroll
        for (int k = 0; k < 8; k++) H.a[k] = x[k];
        H.a[8] = 1;
    }
    return true;
}

// ---- LDS staging --------------------------------------------------------------------------------
// Layout (floats): [object points: 3 * board_np (shared board) | VPB rows of 3 * PS]
//                  [source points float2: board_np | VPB rows of PS]   (pose kernel only)
//                  [image points float2: VPB rows of PS]
struct Stage {
    float* obj;
    float2* src;
    float2* img;
    int PS;
};
__device__ __forceinline__ Stage stage_layout(const InitArgs& a, bool with_src) {
    Stage s;
    s.PS = a.PS;
    size_t nobj = a.shared_board ? (size_t)(3 * a.board_np + 1) / 2 * 2 : (size_t)a.VPB * 3 * a.PS + (a.VPB * a.PS & 1);
    size_t nsrc = with_src ? (a.shared_board ? (size_t)a.board_np : (size_t)a.VPB * a.PS) : 0;
    s.obj = lds_init;
    s.src = reinterpret_cast<float2*>(lds_init + nobj);
    s.img = s.src + nsrc;
    return s;
}
// copies the image points of the workgroup's nv views (view_of(j), j < nv) to the stage; every lane of the
// wavefront takes part
template <class ViewOf>
__device__ __forceinline__ void stage_img(const InitArgs& a, const Stage& s, ViewOf view_of, int nv) {
    const int lane = threadIdx.x;
    for (int j = 0; j < nv; j++) {
        int v = view_of(j);
        int first = a.off[v], n = a.off[v + 1] - first;
        for (int p = lane; p < n; p += 64) s.img[j * s.PS + p] = a.img[first + p];
    }
}
template <class ViewOf>
__device__ __forceinline__ void stage_obj(const InitArgs& a, const Stage& s, ViewOf view_of, int nv) {
    const int lane = threadIdx.x;
    if (a.shared_board) {
        for (int i = lane; i < 3 * a.board_np; i += 64) s.obj[i] = a.obj[i];
    } else {
        for (int j = 0; j < nv; j++) {
            int v = view_of(j);
            int first = a.off[v], n = a.off[v + 1] - first;
            for (int i = lane; i < 3 * n; i += 64) s.obj[j * 3 * s.PS + i] = a.obj[3 * (size_t)first + i];
        }
    }
}

// cvUndistortPoints' fixed-point iteration (5 rounds) to normalised coordinates, rounded to float
// because findHomography works on Point2f.
__device__ __forceinline__ float2 undistort(float2 m, const CamParams& cp) {
    double x0 = ((double)m.x - cp.cx) / cp.fx, y0 = ((double)m.y - cp.cy) / cp.fy, x = x0, y = y0;
    const double k1 = cp.k[0], k2 = cp.k[1], p1 = cp.k[2], p2 = cp.k[3], k3 = cp.k[4], k4 = cp.k[5], k5 = cp.k[6], k6 = cp.k[7];
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icd = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - dx) * icd, y = (y0 - dy) * icd;
    }
    return make_float2((float)x, (float)y);
}

}  // namespace

// CvLevMarq(6 parameters) with criteria (COUNT + EPS, 20, FLT_EPSILON) on the reprojection error of one view:
// the refinement at the end of cvFindExtrinsicCameraParams2.  L lanes share a view: lane `sub` of the group takes
// points sub, sub + L, ... and the sums over points are completed with group_sum, so all L lanes carry the same
// pose and state.  Most views are done after five iterations, one in a hundred needs all twenty, and a wavefront
// runs as long as its slowest view: so the loop can be left at an iteration boundary (stop_at) with its state --
// pose, damping exponent, iteration count, residual norm -- and picked up later by a wavefront filled with such
// views only (fresh = false).  Returns true when done.
template <class M, int L, class Obj>
__device__ bool refine_pose(int n, Obj obj, const float2* irow, const CamParams& cp, double (&ps)[6], int& lambdaLg10,
                            int& iters, double& errNorm, bool fresh, int stop_at, int sub) {
    double JtJ[21], Jte[6], prev[6];
    auto eval_err = [&]() {
        double R[9], e2 = 0;
        rodrigues<false>(ps, R, nullptr);
        for (int p = sub; p < n; p += L) {
            float3 o = obj(p);
            float2 m = irow[p];
            double X = o.x, Y = o.y, Z = o.z;
            double Xc = R[0] * X + R[1] * Y + R[2] * Z + ps[3];
            double Yc = R[3] * X + R[4] * Y + R[5] * Z + ps[4];
            double Zc = R[6] * X + R[7] * Y + R[8] * Z + ps[5];
            PointOut<M> o2;
            point_eval<M, false, false>(Xc, Yc, Zc, (double)m.x, (double)m.y, cp, o2);
            e2 += o2.ex * o2.ex + o2.ey * o2.ey;
        }
        return group_sum<L>(e2);
    };
    auto eval_jac = [&]() {
        const PoseConst q = pose_const(ps);
        double e2 = 0;
#pragma unroll
        for (int k = 0; k < 21; k++) JtJ[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) Jte[k] = 0;
        for (int p = sub; p < n; p += L) {
            float3 o = obj(p);
            float2 m = irow[p];
            PointOut<M> o2;
            double jex[6], jey[6];
            point_jacobian<M>(q, cp, (double)o.x, (double)o.y, (double)o.z, (double)m.x, (double)m.y, o2, jex, jey);
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = i; j < 6; j++) JtJ[tri6(i, j)] += jex[i] * jex[j] + jey[i] * jey[j];
                Jte[i] += jex[i] * o2.ex + jey[i] * o2.ey;
            }
            e2 += o2.ex * o2.ex + o2.ey * o2.ey;
        }
#pragma unroll
        for (int k = 0; k < 21; k++) JtJ[k] = group_sum<L>(JtJ[k]);
#pragma unroll
        for (int k = 0; k < 6; k++) Jte[k] = group_sum<L>(Jte[k]);
        return group_sum<L>(e2);
    };
    auto step = [&]() {
        double lambda = exp(lambdaLg10 * 2.302585092994046);
        double Lc[21], invd[6], x[6];
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = Jte[k];
        bool pd = chol6(JtJ, lambda, Lc, invd);
        chol6_fwd(Lc, invd, x);
        chol6_bwd(Lc, invd, x);
#pragma unroll
        for (int k = 0; k < 6; k++) ps[k] = prev[k] - (pd ? x[k] : 0.);
    };
    double e0 = sqrt(eval_jac());
    double prevErrNorm = fresh ? e0 : errNorm;  // resumed: the norm the loop was left with (the same point)
    for (;;) {
#pragma unroll
        for (int k = 0; k < 6; k++) prev[k] = ps[k];
        step();
        errNorm = sqrt(eval_err());
        while (errNorm > prevErrNorm && ++lambdaLg10 <= 16) {
            step();
            errNorm = sqrt(eval_err());
        }
        lambdaLg10 = max(lambdaLg10 - 1, -16);
        double num = 0, den = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) num += (ps[k] - prev[k]) * (ps[k] - prev[k]), den += prev[k] * prev[k];
        if (++iters >= 20 || sqrt(num) / sqrt(den) < kFltEps) return true;
        if (iters >= stop_at) return false;
        prevErrNorm = errNorm;
        eval_jac();
    }
}

// ---- cvFindExtrinsicCameraParams2 for every view ---------------------------------------------
// Lanes per view in the homography kernels (k_init_pose, k_init_intrinsics): the sums over a view's points can be split
// over L lanes (everything else -- the 9x9 and 8x8 factorisations, the polar iteration -- each lane then does for
// itself).  Measured with L = 4 at 100 000 views: k_init_pose 436 -> 349 us, k_init_intrinsics 417 -> 295 us; a whole
// calibrateCamera call at 1 000 views 2.52 -> 2.32 ms, at 20 views 2.2 -> 1.8 ms.  But findHomography's polish stops
// on steps of FLT_EPSILON size, so a homography is only determined to about 1e-10 by its input, and with the sums taken
// in another order than a serial loop takes them the starting poses move by that much: the residual sum of the first LM
// iteration then agrees with a serial implementation to 1.3e-9 instead of 1e-10 on a 30-view input.  One lane keeps
// the serial order.  -DCCALIB_INIT_LANES=4 builds the split version.
#ifndef CCALIB_INIT_LANES
#define CCALIB_INIT_LANES 1
#endif
constexpr int kInitLanes = CCALIB_INIT_LANES;

template <bool RAT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_init_pose(InitArgs a, CamParams cp, double* __restrict__ pose, int* __restrict__ status, Parked parked) {
    constexpr int L = kInitLanes;
    const int lane = threadIdx.x, sub = lane % L, slot = lane / L;
    const int v0 = blockIdx.x * a.VPB;
    const int nv = min(a.VPB, a.nviews - v0);
    const Stage st = stage_layout(a, true);
    const bool live = slot < nv;
    const int row = live ? slot : 0;  // idle lanes shadow view 0 of the workgroup and store nothing
    const int v = v0 + row;
    const int n = a.off[v + 1] - a.off[v];
    auto view_of = [&](int j) { return v0 + j; };
    stage_img(a, st, view_of, nv);
    stage_obj(a, st, view_of, nv);
    __syncthreads();
    const float* ob = a.shared_board ? st.obj : st.obj + row * 3 * st.PS;
    auto obj = [&](int p) { return make_float3(ob[3 * p], ob[3 * p + 1], ob[3 * p + 2]); };
    Mat3 Rt;
    double Tt[3];
    bool planar = board_frame<L>(n, obj, Rt, Tt, sub);
    // source points of the homography: the rig in its own plane, rounded to float (findHomography
    // converts its inputs to Point2f)
    float2* srow = a.shared_board ? st.src : st.src + row * st.PS;
    if (a.shared_board) {
        for (int p = lane; p < n; p += 64) {
            float3 o = obj(p);
            srow[p] = make_float2((float)(Rt.a[0] * o.x + Rt.a[1] * o.y + Rt.a[2] * o.z + Tt[0]),
                                  (float)(Rt.a[3] * o.x + Rt.a[4] * o.y + Rt.a[5] * o.z + Tt[1]));
        }
    } else if (live) {
        for (int p = sub; p < n; p += L) {
            float3 o = obj(p);
            srow[p] = make_float2((float)(Rt.a[0] * o.x + Rt.a[1] * o.y + Rt.a[2] * o.z + Tt[0]),
                                  (float)(Rt.a[3] * o.x + Rt.a[4] * o.y + Rt.a[5] * o.z + Tt[1]));
        }
    }
    // image points -> undistorted normalised points, in place
    for (int j = 0; j < nv; j++) {
        int nj = a.off[v0 + j + 1] - a.off[v0 + j];
        for (int p = lane; p < nj; p += 64) st.img[j * st.PS + p] = undistort(st.img[j * st.PS + p], cp);
    }
    __syncthreads();
    const float2* irow = st.img + row * st.PS;
    auto src = [&](int p) { return srow[p]; };
    auto dst = [&](int p) { return irow[p]; };
    Mat3 H, Rm;
    double ps[6];
    bool ok = planar && find_homography<L>(n, src, dst, H, sub);
#pragma unroll
    for (int k = 0; k < 9; k++) ok = ok && isfinite(H.a[k]);
    if (ok) {
        double n1 = sqrt(H.a[0] * H.a[0] + H.a[3] * H.a[3] + H.a[6] * H.a[6]);
        double n2 = sqrt(H.a[1] * H.a[1] + H.a[4] * H.a[4] + H.a[7] * H.a[7]);
        double tt[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            tt[i] = H.a[i * 3 + 2] * (2. / fmax(n1 + n2, kDblEps));
            H.a[i * 3] /= fmax(n1, kDblEps);
            H.a[i * 3 + 1] /= fmax(n2, kDblEps);
        }
        H.a[2] = H.a[3] * H.a[7] - H.a[6] * H.a[4];
        H.a[5] = H.a[6] * H.a[1] - H.a[0] * H.a[7];
        H.a[8] = H.a[0] * H.a[4] - H.a[3] * H.a[1];
        double rr[3];
        rodrigues_m2v_polar(H, rr);
        rodrigues<false>(rr, H.a, nullptr);
#pragma unroll
        for (int i = 0; i < 3; i++) ps[3 + i] = H.a[i * 3] * Tt[0] + H.a[i * 3 + 1] * Tt[1] + H.a[i * 3 + 2] * Tt[2] + tt[i];
        Rm = mul3(H, Rt);
    } else {
        Rm = Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
        ps[3] = ps[4] = ps[5] = 0;
    }
    rodrigues_m2v_polar(Rm, ps);
    if (live && sub == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) pose[6 * (size_t)v + k] = planar ? ps[k] : 0.;
        status[v] = planar ? 0 : 1;
    }
}

// The refinement of the poses k_init_pose left (RESUME = false: every view, parked after a.cut iterations if still
// running) or of the views an earlier launch parked (RESUME = true: gathered, run to the end).  L lanes per view,
// 64 / L views per wavefront (a.VPB of them staged; lanes past that shadow the workgroup's first view and store
// nothing).  The resuming grid covers the worst case; workgroups past the parked count leave.
#ifndef CCALIB_REFINE_WAVES
#define CCALIB_REFINE_WAVES 2  // wavefronts per SIMD the register budget of the refinement is set for (3: spills, no faster)
#endif
template <bool RAT, int L, bool RESUME>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CCALIB_REFINE_WAVES, CCALIB_REFINE_WAVES))) void k_refine_pose(InitArgs a, CamParams cp, double* __restrict__ pose, int* __restrict__ status,
                                                    Parked parked) {
    using M = Model<0u, false, RAT>;
    const int lane = threadIdx.x, sub = lane % L, slot = lane / L;
    const int total = RESUME ? *parked.count : a.nviews;
    const int j0 = blockIdx.x * a.VPB;
    if (j0 >= total) return;
    const int nv = min(a.VPB, total - j0);
    const Stage st = stage_layout(a, false);
    const bool live = slot < nv;
    const int row = live ? slot : 0;
    auto view_of = [&](int j) { return RESUME ? parked.view[j0 + j] : j0 + j; };
    const int v = view_of(row);
    const int n = a.off[v + 1] - a.off[v];
    stage_img(a, st, view_of, nv);
    stage_obj(a, st, view_of, nv);
    __syncthreads();
    const float* ob = a.shared_board ? st.obj : st.obj + row * 3 * st.PS;
    auto obj = [&](int p) { return make_float3(ob[3 * p], ob[3 * p + 1], ob[3 * p + 2]); };
    const float2* irow = st.img + row * st.PS;
    const double* d = RESUME ? parked.state + kParkedDoubles * (size_t)(j0 + row) : pose + 6 * (size_t)v;
    double ps[6];
#pragma unroll
    for (int k = 0; k < 6; k++) ps[k] = d[k];
    double errNorm = RESUME ? d[6] : 0.;
    int lambdaLg10 = RESUME ? (int)d[7] : -3, iters = RESUME ? (int)d[8] : 0;
    if (!RESUME && status[v] != 0) return;  // no planar rig in this view: the zero pose stands (all L lanes leave together)
    const bool done = refine_pose<M, L>(n, obj, irow, cp, ps, lambdaLg10, iters, errNorm, !RESUME, RESUME ? 1 << 30 : a.cut, sub);
    if (!live || sub != 0) return;
    if (done) {
#pragma unroll
        for (int k = 0; k < 6; k++) pose[6 * (size_t)v + k] = ps[k];
    } else {  // parked: a later wavefront of slow views takes it from here
        int at = atomicAdd(parked.count, 1);
        parked.view[at] = v;
        double* o = parked.state + kParkedDoubles * (size_t)at;
#pragma unroll
        for (int k = 0; k < 6; k++) o[k] = ps[k];
        o[6] = errNorm, o[7] = lambdaLg10, o[8] = iters;
    }
}

// ---- cvInitIntrinsicParams2D: the two rows every view contributes to the 2-unknown least squares for
// (1/fx^2, 1/fy^2); part receives per workgroup the sums  a0 a0, a0 a1, a1 a1, a0 b, a1 b.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_init_intrinsics(InitArgs a, double cx, double cy,
                                                                                                    double* __restrict__ part) {
    constexpr int L = kInitLanes;
    const int lane = threadIdx.x, sub = lane % L, slot = lane / L;
    const int v0 = blockIdx.x * a.VPB;
    const int nv = min(a.VPB, a.nviews - v0);
    const Stage st = stage_layout(a, false);
    const bool live = slot < nv;
    const int row = live ? slot : 0;
    const int v = v0 + row;
    const int n = a.off[v + 1] - a.off[v];
    auto view_of = [&](int j) { return v0 + j; };
    stage_img(a, st, view_of, nv);
    stage_obj(a, st, view_of, nv);
    __syncthreads();
    const float* ob = a.shared_board ? st.obj : st.obj + row * 3 * st.PS;
    const float2* irow = st.img + row * st.PS;
    auto src = [&](int p) { return make_float2(ob[3 * p], ob[3 * p + 1]); };
    auto dst = [&](int p) { return irow[p]; };
    Mat3 H;
    bool ok = find_homography<L>(n, src, dst, H, sub) && live && sub == 0;  // one lane of the group speaks for the view
#pragma unroll
    for (int c = 0; c < 3; c++) H.a[c] -= H.a[6 + c] * cx, H.a[3 + c] -= H.a[6 + c] * cy;
    double h[3], vv[3], d1[3], d2[3], nn[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double t0 = H.a[j * 3], t1 = H.a[j * 3 + 1];
        h[j] = t0, vv[j] = t1, d1[j] = (t0 + t1) * 0.5, d2[j] = (t0 - t1) * 0.5;
        nn[0] += t0 * t0, nn[1] += t1 * t1, nn[2] += d1[j] * d1[j], nn[3] += d2[j] * d2[j];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) nn[k] = 1. / sqrt(nn[k]);
#pragma unroll
    for (int j = 0; j < 3; j++) h[j] *= nn[0], vv[j] *= nn[1], d1[j] *= nn[2], d2[j] *= nn[3];
    double A0[2] = {h[0] * vv[0], d1[0] * d2[0]}, A1[2] = {h[1] * vv[1], d1[1] * d2[1]}, B[2] = {-h[2] * vv[2], -d1[2] * d2[2]};
    double sums[5] = {0, 0, 0, 0, 0};
    if (ok) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            sums[0] += A0[k] * A0[k], sums[1] += A0[k] * A1[k], sums[2] += A1[k] * A1[k];
            sums[3] += A0[k] * B[k], sums[4] += A1[k] * B[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double s = wave_sum(sums[k]);
        if (lane == 0) part[5 * (size_t)blockIdx.x + k] = s;
    }
}

// ---- stereo: relative pose of camera 1 w.r.t. camera 0 in every view -------------------------
// out: six rows of npad doubles (om_x om_y om_z T_x T_y T_z), R = R1 R0^T, T = t1 - R t0.
__global__ __launch_bounds__(64) void k_relative_pose(const double* __restrict__ pose0, const double* __restrict__ pose1,
                                                      int nviews, int npad, double* __restrict__ out) {
    int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= nviews) return;
    double p0[6], p1[6];
#pragma unroll
    for (int k = 0; k < 6; k++) p0[k] = pose0[6 * (size_t)v + k], p1[k] = pose1[6 * (size_t)v + k];
    Mat3 R0, R1, Rr;
    rodrigues<false>(p0, R0.a, nullptr);
    rodrigues<false>(p1, R1.a, nullptr);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rr.a[i * 3 + j] = R1.a[i * 3] * R0.a[j * 3] + R1.a[i * 3 + 1] * R0.a[j * 3 + 1] + R1.a[i * 3 + 2] * R0.a[j * 3 + 2];
    double om[3];
    rodrigues_m2v_polar(Rr, om);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        out[(size_t)i * npad + v] = om[i];
        out[(size_t)(3 + i) * npad + v] = p1[3 + i] - (Rr.a[i * 3] * p0[3] + Rr.a[i * 3 + 1] * p0[4] + Rr.a[i * 3 + 2] * p0[5]);
    }
}

// ---- launchers -------------------------------------------------------------------------------
static size_t init_lds_floats(bool shared, int board_np, int PS, int VPB, bool with_src) {
    size_t nobj = shared ? (size_t)(3 * board_np + 1) / 2 * 2 : (size_t)VPB * 3 * PS + ((size_t)VPB * PS & 1);
    size_t nsrc = with_src ? (shared ? (size_t)board_np : (size_t)VPB * PS) : 0;
    return nobj + 2 * nsrc + 2 * (size_t)VPB * PS;
}

bool init_geometry(InitArgs& a, bool with_src) {
    const size_t cap = 64 * 1024;
    a.PS = a.maxnp | 1;
    a.cut = 6;
    if (const char* e = std::getenv("CCALIB_INIT_CUT")) a.cut = std::max(1, std::atoi(e));
    for (int vpb = 64 / kInitLanes; vpb >= 1; vpb--) {
        size_t bytes = sizeof(float) * init_lds_floats(a.shared_board != 0, a.board_np, a.PS, vpb, with_src);
        if (bytes <= cap) {
            a.VPB = vpb;
            a.lds_bytes = (unsigned)bytes;
            return true;
        }
    }
    return false;
}

// Lanes per view in the pose refinement.  Four: 16 views to a wavefront, so 100 000 views make 6250 wavefronts (six
// per SIMD, all resident) instead of 1563 that leave most SIMDs with one wavefront and nothing to hide latency with.
constexpr int kRefineLanes = 4;
// ... and in the second round, which takes the few views still running after a.cut iterations to the end.  What is
// left there is a chain of up to 14 dependent iterations per view, so more lanes per view shorten it (16 lanes: 0.27
// instead of 0.37 ms at 100 000 views) -- but the sums over points then run in another order than in the first round,
// the last bits of a resumed view depend on where it was parked, and a refinement that stops one iteration earlier or
// later moves a starting pose by 1e-7.  With the same lane count in both rounds a view's result does not depend on
// a.cut at all.  CCALIB_RESUME_LANES = 8, 16 or 64 is there to measure with.
constexpr int kResumeLanes = kRefineLanes;

template <bool RAT, int L>
static void launch_resume(const InitArgs& a, const CamParams& cp, double* pose, int* status, const Parked& parked, hipStream_t st) {
    InitArgs r = a;
    for (r.VPB = std::min(a.VPB, 64 / L);; r.VPB--) {
        r.lds_bytes = (unsigned)(sizeof(float) * init_lds_floats(a.shared_board != 0, a.board_np, a.PS, r.VPB, false));
        if (r.lds_bytes <= 64 * 1024 || r.VPB == 1) break;
    }
    // the worst case (every view parked) sizes the grid; workgroups past the parked count leave at once
    k_refine_pose<RAT, L, true><<<dim3((a.nviews + r.VPB - 1) / r.VPB), dim3(64), r.lds_bytes, st>>>(r, cp, pose, status, parked);
}
template <bool RAT>
static void launch_resume_any(const InitArgs& a, const CamParams& cp, double* pose, int* status, const Parked& parked,
                              hipStream_t st) {
    int L = kResumeLanes;
    if (const char* e = std::getenv("CCALIB_RESUME_LANES")) L = std::atoi(e);
    switch (L) {
        case 8: return launch_resume<RAT, 8>(a, cp, pose, status, parked, st);
        case 16: return launch_resume<RAT, 16>(a, cp, pose, status, parked, st);
        case 64: return launch_resume<RAT, 64>(a, cp, pose, status, parked, st);
        default: return launch_resume<RAT, kRefineLanes>(a, cp, pose, status, parked, st);
    }
}

void launch_init_pose(const InitArgs& a, const CamParams& cp, bool rational, double* pose, int* status, const Parked& parked,
                      hipStream_t st) {
    int grid = (a.nviews + a.VPB - 1) / a.VPB;
    (void)hipMemsetAsync(parked.count, 0, sizeof(int), st);
    // the refinement stages no homography source points and fewer views per workgroup
    InitArgs r = a;
    for (r.VPB = std::min(a.VPB, 64 / kRefineLanes);; r.VPB--) {  // a.VPB views with source points fit, so this ends at >= 1
        r.lds_bytes = (unsigned)(sizeof(float) * init_lds_floats(a.shared_board != 0, a.board_np, a.PS, r.VPB, false));
        if (r.lds_bytes <= 64 * 1024 || r.VPB == 1) break;
    }
    int rgrid = (a.nviews + r.VPB - 1) / r.VPB;
    if (rational) {
        k_init_pose<true><<<dim3(grid), dim3(64), a.lds_bytes, st>>>(a, cp, pose, status, parked);
        k_refine_pose<true, kRefineLanes, false><<<dim3(rgrid), dim3(64), r.lds_bytes, st>>>(r, cp, pose, status, parked);
        launch_resume_any<true>(a, cp, pose, status, parked, st);
    } else {
        k_init_pose<false><<<dim3(grid), dim3(64), a.lds_bytes, st>>>(a, cp, pose, status, parked);
        k_refine_pose<false, kRefineLanes, false><<<dim3(rgrid), dim3(64), r.lds_bytes, st>>>(r, cp, pose, status, parked);
        launch_resume_any<false>(a, cp, pose, status, parked, st);
    }
}

int launch_init_intrinsics(const InitArgs& a, double cx, double cy, double* part, hipStream_t st) {
    int grid = (a.nviews + a.VPB - 1) / a.VPB;
    k_init_intrinsics<<<dim3(grid), dim3(64), a.lds_bytes, st>>>(a, cx, cy, part);
    return grid;
}

void launch_relative_pose(const double* pose0, const double* pose1, int nviews, int npad, double* out, hipStream_t st) {
    k_relative_pose<<<dim3((nviews + 63) / 64), dim3(64), 0, st>>>(pose0, pose1, nviews, npad, out);
}

}  // namespace ccalib
