/*
 * orc_calib.c -- the calibration loops.  TEST INFRASTRUCTURE ONLY; parity unpinned.
 *
 * Restates, from knowledge of OpenCV 4.1.0 modules/calib3d/src/calibration.cpp:
 *   cvCalibrateCamera2Internal (iFixedPoint = -1) and cvStereoCalibrateImpl,
 * both driven by the CvLevMarq state machine in orc_levmarq.c.
 * Reference call sites: monocular_circle.cpp:93-94, monocular_chessboard.cpp:315-317,
 * binocular_chessboard.cpp:152-162, binocular_circle.cpp:153-163.
 *
 * Two solvers share one state machine:
 *   dense  -- the full (18+6N)^2 (mono) or (6(N+1)+36)^2 (stereo) normal matrix, assembled
 *             block by block exactly as OpenCV does, masked, mirrored, damped
 *             (diag *= 1 + lambda) and solved whole;
 *   schur  -- the same damped system solved by eliminating each view's 6x6 block.
 *             Algebraically identical; feasible for 1e5+ views.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "calib_oracle.h"
#include "orc_levmarq.h"
#include "orc_math.h"

#define NI ORC_NINTRINSIC

int orc_variants = 0;
void orc_set_variants(int bits) { orc_variants = bits; }

static double* g_trace = 0;
static int g_trace_cap = 0, g_trace_n = 0;
void orc_set_trace(double* buf, int cap) { g_trace = buf, g_trace_cap = buf ? cap : 0, g_trace_n = 0; }
int orc_trace_count(void) { return g_trace_n; }
static void trace_attach(orc_levmarq* lm) { lm->trace = g_trace, lm->trace_cap = g_trace_cap, lm->trace_n = 0; }

/* ---- per-view 6x6 elimination shared by the mono and stereo schur paths ---------------- */

typedef struct {
    int nviews;
    int ns;          /* size of the shared (non-view) parameter block: 18 mono, 6 + 36 stereo */
    const int* sidx; /* ns indices of the shared parameters inside lm->param               */
    int eofs0;       /* index of view 0's pose inside lm->param (views are contiguous)      */
    double* U;       /* ns x ns, upper triangle valid (same fill pattern as OpenCV)        */
    double* gs;      /* ns                                                                 */
    double* V;       /* nviews x 36                                                        */
    double* W;       /* nviews x ns x 6                                                    */
    double* ge;      /* nviews x 6                                                         */
    int nthreads;
} schur_sys;

static void schur_alloc(schur_sys* s, int nviews, int ns) {
    s->nviews = nviews;
    s->ns = ns;
    s->U = (double*)calloc((size_t)ns * ns, sizeof(double));
    s->gs = (double*)calloc(ns, sizeof(double));
    s->V = (double*)calloc((size_t)nviews * 36, sizeof(double));
    s->W = (double*)calloc((size_t)nviews * ns * 6, sizeof(double));
    s->ge = (double*)calloc((size_t)nviews * 6, sizeof(double));
}
static void schur_free(schur_sys* s) {
    free(s->U), free(s->gs), free(s->V), free(s->W), free(s->ge);
}
static void schur_zero(schur_sys* s) {
    memset(s->U, 0, sizeof(double) * s->ns * s->ns);
    memset(s->gs, 0, sizeof(double) * s->ns);
    memset(s->V, 0, sizeof(double) * (size_t)s->nviews * 36);
    memset(s->W, 0, sizeof(double) * (size_t)s->nviews * s->ns * 6);
    memset(s->ge, 0, sizeof(double) * (size_t)s->nviews * 6);
}

/* in-place lower Cholesky of a 6x6; returns 0 when not positive definite */
static int chol6(double* A) {
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k];
            if (i == j) {
                if (!(s > 0)) return 0;
                A[i * 6 + i] = sqrt(s);
            } else
                A[i * 6 + j] = s / A[j * 6 + j];
        }
    return 1;
}
static void chol6_solve(const double* L, double* b) {
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * 6 + k] * b[k];
        b[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * b[k];
        b[i] = s / L[i * 6 + i];
    }
}

/* custom CvLevMarq::step for the arrow system */
static void schur_step(orc_levmarq* lm, double lambda, void* user) {
    schur_sys* s = (schur_sys*)user;
    int ns = s->ns, N = s->nviews;
    int fidx[64], nf = 0;
    for (int a = 0; a < ns; a++)
        if (lm->mask[s->sidx[a]]) fidx[nf++] = a;
    int nth = s->nthreads > 0 ? s->nthreads : 1;
    double* Sacc = (double*)calloc((size_t)nth * (nf * nf + nf), sizeof(double));
    double* Lall = (double*)malloc(sizeof(double) * (size_t)N * 36);
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int i = 0; i < N; i++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* Sp = Sacc + (size_t)tid * (nf * nf + nf);
        double* bp = Sp + nf * nf;
        double* L = Lall + (size_t)i * 36;
        const double* Vi = s->V + (size_t)i * 36;
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) L[a * 6 + b] = a >= b ? Vi[b * 6 + a] : 0; /* upper -> lower */
        for (int a = 0; a < 6; a++) L[a * 6 + a] *= 1. + lambda;
        if (!chol6(L)) {
            for (int a = 0; a < 36; a++) L[a] = 0;
            for (int a = 0; a < 6; a++) L[a * 6 + a] = 1e150; /* singular view block: no update */
        }
        const double* Wi = s->W + (size_t)i * ns * 6;
        double y[6];
        memcpy(y, s->ge + (size_t)i * 6, sizeof(y));
        chol6_solve(L, y); /* V'^-1 ge */
        for (int a = 0; a < nf; a++) {
            const double* wa = Wi + fidx[a] * 6;
            double z[6];
            memcpy(z, wa, sizeof(z));
            chol6_solve(L, z); /* V'^-1 W_a^T */
            for (int b = 0; b < nf; b++) {
                const double* wb = Wi + fidx[b] * 6;
                double d = 0;
                for (int k = 0; k < 6; k++) d += wb[k] * z[k];
                Sp[a * nf + b] += d;
            }
            double d = 0;
            for (int k = 0; k < 6; k++) d += wa[k] * y[k];
            bp[a] += d;
        }
    }
    double* S = (double*)malloc(sizeof(double) * nf * nf);
    double* b = (double*)malloc(sizeof(double) * nf);
    double* dc = (double*)malloc(sizeof(double) * nf);
    for (int a = 0; a < nf; a++) {
        for (int c = 0; c < nf; c++) {
            int ia = fidx[a], ic = fidx[c];
            double u = ia <= ic ? s->U[ia * ns + ic] : s->U[ic * ns + ia]; /* completeSymm */
            if (a == c) u *= 1. + lambda;
            double acc = 0;
            for (int t = 0; t < nth; t++) acc += Sacc[(size_t)t * (nf * nf + nf) + a * nf + c];
            S[a * nf + c] = u - acc;
        }
        double acc = 0;
        for (int t = 0; t < nth; t++) acc += Sacc[(size_t)t * (nf * nf + nf) + nf * nf + a];
        b[a] = s->gs[fidx[a]] - acc;
    }
    if (nf > 0) orc_solve(S, b, nf, lm->solveMethod, dc);
    for (int a = 0; a < nf; a++) {
        int pi = s->sidx[fidx[a]];
        lm->param[pi] = lm->prevParam[pi] - dc[a];
    }
    for (int a = 0; a < ns; a++)
        if (!lm->mask[s->sidx[a]]) lm->param[s->sidx[a]] = lm->prevParam[s->sidx[a]];
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int i = 0; i < N; i++) {
        const double* Wi = s->W + (size_t)i * ns * 6;
        double rhs[6];
        memcpy(rhs, s->ge + (size_t)i * 6, sizeof(rhs));
        for (int a = 0; a < nf; a++) {
            const double* wa = Wi + fidx[a] * 6;
            for (int k = 0; k < 6; k++) rhs[k] -= wa[k] * dc[a];
        }
        chol6_solve(Lall + (size_t)i * 36, rhs);
        int e = s->eofs0 + i * 6;
        for (int k = 0; k < 6; k++) lm->param[e + k] = lm->prevParam[e + k] - rhs[k];
    }
    free(Sacc), free(Lall), free(S), free(b), free(dc);
}

/* ---- mono ------------------------------------------------------------------------------ */

/* A^T B for row-major A (rows x ca), B (rows x cb) -> out (ca x cb) */
static void atb(const double* A, int ca, const double* B, int cb, int rows, double* out) {
    for (int a = 0; a < ca; a++)
        for (int b = 0; b < cb; b++) {
            double s = 0;
            for (int i = 0; i < rows; i++) s += A[(size_t)i * ca + a] * B[(size_t)i * cb + b];
            out[a * cb + b] = s;
        }
}

static int mono_flags_and_mask(int flags, int ndist, unsigned char* mask18) {
    if (ndist < 8) {
        if (ndist < 5) flags |= ORC_FIX_K3;
        flags |= ORC_FIX_K4 | ORC_FIX_K5 | ORC_FIX_K6;
    }
    memset(mask18, 1, NI);
    if (flags & ORC_FIX_ASPECT_RATIO) mask18[0] = 0;
    if (flags & ORC_FIX_FOCAL_LENGTH) mask18[0] = mask18[1] = 0;
    if (flags & ORC_FIX_PRINCIPAL_POINT) mask18[2] = mask18[3] = 0;
    if (flags & ORC_ZERO_TANGENT_DIST) mask18[6] = mask18[7] = 0;
    if (!(flags & ORC_RATIONAL_MODEL)) flags |= ORC_FIX_K4 + ORC_FIX_K5 + ORC_FIX_K6;
    if (!(flags & ORC_THIN_PRISM_MODEL)) flags |= ORC_FIX_S1_S2_S3_S4;
    if (!(flags & ORC_TILTED_MODEL)) flags |= ORC_FIX_TAUX_TAUY;
    mask18[4] = !(flags & ORC_FIX_K1);
    mask18[5] = !(flags & ORC_FIX_K2);
    if (flags & ORC_FIX_TANGENT_DIST) mask18[6] = mask18[7] = 0;
    mask18[8] = !(flags & ORC_FIX_K3);
    mask18[9] = !(flags & ORC_FIX_K4);
    mask18[10] = !(flags & ORC_FIX_K5);
    mask18[11] = !(flags & ORC_FIX_K6);
    if (flags & ORC_FIX_S1_S2_S3_S4) mask18[12] = mask18[13] = mask18[14] = mask18[15] = 0;
    if (flags & ORC_FIX_TAUX_TAUY) mask18[16] = mask18[17] = 0;
    return flags;
}

double orc_calibrate_camera(int nviews, const int* npts, const double* obj_in, const double* img,
                            int width, int height, double A[9], double* dist, int ndist,
                            double* rvecs, double* tvecs, int flags, int term_type,
                            int max_iter, double eps, const double* init_rt, int schur,
                            int solve_override, int nthreads, orc_stats* stats) {
    if (nviews < 1) return -1;
    if (!(ndist == 4 || ndist == 5 || ndist == 8 || ndist == 12 || ndist == 14)) return -1;
    if (flags & ORC_TILTED_MODEL) return -1; /* tilt is not restated */
    if ((flags & ORC_THIN_PRISM_MODEL) && ndist != 12 && ndist != 14) return -1;
    int total = 0, maxPoints = 0;
    for (int i = 0; i < nviews; i++) {
        if (npts[i] < 4) return -1;
        total += npts[i];
        if (npts[i] > maxPoints) maxPoints = npts[i];
    }
    double* obj = (double*)malloc(sizeof(double) * 3 * total);
    memcpy(obj, obj_in, sizeof(double) * 3 * total);
    int nparams = NI + nviews * 6;
    double k[14] = {0};
    double aspect = 0.;

    if (flags & ORC_USE_INTRINSIC_GUESS) {
        if (A[0] <= 0 || A[4] <= 0) return -1;
        if (A[2] < 0 || A[2] >= width || A[5] < 0 || A[5] >= height) return -1;
        if (fabs(A[1]) > 1e-5 || fabs(A[3]) > 1e-5 || fabs(A[6]) > 1e-5 || fabs(A[7]) > 1e-5 ||
            fabs(A[8] - 1) > 1e-5)
            return -1;
        A[1] = A[3] = A[6] = A[7] = 0.;
        A[8] = 1.;
        if (flags & ORC_FIX_ASPECT_RATIO) {
            aspect = A[0] / A[4];
            if (aspect < 0.01 || aspect > 100.) return -1;
        }
        for (int i = 0; i < ndist; i++) k[i] = dist[i];
    } else {
        double mean = 0, sq = 0;
        for (int i = 0; i < total; i++) mean += obj[i * 3 + 2];
        mean /= total;
        for (int i = 0; i < total; i++) sq += (obj[i * 3 + 2] - mean) * (obj[i * 3 + 2] - mean);
        if (fabs(mean) > 1e-5 || fabs(sqrt(sq / total)) > 1e-5) return -1; /* non-planar rig needs a guess */
        for (int i = 0; i < total; i++) obj[i * 3 + 2] = 0.;
        if (flags & ORC_FIX_ASPECT_RATIO) {
            aspect = A[0] / A[4];
            if (aspect < 0.01 || aspect > 100.) return -1;
        }
        orc_init_intrinsic_params_2d(nviews, npts, obj, img, width, height, aspect, A);
    }

    orc_levmarq lm;
    orc_lm_init(&lm, nparams, 0, term_type, max_iter, eps, !schur);
    trace_attach(&lm);
    if (flags & ORC_USE_LU) lm.solveMethod = ORC_SOLVE_LU;
    if (solve_override >= 0) lm.solveMethod = solve_override;
    lm.param[0] = A[0], lm.param[1] = A[4], lm.param[2] = A[2], lm.param[3] = A[5];
    memcpy(lm.param + 4, k, sizeof(k));
    if (flags & ORC_ZERO_TANGENT_DIST) lm.param[6] = lm.param[7] = 0;
    flags = mono_flags_and_mask(flags, ndist, lm.mask);

    /* 2. extrinsics */
    {
        double A0[9], k0[14];
        memcpy(A0, A, sizeof(A0));
        memcpy(k0, k, sizeof(k0));
        int* ofs = (int*)malloc(sizeof(int) * nviews);
        for (int i = 0, pos = 0; i < nviews; pos += npts[i], i++) ofs[i] = pos;
        if (init_rt)
            memcpy(lm.param + NI, init_rt, sizeof(double) * 6 * nviews);
        else {
            int nth = nthreads > 0 ? nthreads : 1;
#pragma omp parallel for num_threads(nth) schedule(dynamic, 16)
            for (int i = 0; i < nviews; i++)
                orc_find_extrinsic(npts[i], obj + (size_t)ofs[i] * 3, img + (size_t)ofs[i] * 2, A0, k0,
                                   lm.param + NI + i * 6, lm.param + NI + i * 6 + 3);
        }
        free(ofs);
    }

    schur_sys ss;
    int sidx[NI];
    memset(&ss, 0, sizeof(ss));
    if (schur) {
        schur_alloc(&ss, nviews, NI);
        for (int a = 0; a < NI; a++) sidx[a] = a;
        ss.sidx = sidx;
        ss.eofs0 = NI;
        ss.nthreads = nthreads;
        lm.custom_step = schur_step;
        lm.user = &ss;
    }

    int nth = (schur && nthreads > 0) ? nthreads : 1;
    int* ofs = (int*)malloc(sizeof(int) * nviews);
    for (int i = 0, pos = 0; i < nviews; pos += npts[i], i++) ofs[i] = pos;
    /* per-thread scratch */
    double* scratch = (double*)malloc(sizeof(double) * (size_t)nth * maxPoints * 2 * (1 + 6 + NI));
    double* Uacc = (double*)calloc((size_t)nth * (NI * NI + NI + 1), sizeof(double));
    double reprojErr = 0;
    int jac_passes = 0, err_passes = 0;

    for (;;) {
        int wantJ, wantErr;
        int proceed = orc_lm_update_alt(&lm, &wantJ, &wantErr);
        double* param = lm.param;
        double* pparam = lm.prevParam;
        int calcJ = lm.state == ORC_LM_CALC_J;
        if (flags & ORC_FIX_ASPECT_RATIO) {
            param[0] = param[1] * aspect;
            pparam[0] = pparam[1] * aspect;
        }
        A[0] = param[0], A[4] = param[1], A[2] = param[2], A[5] = param[3];
        memcpy(k, param + 4, sizeof(k));
        if (!proceed) break;
        if (calcJ)
            jac_passes++;
        else
            err_passes++;
        if (calcJ && schur) schur_zero(&ss);
        memset(Uacc, 0, sizeof(double) * (size_t)nth * (NI * NI + NI + 1));
        double fc[4] = {param[0], param[1], param[2], param[3]};
        double asp = (flags & ORC_FIX_ASPECT_RATIO) ? aspect : 0;
#pragma omp parallel for num_threads(nth) schedule(static)
        for (int i = 0; i < nviews; i++) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            int ni = npts[i], pos = ofs[i];
            double* err = scratch + (size_t)tid * maxPoints * 2 * (1 + 6 + NI);
            double* Je = err + maxPoints * 2;
            double* Ji = Je + maxPoints * 2 * 6;
            double* acc = Uacc + (size_t)tid * (NI * NI + NI + 1);
            const double* ri = param + NI + i * 6;
            orc_project_points(ni, obj + (size_t)pos * 3, ri, ri + 3, fc, k, asp, err, calcJ ? Je : 0,
                               calcJ ? Ji : 0);
            for (int j = 0; j < 2 * ni; j++) err[j] -= img[(size_t)pos * 2 + j];
            if (calcJ) {
                /* FIX_FOCAL_LENGTH / FIX_PRINCIPAL_POINT leave dpdf / dpdc unwritten in OpenCV; those
                 * columns are masked out of the solve, so their content never matters. */
                double JiJi[NI * NI], JeJe[36], JiJe[NI * 6], Jie[NI], Jee[6];
                atb(Ji, NI, Ji, NI, 2 * ni, JiJi);
                atb(Je, 6, Je, 6, 2 * ni, JeJe);
                atb(Ji, NI, Je, 6, 2 * ni, JiJe);
                atb(Ji, NI, err, 1, 2 * ni, Jie);
                atb(Je, 6, err, 1, 2 * ni, Jee);
                if (!schur) {
                    int n = lm.nparams, e = NI + i * 6;
                    for (int a = 0; a < NI; a++)
                        for (int b = 0; b < NI; b++) lm.JtJ[(size_t)a * n + b] += JiJi[a * NI + b];
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++) lm.JtJ[(size_t)(e + a) * n + e + b] = JeJe[a * 6 + b];
                    for (int a = 0; a < NI; a++)
                        for (int b = 0; b < 6; b++) lm.JtJ[(size_t)a * n + e + b] = JiJe[a * 6 + b];
                    for (int a = 0; a < NI; a++) lm.JtErr[a] += Jie[a];
                    for (int a = 0; a < 6; a++) lm.JtErr[e + a] = Jee[a];
                } else {
                    for (int a = 0; a < NI * NI; a++) acc[a] += JiJi[a];
                    for (int a = 0; a < NI; a++) acc[NI * NI + a] += Jie[a];
                    memcpy(ss.V + (size_t)i * 36, JeJe, sizeof(JeJe));
                    memcpy(ss.W + (size_t)i * NI * 6, JiJe, sizeof(JiJe));
                    memcpy(ss.ge + (size_t)i * 6, Jee, sizeof(Jee));
                }
            }
            double viewErr = 0;
            for (int j = 0; j < 2 * ni; j++) viewErr += err[j] * err[j];
            acc[NI * NI + NI] += viewErr;
        }
        reprojErr = 0;
        for (int t = 0; t < nth; t++) {
            const double* acc = Uacc + (size_t)t * (NI * NI + NI + 1);
            reprojErr += acc[NI * NI + NI];
            if (calcJ && schur) {
                for (int a = 0; a < NI * NI; a++) ss.U[a] += acc[a];
                for (int a = 0; a < NI; a++) ss.gs[a] += acc[NI * NI + a];
            }
        }
        if (wantErr) lm.errNorm = reprojErr;
    }

    for (int i = 0; i < ndist; i++) dist[i] = k[i];
    for (int i = 0; i < nviews; i++) {
        memcpy(rvecs + i * 3, lm.param + NI + i * 6, 3 * sizeof(double));
        memcpy(tvecs + i * 3, lm.param + NI + i * 6 + 3, 3 * sizeof(double));
    }
    if (stats) {
        stats->iters = lm.iters;
        stats->jac_passes = jac_passes;
        stats->err_passes = err_passes;
        stats->lambda_lg10 = lm.lambdaLg10;
        stats->final_err = reprojErr;
    }
    g_trace_n = lm.trace_n;
    if (schur) schur_free(&ss);
    free(scratch), free(Uacc), free(ofs), free(obj);
    orc_lm_free(&lm);
    return sqrt(reprojErr / total);
}

double orc_reprojection_error(int nviews, const int* npts, const double* obj, const double* img,
                              const double A[9], const double* dist, int ndist,
                              const double* rvecs, const double* tvecs, double* per_view_rms) {
    double k[14] = {0}, fc[4] = {A[0], A[4], A[2], A[5]};
    for (int i = 0; i < ndist && i < 14; i++) k[i] = dist[i];
    double totalErr = 0;
    int totalPoints = 0, pos = 0;
    for (int i = 0; i < nviews; i++) {
        int n = npts[i];
        double* m = (double*)malloc(sizeof(double) * 2 * n);
        orc_project_points(n, obj + (size_t)pos * 3, rvecs + i * 3, tvecs + i * 3, fc, k, 0., m, 0, 0);
        double e = 0;
        for (int j = 0; j < 2 * n; j++) {
            /* projectPoints writes Point2f; the reference compares in single precision */
            double d = img[(size_t)pos * 2 + j] - m[j];
            e += d * d;
        }
        if (per_view_rms) per_view_rms[i] = sqrt(e / n);
        totalErr += e;
        totalPoints += n;
        pos += n;
        free(m);
    }
    return sqrt(totalErr / totalPoints);
}

double orc_mono_blocks(int nviews, const int* npts, const double* obj, const double* img,
                       const double intr[18], double aspect, const double* rt, int nfree,
                       const int* free_idx, double* U, double* gc, double* V, double* W,
                       double* ge) {
    double fc[4] = {intr[0], intr[1], intr[2], intr[3]};
    if (aspect != 0) fc[0] = fc[1] * aspect;
    double total = 0;
    for (int a = 0; a < nfree * nfree; a++) U[a] = 0;
    for (int a = 0; a < nfree; a++) gc[a] = 0;
    int pos = 0;
    for (int i = 0; i < nviews; i++) {
        int ni = npts[i];
        double* err = (double*)malloc(sizeof(double) * 2 * ni * (1 + 6 + NI));
        double* Je = err + 2 * ni;
        double* Ji = Je + 2 * ni * 6;
        orc_project_points(ni, obj + (size_t)pos * 3, rt + i * 6, rt + i * 6 + 3, fc, intr + 4, aspect, err, Je, Ji);
        for (int j = 0; j < 2 * ni; j++) err[j] -= img[(size_t)pos * 2 + j];
        for (int a = 0; a < nfree; a++) {
            for (int b = 0; b < nfree; b++) {
                double s = 0;
                for (int r = 0; r < 2 * ni; r++) s += Ji[r * NI + free_idx[a]] * Ji[r * NI + free_idx[b]];
                U[a * nfree + b] += s;
            }
            double s = 0;
            for (int r = 0; r < 2 * ni; r++) s += Ji[r * NI + free_idx[a]] * err[r];
            gc[a] += s;
            for (int b = 0; b < 6; b++) {
                s = 0;
                for (int r = 0; r < 2 * ni; r++) s += Ji[r * NI + free_idx[a]] * Je[r * 6 + b];
                W[((size_t)i * nfree + a) * 6 + b] = s;
            }
        }
        atb(Je, 6, Je, 6, 2 * ni, V + (size_t)i * 36);
        atb(Je, 6, err, 1, 2 * ni, ge + (size_t)i * 6);
        for (int j = 0; j < 2 * ni; j++) total += err[j] * err[j];
        pos += ni;
        free(err);
    }
    return total;
}

/* ---- stereo ---------------------------------------------------------------------------- */

/* orc_stereo_blocks: when armed, the Schur-path loop below hands out the system of its first Jacobian pass and stops */
static struct {
    int armed;
    double *U, *gs, *V, *W, *ge, *err;
} g_capture;

static int dbl_cmp(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

double orc_stereo_calibrate(int nviews, const int* npts, const double* obj, const double* img1,
                            const double* img2, double A1[9], double dist1[14], double A2[9],
                            double dist2[14], int width, int height, double Rout[9], double Tout[3],
                            double Eout[9], double Fout[9], int flags, int term_type, int max_iter,
                            double eps, const double* init_rt, const double* init_lr,
                            double* rt_out, int schur, int nthreads, orc_stats* stats) {
    if (flags & ORC_TILTED_MODEL) return -1;
    int pointsTotal = 0, maxPoints = 0;
    for (int i = 0; i < nviews; i++) {
        pointsTotal += npts[i];
        if (npts[i] > maxPoints) maxPoints = npts[i];
    }
    const double* imgs[2] = {img1, img2};
    double Ak[2][9], dk[2][14];
    memset(Ak, 0, sizeof(Ak));
    memset(dk, 0, sizeof(dk));
    double* Ain[2] = {A1, A2};
    double* din[2] = {dist1, dist2};
    for (int k = 0; k < 2; k++) {
        if (flags & (ORC_FIX_INTRINSIC | ORC_USE_INTRINSIC_GUESS | ORC_FIX_ASPECT_RATIO | ORC_FIX_FOCAL_LENGTH))
            memcpy(Ak[k], Ain[k], sizeof(Ak[k]));
        if (flags & (ORC_FIX_INTRINSIC | ORC_USE_INTRINSIC_GUESS | ORC_FIX_K1 | ORC_FIX_K2 | ORC_FIX_K3 |
                     ORC_FIX_K4 | ORC_FIX_K5 | ORC_FIX_K6 | ORC_FIX_TANGENT_DIST))
            memcpy(dk[k], din[k], sizeof(dk[k]));
        if (!(flags & (ORC_FIX_INTRINSIC | ORC_USE_INTRINSIC_GUESS))) {
            double* rv = (double*)malloc(sizeof(double) * 3 * nviews);
            double* tv = (double*)malloc(sizeof(double) * 3 * nviews);
            double rr = orc_calibrate_camera(nviews, npts, obj, imgs[k], width, height, Ak[k], dk[k], 14, rv, tv,
                                             flags, ORC_TERM_COUNT | ORC_TERM_EPS, 30, DBL_EPSILON, 0, 0, -1,
                                             1, 0);
            free(rv), free(tv);
            if (rr < 0) return -1;
        }
    }
    if (flags & ORC_SAME_FOCAL_LENGTH) {
        static const int avg_idx[] = {0, 4, 2, 5, -1};
        for (int k = 0; avg_idx[k] >= 0; k++)
            Ak[0][avg_idx[k]] = Ak[1][avg_idx[k]] = (Ak[0][avg_idx[k]] + Ak[1][avg_idx[k]]) * 0.5;
    }
    double aspectRatio[2] = {0, 0};
    if (flags & ORC_FIX_ASPECT_RATIO)
        for (int k = 0; k < 2; k++) aspectRatio[k] = Ak[k][0] / Ak[k][4];
    int recomputeIntrinsics = (flags & ORC_FIX_INTRINSIC) == 0;
    int nparams = 6 * (nviews + 1) + (recomputeIntrinsics ? NI * 2 : 0);

    orc_levmarq lm;
    orc_lm_init(&lm, nparams, 0, term_type, max_iter, eps, !schur);
    trace_attach(&lm);
    if (flags & ORC_USE_LU) lm.solveMethod = ORC_SOLVE_LU;

    if (recomputeIntrinsics) {
        unsigned char* imask = lm.mask + nparams - NI * 2;
        if (!(flags & ORC_RATIONAL_MODEL)) flags |= ORC_FIX_K4 | ORC_FIX_K5 | ORC_FIX_K6;
        if (!(flags & ORC_THIN_PRISM_MODEL)) flags |= ORC_FIX_S1_S2_S3_S4;
        if (!(flags & ORC_TILTED_MODEL)) flags |= ORC_FIX_TAUX_TAUY;
        if (flags & ORC_FIX_ASPECT_RATIO) imask[0] = imask[NI] = 0;
        if (flags & ORC_FIX_FOCAL_LENGTH) imask[0] = imask[1] = imask[NI] = imask[NI + 1] = 0;
        if (flags & ORC_FIX_PRINCIPAL_POINT) imask[2] = imask[3] = imask[NI + 2] = imask[NI + 3] = 0;
        if (flags & (ORC_ZERO_TANGENT_DIST | ORC_FIX_TANGENT_DIST))
            imask[6] = imask[7] = imask[NI + 6] = imask[NI + 7] = 0;
        if (flags & ORC_FIX_K1) imask[4] = imask[NI + 4] = 0;
        if (flags & ORC_FIX_K2) imask[5] = imask[NI + 5] = 0;
        if (flags & ORC_FIX_K3) imask[8] = imask[NI + 8] = 0;
        if (flags & ORC_FIX_K4) imask[9] = imask[NI + 9] = 0;
        if (flags & ORC_FIX_K5) imask[10] = imask[NI + 10] = 0;
        if (flags & ORC_FIX_K6) imask[11] = imask[NI + 11] = 0;
        if (flags & ORC_FIX_S1_S2_S3_S4)
            for (int j = 12; j < 16; j++) imask[j] = imask[NI + j] = 0;
        if (flags & ORC_FIX_TAUX_TAUY)
            for (int j = 16; j < 18; j++) imask[j] = imask[NI + j] = 0;
        if ((flags & ORC_SAME_FOCAL_LENGTH) && (orc_variants & ORC_VARIANT_SAME_FOCAL_MASK))
            imask[NI] = imask[NI + 1] = 0; /* what-if: camera 1's focal lengths are not unknowns */
    }

    /* initial poses: per-view per-camera PnP, relative pose by per-component median */
    if (init_rt && init_lr) {
        memcpy(lm.param, init_lr, 6 * sizeof(double));
        memcpy(lm.param + 6, init_rt, sizeof(double) * 6 * nviews);
    } else {
        double* RT0 = (double*)malloc(sizeof(double) * 6 * nviews);
        int* ofs0 = (int*)malloc(sizeof(int) * nviews);
        for (int i = 0, pos = 0; i < nviews; pos += npts[i], i++) ofs0[i] = pos;
        int nth0 = (schur && nthreads > 0) ? nthreads : 1;
#pragma omp parallel for num_threads(nth0) schedule(dynamic, 16)
        for (int i = 0; i < nviews; i++) {
            int ni = npts[i], ofs = ofs0[i];
            double om[2][3], r[2][9], t[2][3];
            for (int k = 0; k < 2; k++) {
                orc_find_extrinsic(ni, obj + (size_t)ofs * 3, imgs[k] + (size_t)ofs * 2, Ak[k], dk[k], om[k], t[k]);
                orc_rodrigues_v2m(om[k], r[k], 0);
                if (k == 0) {
                    memcpy(lm.param + (i + 1) * 6, om[0], 3 * sizeof(double));
                    memcpy(lm.param + (i + 1) * 6 + 3, t[0], 3 * sizeof(double));
                }
            }
            double Rref[9], R0t[9];
            orc_transpose(r[0], 3, 3, R0t);
            orc_matmul(r[1], 3, 3, R0t, 3, Rref); /* R[0] = R[1] * R[0]^T */
            double Tref[3];
            for (int a = 0; a < 3; a++)
                Tref[a] = t[1][a] - (Rref[a * 3] * t[0][0] + Rref[a * 3 + 1] * t[0][1] + Rref[a * 3 + 2] * t[0][2]);
            double omref[3];
            orc_rodrigues_m2v(Rref, omref, 0);
            for (int a = 0; a < 3; a++) {
                RT0[i + nviews * a] = omref[a];
                RT0[i + nviews * (3 + a)] = Tref[a];
            }
        }
        free(ofs0);
        for (int i = 0; i < 6; i++) {
            qsort(RT0 + i * nviews, nviews, sizeof(double), dbl_cmp);
            lm.param[i] = nviews % 2 != 0 ? RT0[i * nviews + nviews / 2]
                                          : (RT0[i * nviews + nviews / 2 - 1] + RT0[i * nviews + nviews / 2]) * 0.5;
        }
        free(RT0);
    }

    if (recomputeIntrinsics)
        for (int k = 0; k < 2; k++) {
            double* iparam = lm.param + (nviews + 1) * 6 + k * NI;
            if (flags & ORC_ZERO_TANGENT_DIST) dk[k][2] = dk[k][3] = 0;
            iparam[0] = Ak[k][0], iparam[1] = Ak[k][4], iparam[2] = Ak[k][2], iparam[3] = Ak[k][5];
            memcpy(iparam + 4, dk[k], 14 * sizeof(double));
        }

    /* schur bookkeeping: shared block = [om_LR T_LR | intr0 | intr1] */
    schur_sys ss;
    int sidx[6 + 2 * NI];
    int ns = 6 + (recomputeIntrinsics ? 2 * NI : 0);
    memset(&ss, 0, sizeof(ss));
    for (int a = 0; a < 6; a++) sidx[a] = a;
    for (int a = 0; a < 2 * NI; a++) sidx[6 + a] = (nviews + 1) * 6 + a;
    if (schur) {
        schur_alloc(&ss, nviews, ns);
        ss.sidx = sidx;
        ss.eofs0 = 6;
        ss.nthreads = nthreads;
        lm.custom_step = schur_step;
        lm.user = &ss;
    }

    /* the schur path walks the views in parallel: per-thread row buffers and shared-block accumulators
     * (the dense, OpenCV-shaped path writes one big matrix and stays on one thread) */
    int nth = (schur && nthreads > 0) ? nthreads : 1;
    const size_t rowbuf = (size_t)maxPoints * 2 * (1 + 6 + 6 + NI);
    double* scratch = (double*)calloc((size_t)nth * rowbuf, sizeof(double));
    const size_t nacc = (size_t)ns * ns + ns + 1;
    double* Uacc = (double*)calloc((size_t)nth * nacc, sizeof(double));
    int* vofs = (int*)malloc(sizeof(int) * nviews);
    for (int i = 0, pos = 0; i < nviews; pos += npts[i], i++) vofs[i] = pos;
    double reprojErr = 0;
    int jac_passes = 0, err_passes = 0;

    for (;;) {
        int wantJ, wantErr;
        if (!orc_lm_update_alt(&lm, &wantJ, &wantErr)) break;
        int calcJ = lm.state == ORC_LM_CALC_J;
        if (calcJ)
            jac_passes++;
        else
            err_passes++;
        reprojErr = 0;
        double* om_LR = lm.param;
        double* T_LR = lm.param + 3;
        if (recomputeIntrinsics) {
            double* iparam = lm.param + (nviews + 1) * 6;
            double* ipparam = lm.prevParam + (nviews + 1) * 6;
            if (flags & ORC_SAME_FOCAL_LENGTH) {
                iparam[NI] = iparam[0], iparam[NI + 1] = iparam[1];
                ipparam[NI] = ipparam[0], ipparam[NI + 1] = ipparam[1];
            }
            if (flags & ORC_FIX_ASPECT_RATIO) {
                iparam[0] = iparam[1] * aspectRatio[0];
                iparam[NI] = iparam[NI + 1] * aspectRatio[1];
                ipparam[0] = ipparam[1] * aspectRatio[0];
                ipparam[NI] = ipparam[NI + 1] * aspectRatio[1];
            }
            for (int k = 0; k < 2; k++) {
                Ak[k][0] = iparam[k * NI + 0], Ak[k][4] = iparam[k * NI + 1];
                Ak[k][2] = iparam[k * NI + 2], Ak[k][5] = iparam[k * NI + 3];
                memcpy(dk[k], iparam + k * NI + 4, 14 * sizeof(double));
            }
        }
        if (calcJ && schur) schur_zero(&ss);
        memset(Uacc, 0, sizeof(double) * (size_t)nth * nacc);
        int n = lm.nparams;
#pragma omp parallel for num_threads(nth) schedule(static)
        for (int i = 0; i < nviews; i++) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            int ni = npts[i], ofs = vofs[i];
            double* err = scratch + (size_t)tid * rowbuf;
            double* Je = err + maxPoints * 2;
            double* J_LR = Je + maxPoints * 2 * 6;
            double* Ji = J_LR + maxPoints * 2 * 6;
            double* Uloc = Uacc + (size_t)tid * nacc;
            double* gsloc = Uloc + (size_t)ns * ns;
            double* om0 = lm.param + (i + 1) * 6;
            double* T0 = om0 + 3;
            double omR[3], tR[3], dr3dr1[9], dr3dr2[9], dt3dr2[9], dt3dt1[9], dt3dt2[9];
            if (calcJ)
                orc_compose_rt(om0, T0, om_LR, T_LR, omR, tR, dr3dr1, dr3dr2, dt3dr2, dt3dt1, dt3dt2);
            else
                orc_compose_rt(om0, T0, om_LR, T_LR, omR, tR, 0, 0, 0, 0, 0);
            for (int k = 0; k < 2; k++) {
                const double* omk = k == 0 ? om0 : omR;
                const double* Tk = k == 0 ? T0 : tR;
                double fc[4] = {Ak[k][0], Ak[k][4], Ak[k][2], Ak[k][5]};
                orc_project_points(ni, obj + (size_t)ofs * 3, omk, Tk, fc, dk[k],
                                   (flags & ORC_FIX_ASPECT_RATIO) ? aspectRatio[k] : 0, err, calcJ ? Je : 0,
                                   calcJ ? Ji : 0);
                for (int j = 0; j < 2 * ni; j++) err[j] -= imgs[k][(size_t)ofs * 2 + j];
                if (calcJ) {
                    int iofs = (nviews + 1) * 6 + k * NI, eofs = (i + 1) * 6;
                    double JLtJL[36], JLtJe[36], JLte[6], JetJe[36], Jete[6];
                    if (k == 1) {
                        for (int p = 0; p < ni * 2; p++) {
                            double* de3dr3 = Je + p * 6;
                            double* de3dt3 = de3dr3 + 3;
                            double* de3dr2 = J_LR + p * 6;
                            double* de3dt2 = de3dr2 + 3;
                            double de3dr1[3], de3dt1[3];
                            for (int c = 0; c < 3; c++) {
                                de3dr1[c] = de3dr3[0] * dr3dr1[c] + de3dr3[1] * dr3dr1[3 + c] + de3dr3[2] * dr3dr1[6 + c];
                                de3dt1[c] = de3dt3[0] * dt3dt1[c] + de3dt3[1] * dt3dt1[3 + c] + de3dt3[2] * dt3dt1[6 + c];
                                de3dr2[c] = de3dr3[0] * dr3dr2[c] + de3dr3[1] * dr3dr2[3 + c] + de3dr3[2] * dr3dr2[6 + c];
                                de3dr2[c] += de3dt3[0] * dt3dr2[c] + de3dt3[1] * dt3dr2[3 + c] + de3dt3[2] * dt3dr2[6 + c];
                                de3dt2[c] = de3dt3[0] * dt3dt2[c] + de3dt3[1] * dt3dt2[3 + c] + de3dt3[2] * dt3dt2[6 + c];
                            }
                            memcpy(de3dr3, de3dr1, sizeof(de3dr1));
                            memcpy(de3dt3, de3dt1, sizeof(de3dt1));
                        }
                        atb(J_LR, 6, J_LR, 6, 2 * ni, JLtJL);
                        atb(J_LR, 6, Je, 6, 2 * ni, JLtJe);
                        atb(J_LR, 6, err, 1, 2 * ni, JLte);
                    }
                    atb(Je, 6, Je, 6, 2 * ni, JetJe);
                    atb(Je, 6, err, 1, 2 * ni, Jete);
                    double JitJi[NI * NI], JetJi[6 * NI], JLtJi[6 * NI], Jite[NI];
                    if (recomputeIntrinsics) {
                        atb(Ji, NI, Ji, NI, 2 * ni, JitJi);
                        atb(Je, 6, Ji, NI, 2 * ni, JetJi);
                        if (k == 1) atb(J_LR, 6, Ji, NI, 2 * ni, JLtJi);
                        atb(Ji, NI, err, 1, 2 * ni, Jite);
                    }
                    if (!schur) {
                        double* JtJ = lm.JtJ;
                        double* JtErr = lm.JtErr;
                        if (k == 1) {
                            for (int a = 0; a < 6; a++)
                                for (int b = 0; b < 6; b++) {
                                    JtJ[(size_t)a * n + b] += JLtJL[a * 6 + b];
                                    JtJ[(size_t)a * n + eofs + b] = JLtJe[a * 6 + b];
                                }
                            for (int a = 0; a < 6; a++) JtErr[a] += JLte[a];
                        }
                        for (int a = 0; a < 6; a++) {
                            for (int b = 0; b < 6; b++) JtJ[(size_t)(eofs + a) * n + eofs + b] += JetJe[a * 6 + b];
                            JtErr[eofs + a] += Jete[a];
                        }
                        if (recomputeIntrinsics) {
                            for (int a = 0; a < NI; a++) {
                                for (int b = 0; b < NI; b++) JtJ[(size_t)(iofs + a) * n + iofs + b] += JitJi[a * NI + b];
                                JtErr[iofs + a] += Jite[a];
                            }
                            for (int a = 0; a < 6; a++)
                                for (int b = 0; b < NI; b++) {
                                    JtJ[(size_t)(eofs + a) * n + iofs + b] += JetJi[a * NI + b];
                                    if (k == 1) JtJ[(size_t)a * n + iofs + b] += JLtJi[a * NI + b];
                                }
                        }
                    } else {
                        /* shared slots: 0..5 = LR, 6 + k*NI.. = camera k intrinsics */
                        double* Vi = ss.V + (size_t)i * 36;
                        double* Wi = ss.W + (size_t)i * ns * 6;
                        double* gei = ss.ge + (size_t)i * 6;
                        int so = 6 + k * NI;
                        if (k == 1) {
                            for (int a = 0; a < 6; a++) {
                                for (int b = 0; b < 6; b++) {
                                    Uloc[a * ns + b] += JLtJL[a * 6 + b];
                                    Wi[a * 6 + b] = JLtJe[a * 6 + b];
                                }
                                gsloc[a] += JLte[a];
                            }
                        }
                        for (int a = 0; a < 6; a++) {
                            for (int b = 0; b < 6; b++) Vi[a * 6 + b] += JetJe[a * 6 + b];
                            gei[a] += Jete[a];
                        }
                        if (recomputeIntrinsics) {
                            for (int a = 0; a < NI; a++) {
                                for (int b = 0; b < NI; b++) Uloc[(so + a) * ns + so + b] += JitJi[a * NI + b];
                                gsloc[so + a] += Jite[a];
                                for (int b = 0; b < 6; b++) Wi[(so + a) * 6 + b] += JetJi[b * NI + a];
                                if (k == 1)
                                    for (int b = 0; b < 6; b++) Uloc[b * ns + so + a] += JLtJi[b * NI + a];
                            }
                        }
                    }
                }
                double viewErr = 0;
                for (int j = 0; j < 2 * ni; j++) viewErr += err[j] * err[j];
                gsloc[ns] += viewErr;
            }
        }
        for (int t = 0; t < nth; t++) {
            const double* acc = Uacc + (size_t)t * nacc;
            reprojErr += acc[nacc - 1];
            if (calcJ && schur) {
                for (int a = 0; a < ns * ns; a++) ss.U[a] += acc[a];
                for (int a = 0; a < ns; a++) ss.gs[a] += acc[(size_t)ns * ns + a];
            }
        }
        if (wantErr) lm.errNorm = reprojErr;
        if (calcJ && schur && g_capture.armed) {
            g_capture.armed = 0;
            memcpy(g_capture.U, ss.U, sizeof(double) * ns * ns);
            memcpy(g_capture.gs, ss.gs, sizeof(double) * ns);
            memcpy(g_capture.V, ss.V, sizeof(double) * (size_t)nviews * 36);
            memcpy(g_capture.W, ss.W, sizeof(double) * (size_t)nviews * ns * 6);
            memcpy(g_capture.ge, ss.ge, sizeof(double) * (size_t)nviews * 6);
            *g_capture.err = reprojErr;
            break;
        }
    }

    double R_LR[9];
    orc_rodrigues_v2m(lm.param, R_LR, 0);
    memcpy(Rout, R_LR, sizeof(R_LR));
    memcpy(Tout, lm.param + 3, 3 * sizeof(double));
    if (recomputeIntrinsics) {
        memcpy(A1, Ak[0], sizeof(Ak[0]));
        memcpy(A2, Ak[1], sizeof(Ak[1]));
        memcpy(dist1, dk[0], sizeof(dk[0]));
        memcpy(dist2, dk[1], sizeof(dk[1]));
    }
    {
        const double* t = lm.param + 3;
        double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        double E[9], Fm[9], iK[9], tmp[9], iKt[9];
        orc_matmul(tx, 3, 3, R_LR, 3, E);
        if (Eout) memcpy(Eout, E, sizeof(E));
        if (Fout) {
            orc_inv3(Ak[1], iK);
            orc_transpose(iK, 3, 3, iKt);
            orc_matmul(iKt, 3, 3, E, 3, tmp);
            orc_inv3(Ak[0], iK);
            orc_matmul(tmp, 3, 3, iK, 3, Fm);
            double sc = fabs(Fm[8]) > 0 ? 1. / Fm[8] : 1;
            for (int i = 0; i < 9; i++) Fout[i] = Fm[i] * sc;
        }
    }
    if (rt_out) memcpy(rt_out, lm.param + 6, sizeof(double) * 6 * nviews);
    if (stats) {
        stats->iters = lm.iters;
        stats->jac_passes = jac_passes;
        stats->err_passes = err_passes;
        stats->lambda_lg10 = lm.lambdaLg10;
        stats->final_err = reprojErr;
    }
    g_trace_n = lm.trace_n;
    if (schur) schur_free(&ss);
    free(scratch), free(Uacc), free(vofs);
    orc_lm_free(&lm);
    return sqrt(reprojErr / (pointsTotal * 2));
}

double orc_stereo_blocks(int nviews, const int* npts, const double* obj, const double* img1, const double* img2,
                         const double A1[9], const double dist1[14], const double A2[9], const double dist2[14],
                         int flags, const double* rt, const double* lr, double* U, double* gs, double* V, double* W,
                         double* ge) {
    double a1[9], a2[9], d1[14], d2[14], R[9], T[3], err = -1;
    memcpy(a1, A1, sizeof(a1)), memcpy(a2, A2, sizeof(a2));
    memcpy(d1, dist1, sizeof(d1)), memcpy(d2, dist2, sizeof(d2));
    flags = (flags | ORC_USE_INTRINSIC_GUESS) & ~ORC_FIX_INTRINSIC;
    g_capture.armed = 1;
    g_capture.U = U, g_capture.gs = gs, g_capture.V = V, g_capture.W = W, g_capture.ge = ge, g_capture.err = &err;
    orc_stereo_calibrate(nviews, npts, obj, img1, img2, a1, d1, a2, d2, 0, 0, R, T, 0, 0, flags, ORC_TERM_COUNT, 1, 0., rt, lr,
                         0, 1, 1, 0);
    g_capture.armed = 0;
    return err;
}
