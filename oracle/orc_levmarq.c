/* orc_levmarq.c -- restatement of OpenCV 4.1.0 CvLevMarq::{init, updateAlt, update, step}
 * (modules/calib3d/src/compat_ptsetreg.cpp).  TEST INFRASTRUCTURE ONLY; parity unpinned. */
#include "orc_levmarq.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "calib_oracle.h"
#include "orc_math.h"

void orc_lm_init(orc_levmarq* lm, int nparams, int nerrs, int term_type, int max_iter, double eps,
                 int dense) {
    memset(lm, 0, sizeof(*lm));
    lm->nparams = nparams;
    lm->nerrs = nerrs;
    lm->mask = (unsigned char*)malloc(nparams);
    memset(lm->mask, 1, nparams);
    lm->param = (double*)calloc(nparams, sizeof(double));
    lm->prevParam = (double*)calloc(nparams, sizeof(double));
    if (dense) {
        lm->JtJ = (double*)calloc((size_t)nparams * nparams, sizeof(double));
        lm->JtErr = (double*)calloc(nparams, sizeof(double));
    }
    if (nerrs > 0) {
        lm->J = (double*)calloc((size_t)nerrs * nparams, sizeof(double));
        lm->err = (double*)calloc(nerrs, sizeof(double));
    }
    lm->errNorm = lm->prevErrNorm = DBL_MAX;
    lm->lambdaLg10 = -3;
    if (term_type & ORC_TERM_COUNT) {
        int mi = max_iter < 1 ? 1 : max_iter;
        lm->max_iter = mi > 1000 ? 1000 : mi;
    } else
        lm->max_iter = 30;
    if (term_type & ORC_TERM_EPS)
        lm->epsilon = eps > 0 ? eps : 0;
    else
        lm->epsilon = DBL_EPSILON;
    lm->state = ORC_LM_STARTED;
    lm->iters = 0;
    lm->solveMethod = ORC_SOLVE_SVD;
}

/* One record per finished iteration, the fields the product's driver reports for its own runs:
 * iteration, damping exponent of the first trial, exponent of the accepted trial, number of trials,
 * residual sum at the committed point, residual sum at the accepted point. */
static void lm_trace_record(orc_levmarq* lm) {
    if (!lm->trace || lm->trace_n >= lm->trace_cap) return;
    double* r = lm->trace + (size_t)ORC_TRACE_FIELDS * lm->trace_n++;
    r[0] = lm->iters;
    r[1] = lm->tr_first;
    r[2] = lm->lambdaLg10 < 16 ? lm->lambdaLg10 : 16;
    r[3] = lm->tr_trials;
    r[4] = lm->prevErrNorm;
    r[5] = lm->errNorm;
}

void orc_lm_free(orc_levmarq* lm) {
    free(lm->mask);
    free(lm->param);
    free(lm->prevParam);
    free(lm->JtJ);
    free(lm->JtErr);
    free(lm->J);
    free(lm->err);
    memset(lm, 0, sizeof(*lm));
}

static void lm_step(orc_levmarq* lm) {
    const double LOG10 = log(10.);
    double lambda = exp(lm->lambdaLg10 * LOG10);
    if (lm->custom_step) {
        lm->custom_step(lm, lambda, lm->user);
        return;
    }
    int n = lm->nparams, nz = 0;
    int* idx = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++)
        if (lm->mask[i]) idx[nz++] = i;
    double* A = (double*)calloc((size_t)nz * nz + 1, sizeof(double));
    double* b = (double*)calloc((size_t)nz + 1, sizeof(double));
    double* x = (double*)calloc((size_t)nz + 1, sizeof(double));
    for (int i = 0; i < nz; i++) {
        b[i] = lm->JtErr[idx[i]];
        for (int j = 0; j < nz; j++) A[(size_t)i * nz + j] = lm->JtJ[(size_t)idx[i] * n + idx[j]];
    }
    if (!lm->err) /* completeSymm: copy the upper half to the lower half */
        for (int i = 0; i < nz; i++)
            for (int j = 0; j < i; j++) A[(size_t)i * nz + j] = A[(size_t)j * nz + i];
    for (int i = 0; i < nz; i++) A[(size_t)i * nz + i] *= 1. + lambda;
    orc_solve(A, b, nz, lm->solveMethod, x);
    int j = 0;
    for (int i = 0; i < n; i++) lm->param[i] = lm->prevParam[i] - (lm->mask[i] ? x[j++] : 0);
    free(idx);
    free(A);
    free(b);
    free(x);
}

static double rel_l2(const double* a, const double* b, int n) {
    double d = 0, s = 0;
    for (int i = 0; i < n; i++) {
        d += (a[i] - b[i]) * (a[i] - b[i]);
        s += b[i] * b[i];
    }
    return sqrt(d) / sqrt(s); /* cvNorm(a, b, CV_RELATIVE_L2) */
}

int orc_lm_update_alt(orc_levmarq* lm, int* want_J, int* want_err) {
    *want_J = *want_err = 0;
    if (lm->state == ORC_LM_DONE) return 0;
    if (lm->state == ORC_LM_STARTED) {
        if (lm->JtJ) memset(lm->JtJ, 0, sizeof(double) * (size_t)lm->nparams * lm->nparams);
        if (lm->JtErr) memset(lm->JtErr, 0, sizeof(double) * lm->nparams);
        lm->errNorm = 0;
        *want_J = *want_err = 1;
        lm->state = ORC_LM_CALC_J;
        return 1;
    }
    if (lm->state == ORC_LM_CALC_J) {
        memcpy(lm->prevParam, lm->param, sizeof(double) * lm->nparams);
        lm->tr_first = lm->lambdaLg10, lm->tr_trials = 1;
        lm_step(lm);
        lm->prevErrNorm = lm->errNorm;
        lm->errNorm = 0;
        *want_err = 1;
        lm->state = ORC_LM_CHECK_ERR;
        return 1;
    }
    /* CHECK_ERR */
    if (lm->errNorm > lm->prevErrNorm) {
        if (++lm->lambdaLg10 <= 16) {
            lm->tr_trials++;
            lm_step(lm);
            lm->errNorm = 0;
            *want_err = 1;
            lm->state = ORC_LM_CHECK_ERR;
            return 1;
        }
    }
    lm_trace_record(lm);
    lm->lambdaLg10 = lm->lambdaLg10 - 1 > -16 ? lm->lambdaLg10 - 1 : -16;
    if (++lm->iters >= lm->max_iter ||
        rel_l2(lm->param, lm->prevParam, lm->nparams) < lm->epsilon) {
        lm->state = ORC_LM_DONE;
        return 0;
    }
    lm->prevErrNorm = lm->errNorm;
    if (lm->JtJ) memset(lm->JtJ, 0, sizeof(double) * (size_t)lm->nparams * lm->nparams);
    if (lm->JtErr) memset(lm->JtErr, 0, sizeof(double) * lm->nparams);
    *want_J = 1; /* errNorm is not requested on this pass: it keeps the accepted value */
    lm->state = ORC_LM_CALC_J;
    return 1;
}

static double l2(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

int orc_lm_update(orc_levmarq* lm, int* want_J, int* want_err) {
    int n = lm->nparams, m = lm->nerrs;
    *want_J = *want_err = 0;
    if (lm->state == ORC_LM_DONE) return 0;
    if (lm->state == ORC_LM_STARTED) {
        memset(lm->J, 0, sizeof(double) * (size_t)m * n);
        memset(lm->err, 0, sizeof(double) * m);
        *want_J = *want_err = 1;
        lm->state = ORC_LM_CALC_J;
        return 1;
    }
    if (lm->state == ORC_LM_CALC_J) {
        for (int a = 0; a < n; a++) {
            for (int b = 0; b < n; b++) {
                double s = 0;
                for (int i = 0; i < m; i++) s += lm->J[(size_t)i * n + a] * lm->J[(size_t)i * n + b];
                lm->JtJ[a * n + b] = s;
            }
            double s = 0;
            for (int i = 0; i < m; i++) s += lm->J[(size_t)i * n + a] * lm->err[i];
            lm->JtErr[a] = s;
        }
        memcpy(lm->prevParam, lm->param, sizeof(double) * n);
        lm_step(lm);
        if (lm->iters == 0) lm->prevErrNorm = l2(lm->err, m);
        memset(lm->err, 0, sizeof(double) * m);
        *want_err = 1;
        lm->state = ORC_LM_CHECK_ERR;
        return 1;
    }
    lm->errNorm = l2(lm->err, m);
    if (lm->errNorm > lm->prevErrNorm) {
        if (++lm->lambdaLg10 <= 16) {
            lm_step(lm);
            memset(lm->err, 0, sizeof(double) * m);
            *want_err = 1;
            lm->state = ORC_LM_CHECK_ERR;
            return 1;
        }
    }
    lm->lambdaLg10 = lm->lambdaLg10 - 1 > -16 ? lm->lambdaLg10 - 1 : -16;
    if (++lm->iters >= lm->max_iter || rel_l2(lm->param, lm->prevParam, n) < lm->epsilon) {
        lm->state = ORC_LM_DONE;
        return 1; /* returns true with nothing requested; callers stop on !want_err */
    }
    lm->prevErrNorm = lm->errNorm;
    memset(lm->J, 0, sizeof(double) * (size_t)m * n);
    *want_J = *want_err = 1;
    lm->state = ORC_LM_CALC_J;
    return 1;
}
