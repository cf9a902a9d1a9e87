/* orc_levmarq.h -- restatement of OpenCV 4.1.0 CvLevMarq (compat_ptsetreg.cpp).
 * TEST INFRASTRUCTURE ONLY; parity unpinned (see calib_oracle.h). */
#ifndef ORC_LEVMARQ_H
#define ORC_LEVMARQ_H

enum { ORC_LM_DONE = 0, ORC_LM_STARTED = 1, ORC_LM_CALC_J = 2, ORC_LM_CHECK_ERR = 3 };

/* Custom step: solves the damped system for the current lambda and writes
 * param = prevParam - delta.  Used by the Schur variant, which never forms JtJ. */
struct orc_levmarq;
typedef void (*orc_step_fn)(struct orc_levmarq* lm, double lambda, void* user);

typedef struct orc_levmarq {
    int nparams, nerrs;
    unsigned char* mask;
    double *param, *prevParam;
    double *JtJ, *JtErr; /* dense variant only (NULL when custom_step is set) */
    double *J, *err;     /* update() variant only */
    double errNorm, prevErrNorm;
    int lambdaLg10, state, iters, max_iter;
    double epsilon;
    int solveMethod;
    orc_step_fn custom_step;
    void* user;
    /* optional per-iteration record (orc_set_trace): see calib_oracle.h */
    double* trace;
    int trace_cap, trace_n, tr_first, tr_trials;
} orc_levmarq;

/* dense != 0 allocates JtJ/JtErr; nerrs > 0 allocates J/err. */
void orc_lm_init(orc_levmarq* lm, int nparams, int nerrs, int term_type, int max_iter, double eps,
                 int dense);
void orc_lm_free(orc_levmarq* lm);
/* CvLevMarq::updateAlt.  *want_J / *want_err tell the caller what to fill:
 * JtJ+JtErr (zeroed here) and/or errNorm. Returns 0 when finished. */
int orc_lm_update_alt(orc_levmarq* lm, int* want_J, int* want_err);
/* CvLevMarq::update (explicit J and err). */
int orc_lm_update(orc_levmarq* lm, int* want_J, int* want_err);
#endif
