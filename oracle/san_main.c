/*
 * san_main.c -- drives the CPU oracle under AddressSanitizer / UndefinedBehaviorSanitizer.
 * TEST INFRASTRUCTURE ONLY (built by `make -C oracle sancheck`, run by tests/test_oracle_sanitizers.py).
 *
 * Input file (written by the test): int32 nviews, int32 P, then doubles: obj[P*3], img1[nviews*P*2],
 * img2[nviews*P*2].  Runs every oracle entry point the tests use -- mono dense and Schur (two flag sets), stereo
 * dense and Schur, the reprojection re-check, the intrinsic seed, the block evaluation -- and prints the RMS values,
 * which the test compares with the ordinary build's.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "calib_oracle.h"

static void* xmalloc(size_t n) {
    void* p = malloc(n);
    if (!p) {
        fprintf(stderr, "out of memory\n");
        exit(2);
    }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[2];
    if (fread(hdr, sizeof(int), 2, f) != 2) return 2;
    const int N = hdr[0], P = hdr[1];
    double* board = (double*)xmalloc(sizeof(double) * P * 3);
    double* img1 = (double*)xmalloc(sizeof(double) * (size_t)N * P * 2);
    double* img2 = (double*)xmalloc(sizeof(double) * (size_t)N * P * 2);
    if (fread(board, sizeof(double), (size_t)P * 3, f) != (size_t)P * 3) return 2;
    if (fread(img1, sizeof(double), (size_t)N * P * 2, f) != (size_t)N * P * 2) return 2;
    if (fread(img2, sizeof(double), (size_t)N * P * 2, f) != (size_t)N * P * 2) return 2;
    fclose(f);
    int* npts = (int*)xmalloc(sizeof(int) * N);
    double* obj = (double*)xmalloc(sizeof(double) * (size_t)N * P * 3);
    for (int i = 0; i < N; i++) {
        npts[i] = P;
        memcpy(obj + (size_t)i * P * 3, board, sizeof(double) * P * 3);
    }
    double* rv = (double*)xmalloc(sizeof(double) * 3 * N);
    double* tv = (double*)xmalloc(sizeof(double) * 3 * N);
    double* pv = (double*)xmalloc(sizeof(double) * N);
    orc_stats st;
    double A[9], d[14], A1[9], d1[14], A2[9], d2[14];

    for (int schur = 0; schur < 2; schur++) {
        memset(A, 0, sizeof(A)), memset(d, 0, sizeof(d));
        A[0] = A[4] = A[8] = 1;
        double rms = orc_calibrate_camera(N, npts, obj, img1, 640, 480, A, d, 5, rv, tv, 0, 3, 30, 1e-9, 0, schur, -1,
                                          schur ? 2 : 0, &st);
        printf("mono schur=%d rms %.17g iters %d\n", schur, rms, st.iters);
        double rms2 = orc_reprojection_error(N, npts, obj, img1, A, d, 5, rv, tv, pv);
        printf("reproj schur=%d rms %.17g\n", schur, rms2);
        memset(A, 0, sizeof(A)), memset(d, 0, sizeof(d));
        A[0] = A[4] = A[8] = 1;
        rms = orc_calibrate_camera(N, npts, obj, img1, 640, 480, A, d, 5, rv, tv, ORC_FIX_ASPECT_RATIO | ORC_USE_LU, 3, 30,
                                   1e-9, 0, schur, -1, schur ? 2 : 0, &st);
        printf("mono aspect+lu schur=%d rms %.17g iters %d\n", schur, rms, st.iters);
    }
    orc_init_intrinsic_params_2d(N, npts, obj, img1, 640, 480, 0., A1);
    orc_init_intrinsic_params_2d(N, npts, obj, img2, 640, 480, 0., A2);
    printf("seed fx %.17g %.17g\n", A1[0], A2[0]);
    {
        double intr[18] = {0}, U[81], gc[9];
        int free_idx[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
        intr[0] = A[0], intr[1] = A[4], intr[2] = A[2], intr[3] = A[5];
        memcpy(intr + 4, d, sizeof(double) * 5);
        double* rt = (double*)xmalloc(sizeof(double) * 6 * N);
        for (int i = 0; i < N; i++) memcpy(rt + 6 * i, rv + 3 * i, 24), memcpy(rt + 6 * i + 3, tv + 3 * i, 24);
        double* V = (double*)xmalloc(sizeof(double) * 36 * N);
        double* W = (double*)xmalloc(sizeof(double) * 54 * N);
        double* ge = (double*)xmalloc(sizeof(double) * 6 * N);
        double e = orc_mono_blocks(N, npts, obj, img1, intr, 0., rt, 9, free_idx, U, gc, V, W, ge);
        printf("blocks err %.17g\n", e);
        free(rt), free(V), free(W), free(ge);
    }
    const int sflags = ORC_FIX_ASPECT_RATIO | ORC_ZERO_TANGENT_DIST | ORC_USE_INTRINSIC_GUESS | ORC_SAME_FOCAL_LENGTH |
                       ORC_RATIONAL_MODEL | ORC_FIX_K3 | ORC_FIX_K4 | ORC_FIX_K5;
    for (int schur = 0; schur < 2; schur++) {
        double K1[9], K2[9], R[9], T[3], E[9], F[9];
        memcpy(K1, A1, sizeof(K1)), memcpy(K2, A2, sizeof(K2));
        memset(d1, 0, sizeof(d1)), memset(d2, 0, sizeof(d2));
        double* rt = (double*)xmalloc(sizeof(double) * 6 * N);
        double rms = orc_stereo_calibrate(N, npts, obj, img1, img2, K1, d1, K2, d2, 640, 480, R, T, E, F, sflags, 3, 40,
                                          1e-6, 0, 0, rt, schur, schur ? 2 : 0, &st);
        printf("stereo schur=%d rms %.17g iters %d T %.17g\n", schur, rms, st.iters, T[0]);
        free(rt);
    }
    free(board), free(img1), free(img2), free(npts), free(obj), free(rv), free(tv), free(pv);
    return 0;
}
