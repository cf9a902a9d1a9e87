// Measures v_mfma_f64_16x16x4_f64 throughput on gfx950: NCH independent accumulator chains per wave,
// optionally with NV independent FP64 VALU FMAs issued between MFMAs (to see whether the two pipes overlap).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NCH, int NV>
__global__ void mfma_loop(double* out, int iters, double a0, double b0) {
    double4_t acc[NCH];
    double v[NV > 0 ? NV : 1];
#pragma unroll
    for (int k = 0; k < NCH; k++) acc[k] = double4_t{0., 0., 0., 0.};
#pragma unroll
    for (int k = 0; k < (NV > 0 ? NV : 1); k++) v[k] = threadIdx.x * 1e-3 + k;
    double a = a0 + threadIdx.x * 1e-6, b = b0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < NCH; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = v[k] * a0 + b0;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < NCH; k++) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
#pragma unroll
    for (int k = 0; k < NV; k++) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH, int NV>
void run(int waves_per_simd, int iters) {
    int block = 64 * waves_per_simd * 4 > 1024 ? 1024 : 64 * waves_per_simd * 4;
    int blocks_per_cu = (64 * waves_per_simd * 4) / block;
    int grid = 256 * blocks_per_cu;
    double* d;
    hipMalloc(&d, sizeof(double) * grid * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 10; w++) mfma_loop<NCH, NV><<<grid, block>>>(d, iters, 1.0000001, 1e-9);
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        mfma_loop<NCH, NV><<<grid, block>>>(d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
    }
    double nm = (double)NCH * iters * waves_per_simd;              // MFMAs per SIMD
    double cyc = ms * 1e-3 * 2.4e9 / nm;                            // at 2.4 GHz
    double tf = 2048.0 * NCH * iters * (double)grid * block / 64 / ms / 1e9;
    double vtf = 128.0 * NV * iters * (double)grid * block / 64 / ms / 1e9;
    printf("chains=%d valu_per_iter=%d waves/SIMD=%d: %.3f ms  %.1f cyc/mfma/SIMD  mfma %.2f TF  valu %.2f TF\n", NCH, NV,
           waves_per_simd, ms, cyc, tf, vtf);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) run<1, 0>(w, 20000);
    for (int w : {1, 2, 4}) run<2, 0>(w, 10000);
    for (int w : {1, 2}) run<4, 0>(w, 5000);
    for (int w : {1, 2, 4}) run<1, 8>(w, 20000);
    for (int w : {1, 2, 4}) run<1, 16>(w, 20000);
    for (int w : {1, 2}) run<2, 16>(w, 10000);
    return 0;
}
