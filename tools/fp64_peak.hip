// Measures sustained FP64 FMA throughput of the vector ALU as a function of waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o tools/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC>
__global__ void fma_loop(double* out, int iters, double a, double b) {
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = threadIdx.x * 1e-3 + k;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) acc[k] = acc[k] * a + b;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < NACC; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, int iters) {
    int block = 64 * waves_per_simd * 4 > 1024 ? 1024 : 64 * waves_per_simd * 4;  // waves per CU in one block if possible
    int blocks_per_cu = (64 * waves_per_simd * 4) / block;
    int grid = 256 * blocks_per_cu;
    double* d;
    hipMalloc(&d, sizeof(double) * grid * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 20; w++) fma_loop<NACC><<<grid, block>>>(d, iters, 1.0000001, 1e-9);  // warm the clocks
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        fma_loop<NACC><<<grid, block>>>(d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
    }
    double flops = 2.0 * NACC * iters * (double)grid * block;
    printf("NACC=%d waves/SIMD=%d grid=%d block=%d: %.3f ms  %.2f TFLOP/s\n", NACC, waves_per_simd, grid, block, ms, flops / ms / 1e9);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) run<32>(w, 10000);
    for (int w : {1, 2, 4, 8}) run<16>(w, 20000);
    for (int w : {1, 2, 4, 8}) run<8>(w, 20000);
    for (int w : {1, 2, 4, 8}) run<4>(w, 40000);
    for (int w : {1, 2, 4, 8}) run<2>(w, 40000);
    for (int w : {1, 2, 4, 8}) run<1>(w, 80000);
    return 0;
}
