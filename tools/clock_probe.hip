// Sustained shader clock under an FP64 MFMA / FP64 FMA load: shader-cycle counter against the 100 MHz wall clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void load_mfma(double* out, long long* clk, int iters) {
    double4_t acc = {0., 0., 0., 0.};
    double a = 1.0 + threadIdx.x * 1e-6;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0, clk[1] = w1 - w0;
}
__global__ void load_fma(double* out, long long* clk, int iters) {
    double acc[16];
    for (int k = 0; k < 16; k++) acc[k] = threadIdx.x + k;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[k] = acc[k] * 1.0000001 + 1e-9;
    long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int k = 0; k < 16; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0, clk[1] = w1 - w0;
}
int main() {
    double* d;
    long long *c, h[2];
    (void)hipMalloc(&d, sizeof(double) * 2048 * 256);
    (void)hipMalloc(&c, 16);
    for (int rep = 0; rep < 3; rep++) {
        load_mfma<<<2048, 256>>>(d, c, 100000);
        (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        printf("mfma f64: %lld shader ticks / %lld wall ticks (100 MHz) -> %.0f MHz, %.1f ms\n", h[0], h[1], 100.0 * h[0] / h[1], h[1] / 1e5);
    }
    for (int rep = 0; rep < 3; rep++) {
        load_fma<<<2048, 256>>>(d, c, 40000);
        (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        printf("fma f64:  %lld shader ticks / %lld wall ticks (100 MHz) -> %.0f MHz, %.1f ms\n", h[0], h[1], 100.0 * h[0] / h[1], h[1] / 1e5);
    }
    load_mfma<<<2048, 256>>>(d, c, 2000);
    (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("short mfma burst: %.0f MHz over %.3f ms\n", 100.0 * h[0] / h[1], h[1] / 1e5);
    return 0;
}
