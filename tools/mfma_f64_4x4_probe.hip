// Probes v_mfma_f64_4x4x4_4b_f64 on gfx950: operand/result lane layout and throughput.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4_probe.hip -o tools/mfma_f64_4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int* out) {  // out[la * 64 + lb] = lane that received a non-zero (or -1), one block of 64 lanes
    int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
            unsigned long long m = __ballot(d != 0.);
            if (lane == 0) out[la * 64 + lb] = m ? __ffsll((long long)m) - 1 : -1;
        }
}

template <int NCH>
__global__ void rate(double* out, int iters, double a0) {
    double acc[NCH];
    for (int k = 0; k < NCH; k++) acc[k] = 0.;
    double a = a0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int k = 0; k < NCH; k++) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, acc[k], 0, 0, 0);
    double s = 0;
    for (int k = 0; k < NCH; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH>
void run_rate(int wps) {
    int block = 256 * wps > 1024 ? 1024 : 256 * wps, grid = 256 * (256 * wps / block), iters = 20000;
    double* d;
    (void)hipMalloc(&d, sizeof(double) * grid * block);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int w = 0; w < 20; w++) rate<NCH><<<grid, block>>>(d, iters, 1.0);
    (void)hipEventRecord(e0);
    rate<NCH><<<grid, block>>>(d, iters, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("4x4x4_4b chains=%d waves/SIMD=%d: %.1f cycles/instr/SIMD at 2.4 GHz\n", NCH, wps, ms * 1e-3 * 2.4e9 / ((double)NCH * iters * wps));
    (void)hipFree(d);
}

int main() {
    int* d;
    (void)hipMalloc(&d, sizeof(int) * 4096);
    probe<<<1, 64>>>(d);
    std::vector<int> h(4096);
    (void)hipMemcpy(h.data(), d, sizeof(int) * 4096, hipMemcpyDeviceToHost);
    // print for each A lane the set of B lanes that pair with it, and where the result lands
    for (int la = 0; la < 64; la++) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; lb++)
            if (h[la * 64 + lb] >= 0) printf(" B%02d->D%02d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    run_rate<1>(1), run_rate<1>(2), run_rate<3>(1), run_rate<3>(2);
    return 0;
}
