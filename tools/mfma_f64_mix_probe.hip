// Does other work overlap with v_mfma_f64_16x16x4_f64 on gfx950?  One MFMA per iteration plus NV
// independent operations of one kind issued behind it: FP64 FMA, FP32 FMA, integer add, LDS read (b128).
// Prints cycles per iteration per SIMD (at 2.4 GHz) for 1 and 3 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_mix_probe.hip -o tools/mfma_f64_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int KIND, int NV, bool MFMA>
__global__ void mix_loop(double* out, int iters, double a0, double b0, int i0) {
    __shared__ double2 sm[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = make_double2(i * 1e-3, 1.);
    __syncthreads();
    double4_t acc = {0., 0., 0., 0.};
    double vd[NV > 0 ? NV : 1];
    float vf[NV > 0 ? NV : 1];
    int vi[NV > 0 ? NV : 1];
    double2 vl[NV > 0 ? NV : 1];
    for (int k = 0; k < (NV > 0 ? NV : 1); k++) vd[k] = threadIdx.x * 1e-3 + k, vf[k] = threadIdx.x * 1e-3f + k, vi[k] = threadIdx.x + k, vl[k] = make_double2(0., 0.);
    double a = a0 + threadIdx.x * 1e-6, b = b0;
    float fa = (float)a0, fb = (float)b0;
    int idx = (threadIdx.x & 63);
    for (int i = 0; i < iters; i++) {
        if constexpr (MFMA) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; k++) {
            if constexpr (KIND == 0) vd[k] = vd[k] * a0 + b0;
            if constexpr (KIND == 1) vf[k] = vf[k] * fa + fb;
            if constexpr (KIND == 2) vi[k] = vi[k] * 3 + i0;
            if constexpr (KIND == 3) {
                double2 t = sm[(idx + 64 * k) & 1023];
                vl[k].x += t.x, vl[k].y += t.y;   // two FP64 adds ride along
            }
        }
        if constexpr (KIND == 3) idx = (idx + i0) & 1023;
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int k = 0; k < NV; k++) s += vd[k] + vf[k] + vi[k] + vl[k].x + vl[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, bool MFMA>
void run(int waves_per_simd, int iters) {
    int block = 256;
    int grid = 256 * waves_per_simd;
    double* d;
    hipMalloc(&d, sizeof(double) * grid * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) mix_loop<KIND, NV, MFMA><<<grid, block>>>(d, iters, 1.0000001, 1e-9, 1);
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        mix_loop<KIND, NV, MFMA><<<grid, block>>>(d, iters, 1.0000001, 1e-9, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
    }
    static const char* names[] = {"fp64 fma", "fp32 fma", "int mad", "lds b128 + 2 fp64 add"};
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd);
    printf("%-22s x%2d %s waves/SIMD=%d: %7.1f cycles / iteration / SIMD\n", names[KIND], NV, MFMA ? "+ mfma" : "alone ", waves_per_simd, cyc);
    hipFree(d);
}

template <int KIND>
void kind() {
    for (int w : {1, 3}) {
        run<KIND, 8, false>(w, 20000);
        run<KIND, 8, true>(w, 20000);
        run<KIND, 16, false>(w, 20000);
        run<KIND, 16, true>(w, 20000);
    }
}

int main() {
    for (int w : {1, 3}) run<0, 0, true>(w, 20000);
    kind<0>();
    kind<1>();
    kind<2>();
    kind<3>();
    return 0;
}
