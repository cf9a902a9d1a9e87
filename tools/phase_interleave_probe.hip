// Model of one view of the fused Jacobian kernel on one wavefront: NV dependent FP64 FMAs (the point chain, split
// over ILP independent sub-chains) and NM v_mfma_f64_4x4x4_4b on five accumulators (the matrix phase), either one
// after the other ("phased", what the kernel does) or interleaved in one instruction stream.  Twelve wavefronts per
// CU (three per SIMD), as the kernel runs.  Prints cycles per view per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/phase_interleave_probe.hip -o tools/phase_interleave_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NV, int NM, int ILP, bool INTER>
__global__ __launch_bounds__(768, 1) void view_loop(double* out, int views, double a0, double b0) {
    extern __shared__ double lds[];  // only to cap the CU at one workgroup of twelve wavefronts
    double acc[5] = {0., 0., 0., 0., 0.};
    double ch[ILP];
    for (int k = 0; k < ILP; k++) ch[k] = threadIdx.x * 1e-3 + k;
    const double a = a0 + threadIdx.x * 1e-6, b = b0;
    for (int v = 0; v < views; v++) {
        if constexpr (!INTER) {
#pragma unroll
            for (int i = 0; i < NV; i++) ch[i % ILP] = ch[i % ILP] * a0 + b0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NM; i++) acc[i % 5] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i % 5], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            constexpr int per = (NV + NM - 1) / NM;  // FMAs behind each MFMA
#pragma unroll
            for (int i = 0; i < NM; i++) {
                acc[i % 5] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i % 5], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < per; j++)
                    if (i * per + j < NV) ch[(i * per + j) % ILP] = ch[(i * per + j) % ILP] * a0 + b0;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    double s = 0;
    for (int k = 0; k < 5; k++) s += acc[k];
    for (int k = 0; k < ILP; k++) s += ch[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 12345) lds[0] = s;
}

template <int NV, int NM, int ILP, bool INTER>
void run(int views) {
    const int grid = 256, block = 768;
    double* d;
    hipMalloc(&d, sizeof(double) * grid * block);
    auto k = view_loop<NV, NM, ILP, INTER>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<<<grid, block, 150 * 1024>>>(d, views, 1.0000001, 1e-9);
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k<<<grid, block, 150 * 1024>>>(d, views, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
    }
    // three wavefronts per SIMD each do `views` views
    printf("NV=%3d NM=%2d ILP=%d %-11s %8.1f cycles per view per SIMD (three wavefronts share it)\n", NV, NM, ILP,
           INTER ? "interleaved" : "phased", ms * 1e-3 * 2.4e9 / (3.0 * views));
    hipFree(d);
}

int main() {
    run<156, 70, 1, false>(2000);
    run<156, 70, 1, true>(2000);
    run<156, 70, 2, false>(2000);
    run<156, 70, 2, true>(2000);
    run<156, 70, 4, false>(2000);
    run<156, 70, 4, true>(2000);
    run<156, 0, 1, false>(2000);
    run<156, 0, 2, false>(2000);
    run<0, 70, 1, false>(2000);
    return 0;
}
