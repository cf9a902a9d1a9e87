// Host <-> GPU turn-around of three ways to start a dependent kernel once the host has seen a result:
//   A  launch after the result arrives (what a host-driven loop does)
//   B  kernel enqueued in advance behind hipStreamWaitValue64 on a pinned word the host writes
//   C  as B, with a chain of four kernels behind the wait (the shape of one LM iteration)
// Each round: the GPU writes a pinned flag (kernel k_signal), the host polls it, "decides" and releases the next.
// Build: hipcc --offload-arch=gfx950 -O3 tools/waitvalue_probe.hip -o tools/waitvalue_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(e)                                                                         \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) {                                                       \
            std::printf("%s failed: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

__global__ void k_signal(volatile unsigned long long* flag, unsigned long long v, const volatile double* ctrl, double* sink) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        sink[0] = ctrl[0];  // read something the host wrote just before the release
        __threadfence_system();
        *flag = v;
    }
}
__global__ void k_work(double* sink, int n) {
    double a = threadIdx.x;
    for (int i = 0; i < n; i++) a = a * 1.0000001 + 1e-9;
    if (a == 12345.678) sink[1] = a;
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *flag, *go;
    double *ctrl, *sink;
    CK(hipHostMalloc(&flag, 8, hipHostMallocCoherent));
    CK(hipHostMalloc(&go, 8, hipHostMallocCoherent));
    CK(hipHostMalloc(&ctrl, 64, hipHostMallocCoherent));
    CK(hipMalloc(&sink, 64));
    *flag = 0, *go = 0, ctrl[0] = 1;
    const int rounds = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto wait_flag = [&](unsigned long long v) {
        while (*(volatile unsigned long long*)flag != v) __builtin_ia32_pause();
    };
    // A: launch on demand
    {
        unsigned long long seq = 0;
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, flag, ++seq, ctrl, sink);
        wait_flag(seq);
        auto t0 = now();
        for (int i = 0; i < rounds; i++) {
            ctrl[0] = i;
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, flag, ++seq, ctrl, sink);
            wait_flag(seq);
        }
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / rounds;
        std::printf("A  launch-on-demand round trip          %.2f us\n", us);
    }
    if (!can) return 0;
    // B: one kernel parked behind a wait, enqueued one round ahead
    for (int chain = 1; chain <= 4; chain += 3) {
        unsigned long long seq = *flag, g = *go;
        auto enqueue = [&](unsigned long long gv, unsigned long long sv) {
            CK(hipStreamWaitValue64(st, go, gv, hipStreamWaitValueGte, ~0ull));
            for (int k = 1; k < chain; k++) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, st, sink, 1);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, flag, sv, ctrl, sink);
        };
        enqueue(g + 1, seq + 1);
        auto t0 = now();
        for (int i = 0; i < rounds; i++) {
            ctrl[0] = i;
            __atomic_store_n(go, g + 1 + i, __ATOMIC_RELEASE);       // release round i
            enqueue(g + 2 + i, seq + 2 + i);                          // park round i + 1 meanwhile
            wait_flag(seq + 1 + i);
        }
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / rounds;
        __atomic_store_n(go, g + 1 + rounds, __ATOMIC_RELEASE);
        wait_flag(seq + 1 + rounds);
        CK(hipStreamSynchronize(st));
        std::printf("%s  parked behind hipStreamWaitValue64 (%d kernel%s) round trip   %.2f us\n", chain == 1 ? "B" : "C", chain,
                    chain == 1 ? "" : "s", us);
    }
    // E: as A, but every round is a long kernel (~150 us of work on all CUs) followed by the signal: the host
    //    turn-around after real work = round time - kernel time (events)
    {
        const int n = 9000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, st, sink, n);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, st, sink, n);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double kernel_us = ms * 1e3 / 20;
        unsigned long long seq = *flag;
        const int r2 = 300;
        auto t0 = now();
        for (int i = 0; i < r2; i++) {
            ctrl[0] = i;
            hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, st, sink, n);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, flag, ++seq, ctrl, sink);
            wait_flag(seq);
        }
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / r2;
        std::printf("E  long kernel (%.1f us back to back) + signal, launched on demand: %.2f us per round -> turn-around %.2f us\n",
                    kernel_us, us, us - kernel_us);
    }
    // D: four dependent kernels launched back to back, no host in between (the floor for a chain of four)
    {
        unsigned long long seq = *flag;
        auto t0 = now();
        for (int i = 0; i < rounds; i++) {
            for (int k = 1; k < 4; k++) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, st, sink, 1);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, flag, ++seq, ctrl, sink);
        }
        wait_flag(seq);
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / rounds;
        std::printf("D  four kernels back to back, no host wait  %.2f us per chain\n", us);
    }
    return 0;
}
