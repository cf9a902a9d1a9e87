#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    hipFree(0);
    std::vector<float> h(10800000, 1.f);   // 43 MB
    for (int rep = 0; rep < 3; rep++) {
        void *a, *b, *c, *d;
        auto t0 = now();
        hipMalloc(&a, 43200000);
        auto t1 = now();
        hipMemcpy(a, h.data(), 43200000, hipMemcpyHostToDevice);
        auto t2 = now();
        hipMalloc(&b, 64800000); hipMalloc(&c, 64800000);
        auto t3 = now();
        hipMalloc(&d, 4800000);
        auto t4 = now();
        hipFree(a); hipFree(b); hipFree(c); hipFree(d);
        auto t5 = now();
        std::printf("malloc 43MB %.3f ms, memcpy 43MB pageable %.3f ms (%.1f GB/s), 2x malloc 65MB %.3f ms, malloc 4.8MB %.3f ms, 4 frees %.3f ms\n",
                    ms(t0, t1), ms(t1, t2), 43.2 / ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5));
    }
    return 0;
}
