// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result lane layout and issue cost.
//  1. layout: for every pair (la, lb) the instruction runs with A = 1 in lane la only and B = 1 in lane lb
//     only; the lanes whose result is non-zero are recorded.  From the table the program derives, for
//     each lane, which (block, i, k) its A value, which (block, k, j) its B value and which (block, i, j)
//     its result stands for, and checks the derived maps against a random product.
//  2. cost: cycles per instruction for one dependent chain and for three chains issued round-robin,
//     at 1 and 2 wavefronts per SIMD, next to the 16x16x4 form.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4_probe.hip -o tools/mfma_f64_4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void layout_probe(unsigned long long* table) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
            unsigned long long m = __ballot(d != 0.);
            if (lane == 0) table[la * 64 + lb] = m;
        }
}

__global__ void product_probe(const double* a, const double* b, double* d) {
    const int lane = threadIdx.x;
    d[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], 0., 0, 0, 0);
}

template <int CHAINS, bool SMALL>
__global__ void cost_loop(double* out, int iters, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-6, b = b0;
    double acc1[3] = {0., 0., 0.};
    double4_t acc4[3] = {{0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if constexpr (SMALL) acc1[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[c], 0, 0, 0);
            else acc4[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[c], 0, 0, 0);
        }
    }
    double s = 0;
    for (int c = 0; c < 3; c++) s += acc1[c] + acc4[c][0] + acc4[c][1] + acc4[c][2] + acc4[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, bool SMALL>
void cost(int waves_per_simd, int iters) {
    const int block = 256, grid = 256 * waves_per_simd;
    double* d;
    hipMalloc(&d, sizeof(double) * grid * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) cost_loop<CHAINS, SMALL><<<grid, block>>>(d, iters, 1.0000001, 1e-9);
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        cost_loop<CHAINS, SMALL><<<grid, block>>>(d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
    }
    // instructions issued per SIMD = waves_per_simd * iters * CHAINS
    double cyc = ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * CHAINS);
    std::printf("%-9s chains=%d waves/SIMD=%d : %.1f cycles per instruction per SIMD (at 2.4 GHz), %.3f ms\n",
                SMALL ? "4x4x4_4b" : "16x16x4", CHAINS, waves_per_simd, cyc, ms);
    hipFree(d);
}


// 3. the loop shape a symmetric 16-column product would use: per eight rows five 4x4x4 instructions on
//    five accumulators, their operands read from LDS one trip ahead (two 16-byte and two 8-byte reads at
//    per-lane addresses, RS doubles between columns), the four addresses advanced by vector adds.
template <bool BIG>
__global__ void shape_loop(double* out, int iters, int RS, int rows) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* buf = lds + w * 16 * RS;
    for (int i = lane; i < 16 * RS; i += 64) buf[i] = 1e-3 * i;
    __syncthreads();
    const int c = lane & 15, rq = lane >> 4;
    const double2* b2 = reinterpret_cast<const double2*>(buf);
    int aN = (c * RS >> 1) + rq, aN1 = ((((c + 4) & 15) * RS) >> 1) + rq;
    int aA = (c < 8 ? c * RS + 2 * rq : (c - 8) * RS + 2 * rq + 1), aB = (c < 8 ? (c + 8) * RS + 2 * rq : c * RS + 2 * rq + 1);
    double acc[5] = {0., 0., 0., 0., 0.};
    double4_t big = {0., 0., 0., 0.};
    auto small5 = [&](const double2& n, const double2& n1, double ar, double br) {
        acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(n.x, n.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(n.x, n1.x, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, br, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(n.y, n.y, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(n.y, n1.y, acc[4], 0, 0, 0);
    };
    auto big2 = [&](const double2& n) {
        big = __builtin_amdgcn_mfma_f64_16x16x4f64(n.x, n.x, big, 0, 0, 0);
        big = __builtin_amdgcn_mfma_f64_16x16x4f64(n.y, n.y, big, 0, 0, 0);
    };
    const int groups = rows / 8;   // even
    for (int it = 0; it < iters; it++) {
        // two register sets take turns; the reads of a group are issued one group ahead
        int oN = aN, oN1 = aN1, oA = aA, oB = aB;
        double2 pn = b2[oN], pn1, qn, qn1;
        double par = 0, pbr = 0, qar = 0, qbr = 0;
        if constexpr (!BIG) pn1 = b2[oN1], par = buf[oA], pbr = buf[oB];
        for (int g = 0; g < groups; g += 2) {
            qn = b2[oN + 4];
            if constexpr (!BIG) qn1 = b2[oN1 + 4], qar = buf[oA + 8], qbr = buf[oB + 8];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BIG) big2(pn); else small5(pn, pn1, par, pbr);
            oN += 8;
            if constexpr (!BIG) oN1 += 8, oA += 16, oB += 16;
            pn = b2[oN];
            if constexpr (!BIG) pn1 = b2[oN1], par = buf[oA], pbr = buf[oB];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BIG) big2(qn); else small5(qn, qn1, qar, qbr);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + big[0] + big[1] + big[2] + big[3];
}

template <bool BIG>
void shape(int wgs_per_cu, int iters, int RS, int rows) {
    const int block = 256, grid = 256 * wgs_per_cu;
    double* d;
    hipMalloc(&d, sizeof(double) * grid * block);
    const size_t ldsb = sizeof(double) * 4 * 16 * RS + 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) shape_loop<BIG><<<grid, block, ldsb>>>(d, iters, RS, rows);
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        shape_loop<BIG><<<grid, block, ldsb>>>(d, iters, RS, rows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
    }
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * (rows / 8));
    std::printf("%s RS=%d rows=%d waves/SIMD=%d : %.1f cycles per eight rows per SIMD, %.3f ms\n",
                BIG ? "2 x 16x16x4 (one 16-byte read)" : "5 x 4x4x4_4b", RS, rows, wgs_per_cu, cyc, ms);
    hipFree(d);
}

int main() {
    unsigned long long* dt;
    hipMalloc(&dt, sizeof(unsigned long long) * 4096);
    layout_probe<<<1, 64>>>(dt);
    std::vector<unsigned long long> t(4096);
    hipMemcpy(t.data(), dt, sizeof(unsigned long long) * 4096, hipMemcpyDeviceToHost);
    // lanes la and lb "meet" when some result lane is non-zero
    std::printf("A lane -> B lanes it meets -> result lane (first 20 A lanes)\n");
    for (int la = 0; la < 20; la++) {
        std::printf("A%2d:", la);
        for (int lb = 0; lb < 64; lb++)
            if (t[la * 64 + lb]) {
                int first = __builtin_ctzll(t[la * 64 + lb]);
                int cnt = __builtin_popcountll(t[la * 64 + lb]);
                std::printf(" B%d->D%d%s", lb, first, cnt > 1 ? "(+)" : "");
            }
        std::printf("\n");
    }
    // the table reads: A lane = 16 k + 4 block + i, B lane = 16 k + 4 block + j, result lane = 16 i + 4 block + j
    // (k runs over lane / 16 as in the 16x16x4 form; the four blocks sit side by side in each row of 16 lanes)
    std::vector<double> a(64), b(64), d(64);
    for (int l = 0; l < 64; l++) a[l] = std::rand() / (double)RAND_MAX, b[l] = std::rand() / (double)RAND_MAX;
    double *da, *db, *dd;
    hipMalloc(&da, 512), hipMalloc(&db, 512), hipMalloc(&dd, 512);
    hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    product_probe<<<1, 64>>>(da, db, dd);
    hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int blk = 0; blk < 4; blk++)
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += a[16 * k + 4 * blk + i] * b[16 * k + 4 * blk + j];
                double e = s - d[16 * i + 4 * blk + j];
                worst = e < 0 ? (-e > worst ? -e : worst) : (e > worst ? e : worst);
            }
    std::printf("A/B lane = 16 k + 4 blk + (i|j), D lane = 16 i + 4 blk + j : max error %.3g\n", worst);
    const int iters = 20000;
    cost<1, true>(1, iters), cost<3, true>(1, iters), cost<1, true>(2, iters), cost<3, true>(2, iters);
    cost<1, false>(1, iters), cost<3, false>(1, iters), cost<3, false>(2, iters);
    for (int wpc = 1; wpc <= 3; wpc++) shape<false>(wpc, 2000, 114, 112);
    for (int wpc = 1; wpc <= 3; wpc++) shape<true>(wpc, 2000, 114, 112);
    return 0;
}
