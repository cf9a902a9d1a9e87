// fused12.hip -- the fused Jacobian pass for column sets of at most six intrinsics, on a 12-column tile.
//
// Same pass as k_jac_fused (kernels.hip): per view the normal-equation update is A^T A for the view's row matrix
// A = [Ji | Je | e], taken in 4x4 blocks on the FP64 matrix pipe (v_mfma_f64_4x4x4_4b_f64).  What is different:
//
//   * The two rows of a point (its u row and its v row) go through separate accumulators anyway -- lane p stores
//     (u value, v value) as one 16-byte LDS entry and the .x and .y halves feed different instructions.  The
//     principal-point columns are (1, 0) and (0, 1) down those rows, so ONE column of ones serves both: its
//     products over the u rows are cx's entries, its products over the v rows are cy's.  The reference's stereo flag
//     set (binocular_chessboard.cpp:156-161: fy with fx tied, cx, cy, k1, k2, k6 per camera) then has
//     5 + 6 pose + 1 residual = 12 tile columns instead of 13.
//   * Twelve columns are three groups of four.  The diagonal blocks come from (N, N) and the blocks (g, g+1 mod 3)
//     from (N, N shifted by four columns); (2, 0) is the transpose of (0, 2), so these two instructions per row set
//     already cover all six blocks of the upper triangle: four instructions per four points instead of five, two
//     16-byte operand reads instead of two 16-byte and two 8-byte ones.
//   * Both cameras of a stereo pass run in one grid (blockIdx.y = camera): a 10 000-pair problem presents twice the
//     workgroups, and a launch boundary disappears from every pass.
//
// cvStereoCalibrateImpl / cvCalibrateCamera2Internal (OpenCV 4.1.0 calibration.cpp) accumulate the same sums row by
// row; reference call sites binocular_chessboard.cpp:152-162, binocular_circle.cpp:153-163.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_math.hpp"
#include "kernels.hpp"

namespace ccalib {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
extern __shared__ float lds12[];

// Tile columns of a column set: its intrinsic columns with cx|cy merged into one, six pose columns, the residual.
template <class M>
struct Tile12 {
    static constexpr int NC = M::NC;
    static constexpr int ICX = M::index_of(I_CX), ICY = M::index_of(I_CY);
    static constexpr bool fits = NC <= 6 && ICX >= 0 && ICY == ICX + 1;
    static constexpr int NI = NC - 1;                      // intrinsic tile columns
    static constexpr int CPOSE = NI, CRES = NI + 6, NCOL = NI + 7;
    // tile column (< NI) -> intrinsic column; the merged column reports ICX
    __host__ __device__ static constexpr int intr_of(int tc) { return tc <= ICX ? tc : tc + 1; }
    // logical column (0..NC-1 intrinsic, NC..NC+5 pose, NC+6 residual) of tile column tc in the u rows (v = false) or
    // the v rows (v = true); -1 for a column the set does not use
    __host__ __device__ static constexpr int logical(int tc, bool v) {
        if (tc >= NCOL) return -1;
        if (tc >= CRES) return NC + 6;
        if (tc >= CPOSE) return NC + (tc - CPOSE);
        if (tc == ICX) return v ? ICY : ICX;
        return intr_of(tc);
    }
};

// Destination of the product of logical columns (lo <= hi): an entry of the view's blocks (qv: V, ge, W) or of the
// shared sums (qu: U upper triangle, gc, err).
template <class M>
__device__ __forceinline__ void destination(int a, int b, int& qv, int& qu) {
    constexpr int NC = M::NC, NU = M::NU;
    qv = -1, qu = -1;
    if (a < 0 || b < 0) return;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    if (hi < NC)
        qu = lo * NC - lo * (lo - 1) / 2 + (hi - lo);
    else if (hi < NC + 6) {
        if (lo < NC)
            qv = 27 + 6 * lo + (hi - NC);
        else
            qv = tri6(lo - NC, hi - NC);
    } else {
        if (lo < NC)
            qu = NU + lo;
        else if (lo < NC + 6)
            qv = 21 + (lo - NC);
        else
            qu = NU + NC;
    }
}

template <class M, bool SHARED, bool MULTI, int TPB>
__global__ __launch_bounds__(kBlock* TPB, TPB == 1 ? 3 : 1) void k_jac_fused12(PassArgs a, Fused12Cams cams, int RS, int CH) {
    using T = Tile12<M>;
    constexpr int NC = M::NC, NB = M::NBLK, NR = M::NRED_U, NCOL = T::NCOL;
    static_assert(T::fits, "column set does not fit the 12-column tile");
    static_assert(kWaves == kTileViews, "one wavefront per view of a tile");
    const Fused12Cam& cm = cams.c[blockIdx.y];
    const CamParams cp = cm.cp;
    const float2* __restrict__ img = cm.img;
    const double* __restrict__ pc = cm.pc;
    double* __restrict__ blk = cm.blk;
    double* __restrict__ part = cm.part;
    const int* __restrict__ off = a.off;
    const int lane = threadIdx.x & 63, wl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = wl & (kWaves - 1), tb = wl / kWaves;  // view of the tile, tile of the workgroup
    if (a.shared_board) {
        for (int i = threadIdx.x; i < 3 * a.board_np; i += blockDim.x) lds12[i] = a.obj[i];
    }
    const int board_f = SHARED ? (3 * a.board_np + 3) / 4 * 4 : 0;  // floats, 16-byte granules
    double* const ldsd = reinterpret_cast<double*>(lds12);
    const int cbase = board_f / 2;
    const int WS = (12 * RS + 31) & ~31;           // doubles between the wavefronts' buffers
    const int jt = cbase + wl * WS;                // this wavefront's buffer: 12 columns of RS doubles
    const int pslot = cbase + TPB * kWaves * WS + wl * kPoseConstStride;  // its pose-record slot
    for (int c = NCOL; c < 12; c++)  // columns the set does not use stay zero
        for (int r = lane; r < RS; r += 64) ldsd[jt + c * RS + r] = 0.;
    __syncthreads();
    // Result lane = 16 i + 4 block + j.  D: block (g, g), kept for i <= j; E: block (g, (g + 1) mod 3), all of it.
    // Lanes of the fourth block position compute products nobody reads.
    const int col = lane & 15, rq = lane >> 4;
    const int ri = lane >> 4, rg = (lane >> 2) & 3, rj = lane & 3;
    // Each result exists once for the u rows and once for the v rows.  Where neither column is the merged
    // principal-point column both belong to the same sum and are added; where one is, the u-row product is cx's
    // entry and the v-row product cy's.
    int qv[4], qu[4];  // D over u (or u + v), D over v, E over u (or u + v), E over v
    bool split[2];
    {
        const int tp = 4 * rg + ri;
        const int tq[2] = {4 * rg + rj, 4 * ((rg + 1) % 3) + rj};
        const bool keep[2] = {rg < 3 && ri <= rj, rg < 3};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            int vu, uu, vv, uv;
            destination<M>(keep[r] ? T::logical(tp, false) : -1, keep[r] ? T::logical(tq[r], false) : -1, vu, uu);
            destination<M>(keep[r] ? T::logical(tp, true) : -1, keep[r] ? T::logical(tq[r], true) : -1, vv, uv);
            split[r] = vu != vv || uu != uv;
            qv[2 * r] = vu, qu[2 * r] = uu;
            qv[2 * r + 1] = split[r] ? vv : -1, qu[2 * r + 1] = split[r] ? uv : -1;
        }
    }
    int voff[4];  // byte offset of each result inside a tile of blk (past the tile: dropped by the range check)
#pragma unroll
    for (int r = 0; r < 4; r++) voff[r] = qv[r] >= 0 ? (qv[r] * kTileViews + w) * (int)sizeof(double) : 0x7fffffff;
    double usum[4] = {0., 0., 0., 0.};
    const int ntiles = (a.nviews + kTileViews - 1) / kTileViews;
    int first = 0, np = 0;
    float2 mpt = make_float2(0.f, 0.f);
    float bX = 0.f, bY = 0.f, bZ = 0.f;
    unsigned qword = 0;
    auto fetch = [&](int v) {
        if constexpr (SHARED) {
            np = a.board_np, first = v * a.board_np;
        } else {
            first = off[v], np = off[v + 1] - first;
        }
        if (lane < np) {
            mpt = img[first + lane];
            if constexpr (!SHARED) {
                const float* ob = a.obj + 3 * (size_t)(first + lane);
                bX = ob[0], bY = ob[1], bZ = ob[2];
            }
        }
        const unsigned* qs = reinterpret_cast<const unsigned*>(pc + kPoseConstStride * (size_t)v);
        qword = qs[lane & 31];
    };
    if constexpr (SHARED) {
        if (lane < a.board_np) bX = lds12[3 * lane], bY = lds12[3 * lane + 1], bZ = lds12[3 * lane + 2];
    }
    int tile = blockIdx.x * TPB + tb;
    int v = __builtin_amdgcn_readfirstlane(tile * kTileViews + w);
    bool valid = tile < ntiles && v < a.nviews;  // wave-uniform; the loop holds no workgroup barrier
    if (valid) fetch(v);
    {   // as many dropped stores as a view issues, so that the wait before the first point phase is the loop's own
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        auto none = __builtin_amdgcn_make_buffer_rsrc(blk, 0, 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; r++) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, none, 0x7fff0000 + 64 * r + lane, 0, 0);
    }
    const double2* const lds2r = reinterpret_cast<const double2*>(lds12);
    double2* const lds2 = reinterpret_cast<double2*>(lds12);
    // operand addresses (double2 units) of the first four points; each later group of four is 64 bytes further on.
    // Lanes of the fourth block position read what the first reads (same addresses: a broadcast, no conflict).
    const int c0 = col < 12 ? col : col - 12;
    const int c1 = c0 + 4 < 12 ? c0 + 4 : c0 + 4 - 12;
    const int srcN = ((jt + c0 * RS) >> 1) + rq, srcN1 = ((jt + c1 * RS) >> 1) + rq;
    const int dst = (jt >> 1) + lane;  // RS is even
    int dirty = CH;
    double qd[14];
    auto spread_pose = [&]() {
        unsigned* const slotw = reinterpret_cast<unsigned*>(lds12) + 2 * pslot;
        if (lane < 32) slotw[lane] = qword;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 7; k++) {
            double2 t = lds2r[(pslot >> 1) + k];
            qd[2 * k] = t.x, qd[2 * k + 1] = t.y;
        }
    };
    auto point_phase = [&](int n, float2 m, float X, float Y, float Z) {
        if (lane < n) {
            PoseConst q;
            q.ax[0] = qd[0], q.ax[1] = qd[1], q.ax[2] = qd[2], q.c = qd[3], q.c1 = qd[4], q.s = qd[5];
            q.b1 = qd[6], q.b2 = qd[7], q.b3 = qd[8], q.b4 = qd[9], q.t[0] = qd[10], q.t[1] = qd[11], q.t[2] = qd[12];
            PointOut<M> o;
            double jex[6], jey[6];
            point_jacobian<M>(q, cp, (double)X, (double)Y, (double)Z, (double)m.x, (double)m.y, o, jex, jey);
            static_for<0, T::NI>([&](auto tc_) {
                constexpr int tc = decltype(tc_)::value;
                if constexpr (tc == T::ICX)
                    lds2[dst + ((tc * RS) >> 1)] = make_double2(1., 1.);
                else
                    lds2[dst + ((tc * RS) >> 1)] = make_double2(o.jx[T::intr_of(tc)], o.jy[T::intr_of(tc)]);
            });
#pragma unroll
            for (int b = 0; b < 6; b++) lds2[dst + (((T::CPOSE + b) * RS) >> 1)] = make_double2(jex[b], jey[b]);
            lds2[dst + ((T::CRES * RS) >> 1)] = make_double2(o.ex, o.ey);
        }
        if (n < dirty) {  // rows past the chunk's last point must read as zero (wave-uniform branch)
            if (lane >= n && lane < dirty) {
#pragma unroll
                for (int c = 0; c < NCOL; c++) lds2[dst + ((c * RS) >> 1)] = make_double2(0., 0.);
            }
        }
        dirty = n;
    };
    // Matrix phase: per group of four points two operand reads and four instructions on four accumulators
    // (dependent 4x4x4 MFMAs issue every 46 cycles, alternating chains every 19).  The reads of a group are in
    // flight while the group before multiplies; they may run past the chunk's rows (what they fetch is then never
    // used).  CH is a multiple of 4 points, rows up to it exist and are zero.
    struct Acc {
        double d0, d1, e0, e1;
    };
    auto matrix_phase = [&](int n, Acc& acc, bool from_zero) {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(0);
        const int ngroups = (n + 3) >> 2;  // >= 1
        auto mfma4 = [&](const double2& u, const double2& u1) {
            acc.d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.x, u.x, acc.d0, 0, 0, 0);
            acc.e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.x, u1.x, acc.e0, 0, 0, 0);
            acc.d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.y, u.y, acc.d1, 0, 0, 0);
            acc.e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(u.y, u1.y, acc.e1, 0, 0, 0);
        };
        // unrolled over the sixteen groups a chunk can have, so that every read is the base address plus an immediate
        // offset: no address arithmetic on the vector ALU inside the phase, only scalar compares and branches
        double2 pn = lds2r[srcN], pn1 = lds2r[srcN1];
        if (from_zero) acc = Acc{0., 0., 0., 0.};
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
            if (g >= ngroups) break;
            const double2 qn = lds2r[srcN + 4 * (g + 1)], qn1 = lds2r[srcN1 + 4 * (g + 1)];
            __builtin_amdgcn_sched_barrier(0);
            mfma4(pn, pn1);
            if (g + 1 >= ngroups) break;
            pn = lds2r[srcN + 4 * (g + 2)], pn1 = lds2r[srcN1 + 4 * (g + 2)];
            __builtin_amdgcn_sched_barrier(0);
            mfma4(qn, qn1);
        }
        __builtin_amdgcn_s_setprio(3);
        __builtin_amdgcn_wave_barrier();  // the reads are done before the buffer is overwritten
    };
    auto advance = [&]() {  // the next view's loads go out ahead of the matrix phase
        tile += gridDim.x * TPB;
        v = __builtin_amdgcn_readfirstlane(tile * kTileViews + w);
        valid = tile < ntiles && v < a.nviews;
        if (valid) fetch(v);
    };
    if (valid) spread_pose();
    while (valid) {
        const int tile_cur = tile, vfirst = first, vnp = np;
        Acc acc = {0., 0., 0., 0.};
        if constexpr (!MULTI) {  // the launcher guarantees np <= CH <= 64
            point_phase(vnp, mpt, bX, bY, bZ);
            advance();
            matrix_phase(vnp, acc, true);
            if (valid) spread_pose();
        } else {
            for (int p0 = 0; p0 < vnp; p0 += CH) {
                const int n = vnp - p0 < CH ? vnp - p0 : CH;
                float2 m = mpt;
                float X = bX, Y = bY, Z = bZ;
                if (p0 > 0 && lane < n) {  // later chunks are not fetched ahead
                    m = img[vfirst + p0 + lane];
                    if constexpr (SHARED) {
                        X = lds12[3 * (p0 + lane)], Y = lds12[3 * (p0 + lane) + 1], Z = lds12[3 * (p0 + lane) + 2];
                    } else {
                        const float* ob = a.obj + 3 * (size_t)(vfirst + p0 + lane);
                        X = ob[0], Y = ob[1], Z = ob[2];
                    }
                }
                point_phase(n, m, X, Y, Z);
                const bool last = p0 + CH >= vnp;
                if (last) advance();
                matrix_phase(n, acc, p0 == 0);
                if (last && valid) spread_pose();
            }
        }
        const double res[4] = {split[0] ? acc.d0 : acc.d0 + acc.d1, acc.d1, split[1] ? acc.e0 : acc.e0 + acc.e1, acc.e1};
#pragma unroll
        for (int r = 0; r < 4; r++) usum[r] += res[r];
        // unconditional range-checked stores: lanes whose result is not a per-view quantity carry an offset past the tile
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc(blk + (size_t)tile_cur * NB * kTileViews, 0,
                                                      NB * kTileViews * (int)sizeof(double), 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, res[r]), rsrc, voff[r], 0, 0);
        }
    }
    // The four wavefronts of a tile slot add their shared-block sums through LDS (each into its own, now idle, column
    // buffer) and the first of them writes one row of `part` per tile slot, in a fixed order.
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; r++) ldsd[jt + 64 * r + lane] = usum[r];
    __syncthreads();
    if (w == 0) {
        double* po = part + (size_t)(blockIdx.x * TPB + tb) * NR;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double s = 0;
#pragma unroll
            for (int ww = 0; ww < kWaves; ww++) s += ldsd[jt + ww * WS + 64 * r + lane];
            if (qu[r] >= 0) po[qu[r]] = s;
        }
        // cx against cy: no row has both
        if (lane == 0) po[T::ICX * NC - T::ICX * (T::ICX - 1) / 2 + 1] = 0.;
    }
}

}  // namespace

Fused12Geom fused12_geom(int maxnp, bool shared_board, int board_np, int tiles) {
    Fused12Geom g;
    g.CH = maxnp >= 64 ? 64 : (maxnp + 7) / 8 * 8;
    if (maxnp > 64 && (maxnp + 55) / 56 <= (maxnp + 63) / 64) g.CH = 56;
    // RS = 4 (mod 8) doubles between columns and a multiple of 32 doubles between the wavefronts' buffers: the
    // 16-byte operand reads (eight lanes a cycle: columns c .. c+7, or c+4 .. c+11 mod 12) then fall in disjoint banks
    g.RS = 2 * g.CH + 4;
    const size_t board_f = shared_board ? (size_t)(3 * board_np + 3) / 4 * 4 : 0;
    const size_t ws = ((size_t)12 * g.RS + 31) / 32 * 32;
    g.lds_bytes = (unsigned)(sizeof(float) * board_f + sizeof(double) * tiles * kWaves * ws +
                             sizeof(double) * tiles * kWaves * kPoseConstStride + 128);
    return g;
}

template <class M>
static bool launch_fused12_t(const PassArgs& a, const Fused12Cams& cams, int ncams, int grid, hipStream_t st) {
    const Fused12Geom g = fused12_geom(a.maxnp, a.shared_board != 0, a.board_np, 1);
    const Fused12Geom g3 = fused12_geom(a.maxnp, a.shared_board != 0, a.board_np, 3);
    if (g.lds_bytes > 64 * 1024) return false;
    // three one-tile workgroups per CU if their LDS allows it; otherwise, if it fits, one three-tile workgroup
    bool three = 3 * ((g.lds_bytes + 511u) / 512u * 512u) > kLdsPerCU && g3.lds_bytes <= kLdsPerCU && grid % 3 == 0;
    if (const char* e = std::getenv("CCALIB_FUSED_TPB")) three = std::atoi(e) == 3 && g3.lds_bytes <= kLdsPerCU && grid % 3 == 0;
    auto go = [&](auto shared, auto multi) {
        auto f3 = k_jac_fused12<M, decltype(shared)::value, decltype(multi)::value, 3>;
        if (three && hipFuncSetAttribute(reinterpret_cast<const void*>(f3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)g3.lds_bytes) == hipSuccess) {
            f3<<<dim3(grid / 3, ncams), dim3(3 * kBlock), g3.lds_bytes, st>>>(a, cams, g.RS, g.CH);
        } else {
            if (three) (void)hipGetLastError();
            k_jac_fused12<M, decltype(shared)::value, decltype(multi)::value, 1>
                <<<dim3(grid, ncams), dim3(kBlock), g.lds_bytes, st>>>(a, cams, g.RS, g.CH);
        }
    };
    using Tt = std::true_type;
    using Ff = std::false_type;
    const bool multi = a.maxnp > g.CH;
    if (a.shared_board)
        multi ? go(Tt{}, Tt{}) : go(Tt{}, Ff{});
    else
        multi ? go(Ff{}, Tt{}) : go(Ff{}, Ff{});
    return true;
}

bool launch_jac_fused12(unsigned mask, bool aspect, bool rational, const PassArgs& a, const Fused12Cams& cams, int ncams, int grid,
                        hipStream_t st) {
#ifndef CCALIB_FAST_BUILD
    if (mask == 0x83Eu && aspect && rational) return launch_fused12_t<Model<0x83Eu, true, true>>(a, cams, ncams, grid, st);
#endif
    return false;
}

bool has_jac_fused12(unsigned mask, bool aspect, bool rational) {
#ifndef CCALIB_FAST_BUILD
    return mask == 0x83Eu && aspect && rational;
#else
    return false;
#endif
}

}  // namespace ccalib
