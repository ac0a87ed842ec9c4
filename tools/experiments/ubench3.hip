// ubench3.hip -- round-2 issue-cost measurements behind two design questions:
//
//  (1) Can the idle matrix pipe carry the force accumulate?  Round 1 only tried
//      v_mfma_f32_4x4x1_16b_f32 (serialises with VALU).  MI355X_MICROARCH.md (cycle constants)
//      says a LARGE MFMA holds the SIMD's vector issue for only 8 of its cycles, so here the
//      real packed force body (8 v_pk + 2 v_rsq per two pairs: the body WITHOUT the
//      4 accumulate ops) runs with K x v_mfma_f32_16x16x4_f32 or v_mfma_f32_32x32x2_f32 per
//      iteration in the same wave, against today's all-VALU body (12 v_pk + 2 v_rsq).
//      The accumulate of one iteration (8 pairs per lane = 512 pairs per wave) needs
//      512/64 = 8 MFMAs of either form (16x16x4: 16 i x 4 j; 32x32x2: 32 i x 2 j, 4 useful
//      output columns of 16/32).
//  (2) What do the f64 force-loop instructions cost?  v_fma_f64 / v_mul_f64 / v_add_f64 /
//      v_rsq_f64 / v_cvt_f32_f64 / v_cvt_f64_f32 / v_rsq_f32, one wave stream and 2-4 waves
//      per SIMD; plus three whole f64 pair bodies (today's, first-order-expanded on d2, and
//      f32-seeded).
//
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench3 ubench3.hip ; run: ./ubench3 [iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define STAMP0 asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#define STAMP1 asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");

// ------------------------------------------------------------------------------------------
// (1) packed force body + K large MFMAs per iteration
//   MODE 0: 4 groups x (8 v_pk + 2 v_rsq)                          VALU part of an MFMA design
//   MODE 1: MODE 0 + KM x v_mfma_f32_16x16x4_f32 (32 cyc each)
//   MODE 2: MODE 0 + KM x v_mfma_f32_32x32x2_f32 (64 cyc each)
//   MODE 3: 4 groups x (12 v_pk + 2 v_rsq)                         today's all-VALU body
//   MODE 4: KM x 16x16x4 only;  MODE 5: KM x 32x32x2 only
template <int MODE, int KM>
__global__ __launch_bounds__(256) void kmf(const float* seed, float* out, unsigned long long* stamps, int iters)
{
    const float s = seed[threadIdx.x & 63];
    f2 xi[4], yi[4], zi[4], ax[4], ay[4], az[4];
    f4 acc4[8];
    f16v acc16[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        xi[g] = f2{s + g, s - g}; yi[g] = f2{s * 2 + g, s * 3 - g}; zi[g] = f2{s * 5 + g, s * 7 - g};
        ax[g] = ay[g] = az[g] = f2{0, 0};
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) acc4[g] = f4{0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc16[g][q] = 0.f;
    const f2 e2 = f2{1e-4f, 1e-4f};
    float bx = s * 0.3f, by = s * 0.7f, bz = s * 0.11f, bm = 1.f + s, bq = s * 0.01f;
    unsigned long long t0, t1, r0, r1;
    STAMP0
    for (int it = 0; it < iters; ++it) {
        bx += 1e-3f; by -= 1e-3f; bz += 2e-3f;
        f2 rr[4];
        if (MODE <= 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f2 dx = f2{bx, bx} - xi[g], dy = f2{by, by} - yi[g], dz = f2{bz, bz} - zi[g];
                const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
                const f2 d6 = d2 * d2 * d2;
                f2 r = f2{__builtin_amdgcn_rsqf(d6.x), __builtin_amdgcn_rsqf(d6.y)};
                rr[g] = r;
                if (MODE == 3) {
                    const f2 sm = f2{bm, bm} * r;
                    ax[g] = __builtin_elementwise_fma(sm, dx, ax[g]);
                    ay[g] = __builtin_elementwise_fma(sm, dy, ay[g]);
                    az[g] = __builtin_elementwise_fma(sm, dz, az[g]);
                } else if (MODE == 0) {
                    asm volatile("" ::"v"(r));
                }
                // MFMAs interleaved with the groups: KM/4 after each group (rounded), A operand = that group's r
                if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < (KM + 3 - g) / 4; ++q) {
                        const int a = (g * 2 + q) & 7;
                        acc4[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(q & 1 ? r.y : r.x, bq, acc4[a], 0, 0, 0);
                    }
                } else if (MODE == 2) {
#pragma unroll
                    for (int q = 0; q < (KM + 3 - g) / 4; ++q) {
                        const int a = (g + q) & 1;
                        acc16[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(q & 1 ? r.y : r.x, bq, acc16[a], 0, 0, 0);
                    }
                }
            }
        }
        if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < KM; ++q) acc4[q & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(bx, bq, acc4[q & 7], 0, 0, 0);
        } else if (MODE == 5) {
#pragma unroll
            for (int q = 0; q < KM; ++q) acc16[q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bx, bq, acc16[q & 1], 0, 0, 0);
        }
        (void)rr;
    }
    STAMP1
    float v = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) v += ax[g].x + ax[g].y + ay[g].x + ay[g].y + az[g].x + az[g].y;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += acc4[g].x + acc4[g].y + acc4[g].z + acc4[g].w;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 16; ++q) v += acc16[g][q];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = v;
    if ((threadIdx.x & 63) == 0) { stamps[2 * (gid >> 6)] = t1 - t0; stamps[2 * (gid >> 6) + 1] = r1 - r0; }
}

// ------------------------------------------------------------------------------------------
// (1b) the same question across waves: a 512-thread workgroup puts two waves on every SIMD.
//   MIX 0: both run today's all-VALU body;  MIX 1: both run 8 x v_mfma_f32_16x16x4_f32 per iteration;
//   MIX 2: waves 0-3 (one per SIMD) run the VALU body, waves 4-7 the MFMAs.
// Separate pipes would finish MIX 2 in ~max(MIX 0, MIX 1)/2; one shared datapath in ~their mean.
template <int MIX>
__global__ __launch_bounds__(512) void kco(const float* seed, float* out, unsigned long long* stamps, int iters)
{
    const float s = seed[threadIdx.x & 63];
    const bool mfma_role = MIX == 1 || (MIX == 2 && (threadIdx.x >> 6) >= 4);
    f2 xi[4], yi[4], zi[4], ax[4], ay[4], az[4];
    f4 acc4[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        xi[g] = f2{s + g, s - g}; yi[g] = f2{s * 2 + g, s * 3 - g}; zi[g] = f2{s * 5 + g, s * 7 - g};
        ax[g] = ay[g] = az[g] = f2{0, 0};
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) acc4[g] = f4{0, 0, 0, 0};
    const f2 e2 = f2{1e-4f, 1e-4f};
    float bx = s * 0.3f, by = s * 0.7f, bz = s * 0.11f, bm = 1.f + s, bq = s * 0.01f;
    unsigned long long t0, t1, r0, r1;
    STAMP0
    if (mfma_role) {      // wave-uniform branch
        for (int it = 0; it < iters; ++it) {
            bx += 1e-3f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(bx, bq, acc4[q], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            bx += 1e-3f; by -= 1e-3f; bz += 2e-3f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f2 dx = f2{bx, bx} - xi[g], dy = f2{by, by} - yi[g], dz = f2{bz, bz} - zi[g];
                const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
                const f2 d6 = d2 * d2 * d2;
                const f2 r = f2{__builtin_amdgcn_rsqf(d6.x), __builtin_amdgcn_rsqf(d6.y)};
                const f2 sm = f2{bm, bm} * r;
                ax[g] = __builtin_elementwise_fma(sm, dx, ax[g]);
                ay[g] = __builtin_elementwise_fma(sm, dy, ay[g]);
                az[g] = __builtin_elementwise_fma(sm, dz, az[g]);
            }
        }
    }
    STAMP1
    float v = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) v += ax[g].x + ax[g].y + ay[g].x + ay[g].y + az[g].x + az[g].y;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += acc4[g].x + acc4[g].y + acc4[g].z + acc4[g].w;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = v;
    if ((threadIdx.x & 63) == 0) { stamps[2 * (gid >> 6)] = t1 - t0; stamps[2 * (gid >> 6) + 1] = r1 - r0; }
}

// ------------------------------------------------------------------------------------------
// (2a) single-instruction issue cost: 16 independent copies of one instruction per iteration
#define OPK(NAME, TY, INIT, ASM16, CONSTRAINTS)                                                            \
    __global__ __launch_bounds__(256) void NAME(const float* seed, float* out, unsigned long long* stamps, int iters) \
    {                                                                                                       \
        const float s = seed[threadIdx.x & 63];                                                             \
        TY a0 = INIT(1), a1 = INIT(2), a2 = INIT(3), a3 = INIT(4), a4 = INIT(5), a5 = INIT(6), a6 = INIT(7), a7 = INIT(8); \
        TY b = INIT(0.5), c = INIT(0.25);                                                                   \
        unsigned long long t0, t1, r0, r1;                                                                  \
        STAMP0                                                                                              \
        for (int i = 0; i < iters; ++i) { asm volatile(ASM16 CONSTRAINTS); }                                \
        STAMP1                                                                                              \
        const int gid = blockIdx.x * blockDim.x + threadIdx.x;                                              \
        out[gid] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + c);                                  \
        if ((threadIdx.x & 63) == 0) { stamps[2 * (gid >> 6)] = t1 - t0; stamps[2 * (gid >> 6) + 1] = r1 - r0; } \
    }
#define R8(op, tail) op " %0, %0" tail "\n\t" op " %1, %1" tail "\n\t" op " %2, %2" tail "\n\t" op " %3, %3" tail "\n\t" \
                     op " %4, %4" tail "\n\t" op " %5, %5" tail "\n\t" op " %6, %6" tail "\n\t" op " %7, %7" tail "\n\t"
#define C8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
#define DINIT(k) ((double)s + (k))
#define FINIT(k) (s + (float)(k))
OPK(k_fma64, double, DINIT, R8("v_fma_f64", ", %8, %9") R8("v_fma_f64", ", %8, %9"), C8)
OPK(k_mul64, double, DINIT, R8("v_mul_f64", ", %8") R8("v_mul_f64", ", %8"), C8)
OPK(k_add64, double, DINIT, R8("v_add_f64", ", %8") R8("v_add_f64", ", %8"), C8)
OPK(k_rsq64, double, DINIT, R8("v_rsq_f64", "") R8("v_rsq_f64", ""), C8)
OPK(k_rcp64, double, DINIT, R8("v_rcp_f64", "") R8("v_rcp_f64", ""), C8)
OPK(k_rsq32, float, FINIT, R8("v_rsq_f32", "") R8("v_rsq_f32", ""), C8)
OPK(k_fma32, float, FINIT, R8("v_fma_f32", ", %8, %9") R8("v_fma_f32", ", %8, %9"), C8)

// conversions: destination and source have different widths, so separate register sets
__global__ __launch_bounds__(256) void k_cvt(const float* seed, float* out, unsigned long long* stamps, int iters, int dummy)
{
    const float s = seed[threadIdx.x & 63];
    double d0 = s + 1., d1 = s + 2., d2 = s + 3., d3 = s + 4.;
    float f0 = s, f1 = s + 1.f, f2_ = s + 2.f, f3 = s + 3.f;
    unsigned long long t0, t1, r0, r1;
    STAMP0
    for (int i = 0; i < iters; ++i) {
        // 8 x (f64 -> f32) and 8 x (f32 -> f64), all independent of each other within the block
        asm volatile(
            "v_cvt_f32_f64 %4, %0\n\tv_cvt_f32_f64 %5, %1\n\tv_cvt_f32_f64 %6, %2\n\tv_cvt_f32_f64 %7, %3\n\t"
            "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
            "v_cvt_f32_f64 %4, %0\n\tv_cvt_f32_f64 %5, %1\n\tv_cvt_f32_f64 %6, %2\n\tv_cvt_f32_f64 %7, %3\n\t"
            "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(f0), "+v"(f1), "+v"(f2_), "+v"(f3));
    }
    STAMP1
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = (float)(d0 + d1 + d2 + d3) + f0 + f1 + f2_ + f3 + dummy;
    if ((threadIdx.x & 63) == 0) { stamps[2 * (gid >> 6)] = t1 - t0; stamps[2 * (gid >> 6) + 1] = r1 - r0; }
}

// ------------------------------------------------------------------------------------------
// (2b) whole f64 pair bodies, IPL i-bodies per lane, one j per iteration
//   BODY 0: today's: d6 = d2^3, y0 = v_rsq_f64(d6), one Newton step (4 ops), s = gm*y       (16 DP + rsq64)
//   BODY 1: y0 = v_rsq_f64(d2), e = 1 - d2 y0^2, s = gm y0^3 (1 + 1.5 e)                     (15 DP + rsq64)
//   BODY 2: as 1 with y0 = (double)v_rsq_f32((float)d2)                                     (15 DP + 2 cvt + rsq32)
template <int BODY, int IPL>
__global__ __launch_bounds__(256) void kf64(const float* seed, float* out, unsigned long long* stamps, int iters)
{
    const double s = seed[threadIdx.x & 63];
    double xi[IPL], yi[IPL], zi[IPL], ax[IPL], ay[IPL], az[IPL];
#pragma unroll
    for (int k = 0; k < IPL; ++k) { xi[k] = s + k; yi[k] = 2 * s - k; zi[k] = 3 * s + 0.5 * k; ax[k] = ay[k] = az[k] = 0; }
    double bx = s * 0.3, by = s * 0.7, bz = s * 0.11, bm = 1. + s;
    const double eps2 = 1e-4;
    unsigned long long t0, t1, r0, r1;
    STAMP0
    for (int it = 0; it < iters; ++it) {
        bx += 1e-3; by -= 1e-3; bz += 2e-3;
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const double dx = bx - xi[k], dy = by - yi[k], dz = bz - zi[k];
            const double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
            double sc;
            if (BODY == 0) {
                const double d6 = d2 * d2 * d2;
                const double y = __builtin_amdgcn_rsq(d6);
                const double e = __builtin_fma(-d6 * y, y, 1.0);
                sc = bm * __builtin_fma(y * e, 0.5, y);
            } else {
                double y;
                if (BODY == 1) y = __builtin_amdgcn_rsq(d2);
                else y = (double)__builtin_amdgcn_rsqf((float)d2);
                const double y2 = y * y;
                const double e = __builtin_fma(-d2, y2, 1.0);
                const double p3 = (bm * y) * y2;
                sc = __builtin_fma(p3 * e, 1.5, p3);
            }
            ax[k] = __builtin_fma(sc, dx, ax[k]);
            ay[k] = __builtin_fma(sc, dy, ay[k]);
            az[k] = __builtin_fma(sc, dz, az[k]);
        }
    }
    STAMP1
    double v = 0;
#pragma unroll
    for (int k = 0; k < IPL; ++k) v += ax[k] + ay[k] + az[k];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = (float)v;
    if ((threadIdx.x & 63) == 0) { stamps[2 * (gid >> 6)] = t1 - t0; stamps[2 * (gid >> 6) + 1] = r1 - r0; }
}

struct Res { double wall_ms, clk, wave_cyc_iter, simd_cyc_iter; };

template <typename F> Res timeit(F launch, int nb, int iters, unsigned long long* d_st)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    const int nw = nb * 4;
    std::vector<unsigned long long> h(2 * nw);
    CK(hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost));
    std::vector<double> clk(nw), cyc(nw);
    for (int w = 0; w < nw; ++w) { clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 0.1; cyc[w] = (double)h[2 * w]; }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    Res r;
    r.wall_ms = best; r.clk = clk[nw / 2]; r.wave_cyc_iter = cyc[nw / 2] / iters;
    r.simd_cyc_iter = best * 1e-3 * r.clk * 1e9 / iters / (nw / 1024.0);
    return r;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 50000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    float hs[64]; for (int i = 0; i < 64; ++i) hs[i] = 0.5f + 0.01f * i;
    float *d_seed, *d_out; unsigned long long* d_st;
    const int maxnb = ncu * 8;
    CK(hipMalloc(&d_seed, sizeof hs)); CK(hipMemcpy(d_seed, hs, sizeof hs, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)maxnb * 256 * 4)); CK(hipMalloc(&d_st, (size_t)maxnb * 4 * 16));
    printf("CUs %d, iters %d\n", ncu, iters);

#define RUN(label, kern, bpc, unit_per_iter, ...)                                                                      \
    do {                                                                                                               \
        const int nb_ = ncu * (bpc);                                                                                   \
        Res r_ = timeit([&] { hipLaunchKernelGGL(kern, nb_, 256, 0, 0, d_seed, d_out, d_st, iters, ##__VA_ARGS__); }, nb_, iters, d_st); \
        printf("%-44s w/SIMD %d  wall %8.3f ms  clk %.3f GHz  wave cyc/iter %8.1f  SIMD cyc/iter/wave %7.1f  -> %6.2f SIMD cyc per %s\n", \
               label, bpc, r_.wall_ms, r_.clk, r_.wave_cyc_iter, r_.simd_cyc_iter, r_.simd_cyc_iter / (unit_per_iter), #unit_per_iter); \
    } while (0)

    // warm the clocks
    RUN("warmup", (kmf<3, 0>), 4, 1);
    printf("\n== (1) packed force body + large f32 MFMAs in the same wave (per iteration: 8 pairs per lane = 512 pairs per wave;\n"
           "       today's body nominal 256 SIMD cycles, body without accumulate 192; the accumulate needs 8 MFMAs) ==\n");
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        RUN("C  : 4x(12pk+2rsq)  [today]", (kmf<3, 0>), bpc, 1);
        RUN("A  : 4x(8pk+2rsq)", (kmf<0, 0>), bpc, 1);
        RUN("A + 2 mfma16x16x4", (kmf<1, 2>), bpc, 1);
        RUN("A + 4 mfma16x16x4", (kmf<1, 4>), bpc, 1);
        RUN("A + 8 mfma16x16x4  [full accumulate]", (kmf<1, 8>), bpc, 1);
        RUN("A + 2 mfma32x32x2", (kmf<2, 2>), bpc, 1);
        RUN("A + 4 mfma32x32x2", (kmf<2, 4>), bpc, 1);
        RUN("A + 8 mfma32x32x2  [full accumulate]", (kmf<2, 8>), bpc, 1);
        RUN("8 mfma16x16x4 only", (kmf<4, 8>), bpc, 1);
        RUN("8 mfma32x32x2 only", (kmf<5, 8>), bpc, 1);
    }
    printf("\n== (1b) VALU-only waves beside MFMA-only waves on the same SIMD (512-thread workgroups, one per CU: 2 waves per SIMD;\n"
           "        wall ms is what matters: separate pipes -> MIX2 ~ max(MIX0, MIX1) / 2, one datapath -> ~ (MIX0 + MIX1) / 2) ==\n");
    for (int rep = 0; rep < 2; ++rep) {
        const int nb_ = ncu;
        auto wall = [&](auto kern) {
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, nb_, 512, 0, 0, d_seed, d_out, d_st, iters);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            return best;
        };
        const float m0 = wall(kco<0>), m1 = wall(kco<1>), m2 = wall(kco<2>);
        printf("MIX0 VALU|VALU %8.3f ms   MIX1 MFMA|MFMA %8.3f ms   MIX2 VALU|MFMA %8.3f ms   (max/2 = %.3f, mean = %.3f)\n", m0, m1, m2,
               std::max(m0, m1) / 2, (m0 + m1) / 2);
    }
    printf("\n== (2a) instruction issue cost (16 independent instructions per iteration) ==\n");
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        RUN("v_fma_f64", k_fma64, bpc, 16);
        RUN("v_mul_f64", k_mul64, bpc, 16);
        RUN("v_add_f64", k_add64, bpc, 16);
        RUN("v_rsq_f64", k_rsq64, bpc, 16);
        RUN("v_rcp_f64", k_rcp64, bpc, 16);
        RUN("v_rsq_f32", k_rsq32, bpc, 16);
        RUN("v_fma_f32", k_fma32, bpc, 16);
        RUN("v_cvt_f32_f64 + v_cvt_f64_f32 (8+8)", k_cvt, bpc, 16, 0);
    }
    printf("\n== (2b) f64 pair bodies (per iteration: IPL pairs per lane) ==\n");
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        RUN("f64 body 0 (d6, rsq64, Newton) IPL2", (kf64<0, 2>), bpc, 2);
        RUN("f64 body 1 (rsq64(d2), 1st-order) IPL2", (kf64<1, 2>), bpc, 2);
        RUN("f64 body 2 (rsq32 seed, 1st-order) IPL2", (kf64<2, 2>), bpc, 2);
        RUN("f64 body 0 IPL4", (kf64<0, 4>), bpc, 4);
        RUN("f64 body 1 IPL4", (kf64<1, 4>), bpc, 4);
        RUN("f64 body 2 IPL4", (kf64<2, 4>), bpc, 4);
    }
    return 0;
}
