// ubench2.hip -- does v_mfma_f32_4x4x1_16b_f32 co-execute with packed VALU work?
// Four kernels with the real force-loop instruction mix, 4 waves/SIMD, ~tens of ms each:
//   A: 4 groups x (8 v_pk + 2 v_rsq)                      (VALU part of the MFMA design)
//   B: A + 8 v_mfma_f32_4x4x1_16b_f32 (2 per group)       (the MFMA design)
//   C: 4 groups x (12 v_pk + 2 v_rsq)                     (today's all-VALU body)
//   D: 8 v_mfma only
// Reports wall ms (min of 3), in-kernel clock (s_memtime / s_memrealtime) and SIMD cycles per iteration.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* seed, float* out, unsigned long long* stamps, int iters)
{
    const float s = seed[threadIdx.x & 63];
    f2 xi[4], yi[4], zi[4], ax[4], ay[4], az[4];
    f4 acc[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        xi[g] = f2{s + g, s - g}; yi[g] = f2{s * 2 + g, s * 3 - g}; zi[g] = f2{s * 5 + g, s * 7 - g};
        ax[g] = ay[g] = az[g] = f2{0, 0};
        acc[2 * g] = acc[2 * g + 1] = f4{0, 0, 0, 0};
    }
    const f2 e2 = f2{1e-4f, 1e-4f};
    float bx = s * 0.3f, by = s * 0.7f, bz = s * 0.11f, bm = 1.f + s, bq = s * 0.01f;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        bx += 1e-3f; by -= 1e-3f;   // keep the loop body from being hoisted
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f2 dx = f2{bx, bx} - xi[g], dy = f2{by, by} - yi[g], dz = f2{bz, bz} - zi[g];
            const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
            const f2 d6 = d2 * d2 * d2;
            f2 r = f2{__builtin_amdgcn_rsqf(d6.x), __builtin_amdgcn_rsqf(d6.y)};
            if (MODE == 2) {  // C: today's body
                const f2 sm = f2{bm, bm} * r;
                ax[g] = __builtin_elementwise_fma(sm, dx, ax[g]);
                ay[g] = __builtin_elementwise_fma(sm, dy, ay[g]);
                az[g] = __builtin_elementwise_fma(sm, dz, az[g]);
            } else if (MODE == 1) {  // B: MFMA accumulate, A = r, B = per-lane component of the j body
                acc[2 * g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r.x, bq, acc[2 * g], 0, 0, 0);
                acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(r.y, bq, acc[2 * g + 1], 0, 0, 0);
            } else if (MODE == 0) {  // A: keep r alive without extra VALU
                asm volatile("" ::"v"(r));
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int g = 0; g < 8; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(bx, bq, acc[g], 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float v = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) v += ax[g].x + ax[g].y + ay[g].x + ay[g].y + az[g].x + az[g].y + acc[2 * g].x + acc[2 * g].y + acc[2 * g + 1].z + acc[2 * g + 1].w;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = v;
    if ((threadIdx.x & 63) == 0) { stamps[2 * (gid >> 6)] = t1 - t0; stamps[2 * (gid >> 6) + 1] = r1 - r0; }
}

template <int MODE> void run(const char* name, int nb, int iters, float* d_seed, float* d_out, unsigned long long* d_st, int valu_cycles_nominal)
{
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, nb, 256, 0, 0, d_seed, d_out, d_st, iters);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    int nw = nb * 4;
    std::vector<unsigned long long> h(2 * nw);
    CK(hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost));
    std::vector<double> clk(nw), cyc(nw);
    for (int w = 0; w < nw; ++w) { clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 0.1; cyc[w] = (double)h[2 * w]; }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    double waves_per_simd = nw / 1024.0;
    // SIMD cycles per iteration = wall * clock / iters / waves_per_simd
    double simd_cyc = best * 1e-3 * clk[nw / 2] * 1e9 / iters / waves_per_simd;
    printf("%-28s wall %8.3f ms  clk(med) %.3f GHz  wave cyc/iter(med) %8.1f  SIMD cyc/iter/wave %7.1f  (nominal VALU %d)  pairs/s %.3e\n",
           name, best, clk[nw / 2], cyc[nw / 2] / iters, simd_cyc, valu_cycles_nominal, 8.0 * 64 * nw * (double)iters / (best * 1e-3));
}

int main(int argc, char** argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 100000;
    int bpc = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int nb = p.multiProcessorCount * bpc;
    float hs[64]; for (int i = 0; i < 64; ++i) hs[i] = 0.5f + 0.01f * i;
    float *d_seed, *d_out; unsigned long long* d_st;
    CK(hipMalloc(&d_seed, sizeof hs)); CK(hipMemcpy(d_seed, hs, sizeof hs, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, nb * 256 * 4)); CK(hipMalloc(&d_st, nb * 4 * 16));
    printf("CUs %d, blocks/CU %d (waves/SIMD), iters %d; per iteration 8 pairs per lane\n", p.multiProcessorCount, bpc, iters);
    // warm up the clocks
    run<2>("warmup (C)", nb, iters, d_seed, d_out, d_st, 256);
    for (int round = 0; round < 2; ++round) {
        run<0>("A: 8pk+2rsq  x4", nb, iters, d_seed, d_out, d_st, 192);
        run<1>("B: A + 8 mfma4x4x1", nb, iters, d_seed, d_out, d_st, 192);
        run<2>("C: 12pk+2rsq x4", nb, iters, d_seed, d_out, d_st, 256);
        run<3>("D: 8 mfma4x4x1 only", nb, iters, d_seed, d_out, d_st, 0);
    }
    return 0;
}
