// ubench.hip -- issue-rate microbenchmarks that decide the force kernel's shape.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip
// Each kernel runs ITERS iterations of an asm block of REPS instructions per
// wave and stamps s_memtime around the loop.  Reported: SIMD cycles per
// wave-instruction = median_wave(delta) * waves_per_simd ... (see main).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define DECL8 float a0 = s + 1.f, a1 = s + 2.f, a2 = s + 3.f, a3 = s + 4.f, a4 = s + 5.f, a5 = s + 6.f, a6 = s + 7.f, a7 = s + 8.f; float b = s * 0.5f + 1e-3f, c = s * 0.25f + 1e-3f;
#define SUM8 (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)

#define PROLOG \
    float s = seed[threadIdx.x & 63]; \
    unsigned long long t0, t1;
#define T0 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#define T1 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
#define EPILOG(val) \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (val); \
    if ((threadIdx.x & 63) == 0) dt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;

// ---- 1. 16 independent v_fma_f32 per block (8 regs x2)
__global__ void k_fma(const float* seed, float* out, unsigned long long* dt, int iters) {
    PROLOG DECL8
    T0
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
            "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    T1
    EPILOG(SUM8)
}

// ---- 2. 16 v_pk_fma_f32
__global__ void k_pkfma(const float* seed, float* out, unsigned long long* dt, int iters) {
    PROLOG
    f2 a0 = {s, s + 1}, a1 = {s + 2, s + 3}, a2 = {s + 4, s + 5}, a3 = {s + 6, s + 7}, a4 = {s + 8, s}, a5 = {s, s}, a6 = {s, s + 3}, a7 = {s + 1, s};
    f2 b = {s * 0.5f, s * 0.25f}, c = {s * 0.125f, s};
    T0
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
            "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t"
            "v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
            "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    T1
    f2 t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    EPILOG(t.x + t.y)
}

// ---- 3. 16 v_rsq_f32
__global__ void k_rsq(const float* seed, float* out, unsigned long long* dt, int iters) {
    PROLOG DECL8
    (void)b; (void)c;
    T0
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_rsq_f32 %2, %2\n\tv_rsq_f32 %3, %3\n\t"
            "v_rsq_f32 %4, %4\n\tv_rsq_f32 %5, %5\n\tv_rsq_f32 %6, %6\n\tv_rsq_f32 %7, %7\n\t"
            "v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_rsq_f32 %2, %2\n\tv_rsq_f32 %3, %3\n\t"
            "v_rsq_f32 %4, %4\n\tv_rsq_f32 %5, %5\n\tv_rsq_f32 %6, %6\n\tv_rsq_f32 %7, %7\n\t"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    T1
    EPILOG(SUM8)
}

// ---- 4. pair mix: per "pair" 3 sub + 3 fma + 2 mul + rsq + mul + 3 fma = 13 instrs.
// 2 independent pairs per asm block = 26 instrs.  xj.. come from VGPRs.
__global__ void k_pair(const float* seed, float* out, unsigned long long* dt, int iters) {
    PROLOG
    float xi0 = s, yi0 = s + 1, zi0 = s + 2, xi1 = s + 3, yi1 = s + 4, zi1 = s + 5;
    float ax0 = 0, ay0 = 0, az0 = 0, ax1 = 0, ay1 = 0, az1 = 0;
    float xj = s * 0.3f, yj = s * 0.7f, zj = s * 0.11f, mj = 1.f + s, eps = 1e-4f;
    float dx0, dy0, dz0, d0, dx1, dy1, dz1, d1, e0, e1;
    T0
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile(
                "v_sub_f32 %0, %16, %10\n\tv_sub_f32 %1, %17, %11\n\tv_sub_f32 %2, %18, %12\n\t"
                "v_sub_f32 %4, %16, %13\n\tv_sub_f32 %5, %17, %14\n\tv_sub_f32 %6, %18, %15\n\t"
                "v_fma_f32 %3, %0, %0, %20\n\tv_fma_f32 %7, %4, %4, %20\n\t"
                "v_fma_f32 %3, %1, %1, %3\n\tv_fma_f32 %7, %5, %5, %7\n\t"
                "v_fma_f32 %3, %2, %2, %3\n\tv_fma_f32 %7, %6, %6, %7\n\t"
                "v_mul_f32 %8, %3, %3\n\tv_mul_f32 %9, %7, %7\n\t"
                "v_mul_f32 %8, %8, %3\n\tv_mul_f32 %9, %9, %7\n\t"
                "v_rsq_f32 %8, %8\n\tv_rsq_f32 %9, %9\n\t"
                "v_mul_f32 %8, %8, %19\n\tv_mul_f32 %9, %9, %19\n\t"
                "v_fma_f32 %21, %8, %0, %21\n\tv_fma_f32 %24, %9, %4, %24\n\t"
                "v_fma_f32 %22, %8, %1, %22\n\tv_fma_f32 %25, %9, %5, %25\n\t"
                "v_fma_f32 %23, %8, %2, %23\n\tv_fma_f32 %26, %9, %6, %26\n\t"
                : "=&v"(dx0), "=&v"(dy0), "=&v"(dz0), "=&v"(d0), "=&v"(dx1), "=&v"(dy1), "=&v"(dz1), "=&v"(d1), "=&v"(e0), "=&v"(e1)
                : "v"(xi0), "v"(yi0), "v"(zi0), "v"(xi1), "v"(yi1), "v"(zi1), "v"(xj), "v"(yj), "v"(zj), "v"(mj), "v"(eps),
                  "v"(ax0), "v"(ay0), "v"(az0), "v"(ax1), "v"(ay1), "v"(az1));
            // NOTE: accumulators are inputs only ("v") -> written in asm without the compiler
            // knowing; fine for a timing-only kernel, values are summed below to stay live.
        }
    }
    T1
    EPILOG(ax0 + ay0 + az0 + ax1 + ay1 + az1 + dx0 + dx1 + e0 + e1)
}

// ---- 5. same pair mix but xj/yj/zj/mj as SGPR operands (scalar broadcast)
__global__ void k_pair_sgpr(const float* seed, float* out, unsigned long long* dt, int iters, float xj, float yj, float zj, float mj) {
    PROLOG
    float xi0 = s, yi0 = s + 1, zi0 = s + 2, xi1 = s + 3, yi1 = s + 4, zi1 = s + 5;
    float ax0 = 0, ay0 = 0, az0 = 0, ax1 = 0, ay1 = 0, az1 = 0;
    float eps = 1e-4f;
    float dx0, dy0, dz0, d0, dx1, dy1, dz1, d1, e0, e1;
    T0
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile(
                "v_sub_f32 %0, %16, %10\n\tv_sub_f32 %1, %17, %11\n\tv_sub_f32 %2, %18, %12\n\t"
                "v_sub_f32 %4, %16, %13\n\tv_sub_f32 %5, %17, %14\n\tv_sub_f32 %6, %18, %15\n\t"
                "v_fma_f32 %3, %0, %0, %20\n\tv_fma_f32 %7, %4, %4, %20\n\t"
                "v_fma_f32 %3, %1, %1, %3\n\tv_fma_f32 %7, %5, %5, %7\n\t"
                "v_fma_f32 %3, %2, %2, %3\n\tv_fma_f32 %7, %6, %6, %7\n\t"
                "v_mul_f32 %8, %3, %3\n\tv_mul_f32 %9, %7, %7\n\t"
                "v_mul_f32 %8, %8, %3\n\tv_mul_f32 %9, %9, %7\n\t"
                "v_rsq_f32 %8, %8\n\tv_rsq_f32 %9, %9\n\t"
                "v_mul_f32 %8, %19, %8\n\tv_mul_f32 %9, %19, %9\n\t"
                "v_fma_f32 %21, %8, %0, %21\n\tv_fma_f32 %24, %9, %4, %24\n\t"
                "v_fma_f32 %22, %8, %1, %22\n\tv_fma_f32 %25, %9, %5, %25\n\t"
                "v_fma_f32 %23, %8, %2, %23\n\tv_fma_f32 %26, %9, %6, %26\n\t"
                : "=&v"(dx0), "=&v"(dy0), "=&v"(dz0), "=&v"(d0), "=&v"(dx1), "=&v"(dy1), "=&v"(dz1), "=&v"(d1), "=&v"(e0), "=&v"(e1)
                : "v"(xi0), "v"(yi0), "v"(zi0), "v"(xi1), "v"(yi1), "v"(zi1), "s"(xj), "s"(yj), "s"(zj), "s"(mj), "v"(eps),
                  "v"(ax0), "v"(ay0), "v"(az0), "v"(ax1), "v"(ay1), "v"(az1));
        }
    }
    T1
    EPILOG(ax0 + ay0 + az0 + ax1 + ay1 + az1 + dx0 + dx1 + e0 + e1)
}

// ---- 6. packed pair mix: two i-bodies per lane in register pairs, j broadcast by op_sel.
// per 2 pairs: 3 pk_add(sub) + 3 pk_fma + 2 pk_mul + 2 rsq + 1 pk_mul + 3 pk_fma = 14 instrs
__global__ void k_pair_pk(const float* seed, float* out, unsigned long long* dt, int iters) {
    PROLOG
    f2 xi = {s, s + 3}, yi = {s + 1, s + 4}, zi = {s + 2, s + 5};
    f2 ax = {0, 0}, ay = {0, 0}, az = {0, 0};
    f2 xj = {s * 0.3f, s * 0.3f}, yj = {s * 0.7f, s * 0.7f}, zj = {s * 0.11f, s * 0.11f}, mj = {1.f + s, 1.f + s}, eps = {1e-4f, 1e-4f};
    f2 dx, dy, dz, d2, d6;
    T0
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile(
                "v_pk_add_f32 %0, %8, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_add_f32 %1, %9, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_add_f32 %2, %10, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_fma_f32 %3, %0, %0, %11\n\t"
                "v_pk_fma_f32 %3, %1, %1, %3\n\t"
                "v_pk_fma_f32 %3, %2, %2, %3\n\t"
                "v_pk_mul_f32 %4, %3, %3\n\t"
                "v_pk_mul_f32 %4, %4, %3\n\t"
                : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(d2), "=&v"(d6)
                : "v"(xi), "v"(yi), "v"(zi), "v"(xj), "v"(yj), "v"(zj), "v"(eps));
            d6.x = __builtin_amdgcn_rsqf(d6.x);
            d6.y = __builtin_amdgcn_rsqf(d6.y);
            asm volatile(
                "v_pk_mul_f32 %0, %0, %4\n\t"
                "v_pk_fma_f32 %1, %0, %5, %1\n\t"
                "v_pk_fma_f32 %2, %0, %6, %2\n\t"
                "v_pk_fma_f32 %3, %0, %7, %3\n\t"
                : "+v"(d6), "+v"(ax), "+v"(ay), "+v"(az)
                : "v"(mj), "v"(dx), "v"(dy), "v"(dz));
        }
    }
    T1
    f2 t = ax + ay + az + dx + d6;
    EPILOG(t.x + t.y)
}

// ---- 7. pair mix + one LDS broadcast read (ds_read_b128, all lanes same address) per 2 pairs
__global__ void k_pair_lds(const float* seed, float* out, unsigned long long* dt, int iters) {
    __shared__ f4 tile[256];
    PROLOG
    tile[threadIdx.x] = f4{s * 0.3f, s * 0.7f, s * 0.11f, 1.f + s};
    __syncthreads();
    float xi0 = s, yi0 = s + 1, zi0 = s + 2, xi1 = s + 3, yi1 = s + 4, zi1 = s + 5;
    float ax0 = 0, ay0 = 0, az0 = 0, ax1 = 0, ay1 = 0, az1 = 0;
    float eps = 1e-4f;
    T0
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f4 bj = tile[(i * 4 + u) & 255];
            float dx0 = bj.x - xi0, dy0 = bj.y - yi0, dz0 = bj.z - zi0;
            float dx1 = bj.x - xi1, dy1 = bj.y - yi1, dz1 = bj.z - zi1;
            float d0 = __builtin_fmaf(dz0, dz0, __builtin_fmaf(dy0, dy0, __builtin_fmaf(dx0, dx0, eps)));
            float d1 = __builtin_fmaf(dz1, dz1, __builtin_fmaf(dy1, dy1, __builtin_fmaf(dx1, dx1, eps)));
            float e0 = __builtin_amdgcn_rsqf(d0 * d0 * d0) * bj.w;
            float e1 = __builtin_amdgcn_rsqf(d1 * d1 * d1) * bj.w;
            ax0 = __builtin_fmaf(e0, dx0, ax0); ay0 = __builtin_fmaf(e0, dy0, ay0); az0 = __builtin_fmaf(e0, dz0, az0);
            ax1 = __builtin_fmaf(e1, dx1, ax1); ay1 = __builtin_fmaf(e1, dy1, ay1); az1 = __builtin_fmaf(e1, dz1, az1);
        }
    }
    T1
    EPILOG(ax0 + ay0 + az0 + ax1 + ay1 + az1)
}

// ---- 8. mfma 4x4x1 16B f32 alone, and interleaved with fma
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void k_mfma4(const float* seed, float* out, unsigned long long* dt, int iters) {
    PROLOG
    f4v c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    float a = s, b = s + 1;
    T0
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        }
    }
    T1
    f4v t = c0 + c1 + c2 + c3;
    EPILOG(t.x + t.y + t.z + t.w)
}

// 1 mfma4x4x1 per 10 VALU (the "offload the 3 accumulate fmas" idea): 4 groups per asm
__global__ void k_mfma4_mix(const float* seed, float* out, unsigned long long* dt, int iters) {
    PROLOG DECL8
    f4v c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    T0
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
            "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %10, %8, %9, %10\n\t"
            "v_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
            "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
            "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %11, %8, %9, %11\n\t"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(c0), "v"(c1));
    }
    T1
    EPILOG(SUM8 + c0.x + c1.x)
}

struct Res { double cyc_per_iter_wave; double ms; };

template <typename F>
Res run(F launch, int nblocks, int threads, int iters, float* d_out, unsigned long long* d_dt) {
    int nwaves = nblocks * threads / 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(nblocks, threads, iters / 10 + 1);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(nblocks, threads, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nwaves);
    CK(hipMemcpy(h.data(), d_dt, nwaves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    Res r; r.cyc_per_iter_wave = (double)h[nwaves / 2] / iters; r.ms = ms;
    return r;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    int ncu = p.multiProcessorCount;
    float hseed[64]; for (int i = 0; i < 64; ++i) hseed[i] = 0.5f + 0.01f * i;
    float *d_seed, *d_out; unsigned long long* d_dt;
    CK(hipMalloc(&d_seed, sizeof(hseed))); CK(hipMemcpy(d_seed, hseed, sizeof(hseed), hipMemcpyHostToDevice));
    size_t maxthreads = (size_t)ncu * 8 * 256 * 2;
    CK(hipMalloc(&d_out, maxthreads * 4)); CK(hipMalloc(&d_dt, maxthreads / 64 * 8));

    struct K { const char* name; int instrs; int pairs; int kind; };
    // instrs = wave-instructions per loop iteration, pairs = pair-interactions per lane per iteration
    K ks[] = {
        {"fma x16", 16, 0, 0}, {"pk_fma x16", 16, 0, 1}, {"rsq x16", 16, 0, 2},
        {"pair(vgpr j) 4x2", 4 * 26, 8, 3}, {"pair(sgpr j) 4x2", 4 * 26, 8, 4}, {"pair_pk 4x2", 4 * 14, 8, 5},
        {"pair+ds_read_b128 4x2 (compiler)", 4 * 27, 8, 6}, {"mfma4x4x1 x16", 16, 0, 7}, {"20fma+2mfma4x4x1", 22, 0, 8}};
    printf("%-36s %6s %10s %12s %12s %14s\n", "kernel", "w/SIMD", "ms", "cyc/iter/wv", "cyc/instr/SIMD", "pairs/clk/SIMD");
    for (auto& k : ks) {
        for (int wps : {1, 2, 4, 8}) {
            int nblocks = ncu * wps;  // 256 threads = 4 waves = 1 wave per SIMD per block
            auto launch = [&](int nb, int th, int it) {
                switch (k.kind) {
                    case 0: hipLaunchKernelGGL(k_fma, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                    case 1: hipLaunchKernelGGL(k_pkfma, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                    case 2: hipLaunchKernelGGL(k_rsq, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                    case 3: hipLaunchKernelGGL(k_pair, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                    case 4: hipLaunchKernelGGL(k_pair_sgpr, nb, th, 0, 0, d_seed, d_out, d_dt, it, 0.3f, 0.7f, 0.11f, 1.5f); break;
                    case 5: hipLaunchKernelGGL(k_pair_pk, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                    case 6: hipLaunchKernelGGL(k_pair_lds, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                    case 7: hipLaunchKernelGGL(k_mfma4, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                    case 8: hipLaunchKernelGGL(k_mfma4_mix, nb, th, 0, 0, d_seed, d_out, d_dt, it); break;
                }
            };
            Res r = run(launch, nblocks, 256, iters, d_out, d_dt);
            double cyc_instr_simd = r.cyc_per_iter_wave / k.instrs / wps;
            double pairs = k.pairs ? (k.pairs * 64.0 * wps) / r.cyc_per_iter_wave : 0.0;
            // wall-clock derived chip rate
            double gpairs = k.pairs ? (double)k.pairs * 64 * (nblocks * 4.0) * iters / (r.ms * 1e-3) : 0;
            printf("%-36s %6d %10.3f %12.1f %12.3f %14.3f   wall: %.3e pairs/s  eff.clk %.2f GHz\n", k.name, wps, r.ms,
                   r.cyc_per_iter_wave, cyc_instr_simd, pairs, gpairs,
                   r.cyc_per_iter_wave * iters / (r.ms * 1e-3) * 1e-9);
        }
    }
    return 0;
}
