// ubench7.hip -- does the ORDER of v_rsq_f32 among packed-f32 VALU instructions matter on gfx950?
//
// Round 4, VERDICT item 3: the LDS tile=256 kernel (nb_force_pk<4,1,1>) lost 1.9 % between the round-2 tree and HEAD on the
// same box although its loop executes the same 476 instructions -- only their order differs: the round-2 schedule
// interleaves part of its v_rsq_f32 with v_pk_fma / v_pk_mul, HEAD's issues them as blocks of eight.  This file pins the
// order with one volatile asm statement per instruction and times, per wave and loop trip, 48 v_pk_fma_f32 + 8 v_rsq_f32
// (the ordered-pair mix: 12 packed per 2 transcendental) in several orders, at 1 / 2 / 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench7 ubench7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float nb_f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define F(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b))
#define R(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(t[k]))
#define R2(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(t[k]))
#define F6(k) F(k); F(k + 1); F(k + 2); F(k + 3); F(k + 4); F(k + 5)

// MODE 0: RRRRRRRR F x48      1: (R F x6) x8      2: (RR F x12) x4      3: F x48 only      4: R x8 only
// MODE 5: (RRRR F x24) x2     6: (R F F F R F F F ...) rsq every 4th slot, then the remaining F
template <int MODE>
__global__ __launch_bounds__(256) void k_order(float* out, int iters)
{
    nb_f2 acc[48];
    float t[8];
    const float s = 1.0f + 1e-7f * threadIdx.x;
    const nb_f2 a = nb_f2{s, s}, b = nb_f2{1e-9f, 1e-9f};
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = nb_f2{(float)k, 1.0f};
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = 1.0f + k + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            R(0); R(1); R(2); R(3); R(4); R(5); R(6); R(7);
            F6(0); F6(6); F6(12); F6(18); F6(24); F6(30); F6(36); F6(42);
        } else if constexpr (MODE == 1) {
            R(0); F6(0); R(1); F6(6); R(2); F6(12); R(3); F6(18); R(4); F6(24); R(5); F6(30); R(6); F6(36); R(7); F6(42);
        } else if constexpr (MODE == 2) {
            R(0); R(1); F6(0); F6(6); R(2); R(3); F6(12); F6(18); R(4); R(5); F6(24); F6(30); R(6); R(7); F6(36); F6(42);
        } else if constexpr (MODE == 3) {
            F6(0); F6(6); F6(12); F6(18); F6(24); F6(30); F6(36); F6(42);
        } else if constexpr (MODE == 4) {
            R(0); R(1); R(2); R(3); R(4); R(5); R(6); R(7);
        } else if constexpr (MODE == 5) {
            R(0); R(1); R(2); R(3); F6(0); F6(6); F6(12); F6(18); R(4); R(5); R(6); R(7); F6(24); F6(30); F6(36); F6(42);
        } else {
            R(0); F(0); F(1); F(2); R(1); F(3); F(4); F(5); R(2); F(6); F(7); F(8); R(3); F(9); F(10); F(11);
            R(4); F(12); F(13); F(14); R(5); F(15); F(16); F(17); R(6); F(18); F(19); F(20); R(7); F(21); F(22); F(23);
            F6(24); F6(30); F6(36); F6(42);
        }
    }
    float sum = 0;
#pragma unroll
    for (int k = 0; k < 48; ++k) sum += acc[k].x + acc[k].y;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += t[k];
    if (sum == 12345.678f) out[0] = sum;      // never true: keeps the work alive
}


// The symmetric pass's mix: 16 v_rsq_f32 among 144 other VALU instructions per traveler step (NG = 8).  GAP = number of
// v_pk_fma between the two v_rsq of a pair (0: adjacent, as hipcc schedules nb_force_symw today; 9: evenly spread).
#define FF(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(k) % 48]) : "v"(a), "v"(b))
template <int GAP>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters)
{
    nb_f2 acc[48];
    float t[16];
    const float s = 1.0f + 1e-7f * threadIdx.x;
    const nb_f2 a = nb_f2{s, s}, b = nb_f2{1e-9f, 1e-9f};
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = nb_f2{(float)k, 1.0f};
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = 1.0f + k + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if constexpr (GAP >= 0) R2(2 * p);
#pragma unroll
            for (int f = 0; f < 18; ++f) {
                if (GAP >= 0 && f == GAP) R2(2 * p + 1);
                FF(p * 18 + f);
            }
        }
    }
    float sum = 0;
#pragma unroll
    for (int k = 0; k < 48; ++k) sum += acc[k].x + acc[k].y;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += t[k];
    if (sum == 12345.678f) out[0] = sum;
}

template <int GAP>
double run_mix(float* out, int waves_per_simd, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL(k_mix<GAP>, dim3(grid), dim3(256), 0, 0, out, iters / 8);
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mix<GAP>, dim3(grid), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e6 / iters;
}

template <int MODE>
double run(float* out, int waves_per_simd, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * waves_per_simd;          // 256 CUs x one 4-wave workgroup per requested wave per SIMD
    hipLaunchKernelGGL(k_order<MODE>, dim3(grid), dim3(256), 0, 0, out, iters / 8);      // warm-up
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_order<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e6 / iters;      // ns per loop trip (all co-resident waves of a SIMD advance one trip each)
}

int main()
{
    float* out;
    CK(hipMalloc(&out, 64));
    const int iters = 200000;
    const char* names[7] = {"R x8 | F x48", "(R F6) x8", "(RR F12) x4", "F x48 only", "R x8 only", "(RRRR F24) x2", "(R FFF) x8 | F24"};
    for (int w : {1, 2, 4}) {
        double t[7];
        t[0] = run<0>(out, w, iters); t[1] = run<1>(out, w, iters); t[2] = run<2>(out, w, iters); t[3] = run<3>(out, w, iters);
        t[4] = run<4>(out, w, iters); t[5] = run<5>(out, w, iters); t[6] = run<6>(out, w, iters);
        for (int m = 0; m < 7; ++m)
            printf("waves/SIMD %d  %-18s %8.2f ns per trip per SIMD-resident set  (%.1f ns per wave-trip; x%.3f of the blocked order)\n", w, names[m], t[m],
                   t[m] / w, t[m] / t[0]);
    }
    const int it2 = 80000;
    for (int w : {1, 2, 4}) {
        const double base = run_mix<-1>(out, w, it2);
        const double g[6] = {run_mix<0>(out, w, it2), run_mix<1>(out, w, it2), run_mix<2>(out, w, it2), run_mix<3>(out, w, it2), run_mix<4>(out, w, it2),
                             run_mix<9>(out, w, it2)};
        const int gaps[6] = {0, 1, 2, 3, 4, 9};
        printf("mix 144 F + 16 R, waves/SIMD %d: F only %.1f ns per wave-trip;", w, base / w);
        for (int k = 0; k < 6; ++k) printf("  gap %d: %.1f (x%.3f)", gaps[k], g[k] / w, g[k] / g[0]);
        printf("\n");
    }
    return 0;
}
