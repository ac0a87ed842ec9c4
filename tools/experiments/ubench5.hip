// ubench5.hip -- the j-packed SGPR step (nb_step_jpk<WS>) on its own: event-timed us per step for
// N = 2k .. 40k and WS = 4 / 8 / 16, each checked against an fp64 direct sum of the first step
// (max relative acceleration error over all bodies).  Product kernels, no stamps.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../nbody3d-webgpu_amd/csrc -o ubench5 ubench5.hip
#include "nb_kernels.hip.h"

namespace nb {
// Experiment kept out of the product header (measured slower than nb_step_jpk at every size, profiles/r02/ubench5_*):
// ---- j-packed step with a wave-private LDS ring (nb_step_jring) ---------------------------------
// Same decomposition and arithmetic as nb_step_jpk, but the j-pairs reach the wave through LDS-DMA
// instead of the scalar cache: every wave owns a ring of RING 1-KiB blocks (32 pairs each) in LDS,
// filled by global_load_lds_dwordx4 (one wave instruction per block, no VGPR staging) RING-1 blocks
// ahead of the one being read, and read back with wave-uniform ds_read_b128 (LDS broadcast).
//   * vmcnt counts the DMA loads in order, so the wave waits for exactly the block it needs with
//     two younger ones in flight -- a scalar stream has ONE request in flight (SMEM returns out of
//     order: lgkmcnt(0) is its only wait) and measured 1,200 cycles per 4-pair request at any size;
//   * the ring is private to the wave: no s_barrier and no hand-over between waves in the loop (the
//     shared-tile kernels above spend up to 45 % of a tile period there at low occupancy), so ONE wave
//     per SIMD already runs the loop at its issue rate.
// LDS: WS x RING KiB per workgroup.  After the loop the ring memory carries the cross-wave reduction.
template <int WS>
__global__ __launch_bounds__(64 * WS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void nb_step_jring(const float4* __restrict__ bodies_in, const float4* __restrict__ pairs_in, float4* __restrict__ bodies_out,
                   float4* __restrict__ pairs_out, float4* __restrict__ vel, float4* __restrict__ acc, float4* partial,
                   uint32_t* ticket, uint32_t n, uint32_t units_per_wave, uint32_t poison, float G, float eps2, float dt)
{
    static_assert(WS >= 1 && WS <= 16, "a workgroup has at most 16 waves");
    constexpr int RING = 4;                            // blocks per wave; a block = 8 units = 32 pairs = 1 KiB
    __shared__ nb_v4f lds[WS * RING * 64];             // the ONLY LDS object of the kernel
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i = blockIdx.x * 64 + lane;
    const bool valid = i < n;
    const uint32_t ic = valid ? i : n - 1;             // clamped, branch-free (never stored)

    // this wave's share: an even number of 4-pair units (the array ends with 8 spare zero units)
    const uint32_t units = (((n + 1) / 2 + 3) / 4 + 1) & ~1u;
    uint32_t u0 = (blockIdx.y * WS + wv) * units_per_wave, u1 = u0 + units_per_wave;
    if (u0 > units) u0 = units;
    if (u1 > units) u1 = units;
    const uint32_t nu = u1 - u0, nblk = (nu + 7) / 8;

    nb_v4f* const ring = lds + wv * (RING * 64);
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) nb_v4f*)ring;   // LDS byte address (wave-uniform)
    const float4* const gsrc = pairs_in + (size_t)u0 * 8 + lane;       // this lane's 16 B of block 0
    // block b of the share -> ring slot b % RING.  Blocks past the share re-read its last block (into a slot that
    // is not being read): the number of loads in flight stays the same to the end, so one vmcnt value serves.
    auto dma = [&](uint32_t b) {
        const uint32_t bc = b < nblk ? b : nblk - 1;
        const float4* src = gsrc + (size_t)bc * 64;
        const uint32_t dst = ring_lds + (b % RING) * 1024u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    if (nblk) { dma(0); dma(1); dma(2); }

    const float4 bi = ld4(bodies_in + ic);
    nb_v4f v0 = nb_v4f{0, 0, 0, 0}, a0 = nb_v4f{0, 0, 0, 0};
    if (wv == 0) {                                     // in flight under the loop; first use after it
        v0 = *reinterpret_cast<const nb_v4f*>(vel + ic);
        a0 = *reinterpret_cast<const nb_v4f*>(acc + ic);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(a0) : : "memory");   // one round trip for everything above
    const nb_f2 xi = nb_f2{bi.x, bi.x}, yi = nb_f2{bi.y, bi.y}, zi = nb_f2{bi.z, bi.z};
    const nb_f2 e2 = nb_f2{eps2, eps2};
    nb_f2 ax = nb_f2{0, 0}, ay = nb_f2{0, 0}, az = nb_f2{0, 0};

    auto eval = [&](const nb_v4f (&q)[8]) {          // one unit: pairs c = 0..3 as (x0,x1,y0,y1) (z0,z1,Gm0,Gm1)
        nb_f2 dx[4], dy[4], dz[4], d2[4], r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) dx[c] = nb_f2{q[2 * c].x, q[2 * c].y} - xi;
#pragma unroll
        for (int c = 0; c < 4; ++c) dy[c] = nb_f2{q[2 * c].z, q[2 * c].w} - yi;
#pragma unroll
        for (int c = 0; c < 4; ++c) dz[c] = nb_f2{q[2 * c + 1].x, q[2 * c + 1].y} - zi;
#pragma unroll
        for (int c = 0; c < 4; ++c) d2[c] = __builtin_elementwise_fma(dx[c], dx[c], e2);
#pragma unroll
        for (int c = 0; c < 4; ++c) d2[c] = __builtin_elementwise_fma(dy[c], dy[c], d2[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) d2[c] = __builtin_elementwise_fma(dz[c], dz[c], d2[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = d2[c] * d2[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = r[c] * d2[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = nb_f2{q[2 * c + 1].z, q[2 * c + 1].w} * r[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) ax = __builtin_elementwise_fma(r[c], dx[c], ax);
#pragma unroll
        for (int c = 0; c < 4; ++c) ay = __builtin_elementwise_fma(r[c], dy[c], ay);
#pragma unroll
        for (int c = 0; c < 4; ++c) az = __builtin_elementwise_fma(r[c], dz[c], az);
    };

    for (uint32_t b = 0; b < nblk; ++b) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // block b has landed (in-order counter); b+1, b+2 in flight
        dma(b + 3);                                          // into the slot that was read during iteration b-1
        const nb_v4f* blk = ring + (b % RING) * 64;
        const uint32_t left = nu - b * 8;
        const uint32_t cnt = left < 8 ? left : 8;            // units of this block inside the share (even)
        nb_v4f qa[8], qb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) qa[k] = blk[k];
        for (uint32_t u = 0; u < cnt; u += 2) {              // unit u+1 is read while unit u computes, u+2 while u+1 does
#pragma unroll
            for (int k = 0; k < 8; ++k) qb[k] = blk[(u + 1) * 8 + k];
            eval(qa);
            const uint32_t un = u + 2 < 8 ? u + 2 : 7;       // (the last read of a block is unused)
#pragma unroll
            for (int k = 0; k < 8; ++k) qa[k] = blk[un * 8 + k];
            eval(qb);
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the trailing re-reads have landed: the ring is reusable
    float sx = ax.x + ax.y, sy = ay.x + ay.y, sz = az.x + az.y;

    if constexpr (WS > 1) {
        // waves 1.. leave their sums in the first row of their own ring; wave 0 adds them in wave order
        if (wv > 0) ring[lane] = nb_v4f{sx, sy, sz, 0.0f};
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 1; w < WS; ++w) { const nb_v4f t = lds[w * (RING * 64) + lane]; sx += t.x; sy += t.y; sz += t.z; }
    }
    jstep_finish(sx, sy, sz, bi, v0, a0, i, valid, lane, n, bodies_out, pairs_out, vel, acc, partial, ticket, poison, G, dt);
}

}  // namespace nb

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int WS, bool RINGK = true>
void run(uint32_t n, uint32_t js, int steps, bool check)
{
    const uint32_t gx = (n + 63) / 64;
    const dim3 grid(gx, js);
    const uint32_t units = (((n + 1) / 2 + 3) / 4 + 1) & ~1u, upw = 2 * ((units / 2 + WS * js - 1) / (WS * js));
    std::vector<float4> hb(n), hz(n, float4{0, 0, 0, 0});
    srand(1);
    for (uint32_t i = 0; i < n; ++i) hb[i] = float4{(float)rand() / RAND_MAX, (float)rand() / RAND_MAX, (float)rand() / RAND_MAX, (0.5f + (float)rand() / RAND_MAX) / n};
    float4 *b0, *b1, *p0, *p1, *v, *a, *part;
    uint32_t* tick;
    CK(hipMalloc(&part, (size_t)16 * 64 * gx * js)); CK(hipMalloc(&tick, 4 * gx)); CK(hipMemset(tick, 0, 4 * gx));
    CK(hipMemset(part, 0xff, (size_t)16 * 64 * gx * js));
    const size_t pbytes = (size_t)(units + 8) * 8 * 16;   // one spare unit: the loop requests one unit past its range
    CK(hipMalloc(&b0, 16 * n)); CK(hipMalloc(&b1, 16 * n)); CK(hipMalloc(&v, 16 * n)); CK(hipMalloc(&a, 16 * n));
    CK(hipMalloc(&p0, pbytes)); CK(hipMalloc(&p1, pbytes));
    CK(hipMemset(p0, 0, pbytes)); CK(hipMemset(p1, 0, pbytes));
    CK(hipMemcpy(b0, hb.data(), 16 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hz.data(), 16 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, hz.data(), 16 * n, hipMemcpyHostToDevice));
    const float G = 1.0f, eps2 = 1e-4f, dt = 1e-3f;
    hipLaunchKernelGGL(nb::nb_pairs_pack<0>, (n / 2 + 256) / 256, 256, 0, 0, (const float4*)b0, p0, n, G);
    CK(hipDeviceSynchronize());
    double err = -1.0;
    if (check) {
        hipLaunchKernelGGL((RINGK ? nb::nb_step_jring<WS> : nb::nb_step_jpk<WS>), grid, 64 * WS, 0, 0, (const float4*)b0, (const float4*)p0, b1, p1, v, a, part, tick, n, upw, 1u, G, eps2, dt);
        CK(hipDeviceSynchronize());
        std::vector<float4> ha(n), hx(n), hp((size_t)(units + 8) * 8);
        CK(hipMemcpy(ha.data(), a, 16 * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx.data(), b1, 16 * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hp.data(), p1, pbytes, hipMemcpyDeviceToHost));
        double amax = 0, emax = 0;
        std::vector<double> ref(3 * (size_t)n);
        for (uint32_t i = 0; i < n; ++i) {
            double sx = 0, sy = 0, sz = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const double dx = (double)hb[j].x - hb[i].x, dy = (double)hb[j].y - hb[i].y, dz = (double)hb[j].z - hb[i].z;
                const double d2 = dx * dx + dy * dy + dz * dz + 1e-4;
                const double s = (double)hb[j].w / (d2 * std::sqrt(d2));
                sx += s * dx; sy += s * dy; sz += s * dz;
            }
            ref[3 * i] = sx; ref[3 * i + 1] = sy; ref[3 * i + 2] = sz;
            amax = std::max(amax, std::sqrt(sx * sx + sy * sy + sz * sz));
        }
        for (uint32_t i = 0; i < n; ++i)
            emax = std::max({emax, std::fabs(ha[i].x - ref[3 * i]), std::fabs(ha[i].y - ref[3 * i + 1]), std::fabs(ha[i].z - ref[3 * i + 2])});
        err = emax / amax;
        // the pair copy of the new positions must equal the AoS copy
        int bad = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const float4 lo = hp[(i / 2) * 2], hi = hp[(i / 2) * 2 + 1];
            const float x = (i & 1) ? lo.y : lo.x, y = (i & 1) ? lo.w : lo.z, z = (i & 1) ? hi.y : hi.x, m = (i & 1) ? hi.w : hi.z;
            if (x != hx[i].x || y != hx[i].y || z != hx[i].z || m != G * hx[i].w) ++bad;
        }
        if (bad) printf("   !! %d pair rows differ from the AoS rows\n", bad);
        // restore state
        CK(hipMemcpy(v, hz.data(), 16 * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(a, hz.data(), 16 * n, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < steps; ++k) {
            const bool o = k & 1;
            hipLaunchKernelGGL((RINGK ? nb::nb_step_jring<WS> : nb::nb_step_jpk<WS>), grid, 64 * WS, 0, 0, (const float4*)(o ? b1 : b0), (const float4*)(o ? p1 : p0), o ? b0 : b1,
                               o ? p0 : p1, v, a, part, tick, n, upw, 0u, G, eps2, dt);
        }
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const double us = 1e3 * best / steps;
    printf("N=%6u %s<WS%2d> js %2u grid %5u waves %6u  %8.2f us/step  %.3e pairs/s = %4.1f %% of 7.865e12", n, RINGK ? "jring" : "jpk", WS, js, gx * js, gx * js * WS, us,
           (double)n * (n - 1) / (us * 1e-6), 100.0 * (double)n * (n - 1) / (us * 1e-6) / 7.865e12);
    if (check) printf("  | max |a - a_fp64| / max|a| = %.2e", err);
    printf("\n");
    CK(hipFree(b0)); CK(hipFree(b1)); CK(hipFree(v)); CK(hipFree(a)); CK(hipFree(p0)); CK(hipFree(p1)); CK(hipFree(part)); CK(hipFree(tick));
}

template <bool K>
void sweep()
{
    for (uint32_t n : {2048u, 4096u}) {
        for (uint32_t js : {1u, 2u, 4u, 8u}) { run<4, K>(n, js, 400, true); run<8, K>(n, js, 400, true); run<16, K>(n, js, 400, js == 1); }
    }
    for (uint32_t n : {6000u, 8192u}) {
        for (uint32_t js : {1u, 2u, 4u, 8u}) { run<4, K>(n, js, 200, js == 8); run<8, K>(n, js, 200, false); run<16, K>(n, js, 200, false); }
    }
    for (uint32_t n : {12000u, 16384u, 20000u}) {
        for (uint32_t js : {1u, 2u, 3u, 4u, 5u, 6u, 8u}) { run<4, K>(n, js, 100, false); run<8, K>(n, js, 100, false); run<16, K>(n, js, 100, false); }
    }
    for (uint32_t n : {32768u, 40002u, 65536u}) {
        for (uint32_t js : {1u, 2u, 3u, 4u, 5u, 6u}) { run<4, K>(n, js, 30, false); run<8, K>(n, js, 30, n == 40002u && js == 3); run<16, K>(n, js, 30, false); }
    }
}

int main(int argc, char** argv)
{
    if (argc > 1 && argv[1][0] == 's') sweep<false>(); else sweep<true>();
    return 0;
}
