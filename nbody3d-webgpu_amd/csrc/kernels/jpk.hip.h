// kernels/jpk.hip.h -- the j-packed fused step nb_step_jpk (in-launch ticket reduction) and the (x, y, z, G m) / pair-transposed position copies.
// Part of nb_kernels.hip.h (include that, not this file).
#pragma once

namespace nb {

// ---- j-packed SGPR step (mid-size systems) ----------------------------------------------------
// The packed kernels above vectorise across TWO i-BODIES of a lane, so a lane owns at least two
// bodies and a system of N bodies offers N/128 waves of i-work: to fill 1,024 SIMDs below
// N ~ 16k the rest has to come from j-splits through memory (partials + K2) or from lanes sharing
// a body over an LDS tile whose hand-over costs up to 45 % of a tile period (profiles/r02/ubench4_*).
// Here the two halves of a packed instruction are TWO j-BODIES against ONE i-body per lane:
//   * j comes from a pair-transposed copy of the positions, pairs[p] = (x0,x1, y0,y1, z0,z1,
//     G*m0,G*m1) for bodies 2p, 2p+1: one s_load_dwordx8 per j-pair through the scalar cache and
//     the four 64-bit SGPR pairs feed v_pk_add / v_pk_fma / v_pk_mul directly -- no LDS tile, no
//     barrier and no hand-over in the loop, and (G*m_j)*inv is the reference's product
//     (nbody3d.js:236) instead of G applied to the finished sum;
//   * the WS waves of a workgroup (up to 16: 1,024 threads) hold the SAME 64 i-bodies and each streams
//     1/WS of the pairs; the sums meet in LDS once, in wave order (deterministic), and wave 0
//     applies nbody3d.js:274-290 and writes the new positions in both layouts to the OTHER
//     buffers (ping-pong, as nb_step_fused): one launch per step, N/64 * WS waves.
//   * instruction mix per two pairs: the same 12 v_pk + 2 v_rsq_f32 = 64 issue cycles.
// Pairs past the system (zero position, zero mass: set once, never rewritten) pad every wave's
// range to whole 4-pair requests and contribute exactly 0.
typedef float nb_f8 __attribute__((ext_vector_type(8)));

// What follows once wave 0 of a workgroup holds the workgroup's 64 sums (jpk / jring kernels): the
// reduction across the j-splits of gridDim.y workgroups, the integrator, and both position layouts.
__device__ __forceinline__ void jstep_finish(float sx, float sy, float sz, const float4& bi, nb_v4f v0, nb_v4f a0,
                                             const uint32_t i, const bool valid, const int lane, const uint32_t n,
                                             float4* __restrict__ bodies_out, float4* __restrict__ pairs_out,
                                             float4* __restrict__ vel, float4* __restrict__ acc, float4* partial,
                                             uint32_t* ticket, const uint32_t poison /* bit 0: NB_FLAG_POISON, bit 1: NB_FLAG_JPK_FENCED */,
                                             const float G, const float dt)
{
    // j split over gridDim.y workgroups (systems with fewer than ~4 i-blocks per CU): every workgroup
    // stores its 64 partial sums, and the one that arrives LAST at the i-block's ticket adds all of them
    // in ascending split order (deterministic) and integrates -- the in-launch split reduction of
    // cdna_hip_programming.md §5 in its write-through form (sc1 stores, drain, relaxed ticket; the last arriver:
    // agent-scope acquire, then plain vector loads).  One launch per step at any split count, no K2.
    const uint32_t nsplit = gridDim.y;
    if (nsplit > 1) {
        float4* const mine = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + lane;
        // The partial row goes out WRITE-THROUGH (sc1: past this XCD's L2 once drained), so the workgroup needs no release
        // fence.  A release fence is a buffer_wbl2 -- a write-back of the whole L2 -- per workgroup: with it every split form of
        // this kernel was 2-8 us slower per step (N=12,000: 46.3 -> 38.3 us, N=8,192: 22.0 -> 19.8; profiles/r02/
        // ubench5_sc1_vs_fence.txt).  The last arriver still acquires (buffer_inv sc1) before its plain loads.
        uint32_t drawn = 0;
        if (poison & 2u) {
            // NB_FLAG_JPK_FENCED: the textbook form -- plain store, then an agent-scope RELEASE on the ticket (hipcc emits the
            // L2 write-back itself).  Inside the compiler's memory model on any part / partition mode; 2-8 us per step slower.
            *mine = float4{sx, sy, sz, 0.0f};
            if (lane == 0) drawn = __hip_atomic_fetch_add(ticket + blockIdx.x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const nb_v4f pv = nb_v4f{sx, sy, sz, 0.0f};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(mine), "v"(pv) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) drawn = __hip_atomic_fetch_add(ticket + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        drawn = __builtin_amdgcn_readfirstlane(drawn);
        if (drawn != nsplit - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(ticket + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        float4* q = partial + (size_t)blockIdx.x * 64 + lane;
        const size_t stride = (size_t)gridDim.x * 64;
        sx = 0; sy = 0; sz = 0;
        uint32_t sp = 0;
        for (; sp + 4 <= nsplit; sp += 4) {           // 4 independent loads per trip, added in ascending split order
            const nb_v4f p0 = *reinterpret_cast<nb_v4f*>(q), p1 = *reinterpret_cast<nb_v4f*>(q + stride);
            const nb_v4f p2 = *reinterpret_cast<nb_v4f*>(q + 2 * stride), p3 = *reinterpret_cast<nb_v4f*>(q + 3 * stride);
            sx += p0.x; sy += p0.y; sz += p0.z;
            sx += p1.x; sy += p1.y; sz += p1.z;
            sx += p2.x; sy += p2.y; sz += p2.z;
            sx += p3.x; sy += p3.y; sz += p3.z;
            q += 4 * stride;
        }
        for (; sp < nsplit; ++sp) {
            const nb_v4f p0 = *reinterpret_cast<nb_v4f*>(q);
            sx += p0.x; sy += p0.y; sz += p0.z;
            q += stride;
        }
        if (poison & 1u) {     // validation mode (NB_FLAG_POISON): a partial that is ever read stale reads NaN
            const float nan = __builtin_nanf("");
            q = partial + (size_t)blockIdx.x * 64 + lane;
            for (sp = 0; sp < nsplit; ++sp, q += stride) *q = float4{nan, nan, nan, nan};
        }
    }

    asm volatile("" : "+v"(v0), "+v"(a0));          // first use of the prefetched rows: after the loop
    float4 nx = float4{0, 0, 0, 0}, nv, na;
    if (valid) {
        leapfrog<float>(bi, float4{v0.x, v0.y, v0.z, v0.w}, float4{a0.x, a0.y, a0.z, a0.w}, sx, sy, sz, dt, nx, nv, na);
        vel[i] = nv;                                               // :281
        bodies_out[i] = nx;                                        // :283 (other buffer)
        acc[i] = na;                                               // :290
    }
    // the pair-transposed copy of the new positions: lanes 2k, 2k+1 hold one pair; the even lane
    // stores (x0,x1,y0,y1), the odd lane (z0,z1,Gm0,Gm1) -- every lane one 16-B store
    const float gm = G * nx.w;                                     // lanes past the system: zero body
    const bool odd = lane & 1;
    const float s0 = odd ? nx.x : nx.z, s1 = odd ? nx.y : gm;
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
    if ((i & ~1u) < n) pairs_out[i] = odd ? float4{r0, nx.z, r1, gm} : float4{nx.x, r0, nx.y, r1};
}

// AoS positions -> pair-transposed copy with G folded into the mass lanes.
template <int UNUSED = 0>     // a template only so that the header can be included by several translation units
__global__ __launch_bounds__(kBlock) void nb_pairs_pack(const float4* __restrict__ bodies, float4* __restrict__ pairs,
                                                       uint32_t n, float G)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (2 * p >= n) return;
    const float4 a = ld4(bodies + 2 * p);
    float4 b = float4{0, 0, 0, 0};
    if (2 * p + 1 < n) b = ld4(bodies + 2 * p + 1);
    pairs[2 * p] = float4{a.x, b.x, a.y, b.y};
    pairs[2 * p + 1] = float4{a.z, b.z, G * a.w, G * b.w};
}

// (x, y, z, m) -> (x, y, z, G*m): the j-stream of the packed f32 K1 forms when G != 1 (rebuilt when the positions
// were written from outside the step or G changed; the step itself keeps its own rows current).
template <int UNUSED = 0>
__global__ __launch_bounds__(kBlock) void nb_gm_pack(const float4* __restrict__ bodies, float4* __restrict__ gm, uint32_t n, float G)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 b = ld4(bodies + i);
    gm[i] = float4{b.x, b.y, b.z, G * b.w};
}

template <int WS>
__global__ __launch_bounds__(64 * WS) __attribute__((amdgpu_waves_per_eu(4, 8)))
void nb_step_jpk(const float4* __restrict__ bodies_in, const float4* __restrict__ pairs_in, float4* __restrict__ bodies_out,
                 float4* __restrict__ pairs_out, float4* __restrict__ vel, float4* __restrict__ acc, float4* partial,
                 uint32_t* ticket, uint32_t n, uint32_t units_per_wave, uint32_t poison, float G, float eps2, float dt)
{
    static_assert(WS >= 1 && WS <= 16, "a workgroup has at most 16 waves");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i = blockIdx.x * 64 + lane;
    const bool valid = i < n;
    const uint32_t ic = valid ? i : n - 1;          // clamped, branch-free (never stored)
    const float4 bi = ld4(bodies_in + ic);
    nb_f2 xi = nb_f2{bi.x, bi.x}, yi = nb_f2{bi.y, bi.y}, zi = nb_f2{bi.z, bi.z};
    const nb_f2 e2 = nb_f2{eps2, eps2};
    nb_f2 ax = nb_f2{0, 0}, ay = nb_f2{0, 0}, az = nb_f2{0, 0};

    // this wave's share: whole units of 4 pairs (8 bodies), an even number of them; the array holds
    // units_total (+1 spare) units, those past the system all zero
    const uint32_t units = (((n + 1) / 2 + 3) / 4 + 1) & ~1u;
    uint32_t u0 = (blockIdx.y * WS + wv) * units_per_wave, u1 = u0 + units_per_wave;   // splits (grid y) x waves, ascending
    if (u0 > units) u0 = units;
    if (u1 > units) u1 = units;

    // Warm this XCD's L2 with the wave's whole share before streaming it through the scalar cache.
    // The pair array was written by the previous launch (other XCDs' stores are only visible below L2),
    // so the first touch of every line is an Infinity-Cache round trip; a scalar stream exposes it once
    // per unit -- measured 1,200-1,300 cycles per 4-pair unit, N-independent -- where one vector load
    // per 4 KiB (lane stride = one 64-B line, result never used) has all of them in flight at once.
    // The loads complete asynchronously into `sink`: the register stays live ("+v" in every statement)
    // up to the explicit vmcnt(0) below, so the allocator cannot hand it to anything else meanwhile.
    uint32_t sink = 0;
    {
        const char* base = (const char*)(pairs_in + (size_t)u0 * 8);
        const uint32_t bytes = (u1 - u0) * 128u;
        for (uint32_t off = (uint32_t)lane * 64u; off < bytes; off += 4096u)
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(base + off) : "memory");
    }
    nb_v4f v0 = nb_v4f{0, 0, 0, 0}, a0 = nb_v4f{0, 0, 0, 0};
    if (wv == 0) {                                  // in flight under the loop; pinned below so that nothing consumes them early
        v0 = *reinterpret_cast<const nb_v4f*>(vel + ic);
        a0 = *reinterpret_cast<const nb_v4f*>(acc + ic);
    }

    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink), "+v"(v0), "+v"(a0) : : "memory");   // one round trip for everything above

    struct Unit { nb_f8 p0, p1, p2, p3; };          // 4 pairs = 32 SGPRs
    auto eval = [&](const Unit& q) {
        const nb_f8 p[4] = {q.p0, q.p1, q.p2, q.p3};
        nb_f2 dx[4], dy[4], dz[4], d2[4], r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) dx[c] = nb_f2{p[c][0], p[c][1]} - xi;
#pragma unroll
        for (int c = 0; c < 4; ++c) dy[c] = nb_f2{p[c][2], p[c][3]} - yi;
#pragma unroll
        for (int c = 0; c < 4; ++c) dz[c] = nb_f2{p[c][4], p[c][5]} - zi;
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
        for (int c = 0; c < 4; ++c) r[c] = nb_f2{nb_rsq(r[c].x), nb_rsq(r[c].y)};
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = nb_f2{p[c][6], p[c][7]} * r[c];
        // ascending pairs; the even- and odd-j sums of the lane are added after the loop
#pragma unroll
        for (int c = 0; c < 4; ++c) ax = __builtin_elementwise_fma(r[c], dx[c], ax);
#pragma unroll
        for (int c = 0; c < 4; ++c) ay = __builtin_elementwise_fma(r[c], dy[c], ay);
#pragma unroll
        for (int c = 0; c < 4; ++c) az = __builtin_elementwise_fma(r[c], dz[c], az);
    };
    // hand-placed requests and waits, as in nb_force_pk_sgpr: SMEM returns out of order, so lgkmcnt(0)
    // is the only wait; each sits before the next request and drains a load issued one whole unit
    // (256 issue cycles) earlier.  Early-clobber outputs: no destination on the base-address pair.
    // The i-body and the accumulators are threaded through every statement ("+v"): the whole eval of a
    // unit stays between the request of the next unit and its wait.
#define NB_ACC "+v"(ax), "+v"(ay), "+v"(az), "+v"(xi), "+v"(yi), "+v"(zi)
    auto wait_for = [&](Unit& q) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.p0), "+s"(q.p1), "+s"(q.p2), "+s"(q.p3), NB_ACC : : "memory");
    };
    auto request = [&](Unit& q, const float4* p) {
        asm volatile("s_load_dwordx8 %0, %10, 0x0\n\ts_load_dwordx8 %1, %10, 0x20\n\t"
                     "s_load_dwordx8 %2, %10, 0x40\n\ts_load_dwordx8 %3, %10, 0x60"
                     : "=&s"(q.p0), "=&s"(q.p1), "=&s"(q.p2), "=&s"(q.p3), NB_ACC : "s"(p) : "memory");
    };
#undef NB_ACC
    // units_per_wave is even and the pair array ends with one spare (zero) unit: the loop body is
    // branch-free -- the request after the last unit of a wave reads that spare or the next wave's first
    if (u1 > u0) {
        const float4* pj = pairs_in + (size_t)u0 * 8;     // a unit is 8 float4
        Unit A, B;
        request(A, pj);
        const float4* const pend = pairs_in + (size_t)u1 * 8;
        while (pj != pend) {
            wait_for(A);
            request(B, pj + 8);
            eval(A);
            wait_for(B);
            pj += 16;
            request(A, pj);
            eval(B);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(A.p0), "+s"(A.p1), "+s"(A.p2), "+s"(A.p3));   // the spare request lands in dead registers
    }
    float sx = ax.x + ax.y, sy = ay.x + ay.y, sz = az.x + az.y;

    if constexpr (WS > 1) {
        __shared__ float red[WS - 1][3][64];
        if (wv > 0) { red[wv - 1][0][lane] = sx; red[wv - 1][1][lane] = sy; red[wv - 1][2][lane] = sz; }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 0; w < WS - 1; ++w) { sx += red[w][0][lane]; sy += red[w][1][lane]; sz += red[w][2][lane]; }
    }

    jstep_finish(sx, sy, sz, bi, v0, a0, i, valid, lane, n, bodies_out, pairs_out, vel, acc, partial, ticket, poison, G, dt);
}

}  // namespace nb
