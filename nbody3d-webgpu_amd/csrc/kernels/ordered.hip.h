// kernels/ordered.hip.h -- ordered-pair force kernels: nb_force (scalar template), nb_force_pk (LDS tile), nb_step_fused / nb_step_direct (one launch per step), nb_force_pk_sgpr (j broadcast from SGPRs).
// Part of nb_kernels.hip.h (include that, not this file).
#pragma once

namespace nb {

// K1, scalar template.  partial[by * i_count + il] = sum over this block's j-range.
//   IPL: i-bodies per lane group; LS: lanes sharing one i-body (power of two, <= 64).
template <typename T, int IPL, int LS>
__global__ __launch_bounds__(kBlock) void nb_force(const typename vec4<T>::type* __restrict__ bodies,
                                                  typename vec4<T>::type* __restrict__ partial, uint32_t n,
                                                  uint32_t i_begin, uint32_t i_count, T G, T eps2,
                                                  uint32_t j_per_split, SplitWindow win,
                                                  const typename vec4<T>::type* __restrict__ /* zero_row: every K1 form takes the same ten parameters */)
{
    using V4 = typename vec4<T>::type;
    static_assert(LS >= 1 && LS <= 64 && (LS & (LS - 1)) == 0, "LS must be a power of two <= 64");
    const uint32_t bxi = blockIdx.x;
    const uint32_t by = win.split(blockIdx.y);
    constexpr int GROUPS = kBlock / LS;    // i-groups per block per k
    constexpr int IPB = GROUPS * IPL;      // i-bodies per block
    __shared__ V4 tile[2][kTile];

    const int tid = threadIdx.x;
    const int grp = tid / LS;
    const int js = tid % LS;

    T xi[IPL], yi[IPL], zi[IPL], ax[IPL], ay[IPL], az[IPL];
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const uint32_t il = bxi * IPB + k * GROUPS + grp;
        const V4 b = ld4(bodies + i_begin + (il < i_count ? il : i_count - 1));   // clamped, branch-free (sum never stored)
        xi[k] = b.x; yi[k] = b.y; zi[k] = b.z;
        ax[k] = 0; ay[k] = 0; az[k] = 0;
    }

    const uint32_t j0 = by * j_per_split;
    uint32_t j1 = j0 + j_per_split;
    if (j1 > n) j1 = n;
    const uint32_t ntiles = (j1 > j0) ? (j1 - j0 + kTile - 1) / kTile : 0;

    // load: raw, clamped, nothing consumes it until finish() right before the LDS store -- a use
    // next to the load would park the wave on vmcnt(0) and expose the latency every tile
    auto stage = [&](uint32_t t) -> V4 {
        const uint32_t j = j0 + t * kTile + tid;
        return ld4(bodies + (j < j1 ? j : j1 - 1));
    };
    auto finish = [&](uint32_t t, V4 b) -> V4 {
        const uint32_t j = j0 + t * kTile + tid;
        b.w = j < j1 ? b.w * G : T(0);      // past the range: zero mass, contributes exactly 0
        return b;
    };

    if (ntiles) tile[0][tid] = finish(0, stage(0));
    __syncthreads();

    for (uint32_t t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        V4 nxt;
        const bool more = (t + 1 < ntiles);
        if (more) nxt = stage(t + 1);        // global load in flight under the tile's compute
        // j-bodies of this tile that are inside the split (the last tile of a split is
        // usually partial: splits are not tile multiples, see plan_launch in nb_plan.cpp); the loop runs
        // in chunks of CH iterations, entries past the range are staged zero-mass bodies
        const uint32_t left = j1 - (j0 + t * kTile);
        const int cnt = left < (uint32_t)kTile ? (int)left : kTile;
        constexpr int CH = (kTile / LS) < 8 ? (kTile / LS) : 8;   // iterations per chunk (LS = 64: 4 per tile)
        const int chunks = ((cnt + LS - 1) / LS + CH - 1) / CH;
        for (int c = 0; c < chunks; ++c) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const V4 b = tile[cur][(c * CH + u) * LS + js];
#pragma unroll
                for (int k = 0; k < IPL; ++k) pair(b.x, b.y, b.z, b.w, xi[k], yi[k], zi[k], eps2, ax[k], ay[k], az[k]);
            }
        }
        if (more) tile[cur ^ 1][tid] = finish(t + 1, nxt);
        __syncthreads();
    }

    if constexpr (LS > 1) {
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            ax[k] = group_sum<LS>(ax[k]);
            ay[k] = group_sum<LS>(ay[k]);
            az[k] = group_sum<LS>(az[k]);
        }
    }
    if (js == LS - 1) {
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const uint32_t il = bxi * IPB + k * GROUPS + grp;
            if (il < i_count) partial[(size_t)by * i_count + il] = V4{ax[k], ay[k], az[k], 0};
        }
    }
}

// ---- packed f32 force loop -----------------------------------------------------------------
// Same algorithm as nb_force<float,...>, but the arithmetic is vectorised ACROSS TWO i-BODIES
// of the lane with the CDNA packed f32 instructions (v_pk_add_f32 / v_pk_fma_f32 /
// v_pk_mul_f32: two f32 lanes per VGPR pair).  Measured on MI355X (profiles/r01/ubench_run1.txt):
// a wave issues one VALU op per 4 cycles whether it is packed or not, so the packed body
// (12 v_pk + 2 v_rsq per TWO pairs instead of 24 + 2) sustains ~25 % more pairs/s than the
// scalar body at the same occupancy.  The j-body needs no shuffles: the ds_read_b128 result
// quad (x,y | z,m) feeds the packed ops through op_sel (lo/hi broadcast).
//   NG = packed groups per lane -> IPL = 2*NG i-bodies per lane;  LS lanes share the IPL bodies;
//   TL = 256-body tile units staged at once (TL = 4: one exposed load latency per 1024 bodies,
//   what the short loops of small systems need).
typedef float nb_f2 __attribute__((ext_vector_type(2)));

// a.hi * b, both halves: v_pk_mul_f32 with the HIGH half of `a` broadcast (op_sel:[1,0] op_sel_hi:[1,1]).  hipcc
// folds a low-half broadcast into a packed op by itself but copies a high half into a fresh register first
// (one v_mov_b32 per j-body for the mass, which sits in the high half of the (z, m) pair: 1 instruction in 15
// of the two-bodies-per-lane loop).
// The multiply consumes a v_rsq_f32 result, and gfx950 needs one wait state between a transcendental and a
// VALU instruction that reads its result; hipcc inserts it for its own instructions but not in front of an asm
// statement (seen: the scheduler sank each v_rsq_f32 right in front of its asm consumer -- wrong sums).  So the
// reciprocal square roots of a stage and its mass multiplies are BOTH volatile asm: volatile statements keep
// their program order, all 2*NC v_rsq_f32 of a stage come before its NC multiplies (NC >= 4 chains), and the
// nearest producer of a multiply's operand is at least three instructions away.  tests/test_isa_guard.py checks
// every packed kernel for an adjacent pair.
__device__ __forceinline__ nb_f2 rsq_ordered(const nb_f2 a)
{
    nb_f2 o;
    asm volatile("v_rsq_f32 %0, %1" : "=v"(o.x) : "v"(a.x));
    asm volatile("v_rsq_f32 %0, %1" : "=v"(o.y) : "v"(a.y));
    return o;
}
__device__ __forceinline__ nb_f2 mul_hi_bcast_ordered(const nb_f2 a, const nb_f2 b)
{
    nb_f2 o;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(o) : "v"(a), "v"(b));
    return o;
}

template <int NG, int LS, int TL>
struct PkCore {
    static constexpr int TILE = kTile * TL;
    static constexpr int ITER = TILE / LS;              // loop iterations per full tile
    static constexpr int U = ITER < 8 ? ITER : 8;       // iterations per unrolled chunk
    static constexpr int JB0 = NG >= 4 ? 1 : 4 / NG;    // j-bodies per stage-major group
    static constexpr int JB = JB0 < U ? JB0 : U;
    static constexpr int NC = JB * NG;                  // independent dependency chains
    static constexpr int UNR = U / JB;
    // NG < 4: unrolling all UNR stages lets the scheduler interleave them until it spills
    // (228 B/lane of scratch at NG = 2); two stages in flight are enough to cover the LDS reads
    static constexpr int UNROLL = NG >= 4 ? UNR : (UNR < 2 ? UNR : 2);
    static_assert(LS >= 1 && LS <= 64 && (LS & (LS - 1)) == 0, "LS must be a power of two <= 64");
    static_assert(TL == 1 || TL == 4 || TL == 8, "TL is 1, 4 or 8");
    static_assert(NC >= 4, "the ordered rsq / multiply statements of a stage rely on >= 4 chains");

    // Accumulates sum_{j in [j0, j1)} (G m_j) r_ij / (|r_ij|^2 + eps2)^{3/2} for the lane's 2*NG bodies over the
    // lane's share of j (every LS-th body of each tile).  `bodies` is the j-stream: rows (x, y, z, G*m_j), so that
    // every pair multiplies (G*m_j) * inv -- the reference's own product, nbody3d.js:236 -- at no per-pair cost
    // (the engine keeps that copy beside the (x, y, z, m) state whenever G != 1: nb_gm_pack, K2 / the fused epilogues).
    static __device__ __forceinline__ void run(const float4* __restrict__ bodies, const float4* __restrict__ zero_row,
                                               const uint32_t j0, const uint32_t j1, const float eps2,
                                               const nb_f2 (&xi)[NG], const nb_f2 (&yi)[NG], const nb_f2 (&zi)[NG],
                                               nb_f2 (&ax)[NG], nb_f2 (&ay)[NG], nb_f2 (&az)[NG])
    {
        __shared__ float4 tile[2][TILE];                     // the only LDS object of the kernel
        const int tid = threadIdx.x;
        const int js = tid % LS;
        const nb_f2 e2 = nb_f2{eps2, eps2};
        const uint32_t ntiles = (j1 > j0) ? (j1 - j0 + TILE - 1) / TILE : 0;

        // Staging by LDS-DMA (global_load_lds_dwordx4: one wave instruction moves the wave's 64 rows = 1 KiB
        // straight into the tile, no VGPR staging, no ds_write, nothing for the wave to wait on until the
        // barrier).  Round 2 staged through registers (global_load_dwordx4 -> G*m and zero-mass mask -> ds_write_b128):
        // the same loop with DMA staging is 17 / 10 / 10 / 7 / 5 / 4 % faster at N = 2,048 / 4,096 / 8,192 / 16,384 /
        // 32,768 / 65,536 (profiles/r02/ubench4_dma_staging.txt).  Rows past the range come from `zero_row`
        // (a zero-mass body at the origin contributes exactly 0): the source address is per lane, the
        // destination is wave-uniform base + lane * 16 B.  hipcc does not count asm loads: every tile ends
        // with an explicit vmcnt(0) before its barrier.
        // A tile that lies wholly inside the range (all but the last one of a range) needs no per-lane work at all:
        // scalar base of the tile + q * 4 KiB, the lane's constant 16-B offset in a VGPR (saddr form), the LDS
        // destination by scalar adds -- zero VALU instructions per DMA where the per-lane form spends 7 and two
        // hazard nops (compare, select low/high half of the pointer, 64-bit add).  The LDS address of the wave's
        // first row is converted once (a generic -> LDS cast per DMA carried a null check each).
        const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)&tile[0][tid & ~63]);
        const uint32_t lane_off = (uint32_t)tid * 16u;
        auto stage = [&](uint32_t t, int buf) {
            const uint32_t jt = j0 + t * TILE;                       // wave-uniform
            if (jt + TILE <= j1) {
                const float4* base = bodies + jt;
#pragma unroll
                for (int q = 0; q < TL; ++q) {
                    const uint32_t dst = lds_wave + (uint32_t)(buf * TILE + q * kBlock) * 16u;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(lane_off), "s"(base + q * kBlock), "s"(dst) : "memory");
                }
            } else {
#pragma unroll
                for (int q = 0; q < TL; ++q) {
                    const uint32_t j = jt + q * kBlock + tid;
                    const float4* src = j < j1 ? bodies + j : zero_row;
                    const uint32_t dst = lds_wave + (uint32_t)(buf * TILE + q * kBlock) * 16u;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
            }
        };

        // one stage: JB j-bodies (rows p[0], p[LS], ...) against the lane's NG packed groups.
        // (Tried and dropped: issuing a stage's ds_read_b128s one stage ahead through asm statements.
        // One wave alone ran the loop 25 % faster, four per SIMD 3-7 % slower -- the statements fence
        // the scheduler at every stage -- and the step time did not move: at these sizes the waves
        // wait on the next tile's global loads, not on LDS.  profiles/r02/ubench4_*.txt.  Tried again after the
        // LDS-DMA staging removed that wait, as ONE asm statement of eight ds_read_b128 per chunk, a whole chunk
        // ahead, no destination in flight across the back edge: 1-3 % SLOWER from N = 2,002 to 6,000, equal at
        // 8,192 -- profiles/r02/ab_lds_read_pipelining.txt.)
        auto math = [&](const float4* p) {
            nb_f2 bx[JB], by[JB], bz[JB], bzm[JB];
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const float4 b = p[u * LS];
                bx[u] = nb_f2{b.x, b.x}; by[u] = nb_f2{b.y, b.y}; bz[u] = nb_f2{b.z, b.z};
                bzm[u] = nb_f2{b.z, b.w};      // the (z, m) register pair of the ds_read_b128 result, as it lies
            }
            nb_f2 dx[NC], dy[NC], dz[NC], d2[NC], r[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) dx[c] = bx[c / NG] - xi[c % NG];
#pragma unroll
            for (int c = 0; c < NC; ++c) dy[c] = by[c / NG] - yi[c % NG];
#pragma unroll
            for (int c = 0; c < NC; ++c) dz[c] = bz[c / NG] - zi[c % NG];
#pragma unroll
            for (int c = 0; c < NC; ++c) d2[c] = __builtin_elementwise_fma(dx[c], dx[c], e2);
#pragma unroll
            for (int c = 0; c < NC; ++c) d2[c] = __builtin_elementwise_fma(dy[c], dy[c], d2[c]);
#pragma unroll
            for (int c = 0; c < NC; ++c) d2[c] = __builtin_elementwise_fma(dz[c], dz[c], d2[c]);
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = d2[c] * d2[c];
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = r[c] * d2[c];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if constexpr (NG == 1) r[c] = rsq_ordered(r[c]);
                else r[c] = nb_f2{nb_rsq(r[c].x), nb_rsq(r[c].y)};
            }
            // m_j * inv.  One i-pair per lane: the explicit high-half broadcast (no v_mov for the mass; -2..-3.4 % per
            // step from N = 3,000 to 10,000).  More pairs per lane: the plain product -- the v_mov is 1 instruction in
            // 29 / 57 there and the ordered statements cost the scheduler more than that (N = 8,192, 4 per lane: +2 %).
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if constexpr (NG == 1) r[c] = mul_hi_bcast_ordered(bzm[c / NG], r[c]);
                else r[c] = nb_f2{bzm[c / NG].y, bzm[c / NG].y} * r[c];
            }
            // accumulate in ascending j for every group (same order as the plain loop)
#pragma unroll
            for (int c = 0; c < NC; ++c) ax[c % NG] = __builtin_elementwise_fma(r[c], dx[c], ax[c % NG]);
#pragma unroll
            for (int c = 0; c < NC; ++c) ay[c % NG] = __builtin_elementwise_fma(r[c], dy[c], ay[c % NG]);
#pragma unroll
            for (int c = 0; c < NC; ++c) az[c % NG] = __builtin_elementwise_fma(r[c], dz[c], az[c % NG]);
        };

        if (ntiles) stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        NB_STAMP(1);

        for (uint32_t t = 0; t < ntiles; ++t) {
            const int cur = t & 1;
            if (t == 1) NB_STAMP_LIGHT(8);
            if (t + 1 < ntiles) stage(t + 1, cur ^ 1);     // lands under this tile's compute (every wave left cur^1 at the last barrier)
            if (t == 1) NB_STAMP_LIGHT(9);
            // JB j-bodies x NG groups = 4 independent dependency chains, issued stage-major:
            // consecutive packed ops never depend on each other, so the backend needs no s_nop
            // between a v_pk_* / v_rsq result and its consumer (gfx950 VALU hazard) and one wave
            // alone keeps the issue port busy.  Exact trip count on a partial last tile, in
            // chunks of U iterations (entries past the range are zero-mass).
            const uint32_t left = j1 - (j0 + t * TILE);
            const int cnt = left < (uint32_t)TILE ? (int)left : TILE;
            const int chunks = ((cnt + LS - 1) / LS + U - 1) / U;
            for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll UNROLL
                for (int uu = 0; uu < UNR; ++uu) math(&tile[cur][(ch * U + uu * JB) * LS + js]);
            }
            if (t == 1) NB_STAMP_LIGHT(10);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 1) NB_STAMP_LIGHT(11);
            __syncthreads();
            if (t == 1) NB_STAMP_LIGHT(12);
        }
        NB_STAMP(2);

        if constexpr (LS > 1) {
            float r[6 * NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                r[6 * g + 0] = ax[g].x; r[6 * g + 1] = ax[g].y; r[6 * g + 2] = ay[g].x;
                r[6 * g + 3] = ay[g].y; r[6 * g + 4] = az[g].x; r[6 * g + 5] = az[g].y;
            }
            group_sum_all<LS, 6 * NG>(r);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                ax[g] = nb_f2{r[6 * g + 0], r[6 * g + 1]}; ay[g] = nb_f2{r[6 * g + 2], r[6 * g + 3]};
                az[g] = nb_f2{r[6 * g + 4], r[6 * g + 5]};
            }
        }
        NB_STAMP(3);
    }
};

// Occupancy target handed to the register allocator/scheduler: NG = 4 needs ~118 VGPRs
// (4 waves/SIMD); telling the backend so keeps it from re-serialising the stage-major order to
// chase an occupancy it cannot reach anyway.  The NG = 1, 2 bodies get the same 128-VGPR budget:
// at 8 (6) waves per SIMD the allocator spilled 10..64 VGPRs of the loop to scratch.
// (TL = 4 stages 32 KiB of LDS per workgroup: at most 5 workgroups per CU, so the target is 4.)
#define NB_PK_WAVES(NG, TL) ((TL) == 8 ? 3 : 4)
// the small-system shapes (one group, 1024-body stages) also hold a prefetched vel/acc pair and two
// stage register sets: allow them the 168-VGPR budget of 3 waves per SIMD instead of spilling
#define NB_PK_WAVES_MIN(NG, TL) ((TL) == 8 ? 2 : ((NG) == 1 && (TL) == 4 ? 3 : 4))

// K1, packed, j-tile in LDS.  `bodies` = the j-stream rows (x, y, z, G*m); the i-rows come from the same array
// (only x, y, z are used).  G itself is unused here: every K1 form takes the same ten parameters.
template <int NG, int LS, int TL>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(NB_PK_WAVES_MIN(NG, TL), NB_PK_WAVES(NG, TL))))
void nb_force_pk(const float4* __restrict__ bodies, float4* __restrict__ partial, uint32_t n, uint32_t i_begin,
                 uint32_t i_count, float G, float eps2, uint32_t j_per_split, SplitWindow win,
                 const float4* __restrict__ zero_row)
{
    const uint32_t bxi = blockIdx.x;
    const uint32_t by = win.split(blockIdx.y);
    constexpr int GROUPS = kBlock / LS;
    constexpr int IPB = GROUPS * 2 * NG;
    const int tid = threadIdx.x;
    const int grp = tid / LS;
    const int js = tid % LS;

    nb_f2 xi[NG], yi[NG], zi[NG], ax[NG], ay[NG], az[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // rows past the shard are clamped to its last row (their sums are never stored): no bounds
        // branch, so the loads of all groups are in flight together
        const uint32_t il0 = bxi * IPB + (2 * g) * GROUPS + grp;
        const uint32_t il1 = il0 + GROUPS;
        const float4 b0 = ld4(bodies + i_begin + (il0 < i_count ? il0 : i_count - 1));
        const float4 b1 = ld4(bodies + i_begin + (il1 < i_count ? il1 : i_count - 1));
        xi[g] = nb_f2{b0.x, b1.x}; yi[g] = nb_f2{b0.y, b1.y}; zi[g] = nb_f2{b0.z, b1.z};
        ax[g] = nb_f2{0, 0}; ay[g] = nb_f2{0, 0}; az[g] = nb_f2{0, 0};
    }
    const uint32_t j0 = by * j_per_split;
    uint32_t j1 = j0 + j_per_split;
    if (j1 > n) j1 = n;
    PkCore<NG, LS, TL>::run(bodies, zero_row, j0, j1, eps2, xi, yi, zi, ax, ay, az);

    if (js == LS - 1) {
        float4* out = partial + (size_t)by * i_count;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t il0 = bxi * IPB + (2 * g) * GROUPS + grp;
            const uint32_t il1 = il0 + GROUPS;
            if (il0 < i_count) out[il0] = float4{ax[g].x, ay[g].x, az[g].x, 0};
            if (il1 < i_count) out[il1] = float4{ax[g].y, ay[g].y, az[g].y, 0};
        }
    }
}

// The whole step in ONE launch ("ping-pong position buffers to fuse K2 into
// K1's epilogue without the race"): every workgroup accumulates its bodies against ALL n bodies
// of bodies_in (the packed LDS-tile loop above), reduces the LS lane sums in the wave and the
// group's last lane applies nbody3d.js:274-290, writing the new positions to bodies_out --
// a different buffer, so no workgroup can stage a half-updated system (the reference's race,
// nbody3d.js:283 vs :257).  vel/acc are only touched by their own lane: in place.
// Bit-identical to nb_force_pk<NG,LS,TL> with jsplit = 1 followed by nb_integrate.
//   bodies_in / bodies_out : the (x, y, z, m) state (ping-pong);
//   jin                    : the j-stream (x, y, z, G*m) that goes with bodies_in (bodies_in itself when G == 1);
//   gout                   : where the (x, y, z, G*m) rows of the NEW positions go (null when G == 1).
template <int NG, int LS, int TL>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(NB_PK_WAVES_MIN(NG, TL), NB_PK_WAVES(NG, TL))))
void nb_step_fused(const float4* __restrict__ bodies_in, const float4* __restrict__ jin, float4* __restrict__ bodies_out,
                   float4* __restrict__ gout, float4* __restrict__ vel, float4* __restrict__ acc, uint32_t n, float G,
                   float eps2, float dt, const float4* __restrict__ zero_row)
{
    constexpr int GROUPS = kBlock / LS;
    constexpr int IPB = GROUPS * 2 * NG;
    constexpr int IPL = 2 * NG;
    constexpr bool PREFETCH = NG == 1;     // vel/acc of the storing lane loaded before the loop (short loops)
    NB_STAMP(0);
    const uint32_t bxi = blockIdx.x;
    const int tid = threadIdx.x;
    const int grp = tid / LS;
    const int js = tid % LS;
    const bool owner = js == LS - 1;

    nb_f2 xi[NG], yi[NG], zi[NG], ax[NG], ay[NG], az[NG];
    float4 v0[PREFETCH ? IPL : 1], a0[PREFETCH ? IPL : 1];
    float w0[PREFETCH ? IPL : 1];           // .w of the lane's bodies (integrated like xyz, :283)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // rows past n are clamped to the last row (never stored): branch-free, all loads in flight
        // together; every lane of a group loads the same rows (one request per group)
        const uint32_t il0 = bxi * IPB + (2 * g) * GROUPS + grp;
        const uint32_t il1 = il0 + GROUPS;
        const uint32_t c0 = il0 < n ? il0 : n - 1, c1 = il1 < n ? il1 : n - 1;
        const float4 b0 = ld4(bodies_in + c0);
        const float4 b1 = ld4(bodies_in + c1);
        xi[g] = nb_f2{b0.x, b1.x}; yi[g] = nb_f2{b0.y, b1.y}; zi[g] = nb_f2{b0.z, b1.z};
        ax[g] = nb_f2{0, 0}; ay[g] = nb_f2{0, 0}; az[g] = nb_f2{0, 0};
        if constexpr (PREFETCH) {
            w0[2 * g] = b0.w; w0[2 * g + 1] = b1.w;
            v0[2 * g] = ld4(vel + c0); a0[2 * g] = ld4(acc + c0);
            v0[2 * g + 1] = ld4(vel + c1); a0[2 * g + 1] = ld4(acc + c1);
        }
    }
    PkCore<NG, LS, TL>::run(jin, zero_row, 0, n, eps2, xi, yi, zi, ax, ay, az);

    if (owner) {
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const int g = k / 2;
            const uint32_t il = bxi * IPB + k * GROUPS + grp;
            if (il < n) {
                float4 v, ao, x;                     // x with all four components: .w is integrated like xyz (:283)
                if constexpr (PREFETCH) {
                    v = v0[k]; ao = a0[k];
                    x = (k & 1) ? float4{xi[g].y, yi[g].y, zi[g].y, w0[k]} : float4{xi[g].x, yi[g].x, zi[g].x, w0[k]};
                } else { v = ld4(vel + il); ao = ld4(acc + il); x = ld4(bodies_in + il); }
                float4 nx, nv, na;
                if (k & 1) leapfrog<float>(x, v, ao, ax[g].y, ay[g].y, az[g].y, dt, nx, nv, na);
                else leapfrog<float>(x, v, ao, ax[g].x, ay[g].x, az[g].x, dt, nx, nv, na);
                vel[il] = nv;                                              // :281
                bodies_out[il] = nx;                                       // :283 (other buffer)
                acc[il] = na;                                              // :290
                if (gout) gout[il] = float4{nx.x, nx.y, nx.z, G * nx.w};   // next step's j-stream row
            }
        }
    }
    NB_STAMP(4);
}

// The fused step for systems of at most 64*MAXJ bodies (MAXJ = 16: N <= 1,024; 32: N <= 2,048), without
// LDS: a wave's 64 lanes share two bodies (the nb_step_fused<1,64,*> mapping) and lane js needs
// exactly the j-bodies js, js+64, js+128, ... -- at most MAXJ rows, so it loads them straight into
// registers (coalesced: 1 KiB per wave load, every load of the kernel in flight at once) and runs
// the packed loop on registers.  No tile store, no barrier, no ds_read latency: the step is three
// memory round trips (arguments, loads, stores) and 64 issue cycles per j.  Every wave reads all
// N rows itself (4x the L2 traffic of the tiled kernel): only for systems this small.
// Same j order per lane and same reduction as nb_step_fused<1,64,*>: bit-identical results.
template <int MAXJ>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MAXJ > 16 ? 2 : 3, 4)))
void nb_step_direct(const float4* __restrict__ bodies_in, const float4* __restrict__ jin, float4* __restrict__ bodies_out,
                    float4* __restrict__ gout, float4* __restrict__ vel, float4* __restrict__ acc, uint32_t n, float G,
                    float eps2, float dt, const float4* __restrict__ /* zero_row: same parameter list as nb_step_fused */)
{
    constexpr int GROUPS = kBlock / 64;     // one wave per pair of bodies
    constexpr int IPB = GROUPS * 2;
    const int tid = threadIdx.x;
    const int grp = tid / 64, js = tid % 64;
    const uint32_t il0 = blockIdx.x * IPB + grp, il1 = il0 + GROUPS;
    const uint32_t c0 = il0 < n ? il0 : n - 1, c1 = il1 < n ? il1 : n - 1;
    // every global load of the kernel, back to back, nothing consuming them yet
    const float4 b0 = ld4(bodies_in + c0), b1 = ld4(bodies_in + c1);
    const float4 v0 = ld4(vel + c0), v1 = ld4(vel + c1), a0 = ld4(acc + c0), a1 = ld4(acc + c1);
    nb_v4f q[MAXJ];
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
        const uint32_t j = (uint32_t)k * 64u + (uint32_t)js;
        q[k] = *reinterpret_cast<const nb_v4f*>(jin + (j < n ? j : n - 1));      // (x, y, z, G*m)
    }
    // pin all MAXJ loads HERE, ahead of the first stage: left alone the backend sinks the loads of
    // the later stages into those stages' (wave-uniform) branches and pays their latency there
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) asm volatile("" : "+v"(q[k]));
    const nb_f2 xi = nb_f2{b0.x, b1.x}, yi = nb_f2{b0.y, b1.y}, zi = nb_f2{b0.z, b1.z};
    nb_f2 ax = nb_f2{0, 0}, ay = nb_f2{0, 0}, az = nb_f2{0, 0};
    const nb_f2 e2 = nb_f2{eps2, eps2};
    const uint32_t nj = (n + 63) / 64;      // rows of 64 bodies that exist (wave-uniform)
#pragma unroll
    for (int k0 = 0; k0 < MAXJ; k0 += 4) {
        if ((uint32_t)k0 < nj) {            // 4 j-bodies = 4 independent chains, stage-major as in PkCore
            nb_f2 bx[4], by[4], bz[4], bm[4], dx[4], dy[4], dz[4], d2[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t j = (uint32_t)(k0 + u) * 64u + (uint32_t)js;
                const nb_v4f b = q[k0 + u];
                const float gm = j < n ? b.w : 0.0f;              // past the end: zero mass, contributes exactly 0
                bx[u] = nb_f2{b.x, b.x}; by[u] = nb_f2{b.y, b.y}; bz[u] = nb_f2{b.z, b.z}; bm[u] = nb_f2{gm, gm};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) dx[c] = bx[c] - xi;
#pragma unroll
            for (int c = 0; c < 4; ++c) dy[c] = by[c] - yi;
#pragma unroll
            for (int c = 0; c < 4; ++c) dz[c] = bz[c] - zi;
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
            for (int c = 0; c < 4; ++c) r[c] = bm[c] * r[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) ax = __builtin_elementwise_fma(r[c], dx[c], ax);
#pragma unroll
            for (int c = 0; c < 4; ++c) ay = __builtin_elementwise_fma(r[c], dy[c], ay);
#pragma unroll
            for (int c = 0; c < 4; ++c) az = __builtin_elementwise_fma(r[c], dz[c], az);
        }
    }
    float red[6] = {ax.x, ax.y, ay.x, ay.y, az.x, az.y};
    group_sum_all<64, 6>(red);
    if (js == 63) {
        float4 nx, nv, na;
        if (il0 < n) {
            leapfrog<float>(b0, v0, a0, red[0], red[2], red[4], dt, nx, nv, na);
            vel[il0] = nv; bodies_out[il0] = nx; acc[il0] = na;
            if (gout) gout[il0] = float4{nx.x, nx.y, nx.z, G * nx.w};
        }
        if (il1 < n) {
            leapfrog<float>(b1, v1, a1, red[1], red[3], red[5], dt, nx, nv, na);
            vel[il1] = nv; bodies_out[il1] = nx; acc[il1] = na;
            if (gout) gout[il1] = float4{nx.x, nx.y, nx.z, G * nx.w};
        }
    }
}

// K1, packed form with the j-bodies broadcast from SGPRs instead of LDS (SURVEY.md §8 f3
// "scalar-load (SGPR) j-broadcast A/B against the LDS tile").  j is wave-uniform, so
// bodies[j] is fetched with s_load_dwordx4 through the scalar cache and the packed ops
// take the (x,y | z,m) SGPR pairs directly (op_sel broadcast): no LDS tile, no barrier in the
// loop, no v_mov for the mass.  `bodies` holds the j-stream rows (x, y, z, G*m): (G*m_j)*inv per pair is the
// reference's own product (nbody3d.js:236); the parameter G is unused (same ten parameters as every K1 form).
//   WS = 1: the 4 waves of a workgroup hold different i-bodies (256 lanes x 2*NG) and stream the
//           same j-range;
//   WS = 4: the 4 waves hold the SAME 64 x 2*NG i-bodies and each streams a quarter of the
//           workgroup's j-range; their sums are added through LDS in wave order (deterministic)
//           and ONE partial is stored: a quarter of the j-splits, partial arrays and K2 traffic
//           for the same grid size and the same work per wave.
//   PAIRS:  a body arrives as two 64-bit SGPR pairs (x,y) (z,m) -- 8 s_load_dwordx2 per 4 bodies
//           instead of 4 s_load_dwordx4 -- so that the backend folds all four broadcasts into the
//           packed ops (with a quad it copies z and m to VGPRs first: 2 v_mov per body).  Pays on
//           long loops only (+0.4..1.3 % at 8,192 bodies per wave, -1.2 % at 2,048).
template <int NG, int WS, bool PAIRS = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(NG >= 4 ? 4 : 6, NG >= 4 ? 4 : 6)))
void nb_force_pk_sgpr(const float4* __restrict__ bodies, float4* __restrict__ partial, uint32_t n, uint32_t i_begin,
                      uint32_t i_count, float G, float eps2, uint32_t j_per_split, SplitWindow win,
                      const float4* __restrict__ /* zero_row */)
{
    static_assert(WS == 1 || WS == 4, "WS is 1 or 4");
    constexpr int IPL = 2 * NG;
    constexpr int LANES = kBlock / WS;          // i-lanes per workgroup
    constexpr int IPB = LANES * IPL;
    const uint32_t bxi = blockIdx.x;
    const uint32_t by = win.split(blockIdx.y);
    const int tid = threadIdx.x;
    const int lane = tid % LANES;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid / LANES);   // wave-uniform: which j-quarter (WS = 4)

    nb_f2 xi[NG], yi[NG], zi[NG], ax[NG], ay[NG], az[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { ax[g] = nb_f2{0, 0}; ay[g] = nb_f2{0, 0}; az[g] = nb_f2{0, 0}; }
    // The wave's i-rows, all 2*NG loads in flight together.  Called AFTER the first scalar request of the
    // j-stream has been issued (below): neither depends on the other, and a workgroup's prologue is then
    // one memory round trip instead of three (i-rows, i-rows behind a recycled register, first j request).
    auto load_i_rows = [&]() {
        nb_v3f b0[NG], b1[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t il0 = bxi * IPB + (2 * g) * LANES + lane;
            const uint32_t il1 = il0 + LANES;
            b0[g] = ld3(bodies + i_begin + (il0 < i_count ? il0 : i_count - 1));   // clamped, branch-free
            b1[g] = ld3(bodies + i_begin + (il1 < i_count ? il1 : i_count - 1));
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) { xi[g] = nb_f2{b0[g].x, b1[g].x}; yi[g] = nb_f2{b0[g].y, b1[g].y}; zi[g] = nb_f2{b0[g].z, b1[g].z}; }
    };
    const nb_f2 e2 = nb_f2{eps2, eps2};
    uint32_t j0 = by * j_per_split;
    uint32_t j1 = j0 + j_per_split;
    if (j1 > n) j1 = n;
    if constexpr (WS == 4) {
        // quarter of the split, a multiple of 8 bodies (the split itself is one); the last wave takes the rest
        const uint32_t len = j1 > j0 ? j1 - j0 : 0;
        const uint32_t q = ((len / 4 + 7) / 8) * 8;
        uint32_t a = j0 + wv * q, b = a + q;
        if (wv == 3 || b > j1) b = j1;
        if (a > j1) a = j1;
        j0 = a; j1 = b;
    }

    auto eval4 = [&](const float4 q0, const float4 q1, const float4 q2, const float4 q3) {
        const float4 q[4] = {q0, q1, q2, q3};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 b = q[u];
            const nb_f2 bx = nb_f2{b.x, b.x}, by2 = nb_f2{b.y, b.y}, bz = nb_f2{b.z, b.z}, bm = nb_f2{b.w, b.w};
            nb_f2 dx[NG], dy[NG], dz[NG], d2[NG], r[NG];
#pragma unroll
            for (int c = 0; c < NG; ++c) dx[c] = bx - xi[c];
#pragma unroll
            for (int c = 0; c < NG; ++c) dy[c] = by2 - yi[c];
            // z_j - z_i with the LOW half of the (z, m) SGPR pair broadcast, spelled out: left to itself hipcc folds the
            // (x, y) pair of a body that arrived as an SGPR quad into the packed ops but copies z and the mass to VGPRs
            // first (2 v_mov_b32 per body: 2 instructions in 58 at four bodies per lane).  With z taken straight from the
            // pair the mass moves by s_mov_b32 -- a scalar-unit instruction -- and the loop carries no VALU copy at all.
            // Pure function of its inputs and not fed by a transcendental: plain (non-volatile) asm, no hazard to mind.
            if constexpr (!PAIRS) {
                const nb_f2 bzm = nb_f2{b.z, b.w};
#pragma unroll
                for (int c = 0; c < NG; ++c)
                    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz[c]) : "s"(bzm), "v"(zi[c]));
            } else {
#pragma unroll
                for (int c = 0; c < NG; ++c) dz[c] = bz - zi[c];
            }
#pragma unroll
            for (int c = 0; c < NG; ++c) d2[c] = __builtin_elementwise_fma(dx[c], dx[c], e2);
#pragma unroll
            for (int c = 0; c < NG; ++c) d2[c] = __builtin_elementwise_fma(dy[c], dy[c], d2[c]);
#pragma unroll
            for (int c = 0; c < NG; ++c) d2[c] = __builtin_elementwise_fma(dz[c], dz[c], d2[c]);
#pragma unroll
            for (int c = 0; c < NG; ++c) r[c] = d2[c] * d2[c];
#pragma unroll
            for (int c = 0; c < NG; ++c) r[c] = r[c] * d2[c];
#pragma unroll
            for (int c = 0; c < NG; ++c) r[c] = nb_f2{nb_rsq(r[c].x), nb_rsq(r[c].y)};
#pragma unroll
            for (int c = 0; c < NG; ++c) r[c] = bm * r[c];
#pragma unroll
            for (int c = 0; c < NG; ++c) ax[c] = __builtin_elementwise_fma(r[c], dx[c], ax[c]);
#pragma unroll
            for (int c = 0; c < NG; ++c) ay[c] = __builtin_elementwise_fma(r[c], dy[c], ay[c]);
#pragma unroll
            for (int c = 0; c < NG; ++c) az[c] = __builtin_elementwise_fma(r[c], dz[c], az[c]);
        }
    };

    // 2 x 4 bodies live in SGPRs, fetched with hand-placed s_load_dwordx4 (hipcc sinks a
    // plain scalar load next to its first use, which exposes the whole latency).  SMEM
    // returns out of order, so lgkmcnt(0) is the only usable wait; every wait sits BEFORE
    // the next request, so it only drains a load issued one whole eval (4 bodies x NG groups
    // = 1024 issue cycles at NG = 4) earlier.  The accumulators are threaded through every
    // asm statement ("+v") so the packed math cannot drift across a wait or a request;
    // nothing else in the loop uses lgkmcnt (no LDS), so hipcc inserts no waits of its own.
    // The destination quads are early-clobber ("=&s"): none of them may be allocated on the
    // base-address pair, which the later loads of the same statement still read.
    typedef float nb_f4 __attribute__((ext_vector_type(4)));   // native vector: usable as an "s" asm operand
    struct Quad { nb_f4 q0, q1, q2, q3; };   // 4 bodies = 16 SGPRs
    auto f4 = [](const nb_f4& v) { return float4{v.x, v.y, v.z, v.w}; };
#define NB_ACC2 "+v"(ax[0]), "+v"(ax[1]), "+v"(ay[0]), "+v"(ay[1]), "+v"(az[0]), "+v"(az[1])
#define NB_ACC4 "+v"(ax[0]), "+v"(ax[1]), "+v"(ax[2]), "+v"(ax[3]), "+v"(ay[0]), "+v"(ay[1]), "+v"(ay[2]), "+v"(ay[3]), \
                "+v"(az[0]), "+v"(az[1]), "+v"(az[2]), "+v"(az[3])
#define NB_LOAD4(o) "s_load_dwordx4 %0, %" #o ", 0x0\n\ts_load_dwordx4 %1, %" #o ", 0x10\n\t" \
                    "s_load_dwordx4 %2, %" #o ", 0x20\n\ts_load_dwordx4 %3, %" #o ", 0x30"
    auto wait_for = [&](Quad& q) {
        if constexpr (NG == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.q0), "+s"(q.q1), "+s"(q.q2), "+s"(q.q3), NB_ACC4 : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.q0), "+s"(q.q1), "+s"(q.q2), "+s"(q.q3), NB_ACC2 : : "memory");
    };
    auto request = [&](Quad& q, const float4* p) {   // bodies p[0..3]
        if constexpr (NG == 4) asm volatile(NB_LOAD4(16) : "=&s"(q.q0), "=&s"(q.q1), "=&s"(q.q2), "=&s"(q.q3), NB_ACC4 : "s"(p) : "memory");
        else asm volatile(NB_LOAD4(10) : "=&s"(q.q0), "=&s"(q.q1), "=&s"(q.q2), "=&s"(q.q3), NB_ACC2 : "s"(p) : "memory");
    };
#undef NB_LOAD4
#undef NB_ACC2
#undef NB_ACC4
    const uint32_t nb8 = j1 > j0 ? (j1 - j0) / 8 : 0;
    const float4* pj = bodies + j0;
    uint32_t j = j0;
    if constexpr (PAIRS) {
        // A/B arm: every body as two 64-bit SGPR pairs (x,y) and (z,m), 8 s_load_dwordx2 per 4 bodies, so
        // that all four broadcasts fold into the packed ops as SGPR operands (no v_mov for z and m)
        struct Oct { nb_f2 p[8]; };
        auto f4p = [](const nb_f2& xy, const nb_f2& zm) { return float4{xy.x, xy.y, zm.x, zm.y}; };
#define NB_ACC4 "+v"(ax[0]), "+v"(ax[1]), "+v"(ax[2]), "+v"(ax[3]), "+v"(ay[0]), "+v"(ay[1]), "+v"(ay[2]), "+v"(ay[3]), \
                "+v"(az[0]), "+v"(az[1]), "+v"(az[2]), "+v"(az[3])
#define NB_ACC2 "+v"(ax[0]), "+v"(ax[1]), "+v"(ay[0]), "+v"(ay[1]), "+v"(az[0]), "+v"(az[1])
#define NB_OCT(q) (q).p[0], (q).p[1], (q).p[2], (q).p[3], (q).p[4], (q).p[5], (q).p[6], (q).p[7]
        auto wait8 = [&](Oct& q) {
            if constexpr (NG == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.p[0]), "+s"(q.p[1]), "+s"(q.p[2]), "+s"(q.p[3]), "+s"(q.p[4]), "+s"(q.p[5]), "+s"(q.p[6]), "+s"(q.p[7]), NB_ACC4 : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.p[0]), "+s"(q.p[1]), "+s"(q.p[2]), "+s"(q.p[3]), "+s"(q.p[4]), "+s"(q.p[5]), "+s"(q.p[6]), "+s"(q.p[7]), NB_ACC2 : : "memory");
        };
#define NB_LOAD8(o) "s_load_dwordx2 %0, %" #o ", 0x0\n\ts_load_dwordx2 %1, %" #o ", 0x8\n\ts_load_dwordx2 %2, %" #o ", 0x10\n\t" \
                    "s_load_dwordx2 %3, %" #o ", 0x18\n\ts_load_dwordx2 %4, %" #o ", 0x20\n\ts_load_dwordx2 %5, %" #o ", 0x28\n\t" \
                    "s_load_dwordx2 %6, %" #o ", 0x30\n\ts_load_dwordx2 %7, %" #o ", 0x38"
        auto request8 = [&](Oct& q, const float4* p) {
            if constexpr (NG == 4) asm volatile(NB_LOAD8(20) : "=&s"(q.p[0]), "=&s"(q.p[1]), "=&s"(q.p[2]), "=&s"(q.p[3]), "=&s"(q.p[4]), "=&s"(q.p[5]), "=&s"(q.p[6]), "=&s"(q.p[7]), NB_ACC4 : "s"(p) : "memory");
            else asm volatile(NB_LOAD8(14) : "=&s"(q.p[0]), "=&s"(q.p[1]), "=&s"(q.p[2]), "=&s"(q.p[3]), "=&s"(q.p[4]), "=&s"(q.p[5]), "=&s"(q.p[6]), "=&s"(q.p[7]), NB_ACC2 : "s"(p) : "memory");
        };
#undef NB_LOAD8
#undef NB_OCT
#undef NB_ACC2
#undef NB_ACC4
        Oct A, B;
        if (nb8) request8(A, pj);
        load_i_rows();
        if (nb8) {
            for (uint32_t it = 0; it < nb8; ++it) {
                wait8(A);
                request8(B, pj + 4);
                eval4(f4p(A.p[0], A.p[1]), f4p(A.p[2], A.p[3]), f4p(A.p[4], A.p[5]), f4p(A.p[6], A.p[7]));
                wait8(B);
                pj += 8;
                if (it + 1 < nb8) request8(A, pj);
                eval4(f4p(B.p[0], B.p[1]), f4p(B.p[2], B.p[3]), f4p(B.p[4], B.p[5]), f4p(B.p[6], B.p[7]));
            }
            j += nb8 * 8;
        }
    } else {
        Quad A, B;
        if (nb8) request(A, pj);
        load_i_rows();
        // branch-free body: the request after the wave's last 8 bodies re-reads its last 4 (a scalar select on the
        // pointer, never past the range) -- with a conditional request the second eval sat in its own basic block and
        // its SGPR operands were copied to VGPRs at the block boundary
        for (uint32_t it = 0; it < nb8; ++it) {
            wait_for(A);
            request(B, pj + 4);
            eval4(f4(A.q0), f4(A.q1), f4(A.q2), f4(A.q3));
            wait_for(B);
            request(A, it + 1 < nb8 ? pj + 8 : pj + 4);
            pj += 8;
            eval4(f4(B.q0), f4(B.q1), f4(B.q2), f4(B.q3));
        }
        if (nb8) wait_for(A);          // the spare request has landed (in dead registers) before anything else counts lgkm
        j += nb8 * 8;
    }
    for (; j < j1; ++j)      // < 8 bodies left (only when n is not a multiple of 8): one at a time
        eval4(bodies[j], float4{0, 0, 0, 0}, float4{0, 0, 0, 0}, float4{0, 0, 0, 0});

    if constexpr (WS == 4) {
        // waves 1..3 hand their sums to wave 0 through LDS; added in wave order
        __shared__ float red[3][3 * IPL][64];
        if (wv > 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                red[wv - 1][6 * g + 0][lane] = ax[g].x; red[wv - 1][6 * g + 1][lane] = ax[g].y;
                red[wv - 1][6 * g + 2][lane] = ay[g].x; red[wv - 1][6 * g + 3][lane] = ay[g].y;
                red[wv - 1][6 * g + 4][lane] = az[g].x; red[wv - 1][6 * g + 5][lane] = az[g].y;
            }
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                ax[g].x += red[w][6 * g + 0][lane]; ax[g].y += red[w][6 * g + 1][lane];
                ay[g].x += red[w][6 * g + 2][lane]; ay[g].y += red[w][6 * g + 3][lane];
                az[g].x += red[w][6 * g + 4][lane]; az[g].y += red[w][6 * g + 5][lane];
            }
        }
    }

    float4* out = partial + (size_t)by * i_count;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint32_t il0 = bxi * IPB + (2 * g) * LANES + lane;
        const uint32_t il1 = il0 + LANES;
        if (il0 < i_count) out[il0] = float4{ax[g].x, ay[g].x, az[g].x, 0};
        if (il1 < i_count) out[il1] = float4{ax[g].y, ay[g].y, az[g].y, 0};
    }
}

}  // namespace nb
