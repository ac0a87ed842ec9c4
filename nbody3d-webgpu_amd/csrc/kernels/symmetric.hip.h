// kernels/symmetric.hip.h -- the symmetric force pass (each unordered pair once): nb_force_sym / nb_force_symw / nb_force_symw64, the rank form's reduction kernels, nb_integrate_sym*.
// Part of nb_kernels.hip.h (include that, not this file).
#pragma once

namespace nb {

// ---- symmetric force pass (Newton's third law inside a wave) -----------------------------------
// Every kernel above evaluates each ORDERED pair on its own: 12 v_pk + 2 v_rsq_f32 per two pairs, the instruction-mix
// ceiling of 62.5 % of the fp32 vector rate.  r = x_j - x_i, r^2, the cube and the reciprocal square root are the same
// numbers for (i, j) and (j, i) (IEEE subtraction is exactly antisymmetric), so this pass computes them ONCE per
// unordered pair and accumulates both accelerations -- the per-pair products (G m_j) inv r and (G m_i) inv (-r) are
// bit for bit the reference's (nbody3d.js:233-236); only the order of the additions differs:
//   * a lane keeps 8 RESIDENT bodies (4 packed groups, as nb_force_pk_sgpr<4,..>); J = 2 TRAVELING bodies per lane --
//     a chunk of 128 bodies per wave -- rotate through the 64 lanes with v_mov_b32_dpp wave_ror:1 (full rate on gfx950:
//     tools/experiments/ubench6.hip), their six packed sums traveling with them; after 64 steps every resident of the
//     wave has met every traveler of the chunk and the travelers are back in their home lanes;
//   * per (traveler, packed group): 3 v_pk_add, 3 v_pk_fma, 2 v_pk_mul, 2 v_rsq_f32, v_pk_mul + 3 v_pk_fma for the
//     resident side, v_pk_mul + 3 v_pk_fma (negated) for the traveler side = 16 packed + 2 transcendental per FOUR
//     interactions, + 10 v_mov_b32_dpp per traveler and step: 90 issue slots per 16 interactions against 128 --
//     measured 74.7 % of the fp32 roofline for the bare loop (profiles/r03/ubench6_*.txt) against 60 %;
//   * coverage: the bodies form nsb SUPER-BLOCKS of S = 512*WS rows (one 512-row block per wave of a workgroup).
//     Workgroup (g, q) keeps super-block g resident and sweeps segment q (of Q nearly equal ones) of g's chunk list: the chunks of the H =
//     (nsb-1)/2 super-blocks that follow g on the ring (plus the antipodal one for g < nsb/2 when nsb is even) --
//     every unordered pair of different super-blocks exactly once -- and then the chunks of super-block g ITSELF in
//     resident-only mode (traveler sums discarded: every ordered pair inside g once; the self term is exactly 0);
//   * sums: a wave's resident sums go to layer (r_layer0 + q); the traveler sums of a chunk are added over the WS waves
//     in wave order through LDS (one barrier per chunk, double buffered) and go to layer (t_layer0 + ring distance - 1).
//     nb_integrate_sym adds a body's layers in ascending order: deterministic, no float atomics.  A partial row is
//     12 bytes (x, y, z: one global_store_dwordx3 per lane): the layers are the pass's memory traffic.
// Rows [n, np) of `bodies` are zero-mass bodies at the origin (np = nsb * S).
template <typename T> struct SymRowT { T x, y, z; };   // a partial row: 12 bytes in f32 (an ext_vector_type(3) would be padded to 16), 24 in f64
using SymRow = SymRowT<float>;
// struct SymPlan: nb_plan.h (the host's planner fills it)

// rotation steps after which the register sums of the residents move on to their second level (nb_force_symw, nb_force_symw_rank)
constexpr uint32_t kFlushSteps = 4096;

// The register sums of a wave's residents move on to their second level `hi` (the wave's own [6 * NG][64] floats of LDS: rows 6c .. 6c+2
// the x / y / z sums of resident 2c, rows 6c+3 .. 6c+5 of resident 2c+1) and start again from zero.  `flushed` (wave-uniform): the
// second level already holds sums -- one branch around two straight-line forms.
template <int NG>
__device__ __forceinline__ void flush_resident_sums(float (*hi)[64], const int lane, const bool flushed, nb_f2 (&ax)[NG], nb_f2 (&ay)[NG], nb_f2 (&az)[NG])
{
    if (flushed) {
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            hi[6 * c + 0][lane] += ax[c].x; hi[6 * c + 1][lane] += ay[c].x; hi[6 * c + 2][lane] += az[c].x;
            hi[6 * c + 3][lane] += ax[c].y; hi[6 * c + 4][lane] += ay[c].y; hi[6 * c + 5][lane] += az[c].y;
        }
    } else {
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            hi[6 * c + 0][lane] = ax[c].x; hi[6 * c + 1][lane] = ay[c].x; hi[6 * c + 2][lane] = az[c].x;
            hi[6 * c + 3][lane] = ax[c].y; hi[6 * c + 4][lane] = ay[c].y; hi[6 * c + 5][lane] = az[c].y;
        }
    }
#pragma unroll
    for (int c = 0; c < NG; ++c) { ax[c] = nb_f2{0, 0}; ay[c] = nb_f2{0, 0}; az[c] = nb_f2{0, 0}; }
}

__device__ __forceinline__ float wave_rot1(float v)
{
    const int iv = __builtin_bit_cast(int, v);      // old = src: every lane is written, no init move
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x13C /* wave_ror:1 */, 0xF, 0xF, false));
}

// NG packed groups = 2*NG residents per lane (NG = 4: 128 VGPRs, 4 waves per SIMD; NG = 8: the rotation is amortised over
// twice the pairs -- one wave per SIMD already issues this loop at ~90 % of its rate, so 2 waves per SIMD are enough);
// WS waves per workgroup, each with its own 128*NG resident rows; J travelers per lane.
template <int WS, int NG, int J>
__global__ __launch_bounds__(64 * WS) __attribute__((amdgpu_waves_per_eu(NG > 4 ? 2 : 4, NG > 4 ? 2 : 4)))
void nb_force_sym(const float4* __restrict__ bodies, SymRow* __restrict__ partial, const SymPlan pl, const uint32_t n, const float eps2)
{
    constexpr uint32_t RB = 128u * NG;         // resident rows per wave
    constexpr uint32_t S = RB * WS;            // rows per super-block
    constexpr uint32_t CH = 64u * J;           // travelers per chunk
    constexpr uint32_t CPS = S / CH;           // chunks per super-block
    __shared__ float red[WS > 1 ? 2 : 1][WS > 1 ? WS : 1][3 * J][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const uint32_t g = blockIdx.x / pl.q, q = blockIdx.x % pl.q;
    const uint32_t ring = (pl.H + (g < pl.n_hi ? 1u : 0u)) * CPS;     // symmetric chunks of g; CPS resident-only chunks follow
    const uint32_t total = ring + CPS;
    // segment q of Q: chunk ranges of (nearly) equal length, [q * total / Q, (q + 1) * total / Q)
    const uint32_t c0 = (uint32_t)(((uint64_t)q * total) / pl.q), c1 = (uint32_t)(((uint64_t)(q + 1) * total) / pl.q);

    nb_f2 xi[NG], yi[NG], zi[NG], mi[NG], ax[NG], ay[NG], az[NG];
    {
        const float4* rb = bodies + (size_t)g * S + w * RB + lane;
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            const float4 b0 = ld4(rb + (2 * c) * 64), b1 = ld4(rb + (2 * c + 1) * 64);
            xi[c] = nb_f2{b0.x, b1.x}; yi[c] = nb_f2{b0.y, b1.y}; zi[c] = nb_f2{b0.z, b1.z}; mi[c] = nb_f2{b0.w, b1.w};
            ax[c] = nb_f2{0, 0}; ay[c] = nb_f2{0, 0}; az[c] = nb_f2{0, 0};
        }
    }
    const nb_f2 e2 = nb_f2{eps2, eps2};

    uint32_t done = 0;                                               // symmetric chunks processed: alternates the LDS buffer
    for (uint32_t k = c0; k < c1; ++k) {
        const bool sym = k < ring;                                   // wave-uniform
        const uint32_t d = k / CPS;                                  // ring distance - 1 (symmetric chunks)
        uint32_t tb = g + 1 + d;
        if (tb >= pl.nsb) tb -= pl.nsb;
        const uint32_t tstart = sym ? tb * S + (k % CPS) * CH : g * S + (k - ring) * CH;
        if (tstart >= n) continue;       // a chunk of padding rows only (zero mass): exerts nothing, and nobody reads its sums
        float tx[J], ty[J], tz[J], tm[J];
        nb_f2 bx[J], by[J], bz[J];
#pragma unroll
        for (int u = 0; u < J; ++u) {
            const float4 t = ld4(bodies + tstart + u * 64 + lane);
            tx[u] = t.x; ty[u] = t.y; tz[u] = t.z; tm[u] = t.w;
            bx[u] = nb_f2{0, 0}; by[u] = nb_f2{0, 0}; bz[u] = nb_f2{0, 0};
        }
        for (int st = 0; st < 64; ++st) {
#pragma unroll
            for (int u = 0; u < J; ++u) {
                const nb_f2 px = nb_f2{tx[u], tx[u]}, py = nb_f2{ty[u], ty[u]}, pz = nb_f2{tz[u], tz[u]}, pm = nb_f2{tm[u], tm[u]};
                // stage-major over groups of four (as every packed kernel here); NG = 8 runs two such groups
#pragma unroll
                for (int c0g = 0; c0g < NG; c0g += 4) {
                    nb_f2 dx[4], dy[4], dz[4], d2[4], r[4], si[4], sj[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) dx[c] = px - xi[c0g + c];                                   // :233
#pragma unroll
                    for (int c = 0; c < 4; ++c) dy[c] = py - yi[c0g + c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) dz[c] = pz - zi[c0g + c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) d2[c] = __builtin_elementwise_fma(dx[c], dx[c], e2);         // :234
#pragma unroll
                    for (int c = 0; c < 4; ++c) d2[c] = __builtin_elementwise_fma(dy[c], dy[c], d2[c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) d2[c] = __builtin_elementwise_fma(dz[c], dz[c], d2[c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = d2[c] * d2[c];                                       // :235
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = r[c] * d2[c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = nb_f2{nb_rsq(r[c].x), nb_rsq(r[c].y)};
#pragma unroll
                    for (int c = 0; c < 4; ++c) si[c] = pm * r[c];                // (G m_t) inv: resident side, :236
#pragma unroll
                    for (int c = 0; c < 4; ++c) sj[c] = mi[c0g + c] * r[c];       // (G m_i) inv: traveler side
#pragma unroll
                    for (int c = 0; c < 4; ++c) ax[c0g + c] = __builtin_elementwise_fma(si[c], dx[c], ax[c0g + c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) ay[c0g + c] = __builtin_elementwise_fma(si[c], dy[c], ay[c0g + c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) az[c0g + c] = __builtin_elementwise_fma(si[c], dz[c], az[c0g + c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) bx[u] = __builtin_elementwise_fma(-sj[c], dx[c], bx[u]);   // x_i - x_t = -(x_t - x_i), exactly
#pragma unroll
                    for (int c = 0; c < 4; ++c) by[u] = __builtin_elementwise_fma(-sj[c], dy[c], by[u]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) bz[u] = __builtin_elementwise_fma(-sj[c], dz[c], bz[u]);
                }
            }
            // the travelers and their sums move on by one lane
#pragma unroll
            for (int u = 0; u < J; ++u) {
                tx[u] = wave_rot1(tx[u]); ty[u] = wave_rot1(ty[u]); tz[u] = wave_rot1(tz[u]); tm[u] = wave_rot1(tm[u]);
                bx[u] = nb_f2{wave_rot1(bx[u].x), wave_rot1(bx[u].y)};
                by[u] = nb_f2{wave_rot1(by[u].x), wave_rot1(by[u].y)};
                bz[u] = nb_f2{wave_rot1(bz[u].x), wave_rot1(bz[u].y)};
            }
        }
        if (sym) {
            SymRow* out = partial + (size_t)(pl.t_layer0 + d) * pl.np + tstart + lane;
            if constexpr (WS == 1) {
#pragma unroll
                for (int u = 0; u < J; ++u) out[u * 64] = SymRow{bx[u].x + bx[u].y, by[u].x + by[u].y, bz[u].x + bz[u].y};
            } else {
                // traveler sums of the chunk: added over the workgroup's waves in wave order, stored by one of them.
                // One barrier per chunk: the buffers alternate, and the wave that reads buffer b passes the NEXT
                // barrier only after its reads, which is before anybody writes b again.
                const int buf = done++ & 1;
#pragma unroll
                for (int u = 0; u < J; ++u) {
                    red[buf][w][3 * u + 0][lane] = bx[u].x + bx[u].y;
                    red[buf][w][3 * u + 1][lane] = by[u].x + by[u].y;
                    red[buf][w][3 * u + 2][lane] = bz[u].x + bz[u].y;
                }
                __syncthreads();
                if (w == done % WS) {
#pragma unroll
                    for (int u = 0; u < J; ++u) {
                        float sx = red[buf][0][3 * u + 0][lane], sy = red[buf][0][3 * u + 1][lane], sz = red[buf][0][3 * u + 2][lane];
#pragma unroll
                        for (int ww = 1; ww < WS; ++ww) { sx += red[buf][ww][3 * u + 0][lane]; sy += red[buf][ww][3 * u + 1][lane]; sz += red[buf][ww][3 * u + 2][lane]; }
                        out[u * 64] = SymRow{sx, sy, sz};
                    }
                }
            }
        }
    }
    // resident sums of this segment
    SymRow* out = partial + (size_t)(pl.r_layer0 + q) * pl.np + (size_t)g * S + w * RB + lane;
#pragma unroll
    for (int c = 0; c < NG; ++c) {
        out[(2 * c) * 64] = SymRow{ax[c].x, ay[c].x, az[c].x};
        out[(2 * c + 1) * 64] = SymRow{ax[c].y, ay[c].y, az[c].y};
    }
}

// The same pass with the WAVE as the unit of work (no LDS in the sweeps, one barrier at the very end): a super-block is one wave's 128*NG residents, and
// the chunk lists of all super-blocks, laid end to end (L chunk-sweeps), are cut into W contiguous ranges of floor/ceil(L/W)
// sweeps -- one per wave, W a multiple of the chip's SIMD count -- so every SIMD gets the same work to within ONE sweep at
// any N (the workgroup form above needs nsb * Q to land on a multiple of the CU count).  A wave whose range crosses into the
// next super-block stores its resident sums in g's last resident layer, reloads its residents and goes on; the sums of the
// super-block a range ENDS in are added up over the workgroup's four waves in LDS and go to the resident layer the wave's table
// record names (table `gtab`, built by the host: first wave and resident layer count per super-block, then {first unit, end,
// layer, spill row} per wave).
// struct SymWPlan: nb_plan.h
//
// The plan reaches the force kernels as SIX scalar parameters behind the four pointers, the table pointer first: the first 14 dwords of
// the kernel arguments are preloaded into SGPRs by the hardware (-amdgpu-kernarg-preload-count; a struct passed by value ends the
// preloaded part), so the wave's table record -- the first thing a wave needs -- is requested in the wave's first instructions instead
// of behind a round trip for the arguments (-0.4 .. -0.8 % per step from N = 7,000 to 13,000, nothing above:
// profiles/r05/ab_arguments_preloaded_prev_vs_tree.txt -- the argument block is a cheap read).  The other
// plan words follow from these (nb_plan.cpp::lay_out_symw): the ring's half width, the list lengths, the padded row count.
struct SymWK { uint32_t np, nsb, W, total_hi, total_lo, n_hi, zc, ups, r_layer0, t_layer0; };
#define SYMW_PLAN_PARAMS const uint32_t W_, const uint32_t ups_, const uint32_t nsb_, const uint32_t zc_, const uint32_t r_layer0_, const uint32_t t_layer0_
// The wave's table record ({first unit, end, resident layer, spill row}: 16 bytes at `rec_at`) and the kernel's arguments behind the
// preloaded 14 dwords (TAIL dwords at byte 56 of the argument block: the softening, in f64 also G), requested TOGETHER -- left to the
// compiler the arguments' load sinks behind the wait for the record, a second cold round trip in front of the residents' loads.
typedef uint32_t nb_u4 __attribute__((ext_vector_type(4)));
template <int TAIL>
__device__ __forceinline__ void symw_record_and_tail(const uint32_t* rec_at, nb_u4& rec, uint32_t (&tail)[TAIL])
{
    static_assert(TAIL == 1 || TAIL == 4, "f32: eps2; f64: G, eps2");
    constexpr int kTailOffset = 4 * 8 + 6 * 4;       // four pointers and the six words of SYMW_PLAN_PARAMS in front of it
    if constexpr (TAIL == 1) {
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(rec), "=&s"(tail[0]) : "s"(rec_at), "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(kTailOffset) : "memory");
    } else {
        nb_u4 t;
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(rec), "=&s"(t) : "s"(rec_at), "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(kTailOffset) : "memory");
        tail[0] = t.x; tail[1] = t.y; tail[2] = t.z; tail[3] = t.w;
    }
}

__device__ __forceinline__ SymWK symw_plan_words(uint32_t S, uint32_t cps, uint32_t W, uint32_t ups, uint32_t nsb, uint32_t zc, uint32_t r_layer0, uint32_t t_layer0)
{
    const uint32_t H = (nsb - 1u) >> 1, n_hi = (nsb & 1u) ? 0u : nsb >> 1;
    const uint32_t total_lo = (H + 1u) * cps + zc, total_hi = total_lo + (n_hi ? cps : 0u);
    return SymWK{(nsb + (zc ? 1u : 0u)) * S, nsb, W, total_hi, total_lo, n_hi, zc, ups, r_layer0, t_layer0};
}

template <int NG, int J>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NG >= 4 ? 2 : 4, NG > 4 ? 2 : (NG < 4 ? 8 : 4))))
void nb_force_symw(const uint32_t* __restrict__ gtab, const float4* __restrict__ bodies, SymRow* __restrict__ partial, SymRow* __restrict__ spill,
                   SYMW_PLAN_PARAMS, const float eps2_arg /* read by hand, with the table record: symw_record_and_tail */,
                   uint32_t* __restrict__ queue, const uint32_t npieces, const uint32_t pieces_off)
{
    constexpr uint32_t S = 128u * NG;          // rows per super-block = one wave's residents
    constexpr int GW = NG < 4 ? NG : 4;        // packed groups evaluated stage-major together
    constexpr uint32_t CH = 64u * J;           // travelers per chunk
    constexpr uint32_t CPS = S / CH;           // chunks per super-block
    const SymWK pl = symw_plan_words(S, CPS, W_, ups_, nsb_, zc_, r_layer0_, t_layer0_);
    const int lane = threadIdx.x & 63, wi = threadIdx.x >> 6;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wi);     // the four waves of a workgroup sweep independently and meet once, at the end
    const bool active = w < pl.W;
    NB_STAMP(0);
    __shared__ float red[4][6 * NG][64];       // the waves' last resident sums, added up per super-block before they leave the CU
    __shared__ uint32_t fin[4];
    // The wave's range, in UNITS of 64 / ups rotation steps (ups units per chunk-sweep), relative to the handle's part of the list.
    // With ups > 1 a sweep may be shared by consecutive waves: each runs its own rotation steps [s0, s1) of it, starting from
    // travelers loaded s0 lanes ahead (wave_ror:1 moves a traveler from lane l to lane l + 1, so after s steps lane l holds the
    // traveler that started in lane l - s).
    // The range is a record of the planner's table (four words per wave behind the {first wave, resident layers} pair of every block of S
    // rows): consecutive ranges need not belong to consecutive waves (nb_plan.cpp::lay_out_symw pairs an older with a younger wave).
    // A ragged N leaves a SHORT block Z of pl.zc real chunks behind the pl.nsb whole super-blocks of the ring: every super-block sweeps
    // Z's chunks after its ring sweeps (both sides: the traveler sums go to z-row g * zc + c of the spill buffer), and Z -- "super-block"
    // pl.nsb, last in the list -- sweeps only its own chunks (nb_plan.cpp::lay_out_symw).
    const uint32_t ups = pl.ups, ush = (uint32_t)__builtin_ctz(ups), ustep = 64u >> ush,      // (ups is a power of two)
                   tab1 = 2u * (pl.np / S);
    nb_u4 rec;                                 // {first unit, end, resident layer, spill row}: one scalar load
    uint32_t tail[1];
    symw_record_and_tail(gtab + tab1 + 4u * (active ? w : 0u), rec, tail);
    const float eps2 = __builtin_bit_cast(float, tail[0]);
    const nb_f2 e2 = nb_f2{eps2, eps2};
    const uint32_t first_lo = pl.n_hi * pl.total_hi, first_z = first_lo + (pl.nsb - pl.n_hi) * pl.total_lo;
    const uint32_t slot = rec.w;               // the wave's spill row (it has at most one: the sweep its range starts inside)
    NB_STAMP_LIGHT(1);
    bool first_part = true;                    // (diagnostic stamps only; dead in the product build)

    // The units [u, uend) of the list.  piece_layer == ~0u: the wave's own range -- the sums of the super-block it ends in are left in
    // `red` for the workgroup's meeting below, and that super-block is returned; else a piece drawn from the queue (whole sweeps inside ONE
    // super-block's list), whose resident sums go straight to the resident layer the planner gave it.
    auto run = [&](uint32_t u, const uint32_t uend, const uint32_t piece_layer) -> uint32_t {
        uint32_t gfin = ~0u;                       // the super-block the range ends in
        while (u < uend) {
            // which super-block's list the unit lies in, and where
            const uint32_t p = u >> ush;                                  // the sweep
            uint32_t g, k, total;
            if (p < first_lo) { g = p / pl.total_hi; k = p - g * pl.total_hi; total = pl.total_hi; }
            else if (p < first_z) { const uint32_t r = p - first_lo; g = pl.n_hi + r / pl.total_lo; k = r - (g - pl.n_hi) * pl.total_lo; total = pl.total_lo; }
            else { g = pl.nsb; k = p - first_z; total = pl.zc; }         // Z over its own chunks
            const uint32_t ring = g < pl.nsb ? total - CPS - pl.zc : 0u; // sweeps over the chunks of other super-blocks
            const uint32_t both_end = g < pl.nsb ? ring + pl.zc : 0u;    // ... then over Z's: both keep traveler sums; own chunks follow
            uint32_t ug_end = (p - k + total) * ups;                     // end of g's list, in units
            if (ug_end > uend) ug_end = uend;

            nb_f2 xi[NG], yi[NG], zi[NG], mi[NG], ax[NG], ay[NG], az[NG];
            {
                const float4* rb = bodies + (size_t)g * S + lane;
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    const float4 b0 = ld4(rb + (2 * c) * 64), b1 = ld4(rb + (2 * c + 1) * 64);
                    xi[c] = nb_f2{b0.x, b1.x}; yi[c] = nb_f2{b0.y, b1.y}; zi[c] = nb_f2{b0.z, b1.z}; mi[c] = nb_f2{b0.w, b1.w};
                    ax[c] = nb_f2{0, 0}; ay[c] = nb_f2{0, 0}; az[c] = nb_f2{0, 0};
                }
            }
            // Two-level resident sums: a wave that stays with one super-block for thousands of sweeps (N = 2,000,000: one wave per
            // super-block, 15,632 sweeps = 1,000,448 terms per resident) would add terms of ~a / N to a running sum of ~a in ONE
            // binary32 register -- the stagnation that puts the reference's own ascending-j loop at 1e-5 .. 5e-4 there
            // (tests/golden/large_n_row_spread.json; this pass sat at 2e-5).  Every kFlushSteps rotation steps the register sums
            // move on to a second level kept in the wave's own part of `red` (LDS: 48 reads, adds and writes per 4,096 steps), so
            // no accumulator takes more than 4,096 * J terms in sequence.  A range that never gets that far (every N below ~400,000
            // on 256 CUs) never touches LDS here and adds exactly what it added before.
            uint32_t since = 0;
            bool flushed = false;
            while (u < ug_end) {
                // the wave's steps [s0, s1) of sweep k
                const uint32_t q0 = u & (ups - 1u);
                uint32_t nun = ups - q0;
                if (nun > ug_end - u) nun = ug_end - u;
                const uint32_t s0 = q0 * ustep, s1 = s0 + nun * ustep;
                u += nun;
                if (since >= kFlushSteps) {
                    flush_resident_sums<NG>(red[wi], lane, flushed, ax, ay, az);
                    flushed = true;
                    since = 0;
                }
                since += s1 - s0;
                const bool sym = k < both_end, zsweep = k >= ring && sym;
                const uint32_t d = k / CPS;                              // ring distance - 1 (ring sweeps)
                uint32_t tb = g + 1 + d;
                if (tb >= pl.nsb) tb -= pl.nsb;
                const uint32_t tstart = k < ring ? tb * S + (k % CPS) * CH : zsweep ? pl.nsb * S + (k - ring) * CH : g * S + (k - both_end) * CH;
                const uint32_t zrow = g * pl.zc + (k - ring);            // (z sweeps)
                ++k;
                // (Tried in round 5 and dropped: requesting the travelers of sweep k + 1 before the rotation steps of sweep k -- the wait
                // moved behind the loop, in front of the stores.  0.3-1 % SLOWER from N = 10,000 to 65,536 at one and two waves per SIMD,
                // profiles/r05/ab_traveler_prefetch_head_vs_tree.txt: this round trip is not what the short lists wait for.)
                float tx[J], ty[J], tz[J], tm[J];
                nb_f2 bx[J], by[J], bz[J];
                const uint32_t src = ((uint32_t)lane - s0) & 63u;        // the traveler this lane holds after s0 rotation steps
#pragma unroll
                for (int uu = 0; uu < J; ++uu) {
                    const float4 t = ld4(bodies + tstart + uu * 64 + src);
                    tx[uu] = t.x; ty[uu] = t.y; tz[uu] = t.z; tm[uu] = t.w;
                    bx[uu] = nb_f2{0, 0}; by[uu] = nb_f2{0, 0}; bz[uu] = nb_f2{0, 0};
                }
                // (the loop head is 32-byte aligned by -falign-loops=32: a packed instruction that straddles an 8-byte boundary issues
                // more slowly -- 12 % on this loop at one wave per SIMD, profiles/r04/README.md)
                // Two forms of the loop: a sweep over one of the super-block's OWN chunks needs no traveler sums (each of its pairs is met
                // from both sides), so it drops the 4 packed instructions per group and the 6 rotations that keep them: 116 instead of
                // 154 issue slots per step with 16 residents -- 0.88 of the time at one or two waves per SIMD (the planner counts 7/8).
                // (Tried in round 5 and dropped -- commit 4c416e1 has the code: a THIRD, triangular form for the own chunks, chunk c against the
                // resident rows r > c from both sides and row c resident-only, one instantiation per first packed group, traveler sums folded
                // into the accumulators of row c: every pair inside a super-block once, 21 % fewer instructions over the 16 own chunks.
                // Correct (216 GPU tests), but the forms with fewer than four packed groups have fewer than four independent chains and wait
                // on their own results: 3-4 % SLOWER per step at N = 13,000 .. 20,000 with wave ranges weighted by instruction count,
                // level with this form after re-weighting -- profiles/r05/own_chunk_triangular_*.txt.)
                auto steps = [&](auto both) {
                    constexpr bool BOTH = decltype(both)::value;
                    for (uint32_t st = s0; st < s1; ++st) {
#pragma unroll
                        for (int uu = 0; uu < J; ++uu) {
                            const nb_f2 px = nb_f2{tx[uu], tx[uu]}, py = nb_f2{ty[uu], ty[uu]}, pz = nb_f2{tz[uu], tz[uu]}, pm = nb_f2{tm[uu], tm[uu]};
#pragma unroll
                            for (int c0g = 0; c0g < NG; c0g += GW) {         // stage-major over groups of (up to) four
                                nb_f2 dx[GW], dy[GW], dz[GW], d2[GW], r[GW], si[GW], sj[GW];
#pragma unroll
                                for (int c = 0; c < GW; ++c) dx[c] = px - xi[c0g + c];                                   // :233
#pragma unroll
                                for (int c = 0; c < GW; ++c) dy[c] = py - yi[c0g + c];
#pragma unroll
                                for (int c = 0; c < GW; ++c) dz[c] = pz - zi[c0g + c];
#pragma unroll
                                for (int c = 0; c < GW; ++c) d2[c] = __builtin_elementwise_fma(dx[c], dx[c], e2);         // :234
#pragma unroll
                                for (int c = 0; c < GW; ++c) d2[c] = __builtin_elementwise_fma(dy[c], dy[c], d2[c]);
#pragma unroll
                                for (int c = 0; c < GW; ++c) d2[c] = __builtin_elementwise_fma(dz[c], dz[c], d2[c]);
#pragma unroll
                                for (int c = 0; c < GW; ++c) r[c] = d2[c] * d2[c];                                       // :235
#pragma unroll
                                for (int c = 0; c < GW; ++c) r[c] = r[c] * d2[c];
#pragma unroll
                                for (int c = 0; c < GW; ++c) r[c] = nb_f2{nb_rsq(r[c].x), nb_rsq(r[c].y)};
#pragma unroll
                                for (int c = 0; c < GW; ++c) si[c] = pm * r[c];                // (G m_t) inv: resident side, :236
                                if constexpr (BOTH) {
#pragma unroll
                                    for (int c = 0; c < GW; ++c) sj[c] = mi[c0g + c] * r[c];   // (G m_i) inv: traveler side
                                }
#pragma unroll
                                for (int c = 0; c < GW; ++c) ax[c0g + c] = __builtin_elementwise_fma(si[c], dx[c], ax[c0g + c]);
#pragma unroll
                                for (int c = 0; c < GW; ++c) ay[c0g + c] = __builtin_elementwise_fma(si[c], dy[c], ay[c0g + c]);
#pragma unroll
                                for (int c = 0; c < GW; ++c) az[c0g + c] = __builtin_elementwise_fma(si[c], dz[c], az[c0g + c]);
                                if constexpr (BOTH) {
#pragma unroll
                                    for (int c = 0; c < GW; ++c) bx[uu] = __builtin_elementwise_fma(-sj[c], dx[c], bx[uu]);   // x_i - x_t = -(x_t - x_i), exactly
#pragma unroll
                                    for (int c = 0; c < GW; ++c) by[uu] = __builtin_elementwise_fma(-sj[c], dy[c], by[uu]);
#pragma unroll
                                    for (int c = 0; c < GW; ++c) bz[uu] = __builtin_elementwise_fma(-sj[c], dz[c], bz[uu]);
                                }
                            }
                        }
#pragma unroll
                        for (int uu = 0; uu < J; ++uu) {                     // the travelers and their sums move on by one lane
                            tx[uu] = wave_rot1(tx[uu]); ty[uu] = wave_rot1(ty[uu]); tz[uu] = wave_rot1(tz[uu]); tm[uu] = wave_rot1(tm[uu]);
                            if constexpr (BOTH) {
                                bx[uu] = nb_f2{wave_rot1(bx[uu].x), wave_rot1(bx[uu].y)};
                                by[uu] = nb_f2{wave_rot1(by[uu].x), wave_rot1(by[uu].y)};
                                bz[uu] = nb_f2{wave_rot1(bz[uu].x), wave_rot1(bz[uu].y)};
                            }
                        }
                    }
                };
                if (first_part) NB_STAMP(2);       // (drains the loads first: residents + first travelers landed)
                if (sym) steps(std::true_type{}); else steps(std::false_type{});
                if (first_part) { NB_STAMP_LIGHT(3); first_part = false; }
                if (sym) {
                    // the sums of steps [s0, s1) sit s1 lanes past their travelers' home lanes.  The part that starts the sweep owns the
                    // sweep's traveler layer; any later part goes to the wave's own spill row (K2 adds it: the spill rows of a chunk are consecutive)
                    SymRow* out = (s0 != 0 ? spill + (size_t)slot * CH : zsweep ? spill + (size_t)zrow * CH : partial + (size_t)(pl.t_layer0 + d) * pl.np + tstart)
                                  + (((uint32_t)lane - s1) & 63u);
#pragma unroll
                    for (int uu = 0; uu < J; ++uu) out[uu * 64] = SymRow{bx[uu].x + bx[uu].y, by[uu].x + by[uu].y, bz[uu].x + bz[uu].y};
                }
            }
            if (u >= uend && piece_layer != ~0u) {
                SymRow* out = partial + (size_t)(pl.r_layer0 + piece_layer) * pl.np + (size_t)g * S + lane;
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    if (flushed) {         // (a long piece of a very large system: second level + what the registers hold)
                        ax[c].x += red[wi][6 * c + 0][lane]; ay[c].x += red[wi][6 * c + 1][lane]; az[c].x += red[wi][6 * c + 2][lane];
                        ax[c].y += red[wi][6 * c + 3][lane]; ay[c].y += red[wi][6 * c + 4][lane]; az[c].y += red[wi][6 * c + 5][lane];
                    }
                    out[(2 * c) * 64] = SymRow{ax[c].x, ay[c].x, az[c].x};
                    out[(2 * c + 1) * 64] = SymRow{ax[c].y, ay[c].y, az[c].y};
                }
                return g;
            }
            if (u >= uend) {
                // the range ends here: the sums meet those of the workgroup's other waves in LDS (below)
                gfin = g;
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    if (flushed) {         // second level + what the registers hold
                        ax[c].x += red[wi][6 * c + 0][lane]; ay[c].x += red[wi][6 * c + 1][lane]; az[c].x += red[wi][6 * c + 2][lane];
                        ax[c].y += red[wi][6 * c + 3][lane]; ay[c].y += red[wi][6 * c + 4][lane]; az[c].y += red[wi][6 * c + 5][lane];
                    }
                    red[wi][6 * c + 0][lane] = ax[c].x; red[wi][6 * c + 1][lane] = ay[c].x; red[wi][6 * c + 2][lane] = az[c].x;
                    red[wi][6 * c + 3][lane] = ax[c].y; red[wi][6 * c + 4][lane] = ay[c].y; red[wi][6 * c + 5][lane] = az[c].y;
                }
                break;
            }
            // the range goes on into the next super-block: the resident sums of this part go to g's last layer (only the last wave of a
            // super-block's list can go on)
            SymRow* out = partial + (size_t)(pl.r_layer0 + gtab[2 * g + 1] - 1u) * pl.np + (size_t)g * S + lane;
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                if (flushed) {
                    ax[c].x += red[wi][6 * c + 0][lane]; ay[c].x += red[wi][6 * c + 1][lane]; az[c].x += red[wi][6 * c + 2][lane];
                    ax[c].y += red[wi][6 * c + 3][lane]; ay[c].y += red[wi][6 * c + 4][lane]; az[c].y += red[wi][6 * c + 5][lane];
                }
                out[(2 * c) * 64] = SymRow{ax[c].x, ay[c].x, az[c].x};
                out[(2 * c + 1) * 64] = SymRow{ax[c].y, ay[c].y, az[c].y};
            }
        }
        return gfin;
    };
    const uint32_t gfin = run(active ? rec.x : 0u, active ? rec.y : 0u, ~0u);
    // The resident sums of the super-block the range ends in: the waves of the workgroup that end in the same one (consecutive
    // waves share a super-block when there are more waves than super-blocks: N = 16,384 has 64 per super-block) add theirs up in LDS,
    // in wave order, and store ONE row set -- the layer of their table records (the same for all of them) -- a quarter of the resident layers K2 reads.
    NB_STAMP_LIGHT(8);
    if (lane == 0) fin[wi] = gfin;
    __syncthreads();
    NB_STAMP_LIGHT(9);
    if (gfin != ~0u) {
        int members = 0, mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool same = fin[j] == gfin;
            members += same;
            mine += same && j < wi;
        }
        SymRow* out = partial + (size_t)(pl.r_layer0 + rec.z) * pl.np + (size_t)gfin * S + lane;
        for (int r = mine; r < 2 * NG; r += members) {       // the members share the rows out
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (fin[j] == gfin) { sx += red[j][3 * r + 0][lane]; sy += red[j][3 * r + 1][lane]; sz += red[j][3 * r + 2][lane]; }
            out[r * 64] = SymRow{sx, sy, sz};
        }
    }
    // The queue (two waves per SIMD, whole sweeps; nb_plan.cpp::lay_out_symw): the last sweeps of every older wave's range are nobody's
    // own -- a wave that is done draws them one at a time, so the XCDs that hold a higher clock (or started earlier) take more of them
    // and the launch no longer waits for its slowest XCD.  A queued sweep's sums have a layer of their own: which wave ran it does not
    // show in the results.
    if (npieces) {
        for (;;) {
            uint32_t id = 0;
            if (lane == 0) id = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            id = __builtin_amdgcn_readfirstlane(id);
            if (id >= npieces) break;
            const uint32_t pu = gtab[pieces_off + 2u * id], ll = gtab[pieces_off + 2u * id + 1u];      // {first unit, resident layer | sweeps << 16}
            (void)run(pu, pu + (ll >> 16), ll & 0xffffu);
        }
    }
    NB_STAMP(4);
}

// The fp64 form (BASELINE config 5): non-packed, IPL residents per lane, one traveler per lane.  Per unordered pair: 3 adds,
// 3 fma (r^2 + eps2), v_rsq_f64 + first-order correction as in pair(double...) -- y = rsq(d2), e = 1 - d2 y^2,
// u = y^3 (1 + 3e/2) -- then (G m_t) u and (G m_i) u and six fma: 19 DP instructions + the seed for TWO interactions where
// nb_force<double,...> spends 15 + the seed on one; 14 v_mov_b32_dpp per traveler and step rotate the seven doubles.
__device__ __forceinline__ double wave_rot1(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    const unsigned rlo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x13C, 0xF, 0xF, false);
    const unsigned rhi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xF, 0xF, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)rhi << 32) | rlo));
}

// (Tried in round 5 and dropped: IPL = 16 -- the 14 rotations of a step shared by twice the pairs, (16 * 92 + 56) issue cycles per 1,024
// unordered pairs = a ceiling of 83.8 % of the fp64 vector rate against 80.8 % -- which needs 16 x 14 registers of residents, i.e. ONE
// wave per SIMD: N = 262,144 ran 24.7 ms against 24.0 with 8 residents and two waves per SIMD (8 residents and one wave: 24.7 too;
// profiles/r05/f64_16_residents_one_wave_per_simd_ab.txt).  The loop below already issues at its count: 80.8 % x the 2.17 of 2.4 GHz the
// chip holds under this load = 73 %, what it measures.)
template <int IPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void nb_force_symw64(const uint32_t* __restrict__ gtab, const double4* __restrict__ bodies, SymRowT<double>* __restrict__ partial, SymRowT<double>* __restrict__ spill,
                     SYMW_PLAN_PARAMS, const double G_arg, const double eps2_arg /* both read by hand: symw_record_and_tail */,
                     uint32_t* __restrict__ queue, const uint32_t npieces, const uint32_t pieces_off)
{
    constexpr uint32_t S = 64u * IPL, CH = 64u, CPS = S / CH;
    const SymWK pl = symw_plan_words(S, CPS, W_, ups_, nsb_, zc_, r_layer0_, t_layer0_);
    constexpr int GW = 4;                      // residents evaluated stage-major together
    const int lane = threadIdx.x & 63, wi = threadIdx.x >> 6;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wi);
    const bool active = w < pl.W;
    __shared__ double red[4][3 * IPL][64];     // see nb_force_symw: the last resident sums of the workgroup's waves meet here
    __shared__ uint32_t fin[4];
    const uint32_t ups = pl.ups, ush = (uint32_t)__builtin_ctz(ups), ustep = 64u >> ush,      // (ups is a power of two)
                   tab1 = 2u * (pl.np / S);          // wave ranges in units of 64 / ups rotation steps, starts from the table; the short block Z: see nb_force_symw
    nb_u4 rec;
    uint32_t tail[4];
    symw_record_and_tail(gtab + tab1 + 4u * (active ? w : 0u), rec, tail);
    const double G = __builtin_bit_cast(double, (unsigned long long)tail[1] << 32 | tail[0]), eps2 = __builtin_bit_cast(double, (unsigned long long)tail[3] << 32 | tail[2]);
    const uint32_t first_lo = pl.n_hi * pl.total_hi, first_z = first_lo + (pl.nsb - pl.n_hi) * pl.total_lo;
    const uint32_t slot = rec.w;
    auto run = [&](uint32_t u, const uint32_t uend, const uint32_t piece_layer) -> uint32_t {      // see nb_force_symw: the wave's own range, or a piece drawn from the queue
        uint32_t gfin = ~0u;
        while (u < uend) {
            const uint32_t p = u >> ush;
            uint32_t g, k, total;
            if (p < first_lo) { g = p / pl.total_hi; k = p - g * pl.total_hi; total = pl.total_hi; }
            else if (p < first_z) { const uint32_t r = p - first_lo; g = pl.n_hi + r / pl.total_lo; k = r - (g - pl.n_hi) * pl.total_lo; total = pl.total_lo; }
            else { g = pl.nsb; k = p - first_z; total = pl.zc; }
            const uint32_t ring = g < pl.nsb ? total - CPS - pl.zc : 0u, both_end = g < pl.nsb ? ring + pl.zc : 0u;
            uint32_t ug_end = (p - k + total) * ups;
            if (ug_end > uend) ug_end = uend;
            double xi[IPL], yi[IPL], zi[IPL], mi[IPL], ax[IPL], ay[IPL], az[IPL];
#pragma unroll
            for (int c = 0; c < IPL; ++c) {
                const double4 b = ld4(bodies + (size_t)g * S + c * 64 + lane);
                xi[c] = b.x; yi[c] = b.y; zi[c] = b.z; mi[c] = b.w * G;
                ax[c] = 0; ay[c] = 0; az[c] = 0;
            }
            while (u < ug_end) {
                const uint32_t q0 = u & (ups - 1u);
                uint32_t nun = ups - q0;
                if (nun > ug_end - u) nun = ug_end - u;
                const uint32_t s0 = q0 * ustep, s1 = s0 + nun * ustep;
                u += nun;
                const bool sym = k < both_end, zsweep = k >= ring && sym;
                const uint32_t d = k / CPS;
                uint32_t tb = g + 1 + d;
                if (tb >= pl.nsb) tb -= pl.nsb;
                const uint32_t tstart = k < ring ? tb * S + (k % CPS) * CH : zsweep ? pl.nsb * S + (k - ring) * CH : g * S + (k - both_end) * CH;
                const uint32_t zrow = g * pl.zc + (k - ring);
                ++k;
                const double4 t = ld4(bodies + tstart + (((uint32_t)lane - s0) & 63u));
                double tx = t.x, ty = t.y, tz = t.z, tm = t.w * G, bx = 0, by = 0, bz = 0;
                auto steps = [&](auto both) {                                // see nb_force_symw: a sweep over an own chunk keeps no traveler sums
                    constexpr bool BOTH = decltype(both)::value;
                    for (uint32_t st = s0; st < s1; ++st) {
#pragma unroll
                        for (int c0g = 0; c0g < IPL; c0g += GW) {            // stage-major over four residents
                            double dx[GW], dy[GW], dz[GW], d2[GW], y[GW], uu[GW];
#pragma unroll
                            for (int c = 0; c < GW; ++c) dx[c] = tx - xi[c0g + c];
#pragma unroll
                            for (int c = 0; c < GW; ++c) dy[c] = ty - yi[c0g + c];
#pragma unroll
                            for (int c = 0; c < GW; ++c) dz[c] = tz - zi[c0g + c];
#pragma unroll
                            for (int c = 0; c < GW; ++c) d2[c] = nb_fma(dz[c], dz[c], nb_fma(dy[c], dy[c], nb_fma(dx[c], dx[c], eps2)));
#pragma unroll
                            for (int c = 0; c < GW; ++c) y[c] = __builtin_amdgcn_rsq(d2[c]);
#pragma unroll
                            for (int c = 0; c < GW; ++c) {
                                const double y2 = y[c] * y[c];
                                const double e = nb_fma(-d2[c], y2, 1.0);
                                const double t3 = y[c] * y2;
                                uu[c] = nb_fma(t3 * e, 1.5, t3);
                            }
#pragma unroll
                            for (int c = 0; c < GW; ++c) {
                                const double si = tm * uu[c];
                                ax[c0g + c] = nb_fma(si, dx[c], ax[c0g + c]); ay[c0g + c] = nb_fma(si, dy[c], ay[c0g + c]); az[c0g + c] = nb_fma(si, dz[c], az[c0g + c]);
                                if constexpr (BOTH) {
                                    const double sj = mi[c0g + c] * uu[c];
                                    bx = nb_fma(-sj, dx[c], bx); by = nb_fma(-sj, dy[c], by); bz = nb_fma(-sj, dz[c], bz);
                                }
                            }
                        }
                        tx = wave_rot1(tx); ty = wave_rot1(ty); tz = wave_rot1(tz); tm = wave_rot1(tm);
                        if constexpr (BOTH) { bx = wave_rot1(bx); by = wave_rot1(by); bz = wave_rot1(bz); }
                    }
                };
                if (sym) steps(std::true_type{}); else steps(std::false_type{});
                if (sym) {
                    SymRowT<double>* out = (s0 != 0 ? spill + (size_t)slot * CH : zsweep ? spill + (size_t)zrow * CH : partial + (size_t)(pl.t_layer0 + d) * pl.np + tstart)
                                           + (((uint32_t)lane - s1) & 63u);
                    *out = SymRowT<double>{bx, by, bz};
                }
            }
            if (u >= uend && piece_layer != ~0u) {
                SymRowT<double>* out = partial + (size_t)(pl.r_layer0 + piece_layer) * pl.np + (size_t)g * S + lane;
#pragma unroll
                for (int c = 0; c < IPL; ++c) out[c * 64] = SymRowT<double>{ax[c], ay[c], az[c]};
                return g;
            }
            if (u >= uend) {
                gfin = g;
#pragma unroll
                for (int c = 0; c < IPL; ++c) { red[wi][3 * c + 0][lane] = ax[c]; red[wi][3 * c + 1][lane] = ay[c]; red[wi][3 * c + 2][lane] = az[c]; }
                break;
            }
            SymRowT<double>* out = partial + (size_t)(pl.r_layer0 + gtab[2 * g + 1] - 1u) * pl.np + (size_t)g * S + lane;     // the range goes on: g's last layer
#pragma unroll
            for (int c = 0; c < IPL; ++c) out[c * 64] = SymRowT<double>{ax[c], ay[c], az[c]};
        }
        return gfin;
    };
    const uint32_t gfin = run(active ? rec.x : 0u, active ? rec.y : 0u, ~0u);
    if (lane == 0) fin[wi] = gfin;
    __syncthreads();
    if (gfin != ~0u) {
        int members = 0, mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool same = fin[j] == gfin;
            members += same;
            mine += same && j < wi;
        }
        SymRowT<double>* out = partial + (size_t)(pl.r_layer0 + rec.z) * pl.np + (size_t)gfin * S + lane;
        for (int r = mine; r < IPL; r += members) {
            double sx = 0, sy = 0, sz = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (fin[j] == gfin) { sx += red[j][3 * r + 0][lane]; sy += red[j][3 * r + 1][lane]; sz += red[j][3 * r + 2][lane]; }
            out[r * 64] = SymRowT<double>{sx, sy, sz};
        }
    }
    if (npieces) {                             // the queue: see nb_force_symw
        for (;;) {
            uint32_t id = 0;
            if (lane == 0) id = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            id = __builtin_amdgcn_readfirstlane(id);
            if (id >= npieces) break;
            const uint32_t pu = gtab[pieces_off + 2u * id], ll = gtab[pieces_off + 2u * id + 1u];
            (void)run(pu, pu + (ll >> 16), ll & 0xffffu);
        }
    }
}

// ---- the rank form's force kernels (nb::SymRankPlan, nb_plan.h): whole sweeps, one traveler per lane, two phases -------------
// Waves [0, WA) sweep phase A (travelers = the rank's own rows), waves [WA, WA + WB) phase B; `wave0` / `wave_end` select the part
// of them a launch runs (everything, or A before the wait for the all-gather and B after it).  A phase's sweeps are laid end to
// end; prefix[gi] is where own super-block g0 + gi begins.
// Layers are COMPACT: a layer holds the (g1 - g0) * S rows this rank can write -- resident layers the rows of its own super-blocks,
// traveler layer d the sums super-block g produced for the rows of block g + 1 + d, filed under the SOURCE g -- so a rank of 8
// allocates an eighth of what the whole-system form does (N = 1,048,576: 0.8 GB instead of 6.4).
__device__ __forceinline__ uint32_t rank_find(const uint32_t* __restrict__ prefix, uint32_t ng, uint32_t p)
{
    uint32_t lo = 0, hi = ng;                         // largest gi with prefix[gi] <= p  (prefix[ng] > p)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NG > 4 ? 2 : 4, NG > 4 ? 2 : 4)))
void nb_force_symw_rank(const float4* __restrict__ bodies, SymRow* __restrict__ partial, const uint32_t* __restrict__ tab, const SymRankPlan pl,
                        const uint32_t n, const float eps2, const uint32_t wave0, const uint32_t wave_end, SymRow* __restrict__ spill,
                        const uint32_t k_lo, const uint32_t k_hi, const uint32_t d0)
{
    constexpr uint32_t S = 128u * NG, CPS = S / 64u;
    constexpr int GW = NG < 4 ? NG : 4;
    const int lane = threadIdx.x & 63;
    const uint32_t w = wave0 + __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (w >= wave_end) return;
    const bool phase_b = w >= pl.WA;
    const uint32_t wl = phase_b ? w - pl.WA : w, Lp = phase_b ? pl.LB : pl.LA, Wp = phase_b ? pl.WB : pl.WA;
    const uint32_t ng = pl.g1 - pl.g0;
    const size_t lstride = (size_t)ng * S;             // rows per (compact) layer
    const uint32_t* __restrict__ prefix = tab + 4 * pl.nsb + (phase_b ? ng + 1 : 0);
    // the wave's range in units of 64 / ups rotation steps of its phase (see nb_force_symw)
    const uint32_t ups = pl.ups, ustep = 64u / ups;
    const uint64_t Lu = (uint64_t)Lp * ups;
    uint32_t u = (uint32_t)(((uint64_t)wl * Lu) / Wp);
    const uint32_t uend = (uint32_t)(((uint64_t)(wl + 1) * Lu) / Wp);
    const nb_f2 e2 = nb_f2{eps2, eps2};
    __shared__ float hi[4][6 * NG][64];          // second level of the resident sums (see nb_force_symw): a whole system in passes keeps a wave
    const int wi = threadIdx.x >> 6;             // on one super-block for ~16,000 sweeps (N = 4,194,304)
    while (u < uend) {
        const uint32_t ps = u / ups;
        const uint32_t gi = rank_find(prefix, ng, ps), g = pl.g0 + gi;
        uint32_t j = ps - prefix[gi];
        uint32_t ug_end = prefix[gi + 1] * ups;                      // end of g's part of the phase, in units
        if (ug_end > uend) ug_end = uend;
        const uint32_t total = g < pl.n_hi ? pl.total_hi : pl.total_lo, ring = total - CPS;
        uint32_t a = (pl.g1 - 1 - g) * CPS;                          // sweeps of g whose travelers are own rows
        if (a > ring) a = ring;
        // the pass's window [k_lo, k_hi) of the ring sweeps (one pass: everything): phase A runs [a_lo, a_lo + a_len) and then the
        // resident-only sweeps (if the pass has them: its prefix table says so), phase B [b_lo, ...)
        const uint32_t a_lo = a < k_lo ? a : k_lo, a_len = (a < k_hi ? a : k_hi) - a_lo;
        uint32_t b_lo = a > k_lo ? a : k_lo;
        if (b_lo > ring) b_lo = ring;

        nb_f2 xi[NG], yi[NG], zi[NG], mi[NG], ax[NG], ay[NG], az[NG];
        {
            const float4* rb = bodies + (size_t)g * S + lane;
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                const float4 b0 = ld4(rb + (2 * c) * 64), b1 = ld4(rb + (2 * c + 1) * 64);
                xi[c] = nb_f2{b0.x, b1.x}; yi[c] = nb_f2{b0.y, b1.y}; zi[c] = nb_f2{b0.z, b1.z}; mi[c] = nb_f2{b0.w, b1.w};
                ax[c] = nb_f2{0, 0}; ay[c] = nb_f2{0, 0}; az[c] = nb_f2{0, 0};
            }
        }
        uint32_t since = 0;
        bool flushed = false;
        while (u < ug_end) {
            const uint32_t q0 = u % ups;                             // the wave's steps [s0, s1) of this sweep
            uint32_t nun = ups - q0;
            if (nun > ug_end - u) nun = ug_end - u;
            const uint32_t s0 = q0 * ustep, s1 = s0 + nun * ustep;
            u += nun;
            const uint32_t k = phase_b ? b_lo + j : (j < a_len ? a_lo + j : ring + (j - a_len));       // position in g's list
            ++j;
            const bool sym = k < ring;
            const uint32_t d = k / CPS;
            uint32_t tb = g + 1 + d;
            if (tb >= pl.nsb) tb -= pl.nsb;
            const uint32_t tstart = sym ? tb * S + (k % CPS) * 64u : g * S + (k - ring) * 64u;
            if (tstart >= n) continue;
            if (since >= kFlushSteps) {                              // register sums -> second level (LDS), as nb_force_symw
                flush_resident_sums<NG>(hi[wi], lane, flushed, ax, ay, az);
                flushed = true;
                since = 0;
            }
            since += s1 - s0;
            const float4 t = ld4(bodies + tstart + (((uint32_t)lane - s0) & 63u));
            float tx = t.x, ty = t.y, tz = t.z, tm = t.w;
            nb_f2 bx = nb_f2{0, 0}, by = nb_f2{0, 0}, bz = nb_f2{0, 0};
            for (uint32_t st = s0; st < s1; ++st) {
                const nb_f2 px = nb_f2{tx, tx}, py = nb_f2{ty, ty}, pz = nb_f2{tz, tz}, pm = nb_f2{tm, tm};
#pragma unroll
                for (int c0g = 0; c0g < NG; c0g += GW) {
                    nb_f2 dx[GW], dy[GW], dz[GW], d2[GW], r[GW], si[GW], sj[GW];
#pragma unroll
                    for (int c = 0; c < GW; ++c) dx[c] = px - xi[c0g + c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) dy[c] = py - yi[c0g + c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) dz[c] = pz - zi[c0g + c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) d2[c] = __builtin_elementwise_fma(dx[c], dx[c], e2);
#pragma unroll
                    for (int c = 0; c < GW; ++c) d2[c] = __builtin_elementwise_fma(dy[c], dy[c], d2[c]);
#pragma unroll
                    for (int c = 0; c < GW; ++c) d2[c] = __builtin_elementwise_fma(dz[c], dz[c], d2[c]);
#pragma unroll
                    for (int c = 0; c < GW; ++c) r[c] = d2[c] * d2[c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) r[c] = r[c] * d2[c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) r[c] = nb_f2{nb_rsq(r[c].x), nb_rsq(r[c].y)};
#pragma unroll
                    for (int c = 0; c < GW; ++c) si[c] = pm * r[c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) sj[c] = mi[c0g + c] * r[c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) ax[c0g + c] = __builtin_elementwise_fma(si[c], dx[c], ax[c0g + c]);
#pragma unroll
                    for (int c = 0; c < GW; ++c) ay[c0g + c] = __builtin_elementwise_fma(si[c], dy[c], ay[c0g + c]);
#pragma unroll
                    for (int c = 0; c < GW; ++c) az[c0g + c] = __builtin_elementwise_fma(si[c], dz[c], az[c0g + c]);
#pragma unroll
                    for (int c = 0; c < GW; ++c) bx = __builtin_elementwise_fma(-sj[c], dx[c], bx);
#pragma unroll
                    for (int c = 0; c < GW; ++c) by = __builtin_elementwise_fma(-sj[c], dy[c], by);
#pragma unroll
                    for (int c = 0; c < GW; ++c) bz = __builtin_elementwise_fma(-sj[c], dz[c], bz);
                }
                tx = wave_rot1(tx); ty = wave_rot1(ty); tz = wave_rot1(tz); tm = wave_rot1(tm);
                bx = nb_f2{wave_rot1(bx.x), wave_rot1(bx.y)};
                by = nb_f2{wave_rot1(by.x), wave_rot1(by.y)};
                bz = nb_f2{wave_rot1(bz.x), wave_rot1(bz.y)};
            }
            if (sym) {
                SymRow* out = (s0 == 0 ? partial + (size_t)(pl.t_layer0 + d - d0) * lstride + (size_t)gi * S + (k % CPS) * 64u : spill + (size_t)w * 64u) + (((uint32_t)lane - s1) & 63u);
                *out = SymRow{bx.x + bx.y, by.x + by.y, bz.x + bz.y};
            }
        }
        // resident sums of this wave's part of g's list in this phase
        const uint32_t* gt = tab + 4 * g;
        SymRow* out = partial + (size_t)(phase_b ? pl.rb_layer0 + (wl - gt[2]) : pl.r_layer0 + (wl - gt[0])) * lstride + (size_t)gi * S + lane;
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            if (flushed) {
                ax[c].x += hi[wi][6 * c + 0][lane]; ay[c].x += hi[wi][6 * c + 1][lane]; az[c].x += hi[wi][6 * c + 2][lane];
                ax[c].y += hi[wi][6 * c + 3][lane]; ay[c].y += hi[wi][6 * c + 4][lane]; az[c].y += hi[wi][6 * c + 5][lane];
            }
            out[(2 * c) * 64] = SymRow{ax[c].x, ay[c].x, az[c].x};
            out[(2 * c + 1) * 64] = SymRow{ax[c].y, ay[c].y, az[c].y};
        }
    }
}

template <int IPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void nb_force_symw64_rank(const double4* __restrict__ bodies, SymRowT<double>* __restrict__ partial, const uint32_t* __restrict__ tab,
                          const SymRankPlan pl, const uint32_t n, const double G, const double eps2, const uint32_t wave0, const uint32_t wave_end,
                          SymRowT<double>* __restrict__ spill, const uint32_t k_lo, const uint32_t k_hi, const uint32_t d0)
{
    constexpr uint32_t S = 64u * IPL, CPS = S / 64u;
    constexpr int GW = 4;
    const int lane = threadIdx.x & 63;
    const uint32_t w = wave0 + __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (w >= wave_end) return;
    const bool phase_b = w >= pl.WA;
    const uint32_t wl = phase_b ? w - pl.WA : w, Lp = phase_b ? pl.LB : pl.LA, Wp = phase_b ? pl.WB : pl.WA;
    const uint32_t ng = pl.g1 - pl.g0;
    const size_t lstride = (size_t)ng * S;             // rows per (compact) layer
    const uint32_t* __restrict__ prefix = tab + 4 * pl.nsb + (phase_b ? ng + 1 : 0);
    const uint32_t ups = pl.ups, ustep = 64u / ups;
    const uint64_t Lu = (uint64_t)Lp * ups;
    uint32_t u = (uint32_t)(((uint64_t)wl * Lu) / Wp);
    const uint32_t uend = (uint32_t)(((uint64_t)(wl + 1) * Lu) / Wp);
    while (u < uend) {
        const uint32_t ps = u / ups;
        const uint32_t gi = rank_find(prefix, ng, ps), g = pl.g0 + gi;
        uint32_t j = ps - prefix[gi];
        uint32_t ug_end = prefix[gi + 1] * ups;
        if (ug_end > uend) ug_end = uend;
        const uint32_t total = g < pl.n_hi ? pl.total_hi : pl.total_lo, ring = total - CPS;
        uint32_t a = (pl.g1 - 1 - g) * CPS;
        if (a > ring) a = ring;
        const uint32_t a_lo = a < k_lo ? a : k_lo, a_len = (a < k_hi ? a : k_hi) - a_lo;         // the pass's window: see nb_force_symw_rank
        uint32_t b_lo = a > k_lo ? a : k_lo;
        if (b_lo > ring) b_lo = ring;
        double xi[IPL], yi[IPL], zi[IPL], mi[IPL], ax[IPL], ay[IPL], az[IPL];
#pragma unroll
        for (int c = 0; c < IPL; ++c) {
            const double4 b = ld4(bodies + (size_t)g * S + c * 64 + lane);
            xi[c] = b.x; yi[c] = b.y; zi[c] = b.z; mi[c] = b.w * G;
            ax[c] = 0; ay[c] = 0; az[c] = 0;
        }
        while (u < ug_end) {
            const uint32_t q0 = u % ups;
            uint32_t nun = ups - q0;
            if (nun > ug_end - u) nun = ug_end - u;
            const uint32_t s0 = q0 * ustep, s1 = s0 + nun * ustep;
            u += nun;
            const uint32_t k = phase_b ? b_lo + j : (j < a_len ? a_lo + j : ring + (j - a_len));
            ++j;
            const bool sym = k < ring;
            const uint32_t d = k / CPS;
            uint32_t tb = g + 1 + d;
            if (tb >= pl.nsb) tb -= pl.nsb;
            const uint32_t tstart = sym ? tb * S + (k % CPS) * 64u : g * S + (k - ring) * 64u;
            if (tstart >= n) continue;
            const double4 t = ld4(bodies + tstart + (((uint32_t)lane - s0) & 63u));
            double tx = t.x, ty = t.y, tz = t.z, tm = t.w * G, bx = 0, by = 0, bz = 0;
            for (uint32_t st = s0; st < s1; ++st) {
#pragma unroll
                for (int c0g = 0; c0g < IPL; c0g += GW) {
                    double dx[GW], dy[GW], dz[GW], d2[GW], y[GW], uu[GW];
#pragma unroll
                    for (int c = 0; c < GW; ++c) dx[c] = tx - xi[c0g + c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) dy[c] = ty - yi[c0g + c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) dz[c] = tz - zi[c0g + c];
#pragma unroll
                    for (int c = 0; c < GW; ++c) d2[c] = nb_fma(dz[c], dz[c], nb_fma(dy[c], dy[c], nb_fma(dx[c], dx[c], eps2)));
#pragma unroll
                    for (int c = 0; c < GW; ++c) y[c] = __builtin_amdgcn_rsq(d2[c]);
#pragma unroll
                    for (int c = 0; c < GW; ++c) {
                        const double y2 = y[c] * y[c];
                        const double e = nb_fma(-d2[c], y2, 1.0);
                        const double t3 = y[c] * y2;
                        uu[c] = nb_fma(t3 * e, 1.5, t3);
                    }
#pragma unroll
                    for (int c = 0; c < GW; ++c) {
                        const double si = tm * uu[c], sj = mi[c0g + c] * uu[c];
                        ax[c0g + c] = nb_fma(si, dx[c], ax[c0g + c]); ay[c0g + c] = nb_fma(si, dy[c], ay[c0g + c]); az[c0g + c] = nb_fma(si, dz[c], az[c0g + c]);
                        bx = nb_fma(-sj, dx[c], bx); by = nb_fma(-sj, dy[c], by); bz = nb_fma(-sj, dz[c], bz);
                    }
                }
                tx = wave_rot1(tx); ty = wave_rot1(ty); tz = wave_rot1(tz); tm = wave_rot1(tm);
                bx = wave_rot1(bx); by = wave_rot1(by); bz = wave_rot1(bz);
            }
            if (sym) {
                SymRowT<double>* out = (s0 == 0 ? partial + (size_t)(pl.t_layer0 + d - d0) * lstride + (size_t)gi * S + (k % CPS) * 64u : spill + (size_t)w * 64u) + (((uint32_t)lane - s1) & 63u);
                *out = SymRowT<double>{bx, by, bz};
            }
        }
        const uint32_t* gt = tab + 4 * g;
        SymRowT<double>* out = partial + (size_t)(phase_b ? pl.rb_layer0 + (wl - gt[2]) : pl.r_layer0 + (wl - gt[0])) * lstride + (size_t)gi * S + lane;
#pragma unroll
        for (int c = 0; c < IPL; ++c) out[c * 64] = SymRowT<double>{ax[c], ay[c], az[c]};
    }
}

// The RANK form of the pass (multi-GPU: rank r keeps the super-blocks [g0, g1) of its own rows resident and sweeps THEIR chunk
// lists, so every unordered pair of the system is evaluated by exactly one rank): the traveler sums a rank produces belong
// to bodies of other ranks as well.  This kernel adds up, for EVERY row of the system, what this rank has for it -- its
// resident layers (own rows only) and the traveler layers written by the rank's own super-blocks, in the order
// nb_integrate_symw uses -- into one array A[np]; the ranks then reduce-scatter A (ncclReduceScatter, or peer copies + a
// fixed-order sum in the single-process handle) and the plain integrate kernel reads the rank's rows of the result.
template <typename T>
__global__ __launch_bounds__(kBlock) void nb_sym_reduce(const SymRowT<T>* __restrict__ partial, const uint32_t* __restrict__ tab,
                                                       typename vec4<T>::type* __restrict__ A, const SymRankPlan pl, uint32_t S,
                                                       const SymRowT<T>* __restrict__ spill, uint32_t d0, uint32_t d1, uint32_t accumulate)
{
    using SymRow = SymRowT<T>;
    using V4 = typename vec4<T>::type;
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= pl.np) return;
    const uint32_t b = j / S, g0 = pl.g0, g1 = pl.g1, within = j - b * S;
    const size_t lstride = (size_t)(g1 - g0) * S;                // compact layers: rows filed under the own super-block that wrote them
    T sx = 0, sy = 0, sz = 0;
    // FOUR rows requested per trip, added in the same fixed order as one by one (round 5: a thread that loaded a row, added it and only
    // then asked for the next was a chain of dependent round trips -- 61 us per call for a rank of 8 at N = 262,144).  `at(e)` names the
    // e-th row of a section, or nullptr where the section has none at e: row 0 of `partial` is requested in its place and 0 added.
    auto add_rows = [&](uint32_t count, auto at) {
        for (uint32_t e = 0; e < count; e += 4) {
            const SymRow* q[4];
            SymRow r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { q[i] = e + i < count ? at(e + i) : nullptr; r[i] = *(q[i] ? q[i] : partial); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { sx += q[i] ? r[i].x : T(0); sy += q[i] ? r[i].y : T(0); sz += q[i] ? r[i].z : T(0); }
        }
    };
    if (b >= g0 && b < g1) {                                     // resident layers: phase A's waves, then phase B's
        const uint32_t na = tab[4 * b + 1], nb_ = tab[4 * b + 3];
        const size_t row = (size_t)(b - g0) * S + within;
        add_rows(na, [&](uint32_t e) { return partial + (size_t)(pl.r_layer0 + e) * lstride + row; });
        add_rows(nb_, [&](uint32_t e) { return partial + (size_t)(pl.rb_layer0 + e) * lstride + row; });
    }
    if (g1 - g0 > pl.H) {
        // few ring distances, many own super-blocks: ascending distance
        add_rows(pl.H + 1 > d0 ? std::min(pl.H + 1, d1) - d0 : 0u, [&](uint32_t e) -> const SymRow* {
            const uint32_t d = d0 + e;
            uint32_t g = b + pl.nsb - 1 - d;
            if (g >= pl.nsb) g -= pl.nsb;
            if (g < g0 || g >= g1 || d >= pl.H + (g < pl.n_hi ? 1u : 0u)) return nullptr;       // [d0, d1): the ring distances of this pass
            return partial + (size_t)(pl.t_layer0 + d - d0) * lstride + (size_t)(g - g0) * S + within;
        });
    } else {
        // a rank of many: only its own super-blocks can have written a layer of row j
        add_rows(g1 - g0, [&](uint32_t e) -> const SymRow* {
            const uint32_t g = g0 + e;
            uint32_t d = b + pl.nsb - 1 - g;
            if (d >= pl.nsb) d -= pl.nsb;
            if (d < d0 || d >= d1 || d >= pl.H + (g < pl.n_hi ? 1u : 0u)) return nullptr;
            return partial + (size_t)(pl.t_layer0 + d - d0) * lstride + (size_t)(g - g0) * S + within;
        });
    }
    if (pl.ups > 1) {                                            // later parts of sweeps shared by two waves: the chunk's spill list
        const uint32_t base = 4 * pl.nsb + 2 * (g1 - g0 + 1), ci = j >> 6;
        const uint32_t so = tab[base + 2 * ci], ns = tab[base + 2 * ci + 1];
        const uint32_t* ids = tab + base + 2 * (pl.np >> 6) + so;
        add_rows(ns, [&](uint32_t e) { return spill + (size_t)ids[e] * 64u + (j & 63u); });
    }
    if (accumulate) { const V4 o = ld4(A + j); sx += o.x; sy += o.y; sz += o.z; }      // a later pass over the ring distances: layers reused, sums carried in A
    A[j] = V4{sx, sy, sz, 0};
}

// The single-process multi-device handle's reduce-scatter by peer copies: stage[d] holds shard d's A rows for THIS shard's
// row block (shard d = own: its own A); summed in ascending shard order -- deterministic.
template <typename T>
__global__ __launch_bounds__(kBlock) void nb_sym_sum_shards(const typename vec4<T>::type* __restrict__ stage, typename vec4<T>::type* __restrict__ out,
                                                           uint32_t rows, uint32_t shards)
{
    using V4 = typename vec4<T>::type;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= rows) return;
    T sx = 0, sy = 0, sz = 0;
    for (uint32_t d = 0; d < shards; ++d) { const V4 r = ld4(stage + (size_t)d * rows + i); sx += r.x; sy += r.y; sz += r.z; }
    out[i] = V4{sx, sy, sz, 0};
}

// The single-process multi-device handle's exchanges as PULL kernels (peer access: a kernel on device e reads the other
// shards' arrays directly): one launch per shard instead of g - 1 hipMemcpyAsync -- the host thread that drives all g devices
// issued ~120 copies per step at g = 8 (0.9-1.4 ms of host time against a 1.4 ms step, profiles/r03/multi_host_cost.txt).
struct PeerPtrs { const void* p[16]; };

// reduce-scatter: out[i] = sum over shards d (ascending: deterministic) of A_d[e * rows + i]
template <typename T>
__global__ __launch_bounds__(kBlock) void nb_peer_sum(const PeerPtrs src, typename vec4<T>::type* __restrict__ out, uint32_t rows, uint32_t shards, uint32_t e)
{
    using V4 = typename vec4<T>::type;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= rows) return;
    T sx = 0, sy = 0, sz = 0;
    for (uint32_t d = 0; d < shards; ++d) {
        const V4 r = ld4((const V4*)src.p[d] + (size_t)e * rows + i);
        sx += r.x; sy += r.y; sz += r.z;
    }
    out[i] = V4{sx, sy, sz, 0};
}

// all-gather: dst (shard e's replicated array) takes every other shard's own row block from that shard's array
template <typename T>
__global__ __launch_bounds__(kBlock) void nb_peer_gather(const PeerPtrs src, typename vec4<T>::type* __restrict__ dst, uint32_t rows, uint32_t shards, uint32_t e)
{
    using V4 = typename vec4<T>::type;
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= rows * shards) return;
    const uint32_t d = idx / rows;
    if (d == e) return;
    dst[idx] = ld4((const V4*)src.p[d] + idx);
}

// K2 for the wave-granular form: resident layers gtab[2g+1] (workgroups whose waves ended in g's list, + the wave that went on), then the traveler layers, then (wave
// ranges cut inside sweeps, pl.ups > 1) the spill rows of the waves that ran a later part of a sweep over the body's chunk:
// {first spill row, count} per chunk of CH rows at gtab[spill_off + 2 chunk] (spill_off = 2 np / S + 4 W) (a chunk's spill rows are consecutive, in list order: the
// row addresses hang on ONE table load, like the layers').  Fixed order.  The body's own state is requested before the sums.
// The arguments: the table pointer, the shifts and the spill lists' offset lie inside the 14 dwords the hardware preloads into SGPRs, so
// the two table reads go out in the wave's first instructions, together with the rest of the arguments (round 5: the plan struct by
// value and `S` behind it put the table read behind a round trip for the arguments, and a software division in front of it).
template <typename T, int R>
__global__ __launch_bounds__(kBlock) void nb_integrate_symw(const uint32_t* __restrict__ gtab, const SymRowT<T>* __restrict__ partial,
                                                           typename vec4<T>::type* __restrict__ bodies, typename vec4<T>::type* __restrict__ vel,
                                                           typename vec4<T>::type* __restrict__ acc,
                                                           const uint32_t n, const uint32_t shifts /* log2 S | log2 (travelers per chunk) << 8 */,
                                                           const uint32_t np, const uint32_t spill_off /* word offset of the spill lists in gtab; 0: whole sweeps */,
                                                           const SymRowT<T>* __restrict__ spill_arg /* read by hand below */, typename vec4<T>::type* __restrict__ gout, const T dt, const T G,
                                                           const uint32_t t_layer0, const uint32_t r_layer0, const uint32_t nsb, const uint32_t n_hi,
                                                           const uint32_t H, const uint32_t zc)
{
    using V4 = typename vec4<T>::type;
    const struct { uint32_t np, nsb, n_hi, H, r_layer0, t_layer0, zc; } pl{np, nsb, n_hi, H, r_layer0, t_layer0, zc};
    const uint32_t s_shift = shifts & 0xffu, ch_shift = shifts >> 8;
    constexpr int kSpillArgOffset = 5 * 8 + 4 * 4;      // byte offset of `spill_arg` in the kernel arguments: five pointers and four words in front of it
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t il = gid / R, r = gid % R;
    const bool valid = il < n;
    T sx = 0, sy = 0, sz = 0;
    typename vec4<T>::type b0{}, v0{}, a0{};
    if (valid) {
        if (r == 0) { b0 = ld4(bodies + il); v0 = ld4(vel + il); a0 = ld4(acc + il); }
        // a workgroup's kBlock / R bodies lie in ONE super-block and ONE traveler chunk (both are multiples of 64 rows): the table
        // entries are the same for the whole workgroup -- scalar loads
        static_assert((kBlock / R) <= 64 && 64 % (kBlock / R) == 0, "a workgroup's bodies stay inside one 64-row chunk");
        const uint32_t il0 = blockIdx.x * (kBlock / R);
        const uint32_t b = il0 >> s_shift;
        // (the two table reads -- resident layers of the block, spill rows of the chunk -- are requested TOGETHER: with the second one
        // under `if (ups > 1)` the compiler waited for the first before it issued it, one more scalar round trip in front of the rows;
        // whole-sweep plans have no spill table: they re-read the block's own entry and ignore it)
        const uint32_t ci0 = il0 >> ch_shift;
        // (as explicit instructions: the compiler split two plain loads into three single-word loads, each behind the wait for the one
        // before, and sank the load of the `spill` argument -- the first one behind the preloaded part, used under conditions only --
        // behind that wait: it is requested here, from its place in the kernel arguments, together with the table entries)
        unsigned long long blk, ent, spill_bits;
        asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dwordx2 %2, %5, %6\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(blk), "=&s"(ent), "=&s"(spill_bits)
                     : "s"(gtab + 2 * b), "s"(gtab + (spill_off ? spill_off + 2 * ci0 : 2 * b)), "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(kSpillArgOffset) : "memory");
        const SymRowT<T>* const spill = (const SymRowT<T>*)spill_bits;
        const uint32_t nr = (uint32_t)(blk >> 32), ent0 = (uint32_t)ent, ent1 = (uint32_t)(ent >> 32);
        // traveler sums: a row of a whole super-block has one layer per ring distance; a row of the short block Z (b == pl.nsb) has
        // the z-rows instead -- one per whole super-block, row g * zc + c of the spill buffer (c: its chunk inside Z)
        const bool zb = b >= pl.nsb;
        const uint32_t nt = zb ? pl.nsb : pl.H + ((pl.n_hi && b >= pl.n_hi) ? 1u : 0u);
        uint32_t ns = 0, s_first = 0;
        const uint32_t ci = il0 >> ch_shift;
        const uint32_t zci = ci - (pl.nsb << (s_shift - ch_shift));
        if (spill_off) { s_first = ent0; ns = ent1; }
        const uint32_t total = nr + nt + ns;
        auto row = [&](uint32_t e) {
            if (e < nr) return partial + (size_t)(pl.r_layer0 + e) * pl.np + il;
            if (e < nr + nt) return zb ? spill + (((size_t)((e - nr) * pl.zc + zci) << ch_shift) + (il - (ci << ch_shift)))
                                       : partial + (size_t)(pl.t_layer0 + (e - nr)) * pl.np + il;
            return spill + (((size_t)(s_first + (e - nr - nt)) << ch_shift) + (il - (ci << ch_shift)));
        };
        // Rows e = r, r + R, ... in ascending order, FOUR requests in flight per trip for EVERY lane: a lane whose share is not a
        // multiple of four rows re-requests row 0 for the missing ones and adds 0 in their place.  (Round 4 ran a one-row tail loop
        // instead: load, wait, add, next load -- with ~30 rows over 8 lanes two lanes of every body took THREE round trips in
        // sequence where the others took one, and the whole launch waited for them: K2 4.4 -> 3.x us at N = 9,000 .. 20,000.)
        for (uint32_t e = r; e < total; e += 4 * R) {
            SymRowT<T> p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t ee = e + q * R;
                p[q] = *row(ee < total ? ee : 0u);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = e + q * R < total;
                sx += ok ? p[q].x : T(0); sy += ok ? p[q].y : T(0); sz += ok ? p[q].z : T(0);
            }
        }
    }
    if constexpr (R > 1) {
#pragma unroll
        for (int m = 1; m < R; m <<= 1) {
            sx += __shfl_xor(sx, m, 64);
            sy += __shfl_xor(sy, m, 64);
            sz += __shfl_xor(sz, m, 64);
        }
    }
    if (!valid || r != 0) return;
    V4 nx, nv, na;
    leapfrog<T>(b0, v0, a0, sx, sy, sz, dt, nx, nv, na);
    vel[il] = nv;                                                       // :281
    bodies[il] = nx;                                                    // :283
    acc[il] = na;                                                       // :290
    if (gout) gout[il] = V4{nx.x, nx.y, nx.z, G * nx.w};
}

// K2 for the symmetric pass: a body's acceleration is the sum of its resident layers (one per segment of its
// super-block's chunk list) and its traveler layers (one per ring distance), in ascending layer order.
template <int R>
__global__ __launch_bounds__(kBlock) void nb_integrate_sym(float4* __restrict__ bodies, float4* __restrict__ vel, float4* __restrict__ acc,
                                                          const SymRow* __restrict__ partial, uint32_t n, const SymPlan pl, uint32_t S, float dt,
                                                          float4* __restrict__ gout, float G)
{
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t il = gid / R, r = gid % R;
    const bool valid = il < n;
    float sx = 0, sy = 0, sz = 0;
    if (valid) {
        const uint32_t b = il / S;
        const uint32_t nr = pl.q;
        const uint32_t nt = pl.H + ((pl.n_hi && b >= pl.n_hi) ? 1u : 0u);
        const uint32_t total = nr + nt;
        auto row = [&](uint32_t e) { return partial + (size_t)(e < nr ? pl.r_layer0 + e : pl.t_layer0 + (e - nr)) * pl.np + il; };
        uint32_t e = r;
        for (; e + 3 * R < total; e += 4 * R) {
            const SymRow p0 = *row(e), p1 = *row(e + R), p2 = *row(e + 2 * R), p3 = *row(e + 3 * R);
            sx += p0.x; sy += p0.y; sz += p0.z;
            sx += p1.x; sy += p1.y; sz += p1.z;
            sx += p2.x; sy += p2.y; sz += p2.z;
            sx += p3.x; sy += p3.y; sz += p3.z;
        }
        for (; e < total; e += R) {
            const SymRow p0 = *row(e);
            sx += p0.x; sy += p0.y; sz += p0.z;
        }
    }
    if constexpr (R > 1) {
#pragma unroll
        for (int m = 1; m < R; m <<= 1) {
            sx += __shfl_xor(sx, m, 64);
            sy += __shfl_xor(sy, m, 64);
            sz += __shfl_xor(sz, m, 64);
        }
    }
    if (!valid || r != 0) return;
    float4 nx, nv, na;
    leapfrog<float>(ld4(bodies + il), ld4(vel + il), ld4(acc + il), sx, sy, sz, dt, nx, nv, na);
    vel[il] = nv;                                                       // :281
    bodies[il] = nx;                                                    // :283
    acc[il] = na;                                                       // :290
    if (gout) gout[il] = float4{nx.x, nx.y, nx.z, G * nx.w};
}

}  // namespace nb
