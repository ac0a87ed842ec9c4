// nb_kernels.hip.h -- device code of the MI355X (gfx950) direct N-body engine.
//
// The reference runs ONE WGSL compute pass per frame (/root/reference nbody3d.js:219-292):
// tiled O(N^2) softened-gravity accumulation (:232-237 pair force, :255-272 tile loop)
// followed by the "velocity verlet with frame shift" update (:274-290), writing positions in
// place (:283) while other workgroups still stage them (:257) -- a cross-workgroup race.
// Here a step is well defined in both of its forms:
//
//   two kernels   K1 nb_force*   (reads positions only)  ->  K2 nb_integrate* (after all of K1)
//   one kernel    nb_step_fused  (reads bodies_in, writes bodies_out: ping-pong buffers)
//
//   K1 forms:   nb_force_symw<NG,J>      f32, SYMMETRIC pass (default from N ~ 13,000): every unordered pair once, both
//               nb_force_symw64<8>       accelerations; residents in registers, travelers rotate through the wave by DPP
//               nb_force_sym<WS,NG,J>    (f64 form; workgroup form with LDS-combined traveler sums: A/B arm)
//               nb_force_pk_sgpr<NG,WS>  f32, packed math, ordered pairs, j broadcast from SGPRs (shards without the native exchange)
//               nb_force_pk<NG,LS,TL>    f32, packed math, j-tile staged in LDS
//               nb_force<T,IPL,LS>       scalar template: f64, and the unpacked f32 shapes
//   fused form: nb_step_fused<NG,LS,TL>  nb_force_pk's loop over ALL j + the integrator in the
//                                        epilogue (no j-split across workgroups: LS lanes of a
//                                        wave share an i-body instead) -- one launch per step,
//                                        no partial sums through memory
//               nb_step_direct<MAXJ>     the same for N <= 2,048 with each lane's j-bodies loaded
//                                        straight into registers (no LDS tile, no barrier)
//               nb_step_jpk<WS>          A/B arm: packed across TWO j-BODIES streamed as SGPR pairs from a
//                                        pair-transposed position copy; j split over the waves of a workgroup
//                                        and over workgroups that meet at a ticket inside the launch
//
// CDNA4 mapping of the force loop (wave = 64 lanes, 4 SIMDs/CU, 160 KiB LDS/CU):
//   * a 256-thread workgroup (4 waves, one per SIMD) stages a j-tile of 256*TL bodies
//     (x, y, z, m) in LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write),
//     double buffered, ONE s_barrier per tile; the next tile lands while the current one computes
//     (the scalar template nb_force<T,...> stages (x, y, z, G*m) through registers);
//   * the inner loop reads the tile with ds_read_b128 at a wave-uniform address
//     (LDS broadcast: one read feeds 64*IPL pair evaluations) -- LS == 1 -- or
//     at LS consecutive addresses when LS lanes share one i-body;
//   * each lane keeps IPL i-bodies in VGPRs (register blocking), loaded with coalesced 16-B accesses;
//   * f32: the arithmetic is packed across TWO i-bodies of the lane (v_pk_add/fma/mul_f32):
//     per two pairs 3 v_pk_add, 3 v_pk_fma (r^2 + eps2), 2 v_pk_mul (cube), 2 v_rsq_f32,
//     1 v_pk_mul (m_j), 3 v_pk_fma (accumulate) = 12 packed (4 cycles each) + 2 transcendental
//     (8 cycles each) = 64 issue cycles per 128 pairs, issued stage-major over 4 independent
//     chains so no hazard s_nop is needed;
//   * no branch in the loop: with eps2 > 0 the self term is exactly 0*finite = 0 and bodies
//     past the range are staged as zero-mass;
//   * LS > 1: the LS partial sums of a body are reduced inside the wave (DPP row operations
//     and row broadcasts for f32, wavefront shuffles for f64) before ONE lane stores/integrates;
//   * grid = (i-blocks, jsplit): j may also be partitioned over blockIdx.y (any multiple of 8
//     bodies per split); K2 sums the jsplit partials in ascending order (deterministic, no atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nb_plan.h"

namespace nb {

template <typename T> struct vec4;
template <> struct vec4<float> { using type = float4; };
template <> struct vec4<double> { using type = double4; };

// Diagnostic build only (tools/ubench4.hip, -DNB_STAMPS): per-wave s_memtime stamps at the phase
// boundaries of the LDS-tile kernels, written to a buffer nothing else reads.  The product build
// compiles NB_STAMP to nothing (MI355X_MICROARCH.md 'DVFS give-back' item 6: no stamp executes
// in the real kernel).
#ifdef NB_STAMPS
__device__ unsigned long long* nb_stamp_buf;
#define NB_STAMP(k)                                                                                         \
    do {                                                                                                    \
        unsigned long long t_;                                                                              \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        if ((threadIdx.x & 63) == 0) nb_stamp_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k)] = t_; \
        if ((k) == 0) {                 /* where the wave runs: HW_ID (wave/simd/cu/sh/se) and XCC_ID */       \
            unsigned hw_, xcc_;                                                                                 \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xcc_)); \
            if ((threadIdx.x & 63) == 0) nb_stamp_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 7] = ((unsigned long long)xcc_ << 32) | hw_; \
        }                                                                                                   \
        if ((k) == 0 || (k) == 4) {     /* 100 MHz wall clock beside the first and last stamp */                \
            unsigned long long r_;                                                                              \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_)::"memory");                    \
            if ((threadIdx.x & 63) == 0) nb_stamp_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + ((k) == 0 ? 5 : 6)] = r_; \
        }                                                                                                   \
    } while (0)
// without the drain: for points inside the tile loop
#define NB_STAMP_LIGHT(k)                                                                                   \
    do {                                                                                                    \
        unsigned long long t_;                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
        if ((threadIdx.x & 63) == 0) nb_stamp_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k)] = t_; \
    } while (0)
#else
#define NB_STAMP(k) do { } while (0)
#define NB_STAMP_LIGHT(k) do { } while (0)
#endif

// kBlock (threads per workgroup = reference TILE_SIZE) and kTile (j-bodies per LDS tile unit): nb_plan.h

// Whole-row global loads.  HIP's float4/double4 are structs of scalars: a plain `bodies[j]` is
// four scalar loads that the backend re-merges as it sees fit (seen: dwordx2 + dwordx3 + dwordx2
// for one row).  Going through the native vector type keeps ONE global_load_dwordx4 (two for f64).
typedef float nb_v4f __attribute__((ext_vector_type(4)));
typedef double nb_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float4* p)
{
    const nb_v4f v = *reinterpret_cast<const nb_v4f*>(p);
    return float4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ double4 ld4(const double4* p)
{
    const nb_v4d v = *reinterpret_cast<const nb_v4d*>(p);
    return double4{v.x, v.y, v.z, v.w};
}

// x, y, z of a row as ONE global_load_dwordx3: for i-rows whose mass is never used.  (With a dwordx4 the backend
// recycles the dead fourth register while the load is still in flight and has to wait for it first --
// seen as an s_waitcnt vmcnt(1) between the i-row loads of nb_force_pk_sgpr, which serialised them.)
typedef float nb_v3f __attribute__((ext_vector_type(3)));
__device__ __forceinline__ nb_v3f ld3(const float4* p) { return *reinterpret_cast<const nb_v3f*>(p); }

__device__ __forceinline__ float nb_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double nb_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Which j-splits a launch covers.  A launch normally covers all of them
// (base 0, no hole).  The overlapped multi-GPU step issues the splits that lie
// inside the rank's OWN rows first (base = first own split) and, once the
// all-gather of the other ranks' rows has landed, the rest (hole = own splits).
struct SplitWindow {
    uint32_t base, hole_begin, hole_count;
    __device__ __forceinline__ uint32_t split(uint32_t y) const
    {
        uint32_t b = y + base;
        if (b >= hole_begin) b += hole_count;
        return b;
    }
};

// One pair: nbody3d.js:232-237 with b.w already multiplied by G at staging
// time ((G*m)*inv is the reference's left-associated product, :236).
__device__ __forceinline__ void pair(const float bx, const float by, const float bz, const float bgm, const float xi,
                                     const float yi, const float zi, const float eps2, float& ax, float& ay, float& az)
{
    const float dx = bx - xi, dy = by - yi, dz = bz - zi;                    // :233
    const float d2 = nb_fma(dz, dz, nb_fma(dy, dy, nb_fma(dx, dx, eps2)));    // :234 (contracted; WGSL permits it)
    const float d6 = d2 * d2 * d2;                                           // :235
    const float s = bgm * __builtin_amdgcn_rsqf(d6);                         // :235-236, bare v_rsq_f32 (1 ulp)
    ax = nb_fma(s, dx, ax);                                                  // :266
    ay = nb_fma(s, dy, ay);
    az = nb_fma(s, dz, az);
}

// f64 pair.  v_rsq_f64 costs 16 issue cycles and every other DP instruction 4 (measured,
// profiles/r02/ubench3_*.txt), so the body is built to need the fewest DP instructions:
//   y0 = v_rsq_f64(d2)  (relative error |e|/2, e = 1 - d2*y0^2, |e| <~ 2^-26)
//   d2^(-3/2) = y0^3 (1 - e)^(-3/2) = y0^3 (1 + 3e/2 + 15e^2/8 + ...)   -> first order: error < 2 e^2 ~ 4e-16
// = 15 DP instructions + the seed per pair (round 1: d2^3, seed, one Newton step = 16 + seed + a clamp).
// d2 must stay finite (|x| < 1e150): an infinite d2 would give 0*inf in e.
__device__ __forceinline__ void pair(const double bx, const double by, const double bz, const double bgm,
                                     const double xi, const double yi, const double zi, const double eps2, double& ax,
                                     double& ay, double& az)
{
    const double dx = bx - xi, dy = by - yi, dz = bz - zi;
    const double d2 = nb_fma(dz, dz, nb_fma(dy, dy, nb_fma(dx, dx, eps2)));
    const double y = __builtin_amdgcn_rsq(d2);
    const double y2 = y * y;
    const double e = nb_fma(-d2, y2, 1.0);
    const double p3 = (bgm * y) * y2;
    const double s = nb_fma(p3 * e, 1.5, p3);
    ax = nb_fma(s, dx, ax);
    ay = nb_fma(s, dy, ay);
    az = nb_fma(s, dz, az);
}

// ---- sum over the LS consecutive lanes that share an i-body ------------------------------
// f32: DPP row operations inside a 16-lane row (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror: one v_add_f32 with a DPP operand per step, no LDS traffic), then row_bcast15 /
// row_bcast31 across rows.  The full sum is valid in the LAST lane of the group (js == LS-1);
// for LS <= 16 in every lane.  Fixed order: deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}
template <int LS>
__device__ __forceinline__ float group_sum(float v)
{
    if constexpr (LS >= 2) v = dpp_add<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    if constexpr (LS >= 4) v = dpp_add<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    if constexpr (LS >= 8) v = dpp_add<0x141, 0xF>(v);    // row_half_mirror
    if constexpr (LS >= 16) v = dpp_add<0x140, 0xF>(v);   // row_mirror
    if constexpr (LS >= 32) v = dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    if constexpr (LS >= 64) v = dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return v;
}
// The same reduction for NV values at once, step-major: the NV adds of a step are independent, so
// no DPP hazard wait falls between them (value by value every add waits on the one before).
// Each add is ONE v_add_f32_dpp (the DPP-selected lane is the add's first operand); written as asm because
// hipcc emits v_mov_b32_dpp + v_add_f32 for update_dpp + add (it cannot fold a +0.0 `old` into an fadd:
// 72 instead of 36 instructions per wave for a pair of bodies shared by 64 lanes).  Lanes a row mask switches
// off keep their value (update_dpp gave them v + 0).  volatile: the statements keep this step-major order, so
// an add reads a register written at least NV >= 6 instructions earlier (a DPP read needs 2 wait states after
// the VALU write, and hipcc inserts none in front of asm); the s_nop covers the first step.
template <int CTRL>
__device__ __forceinline__ void dpp_add_inplace(float& v)
{
    if constexpr (CTRL == 0xB1) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    else if constexpr (CTRL == 0x4E) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    else if constexpr (CTRL == 0x141) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    else if constexpr (CTRL == 0x140) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    else if constexpr (CTRL == 0x142) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    else if constexpr (CTRL == 0x143) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
}
template <int LS, int NV>
__device__ __forceinline__ void group_sum_all(float (&v)[NV])
{
    static_assert(NV >= 6, "step-major order is what keeps dependent DPP adds apart");
    if constexpr (LS >= 2) asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[NV - 1]));
#define NB_DPP_STEP(MIN_LS, CTRL)                                        \
    if constexpr (LS >= MIN_LS) {                                        \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) dpp_add_inplace<CTRL>(v[i]); \
    }
    NB_DPP_STEP(2, 0xB1)
    NB_DPP_STEP(4, 0x4E)
    NB_DPP_STEP(8, 0x141)
    NB_DPP_STEP(16, 0x140)
    NB_DPP_STEP(32, 0x142)
    NB_DPP_STEP(64, 0x143)
#undef NB_DPP_STEP
}
template <int LS>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = 1; m < LS; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---- the integrator: nbody3d.js:274-290 on all four components (the .w lane is integrated
// too, exactly as the reference does; mass stays constant because vel.w = 0) -------------------
template <typename T>
__device__ __forceinline__ void leapfrog(const typename vec4<T>::type& x, const typename vec4<T>::type& v,
                                         const typename vec4<T>::type& ao, const T ax, const T ay, const T az, const T dt,
                                         typename vec4<T>::type& nx, typename vec4<T>::type& nv,
                                         typename vec4<T>::type& na)
{
    na.x = ax; na.y = ay; na.z = az;
    na.w = 0;                                                           // :274
    const T h = dt * T(0.5);                                            // :276
    nv.x = nb_fma(ao.x + na.x, h, v.x);                                 // :280
    nv.y = nb_fma(ao.y + na.y, h, v.y);
    nv.z = nb_fma(ao.z + na.z, h, v.z);
    nv.w = nb_fma(ao.w + na.w, h, v.w);
    nx.x = nb_fma(nb_fma(h, na.x, nv.x), dt, x.x);                      // :283
    nx.y = nb_fma(nb_fma(h, na.y, nv.y), dt, x.y);
    nx.z = nb_fma(nb_fma(h, na.z, nv.z), dt, x.z);
    nx.w = nb_fma(nb_fma(h, na.w, nv.w), dt, x.w);
}

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
                else r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
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
            for (int c = 0; c < 4; ++c) r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
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
            for (int c = 0; c < NG; ++c) r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
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
                    for (int c = 0; c < 4; ++c) r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
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

// The same pass with the WAVE as the unit of work (no LDS, no barrier): a super-block is one wave's 128*NG residents, and
// the chunk lists of all super-blocks, laid end to end (L chunk-sweeps), are cut into W contiguous ranges of floor/ceil(L/W)
// sweeps -- one per wave, W a multiple of the chip's SIMD count -- so every SIMD gets the same work to within ONE sweep at
// any N (the workgroup form above needs nsb * Q to land on a multiple of the CU count).  A wave whose range crosses into the
// next super-block stores its resident sums, reloads its residents and goes on; its resident sums of super-block g go to
// layer r_layer0 + (w - first wave of g) (table `gtab`: first wave and wave count per super-block, built by the host).
// struct SymWPlan: nb_plan.h

template <int NG, int J>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NG > 4 ? 2 : 4, NG > 4 ? 2 : (NG < 4 ? 8 : 4))))
void nb_force_symw(const float4* __restrict__ bodies, SymRow* __restrict__ partial, const uint32_t* __restrict__ gtab, const SymWPlan pl,
                   const uint32_t n, const float eps2)
{
    constexpr uint32_t S = 128u * NG;          // rows per super-block = one wave's residents
    constexpr int GW = NG < 4 ? NG : 4;        // packed groups evaluated stage-major together
    constexpr uint32_t CH = 64u * J;           // travelers per chunk
    constexpr uint32_t CPS = S / CH;           // chunks per super-block
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));     // the four waves of a workgroup are independent
    if (w >= pl.W) return;
    uint32_t p = pl.p0 + (uint32_t)(((uint64_t)w * pl.L) / pl.W);
    const uint32_t pend = pl.p0 + (uint32_t)(((uint64_t)(w + 1) * pl.L) / pl.W);
    const nb_f2 e2 = nb_f2{eps2, eps2};
    const uint32_t first_lo = pl.n_hi * pl.total_hi;

    while (p < pend) {
        // which super-block's list p lies in, and where
        uint32_t g, k, total;
        if (p < first_lo) { g = p / pl.total_hi; k = p - g * pl.total_hi; total = pl.total_hi; }
        else { const uint32_t r = p - first_lo; g = pl.n_hi + r / pl.total_lo; k = r - (g - pl.n_hi) * pl.total_lo; total = pl.total_lo; }
        const uint32_t ring = total - CPS;                           // symmetric chunks of g; CPS resident-only chunks follow
        uint32_t kend = k + (pend - p);
        if (kend > total) kend = total;
        p += kend - k;

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
        for (; k < kend; ++k) {
            const bool sym = k < ring;
            const uint32_t d = k / CPS;                              // ring distance - 1 (symmetric chunks)
            uint32_t tb = g + 1 + d;
            if (tb >= pl.nsb) tb -= pl.nsb;
            const uint32_t tstart = sym ? tb * S + (k % CPS) * CH : g * S + (k - ring) * CH;
            if (tstart >= n) continue;   // a chunk of padding rows only (zero mass): exerts nothing, and nobody reads its sums
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
                        for (int c = 0; c < GW; ++c) r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
#pragma unroll
                        for (int c = 0; c < GW; ++c) si[c] = pm * r[c];                // (G m_t) inv: resident side, :236
#pragma unroll
                        for (int c = 0; c < GW; ++c) sj[c] = mi[c0g + c] * r[c];       // (G m_i) inv: traveler side
#pragma unroll
                        for (int c = 0; c < GW; ++c) ax[c0g + c] = __builtin_elementwise_fma(si[c], dx[c], ax[c0g + c]);
#pragma unroll
                        for (int c = 0; c < GW; ++c) ay[c0g + c] = __builtin_elementwise_fma(si[c], dy[c], ay[c0g + c]);
#pragma unroll
                        for (int c = 0; c < GW; ++c) az[c0g + c] = __builtin_elementwise_fma(si[c], dz[c], az[c0g + c]);
#pragma unroll
                        for (int c = 0; c < GW; ++c) bx[u] = __builtin_elementwise_fma(-sj[c], dx[c], bx[u]);   // x_i - x_t = -(x_t - x_i), exactly
#pragma unroll
                        for (int c = 0; c < GW; ++c) by[u] = __builtin_elementwise_fma(-sj[c], dy[c], by[u]);
#pragma unroll
                        for (int c = 0; c < GW; ++c) bz[u] = __builtin_elementwise_fma(-sj[c], dz[c], bz[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < J; ++u) {                        // the travelers and their sums move on by one lane
                    tx[u] = wave_rot1(tx[u]); ty[u] = wave_rot1(ty[u]); tz[u] = wave_rot1(tz[u]); tm[u] = wave_rot1(tm[u]);
                    bx[u] = nb_f2{wave_rot1(bx[u].x), wave_rot1(bx[u].y)};
                    by[u] = nb_f2{wave_rot1(by[u].x), wave_rot1(by[u].y)};
                    bz[u] = nb_f2{wave_rot1(bz[u].x), wave_rot1(bz[u].y)};
                }
            }
            if (sym) {
                SymRow* out = partial + (size_t)(pl.t_layer0 + d) * pl.np + tstart + lane;
#pragma unroll
                for (int u = 0; u < J; ++u) out[u * 64] = SymRow{bx[u].x + bx[u].y, by[u].x + by[u].y, bz[u].x + bz[u].y};
            }
        }
        // resident sums of this wave's part of g's list
        SymRow* out = partial + (size_t)(pl.r_layer0 + (w - gtab[2 * g])) * pl.np + (size_t)g * S + lane;
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            out[(2 * c) * 64] = SymRow{ax[c].x, ay[c].x, az[c].x};
            out[(2 * c + 1) * 64] = SymRow{ax[c].y, ay[c].y, az[c].y};
        }
    }
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

template <int IPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void nb_force_symw64(const double4* __restrict__ bodies, SymRowT<double>* __restrict__ partial, const uint32_t* __restrict__ gtab,
                     const SymWPlan pl, const uint32_t n, const double G, const double eps2)
{
    constexpr uint32_t S = 64u * IPL, CH = 64u, CPS = S / CH;
    constexpr int GW = 4;                      // residents evaluated stage-major together
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (w >= pl.W) return;
    uint32_t p = pl.p0 + (uint32_t)(((uint64_t)w * pl.L) / pl.W);
    const uint32_t pend = pl.p0 + (uint32_t)(((uint64_t)(w + 1) * pl.L) / pl.W);
    const uint32_t first_lo = pl.n_hi * pl.total_hi;
    while (p < pend) {
        uint32_t g, k, total;
        if (p < first_lo) { g = p / pl.total_hi; k = p - g * pl.total_hi; total = pl.total_hi; }
        else { const uint32_t r = p - first_lo; g = pl.n_hi + r / pl.total_lo; k = r - (g - pl.n_hi) * pl.total_lo; total = pl.total_lo; }
        const uint32_t ring = total - CPS;
        uint32_t kend = k + (pend - p);
        if (kend > total) kend = total;
        p += kend - k;
        double xi[IPL], yi[IPL], zi[IPL], mi[IPL], ax[IPL], ay[IPL], az[IPL];
#pragma unroll
        for (int c = 0; c < IPL; ++c) {
            const double4 b = ld4(bodies + (size_t)g * S + c * 64 + lane);
            xi[c] = b.x; yi[c] = b.y; zi[c] = b.z; mi[c] = b.w * G;
            ax[c] = 0; ay[c] = 0; az[c] = 0;
        }
        for (; k < kend; ++k) {
            const bool sym = k < ring;
            const uint32_t d = k / CPS;
            uint32_t tb = g + 1 + d;
            if (tb >= pl.nsb) tb -= pl.nsb;
            const uint32_t tstart = sym ? tb * S + (k % CPS) * CH : g * S + (k - ring) * CH;
            if (tstart >= n) continue;
            const double4 t = ld4(bodies + tstart + lane);
            double tx = t.x, ty = t.y, tz = t.z, tm = t.w * G, bx = 0, by = 0, bz = 0;
            for (int st = 0; st < 64; ++st) {
#pragma unroll
                for (int c0g = 0; c0g < IPL; c0g += GW) {            // stage-major over four residents
                    double dx[GW], dy[GW], dz[GW], d2[GW], y[GW], u[GW];
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
                        u[c] = nb_fma(t3 * e, 1.5, t3);
                    }
#pragma unroll
                    for (int c = 0; c < GW; ++c) {
                        const double si = tm * u[c], sj = mi[c0g + c] * u[c];
                        ax[c0g + c] = nb_fma(si, dx[c], ax[c0g + c]); ay[c0g + c] = nb_fma(si, dy[c], ay[c0g + c]); az[c0g + c] = nb_fma(si, dz[c], az[c0g + c]);
                        bx = nb_fma(-sj, dx[c], bx); by = nb_fma(-sj, dy[c], by); bz = nb_fma(-sj, dz[c], bz);
                    }
                }
                tx = wave_rot1(tx); ty = wave_rot1(ty); tz = wave_rot1(tz); tm = wave_rot1(tm);
                bx = wave_rot1(bx); by = wave_rot1(by); bz = wave_rot1(bz);
            }
            if (sym) partial[(size_t)(pl.t_layer0 + d) * pl.np + tstart + lane] = SymRowT<double>{bx, by, bz};
        }
        SymRowT<double>* out = partial + (size_t)(pl.r_layer0 + (w - gtab[2 * g])) * pl.np + (size_t)g * S + lane;
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
__global__ __launch_bounds__(kBlock) void nb_sym_reduce(const SymRowT<T>* __restrict__ partial, const uint32_t* __restrict__ gtab,
                                                       typename vec4<T>::type* __restrict__ A, const SymWPlan pl, uint32_t S, uint32_t g0, uint32_t g1)
{
    using SymRow = SymRowT<T>;
    using V4 = typename vec4<T>::type;
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= pl.np) return;
    const uint32_t b = j / S;
    T sx = 0, sy = 0, sz = 0;
    if (b >= g0 && b < g1) {
        const uint32_t nr = gtab[2 * b + 1];
        for (uint32_t e = 0; e < nr; ++e) { const SymRow r = partial[(size_t)(pl.r_layer0 + e) * pl.np + j]; sx += r.x; sy += r.y; sz += r.z; }
    }
    if (g1 - g0 > pl.H) {
        for (uint32_t d = 0; d <= pl.H; ++d) {                   // few ring distances, many own super-blocks: ascending distance
            uint32_t g = b + pl.nsb - 1 - d;
            if (g >= pl.nsb) g -= pl.nsb;
            if (g < g0 || g >= g1 || d >= pl.H + (g < pl.n_hi ? 1u : 0u)) continue;
            const SymRow r = partial[(size_t)(pl.t_layer0 + d) * pl.np + j];
            sx += r.x; sy += r.y; sz += r.z;
        }
    } else {
        for (uint32_t g = g0; g < g1; ++g) {                     // a rank of many: only its own super-blocks can have written a layer of row j
            uint32_t d = b + pl.nsb - 1 - g;
            if (d >= pl.nsb) d -= pl.nsb;
            if (d >= pl.H + (g < pl.n_hi ? 1u : 0u)) continue;
            const SymRow r = partial[(size_t)(pl.t_layer0 + d) * pl.np + j];
            sx += r.x; sy += r.y; sz += r.z;
        }
    }
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

// K2 for the wave-granular form: resident layers gtab[2g+1] (waves that worked on g's list), then the traveler layers.
template <typename T, int R>
__global__ __launch_bounds__(kBlock) void nb_integrate_symw(typename vec4<T>::type* __restrict__ bodies, typename vec4<T>::type* __restrict__ vel,
                                                           typename vec4<T>::type* __restrict__ acc, const SymRowT<T>* __restrict__ partial,
                                                           const uint32_t* __restrict__ gtab, uint32_t n, const SymWPlan pl, uint32_t S, T dt,
                                                           typename vec4<T>::type* __restrict__ gout, T G)
{
    using V4 = typename vec4<T>::type;
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t il = gid / R, r = gid % R;
    const bool valid = il < n;
    T sx = 0, sy = 0, sz = 0;
    if (valid) {
        const uint32_t b = il / S;
        const uint32_t nr = gtab[2 * b + 1];
        const uint32_t nt = pl.H + ((pl.n_hi && b >= pl.n_hi) ? 1u : 0u);
        const uint32_t total = nr + nt;
        auto row = [&](uint32_t e) { return partial + (size_t)(e < nr ? pl.r_layer0 + e : pl.t_layer0 + (e - nr)) * pl.np + il; };
        uint32_t e = r;
        for (; e + 3 * R < total; e += 4 * R) {
            const SymRowT<T> p0 = *row(e), p1 = *row(e + R), p2 = *row(e + 2 * R), p3 = *row(e + 3 * R);
            sx += p0.x; sy += p0.y; sz += p0.z;
            sx += p1.x; sy += p1.y; sz += p1.z;
            sx += p2.x; sy += p2.y; sz += p2.z;
            sx += p3.x; sy += p3.y; sz += p3.z;
        }
        for (; e < total; e += R) {
            const SymRowT<T> p0 = *row(e);
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
    V4 nx, nv, na;
    leapfrog<T>(ld4(bodies + il), ld4(vel + il), ld4(acc + il), sx, sy, sz, dt, nx, nv, na);
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
        for (int c = 0; c < 4; ++c) r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
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

// K2.  nbody3d.js:274-290.  R lanes cooperate on one body: lane r sums partials r, r+R,
// r+2R, ... (ascending, independent 16-B loads in flight), the R sums are combined by wavefront
// shuffles in a fixed order (deterministic), and lane 0 of the group applies the update.
// With jsplit = 64 partials a single lane per body is latency-bound (20 us at
// 16,384 rows); R = 8 brings it to the launch floor.
template <typename T, int R>
__global__ __launch_bounds__(kBlock) void nb_integrate(typename vec4<T>::type* __restrict__ bodies,
                                                      typename vec4<T>::type* __restrict__ vel,
                                                      typename vec4<T>::type* __restrict__ acc,
                                                      const typename vec4<T>::type* __restrict__ partial,
                                                      uint32_t i_begin, uint32_t i_count, uint32_t jsplit, T dt,
                                                      typename vec4<T>::type* __restrict__ gout, T G)
{
    using V4 = typename vec4<T>::type;
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t il = gid / R;
    const uint32_t r = gid % R;
    const bool valid = il < i_count;
    T sx = 0, sy = 0, sz = 0;
    if (valid) {
        uint32_t sp = r;
        // 4 independent loads per trip, summed in ascending split order
        for (; sp + 3 * R < jsplit; sp += 4 * R) {
            const V4 p0 = ld4(partial + (size_t)sp * i_count + il);
            const V4 p1 = ld4(partial + (size_t)(sp + R) * i_count + il);
            const V4 p2 = ld4(partial + (size_t)(sp + 2 * R) * i_count + il);
            const V4 p3 = ld4(partial + (size_t)(sp + 3 * R) * i_count + il);
            sx += p0.x; sy += p0.y; sz += p0.z;
            sx += p1.x; sy += p1.y; sz += p1.z;
            sx += p2.x; sy += p2.y; sz += p2.z;
            sx += p3.x; sy += p3.y; sz += p3.z;
        }
        for (; sp < jsplit; sp += R) {
            const V4 p = ld4(partial + (size_t)sp * i_count + il);
            sx += p.x; sy += p.y; sz += p.z;
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
    leapfrog<T>(ld4(bodies + i_begin + il), ld4(vel + il), ld4(acc + il), sx, sy, sz, dt, nx, nv, na);
    vel[il] = nv;                                                       // :281
    bodies[i_begin + il] = nx;                                          // :283
    acc[il] = na;                                                       // :290
    if (gout) gout[i_begin + il] = V4{nx.x, nx.y, nx.z, G * nx.w};      // the packed f32 K1's j-stream row (G != 1 only)
}

// K2 for jsplit == 1 with the a_old / a_new buffers swapped by pointer (SURVEY.md §8(d) "K2
// roofline": read x, v, a_old, a_new = 64 B, write x, v = 32 B -> 96 B per body, nothing else):
// `anew` is K1's single partial array and becomes the next step's `aold` on the host side.
template <typename T>
__global__ __launch_bounds__(kBlock) void nb_integrate_swap(typename vec4<T>::type* __restrict__ bodies,
                                                           typename vec4<T>::type* __restrict__ vel,
                                                           const typename vec4<T>::type* __restrict__ aold,
                                                           const typename vec4<T>::type* __restrict__ anew,
                                                           uint32_t i_begin, uint32_t i_count, T dt,
                                                           typename vec4<T>::type* __restrict__ gout, T G)
{
    using V4 = typename vec4<T>::type;
    const uint32_t il = blockIdx.x * kBlock + threadIdx.x;
    if (il >= i_count) return;
    const V4 a = ld4(anew + il);
    V4 nx, nv, na;
    leapfrog<T>(ld4(bodies + i_begin + il), ld4(vel + il), ld4(aold + il), a.x, a.y, a.z, dt, nx, nv, na);
    vel[il] = nv;
    bodies[i_begin + il] = nx;
    if (gout) gout[i_begin + il] = V4{nx.x, nx.y, nx.z, G * nx.w};
}

// Viewer frame: what the reference's render pass reads every frame -- bodies
// (x, y, z, mass -> billboard position and radius, nbody3d.js:331,345) and the speed
// length(vel.xyz) that feeds its colour map (:380) -- packed as f32 into a staging buffer the
// step stream never writes again, so the copy to the host can run beside the next steps.
template <typename T>
__global__ __launch_bounds__(kBlock) void nb_frame_pack(const typename vec4<T>::type* __restrict__ bodies,
                                                       const typename vec4<T>::type* __restrict__ vel, uint32_t n,
                                                       uint32_t i_begin, uint32_t i_count, float4* __restrict__ out_b,
                                                       float* __restrict__ out_speed)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const auto b = bodies[i];
        out_b[i] = float4{(float)b.x, (float)b.y, (float)b.z, (float)b.w};
    }
    if (i < i_count) {
        const auto v = vel[i];
        const float vx = (float)v.x, vy = (float)v.y, vz = (float)v.z;
        out_speed[i_begin + i] = __builtin_sqrtf(vx * vx + vy * vy + vz * vz);
    }
}

// Diagnostics (no reference analogue): per-block fp64 partial
// sums of kinetic energy, momentum, and the shard's share of the softened
// potential; finished on the host (a few hundred doubles).
template <typename T>
__global__ __launch_bounds__(kBlock) void nb_diag(const typename vec4<T>::type* __restrict__ bodies,
                                                 const typename vec4<T>::type* __restrict__ vel, uint32_t n,
                                                 uint32_t i_begin, uint32_t i_count, double G, double eps2,
                                                 double* __restrict__ out /* [gridDim.x][5] */)
{
    using V4 = typename vec4<T>::type;
    __shared__ V4 tile[kTile];
    __shared__ double red[5][kBlock / 64];
    const int tid = threadIdx.x;
    const uint32_t il = blockIdx.x * kBlock + tid;
    const bool valid = il < i_count;
    V4 bi = V4{0, 0, 0, 0}, vi = V4{0, 0, 0, 0};
    if (valid) { bi = bodies[i_begin + il]; vi = vel[il]; }
    double pot = 0.0;
    for (uint32_t j0 = 0; j0 < n; j0 += kTile) {
        const uint32_t j = j0 + tid;
        tile[tid] = (j < n) ? bodies[j] : V4{0, 0, 0, 0};
        __syncthreads();
        double p = 0.0;
#pragma unroll 4
        for (int jj = 0; jj < kTile; ++jj) {
            const V4 b = tile[jj];
            const double dx = (double)b.x - (double)bi.x, dy = (double)b.y - (double)bi.y, dz = (double)b.z - (double)bi.z;
            const double r2 = dx * dx + dy * dy + dz * dz;
            // exclude the self term exactly (j == i), keep everything else
            const double w = (j0 + jj == i_begin + il) ? 0.0 : (double)b.w;
            p += w * rsqrt(r2 + eps2);
        }
        pot += p;
        __syncthreads();
    }
    double vals[5];
    const double m = valid ? (double)bi.w : 0.0;
    vals[0] = 0.5 * m * ((double)vi.x * vi.x + (double)vi.y * vi.y + (double)vi.z * vi.z);
    vals[1] = valid ? -0.5 * G * m * pot : 0.0;
    vals[2] = m * vi.x; vals[3] = m * vi.y; vals[4] = m * vi.z;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double v = vals[q];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
        if ((tid & 63) == 0) red[q][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 5) {
        double v = 0;
        for (int w = 0; w < kBlock / 64; ++w) v += red[tid][w];
        out[(size_t)blockIdx.x * 5 + tid] = v;
    }
}

}  // namespace nb
