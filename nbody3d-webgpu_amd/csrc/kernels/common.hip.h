// kernels/common.hip.h -- shared device helpers: row loads, the pair interaction (nbody3d.js:232-237), in-wave sums, the integrator (:274-290).
// Part of nb_kernels.hip.h (include that, not this file).
#pragma once

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

// v_rsq_f32 in its 64-bit (VOP3) encoding.  On gfx950 a 64-bit instruction that does not start on an 8-byte boundary issues more
// slowly (profiles/r04/README.md "instruction alignment": the symmetric pass's loop +12 % at one wave per SIMD, +5 % at two), the
// hot loops are all packed-f32 / DPP instructions (64-bit encodings), and a 32-bit v_rsq_f32_e32 among them flips the parity of
// everything behind it.  An |x| source modifier needs the VOP3 form, so the compiler emits v_rsq_f32_e64 -- still ITS instruction
// (it keeps the wait state gfx950 wants between a transcendental and a reader of its result; an asm statement would not get it).
// The argument is a cube of r^2 + eps2 > 0: |x| = x, the value is bit for bit v_rsq_f32(x).
__device__ __forceinline__ float nb_rsq(float x) { return __builtin_amdgcn_rsqf(__builtin_fabsf(x)); }

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

}  // namespace nb
