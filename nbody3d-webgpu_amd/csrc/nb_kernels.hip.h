// nb_kernels.hip.h -- device code of the MI355X (gfx950) direct N-body engine.
//
// Two kernels replace the reference's single WGSL compute pass
// (/root/reference nbody3d.js:219-292):
//
//   K1 nb_force*            tiled O(N^2) softened-gravity accumulation
//                           (nbody3d.js:232-237 pair force, :255-272 tile loop), three forms:
//                             nb_force_pk_sgpr<NG>   f32, packed math, j broadcast from SGPRs (default for
//                                                    large systems: +3..4 % over the LDS tile)
//                             nb_force_pk<NG,LS>     f32, packed math, j-tile staged in LDS
//                             nb_force<T,IPL,LS>     scalar template: f64, and LS lanes per body (small N)
//   K2 nb_integrate<T>      the "velocity verlet with frame shift" update
//                           (nbody3d.js:274-290)
//
// The split is what makes a step well defined: the reference writes positions
// in place (:283) while other workgroups still stage them (:257); here K1 only
// reads positions and K2 only runs after every K1 block has finished.
//
// CDNA4 mapping of K1 (wave = 64 lanes, 4 SIMDs/CU, 160 KiB LDS/CU):
//   * a 256-thread workgroup (4 waves, one per SIMD) stages a 256-body j-tile
//     (x, y, z, G*m) in LDS, double buffered, ONE s_barrier per tile; the next
//     tile's global_load_dwordx4 is in flight while the current tile computes;
//   * the inner loop reads the tile with ds_read_b128 at a wave-uniform address
//     (LDS broadcast: one read feeds 64*IPL pair evaluations) -- LS == 1 -- or
//     at LS consecutive addresses when LS lanes share one i-body (small N);
//   * each lane keeps IPL i-bodies in VGPRs (register blocking: 1 LDS read per
//     IPL*64 pairs), loaded with coalesced 16-B accesses (lane stride 16 B);
//   * f32, nb_force_pk / nb_force_pk_sgpr: the arithmetic is packed across TWO i-bodies of
//     the lane (v_pk_add/fma/mul_f32): per two pairs 3 v_pk_add, 3 v_pk_fma
//     (r^2 + eps2), 2 v_pk_mul (cube), 2 v_rsq_f32, 1 v_pk_mul (G*m_j), 3 v_pk_fma
//     (accumulate) = 12 packed (4 cycles each) + 2 transcendental (8 cycles each)
//     = 64 issue cycles per 128 pairs, issued stage-major over 4 independent
//     chains so no hazard s_nop is needed.  nb_force is the scalar template
//     (13 VALU per pair) used for f64 and for the LS > 1 shapes;
//   * no branch in the loop: with eps2 > 0 the self term is exactly 0*finite = 0
//     and bodies past the split are staged as zero-mass;
//   * when LS > 1 the LS partial sums of a body are reduced with wavefront
//     shuffles before one lane stores;
//   * grid = (i-blocks, jsplit): j is also partitioned over blockIdx.y (any
//     multiple of 8 bodies per split, exact trip count on the last partial tile)
//     so that small i-counts (N = 65,536, or a 1/8 shard) still fill every SIMD;
//     K2 sums the jsplit partials in ascending order (deterministic, no atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nb {

template <typename T> struct vec4;
template <> struct vec4<float> { using type = float4; };
template <> struct vec4<double> { using type = double4; };

constexpr int kBlock = 256;  // threads per workgroup = reference TILE_SIZE (nbody3d.js:4,240)
constexpr int kTile = 256;   // j-bodies per LDS tile (nbody3d.js:229)

__device__ __forceinline__ float nb_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }  // bare v_rsq_f32 (1 ulp)
// v_rsq_f64 seed (~2^-26) + one Newton step y(1 + e/2), e = 1 - x y^2: 4 DP ops
// instead of ocml rsqrt()'s 5 + class test + 2 selects.  x is clamped so an
// overflowed d^6 cannot turn into inf*0 = NaN (such pairs then contribute ~1e-150*m,
// i.e. nothing; the f32 path gets the reference's exact 0 from v_rsq_f32(inf)).
__device__ __forceinline__ double nb_rsqrt(double x)
{
    x = __builtin_fmin(x, 1e300);
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * e, 0.5, y);
}
__device__ __forceinline__ float nb_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double nb_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Which j-splits a launch covers.  A launch normally covers all of them
// (base 0, no hole).  The overlapped multi-GPU step issues the splits that lie
// inside the rank's OWN rows first (base = first own split) and, once the
// all-gather of the other ranks' rows has landed, the rest (hole = own splits).
struct SplitWindow {
    uint32_t base, hole_begin, hole_count;
    uint32_t xcd_remap;   // 1: XCD-aware workgroup mapping (see xcd_remap below)
    __device__ __forceinline__ uint32_t split(uint32_t y) const
    {
        uint32_t b = y + base;
        if (b >= hole_begin) b += hole_count;
        return b;
    }
};

// XCD-aware workgroup -> (i-block, j-split) mapping.  MI355X deals workgroups round-robin
// over its 8 XCDs in dispatch order (x fastest), and every XCD has its own 4 MiB L2.  With
// the plain mapping each j-split (one 16-B row per body, streamed by all gridDim.x i-blocks)
// is pulled into all 8 L2s; with this remap all workgroups that stream a given j-split sit
// on ONE XCD (splits k, k+8, k+16, ... belong to XCD k), so the replicated bodies array is
// fetched once per step instead of 8 times (FETCH_SIZE 33.7 MB -> 4.x MB at N=262,144).
// Placement only changes speed/traffic, never results (MI355X_MICROARCH.md, XCD placement).
__device__ __forceinline__ void xcd_remap(uint32_t& bx, uint32_t& by, uint32_t enable)
{
    const uint32_t gx = gridDim.x, gy = gridDim.y;
    bx = blockIdx.x; by = blockIdx.y;
    if (enable && (gy & 7u) == 0) {
        const uint32_t lin = bx + by * gx;
        const uint32_t xcd = lin & 7u, slot = lin >> 3;
        by = xcd + 8u * (slot / gx);
        bx = slot % gx;
    }
}

// One pair: nbody3d.js:232-237 with b.w already multiplied by G at staging
// time ((G*m)*inv is the reference's left-associated product, :236).
template <typename T>
__device__ __forceinline__ void pair(const T bx, const T by, const T bz, const T bgm, const T xi, const T yi, const T zi,
                                     const T eps2, T& ax, T& ay, T& az)
{
    const T dx = bx - xi, dy = by - yi, dz = bz - zi;                  // :233
    const T d2 = nb_fma(dz, dz, nb_fma(dy, dy, nb_fma(dx, dx, eps2)));  // :234 (contracted; WGSL permits it)
    const T d6 = d2 * d2 * d2;                                         // :235
    const T s = bgm * nb_rsqrt(d6);                                    // :235-236
    ax = nb_fma(s, dx, ax);                                            // :266
    ay = nb_fma(s, dy, ay);
    az = nb_fma(s, dz, az);
}

// K1.  partial[by * i_count + il] = sum over this block's j-range.
//   IPL: i-bodies per lane group; LS: lanes sharing one i-body (power of two, <= 64).
template <typename T, int IPL, int LS>
__global__ __launch_bounds__(kBlock) void nb_force(const typename vec4<T>::type* __restrict__ bodies,
                                                  typename vec4<T>::type* __restrict__ partial, uint32_t n,
                                                  uint32_t i_begin, uint32_t i_count, T G, T eps2,
                                                  uint32_t j_per_split, SplitWindow win)
{
    using V4 = typename vec4<T>::type;
    static_assert(LS >= 1 && LS <= 64 && (LS & (LS - 1)) == 0, "LS must be a power of two <= 64");
    uint32_t bxi, byi;
    xcd_remap(bxi, byi, win.xcd_remap);
    const uint32_t by = win.split(byi);
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
        V4 b = V4{0, 0, 0, 0};
        if (il < i_count) b = bodies[i_begin + il];
        xi[k] = b.x; yi[k] = b.y; zi[k] = b.z;
        ax[k] = 0; ay[k] = 0; az[k] = 0;
    }

    const uint32_t j0 = by * j_per_split;
    uint32_t j1 = j0 + j_per_split;
    if (j1 > n) j1 = n;
    const uint32_t ntiles = (j1 > j0) ? (j1 - j0 + kTile - 1) / kTile : 0;

    auto stage = [&](uint32_t t) -> V4 {
        const uint32_t j = j0 + t * kTile + tid;
        V4 b = V4{0, 0, 0, 0};              // past the range: zero mass, contributes exactly 0
        if (j < j1) { b = bodies[j]; b.w *= G; }
        return b;
    };

    if (ntiles) tile[0][tid] = stage(0);
    __syncthreads();

    for (uint32_t t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        V4 nxt;
        const bool more = (t + 1 < ntiles);
        if (more) nxt = stage(t + 1);        // global load in flight under the tile's compute
        // j-bodies of this tile that are inside the split (the last tile of a split is
        // usually partial: splits are not tile multiples, see choose_shape); the loop runs
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
                for (int k = 0; k < IPL; ++k) pair<T>(b.x, b.y, b.z, b.w, xi[k], yi[k], zi[k], eps2, ax[k], ay[k], az[k]);
            }
        }
        if (more) tile[cur ^ 1][tid] = nxt;
        __syncthreads();
    }

    // wavefront-shuffle reduction of the LS partial sums that share a body
    if constexpr (LS > 1) {
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
#pragma unroll
            for (int m = LS / 2; m >= 1; m >>= 1) {
                ax[k] += __shfl_xor(ax[k], m, 64);
                ay[k] += __shfl_xor(ay[k], m, 64);
                az[k] += __shfl_xor(az[k], m, 64);
            }
        }
    }
    if (js == 0) {
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            const uint32_t il = bxi * IPB + k * GROUPS + grp;
            if (il < i_count) partial[(size_t)by * i_count + il] = V4{ax[k], ay[k], az[k], 0};
        }
    }
}

// K1, packed form (f32 only).  Same algorithm as nb_force<float,...>, but the
// arithmetic is vectorised ACROSS TWO i-BODIES of the lane with the CDNA packed
// f32 instructions (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32: two f32 lanes
// per VGPR pair).  Measured on MI355X (profiles/r01/ubench_run1.txt): a wave
// issues one VALU op per 4 cycles whether it is packed or not, so the packed
// body (12 v_pk + 2 v_rsq per TWO pairs instead of 24 + 2) sustains ~25 % more
// pairs/s than the scalar body at the same occupancy.  The j-body needs no
// shuffles: the ds_read_b128 result quad (x,y | z,m) feeds the packed ops
// through op_sel (lo/hi broadcast), which the backend folds from the splats.
//   NG = packed groups per lane -> IPL = 2*NG i-bodies per lane.
typedef float nb_f2 __attribute__((ext_vector_type(2)));

// Occupancy target handed to the register allocator/scheduler: NG = 4 needs
// ~118 VGPRs (4 waves/SIMD); telling the backend so keeps it from re-serialising
// the stage-major order to chase an occupancy it cannot reach anyway.
template <int NG, int LS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(NG >= 4 ? 4 : (NG == 2 ? 6 : 8), NG >= 4 ? 4 : (NG == 2 ? 6 : 8))))
void nb_force_pk(const float4* __restrict__ bodies, float4* __restrict__ partial,
                                                     uint32_t n, uint32_t i_begin, uint32_t i_count, float G,
                                                     float eps2, uint32_t j_per_split, SplitWindow win)
{
    static_assert(LS >= 1 && LS <= 64 && (LS & (LS - 1)) == 0, "LS must be a power of two <= 64");
    uint32_t bxi, byi;
    xcd_remap(bxi, byi, win.xcd_remap);
    const uint32_t by = win.split(byi);
    constexpr int IPL = 2 * NG;
    constexpr int GROUPS = kBlock / LS;
    constexpr int IPB = GROUPS * IPL;
    __shared__ float4 tile[2][kTile];

    const int tid = threadIdx.x;
    const int grp = tid / LS;
    const int js = tid % LS;

    nb_f2 xi[NG], yi[NG], zi[NG], ax[NG], ay[NG], az[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float4 b0 = float4{0, 0, 0, 0}, b1 = float4{0, 0, 0, 0};
        const uint32_t il0 = bxi * IPB + (2 * g) * GROUPS + grp;
        const uint32_t il1 = il0 + GROUPS;
        if (il0 < i_count) b0 = bodies[i_begin + il0];
        if (il1 < i_count) b1 = bodies[i_begin + il1];
        xi[g] = nb_f2{b0.x, b1.x}; yi[g] = nb_f2{b0.y, b1.y}; zi[g] = nb_f2{b0.z, b1.z};
        ax[g] = nb_f2{0, 0}; ay[g] = nb_f2{0, 0}; az[g] = nb_f2{0, 0};
    }
    const nb_f2 e2 = nb_f2{eps2, eps2};

    const uint32_t j0 = by * j_per_split;
    uint32_t j1 = j0 + j_per_split;
    if (j1 > n) j1 = n;
    const uint32_t ntiles = (j1 > j0) ? (j1 - j0 + kTile - 1) / kTile : 0;

    auto stage = [&](uint32_t t) -> float4 {
        const uint32_t j = j0 + t * kTile + tid;
        float4 b = float4{0, 0, 0, 0};
        if (j < j1) { b = bodies[j]; b.w *= G; }
        return b;
    };

    if (ntiles) tile[0][tid] = stage(0);
    __syncthreads();

    for (uint32_t t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        float4 nxt;
        const bool more = (t + 1 < ntiles);
        if (more) nxt = stage(t + 1);
        // JB j-bodies x NG groups = 4 independent dependency chains, issued
        // stage-major: consecutive packed ops never depend on each other, so the
        // backend needs no s_nop between a v_pk_* / v_rsq result and its consumer
        // (gfx950 VALU hazard) and one wave alone keeps the issue port busy.
        constexpr int JB = NG >= 4 ? 1 : 4 / NG;
        constexpr int NC = JB * NG;
        constexpr int UNR = 8 / JB;
        // exact trip count on a partial last tile (see nb_force), in chunks of 8 j-bodies
        const uint32_t left = j1 - (j0 + t * kTile);
        const int cnt = left < (uint32_t)kTile ? (int)left : kTile;
        const int chunks = ((cnt + LS - 1) / LS + 7) / 8;
        for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
        for (int uu = 0; uu < UNR; ++uu) {
            const int jj = ch * 8 + uu * JB;
            nb_f2 bx[JB], by[JB], bz[JB], bm[JB];
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const float4 b = tile[cur][(jj + u) * LS + js];
                bx[u] = nb_f2{b.x, b.x}; by[u] = nb_f2{b.y, b.y}; bz[u] = nb_f2{b.z, b.z}; bm[u] = nb_f2{b.w, b.w};
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
            for (int c = 0; c < NC; ++c) r[c] = nb_f2{__builtin_amdgcn_rsqf(r[c].x), __builtin_amdgcn_rsqf(r[c].y)};
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = bm[c / NG] * r[c];
            // accumulate in ascending j for every group (same order as the plain loop)
#pragma unroll
            for (int c = 0; c < NC; ++c) ax[c % NG] = __builtin_elementwise_fma(r[c], dx[c], ax[c % NG]);
#pragma unroll
            for (int c = 0; c < NC; ++c) ay[c % NG] = __builtin_elementwise_fma(r[c], dy[c], ay[c % NG]);
#pragma unroll
            for (int c = 0; c < NC; ++c) az[c % NG] = __builtin_elementwise_fma(r[c], dz[c], az[c % NG]);
        }
        }
        if (more) tile[cur ^ 1][tid] = nxt;
        __syncthreads();
    }

    if constexpr (LS > 1) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int m = LS / 2; m >= 1; m >>= 1) {
                ax[g].x += __shfl_xor(ax[g].x, m, 64); ax[g].y += __shfl_xor(ax[g].y, m, 64);
                ay[g].x += __shfl_xor(ay[g].x, m, 64); ay[g].y += __shfl_xor(ay[g].y, m, 64);
                az[g].x += __shfl_xor(az[g].x, m, 64); az[g].y += __shfl_xor(az[g].y, m, 64);
            }
        }
    }
    if (js == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t il0 = bxi * IPB + (2 * g) * GROUPS + grp;
            const uint32_t il1 = il0 + GROUPS;
            float4* out = partial + (size_t)by * i_count;
            if (il0 < i_count) out[il0] = float4{ax[g].x, ay[g].x, az[g].x, 0};
            if (il1 < i_count) out[il1] = float4{ax[g].y, ay[g].y, az[g].y, 0};
        }
    }
}

// K1, packed form with the j-bodies broadcast from SGPRs instead of LDS (SURVEY.md §8 f3
// "scalar-load (SGPR) j-broadcast A/B against the LDS tile").  j is wave-uniform, so
// bodies[j] is fetched with s_load_dwordx4 through the scalar cache and the packed ops
// take the (x,y | z,m) SGPR pairs directly (op_sel broadcast): no LDS, no barrier, no
// v_mov for the mass, workgroups need no tile synchronisation.  G is applied
// once to the finished sums (G * sum(m r^-3 d) instead of sum((G m) r^-3 d): rounding only).
// Opt-in (variant 34/38); the LDS kernel stays the default -- measurement in DESIGN.md.
template <int NG>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(NG >= 4 ? 4 : 6, NG >= 4 ? 4 : 6)))
void nb_force_pk_sgpr(const float4* __restrict__ bodies, float4* __restrict__ partial, uint32_t n, uint32_t i_begin,
                      uint32_t i_count, float G, float eps2, uint32_t j_per_split, SplitWindow win)
{
    constexpr int IPL = 2 * NG;
    constexpr int IPB = kBlock * IPL;
    uint32_t bxi, byi;
    xcd_remap(bxi, byi, win.xcd_remap);
    const uint32_t by = win.split(byi);
    const int tid = threadIdx.x;

    nb_f2 xi[NG], yi[NG], zi[NG], ax[NG], ay[NG], az[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float4 b0 = float4{0, 0, 0, 0}, b1 = float4{0, 0, 0, 0};
        const uint32_t il0 = bxi * IPB + (2 * g) * kBlock + tid;
        const uint32_t il1 = il0 + kBlock;
        if (il0 < i_count) b0 = bodies[i_begin + il0];
        if (il1 < i_count) b1 = bodies[i_begin + il1];
        xi[g] = nb_f2{b0.x, b1.x}; yi[g] = nb_f2{b0.y, b1.y}; zi[g] = nb_f2{b0.z, b1.z};
        ax[g] = nb_f2{0, 0}; ay[g] = nb_f2{0, 0}; az[g] = nb_f2{0, 0};
    }
    const nb_f2 e2 = nb_f2{eps2, eps2};
    const uint32_t j0 = by * j_per_split;
    uint32_t j1 = j0 + j_per_split;
    if (j1 > n) j1 = n;

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
#pragma unroll
            for (int c = 0; c < NG; ++c) dz[c] = bz - zi[c];
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
        if constexpr (NG == 4) asm volatile(NB_LOAD4(16) : "=s"(q.q0), "=s"(q.q1), "=s"(q.q2), "=s"(q.q3), NB_ACC4 : "s"(p) : "memory");
        else asm volatile(NB_LOAD4(10) : "=s"(q.q0), "=s"(q.q1), "=s"(q.q2), "=s"(q.q3), NB_ACC2 : "s"(p) : "memory");
    };
#undef NB_LOAD4
#undef NB_ACC2
#undef NB_ACC4
    const uint32_t nb8 = j1 > j0 ? (j1 - j0) / 8 : 0;
    const float4* pj = bodies + j0;
    uint32_t j = j0;
    if (nb8) {
        Quad A, B;
        request(A, pj);
        for (uint32_t it = 0; it < nb8; ++it) {
            wait_for(A);
            request(B, pj + 4);
            eval4(f4(A.q0), f4(A.q1), f4(A.q2), f4(A.q3));
            wait_for(B);
            pj += 8;
            if (it + 1 < nb8) request(A, pj);
            eval4(f4(B.q0), f4(B.q1), f4(B.q2), f4(B.q3));
        }
        j += nb8 * 8;
    }
    for (; j < j1; ++j)      // < 8 bodies left (only when n is not a multiple of 8): one at a time
        eval4(bodies[j], float4{0, 0, 0, 0}, float4{0, 0, 0, 0}, float4{0, 0, 0, 0});

    float4* out = partial + (size_t)by * i_count;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint32_t il0 = bxi * IPB + (2 * g) * kBlock + tid;
        const uint32_t il1 = il0 + kBlock;
        if (il0 < i_count) out[il0] = float4{G * ax[g].x, G * ay[g].x, G * az[g].x, 0};
        if (il1 < i_count) out[il1] = float4{G * ax[g].y, G * ay[g].y, G * az[g].y, 0};
    }
}

// K2.  nbody3d.js:274-290 on all four components (the .w lane is integrated
// too, exactly as the reference does; mass stays constant because vel.w = 0).
// R lanes cooperate on one body: lane r sums partials r, r+R, r+2R, ... (ascending,
// independent 16-B loads in flight), the R sums are combined by wavefront shuffles
// in a fixed order (deterministic), and lane 0 of the group applies the update.
// With jsplit = 64 partials a single lane per body is latency-bound (20 us at
// 16,384 rows); R = 8 brings it to the launch floor.
template <typename T, int R>
__global__ __launch_bounds__(kBlock) void nb_integrate(typename vec4<T>::type* __restrict__ bodies,
                                                      typename vec4<T>::type* __restrict__ vel,
                                                      typename vec4<T>::type* __restrict__ acc,
                                                      const typename vec4<T>::type* __restrict__ partial,
                                                      uint32_t i_begin, uint32_t i_count, uint32_t jsplit, T dt)
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
            const V4 p0 = partial[(size_t)sp * i_count + il];
            const V4 p1 = partial[(size_t)(sp + R) * i_count + il];
            const V4 p2 = partial[(size_t)(sp + 2 * R) * i_count + il];
            const V4 p3 = partial[(size_t)(sp + 3 * R) * i_count + il];
            sx += p0.x; sy += p0.y; sz += p0.z;
            sx += p1.x; sy += p1.y; sz += p1.z;
            sx += p2.x; sy += p2.y; sz += p2.z;
            sx += p3.x; sy += p3.y; sz += p3.z;
        }
        for (; sp < jsplit; sp += R) {
            const V4 p = partial[(size_t)sp * i_count + il];
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
    V4 a;
    a.x = sx; a.y = sy; a.z = sz;
    a.w = 0;                                                            // :274
    const T h = dt * T(0.5);                                            // :276
    const V4 ao = acc[il];
    const V4 v = vel[il];
    const V4 x = bodies[i_begin + il];
    V4 nv, nx;
    nv.x = nb_fma(ao.x + a.x, h, v.x);                                  // :280
    nv.y = nb_fma(ao.y + a.y, h, v.y);
    nv.z = nb_fma(ao.z + a.z, h, v.z);
    nv.w = nb_fma(ao.w + a.w, h, v.w);
    nx.x = nb_fma(nb_fma(h, a.x, nv.x), dt, x.x);                       // :283
    nx.y = nb_fma(nb_fma(h, a.y, nv.y), dt, x.y);
    nx.z = nb_fma(nb_fma(h, a.z, nv.z), dt, x.z);
    nx.w = nb_fma(nb_fma(h, a.w, nv.w), dt, x.w);
    vel[il] = nv;                                                       // :281
    bodies[i_begin + il] = nx;                                          // :283
    acc[il] = a;                                                        // :290
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
