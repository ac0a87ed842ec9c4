// kernels/integrate.hip.h -- K2 (nb_integrate*), the viewer frame pack and the on-device diagnostics.
// Part of nb_kernels.hip.h (include that, not this file).
#pragma once

namespace nb {

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

// Diagnostics (no reference analogue; north_star "total-energy drift reported"): per-workgroup fp64
// partial sums of kinetic energy, momentum and the handle's share of the softened potential; finished on the host in a
// fixed order (deterministic).
//   The potential is a scalar, so Newton's third law costs nothing here: workgroup (bi, c) holds kDiagRows i-rows (4 per lane)
// and sweeps the 256-body tiles of j-chunk c, counting a pair only when j > i -- every UNORDERED pair of the system once
// (a shard handle: the pairs whose LOWER index lies in its rows, so the shards' shares add up to the whole).  Tiles entirely
// at or below the block's first row are skipped, tiles entirely above its last row take the unmasked loop, the few that
// cross the diagonal the masked one.  f32 handles: r^2 and v_rsq_f32 in packed f32 per pair (relative error ~1e-7 per pair,
// averaging out over the sum), every m_j / r accumulated in fp64 per lane -- 12 v_pk + 2 v_rsq_f32 + 2 v_cvt + 2 v_fma_f64 per
// two i-rows and j; f64 handles: all fp64, v_rsq_f64 + one correction step.  N = 262,144: ~8 ms (the ordered-pair fp64 loop
// it replaces: 52 ms; a force step: 10.5 ms).
constexpr uint32_t kDiagRows = 4 * kBlock;

template <typename T>
__global__ __launch_bounds__(kBlock) void nb_diag(const typename vec4<T>::type* __restrict__ bodies,
                                                 const typename vec4<T>::type* __restrict__ vel, uint32_t n,
                                                 uint32_t i_begin, uint32_t i_count, uint32_t j_chunk, double G, T eps2,
                                                 double* __restrict__ out /* [gridDim.y][gridDim.x][5] */)
{
    using V4 = typename vec4<T>::type;
    __shared__ V4 tile[kTile];
    __shared__ double red[5][kBlock / 64];
    const int tid = threadIdx.x;
    const uint32_t ib0 = blockIdx.x * kDiagRows;                  // first row of the block, shard-local
    const uint32_t gi0 = i_begin + ib0;                           // ... and as an index into `bodies`
    const uint32_t rows = i_count - ib0 < kDiagRows ? i_count - ib0 : kDiagRows;
    const uint32_t gi_end = gi0 + rows;
    const uint32_t jc0 = blockIdx.y * j_chunk, jc1 = (jc0 + j_chunk < n) ? jc0 + j_chunk : n;

    T xi[4], yi[4], zi[4];
    double mi[4], acc[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t gi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t il = (uint32_t)tid + k * kBlock;
        gi[k] = gi0 + il;
        const bool valid = il < rows;
        const V4 b = valid ? ld4(bodies + gi[k]) : V4{0, 0, 0, 0};
        xi[k] = b.x; yi[k] = b.y; zi[k] = b.z; mi[k] = valid ? (double)b.w : 0.0;     // rows past the block: zero mass
    }

    for (uint32_t j0 = jc0; j0 < jc1; j0 += kTile) {
        if (j0 + kTile <= gi0 + 1) continue;                      // every j of the tile <= every i of the block: counted elsewhere
        const uint32_t j = j0 + tid;
        __syncthreads();                                          // the previous tile has been read
        tile[tid] = (j < jc1) ? ld4(bodies + j) : V4{0, 0, 0, 0};  // past the chunk / the system: zero mass
        __syncthreads();
        const bool masked = j0 < gi_end;                          // some j <= some i: per-pair test
        if constexpr (sizeof(T) == 4) {
            const nb_f2 e2 = nb_f2{eps2, eps2};
            const nb_f2 xa = nb_f2{xi[0], xi[1]}, ya = nb_f2{yi[0], yi[1]}, za = nb_f2{zi[0], zi[1]};
            const nb_f2 xb = nb_f2{xi[2], xi[3]}, yb = nb_f2{yi[2], yi[3]}, zb = nb_f2{zi[2], zi[3]};
            if (!masked) {
#pragma unroll 4
                for (int jj = 0; jj < kTile; ++jj) {
                    const float4 b = tile[jj];
                    const nb_f2 px = nb_f2{b.x, b.x}, py = nb_f2{b.y, b.y}, pz = nb_f2{b.z, b.z};
                    const nb_f2 dxa = px - xa, dya = py - ya, dza = pz - za, dxb = px - xb, dyb = py - yb, dzb = pz - zb;
                    const nb_f2 ra = __builtin_elementwise_fma(dza, dza, __builtin_elementwise_fma(dya, dya, __builtin_elementwise_fma(dxa, dxa, e2)));
                    const nb_f2 rb = __builtin_elementwise_fma(dzb, dzb, __builtin_elementwise_fma(dyb, dyb, __builtin_elementwise_fma(dxb, dxb, e2)));
                    const double mj = (double)b.w;
                    acc[0] = __builtin_fma((double)__builtin_amdgcn_rsqf(ra.x), mj, acc[0]);
                    acc[1] = __builtin_fma((double)__builtin_amdgcn_rsqf(ra.y), mj, acc[1]);
                    acc[2] = __builtin_fma((double)__builtin_amdgcn_rsqf(rb.x), mj, acc[2]);
                    acc[3] = __builtin_fma((double)__builtin_amdgcn_rsqf(rb.y), mj, acc[3]);
                }
            } else {
                for (int jj = 0; jj < kTile; ++jj) {
                    const float4 b = tile[jj];
                    const uint32_t jg = j0 + (uint32_t)jj;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dx = b.x - xi[k], dy = b.y - yi[k], dz = b.z - zi[k];
                        const float r2 = nb_fma(dz, dz, nb_fma(dy, dy, nb_fma(dx, dx, eps2)));
                        const double w = jg > gi[k] ? (double)b.w : 0.0;
                        acc[k] = __builtin_fma((double)__builtin_amdgcn_rsqf(r2), w, acc[k]);
                    }
                }
            }
        } else {
            for (int jj = 0; jj < kTile; ++jj) {
                const double4 b = tile[jj];
                const uint32_t jg = j0 + (uint32_t)jj;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double dx = b.x - xi[k], dy = b.y - yi[k], dz = b.z - zi[k];
                    const double r2 = nb_fma(dz, dz, nb_fma(dy, dy, nb_fma(dx, dx, eps2)));
                    const double y0 = __builtin_amdgcn_rsq(r2);               // ~2^-26 relative; one correction: y0 (1 + e/2), e = 1 - r2 y0^2
                    const double e = nb_fma(-(r2 * y0), y0, 1.0);
                    const double y = nb_fma(0.5 * y0, e, y0);
                    const double w = (!masked || jg > gi[k]) ? b.w : 0.0;
                    acc[k] = nb_fma(y, w, acc[k]);
                }
            }
        }
    }

    double vals[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    vals[1] = -G * (mi[0] * acc[0] + mi[1] * acc[1] + mi[2] * acc[2] + mi[3] * acc[3]);
    if (blockIdx.y == 0) {                                         // kinetic energy and momentum of the block's rows: once
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t il = ib0 + (uint32_t)tid + k * kBlock;
            if (il >= i_count) continue;
            const V4 v = ld4(vel + il);
            const double vx = (double)v.x, vy = (double)v.y, vz = (double)v.z;
            vals[0] += 0.5 * mi[k] * (vx * vx + vy * vy + vz * vz);
            vals[2] += mi[k] * vx; vals[3] += mi[k] * vy; vals[4] += mi[k] * vz;
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double v = vals[q];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
        if ((tid & 63) == 0) red[q][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 5) {
        double v = 0;
        for (int w = 0; w < kBlock / 64; ++w) v += red[tid][w];
        out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5 + tid] = v;
    }
}

}  // namespace nb
