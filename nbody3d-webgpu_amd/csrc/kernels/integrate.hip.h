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
