// ubench6.hip -- feasibility of a SYMMETRIC force step (Newton's third law inside a wave).
//
// Every kernel of the engine evaluates each ordered pair (i, j) on its own: 12 v_pk + 2 v_rsq_f32 per two pairs
// = 64 issue cycles per 128 lane-pairs, the instruction-mix ceiling of 62.5 % of the fp32 vector rate.  Here a lane
// keeps 2*NG resident bodies while J "traveling" bodies per lane rotate through the 64 lanes of the wave
// (v_mov_b32_dpp wave_ror:1); r = x_t - x_i, r^2, the cube and the reciprocal square root are computed ONCE per
// unordered pair and both accelerations are accumulated (the traveler's sums travel with it):
//     per (traveler, packed group): 3 v_pk_add, 3 v_pk_fma, 2 v_pk_mul, 2 v_rsq_f32,
//                                   v_pk_mul (m_t s) + 3 v_pk_fma  -> resident side
//                                   v_pk_mul (m_i s) + 3 v_pk_fma  -> traveler side
//     = 16 packed + 2 transcendental = 80 issue cycles per FOUR lane-pairs-with-both-directions ... per 256 lane
//     interactions, plus 10 v_mov_b32_dpp per traveler and step (x, y, z, m and six packed sums).
// This file measures (1) that wave_ror:1 rotates the whole wave on gfx950 and in which direction, (2) the issue rate
// of that loop on a full chip, (3) its sums against an fp64 direct sum.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench6 ubench6.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float nb_f2 __attribute__((ext_vector_type(2)));
typedef float nb_v4f __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float rot1(float v)
{
    // wave_ror:1 (DPP_WF_RR1 = 0x13C): lane l receives the value of lane l-1 (mod 64) -- checked by part 1
    const int iv = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x13C, 0xF, 0xF, false));   // old = src: every lane is written, no init mov
}

__global__ void k_rot(int* out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x13C, 0xF, 0xF, false);
    out[64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x134, 0xF, 0xF, false);     // wave_rol:1
}

// One wave: resident bodies res[wave*64*2NG + ...], travelers trv[...]; `sweeps` full rotations (64 steps each).
// Resident sums -> out_r, traveler sums -> out_t (after the last sweep the travelers are back in their home lanes).
template <int NG, int J>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_sym(const float4* __restrict__ res, const float4* __restrict__ trv, float4* __restrict__ out_r, float4* __restrict__ out_t,
           float eps2, int sweeps)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    nb_f2 xi[NG], yi[NG], zi[NG], mi[NG], ax[NG], ay[NG], az[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float4 b0 = res[(size_t)wave * 128 * NG + (2 * g) * 64 + lane], b1 = res[(size_t)wave * 128 * NG + (2 * g + 1) * 64 + lane];
        xi[g] = nb_f2{b0.x, b1.x}; yi[g] = nb_f2{b0.y, b1.y}; zi[g] = nb_f2{b0.z, b1.z}; mi[g] = nb_f2{b0.w, b1.w};
        ax[g] = nb_f2{0, 0}; ay[g] = nb_f2{0, 0}; az[g] = nb_f2{0, 0};
    }
    float tx[J], ty[J], tz[J], tm[J];
    nb_f2 bx[J], by[J], bz[J];
#pragma unroll
    for (int u = 0; u < J; ++u) {
        const float4 t = trv[(size_t)wave * 64 * J + u * 64 + lane];
        tx[u] = t.x; ty[u] = t.y; tz[u] = t.z; tm[u] = t.w;
        bx[u] = nb_f2{0, 0}; by[u] = nb_f2{0, 0}; bz[u] = nb_f2{0, 0};
    }
    const nb_f2 e2 = nb_f2{eps2, eps2};
    for (int s = 0; s < sweeps * 64; ++s) {
#pragma unroll
        for (int u = 0; u < J; ++u) {
            const nb_f2 px = nb_f2{tx[u], tx[u]}, py = nb_f2{ty[u], ty[u]}, pz = nb_f2{tz[u], tz[u]}, pm = nb_f2{tm[u], tm[u]};
            nb_f2 dx[NG], dy[NG], dz[NG], d2[NG], r[NG], si[NG], st[NG];
#pragma unroll
            for (int c = 0; c < NG; ++c) dx[c] = px - xi[c];
#pragma unroll
            for (int c = 0; c < NG; ++c) dy[c] = py - yi[c];
#pragma unroll
            for (int c = 0; c < NG; ++c) dz[c] = pz - zi[c];
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
            for (int c = 0; c < NG; ++c) si[c] = pm * r[c];          // m_t * inv  -> resident side
#pragma unroll
            for (int c = 0; c < NG; ++c) st[c] = mi[c] * r[c];       // m_i * inv  -> traveler side
#pragma unroll
            for (int c = 0; c < NG; ++c) ax[c] = __builtin_elementwise_fma(si[c], dx[c], ax[c]);
#pragma unroll
            for (int c = 0; c < NG; ++c) ay[c] = __builtin_elementwise_fma(si[c], dy[c], ay[c]);
#pragma unroll
            for (int c = 0; c < NG; ++c) az[c] = __builtin_elementwise_fma(si[c], dz[c], az[c]);
#pragma unroll
            for (int c = 0; c < NG; ++c) bx[u] = __builtin_elementwise_fma(-st[c], dx[c], bx[u]);
#pragma unroll
            for (int c = 0; c < NG; ++c) by[u] = __builtin_elementwise_fma(-st[c], dy[c], by[u]);
#pragma unroll
            for (int c = 0; c < NG; ++c) bz[u] = __builtin_elementwise_fma(-st[c], dz[c], bz[u]);
        }
        // the travelers and their sums move on by one lane
#pragma unroll
        for (int u = 0; u < J; ++u) {
            tx[u] = rot1(tx[u]); ty[u] = rot1(ty[u]); tz[u] = rot1(tz[u]); tm[u] = rot1(tm[u]);
            bx[u] = nb_f2{rot1(bx[u].x), rot1(bx[u].y)}; by[u] = nb_f2{rot1(by[u].x), rot1(by[u].y)}; bz[u] = nb_f2{rot1(bz[u].x), rot1(bz[u].y)};
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        out_r[(size_t)wave * 128 * NG + (2 * g) * 64 + lane] = float4{ax[g].x, ay[g].x, az[g].x, 0};
        out_r[(size_t)wave * 128 * NG + (2 * g + 1) * 64 + lane] = float4{ax[g].y, ay[g].y, az[g].y, 0};
    }
#pragma unroll
    for (int u = 0; u < J; ++u)
        out_t[(size_t)wave * 64 * J + u * 64 + lane] = float4{bx[u].x + bx[u].y, by[u].x + by[u].y, bz[u].x + bz[u].y, 0};
}

// issue cost of one v_mov_b32_dpp by control: 8 independent registers rotated `iters` times, 4 waves per SIMD
template <int CTRL>
__global__ __launch_bounds__(256) void k_dpp(float* out, int iters)
{
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)(threadIdx.x + k);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int iv = __builtin_bit_cast(int, v[k]);
            v[k] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xF, 0xF, false));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CTRL>
void dpp_cost(const char* name, int n_cu)
{
    float* d; CK(hipMalloc(&d, (size_t)n_cu * 4 * 256 * 4));
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_dpp<CTRL>), dim3(n_cu * 4), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_dpp<CTRL>), dim3(n_cu * 4), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // 4 waves per SIMD, 8 DPP movs per iteration each
    printf("%-22s %.3f ms -> %.2f ns per wave-instruction (x 2.3 GHz = %.1f cycles)\n", name, ms, ms * 1e6 / (iters * 8.0 * 4), ms * 1e6 / (iters * 8.0 * 4) * 2.3);
    CK(hipFree(d));
}

// Variant: the travelers do not move.  Their positions sit in a wave-private LDS slab and lane l reads slot (l - step) & 63
// (4 ds_read_b32: the LDS pipe, not the VALU); their sums are accumulated IN LDS with ds_add_f32 (conflict-free: 64 lanes, 64
// slots; one wave's LDS operations execute in order, so step s+1 sees step s).  VALU work per step: the pair arithmetic only.
template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_sym_lds(const float4* __restrict__ res, const float4* __restrict__ trv, float4* __restrict__ out_r, float4* __restrict__ out_t,
               float eps2, int sweeps)
{
    __shared__ float tpos[4][4][64];
    __shared__ float tsum[4][3][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    nb_f2 xi[NG], yi[NG], zi[NG], mi[NG], ax[NG], ay[NG], az[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float4 b0 = res[(size_t)wave * 128 * NG + (2 * g) * 64 + lane], b1 = res[(size_t)wave * 128 * NG + (2 * g + 1) * 64 + lane];
        xi[g] = nb_f2{b0.x, b1.x}; yi[g] = nb_f2{b0.y, b1.y}; zi[g] = nb_f2{b0.z, b1.z}; mi[g] = nb_f2{b0.w, b1.w};
        ax[g] = nb_f2{0, 0}; ay[g] = nb_f2{0, 0}; az[g] = nb_f2{0, 0};
    }
    const float4 t = trv[(size_t)wave * 64 + lane];
    tpos[wv][0][lane] = t.x; tpos[wv][1][lane] = t.y; tpos[wv][2][lane] = t.z; tpos[wv][3][lane] = t.w;
    tsum[wv][0][lane] = 0; tsum[wv][1][lane] = 0; tsum[wv][2][lane] = 0;
    const nb_f2 e2 = nb_f2{eps2, eps2};
    for (int s = 0; s < sweeps * 64; ++s) {
        const int slot = (lane - s) & 63;
        const float tx = tpos[wv][0][slot], ty = tpos[wv][1][slot], tz = tpos[wv][2][slot], tm = tpos[wv][3][slot];
        const nb_f2 px = nb_f2{tx, tx}, py = nb_f2{ty, ty}, pz = nb_f2{tz, tz}, pm = nb_f2{tm, tm};
        nb_f2 bx, by, bz;
#pragma unroll
        for (int c0g = 0; c0g < NG; c0g += 4) {
            nb_f2 dx[4], dy[4], dz[4], d2[4], r[4], si[4], st[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) dx[c] = px - xi[c0g + c];
#pragma unroll
            for (int c = 0; c < 4; ++c) dy[c] = py - yi[c0g + c];
#pragma unroll
            for (int c = 0; c < 4; ++c) dz[c] = pz - zi[c0g + c];
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
            for (int c = 0; c < 4; ++c) si[c] = pm * r[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) st[c] = mi[c0g + c] * r[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) ax[c0g + c] = __builtin_elementwise_fma(si[c], dx[c], ax[c0g + c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) ay[c0g + c] = __builtin_elementwise_fma(si[c], dy[c], ay[c0g + c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) az[c0g + c] = __builtin_elementwise_fma(si[c], dz[c], az[c0g + c]);
            // traveler side: the first product starts the chain (no zeroing), the rest accumulate
            if (c0g == 0) { bx = -st[0] * dx[0]; by = -st[0] * dy[0]; bz = -st[0] * dz[0]; }
#pragma unroll
            for (int c = (c0g == 0 ? 1 : 0); c < 4; ++c) bx = __builtin_elementwise_fma(-st[c], dx[c], bx);
#pragma unroll
            for (int c = (c0g == 0 ? 1 : 0); c < 4; ++c) by = __builtin_elementwise_fma(-st[c], dy[c], by);
#pragma unroll
            for (int c = (c0g == 0 ? 1 : 0); c < 4; ++c) bz = __builtin_elementwise_fma(-st[c], dz[c], bz);
        }
        // both halves of the packed sums go to the traveler's slot (ds_add_f32, no return value)
        __hip_atomic_fetch_add(&tsum[wv][0][slot], bx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&tsum[wv][0][slot], bx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&tsum[wv][1][slot], by.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&tsum[wv][1][slot], by.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&tsum[wv][2][slot], bz.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&tsum[wv][2][slot], bz.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        out_r[(size_t)wave * 128 * NG + (2 * g) * 64 + lane] = float4{ax[g].x, ay[g].x, az[g].x, 0};
        out_r[(size_t)wave * 128 * NG + (2 * g + 1) * 64 + lane] = float4{ax[g].y, ay[g].y, az[g].y, 0};
    }
    out_t[(size_t)wave * 64 + lane] = float4{tsum[wv][0][lane], tsum[wv][1][lane], tsum[wv][2][lane], 0};
}

template <int NG, int J, bool LDS = false>
void run(const char* name, int n_cu)
{
    auto kern = [](auto... a) {};
    (void)kern;
    const int waves = n_cu * 16, wgs = waves / 4;                 // 4 waves per SIMD
    const size_t nr = (size_t)waves * 128 * NG, nt = (size_t)waves * 64 * J;
    std::vector<float4> hr(nr), ht(nt);
    srand(1);
    auto rnd = [] { return (float)(rand() & 0xffffff) / (float)0x1000000 * 2.f - 1.f; };
    for (auto& b : hr) b = float4{rnd(), rnd(), rnd(), 0.5f + 0.5f * fabsf(rnd())};
    for (auto& b : ht) b = float4{rnd(), rnd(), rnd(), 0.5f + 0.5f * fabsf(rnd())};
    float4 *dr, *dt, *orr, *ott;
    CK(hipMalloc(&dr, nr * 16)); CK(hipMalloc(&dt, nt * 16)); CK(hipMalloc(&orr, nr * 16)); CK(hipMalloc(&ott, nt * 16));
    CK(hipMemcpy(dr, hr.data(), nr * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, ht.data(), nt * 16, hipMemcpyHostToDevice));
    // correctness: one sweep, wave 0 and the last wave against an fp64 direct sum
    if constexpr (LDS) hipLaunchKernelGGL((k_sym_lds<NG>), dim3(wgs), dim3(256), 0, 0, dr, dt, orr, ott, 1e-4f, 1); else hipLaunchKernelGGL((k_sym<NG, J>), dim3(wgs), dim3(256), 0, 0, dr, dt, orr, ott, 1e-4f, 1);
    CK(hipDeviceSynchronize());
    std::vector<float4> gr(nr), gt(nt);
    CK(hipMemcpy(gr.data(), orr, nr * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(gt.data(), ott, nt * 16, hipMemcpyDeviceToHost));
    double worst_r = 0, worst_t = 0;
    for (int w : {0, waves - 1}) {
        const float4* R = hr.data() + (size_t)w * 128 * NG; const float4* T = ht.data() + (size_t)w * 64 * J;
        for (int i = 0; i < 128 * NG; ++i) {
            double a[3] = {0, 0, 0};
            for (int j = 0; j < 64 * J; ++j) {
                const double dx = (double)T[j].x - R[i].x, dy = (double)T[j].y - R[i].y, dz = (double)T[j].z - R[i].z;
                const double d2 = dx * dx + dy * dy + dz * dz + 1e-4, s = T[j].w / (d2 * sqrt(d2));
                a[0] += s * dx; a[1] += s * dy; a[2] += s * dz;
            }
            const float4 g = gr[(size_t)w * 128 * NG + i];
            const double sc = fmax(fmax(fabs(a[0]), fabs(a[1])), fabs(a[2]));
            worst_r = fmax(worst_r, fmax(fmax(fabs(g.x - a[0]), fabs(g.y - a[1])), fabs(g.z - a[2])) / sc);
        }
        for (int j = 0; j < 64 * J; ++j) {
            double a[3] = {0, 0, 0};
            for (int i = 0; i < 128 * NG; ++i) {
                const double dx = (double)R[i].x - T[j].x, dy = (double)R[i].y - T[j].y, dz = (double)R[i].z - T[j].z;
                const double d2 = dx * dx + dy * dy + dz * dz + 1e-4, s = R[i].w / (d2 * sqrt(d2));
                a[0] += s * dx; a[1] += s * dy; a[2] += s * dz;
            }
            const float4 g = gt[(size_t)w * 64 * J + j];
            const double sc = fmax(fmax(fabs(a[0]), fabs(a[1])), fabs(a[2]));
            worst_t = fmax(worst_t, fmax(fmax(fabs(g.x - a[0]), fabs(g.y - a[1])), fabs(g.z - a[2])) / sc);
        }
    }
    // timing
    const int sweeps = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 3; ++k) { if constexpr (LDS) hipLaunchKernelGGL((k_sym_lds<NG>), dim3(wgs), dim3(256), 0, 0, dr, dt, orr, ott, 1e-4f, sweeps); else hipLaunchKernelGGL((k_sym<NG, J>), dim3(wgs), dim3(256), 0, 0, dr, dt, orr, ott, 1e-4f, sweeps); }
    CK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int k = 0; k < reps; ++k) { if constexpr (LDS) hipLaunchKernelGGL((k_sym_lds<NG>), dim3(wgs), dim3(256), 0, 0, dr, dt, orr, ott, 1e-4f, sweeps); else hipLaunchKernelGGL((k_sym<NG, J>), dim3(wgs), dim3(256), 0, 0, dr, dt, orr, ott, 1e-4f, sweeps); }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double inter = 2.0 * waves * (double)sweeps * 64 * 64 * (2.0 * NG * J);     // ordered interactions (both directions)
    const double rate = inter / (ms * 1e-3);
    printf("%-12s waves=%d  %.3f ms  %.3e interactions/s = %5.1f %% of 7.865e12   err resident %.2e traveler %.2e\n", name, waves, ms, rate,
           100 * rate / 7.865e12, worst_r, worst_t);
    CK(hipFree(dr)); CK(hipFree(dt)); CK(hipFree(orr)); CK(hipFree(ott));
}

int main()
{
    int* d; CK(hipMalloc(&d, 128 * 4));
    hipLaunchKernelGGL(k_rot, dim3(1), dim3(64), 0, 0, d);
    int h[128]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("wave_ror:1 lanes 0..3,15..17,31..33,63: %d %d %d %d | %d %d %d | %d %d %d | %d\n", h[0], h[1], h[2], h[3], h[15], h[16], h[17], h[31], h[32], h[33], h[63]);
    printf("wave_rol:1 lanes 0..3,15..17,31..33,63: %d %d %d %d | %d %d %d | %d %d %d | %d\n", h[64], h[65], h[66], h[67], h[79], h[80], h[81], h[95], h[96], h[97], h[127]);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cu = p.multiProcessorCount;
    dpp_cost<0x13C>("wave_ror:1", cu);
    dpp_cost<0x121>("row_ror:1", cu);
    dpp_cost<0xB1>("quad_perm[1,0,3,2]", cu);
    dpp_cost<0x138>("wave_shr:1", cu);
    dpp_cost<0x142>("row_bcast:15", cu);
    run<4, 1>("NG4 J1", cu);
    run<4, 2>("NG4 J2", cu);
    run<2, 2>("NG2 J2", cu);
    run<2, 4>("NG2 J4", cu);
    run<8, 1>("NG8 J1 dpp", cu);
    run<8, 1, true>("NG8 J1 lds", cu);
    run<4, 1, true>("NG4 J1 lds", cu);
    return 0;
}
