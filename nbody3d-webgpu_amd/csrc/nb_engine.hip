// nb_engine.hip -- C ABI (include/nbody3d_hip.h) over the HIP particle pool
// and the gfx950 kernels in nb_kernels.hip.h: the single-handle entry points.
// (nb_comm.hip holds the RCCL collective and the single-process multi-device handle.)
//
// Device state per handle (reference layout float4 AoS,
// nbody3d.js:179-199, kept as-is on the device because one 16-B lane access is
// the widest coalesced load and the j-tile is read back as one ds_read_b128):
//   bodies  : 4*n elements, replicated on every shard (x, y, z, mass); a fused handle keeps two
//             (ping-pong: a step reads one and writes the other)
//   vel     : 4*shard_count elements
//   accel   : 4*shard_count elements (acceleration of the previous step)
//   partial : jsplit * 4*shard_count elements (K1 output, summed by K2; none when fused)
#include "nb_internal.h"
#include "nb_kernels.hip.h"
#include "../../include/nbody3d_hip_plan.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

namespace {
thread_local std::string g_create_error = "";
}

namespace nbi {
int fail(nb_sim* s, int code, const std::string& msg)
{
    if (s) s->err = msg; else g_create_error = msg;
    return code;
}
void set_create_error(const std::string& msg) { g_create_error = msg; }
const std::string& create_error() { return g_create_error; }
}  // namespace nbi

namespace {

using nbi::fail;

#define NB_HIP(s, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail((s), NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));              \
    } while (0)

using namespace nbp;     // Kind, Shape, ceil_div, ipb_of, ... (nb_plan.h)

// ---- kernel tables ---------------------------------------------------------------------------
// packed LDS kernels: NG in {1,2,4}, LS in {1,2,4,8,16,32,64}, TL = 1; TL = 4 for LS >= 16
// (a tile of 256 bodies is only 256/LS loop iterations: short loops want 1024 staged at once)
#define NB_LS_CASES(F, NG, TL, ls)                                      \
    switch (ls) {                                                        \
        case 1: return (const void*)&nb::F<NG, 1, TL>;                   \
        case 2: return (const void*)&nb::F<NG, 2, TL>;                   \
        case 4: return (const void*)&nb::F<NG, 4, TL>;                   \
        case 8: return (const void*)&nb::F<NG, 8, TL>;                   \
        case 16: return (const void*)&nb::F<NG, 16, TL>;                 \
        case 32: return (const void*)&nb::F<NG, 32, TL>;                 \
        case 64: return (const void*)&nb::F<NG, 64, TL>;                 \
        default: return nullptr;                                         \
    }
#define NB_LS_CASES_T4(F, NG, ls)                                       \
    switch (ls) {                                                        \
        case 16: return (const void*)&nb::F<NG, 16, 4>;                  \
        case 32: return (const void*)&nb::F<NG, 32, 4>;                  \
        case 64: return (const void*)&nb::F<NG, 64, 4>;                  \
        default: return nullptr;                                         \
    }
#define NB_LS_CASES_T8(F, NG, ls)                                       \
    switch (ls) {                                                        \
        case 32: return (const void*)&nb::F<NG, 32, 8>;                  \
        case 64: return (const void*)&nb::F<NG, 64, 8>;                  \
        default: return nullptr;                                         \
    }
#define NB_PK_TABLE(NAME, F)                                            \
    const void* NAME(int ng, int ls, int tl)                            \
    {                                                                    \
        if (tl == 8) {          /* 2048-body stages: 64 KiB of LDS, <= 2 workgroups per CU */ \
            if (ng == 1) { NB_LS_CASES_T8(F, 1, ls) }                    \
            if (ng == 2) { NB_LS_CASES_T8(F, 2, ls) }                    \
            if (ng == 4) { NB_LS_CASES_T8(F, 4, ls) }                    \
        }                                                                \
        if (tl == 1) {                                                   \
            if (ng == 1) { NB_LS_CASES(F, 1, 1, ls) }                    \
            if (ng == 2) { NB_LS_CASES(F, 2, 1, ls) }                    \
            if (ng == 4) { NB_LS_CASES(F, 4, 1, ls) }                    \
        } else if (tl == 4) {                                            \
            if (ng == 1) { NB_LS_CASES_T4(F, 1, ls) }                    \
            if (ng == 2) { NB_LS_CASES_T4(F, 2, ls) }                    \
            if (ng == 4) { NB_LS_CASES_T4(F, 4, ls) }                    \
        }                                                                \
        return nullptr;                                                  \
    }
NB_PK_TABLE(pk_force_kernel, nb_force_pk)
NB_PK_TABLE(pk_fused_kernel, nb_step_fused)

template <typename T>
const void* scalar_kernel(int ipl, int ls)
{
    if (ls == 1) {
        if (ipl == 1) return (const void*)&nb::nb_force<T, 1, 1>;
        if (ipl == 2) return (const void*)&nb::nb_force<T, 2, 1>;
        if (ipl == 4) return (const void*)&nb::nb_force<T, 4, 1>;
        return nullptr;
    }
    if (ipl != 1) return nullptr;
    if (ls == 4) return (const void*)&nb::nb_force<T, 1, 4>;
    if (ls == 16) return (const void*)&nb::nb_force<T, 1, 16>;
    if (ls == 64) return (const void*)&nb::nb_force<T, 1, 64>;
    return nullptr;
}

const void* sgpr_kernel(int ipl, int ws)
{
    if (ipl == 4 && ws == 5) return (const void*)&nb::nb_force_pk_sgpr<2, 4, true>;     // 64-bit pair loads
    if (ipl == 4) return ws == 4 ? (const void*)&nb::nb_force_pk_sgpr<2, 4> : (const void*)&nb::nb_force_pk_sgpr<2, 1>;
    if (ipl == 8 && ws == 5) return (const void*)&nb::nb_force_pk_sgpr<4, 4, true>;     // A/B arm: 64-bit pair loads
    if (ipl == 8) return ws == 4 ? (const void*)&nb::nb_force_pk_sgpr<4, 4> : (const void*)&nb::nb_force_pk_sgpr<4, 1>;
    return nullptr;
}

// The rank form's force kernel (two phases: nb::SymRankPlan); ipl = residents per lane.
const void* rank_kernel_of(bool f64, int ipl)
{
    if (f64) return ipl == 8 ? (const void*)&nb::nb_force_symw64_rank<8> : nullptr;
    return ipl == 16 ? (const void*)&nb::nb_force_symw_rank<8> : ipl == 8 ? (const void*)&nb::nb_force_symw_rank<4> : nullptr;
}

// The kernel a shape launches (nullptr: no such instantiation).
const void* kernel_of(bool f64, const Shape& sh)
{
    switch (sh.kind) {
        case kScalar: return f64 ? scalar_kernel<double>(sh.ipl, sh.ls) : scalar_kernel<float>(sh.ipl, sh.ls);
        case kPkLds: return f64 || (sh.ipl & 1) ? nullptr : pk_force_kernel(sh.ipl / 2, sh.ls, sh.x);
        case kFused: return f64 || (sh.ipl & 1) ? nullptr : pk_fused_kernel(sh.ipl / 2, sh.ls, sh.x);
        case kPkSgpr: return f64 || sh.ls != 1 || !(sh.x == 1 || sh.x == 4 || sh.x == 5) ? nullptr : sgpr_kernel(sh.ipl, sh.x);
        case kDirect:
            if (f64 || sh.ipl != 2 || sh.ls != 64) return nullptr;
            return sh.x == 1 ? (const void*)&nb::nb_step_direct<16> : sh.x == 2 ? (const void*)&nb::nb_step_direct<32> : nullptr;
        case kSym:       // ipl = residents per lane (8 or 16); x = 1 / 3: wave-granular form with 2 / 1 travelers per lane,
                         // x = 4: workgroup form (4 waves, 8 residents per lane)
            if (sh.ls != 1) return nullptr;
            if (f64) return sh.ipl == 8 && sh.x == 3 ? (const void*)&nb::nb_force_symw64<8> : nullptr;      // 8 residents, 1 traveler per lane
            if (sh.x == 4) return sh.ipl == 8 ? (const void*)&nb::nb_force_sym<4, 4, 2> : nullptr;
            if (sh.ipl == 4) return sh.x == 3 ? (const void*)&nb::nb_force_symw<2, 1> : nullptr;      // (an arm: whole sweeps round finer with 4 residents)
            if (sh.ipl == 8) return sh.x == 1 ? (const void*)&nb::nb_force_symw<4, 2> : sh.x == 3 ? (const void*)&nb::nb_force_symw<4, 1> : nullptr;
            if (sh.ipl == 16) return sh.x == 1 ? (const void*)&nb::nb_force_symw<8, 2> : sh.x == 3 ? (const void*)&nb::nb_force_symw<8, 1> : nullptr;
            return nullptr;
        case kJpk:
            if (f64 || sh.ipl != 1 || sh.ls != 1) return nullptr;
            return sh.x == 4 ? (const void*)&nb::nb_step_jpk<4> : sh.x == 8 ? (const void*)&nb::nb_step_jpk<8>
                   : sh.x == 6 ? (const void*)&nb::nb_step_jpk<16> : nullptr;
        default: return nullptr;
    }
}

// Runs the planner (nb_plan.cpp) for a handle and copies its answer into the handle's launch fields.
void plan_handle(nb_sim* s, const nb_config& cfg, int n_cu, double clock_hz, double device_mem)
{
    PlanInput in;
    if (device_mem > 0) in.device_mem = device_mem;
    in.n = s->n; in.sb = s->sb; in.sc = s->sc; in.f64 = s->f64; in.cfg = cfg; in.n_cu = n_cu; in.clock_hz = clock_hz;
    if (!s->no_device)
        in.occupancy = [f64 = s->f64](const Shape& sh, int block) {
            int occ = 0;
            const void* fn = kernel_of(f64, sh);
            if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, block, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
            return occ;
        };
    LaunchPlan p = plan_launch(in);
    s->ipl = p.ipl; s->ls = p.ls; s->ws = p.ws; s->tl = p.tl;
    s->packed = p.packed; s->sgpr = p.sgpr; s->fused = p.fused; s->direct = p.direct; s->jpk = p.jpk; s->swap_acc = p.swap_acc;
    s->jsplit = p.jsplit; s->j_per_split = p.j_per_split; s->junits = p.junits; s->own_split0 = p.own_split0; s->own_splits = p.own_splits;
    s->sym = p.sym; s->symw = p.symw; s->sym_rank = p.sym_rank;
    s->sym_np = p.sym_np; s->sym_layers = p.sym_layers; s->sym_g0 = p.sym_g0; s->sym_g1 = p.sym_g1;
    static_assert(sizeof(s->sym_plan) == sizeof(p.sym_plan), "nb_sim::sym_plan mirrors LaunchPlan::sym_plan");
    memcpy(s->sym_plan, p.sym_plan, sizeof s->sym_plan);
    static_assert(sizeof(s->sym_rank_plan) == sizeof(p.sym_rank_plan), "nb_sim::sym_rank_plan mirrors LaunchPlan::sym_rank_plan");
    memcpy(s->sym_rank_plan, p.sym_rank_plan, sizeof s->sym_rank_plan);
    s->sym_passes = std::move(p.sym_passes);
    s->sym_local = p.sym_local;
    s->sym_tab_host = std::move(p.sym_tab_host);
    s->sym_spill_rows = p.sym_spill_rows;
    s->sym_pieces = p.sym_pieces;
    s->variant = std::move(p.variant);
}

Shape shape_of(const nb_sim* s)
{
    if (s->sym) return {kSym, s->ipl, 1, s->ws};
    if (s->jpk) return {kJpk, 1, 1, s->ws};
    if (s->direct) return {kDirect, s->ipl, s->ls, s->tl};
    if (s->fused) return {kFused, s->ipl, s->ls, s->tl};
    if (s->sgpr) return {kPkSgpr, s->ipl, 1, s->ws};
    if (s->packed) return {kPkLds, s->ipl, s->ls, s->tl};
    return {kScalar, s->ipl, s->ls, 1};
}

// Rows of one partial-sum layer of a symmetric handle: the padded system, or (rank form: compact layers) the handle's own super-blocks.
size_t sym_layer_rows(const nb_sim* s) { return s->sym_rank ? (size_t)(s->sym_g1 - s->sym_g0) * ipb_of(shape_of(s)) : (size_t)s->sym_np; }

// The packed f32 K1 forms stream their j-bodies as (x, y, z, G*m) rows (nb_internal.h, `gm`).
bool streams_gm(const nb_sim* s) { return !s->f64 && s->packed && !s->jpk; }
bool gm_active(const nb_sim* s) { return streams_gm(s) && (float)s->G != 1.0f; }
const void* jstream(const nb_sim* s, int k) { return gm_active(s) ? s->gm[k] : s->bodies[k]; }

// Makes gm[cur] current (allocates on first use; a no-op for G == 1 and for the kernels that fold G themselves).
int ensure_gm(nb_sim* s)
{
    if (!gm_active(s)) { s->gm_ok = false; return NB_OK; }      // steps taken meanwhile leave any old copy behind
    const size_t bytes = (size_t)16 * (s->sym ? s->sym_np : s->n);
    for (int k = 0; k < (s->fused ? 2 : 1); ++k)
        if (!s->gm[k]) {
            NB_HIP(s, hipMalloc(&s->gm[k], bytes));
            if (s->sym) NB_HIP(s, hipMemsetAsync(s->gm[k], 0, bytes, s->stream));     // zero-mass rows past n
        }
    if (s->gm_ok && s->gm_G == s->G) return NB_OK;
    const float4* b = (const float4*)s->bodies[s->cur];
    float4* g = (float4*)s->gm[s->cur];
    uint32_t n = s->n;
    float G = (float)s->G;
    void* args[] = {&b, &g, &n, &G};
    NB_HIP(s, hipLaunchKernel((const void*)&nb::nb_gm_pack<0>, dim3(ceil_div(n, nb::kBlock)), dim3(nb::kBlock), args, 0, s->stream));
    s->gm_ok = true; s->gm_G = s->G;
    return NB_OK;
}

// part: 0 = all splits, 1 = only the splits inside this shard's own rows,
//       2 = all the others
// t0/t1 (optional): events stamped at the kernel's begin / end (hipExtLaunchKernel)
void launch_kernel(const void* fn, dim3 grid, dim3 block, void** args, hipStream_t stream, hipEvent_t t0, hipEvent_t t1)
{
    // error picked up by hipGetLastError in nb_step
    if (t0 || t1) (void)hipExtLaunchKernel(fn, grid, block, args, 0, stream, t0, t1, 0);
    else (void)hipLaunchKernel(fn, grid, block, args, 0, stream);
}

template <typename T>
void launch_force(nb_sim* s, int part = 0, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr)
{
    using V4 = typename nb::vec4<T>::type;
    const Shape sh = shape_of(s);
    if (s->sym_rank) {
        // part 0: every wave; 1: phase A (travelers = own rows; nothing of the other ranks is read); 2: phase B
        const nbp::LaunchPlan::SymPass& ps = s->sym_passes[s->sym_pass];
        nb::SymRankPlan rp;
        memcpy(&rp, ps.plan, sizeof rp);
        uint32_t w0 = part == 2 ? rp.WA : 0u, w1 = part == 1 ? rp.WA : rp.WA + rp.WB;
        if (w1 <= w0) return;
        const void* b = jstream(s, s->cur);
        void* p = s->partial;
        const uint32_t* tab = s->sym_tab + ps.tab_off;
        uint32_t n = s->n, k_lo = ps.k_lo, k_hi = ps.k_hi, d0 = ps.d0;
        void* sp = s->sym_spill;
        if (s->f64) {
            double G = s->G, e2 = s->eps2;
            void* args[] = {&b, &p, &tab, &rp, &n, &G, &e2, &w0, &w1, &sp, &k_lo, &k_hi, &d0};
            launch_kernel(rank_kernel_of(true, sh.ipl), dim3(ceil_div(w1 - w0, 4u)), dim3(256), args, s->stream, t0, t1);
        } else {
            float e2 = (float)s->eps2;
            void* args[] = {&b, &p, &tab, &rp, &n, &e2, &w0, &w1, &sp, &k_lo, &k_hi, &d0};
            launch_kernel(rank_kernel_of(false, sh.ipl), dim3(ceil_div(w1 - w0, 4u)), dim3(256), args, s->stream, t0, t1);
        }
        return;
    }
    if (s->symw) {
        nb::SymWPlan pl;
        memcpy(&pl, s->sym_plan, sizeof pl);
        const void* b = jstream(s, s->cur);                // f32: rows (x, y, z, G*m); f64: (x, y, z, m) and G
        void* p = s->partial;
        const uint32_t* tab = s->sym_tab;
        void* sp = s->sym_spill;                           // one row set per wave (wave ranges cut inside sweeps); null with whole sweeps
        // (the order of kernels/symmetric.hip.h SYMW_PLAN_PARAMS: the table pointer and the plan words inside the preloaded 14 dwords)
        uint32_t* queue = s->sym_queue;
        uint32_t npieces = s->sym_pieces, pieces_off = 2u * (pl.np / ipb_of(sh)) + 4u * pl.W;      // the queued sweeps behind the wave records (lay_out_symw)
        if (npieces) (void)hipMemsetAsync(queue, 0, sizeof(uint32_t), s->stream);                  // the queue's draw counter
        if (s->f64) {
            double G = s->G, e2 = s->eps2;
            void* args[] = {&tab, &b, &p, &sp, &pl.W, &pl.ups, &pl.nsb, &pl.zc, &pl.r_layer0, &pl.t_layer0, &G, &e2, &queue, &npieces, &pieces_off};
            launch_kernel(kernel_of(true, sh), dim3(ceil_div(pl.W, 4u)), dim3(256), args, s->stream, t0, t1);
        } else {
            float e2 = (float)s->eps2;
            void* args[] = {&tab, &b, &p, &sp, &pl.W, &pl.ups, &pl.nsb, &pl.zc, &pl.r_layer0, &pl.t_layer0, &e2, &queue, &npieces, &pieces_off};
            launch_kernel(kernel_of(false, sh), dim3(ceil_div(pl.W, 4u)), dim3(256), args, s->stream, t0, t1);
        }
        return;
    }
    if (s->sym) {
        nb::SymPlan pl;
        memcpy(&pl, s->sym_plan, sizeof pl);
        const float4* b = (const float4*)jstream(s, s->cur);
        nb::SymRow* p = (nb::SymRow*)s->partial;       // 12-byte rows
        float e2 = (float)s->eps2;
        uint32_t n = s->n;
        void* args[] = {&b, &p, &pl, &n, &e2};
        launch_kernel(kernel_of(false, sh), dim3(pl.nsb * pl.q), dim3(64 * sh.x), args, s->stream, t0, t1);
        return;
    }
    nb::SplitWindow win{0, 0xffffffffu, 0};
    uint32_t ny = s->jsplit;
    if (part == 1) { win.base = s->own_split0; ny = s->own_splits; }
    else if (part == 2) { win.hole_begin = s->own_split0; win.hole_count = s->own_splits; ny = s->jsplit - s->own_splits; }
    if (ny == 0) return;
    dim3 grid(ceil_div(s->sc, ipb_of(sh)), ny), block(nb::kBlock);
    const V4* b = (const V4*)jstream(s, s->cur);    // packed f32 forms: rows (x, y, z, G*m); scalar template: (x, y, z, m) and G
    V4* p = (V4*)s->partial;
    T G = (T)s->G, e2 = (T)s->eps2;
    uint32_t n = s->n, sb = s->sb, sc = s->sc, jps = s->j_per_split;
    const V4* zr = (const V4*)s->zero_row;          // LDS-DMA source for j past the range (packed LDS-tile kernels)
    void* args[] = {&b, &p, &n, &sb, &sc, &G, &e2, &jps, &win, &zr};   // every K1 form declares exactly these ten parameters
    launch_kernel(kernel_of(s->f64, sh), grid, block, args, s->stream, t0, t1);
}

// The fused one-launch step: reads bodies[cur], writes bodies[cur ^ 1], then the roles flip.
// The pair-transposed copy the j-packed step streams from: rebuilt from bodies[cur] whenever the positions
// were written from outside the step (upload, a raw device pointer handed out) or G changed (it is folded
// into the mass lanes); the step itself keeps it current.
void ensure_pairs(nb_sim* s)
{
    if (!s->jpk || (s->pairs_ok && s->pairs_G == s->G)) return;
    const float4* b = (const float4*)s->bodies[s->cur];
    float4* p = (float4*)s->pairs[s->cur];
    uint32_t n = s->n;
    float G = (float)s->G;
    void* args[] = {&b, &p, &n, &G};
    (void)hipLaunchKernel((const void*)&nb::nb_pairs_pack<0>, dim3(ceil_div(ceil_div(n, 2u), nb::kBlock)), dim3(nb::kBlock), args, 0, s->stream);
    s->pairs_ok = true; s->pairs_G = s->G;
}

void launch_jpk(nb_sim* s, hipEvent_t t0, hipEvent_t t1)
{
    const Shape sh = shape_of(s);
    dim3 grid(ceil_div(s->n, 64u), s->jsplit), block(64 * jpk_ws(sh.x));
    const float4 *bin = (const float4*)s->bodies[s->cur], *pin = (const float4*)s->pairs[s->cur];
    float4 *bout = (float4*)s->bodies[s->cur ^ 1], *pout = (float4*)s->pairs[s->cur ^ 1];
    float4 *v = (float4*)s->vel, *a = (float4*)s->acc, *part = (float4*)s->jpartial;
    uint32_t* tk = s->tickets;
    uint32_t n = s->n, upw = s->junits, poison = (s->poison ? 1u : 0u) | (s->jpk_fenced ? 2u : 0u);
    float G = (float)s->G, e2 = (float)s->eps2, dt = (float)s->dt;
    void* args[] = {&bin, &pin, &bout, &pout, &v, &a, &part, &tk, &n, &upw, &poison, &G, &e2, &dt};
    launch_kernel(kernel_of(false, sh), grid, block, args, s->stream, t0, t1);
    s->cur ^= 1;
}

void launch_fused(nb_sim* s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr)
{
    if (s->jpk) { ensure_pairs(s); launch_jpk(s, t0, t1); return; }
    const Shape sh = shape_of(s);
    dim3 grid(ceil_div(s->n, ipb_of(sh))), block(nb::kBlock);
    const float4 *bin = (const float4*)s->bodies[s->cur], *jin = (const float4*)jstream(s, s->cur);
    float4 *bout = (float4*)s->bodies[s->cur ^ 1], *gout = gm_active(s) ? (float4*)s->gm[s->cur ^ 1] : nullptr;
    float4 *v = (float4*)s->vel, *a = (float4*)s->acc;
    uint32_t n = s->n;
    float G = (float)s->G, e2 = (float)s->eps2, dt = (float)s->dt;
    const float4* zr = (const float4*)s->zero_row;
    void* args[] = {&bin, &jin, &bout, &gout, &v, &a, &n, &G, &e2, &dt, &zr};   // nb_step_fused and nb_step_direct: the same eleven
    launch_kernel(kernel_of(false, sh), grid, block, args, s->stream, t0, t1);
    s->cur ^= 1;
}

template <typename T>
void launch_integrate(nb_sim* s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr)
{
    using V4 = typename nb::vec4<T>::type;
    V4 *b = (V4*)s->bodies[s->cur], *v = (V4*)s->vel;
    uint32_t sb = s->sb, sc = s->sc, js = s->jsplit;
    T dt = (T)s->dt, G = (T)s->G;
    // the j-stream rows of the new positions; a handle whose rows are exchanged rebuilds the whole copy after the gather instead
    V4* gout = gm_active(s) && !(s->xfn || s->rccl) ? (V4*)s->gm[s->cur] : nullptr;
    if (s->sym_rank) {
        // A rank-form handle (a shard, or a whole system whose ring distances go in passes) has no layered integrate: its sums
        // are reduced into sym_A by nb_sym_reduce and the PLAIN integrate kernel reads the handle's rows of it (sym_rank_phase_b_t).
        // lay_out_symw_rank also sets `symw` (its SymWPlan is a summary for the reports): without this branch nb_integrate_symw would
        // walk the rank table as {first wave, layers} pairs and stride compact layers by np -- reads far past `partial`.
        V4* a = (V4*)s->acc;
        const V4* p = (const V4*)s->sym_A + sb;
        uint32_t one = 1;
        V4* none = nullptr;                         // the (x, y, z, G*m) copy is rebuilt whole after the position all-gather
        void* args[] = {&b, &v, &a, &p, &sb, &sc, &one, &dt, &none, &G};
        launch_kernel((const void*)&nb::nb_integrate<T, 1>, dim3(ceil_div(sc, nb::kBlock)), dim3(nb::kBlock), args, s->stream, t0, t1);
        return;
    }
    if (s->symw) {
        nb::SymWPlan pl;
        memcpy(&pl, s->sym_plan, sizeof pl);
        V4* aa = (V4*)s->acc;
        const nb::SymRowT<T>* pp = (const nb::SymRowT<T>*)s->partial;
        const uint32_t* tab = s->sym_tab;
        uint32_t n = s->n, S = ipb_of(shape_of(s)), ch_shift = s->ws == 3 ? 6u : 7u;      // travelers per chunk: X = 3 one per lane (64), X = 1 two (128)
        uint32_t shifts = (uint32_t)__builtin_ctz(S) | ch_shift << 8;                       // (S is a power of two: 64 * residents per lane)
        uint32_t spill_off = pl.ups > 1 ? 2u * (pl.np / S) + 4u * pl.W : 0u;                // the spill lists behind the wave records (lay_out_symw)
        const void* sp = s->sym_spill;
        void* args[] = {&tab, &pp, &b, &v, &aa, &n, &shifts, &pl.np, &spill_off, &sp, &gout, &dt, &G, &pl.t_layer0, &pl.r_layer0, &pl.nsb, &pl.n_hi, &pl.H, &pl.zc};
        launch_kernel((const void*)&nb::nb_integrate_symw<T, 8>, dim3(ceil_div(n * 8u, nb::kBlock)), dim3(nb::kBlock), args, s->stream, t0, t1);
        return;
    }
    if (s->sym) {
        if constexpr (std::is_same<T, float>::value) {
            nb::SymPlan pl;
            memcpy(&pl, s->sym_plan, sizeof pl);
            float4 *bb = (float4*)b, *vv = (float4*)v, *aa = (float4*)s->acc, *gg = (float4*)gout;
            const nb::SymRow* pp = (const nb::SymRow*)s->partial;
            uint32_t n = s->n, S = ipb_of(shape_of(s));
            float fdt = (float)s->dt, fG = (float)s->G;
            void* args[] = {&bb, &vv, &aa, &pp, &n, &pl, &S, &fdt, &gg, &fG};
            launch_kernel((const void*)&nb::nb_integrate_sym<8>, dim3(ceil_div(n * 8u, nb::kBlock)), dim3(nb::kBlock), args, s->stream, t0, t1);
        }
        return;
    }
    if (s->swap_acc) {
        // jsplit == 1: the single partial array IS a_new; K2 reads it beside a_old and the two
        // buffers swap roles (no 16-B store of a per body: 96 B per body in all)
        const V4 *ao = (const V4*)s->acc, *an = (const V4*)s->partial;
        void* args[] = {&b, &v, &ao, &an, &sb, &sc, &dt, &gout, &G};
        launch_kernel((const void*)&nb::nb_integrate_swap<T>, dim3(ceil_div(sc, nb::kBlock)), dim3(nb::kBlock), args, s->stream, t0, t1);
        std::swap(s->acc, s->partial);
        s->acc_parity ^= 1;
        return;
    }
    // lanes per body: enough to keep ~8 partial loads per lane at most
    const int R = js >= 32 ? 8 : js >= 8 ? 4 : 1;
    V4* a = (V4*)s->acc;
    const V4* p = (const V4*)s->partial;
    void* args[] = {&b, &v, &a, &p, &sb, &sc, &js, &dt, &gout, &G};
    const void* fn = R == 8 ? (const void*)&nb::nb_integrate<T, 8> : R == 4 ? (const void*)&nb::nb_integrate<T, 4> : (const void*)&nb::nb_integrate<T, 1>;
    launch_kernel(fn, dim3(ceil_div(sc * (uint32_t)R, nb::kBlock)), dim3(nb::kBlock), args, s->stream, t0, t1);
}

void launch_step(nb_sim* s)
{
    if (s->fused) { launch_fused(s); return; }
    if (s->f64) { launch_force<double>(s); launch_integrate<double>(s); }
    else { launch_force<float>(s); launch_integrate<float>(s); }
}

constexpr uint32_t kGraphChunk = 16;   // even: buffer roles (ping-pong, acc swap) are back where they started
constexpr uint32_t kGraphBig = 128;
constexpr uint32_t kGraphSteps[2] = {kGraphChunk, kGraphBig};

void drop_graph(nb_sim* s)
{
    for (auto& g : s->graphs) {
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
        if (g.graph) { (void)hipGraphDestroy(g.graph); g.graph = nullptr; }
    }
}

// buffer-role parity a captured graph is valid for
int role_parity(const nb_sim* s) { return s->cur | (s->acc_parity << 1); }

// Captures kGraphSteps[which] steps on the engine's own stream.  Returns false (and disables graphs
// for the handle) if anything goes wrong; the caller then issues plain launches -- same
// kernels, same results.
bool ensure_graph(nb_sim* s, int which)
{
    auto& slot = s->graphs[which];
    if (slot.exec && slot.dt == s->dt && slot.G == s->G && slot.parity == role_parity(s)) return true;
    if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
    if (slot.graph) { (void)hipGraphDestroy(slot.graph); slot.graph = nullptr; }
    void *acc0 = s->acc, *par0 = s->partial;
    const int cur0 = s->cur, par_bit0 = s->acc_parity;
    ensure_pairs(s);             // not part of the captured steps
    if (ensure_gm(s) != NB_OK) { s->graphs_ok = false; return false; }
    if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { s->graphs_ok = false; return false; }
    for (uint32_t k = 0; k < kGraphSteps[which]; ++k) launch_step(s);
    hipGraph_t g = nullptr;
    const bool ok = hipStreamEndCapture(s->stream, &g) == hipSuccess && g;
    s->acc = acc0; s->partial = par0; s->cur = cur0; s->acc_parity = par_bit0;   // an even number of role flips: explicit for clarity
    if (!ok) { (void)hipGetLastError(); s->graphs_ok = false; return false; }
    hipGraphExec_t ge = nullptr;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); (void)hipGetLastError(); s->graphs_ok = false; return false; }
    slot.graph = g; slot.exec = ge; slot.dt = s->dt; slot.G = s->G; slot.parity = role_parity(s);
    return true;
}

// Makes the engine stream wait for an all-gather started by the two-phase hook / the
// overlapped native collective.
int finish_gather(nb_sim* s)
{
    if (!s->gather_pending) return NB_OK;
    s->gather_pending = false;
    if (s->rccl) return nbi::rccl_exchange_wait(s);
    if (!s->xwait) return NB_OK;
    const int rc = s->xwait(s->xuser, (void*)s->stream);
    if (rc != 0) return fail(s, NB_ERR_COMM, "exchange wait hook failed with code " + std::to_string(rc));
    return NB_OK;
}

int get_events(nb_sim* s, nb_events* out)
{
    if (s->pool_next == s->pool.size()) {
        if (s->pool.size() >= 4096) return 1;   // stop recording, keep running
        nb_events t;
        t.two = t.xchg = t.rs = false;
        for (auto& e : t.e)
            if (hipEventCreate(&e) != hipSuccess) return 1;
        s->pool.push_back(t);
    }
    *out = s->pool[s->pool_next++];
    out->two = out->xchg = out->rs = false;
    return 0;
}

void free_frames(nb_sim* s)
{
    for (auto& f : s->frame) {
        if (f.landed) (void)hipEventSynchronize(f.landed);
        if (f.h_bodies) (void)hipHostFree(f.h_bodies);     // h_speed / d_speed point into the same allocations
        if (f.d_bodies) (void)hipFree(f.d_bodies);
        if (f.packed) (void)hipEventDestroy(f.packed);
        if (f.landed) (void)hipEventDestroy(f.landed);
        f = nb_frame_slot();
    }
    if (s->frame_stream) { (void)hipStreamDestroy(s->frame_stream); s->frame_stream = nullptr; }
    s->frame_next = 0; s->frame_latest = -1;
}

}  // namespace

namespace nbi {

// Rank form of the symmetric pass, first half of a step: the force pass over the chunk lists of the handle's own super-blocks,
// then this rank's sums for every row of the system into sym_A.
template <typename T>
int sym_rank_phase_a_t(nb_sim* s, hipEvent_t after_force, bool split_at_gather, nb_events* stamps = nullptr)
{
    using V4 = typename nb::vec4<T>::type;
    const uint32_t npass = (uint32_t)s->sym_passes.size();
    for (uint32_t q = 0; q < npass; ++q) {
        s->sym_pass = q;
        if (split_at_gather && q == 0) {
            if (s->sym_passes[0].plan[13] == 0 || s->sym_passes[0].plan[14] == 0) stamps = nullptr;      // (WA, WB: a phase without waves launches nothing to stamp)
            // Two launches with the wait for the all-gather between them.  Timed steps stamp each launch at its own begin and end
            // (hipExtLaunchKernel: e[0]..e[7] and e[3]..e[4]), so that the wait for the other ranks' rows is NOT counted as force time.
            launch_force<T>(s, 1, stamps ? stamps->e[0] : nullptr, stamps ? stamps->e[7] : nullptr);     // own-row travelers: needs nothing from the other ranks
            if (int rc = finish_gather(s)) return rc;     // the engine stream waits for their rows here
            launch_force<T>(s, 2, stamps ? stamps->e[3] : nullptr, stamps ? stamps->e[4] : nullptr);
            if (stamps) { stamps->two = true; after_force = nullptr; }
        } else {
            launch_force<T>(s);
        }
        if (after_force && q + 1 == npass) NB_HIP(s, hipEventRecord(after_force, s->stream));
        // this pass's sums for every row: into sym_A (first pass) or on top of it
        const nbp::LaunchPlan::SymPass& ps = s->sym_passes[q];
        nb::SymRankPlan rp;
        memcpy(&rp, ps.plan, sizeof rp);
        const nb::SymRowT<T>* p = (const nb::SymRowT<T>*)s->partial;
        const uint32_t* tab = s->sym_tab + ps.tab_off;
        V4* A = (V4*)s->sym_A;
        uint32_t S = ipb_of(shape_of(s));
        const void* sp = s->sym_spill;
        uint32_t d0 = ps.d0, d1 = ps.k_hi == 0xffffffffu ? 0xffffffffu : ps.k_hi / (S / 64u), acc = q ? 1u : 0u;
        void* args[] = {&p, &tab, &A, &rp, &S, &sp, &d0, &d1, &acc};
        NB_HIP(s, hipLaunchKernel((const void*)&nb::nb_sym_reduce<T>, dim3(ceil_div(rp.np, nb::kBlock)), dim3(nb::kBlock), args, 0, s->stream));
    }
    s->sym_pass = 0;
    return NB_OK;
}

int sym_rank_phase_a(nb_sim* s, void* after_force, bool split_at_gather, nb_events* stamps)
{
    if (!s->sym_rank) return fail(s, NB_ERR_STATE, "sym_rank_phase_a: not a rank-form handle");
    if (int rc = ensure_gm(s)) return rc;
    return s->f64 ? sym_rank_phase_a_t<double>(s, (hipEvent_t)after_force, split_at_gather, stamps) : sym_rank_phase_a_t<float>(s, (hipEvent_t)after_force, split_at_gather, stamps);
}

template <typename T>
int sym_rank_phase_b_t(nb_sim* s)
{
    launch_integrate<T>(s);                          // its rank-form branch: nb_integrate<T, 1> on the handle's rows of sym_A
    NB_HIP(s, hipGetLastError());
    return NB_OK;
}

// Second half: the plain integrate kernel on the handle's rows of the (reduce-scattered) sym_A.
int sym_rank_phase_b(nb_sim* s)
{
    if (int rc = s->f64 ? sym_rank_phase_b_t<double>(s) : sym_rank_phase_b_t<float>(s)) return rc;
    ++s->steps_done;
    s->gm_ok = false;
    return NB_OK;
}

}  // namespace nbi

extern "C" {

#ifdef NB_STAMPS
/* Diagnostic build only (`make stamps`): the per-wave s_memtime stamps of the last launch that carries them (kernels/common.hip.h,
 * NB_STAMP): 16 words per wave, `waves` waves.  tools/stamps_symw.py. */
__attribute__((visibility("default"))) int nb_debug_stamps(uint64_t* out, uint32_t waves)
{
    static unsigned long long* buf = nullptr;
    constexpr size_t kWaves = 16384;
    if (!buf) {
        if (hipMalloc((void**)&buf, kWaves * 16 * sizeof(unsigned long long)) != hipSuccess) return NB_ERR_HIP;
        (void)hipMemset(buf, 0, kWaves * 16 * sizeof(unsigned long long));
        if (hipMemcpyToSymbol(HIP_SYMBOL(nb::nb_stamp_buf), &buf, sizeof buf) != hipSuccess) return NB_ERR_HIP;
    }
    if (out && waves) {
        if (waves > kWaves) waves = kWaves;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, buf, (size_t)waves * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return NB_ERR_HIP;
    }
    return NB_OK;
}
#endif

uint32_t nb_abi_version(void) { return NB_ABI_VERSION; }
uint32_t nb_abi_minor(void) { return NB_ABI_MINOR; }

int nb_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

// nb_create's and nb_plan_query's reading of a caller's nb_config: a normalised copy, the shard and the softening
static int read_config(const nb_config* cfg_in, const char* who, nb_config* cfg, uint32_t* sb_out, uint32_t* sc_out, double* eps2_out)
{
    const std::string w(who);
    if (cfg_in->struct_size < offsetof(nb_config, reserved))
        return fail(nullptr, NB_ERR_INVALID, w + ": struct_size too small (set it to sizeof(nb_config))");
    memset(cfg, 0, sizeof *cfg);
    memcpy(cfg, cfg_in, cfg_in->struct_size < sizeof *cfg ? cfg_in->struct_size : sizeof *cfg);
    if (cfg->n == 0) return fail(nullptr, NB_ERR_INVALID, w + ": n must be >= 1");
    if (cfg->n > (1u << 30)) return fail(nullptr, NB_ERR_INVALID, w + ": n must be <= 2^30 (32-bit row arithmetic; the reference passes N as an f32, exact to 2^24: nbody3d.js:246)");
    if (cfg->precision > NB_F64) return fail(nullptr, NB_ERR_INVALID, w + ": unknown precision");
    if (cfg->tile != 0 && cfg->tile != (uint32_t)nb::kTile)
        return fail(nullptr, NB_ERR_INVALID, w + ": only tile = 256 is built (reference TILE_SIZE)");
    const double eps2 = cfg->eps2 == 0.0 ? 1e-4 : cfg->eps2;
    if (!(eps2 >= 1e-12))
        return fail(nullptr, NB_ERR_INVALID, w + ": eps2 must be >= 1e-12 (branch-free self term needs it)");
    uint32_t sb = cfg->shard_begin, sc = cfg->shard_count;
    if (sc == 0) { sb = 0; sc = cfg->n; }
    if ((uint64_t)sb + sc > cfg->n) return fail(nullptr, NB_ERR_INVALID, w + ": shard exceeds n");
    *sb_out = sb; *sc_out = sc; *eps2_out = eps2;
    return NB_OK;
}

int nb_create(const nb_config* cfg_in, nb_sim** out)
{
    if (out) *out = nullptr;
    if (!cfg_in || !out) return fail(nullptr, NB_ERR_INVALID, "nb_create: null argument");
    nb_config cfg;
    uint32_t sb, sc;
    double eps2;
    if (const int rc = read_config(cfg_in, "nb_create", &cfg, &sb, &sc, &eps2)) return rc;

    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, NB_ERR_NO_DEVICE,
                    std::string("nb_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                        "); this engine has no CPU fallback");
    int dev = cfg.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= count) return fail(nullptr, NB_ERR_INVALID, "nb_create: device ordinal out of range");

    nb_sim* s = new (std::nothrow) nb_sim;
    if (!s) return fail(nullptr, NB_ERR_NOMEM, "nb_create: out of host memory");
    s->n = cfg.n; s->sb = sb; s->sc = sc;
    s->f64 = cfg.precision == NB_F64;
    s->esz = s->f64 ? 8 : 4;
    s->eps2 = eps2;
    s->device = dev;

    auto bail = [&](int code, const std::string& msg) {
        std::string m = msg;
        nb_destroy(s);
        return fail(nullptr, code, m);
    };
#define NB_HIPC(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return bail(NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

    NB_HIPC(hipSetDevice(dev));
    hipDeviceProp_t prop;
    NB_HIPC(hipGetDeviceProperties(&prop, dev));
    if (cfg.ext_stream || (cfg.flags & NB_FLAG_EXT_STREAM)) { s->stream = (hipStream_t)cfg.ext_stream; s->own_stream = false; }
    else { NB_HIPC(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)); s->own_stream = true; }

    const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const double clock_hz = prop.clockRate > 0 ? 1e3 * prop.clockRate : 2.4e9;     // clockRate is in kHz
    plan_handle(s, cfg, n_cu, clock_hz, (double)prop.totalGlobalMem);
    if (s->sym && (!s->sym_rank || s->sym_local) && !cfg.force_variant) {
        // The planner budgets the symmetric pass's layers against the device's TOTAL memory; what is FREE right now may be less
        // (other handles, other processes).  A whole-system handle then re-plans against the free memory instead of failing in
        // hipMalloc.  (A rank-form shard does not: its peers would still expect the reduce-scatter -- it fails loudly below.)
        size_t free_b = 0, total_b = 0;
        auto need = [&]() { return 3.0 * s->esz * (double)sym_layer_rows(s) * s->sym_layers + 12.0 * s->esz * s->sym_np; };
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need() > 0.9 * (double)free_b) {
            // first with a layer budget of what IS free: the ring distances then go in passes that reuse the layers (still every
            // unordered pair once); only if even that does not fit, the ordered-pair kernels
            nb_config fit = cfg;
            fit.layer_budget_mib = (uint32_t)std::max(1.0, 0.6 * (double)free_b / 1048576.0);
            plan_handle(s, fit, n_cu, clock_hz, (double)prop.totalGlobalMem);
            if (s->sym && need() > 0.9 * (double)free_b) {
                cfg.flags |= NB_FLAG_NO_SYM;
                plan_handle(s, cfg, n_cu, clock_hz, (double)prop.totalGlobalMem);
            }
        }
    }
    if (!kernel_of(s->f64, shape_of(s))) return bail(NB_ERR_INVALID, "nb_create: no kernel for shape " + s->variant);

    const size_t row = 4 * s->esz;
    if (cfg.ext_bodies) { s->bodies[0] = cfg.ext_bodies; s->own_bodies = false; }
    else {
        const size_t rows = s->sym ? s->sym_np : s->n;        // the symmetric pass reads whole super-blocks: zero-mass rows past n
        NB_HIPC(hipMalloc(&s->bodies[0], row * rows));
        if (s->sym) NB_HIPC(hipMemset(s->bodies[0], 0, row * rows));
        s->own_bodies = true;
        if (s->fused) NB_HIPC(hipMalloc(&s->bodies[1], row * s->n));
    }
    NB_HIPC(hipMalloc(&s->vel, row * s->sc));
    NB_HIPC(hipMalloc(&s->acc, row * s->sc));
    // `partial` is allocated exactly ONCE, with the size the handle's own kernels index (round 3's memory fault: an if / else-if
    // split allocated it a second time with the ordered-pair size -- 16 n q = 8.4 MB at N = 32,768, q = 16 -- under a
    // symmetric handle that needed 12 np (q + H + 1) = 9.4 MB; profiles/r03/README.md)
    auto alloc_partial = [&](size_t bytes) -> hipError_t {
        if (s->partial) return hipErrorInvalidValue;
        s->partial_bytes = bytes;
        return hipMalloc(&s->partial, bytes);
    };
    if (s->sym) {
        // layers of (x, y, z) rows: 12 bytes (24 in f64); a rank-form handle's layers hold the rows of its own super-blocks only
        NB_HIPC(alloc_partial((size_t)3 * s->esz * sym_layer_rows(s) * s->sym_layers));
        if (s->sym_rank) NB_HIPC(hipMalloc(&s->sym_A, 4 * s->esz * s->sym_np));
        if (s->sym_spill_rows) {
            // zeroed once: a wave that never spills (its range starts at a sweep boundary) leaves its row alone, and nobody reads it
            NB_HIPC(hipMalloc(&s->sym_spill, (size_t)3 * s->esz * s->sym_spill_rows));
            NB_HIPC(hipMemset(s->sym_spill, 0, (size_t)3 * s->esz * s->sym_spill_rows));
        }
        if (s->symw && s->sym_pieces) {
            NB_HIPC(hipMalloc((void**)&s->sym_queue, 64));
            NB_HIPC(hipMemset(s->sym_queue, 0, 64));
        }
        if (s->symw) {
            NB_HIPC(hipMalloc((void**)&s->sym_tab, sizeof(uint32_t) * s->sym_tab_host.size()));
            NB_HIPC(hipMemcpy(s->sym_tab, s->sym_tab_host.data(), sizeof(uint32_t) * s->sym_tab_host.size(), hipMemcpyHostToDevice));
        }
    } else if (!s->fused) {
        NB_HIPC(alloc_partial(row * s->sc * s->jsplit));
    }
    if (s->partial_bytes != (s->sym ? (size_t)3 * s->esz * sym_layer_rows(s) * s->sym_layers : s->fused ? (size_t)0 : row * s->sc * s->jsplit))
        return bail(NB_ERR_STATE, "nb_create: the partial-sum buffer does not have the size this handle's kernels index");
    if (s->jpk) {
        // pairs: whole 4-pair units (128 B) plus one spare the loop's last request may touch; everything past the
        // system stays zero (zero-mass bodies at the origin).  Partials: 64 rows per (split, i-block).
        const size_t pbytes = ((size_t)ceil_div(ceil_div(s->n, 2u), 4u) + 3) * 128;
        const uint32_t iblocks = ceil_div(s->n, 64u);
        for (auto& p : s->pairs) { NB_HIPC(hipMalloc(&p, pbytes)); NB_HIPC(hipMemset(p, 0, pbytes)); }
        NB_HIPC(hipMalloc((void**)&s->tickets, sizeof(uint32_t) * iblocks));
        NB_HIPC(hipMemset(s->tickets, 0, sizeof(uint32_t) * iblocks));
        if (s->jsplit > 1) {
            NB_HIPC(hipMalloc(&s->jpartial, (size_t)16 * 64 * iblocks * s->jsplit));
            NB_HIPC(hipMemset(s->jpartial, 0xff, (size_t)16 * 64 * iblocks * s->jsplit));   // NaN until written
        }
        NB_HIPC(hipDeviceSynchronize());
        s->poison = (cfg.flags & NB_FLAG_POISON) != 0;
        s->jpk_fenced = (cfg.flags & NB_FLAG_JPK_FENCED) != 0;
    }
    NB_HIPC(hipMalloc(&s->zero_row, 64));                     // a zero-mass body at the origin: what the LDS-DMA staging
    NB_HIPC(hipMemsetAsync(s->zero_row, 0, 64, s->stream));   // of the packed tile kernels reads for rows past the range (stream-ordered)
    // diagnostics: a grid of (row blocks) x (j-chunks); the chunk is sized so that a few thousand workgroups share the pairs
    s->diag_chunk = nb::kTile * std::min(16u, std::max(1u, s->n / 16384u));
    s->diag_blocks = ceil_div(s->sc, nb::kDiagRows) * ceil_div(s->n, s->diag_chunk);
    NB_HIPC(hipMalloc((void**)&s->diag, sizeof(double) * 5 * s->diag_blocks));
#undef NB_HIPC
    *out = s;
    return NB_OK;
}

void nb_destroy(nb_sim* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)finish_gather(s);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    nbi::rccl_release(s);
    drop_graph(s);
    free_frames(s);
    for (auto& t : s->pool)
        for (auto& e : t.e) (void)hipEventDestroy(e);
    if (s->own_bodies) {
        if (s->bodies[0]) (void)hipFree(s->bodies[0]);
        if (s->bodies[1]) (void)hipFree(s->bodies[1]);
    }
    if (s->vel) (void)hipFree(s->vel);
    if (s->acc) (void)hipFree(s->acc);
    if (s->partial) (void)hipFree(s->partial);
    for (auto& p : s->pairs) if (p) (void)hipFree(p);
    for (auto& p : s->gm) if (p) (void)hipFree(p);
    if (s->jpartial) (void)hipFree(s->jpartial);
    if (s->tickets) (void)hipFree(s->tickets);
    if (s->sym_tab) (void)hipFree(s->sym_tab);
    if (s->sym_spill) (void)hipFree(s->sym_spill);
    if (s->sym_queue) (void)hipFree(s->sym_queue);
    if (s->sym_A) (void)hipFree(s->sym_A);
    if (s->diag) (void)hipFree(s->diag);
    if (s->zero_row) (void)hipFree(s->zero_row);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

const char* nb_last_error(nb_sim* s) { return s ? s->err.c_str() : g_create_error.c_str(); }

int nb_upload(nb_sim* s, const void* bodies, const void* vel, const void* accel)
{
    if (!s) return NB_ERR_INVALID;
    if (!bodies || !vel) return fail(s, NB_ERR_INVALID, "nb_upload: bodies and vel are required");
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    const size_t row = 4 * s->esz;
    // the reference's writeBuffer copies out of the typed array before returning
    // (nbody3d.js:186,193): synchronous copies, host pointers are not retained
    NB_HIP(s, hipStreamSynchronize(s->stream));
    NB_HIP(s, hipMemcpy(s->bodies[s->cur], bodies, row * s->n, hipMemcpyHostToDevice));
    NB_HIP(s, hipMemcpy(s->vel, (const char*)vel + row * s->sb, row * s->sc, hipMemcpyHostToDevice));
    if (accel) NB_HIP(s, hipMemcpy(s->acc, (const char*)accel + row * s->sb, row * s->sc, hipMemcpyHostToDevice));
    else {
        // WebGPU zero-init, nbody3d.js:195-199.  On the ENGINE stream: a hipMemset on the null stream is
        // asynchronous for device memory and the engine's non-blocking stream does not wait for it
        // -- the fused step reads accel in its first microsecond.
        NB_HIP(s, hipMemsetAsync(s->acc, 0, row * s->sc, s->stream));
        NB_HIP(s, hipStreamSynchronize(s->stream));
    }
    s->uploaded = true;
    s->pairs_ok = false;
    s->gm_ok = false;
    return NB_OK;
}

int nb_set_params(nb_sim* s, double dt, double G)
{
    if (!s) return NB_ERR_INVALID;
    if (!(dt == dt) || !(G == G)) return fail(s, NB_ERR_INVALID, "nb_set_params: NaN");
    s->dt = dt; s->G = G; s->params_set = true;
    return NB_OK;
}

int nb_step(nb_sim* s, uint32_t nsteps)
{
    if (!s) return NB_ERR_INVALID;
    if (!s->uploaded) return fail(s, NB_ERR_STATE, "nb_step: nb_upload has not been called");
    if (!s->params_set) return fail(s, NB_ERR_STATE, "nb_step: nb_set_params has not been called");
    if (!(s->dt > 0.0)) return NB_OK;   // `if (dt > 0)` gate, nbody3d.js:474
    NB_HIP(s, hipSetDevice(s->device));
    const bool exchange = s->xfn || s->rccl;
    ensure_pairs(s);      // the j-packed step's position copy, if something outside the step rewrote the positions or G
    if (gm_active(s) && !s->gm_ok) { if (int rc = finish_gather(s)) return rc; }
    if (int rc = ensure_gm(s)) return rc;   // the packed K1 forms' (x, y, z, G*m) j-stream, likewise
    // Multi-step calls on the engine's own stream replay a captured graph of
    // kGraphChunk steps (no exchange, no per-kernel timing requested).
    if (s->own_stream && s->graphs_ok && !exchange && !s->timing && !s->sym_rank && nsteps >= kGraphChunk) {
        for (int which = 1; which >= 0; --which) {
            while (nsteps >= kGraphSteps[which] && ensure_graph(s, which)) {
                NB_HIP(s, hipGraphLaunch(s->graphs[which].exec, s->stream));
                nsteps -= kGraphSteps[which];
                s->steps_done += kGraphSteps[which];
            }
        }
    }
    for (uint32_t k = 0; k < nsteps; ++k) {
        if (exchange && gm_active(s) && !s->gm_ok) {
            // G != 1 on a handle whose rows are exchanged: the other ranks' rows of the j-stream copy are rebuilt from the
            // gathered positions (one small launch per step; the overlapped form waits here, so it only overlaps at G == 1,
            // which every multi-GPU configuration of BASELINE.json uses)
            if (int rc = finish_gather(s)) return rc;
            if (int rc = ensure_gm(s)) return rc;
        }
        if (s->sym_rank) {
            // rank form of the symmetric pass: force pass -> this rank's sums for every row -> reduce-scatter across the ranks
            // -> integrate own rows -> all-gather of the new positions
            if (!s->rccl && !s->sym_local) return fail(s, NB_ERR_STATE, "nb_step: an NB_FLAG_SYM_SHARD handle needs nb_rccl_attach (or nb_multi) for its reduce-scatter");
            // an overlapped all-gather of the previous step still in flight: the sweeps whose travelers are this rank's own rows
            // (phase A, ~1 / ranks of the work) are issued before the engine stream waits for it
            const bool split = s->gather_pending;
            if (!split) { if (int rc = finish_gather(s)) return rc; }
            nb_events evr;
            const bool recr = s->timing && get_events(s, &evr) == 0;
            // one force launch: plain records around it; split at the gather: each launch stamped at its own begin / end (evr.two)
            if (recr) NB_HIP(s, hipEventRecord(evr.e[0], s->stream));
            if (int rc = nbi::sym_rank_phase_a(s, recr ? evr.e[7] : nullptr, split, recr && split ? &evr : nullptr)) return rc;
            if (recr) { NB_HIP(s, hipEventRecord(evr.e[1], s->stream)); evr.rs = true; }
            if (!s->sym_local) { if (int rc = nbi::rccl_reduce_scatter_A(s)) return rc; }
            if (recr) NB_HIP(s, hipEventRecord(evr.e[6], s->stream));
            if (int rc = nbi::sym_rank_phase_b(s)) return rc;
            if (recr) NB_HIP(s, hipEventRecord(evr.e[2], s->stream));
            NB_HIP(s, hipGetLastError());
            if (s->sym_local) { if (recr) s->pending.push_back(evr); continue; }      // a whole system on this device: nothing to exchange
            if (int rc = nbi::rccl_exchange_begin(s)) return rc;
            if (nbi::rccl_overlapped(s)) s->gather_pending = true;        // waited for inside the next force pass (or by whoever reads the positions first)
            else if (recr) { NB_HIP(s, hipEventRecord(evr.e[5], s->stream)); evr.xchg = true; }
            if (recr) s->pending.push_back(evr);
            continue;
        }
        nb_events ev;
        const bool rec = s->timing && get_events(s, &ev) == 0;
        hipEvent_t* e = rec ? ev.e : nullptr;
        if (s->fused) {
            launch_fused(s, e ? e[0] : nullptr, e ? e[1] : nullptr);
        } else {
            if (s->gather_pending) {
                // the previous step's all-gather is still in flight: own-row splits first
                if (s->f64) launch_force<double>(s, 1, e ? e[0] : nullptr, e ? e[1] : nullptr);
                else launch_force<float>(s, 1, e ? e[0] : nullptr, e ? e[1] : nullptr);
                if (int rc = finish_gather(s)) return rc;
                if (s->f64) launch_force<double>(s, 2, e ? e[3] : nullptr, e ? e[4] : nullptr);
                else launch_force<float>(s, 2, e ? e[3] : nullptr, e ? e[4] : nullptr);
                if (rec) ev.two = s->own_splits > 0 && s->own_splits < s->jsplit;
            } else {
                if (s->f64) launch_force<double>(s, 0, e ? e[0] : nullptr, e ? e[1] : nullptr);
                else launch_force<float>(s, 0, e ? e[0] : nullptr, e ? e[1] : nullptr);
            }
            if (s->f64) launch_integrate<double>(s, e ? e[6] : nullptr, e ? e[2] : nullptr);
            else launch_integrate<float>(s, e ? e[6] : nullptr, e ? e[2] : nullptr);
        }
        NB_HIP(s, hipGetLastError());
        ++s->steps_done;
        if (exchange) s->gm_ok = false;
        if (s->rccl) {
            if (int rc = nbi::rccl_exchange_begin(s)) return rc;
            if (nbi::rccl_overlapped(s)) {
                s->gather_pending = true;
                if (s->own_splits == 0) { if (int rc2 = finish_gather(s)) return rc2; }
            } else if (rec) {
                NB_HIP(s, hipEventRecord(ev.e[5], s->stream));
                ev.xchg = true;
            }
        } else if (s->xfn) {
            int rc = s->xfn(s->xuser, s->bodies[s->cur], s->esz, s->n, s->sb, s->sc, (void*)s->stream);
            if (rc != 0) return fail(s, NB_ERR_COMM, "nb_step: exchange hook failed with code " + std::to_string(rc));
            if (s->xwait) {
                s->gather_pending = true;
                // nothing to overlap with: splits do not line up with the shard
                if (s->own_splits == 0) { if (int rc2 = finish_gather(s)) return rc2; }
            }
        }
        if (rec) s->pending.push_back(ev);
    }
    return NB_OK;
}

int nb_sync(nb_sim* s)
{
    if (!s) return NB_ERR_INVALID;
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    NB_HIP(s, hipStreamSynchronize(s->stream));
    return NB_OK;
}

int nb_download(nb_sim* s, void* bodies, void* vel, void* accel)
{
    if (!s) return NB_ERR_INVALID;
    if (!s->uploaded) return fail(s, NB_ERR_STATE, "nb_download: nothing uploaded yet");
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    NB_HIP(s, hipStreamSynchronize(s->stream));
    const size_t row = 4 * s->esz;
    if (bodies) NB_HIP(s, hipMemcpy(bodies, s->bodies[s->cur], row * s->n, hipMemcpyDeviceToHost));
    if (vel) NB_HIP(s, hipMemcpy((char*)vel + row * s->sb, s->vel, row * s->sc, hipMemcpyDeviceToHost));
    if (accel) NB_HIP(s, hipMemcpy((char*)accel + row * s->sb, s->acc, row * s->sc, hipMemcpyDeviceToHost));
    return NB_OK;
}

int nb_device_ptr(nb_sim* s, int which, void** out)
{
    if (!s || !out) return NB_ERR_INVALID;
    switch (which) {
        case NB_BODIES: *out = s->bodies[s->cur]; s->pairs_ok = false; s->gm_ok = false; break;   // the caller may write through it
        case NB_VEL: *out = s->vel; break;
        case NB_ACCEL: *out = s->acc; break;
        default: return fail(s, NB_ERR_INVALID, "nb_device_ptr: unknown array");
    }
    return NB_OK;
}

int nb_set_exchange(nb_sim* s, nb_exchange_fn fn, void* user)
{
    if (!s) return NB_ERR_INVALID;
    if (fn && s->fused) return fail(s, NB_ERR_STATE, "nb_set_exchange: a fused whole-system handle has nothing to exchange");
    if (fn && s->rccl) return fail(s, NB_ERR_STATE, "nb_set_exchange: a native RCCL communicator is attached (nb_rccl_detach first)");
    if (int rc = finish_gather(s)) return rc;
    s->xfn = fn; s->xwait = nullptr; s->xuser = user;
    return NB_OK;
}

int nb_set_exchange_overlapped(nb_sim* s, nb_exchange_fn begin, nb_exchange_wait_fn wait, void* user)
{
    if (!s) return NB_ERR_INVALID;
    if (!begin || !wait) return fail(s, NB_ERR_INVALID, "nb_set_exchange_overlapped: both hooks are required");
    if (s->fused) return fail(s, NB_ERR_STATE, "nb_set_exchange_overlapped: a fused whole-system handle has nothing to exchange");
    if (s->rccl) return fail(s, NB_ERR_STATE, "nb_set_exchange_overlapped: a native RCCL communicator is attached");
    if (int rc = finish_gather(s)) return rc;
    s->xfn = begin; s->xwait = wait; s->xuser = user;
    return NB_OK;
}

int nb_enable_timing(nb_sim* s, int on)
{
    if (!s) return NB_ERR_INVALID;
    s->timing = on != 0;
    return NB_OK;
}

// Averages of the recorded steps since the last call; clears the record.
static int collect_times(nb_sim* s, nb_step_timing* t)
{
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    NB_HIP(s, hipStreamSynchronize(s->stream));
    double f = 0, red = 0, rs = 0, g = 0, x = 0, span = 0;
    uint32_t nx = 0, nrs = 0;
    for (auto& ev : s->pending) {
        float a = 0, b = 0, c = 0, d = 0, r1 = 0, r2 = 0, sp = 0;
        if (ev.rs) {
            // rank form: e0 force e7 nb_sym_reduce e1 reduce-scatter e6 integrate e2 [all-gather e5]; with the force pass split at the
            // gather: e0 own-row sweeps e7 [wait for the other ranks' rows] e3 the rest e4 nb_sym_reduce e1 ...
            NB_HIP(s, hipEventElapsedTime(&a, ev.e[0], ev.e[7]));
            NB_HIP(s, hipEventElapsedTime(&r1, ev.two ? ev.e[4] : ev.e[7], ev.e[1]));
            NB_HIP(s, hipEventElapsedTime(&r2, ev.e[1], ev.e[6]));
            ++nrs;
        } else {
            NB_HIP(s, hipEventElapsedTime(&a, ev.e[0], ev.e[1]));
        }
        if (ev.two) NB_HIP(s, hipEventElapsedTime(&c, ev.e[3], ev.e[4]));   // own-row work, [gather wait], the rest
        if (!s->fused) NB_HIP(s, hipEventElapsedTime(&b, ev.e[6], ev.e[2]));
        if (ev.xchg) { NB_HIP(s, hipEventElapsedTime(&d, ev.e[2], ev.e[5])); ++nx; }
        NB_HIP(s, hipEventElapsedTime(&sp, ev.e[0], ev.xchg ? ev.e[5] : s->fused ? ev.e[1] : ev.e[2]));
        f += a + c; red += r1; rs += r2; g += b; x += d; span += sp;
    }
    const uint32_t cnt = (uint32_t)s->pending.size();
    t->launches = cnt;
    t->force_ms = cnt ? f / cnt : 0.0;
    t->sym_reduce_ms = nrs ? red / nrs : 0.0;
    t->reduce_scatter_ms = nrs ? rs / nrs : 0.0;
    t->integrate_ms = cnt ? g / cnt : 0.0;
    t->allgather_ms = nx ? x / nx : 0.0;
    t->span_ms = cnt ? span / cnt : 0.0;
    t->reduce_scatters = nrs;
    t->allgathers = nx;
    s->pending.clear();
    s->pool_next = 0;
    return NB_OK;
}

int nb_step_times2(nb_sim* s, nb_step_timing* out)
{
    if (!s) return NB_ERR_INVALID;
    if (!out || out->struct_size < sizeof(nb_step_timing)) return fail(s, NB_ERR_INVALID, "nb_step_times2: set out->struct_size to sizeof(nb_step_timing)");
    nb_step_timing t;
    memset(&t, 0, sizeof t);
    if (int rc = collect_times(s, &t)) return rc;
    t.struct_size = out->struct_size;
    memcpy(out, &t, sizeof t);
    return NB_OK;
}

int nb_step_times(nb_sim* s, double* force_ms, double* integrate_ms, double* exchange_ms, uint32_t* launches)
{
    if (!s) return NB_ERR_INVALID;
    nb_step_timing t;
    memset(&t, 0, sizeof t);
    if (int rc = collect_times(s, &t)) return rc;
    if (force_ms) *force_ms = t.force_ms + t.sym_reduce_ms;        // the rank form's nb_sym_reduce counts as force work here
    if (integrate_ms) *integrate_ms = t.integrate_ms;
    if (exchange_ms) *exchange_ms = t.reduce_scatter_ms + t.allgather_ms;   // every native collective of a step
    if (launches) *launches = t.launches;
    return NB_OK;
}

int nb_kernel_times(nb_sim* s, double* force_ms, double* integrate_ms, uint32_t* launches)
{
    return nb_step_times(s, force_ms, integrate_ms, nullptr, launches);
}

int nb_integrate_pass(nb_sim* s, uint32_t reps, double* avg_ms)
{
    if (!s || !avg_ms) return NB_ERR_INVALID;
    if (!s->uploaded) return fail(s, NB_ERR_STATE, "nb_integrate_pass: nothing uploaded yet");
    if (s->fused) return fail(s, NB_ERR_STATE, "nb_integrate_pass: a fused handle has no integrate kernel (create it with NB_FLAG_NO_FUSE)");
    if (reps == 0) return fail(s, NB_ERR_INVALID, "nb_integrate_pass: reps must be >= 1");
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    const double dt = s->dt > 0 ? s->dt : 1e-3;
    const double keep = s->dt;
    s->dt = dt;
    if (s->steps_done == 0) {
        NB_HIP(s, hipMemsetAsync(s->partial, 0, s->sym ? (size_t)3 * s->esz * sym_layer_rows(s) * s->sym_layers : 4 * s->esz * s->sc * s->jsplit, s->stream));
        if (s->sym_rank) NB_HIP(s, hipMemsetAsync(s->sym_A, 0, 4 * s->esz * s->sym_np, s->stream));      // what a rank-form handle integrates from
    }
    hipEvent_t e0, e1;
    NB_HIP(s, hipEventCreate(&e0));
    NB_HIP(s, hipEventCreate(&e1));
    if (s->f64) launch_integrate<double>(s); else launch_integrate<float>(s);     // warm-up launch
    NB_HIP(s, hipEventRecord(e0, s->stream));
    for (uint32_t k = 0; k < reps; ++k) { if (s->f64) launch_integrate<double>(s); else launch_integrate<float>(s); }
    NB_HIP(s, hipEventRecord(e1, s->stream));
    NB_HIP(s, hipEventSynchronize(e1));
    float ms = 0;
    NB_HIP(s, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    NB_HIP(s, hipGetLastError());
    s->dt = keep;
    drop_graph(s);          // buffer roles may have changed parity
    *avg_ms = ms / reps;
    return NB_OK;
}

int nb_force_pass(nb_sim* s, uint32_t reps, double* avg_ms)
{
    if (!s || !avg_ms) return NB_ERR_INVALID;
    if (!s->uploaded) return fail(s, NB_ERR_STATE, "nb_force_pass: nothing uploaded yet");
    if (s->fused) return fail(s, NB_ERR_STATE, "nb_force_pass: a fused handle has no separate force kernel (create it with NB_FLAG_NO_FUSE)");
    if (reps == 0) return fail(s, NB_ERR_INVALID, "nb_force_pass: reps must be >= 1");
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    if (int rc = ensure_gm(s)) return rc;
    auto once = [&]() -> int {
        if (s->sym_rank) return s->f64 ? nbi::sym_rank_phase_a_t<double>(s, nullptr, false) : nbi::sym_rank_phase_a_t<float>(s, nullptr, false);
        if (s->f64) launch_force<double>(s); else launch_force<float>(s);
        return NB_OK;
    };
    hipEvent_t e0, e1;
    NB_HIP(s, hipEventCreate(&e0));
    NB_HIP(s, hipEventCreate(&e1));
    if (int rc = once()) return rc;               // warm-up
    NB_HIP(s, hipEventRecord(e0, s->stream));
    for (uint32_t k = 0; k < reps; ++k) { if (int rc = once()) return rc; }
    NB_HIP(s, hipEventRecord(e1, s->stream));
    NB_HIP(s, hipEventSynchronize(e1));
    float ms = 0;
    NB_HIP(s, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    NB_HIP(s, hipGetLastError());
    *avg_ms = ms / reps;
    return NB_OK;
}

const char* nb_variant_name(nb_sim* s) { return s ? s->variant.c_str() : ""; }

int nb_shape_info(nb_sim* s, uint32_t* jsplit, uint32_t* j_per_split, uint32_t* own_split0, uint32_t* own_splits)
{
    if (!s) return NB_ERR_INVALID;
    if (jsplit) *jsplit = s->jsplit;
    if (j_per_split) *j_per_split = s->j_per_split;
    if (own_split0) *own_split0 = s->own_split0;
    if (own_splits) *own_splits = s->own_splits;
    return NB_OK;
}

int nb_plan_query(const nb_config* cfg_in, int n_cu, double clock_hz, nb_plan_info* out, uint32_t* tab, uint32_t tab_cap)
{
    if (!cfg_in || !out) return fail(nullptr, NB_ERR_INVALID, "nb_plan_query: null argument");
    if (out->struct_size < sizeof(nb_plan_info)) return fail(nullptr, NB_ERR_INVALID, "nb_plan_query: set out->struct_size to sizeof(nb_plan_info)");
    nb_config cfg;
    uint32_t sb, sc;
    double eps2;
    if (const int rc = read_config(cfg_in, "nb_plan_query", &cfg, &sb, &sc, &eps2)) return rc;
    int count = 0;
    double device_mem = 0.0;                      // 0: the planner's default (an MI355X's 288 GB)
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    if (n_cu <= 0 || !(clock_hz > 0)) {           // "as on the current device"
        if (count <= 0) return fail(nullptr, NB_ERR_NO_DEVICE, "nb_plan_query: n_cu / clock_hz not given and there is no HIP device to read them from");
        int dev = cfg.device;
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
        hipDeviceProp_t prop;
        if (dev >= count || hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail(nullptr, NB_ERR_HIP, "nb_plan_query: cannot read the device properties");
        if (n_cu <= 0) n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (!(clock_hz > 0)) clock_hz = prop.clockRate > 0 ? 1e3 * prop.clockRate : 2.4e9;
        device_mem = (double)prop.totalGlobalMem;
    }
    nb_sim tmp;                                    // host fields only: nothing of it is ever allocated on a device
    tmp.n = cfg.n; tmp.sb = sb; tmp.sc = sc; tmp.f64 = cfg.precision == NB_F64; tmp.esz = tmp.f64 ? 8 : 4; tmp.eps2 = eps2;
    tmp.no_device = count <= 0;
    plan_handle(&tmp, cfg, n_cu, clock_hz, device_mem);
    const Shape sh = shape_of(&tmp);
    if (!kernel_of(tmp.f64, sh)) return fail(nullptr, NB_ERR_INVALID, "nb_plan_query: no kernel for shape " + tmp.variant);
    const uint32_t size = out->struct_size, want_pass = out->sym_pass;
    memset(out, 0, sizeof *out);
    out->struct_size = size;
    out->kind = (uint32_t)sh.kind; out->ipl = (uint32_t)sh.ipl; out->ls = (uint32_t)sh.ls; out->x = (uint32_t)sh.x;
    out->jsplit = tmp.jsplit; out->j_per_split = tmp.j_per_split; out->own_split0 = tmp.own_split0; out->own_splits = tmp.own_splits;
    out->sym = tmp.sym; out->symw = tmp.symw; out->sym_rank = tmp.sym_rank;
    out->sym_np = tmp.sym_np; out->sym_layers = tmp.sym_layers; out->sym_g0 = tmp.sym_g0; out->sym_g1 = tmp.sym_g1;
    static_assert(sizeof(out->sym_plan) == 11 * sizeof(uint32_t) && sizeof(tmp.sym_plan) >= 12 * sizeof(uint32_t), "nb_plan_info::sym_plan holds the first eleven words of nb_sim::sym_plan; the twelfth (ups) is sym_ups");
    memcpy(out->sym_plan, tmp.sym_plan, sizeof out->sym_plan);
    out->sym_ups = tmp.symw ? tmp.sym_plan[11] : 0;
    out->sym_spill_rows = tmp.sym_spill_rows;
    static_assert(sizeof(out->sym_rank_plan) == sizeof(tmp.sym_rank_plan), "nb_plan_info::sym_rank_plan mirrors nb_sim::sym_rank_plan");
    memcpy(out->sym_rank_plan, tmp.sym_rank_plan, sizeof out->sym_rank_plan);
    out->sym_passes = (uint32_t)tmp.sym_passes.size();
    out->sym_local = tmp.sym_local;
    size_t tab_from = 0, tab_to = tmp.sym_tab_host.size();
    if (!tmp.sym_passes.empty()) {
        if (want_pass >= tmp.sym_passes.size()) return fail(nullptr, NB_ERR_INVALID, "nb_plan_query: sym_pass out of range");
        const auto& ps = tmp.sym_passes[want_pass];
        out->sym_pass = want_pass;
        memcpy(out->sym_rank_plan, ps.plan, sizeof out->sym_rank_plan);
        out->sym_pass_k_lo = ps.k_lo; out->sym_pass_k_hi = ps.k_hi; out->sym_pass_d0 = ps.d0;
        tab_from = ps.tab_off;
        tab_to = want_pass + 1 < tmp.sym_passes.size() ? tmp.sym_passes[want_pass + 1].tab_off : tmp.sym_tab_host.size();
    }
    out->tab_len = (uint32_t)(tab_to - tab_from);
    snprintf(out->variant, sizeof out->variant, "%s", tmp.variant.c_str());
    if (tab) memcpy(tab, tmp.sym_tab_host.data() + tab_from, sizeof(uint32_t) * (out->tab_len < tab_cap ? out->tab_len : tab_cap));
    return NB_OK;
}

int nb_diagnostics(nb_sim* s, double out[5])
{
    if (!s || !out) return NB_ERR_INVALID;
    if (!s->uploaded) return fail(s, NB_ERR_STATE, "nb_diagnostics: nothing uploaded yet");
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    dim3 grid(ceil_div(s->sc, nb::kDiagRows), ceil_div(s->n, s->diag_chunk)), block(nb::kBlock);
    if (s->f64)
        hipLaunchKernelGGL((nb::nb_diag<double>), grid, block, 0, s->stream, (const double4*)s->bodies[s->cur],
                           (const double4*)s->vel, s->n, s->sb, s->sc, s->diag_chunk, s->G, s->eps2, s->diag);
    else
        hipLaunchKernelGGL((nb::nb_diag<float>), grid, block, 0, s->stream, (const float4*)s->bodies[s->cur],
                           (const float4*)s->vel, s->n, s->sb, s->sc, s->diag_chunk, s->G, (float)s->eps2, s->diag);
    NB_HIP(s, hipGetLastError());
    std::vector<double> h((size_t)5 * s->diag_blocks);
    NB_HIP(s, hipMemcpyAsync(h.data(), s->diag, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    for (int q = 0; q < 5; ++q) out[q] = 0.0;
    for (uint32_t b = 0; b < s->diag_blocks; ++b)
        for (int q = 0; q < 5; ++q) out[q] += h[(size_t)b * 5 + q];
    return NB_OK;
}

/* ---- viewer frame feed -------------------------------------------------------------------- */

int nb_frame_request(nb_sim* s)
{
    if (!s) return NB_ERR_INVALID;
    if (!s->uploaded) return fail(s, NB_ERR_STATE, "nb_frame_request: nothing uploaded yet");
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;     // other ranks' rows must have landed
    if (!s->frame_stream) {
        // all or nothing: the stream is only published once every slot has its buffers and events -- a failed
        // allocation (4 x 20*n bytes pinned + device) returns an error and leaves the handle without a frame feed,
        // so that a later request starts over instead of packing into null buffers
        hipStream_t fs = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&fs, hipStreamNonBlocking);
        int slot = 0;
        for (auto& f : s->frame) {
            if (e != hipSuccess) break;
#ifdef NB_TUNING    // calibration / test build only: fail the k-th slot's allocation (tests/test_round3_gpu.py)
            if (const char* inj = getenv("NB_TEST_FAIL_FRAME_SLOT")) { if (atoi(inj) == slot) { e = hipErrorOutOfMemory; break; } }
#endif
            ++slot; (void)slot;
            // one allocation per side (bodies[4n] then speed[n]): ONE device-to-host copy per frame
            e = hipHostMalloc((void**)&f.h_bodies, sizeof(float) * 5 * s->n, hipHostMallocDefault);
            if (e != hipSuccess) break;
            f.h_speed = f.h_bodies + (size_t)4 * s->n;
            memset(f.h_speed, 0, sizeof(float) * s->n);
            e = hipMalloc((void**)&f.d_bodies, sizeof(float) * 5 * s->n);
            if (e != hipSuccess) break;
            f.d_speed = f.d_bodies + (size_t)4 * s->n;
            e = hipMemsetAsync(f.d_speed, 0, sizeof(float) * s->n, s->stream);   // ordered before the pack kernel
            if (e == hipSuccess) e = hipEventCreate(&f.packed);     // stamped by hipExtLaunchKernel
            if (e == hipSuccess) e = hipEventCreateWithFlags(&f.landed, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s->stream);       // the memsets above
            s->frame_stream = fs;                        // free_frames destroys it with the partly built slots
            free_frames(s);
            return fail(s, e == hipErrorOutOfMemory ? NB_ERR_NOMEM : NB_ERR_HIP,
                        std::string("nb_frame_request: cannot set up the frame slots: ") + hipGetErrorString(e));
        }
        s->frame_stream = fs;
    }
    nb_frame_slot& f = s->frame[s->frame_next];
    if (!f.h_bodies || !f.d_bodies || !f.packed || !f.landed) return fail(s, NB_ERR_STATE, "nb_frame_request: frame slot is not initialised");
    // the host copy issued from this slot kFrameSlots requests ago must have finished reading its
    // staging buffer (normally long done; a host that runs further ahead than that is held back
    // here, on the host side); the step stream itself never waits for a copy
    if (f.in_flight) NB_HIP(s, hipEventSynchronize(f.landed));
    // the "packed" event rides on the pack kernel's own completion signal (hipExtLaunchKernel):
    // no marker packet on the step stream, so the next step's kernel is not held behind one
    {
        dim3 grid(ceil_div(s->n, nb::kBlock)), block(nb::kBlock);
        const void* b = s->bodies[s->cur];
        const void* v = s->vel;
        uint32_t n = s->n, sb = s->sb, sc = s->sc;
        float4* ob = (float4*)f.d_bodies;
        float* os = f.d_speed;
        void* args[] = {&b, &v, &n, &sb, &sc, &ob, &os};
        const void* fn = s->f64 ? (const void*)&nb::nb_frame_pack<double> : (const void*)&nb::nb_frame_pack<float>;
        NB_HIP(s, hipExtLaunchKernel(fn, grid, block, args, 0, s->stream, nullptr, f.packed, 0));
    }
    NB_HIP(s, hipStreamWaitEvent(s->frame_stream, f.packed, 0));
    NB_HIP(s, hipMemcpyAsync(f.h_bodies, f.d_bodies, sizeof(float) * 5 * s->n, hipMemcpyDeviceToHost, s->frame_stream));
    NB_HIP(s, hipEventRecord(f.landed, s->frame_stream));
    f.step = s->steps_done;
    f.in_flight = true; f.valid = true;
    s->frame_latest = s->frame_next;
    s->frame_next = (s->frame_next + 1) % nb_sim::kFrameSlots;
    return NB_OK;
}

int nb_frame_acquire(nb_sim* s, int wait, const float** bodies, const float** speed, uint64_t* step_index)
{
    if (!s) return NB_ERR_INVALID;
    if (s->frame_latest < 0) return fail(s, NB_ERR_STATE, "nb_frame_acquire: nb_frame_request has not been called");
    NB_HIP(s, hipSetDevice(s->device));
    int pick = -1;
    for (int k = 0; k < nb_sim::kFrameSlots && pick < 0; ++k) {          // newest first
        const int idx = (s->frame_latest - k + nb_sim::kFrameSlots) % nb_sim::kFrameSlots;
        nb_frame_slot& f = s->frame[idx];
        if (!f.valid) continue;
        if (f.in_flight) {
            if (wait && k == 0) NB_HIP(s, hipEventSynchronize(f.landed));
            const hipError_t q = hipEventQuery(f.landed);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
            if (q != hipSuccess) return fail(s, NB_ERR_HIP, std::string("nb_frame_acquire: ") + hipGetErrorString(q));
            f.in_flight = false;
        }
        pick = idx;
    }
    if (pick < 0) return NB_NOT_READY;
    const nb_frame_slot& f = s->frame[pick];
    if (bodies) *bodies = f.h_bodies;
    if (speed) *speed = f.h_speed;
    if (step_index) *step_index = f.step;
    return NB_OK;
}

}  // extern "C"
