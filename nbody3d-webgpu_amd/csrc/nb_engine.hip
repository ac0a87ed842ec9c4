// nb_engine.hip -- C ABI (include/nbody3d_hip.h) over the HIP particle pool
// and the two gfx950 kernels in nb_kernels.hip.h.
//
// Device state per handle (reference layout float4 AoS,
// nbody3d.js:179-199, kept as-is on the device because one 16-B lane access is
// the widest coalesced load and the j-tile is read back as one ds_read_b128):
//   bodies  : 4*n elements, replicated on every shard (x, y, z, mass)
//   vel     : 4*shard_count elements
//   accel   : 4*shard_count elements (acceleration of the previous step)
//   partial : jsplit * 4*shard_count elements (K1 output, summed by K2)
#include "../../include/nbody3d_hip.h"
#include "nb_kernels.hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

namespace {

thread_local std::string g_create_error = "";

// e0..e1: force launch(es) issued before a pending gather is waited for (or the
// only force launch); e3..e4: force launch issued after it; e1/e4..e2: integrate.
struct EventTriple { hipEvent_t e0, e1, e2, e3, e4; bool two; };

}  // namespace

struct nb_sim {
    uint32_t n = 0, sb = 0, sc = 0;
    bool f64 = false;
    size_t esz = 4;
    int device = 0;
    double eps2 = 1e-4;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* bodies = nullptr;
    bool own_bodies = false;
    void* vel = nullptr;
    void* acc = nullptr;
    void* partial = nullptr;
    double* diag = nullptr;
    uint32_t diag_blocks = 0;
    double dt = 0.0, G = 0.0;
    bool params_set = false, uploaded = false;
    int ipl = 1, ls = 1;
    bool packed = false;   // nb_force_pk (f32 only)
    bool sgpr = false;     // nb_force_pk_sgpr: j broadcast from SGPRs instead of the LDS tile
    bool xcd_remap = false;
    uint32_t jsplit = 1, j_per_split = 0;
    std::string variant, err;
    nb_exchange_fn xfn = nullptr;
    nb_exchange_wait_fn xwait = nullptr;   // non-null: two-phase (overlapped) exchange
    void* xuser = nullptr;
    bool gather_pending = false;           // begin() called, wait() not yet
    uint32_t own_split0 = 0, own_splits = 0;   // j-splits lying entirely inside this shard's rows
    bool timing = false;
    // HIP-graph replay of multi-step calls (launch-bound small N): kGraphChunk
    // [K1,K2] pairs captured once per (dt, G) and replayed
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    double graph_dt = 0.0, graph_G = 0.0;
    bool graphs_ok = true;           // cleared if capture ever fails: fall back to plain launches
    std::vector<EventTriple> pool;   // recycled events
    std::vector<EventTriple> pending;
    size_t pool_next = 0;
};

namespace {

int fail(nb_sim* s, int code, const std::string& msg)
{
    if (s) s->err = msg; else g_create_error = msg;
    return code;
}

#define NB_HIP(s, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail((s), NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));              \
    } while (0)

uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

struct Shape { int ipl, ls; bool pk; bool sgpr = false; };

// The force kernel instantiation for a shape (used for launching and for the
// occupancy query of the launch-shape model).
template <typename T>
const void* force_kernel(const Shape& sh)
{
    if constexpr (std::is_same<T, float>::value) {
        if (sh.pk && sh.sgpr) {
            if (sh.ipl == 4) return (const void*)&nb::nb_force_pk_sgpr<2>;
            return (const void*)&nb::nb_force_pk_sgpr<4>;
        }
        if (sh.pk) {
            if (sh.ipl == 2) return (const void*)&nb::nb_force_pk<1, 1>;
            if (sh.ipl == 4) return (const void*)&nb::nb_force_pk<2, 1>;
            return (const void*)&nb::nb_force_pk<4, 1>;
        }
    }
    if (sh.ls == 1) {
        if (sh.ipl == 1) return (const void*)&nb::nb_force<T, 1, 1>;
        if (sh.ipl == 2) return (const void*)&nb::nb_force<T, 2, 1>;
        return (const void*)&nb::nb_force<T, 4, 1>;
    }
    if (sh.ls == 4) return (const void*)&nb::nb_force<T, 1, 4>;
    if (sh.ls == 16) return (const void*)&nb::nb_force<T, 1, 16>;
    return (const void*)&nb::nb_force<T, 1, 64>;
}

const void* force_kernel_of(const nb_sim* s, const Shape& sh)
{
    return s->f64 ? force_kernel<double>(sh) : force_kernel<float>(sh);
}

// Launch-shape model (inputs measured on MI355X: profiles/r01/sweep_*.txt).
//   grid = (i-blocks, jsplit) workgroups of 4 waves, all with the same amount of work,
//   so a launch runs in rounds of `slots` resident workgroups.  For every kernel shape
//   and every split count the model estimates
//     t = sum over rounds [ max(compute, latency) + prologue ] / balance + K2 time
//       compute  = (split length / 256) * resident workgroups per CU * cycles per tile
//                  (the waves of a SIMD share its issue port)
//       latency  = tiles per split * ~3000 cycles (global load + LDS store + barrier; what
//                  bounds small systems: N = 4,096 with 64 lanes per body is all latency)
//       balance  = 1 - 0.03 / rounds (more rounds even out DVFS/tail: +3..4 % from 1 to 4)
//   and keeps the minimum.  It reproduces the measured optimum at the aligned sizes
//   (N = 262,144: 8 bodies/lane, 4 rounds of 1024) and removes the round-quantisation
//   loss at the others (N = 40,002, the reference's default: 1,580 workgroups on 1,024
//   slots = 0.77 -> 1,020 = 0.996).  A split is any multiple of 8 bodies >= 128 -- not a
//   multiple of the 256-body tile: the kernels run an exact trip count on the last,
//   partial tile -- and there are at most 128 splits.
void choose_shape(nb_sim* s, const nb_config& cfg, int n_cu)
{
    const uint32_t sc = s->sc, n = s->n;
    const uint32_t kMaxSplit = 128, kMinSplitLen = 128;
    const double kTileLatency = 3000.0, kPrologue = 3000.0, kClock = 2.3e9;
    auto ipb_of = [](const Shape& sh) { return (uint32_t)(nb::kBlock / sh.ls) * sh.ipl; };
    auto split_len = [&](uint32_t js) { return ceil_div(ceil_div(n, js), 8u) * 8u; };

    struct Cand { Shape sh; double tile_cycles; };   // SIMD cycles one wave needs for a full 256-body tile
    const Cand f32c[] = {{{8, 1, true}, 65536}, {{4, 1, true}, 34600}, {{2, 1, true}, 17900},
                         {{1, 4, false}, 2400}, {{1, 16, false}, 800}, {{1, 64, false}, 280}};
    const Cand f64c[] = {{{2, 1, false}, 45000}, {{1, 1, false}, 23200},
                         {{1, 4, false}, 6400}, {{1, 16, false}, 1600}, {{1, 64, false}, 400}};
    const Cand* cands = s->f64 ? f64c : f32c;
    const int ncand = s->f64 ? 5 : 6;

    Shape sh{2, 1, false};
    uint32_t js = cfg.jsplit;
    const uint32_t variant = cfg.force_variant;
    if (variant != 0) {
        switch (variant) {
            case 1: sh = {1, 1, false}; break;
            case 2: sh = {2, 1, false}; break;
            case 4: sh = {4, 1, false}; break;
            case 14: sh = {1, 4, false}; break;
            case 116: sh = {1, 16, false}; break;
            case 164: sh = {1, 64, false}; break;
            case 22: sh = {2, 1, true}; break;     // packed across 2 i-bodies
            case 24: sh = {4, 1, true}; break;
            case 28: sh = {8, 1, true}; break;
            case 34: sh = {4, 1, true, true}; break;   // packed, j broadcast from SGPRs (no LDS)
            case 38: sh = {8, 1, true, true}; break;
            default: sh = {2, 1, false}; break;
        }
        if (s->f64) { sh.pk = false; sh.sgpr = false; }
    }
    if (variant == 0 || js == 0) {
        double best_t = 1e300;
        for (int k = 0; k < ncand; ++k) {
            const Cand& c = cands[k];
            if (variant != 0 && !(c.sh.ipl == sh.ipl && c.sh.ls == sh.ls && c.sh.pk == sh.pk)) continue;
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, force_kernel_of(s, c.sh), nb::kBlock, 0) != hipSuccess || occ < 1) {
                (void)hipGetLastError();
                occ = 4;
            }
            if (occ > 8) occ = 8;
            const uint64_t slots = (uint64_t)occ * n_cu;
            const uint32_t iblocks = ceil_div(sc, ipb_of(c.sh));
            uint32_t js_hi = n / kMinSplitLen;
            if (js_hi < 1) js_hi = 1;
            if (js_hi > kMaxSplit) js_hi = kMaxSplit;
            const uint32_t js_lo = cfg.jsplit ? cfg.jsplit : 1, js_top = cfg.jsplit ? cfg.jsplit : js_hi;
            for (uint32_t q = js_lo; q <= js_top; ++q) {
                const uint32_t len = split_len(q), used = ceil_div(n, len);
                const uint64_t blocks = (uint64_t)iblocks * used;
                const uint64_t full = blocks / slots, rem = blocks % slots;
                const double tiles = std::ceil(len / 256.0), frac = len / 256.0;
                // a SIMD with fewer than 4 resident waves cannot keep its issue port full
                // (measured with the pure-ALU loop, profiles/r01/ubench2_mfma_coexec.txt and ubench_run1.txt)
                auto round_cycles = [&](double per_cu) {
                    const double fill = per_cu >= 4 ? 1.0 : per_cu >= 3 ? 0.92 : per_cu >= 2 ? 0.82 : 0.62;
                    return std::max(frac * per_cu * c.tile_cycles / fill, tiles * kTileLatency) + kPrologue;
                };
                double cyc = full * round_cycles(occ);
                if (rem) cyc += round_cycles((double)ceil_div((uint32_t)rem, (uint32_t)n_cu));
                const double rounds = (double)full + (rem ? 1 : 0);
                // every i-block streams all n rows through L2 once per step: what rules out many
                // lanes per body at large N (f64, 64 lanes per body, N=262,144: 550 GB per step)
                const double stream_s = (double)iblocks * n * 4 * s->esz / 8.0e12;
                // K2 reads every split's partial back (and K1 writes it): priced at 2 TB/s so that, when the
                // balance gain is a wash (N = 262,144: 32 vs 64 splits), the smaller HBM footprint wins
                const double t = std::max(cyc / kClock, stream_s) / (1.0 - 0.03 / rounds) + (double)used * sc * 4 * s->esz / 2.0e12 + 3e-6;
                if (t < best_t) { best_t = t; if (variant == 0) sh = c.sh; js = q; }
            }
        }
    }
    if (js < 1) {   // pinned shape outside the model's candidate list: fill ~4096 workgroups
        js = ceil_div((uint32_t)n_cu * 16, ceil_div(sc, ipb_of(sh)));
        const uint32_t hi = n / kMinSplitLen < 1 ? 1 : (n / kMinSplitLen > kMaxSplit ? kMaxSplit : n / kMinSplitLen);
        if (js > hi) js = hi;
        if (js < 1) js = 1;
    }
    // The packed shapes with 4 or 8 bodies per lane run the SGPR-broadcast kernel: measured
    // +3..4 % at N=262,144, +2 % on the 1/8-shard shape, +3.5 % on the 785-body splits of
    // N=40,002 (profiles/r01/sweep_sgpr_vs_lds.txt); the LDS-tile kernel keeps the short
    // splits, the 2-bodies-per-lane shape, f64 and the LS shapes
    if (variant == 0 && sh.pk && sh.ipl >= 4 && !s->f64 && !(cfg.flags & NB_FLAG_LDS_ONLY) && split_len(js) >= 512)
        sh.sgpr = true;
    s->ipl = sh.ipl; s->ls = sh.ls; s->packed = sh.pk; s->sgpr = sh.sgpr;
    s->j_per_split = split_len(js);
    s->jsplit = ceil_div(n, s->j_per_split);   // a split may end up empty after rounding
    // j-splits that lie entirely inside this shard's own rows (overlapped exchange)
    s->own_split0 = 0; s->own_splits = 0;
    if (sc < n && s->sb % s->j_per_split == 0) {
        const uint32_t end = s->sb + sc;
        if (end % s->j_per_split == 0 || end == n) {
            s->own_split0 = s->sb / s->j_per_split;
            s->own_splits = ceil_div(end, s->j_per_split) - s->own_split0;
        }
    }
    char buf[96];
    if (sh.sgpr)
        snprintf(buf, sizeof buf, "f32pk_sgpr_ipl%d_js%u", sh.ipl, s->jsplit);
    else
        snprintf(buf, sizeof buf, "%s%s_lds%d_ipl%d_ls%d_js%u", s->f64 ? "f64" : "f32", sh.pk ? "pk" : "", nb::kTile,
                 sh.ipl, sh.ls, s->jsplit);
    s->variant = buf;
}

// part: 0 = all splits, 1 = only the splits inside this shard's own rows,
//       2 = all the others
template <typename T>
void launch_force(nb_sim* s, int part = 0)
{
    using V4 = typename nb::vec4<T>::type;
    const Shape sh{s->ipl, s->ls, s->packed, s->sgpr};
    const uint32_t ipb = (nb::kBlock / s->ls) * s->ipl;
    nb::SplitWindow win{0, 0xffffffffu, 0, s->xcd_remap ? 1u : 0u};
    uint32_t ny = s->jsplit;
    if (part == 1) { win.base = s->own_split0; ny = s->own_splits; }
    else if (part == 2) { win.hole_begin = s->own_split0; win.hole_count = s->own_splits; ny = s->jsplit - s->own_splits; }
    if (ny == 0) return;
    dim3 grid(ceil_div(s->sc, ipb), ny), block(nb::kBlock);
    const V4* b = (const V4*)s->bodies;
    V4* p = (V4*)s->partial;
    T G = (T)s->G, e2 = (T)s->eps2;
    uint32_t n = s->n, sb = s->sb, sc = s->sc, jps = s->j_per_split;
    void* args[] = {&b, &p, &n, &sb, &sc, &G, &e2, &jps, &win};
    (void)hipLaunchKernel(force_kernel<T>(sh), grid, block, args, 0, s->stream);   // error picked up by hipGetLastError
}

template <typename T>
void launch_integrate(nb_sim* s)
{
    using V4 = typename nb::vec4<T>::type;
    // lanes per body: enough to keep ~8 partial loads per lane at most
    const int R = s->jsplit >= 32 ? 8 : s->jsplit >= 8 ? 4 : 1;
    dim3 grid(ceil_div(s->sc * (uint32_t)R, nb::kBlock)), block(nb::kBlock);
#define NB_K2(RR)                                                                                                   \
    hipLaunchKernelGGL((nb::nb_integrate<T, RR>), grid, block, 0, s->stream, (V4*)s->bodies, (V4*)s->vel,           \
                       (V4*)s->acc, (const V4*)s->partial, s->sb, s->sc, s->jsplit, (T)s->dt)
    if (R == 8) NB_K2(8); else if (R == 4) NB_K2(4); else NB_K2(1);
#undef NB_K2
}

constexpr uint32_t kGraphChunk = 16;

void drop_graph(nb_sim* s)
{
    if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
    if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
}

// Captures kGraphChunk steps of [force, integrate] on the engine's own stream.
// Returns false (and disables graphs for the handle) if anything goes wrong;
// the caller then issues plain launches -- same kernels, same results.
bool ensure_graph(nb_sim* s)
{
    if (s->graph_exec && s->graph_dt == s->dt && s->graph_G == s->G) return true;
    drop_graph(s);
    if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { s->graphs_ok = false; return false; }
    for (uint32_t k = 0; k < kGraphChunk; ++k) {
        if (s->f64) { launch_force<double>(s); launch_integrate<double>(s); }
        else { launch_force<float>(s); launch_integrate<float>(s); }
    }
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(s->stream, &g) != hipSuccess || !g) { (void)hipGetLastError(); s->graphs_ok = false; return false; }
    hipGraphExec_t ge = nullptr;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); (void)hipGetLastError(); s->graphs_ok = false; return false; }
    s->graph = g; s->graph_exec = ge; s->graph_dt = s->dt; s->graph_G = s->G;
    return true;
}

// Makes the engine stream wait for an all-gather started by the two-phase hook.
int finish_gather(nb_sim* s)
{
    if (!s->gather_pending) return NB_OK;
    s->gather_pending = false;
    if (!s->xwait) return NB_OK;
    const int rc = s->xwait(s->xuser, (void*)s->stream);
    if (rc != 0) return fail(s, NB_ERR_COMM, "exchange wait hook failed with code " + std::to_string(rc));
    return NB_OK;
}

int get_events(nb_sim* s, EventTriple* out)
{
    if (s->pool_next == s->pool.size()) {
        if (s->pool.size() >= 4096) return 1;   // stop recording, keep running
        EventTriple t;
        t.two = false;
        if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess ||
            hipEventCreate(&t.e2) != hipSuccess || hipEventCreate(&t.e3) != hipSuccess ||
            hipEventCreate(&t.e4) != hipSuccess)
            return 1;
        s->pool.push_back(t);
    }
    *out = s->pool[s->pool_next++];
    return 0;
}

}  // namespace

extern "C" {

uint32_t nb_abi_version(void) { return NB_ABI_VERSION; }

int nb_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int nb_create(const nb_config* cfg_in, nb_sim** out)
{
    if (out) *out = nullptr;
    if (!cfg_in || !out) return fail(nullptr, NB_ERR_INVALID, "nb_create: null argument");
    if (cfg_in->struct_size < offsetof(nb_config, reserved))
        return fail(nullptr, NB_ERR_INVALID, "nb_create: struct_size too small (set it to sizeof(nb_config))");
    nb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    memcpy(&cfg, cfg_in, cfg_in->struct_size < sizeof cfg ? cfg_in->struct_size : sizeof cfg);
    if (cfg.n == 0) return fail(nullptr, NB_ERR_INVALID, "nb_create: n must be >= 1");
    if (cfg.precision > NB_F64) return fail(nullptr, NB_ERR_INVALID, "nb_create: unknown precision");
    if (cfg.tile != 0 && cfg.tile != (uint32_t)nb::kTile)
        return fail(nullptr, NB_ERR_INVALID, "nb_create: only tile = 256 is built (reference TILE_SIZE)");
    const double eps2 = cfg.eps2 == 0.0 ? 1e-4 : cfg.eps2;
    if (!(eps2 >= 1e-12))
        return fail(nullptr, NB_ERR_INVALID, "nb_create: eps2 must be >= 1e-12 (branch-free self term needs it)");
    uint32_t sb = cfg.shard_begin, sc = cfg.shard_count;
    if (sc == 0) { sb = 0; sc = cfg.n; }
    if ((uint64_t)sb + sc > cfg.n) return fail(nullptr, NB_ERR_INVALID, "nb_create: shard exceeds n");

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
    s->xcd_remap = (cfg.flags & NB_FLAG_XCD_REMAP) != 0;

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

    choose_shape(s, cfg, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);

    const size_t row = 4 * s->esz;
    if (cfg.ext_bodies) { s->bodies = cfg.ext_bodies; s->own_bodies = false; }
    else { NB_HIPC(hipMalloc(&s->bodies, row * s->n)); s->own_bodies = true; }
    NB_HIPC(hipMalloc(&s->vel, row * s->sc));
    NB_HIPC(hipMalloc(&s->acc, row * s->sc));
    NB_HIPC(hipMalloc(&s->partial, row * s->sc * s->jsplit));
    s->diag_blocks = ceil_div(s->sc, nb::kBlock);
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
    drop_graph(s);
    for (auto& t : s->pool) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); (void)hipEventDestroy(t.e2); (void)hipEventDestroy(t.e3); (void)hipEventDestroy(t.e4); }
    if (s->own_bodies && s->bodies) (void)hipFree(s->bodies);
    if (s->vel) (void)hipFree(s->vel);
    if (s->acc) (void)hipFree(s->acc);
    if (s->partial) (void)hipFree(s->partial);
    if (s->diag) (void)hipFree(s->diag);
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
    NB_HIP(s, hipMemcpy(s->bodies, bodies, row * s->n, hipMemcpyHostToDevice));
    NB_HIP(s, hipMemcpy(s->vel, (const char*)vel + row * s->sb, row * s->sc, hipMemcpyHostToDevice));
    if (accel) NB_HIP(s, hipMemcpy(s->acc, (const char*)accel + row * s->sb, row * s->sc, hipMemcpyHostToDevice));
    else NB_HIP(s, hipMemset(s->acc, 0, row * s->sc));   // WebGPU zero-init, nbody3d.js:195-199
    s->uploaded = true;
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
    // Multi-step calls on the engine's own stream replay a captured graph of
    // kGraphChunk steps (no exchange hook, no per-kernel timing requested).
    if (s->own_stream && s->graphs_ok && !s->xfn && !s->timing && nsteps >= kGraphChunk) {
        while (nsteps >= kGraphChunk && ensure_graph(s)) {
            NB_HIP(s, hipGraphLaunch(s->graph_exec, s->stream));
            nsteps -= kGraphChunk;
        }
    }
    for (uint32_t k = 0; k < nsteps; ++k) {
        EventTriple ev;
        const bool rec = s->timing && get_events(s, &ev) == 0;
        if (rec) NB_HIP(s, hipEventRecord(ev.e0, s->stream));
        if (s->gather_pending) {
            // the previous step's all-gather is still in flight: own-row splits first
            if (s->f64) launch_force<double>(s, 1); else launch_force<float>(s, 1);
            if (rec) NB_HIP(s, hipEventRecord(ev.e1, s->stream));
            if (int rc = finish_gather(s)) return rc;
            if (rec) NB_HIP(s, hipEventRecord(ev.e3, s->stream));
            if (s->f64) launch_force<double>(s, 2); else launch_force<float>(s, 2);
            if (rec) { NB_HIP(s, hipEventRecord(ev.e4, s->stream)); ev.two = true; }
        } else {
            ev.two = false;
            if (s->f64) launch_force<double>(s); else launch_force<float>(s);
            if (rec) NB_HIP(s, hipEventRecord(ev.e1, s->stream));
        }
        if (s->f64) launch_integrate<double>(s); else launch_integrate<float>(s);
        if (rec) NB_HIP(s, hipEventRecord(ev.e2, s->stream));
        NB_HIP(s, hipGetLastError());
        if (s->xfn) {
            int rc = s->xfn(s->xuser, s->bodies, s->esz, s->n, s->sb, s->sc, (void*)s->stream);
            if (rc != 0) return fail(s, NB_ERR_COMM, "nb_step: exchange hook failed with code " + std::to_string(rc));
            if (s->xwait) {
                s->gather_pending = true;
                // nothing to overlap with: splits do not line up with the shard, or last step of the call
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
    if (bodies) NB_HIP(s, hipMemcpy(bodies, s->bodies, row * s->n, hipMemcpyDeviceToHost));
    if (vel) NB_HIP(s, hipMemcpy((char*)vel + row * s->sb, s->vel, row * s->sc, hipMemcpyDeviceToHost));
    if (accel) NB_HIP(s, hipMemcpy((char*)accel + row * s->sb, s->acc, row * s->sc, hipMemcpyDeviceToHost));
    return NB_OK;
}

int nb_device_ptr(nb_sim* s, int which, void** out)
{
    if (!s || !out) return NB_ERR_INVALID;
    switch (which) {
        case NB_BODIES: *out = s->bodies; break;
        case NB_VEL: *out = s->vel; break;
        case NB_ACCEL: *out = s->acc; break;
        default: return fail(s, NB_ERR_INVALID, "nb_device_ptr: unknown array");
    }
    return NB_OK;
}

int nb_set_exchange(nb_sim* s, nb_exchange_fn fn, void* user)
{
    if (!s) return NB_ERR_INVALID;
    if (int rc = finish_gather(s)) return rc;
    s->xfn = fn; s->xwait = nullptr; s->xuser = user;
    return NB_OK;
}

int nb_set_exchange_overlapped(nb_sim* s, nb_exchange_fn begin, nb_exchange_wait_fn wait, void* user)
{
    if (!s) return NB_ERR_INVALID;
    if (!begin || !wait) return fail(s, NB_ERR_INVALID, "nb_set_exchange_overlapped: both hooks are required");
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

int nb_kernel_times(nb_sim* s, double* force_ms, double* integrate_ms, uint32_t* launches)
{
    if (!s) return NB_ERR_INVALID;
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    NB_HIP(s, hipStreamSynchronize(s->stream));
    double f = 0, g = 0;
    for (auto& ev : s->pending) {
        float a = 0, b = 0, c = 0;
        NB_HIP(s, hipEventElapsedTime(&a, ev.e0, ev.e1));
        if (ev.two) {   // own-row splits, [gather wait], remaining splits
            NB_HIP(s, hipEventElapsedTime(&c, ev.e3, ev.e4));
            NB_HIP(s, hipEventElapsedTime(&b, ev.e4, ev.e2));
        } else {
            NB_HIP(s, hipEventElapsedTime(&b, ev.e1, ev.e2));
        }
        f += a + c; g += b;
    }
    const uint32_t cnt = (uint32_t)s->pending.size();
    if (force_ms) *force_ms = cnt ? f / cnt : 0.0;
    if (integrate_ms) *integrate_ms = cnt ? g / cnt : 0.0;
    if (launches) *launches = cnt;
    s->pending.clear();
    s->pool_next = 0;
    return NB_OK;
}

const char* nb_variant_name(nb_sim* s) { return s ? s->variant.c_str() : ""; }

int nb_diagnostics(nb_sim* s, double out[5])
{
    if (!s || !out) return NB_ERR_INVALID;
    if (!s->uploaded) return fail(s, NB_ERR_STATE, "nb_diagnostics: nothing uploaded yet");
    NB_HIP(s, hipSetDevice(s->device));
    if (int rc = finish_gather(s)) return rc;
    dim3 grid(s->diag_blocks), block(nb::kBlock);
    if (s->f64)
        hipLaunchKernelGGL((nb::nb_diag<double>), grid, block, 0, s->stream, (const double4*)s->bodies,
                           (const double4*)s->vel, s->n, s->sb, s->sc, s->G, s->eps2, s->diag);
    else
        hipLaunchKernelGGL((nb::nb_diag<float>), grid, block, 0, s->stream, (const float4*)s->bodies,
                           (const float4*)s->vel, s->n, s->sb, s->sc, s->G, s->eps2, s->diag);
    NB_HIP(s, hipGetLastError());
    std::vector<double> h((size_t)5 * s->diag_blocks);
    NB_HIP(s, hipMemcpyAsync(h.data(), s->diag, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    for (int q = 0; q < 5; ++q) out[q] = 0.0;
    for (uint32_t b = 0; b < s->diag_blocks; ++b)
        for (int q = 0; q < 5; ++q) out[q] += h[(size_t)b * 5 + q];
    return NB_OK;
}

/* ------------------------------------------------------------------------- *
 * nb_multi: g shard handles in one process, peer-copy all-gather             *
 * ------------------------------------------------------------------------- */
}  // extern "C" (nb_multi struct below needs C++ members)

struct nb_multi {
    uint32_t n = 0, rows = 0, padded_n = 0, g = 0;
    size_t esz = 4;
    std::vector<nb_sim*> shard;
    std::vector<hipEvent_t> ev_k2, ev_copied;   // per shard: "own rows written", "all foreign rows received"
    bool copied_pending = false;
    std::vector<char> pad_b, pad_v, pad_a;       // host staging for the zero-mass padding rows
    std::string err;
};

namespace {

int mfail(nb_multi* m, int code, const std::string& msg) { if (m) m->err = msg; else g_create_error = msg; return code; }

#define NB_MHIP(m, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return mfail((m), NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

extern "C" {

int nb_multi_create(const nb_config* cfg_in, uint32_t n_shards, const int32_t* devices, nb_multi** out)
{
    if (out) *out = nullptr;
    if (!cfg_in || !out || n_shards == 0) return mfail(nullptr, NB_ERR_INVALID, "nb_multi_create: bad argument");
    if (cfg_in->struct_size < offsetof(nb_config, reserved))
        return mfail(nullptr, NB_ERR_INVALID, "nb_multi_create: struct_size too small");
    nb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    memcpy(&cfg, cfg_in, cfg_in->struct_size < sizeof cfg ? cfg_in->struct_size : sizeof cfg);
    if (cfg.n == 0) return mfail(nullptr, NB_ERR_INVALID, "nb_multi_create: n must be >= 1");
    if (cfg.shard_count || cfg.ext_bodies || cfg.ext_stream)
        return mfail(nullptr, NB_ERR_INVALID, "nb_multi_create: shard/ext_* fields are managed by the multi handle");
    const int count = nb_device_count();
    if (count <= 0) return mfail(nullptr, NB_ERR_NO_DEVICE, "nb_multi_create: no HIP device; this engine has no CPU fallback");
    nb_multi* m = new (std::nothrow) nb_multi;
    if (!m) return mfail(nullptr, NB_ERR_NOMEM, "nb_multi_create: out of host memory");
    m->n = cfg.n; m->g = n_shards;
    m->esz = cfg.precision == NB_F64 ? 8 : 4;
    uint32_t rows = ceil_div(cfg.n, n_shards);
    rows = ceil_div(rows, (uint32_t)nb::kTile) * nb::kTile;     // 256-aligned blocks (reference tile, nbody3d.js:4)
    m->rows = rows; m->padded_n = rows * n_shards;
    for (uint32_t k = 0; k < n_shards; ++k) {
        nb_config c = cfg;
        c.struct_size = sizeof c;
        c.n = m->padded_n;
        c.shard_begin = k * rows; c.shard_count = rows;
        c.device = devices ? devices[k] : (int32_t)(k % (uint32_t)count);
        nb_sim* s = nullptr;
        int rc = nb_create(&c, &s);
        if (rc != NB_OK) { std::string e = g_create_error; nb_multi_destroy(m); return mfail(nullptr, rc, "nb_multi_create: shard " + std::to_string(k) + ": " + e); }
        m->shard.push_back(s);
    }
    // peer access between every pair of distinct devices (ignore "already enabled")
    for (uint32_t a = 0; a < n_shards; ++a)
        for (uint32_t b = 0; b < n_shards; ++b) {
            const int da = m->shard[a]->device, db = m->shard[b]->device;
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                (void)hipSetDevice(da);
                hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e != hipSuccess) (void)hipGetLastError();   // hipErrorPeerAccessAlreadyEnabled is fine
            }
        }
    m->ev_k2.resize(n_shards); m->ev_copied.resize(n_shards);
    for (uint32_t k = 0; k < n_shards; ++k) {
        if (hipSetDevice(m->shard[k]->device) != hipSuccess ||
            hipEventCreateWithFlags(&m->ev_k2[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->ev_copied[k], hipEventDisableTiming) != hipSuccess) {
            nb_multi_destroy(m);
            return mfail(nullptr, NB_ERR_HIP, "nb_multi_create: event creation failed");
        }
    }
    *out = m;
    return NB_OK;
}

void nb_multi_destroy(nb_multi* m)
{
    if (!m) return;
    for (size_t k = 0; k < m->shard.size(); ++k) {
        (void)hipSetDevice(m->shard[k]->device);
        (void)hipStreamSynchronize(m->shard[k]->stream);
    }
    for (size_t k = 0; k < m->ev_k2.size(); ++k) {
        if (k < m->shard.size()) (void)hipSetDevice(m->shard[k]->device);
        if (m->ev_k2[k]) (void)hipEventDestroy(m->ev_k2[k]);
        if (m->ev_copied[k]) (void)hipEventDestroy(m->ev_copied[k]);
    }
    for (nb_sim* s : m->shard) nb_destroy(s);
    delete m;
}

const char* nb_multi_last_error(nb_multi* m) { return m ? m->err.c_str() : g_create_error.c_str(); }
const char* nb_multi_variant_name(nb_multi* m) { return (m && !m->shard.empty()) ? m->shard[0]->variant.c_str() : ""; }

int nb_multi_sync(nb_multi* m)
{
    if (!m) return NB_ERR_INVALID;
    for (nb_sim* s : m->shard) { int rc = nb_sync(s); if (rc != NB_OK) return mfail(m, rc, s->err); }
    return NB_OK;
}

int nb_multi_upload(nb_multi* m, const void* bodies, const void* vel, const void* accel)
{
    if (!m) return NB_ERR_INVALID;
    if (!bodies || !vel) return mfail(m, NB_ERR_INVALID, "nb_multi_upload: bodies and vel are required");
    if (int rc = nb_multi_sync(m)) return rc;
    m->copied_pending = false;
    const size_t row = 4 * m->esz, real = row * m->n, padded = row * m->padded_n;
    const void *b = bodies, *v = vel, *a = accel;
    if (m->padded_n != m->n) {     // zero-mass rows at the origin, zero velocity
        m->pad_b.assign(padded, 0); memcpy(m->pad_b.data(), bodies, real); b = m->pad_b.data();
        m->pad_v.assign(padded, 0); memcpy(m->pad_v.data(), vel, real); v = m->pad_v.data();
        if (accel) { m->pad_a.assign(padded, 0); memcpy(m->pad_a.data(), accel, real); a = m->pad_a.data(); }
    }
    for (nb_sim* s : m->shard) { int rc = nb_upload(s, b, v, a); if (rc != NB_OK) return mfail(m, rc, s->err); }
    return NB_OK;
}

int nb_multi_set_params(nb_multi* m, double dt, double G)
{
    if (!m) return NB_ERR_INVALID;
    for (nb_sim* s : m->shard) { int rc = nb_set_params(s, dt, G); if (rc != NB_OK) return mfail(m, rc, s->err); }
    return NB_OK;
}

int nb_multi_step(nb_multi* m, uint32_t nsteps)
{
    if (!m) return NB_ERR_INVALID;
    const uint32_t g = m->g;
    const size_t row = 4 * m->esz, blk = row * m->rows;
    if (g == 1) { int rc = nb_step(m->shard[0], nsteps); return rc == NB_OK ? rc : mfail(m, rc, m->shard[0]->err); }
    if (!m->shard[0]->params_set || !(m->shard[0]->dt > 0.0)) {
        int rc = nb_step(m->shard[0], 0);      // reports state errors; dt <= 0 is the reference's no-op gate
        return rc == NB_OK ? rc : mfail(m, rc, m->shard[0]->err);
    }
    for (uint32_t k = 0; k < nsteps; ++k) {
        // force + integrate on every shard (asynchronous on the shard's own stream)
        for (uint32_t d = 0; d < g; ++d) {
            nb_sim* s = m->shard[d];
            NB_MHIP(m, hipSetDevice(s->device));
            if (m->copied_pending)      // nobody may still be reading the rows this shard is about to overwrite
                for (uint32_t e = 0; e < g; ++e)
                    if (e != d) NB_MHIP(m, hipStreamWaitEvent(s->stream, m->ev_copied[e], 0));
            int rc = nb_step(s, 1);
            if (rc != NB_OK) return mfail(m, rc, s->err);
            NB_MHIP(m, hipEventRecord(m->ev_k2[d], s->stream));
        }
        // all-gather by direct copies: shard e pulls the new rows of every other shard d
        for (uint32_t e = 0; e < g; ++e) {
            nb_sim* dst = m->shard[e];
            NB_MHIP(m, hipSetDevice(dst->device));
            for (uint32_t d = 0; d < g; ++d) {
                if (d == e) continue;
                nb_sim* src = m->shard[d];
                NB_MHIP(m, hipStreamWaitEvent(dst->stream, m->ev_k2[d], 0));
                NB_MHIP(m, hipMemcpyAsync((char*)dst->bodies + blk * d, (const char*)src->bodies + blk * d, blk,
                                          hipMemcpyDeviceToDevice, dst->stream));
            }
            NB_MHIP(m, hipEventRecord(m->ev_copied[e], dst->stream));
        }
        m->copied_pending = true;
    }
    return NB_OK;
}

int nb_multi_diagnostics(nb_multi* m, double out[5])
{
    if (!m || !out) return NB_ERR_INVALID;
    if (int rc = nb_multi_sync(m)) return rc;
    for (int q = 0; q < 5; ++q) out[q] = 0.0;
    for (nb_sim* s : m->shard) {
        double part[5];
        int rc = nb_diagnostics(s, part);      // zero-mass padding rows contribute exactly 0
        if (rc != NB_OK) return mfail(m, rc, s->err);
        for (int q = 0; q < 5; ++q) out[q] += part[q];
    }
    return NB_OK;
}

int nb_multi_download(nb_multi* m, void* bodies, void* vel, void* accel)
{
    if (!m) return NB_ERR_INVALID;
    if (int rc = nb_multi_sync(m)) return rc;
    const size_t row = 4 * m->esz, real = row * m->n, padded = row * m->padded_n;
    const bool pad = m->padded_n != m->n;
    void *b = bodies, *v = vel, *a = accel;
    if (pad) {
        if (bodies) { m->pad_b.assign(padded, 0); b = m->pad_b.data(); }
        if (vel) { m->pad_v.assign(padded, 0); v = m->pad_v.data(); }
        if (accel) { m->pad_a.assign(padded, 0); a = m->pad_a.data(); }
    }
    for (uint32_t k = 0; k < m->g; ++k) {
        // bodies: every shard holds the full array; take it from shard 0 only
        int rc = nb_download(m->shard[k], k == 0 ? b : nullptr, v, a);
        if (rc != NB_OK) return mfail(m, rc, m->shard[k]->err);
    }
    if (pad) {
        if (bodies) memcpy(bodies, b, real);
        if (vel) memcpy(vel, v, real);
        if (accel) memcpy(accel, a, real);
    }
    return NB_OK;
}

}  // extern "C"
