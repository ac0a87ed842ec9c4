// nb_comm.hip -- the ONE exchange step of the sharded force pass: after the
// integrate kernel every shard's new position rows are all-gathered into every shard's
// replicated bodies array.  Three native forms (no reference analogue: the reference is
// single-device, nbody3d.js:2,140-149):
//
//   nb_rccl_attach      one process per GPU: ncclCommInitRank, then per step an in-place
//                       ncclAllGather on the engine stream (or on a second stream, hidden behind
//                       the next step's own-row force work: NB_RCCL_OVERLAP)
//   nb_multi, RCCL      one process, g devices: ncclCommInitAll, per step
//                       ncclGroupStart / ncclAllGather x g / ncclGroupEnd
//   nb_multi, PEER      one process, g devices: g*(g-1) hipMemcpyAsync peer copies ordered by events
//
// librccl is NOT a link-time dependency: it is loaded on first use, and a copy that is already
// mapped into the process (PyTorch ships its own librccl.so) is reused rather than loading a
// second RCCL beside it.
#include "nb_internal.h"
#include "nb_kernels.hip.h"   // nb::kTile

#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

using nbi::fail;

namespace {

// ---- RCCL entry points, resolved once --------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    std::string path, error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    int version = 0;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out)
{
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        *static_cast<std::string*>(out) = info->dlpi_name;
        return 1;
    }
    return 0;
}

void load_rccl_once()
{
    RcclApi& a = g_rccl;
    std::vector<std::string> tries;
    if (const char* env = getenv("NB_RCCL_LIB")) tries.push_back(env);
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);      // e.g. torch/lib/librccl.so in a PyTorch process
    if (!loaded.empty()) tries.push_back(loaded);
    tries.push_back("librccl.so.1");
    tries.push_back("/opt/rocm/lib/librccl.so.1");
    tries.push_back("librccl.so");
    for (const std::string& p : tries) {
        a.lib = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (a.lib) { a.path = p; break; }
        a.error += p + ": " + (dlerror() ? dlerror() : "?") + "; ";
    }
    if (!a.lib) return;
#define NB_SYM(field, name)                                                                       \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name));                            \
    if (!a.field) { a.error = std::string("missing symbol ") + name + " in " + a.path; a.lib = nullptr; return; }
    NB_SYM(GetUniqueId, "ncclGetUniqueId") NB_SYM(CommInitRank, "ncclCommInitRank") NB_SYM(CommInitAll, "ncclCommInitAll")
    NB_SYM(CommDestroy, "ncclCommDestroy") NB_SYM(CommCount, "ncclCommCount") NB_SYM(CommUserRank, "ncclCommUserRank")
    NB_SYM(AllGather, "ncclAllGather") NB_SYM(ReduceScatter, "ncclReduceScatter") NB_SYM(GroupStart, "ncclGroupStart") NB_SYM(GroupEnd, "ncclGroupEnd")
    NB_SYM(GetErrorString, "ncclGetErrorString") NB_SYM(GetVersion, "ncclGetVersion")
#undef NB_SYM
    (void)a.GetVersion(&a.version);
}

// nullptr + message when RCCL cannot be loaded
const RcclApi* rccl_api(std::string* why)
{
    std::call_once(g_rccl_once, load_rccl_once);
    if (!g_rccl.lib) { if (why) *why = "cannot load librccl: " + g_rccl.error; return nullptr; }
    return &g_rccl;
}

static_assert(sizeof(ncclUniqueId) == NB_RCCL_ID_BYTES, "NB_RCCL_ID_BYTES must match ncclUniqueId");

}  // namespace

// per-process communicator of one handle
struct nb_rccl {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    bool overlap = false;
    hipStream_t stream = nullptr;          // overlap: the collective's own stream
    hipEvent_t ev_ready = nullptr;         // own rows written (engine stream)
    hipEvent_t ev_done = nullptr;          // all rows received (collective stream)
};

namespace nbi {

bool rccl_overlapped(const nb_sim* s) { return s->rccl && s->rccl->overlap; }

// In-place all-gather of this rank's rows: send = bodies + rank * rows, recv = bodies.
// RCCL skips the self copy for the in-place form, so the rank's own rows are only read.
int rccl_exchange_begin(nb_sim* s)
{
    nb_rccl* c = s->rccl;
    const RcclApi* api = rccl_api(nullptr);
    if (!c || !api) return fail(s, NB_ERR_COMM, "rccl_exchange_begin: no communicator");
    char* base = (char*)s->bodies[s->cur];
    const size_t row = 4 * s->esz;
    const ncclDataType_t ty = s->f64 ? ncclDouble : ncclFloat;
    hipStream_t st = s->stream;
    if (c->overlap) {
        hipError_t e = hipEventRecord(c->ev_ready, s->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_ready, 0);
        if (e != hipSuccess) return fail(s, NB_ERR_HIP, std::string("rccl_exchange_begin: ") + hipGetErrorString(e));
        st = c->stream;
    }
    const ncclResult_t r = api->AllGather(base + row * s->sb, base, (size_t)4 * s->sc, ty, c->comm, st);
    if (r != ncclSuccess) return fail(s, NB_ERR_COMM, std::string("ncclAllGather: ") + api->GetErrorString(r));
    if (c->overlap) {
        const hipError_t e = hipEventRecord(c->ev_done, c->stream);
        if (e != hipSuccess) return fail(s, NB_ERR_HIP, std::string("rccl_exchange_begin: ") + hipGetErrorString(e));
    }
    return NB_OK;
}

// Rank form of the symmetric pass: sym_A holds this rank's sums for every row of the system; in place, rank r receives the
// sum over all ranks of rows [r * rows, (r + 1) * rows) at that position of its own array (recv = send + rank * count, the
// in-place form RCCL documents for ncclReduceScatter).  On the engine stream: ordered after nb_sym_reduce, before the integrate kernel.
int rccl_reduce_scatter_A(nb_sim* s)
{
    nb_rccl* c = s->rccl;
    const RcclApi* api = rccl_api(nullptr);
    if (!c || !api) return fail(s, NB_ERR_COMM, "rccl_reduce_scatter_A: no communicator");
    char* A = (char*)s->sym_A;
    const size_t count = (size_t)4 * s->sc;
    const ncclResult_t r = api->ReduceScatter(A, A + 4 * s->esz * s->sb, count, s->f64 ? ncclDouble : ncclFloat, ncclSum, c->comm, s->stream);
    if (r != ncclSuccess) return fail(s, NB_ERR_COMM, std::string("ncclReduceScatter: ") + api->GetErrorString(r));
    return NB_OK;
}

int rccl_exchange_wait(nb_sim* s)
{
    nb_rccl* c = s->rccl;
    if (!c || !c->overlap) return NB_OK;
    const hipError_t e = hipStreamWaitEvent(s->stream, c->ev_done, 0);
    if (e != hipSuccess) return fail(s, NB_ERR_HIP, std::string("rccl_exchange_wait: ") + hipGetErrorString(e));
    return NB_OK;
}

void rccl_release(nb_sim* s)
{
    nb_rccl* c = s->rccl;
    if (!c) return;
    const RcclApi* api = rccl_api(nullptr);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (api && c->comm) (void)api->CommDestroy(c->comm);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    s->rccl = nullptr;
}

}  // namespace nbi

extern "C" {

int nb_rccl_unique_id(void* id_out)
{
    if (!id_out) return fail(nullptr, NB_ERR_INVALID, "nb_rccl_unique_id: null argument");
    std::string why;
    const RcclApi* api = rccl_api(&why);
    if (!api) return fail(nullptr, NB_ERR_COMM, "nb_rccl_unique_id: " + why);
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, NB_ERR_COMM, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return NB_OK;
}

int nb_rccl_attach(nb_sim* s, const void* id_in, int nranks, int rank, uint32_t flags)
{
    if (!s) return NB_ERR_INVALID;
    if (!id_in || nranks < 1 || rank < 0 || rank >= nranks) return fail(s, NB_ERR_INVALID, "nb_rccl_attach: bad argument");
    if (s->rccl) return fail(s, NB_ERR_STATE, "nb_rccl_attach: a communicator is already attached");
    if (s->xfn) return fail(s, NB_ERR_STATE, "nb_rccl_attach: an exchange hook is set (clear it with nb_set_exchange(s, NULL, NULL))");
    if (s->fused) return fail(s, NB_ERR_STATE, "nb_rccl_attach: a fused whole-system handle has nothing to exchange (create the shard handle with shard_count set)");
    // equal row blocks, as ncclAllGather wants them
    if ((uint64_t)s->sc * (uint32_t)nranks != s->n || s->sb != (uint32_t)rank * s->sc)
        return fail(s, NB_ERR_INVALID, "nb_rccl_attach: needs n == nranks * shard_count and shard_begin == rank * shard_count "
                                       "(pad the system with zero-mass rows)");
    std::string why;
    const RcclApi* api = rccl_api(&why);
    if (!api) return fail(s, NB_ERR_COMM, "nb_rccl_attach: " + why);
    hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) return fail(s, NB_ERR_HIP, std::string("nb_rccl_attach: hipSetDevice: ") + hipGetErrorString(e));
    nb_rccl* c = new (std::nothrow) nb_rccl;
    if (!c) return fail(s, NB_ERR_NOMEM, "nb_rccl_attach: out of host memory");
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    const ncclResult_t r = api->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return fail(s, NB_ERR_COMM, std::string("ncclCommInitRank: ") + api->GetErrorString(r)); }
    (void)api->CommCount(c->comm, &c->nranks);
    (void)api->CommUserRank(c->comm, &c->rank);
    c->overlap = (flags & NB_RCCL_OVERLAP) != 0;
    if (c->overlap) {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
            s->rccl = c;
            nbi::rccl_release(s);
            return fail(s, NB_ERR_HIP, "nb_rccl_attach: stream/event creation failed");
        }
    }
    s->rccl = c;
    return NB_OK;
}

int nb_rccl_detach(nb_sim* s)
{
    if (!s) return NB_ERR_INVALID;
    if (int rc = nb_sync(s)) return rc;
    nbi::rccl_release(s);
    return NB_OK;
}

int nb_rccl_info(nb_sim* s, int* nranks, int* rank, int* rccl_version)
{
    if (!s) return NB_ERR_INVALID;
    if (nranks) *nranks = s->rccl ? s->rccl->nranks : 0;
    if (rank) *rank = s->rccl ? s->rccl->rank : 0;
    if (rccl_version) *rccl_version = s->rccl ? g_rccl.version : 0;
    return NB_OK;
}

}  // extern "C"

/* ------------------------------------------------------------------------- *
 * nb_multi: g shard handles in one process                                   *
 * ------------------------------------------------------------------------- */
struct nb_multi {
    uint32_t n = 0, rows = 0, padded_n = 0, g = 0;
    size_t esz = 4;
    std::vector<nb_sim*> shard;
    std::vector<hipEvent_t> ev_k2, ev_copied;   // per shard: "own rows written", "all foreign rows received"
    bool copied_pending = false;
    // rank form of the symmetric force pass (f32, rows in whole super-blocks): each shard sweeps the pair lists of its own rows,
    // then the shards reduce-scatter their partial accelerations (peer copies into `stage` + a fixed-order sum, or ncclReduceScatter)
    bool sym = false;
    std::vector<hipEvent_t> ev_a, ev_rs;        // per shard: "sym_A complete", "this shard has copied what it needs of the others' sym_A"
    std::vector<void*> stage;                   // per shard: g x rows float4
    bool rs_pending = false;
    bool pull = false;                          // peer exchanges as one pull kernel per shard (g <= 16, every pair of devices peer-accessible)
    // "every shard has reached X" as ONE event: a hub stream (on shard 0's device) waits for the g per-shard events and records
    // the hub event every shard then waits for -- 2 g stream waits per dependency instead of g (g - 1) (at g = 8: 64 instead of 224
    // per step; the host thread spends ~1.7 us per call)
    hipStream_t hub = nullptr;
    hipEvent_t ev_hub[4] = {nullptr, nullptr, nullptr, nullptr};   // 0: sym_A complete, 1: sym_A consumed, 2: rows written, 3: rows gathered
    int mode = NB_MULTI_PEER;
    // NB_MULTI_PEER_OVERLAP: the gather pull of a step runs on xs[e]; shard e's own stream waits for ev_copied[e] only in front of
    // the sweeps that read the other shards' rows (the shard handle's exchange-wait hook, armed per step)
    std::vector<hipStream_t> xs;
    struct Wait { hipEvent_t ev; };
    std::vector<Wait> waits;
    std::vector<ncclComm_t> comms;               // NB_MULTI_RCCL: one per shard (ncclCommInitAll)
    std::vector<char> pad_b, pad_v, pad_a;       // host staging for the zero-mass padding rows
    std::string err;
};

namespace {

uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

int mfail(nb_multi* m, int code, const std::string& msg) { if (m) m->err = msg; else nbi::set_create_error(msg); return code; }

#define NB_MHIP(m, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return mfail((m), NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// the shard handle's exchange-wait hook in NB_MULTI_PEER_OVERLAP: the engine stream waits for this shard's gather pull
int multi_gather_wait(void* user, void* stream)
{
    const nb_multi::Wait* w = (const nb_multi::Wait*)user;
    return hipStreamWaitEvent((hipStream_t)stream, w->ev, 0) == hipSuccess ? 0 : 1;
}

void drop_overlap(nb_multi* m)
{
    for (size_t k = 0; k < m->xs.size(); ++k) {
        if (k < m->shard.size()) {
            (void)hipSetDevice(m->shard[k]->device);
            m->shard[k]->xwait = nullptr; m->shard[k]->xuser = nullptr; m->shard[k]->gather_pending = false;
        }
        if (m->xs[k]) { (void)hipStreamSynchronize(m->xs[k]); (void)hipStreamDestroy(m->xs[k]); }
    }
    m->xs.clear();
    m->waits.clear();
}

void drop_comms(nb_multi* m)
{
    const RcclApi* api = rccl_api(nullptr);
    for (size_t k = 0; k < m->comms.size(); ++k)
        if (api && m->comms[k]) { (void)hipSetDevice(m->shard[k]->device); (void)api->CommDestroy(m->comms[k]); }
    m->comms.clear();
}

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
    if (cfg.n > (1u << 30) - 1024u * n_shards) return mfail(nullptr, NB_ERR_INVALID, "nb_multi_create: n too large (the padded system must stay <= 2^30 rows)");
    if (cfg.shard_count || cfg.ext_bodies || cfg.ext_stream)
        return mfail(nullptr, NB_ERR_INVALID, "nb_multi_create: shard/ext_* fields are managed by the multi handle");
    const int count = nb_device_count();
    if (count <= 0) return mfail(nullptr, NB_ERR_NO_DEVICE, "nb_multi_create: no HIP device; this engine has no CPU fallback");
    nb_multi* m = new (std::nothrow) nb_multi;
    if (!m) return mfail(nullptr, NB_ERR_NOMEM, "nb_multi_create: out of host memory");
    m->n = cfg.n; m->g = n_shards;
    m->esz = cfg.precision == NB_F64 ? 8 : 4;
    uint32_t rows = ceil_div(cfg.n, n_shards);
    // the rank form of the symmetric pass wants rows in whole super-blocks of 1,024 (f32, >= 2 shards, systems worth it)
    const bool want_sym = n_shards >= 2 && rows >= 2048 && !(cfg.flags & NB_FLAG_NO_SYM) && cfg.force_variant == 0;
    const uint32_t align = want_sym ? (cfg.precision == NB_F64 ? 512u : 1024u) : (uint32_t)nb::kTile;      // whole super-blocks
    rows = ceil_div(rows, align) * align;                       // 256-aligned blocks (reference tile, nbody3d.js:4) at least
    m->rows = rows; m->padded_n = rows * n_shards;
    for (uint32_t k = 0; k < n_shards; ++k) {
        nb_config c = cfg;
        c.struct_size = sizeof c;
        c.n = m->padded_n;
        c.shard_begin = k * rows; c.shard_count = rows;
        c.device = devices ? devices[k] : (int32_t)(k % (uint32_t)count);
        if (n_shards == 1) c.flags |= NB_FLAG_NO_FUSE;   // keep the shard code path (and its exchange) even for g = 1
        if (want_sym) c.flags |= NB_FLAG_SYM_SHARD;
        nb_sim* s = nullptr;
        int rc = nb_create(&c, &s);
        if (rc != NB_OK) { std::string e = nbi::create_error(); nb_multi_destroy(m); return mfail(nullptr, rc, "nb_multi_create: shard " + std::to_string(k) + ": " + e); }
        m->shard.push_back(s);
    }
    // Peer access between every pair of distinct devices.  The pull kernels (nb_peer_sum / nb_peer_gather) dereference the other
    // shards' arrays directly, so they are only used when EVERY ordered pair of devices is accessible AND was enabled (success or
    // "already enabled"); any other outcome keeps the hipMemcpyAsync / staging path, which needs no mapping.
    m->pull = n_shards <= 16;
    for (uint32_t a = 0; a < n_shards; ++a)
        for (uint32_t b = 0; b < n_shards; ++b) {
            const int da = m->shard[a]->device, db = m->shard[b]->device;
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) { (void)hipGetLastError(); m->pull = false; continue; }
            hipError_t e = hipSetDevice(da);
            if (e == hipSuccess) e = hipDeviceEnablePeerAccess(db, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            if (e != hipSuccess) { (void)hipGetLastError(); m->pull = false; }
        }
    m->ev_k2.resize(n_shards); m->ev_copied.resize(n_shards);
    m->sym = want_sym;
    for (nb_sim* s : m->shard) if (!s->sym_rank) m->sym = false;
    m->ev_a.assign(n_shards, nullptr); m->ev_rs.assign(n_shards, nullptr); m->stage.assign(n_shards, nullptr);
    {
        bool ok = hipSetDevice(m->shard[0]->device) == hipSuccess && hipStreamCreateWithFlags(&m->hub, hipStreamNonBlocking) == hipSuccess;
        for (auto& e : m->ev_hub) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (!ok) { nb_multi_destroy(m); return mfail(nullptr, NB_ERR_HIP, "nb_multi_create: hub stream / events"); }
    }
    for (uint32_t k = 0; k < n_shards; ++k) {
        if (hipSetDevice(m->shard[k]->device) != hipSuccess ||
            hipEventCreateWithFlags(&m->ev_k2[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->ev_copied[k], hipEventDisableTiming) != hipSuccess ||
            (m->sym && (hipEventCreateWithFlags(&m->ev_a[k], hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&m->ev_rs[k], hipEventDisableTiming) != hipSuccess ||
                        hipMalloc(&m->stage[k], 4 * m->esz * rows * n_shards) != hipSuccess))) {
            nb_multi_destroy(m);
            return mfail(nullptr, NB_ERR_HIP, "nb_multi_create: event / staging creation failed");
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
    drop_comms(m);
    drop_overlap(m);
    for (size_t k = 0; k < m->ev_k2.size(); ++k) {
        if (k < m->shard.size()) (void)hipSetDevice(m->shard[k]->device);
        if (m->ev_k2[k]) (void)hipEventDestroy(m->ev_k2[k]);
        if (m->ev_copied[k]) (void)hipEventDestroy(m->ev_copied[k]);
        if (k < m->ev_a.size() && m->ev_a[k]) (void)hipEventDestroy(m->ev_a[k]);
        if (k < m->ev_rs.size() && m->ev_rs[k]) (void)hipEventDestroy(m->ev_rs[k]);
        if (k < m->stage.size() && m->stage[k]) (void)hipFree(m->stage[k]);
    }
    if (!m->shard.empty()) (void)hipSetDevice(m->shard[0]->device);
    for (auto& e : m->ev_hub) if (e) (void)hipEventDestroy(e);
    if (m->hub) { (void)hipStreamSynchronize(m->hub); (void)hipStreamDestroy(m->hub); }
    for (nb_sim* s : m->shard) nb_destroy(s);
    delete m;
}

const char* nb_multi_last_error(nb_multi* m) { return m ? m->err.c_str() : nbi::create_error().c_str(); }
const char* nb_multi_variant_name(nb_multi* m) { return (m && !m->shard.empty()) ? m->shard[0]->variant.c_str() : ""; }

int nb_multi_sync(nb_multi* m)
{
    if (!m) return NB_ERR_INVALID;
    for (nb_sim* s : m->shard) { int rc = nb_sync(s); if (rc != NB_OK) return mfail(m, rc, s->err); }
    return NB_OK;
}

int nb_multi_set_collective(nb_multi* m, int mode)
{
    if (!m) return NB_ERR_INVALID;
    if (mode != NB_MULTI_PEER && mode != NB_MULTI_RCCL && mode != NB_MULTI_PEER_OVERLAP) return mfail(m, NB_ERR_INVALID, "nb_multi_set_collective: unknown mode");
    if (int rc = nb_multi_sync(m)) return rc;
    if (mode == m->mode) return NB_OK;
    if (mode == NB_MULTI_PEER) { drop_comms(m); drop_overlap(m); m->mode = mode; m->copied_pending = false; m->rs_pending = false; return NB_OK; }
    if (mode == NB_MULTI_PEER_OVERLAP) {
        if (!m->sym || !m->pull) return mfail(m, NB_ERR_STATE, "nb_multi_set_collective: NB_MULTI_PEER_OVERLAP needs the rank form of the symmetric pass and peer-accessible shards");
        drop_comms(m);
        m->xs.assign(m->g, nullptr);
        m->waits.assign(m->g, nb_multi::Wait{nullptr});
        for (uint32_t k = 0; k < m->g; ++k) {
            if (hipSetDevice(m->shard[k]->device) != hipSuccess || hipStreamCreateWithFlags(&m->xs[k], hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError(); drop_overlap(m);
                return mfail(m, NB_ERR_HIP, "nb_multi_set_collective: cannot create the gather streams");
            }
            m->waits[k].ev = m->ev_copied[k];
        }
        m->mode = mode; m->copied_pending = false; m->rs_pending = false;
        return NB_OK;
    }
    drop_overlap(m);
    // one communicator per shard, one shard per device (RCCL refuses two ranks on one GPU)
    std::vector<int> devs;
    for (nb_sim* s : m->shard) {
        for (int d : devs)
            if (d == s->device) return mfail(m, NB_ERR_INVALID, "nb_multi_set_collective: NB_MULTI_RCCL needs every shard on its own device");
        devs.push_back(s->device);
    }
    std::string why;
    const RcclApi* api = rccl_api(&why);
    if (!api) return mfail(m, NB_ERR_COMM, "nb_multi_set_collective: " + why);
    m->comms.assign(m->g, nullptr);
    const ncclResult_t r = api->CommInitAll(m->comms.data(), (int)m->g, devs.data());
    if (r != ncclSuccess) { m->comms.clear(); return mfail(m, NB_ERR_COMM, std::string("ncclCommInitAll: ") + api->GetErrorString(r)); }
    m->mode = mode;
    m->copied_pending = false;
    m->rs_pending = false;
    return NB_OK;
}

int nb_multi_collective_info(nb_multi* m, int* mode, int* nranks, int* rccl_version)
{
    if (!m) return NB_ERR_INVALID;
    if (mode) *mode = m->mode;
    int nr = 0;
    if (m->mode == NB_MULTI_RCCL && !m->comms.empty()) {
        const RcclApi* api = rccl_api(nullptr);
        if (api) (void)api->CommCount(m->comms[0], &nr);
    }
    if (nranks) *nranks = nr;
    if (rccl_version) *rccl_version = m->mode == NB_MULTI_RCCL ? g_rccl.version : 0;
    return NB_OK;
}

int nb_multi_upload(nb_multi* m, const void* bodies, const void* vel, const void* accel)
{
    if (!m) return NB_ERR_INVALID;
    if (!bodies || !vel) return mfail(m, NB_ERR_INVALID, "nb_multi_upload: bodies and vel are required");
    if (int rc = nb_multi_sync(m)) return rc;
    m->copied_pending = false;
    m->rs_pending = false;
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
    if (!m->shard[0]->params_set || !(m->shard[0]->dt > 0.0) || (g == 1 && m->mode == NB_MULTI_PEER)) {
        // state errors, the reference's dt <= 0 no-op gate, and the trivial one-shard case
        int rc = nb_step(m->shard[0], (m->shard[0]->params_set && m->shard[0]->dt > 0.0) ? nsteps : 0);
        return rc == NB_OK ? rc : mfail(m, rc, m->shard[0]->err);
    }
    const RcclApi* api = m->mode == NB_MULTI_RCCL ? rccl_api(nullptr) : nullptr;
    if (m->mode == NB_MULTI_RCCL && !api) return mfail(m, NB_ERR_COMM, "nb_multi_step: RCCL is not loaded");
    if (m->sym && !m->shard[0]->uploaded) return mfail(m, NB_ERR_STATE, "nb_multi_step: nb_multi_upload has not been called");
    // hub event `which` := all of `evs` (already recorded on the shards' streams)
    auto join = [&](const std::vector<hipEvent_t>& evs, int which) -> hipError_t {
        hipError_t e = hipSetDevice(m->shard[0]->device);
        for (uint32_t d = 0; d < g && e == hipSuccess; ++d) e = hipStreamWaitEvent(m->hub, evs[d], 0);
        if (e == hipSuccess) e = hipEventRecord(m->ev_hub[which], m->hub);
        return e;
    };
    for (uint32_t k = 0; k < nsteps; ++k) {
        // force + integrate on every shard (asynchronous on the shard's own stream)
        for (uint32_t d = 0; d < g && !m->sym; ++d) {
            nb_sim* s = m->shard[d];
            NB_MHIP(m, hipSetDevice(s->device));
            if (m->copied_pending)      // nobody may still be reading the rows this shard is about to overwrite
                NB_MHIP(m, hipStreamWaitEvent(s->stream, m->ev_hub[3], 0));
            int rc = nb_step(s, 1);
            if (rc != NB_OK) return mfail(m, rc, s->err);
            if (m->mode != NB_MULTI_RCCL) NB_MHIP(m, hipEventRecord(m->ev_k2[d], s->stream));
        }
        if (m->sym) {
            // rank form of the symmetric pass.  (a) every shard: force pass over the pair lists of its own rows, then its sums
            // for every row of the system (sym_A)
            for (uint32_t d = 0; d < g; ++d) {
                nb_sim* s = m->shard[d];
                NB_MHIP(m, hipSetDevice(s->device));
                if (m->rs_pending)       // the other shards may still be reading this shard's sym_A (previous step)
                    NB_MHIP(m, hipStreamWaitEvent(s->stream, m->ev_hub[1], 0));
                // (the positions this pass reads were completed on this stream by the previous step's gather -- or, overlapped, the
                // gather is still running on the shard's second stream: the sweeps over the shard's own rows go first and the
                // stream waits for the pull in front of the rest)
                int rc = nbi::sym_rank_phase_a(s, nullptr, s->gather_pending);
                if (rc != NB_OK) return mfail(m, rc, s->err);
                if (m->mode != NB_MULTI_RCCL) NB_MHIP(m, hipEventRecord(m->ev_a[d], s->stream));
            }
            // (b) reduce-scatter: shard e ends up with the sum over all shards of the rows it owns
            if (m->mode == NB_MULTI_RCCL) {
                ncclResult_t r = api->GroupStart();
                for (uint32_t d = 0; d < g && r == ncclSuccess; ++d) {
                    nb_sim* s = m->shard[d];
                    char* A = (char*)s->sym_A;
                    r = api->ReduceScatter(A, A + blk * d, (size_t)4 * m->rows, m->esz == 8 ? ncclDouble : ncclFloat, ncclSum, m->comms[d], s->stream);
                }
                const ncclResult_t r2 = api->GroupEnd();
                if (r == ncclSuccess) r = r2;
                if (r != ncclSuccess) return mfail(m, NB_ERR_COMM, std::string("nb_multi_step: ncclReduceScatter: ") + api->GetErrorString(r));
            } else {
                NB_MHIP(m, join(m->ev_a, 0));
                for (uint32_t e = 0; e < g; ++e) {
                    nb_sim* dst = m->shard[e];
                    NB_MHIP(m, hipSetDevice(dst->device));
                    NB_MHIP(m, hipStreamWaitEvent(dst->stream, m->ev_hub[0], 0));        // every shard's sym_A is complete
                    if (m->pull) {
                        // ONE pull kernel: reads the rows this shard owns out of every shard's sym_A, sums in shard order
                        nb::PeerPtrs pp{};
                        for (uint32_t d = 0; d < g; ++d) pp.p[d] = m->shard[d]->sym_A;
                        void* out = (char*)dst->sym_A + blk * e;   // in place: the other shards read THEIR row blocks of this array, never this one
                        uint32_t rows = m->rows, shards = g, me = e;
                        void* args[] = {&pp, &out, &rows, &shards, &me};
                        const void* fn = m->esz == 8 ? (const void*)&nb::nb_peer_sum<double> : (const void*)&nb::nb_peer_sum<float>;
                        NB_MHIP(m, hipLaunchKernel(fn, dim3((rows + nb::kBlock - 1) / nb::kBlock), dim3(nb::kBlock), args, 0, dst->stream));
                        NB_MHIP(m, hipEventRecord(m->ev_rs[e], dst->stream));
                        continue;
                    }
                    for (uint32_t d = 0; d < g; ++d) {
                        nb_sim* src = m->shard[d];
                        NB_MHIP(m, hipMemcpyAsync((char*)m->stage[e] + blk * d, (const char*)src->sym_A + blk * e, blk, hipMemcpyDeviceToDevice, dst->stream));
                    }
                    NB_MHIP(m, hipEventRecord(m->ev_rs[e], dst->stream));
                    const void* st = m->stage[e];
                    void* out = (char*)dst->sym_A + blk * e;
                    uint32_t rows = m->rows, shards = g;
                    void* args[] = {&st, &out, &rows, &shards};
                    const void* fn = m->esz == 8 ? (const void*)&nb::nb_sym_sum_shards<double> : (const void*)&nb::nb_sym_sum_shards<float>;
                    NB_MHIP(m, hipLaunchKernel(fn, dim3((rows + nb::kBlock - 1) / nb::kBlock), dim3(nb::kBlock), args, 0, dst->stream));
                }
                NB_MHIP(m, join(m->ev_rs, 1));
                m->rs_pending = true;
            }
            // (c) integrate every shard's own rows
            for (uint32_t d = 0; d < g; ++d) {
                nb_sim* s = m->shard[d];
                NB_MHIP(m, hipSetDevice(s->device));
                if (m->copied_pending)      // nobody may still be reading the rows this shard is about to overwrite
                    NB_MHIP(m, hipStreamWaitEvent(s->stream, m->ev_hub[3], 0));
                int rc = nbi::sym_rank_phase_b(s);
                if (rc != NB_OK) return mfail(m, rc, s->err);
                if (m->mode != NB_MULTI_RCCL) NB_MHIP(m, hipEventRecord(m->ev_k2[d], s->stream));
            }
        }
        if (m->mode == NB_MULTI_RCCL) {
            // ncclGroupStart / per-device in-place ncclAllGather / ncclGroupEnd.
            // The collective orders the shards against each other: a rank's call completes on
            // its stream once its rows have been delivered and every other block has arrived.
            ncclResult_t r = api->GroupStart();
            for (uint32_t d = 0; d < g && r == ncclSuccess; ++d) {
                nb_sim* s = m->shard[d];
                char* base = (char*)s->bodies[s->cur];
                r = api->AllGather(base + blk * d, base, (size_t)4 * m->rows, m->esz == 8 ? ncclDouble : ncclFloat,
                                   m->comms[d], s->stream);
            }
            const ncclResult_t r2 = api->GroupEnd();
            if (r == ncclSuccess) r = r2;
            if (r != ncclSuccess) return mfail(m, NB_ERR_COMM, std::string("nb_multi_step: ncclAllGather: ") + api->GetErrorString(r));
            for (nb_sim* s : m->shard) s->gm_ok = false;     // the other shards' rows of the (x, y, z, G*m) copy are stale now
            continue;
        }
        // all-gather by direct copies: shard e pulls the new rows of every other shard d
        NB_MHIP(m, join(m->ev_k2, 2));
        for (uint32_t e = 0; e < g; ++e) {
            nb_sim* dst = m->shard[e];
            NB_MHIP(m, hipSetDevice(dst->device));
            const bool over = m->mode == NB_MULTI_PEER_OVERLAP && m->pull && (float)dst->G == 1.0f;
            if (!over) NB_MHIP(m, hipStreamWaitEvent(dst->stream, m->ev_hub[2], 0));        // every shard has written its new rows
            if (m->pull) {               // one pull kernel instead of g - 1 copies
                nb::PeerPtrs pp{};
                for (uint32_t d = 0; d < g; ++d) pp.p[d] = m->shard[d]->bodies[m->shard[d]->cur];
                void* out = dst->bodies[dst->cur];
                uint32_t rows = m->rows, shards = g, me = e;
                void* args[] = {&pp, &out, &rows, &shards, &me};
                const void* fn = m->esz == 8 ? (const void*)&nb::nb_peer_gather<double> : (const void*)&nb::nb_peer_gather<float>;
                // overlapped: on the shard's second stream (which waited for "every shard has written its rows" instead of the
                // shard's own stream); the shard handle is told a gather is pending and waits for ev_copied[e] where it must.
                // With G != 1 the (x, y, z, G m) copy is rebuilt from the gathered rows first thing next step: nothing to overlap
                hipStream_t on = over ? m->xs[e] : dst->stream;
                if (over) NB_MHIP(m, hipStreamWaitEvent(on, m->ev_hub[2], 0));
                NB_MHIP(m, hipLaunchKernel(fn, dim3((rows * shards + nb::kBlock - 1) / nb::kBlock), dim3(nb::kBlock), args, 0, on));
                NB_MHIP(m, hipEventRecord(m->ev_copied[e], on));
                if (over) { dst->xwait = multi_gather_wait; dst->xuser = &m->waits[e]; dst->gather_pending = true; }
                continue;
            }
            for (uint32_t d = 0; d < g; ++d) {
                if (d == e) continue;
                nb_sim* src = m->shard[d];
                NB_MHIP(m, hipMemcpyAsync((char*)dst->bodies[dst->cur] + blk * d, (const char*)src->bodies[src->cur] + blk * d, blk,
                                          hipMemcpyDeviceToDevice, dst->stream));
            }
            NB_MHIP(m, hipEventRecord(m->ev_copied[e], dst->stream));
        }
        NB_MHIP(m, join(m->ev_copied, 3));
        m->copied_pending = true;
        for (nb_sim* s : m->shard) s->gm_ok = false;         // as above
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
