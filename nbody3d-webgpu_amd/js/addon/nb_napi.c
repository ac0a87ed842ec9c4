/*
 * nb_napi.c -- raw N-API binding of include/nbody3d_hip.h for Node.js.
 *
 * The reference's host side is browser JavaScript talking to WebGPU
 * (/root/reference nbody3d.js:136-521).  This addon is the thin FFI that lets
 * the same JavaScript-side calls reach the HIP engine: typed arrays are passed
 * by pointer (napi_get_typedarray_info, zero copy on the JS side), every C
 * status code becomes a thrown Error carrying nb_last_error().
 *
 * Built with plain gcc against /usr/include/node (no node-gyp, no network):
 *   gcc -O2 -fPIC -shared -I/usr/include/node -o nb_napi.node nb_napi.c -ldl
 * The engine library is dlopen()ed at load(path) time, so the addon itself
 * loads on a machine without ROCm and reports that as an ordinary error.
 */
#define NAPI_VERSION 6
#include <node_api.h>

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/nbody3d_hip.h"
#include "../../../include/nbody3d_hip_plan.h"

/* ---- engine entry points, resolved at load() ---------------------------- */
static void *g_lib;
static uint32_t (*p_abi_version)(void);
static int (*p_device_count)(void);
static int (*p_create)(const nb_config *, nb_sim **);
static void (*p_destroy)(nb_sim *);
static int (*p_upload)(nb_sim *, const void *, const void *, const void *);
static int (*p_set_params)(nb_sim *, double, double);
static int (*p_step)(nb_sim *, uint32_t);
static int (*p_download)(nb_sim *, void *, void *, void *);
static int (*p_sync)(nb_sim *);
static const char *(*p_last_error)(nb_sim *);
static int (*p_enable_timing)(nb_sim *, int);
static int (*p_kernel_times)(nb_sim *, double *, double *, uint32_t *);
static const char *(*p_variant_name)(nb_sim *);
static int (*p_diagnostics)(nb_sim *, double *);
static int (*p_multi_create)(const nb_config *, uint32_t, const int32_t *, nb_multi **);
static void (*p_multi_destroy)(nb_multi *);
static int (*p_multi_upload)(nb_multi *, const void *, const void *, const void *);
static int (*p_multi_set_params)(nb_multi *, double, double);
static int (*p_multi_step)(nb_multi *, uint32_t);
static int (*p_multi_download)(nb_multi *, void *, void *, void *);
static int (*p_multi_sync)(nb_multi *);
static const char *(*p_multi_last_error)(nb_multi *);
static const char *(*p_multi_variant_name)(nb_multi *);
static int (*p_multi_diagnostics)(nb_multi *, double *);
static int (*p_multi_set_collective)(nb_multi *, int);
static int (*p_multi_collective_info)(nb_multi *, int *, int *, int *);
static int (*p_step_times)(nb_sim *, double *, double *, double *, uint32_t *);
static int (*p_step_times2)(nb_sim *, nb_step_timing *);
static int (*p_frame_request)(nb_sim *);
static int (*p_frame_acquire)(nb_sim *, int, const float **, const float **, uint64_t *);
static int (*p_plan_query)(const nb_config *, int, double, nb_plan_info *, uint32_t *, uint32_t);

/* one JS handle = a single-device nb_sim or a single-process multi-device nb_multi */
typedef struct { nb_sim *sim; nb_multi *multi; uint32_t n; int f64; } handle_t;

#define CHECK_NAPI(env, call)                                                        \
    do {                                                                             \
        if ((call) != napi_ok) {                                                     \
            napi_throw_error((env), NULL, "N-API call failed: " #call);              \
            return NULL;                                                             \
        }                                                                            \
    } while (0)

static napi_value throw_msg(napi_env env, int code, const char *msg, const char *where)
{
    char buf[768];
    snprintf(buf, sizeof buf, "%s: status %d: %s", where, code, msg ? msg : "");
    char codes[16];
    snprintf(codes, sizeof codes, "NB_%d", code);
    napi_throw_error(env, codes, buf);
    return NULL;
}

static napi_value throw_nb(napi_env env, int code, nb_sim *s, const char *where)
{
    char buf[768];
    const char *msg = p_last_error ? p_last_error(s) : "engine not loaded";
    snprintf(buf, sizeof buf, "%s: status %d: %s", where, code, msg ? msg : "");
    char codes[16];
    snprintf(codes, sizeof codes, "NB_%d", code);
    napi_throw_error(env, codes, buf);
    return NULL;
}

static int need_lib(napi_env env)
{
    if (!g_lib) { napi_throw_error(env, "NB_NOLIB", "libnbody3d_hip.so is not loaded; call load(path) first"); return 0; }
    return 1;
}

static napi_value undefined(napi_env env) { napi_value u; napi_get_undefined(env, &u); return u; }

/* load(path) -> abi version */
static napi_value js_load(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    char path[4096]; size_t len = 0;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof path, &len) != napi_ok) {
        napi_throw_type_error(env, NULL, "load(path): path string required"); return NULL;
    }
    if (!g_lib) {
        void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            char buf[4600]; snprintf(buf, sizeof buf, "cannot load HIP engine %s: %s", path, dlerror());
            napi_throw_error(env, "NB_NOLIB", buf); return NULL;
        }
#define SYM(var, name)                                                                            \
    do {                                                                                          \
        *(void **)(&var) = dlsym(h, name);                                                        \
        if (!var) { dlclose(h); napi_throw_error(env, "NB_NOLIB", "missing symbol " name); return NULL; } \
    } while (0)
        SYM(p_abi_version, "nb_abi_version"); SYM(p_device_count, "nb_device_count");
        SYM(p_create, "nb_create"); SYM(p_destroy, "nb_destroy"); SYM(p_upload, "nb_upload");
        SYM(p_set_params, "nb_set_params"); SYM(p_step, "nb_step"); SYM(p_download, "nb_download");
        SYM(p_sync, "nb_sync"); SYM(p_last_error, "nb_last_error"); SYM(p_enable_timing, "nb_enable_timing");
        SYM(p_kernel_times, "nb_kernel_times"); SYM(p_variant_name, "nb_variant_name");
        SYM(p_diagnostics, "nb_diagnostics");
        SYM(p_multi_create, "nb_multi_create"); SYM(p_multi_destroy, "nb_multi_destroy");
        SYM(p_multi_upload, "nb_multi_upload"); SYM(p_multi_set_params, "nb_multi_set_params");
        SYM(p_multi_step, "nb_multi_step"); SYM(p_multi_download, "nb_multi_download");
        SYM(p_multi_sync, "nb_multi_sync"); SYM(p_multi_last_error, "nb_multi_last_error");
        SYM(p_multi_variant_name, "nb_multi_variant_name"); SYM(p_multi_diagnostics, "nb_multi_diagnostics");
        SYM(p_multi_set_collective, "nb_multi_set_collective"); SYM(p_multi_collective_info, "nb_multi_collective_info");
        SYM(p_step_times, "nb_step_times"); SYM(p_step_times2, "nb_step_times2"); SYM(p_frame_request, "nb_frame_request"); SYM(p_frame_acquire, "nb_frame_acquire");
        SYM(p_plan_query, "nb_plan_query");
#undef SYM
        g_lib = h;
    }
    if (p_abi_version() != NB_ABI_VERSION) { napi_throw_error(env, "NB_ABI", "ABI version mismatch"); return NULL; }
    napi_value v; CHECK_NAPI(env, napi_create_uint32(env, p_abi_version(), &v)); return v;
}

static napi_value js_device_count(napi_env env, napi_callback_info info)
{
    (void)info;
    if (!need_lib(env)) return NULL;
    napi_value v; CHECK_NAPI(env, napi_create_int32(env, p_device_count(), &v)); return v;
}

static void finalize_handle(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    handle_t *h = (handle_t *)data;
    if (h) {
        if (h->sim && p_destroy) p_destroy(h->sim);
        if (h->multi && p_multi_destroy) p_multi_destroy(h->multi);
        free(h);
    }
}

static int get_u32_prop(napi_env env, napi_value obj, const char *name, uint32_t *out)
{
    bool has = false; napi_value v;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    napi_valuetype t; napi_typeof(env, v, &t);
    if (t != napi_number) return 0;
    double d; napi_get_value_double(env, v, &d);
    *out = (uint32_t)d; return 1;
}

static int get_f64_prop(napi_env env, napi_value obj, const char *name, double *out)
{
    bool has = false; napi_value v;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    napi_valuetype t; napi_typeof(env, v, &t);
    if (t != napi_number) return 0;
    napi_get_value_double(env, v, out); return 1;
}

/* create({n, f64, eps2, device, shardBegin, shardCount, variant, jsplit, flags, shards, collective}) -> external */
/* options object -> nb_config (create and planQuery read the same keys) */
static void read_config(napi_env env, napi_value opts, nb_config *cfg)
{
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg; cfg->device = -1;
    uint32_t u; double d;
    if (get_u32_prop(env, opts, "n", &u)) cfg->n = u;
    if (get_u32_prop(env, opts, "f64", &u)) cfg->precision = u ? NB_F64 : NB_F32;
    if (get_f64_prop(env, opts, "eps2", &d)) cfg->eps2 = d;
    if (get_f64_prop(env, opts, "device", &d)) cfg->device = (int32_t)d;
    if (get_u32_prop(env, opts, "shardBegin", &u)) cfg->shard_begin = u;
    if (get_u32_prop(env, opts, "shardCount", &u)) cfg->shard_count = u;
    if (get_u32_prop(env, opts, "variant", &u)) cfg->force_variant = u;
    if (get_u32_prop(env, opts, "jsplit", &u)) cfg->jsplit = u;
    if (get_u32_prop(env, opts, "tile", &u)) cfg->tile = u;
    if (get_u32_prop(env, opts, "flags", &u)) cfg->flags = u;
    if (get_u32_prop(env, opts, "layerBudgetMiB", &u)) cfg->layer_budget_mib = u;
}

/* planQuery(options) -> {variant, kind, ipl, ls, x, jsplit, jPerSplit, sym, symRows, symLayers, symUnitsPerSweep, symSpillRows, layerBytes}: nb_plan_query -- the launch
 * plan nb_create would build, from host arithmetic alone (options.nCU + options.clockHz given: no GPU needed) */
static napi_value js_plan_query(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1) { napi_throw_type_error(env, NULL, "planQuery(options) requires an object"); return NULL; }
    nb_config cfg; read_config(env, argv[0], &cfg);
    uint32_t n_cu = 0; double clock = 0.0;
    get_u32_prop(env, argv[0], "nCU", &n_cu);
    get_f64_prop(env, argv[0], "clockHz", &clock);
    nb_plan_info pi; memset(&pi, 0, sizeof pi); pi.struct_size = sizeof pi;
    int rc = p_plan_query(&cfg, (int)n_cu, clock, &pi, NULL, 0);
    if (rc != NB_OK) return throw_nb(env, rc, NULL, "nb_plan_query");
    napi_value o, v;
    CHECK_NAPI(env, napi_create_object(env, &o));
    CHECK_NAPI(env, napi_create_string_utf8(env, pi.variant, NAPI_AUTO_LENGTH, &v)); napi_set_named_property(env, o, "variant", v);
#define PUT_U32(name, val) do { napi_create_uint32(env, (val), &v); napi_set_named_property(env, o, name, v); } while (0)
    PUT_U32("kind", pi.kind); PUT_U32("ipl", pi.ipl); PUT_U32("ls", pi.ls); PUT_U32("x", pi.x);
    PUT_U32("jsplit", pi.jsplit); PUT_U32("jPerSplit", pi.j_per_split);
    PUT_U32("sym", pi.sym); PUT_U32("symRows", pi.sym_np); PUT_U32("symLayers", pi.sym_layers);
    PUT_U32("symUnitsPerSweep", pi.sym_ups); PUT_U32("symSpillRows", pi.sym_spill_rows);
#undef PUT_U32
    napi_create_double(env, 3.0 * (cfg.precision == NB_F64 ? 8.0 : 4.0) * (double)pi.sym_np * (double)pi.sym_layers, &v);
    napi_set_named_property(env, o, "layerBytes", v);
    return o;
}

static napi_value js_create(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1) { napi_throw_type_error(env, NULL, "create(options) requires an object"); return NULL; }
    nb_config cfg; read_config(env, argv[0], &cfg);
    uint32_t shards = 0, collective = 0;
    get_u32_prop(env, argv[0], "shards", &shards);
    get_u32_prop(env, argv[0], "collective", &collective);   /* 0 peer copies, 1 RCCL (nb_multi_collective) */
    nb_sim *sim = NULL;
    nb_multi *multi = NULL;
    if (shards > 1 || (shards == 1 && collective)) {   /* single-process multi-device: shards round-robin over the visible GPUs */
        cfg.device = -1; cfg.shard_begin = cfg.shard_count = 0;
        int rc = p_multi_create(&cfg, shards, NULL, &multi);
        if (rc != NB_OK) return throw_msg(env, rc, p_multi_last_error(NULL), "nb_multi_create");
        if (collective) {
            rc = p_multi_set_collective(multi, (int)collective);
            if (rc != NB_OK) {
                char msg[512]; snprintf(msg, sizeof msg, "%s", p_multi_last_error(multi));
                p_multi_destroy(multi);
                return throw_msg(env, rc, msg, "nb_multi_set_collective");
            }
        }
    } else {
        int rc = p_create(&cfg, &sim);
        if (rc != NB_OK) return throw_nb(env, rc, NULL, "nb_create");
    }
    handle_t *h = (handle_t *)calloc(1, sizeof *h);
    if (!h) { if (sim) p_destroy(sim); if (multi) p_multi_destroy(multi); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    h->sim = sim; h->multi = multi; h->n = cfg.n; h->f64 = cfg.precision == NB_F64;
    napi_value ext;
    if (napi_create_external(env, h, finalize_handle, NULL, &ext) != napi_ok) {
        finalize_handle(env, h, NULL); napi_throw_error(env, NULL, "napi_create_external failed"); return NULL;
    }
    return ext;
}

static handle_t *get_handle(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || (!((handle_t *)p)->sim && !((handle_t *)p)->multi)) {
        napi_throw_error(env, "NB_1", "invalid or destroyed simulation handle"); return NULL;
    }
    return (handle_t *)p;
}

/* typed array -> pointer; NULL allowed when nullable; checks element type + length */
static int get_array(napi_env env, napi_value v, const handle_t *h, int nullable, void **out, const char *name)
{
    napi_valuetype t; napi_typeof(env, v, &t);
    *out = NULL;
    if (t == napi_null || t == napi_undefined) {
        if (nullable) return 1;
        napi_throw_type_error(env, NULL, name); return 0;
    }
    bool is_ta = false; napi_is_typedarray(env, v, &is_ta);
    if (!is_ta) { napi_throw_type_error(env, NULL, name); return 0; }
    napi_typedarray_type tt; size_t len; void *data; napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, v, &tt, &len, &data, &ab, &off) != napi_ok) { napi_throw_type_error(env, NULL, name); return 0; }
    if ((h->f64 && tt != napi_float64_array) || (!h->f64 && tt != napi_float32_array)) {
        napi_throw_type_error(env, NULL, h->f64 ? "expected Float64Array (f64 simulation)" : "expected Float32Array");
        return 0;
    }
    if (len != (size_t)4 * h->n) {
        char buf[160]; snprintf(buf, sizeof buf, "%s: expected %zu elements (4*n), got %zu", name, (size_t)4 * h->n, len);
        napi_throw_range_error(env, NULL, buf); return 0;
    }
    *out = data; return 1;
}

static napi_value js_upload(napi_env env, napi_callback_info info)
{
    size_t argc = 4; napi_value argv[4];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 3) { napi_throw_type_error(env, NULL, "upload(handle, bodies, vel[, accel])"); return NULL; }
    handle_t *h = get_handle(env, argv[0]); if (!h) return NULL;
    void *b, *v, *a = NULL;
    if (!get_array(env, argv[1], h, 0, &b, "bodies must be a typed array of 4*n elements")) return NULL;
    if (!get_array(env, argv[2], h, 0, &v, "vel must be a typed array of 4*n elements")) return NULL;
    if (argc >= 4 && !get_array(env, argv[3], h, 1, &a, "accel must be a typed array of 4*n elements or null")) return NULL;
    if (h->multi) {
        int rc = p_multi_upload(h->multi, b, v, a);
        if (rc != NB_OK) return throw_msg(env, rc, p_multi_last_error(h->multi), "nb_multi_upload");
        return undefined(env);
    }
    int rc = p_upload(h->sim, b, v, a);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_upload");
    return undefined(env);
}

static napi_value js_set_params(napi_env env, napi_callback_info info)
{
    size_t argc = 3; napi_value argv[3];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 3) { napi_throw_type_error(env, NULL, "setParams(handle, dt, G)"); return NULL; }
    handle_t *h = get_handle(env, argv[0]); if (!h) return NULL;
    double dt, G;
    if (napi_get_value_double(env, argv[1], &dt) != napi_ok || napi_get_value_double(env, argv[2], &G) != napi_ok) {
        napi_throw_type_error(env, NULL, "dt and G must be numbers"); return NULL;
    }
    if (h->multi) {
        int rc = p_multi_set_params(h->multi, dt, G);
        if (rc != NB_OK) return throw_msg(env, rc, p_multi_last_error(h->multi), "nb_multi_set_params");
        return undefined(env);
    }
    int rc = p_set_params(h->sim, dt, G);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_set_params");
    return undefined(env);
}

static napi_value js_step(napi_env env, napi_callback_info info)
{
    size_t argc = 2; napi_value argv[2];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1) { napi_throw_type_error(env, NULL, "step(handle[, nsteps])"); return NULL; }
    handle_t *h = get_handle(env, argv[0]); if (!h) return NULL;
    uint32_t n = 1;
    if (argc >= 2) { double d; if (napi_get_value_double(env, argv[1], &d) == napi_ok && d >= 0) n = (uint32_t)d; }
    if (h->multi) {
        int rc = p_multi_step(h->multi, n);
        if (rc != NB_OK) return throw_msg(env, rc, p_multi_last_error(h->multi), "nb_multi_step");
        return undefined(env);
    }
    int rc = p_step(h->sim, n);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_step");
    return undefined(env);
}

static napi_value js_download(napi_env env, napi_callback_info info)
{
    size_t argc = 4; napi_value argv[4];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 4) { napi_throw_type_error(env, NULL, "download(handle, bodies|null, vel|null, accel|null)"); return NULL; }
    handle_t *h = get_handle(env, argv[0]); if (!h) return NULL;
    void *b, *v, *a;
    if (!get_array(env, argv[1], h, 1, &b, "bodies: typed array of 4*n elements or null")) return NULL;
    if (!get_array(env, argv[2], h, 1, &v, "vel: typed array of 4*n elements or null")) return NULL;
    if (!get_array(env, argv[3], h, 1, &a, "accel: typed array of 4*n elements or null")) return NULL;
    if (h->multi) {
        int rc = p_multi_download(h->multi, b, v, a);
        if (rc != NB_OK) return throw_msg(env, rc, p_multi_last_error(h->multi), "nb_multi_download");
        return undefined(env);
    }
    int rc = p_download(h->sim, b, v, a);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_download");
    return undefined(env);
}

static napi_value js_sync(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    if (h->multi) {
        int rc = p_multi_sync(h->multi);
        if (rc != NB_OK) return throw_msg(env, rc, p_multi_last_error(h->multi), "nb_multi_sync");
        return undefined(env);
    }
    int rc = p_sync(h->sim);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_sync");
    return undefined(env);
}

static napi_value js_destroy(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void *p = NULL;
    if (argc && napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        handle_t *h = (handle_t *)p;
        if (h->sim) { p_destroy(h->sim); h->sim = NULL; }   /* idempotent; finalizer frees the shell */
        if (h->multi) { p_multi_destroy(h->multi); h->multi = NULL; }
    }
    return undefined(env);
}

static napi_value js_enable_timing(napi_env env, napi_callback_info info)
{
    size_t argc = 2; napi_value argv[2];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    if (h->multi) { napi_throw_error(env, "NB_1", "per-kernel timing is not available on a multi-device handle"); return NULL; }
    bool on = true; if (argc >= 2) napi_get_value_bool(env, argv[1], &on);
    int rc = p_enable_timing(h->sim, on ? 1 : 0);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_enable_timing");
    return undefined(env);
}

/* kernelTimes(handle) -> {forceMs, integrateMs, launches} */
static napi_value js_kernel_times(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    if (h->multi) { napi_throw_error(env, "NB_1", "per-kernel timing is not available on a multi-device handle"); return NULL; }
    double f, g; uint32_t c;
    int rc = p_kernel_times(h->sim, &f, &g, &c);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_kernel_times");
    napi_value o, v;
    CHECK_NAPI(env, napi_create_object(env, &o));
    napi_create_double(env, f, &v); napi_set_named_property(env, o, "forceMs", v);
    napi_create_double(env, g, &v); napi_set_named_property(env, o, "integrateMs", v);
    napi_create_uint32(env, c, &v); napi_set_named_property(env, o, "launches", v);
    return o;
}

static napi_value js_variant(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    napi_value v;
    CHECK_NAPI(env, napi_create_string_utf8(env, h->multi ? p_multi_variant_name(h->multi) : p_variant_name(h->sim), NAPI_AUTO_LENGTH, &v));
    return v;
}

/* diagnostics(handle) -> {kinetic, potential, momentum:[3]} */
static napi_value js_diagnostics(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    double out[5];
    if (h->multi) {
        int rcm = p_multi_diagnostics(h->multi, out);
        if (rcm != NB_OK) return throw_msg(env, rcm, p_multi_last_error(h->multi), "nb_multi_diagnostics");
    } else {
        int rc = p_diagnostics(h->sim, out);
        if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_diagnostics");
    }
    napi_value o, v, arr;
    CHECK_NAPI(env, napi_create_object(env, &o));
    napi_create_double(env, out[0], &v); napi_set_named_property(env, o, "kinetic", v);
    napi_create_double(env, out[1], &v); napi_set_named_property(env, o, "potential", v);
    napi_create_array_with_length(env, 3, &arr);
    for (uint32_t i = 0; i < 3; ++i) { napi_create_double(env, out[2 + i], &v); napi_set_element(env, arr, i, v); }
    napi_set_named_property(env, o, "momentum", arr);
    return o;
}


/* stepTimes(handle) -> {forceMs, integrateMs, exchangeMs, launches, symReduceMs, reduceScatterMs, allgatherMs, spanMs}
 * (nb_step_times2; forceMs / exchangeMs keep nb_step_times's meaning: force pass + nb_sym_reduce, every native collective) */
static napi_value js_step_times(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    if (h->multi) { napi_throw_error(env, "NB_1", "per-kernel timing is not available on a multi-device handle"); return NULL; }
    nb_step_timing t;
    memset(&t, 0, sizeof t);
    t.struct_size = sizeof t;
    int rc = p_step_times2(h->sim, &t);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_step_times2");
    const double f = t.force_ms + t.sym_reduce_ms, g = t.integrate_ms, x = t.reduce_scatter_ms + t.allgather_ms;
    const uint32_t c = t.launches;
    napi_value o, v;
    CHECK_NAPI(env, napi_create_object(env, &o));
    napi_create_double(env, t.sym_reduce_ms, &v); napi_set_named_property(env, o, "symReduceMs", v);
    napi_create_double(env, t.reduce_scatter_ms, &v); napi_set_named_property(env, o, "reduceScatterMs", v);
    napi_create_double(env, t.allgather_ms, &v); napi_set_named_property(env, o, "allgatherMs", v);
    napi_create_double(env, t.span_ms, &v); napi_set_named_property(env, o, "spanMs", v);
    napi_create_double(env, f, &v); napi_set_named_property(env, o, "forceMs", v);
    napi_create_double(env, g, &v); napi_set_named_property(env, o, "integrateMs", v);
    napi_create_double(env, x, &v); napi_set_named_property(env, o, "exchangeMs", v);
    napi_create_uint32(env, c, &v); napi_set_named_property(env, o, "launches", v);
    return o;
}

/* collectiveInfo(handle) -> {mode: 'peer'|'rccl'|'none', nranks, rcclVersion} */
static napi_value js_collective_info(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    int mode = -1, nranks = 0, ver = 0;
    if (h->multi) {
        int rc = p_multi_collective_info(h->multi, &mode, &nranks, &ver);
        if (rc != NB_OK) return throw_msg(env, rc, p_multi_last_error(h->multi), "nb_multi_collective_info");
    }
    napi_value o, v;
    CHECK_NAPI(env, napi_create_object(env, &o));
    napi_create_string_utf8(env, mode == NB_MULTI_RCCL ? "rccl" : (mode == NB_MULTI_PEER ? "peer" : "none"), NAPI_AUTO_LENGTH, &v);
    napi_set_named_property(env, o, "mode", v);
    napi_create_int32(env, nranks, &v); napi_set_named_property(env, o, "nranks", v);
    napi_create_int32(env, ver, &v); napi_set_named_property(env, o, "rcclVersion", v);
    return o;
}

/* requestFrame(handle): enqueue a viewer snapshot behind the steps issued so far (returns at once) */
static napi_value js_request_frame(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_t *h = argc ? get_handle(env, argv[0]) : NULL; if (!h) return NULL;
    if (h->multi) { napi_throw_error(env, "NB_1", "the frame feed is not available on a multi-device handle"); return NULL; }
    int rc = p_frame_request(h->sim);
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_frame_request");
    return undefined(env);
}

/* frame(handle, wait, bodiesOut Float32Array(4n), speedOut Float32Array(n)) -> step index, or -1 when
 * wait is false and no frame has landed yet.  Copies out of the engine's pinned frame slot. */
static napi_value js_frame(napi_env env, napi_callback_info info)
{
    size_t argc = 4; napi_value argv[4];
    CHECK_NAPI(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 4) { napi_throw_type_error(env, NULL, "frame(handle, wait, bodiesOut, speedOut)"); return NULL; }
    handle_t *h = get_handle(env, argv[0]); if (!h) return NULL;
    if (h->multi) { napi_throw_error(env, "NB_1", "the frame feed is not available on a multi-device handle"); return NULL; }
    bool wait = true; napi_get_value_bool(env, argv[1], &wait);
    void *dst[2]; const size_t want[2] = {(size_t)4 * h->n, (size_t)h->n};
    for (int k = 0; k < 2; ++k) {
        bool is_ta = false; napi_is_typedarray(env, argv[2 + k], &is_ta);
        napi_typedarray_type tt; size_t len = 0; napi_value ab; size_t off;
        if (!is_ta || napi_get_typedarray_info(env, argv[2 + k], &tt, &len, &dst[k], &ab, &off) != napi_ok ||
            tt != napi_float32_array || len != want[k]) {
            napi_throw_type_error(env, NULL, k ? "speedOut must be a Float32Array of n elements" : "bodiesOut must be a Float32Array of 4*n elements");
            return NULL;
        }
    }
    const float *b = NULL, *sp = NULL; uint64_t step = 0;
    int rc = p_frame_acquire(h->sim, wait ? 1 : 0, &b, &sp, &step);
    napi_value v;
    if (rc == NB_NOT_READY) { napi_create_double(env, -1.0, &v); return v; }
    if (rc != NB_OK) return throw_nb(env, rc, h->sim, "nb_frame_acquire");
    memcpy(dst[0], b, sizeof(float) * want[0]);
    memcpy(dst[1], sp, sizeof(float) * want[1]);
    napi_create_double(env, (double)step, &v);
    return v;
}

static napi_value init_module(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"load", js_load}, {"deviceCount", js_device_count}, {"create", js_create}, {"upload", js_upload},
        {"setParams", js_set_params}, {"step", js_step}, {"download", js_download}, {"sync", js_sync},
        {"destroy", js_destroy}, {"enableTiming", js_enable_timing}, {"kernelTimes", js_kernel_times},
        {"variant", js_variant}, {"diagnostics", js_diagnostics}, {"stepTimes", js_step_times},
        {"collectiveInfo", js_collective_info}, {"requestFrame", js_request_frame}, {"frame", js_frame}, {"planQuery", js_plan_query},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; ++i) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}

NAPI_MODULE(nb_napi, init_module)
