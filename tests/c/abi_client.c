/* A plain-C client of include/nbody3d_hip.h (no Python, no Node): what a C host of the
 * reference's hot path would link.  Driven by tests/test_c_client.py.
 *
 *   abi_client <golden dir>    exit 0 = all checks passed; prints one line per check.
 * Without a GPU it checks the no-device contract; with one it runs the N=1,024 Plummer
 * fixture for 10 steps and compares with the committed oracle vector, then the frame feed, the
 * kernel timing, and both native-RCCL forms with one rank. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbody3d_hip.h"
#include "nbody3d_hip_plan.h" /* planner introspection: not part of the drop-in surface, checked here because the library exports it */

static float *load_f32(const char *dir, const char *name, size_t count)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s.f32", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    float *p = (float *)malloc(count * sizeof(float));
    if (fread(p, sizeof(float), count, f) != count) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    const char *gold = argc > 1 ? argv[1] : "tests/golden";
    const uint32_t n = 1024;
    int fails = 0;
    printf("abi_version %u\n", nb_abi_version());
    if (nb_abi_version() != NB_ABI_VERSION || nb_abi_minor() < NB_ABI_MINOR) { printf("FAIL abi version\n"); return 1; }

    nb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.n = n;
    cfg.device = -1;
    /* the launch planner needs no device: what an MI355X (256 CUs, 2.4 GHz) would run for the headline system */
    {
        nb_config big = cfg;
        big.n = 262144;
        nb_plan_info pi;
        memset(&pi, 0, sizeof pi);
        pi.struct_size = sizeof pi;
        uint32_t tab[512];
        int prc = nb_plan_query(&big, 256, 2.4e9, &pi, tab, 512);
        int ok = prc == NB_OK && pi.sym == 1 && pi.symw == 1 && pi.sym_np == 262144 && pi.tab_len >= 512 + 4 * 2048 && (pi.tab_len - 512 - 4 * 2048) % 2 == 0 /* {first wave, resident layers} per super-block + {first unit, end, resident layer, spill row} per wave + {first unit, layer | sweeps << 16} per queued piece */ && tab[0] == 0 && tab[1] >= 1 &&
                 strncmp(pi.variant, "f32pk_symw_ipl16_j1_w2048", 25) == 0;
        printf("%s plan query without a device: %s, %u layers\n", ok ? "ok" : "FAIL", pi.variant, pi.sym_layers);
        if (!ok) fails++;
    }
    nb_sim *sim = NULL;
    int rc = nb_create(&cfg, &sim);
    if (nb_device_count() == 0) {
        /* reference: alert + return when WebGPU is missing (nbody3d.js:151-155) */
        int ok = rc == NB_ERR_NO_DEVICE && sim == NULL && strstr(nb_last_error(NULL), "no CPU fallback") != NULL;
        printf("%s no-device contract (rc=%d, \"%s\")\n", ok ? "ok" : "FAIL", rc, nb_last_error(NULL));
        return ok && !fails ? 0 : 1;
    }
    if (rc != NB_OK) { printf("FAIL nb_create: %s\n", nb_last_error(NULL)); return 1; }

    float *b = load_f32(gold, "plummer1024_bodies0", 4 * n), *v = load_f32(gold, "plummer1024_vel0", 4 * n);
    float *ref = load_f32(gold, "plummer1024_s10_bodies", 4 * n);
    float *out = (float *)malloc(sizeof(float) * 4 * n), *acc = (float *)malloc(sizeof(float) * 4 * n);

    rc = nb_step(sim, 1);                          /* before upload: a state error, not a crash */
    if (rc != NB_ERR_STATE) { printf("FAIL step-before-upload rc=%d\n", rc); fails++; } else printf("ok step-before-upload -> NB_ERR_STATE\n");
    if (nb_upload(sim, b, v, NULL) != NB_OK || nb_set_params(sim, 1e-3, 1.0) != NB_OK || nb_step(sim, 10) != NB_OK ||
        nb_download(sim, out, NULL, acc) != NB_OK) {
        printf("FAIL run: %s\n", nb_last_error(sim));
        return 1;
    }
    double worst = 0.0;
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            double d = fabs((double)out[4 * i + c] - (double)ref[4 * i + c]);
            if (d > worst) worst = d;
        }
    printf("%s fixture after 10 steps: max |dx| vs oracle = %.3g (variant %s)\n", worst < 1e-5 ? "ok" : "FAIL", worst,
           nb_variant_name(sim));
    if (!(worst < 1e-5)) fails++;
    for (uint32_t i = 0; i < n; ++i)
        if (out[4 * i + 3] != b[4 * i + 3] || acc[4 * i + 3] != 0.0f) { printf("FAIL mass/accel.w lane\n"); fails++; break; }
    double diag[5];
    if (nb_diagnostics(sim, diag) != NB_OK || !(diag[1] < 0.0)) { printf("FAIL diagnostics\n"); fails++; } else printf("ok diagnostics KE=%.6f PE=%.6f\n", diag[0], diag[1]);

    /* viewer frame feed (nbody3d.js:408-415,482-487): snapshot of the state now, read back after more steps */
    const float *fb = NULL, *fs = NULL;
    uint64_t fstep = 0;
    if (nb_frame_request(sim) != NB_OK || nb_step(sim, 3) != NB_OK || nb_frame_acquire(sim, 1, &fb, &fs, &fstep) != NB_OK) {
        printf("FAIL frame feed: %s\n", nb_last_error(sim)); fails++;
    } else {
        int same = fstep == 10 && memcmp(fb, out, sizeof(float) * 4 * n) == 0 && fs[0] >= 0.0f;
        printf("%s frame feed: snapshot of step %llu equals the download of that step\n", same ? "ok" : "FAIL", (unsigned long long)fstep);
        if (!same) fails++;
    }
    /* kernel-exact timing (role of TimingHelper, util.js:297-423) */
    double f_ms = 0, i_ms = 0, x_ms = 0; uint32_t launches = 0;
    if (nb_enable_timing(sim, 1) != NB_OK || nb_step(sim, 4) != NB_OK || nb_step_times(sim, &f_ms, &i_ms, &x_ms, &launches) != NB_OK ||
        launches != 4 || !(f_ms > 0.0)) { printf("FAIL step times\n"); fails++; }
    else printf("ok step times: %u launches, force %.4f ms, integrate %.4f ms\n", launches, f_ms, i_ms);
    /* the part-by-part form: on a whole-system handle only the force (and integrate) parts are non-zero */
    nb_step_timing tm;
    memset(&tm, 0, sizeof tm);
    tm.struct_size = sizeof tm;
    if (nb_step(sim, 2) != NB_OK || nb_step_times2(sim, &tm) != NB_OK || tm.launches != 2 || !(tm.force_ms > 0.0) ||
        tm.reduce_scatters != 0 || tm.allgathers != 0 || !(tm.span_ms >= tm.force_ms + tm.integrate_ms - 1e-9)) { printf("FAIL step times2\n"); fails++; }
    else printf("ok step times2: force %.4f + integrate %.4f ms inside a span of %.4f ms\n", tm.force_ms, tm.integrate_ms, tm.span_ms);
    tm.struct_size = 4;
    if (nb_step_times2(sim, &tm) != NB_ERR_INVALID) { printf("FAIL step times2 accepts a short struct\n"); fails++; }
    nb_destroy(sim);

    /* native RCCL collective with one rank: a shard handle that owns every row */
    cfg.shard_begin = 0; cfg.shard_count = n;
    nb_sim *sh = NULL;
    unsigned char id[NB_RCCL_ID_BYTES];
    int nr = 0, rk = -1, ver = 0;
    if (nb_create(&cfg, &sh) != NB_OK || nb_rccl_unique_id(id) != NB_OK || nb_rccl_attach(sh, id, 1, 0, 0) != NB_OK ||
        nb_rccl_info(sh, &nr, &rk, &ver) != NB_OK || nr != 1 || rk != 0 ||
        nb_upload(sh, b, v, NULL) != NB_OK || nb_set_params(sh, 1e-3, 1.0) != NB_OK || nb_step(sh, 10) != NB_OK ||
        nb_download(sh, acc, NULL, NULL) != NB_OK) {
        printf("FAIL rccl single rank: %s\n", sh ? nb_last_error(sh) : nb_last_error(NULL)); fails++;
    } else {
        double w2 = 0.0;
        for (uint32_t i = 0; i < 4 * n; ++i) { double d = fabs((double)acc[i] - (double)ref[i]); if (i % 4 != 3 && d > w2) w2 = d; }
        printf("%s rccl-attached handle (nranks %d, rccl %d): max |dx| vs oracle = %.3g\n", w2 < 1e-5 ? "ok" : "FAIL", nr, ver, w2);
        if (!(w2 < 1e-5)) fails++;
    }
    if (sh) { nb_rccl_detach(sh); nb_destroy(sh); }

    /* one process, one shard, RCCL mode of the multi handle: ncclCommInitAll + grouped all-gather */
    cfg.shard_begin = cfg.shard_count = 0;
    nb_multi *m = NULL;
    int mode = -1;
    if (nb_multi_create(&cfg, 1, NULL, &m) != NB_OK || nb_multi_set_collective(m, NB_MULTI_RCCL) != NB_OK ||
        nb_multi_collective_info(m, &mode, &nr, &ver) != NB_OK || mode != NB_MULTI_RCCL || nr != 1 ||
        nb_multi_upload(m, b, v, NULL) != NB_OK || nb_multi_set_params(m, 1e-3, 1.0) != NB_OK || nb_multi_step(m, 10) != NB_OK ||
        nb_multi_download(m, acc, NULL, NULL) != NB_OK) {
        printf("FAIL multi rccl: %s\n", nb_multi_last_error(m)); fails++;
    } else {
        double w3 = 0.0;
        for (uint32_t i = 0; i < 4 * n; ++i) { double d = fabs((double)acc[i] - (double)ref[i]); if (i % 4 != 3 && d > w3) w3 = d; }
        printf("%s multi handle in RCCL mode: max |dx| vs oracle = %.3g\n", w3 < 1e-5 ? "ok" : "FAIL", w3);
        if (!(w3 < 1e-5)) fails++;
    }
    if (m) nb_multi_destroy(m);
    free(b); free(v); free(ref); free(out); free(acc);
    return fails ? 1 : 0;
}
