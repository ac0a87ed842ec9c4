/* A plain-C client of include/nbody3d_hip.h (no Python, no Node): what a C host of the
 * reference's hot path would link.  Driven by tests/test_c_client.py.
 *
 *   abi_client <golden dir>    exit 0 = all checks passed; prints one line per check.
 * Without a GPU it checks the no-device contract; with one it runs the N=1,024 Plummer
 * fixture for 10 steps and compares with the committed oracle vector. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbody3d_hip.h"

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
    if (nb_abi_version() != NB_ABI_VERSION) { printf("FAIL abi version\n"); return 1; }

    nb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.n = n;
    cfg.device = -1;
    nb_sim *sim = NULL;
    int rc = nb_create(&cfg, &sim);
    if (nb_device_count() == 0) {
        /* reference: alert + return when WebGPU is missing (nbody3d.js:151-155) */
        int ok = rc == NB_ERR_NO_DEVICE && sim == NULL && strstr(nb_last_error(NULL), "no CPU fallback") != NULL;
        printf("%s no-device contract (rc=%d, \"%s\")\n", ok ? "ok" : "FAIL", rc, nb_last_error(NULL));
        return ok ? 0 : 1;
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
    nb_destroy(sim);
    free(b); free(v); free(ref); free(out); free(acc);
    return fails ? 1 : 0;
}
