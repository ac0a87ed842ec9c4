/* The CPU oracle (oracle/nb_oracle.c, test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer: the golden
 * fixture must come out bit for bit, ragged ranges and the threaded baseline kernel must stay inside their arrays.
 * Built and run by tests/test_oracle.py (no GPU). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void nbo_accel_f32(const float *, uint32_t, float, float, uint32_t, uint32_t, float *);
int nbo_run_f32(float *, float *, float *, uint32_t, float, float, float, uint32_t);
void nbo_accel_f64(const double *, uint32_t, double, double, uint32_t, uint32_t, double *);
int nbo_run_f64(double *, double *, double *, uint32_t, double, double, double, uint32_t);
int nbo_accel_f32_mt(const float *, uint32_t, float, float, uint32_t, uint32_t, float *, int);

static float *load(const char *dir, const char *name, size_t count)
{
    char path[1024];
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
    float *b = load(gold, "plummer1024_bodies0", 4 * n), *v = load(gold, "plummer1024_vel0", 4 * n);
    float *a = (float *)calloc(4 * n, sizeof(float));
    float *want = load(gold, "plummer1024_s10_bodies", 4 * n);
    if (nbo_run_f32(b, v, a, n, 1e-3f, 1.0f, 1e-4f, 10) != 0 || memcmp(b, want, sizeof(float) * 4 * n) != 0) { printf("FAIL golden s10\n"); fails++; }
    else printf("ok golden fixture, 10 steps, bit for bit\n");
    /* ragged ranges: exactly-sized output arrays, so an index past the range is an ASan report */
    for (uint32_t i0 = 0; i0 < n; i0 += 333) {
        uint32_t i1 = i0 + 77 > n ? n : i0 + 77;
        float *o = (float *)malloc(sizeof(float) * 4 * (i1 - i0));
        float *om = (float *)malloc(sizeof(float) * 4 * (i1 - i0));
        nbo_accel_f32(b, n, 1.0f, 1e-4f, i0, i1, o);
        if (nbo_accel_f32_mt(b, n, 1.0f, 1e-4f, i0, i1, om, 3) <= 0) { printf("FAIL mt\n"); fails++; }
        for (uint32_t k = 0; k < 4 * (i1 - i0); ++k)
            if (fabsf(o[k] - om[k]) > 2e-5f * (fabsf(o[k]) + 1e-3f)) { printf("FAIL mt vs scalar at %u\n", k); fails++; break; }
        free(o); free(om);
    }
    /* odd sizes incl. n = 1, f64 twin */
    for (uint32_t m = 1; m <= 70; m += 23) {
        double *bd = (double *)malloc(sizeof(double) * 4 * m), *vd = (double *)calloc(4 * m, sizeof(double)), *ad = (double *)calloc(4 * m, sizeof(double));
        for (uint32_t k = 0; k < 4 * m; ++k) bd[k] = b[k];
        if (nbo_run_f64(bd, vd, ad, m, 1e-3, 1.0, 1e-4, 3) != 0) { printf("FAIL f64 run\n"); fails++; }
        double *od = (double *)malloc(sizeof(double) * 4 * m);
        nbo_accel_f64(bd, m, 1.0, 1e-4, 0, m, od);
        for (uint32_t k = 0; k < 4 * m; ++k) if (!isfinite(od[k])) { printf("FAIL f64 finite\n"); fails++; break; }
        free(bd); free(vd); free(ad); free(od);
    }
    printf("%s oracle under ASan + UBSan\n", fails ? "FAIL" : "ok");
    free(b); free(v); free(a); free(want);
    return fails ? 1 : 0;
}
