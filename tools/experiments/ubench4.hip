// ubench4.hip -- where does a small-N fused step spend its time?  Diagnostic build of the product
// kernels (-DNB_STAMPS: per-wave s_memtime at phase boundaries; the product build has no stamps).
//   phases per wave:  0 entry -> 1 first tile staged (args, i-bodies, global loads, LDS store, barrier)
//                     -> 2 j-loop done -> 3 in-wave reduction done -> 4 integrated and stored
// Reports, per shape and N: event-timed us per launch (un-stamped numbers come from
// tools/shape_scan.py), the median phase lengths in shader cycles, and the spread of the entry
// stamps over the grid (launch ramp).
// Build: hipcc -O3 --offload-arch=gfx950 -DNB_STAMPS -I../../nbody3d-webgpu_amd/csrc -o ubench4 ubench4.hip
#include "nb_kernels.hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NG, int LS, int TL>
void run(uint32_t n, int steps)
{
    constexpr int IPB = (256 / LS) * 2 * NG;
    const uint32_t grid = (n + IPB - 1) / IPB;
    std::vector<float4> hb(n), hz(n, float4{0, 0, 0, 0});
    srand(1);
    for (uint32_t i = 0; i < n; ++i) hb[i] = float4{(float)rand() / RAND_MAX, (float)rand() / RAND_MAX, (float)rand() / RAND_MAX, 1.0f / n};
    float4 *b0, *b1, *v, *a;
    unsigned long long* st;
    CK(hipMalloc(&b0, 16 * n)); CK(hipMalloc(&b1, 16 * n)); CK(hipMalloc(&v, 16 * n)); CK(hipMalloc(&a, 16 * n));
    CK(hipMalloc(&st, sizeof(unsigned long long) * 16 * 4 * grid));
    CK(hipMemcpy(b0, hb.data(), 16 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hz.data(), 16 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, hz.data(), 16 * n, hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(nb::nb_stamp_buf), &st, sizeof st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < steps; ++k) {
            hipLaunchKernelGGL((nb::nb_step_fused<NG, LS, TL>), grid, 256, 0, 0, (const float4*)((k & 1) ? b1 : b0), (k & 1) ? b0 : b1, v, a, n, 1.0f, 1e-4f, 1e-3f);
        }
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h((size_t)16 * 4 * grid);
    CK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    const size_t nw = (size_t)4 * grid;
    std::vector<double> ph[4], t0, t4, clk;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * 16] || !h[w * 16 + 4]) continue;
        for (int p = 0; p < 4; ++p) ph[p].push_back((double)(h[w * 16 + p + 1] - h[w * 16 + p]));
        t0.push_back((double)h[w * 16]); t4.push_back((double)h[w * 16 + 4]);
        if (h[w * 16 + 6] > h[w * 16 + 5]) clk.push_back((double)(h[w * 16 + 4] - h[w * 16]) / (double)(h[w * 16 + 6] - h[w * 16 + 5]) * 0.1);
    }
    for (auto& p : ph) std::sort(p.begin(), p.end());
    std::sort(clk.begin(), clk.end());
    const size_t m = ph[0].size() / 2;
    const double first = *std::min_element(t0.begin(), t0.end()), lastin = *std::max_element(t0.begin(), t0.end());
    const double lastout = *std::max_element(t4.begin(), t4.end());
    const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    printf("N=%6u fused<NG%d,LS%2d,TL%d> grid %5u waves %5zu/%5zu  %7.2f us/launch | cycles med: stage %6.0f loop %7.0f reduce %5.0f integrate %5.0f | "
           "entry spread %6.0f, first-in->last-out %7.0f cycles; in-kernel clock %.2f GHz -> %.2f us\n",
           n, NG, LS, TL, grid, ph[0].size(), nw, 1e3 * best / steps, ph[0][m], ph[1][m], ph[2][m], ph[3][m], lastin - first, lastout - first,
           ghz, ghz > 0 ? (lastout - first) / (ghz * 1e3) : 0.0);
    {   // inside tile 1 of the loop: issue of the next tile's loads, the tile's math, LDS store of the next tile, barrier
        std::vector<double> a, b, c, d;
        for (size_t w = 0; w < nw; ++w) {
            if (!h[w * 16 + 8] || !h[w * 16 + 12]) continue;
            a.push_back((double)(h[w * 16 + 9] - h[w * 16 + 8])); b.push_back((double)(h[w * 16 + 10] - h[w * 16 + 9]));
            c.push_back((double)(h[w * 16 + 11] - h[w * 16 + 10])); d.push_back((double)(h[w * 16 + 12] - h[w * 16 + 11]));
        }
        if (!a.empty()) {
            for (auto* v : {&a, &b, &c, &d}) std::sort(v->begin(), v->end());
            const size_t mm = a.size() / 2, p9 = a.size() * 9 / 10;
            printf("        tile 1, cycles med (p90): issue loads %.0f (%.0f)  math %.0f (%.0f)  wait+LDS store %.0f (%.0f)  barrier %.0f (%.0f)\n",
                   a[mm], a[p9], b[mm], b[p9], c[mm], c[p9], d[mm], d[p9]);
        }
    }
    // residency: workgroups per CU (HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13; XCC_ID bits 3:0) and the
    // slowest wave against the median one
    {
        std::vector<int> per_cu(8 * 64, 0);
        std::vector<double> tot;
        for (size_t w = 0; w < nw; ++w) {
            if (!h[w * 16] || !h[w * 16 + 4]) continue;
            tot.push_back((double)(h[w * 16 + 4] - h[w * 16]));
            if (w % 4 == 0) {
                const unsigned hw = (unsigned)(h[w * 16 + 7] & 0xffffffffu), xcc = (unsigned)(h[w * 16 + 7] >> 32) & 0xf;
                const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                per_cu[(xcc & 7) * 64 + ((se * 2 + sh) * 16 + cu) % 64]++;
            }
        }
        // who is slow?  mean lifetime by XCC, by blockIdx quartile, by wave slot of the workgroup
        double sx[8] = {0}, sq[4] = {0}, sw[4] = {0}; int nx[8] = {0}, nq[4] = {0}, nwv[4] = {0};
        for (size_t w = 0; w < nw; ++w) {
            if (!h[w * 16] || !h[w * 16 + 4]) continue;
            const double life = (double)(h[w * 16 + 4] - h[w * 16]);
            const unsigned xcc = (unsigned)(h[(w / 4 * 4) * 16 + 7] >> 32) & 7;
            sx[xcc] += life; nx[xcc]++;
            const int q = (int)((w / 4) * 4 / grid); sq[q] += life; nq[q]++;
            sw[w % 4] += life; nwv[w % 4]++;
        }
        printf("        mean lifetime by XCC:");
        for (int k = 0; k < 8; ++k) printf(" %.0f", nx[k] ? sx[k] / nx[k] : 0.0);
        printf(" | by blockIdx quartile:");
        for (int k = 0; k < 4; ++k) printf(" %.0f", nq[k] ? sq[k] / nq[k] : 0.0);
        printf(" | by wave slot:");
        for (int k = 0; k < 4; ++k) printf(" %.0f", nwv[k] ? sw[k] / nwv[k] : 0.0);
        printf("\n");
        std::sort(tot.begin(), tot.end());
        int hist[12] = {0}, used = 0;
        for (int c : per_cu) { if (c) ++used; hist[c > 11 ? 11 : c]++; }
        printf("        wave lifetime cycles: med %.0f  p95 %.0f  max %.0f | CUs used %d; CUs holding k workgroups:", tot[tot.size() / 2],
               tot[tot.size() * 95 / 100], tot.back(), used);
        for (int k = 1; k < 12; ++k) if (hist[k]) printf(" %d:%d", k, hist[k]);
        printf("\n");
    }
    CK(hipFree(b0)); CK(hipFree(b1)); CK(hipFree(v)); CK(hipFree(a)); CK(hipFree(st));
}

int main()
{
    const int steps = 200;
    printf("note: s_memtime counts shader-clock cycles on gfx950 (MI355X_MICROARCH.md cycle constants)\n");
    run<1, 64, 4>(2048, steps);
    run<1, 64, 4>(4096, steps);
    run<1, 32, 4>(4096, steps);
    run<1, 64, 4>(8192, steps);
    run<1, 64, 1>(8192, steps);
    run<1, 32, 1>(8192, steps);
    run<2, 64, 4>(16384, steps);
    run<2, 64, 1>(16384, steps);
    run<4, 64, 4>(16384, steps);
    run<4, 32, 1>(32768, 50);
    run<4, 64, 1>(32768, 50);
    run<4, 16, 1>(65536, 20);
    return 0;
}
