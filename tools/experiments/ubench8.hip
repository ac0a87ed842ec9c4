// ubench8.hip -- can the integrate kernel of step s run UNDER the force kernel of step s + 1?
//
// Today a step is K1 (force) -> K2 (integrate) -> K1 ... on one stream: two dependent kernel boundaries (~1.45 us each) and K2's ~3 us
// per step, a quarter of the step at N = 8,192.  K1(s+1) needs K2(s)'s rows, but only chunk by chunk: if K1(s+1) were launched beside
// K2(s) (both behind K1(s)) and waited for per-chunk flags that K2(s) raises, one boundary and most of K2 would leave the critical path.
// This file measures, on a full chip:
//   (1) a chain graph A -> B -> A -> B ... against the forked graph A(s) -> {B(s), A(s+1)}, B(s) -> B(s+1): does the runtime run B(s)
//       beside A(s+1), and what does a step cost then?  (A: 256 workgroups of 256 threads busy for `ta` us, B: 256 x 256 busy for `tb`.)
//   (2) the hand-over: B(s) writes 64-row chunks (write-through) and raises a flag per chunk; A(s+1), launched beside it, polls the
//       flag (bounded) and checks every row it then reads -- a stale row or a timeout is counted.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench8 ubench8.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long realtime() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

__global__ __launch_bounds__(256) void k_busy(unsigned ticks, unsigned* sink)
{
    const unsigned long long t0 = realtime();
    unsigned n = 0;
    while (realtime() - t0 < ticks) ++n;
    if (n == 0xffffffffu) *sink = n;
}

// B(s): rows [64 c, 64 c + 64) of chunk c get the value (s, row), write-through; then the chunk's flag becomes s
__global__ __launch_bounds__(256) void k_write(float4* rows, unsigned* flags, unsigned chunks, unsigned step, unsigned delay_ticks)
{
    const unsigned long long t0 = realtime();
    while (realtime() - t0 < delay_ticks) { }
    const unsigned c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= chunks) return;
    const v4f v = v4f{(float)step, (float)(c * 64 + lane), 1.f, 2.f};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(&rows[c * 64 + lane]), "v"(v) : "memory");
    if (lane == 0) {
        asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(&flags[c]), "v"(step) : "memory");
    }
}

// A(s+1): every wave polls the flag of `per_wave` chunks (those a real K1 wave would sweep first), then reads the rows and checks them
__global__ __launch_bounds__(256) void k_read(const float4* rows, const unsigned* flags, unsigned chunks, unsigned step, unsigned per_wave, unsigned busy_ticks,
                                             unsigned* bad, unsigned* timeouts, unsigned long long* wait_ticks)
{
    const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned long long waited = 0;
    for (unsigned q = 0; q < per_wave; ++q) {
        const unsigned c = (w * 7 + q * 13) % chunks;
        const unsigned long long t0 = realtime();
        unsigned f, polls = 0;
        do {
            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(&flags[c]) : "memory");
            f = __builtin_amdgcn_readfirstlane(f);
        } while (f != step && ++polls < 200000u);
        waited += realtime() - t0;
        if (f != step) { if (lane == 0) atomicAdd(timeouts, 1u); continue; }
        v4f v;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&rows[c * 64 + lane]) : "memory");
        if (v.x != (float)step || v.y != (float)(c * 64 + lane)) atomicAdd(bad, 1u);
    }
    if (lane == 0) atomicAdd(wait_ticks, waited);
    const unsigned long long t0 = realtime();
    while (realtime() - t0 < busy_ticks) { }
}

static double replay(hipGraphExec_t g, hipStream_t s, int reps, int steps)
{
    CK(hipGraphLaunch(g, s)); CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(g, s));
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / (reps * steps);
}

int main()
{
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    const int steps = 16;
    // ---- (1) chain against fork
    for (unsigned ta : {800u, 1500u, 4000u}) for (unsigned tb : {150u, 300u}) {
        hipGraph_t gc, gf; hipGraphExec_t ec, ef;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        for (int s = 0; s < steps; ++s) { k_busy<<<256, 256, 0, s1>>>(ta, sink); k_busy<<<256, 256, 0, s1>>>(tb, sink); }
        CK(hipStreamEndCapture(s1, &gc)); CK(hipGraphInstantiate(&ec, gc, nullptr, nullptr, 0));
        std::vector<hipEvent_t> ev(2 * steps + 2);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        for (int s = 0; s < steps; ++s) {
            k_busy<<<256, 256, 0, s1>>>(ta, sink);                     // A(s): behind A(s-1) on s1
            CK(hipEventRecord(ev[2 * s], s1)); CK(hipStreamWaitEvent(s2, ev[2 * s], 0));
            k_busy<<<256, 256, 0, s2>>>(tb, sink);                     // B(s): behind A(s) and B(s-1), beside A(s+1)
        }
        CK(hipEventRecord(ev[2 * steps], s2)); CK(hipStreamWaitEvent(s1, ev[2 * steps], 0));
        CK(hipStreamEndCapture(s1, &gf)); CK(hipGraphInstantiate(&ef, gf, nullptr, nullptr, 0));
        const double c = replay(ec, s1, 200, steps), f = replay(ef, s1, 200, steps);
        printf("A %5.1f us + B %4.1f us per step: chain %6.2f us/step, fork %6.2f us/step\n", ta / 100.0, tb / 100.0, c, f);
        fflush(stdout);
    }
    // ---- (2) the hand-over beside a running reader
    const unsigned chunks = 256;             // N = 16,384
    float4* rows; unsigned *flags, *bad, *timeouts; unsigned long long* wait_ticks;
    CK(hipMalloc(&rows, chunks * 64 * sizeof(float4))); CK(hipMalloc(&flags, chunks * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&timeouts, 4)); CK(hipMalloc(&wait_ticks, 8));
    CK(hipMemset(rows, 0, chunks * 64 * sizeof(float4))); CK(hipMemset(flags, 0, chunks * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(timeouts, 0, 4)); CK(hipMemset(wait_ticks, 0, 8));
    for (unsigned delay : {0u, 200u}) {
        hipGraph_t g; hipGraphExec_t e;
        std::vector<hipEvent_t> ev(steps + 2);
        for (auto& x : ev) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
        static unsigned base = 1;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        k_busy<<<256, 256, 0, s1>>>(800u, sink);
        for (int s = 0; s < steps; ++s) {
            const unsigned step = base + s;
            CK(hipEventRecord(ev[s], s1)); CK(hipStreamWaitEvent(s2, ev[s], 0));
            k_write<<<chunks / 4, 256, 0, s2>>>(rows, flags, chunks, step, delay);                               // "K2(s)"
            k_read<<<256, 256, 0, s1>>>(rows, flags, chunks, step, 4, 800u, bad, timeouts, wait_ticks);            // "K1(s+1)" beside it
        }
        base += steps;
        CK(hipEventRecord(ev[steps], s2)); CK(hipStreamWaitEvent(s1, ev[steps], 0));
        CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(e, s1)); CK(hipStreamSynchronize(s1));          // (one replay only: the flags count steps up)
        unsigned hb, ht; unsigned long long hw;
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, timeouts, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hw, wait_ticks, 8, hipMemcpyDeviceToHost));
        printf("hand-over, writer delayed %.1f us: %u stale rows, %u timeouts, mean wait per polled chunk %.2f us (16 steps, 1024 waves x 4 chunks)\n",
               delay / 100.0, hb, ht, hw / 100.0 / (16.0 * 1024 * 4));
        CK(hipMemset(bad, 0, 4)); CK(hipMemset(timeouts, 0, 4)); CK(hipMemset(wait_ticks, 0, 8));
    }
    return 0;
}
