// The launch planner (csrc/nb_plan.cpp: plain host C++) over a grid of sizes, precisions, pinned shapes, split counts, flags and
// shards, built with AddressSanitizer + UndefinedBehaviorSanitizer by tests/test_planner_cpu.py (sanitizers run on the CPU build
// only: there is no GPU ASan on this pool).  Also checks that every plan names a shape that has a kernel.
#include "nb_plan.h"
#include <cstdio>
#include <cstring>
int main() {
    const uint32_t sizes[] = {1, 2, 63, 64, 255, 256, 1000, 1024, 1536, 2048, 3000, 4096, 5000, 8192, 12000, 13000, 14000, 16384, 20000, 32768, 40002, 65536,
                              100000, 262144, 370688, 1048576, 2000000, 2500000, 4194304, 7000000, 1u << 30};
    const uint32_t variants[] = {0, 1, 2, 4, 14, 116, 164, 22, 24, 28, 34, 38, 304014, 308014, 308015, 402644, 502641, 601014, 601018, 601016, 704013, 708013, 708011, 716013, 716011, 708014, 999999, 123456};
    unsigned long long count = 0, syms = 0;
    for (uint32_t n : sizes)
        for (int f64 = 0; f64 < 2; ++f64)
            for (uint32_t v : variants)
                for (uint32_t js : {0u, 1u, 3u, 16u, 200u})
                    for (uint32_t flags : {0u, 4u, 8u, 64u, 128u})
                        for (int shard = 0; shard < 4; ++shard) {
                            nbp::PlanInput in;
                            in.n = n; in.f64 = f64;
                            memset(&in.cfg, 0, sizeof in.cfg);
                            in.cfg.n = n; in.cfg.force_variant = v; in.cfg.jsplit = js; in.cfg.flags = flags;
                            uint32_t g = shard == 0 ? 1 : shard == 1 ? 2 : shard == 2 ? 3 : 8;
                            uint32_t align = shard == 3 ? 1024 : 1;
                            uint32_t rows = (n + g - 1) / g; rows = (rows + align - 1) / align * align;
                            in.sb = shard ? (rows * (g - 1) < n ? rows * (g - 1) : 0) : 0;
                            in.sc = shard ? (n - in.sb < rows ? n - in.sb : rows) : n;
                            if (shard) { in.cfg.shard_begin = in.sb; in.cfg.shard_count = in.sc; }
                            if (in.sc == 0) continue;
                            nbp::LaunchPlan p = nbp::plan_launch(in);
                            ++count; syms += p.sym;
                            if (!nbp::shape_exists(f64, p.sh)) { printf("no kernel for %s\n", p.variant.c_str()); return 1; }
                        }
    printf("planned %llu configurations (%llu symmetric) under ASan + UBSan\n", count, syms);
    return 0;
}
