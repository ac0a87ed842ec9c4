#!/usr/bin/env python3
"""Dump the launch planner's answers (nb_plan_query, no GPU needed) over a grid of sizes, precisions, flags, pinned variants
and shards -- one line each.  Diff two dumps to see what a change to the planner / cost model moved:
    python tools/plan_dump.py > /tmp/before.txt;  <edit, make>;  python tools/plan_dump.py | diff /tmp/before.txt -"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import capi  # noqa: E402

SIZES = [1, 2, 63, 64, 255, 256, 1000, 1024, 1536, 2048, 3000, 4096, 5000, 6000, 7000, 8192, 10000, 12000, 12500, 13000, 14000, 15000,
         16384, 18000, 20000, 22000, 24000, 28000, 32768, 40002, 50000, 65536, 100000, 131072, 200000, 262144, 370688, 524288, 741376,
         1048576, 2000000, 2500000, 4194304]
VARIANTS = [0, 1, 2, 4, 14, 116, 164, 22, 24, 28, 34, 38, 304014, 308014, 308015, 402644, 502641, 601014, 601018, 601016,
            704013, 708013, 708011, 716013, 716011, 708014]


def line(tag, q):
    plan = " ".join("%s=%d" % kv for kv in q.get("plan", {}).items())
    tab = (" tab=%d:%d" % (int(q["tab"][:, 0].sum()), int(q["tab"][:, 1].sum()))) if "tab" in q else ""
    print("%-44s %-44s K%d ipl%d ls%d x%d js=%d jps=%d own=%d+%d layers=%d %s%s" % (
        tag, q["variant"], q["kind"], q["ipl"], q["ls"], q["x"], q["jsplit"], q["j_per_split"], q["own_split0"], q["own_splits"],
        q["sym_layers"], plan, tab))


for n_cu, clock in ((256, 2.4e9), (304, 2.1e9), (64, 2.4e9)):
    for n in SIZES:
        for prec in ("f32", "f64"):
            for flags in (0, 4, 8, 64, 72):
                line("cu%d n=%d %s flags=%d" % (n_cu, n, prec, flags), capi.plan_query(n, prec, flags=flags, n_cu=n_cu, clock_hz=clock))
            if n_cu != 256:
                continue
            for js in (1, 3, 16):
                line("n=%d %s js=%d" % (n, prec, js), capi.plan_query(n, prec, jsplit=js))
            for v in VARIANTS[1:]:
                for js in (0, 2):
                    line("n=%d %s v=%d js=%d" % (n, prec, v, js), capi.plan_query(n, prec, force_variant=v, jsplit=js))
            for g in (2, 3, 8):
                for align in (1, 256, 1024):
                    rows = -(-(-(-n // g)) // align) * align
                    for r in (0, g - 1):
                        b = min(r * rows, n)
                        cnt = min(rows, n - b)
                        if cnt:
                            for flags in (0, 128):
                                line("n=%d %s shard=%d+%d flags=%d" % (n, prec, b, cnt, flags),
                                     capi.plan_query(n, prec, shard=(b, cnt), flags=flags))
