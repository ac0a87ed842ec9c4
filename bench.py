#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json:

    pair-interactions/s (and % fp32 roofline) at N=262,144; 1/2/4/8 GPU

One "step" = one pass of the hot path (force kernel + integrate kernel, and for
N>1 the position all-gather) over the whole particle set, inputs already
resident in HBM.  Prints ONE JSON line on rank 0.

    python bench.py                      # 1 GPU, N=262,144 Plummer sphere
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8

Multi-GPU: one process per GPU; the i-bodies are sharded (rank r owns a
contiguous row block); every rank keeps the full bodies array
in HBM and the ranks all-gather their new rows each step through
torch.distributed (backend nccl = RCCL over xGMI).  Default --scaling strong:
the metric is quoted at N=262,144 for every GPU count.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "nbody3d-webgpu_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOPS_PER_PAIR = 20            # literal op count of nbody3d.js:233-236,266
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md:41 'Peak FP32 (vector)' (spec)
N_HEADLINE = 262144            # BASELINE.json metric / configs[2]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nbodies", dest="n", type=int, default=0,
                    help="override N (default 262144); not spelled --n: torch.distributed.run would claim it")
    ap.add_argument("--workload", default="plummer", choices=["plummer", "cube"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--jsplit", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--exchange", default="nccl", choices=["nccl", "host"],
                    help="host: REHEARSAL ONLY -- gloo group + host staging so several ranks can share one GPU "
                         "(all ranks use device 0); the JSON line is marked and must not be quoted as a result")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the torch.distributed/RCCL exchange path even with one rank (plumbing test)")
    return ap.parse_args()


def weak_n(g):
    """BASELINE.json config 4 weak-scaling series: constant pairs per GPU,
    anchored at N = 1,048,576 on 8 GPUs: N_g = 1,048,576 * sqrt(g/8), rounded to 256*g rows
    -> 370,688 / 524,288 / 741,376 / 1,048,576 for g = 1 / 2 / 4 / 8."""
    q = 256 * g
    return int(round(1048576 * math.sqrt(g / 8.0) / q)) * q


def _jsplit(variant_name):
    try:
        return int(variant_name.rsplit("_js", 1)[1])
    except (IndexError, ValueError):
        return 1


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(bodies, G, seconds):
    """The oracle's threaded f32 kernel (kind 'port': the reference has no CPU
    path) on a bounded i-slice of the SAME workload."""
    from oracle import oracle
    n = bodies.shape[0]
    t0 = time.perf_counter()
    _, used = oracle.accel_f32_mt(bodies, G, i0=0, i1=min(256, n))
    t_probe = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, used = oracle.accel_f32_mt(bodies, G, i0=0, i1=min(256, n))   # warm
    t_probe = min(t_probe, time.perf_counter() - t0)
    rows = int(min(n, max(256, 256 * seconds / max(t_probe, 1e-6))))
    rows = max(256, (rows // 256) * 256) if n >= 256 else n
    t0 = time.perf_counter()
    _, used = oracle.accel_f32_mt(bodies, G, i0=0, i1=rows)
    dt = time.perf_counter() - t0
    out = {"value": rows * (n - 1) / dt, "unit": "pair-interactions/s", "cores": int(used), "kind": "port",
           "sample": "oracle/nb_oracle.c nbo_accel_f32_mt (f32, AVX clones, OpenMP over i): rows [0,%d) of the "
                     "N=%d workload against all N, %.1f s" % (rows, n, dt),
           "host_cpus": os.cpu_count(), "cpu_model": _cpu_model()}
    # BASELINE.md §4 'CPU-JS': the single-thread JavaScript restatement on config 1 (N=1,024)
    try:
        import shutil
        import subprocess
        node = shutil.which("node")
        if node:
            p = subprocess.run([node, os.path.join(ROOT, "oracle", "js_baseline.js"), "40"], capture_output=True,
                               text=True, timeout=120)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode == 0 and line:
                out["js_single_thread"] = json.loads(line[-1])
    except Exception as e:  # the JS figure is informative; never fail the bench over it
        out["js_single_thread"] = {"error": str(e)}
    return out


def fixture_check():
    """Correctness gate in the same run: the N=1,024 Plummer
    fixture, 100 steps, against the committed fp64 golden vector (data files
    only -- the oracle is not imported here)."""
    from nbody3d_amd import Simulation
    g = os.path.join(ROOT, "tests", "golden")
    man = json.load(open(os.path.join(g, "manifest.json")))["plummer1024"]
    b0 = np.fromfile(os.path.join(g, "plummer1024_bodies0.f32"), "<f4").reshape(-1, 4)
    v0 = np.fromfile(os.path.join(g, "plummer1024_vel0.f32"), "<f4").reshape(-1, 4)
    ref = np.fromfile(os.path.join(g, "plummer1024_s100_bodies.f64"), "<f8").reshape(-1, 4)
    with Simulation(1024) as sim:
        sim.init(b0, v0)
        sim.set_params(man["dt"], man["G"])
        ke0, pe0, _ = sim.diagnostics()
        sim.simulate(99)
        _, pe_prev, _ = sim.diagnostics()      # PE of positions before the last call
        sim.step()
        ke, _, _ = sim.diagnostics()           # KE of vel after it
        b = sim.read(vel=False, accel=False)[0]
    d = np.abs(b[:, :3].astype(np.float64) - ref[:, :3]).max(1)
    err = float((d / np.maximum(np.sqrt((ref[:, :3] ** 2).sum(1)), man["r_scale"])).max())
    e0 = ke0 + pe0
    return {"fixture": "plummer1024 dt=1e-3 100 steps", "max_rel_pos_err_vs_f64_oracle": err, "tolerance": 1e-4,
            "energy_drift": abs((ke + pe_prev - e0) / e0), "pass": bool(err < 1e-4)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs a torch.distributed.run launch (one process per GPU)" % args.gpus)
        args.gpus = world

    import torch
    from nbody3d_amd import Simulation, capi, ic
    if not os.path.exists(capi.library_path()) and rank == 0:
        import __graft_entry__
        __graft_entry__.build()       # fresh checkout: compile the engine (never a CPU fallback)
    from nbody3d_amd.shard import (ShardPlan, torch_allgather_hook, torch_allgather_overlapped_hooks,
                                   torch_allgather_via_host_hook)

    if not torch.cuda.is_available() or capi.device_count() < 1:
        sys.exit("bench.py: no GPU visible -- the engine has no CPU fallback")
    rehearsal = args.exchange == "host"
    if rehearsal:
        local_rank = 0          # every rank shares device 0 (1-GPU development box)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:      # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.n or (N_HEADLINE if args.scaling == "strong" else weak_n(world))
    G, dt = 1.0, 1e-3
    bodies, vel = (ic.plummer(n, seed=1) if args.workload == "plummer" else ic.uniform_cube(n, seed=2))
    np_dtype = np.float64 if args.precision == "f64" else np.float32
    plan = ShardPlan(n, world, rank)
    bodies_p, vel_p = plan.pad(bodies.astype(np_dtype)), plan.pad(vel.astype(np_dtype))

    stream = torch.cuda.current_stream()
    kw = dict(precision=args.precision, device=local_rank, force_variant=args.variant, jsplit=args.jsplit)
    if dist is not None:
        # torch owns the replicated bodies array so the collective runs on it directly
        t_bodies = torch.empty((plan.padded_n, 4), device="cuda",
                               dtype=torch.float64 if args.precision == "f64" else torch.float32)
        sim = Simulation(plan.padded_n, shard=(plan.begin, plan.count), stream=stream.cuda_stream,
                         ext_bodies=t_bodies.data_ptr(), **kw)
        if rehearsal:
            sim.set_exchange(torch_allgather_via_host_hook(t_bodies, plan))
        elif os.environ.get("NB_OVERLAP") == "1":
            # opt-in: all-gather of step n issued async, waited for only after the own-rows force
            # splits of step n+1 (bit-identical results; not measurable on the 1-GPU dev box, so off by default)
            sim.set_exchange_overlapped(*torch_allgather_overlapped_hooks(t_bodies, plan))
        else:
            sim.set_exchange(torch_allgather_hook(t_bodies, plan))
    else:
        sim = Simulation(plan.padded_n, stream=stream.cuda_stream, **kw)
    sim.init(bodies_p, vel_p)
    sim.set_params(dt, G)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    e_start = None
    if world == 1 and dist is None and not args.no_check:
        ke0, pe0, _ = sim.diagnostics()          # fp64 on the device; outside the timed region
        e_start = ke0 + pe0
    sim.simulate(args.warmup)
    barrier()
    sim.enable_timing(True)
    t0 = time.perf_counter()
    sim.simulate(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    f_ms, i_ms, launches = sim.kernel_times()
    sim.enable_timing(False)

    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs_step = n * (n - 1)
    value = pairs_step * args.steps / elapsed
    roof_pairs = PEAK_FP32_TFLOPS * 1e12 / FLOPS_PER_PAIR * (0.5 if args.precision == "f64" else 1.0)
    out = {
        "metric": "pair-interactions/s", "value": value, "unit": "pair-interactions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "N=%d %s, dt=1e-3, G=1, eps2=1e-4, i-sharded over %d GPU(s)" % (
            n, "Plummer sphere" if args.workload == "plummer" else "uniform cube", world),
            "n": n, "kernel_variant": sim.variant, "parallelism": ("ishard%d+allgather%s" % (world, "(overlapped)" if os.environ.get("NB_OVERLAP") == "1" else ""))
                   if world > 1 else "1gpu"},
        "frac_of_fp32_roofline": value / (roof_pairs * world),
    }
    if rehearsal:
        out["REHEARSAL"] = "ranks share one GPU, exchange staged through host memory over gloo: not a result"
        if rank == 0:   # the sharded state must equal an unsharded run of the same number of steps
            got = sim.read(vel=False, accel=False)[0][:n]
            with Simulation(n, precision=args.precision, device=local_rank) as ref:
                ref.init(bodies.astype(np_dtype), vel.astype(np_dtype))
                ref.simulate(args.warmup + args.steps, dt, G)
                want = ref.read(vel=False, accel=False)[0]
            out["rehearsal_max_rel_diff_vs_unsharded"] = float(
                np.abs(got[:, :3] - want[:, :3]).max() / np.abs(want[:, :3]).max())
    if launches:
        # K1 on THIS rank: algorithmic flops of one launch / measured launch time
        flops_launch = FLOPS_PER_PAIR * plan.count * (n - 1) if world > 1 else FLOPS_PER_PAIR * pairs_step
        peak = PEAK_FP32_TFLOPS * (0.5 if args.precision == "f64" else 1.0)
        achieved = flops_launch / (f_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k1_hbm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("n") == n and tj.get("n_gpus", 1) == world and tj.get("dtype") == args.precision:
                traffic = tj.get("bytes_per_launch")
        out["roofline"] = {"kernel": "nb_force<%s> (%s)" % (args.precision, sim.variant), "bound": "valu",
                           "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                           "traffic": traffic, "avg_launch_ms": f_ms, "launches": launches,
                           "flops_per_pair": FLOPS_PER_PAIR,
                           "note": "compute-bound on the fp32 vector-FMA rate (157.3 TFLOP/s spec, equal to the "
                                   "dense f32 MFMA peak); not HBM and not MFMA: rsqrt-bound scalar FMA",
                           "integrate_kernel_avg_ms": i_ms,
                           # algorithmic 96 B per body; "moved" adds the jsplit partials K2 sums
                           "integrate_kernel_GBps": 96.0 * plan.count / (i_ms * 1e-3) / 1e9 if i_ms > 0 else None,
                           "integrate_kernel_GBps_moved": (96.0 + 16.0 * _jsplit(sim.variant)) * plan.count /
                                                          (i_ms * 1e-3) / 1e9 if i_ms > 0 else None}
    if e_start is not None:
        # total-energy drift of THIS run (north_star: "with total-energy drift reported"): one extra
        # untimed step so that KE(vel after call n) pairs with PE(positions before call n)
        _, pe_prev, _ = sim.diagnostics()
        sim.step()
        ke, _, _ = sim.diagnostics()
        out["energy_drift_over_run"] = {"steps": args.warmup + args.steps + 1,
                                        "dE_rel": abs((ke + pe_prev - e_start) / e_start)}
    if rank == 0 and world == 1 and not args.no_check:
        out["check"] = fixture_check()
    sim.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(bodies, G, args.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
