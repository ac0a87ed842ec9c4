#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json:

    pair-interactions/s (and % fp32 roofline) at N=262,144; 1/2/4/8 GPU

One "step" = one pass of the hot path (force kernel + integrate kernel, and for
N>1 the exchange: reduce-scatter of the partial accelerations + all-gather of the
positions) over the whole particle set, inputs already resident in HBM.  Prints
ONE JSON line on rank 0.

    python bench.py                      # 1 GPU, N=262,144 Plummer sphere
    python bench.py --gpus 8             # starts its own 8 ranks (torch.distributed.run child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8     # or under a launcher

Multi-GPU: one process per GPU; rank r owns a contiguous block of rows
 and every rank keeps the full bodies array in HBM.  Two
protocols, both measured by ONE N>1 launch:
  * default (`config.parallelism` = pairshardN+reducescatter+allgather): the
    rank form of the symmetric force pass -- every UNORDERED pair of the system
    is evaluated by exactly one rank (the one whose rows keep it resident), each
    rank adds up what it has for every row, the engine reduce-scatters those
    partial accelerations (in-place ncclReduceScatter, RCCL over xGMI), integrates
    its rows and all-gathers the new positions (in-place ncclAllGather);
  * `also[0]` (ishardN+allgather, NB_FLAG_NO_SYM = --flags 64): the protocol
    north_star spells out -- ordered-pair force pass over the rank's own rows x
    all j, integrate, all-gather of the positions.
--exchange torch routes the all-gather through torch.distributed instead of the
engine's own RCCL calls.  `per_rank` breaks a step into force / sym_reduce /
reduce_scatter / integrate / allgather (nb_step_times2), next to ms_per_step.
Default --scaling strong: the metric is quoted at N=262,144 for every GPU count.

The line is only printed with a value when the run's own correctness checks
pass (a sampled row of the benchmarked launch shape against an fp64 direct sum,
replica agreement across ranks, the N=1,024 golden fixture); otherwise the
value is null and the exit code 1.
"""
import argparse
import json
import math
import os
import socket
import subprocess
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
EPS2 = 1e-4                    # nbody3d.js:234


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
    ap.add_argument("--flags", type=int, default=0, help="nb_config.flags (NB_FLAG_LDS_ONLY = 4, NB_FLAG_NO_FUSE = 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-check", action="store_true", help="skip the N=1,024 fixture check (the sampled-row check of "
                                                            "the benchmarked shape always runs)")
    ap.add_argument("--exchange", default="native", choices=["native", "torch", "host"],
                    help="native: the engine's own in-place ncclAllGather; torch: torch.distributed all-gather through "
                         "the exchange hook; host: REHEARSAL ONLY -- gloo group + host staging so several ranks can "
                         "share one GPU (all ranks use device 0); the JSON line is marked and must not be quoted")
    ap.add_argument("--overlap", action="store_true",
                    help="start the all-gather asynchronously and hide it behind the next step's own-row force work")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed exchange path even with one rank (plumbing test)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the secondary single-GPU measurements (BASELINE configs 2 and 5, config 4's N=1,048,576 on one GPU, the "
                         "reference's default N=40,002 system) that the default 1-GPU headline run appends under the `also` key")
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


def _k2_bytes_per_body(variant_name):
    """Bytes the integrate kernel moves per body: the algorithmic 96 plus the partial sums it adds up --
    16 B per j-split of the ordered-pair kernels, 12 B per resident / traveler layer of the symmetric pass."""
    import re
    m = re.search(r"_r(\d+)t(\d+)(?:_u\d+)?$", variant_name)
    if "sym" in variant_name and m:
        return 96.0 + 12.0 * (int(m.group(1)) + int(m.group(2)))
    js = _jsplit(variant_name)
    return 96.0 if js == 1 else 96.0 + 16.0 * js


def spawn_ranks(args):
    """`bench.py --gpus N` without a launcher: this process stays GPU-free (no HIP call, no torch
    import) and starts the N ranks as a child `python -m torch.distributed.run`; it exits with
    the child's code.  Never exec()s: a process that has touched the GPU must not be replaced."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ---- CPU baseline -------------------------------------------------------------------------------

def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    info["sched_affinity"] = aff
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                info["cgroup_cfs_quota"] = "%g/%g" % (q, per)
                if q > 0:
                    quota = q / per
            break
        except (OSError, ValueError, IndexError):
            continue
    use = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    info["usable"] = use
    for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES"):
        info[k] = os.environ.get(k)
    return info


def cpu_baseline(bodies, G, seconds):
    """The oracle's threaded f32 kernel (kind 'port': the reference has no CPU
    path) on a bounded i-slice of the SAME workload: once on one thread, once on
    every CPU this process can use (affinity mask and cgroup quota, not the host's core count)."""
    from oracle import oracle
    n = bodies.shape[0]
    cpus = usable_cpus()
    nthreads = cpus["usable"]

    def timed(rows, threads):
        t0 = time.perf_counter()
        _, used = oracle.accel_f32_mt(bodies, G, i0=0, i1=rows, nthreads=threads)
        return time.perf_counter() - t0, used

    probe = min(256, n)
    timed(probe, 1)
    t1, _ = timed(probe, 1)                                             # warm, one thread
    rows1 = int(min(n, max(probe, probe * 0.25 * seconds / max(t1, 1e-6))))
    d1, _ = timed(rows1, 1)
    one = rows1 * (n - 1) / d1
    timed(probe, nthreads)
    tm, _ = timed(min(n, probe * nthreads), nthreads)
    rows = int(min(n, max(probe, probe * nthreads * 0.75 * seconds / max(tm, 1e-6))))
    rows = max(256, (rows // 256) * 256) if n >= 256 else n
    dm, used = timed(rows, nthreads)
    allv = rows * (n - 1) / dm
    out = {"value": allv, "unit": "pair-interactions/s", "cores": int(used), "kind": "port",
           "sample": "oracle/nb_oracle.c nbo_accel_f32_mt (f32, AVX clones, OpenMP over i): rows [0,%d) of the "
                     "N=%d workload against all N on %d threads, %.1f s; one thread: rows [0,%d), %.1f s"
                     % (rows, n, used, dm, rows1, d1),
           "one_thread_value": one, "threads_x_one_thread": used * one, "parallel_efficiency": allv / (used * one),
           "cpus": cpus, "cpu_model": _cpu_model()}
    # BASELINE.md §4 'CPU-JS': the single-thread JavaScript restatement on config 1 (N=1,024)
    try:
        import shutil
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


# ---- correctness gates ----------------------------------------------------------------------------

def fixture_check():
    """The N=1,024 Plummer fixture, 100 steps, against the committed fp64 golden vector (data
    files only -- the oracle is not imported here)."""
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
        name = sim.variant
    d = np.abs(b[:, :3].astype(np.float64) - ref[:, :3]).max(1)
    err = float((d / np.maximum(np.sqrt((ref[:, :3] ** 2).sum(1)), man["r_scale"])).max())
    e0 = ke0 + pe0
    return {"fixture": "plummer1024 dt=1e-3 100 steps", "kernel_variant": name, "max_rel_pos_err_vs_f64_oracle": err,
            "tolerance": 1e-4, "energy_drift": abs((ke + pe_prev - e0) / e0), "pass": bool(err < 1e-4)}


def sampled_rows_check(bodies64, accel, rows, G, tol):
    """Accelerations of a few rows of the BENCHMARKED launch shape against an fp64 direct sum
    over all N bodies done here in numpy (nbody3d.js:232-237 arithmetic, j != i)."""
    worst = 0.0
    x = bodies64[:, :3]
    gm = G * bodies64[:, 3]
    for i in rows:
        d = x - x[i]
        r2 = (d * d).sum(1) + EPS2
        w = gm / (r2 * np.sqrt(r2))
        w[i] = 0.0
        ref = (w[:, None] * d).sum(0)
        err = np.abs(accel[i, :3].astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-3)
        worst = max(worst, float(err))
    return {"rows": [int(r) for r in rows], "max_rel_err_vs_fp64_direct_sum": worst, "tolerance": tol,
            "pass": bool(worst < tol)}


def also_measurements(Simulation, ic, device):
    """Secondary measurements appended to the default 1-GPU headline line (same process, same gates): the other
    single-GPU BASELINE configs and the reference's own default workload, each with its own sampled-row check
    against an fp64 direct sum.  ~3 s of GPU time in all; the headline fields are computed before this runs.

      config 2   N=65,536 uniform cube, fp32: the default shape, and the "LDS tile=256" kernel BASELINE names (variant 28)
      config 5   N=262,144 Plummer, fp64
      config 4   its N=1,048,576 system on this one GPU (3 steps), fp32
      default    the reference's UI defaults (index.html:68-74, nbody3d.js:62-64,163-177): 2 galaxies x 20,000 + 2 = N 40,002,
                 G = dt = 1e-4, central masses 1e7 -- built by js/ic.js::galaxies under Node, digest-checked against the
                 reference generator's own output (tests/golden/galaxy40002_params.json)
      UI range   N=13,000 / 16,384 / 20,000 Plummer spheres, fp32: mid sizes of the reference's UI range; 2,048 / 4,096 / 8,192 / 10,000: its low half
      weak g=1   N=370,688: the g = 1 anchor of config 4's weak-scaling series
    """
    out = []

    def run(label, bodies, vel, dt, G, precision, variant, warm, steps, k1_steps):
        n = bodies.shape[0]
        np_dtype = np.float64 if precision == "f64" else np.float32
        roof = PEAK_FP32_TFLOPS * 1e12 / FLOPS_PER_PAIR * (0.5 if precision == "f64" else 1.0)
        entry = {"workload": label, "n": n, "dtype": precision}
        try:
            with Simulation(n, precision=precision, device=device, force_variant=variant) as sim:
                sim.init(bodies.astype(np_dtype), vel.astype(np_dtype))
                sim.set_params(dt, G)
                sim.simulate(1)
                acc = sim.read(bodies=False, vel=False)[2]
                rows = np.unique(np.linspace(0, n - 1, 4).astype(np.int64))
                chk = sampled_rows_check(bodies.astype(np.float64), acc, rows, G, 1e-11 if precision == "f64" else 2e-5)
                sim.simulate(warm)
                sim.sync()
                t0 = time.perf_counter()
                sim.simulate(steps)
                sim.sync()
                wall = time.perf_counter() - t0
                sim.enable_timing(True)
                sim.simulate(k1_steps)
                f_ms, i_ms, _, launches = sim.step_times()
                sim.enable_timing(False)
                rate = n * (n - 1) * steps / wall
                entry.update({"kernel_variant": sim.variant, "ms_per_step": 1e3 * wall / steps, "steps": steps,
                              "pairs_per_s": rate, "frac": rate / roof,
                              "k1_avg_launch_ms": f_ms, "k1_frac": FLOPS_PER_PAIR * n * (n - 1) / (f_ms * 1e-3) / 1e12 /
                              (PEAK_FP32_TFLOPS * (0.5 if precision == "f64" else 1.0)) if f_ms > 0 else None,
                              "max_rel_err_vs_fp64_direct_sum": chk["max_rel_err_vs_fp64_direct_sum"],
                              "tolerance": chk["tolerance"], "pass": chk["pass"]})
                if not chk["pass"]:
                    entry.update({"withheld_frac": entry["frac"], "frac": None, "pairs_per_s": None})
        except Exception as e:      # a secondary measurement never takes the headline down silently: it fails the run
            entry.update({"error": str(e), "pass": False, "frac": None})
        out.append(entry)

    cb, cv = ic.uniform_cube(65536, seed=2)
    run("config 2: N=65536 uniform cube, fp32, default shape", cb, cv, 1e-3, 1.0, "f32", 0, 150, 250, 20)
    run("config 2: N=65536 uniform cube, fp32, LDS tile=256 kernel (variant 28)", cb, cv, 1e-3, 1.0, "f32", 28, 150, 250, 20)
    pb, pv = ic.plummer(N_HEADLINE, seed=1)
    run("config 5: N=262144 Plummer sphere, fp64", pb, pv, 1e-3, 1.0, "f64", 0, 1, 3, 3)
    mb, mv = ic.plummer(1048576, seed=1)
    run("config 4's system on ONE GPU: N=1048576 Plummer sphere, fp32 (the 8-GPU run shards this)", mb, mv, 1e-3, 1.0, "f32", 0, 1, 3, 3)
    del mb, mv
    # config 5's second half: "long-horizon energy conservation vs fp32" -- the same system through both engines, energy sampled on
    # the device (nb_diagnostics: KE after call n with PE before it)
    try:
        hn, hsteps, hevery = 65536, 400, 50
        hb, hv = ic.plummer(hn, seed=1)
        drift, final = {}, {}
        for prec, dt_np in (("f32", np.float32), ("f64", np.float64)):
            with Simulation(hn, precision=prec, device=device) as sim:
                sim.init(hb.astype(dt_np), hv.astype(dt_np))
                sim.set_params(1e-3, 1.0)
                drift[prec] = max(sim.energy_drift(hsteps, hevery))
                final[prec] = sim.read(vel=False, accel=False)[0]
        dpos = np.abs(final["f32"][:, :3].astype(np.float64) - final["f64"][:, :3]).max(1)
        rel = float((dpos / np.maximum(np.sqrt((final["f64"][:, :3] ** 2).sum(1)), 1.0)).max())
        ok = bool(drift["f32"] < 1e-6 and drift["f64"] < 1e-6 and 0.5 * drift["f64"] < drift["f32"] < 2.0 * drift["f64"] and rel < 2e-5)
        out.append({"workload": "config 5 energy horizon: N=65536 Plummer sphere, dt=1e-3, f32 and f64 engines from the same initial conditions",
                    "n": hn, "steps": hsteps, "sampled_every": hevery, "max_dE_rel_f32": drift["f32"], "max_dE_rel_f64": drift["f64"],
                    "max_rel_pos_diff_f32_vs_f64": rel, "tolerance": "both drifts < 1e-6 and within 2x of each other; positions < 2e-5",
                    "pass": ok, "frac": None})
    except Exception as e:
        out.append({"workload": "config 5 energy horizon", "error": str(e), "pass": False, "frac": None})
    try:
        gb, gv, gp = ic.reference_galaxies(os.path.join(ROOT, "tests", "golden", "galaxy40002_params.json"))
        run("reference default: N=40002, 2 galaxies x 20000 + central masses 1e7, G=dt=1e-4 (index.html:68-74), fp32",
            gb, gv, 1e-4, gp["G"], "f32", 0, 400, 700, 40)
    except Exception as e:
        # the generator runs under Node (JavaScript's Math functions are part of the bits): without it the entry is skipped,
        # not failed; a digest mismatch or a failed check does fail the line
        skipped = "node is not installed" in str(e)
        out.append({"workload": "reference default: N=40002 galaxies", "skipped" if skipped else "error": str(e), "pass": skipped, "frac": None})
    # the reference's UI range below its default (index.html:68-74: 1,001 .. 500,010 bodies): three mid sizes where the fixed cost of a
    # step (launch ramp, K1 -> K2 boundary, K2) weighs most -- Plummer spheres, the headline's dt and G
    for un in (13000, 16384, 20000):
        ub, uv = ic.plummer(un, seed=1)
        run("reference UI range: N=%d Plummer sphere, fp32" % un, ub, uv, 1e-3, 1.0, "f32", 0, 600, 1600, 48)
    # ... and its low half (the UI starts at 1,001 bodies per galaxy): one-launch steps, launch- and latency-bound
    for un, usteps in ((2048, 16000), (4096, 12800), (8192, 6400), (10000, 4800)):
        ub, uv = ic.plummer(un, seed=1)
        run("reference UI range, low half: N=%d Plummer sphere, fp32" % un, ub, uv, 1e-3, 1.0, "f32", 0, usteps // 2, usteps, 48)
    # config 4's weak-scaling series (constant pairs per GPU, N_g = 1,048,576 sqrt(g / 8)) has its g = 1 point on this
    # one GPU: `bench.py --scaling weak --gpus 1` runs the same N -- here so that the curve has its anchor in every default line
    wn = weak_n(1)
    wb, wv = ic.plummer(wn, seed=1)
    run("config 4 weak-scaling series, g=1 anchor: N=%d Plummer sphere, fp32 (= bench.py --scaling weak --gpus 1)" % wn, wb, wv, 1e-3, 1.0, "f32", 0, 2, 6, 4)
    return out


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import torch
    from nbody3d_amd import Simulation, capi, ic
    if not os.path.exists(capi.library_path()):
        # fresh checkout: rank 0 compiles the engine (never a CPU fallback); the other ranks wait until it has finished
        # (the process group does not exist yet, so the rendezvous is a stamp file of this launch, written after the build)
        # The stamp is unique to this launch (torchrun's run id + master port + the launcher's pid) and node-local: each
        # node's LOCAL_RANK 0 builds for its node; it is removed before the build and again at exit.
        import atexit
        import tempfile
        launch = "%s_%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MASTER_PORT", "0"), os.getppid())
        stamp = os.path.join(tempfile.gettempdir(), "nb_engine_built_" + "".join(c if c.isalnum() else "_" for c in launch))
        if local_rank == 0:
            if os.path.exists(stamp):
                os.unlink(stamp)
            import __graft_entry__
            __graft_entry__.build()
            open(stamp, "w").write("ok\n")
            atexit.register(lambda: os.path.exists(stamp) and os.unlink(stamp))
        else:
            t_wait = time.time()
            while not (os.path.exists(stamp) and os.path.exists(capi.library_path())):
                if time.time() - t_wait > 900:
                    sys.exit("bench.py: rank %d: the engine library was not built within 15 min" % rank)
                time.sleep(0.5)
    from nbody3d_amd.shard import (ShardPlan, torch_allgather_hook, torch_allgather_overlapped_hooks,
                                   torch_allgather_via_host_hook)

    rehearsal = args.exchange == "host"
    ndev = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if ndev < 1:
        sys.exit("bench.py: no GPU visible -- the engine has no CPU fallback")
    if world > ndev and not rehearsal:
        sys.exit("bench.py: %d ranks need %d GPUs, %d visible (one process per GPU)" % (world, world, ndev))
    if not torch.cuda.is_available() or capi.device_count() < 1:
        sys.exit("bench.py: no GPU visible -- the engine has no CPU fallback")
    if rehearsal:
        local_rank = 0          # every rank shares device 0 (1-GPU development box)
    torch.cuda.set_device(local_rank)
    dist = None
    dist_wanted = world > 1 or args.force_dist
    if dist_wanted:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:      # --force-dist without a launcher: a free port, not a fixed one
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                free_port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port), RANK="0", WORLD_SIZE="1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.n or (N_HEADLINE if args.scaling == "strong" else weak_n(world))
    G, dt = 1.0, 1e-3
    bodies, vel = (ic.plummer(n, seed=1) if args.workload == "plummer" else ic.uniform_cube(n, seed=2))
    np_dtype = np.float64 if args.precision == "f64" else np.float32
    esz = 8 if args.precision == "f64" else 4
    # native exchange: rows in whole super-blocks of 1,024 (f64: 512), so that the ranks can take the rank form of the symmetric force
    # pass (NB_FLAG_SYM_SHARD: each unordered pair evaluated by ONE rank, partial accelerations reduce-scattered by the engine)
    sym_shard = dist_wanted and args.exchange == "native" and not args.variant and not (args.flags & capi.NB_FLAG_NO_SYM)
    plan = ShardPlan(n, world, rank, align=1024 if sym_shard else 256)
    bodies_p, vel_p = plan.pad(bodies.astype(np_dtype)), plan.pad(vel.astype(np_dtype))
    overlap = args.overlap or os.environ.get("NB_OVERLAP") == "1"
    cpu_dev = "cpu" if rehearsal else "cuda"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(flag):
        if dist is None:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], device=cpu_dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def open_sim(flags):
        """The handle of one protocol: (sim, exchange description, torch-owned bodies or None)."""
        kw = dict(precision=args.precision, device=local_rank, force_variant=args.variant, jsplit=args.jsplit, flags=flags)
        exchange = {"kind": "none"}
        if dist is None:
            return Simulation(plan.padded_n, **kw), exchange, None
        sim = None
        if args.exchange == "native":
            # the engine owns stream and buffers; the only thing the host moves is the ncclUniqueId.
            # Every rank must take the SAME path: ncclCommInitRank blocks until all ranks have called it, so the ranks
            # first agree that each of them CAN call it (librccl loads, the handle exists, the partition is the one
            # ncclAllGather needs), then all call it, then agree on the outcome; one failure anywhere sends every rank
            # to the torch path together (a lone rank falling back would leave the others in mismatched collectives).
            why = None
            try:
                my_uid = capi.rccl_unique_id()          # loads librccl on EVERY rank (rank 0's id is the one used)
                sim = Simulation(plan.padded_n, shard=(plan.begin, plan.count),
                                 **dict(kw, flags=flags | (capi.NB_FLAG_SYM_SHARD if sym_shard else 0)))
                if plan.padded_n != world * plan.count or plan.begin != rank * plan.count:
                    raise RuntimeError("partition is not nranks equal row blocks")
            except Exception as e:
                why = "precondition: " + str(e)
            if all_ranks(why is None):
                uid = torch.zeros(capi.NB_RCCL_ID_BYTES, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(my_uid), dtype=torch.uint8))
                dist.broadcast(uid, src=0)
                try:
                    sim.rccl_attach(bytes(uid.cpu().numpy().tobytes()), world, rank, overlap=overlap)
                except Exception as e:
                    why = "attach: " + str(e)
                if not all_ranks(why is None) and why is None:
                    why = "attach failed on another rank"
            elif why is None:
                why = "precondition failed on another rank"
            if why is None:
                nr, rk, ver = sim.rccl_info()
                shp = sim.shape_info()
                rank_form = "symwrank" in sim.variant
                exchange = {"kind": ("rccl-native: in-place ncclReduceScatter of the partial accelerations (rank form of the symmetric pass) + "
                                     if rank_form else "rccl-native ") + "in-place ncclAllGather of the positions on the engine stream" +
                                    (" (overlapped: own-row force work first)" if overlap else ""),
                            "rccl_nranks": nr, "rccl_rank": rk, "rccl_version": ver,
                            "overlap_requested": bool(overlap),
                            # the overlapped form only hides the gather when some of the force work reads the rank's own rows only
                            "overlap_engaged": bool(overlap and shp["own_splits"] > 0),
                            "own_splits": shp["own_splits"], "jsplit": shp["jsplit"]}
                return sim, exchange, None
            if sim is not None:                # keep the run alive on torch's collective -- every rank together -- and say so
                sim.close()
            sim = None
            exchange = {"native_attach_failed": why}
        # torch owns the replicated bodies array so its collective runs on it directly
        stream = torch.cuda.current_stream()
        t_bodies = torch.empty((plan.padded_n, 4), device="cuda", dtype=torch.float64 if args.precision == "f64" else torch.float32)
        sim = Simulation(plan.padded_n, shard=(plan.begin, plan.count), stream=stream.cuda_stream, ext_bodies=t_bodies.data_ptr(), **kw)
        if rehearsal:
            sim.set_exchange(torch_allgather_via_host_hook(t_bodies, plan))
            exchange["kind"] = "REHEARSAL host-staged gloo"
        elif overlap:
            sim.set_exchange_overlapped(*torch_allgather_overlapped_hooks(t_bodies, plan))
            exchange["kind"] = "torch.distributed all_gather_into_tensor (nccl = RCCL), async, via exchange hooks"
            shp = sim.shape_info()
            exchange.update({"overlap_requested": True, "overlap_engaged": shp["own_splits"] > 0,
                             "own_splits": shp["own_splits"], "jsplit": shp["jsplit"]})
        else:
            sim.set_exchange(torch_allgather_hook(t_bodies, plan))
            exchange["kind"] = "torch.distributed all_gather_into_tensor (nccl = RCCL) via exchange hook"
        return sim, exchange, t_bodies

    def protocol_name(variant):
        if world == 1 and dist is None:
            return "1gpu"
        if "symwrank" in variant:       # each unordered pair evaluated by ONE rank; partial accelerations reduce-scattered, positions all-gathered
            return "pairshard%d+reducescatter+allgather%s" % (world, "(overlapped)" if overlap else "")
        return "ishard%d+allgather%s" % (world, "(overlapped)" if overlap else "")

    def measure(flags, want_energy):
        """One protocol, start to finish: handle, gates, warm-up, the timed steps, the per-part breakdown.  Returns (result, sim)
        with the handle still open (the caller closes it)."""
        sim, exchange, t_bodies = open_sim(flags)
        sim._bench_keepalive = t_bodies          # torch-owned position array (exchange through torch): lives as long as the handle
        sim.init(bodies_p, vel_p)
        sim.set_params(dt, G)
        res = {"variant": sim.variant, "exchange": exchange, "e_start": None}
        if want_energy:
            ke0, pe0, _ = sim.diagnostics()          # fp64 on the device; outside the timed region
            res["e_start"] = ke0 + pe0
        # gate 1: the benchmarked launch shape itself -- accelerations of the first step, a few rows of
        # this rank's shard, against an fp64 direct sum (done in numpy here; not the oracle)
        sim.simulate(1)
        acc = sim.read(bodies=False, vel=False)[2]
        lo, hi = plan.begin, min(plan.begin + plan.count, n)
        rows = np.unique(np.linspace(lo, hi - 1, 6).astype(np.int64)) if hi > lo else np.array([], np.int64)
        shape_check = sampled_rows_check(bodies.astype(np.float64), acc, rows, G, 1e-11 if args.precision == "f64" else 2e-5)
        shape_check["kernel_variant"] = sim.variant
        res["shape_check"] = shape_check
        res["shape_ok"] = all_ranks(shape_check["pass"])

        sim.simulate(max(args.warmup - 1, 0))
        barrier()
        # The timed region: EXACTLY args.steps steps between two barriers, per-kernel event timing OFF -- the regime a user gets
        # (multi-step calls on the engine's own stream replay a captured graph), and the one every `also` entry is timed in.
        t0 = time.perf_counter()
        sim.simulate(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        # ... and a separate leg of the same number of steps with the kernels stamped at their own begin / end (HIP events on the
        # engine's stream: plain launches, no graph) for roofline.avg_launch_ms and the per-part breakdown; outside the timed region
        sim.enable_timing(True)
        sim.simulate(args.steps)
        barrier()
        parts = sim.step_breakdown()
        sim.enable_timing(False)
        if dist is not None:
            t = torch.tensor([elapsed], device=cpu_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        res["elapsed"], res["parts"] = elapsed, parts
        ms_step = 1e3 * elapsed / args.steps

        # gate 2 (distributed runs): every rank must hold the same replicated position array
        res["replicas_ok"], res["replica_check"] = True, None
        if dist is not None:
            mine = sim.read(vel=False, accel=False)[0]
            digest = float(np.abs(mine[:n, :3].astype(np.float64)).sum())
            t = torch.tensor([digest, -digest], device=cpu_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res["replicas_ok"] = bool(t[0].item() == -t[1].item()) and bool(np.isfinite(digest))
            res["replica_check"] = {"abs_sum_of_positions_max_over_ranks": float(t[0].item()),
                                    "abs_sum_of_positions_min_over_ranks": float(-t[1].item()), "pass": res["replicas_ok"]}
            # what the collectives move, per rank and step (payload; a ring forwards each block g - 1 times):
            #   all-gather: this rank's new position rows go to every other rank;
            #   reduce-scatter (rank form): g - 1 of the g row blocks of this rank's partial-acceleration array go out, one block comes back summed
            blk = int(plan.count * 4 * esz)
            rank_form = "symwrank" in sim.variant
            timed_ag = parts["allgathers"] > 0
            exchange.update({
                "allgather_ms": parts["allgather_ms"] if timed_ag else None,
                "reduce_scatter_ms": parts["reduce_scatter_ms"] if rank_form else None,
                "avg_ms": parts["allgather_ms"] + parts["reduce_scatter_ms"],            # every native collective of a step
                "allgather_bytes_sent_per_rank": blk,
                "reduce_scatter_bytes_sent_per_rank": blk * (world - 1) if rank_form else 0,
                "bytes_sent_per_rank": blk + (blk * (world - 1) if rank_form else 0),
                "note": None if timed_ag else "the overlapped all-gather runs on its own stream and is not timed by the engine"})
            names = ["force_ms", "sym_reduce_ms", "reduce_scatter_ms", "integrate_ms", "allgather_ms", "span_ms"]
            mine_parts = [parts[k] for k in names]
            tmax = torch.tensor(mine_parts, device=cpu_dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            total = sum(mine_parts[:5])
            res["per_rank"] = {"rank": rank, "rows": plan.count,
                               "force_kernel_avg_ms": parts["force_ms"], "sym_reduce_kernel_avg_ms": parts["sym_reduce_ms"],
                               "reduce_scatter_avg_ms": parts["reduce_scatter_ms"], "integrate_kernel_avg_ms": parts["integrate_ms"],
                               "allgather_avg_ms": parts["allgather_ms"], "exchange_avg_ms": parts["allgather_ms"] + parts["reduce_scatter_ms"],
                               "sum_of_parts_ms": total, "span_ms": parts["span_ms"], "ms_per_step": ms_step,
                               # the parts and the span come from the event-timed leg, ms_per_step from the wall-timed one before it
                               "sum_of_parts_over_span": total / parts["span_ms"] if parts["span_ms"] > 0 else None,
                               "sum_of_parts_over_ms_per_step": total / ms_step if ms_step > 0 else None,
                               "max_over_ranks": dict(zip(names, [float(v) for v in tmax.tolist()]))}
        if rehearsal and rank == 0:   # the sharded state must equal an unsharded run of the same number of steps
            got = sim.read(vel=False, accel=False)[0][:n]
            with Simulation(n, precision=args.precision, device=local_rank) as ref:
                ref.init(bodies.astype(np_dtype), vel.astype(np_dtype))
                ref.simulate(max(args.warmup, 1) + 2 * args.steps, dt, G)      # gate + warm-up, the wall-timed leg, the event-timed leg
                want = ref.read(vel=False, accel=False)[0]
            res["rehearsal_max_rel_diff_vs_unsharded"] = float(np.abs(got[:, :3] - want[:, :3]).max() / np.abs(want[:, :3]).max())
        return res, sim

    pairs_step = n * (n - 1)
    roof_pairs = PEAK_FP32_TFLOPS * 1e12 / FLOPS_PER_PAIR * (0.5 if args.precision == "f64" else 1.0)
    main_res, sim = measure(args.flags, world == 1 and dist is None)
    elapsed, parts, variant = main_res["elapsed"], main_res["parts"], main_res["variant"]
    f_ms, i_ms, launches = parts["force_ms"], parts["integrate_ms"], parts["launches"]
    value = pairs_step * args.steps / elapsed
    out = {
        "metric": "pair-interactions/s", "value": value, "unit": "pair-interactions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "N=%d %s, dt=1e-3, G=1, eps2=1e-4, %s over %d GPU(s)" % (
            n, "Plummer sphere" if args.workload == "plummer" else "uniform cube",
            "unordered pairs sharded by the rows that keep them resident" if "symwrank" in variant else "i-sharded", world),
            "n": n, "kernel_variant": variant, "parallelism": protocol_name(variant)},
        "frac_of_fp32_roofline": value / (roof_pairs * world),
        "shape_check": main_res["shape_check"],
    }
    if dist is not None:
        out["exchange"] = main_res["exchange"]
        out["per_rank"] = main_res["per_rank"]
        out["replica_check"] = main_res["replica_check"]
    if rehearsal:
        out["REHEARSAL"] = "ranks share one GPU, exchange staged through host memory over gloo: not a result"
        if "rehearsal_max_rel_diff_vs_unsharded" in main_res:
            out["rehearsal_max_rel_diff_vs_unsharded"] = main_res["rehearsal_max_rel_diff_vs_unsharded"]
    if launches:
        # K1 on THIS rank: algorithmic flops of one launch / measured launch time.  The rank form divides the system's
        # UNORDERED pairs equally over the ranks: a rank's launch delivers 1/world of the N(N-1) ordered interactions.
        flops_launch = FLOPS_PER_PAIR * pairs_step / world if "symwrank" in variant else (
            FLOPS_PER_PAIR * plan.count * (n - 1) if world > 1 else FLOPS_PER_PAIR * pairs_step)
        peak = PEAK_FP32_TFLOPS * (0.5 if args.precision == "f64" else 1.0)
        achieved = flops_launch / (f_ms * 1e-3) / 1e12
        # HBM bytes per launch come from the PMC profile of the SAME kernel variant (separate
        # rocprofv3 --pmc passes, tools/gpu_prof.sh); any other shape reports null
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "k1_hbm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if (tj.get("n") == n and tj.get("n_gpus", 1) == world and tj.get("dtype") == args.precision and
                    tj.get("kernel_variant") == variant):
                traffic, traffic_src = tj.get("bytes_per_launch"), "profiles/k1_hbm_traffic.json (" + tj.get("source", "") + ")"
        fused = "fused" in variant
        kname = ("nb_force_symw" if "symw" in variant else "nb_force_sym" if "_sym_" in variant
                 else "nb_step_jpk" if "jpairs" in variant else "nb_step_direct" if "fused_regs" in variant else "nb_step_fused" if fused else "nb_force")
        out["roofline"] = {"kernel": kname + "<%s> (%s)" % (args.precision, variant),
                           "bound": "valu",
                           "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                           "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": f_ms, "launches": launches,
                           "timing": "HIP events stamped at the kernel's own begin / end on the engine's stream, over a separate leg of "
                                     "%d steps right after the wall-timed ones (value / ms_per_step come from the wall-timed leg: graph "
                                     "replay, no per-kernel events)" % args.steps,
                           "flops_per_pair": FLOPS_PER_PAIR,
                           "note": "compute-bound on the fp32 vector-FMA rate (157.3 TFLOP/s spec, equal to the "
                                   "dense f32 MFMA peak); not HBM and not MFMA: rsqrt-bound scalar FMA.  achieved = 20 flop x "
                                   "N(N-1) ORDERED pair interactions (SURVEY.md §8(d)) / launch time" +
                                   ("; the symmetric pass delivers them by evaluating each unordered pair once (r, r^2, the cube and "
                                    "the rsqrt are shared; both accelerations accumulated), 18 packed/transcendental issue slots + "
                                    "rotation per TWO interactions instead of 16 per one" if "sym" in variant else ""),
                           "integrate_kernel_avg_ms": i_ms}
        if i_ms > 0:
            # algorithmic 96 B per body; "moved" adds the jsplit partials K2 sums.
            # At this size the state is cache-resident: the HBM figure of K2 is tools/k2_hbm.py (N >= 4M)
            moved = _k2_bytes_per_body(variant)
            out["roofline"]["integrate_kernel_GBps"] = 96.0 * plan.count / (i_ms * 1e-3) / 1e9
            out["roofline"]["integrate_kernel_GBps_moved"] = moved * plan.count / (i_ms * 1e-3) / 1e9
            out["roofline"]["integrate_bytes_moved_over_algorithmic"] = moved / 96.0
    if main_res["e_start"] is not None:
        # total-energy drift of THIS run (north_star: "with total-energy drift reported"): one extra
        # untimed step so that KE(vel after call n) pairs with PE(positions before call n)
        _, pe_prev, _ = sim.diagnostics()
        sim.step()
        ke, _, _ = sim.diagnostics()
        out["energy_drift_over_run"] = {"steps": max(args.warmup, 1) + 2 * args.steps + 1,
                                        "dE_rel": abs((ke + pe_prev - main_res["e_start"]) / main_res["e_start"])}
    sim.close()
    ok = main_res["shape_ok"] and main_res["replicas_ok"]
    if dist is not None and "symwrank" in variant and not rehearsal:
        # The same launch also measures the protocol north_star spells out -- i-sharded ordered-pair force pass + per-step
        # all-gather of the positions (NB_FLAG_NO_SYM) -- so that both are on record from ONE run: same ranks, same
        # workload, same gates; its own communicator.
        lit, sim2 = measure(args.flags | capi.NB_FLAG_NO_SYM, False)
        sim2.close()
        lval = pairs_step * args.steps / lit["elapsed"]
        lp = lit["parts"]
        out["also"] = [{"workload": "the same system through the north_star-literal protocol: i-shard + per-step RCCL all-gather(positions)",
                        "config": {"kernel_variant": lit["variant"], "parallelism": protocol_name(lit["variant"]), "n": n},
                        "value": lval if lit["shape_ok"] and lit["replicas_ok"] else None, "unit": "pair-interactions/s",
                        "ms_per_step": 1e3 * lit["elapsed"] / args.steps, "steps": args.steps,
                        "frac_of_fp32_roofline": lval / (roof_pairs * world),
                        "k1_frac": (FLOPS_PER_PAIR * plan.count * (n - 1) / (lp["force_ms"] * 1e-3) / 1e12 /
                                    (PEAK_FP32_TFLOPS * (0.5 if args.precision == "f64" else 1.0))) if lp["force_ms"] > 0 else None,
                        "exchange": lit["exchange"], "per_rank": lit["per_rank"], "shape_check": lit["shape_check"],
                        "replica_check": lit["replica_check"], "pass": bool(lit["shape_ok"] and lit["replicas_ok"])}]
        ok = ok and out["also"][0]["pass"]
    if rank == 0 and not args.no_check:
        out["check"] = fixture_check()
        ok = ok and out["check"]["pass"]
    default_run = (world == 1 and dist is None and not args.n and args.workload == "plummer" and args.precision == "f32"
                   and not args.variant and not args.jsplit and not args.flags)
    if default_run and not args.no_also:
        out["also"] = also_measurements(Simulation, ic, local_rank)
        ok = ok and all(e.get("pass") for e in out["also"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(bodies, G, args.cpu_seconds)
    if dist is not None:
        ok = all_ranks(ok)
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        out["invalid"] = "a correctness check failed in this run: the throughput is withheld"
        out["withheld_value"] = out["value"]
        out["value"] = None
    if rank == 0:
        print(json.dumps(out))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
