"""GPU tests of the multi-GPU machinery on the one GPU a test box has, through the C ABI: nb_multi in RCCL mode and nb_rccl_attach with
one rank (the partition / offset logic is covered by the virtual-shard tests here and the gloo tests in test_shard_gloo.py),
bench.py's distributed path, virtual shards and the overlapped exchange with the planner's own split counts (BASELINE config 4's
shard shapes).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden32, load_golden64, rel_pos_err
from oracle import oracle
from nbody3d_amd import MultiSimulation, Simulation, capi, ic

pytestmark = pytest.mark.gpu

TOL_ACC, TOL_TIGHT = 2e-5, 2e-5
TOL_F64 = 1e-12


def run(b, v, steps, dt=1e-3, G=1.0, **kw):
    with Simulation(b.shape[0], **kw) as sim:
        sim.init(b, v)
        sim.simulate(steps, dt, G)
        return sim.read() + (sim.variant,)


# ---- native RCCL ------------------------------------------------------------------------------

def test_multi_handle_rccl_mode_one_device():
    """nb_multi in NB_MULTI_RCCL mode with one shard on the one GPU of the box: ncclCommInitAll,
    ncclGroupStart / in-place ncclAllGather / ncclGroupEnd every step -- the calls an 8-GPU
    node makes -- bit-identical to the peer-copy mode and to a plain handle."""
    n, steps = 4096, 6
    b, v = ic.plummer(n, seed=51)
    kw = dict(force_variant=28, jsplit=4)
    with Simulation(n, **kw) as one:
        one.init(b, v)
        one.simulate(steps, 1e-3, 1.0)
        ref = one.read()
    with MultiSimulation(n, 1, collective="rccl", **kw) as ms:
        info = ms.collective_info()
        assert info["mode"] == "rccl" and info["nranks"] == 1 and info["rccl_version"] > 20000, info
        ms.init(b, v)
        ms.simulate(steps, 1e-3, 1.0)
        got = ms.read()
        ms.set_collective("peer")
        assert ms.collective_info() == {"mode": "peer", "nranks": 0, "rccl_version": 0}
        ms.simulate(2)
        ms.set_collective("rccl")
        ms.simulate(2)
        later = ms.read()
    for x, y in zip(got, ref):
        assert x.tobytes() == y.tobytes()
    rb, _, _ = oracle.run_f32(b, v, None, 1e-3, 1.0, steps + 4)
    assert rel_pos_err(later[0], rb, 1.0) < 1e-6


def test_multi_handle_rccl_mode_refuses_shared_devices():
    b, v = ic.plummer(1024, seed=52)
    with MultiSimulation(1024, 2) as ms:             # two shards on the one GPU: fine for peer copies
        with pytest.raises(Exception) as e:
            ms.set_collective("rccl")
        assert "own device" in str(e.value)
        ms.init(b, v)
        ms.simulate(3, 1e-3, 1.0)                    # still usable in peer mode
        rb, _, _ = oracle.run_f32(b, v, None, 1e-3, 1.0, 3)
        assert rel_pos_err(ms.read()[0], rb, 1.0) < 1e-6


@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_attach_single_rank(overlap):
    """nb_rccl_attach: the engine's own in-place ncclAllGather after every integrate kernel
    (one process per GPU).  One rank here; results equal the handle without a communicator."""
    n, steps = 8192, 5
    b, v = ic.plummer(n, seed=53)
    kw = dict(force_variant=308014, jsplit=4, flags=capi.NB_FLAG_NO_FUSE)
    with Simulation(n, **kw) as one:
        one.init(b, v)
        one.simulate(steps, 1e-3, 1.0)
        ref = one.read()
    with Simulation(n, shard=(0, n), **kw) as sim:
        uid = capi.rccl_unique_id()
        assert len(uid) == 128 and any(uid)
        sim.rccl_attach(uid, 1, 0, overlap=overlap)
        nranks, rank, ver = sim.rccl_info()
        assert (nranks, rank) == (1, 0) and ver > 20000
        with pytest.raises(Exception):
            sim.set_exchange(lambda *a: 0)           # hook and native collective are exclusive
        sim.init(b, v)
        sim.enable_timing(True)
        sim.simulate(steps, 1e-3, 1.0)
        f_ms, i_ms, x_ms, launches = sim.step_times()
        got = sim.read()
        assert launches == steps and f_ms > 0 and i_ms > 0
        if not overlap:
            assert x_ms > 0
        sim.rccl_detach()
        assert sim.rccl_info() == (0, 0, 0)
    for x, y in zip(got, ref):
        assert x.tobytes() == y.tobytes()


def test_rccl_attach_checks_the_partition():
    with Simulation(1024, shard=(256, 256)) as sim:
        uid = capi.rccl_unique_id()
        with pytest.raises(Exception) as e:
            sim.rccl_attach(uid, 1, 0)               # 1 rank must own all rows
        assert "NB_ERR_INVALID" in str(e.value)


def _bench(*args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                       timeout=timeout, env=e)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("exchange", ["native", "torch"])
def test_bench_distributed_path_with_one_rank(exchange):
    """bench.py --force-dist: process group on the nccl (= RCCL) backend, sharded handle on torch's
    stream, per-step all-gather (the engine's own ncclAllGather, or torch's through the hook)."""
    p, out = _bench("--force-dist", "--exchange", exchange, "--nbodies", "16384", "--steps", "4", "--warmup", "1",
                    "--no-cpu-baseline")
    assert p.returncode == 0 and out, p.stderr[-2000:]
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["exchange"]["kind"].startswith("rccl-native" if exchange == "native" else "torch"), out["exchange"]
    if exchange == "native":
        assert out["exchange"]["rccl_nranks"] == 1 and out["exchange"]["avg_ms"] > 0
    assert out["check"]["pass"], out["check"]


def test_bench_multi_gpu_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the GPU-free parent starts the two ranks
    itself.  On a one-GPU box they must get as far as the device count and fail there."""
    p, out = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", timeout=300)
    if capi.device_count() >= 2:
        assert p.returncode == 0 and out and out["n_gpus"] == 2
    else:
        assert p.returncode != 0 and out is None
        assert "2 ranks need 2 GPUs" in (p.stderr + p.stdout), (p.stderr + p.stdout)[-1500:]


def test_bench_two_rank_control_flow_rehearsal():
    """The complete multi-rank flow of bench.py -- self-started ranks, shard plan, per-rank gates,
    replica agreement, max-over-ranks timing -- with two ranks SHARING the one GPU (exchange staged
    through host memory over gloo: RCCL refuses two ranks on one device).  Marked REHEARSAL in the
    line; the sharded state must equal an unsharded run."""
    p, out = _bench("--gpus", "2", "--exchange", "host", "--nbodies", "16384", "--steps", "3", "--warmup", "2",
                    "--no-cpu-baseline", timeout=600)
    assert p.returncode == 0 and out, (p.stderr + p.stdout)[-2000:]
    assert out["n_gpus"] == 2 and "REHEARSAL" in out and out["value"] > 0
    assert out["replica_check"]["pass"] and out["shape_check"]["pass"] and out["check"]["pass"]
    assert out["rehearsal_max_rel_diff_vs_unsharded"] < 1e-6
    assert out["per_rank"]["rows"] == 8192 and out["exchange"]["bytes_sent_per_rank"] == 8192 * 16


# ---- the overlapped exchange with the model's own split count --------------------------------------

@pytest.mark.parametrize("n,g", [(262144, 2), (262144, 4), (262144, 8), (1048576, 8), (40448, 2)])
def test_model_chosen_split_count_leaves_own_row_splits_for_every_rank(n, g):
    """BASELINE config 4's shard shapes (strong scaling at N=262,144 over 2/4/8 ranks, the weak-scaling end point
    N=1,048,576 over 8): with NO jsplit pin, every rank's handle has j-partitions lying entirely inside its own rows, i.e.
    NB_RCCL_OVERLAP / nb_set_exchange_overlapped really issue force work before waiting for the gather.  (Round 2 required
    the shard to be a whole number of partitions: 24 and 44 partitions on the 1/4 and 1/8 shards gave own_splits = 0.)"""
    n = (n // (256 * g)) * 256 * g
    per = n // g
    for r in sorted({0, 1, g // 2, g - 1}):
        with Simulation(n, shard=(r * per, per)) as s:
            info = s.shape_info()
            name = s.variant
        assert info["jsplit"] >= g, (name, info)
        assert info["own_splits"] >= 1, (n, g, r, name, info)
        lo, hi = info["own_split0"] * info["j_per_split"], (info["own_split0"] + info["own_splits"]) * info["j_per_split"]
        assert r * per <= lo and min(hi, n) <= (r + 1) * per, (n, g, r, info)      # inside the rank's own rows
        assert info["own_splits"] * info["j_per_split"] > per - 2 * info["j_per_split"]   # all but the straddlers


@pytest.mark.parametrize("n,g,variant", [(16384, 4, 304014), (20480, 8, 304014), (12288, 3, 28), (65536, 8, 0)])
def test_overlapped_exchange_with_model_chosen_splits_and_virtual_shards(n, g, variant):
    """The overlapped hooks on ONE GPU with g shard handles and the split count the MODEL picks (no jsplit pin; shards that
    are not a whole number of partitions): own-row partitions first, the straddling and foreign ones after wait().
    Bit-identical to an unsharded handle running the same kernel with the same number of partitions."""
    import torch
    steps = 5
    per = n // g
    assert per * g == n and per % 256 == 0
    b, v = ic.plummer(n, seed=75)
    stream = torch.cuda.current_stream().cuda_stream
    bufs = [torch.empty((n, 4), device="cuda", dtype=torch.float32) for _ in range(g)]
    sims = [Simulation(n, shard=(r * per, per), stream=stream, ext_bodies=bufs[r].data_ptr(), force_variant=variant) for r in range(g)]
    infos = [s.shape_info() for s in sims]
    names = [s.variant for s in sims]
    snap = {}
    calls = {"begin": 0, "wait": 0}
    try:
        assert len(set(names)) == 1 and all(i["own_splits"] >= 1 for i in infos), (names, infos)
        for r, s in enumerate(sims):
            s.init(b, v)
            s.set_params(1e-3, 1.0)

            def begin(ptr, esz, nn, sb, sc, st):
                calls["begin"] += 1
                return 0

            def wait(st, r=r):
                calls["wait"] += 1
                for q in range(g):
                    if q != r:
                        bufs[r][q * per:(q + 1) * per].copy_(snap[q])
                return 0

            s.set_exchange_overlapped(begin, wait)
        for _ in range(steps):
            snap = {q: bufs[q][q * per:(q + 1) * per].clone() for q in range(g)}
            for s in sims:
                s.step()
        for s in sims:
            s.sync()
        bodies = np.concatenate([bufs[r][r * per:(r + 1) * per].cpu().numpy() for r in range(g)])
        vel = np.zeros((n, 4), np.float32)
        for r, s in enumerate(sims):
            vel[r * per:(r + 1) * per] = s.read(bodies=False, accel=False)[1][r * per:(r + 1) * per]
    finally:
        for s in sims:
            s.close()
    assert calls["begin"] == g * steps and calls["wait"] == g * steps
    # the unsharded twin: same kernel family / bodies per lane / waves, same number of j-partitions
    js = infos[0]["jsplit"]
    twin = variant
    if variant == 0:
        nm = names[0]
        assert "sgpr_ipl" in nm, nm
        twin = 300000 + int(nm.split("ipl")[1].split("_")[0]) * 1000 + 10 + (4 if "_ws4" in nm else 1)
    with Simulation(n, force_variant=twin, jsplit=js, flags=capi.NB_FLAG_NO_FUSE) as one:
        assert one.shape_info()["j_per_split"] == infos[0]["j_per_split"], (one.variant, names[0])
        one.init(b, v)
        one.simulate(steps, 1e-3, 1.0)
        ref = one.read()
    assert bodies.tobytes() == ref[0].tobytes(), names[0]
    assert vel.tobytes() == ref[1].tobytes(), names[0]
