"""The N>1 path on CPU: world_size 2 and 3 over gloo (prompt section 5).
Checks that i-sharding + per-step all-gather reproduces the unsharded oracle
bit for bit, including a ragged N that needs zero-mass padding rows."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle
from nbody3d_amd import ic
from nbody3d_amd.shard import ShardPlan

WORKER = os.path.join(ROOT, "tests", "dist", "gloo_shard_worker.py")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_plan_partitions_and_pads():
    for n, w in [(1024, 2), (1000, 2), (262144, 8), (1048576, 8), (40002, 8), (5, 3)]:
        plans = [ShardPlan(n, w, r) for r in range(w)]
        assert all(p.count == plans[0].count and p.count % 256 == 0 for p in plans)
        assert [p.begin for p in plans] == [r * plans[0].count for r in range(w)]
        assert plans[0].padded_n == plans[0].count * w >= n
        assert plans[0].padded_n - n < 256 * w
        x = plans[0].pad(np.ones((n, 4), np.float32))
        assert x.shape == (plans[0].padded_n, 4) and x[n:].sum() == 0 and x[:n].sum() == 4 * n
    assert ShardPlan(262144, 8, 3).begin == 3 * 32768
    with pytest.raises(ValueError):
        ShardPlan(10, 2, 2)


@pytest.mark.parametrize("n,world,mode", [(1024, 2, "plain"), (1000, 2, "plain"), (1536, 3, "plain"), (1000, 2, "overlapped"),
                                           (1536, 3, "overlapped")])
def test_sharded_steps_equal_unsharded_oracle(tmp_path, n, world, mode):
    """mode 'overlapped' drives the two-phase hooks, whose all-gather is in place (the input is a
    view of the rank's own rows inside the output tensor)."""
    steps = 4
    out = str(tmp_path / "result.npz")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), WORKER, out, str(n), str(steps), mode]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    got = np.load(out)
    b, v = ic.plummer(n, seed=21)
    rb, rv, ra = oracle.run_f32(b, v, None, 1e-3, 1.0, steps)
    # padding rows are zero-mass bodies at the origin: they add exactly 0 to every sum,
    # so the sharded result is bit-identical to the unsharded, unpadded oracle
    assert got["bodies"].tobytes() == rb.tobytes()
    assert got["vel"].tobytes() == rv.tobytes()
    assert got["acc"].tobytes() == ra.tobytes()


@pytest.mark.parametrize("n,world", [(512, 2), (768, 3), (1024, 4), (700, 2)])
def test_rank_form_protocol_reproduces_the_unsharded_oracle(tmp_path, n, world):
    """The multi-GPU protocol of the symmetric pass (pairs divided among the ranks on a ring, partial accelerations reduced
    across ranks, own rows integrated, positions all-gathered) over gloo with numpy standing in for the kernels: every ordered
    pair accounted for exactly once, to rounding against the unsharded fp32 oracle, momentum conserved by construction."""
    steps = 3
    out = str(tmp_path / "result.npz")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), WORKER, out, str(n), str(steps), "rankform"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    got = np.load(out)
    b, v = ic.plummer(n, seed=21)
    rb, rv, ra = oracle.run_f32(b, v, None, 1e-3, 1.0, steps)
    assert np.abs(got["acc"][:, :3] - ra[:, :3]).max() < 5e-6 * np.abs(ra[:, :3]).max()
    assert np.abs(got["bodies"][:, :3] - rb[:, :3]).max() < 1e-6 * np.abs(rb[:, :3]).max()
    f = b[:, 3:4].astype(np.float64) * got["acc"][:, :3]
    assert np.all(np.abs(f.sum(0)) < 1e-6 * np.abs(f).sum(0))
