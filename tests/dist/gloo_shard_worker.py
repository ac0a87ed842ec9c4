"""world_size-2 (or more) CPU worker for tests/test_shard_gloo.py.

Runs the PRODUCT's shard protocol -- nbody3d_amd.shard.ShardPlan for the
partition, torch_allgather_hook for the per-step position all-gather -- over
the gloo backend, with the oracle standing in for the two HIP kernels (this is
a test: there is no GPU here).  Rank 0 writes the final state to argv[1].
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))

from nbody3d_amd import ic  # noqa: E402
from nbody3d_amd.shard import ShardPlan, torch_allgather_hook, torch_allgather_overlapped_hooks  # noqa: E402
from oracle import oracle  # noqa: E402


def pair_block(bodies, ia, ib, G, eps2=1e-4):
    """fp32 sums of one block pair, both directions from the SAME inv * r (the symmetric pass's arithmetic, nbody3d.js:233-236):
    returns (acceleration of rows ia from rows ib, acceleration of rows ib from rows ia)."""
    xa, xb = bodies[ia, :3].astype(np.float32), bodies[ib, :3].astype(np.float32)
    ma, mb = (np.float32(G) * bodies[ia, 3]).astype(np.float32), (np.float32(G) * bodies[ib, 3]).astype(np.float32)
    d = xb[None, :, :] - xa[:, None, :]                                    # r = x_b - x_a
    d2 = (d * d).sum(2, dtype=np.float32) + np.float32(eps2)
    inv = (np.float32(1) / np.sqrt(d2 * d2 * d2)).astype(np.float32)
    fa = ((mb[None, :] * inv)[:, :, None] * d).sum(1, dtype=np.float32)    # on a: (G m_b) inv r
    fb = -((ma[:, None] * inv)[:, :, None] * d).sum(0, dtype=np.float32)   # on b: (G m_a) inv (-r)
    return fa, fb


def rank_form(out, n, steps):
    """The rank form of the symmetric pass as a protocol (csrc/nb_engine.hip sym_rank_phase_a/b, nb_comm.hip): row blocks on a
    ring, rank r evaluates the pairs of ITS block with the (world-1)/2 blocks after it (+ the antipodal one for r < world/2
    when world is even) and inside its own block; every rank then holds partial accelerations for rows of other ranks too:
    reduce (all_reduce over gloo standing in for ncclReduceScatter: each rank keeps its own rows), integrate own rows,
    all-gather positions.  Must reproduce the unsharded oracle to rounding (only the order of additions differs)."""
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = ShardPlan(n, world, rank)
    b, v = ic.plummer(n, seed=21)
    bodies = torch.from_numpy(plan.pad(b).copy())
    vel = plan.pad(v).copy()
    acc = np.zeros_like(vel)
    G, dt = 1.0, 1e-3
    bnp = bodies.numpy()
    rows = lambda r: np.arange(r * plan.count, (r + 1) * plan.count)       # noqa: E731
    H, hi = (world - 1) // 2, (0 if world & 1 else world // 2)
    hook = torch_allgather_hook(bodies, plan)
    for _ in range(steps):
        A = np.zeros((plan.padded_n, 4), np.float32)
        for dd in range(1, H + 1 + (1 if rank < hi else 0)):
            t = (rank + dd) % world
            fa, fb = pair_block(bnp, rows(rank), rows(t), G)
            A[rows(rank), :3] += fa
            A[rows(t), :3] += fb
        own = rows(rank)
        d = bnp[own][None, :, :3] - bnp[own][:, None, :3]
        d2 = (d * d).sum(2, dtype=np.float32) + np.float32(1e-4)
        inv = (np.float32(1) / np.sqrt(d2 * d2 * d2)).astype(np.float32)
        A[own, :3] += (((np.float32(G) * bnp[own, 3])[None, :] * inv)[:, :, None] * d).sum(1, dtype=np.float32)   # own block: self term is exactly 0
        tA = torch.from_numpy(A)
        dist.all_reduce(tA)                                                # the reduce-scatter: a rank only uses its own rows
        oracle.integrate_range_f32(bnp, vel, acc, np.ascontiguousarray(A[own]), plan.begin, plan.begin + plan.count, dt)
        assert hook(0, 4, plan.padded_n, plan.begin, plan.count, 0) == 0   # all-gather of the new positions
    tv, ta = torch.from_numpy(vel), torch.from_numpy(acc)
    dist.all_gather_into_tensor(tv, tv[plan.begin: plan.begin + plan.count].clone())
    dist.all_gather_into_tensor(ta, ta[plan.begin: plan.begin + plan.count].clone())
    if rank == 0:
        np.savez(out, bodies=bnp[:n], vel=vel[:n], acc=acc[:n], padded_n=plan.padded_n, rows=plan.rows)
    dist.barrier()
    dist.destroy_process_group()


def main():
    out, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == "rankform":
        return rank_form(out, n, steps)
    overlapped = len(sys.argv) > 4 and sys.argv[4] == "overlapped"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = ShardPlan(n, world, rank)
    b, v = ic.plummer(n, seed=21)
    bodies = torch.from_numpy(plan.pad(b).copy())           # the replicated array
    vel = plan.pad(v).copy()
    acc = np.zeros_like(vel)
    if overlapped:
        # the two-phase hooks: begin() starts the IN-PLACE all-gather (input = this rank's rows of the
        # replicated array, never written by the collective), wait() completes it
        begin, wait = torch_allgather_overlapped_hooks(bodies, plan)

        def hook(*a):
            rc = begin(*a)
            return rc or wait(0)
    else:
        hook = torch_allgather_hook(bodies, plan)
    G, dt = 1.0, 1e-3
    bnp = bodies.numpy()                                    # shares storage with the tensor
    for _ in range(steps):
        a = oracle.accel_f32(bnp, G, i0=plan.begin, i1=plan.begin + plan.count)   # "K1" on this shard
        oracle.integrate_range_f32(bnp, vel, acc, a, plan.begin, plan.begin + plan.count, dt)  # "K2"
        assert hook(0, 4, plan.padded_n, plan.begin, plan.count, 0) == 0          # exchange
    # collect vel/acc rows on rank 0 for the comparison
    tv, ta = torch.from_numpy(vel), torch.from_numpy(acc)
    mine_v = tv[plan.begin: plan.begin + plan.count].clone()
    mine_a = ta[plan.begin: plan.begin + plan.count].clone()
    dist.all_gather_into_tensor(tv, mine_v)
    dist.all_gather_into_tensor(ta, mine_a)
    if rank == 0:
        np.savez(out, bodies=bnp[:n], vel=vel[:n], acc=acc[:n], padded_n=plan.padded_n, rows=plan.rows)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
