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


def main():
    out, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
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
