"""i-shard planning and the per-step position all-gather.

The force pass shards over i-bodies: rank r of g owns a contiguous block of
rows, holds the FULL replicated bodies array, and after its integrate kernel
all ranks all-gather their rows.  One process per GPU; the collective goes
through torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" in
the CPU tests).  The reference has no multi-device code at all.
"""
import numpy as np

ROW_ALIGN = 256  # keep shard boundaries on the reference's tile size (nbody3d.js:4)


class ShardPlan:
    """Contiguous, ROW_ALIGN-aligned i-blocks; the last rank takes the remainder.

    Equal counts are required by all_gather_into_tensor; when n is not divisible
    the plan pads: every rank owns ``rows`` rows of a padded array of
    ``padded_n = rows * world`` bodies, and rows >= n are zero-mass bodies at the
    origin that exert and feel no force (zero mass, finite distance).
    """

    def __init__(self, n, world, rank=0, align=ROW_ALIGN):
        if world < 1 or not (0 <= rank < world):
            raise ValueError("bad world/rank")
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        per = -(-self.n // self.world)
        per = -(-per // align) * align
        self.rows = per
        self.padded_n = per * self.world
        self.begin = per * self.rank
        self.count = per

    def pad(self, a):
        """(n,4) -> (padded_n,4) with zero rows appended."""
        a = np.asarray(a).reshape(-1, 4)
        if a.shape[0] == self.padded_n:
            return np.ascontiguousarray(a)
        out = np.zeros((self.padded_n, 4), a.dtype)
        out[: a.shape[0]] = a
        return out

    def pairs_per_step(self):
        """Real pair interactions of one step of the whole job: N(N-1)."""
        return self.n * (self.n - 1)


def torch_allgather_hook(bodies_tensor, plan, group=None):
    """Exchange hook for Simulation.set_exchange: IN-PLACE all-gather of each
    rank's rows of ``bodies_tensor`` (shape (padded_n, 4), the tensor whose
    storage the engine uses as its replicated bodies array).

    The engine enqueues its kernels on torch's current stream (it is created
    with stream=torch.cuda.current_stream().cuda_stream), so the collective --
    which torch orders after the current stream and makes the current stream
    wait for -- is correctly ordered against the integrate kernel before it and
    the next step's force kernel after it.
    """
    import torch.distributed as dist

    # IN PLACE, as in all three exchange kinds (this hook, the overlapped pair below, the engine's native
    # ncclAllGather): the input is the view of this rank's rows inside the output tensor (input pointer =
    # output pointer + rank * rows, the in-place form RCCL documents for ncclAllGather), so the collective never
    # rewrites the rank's own rows and no per-step device copy is needed.  Covered with 2 and 3 ranks over gloo
    # (tests/test_shard_gloo.py) and with one rank on the nccl backend (bench.py --force-dist --exchange torch).
    mine = bodies_tensor[plan.begin: plan.begin + plan.count]

    def hook(bodies_ptr, esz, n, sb, sc, stream):
        dist.all_gather_into_tensor(bodies_tensor, mine, group=group)
        return 0

    return hook


def torch_allgather_overlapped_hooks(bodies_tensor, plan, group=None):
    """(begin, wait) for Simulation.set_exchange_overlapped: the all-gather is issued
    with async_op=True right after the integrate kernel; the engine then enqueues
    the next step's force work on the j-range of its OWN rows and only then calls wait(),
    which makes the current stream wait for the collective.  Hides the collective behind
    1/world of the force pass ("hide it by starting K1 on the rank's own j-block").

    The gather is IN PLACE: the input is the view of this rank's rows inside ``bodies_tensor``
    (input pointer = output pointer + rank * rows, the form RCCL documents for ncclAllGather),
    so the collective never writes the rank's own rows -- the rows the early force launch is
    reading while the collective runs.  (A separate send buffer would make the collective copy
    them back onto themselves: same values, but an unsynchronised write under a concurrent read.)
    The native path (nb_rccl_attach with NB_RCCL_OVERLAP) does the same inside the engine."""
    import torch.distributed as dist

    mine = bodies_tensor[plan.begin: plan.begin + plan.count]
    state = {"work": None}

    def begin(bodies_ptr, esz, n, sb, sc, stream):
        state["work"] = dist.all_gather_into_tensor(bodies_tensor, mine, group=group, async_op=True)
        return 0

    def wait(stream):
        w, state["work"] = state["work"], None
        if w is not None:
            w.wait()
        return 0

    return begin, wait


def torch_allgather_via_host_hook(bodies_tensor, plan, group=None):
    """REHEARSAL ONLY (bench.py --exchange host): same exchange as torch_allgather_hook but
    staged through host memory over a gloo group, so that several ranks can share ONE GPU
    (RCCL refuses two ranks on one device).  Used to walk the whole multi-rank control flow
    -- shard plan, engine on torch's stream with a torch-owned bodies buffer, hook, barrier,
    max-over-ranks timing -- on the 1-GPU development box.  Never the measured path."""
    import torch
    import torch.distributed as dist

    mine = bodies_tensor[plan.begin: plan.begin + plan.count]
    host_all = torch.empty((plan.padded_n, 4), dtype=bodies_tensor.dtype).pin_memory()
    host_mine = torch.empty((plan.count, 4), dtype=bodies_tensor.dtype).pin_memory()

    def hook(bodies_ptr, esz, n, sb, sc, stream):
        host_mine.copy_(mine)                       # D2H on the current stream, synchronous for pageable->pinned
        torch.cuda.current_stream().synchronize()
        dist.all_gather_into_tensor(host_all, host_mine, group=group)
        bodies_tensor.copy_(host_all, non_blocking=False)
        return 0

    return hook
