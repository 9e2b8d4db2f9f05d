"""Multi-GPU plumbing: one process per GPU, reads shard embarrassingly, and the ONLY collective of the path is
one all-reduce(sum) of the additive u64 counter block -- RCCL over xGMI when the
process group is `nccl`, gloo in the CPU tests.

The reference has no counterpart (it is a single OpenMP process whose only reduction is the `omp critical`
merge at trim.cpp:120-154); this is what that merge becomes across devices.
"""
import numpy as np


class _DevArray:
    """Minimal __cuda_array_interface__ holder so torch can wrap a raw device pointer without copying."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 3}


def shard_bounds(n_items, rank, world_size):
    """Contiguous split of [0, n_items) into world_size nearly equal shards (reference batch order is kept
    inside a shard; counters are order-independent)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_counters_device(engine, group=None):
    """In-place all-reduce(sum) of the engine's device-resident counter block.  uint64 sums are done on the
    int64 view (two's-complement addition is the same operation)."""
    import torch
    import torch.distributed as dist

    ptr, n = engine.counters_device()
    engine.sync()
    t = torch.as_tensor(_DevArray(ptr, n), device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    torch.cuda.synchronize()
    return t


def allreduce_counters_host(block, group=None):
    """Same collective on a host copy of the block (gloo path; also used for the k-mer-free CPU tests)."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(block).view(np.int64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy().view(np.uint64)
