"""Multi-GPU sharding of the path set: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm), paths partitioned by GLOBAL id, one all-reduce of the moment vector per pricing call.

The reference is single-GPU (no NCCL/MPI anywhere); this is the data-parallel layer the build adds.
Because every pair's Philox subsequence is its global id, the union of the shards is the same
sample set whatever the number of ranks: results differ only by float64 summation order.
Message sizes: 5 doubles (ZBC), n_mat doubles (curve), 1 double (Vega) -> pure latency on xGMI.
"""
import torch
import torch.distributed as dist


def shard(n_total, rank, world):
    """Contiguous block partition of [0, n_total): returns (offset, count) of `rank`."""
    base, rem = divmod(int(n_total), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def allreduce_sums(t, async_op=False, group=None, force=False):
    """Sum a tensor of moment sums over the ranks in place (no-op in a single process unless `force`: a one-rank
    group still runs the collective, which is how a one-GPU box exercises the RCCL path)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op, group=group)
    return None


def sharded_sums(local_fn, n_total, device="cpu", group=None):
    """Run local_fn(count, path_offset) -> sequence of float64 sums on this rank's shard of
    [0, n_total) and return the all-reduced vector (identical on every rank)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    off, cnt = shard(n_total, rank, world)
    t = torch.as_tensor(local_fn(cnt, off), dtype=torch.float64, device=device).clone()
    allreduce_sums(t, group=group)
    return t
