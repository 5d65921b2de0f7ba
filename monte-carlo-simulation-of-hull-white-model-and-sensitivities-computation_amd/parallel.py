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


class StepReducer:
    """Which steps' sums travel together.  A pricing run makes K steps; step i leaves its sums in row i of a
    [K, n] tensor.  `every` = 1 sends ONE all-reduce per pricing call (the default), `every` = B
    one message per B consecutive steps, `every` = 0 a single message after the last step.  The caller supplies
    `reduce_rows(i0, i1, producers)`: all-reduce rows i0..i1 (contiguous) once the launches of `producers` (opaque
    handles, e.g. stream indices) have finished, returning a work handle or None.  bench.py passes an RCCL
    all-reduce on a stream of its own; tests/test_distributed_gloo.py a gloo one."""

    def __init__(self, every, n_steps, reduce_rows):
        self.bucket = max(1, int(every)) if every > 0 else max(1, int(n_steps))
        self.reduce_rows = reduce_rows
        self.first = None
        self.producers = []

    def after_step(self, i, producer=None, last=False):
        """Call after step i has been enqueued; returns the work handle of the message it completed, if any."""
        if self.first is None:
            self.first = i
        if producer is not None and producer not in self.producers:
            self.producers.append(producer)
        if last or i - self.first + 1 >= self.bucket:
            i0, prods = self.first, self.producers
            self.first, self.producers = None, []
            return self.reduce_rows(i0, i, prods)
        return None


def pricing_workload(n_gpus, scaling="strong", pairs_log2=20, total_log2=24):
    """Pairs per pricing call over all GPUs: N = 1 and weak scaling price 2^pairs_log2 pairs per GPU (BASELINE
    configs[1]); N > 1 strong scaling shards ONE 2^total_log2-pair call (configs[3]: 2^24)."""
    if n_gpus == 1 or scaling == "weak":
        return (1 << pairs_log2) * n_gpus
    return 1 << total_log2
