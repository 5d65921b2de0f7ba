"""Multi-GPU sharding of the path set: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm), paths partitioned by GLOBAL id, one all-reduce of the moment vector per pricing call.

The reference is single-GPU (no NCCL/MPI anywhere); this is the data-parallel layer the build adds.
Because every pair's Philox subsequence is its global id, the union of the shards is the same
sample set whatever the number of ranks: results differ only by float64 summation order.
Message sizes: 5 doubles (ZBC), n_mat doubles (curve), 1 double (Vega) -> pure latency on xGMI.
"""
import json
import os

import torch
import torch.distributed as dist

REFERENCE_SUMS_FILE = os.environ.get("HWMC_REFERENCE_SUMS") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_sums.json")
_reference_cache = None


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


# ---- a multi-rank run that checks itself (no 8-GPU node was ever available to the authoring rounds: the first run on
# ---- one has to tell, from its own output line, whether it priced the right sample set and where its time went) ----------
def world_size_seen(device="cpu", group=None):
    """How many ranks really take part: torch's own count and the sum of an all-reduce of ones (a rank that silently
    dropped out of the collective, or a launcher that started fewer processes than --gpus, shows as a mismatch)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"dist": 1, "allreduce_of_ones": 1}
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, group=group)
    return {"dist": int(dist.get_world_size(group)), "allreduce_of_ones": int(round(float(one.item())))}


def gather_floats(values, device="cpu", group=None):
    """values (a list of floats of the same length on every rank) from all ranks -> [world][len] (rank order)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [x.tolist() for x in out]


def launch_skew(t_start_wall_s, device="cpu", group=None):
    """Spread of the ranks' host clocks (time.time(): one node, one clock) at the moment each rank began to enqueue a timed
    region, behind the barrier: what the barrier's own release jitter costs a region of K steps.  Input: this rank's
    start times, one per region."""
    rows = gather_floats(t_start_wall_s, device, group)
    per_region = [max(col) - min(col) for col in zip(*rows)]
    per_region.sort()
    med = per_region[len(per_region) // 2] if per_region else 0.0
    return {"median_ms": med * 1e3, "max_ms": (per_region[-1] if per_region else 0.0) * 1e3, "regions": len(per_region),
            "how": "max - min over ranks of time.time() when the rank enqueued the first launch of a timed region (same host clock)"}


# ---- committed one-GPU constants: an N-GPU line proves itself WITHOUT a companion N = 1 line -----------------------------------
def reference_key(kernel, profile, precision, stepper, grid_steps, pairs_log2, seed):
    return "%s|%s|%s|%s|%d|%d|%d" % (kernel, profile, precision, stepper, grid_steps, pairs_log2, seed)


def load_reference_sums():
    """reference_sums.json beside this file (written by tools/make_reference_sums.py on one MI355X): whole-call moment sums
    per configuration, plus `_one_gpu_ms` (the committed one-GPU times of the same calls) and `_predicted_ms_per_step`."""
    global _reference_cache
    if _reference_cache is None:
        try:
            with open(REFERENCE_SUMS_FILE) as fh:
                _reference_cache = json.load(fh)
        except (OSError, ValueError):
            _reference_cache = {}
    return _reference_cache


def committed_one_gpu_line(lineN):
    """The N = 1 side of check_scaling_lines for `lineN`'s configuration, built from the committed constants; None when the
    store has no entry for it (another size, grid, profile or stepper)."""
    c = lineN["config"]
    n = int(c["pairs_per_step"])
    if n <= 0 or n & (n - 1):
        return None, None
    store = load_reference_sums()
    base = reference_key("zbc", c["theta_profile"], lineN["dtype"], c["stepper"], c["grid_steps"], n.bit_length() - 1, 1234)
    sums = store.get(base)
    if sums is None:
        return None, base
    t = (store.get("_one_gpu_ms") or {}).get(base.rsplit("|", 1)[0]) or {}
    ref = {"pairs_per_step": n, "n_gpus": 1, "sums_step0": sums, "ms_per_step": t.get("ms_per_step"),
           "serial_ms_per_step": t.get("serial_ms_per_step"), "kernel_ms": t.get("kernel_ms"), "source": t.get("source")}
    line1 = {"metric": lineN["metric"], "unit": lineN["unit"], "dtype": lineN["dtype"], "n_gpus": 1,
             "config": {k: c[k] for k in ("grid_steps", "steps_executed", "stepper", "theta_profile")},
             "strong_scaling_reference": ref}
    return line1, base


def self_check(lineN, gpus_requested, rel=1e-13):
    """What bench.py --gpus N puts into multi_gpu.self_check (rank 0, before it prints): did every requested rank take part,
    did the N ranks price the sample set ONE GPU prices (sums_step0 against the committed constant of the configuration, 1e-13),
    and how does ms_per_step compare with the committed one-GPU time and with the prediction written down before any N > 1
    run existed.  `failed` lists what makes bench.py exit non-zero: a rank count that is not --gpus, or
    sums that differ.  A configuration without a committed constant (e.g. the 2^18-pair rehearsal) is reported, not failed."""
    mg = lineN.get("multi_gpu") or {}
    seen = mg.get("world_size_seen") or {}
    n = lineN["n_gpus"]
    out = {"gpus_requested": gpus_requested, "world_size_seen": seen, "failed": []}
    if not (seen.get("dist") == seen.get("allreduce_of_ones") == n == gpus_requested):
        out["failed"].append("ranks seen %r, --gpus %d, n_gpus %d" % (seen, gpus_requested, n))
    line1, key = committed_one_gpu_line(lineN) if lineN.get("scaling") == "strong" else (None, None)
    out["reference"] = key
    if line1 is None:
        out["sums_max_rel_diff"] = None
        out["verdict"] = "no committed one-GPU constant for this configuration (%s): sample set unchecked" % (key or "weak scaling or odd size")
        return out
    a, b = lineN["sums_step0"], line1["strong_scaling_reference"]["sums_step0"]
    worst = max(abs(x - y) / max(abs(y), 1e-300) for x, y in zip(a, b))
    out["sums_max_rel_diff"] = worst
    if worst > rel:
        out["failed"].append("sums_step0 differ from the committed one-GPU sums by %.3e (relative): another sample set was priced" % worst)
    ref = line1["strong_scaling_reference"]
    out["one_gpu_ms_per_step"] = ref["ms_per_step"]
    out["one_gpu_source"] = ref.get("source")
    pred = ((load_reference_sums().get("_predicted_ms_per_step") or {}).get(key.rsplit("|", 1)[0]) or {}).get(str(n))
    if pred:
        out["predicted_ms_per_step"] = pred
        out["ms_per_step_over_prediction"] = lineN["ms_per_step"] / pred["throughput_mode"] if pred.get("throughput_mode") else None
    if n < 2:
        out["verdict"] = "sample set verified on one rank (the forced-RCCL path of a one-GPU box)" if not out["failed"] else "FAILED: " + "; ".join(out["failed"])
        return out
    if out["failed"] or not ref["ms_per_step"]:
        out["verdict"] = "FAILED: " + "; ".join(out["failed"]) if out["failed"] else "sample set verified; no committed one-GPU time for this configuration"
        return out
    try:
        rep = check_scaling_lines(line1, lineN, rel)
        out.update(strong_scaling=rep["strong_scaling"], compute_scaling=rep["compute_scaling"], rank_imbalance=rep["rank_imbalance"],
                   verdict=rep["verdict"])
    except AssertionError as exc:                     # an invariant beyond ranks and sums (value != paths / time, missing fields)
        out["failed"].append(str(exc))
        out["verdict"] = "FAILED: %s" % exc
    return out


def check_scaling_lines(line1, lineN, rel=1e-13):
    """Invariants an N-GPU bench line must satisfy against the N = 1 line of the same build (both are bench.py JSON lines).
    Raises AssertionError naming the violated invariant; returns the scaling report.

      * same metric, dtype, grid and stepper; lineN prices the workload line1 carries as `strong_scaling_reference`
      * every rank took part: world_size_seen.dist == world_size_seen.allreduce_of_ones == n_gpus
      * SAME SAMPLE SET: the reduced sums of the first timed step equal the one-GPU sums of the same pricing call to `rel`
        (subsequence = global pair id: only the order of the float64 additions differs)
      * value == effective_paths / ms_per_step, and ms_per_step is not below the slowest rank's kernel time
    The report names what decides between ">= 7x" and "collective exposed": strong_scaling (reference ms / N-GPU ms),
    compute_scaling (reference kernel ms / slowest rank's kernel ms), exposed_allreduce_ms_per_step, launch_skew."""
    ref = line1.get("strong_scaling_reference")
    assert ref, "the N = 1 line carries no strong_scaling_reference (run bench.py without --no-extras)"
    n = lineN["n_gpus"]
    assert n > 1 and line1["n_gpus"] == 1, "need an N = 1 and an N > 1 line"
    for key in ("metric", "unit", "dtype"):
        assert line1[key] == lineN[key], "lines differ in %s" % key
    c1, cN = line1["config"], lineN["config"]
    for key in ("grid_steps", "steps_executed", "stepper", "theta_profile"):
        assert c1[key] == cN[key], "lines differ in config.%s" % key
    assert lineN["scaling"] == "strong" and cN["pairs_per_step"] == ref["pairs_per_step"], \
        "the N-GPU line prices %d pairs per call, the N = 1 reference %d" % (cN["pairs_per_step"], ref["pairs_per_step"])
    mg = lineN.get("multi_gpu")
    assert mg, "the N-GPU line has no multi_gpu block"
    seen = mg["world_size_seen"]
    assert seen["dist"] == seen["allreduce_of_ones"] == n, "ranks seen %r, line says n_gpus = %d" % (seen, n)
    assert len(mg["per_rank_kernel_ms"]) == n
    a, b = lineN["sums_step0"], ref["sums_step0"]
    assert len(a) == len(b) == 5
    worst = max(abs(x - y) / max(abs(y), 1e-300) for x, y in zip(a, b))
    assert worst <= rel, "the N-GPU run did not price the N = 1 sample set: sums_step0 differ by %.3e (relative)" % worst
    eff = cN["effective_paths"]
    assert abs(lineN["value"] - eff / lineN["ms_per_step"] / 1000.0) <= 1e-3 * lineN["value"], "value is not effective_paths / ms_per_step"
    slowest = max(mg["per_rank_kernel_ms"])
    report = {
        "n_gpus": n, "sums_step0_max_rel_diff": worst,
        "strong_scaling": ref["ms_per_step"] / lineN["ms_per_step"],
        "strong_scaling_serial_reference": (ref.get("serial_ms_per_step") or ref["ms_per_step"]) / lineN["ms_per_step"],
        "compute_scaling": (ref.get("kernel_ms") or ref["serial_ms_per_step"]) / slowest,
        "exposed_allreduce_ms_per_step": mg.get("exposed_allreduce_ms_per_step"),
        "launch_skew_median_ms": (mg.get("launch_skew") or {}).get("median_ms"),
        "single_call_ms": {"rccl": mg.get("single_call_ms"), "host_sum": mg.get("single_call_host_sum_ms")},
        "rank_imbalance": max(mg["per_rank_kernel_ms"]) / min(mg["per_rank_kernel_ms"]),
    }
    if not lineN.get("rehearsal"):
        assert lineN["ms_per_step"] >= 0.9 * slowest / max(1, cN.get("streams_per_gpu", 1)), \
            "ms_per_step %.4f is below the slowest rank's kernel time %.4f: the timed region cannot have held K full steps" % (lineN["ms_per_step"], slowest)
    # the verdict in words: where a shortfall against N x comes from
    target = 7.0 / 8.0 * n
    if report["strong_scaling"] >= target:
        report["verdict"] = "scales: %.2fx on %d GPUs (>= %.2fx)" % (report["strong_scaling"], n, target)
    elif report["compute_scaling"] < target:
        report["verdict"] = "compute-bound shortfall: the slowest rank's kernel alone scales %.2fx (shard too small / rank imbalance %.3f)" % (report["compute_scaling"], report["rank_imbalance"])
    else:
        report["verdict"] = "collective exposed: kernels scale %.2fx but the step %.2fx; exposed all-reduce %.4f ms per step, launch skew %.4f ms" % (
            report["compute_scaling"], report["strong_scaling"], report["exposed_allreduce_ms_per_step"] or float("nan"), report["launch_skew_median_ms"] or float("nan"))
    return report
