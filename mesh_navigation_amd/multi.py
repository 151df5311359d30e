"""Multi-GPU layer: one process per GPU, plans sharded by goal.

The wavefront of ONE plan does not shard without an exchange step per band; the
path that shards naturally is the batch of independent goals (BASELINE config 5): the mesh and the
cost arrays are replicated on every GPU (a 1M-vertex mesh is ~0.2 GB of the 288 GB), goal g of a
global list goes to rank (g * world) // n, and there is NO data-path collective -- only the final
gather of the per-plan results (codes, path lengths, paths) that a single consumer needs.
`torch.distributed` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) is plumbing here.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of n items: rank r owns [lo, hi)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def plan_goals_sharded(plan_batch: Callable[[np.ndarray, np.ndarray], dict], goals: Sequence[int], targets: Sequence[int],
                       rank: int = 0, world: int = 1, dist=None, gather_to: int | None = 0):
    """Plans this rank's share of (goals, targets) with `plan_batch` (e.g. MnavContext.plan_dijkstra_batch)
    and, if `dist` (torch.distributed) is given, gathers codes / paths on rank `gather_to` in the
    global goal order.  Returns (local_result, gathered or None)."""
    goals = np.asarray(goals, np.uint32)
    targets = np.asarray(targets, np.uint32)
    lo, hi = shard_range(len(goals), rank, world)
    local = plan_batch(goals[lo:hi], targets[lo:hi]) if hi > lo else dict(codes=np.zeros(0, np.uint32), paths=[])
    local = dict(local)
    local["range"] = (lo, hi)
    if dist is None or world == 1:
        return local, dict(codes=np.asarray(local["codes"]), paths=list(local["paths"]))
    payload = (lo, hi, np.asarray(local["codes"]).tolist(), [np.asarray(p).tolist() for p in local["paths"]])
    if gather_to is None:
        out = [None] * world
        dist.all_gather_object(out, payload)
    else:
        out = [None] * world if rank == gather_to else None
        dist.gather_object(payload, out, dst=gather_to)
        if rank != gather_to:
            return local, None
    codes = np.zeros(len(goals), np.uint32)
    paths: list = [None] * len(goals)
    for plo, phi, pc, pp in out:
        codes[plo:phi] = pc
        for k, path in enumerate(pp):
            paths[plo + k] = np.asarray(path, np.uint32)
    return local, dict(codes=codes, paths=paths)


def aggregate_throughput(n_local_plans: int, elapsed_s: float, dist=None) -> tuple[int, float]:
    """(total plans over all ranks, max elapsed over ranks): value = total / max time (bench.py)."""
    if dist is None:
        return n_local_plans, elapsed_s
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    n = torch.tensor([n_local_plans], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return int(n.item()), float(t.item())
