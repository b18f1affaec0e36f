"""Multi-GPU host logic: one process per GPU, utterances sharded across ranks, no data-path collective.

Every utterance is independent (the reference serves one at a time, src/pocket_tts.cpp:532-664), so the
only cross-rank traffic is the control plane: a barrier around the timed region and two scalar reductions
(MAX of the wall time, SUM of the frames produced).  They run over gloo on host tensors, on the GPU box as in the
CPU tests: the data path has no collective, so RCCL is not loaded at all.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def env_ranks() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) when run alone."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def partition(costs: Sequence[float], world: int) -> List[List[int]]:
    """Assigns job i (cost = expected frames, i.e. prefill + decode work) to a rank: longest job first onto
    the least-loaded rank, ties to the lower rank.  Deterministic, so every rank computes the same table
    without talking to the others.  Returns per-rank lists of job indices in submission order."""
    if world < 1:
        raise ValueError("world must be >= 1")
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda j: (-float(costs[j]), j)):
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += float(costs[i])
        out[r].append(i)
    return [sorted(x) for x in out]


def batches(jobs: Sequence[int], costs: Sequence[float], max_batch: int) -> List[List[int]]:
    """Groups a rank's jobs into lock-step batches of at most max_batch, similar lengths together so a
    batch's padded frame loop wastes little (the batch runs until its longest utterance ends)."""
    order = sorted(jobs, key=lambda j: (-float(costs[j]), j))
    return [order[i:i + max_batch] for i in range(0, len(order), max_batch)]


def barrier(dist, device=None) -> None:
    """Every engine call returns with its stream drained, so a host barrier brackets the timed region exactly."""
    if dist is None:
        return
    dist.barrier()


def reduce_job(dt_local: float, units_local: float, dist, device=None) -> Tuple[float, float]:
    """(max over ranks of the wall time, sum over ranks of the units processed)."""
    if dist is None:
        return float(dt_local), float(units_local)
    import torch
    t = torch.tensor([dt_local], dtype=torch.float64)
    u = torch.tensor([units_local], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
