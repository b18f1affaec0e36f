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


def spawn_plan(n_ranks: int, argv: Sequence[str], port: int, base_env=None) -> List[Tuple[dict, List[str]]]:
    """One (environment, command line) pair per rank for a single-node run of `argv` as n_ranks processes, one GPU each:
    the variables torch.distributed.run would export (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT), the
    rendezvous on 127.0.0.1.  The launcher that uses this starts fresh child processes and must not have touched the GPU."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    import sys
    plan = []
    for r in range(n_ranks):
        env = dict(os.environ if base_env is None else base_env)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        plan.append((env, [sys.executable] + list(argv)))
    return plan


def visible_gpus() -> int:
    """GPUs this process could open, counted without initialising the HIP runtime (torch's device_count does not)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def launch_ranks(n_ranks: int, argv: Sequence[str], timeout_s: float = 3000.0, gpus_visible=None) -> int:
    """Runs `argv` as n_ranks child processes (spawn_plan), relays rank 0's standard output, and returns 0 only if every
    rank exited with 0.  A rank that fails ends the others: nothing is retried.  gpus_visible overrides the device count
    (the CPU tests run ranks that open no GPU)."""
    import socket
    import subprocess
    import sys
    import time
    have = visible_gpus() if gpus_visible is None else int(gpus_visible)
    if have < n_ranks:
        print(f"bench: {n_ranks} GPUs requested, {have} visible", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r, (env, cmd) in enumerate(spawn_plan(n_ranks, argv, port)):
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    t0 = time.time()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        if time.time() - t0 > timeout_s:
            failed = -1
        time.sleep(0.05)
    if failed is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                failed = r
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        what = "timed out" if failed < 0 else f"rank {failed} exited with {procs[failed].returncode}"
        print(f"bench: {what}; the other ranks were stopped", file=sys.stderr, flush=True)
        return 1
    sys.stdout.write(procs[0].stdout.read())
    sys.stdout.flush()
    return 0


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


def reduce_job(dt_local: float, units_local: float, dist, device=None) -> Tuple[float, float, int]:
    """(max over ranks of the wall time, sum over ranks of the units processed, number of ranks that reported)."""
    if dist is None:
        return float(dt_local), float(units_local), 1
    import torch
    t = torch.tensor([dt_local], dtype=torch.float64)
    u = torch.tensor([units_local, 1.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u[0].item()), int(round(u[1].item()))
