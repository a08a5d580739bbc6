"""Data-parallel utterance sharding: one engine process per GPU, a full weight replica
and a private KV pool each, NO cross-GPU traffic inside a step.  torch.distributed (RCCL on ROCm, gloo on
CPU) is used only outside the step loop: one gather of the emitted ids.  The reference has no
multi-GPU path (no call site to mirror); utterances are independent, and the RNG is keyed by the GLOBAL
utterance id, so the 1-GPU and N-GPU token streams are identical.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple


def shard_indices(costs: Sequence[float], world: int) -> List[List[int]]:
    """Length-aware greedy bin packing (longest first onto the lightest rank); ties by index, deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i); loads[r] += costs[i]
    for s in shards:
        s.sort()
    return shards


def generate_sharded(generate_fn: Callable[[List[int]], List[List[int]]], n_items: int, costs: Sequence[float],
                     rank: int, world: int, gather: bool = True) -> Tuple[List[int], List[List[int]]]:
    """Run ``generate_fn(global indices of this rank's shard)`` and gather every rank's token lists.

    Returns (my indices, results): results has n_items entries on every rank when gather=True (others'
    entries filled in from the gather), else only this rank's entries are non-None.
    """
    shards = shard_indices(costs, world)
    mine = shards[rank]
    local = generate_fn(mine) if mine else []
    results: List = [None] * n_items
    for i, toks in zip(mine, local):
        results[i] = toks
    if gather and world > 1:
        import torch.distributed as dist

        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, local))
        for idxs, outs in gathered:
            for i, toks in zip(idxs, outs):
                results[i] = toks
    return mine, results
