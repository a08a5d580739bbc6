"""Data-parallel utterance sharding: one engine process per GPU, a full weight replica
and a private KV pool each, NO cross-GPU traffic inside a step.  torch.distributed (RCCL on ROCm, gloo on
CPU) is used only outside the step loop: one all-gather of the emitted ids as a fixed-shape int32 tensor
([B_max, 2 + max_tokens] per rank: global index, length, ids -- ~0.5 MB per rank at 128 x 1000).  The reference has no
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
                     rank: int, world: int, gather: bool = True, max_len: int = 8192) -> Tuple[List[int], List[List[int]]]:
    """Run ``generate_fn(global indices of this rank's shard)`` and gather every rank's token lists (each at most max_len ids).

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
        for i, toks in gather_token_lists(mine, local, max(len(s) for s in shards), max_len, world):
            results[i] = toks
    return mine, results


def gather_token_lists(idxs: Sequence[int], lists: Sequence[Sequence[int]], rows: int, max_len: int, world: int):
    """One all-gather of every rank's (global index, ids) pairs as a fixed-shape int32 tensor [rows, 2 + max_len] (column 0 the
    global index or -1 for padding rows, column 1 the length): the same buffer shape on every rank, as RCCL needs -- on the
    current CUDA device under the nccl backend (the caller's process has set it: LLM.__init__ does), on the CPU under gloo."""
    import torch
    import torch.distributed as dist

    on_gpu = dist.get_backend() == "nccl"
    buf = torch.full((rows, 2 + max_len), -1, dtype=torch.int32)
    for r, (i, toks) in enumerate(zip(idxs, lists)):
        if len(toks) > max_len:
            raise ValueError(f"utterance {i} has {len(toks)} tokens, more than the gather width {max_len}")
        buf[r, 0] = i; buf[r, 1] = len(toks)
        buf[r, 2:2 + len(toks)] = torch.tensor(list(toks), dtype=torch.int32)
    if on_gpu:
        buf = buf.cuda()
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    for t in out:
        t = t.cpu()
        for r in range(rows):
            i, n = int(t[r, 0]), int(t[r, 1])
            if i >= 0:
                yield i, t[r, 2:2 + n].tolist()


def generate_data_parallel(llm, prompts: Sequence, sampling_params, rank: int, world: int, gather: bool = True):
    """`LLM.generate` over a list of utterances sharded across `world` engine processes (one per GPU).

    Every rank calls this with the SAME prompt list; each generates only its shard, with RNG streams keyed by the utterance's
    global index (`uids`; a request that carries its own SamplingParams.seed keeps that seed's stream, as in `LLM.generate`), and
    the token ids are exchanged once at the end (one all-gather of a fixed-shape int32 tensor, a few hundred KB) -- nothing crosses
    GPUs while the engines step.  Returns a list aligned with `prompts` of offset-space token-id lists
    (None for other ranks' utterances when gather=False).  The reference has no multi-GPU path; this is SURVEY.md 8(e)."""
    sps = list(sampling_params) if isinstance(sampling_params, (list, tuple)) else [sampling_params] * len(prompts)

    def cost(p, sp):
        n_text = len(p["prompt_token_ids"]) if isinstance(p, dict) and "prompt_token_ids" in p else len(str(p.get("prompt", ""))) if isinstance(p, dict) else len(str(p))
        return float(n_text + (sp.max_tokens or llm.max_model_len))

    costs = [cost(p, sp) for p, sp in zip(prompts, sps)]

    def run(idxs):
        outs = llm.generate([prompts[i] for i in idxs], [sps[i] for i in idxs], uids=idxs)
        return [o.outputs[0].token_ids for o in outs]

    max_len = max(int(sp.max_tokens or llm.max_model_len) for sp in sps) if sps else 1
    return generate_sharded(run, len(prompts), costs, rank, world, gather, max_len=min(max_len, llm.max_model_len))[1]
