"""Data-parallel sharding of utterances across the GPUs of a node.

The path shards embarrassingly: utterances share nothing but read-only weights, so the only collective is ONE
broadcast of the packed weight blob from rank 0 at start-up (RCCL over xGMI on GPUs; gloo on CPU in the tests).
"""
from __future__ import annotations

import numpy as np


def shard_utterances(lengths, world_size):
    """Assign utterance indices to ranks, balancing the expected work (longest-first greedy).
    lengths: expected output length of every utterance -> list (per rank) of index lists, each sorted."""
    order = np.argsort(-np.asarray(lengths, np.int64), kind="stable")
    load = np.zeros(world_size, np.int64)
    out = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += int(lengths[i])
    return [sorted(x) for x in out]


def broadcast_blob(blob, src=0):
    """Broadcast the packed weight tensor in place (no-op without an initialised process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def gather_results(local_items, world_size):
    """all_gather_object of per-rank result lists (small host objects: lengths, timings)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world_size == 1:
        return [local_items]
    out = [None] * world_size
    dist.all_gather_object(out, local_items)
    return out
