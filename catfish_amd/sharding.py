"""Multi-GPU: reads shard across ranks, one process per GPU, no collective on the data path.

Every 35-sample window is independent (zero GRU state, window-local padding:
rnn_class.py:159,170-171) and every read is independent (catfish/catfish:55-56), so the
reference's sequential per-file loop partitions into disjoint shards.  Each rank runs its
shard on its own MI355X with its own copy of the 0.79 MB weights; the only communication is
the final HOST gather of the per-read results to rank 0 (pickled Python objects over a gloo
group) -- RCCL/xGMI is never on the critical path.
"""
from __future__ import annotations

import numpy as np

from .infer import WINDOW_SIZE, padding_size_for


def windows_of(length, window=WINDOW_SIZE):
    return (int(length) + padding_size_for(int(length), window)) // window


def shard_reads(lengths, world_size):
    """Greedy longest-processing-time partition of read indices by window count.

    Returns ``world_size`` lists of indices (each ascending).  Equal-length reads degenerate to
    near-equal contiguous-count shards.
    """
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    cost = np.array([windows_of(n) for n in lengths], dtype=np.int64)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += cost[i]
    return [sorted(s) for s in shards]


def run_sharded(signals, infer_fn, rank=None, world_size=None, gather_group=None):
    """Run ``infer_fn(list of signals) -> list of results`` on this rank's shard and gather on rank 0.

    All ranks pass the same ``signals`` list (or at least the same lengths: a rank only touches
    its own shard's entries).  Returns the full result list in input order on rank 0 and None on
    the other ranks.  Without an initialised process group it simply runs everything locally.
    """
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank() if distributed else 0
    if world_size is None:
        world_size = dist.get_world_size() if distributed else 1
    shards = shard_reads([len(s) if s is not None else 0 for s in signals], world_size)
    mine = shards[rank]
    local = infer_fn([signals[i] for i in mine]) if mine else []
    if len(local) != len(mine):
        raise RuntimeError("infer_fn returned %d results for %d reads" % (len(local), len(mine)))
    if world_size == 1:
        out = [None] * len(signals)
        for i, res in zip(mine, local):
            out[i] = res
        return out
    if not distributed:
        raise RuntimeError("world_size > 1 needs an initialised torch.distributed process group")
    gathered = [None] * world_size if rank == 0 else None
    dist.gather_object(list(zip(mine, local)), gathered, dst=0, group=gather_group)
    if rank != 0:
        return None
    out = [None] * len(signals)
    for part in gathered:
        for i, res in part:
            out[i] = res
    return out


def host_gather_group():
    """A gloo group for the final host gather (object pickles travel over TCP/shared memory, never
    through RCCL); falls back to the default group when that already is gloo."""
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        return None
    return dist.new_group(backend="gloo")
