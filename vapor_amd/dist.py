"""Locus sharding across the GPUs of one node and the all-gather of per-locus results
(SURVEY.md §8e).

Loci are independent, so there is no data-path collective: each rank scores its share and one
all-gather of fixed-size float64 records (RCCL over xGMI when the ranks own GPUs, gloo on CPU in
the tests) gives every rank the complete table.  Record layout (RECORD_WIDTH float64):
    [0] = n scores (or -1: the locus ended in an exception, sent separately as an object)
    [1 .. 1+n) = the read scores in read order.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import numpy as np

RECORD_WIDTH = 32      # the reference keeps at most 20 reads per locus (SF:1091)

_pg = None


def init_from_env(backend: str = None) -> None:
    """Join the process group torchrun describes (no-op for a single process)."""
    global _pg
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or _pg is not None:
        return
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if not dist.is_initialized():
        dist.init_process_group(backend)
    _pg = backend


def finalize() -> None:
    global _pg
    if _pg is not None:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        _pg = None


def world() -> int:
    if _pg is None:
        return 1
    import torch.distributed as dist
    return dist.get_world_size()


def rank() -> int:
    if _pg is None:
        return 0
    import torch.distributed as dist
    return dist.get_rank()


def partition(costs: Sequence[float], n_parts: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of items to parts; each part keeps its items in
    input order."""
    order = sorted(range(len(costs)), key=lambda t: (-costs[t], t))
    load = [0.0] * n_parts
    parts: List[List[int]] = [[] for _ in range(n_parts)]
    for t in order:
        p = min(range(n_parts), key=lambda q: (load[q], q))
        parts[p].append(t)
        load[p] += costs[t]
    return [sorted(p) for p in parts]


def my_share(n_items: int, costs: Sequence[float] = None) -> List[int]:
    w = world()
    if w == 1:
        return list(range(n_items))
    if costs is None:
        costs = [1.0] * n_items
    return partition(costs, w)[rank()]


def pack_records(local: Dict[int, object], n_items: int):
    """(records (n_items, RECORD_WIDTH) with NaN rows for other ranks' items, {index: object}
    for what does not fit a record)."""
    rec = np.full((n_items, RECORD_WIDTH), np.nan, dtype=np.float64)
    extra = {}
    for t, v in local.items():
        if isinstance(v, BaseException) or v is None or len(v) > RECORD_WIDTH - 1:
            rec[t, 0] = -1
            extra[t] = v
        else:
            rec[t, 0] = len(v)
            rec[t, 1:1 + len(v)] = v
    return rec, extra


def unpack_records(rec: np.ndarray, extra: Dict[int, object]) -> List[object]:
    out: List[object] = []
    for t in range(rec.shape[0]):
        n = rec[t, 0]
        if n == -1:
            out.append(extra[t])
        else:
            out.append([float(x) for x in rec[t, 1:1 + int(n)]])
    return out


def gather_results(local: Dict[int, object], n_items: int) -> List[object]:
    """Every rank contributes the items it scored; every rank gets the full list back."""
    rec, extra = pack_records(local, n_items)
    if _pg is None:
        return unpack_records(rec, extra)
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if _pg == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(rec).to(dev)
    nw = dist.get_world_size()
    allr = torch.empty((nw * mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=dev)
    dist.all_gather_into_tensor(allr, mine.contiguous())
    allr = allr.cpu().numpy().reshape(nw, mine.shape[0], mine.shape[1])
    # each item was scored by exactly one rank: take the row that is not NaN
    have = ~np.isnan(allr[:, :, 0])
    owner = have.argmax(axis=0)
    assert have.sum(axis=0).min() == 1 and have.sum(axis=0).max() == 1, "every locus must be scored exactly once"
    merged = allr[owner, np.arange(n_items)]
    extras = [None] * dist.get_world_size()
    dist.all_gather_object(extras, extra)
    all_extra = {}
    for e in extras:
        all_extra.update(e)
    return unpack_records(merged, all_extra)
