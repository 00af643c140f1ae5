"""Locus sharding across the GPUs of one node and the all-gather of per-locus results
(SURVEY.md §8e).

Loci are independent, so there is no data-path collective: each rank scores its share and one
all-gather of the ranks' score vectors (RCCL over xGMI when every rank owns a GPU; gloo on CPU in the tests and
between ranks that share a GPU) gives every rank the complete table: per locus [n, score_1 .. score_n] in read order (n = -1: the locus
ended in an exception, which is sent separately as an object), each rank's own loci only.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import numpy as np

_pg = None
_files = None           # the "files" backend: {"dir", "rank", "world", "step"}


def _device_ordinal() -> int:
    """LOCAL_RANK folded onto the GPUs present (several ranks may share one GPU)."""
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if lr == 0:
        return 0                       # (a single process never pays the import of torch: 1.5 s of a short run)
    told = os.environ.get("VAPOR_LOCAL_GPUS")          # (the workflow launcher says how many GPUs it was given)
    if told and int(told) > 0:
        return lr % int(told)
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:       # noqa: BLE001
        n = 0
    return lr % n if n > 0 else lr


def init_from_env(backend: str = None) -> None:
    """Join the process group torchrun describes (no-op for a single process)."""
    global _pg, _files
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or _pg is not None:
        return
    if (backend or os.environ.get("VAPOR_DIST_BACKEND")) == "files":
        # Ranks that the workflow launcher started itself on one node (several per GPU: host-side parallelism).  Their only
        # exchange is the gather of a few kilobytes of scores per rank at the end of a run: a file per rank in a directory
        # of the launcher's, no process group - and no import of torch, which costs such a run more than its scoring.
        _files = {"dir": os.environ["VAPOR_DIST_DIR"], "rank": int(os.environ.get("RANK", "0")), "world": world, "step": 0}
        _pg = "files"
        return
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = os.environ.get("VAPOR_DIST_BACKEND")
    if backend is None:
        # RCCL wants a GPU per rank; ranks that share a GPU (host-side parallelism: the per-locus Python, BAM
        # decompression) exchange their few kilobytes of scores through gloo
        n_gpu = torch.cuda.device_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        backend = "nccl" if n_gpu > 0 and local_world <= n_gpu else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(_device_ordinal())
    if not dist.is_initialized():
        dist.init_process_group(backend)
    _pg = backend


def adopt_process_group(backend: str) -> None:
    """Use the torch.distributed process group the caller has already initialised (bench.py's ranks) for the gather."""
    global _pg
    _pg = backend


def release_process_group() -> None:
    """Forget an adopted group without destroying it (it is the caller's)."""
    global _pg
    _pg = None


def finalize() -> None:
    global _pg, _files
    if _pg == "files":
        _pg = _files = None            # (the directory is the launcher's)
        return
    if _pg is not None:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        _pg = None


def world() -> int:
    if _pg is None:
        return 1
    if _pg == "files":
        return _files["world"]
    import torch.distributed as dist
    return dist.get_world_size()


def rank() -> int:
    if _pg is None:
        return 0
    if _pg == "files":
        return _files["rank"]
    import torch.distributed as dist
    return dist.get_rank()


def _gather_through_files(flat: np.ndarray, extra: Dict[int, object]):
    """Every rank's (vector, objects) through the launcher's directory: written under a temporary name and renamed, so a
    reader sees a whole file or none; a rank that waits gives up when the launcher has marked the run as failed (a rank
    ended badly) or after VAPOR_DIST_TIMEOUT seconds."""
    import pickle
    import time
    d, me, nw = _files["dir"], _files["rank"], _files["world"]
    step = _files["step"]
    _files["step"] += 1
    tmp = os.path.join(d, "s%d_r%d.tmp" % (step, me))
    with open(tmp, "wb") as f:
        pickle.dump((np.ascontiguousarray(flat, dtype=np.float64), extra), f, protocol=4)
    os.replace(tmp, os.path.join(d, "s%d_r%d.bin" % (step, me)))
    out = []
    deadline = time.monotonic() + float(os.environ.get("VAPOR_DIST_TIMEOUT", "86400"))
    for r in range(nw):
        path = os.path.join(d, "s%d_r%d.bin" % (step, r))
        pause = 0.0005
        while not os.path.exists(path):
            if os.path.exists(os.path.join(d, "abort")):
                raise RuntimeError("vapor_amd.dist: another rank of this run failed")
            if time.monotonic() > deadline:
                raise RuntimeError("vapor_amd.dist: rank %d did not deliver its scores" % r)
            time.sleep(pause)
            pause = min(pause * 2, 0.05)
        with open(path, "rb") as f:
            out.append(pickle.load(f))
    return out


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


def pack_records(local: Dict[int, object], items: Sequence[int]):
    """This rank's share as one flat float64 vector - per item, in `items` order, [n, score_1 .. score_n], or [-1]
    for an item that ended in an exception (or produced no list), which travels separately as an object - and
    {index: object} for those."""
    flat: List[float] = []
    extra = {}
    for t in items:
        v = local[t]
        if isinstance(v, BaseException) or v is None:
            flat.append(-1.0)
            extra[t] = v
        else:
            flat.append(float(len(v)))
            flat.extend(v)
    return np.asarray(flat, dtype=np.float64), extra


def unpack_records(flat: np.ndarray, items: Sequence[int], extra: Dict[int, object], out: List[object]) -> None:
    p = 0
    fl = flat.tolist()
    for t in items:
        n = int(fl[p])
        p += 1
        if n < 0:
            out[t] = extra[t]
        else:
            out[t] = fl[p:p + n]
            p += n


def gather_results(local: Dict[int, object], n_items: int, costs: Sequence[float] = None) -> List[object]:
    """Every rank contributes the items it scored (its share of `my_share(n_items, costs)`); every rank gets the
    full list back.  What travels: one all-gather of (vector length, number of exceptions) per rank, then one
    all-gather of the ranks' own score vectors padded to the longest - a few hundred bytes per locus, no NaN rows
    for other ranks' loci, and no pickling unless some locus raised."""
    out: List[object] = [None] * n_items
    if _pg is None:                               # one process: the lists as they are (as floats, as the vector would give them)
        for t in range(n_items):
            v = local[t]
            out[t] = v if isinstance(v, BaseException) or v is None else [float(x) for x in v]
        return out
    if _pg == "files":
        shares = partition([1.0] * n_items if costs is None else costs, _files["world"])
        flat, extra = pack_records(local, shares[_files["rank"]])
        all_extra: Dict[int, object] = {}
        got = _gather_through_files(flat, extra)
        for _v, e in got:
            all_extra.update(e)
        for r, (v, _e) in enumerate(got):
            unpack_records(v, shares[r], all_extra, out)
        return out
    import torch
    import torch.distributed as dist
    nw = dist.get_world_size()
    shares = partition([1.0] * n_items if costs is None else costs, nw)
    flat, extra = pack_records(local, shares[dist.get_rank()])
    dev = torch.device("cuda", _device_ordinal()) if _pg == "nccl" else torch.device("cpu")
    head = torch.tensor([len(flat), len(extra)], dtype=torch.int64, device=dev)
    heads = torch.empty(nw * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(heads, head)
    heads = heads.cpu().numpy().reshape(nw, 2)
    width = max(int(heads[:, 0].max()), 1)
    mine = torch.zeros(width, dtype=torch.float64, device=dev)
    mine[:len(flat)] = torch.from_numpy(flat).to(dev)
    allv = torch.empty(nw * width, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allv, mine)
    allv = allv.cpu().numpy().reshape(nw, width)
    all_extra: Dict[int, object] = {}
    if int(heads[:, 1].sum()) > 0:              # some locus ended in an exception: those objects are pickled
        extras = [None] * nw
        dist.all_gather_object(extras, extra)
        for e in extras:
            all_extra.update(e)
    for r in range(nw):
        unpack_records(allv[r, :int(heads[r, 0])], shares[r], all_extra, out)
    return out
