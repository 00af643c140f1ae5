"""Thin object layer over the C ABI: contexts, resident sequence sets and plans.

Device memory stays inside libvapor_hip.so; numpy arrays cross the boundary.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


_utf8 = ctypes.pythonapi.PyUnicode_AsUTF8AndSize
_utf8.restype = ctypes.c_void_p
_utf8.argtypes = [ctypes.py_object, ctypes.POINTER(ctypes.c_ssize_t)]


def _ascii_data_offset() -> int:
    """Where the characters of a compact ASCII str lie relative to id(str) in this interpreter (measured, not assumed)."""
    probe = "ACGTACGTACGTACGT" * 4
    size = ctypes.c_ssize_t()
    p = _utf8(probe, ctypes.byref(size))
    return (p - id(probe)) if p and size.value == len(probe) else -1


_ASCII_OFF = _ascii_data_offset()


def _as_bytes(s) -> bytes:
    return s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1", "replace")


class SeqSet:
    """Sequences packed on the device.  `n_exc[s]` = symbols outside upper-case ACGT,
    `n_invalid[s]` = symbols outside invert_base's alphabet after IUPAC folding."""

    def __init__(self, engine: "Engine", seqs: Sequence, upper: Optional[Sequence[bool]] = None, derived=None):
        """`derived`: sequences described instead of uploaded (include/vapor_hip.h, vapor_seqset_create_derived) - a list of
        (segments, upper) with segments = [(parent index into `seqs`, off, len, revcomp), ...]; derived sequence d is index
        len(seqs) + d of the set.  The device assembles their planes from the parents'."""
        self.engine = engine
        self.n_lit = len(seqs)
        self.n = len(seqs)
        n1 = max(self.n, 1)
        # One pointer per sequence, no concatenated copy: an ASCII str is handed over as it lies in memory (CPython
        # keeps it one byte per character; PyUnicode_AsUTF8AndSize returns that buffer), anything else as the
        # bytes it is or encodes to (characters outside Latin-1 become '?', which matches nothing anyway).
        keep = []
        self.lens = np.zeros(self.n, dtype=np.int32)
        size = ctypes.c_ssize_t()
        fast = (self.n > 64 and 0 < _ASCII_OFF < 256 and isinstance(seqs, (list, tuple)) and set(map(type, seqs)) == {str}
                and all(map(str.isascii, seqs)))
        if fast:
            # all ASCII str (the usual case): the characters of every one lie at the same offset behind the object, so the
            # pointers are id() + offset - four passes of map() instead of a ctypes call per sequence (2 200 sequences:
            # 1.1 ms -> 0.3 ms of a 2.4 ms upload)
            addr = np.fromiter(map(id, seqs), dtype=np.uint64, count=self.n) + np.uint64(_ASCII_OFF)
            self.lens = np.fromiter(map(len, seqs), dtype=np.int32, count=self.n)
            for t in (0, self.n // 2, self.n - 1):          # (spot check against the interpreter's own answer)
                if _utf8(seqs[t], ctypes.byref(size)) != int(addr[t]) or size.value != int(self.lens[t]):
                    fast = False
        if fast:
            ptrs = ctypes.cast(addr.ctypes.data, ctypes.POINTER(ctypes.c_void_p))
            keep.append(addr)
        else:
            ptrs = (ctypes.c_void_p * n1)()
        for t, sq in enumerate(() if fast else seqs):
            if isinstance(sq, str):
                p = _utf8(sq, ctypes.byref(size))
                if p and size.value == len(sq):
                    ptrs[t] = p
                    self.lens[t] = len(sq)
                    continue
                sq = sq.encode("latin-1", "replace")
            elif not isinstance(sq, bytes):
                sq = bytes(sq)
            keep.append(sq)
            ptrs[t] = ctypes.cast(ctypes.c_char_p(sq), ctypes.c_void_p).value
            self.lens[t] = len(sq)
        flags = np.zeros(n1, dtype=np.uint8)
        if upper is not None:
            flags[:self.n] = np.asarray(upper, dtype=bool).astype(np.uint8) * L.SEQ_UPPER
        h = ctypes.c_void_p()
        lib = L.load()
        lens = self.lens if self.n else np.zeros(1, np.int32)
        if derived:
            nd = len(derived)
            if isinstance(derived, tuple) and len(derived) == 3 and isinstance(derived[0], np.ndarray):
                seg_first, segs, dflags = derived                      # (already as arrays: pipeline builds them in one go)
                nd = len(seg_first) - 1
            else:
                seg_first = np.zeros(nd + 1, dtype=np.int32)
                np.cumsum([len(sg) for sg, _u in derived], out=seg_first[1:])
                segs = np.zeros(max(int(seg_first[-1]), 1), dtype=L.SEG_DTYPE)
                w = 0
                for sg, _u in derived:
                    for par, off, ln, rc in sg:
                        segs[w] = (par, off, ln, L.SEG_REVCOMP if rc else 0)
                        w += 1
                dflags = np.asarray([L.SEQ_UPPER if u else 0 for _sg, u in derived], dtype=np.uint8)
            info = np.zeros(2 * (self.n + nd), dtype=np.int32)
            L.check(lib.vapor_seqset_create_derived(engine._ctx, self.n, ptrs, L.ptr(lens, ctypes.c_int32), L.ptr(flags, ctypes.c_uint8),
                                                    nd, L.ptr(np.ascontiguousarray(seg_first, dtype=np.int32), ctypes.c_int32),
                                                    np.ascontiguousarray(segs, dtype=L.SEG_DTYPE).ctypes.data_as(ctypes.c_void_p),
                                                    L.ptr(np.ascontiguousarray(dflags, dtype=np.uint8), ctypes.c_uint8),
                                                    L.ptr(info, ctypes.c_int32), ctypes.byref(h)))
            dl = np.zeros(nd, dtype=np.int32)
            np.add.at(dl, np.repeat(np.arange(nd), np.diff(seg_first)), segs["len"][:int(seg_first[-1])])
            self.lens = np.concatenate([self.lens[:self.n], dl])
            self.n += nd
        else:
            info = np.zeros(2 * n1, dtype=np.int32)
            L.check(lib.vapor_seqset_create_ptrs(engine._ctx, self.n, ptrs, L.ptr(lens, ctypes.c_int32),
                                                 L.ptr(flags, ctypes.c_uint8), L.ptr(info, ctypes.c_int32), ctypes.byref(h)))
        del keep
        self._h = h
        engine._live.add(self)
        self.n_exc = info[0::2][:self.n].copy()
        self.n_invalid = info[1::2][:self.n].copy()

    @classmethod
    def from_addresses(cls, engine: "Engine", addr: np.ndarray, lens: np.ndarray, derived=None, keepalive=None,
                       src_kind=None, src_first=None) -> "SeqSet":
        """A set whose byte sequences are given as (address, length) pairs - slices of strings and buffers the caller keeps
        alive (`keepalive`) until this returns: a read is a slice of its record's sequence, a window a slice of its contig;
        nothing is copied on the Python side.  `derived` as (seg_first, segs, flags) arrays.  `src_kind` (uint8 per sequence;
        vapor_seqset_create_mixed): 1 where `addr` is the DEVICE address of BAM-packed bases inside a live BamBatch and the
        sequence the `lens` bases from base `src_first` on (Engine.bam_chop_device's reads); those never cross the link."""
        self = cls.__new__(cls)
        self.engine = engine
        self.n_lit = self.n = int(len(addr))
        addr = np.ascontiguousarray(addr, dtype=np.uint64)
        self.lens = np.ascontiguousarray(lens, dtype=np.int32)
        ptrs = ctypes.cast(addr.ctypes.data, ctypes.POINTER(ctypes.c_void_p))
        h = ctypes.c_void_p()
        lib = L.load()
        nd = 0
        if derived is not None:
            seg_first, segs, dflags = derived
            nd = len(seg_first) - 1
        info = np.zeros(2 * max(self.n + nd, 1), dtype=np.int32)
        lens_p = L.ptr(self.lens if self.n else np.zeros(1, np.int32), ctypes.c_int32)
        if src_kind is not None:
            kind = np.ascontiguousarray(src_kind, dtype=np.uint8)
            first = np.ascontiguousarray(src_first, dtype=np.int64)
            if len(kind) != self.n or len(first) != self.n:
                raise ValueError("src_kind / src_first: one entry per sequence")
            if nd:
                seg_first = np.ascontiguousarray(seg_first, dtype=np.int32)
                segs = np.ascontiguousarray(segs, dtype=L.SEG_DTYPE)
                dfl = np.ascontiguousarray(dflags, dtype=np.uint8)
            L.check(lib.vapor_seqset_create_mixed(engine._ctx, self.n, ptrs, lens_p, None, L.ptr(kind, ctypes.c_uint8), first.ctypes.data_as(ctypes.c_void_p),
                                                  nd, L.ptr(seg_first, ctypes.c_int32) if nd else None,
                                                  segs.ctypes.data_as(ctypes.c_void_p) if nd else None,
                                                  L.ptr(dfl, ctypes.c_uint8) if nd else None, L.ptr(info, ctypes.c_int32), ctypes.byref(h)))
            if nd:
                dl = np.zeros(nd, dtype=np.int32)
                np.add.at(dl, np.repeat(np.arange(nd), np.diff(seg_first)), segs["len"][:int(seg_first[-1])])
                self.lens = np.concatenate([self.lens, dl])
                self.n += nd
        elif nd:
            seg_first = np.ascontiguousarray(seg_first, dtype=np.int32)
            segs = np.ascontiguousarray(segs, dtype=L.SEG_DTYPE)
            L.check(lib.vapor_seqset_create_derived(engine._ctx, self.n, ptrs, lens_p, None, nd, L.ptr(seg_first, ctypes.c_int32),
                                                    segs.ctypes.data_as(ctypes.c_void_p), L.ptr(np.ascontiguousarray(dflags, dtype=np.uint8), ctypes.c_uint8),
                                                    L.ptr(info, ctypes.c_int32), ctypes.byref(h)))
            dl = np.zeros(nd, dtype=np.int32)
            np.add.at(dl, np.repeat(np.arange(nd), np.diff(seg_first)), segs["len"][:int(seg_first[-1])])
            self.lens = np.concatenate([self.lens, dl])
            self.n += nd
        else:
            L.check(lib.vapor_seqset_create_ptrs(engine._ctx, self.n, ptrs, lens_p, None, L.ptr(info, ctypes.c_int32), ctypes.byref(h)))
        del keepalive
        self._h = h
        engine._live.add(self)
        self.n_exc = info[0::2][:self.n].copy()
        self.n_invalid = info[1::2][:self.n].copy()
        return self

    def planes(self, idx: int):
        """(p2, e1, x4) uint32 arrays of sequence `idx` as they lie in HBM: 2, 1 and 4 words per 32-symbol chunk."""
        ch = (int(self.lens[idx]) + 31) // 32
        p2, e1, x4 = np.zeros(2 * ch, np.uint32), np.zeros(ch, np.uint32), np.zeros(4 * ch, np.uint32)
        L.check(L.load().vapor_seqset_planes(self._h, int(idx), p2.ctypes.data_as(ctypes.c_void_p), e1.ctypes.data_as(ctypes.c_void_p),
                                             x4.ctypes.data_as(ctypes.c_void_p)))
        return p2, e1, x4

    def close(self) -> None:
        if self._h:
            L.load().vapor_seqset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BamBatch:
    """The inflated blocks of one vapor_bam_chop_device call on the device (the reads' packed bases lie in them)."""

    def __init__(self, handle):
        self._h = handle

    def close(self) -> None:
        if self._h:
            L.load().vapor_bam_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass


class Plan:
    def __init__(self, engine: "Engine", seqset: SeqSet, pairs: np.ndarray):
        self.engine = engine
        self.seqset = seqset
        self.pairs = np.ascontiguousarray(pairs, dtype=L.PAIR_DTYPE)
        self.n = len(self.pairs)
        h = ctypes.c_void_p()
        L.check(L.load().vapor_plan_create(engine._ctx, seqset._h, self.n, self.pairs.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.byref(h)))
        self._h = h
        engine._live.add(self)
        self.stats = np.zeros((max(self.n, 1), L.STATS_STRIDE), dtype=np.int64)

    def run(self) -> np.ndarray:
        """One pass of the hot path; returns the (n_pairs, 16) int64 statistics."""
        L.check(L.load().vapor_plan_run(self._h, L.ptr(self.stats, ctypes.c_int64)))
        return self.stats[:self.n]

    def timings(self) -> dict:
        ms = np.zeros(10, dtype=np.float64)
        L.check(L.load().vapor_plan_timings(self._h, L.ptr(ms, ctypes.c_double), 10))
        return {"join_ms": ms[0], "clean_ms": ms[1], "total_ms": ms[2], "join_launches": int(ms[3]),
                "retried_pairs": int(ms[4]), "finish_ms": ms[5], "pairs_served_by_shared_joins": int(ms[6]), "shared_joins": int(ms[7]),
                "clean_workgroups_per_cu": int(ms[8]), "remap_in_clean": int(ms[9])}

    def record_counts(self) -> np.ndarray:
        """Run records per pair of the last run (the device stores runs of consecutive dots as one record)."""
        out = np.zeros(max(self.n, 1), dtype=np.int64)
        L.check(L.load().vapor_plan_record_counts(self._h, L.ptr(out, ctypes.c_int64)))
        return out[:self.n]

    def set_reads(self, reads: np.ndarray, n_loci: int) -> None:
        """Describe which pairs score which read of which locus (READ_DTYPE rows sorted by locus) so that
        run_loci() can finish on the device."""
        from . import finish
        self.reads = np.ascontiguousarray(reads, dtype=L.READ_DTYPE)
        self.n_loci = int(n_loci)
        L.check(L.load().vapor_plan_set_reads(self._h, len(self.reads), self.reads.ctypes.data_as(ctypes.c_void_p),
                                              self.n_loci, L.ptr(finish.gt_table(), ctypes.c_double)))
        self.loci = np.zeros((max(self.n_loci, 1), L.LOCUS_STRIDE), dtype=np.float64)
        self.read_scores = np.zeros(max(len(self.reads), 1), dtype=np.float64)

    def run_loci(self, device_out: int = 0, want_host: bool = True, want_scores: bool = False):
        """join -> clean -> finish on the device.  Returns the (n_loci, 8) float64 records
        [QS, GS, GT index, GQ, n scored, n positive, n rounding to <= 0, 0] (NaN row = 'NA'); with
        `device_out` (a device pointer, e.g. tensor.data_ptr()) they are also written there."""
        L.check(L.load().vapor_plan_run_loci(self._h, ctypes.c_void_p(device_out) if device_out else None,
                                             L.ptr(self.loci, ctypes.c_double) if want_host else None,
                                             L.ptr(self.read_scores, ctypes.c_double) if want_scores else None))
        return self.loci[:self.n_loci]

    def run_loci_async(self, device_out: int = 0) -> None:
        """Enqueue join -> clean -> finish without waiting (after one run_loci(), which sizes the slots)."""
        L.check(L.load().vapor_plan_run_loci_async(self._h, ctypes.c_void_p(device_out) if device_out else None))

    def then(self, hip_stream: int) -> None:
        """Make `hip_stream` (a caller's stream) wait on the device for this plan's most recently enqueued step."""
        L.check(L.load().vapor_plan_then(self._h, ctypes.c_void_p(hip_stream)))

    def after(self, hip_stream: int) -> None:
        """Make this plan's next step wait for what has been enqueued on `hip_stream` so far."""
        L.check(L.load().vapor_plan_after(self._h, ctypes.c_void_p(hip_stream)))

    def sync(self, want_host: bool = True):
        """Wait for the enqueued steps; timings() then holds their averages.  Returns the last step's records."""
        L.check(L.load().vapor_plan_sync(self._h, L.ptr(self.loci, ctypes.c_double) if want_host else None))
        return self.loci[:self.n_loci]

    def algorithmic(self) -> Tuple[int, int]:
        b = ctypes.c_int64()
        c = ctypes.c_int64()
        L.check(L.load().vapor_plan_algorithmic_bytes(self._h, ctypes.byref(b), ctypes.byref(c)))
        return b.value, c.value

    def fetch_hits(self, idx: Iterable[int], want_flags: bool = True):
        """(hits (m,2) int32 [j,i], flags (m,) uint8 or None, off (len(idx)+1,) int64)."""
        idx = np.ascontiguousarray(list(idx), dtype=np.int64)
        off = np.zeros(len(idx) + 1, dtype=np.int64)
        if len(idx) == 0:
            return np.zeros((0, 2), np.int32), (np.zeros(0, np.uint8) if want_flags else None), off
        tot = int(self.stats[idx, L.ST_N_HITS][self.stats[idx, L.ST_STATUS] == 0].sum())
        hits = np.zeros((max(tot, 1), 2), dtype=np.int32)
        flags = np.zeros(max(tot, 1), dtype=np.uint8) if want_flags else None
        L.check(L.load().vapor_plan_fetch_hits(self._h, len(idx), L.ptr(idx, ctypes.c_int64), L.ptr(hits, ctypes.c_int32),
                                               L.ptr(flags, ctypes.c_uint8) if want_flags else None, tot,
                                               L.ptr(off, ctypes.c_int64)))
        return hits[:tot], (flags[:tot] if want_flags else None), off

    def close(self) -> None:
        if self._h:
            L.load().vapor_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One HIP context (device + stream).  Raises if the library or the GPU is missing."""

    def __init__(self, device: int = 0):
        lib = L.load()
        self._ctx = ctypes.c_void_p()
        L.check(lib.vapor_init(device, ctypes.byref(self._ctx)))
        self.device = device
        self._live = weakref.WeakSet()      # sequence sets and plans of this context: closed with it, plans first

    def set_param(self, name: str, value: int) -> None:
        L.check(L.load().vapor_set_param(self._ctx, name.encode(), int(value)))

    def set_stream(self, hip_stream: int) -> None:
        """Enqueue on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream); 0: the library's own."""
        L.check(L.load().vapor_set_stream(self._ctx, ctypes.c_void_p(hip_stream) if hip_stream else None))

    def seqset(self, seqs: Sequence, upper: Optional[Sequence[bool]] = None, derived=None) -> SeqSet:
        return SeqSet(self, seqs, upper, derived)

    def bam_chop_device(self, native_bam, tids, starts, ends, flanks, chunk_first, chunks, max_keep: int = 20):
        """vapor_bam_chop_device: the read selection of many regions of an open BAM file on the device.  Returns (kept_first,
        device addresses of the kept reads' packed bases, q0, miss_bp, status per region, BamBatch); the batch owns the data the
        addresses point into - close it after the sequence sets made from them."""
        n = len(tids)
        tids = np.ascontiguousarray(tids, dtype=np.int32)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        ends = np.ascontiguousarray(ends, dtype=np.int64)
        flanks = np.ascontiguousarray(flanks, dtype=np.int64)
        chunk_first = np.ascontiguousarray(chunk_first, dtype=np.int32)
        chunks = np.ascontiguousarray(chunks, dtype=np.uint64).reshape(-1)
        if len(chunk_first) != n + 1 or len(chunks) != 2 * int(chunk_first[-1]):
            raise ValueError("chunk_first / chunks do not describe %d regions" % n)
        kept_first = np.zeros(n + 1, dtype=np.int32)
        cap = max(n * max_keep, 1)
        addr = np.zeros(cap, dtype=np.uint64)
        q0 = np.zeros(cap, dtype=np.int64)
        miss = np.zeros(cap, dtype=np.int64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        h = ctypes.c_void_p()
        vp = ctypes.c_void_p
        L.check(L.load().vapor_bam_chop_device(self._ctx, native_bam, n, tids.ctypes.data_as(vp), starts.ctypes.data_as(vp), ends.ctypes.data_as(vp),
                                               flanks.ctypes.data_as(vp), chunk_first.ctypes.data_as(vp), chunks.ctypes.data_as(vp) if len(chunks) else None,
                                               int(max_keep), kept_first.ctypes.data_as(vp), addr.ctypes.data_as(vp), q0.ctypes.data_as(vp),
                                               miss.ctypes.data_as(vp), status.ctypes.data_as(vp), ctypes.byref(h)))
        w = int(kept_first[n])
        return kept_first, addr[:w], q0[:w], miss[:w], status[:n], BamBatch(h)

    def bam_last_stats(self) -> dict:
        """What this engine's last bam_chop_device did (vapor_bam_last_stats)."""
        out = np.zeros(6, dtype=np.float64)
        L.check(L.load().vapor_bam_last_stats(self._ctx, out.ctypes.data_as(ctypes.c_void_p), 6))
        return {"regions": int(out[0]), "blocks": int(out[1]), "compressed_bytes": int(out[2]), "inflated_bytes": int(out[3]),
                "inflate_ms": float(out[4]), "call_ms": float(out[5])}

    def seqset_raw(self, addr: np.ndarray, lens: np.ndarray, derived=None, keepalive=None, src_kind=None, src_first=None) -> SeqSet:
        """A set from (address, length) pairs - slices of strings the caller keeps alive - and derived sequences as arrays."""
        return SeqSet.from_addresses(self, addr, lens, derived, keepalive, src_kind, src_first)

    def plan(self, seqset: SeqSet, pairs: np.ndarray) -> Plan:
        return Plan(self, seqset, pairs)

    @staticmethod
    def make_pairs(rows: Sequence[Tuple[int, int, int, int, int]]) -> np.ndarray:
        a = np.zeros(len(rows), dtype=L.PAIR_DTYPE)
        for t, r in enumerate(rows):
            a[t] = tuple(r)
        return a

    def score(self, seqset: SeqSet, pairs: np.ndarray) -> np.ndarray:
        p = Plan(self, seqset, pairs)
        try:
            return p.run().copy()
        finally:
            p.close()

    def dotplots(self, seqset: SeqSet, pairs: np.ndarray) -> Tuple[np.ndarray, List[np.ndarray]]:
        """Statistics plus, per pair, the (n,2) [j,i] hit array sorted the way dotdata() lists it."""
        p = Plan(self, seqset, pairs)
        try:
            st = p.run().copy()
            hits, _f, off = p.fetch_hits(range(p.n), want_flags=False)
        finally:
            p.close()
        out = []
        for t in range(len(off) - 1):
            h = hits[off[t]:off[t + 1]]
            out.append(h[np.lexsort((h[:, 1], h[:, 0]))])
        return st, out

    def clean_hits(self, lists: Sequence[np.ndarray], flags: Optional[Sequence[int]] = None):
        """Cleaning + reductions on explicit dot lists -> (stats (n,16), [flag bytes per list])."""
        n = len(lists)
        arrs = [np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 2) for a in lists]
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(a) for a in arrs], out=off[1:])
        allh = np.concatenate(arrs + [np.zeros((1, 2), np.int32)])
        fl = np.asarray(flags if flags is not None else [3] * n, dtype=np.uint32)
        st = np.zeros((max(n, 1), 16), dtype=np.int64)
        hf = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        L.check(L.load().vapor_clean_hits(self._ctx, n, L.ptr(allh, ctypes.c_int32), L.ptr(off, ctypes.c_int64),
                                          L.ptr(fl if n else np.zeros(1, np.uint32), ctypes.c_uint32),
                                          L.ptr(st, ctypes.c_int64), L.ptr(hf, ctypes.c_uint8)))
        return st[:n], [hf[off[t]:off[t + 1]] for t in range(n)]

    def close(self) -> None:
        if self._ctx:
            live = list(self._live)
            for obj in sorted(live, key=lambda o: isinstance(o, SeqSet)):
                obj.close()
            L.load().vapor_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
