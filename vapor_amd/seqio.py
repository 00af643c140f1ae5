"""Reference/read extraction around an SV window (component #3).

The reference shells out to samtools twice or more per locus and parses the text
(`ref_seq_readin` SF:1203-1217, `chop_pacbio_read_by_pos` SF:339-354).  Here the same
text protocol is served by a pluggable backend so that whole shards of loci can be
prepared in-process before anything goes to the GPU:

* `SamtoolsCLI`     - the real `samtools faidx` / `samtools view` (when installed);
* `MemorySamtools`  - an in-memory world (`vapor_amd.synth.SynthWorld`) answering the
                      same two queries with the same line formats.

The trimming rules (POS filter, CIGAR walk, miss_bp cut, 20-read cap) keep the
reference's semantics exactly; they are host-side integer/string work.
"""
from __future__ import annotations

import os
import threading
import re
import shutil
import subprocess
from typing import Iterable, List, Optional

_backend = None


class SamtoolsCLI:
    """Runs the samtools binary; raises if it is missing instead of yielding nothing."""

    def __init__(self, exe: str = "samtools") -> None:
        self.exe = shutil.which(exe)
        if self.exe is None:
            raise RuntimeError("samtools not found on PATH; install it or select "
                               "vapor_amd.seqio.MemorySamtools via set_backend()")

    def faidx_lines(self, ref: str, region: str) -> Iterable[str]:
        p = subprocess.run([self.exe, "faidx", ref, region], capture_output=True, text=True)
        return p.stdout.splitlines()

    def view_lines(self, bam: str, region: str) -> Iterable[str]:
        p = subprocess.run([self.exe, "view", bam, region], capture_output=True, text=True)
        return p.stdout.splitlines()

    def isfile(self, path: str) -> bool:
        return os.path.isfile(path)

    def fai_lines(self, ref: str) -> Iterable[str]:
        with open(ref + ".fai") as f:
            return f.read().splitlines()


class FaiFasta:
    """In-process `samtools faidx ref chrom:start-end` through the .fai index (no process per locus)."""

    def __init__(self, path: str):
        self.path = path
        self.index = {}
        with open(path + ".fai") as f:
            for ln in f:
                t = ln.rstrip("\n").split("\t")
                if len(t) >= 5:
                    self.index[t[0]] = (int(t[1]), int(t[2]), int(t[3]), int(t[4]))
        self._fh = open(path, "rb")

    def fetch(self, chrom: str, start: int, end: int) -> str:
        """1-based inclusive, clipped to the contig like samtools."""
        if chrom not in self.index:
            return ""
        length, offset, linebases, linewidth = self.index[chrom]
        start = max(int(start), 1)
        end = min(int(end), length)
        if end < start:
            return ""
        a, b = start - 1, end
        first = offset + (a // linebases) * linewidth + a % linebases
        last = offset + ((b - 1) // linebases) * linewidth + (b - 1) % linebases + 1
        raw = os.pread(self._fh.fileno(), last - first, first)      # (positioned read: several threads may fetch at once)
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode("ascii")

    def lines(self, region: str) -> List[str]:
        """`samtools faidx` text: header line, then the sequence in lines of 60."""
        chrom, _, span = region.rpartition(":")
        if not chrom:
            chrom, a, b = region, 1, self.index.get(region, (0,))[0]
        else:
            a, _, b = span.partition("-")
            a, b = int(a.replace(",", "")), int(b.replace(",", ""))
        seq = self.fetch(chrom, a, b)
        return [">" + region] + [seq[i:i + 60] for i in range(0, len(seq), 60)]


class SamtoolsHybrid(SamtoolsCLI):
    """samtools for the BAM, the .fai index read in-process for the reference windows: one process per
    locus instead of two or more."""

    def __init__(self, exe: str = "samtools") -> None:
        super().__init__(exe)
        self._fa = {}

    def faidx_lines(self, ref: str, region: str) -> Iterable[str]:
        fa = self._fa.get(ref)
        if fa is None:
            if not os.path.exists(ref + ".fai"):
                return super().faidx_lines(ref, region)
            fa = self._fa[ref] = FaiFasta(ref)
        return fa.lines(region)


class InProcessBam(SamtoolsHybrid):
    """BAM and BAI read in-process as well (vapor_amd.bamio): no process per locus at all, and the records reach
    the trimming code as fields, not as text to be split again.  The default backend."""

    threads_ok = True                  # pipeline.run_batch may start loci on several threads (native chop, positioned reads)
    chunk_threads_ok = True            # cli.score_jobs may score several chunks at once, a thread each

    def __init__(self) -> None:        # noqa: D401 - does not require the samtools binary
        self.exe = None
        self._fa = {}
        self._bam = {}
        self._open_lock = threading.Lock()

    def _open(self, bam: str):
        from . import bamio
        b = self._bam.get(bam)
        if b is None:
            with self._open_lock:
                b = self._bam.get(bam)
                if b is None:
                    b = self._bam[bam] = bamio.BamFile(bam)
        return b

    def view_lines(self, bam: str, region: str) -> Iterable[str]:
        chrom, _, span = region.rpartition(":")
        a, _, e = span.partition("-")
        return self._open(bam).fetch_lines(chrom, int(a), int(e))

    def records(self, bam: str, chrom: str, start: int, end: int):
        """(QNAME, POS, CIGAR, SEQ) per alignment overlapping chrom:start-end."""
        return [r[:4] for r in self._open(bam).fetch_records(chrom, int(start), int(end))]

    def chop(self, bam: str, chrom: str, start: int, end: int, flank_length: int):
        """chop_pacbio_read_by_pos (SF:339-354) straight from the BAM file: the library's native reader
        (vapor_bam_chop), or with VAPOR_BAM_NATIVE=0 the Python statement of the same steps below."""
        if not _env_is(b"VAPOR_BAM_NATIVE", b"0"):
            return self._open(bam).chop_native(chrom, int(start), int(end), int(flank_length))
        return self.chop_python(bam, chrom, start, end, flank_length)

    def chop_many(self, bam: str, chroms, starts, ends, flanks, max_keep: int = 20):
        """MemorySamtools.chop_many's contract from a BAM file: every region through the library's native reader (vapor_bam_chop:
        threaded inflate, binary CIGAR walk, only the kept bases decoded) on a few threads, the kept reads of a region as slices
        of ONE text per region - (kept_first, addr, q0 = 0, miss, status, keepalive).  minimize_pacbio_read_list (SF:1091-1102)
        on the numbers: the first max_keep in a stable order by miss_bp."""
        import numpy as np
        from .engine import _ASCII_OFF
        if _env_is(b"VAPOR_BAM_NATIVE", b"0") or not (0 < _ASCII_OFF < 256):
            raise NotImplementedError("chop_many needs the native reader")
        n = len(chroms)
        b = self._open(bam)
        st, en, fl = [int(x) for x in starts], [int(x) for x in ends], [int(x) for x in flanks]

        def one(g):
            try:
                return b.chop_native_raw(chroms[g], st[g], en[g], fl[g])
            except IndexError:
                return IndexError                       # (a record without CIGAR: the drivers' route raises it where the reference does)
        from . import pipeline
        n_thr = pipeline._prefetch_threads(n)
        if n_thr > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=n_thr) as pool:
                got = list(pool.map(one, range(n), chunksize=max(1, n // (n_thr * 8))))
        else:
            got = [one(g) for g in range(n)]
        kept_first = np.zeros(n + 1, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        pa, pm, keep = [], [], []
        w = 0
        for g, r in enumerate(got):
            kept_first[g] = w
            if r is IndexError:
                status[g] = -4
                continue
            if r is None:
                continue
            whole, off, ln, miss = r
            order = np.arange(len(off))
            if len(order) > max_keep:
                order = np.argsort(miss, kind="stable")[:max_keep]
            keep.append(whole)
            pa.append((off[order] + (id(whole) + _ASCII_OFF)).astype(np.uint64))
            pm.append(miss[order])
            w += len(order)
        kept_first[n] = w
        addr = np.concatenate(pa) if pa else np.zeros(0, dtype=np.uint64)
        miss_a = np.concatenate(pm).astype(np.int64) if pm else np.zeros(0, dtype=np.int64)
        return kept_first, addr, np.zeros(w, dtype=np.int64), miss_a, status, keep

    def chop_many_device(self, engine, bam: str, chroms, starts, ends, flanks, max_keep: int = 20):
        """chop_many with the work on the device (vapor_bam_chop_device: the regions' BGZF blocks go over the link compressed,
        one wavefront inflates a block, one walks a region's records): (kept_first, DEVICE addresses of the kept reads' packed
        bases, q0 = first base of each read's part, miss, status, keepalive).  A region the device leaves to the host route
        (status != 0: a damaged block, a record without CIGAR, ...) is answered by the caller's per-locus route, which words
        the reference's errors."""
        import numpy as np
        if _env_is(b"VAPOR_BAM_NATIVE", b"0") or _env_is(b"VAPOR_BAM_DEVICE", b"0") or not hasattr(engine, "bam_chop_device"):
            raise NotImplementedError("no device reader")
        from . import _lib
        lib = _lib.load()
        if not hasattr(lib, "vapor_bam_chop_device"):
            raise NotImplementedError("no device reader")
        b = self._open(bam)
        n = len(chroms)
        tids = np.zeros(n, dtype=np.int32)
        chunk_first = np.zeros(n + 1, dtype=np.int32)
        flat = []
        index_chunks = b.index.chunks
        tid_of = b.tid
        for g in range(n):
            t = tid_of.get(chroms[g])
            if t is not None:
                tids[g] = t
                for c in index_chunks(t, max(int(starts[g]) - 1, 0), int(ends[g])):
                    flat.append(c[0])
                    flat.append(c[1])
            chunk_first[g + 1] = len(flat) >> 1
        # One call holds the blocks of all its regions on the device at once (compressed and inflated): regions in groups of
        # at most ~192 MB of compressed blocks (their inflated data stays far below the library's 2 GB a call), a group that the
        # library still refuses for its size in halves.
        flat_a = np.asarray(flat, dtype=np.uint64).reshape(-1, 2)
        comp = np.zeros(n, dtype=np.int64)
        if len(flat_a):
            per_chunk = ((flat_a[:, 1] >> np.uint64(16)) - (flat_a[:, 0] >> np.uint64(16))).astype(np.int64) + 65600
            np.add.at(comp, np.repeat(np.arange(n), np.diff(chunk_first)), per_chunk)
        cap = int(os.environ.get("VAPOR_BAM_DEVICE_BATCH_MB", "192")) << 20
        groups = []
        a = 0
        while a < n:
            e, tot = a, 0
            while e < n and (e == a or tot + int(comp[e]) <= cap):
                tot += int(comp[e])
                e += 1
            groups.append((a, e))
            a = e
        tl = b._take_handle(lib)
        parts, batches = [], []
        try:
            while groups:
                a, e = groups.pop(0)
                c0, c1 = int(chunk_first[a]), int(chunk_first[e])
                try:
                    got = engine.bam_chop_device(tl["native"], tids[a:e], starts[a:e], ends[a:e], flanks[a:e], chunk_first[a:e + 1] - c0,
                                                 flat_a[c0:c1].reshape(-1), max_keep)
                except _lib.VaporHipError as err:
                    if "in one call" in str(err) and e - a >= 2:
                        groups[:0] = [(a, (a + e) // 2), ((a + e) // 2, e)]
                        continue
                    if "in one call" in str(err):
                        # one region whose blocks alone are more than a call takes: the host route's (it streams them)
                        parts.append((np.zeros(2, dtype=np.int32), np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.int64),
                                      np.zeros(0, dtype=np.int64), np.ones(1, dtype=np.int32)))
                        continue
                    for bt in batches:
                        bt.close()
                    raise
                parts.append(got[:5])
                batches.append(got[5])
        finally:
            with b._lock:
                b._free.append(tl)
        kf = np.zeros(n + 1, dtype=np.int32)
        w = 0
        g = 0
        for p in parts:
            m = len(p[0]) - 1
            kf[g:g + m + 1] = p[0] + w
            w += int(p[0][-1])
            g += m
        cat = lambda k, dt: np.concatenate([p[k] for p in parts]) if parts else np.zeros(0, dtype=dt)    # noqa: E731
        return kf, cat(1, np.uint64), cat(2, np.int64), cat(3, np.int64), cat(4, np.int32), batches

    def isfile(self, path: str) -> bool:
        # (bam_in_decide, SF:69-89, asks once per locus: a file this reader holds open is a file - no stat, and no release of
        # the interpreter lock around one, for the loci after the first)
        return path in self._bam or os.path.isfile(path)

    def chop_python(self, bam: str, chrom: str, start: int, end: int, flank_length: int):
        """The same from the records as Python parses them: the CIGAR is walked in its binary form (the library's
        host helper) and only the reads that are kept have their bases decoded."""
        import ctypes
        import numpy as np
        from . import _lib, bamio
        walk = _lib.load().vapor_cigar2alignstart_ops
        res = np.zeros(2, dtype=np.int64)
        res_p = res.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        out = []
        for qname, pos, cig, sq, l_seq, _flag in self._open(bam).fetch_raw(chrom, int(start), int(end)):
            if not pos < start + 1:
                continue
            ops = np.ascontiguousarray(cig, dtype=np.uint32)
            rc = walk(ops.ctypes.data, len(ops), int(pos), int(start), res_p)
            if rc != 0:
                raise IndexError("string index out of range")  # what '' [1] raises in SF:331 for a record without CIGAR
            q0, miss_bp = int(res[0]), int(res[1])
            if not miss_bp > flank_length / 2:
                seq = bamio._decode_seq(sq, l_seq) or "*"
                tail = seq[q0:]
                want = end - start - miss_bp
                if len(tail) > want:
                    out.append([tail[:want], miss_bp, qname])
        return out

    def _fasta(self, ref: str) -> FaiFasta:
        fa = self._fa.get(ref)
        if fa is None:
            with self._open_lock:
                fa = self._fa.get(ref)
                if fa is None:
                    fa = self._fa[ref] = FaiFasta(ref)
        return fa

    def faidx_lines(self, ref: str, region: str) -> Iterable[str]:
        return self._fasta(ref).lines(region)

    def fetch_seq(self, ref: str, chrom: str, start: int, end: int) -> str:
        return self._fasta(ref).fetch(chrom, start, end)


class MemorySamtools:
    # cli.score_jobs may score several chunks at once, a thread each: the native chop writes to arrays of the call's own
    # (not with VAPOR_MEMORY_CHOP=records: that path's CIGAR walk answers into one module-level array, see cli._chunk_threads_ok)
    chunk_threads_ok = True

    """Answers faidx/view from a `SynthWorld`; file names are ignored."""

    def __init__(self, world) -> None:
        self.world = world

    @staticmethod
    def _region(region: str):
        chrom, _, span = region.rpartition(":")
        if not chrom:
            return region, None, None
        a, _, b = span.partition("-")
        return chrom, int(a), int(b)

    def faidx_lines(self, ref: str, region: str) -> Iterable[str]:
        chrom, a, b = self._region(region)
        if a is None:
            seq = self.world.contigs.get(chrom, "")
        else:
            seq = self.world.fetch(chrom, a, b) if chrom in self.world.contigs else ""
        out = [">" + region]
        out.extend(seq[i:i + 60] for i in range(0, len(seq), 60))
        return out

    def view_lines(self, bam: str, region: str) -> Iterable[str]:
        chrom, a, b = self._region(region)
        return [r.line() for r in self.world.overlapping(chrom, a, b)]

    def records(self, bam: str, chrom: str, start: int, end: int):
        return [(r.qname, r.pos, r.cigar, r.seq) for r in self.world.overlapping(chrom, int(start), int(end))]

    def fetch_seq(self, ref: str, chrom: str, start: int, end: int) -> str:
        return self.world.fetch(chrom, start, end) if chrom in self.world.contigs else ""

    # chop_pacbio_read_by_pos in one native call per region: the contig's records as small arrays (positions, spans, read
    # lengths, pointers to the CIGAR texts - made once per record list; nothing is parsed ahead: the walk reads a CIGAR only as
    # far as the window start), the region rule / CIGAR walk / keep rules in the library's host helper (vapor_chop_records),
    # only the kept reads sliced here.  The per-record path (records + cigar2alignstart_by_pos per read) stays the statement
    # it is tested against; VAPOR_MEMORY_CHOP=records selects it.
    def _arrays(self, chrom: str):
        cache = self.__dict__.setdefault("_chop_cache", {})
        recs = self.world.reads.get(chrom, ())
        key = id(recs)                       # (contigs that share one record list - tiled worlds - share its arrays)
        keep_it = getattr(self.world, "cache_ok", True)      # (a world that makes its record lists on demand: nothing is kept)
        got = cache.get(key) if keep_it else None
        if got is None or got[0] is not recs or got[4] != len(recs):
            import ctypes
            import numpy as np
            # (a str's own UTF-8 buffer, NUL-terminated and alive as long as the str: no copy of a 10 kb CIGAR per read)
            from .engine import _utf8
            size = ctypes.c_ssize_t()
            cig = [r.cigar for r in recs]
            ptrs = (ctypes.c_void_p * max(len(recs), 1))(*[_utf8(c, ctypes.byref(size)) for c in cig])
            arrs = (np.array([r.pos for r in recs], dtype=np.int64), np.array([r.ref_span for r in recs], dtype=np.int64),
                    np.array([len(r.seq) for r in recs], dtype=np.int64))
            got = (recs, arrs, (arrs[0].ctypes.data, arrs[1].ctypes.data, ctypes.addressof(ptrs), arrs[2].ctypes.data), (cig, ptrs), len(recs))
            if len(cache) > 200000:
                cache.clear()
            if keep_it:
                cache[key] = got
        return got

    def chop(self, bam: str, chrom: str, start: int, end: int, flank_length):
        if _memory_chop_by_records():
            return _chop_records(self.records(bam, chrom, start, end), start, end, flank_length)
        recs, arrs, ptr, _keep, _n = self._arrays(chrom)
        if not recs:
            return []
        fn = self.__dict__.get("_chop_fn")
        if fn is None:
            from . import _lib
            fn = self.__dict__["_chop_fn"] = _lib.load_holding_gil().vapor_chop_records      # (a ~10 us call: see there)
        import numpy as np
        # (the answers go to arrays of the call's own: chunks of a run are scored on two threads, and tiled worlds share a
        # record list between contigs)
        qm_a, keep_a = np.empty(2 * len(recs), dtype=np.int64), np.empty(len(recs), dtype=np.uint8)
        start, end = int(start), int(end)
        if fn(len(recs), ptr[0], ptr[1], ptr[2], ptr[3], start, end, int(flank_length), qm_a.ctypes.data, keep_a.ctypes.data) != 0:
            raise IndexError("string index out of range")      # what '' [1] raises in SF:331
        kept = keep_a.nonzero()[0]
        if not len(kept):
            return []
        qm = qm_a.tolist()
        out = []
        for t in kept.tolist():
            q0, miss = qm[2 * t], qm[2 * t + 1]
            r = recs[t]
            out.append([r.seq[q0:q0 + (end - start - miss)] if q0 >= 0 else r.seq[q0:][:end - start - miss], miss, r.qname])
        return out

    def chop_many(self, bam: str, chroms, starts, ends, flanks, max_keep: int = 20):
        """chop_pacbio_read_by_pos (SF:339-354) + minimize_pacbio_read_list (SF:1091-1102) for many regions in ONE native call
        (vapor_chop_records_many): per region its kept reads as numbers, not as lists of strings -
        (kept_first [n + 1], addr, q0, miss, status [n], keepalive): read t of region g (kept_first[g] <= t < kept_first[g + 1])
        is the `end - start - miss[t]` bytes at address addr[t] + q0[t] (inside the record's own sequence string, which
        `keepalive` holds); status[g] != 0: the region needs the per-record route (a record without CIGAR, a record whose
        sequence is not ASCII text)."""
        import ctypes
        import numpy as np
        from . import _lib
        from .engine import _ASCII_OFF
        n = len(chroms)
        # per contig, once: (records, pos*, span*, cigar**, seq_len*, addresses of the records' sequences or None)
        keep_it = getattr(self.world, "cache_ok", True)
        per = self.__dict__.setdefault("_many_cache", {}) if keep_it else {}

        def entry(c):
            recs, arrs, p, keep, _cnt = self._arrays(c)
            ok = 0 < _ASCII_OFF < 256 and all(type(r.seq) is str and r.seq.isascii() for r in recs)
            sa_c = (np.fromiter(map(id, (r.seq for r in recs)), dtype=np.uint64, count=len(recs)) + np.uint64(_ASCII_OFF)) if ok else None
            # (the arrays behind the pointers travel with the entry: they live as long as it does)
            e = per[c] = (len(recs), p[0], p[1], p[2], p[3], sa_c, recs, arrs, keep)
            return e
        ent = [per.get(c) or entry(c) for c in chroms]
        if keep_it:
            for g, e in enumerate(ent):                        # (a contig whose record list was replaced since)
                if e[6] is not self.world.reads.get(chroms[g], ()) or e[0] != len(e[6]):
                    ent[g] = entry(chroms[g])
        n_rec = np.fromiter((e[0] for e in ent), dtype=np.int32, count=n)
        ptr = np.asarray([(e[1], e[2], e[3], e[4]) for e in ent], dtype=np.uint64).reshape(n, 4).T.copy()
        sa_ptr = np.fromiter((e[5].ctypes.data if e[5] is not None else 0 for e in ent), dtype=np.uint64, count=n)
        bad = np.fromiter((e[5] is None and e[0] > 0 for e in ent), dtype=bool, count=n)
        cap = max_keep * max(n, 1)
        kept_first = np.zeros(n + 1, dtype=np.int32)
        rec_idx = np.zeros(cap, dtype=np.int32)
        q0 = np.zeros(cap, dtype=np.int64)
        miss = np.zeros(cap, dtype=np.int64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        addr = np.zeros(cap, dtype=np.uint64)
        st = np.ascontiguousarray(starts, dtype=np.int64)
        en = np.ascontiguousarray(ends, dtype=np.int64)
        fl = np.ascontiguousarray(flanks, dtype=np.int64)
        rc = _lib.load().vapor_chop_records_many(n, n_rec.ctypes.data, ptr[0].ctypes.data, ptr[1].ctypes.data, ptr[2].ctypes.data,
                                                 ptr[3].ctypes.data, st.ctypes.data, en.ctypes.data, fl.ctypes.data, max_keep,
                                                 kept_first.ctypes.data, rec_idx.ctypes.data, q0.ctypes.data, miss.ctypes.data,
                                                 status.ctypes.data, sa_ptr.ctypes.data, addr.ctypes.data)
        if rc != 0:
            raise RuntimeError(_lib.load().vapor_bam_last_error().decode())
        tot = int(kept_first[n])
        addr = addr[:tot]
        status[:n][bad] = -1
        return kept_first, addr, q0[:tot], miss[:tot], status[:n], ent      # (the entries hold the records the addresses point into)

    def isfile(self, path: str) -> bool:
        return True

    def fai_lines(self, ref: str) -> Iterable[str]:
        rows = self.world.fai_rows() if hasattr(self.world, "fai_rows") else [(k, len(v)) for k, v in self.world.contigs.items()]
        return ["%s\t%d\t0\t60\t61" % (k, n) for k, n in rows]


def _env_is(name: bytes, value: bytes) -> bool:
    # (os.environ is a mapping with an encode per lookup: 5 us a call, and these are asked once per locus)
    e = os.environ
    v = e._data.get(name) if hasattr(e, "_data") else e.get(name.decode())
    return v in (value, value.decode())


def _memory_chop_by_records() -> bool:
    return _env_is(b"VAPOR_MEMORY_CHOP", b"records")


def set_backend(b) -> None:
    global _backend
    _backend = b


def get_backend():
    global _backend
    if _backend is None:
        # In-process readers (.fai, BGZF/BAM + .bai) unless samtools is asked for: no process per locus, which is the
        # wall-clock floor once scoring runs on the GPU.  They raise on a missing .bai/.fai instead of returning nothing.
        want = os.environ.get("VAPOR_BAM_BACKEND", "inprocess")
        _backend = SamtoolsHybrid() if want == "samtools" else InProcessBam()
        import sys
        print("vapor_amd.seqio: %s backend for BAM regions (VAPOR_BAM_BACKEND=%s)"
              % ("samtools" if want == "samtools" else "in-process BGZF/BAI", want), file=sys.stderr)
    return _backend


# ---------------------------------------------------------------------------
# reference-named helpers
# ---------------------------------------------------------------------------
_COMP = {i: None for i in range(256)}
_COMP.update({ord(a): b for a, b in zip("ATGCNatgcn", "TACGNtacgn")})


def complementary(seq: str) -> str:
    """SF:471-478 - complements ATGCN/atgcn and silently drops everything else."""
    return seq.translate(_COMP)


def reverse(seq: str) -> str:
    return seq[::-1]


def ref_seq_readin(ref, chrom, start, end, reverse_flag="FALSE") -> str:
    """SF:1203-1217: `samtools faidx ref chrom:start-end`, header dropped, the first
    whitespace-separated token of every following line joined, stopping at a blank line."""
    be = get_backend()
    if hasattr(be, "fetch_seq"):
        seq = be.fetch_seq(ref, chrom, int(start), int(end))     # the same bases without the 60-column text in between
    else:
        lines = iter(be.faidx_lines(ref, "%s:%d-%d" % (chrom, int(start), int(end))))
        next(lines, None)
        parts: List[str] = []
        for ln in lines:
            tok = ln.strip().split()
            if not tok:
                break
            parts.append(tok[0])
        seq = "".join(parts)
    if reverse_flag == "FALSE":
        return seq
    return reverse(complementary(seq))


_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=X])")


def _cigar2alignstart_py(cigar: str, align_start: int, start: int, end: int):
    """SF:309-337 in Python (the statement the native helper is tested against)."""
    q = 0
    r = align_start
    last = None
    for m in _CIGAR_RE.finditer(cigar):
        n = int(m.group(1))
        op = m.group(2)
        if op == "S" or op == "I":
            q += n
        elif op == "M" or op == "=":
            q += n
            r += n
        elif op == "D":
            r += n
        last = op
        if r > start - 1:
            break
    if last is None:
        raise IndexError("string index out of range")  # what '' [1] raises in SF:331
    over = int(r) - start
    if last in ("M", "="):
        return [q - over, 0]
    return [q, over]


_cigar_out = None
_cigar_ptr = None
_cigar_fn = None


def cigar2alignstart_by_pos(cigar: str, align_start: int, start: int, end: int):
    """SF:309-337: walk the CIGAR until the reference cursor passes `start-1`; returns
    [offset into the read, miss_bp].  Only S/M/=/I advance the read and M/=/D the
    reference (N, H, P and X advance nothing, as in the reference).  Long-read CIGARs hold thousands of
    operations, so the walk is the library's host helper `vapor_cigar2alignstart` (a dozen times faster than the
    interpreter loop); `_cigar2alignstart_py` is the same in Python."""
    global _cigar_out, _cigar_ptr, _cigar_fn
    if _cigar_out is None:
        import ctypes
        import numpy as np
        from . import _lib
        _cigar_out = np.zeros(2, dtype=np.int64)
        _cigar_ptr = _cigar_out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))     # made once: a ctypes view per call costs more than the walk
        _cigar_fn = _lib.load().vapor_cigar2alignstart
    rc = _cigar_fn(cigar.encode("ascii", "replace"), int(align_start), int(start), _cigar_ptr)
    if rc != 0:
        raise IndexError("string index out of range")  # what '' [1] raises in SF:331
    return [int(_cigar_out[0]), int(_cigar_out[1])]


def chop_pacbio_read_by_pos(bam_in_new, chrom, start, end, flank_length):
    """SF:339-354."""
    out = []
    be = get_backend()
    if hasattr(be, "chop"):
        return be.chop(bam_in_new, chrom, start, end, flank_length)
    if hasattr(be, "records"):
        recs = be.records(bam_in_new, chrom, start, end)
    else:
        recs = []
        for line in be.view_lines(bam_in_new, "%s:%d-%d" % (chrom, start, end)):
            f = line.strip().split()
            if not f or f[0] == "@":
                continue
            recs.append((f[0], f[3], f[5], f[9]))
    return _chop_records(recs, start, end, flank_length)


def _chop_records(recs, start, end, flank_length):
    """The body of chop_pacbio_read_by_pos (SF:345-353) over (qname, pos, cigar, seq) records."""
    out = []
    for qname, pos, cigar, seq in recs:
        if int(pos) < start + 1:
            q0, miss_bp = cigar2alignstart_by_pos(cigar, int(pos), start, end)
            if not miss_bp > flank_length / 2:
                tail = seq[q0:]
                want = end - start - miss_bp
                if len(tail) > want:
                    out.append([tail[:want], miss_bp, qname])
    return out


def minimize_pacbio_read_list(x, ideal_list_length=20):
    """SF:1091-1102: keep at most 20 reads, smallest miss_bp first, input order inside
    one miss_bp value."""
    if len(x) <= ideal_list_length:
        return x
    by_miss = {}
    for rec in x:
        by_miss.setdefault(rec[1], []).append(rec)
    out = []
    for k in sorted(by_miss):
        if len(out) < ideal_list_length:
            out += by_miss[k]
    return out[:ideal_list_length]


def bam_in_decide(bam_in, bps):
    """SF:69-89: a file, or a per-chromosome pattern with XXX or * in the basename."""
    be = get_backend()
    if be.isfile(bam_in):
        return [bam_in]
    d = "/".join(bam_in.split("/")[:-1]) + "/"
    base = bam_in.split("/")[-1]
    if "XXX" in base:
        keys = base.split("XXX")
    elif "*" in base:
        keys = base.split("*")
    else:
        print("Error: invalid name for pacbio files !")
        raise NameError("bam_in_keys")  # the reference dies on the unbound name (SF:82)
    ext = bam_in.split(".")[-1]
    return [d + k for k in os.listdir(d)
            if k.split(".")[-1] == ext and all(y in k for y in keys)]


def simple_del_chop_pacbio_read_simple_short(bam_in, sv_info, flank_length):
    """SF:1378-1390: reads around the left breakpoint only."""
    bams = bam_in_decide(bam_in, sv_info)
    if bams == "":
        return [[], [], []]
    x = []
    for b in bams:
        x += chop_pacbio_read_by_pos(b, sv_info[0], int(sv_info[1]) - flank_length,
                                     int(sv_info[1]) + flank_length, flank_length)
    return minimize_pacbio_read_list(x)


def simple_chop_pacbio_read_simple_short(bam_in, sv_info, flank_length):
    """SF:1392-1401: reads spanning first to last breakpoint."""
    bams = bam_in_decide(bam_in, sv_info)
    if bams == "":
        return [[], [], []]
    x = []
    for b in bams:
        x += chop_pacbio_read_by_pos(b, sv_info[0], int(sv_info[1]) - flank_length,
                                     int(sv_info[-1]) + flank_length, flank_length)
    return minimize_pacbio_read_list(x)


class _Chromos(list):
    """The contig names as the list the reference builds, with `in` answered from a set (the drivers of the unclassified
    structures test every breakpoint token against it, SF:1490-1555: a scan of thousands of names each)."""

    def __init__(self, names):
        super().__init__(names)
        self._names = frozenset(names)

    def __contains__(self, x):
        try:
            return x in self._names
        except TypeError:                     # (an unhashable token: the list's own comparison)
            return list.__contains__(self, x)


_chromos_cache: dict = {}


def chromos_readin(ref) -> List[str]:
    """SF:356-363: contig names from the .fai.  The reference reads the file again for every unclassified record; the names
    are kept here per backend and index file as long as the file's size and modification time (or, for an in-memory world,
    its contig table) stay what they were - 2.5 ms a call on an index of 5 000 contigs otherwise."""
    be = get_backend()
    world = getattr(be, "world", None)
    if world is not None:
        stamp = (id(world.contigs), len(world.contigs))
    else:
        try:
            st = os.stat(str(ref) + ".fai")
            stamp = (st.st_mtime_ns, st.st_size)
        except OSError:
            stamp = None
    key = (id(be), ref)
    got = _chromos_cache.get(key)
    if got is not None and stamp is not None and got[0] == stamp:
        return got[1]
    out = []
    for ln in be.fai_lines(ref):
        f = ln.strip().split()
        if f:
            out.append(f[0])
    out = _Chromos(out)
    if stamp is not None:
        if len(_chromos_cache) > 64:
            _chromos_cache.clear()
        _chromos_cache[key] = (stamp, out, be)          # (the backend kept alive: its id() is part of the key)
    return out


def flank_length_calculate(bps) -> int:
    """SF:794-802: min(500, last - first breakpoint)."""
    span = int(bps[-1]) - int(bps[1])
    return span if span < 500 else 500
