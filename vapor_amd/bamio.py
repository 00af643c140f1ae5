"""In-process BAM access for `samtools view bam chrom:start-end`.

The reference starts one samtools process per locus and BAM (SF:340); once scoring runs on the GPU
that popen is the wall-clock floor.  This module reads BGZF/BAM and the .bai index directly and returns
the four fields the reference uses from every SAM line (QNAME, POS, CIGAR, SEQ; SF:342-352), as records
(`fetch_records`) or as text lines in SAM column order (`fetch_lines`).

This is the default BAM backend (`seqio.get_backend`; VAPOR_BAM_BACKEND=samtools selects the samtools binary
instead).  It follows the SAM/BAM specification (v1 BAM incl. the CG:B,I long-CIGAR convention, BAI bins and
linear index).  Checks (tests/test_bamio.py): its BGZF blocks against Python's gzip module, reg2bin / reg2bins
against a brute-force walk of the bin hierarchy, records against a second, independent BAM encoder written from
the specification in the test, and against this module's own writer.  No third-party BAM file exists in this
environment (the reference bundles none), so it has not been run on one.
"""
from __future__ import annotations

import struct
import threading
import zlib
from typing import Dict, Iterator, List, Tuple

import numpy as np

_SEQ = "=ACMGRSVTWYHKDBN"
_CIG = "MIDNSHP=X"
_REF_OP = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint32)
_SEQ_LUT16 = None       # byte -> its two bases as one little-endian uint16 (numpy, built on first use)


def _decode_seq(sq: bytes, l_seq: int) -> str:
    """4-bit packed bases -> text, a table lookup per byte done by numpy (reads are tens of kilobases)."""
    global _SEQ_LUT16
    if _SEQ_LUT16 is None:
        _SEQ_LUT16 = np.array([ord(a) | (ord(b) << 8) for a in _SEQ for b in _SEQ], dtype="<u2")
    return _SEQ_LUT16[np.frombuffer(sq, dtype=np.uint8)].tobytes()[:l_seq].decode("ascii")


# ---------------------------------------------------------------------------------------------
# BGZF
# ---------------------------------------------------------------------------------------------
class BgzfReader:
    """Random access by virtual offset (compressed block offset << 16 | offset inside the block)."""

    def __init__(self, path: str):
        self.f = open(path, "rb")
        self._cache: Dict[int, Tuple[bytes, int]] = {}

    def _block(self, coff: int) -> Tuple[bytes, int]:
        got = self._cache.get(coff)
        if got is not None:
            return got
        self.f.seek(coff)
        hdr = self.f.read(18)
        if len(hdr) < 18:
            return b"", 0
        if hdr[:4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at offset %d" % coff)
        xlen = struct.unpack("<H", hdr[10:12])[0]
        extra = hdr[12:] + self.f.read(xlen - 6)
        bsize = None
        p = 0
        while p + 4 <= len(extra):
            si1, si2, slen = extra[p], extra[p + 1], struct.unpack("<H", extra[p + 2:p + 4])[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack("<H", extra[p + 4:p + 6])[0] + 1
            p += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        if bsize < xlen + 20:
            raise ValueError("BGZF block at offset %d is shorter than its own header and trailer" % coff)
        cdata = self.f.read(bsize - 12 - xlen - 8)
        trailer = self.f.read(8)
        if len(trailer) < 8:
            raise ValueError("truncated BGZF block at offset %d" % coff)
        crc, isize = struct.unpack("<II", trailer)
        if isize > 65536:
            raise ValueError("BGZF block at offset %d claims %d bytes (at most 65536)" % (coff, isize))
        try:
            data = zlib.decompress(cdata, -15)
        except zlib.error as e:
            raise ValueError("BGZF block at offset %d does not inflate: %s" % (coff, e)) from None
        # the block's own CRC32 and size, as htslib checks them: a damaged block must not become reads
        if len(data) != isize or (zlib.crc32(data) & 0xFFFFFFFF) != crc:
            raise ValueError("BGZF block at offset %d fails its CRC32 / size check" % coff)
        if len(self._cache) > 64:
            self._cache.clear()
        self._cache[coff] = (data, bsize)
        return data, bsize

    def read_from(self, voff: int):
        """Generator of (bytes, virtual offset after them) is awkward for records; use Cursor."""
        return BgzfCursor(self, voff)


class BgzfCursor:
    def __init__(self, rd: BgzfReader, voff: int):
        self.rd = rd
        self.coff, self.uoff = voff >> 16, voff & 0xFFFF
        self.data, self.bsize = rd._block(self.coff)

    def tell(self) -> int:
        return (self.coff << 16) | self.uoff

    def read(self, n: int) -> bytes:
        out = []
        while n > 0:
            if self.uoff >= len(self.data):
                if self.bsize == 0:
                    break
                self.coff += self.bsize
                self.uoff = 0
                self.data, self.bsize = self.rd._block(self.coff)
                if not self.data and self.bsize == 0:
                    break
                continue
            take = self.data[self.uoff:self.uoff + n]
            out.append(take)
            self.uoff += len(take)
            n -= len(take)
        return b"".join(out)


# ---------------------------------------------------------------------------------------------
# BAI
# ---------------------------------------------------------------------------------------------
def reg2bins(beg: int, end: int) -> List[int]:
    """Bins overlapping the 0-based half-open interval [beg, end) (SAM spec, section 5.3)."""
    end -= 1
    out = [0]
    for shift, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out.extend(range(off + (beg >> shift), off + (end >> shift) + 1))
    return out


def reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


class BaiIndex:
    def __init__(self, path: str):
        raw = open(path, "rb").read()
        if raw[:4] != b"BAI\x01":
            raise ValueError("not a BAI index: " + path)
        p = 4
        n_ref = struct.unpack_from("<i", raw, p)[0]
        p += 4
        self.bins: List[Dict[int, List[Tuple[int, int]]]] = []
        self.linear: List[List[int]] = []
        for _ in range(n_ref):
            n_bin = struct.unpack_from("<i", raw, p)[0]
            p += 4
            d = {}
            for _b in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", raw, p)
                p += 8
                ch = list(struct.iter_unpack("<QQ", raw[p:p + 16 * n_chunk]))
                p += 16 * n_chunk
                d[b] = ch
            n_intv = struct.unpack_from("<i", raw, p)[0]
            p += 4
            lin = list(struct.unpack_from("<%dQ" % n_intv, raw, p)) if n_intv else []
            p += 8 * n_intv
            self.bins.append(d)
            self.linear.append(lin)

    def chunks(self, tid: int, beg: int, end: int) -> List[Tuple[int, int]]:
        d = self.bins[tid]
        lin = self.linear[tid]
        min_off = lin[min(beg >> 14, len(lin) - 1)] if lin else 0
        out = []
        for b in reg2bins(beg, end):
            for s, e in d.get(b, ()):
                if e > min_off:
                    out.append((max(s, min_off), e))
        out.sort()
        merged: List[Tuple[int, int]] = []
        for s, e in out:
            if merged and s <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], e))
            else:
                merged.append((s, e))
        return merged


# ---------------------------------------------------------------------------------------------
# BAM
# ---------------------------------------------------------------------------------------------
class BamFile:
    def __init__(self, path: str):
        self.path = path
        self.bgzf = BgzfReader(path)
        c = self.bgzf.read_from(0)
        if c.read(4) != b"BAM\x01":
            raise ValueError("not a BAM file: " + path)
        l_text = struct.unpack("<i", c.read(4))[0]
        c.read(l_text)
        n_ref = struct.unpack("<i", c.read(4))[0]
        self.refs: List[Tuple[str, int]] = []
        self.tid: Dict[str, int] = {}
        for t in range(n_ref):
            l_name = struct.unpack("<i", c.read(4))[0]
            name = c.read(l_name)[:-1].decode()
            l_ref = struct.unpack("<i", c.read(4))[0]
            self.refs.append((name, l_ref))
            self.tid[name] = t
        self.first_record = c.tell()
        # vapor_bam handles of the HIP library's host helper: as many as calls were ever in flight at once (a handle owns
        # its file descriptor and inflate buffers, the .bai index above is shared; `_free` holds the idle ones):
        # pipeline.run_batch advances the loci of a batch on a few threads
        self._free = []
        self._handles = []
        self._lock = threading.Lock()
        import os
        bai = path + ".bai" if os.path.exists(path + ".bai") else path[:-4] + ".bai"
        self.index = BaiIndex(bai)

    @staticmethod
    def _aux_cigar(rec: bytes, p: int):
        """The real CIGAR of a record that carries more than 65535 operations: the CG:B,I tag (SAM spec 4.2.2)."""
        n = len(rec)
        size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
        while p + 3 <= n:
            tag, typ = rec[p:p + 2], chr(rec[p + 2])
            p += 3
            if typ in size:
                p += size[typ]
            elif typ in "ZH":
                p = rec.index(b"\x00", p) + 1
            elif typ == "B":
                sub, cnt = chr(rec[p]), struct.unpack_from("<i", rec, p + 1)[0]
                p += 5
                if tag == b"CG" and sub == "I":
                    return np.frombuffer(rec, dtype="<u4", count=cnt, offset=p)
                p += cnt * size[sub]
            else:
                break
        return None

    @classmethod
    def _parse(cls, rec: bytes):
        ref_id, pos, l_name, _mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
        p = 32
        name = rec[p:p + l_name - 1].decode()
        p += l_name
        cig = np.frombuffer(rec, dtype="<u4", count=n_cig, offset=p)
        p += 4 * n_cig
        nb = (l_seq + 1) // 2
        sq = rec[p:p + nb]
        if n_cig == 2 and (int(cig[0]) & 15) == 4 and (int(cig[0]) >> 4) == l_seq and (int(cig[1]) & 15) == 3:
            real = cls._aux_cigar(rec, p + nb + l_seq)      # placeholder <l_seq>S<ref len>N: the operations are in CG
            if real is not None:
                cig = real
        return ref_id, pos, name, flag, cig, l_seq, sq

    def chop_native(self, chrom: str, start: int, end: int, flank_length: int):
        """chop_pacbio_read_by_pos (SF:339-354) for one region through the library's native reader (vapor_bam_chop:
        threaded inflate, binary CIGAR walk, only kept bases decoded); the .bai lookup stays here.  Returns the same
        [[read tail, miss_bp, qname], ...] as the Python statement of it (seqio.InProcessBam.chop_python)."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        tid = self.tid.get(chrom)
        if tid is None:
            return []
        ch = self.index.chunks(tid, max(int(start) - 1, 0), int(end))
        if not ch:
            return []
        tl = self._take_handle(lib)
        try:
            return self._chop_with(lib, tl, tid, ch, start, end, flank_length)
        finally:
            with self._lock:
                self._free.append(tl)

    def chop_native_raw(self, chrom: str, start: int, end: int, flank_length: int):
        """chop_native's answer as numbers: (text, offsets, lengths, miss_bp) - read r is text[offsets[r] : offsets[r] + lengths[r]]
        - for callers that hand the reads on by address (vapor_amd.fastpath) instead of making a string per read."""
        from . import _lib
        lib = _lib.load()
        tid = self.tid.get(chrom)
        if tid is None:
            return None
        ch = self.index.chunks(tid, max(int(start) - 1, 0), int(end))
        if not ch:
            return None
        tl = self._take_handle(lib)
        try:
            return self._chop_with(lib, tl, tid, ch, start, end, flank_length, raw=True)
        finally:
            with self._lock:
                self._free.append(tl)

    def _take_handle(self, lib):
        """A native handle with its output buffers, for the duration of one call: from the free list, else a new one.
        (Handles are not tied to threads: a pool of threads that lives for one batch would leave its handles - a file
        descriptor and the inflate buffers each - behind for every batch of a long run.)"""
        import ctypes
        with self._lock:
            if self._free:
                return self._free.pop()
        h = ctypes.c_void_p()
        if lib.vapor_bam_open(self.path.encode(), ctypes.byref(h)) != 0:
            raise OSError(lib.vapor_bam_last_error().decode())
        if threading.current_thread() is not threading.main_thread():
            import os
            lib.vapor_bam_set_threads(h, int(os.environ.get("VAPOR_BAM_INFLATE_THREADS", "2")))   # several readers at once: fewer inflate threads each
        tl = {"native": h, "buf": {"seq": np.empty(1 << 20, dtype=np.uint8), "names": ctypes.create_string_buffer(1 << 16),
                                   "meta": np.empty(4 * 256, dtype=np.int64), "need": np.zeros(3, dtype=np.int64)}}
        with self._lock:
            self._handles.append(h)
        return tl

    def _chop_with(self, lib, tl, tid, ch, start, end, flank_length, raw=False):
        import ctypes
        from . import _lib
        chunks = np.asarray(ch, dtype=np.uint64).reshape(-1)
        n = ctypes.c_int32(0)
        while True:
            bf = tl["buf"]
            rc = lib.vapor_bam_chop(tl["native"], tid, int(start), int(end), int(flank_length), len(ch), chunks.ctypes.data,
                                    bf["seq"].ctypes.data, bf["seq"].size, ctypes.cast(bf["names"], ctypes.c_void_p), len(bf["names"]),
                                    bf["meta"].ctypes.data, bf["meta"].size // 4, ctypes.byref(n), bf["need"].ctypes.data)
            if rc == 0:
                break
            if rc != _lib.E_OVERFLOW:
                msg = lib.vapor_bam_last_error().decode()
                if "IndexError" in msg:
                    raise IndexError("string index out of range")      # what '' [1] raises in SF:331
                raise ValueError(msg)
            need = bf["need"]
            tl["buf"] = {"seq": np.empty(int(need[0]) * 2 + 1024, dtype=np.uint8),
                         "names": ctypes.create_string_buffer(int(need[1]) * 2 + 256),
                         "meta": np.empty(4 * (int(need[2]) * 2 + 16), dtype=np.int64), "need": need}
        if n.value == 0:
            return None if raw else []
        if raw:
            meta = bf["meta"][:4 * n.value].reshape(-1, 4).copy()
            whole = bf["seq"][:int((meta[:, 0] + meta[:, 1]).max())].tobytes().decode("ascii")
            return whole, meta[:, 0], meta[:, 1], meta[:, 2]
        # (one conversion of the numbers, one of the bases: this runs under the interpreter lock on every pool thread)
        m = bf["meta"][:4 * n.value].tolist()
        whole = bf["seq"][:max(m[4 * r] + m[4 * r + 1] for r in range(n.value))].tobytes().decode("ascii")
        names = bf["names"].raw
        out = []
        for r in range(n.value):
            o, ln, miss, no = m[4 * r:4 * r + 4]
            out.append([whole[o:o + ln], miss, names[no:names.index(b"\0", no)].decode()])
        return out

    def close(self) -> None:
        with self._lock:
            hs, self._handles, self._free = self._handles, [], []
        if hs:
            from . import _lib
            for h in hs:
                _lib.load().vapor_bam_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

    def fetch_raw(self, chrom: str, start: int, end: int):
        """(QNAME, 1-based POS, CIGAR operations as a uint32 tuple, packed SEQ bytes, l_seq, FLAG) of the alignments
        that overlap the 1-based inclusive region, in file order; nothing is decoded to text."""
        tid = self.tid.get(chrom)
        if tid is None:
            return []
        beg, stop = max(start - 1, 0), end               # 0-based half-open
        out = []
        for cs, ce in self.index.chunks(tid, beg, stop):
            cur = self.bgzf.read_from(cs)
            while cur.tell() < ce:
                hdr = cur.read(4)
                if len(hdr) < 4:
                    break
                rec = cur.read(struct.unpack("<i", hdr)[0])
                ref_id, pos, name, flag, cig, l_seq, sq = self._parse(rec)
                if ref_id != tid or pos >= stop:
                    if ref_id > tid or (ref_id == tid and pos >= stop):
                        break
                    continue
                rlen = int(((cig >> 4) * _REF_OP[cig & 15]).sum()) if len(cig) else 0     # M, D, N, =, X consume reference
                if pos + max(rlen, 1) <= beg:
                    continue
                out.append((name, pos + 1, cig, sq, l_seq, flag))
        return out

    def fetch_records(self, chrom: str, start: int, end: int) -> List[Tuple[str, int, str, str, int]]:
        """(QNAME, 1-based POS, CIGAR, SEQ, FLAG) as text fields - what `samtools view bam chrom:start-end` lists."""
        return [(name, pos, "".join("%d%s" % (c >> 4, _CIG[c & 15]) for c in cig.tolist()) or "*", _decode_seq(sq, l_seq) or "*", flag)
                for name, pos, cig, sq, l_seq, flag in self.fetch_raw(chrom, start, end)]

    def fetch_lines(self, chrom: str, start: int, end: int) -> List[str]:
        """The same as SAM-ordered text lines (QNAME FLAG RNAME POS MAPQ CIGAR * 0 0 SEQ *)."""
        return ["\t".join([name, str(flag), chrom, str(pos), "0", cigar, "*", "0", "0", seq, "*"])
                for name, pos, cigar, seq, flag in self.fetch_records(chrom, start, end)]


# ---------------------------------------------------------------------------------------------
# writer (tests and synthetic worlds): coordinate-sorted BAM + BAI
# ---------------------------------------------------------------------------------------------
def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path: str, refs: List[Tuple[str, int]], records: List[Tuple[str, int, int, str, str]],
              block_size: int = 16384, qual_seed=None) -> None:
    """records: (qname, tid, pos0, cigar string, seq), will be sorted by (tid, pos).  Writes path and
    path + '.bai'.  Qualities are 0xFF ("absent") - or, with qual_seed, seeded values in runs of a few bases between 2 and 60,
    which is what makes the blocks of a sequencer's file literal-heavy for its DEFLATE decoder."""
    import re
    import numpy as np
    lut = np.full(256, 15, dtype=np.uint8)               # 4-bit codes of "=ACMGRSVTWYHKDBN", anything else N
    for i, c in enumerate(_SEQ):
        lut[ord(c)] = i
        lut[ord(c.lower())] = i
    recs = sorted(records, key=lambda r: (r[1], r[2]))
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    head = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    head += b"".join(struct.pack("<i", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<i", ln) for name, ln in refs)
    blobs = []
    meta = []
    qrng = np.random.default_rng(qual_seed) if qual_seed is not None else None
    for qname, tid, pos, cigar, seq in recs:
        ops = [(int(n), _CIG.index(o)) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", cigar)]
        rlen = sum(n for n, o in ops if o in (0, 2, 3, 7, 8))
        end = pos + max(rlen, 1)
        codes = lut[np.frombuffer(seq.encode("latin-1", "replace"), dtype=np.uint8)]
        if len(codes) & 1:
            codes = np.concatenate((codes, np.zeros(1, dtype=np.uint8)))
        sq = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()
        packed = [(n << 4) | o for n, o in ops]
        aux = b""
        if len(packed) > 65535:
            # SAM spec 4.2.2: the operations go to CG:B,I, the CIGAR field holds <l_seq>S<reference length>N
            aux = b"CGBI" + struct.pack("<i", len(packed)) + struct.pack("<%dI" % len(packed), *packed)
            packed = [(len(seq) << 4) | 4, (rlen << 4) | 3]
        body = struct.pack("<iiBBHHHiiii", tid, pos, len(qname) + 1, 60, reg2bin(pos, end), len(packed), 0, len(seq), -1, -1, 0)
        if qrng is None:
            qual = b"\xff" * len(seq)
        else:
            runs = qrng.integers(1, 6, size=len(seq) // 2 + 1)
            qual = np.repeat(qrng.integers(2, 61, size=len(runs)).astype(np.uint8), runs)[:len(seq)].tobytes()
        body += qname.encode() + b"\x00" + b"".join(struct.pack("<I", c) for c in packed) + bytes(sq) + qual + aux
        blobs.append(struct.pack("<i", len(body)) + body)
        meta.append((tid, pos, end))
    # lay the stream out in BGZF blocks, remembering the virtual offset of every record
    starts = []
    pos_in_stream = len(head)
    for b in blobs:
        starts.append(pos_in_stream)
        pos_in_stream += len(b)
    stream = b"".join([head] + blobs)
    ends = starts[1:] + [len(stream)]
    pieces = []
    block_coff = []
    out_len = 0
    for o in range(0, len(stream), block_size):
        block_coff.append(out_len)
        blk = _bgzf_block(stream[o:o + block_size])
        pieces.append(blk)
        out_len += len(blk)
    pieces.append(_BGZF_EOF)
    out = b"".join(pieces)

    def voff(u: int) -> int:
        b = u // block_size
        if b >= len(block_coff):
            return (len(out) - len(_BGZF_EOF)) << 16
        return (block_coff[b] << 16) | (u % block_size)

    open(path, "wb").write(out)
    bins: List[Dict[int, List[List[int]]]] = [dict() for _ in refs]
    linear: List[List[int]] = [[] for _ in refs]
    for (tid, pos, end), s, e in zip(meta, starts, ends):
        vs, ve = voff(s), voff(e)
        ch = bins[tid].setdefault(reg2bin(pos, end), [])
        if ch and ch[-1][1] == vs:
            ch[-1][1] = ve
        else:
            ch.append([vs, ve])
        lin = linear[tid]
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            while len(lin) <= w:
                lin.append(0)
            if lin[w] == 0:
                lin[w] = vs
        # windows without their own first record inherit the next known offset backwards (spec: the
        # linear index gives the smallest offset of any record overlapping the window)
    bai = b"BAI\x01" + struct.pack("<i", len(refs))
    for t in range(len(refs)):
        lin = linear[t]
        last = 0
        for w in range(len(lin)):
            if lin[w] == 0:
                lin[w] = last
            last = lin[w]
        bai += struct.pack("<i", len(bins[t]))
        for b, ch in sorted(bins[t].items()):
            bai += struct.pack("<Ii", b, len(ch)) + b"".join(struct.pack("<QQ", s, e) for s, e in ch)
        bai += struct.pack("<i", len(lin)) + b"".join(struct.pack("<Q", v) for v in lin)
    bai += struct.pack("<Q", 0)
    open(path + ".bai", "wb").write(bai)
