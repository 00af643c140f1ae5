"""Single-node workflow layer.

The reference scales out with a WDL workflow: split the BED per contig, run `vapor bed` once per contig,
then concatenate the per-contig tables, drop the header, `sort -Vk1,1 -k2,2n -k3,3n`, `bgzip` and `tabix -p bed`
(wdl/VaPoRBedPerContig.wdl:45-85, wdl/TasksBenchmark.wdl:249-313).  On one node with several GPUs none of
the scatter is needed - `vapor_amd.cli` shards the loci over the ranks of a torchrun launch - so this module
is the gather side only: the same sorted, block-gzipped table plus its tabix index, written in-process
(no bgzip / tabix binaries), and a launcher that runs the CLI on N GPUs and then produces them.

    python -m vapor_amd.workflow --gpus 8 [--ranks-per-gpu R|auto] --prefix sample1 bed --sv-input x.bed --reference ref.fa \\
           --pacbio-input reads.bam --output-path figs/ --output-file sample1.vapor

writes sample1.vapor (the CLI's table, unchanged), sample1.bed.gz and sample1.bed.gz.tbi.

Deliberate difference: the WDL's `tail -n+2` removes only the first shard's header line, so its merged table
keeps one stray header row per further contig; here the header is dropped once and for all.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import zlib
from typing import Dict, Iterable, List, Sequence, Tuple

from .bamio import _BGZF_EOF, BgzfReader, reg2bin, reg2bins

_BLOCK = 0xFF00            # uncompressed bytes per BGZF block, as bgzip writes them
_NUM = re.compile(r"(\d+)")


# ---------------------------------------------------------------------------------------------
# sort -Vk1,1 -k2,2n -k3,3n
# ---------------------------------------------------------------------------------------------
def _order(c: str) -> int:
    """filevercmp's order(): digits 0, letters their code, '~' before everything, the rest after the letters."""
    if c.isdigit():
        return 0
    if c.isalpha():
        return ord(c)
    return -1 if c == "~" else ord(c) + 256


def _match_suffix(s: str) -> int:
    """filevercmp's match_suffix(): where a trailing run of `.[A-Za-z~][A-Za-z0-9~]*` groups starts, -1 when there is none."""
    match, read_alpha = -1, False
    for t, c in enumerate(s):
        if read_alpha:
            read_alpha = False
            if not (c.isalpha() or c == "~"):
                match = -1
        elif c == ".":
            read_alpha = True
            if match < 0:
                match = t
        elif not (c.isalnum() or c == "~"):
            match = -1
    return match


def _verrevcmp(a: str, la: int, b: str, lb: int) -> int:
    """filevercmp's verrevcmp() on a[:la], b[:lb]; like the C, the digit loops look at the characters behind those lengths
    (the strings end at their NUL there: an index past the end reads as a non-digit here)."""
    def dig(s, t):
        return t < len(s) and s[t].isdigit()
    pa = pb = 0
    while pa < la or pb < lb:
        first = 0
        while (pa < la and not a[pa].isdigit()) or (pb < lb and not b[pb].isdigit()):
            ca = 0 if pa == la else _order(a[pa])
            cb = 0 if pb == lb else _order(b[pb])
            if ca != cb:
                return ca - cb
            pa += 1
            pb += 1
        while pa < len(a) and a[pa] == "0":
            pa += 1
        while pb < len(b) and b[pb] == "0":
            pb += 1
        while dig(a, pa) and dig(b, pb):
            if not first:
                first = ord(a[pa]) - ord(b[pb])
            pa += 1
            pb += 1
        if dig(a, pa):
            return 1
        if dig(b, pb):
            return -1
        if first:
            return first
    return 0


def filevercmp(a: str, b: str) -> int:
    """GNU `sort -V`'s comparison (gnulib filevercmp.c as of coreutils 8.32, the version in the reference's images), restated:
    digit runs compare as numbers without their leading zeros, letters sort before other characters, '~' before everything,
    a trailing `.suffix` run is set aside unless the names are equal without it, names that compare equal fall back to
    strcmp().  ASCII names (LC_ALL=C), which contig names are."""
    simple = (a > b) - (a < b)
    if simple == 0:
        return 0
    if not a:
        return -1
    if not b:
        return 1
    for dots in (".", ".."):
        if a == dots:
            return -1
        if b == dots:
            return 1
    if a[0] == "." and b[0] != ".":
        return -1
    if a[0] != "." and b[0] == ".":
        return 1
    if a[0] == "." and b[0] == ".":
        a, b = a[1:], b[1:]
    sa, sb = _match_suffix(a), _match_suffix(b)
    la = sa if sa >= 0 else len(a)
    lb = sb if sb >= 0 else len(b)
    if (sa >= 0 or sb >= 0) and la == lb and a[:la] == b[:lb]:
        la, lb = len(a), len(b)
    r = _verrevcmp(a, la, b, lb)
    return r if r else simple


def version_key(s: str):
    """A sort key with filevercmp's order (for callers that sort names by themselves)."""
    import functools
    return functools.cmp_to_key(filevercmp)(s)


def _num(field: str) -> float:
    """`sort -n`: leading blanks, an optional '-', digits with an optional fraction; 0 when there is no number."""
    m = re.match(r"[ \t]*(-?\d*\.?\d*)", field)
    txt = m.group(1) if m else ""
    try:
        return float(txt) if any(ch.isdigit() for ch in txt) else 0.0
    except ValueError:
        return 0.0


def sort_rows(lines: Iterable[str]) -> List[str]:
    """`sort -Vk1,1 -k2,2n -k3,3n` (wdl/TasksBenchmark.wdl:286-301, LC_ALL=C) of tab-separated rows without blanks inside
    their first three fields: version order on the contig name, numeric on start and end, the whole line byte-wise as the
    last resort (sort's default when -s is not given).  The distinct contig names are ranked once with filevercmp (a
    comparison function is slow in Python; a table has few names and many rows), the rows sort on plain tuples."""
    import functools
    rows = [ln for ln in lines if ln]
    names = sorted({ln.split("\t", 1)[0] for ln in rows}, key=functools.cmp_to_key(filevercmp))
    rank = {n: t for t, n in enumerate(names)}

    def key(ln: str):
        f = ln.split("\t")
        return (rank[f[0]], _num(f[1]) if len(f) > 1 else 0.0, _num(f[2]) if len(f) > 2 else 0.0, ln)
    return sorted(rows, key=key)


# ---------------------------------------------------------------------------------------------
# BGZF + tabix
# ---------------------------------------------------------------------------------------------
def _block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_bgzf(path: str, payload: bytes) -> List[int]:
    """Writes `payload` as BGZF; returns the compressed offset of every block (for virtual offsets)."""
    offs, out = [], bytearray()
    for o in range(0, len(payload), _BLOCK):
        offs.append(len(out))
        out += _block(payload[o:o + _BLOCK])
    offs.append(len(out))          # where the next block would start (the EOF block)
    out += _BGZF_EOF
    with open(path, "wb") as f:
        f.write(bytes(out))
    return offs


def _voff(block_offs: Sequence[int], u: int) -> int:
    b = u // _BLOCK
    if u % _BLOCK == 0 and b >= len(block_offs) - 1:
        return block_offs[-1] << 16
    return (block_offs[b] << 16) | (u % _BLOCK)


def write_bed_gz_with_index(path: str, lines: Sequence[str], index: bool = True) -> None:
    """`bgzip -c > path` and `tabix -f -p bed path` for already sorted BED-like rows (0-based start in column 2,
    end in column 3, as tabix's bed preset reads them)."""
    text = "".join(ln + "\n" for ln in lines).encode()
    offs = write_bgzf(path, text)
    if not index:
        open(path + ".tbi", "wb").close()            # the WDL touches an empty index in that case
        return
    names: List[str] = []
    seen = set()
    bins: List[Dict[int, List[List[int]]]] = []
    linear: List[List[int]] = []
    pos = 0
    for ln in lines:
        n = len(ln.encode()) + 1
        f = ln.split("\t")
        if ln.startswith("#") or len(f) < 3:
            pos += n
            continue
        if not names or names[-1] != f[0]:
            if f[0] in seen:
                raise ValueError("rows are not grouped by contig: " + f[0])
            seen.add(f[0])
            names.append(f[0]); bins.append({}); linear.append([])
        beg, end = int(_num(f[1])), int(_num(f[2]))
        if end <= beg:
            end = beg + 1
        vs, ve = _voff(offs, pos), _voff(offs, pos + n)
        ch = bins[-1].setdefault(reg2bin(beg, end), [])
        if ch and ch[-1][1] == vs:
            ch[-1][1] = ve
        else:
            ch.append([vs, ve])
        lin = linear[-1]
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            while len(lin) <= w:
                lin.append(0)
            if lin[w] == 0:
                lin[w] = vs
        pos += n
    nm = b"".join(x.encode() + b"\0" for x in names)
    # format 0x10000 = UCSC/BED coordinates (0-based, half-open); columns 1, 2, 3; comment char '#'; skip 0
    raw = bytearray(b"TBI\x01" + struct.pack("<iiiiiiii", len(names), 0x10000, 1, 2, 3, ord("#"), 0, len(nm)) + nm)
    for t in range(len(names)):
        lin, last = linear[t], 0
        for w in range(len(lin)):
            if lin[w] == 0:
                lin[w] = last
            last = lin[w]
        raw += struct.pack("<i", len(bins[t]))
        for b, ch in sorted(bins[t].items()):
            raw += struct.pack("<Ii", b, len(ch)) + b"".join(struct.pack("<QQ", s, e) for s, e in ch)
        raw += struct.pack("<i", len(lin)) + struct.pack("<%dQ" % len(lin), *lin)
    write_bgzf(path + ".tbi", bytes(raw))


def read_bgzf(path: str) -> bytes:
    """Whole content of a BGZF file (any gzip reader can do this; used by the tests and tabix_query)."""
    import gzip
    with gzip.open(path, "rb") as f:
        return f.read()


def tabix_query(path: str, chrom: str, start: int, end: int) -> List[str]:
    """Rows of `path` overlapping chrom:start-end (1-based inclusive, like `tabix path chrom:start-end`), found
    through the .tbi index."""
    raw = read_bgzf(path + ".tbi")
    if raw[:4] != b"TBI\x01":
        raise ValueError("not a tabix index")
    n_ref, _fmt, c_seq, c_beg, c_end, _meta, _skip, l_nm = struct.unpack_from("<iiiiiiii", raw, 4)
    p = 36
    names = raw[p:p + l_nm].split(b"\0")[:-1]
    p += l_nm
    want = names.index(chrom.encode()) if chrom.encode() in names else -1
    chunks: List[Tuple[int, int]] = []
    min_off = 0
    beg, stop = max(start - 1, 0), end
    for t in range(n_ref):
        n_bin = struct.unpack_from("<i", raw, p)[0]; p += 4
        d = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", raw, p); p += 8
            d[b] = list(struct.iter_unpack("<QQ", raw[p:p + 16 * n_chunk])); p += 16 * n_chunk
        n_intv = struct.unpack_from("<i", raw, p)[0]; p += 4
        lin = struct.unpack_from("<%dQ" % n_intv, raw, p) if n_intv else ()
        p += 8 * n_intv
        if t == want:
            min_off = lin[min(beg >> 14, len(lin) - 1)] if lin else 0
            for b in reg2bins(beg, stop):
                chunks += [(max(s, min_off), e) for s, e in d.get(b, ()) if e > min_off]
    out: List[str] = []
    rd = BgzfReader(path)
    merged: List[Tuple[int, int]] = []
    for cs, ce in sorted(set(chunks)):
        if merged and cs <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(merged[-1][1], ce))
        else:
            merged.append((cs, ce))
    for cs, ce in merged:
        cur = rd.read_from(cs)
        buf = b""
        while cur.tell() < ce:
            piece = cur.read(1)
            if not piece:
                break
            buf += piece
            if piece == b"\n":
                f = buf[:-1].decode().split("\t")
                if f[c_seq - 1] == chrom and int(_num(f[c_beg - 1])) < stop and max(int(_num(f[c_end - 1])), int(_num(f[c_beg - 1])) + 1) > beg:
                    out.append(buf[:-1].decode())
                buf = b""
    return out


def merge_tables(tables: Sequence[str], prefix: str, index: bool = True) -> str:
    """ConcatVaPoR (TasksBenchmark.wdl:249-313) for tables on local disk: header lines dropped, rows sorted,
    written as <prefix>.bed.gz (+ .tbi).  Returns the path."""
    rows: List[str] = []
    for t in tables:
        with open(t) as f:
            for k, ln in enumerate(f):
                ln = ln.rstrip("\n")
                if k == 0 and ln.lstrip("#").startswith("CHR"):
                    continue
                rows.append(ln)
    out = prefix + ".bed.gz"
    write_bed_gz_with_index(out, sort_rows(rows), index)
    return out


# ---------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------
def auto_ranks_per_gpu(n_records: int, gpus: int, cores: int) -> int:
    """Ranks per GPU when the caller does not say: a whole run waits for the host side (the interpreter's share of every
    locus, BAM decompression) far longer than for the kernels, and that share does not thread - so ranks share a GPU, one per
    four cores of the host's quota, four at most (12 000 loci from files on a 16-core box: 2.0 s instead of 2.9 s, scored at
    11 800-13 500 instead of 6 400 loci/s), but only where every rank gets 3 000 records at least: each rank pays the start
    of an interpreter and of a device context."""
    gpus = max(1, gpus)
    return max(1, min(4, cores // (4 * gpus), n_records // (3000 * gpus)))


def _count_records(path: str) -> int:
    try:
        with open(path, "rb") as f:
            return sum(1 for ln in f if ln.strip() and not ln.startswith(b"#"))
    except OSError:
        return 0


def _run_local_ranks(gpus: int, world: int, argv: List[str]) -> int:
    """`world` processes of `python -m vapor_amd.cli argv` on this node, rank r on GPU r modulo `gpus`; returns the first
    non-zero exit code (the others are told to stop waiting for that rank's scores, then ended)."""
    import shutil
    import tempfile
    import time
    d = tempfile.mkdtemp(prefix="vapor_ranks_")
    procs = []
    try:
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world),
                       VAPOR_DIST_BACKEND="files", VAPOR_DIST_DIR=d, VAPOR_LOCAL_GPUS=str(max(gpus, 1)))
            procs.append(subprocess.Popen([sys.executable, "-m", "vapor_amd.cli"] + argv, env=env))
        rc = 0
        left = set(range(world))
        while left:
            for r in sorted(left):
                code = procs[r].poll()
                if code is None:
                    continue
                left.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    open(os.path.join(d, "abort"), "w").close()
            if left:
                time.sleep(0.01)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)


def main(argv: List[str] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    gpus, prefix, index, per_gpu = 1, None, True, "auto"
    while argv and argv[0].startswith("--"):
        if argv[0] == "--gpus":
            gpus = int(argv[1]); argv = argv[2:]
        elif argv[0] == "--ranks-per-gpu":
            per_gpu = argv[1] if argv[1] == "auto" else int(argv[1]); argv = argv[2:]
        elif argv[0] == "--prefix":
            prefix = argv[1]; argv = argv[2:]
        elif argv[0] == "--no-index":
            index = False; argv = argv[1:]
        else:
            break
    if not argv or argv[0] not in ("bed", "vcf", "svelter"):
        print(__doc__)
        return 2
    mode = argv[0]

    def opt(name):
        return argv[argv.index(name) + 1] if name in argv else None
    table = opt("--sv-input") + ".vapor" if mode == "vcf" else opt("--output-file")
    if prefix is None:
        prefix = re.sub(r"\.vapor$", "", table)
    if per_gpu == "auto":
        from . import pipeline
        per_gpu = auto_ranks_per_gpu(_count_records(opt("--sv-input")), gpus, pipeline._usable_cores())
        if per_gpu > 1:
            print("vapor_amd.workflow: %d ranks per GPU (--ranks-per-gpu to choose)" % per_gpu, file=sys.stderr)
    if per_gpu > 1 and os.environ.get("VAPOR_LAUNCHER", "files") != "torchrun":
        # several ranks per GPU (LOCAL_RANK modulo the GPUs): a whole `vapor` run waits for the host side - BAM
        # decompression, CIGAR walks, the per-locus Python - far longer than for the kernels, so ranks that share a GPU scale
        # it until the GPU is busy.  They are started here and hand their scores over through a directory (vapor_amd.dist,
        # the "files" backend): no torch.distributed.run and no process group, whose imports cost 2.5 s per launch.
        rc = _run_local_ranks(gpus, gpus * per_gpu, argv)
    elif gpus * per_gpu > 1:
        # one rank per GPU over RCCL (or, VAPOR_LAUNCHER=torchrun, several per GPU with gloo between them)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus * per_gpu),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29513"),
               "-m", "vapor_amd.cli"] + argv
        rc = subprocess.call(cmd)
    else:
        from . import cli
        rc = cli.main(argv)
    if rc != 0:
        return rc
    print(merge_tables([table], prefix, index))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
