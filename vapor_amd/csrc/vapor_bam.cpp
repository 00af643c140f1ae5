// vapor_bam.cpp - host side of the read extraction: `samtools view bam chrom:start-end` + chop_pacbio_read_by_pos
// (SF:339-354) for one region, straight from a BGZF/BAM file.  No device code.
//
// The reference starts a samtools process per locus and parses its text; vapor_amd/bamio.py does the same work
// in-process in Python (and stays the statement this file is tested against, tests/test_bamio.py); this is the native
// form of its hot loop: the region's BGZF blocks are read with one pread, inflated by a few host threads (vapor_inflate.h),
// the records of the wanted reference are walked in file order, each CIGAR is walked in its binary form up to the window
// start (cigar2alignstart_by_pos, SF:309-337; the CG:B,I long-CIGAR convention included) and only the bases that are
// kept are decoded.  The .bai lookup (bins, linear index) stays in Python: it is a few dictionary reads per locus.
#include "vapor_hip.h"
#include "vapor_inflate.h"

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct vapor_bam {
    int fd = -1;
    std::string path;
    int n_threads = 4;
    // the blocks of the chunk being walked: `comp` holds the file bytes from `comp_base` on, `data` their inflated bytes
    std::vector<uint8_t> comp, data;
    int64_t comp_base = 0;
    size_t scan_pos = 0;                // first byte of `comp` that is not part of a parsed block
    std::vector<int64_t> blk_coff;      // compressed file offset of every block
    std::vector<int64_t> blk_cpos;      // its position inside `comp`
    std::vector<int32_t> blk_csize;     // whole block size (header .. trailer)
    std::vector<int64_t> blk_ustart;    // where its inflated bytes start in `data`
    std::vector<int32_t> blk_usize;
    std::vector<vapor_inflate::Decoder> dec;   // decoder tables, one per inflate thread
};

static thread_local std::string g_bam_err;
extern "C" const char* vapor_bam_last_error(void) { return g_bam_err.c_str(); }
static int bfail(int code, const std::string& m) { g_bam_err = m; return code; }

extern "C" int vapor_bam_open(const char* path, vapor_bam** out)
{
    if (!path || !out) return bfail(VAPOR_E_ARG, "vapor_bam_open: null argument");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return bfail(VAPOR_E_ARG, std::string("vapor_bam_open: cannot open ") + path);
    vapor_bam* b = new vapor_bam();
    b->fd = fd;
    b->path = path;
    const unsigned hc = std::thread::hardware_concurrency();
    b->n_threads = (int)std::max(1u, std::min(4u, hc ? hc / 2 : 1u));
    *out = b;
    return VAPOR_OK;
}

extern "C" int vapor_bam_close(vapor_bam* b)
{
    if (b) {
        if (b->fd >= 0) close(b->fd);
        delete b;
    }
    return VAPOR_OK;
}

// the descriptor and inflate-thread count of an open file, for vapor_bam_chop_device (vapor_hip.hip: it reads the file's blocks with
// positioned reads of its own and leaves the handle's buffers alone)
extern "C" int vapor_bam_fileno(vapor_bam* b) { return b ? b->fd : -1; }
extern "C" int vapor_bam_threads(vapor_bam* b) { return b ? b->n_threads : 1; }

extern "C" int vapor_bam_set_threads(vapor_bam* b, int32_t n)
{
    if (!b || n < 1 || n > 64) return bfail(VAPOR_E_ARG, "vapor_bam_set_threads: out of range");
    b->n_threads = n;
    return VAPOR_OK;
}

// Parses the BGZF block headers from scan_pos on and appends the whole blocks found to the block lists (a truncated
// last block is left for the next read); -1 when the bytes are not BGZF.
static int scan_blocks(vapor_bam* b)
{
    int n = 0;
    size_t p = b->scan_pos;
    while (p + 18 <= b->comp.size()) {
        const uint8_t* h = b->comp.data() + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return -1;
        const int xlen = h[10] | (h[11] << 8);
        if (p + 12 + (size_t)xlen > b->comp.size()) break;
        int bsize = -1;
        for (int q = 0; q + 4 <= xlen;) {
            const uint8_t* e = h + 12 + q;
            const int slen = e[2] | (e[3] << 8);
            if (e[0] == 66 && e[1] == 67 && slen == 2) bsize = (e[4] | (e[5] << 8)) + 1;
            q += 4 + slen;
        }
        // (a block is its 12-byte header, the extra field, the payload, CRC32 and ISIZE: anything shorter is not one, and its
        // trailer would be read from before the block)
        if (bsize < 0 || bsize < xlen + 20) return -1;
        if (p + (size_t)bsize > b->comp.size()) break;
        const uint8_t* t = h + bsize - 4;
        const uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > 65536u) return -1;                       // BGZF: at most 64 KB of data per block
        b->blk_coff.push_back(b->comp_base + (int64_t)p);
        b->blk_cpos.push_back((int64_t)p);
        b->blk_csize.push_back(bsize);
        b->blk_usize.push_back((int32_t)isize);
        p += (size_t)bsize;
        ++n;
    }
    b->scan_pos = p;
    return n;
}

// the block's own CRC32 (the four bytes before ISIZE) against the inflated bytes, as htslib checks it: a decoder bug or a
// damaged block that still inflates to ISIZE bytes must not become reads
static bool crc_ok(const uint8_t* blk, int bsize, const uint8_t* out, int isize)
{
    const uint8_t* t = blk + bsize - 8;
    const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    return vapor_inflate::crc32_fast(out, (size_t)isize) == want;
}

static bool inflate_block(const uint8_t* blk, int bsize, uint8_t* out, int isize, vapor_inflate::Decoder& dec)
{
    const int xlen = blk[10] | (blk[11] << 8);
    if (bsize - xlen - 20 < 0 || isize < 0 || isize > 65536) return false;
    if (isize == 0) return crc_ok(blk, bsize, out, 0);
    if (vapor_inflate::inflate_raw(blk + 12 + xlen, (size_t)(bsize - xlen - 20), out, (size_t)isize, dec)) return crc_ok(blk, bsize, out, isize);
    // refused: zlib has the last word on whether the block is damaged
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(blk + 12 + xlen);
    zs.avail_in = (uInt)(bsize - xlen - 20);
    zs.next_out = out;
    zs.avail_out = (uInt)isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0 && crc_ok(blk, bsize, out, isize);
}

extern "C" int vapor_inflate_raw(const uint8_t* in, int64_t in_n, uint8_t* out, int64_t out_n)
{
    if (in_n < 0 || out_n < 0 || (in_n && !in) || (out_n && !out)) return bfail(VAPOR_E_ARG, "vapor_inflate_raw: bad argument");
    static thread_local vapor_inflate::Decoder* dec = nullptr;
    if (!dec) dec = new vapor_inflate::Decoder();
    uint8_t none = 0;
    if (!vapor_inflate::inflate_raw(in ? in : &none, (size_t)in_n, out ? out : &none, (size_t)out_n, *dec))
        return bfail(VAPOR_E_ARG, "vapor_inflate_raw: not a DEFLATE stream of that size");
    return VAPOR_OK;
}

// the CRC-32 the block trailers are checked with, by itself (tests compare it with zlib's; `tables_only` takes the path of a
// host without carry-less multiply)
extern "C" uint32_t vapor_crc32(const uint8_t* data, int64_t n, int32_t tables_only)
{
    if (!data || n <= 0) return 0u;
    return vapor_inflate::crc32_fast(data, (size_t)n, !tables_only);
}

// inflates blocks [b0, b1) into data (their ustart already laid out), a few threads when there are several
static bool inflate_range(vapor_bam* b, size_t b0, size_t b1)
{
    const size_t n = b1 - b0;
    const int nt = (int)std::min<size_t>((size_t)b->n_threads, n);
    if (b->dec.size() < (size_t)std::max(nt, 1)) b->dec.resize((size_t)std::max(nt, 1));
    std::vector<char> ok((size_t)std::max(nt, 1), 1);
    auto work = [&](int t) {
        for (size_t i = b0 + (size_t)t; i < b1; i += (size_t)nt)
            if (!inflate_block(b->comp.data() + b->blk_cpos[i], b->blk_csize[i], b->data.data() + b->blk_ustart[i], b->blk_usize[i], b->dec[(size_t)t])) ok[(size_t)t] = 0;
    };
    if (nt <= 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    for (char c : ok)
        if (!c) return false;
    return true;
}

// lays the next scanned blocks, up to block b1, out behind the inflated bytes and inflates them (blocks are inflated in file
// order: the first blk_ustart.size() of the scanned ones are)
static bool take_blocks(vapor_bam* b, size_t b1)
{
    const size_t b0 = b->blk_ustart.size();
    b1 = std::min(b1, b->blk_coff.size());
    if (b1 <= b0) return true;
    int64_t u = (int64_t)b->data.size();
    for (size_t i = b0; i < b1; ++i) { b->blk_ustart.push_back(u); u += b->blk_usize[i]; }
    b->data.resize((size_t)u);
    return inflate_range(b, b0, b1);
}

// reads `bytes` more file bytes behind `comp`; false at end of file
static bool read_more(vapor_bam* b, size_t bytes)
{
    const size_t old = b->comp.size();
    b->comp.resize(old + bytes);
    const ssize_t got = pread(b->fd, b->comp.data() + old, bytes, (off_t)(b->comp_base + (int64_t)old));
    b->comp.resize(old + (size_t)std::max<ssize_t>(got, 0));
    return got > 0;
}

// makes sure `data` holds at least `upto` inflated bytes (a record may run past the chunk's own blocks); false at the end
// of the file or on bytes that are not BGZF
static thread_local bool g_bam_damaged = false;          // ensure() failed on damaged bytes, not at the end of the file
static bool ensure(vapor_bam* b, int64_t upto)
{
    g_bam_damaged = false;
    while ((int64_t)b->data.size() < upto) {
        // blocks that were read with the chunk but lie behind its end (a record runs into them), a few at a time
        if (b->blk_ustart.size() < b->blk_coff.size()) {
            if (!take_blocks(b, b->blk_ustart.size() + (size_t)std::max(b->n_threads, 1))) { g_bam_damaged = true; return false; }
            continue;
        }
        if (!read_more(b, (size_t)1 << 17)) return false;
        if (scan_blocks(b) < 0) { g_bam_damaged = true; return false; }
    }
    return true;
}

static inline int32_t rd32(const uint8_t* p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }

// the CG:B,I array of a record's aux fields, or null
static const uint8_t* find_cg(const uint8_t* p, const uint8_t* end, int32_t* count)
{
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        int sz = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': { while (p < end && *p) ++p; if (p >= end) return nullptr; ++p; continue; }
        case 'B': {
            if (p + 5 > end) return nullptr;
            const uint8_t sub = p[0];
            const int32_t cnt = rd32(p + 1);
            p += 5;
            const int es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            // (the array must lie inside the record)
            if (cnt < 0 || (int64_t)cnt * es > (int64_t)(end - p)) return nullptr;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') { *count = cnt; return p; }
            p += (size_t)cnt * (size_t)es;
            continue;
        }
        default: return nullptr;
        }
        if (sz > end - p) return nullptr;
        p += sz;
    }
    return nullptr;
}

// Every size field of the file is checked before it is used (a damaged ISIZE reached the inflaters as a
// buffer size); a file that breaks a rule is VAPOR_E_ARG with a message, never a read or write outside `comp` / `data`.
static int bam_chop_impl(vapor_bam* b, int32_t tid, int64_t start, int64_t end, int64_t flank, int32_t n_chunks,
                         const uint64_t* chunks, uint8_t* seq_out, int64_t seq_cap, char* names_out, int64_t names_cap,
                         int64_t* meta, int32_t max_reads, int32_t* n_reads, int64_t* need)
{
    if (!b || !n_reads || (n_chunks && !chunks)) return bfail(VAPOR_E_ARG, "vapor_bam_chop: null argument");
    if (seq_cap < 0 || names_cap < 0 || max_reads < 0 || (max_reads && (!meta || !seq_out || !names_out)))
        return bfail(VAPOR_E_ARG, "vapor_bam_chop: bad output buffers");
    static const char* NT16 = "=ACMGRSVTWYHKDBN";
    const int64_t beg = std::max<int64_t>(start - 1, 0), stop = end;     // 0-based half-open region
    int64_t seq_used = 0, names_used = 0;
    int32_t nr = 0;
    bool overflow = false;
    for (int32_t c = 0; c < n_chunks; ++c) {
        const uint64_t cs = chunks[2 * c], ce = chunks[2 * c + 1];
        // the chunk's own compressed range in one read, its blocks inflated together
        b->comp.clear(); b->data.clear(); b->scan_pos = 0;
        b->blk_coff.clear(); b->blk_cpos.clear(); b->blk_csize.clear(); b->blk_ustart.clear(); b->blk_usize.clear();
        b->comp_base = (int64_t)(cs >> 16);
        // (index chunks come from a file as well: an end before the start, or gigabytes for one locus, is a damaged .bai)
        if (ce < cs || (ce >> 16) - (cs >> 16) > ((uint64_t)1 << 31))
            return bfail(VAPOR_E_ARG, "vapor_bam_chop: implausible index chunk for " + b->path);
        // through the block that holds the chunk's end (none of it when the chunk ends on a block boundary)
        const size_t span = (size_t)((int64_t)(ce >> 16) - b->comp_base) + ((ce & 0xFFFFu) ? ((size_t)1 << 16) + 64 : 0);
        if (span == 0 || !read_more(b, span)) continue;
        if (scan_blocks(b) < 0) return bfail(VAPOR_E_ARG, "vapor_bam_chop: not a BGZF block in " + b->path);
        // the blocks the chunk covers (the read holds a few more behind its end block: those wait until a record needs them)
        size_t own = 0;
        while (own < b->blk_coff.size() && ((uint64_t)b->blk_coff[own] << 16) < ce) ++own;
        if (!take_blocks(b, own)) return bfail(VAPOR_E_ARG, "vapor_bam_chop: inflate failed in " + b->path);
        int64_t pos_u = (int64_t)(cs & 0xFFFF);                              // position in `data`
        size_t blk = 0;
        for (;;) {
            // virtual offset of pos_u
            // virtual offset of pos_u: inside a block that is here, or the start of the block that follows them - decided
            // before anything more is read, so that the end of the chunk does not cost another read and inflate
            uint64_t voff;
            if (pos_u >= (int64_t)b->data.size()) {
                const size_t taken = b->blk_ustart.size();
                if (taken == 0) break;
                voff = (uint64_t)(b->blk_coff[taken - 1] + b->blk_csize[taken - 1]) << 16;
            } else {
                while (blk + 1 < b->blk_ustart.size() && pos_u >= b->blk_ustart[blk + 1]) ++blk;
                voff = ((uint64_t)b->blk_coff[blk] << 16) | (uint64_t)(pos_u - b->blk_ustart[blk]);
            }
            if (voff >= ce) break;
            if (!ensure(b, pos_u + 4)) {
                // the end of the file inside a chunk is the end of its records; bytes that are not BGZF, or a block that
                // does not inflate to its CRC, are an error
                if (!g_bam_damaged) break;
                return bfail(VAPOR_E_ARG, "vapor_bam_chop: damaged BGZF block in " + b->path);
            }
            const int32_t bs = rd32(b->data.data() + pos_u);
            if (bs < 32 || bs > (1 << 29)) return bfail(VAPOR_E_ARG, "vapor_bam_chop: implausible record size in " + b->path);
            if (!ensure(b, pos_u + 4 + bs)) return bfail(VAPOR_E_ARG, "vapor_bam_chop: truncated record (or damaged block) in " + b->path);
            const uint8_t* r = b->data.data() + pos_u + 4;
            pos_u += 4 + bs;
            const int32_t ref_id = rd32(r), pos = rd32(r + 4);
            const int l_name = r[8];
            const int n_cig = r[12] | (r[13] << 8);
            const int32_t l_seq = rd32(r + 16);
            // name, CIGAR, packed bases and qualities must lie inside the record
            if (l_seq < 0 || 32 + (int64_t)l_name + 4 * (int64_t)n_cig + ((int64_t)l_seq + 1) / 2 + (int64_t)l_seq > (int64_t)bs)
                return bfail(VAPOR_E_ARG, "vapor_bam_chop: record fields exceed the record in " + b->path);
            if (ref_id != tid || pos >= stop) {
                if (ref_id > tid || (ref_id == tid && pos >= stop)) break;
                continue;
            }
            const uint8_t* name = r + 32;
            const uint8_t* cig = name + l_name;
            const uint8_t* sq = cig + 4 * n_cig;
            const uint8_t* rec_end = r + bs;
            const uint8_t* ops = cig;
            int32_t n_ops = n_cig;
            if (n_cig == 2) {
                const uint32_t o0 = (uint32_t)rd32(cig), o1 = (uint32_t)rd32(cig + 4);
                if ((o0 & 15u) == 4u && (int32_t)(o0 >> 4) == l_seq && (o1 & 15u) == 3u) {
                    int32_t cnt = 0;
                    const uint8_t* cg = find_cg(sq + (l_seq + 1) / 2 + l_seq, rec_end, &cnt);
                    if (cg) { ops = cg; n_ops = cnt; }
                }
            }
            // reference length; the region-overlap rule of `samtools view`
            // (only as far as the answer: a long read's thousands of operations end far behind the window)
            int64_t rlen = 0;
            for (int32_t t = 0; t < n_ops && (int64_t)pos + rlen <= beg; ++t) {
                const uint32_t o = (uint32_t)rd32(ops + 4 * t), code = o & 15u;
                if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) rlen += o >> 4;
            }
            if ((int64_t)pos + std::max<int64_t>(rlen, 1) <= beg) continue;
            // chop_pacbio_read_by_pos: only alignments that start at or before the window start
            if (!((int64_t)pos + 1 < start + 1)) continue;
            if (n_ops <= 0) return bfail(VAPOR_E_ARG, "vapor_bam_chop: record without CIGAR (the reference raises IndexError, SF:331)");
            int64_t q = 0, rr = (int64_t)pos + 1;
            uint32_t last = 0;
            for (int32_t t = 0; t < n_ops; ++t) {
                const uint32_t o = (uint32_t)rd32(ops + 4 * t);
                const int64_t n = o >> 4;
                last = o & 15u;
                if (last == 4u || last == 1u) q += n;
                else if (last == 0u || last == 7u) { q += n; rr += n; }
                else if (last == 2u) rr += n;
                if (rr > start - 1) break;
            }
            const int64_t over = rr - start;
            int64_t q0, miss;
            if (last == 0u || last == 7u) { q0 = q - over; miss = 0; } else { q0 = q; miss = over; }
            if (2 * miss > flank) continue;                                   // miss_bp > flank_length / 2
            const int64_t seq_len = l_seq > 0 ? l_seq : 1;                    // an absent sequence reads "*"
            const int64_t tail = q0 < seq_len ? seq_len - std::max<int64_t>(q0, 0) : 0;
            const int64_t want_len = end - start - miss;
            if (q0 < 0) return bfail(VAPOR_E_ARG, "vapor_bam_chop: negative read offset");
            if (want_len < 0 || !(tail > want_len)) continue;
            const int64_t nl = l_name > 0 ? l_name - 1 : 0;
            if (nr >= max_reads || seq_used + want_len > seq_cap || names_used + nl + 1 > names_cap) {
                overflow = true;
                seq_used += want_len; names_used += nl + 1; ++nr;
                continue;
            }
            uint8_t* dst = seq_out + seq_used;
            if (l_seq > 0) {
                int64_t t = 0, i = q0;
                if ((i & 1) && t < want_len) { dst[t++] = (uint8_t)NT16[sq[i >> 1] & 15]; ++i; }
                for (; t + 2 <= want_len; t += 2, i += 2) {              // two bases a byte
                    const uint8_t byte = sq[i >> 1];
                    dst[t] = (uint8_t)NT16[byte >> 4];
                    dst[t + 1] = (uint8_t)NT16[byte & 15];
                }
                if (t < want_len) dst[t] = (uint8_t)NT16[sq[i >> 1] >> 4];
            } else if (want_len > 0) {
                dst[0] = '*';
            }
            memcpy(names_out + names_used, name, (size_t)nl);
            names_out[names_used + nl] = 0;
            meta[4 * nr] = seq_used; meta[4 * nr + 1] = want_len; meta[4 * nr + 2] = miss; meta[4 * nr + 3] = names_used;
            seq_used += want_len; names_used += nl + 1; ++nr;
        }
    }
    *n_reads = nr;
    if (need) { need[0] = seq_used; need[1] = names_used; need[2] = nr; }
    if (overflow) return bfail(VAPOR_E_OVERFLOW, "vapor_bam_chop: output buffers too small");
    return VAPOR_OK;
}

extern "C" int vapor_bam_chop(vapor_bam* b, int32_t tid, int64_t start, int64_t end, int64_t flank, int32_t n_chunks,
                              const uint64_t* chunks, uint8_t* seq_out, int64_t seq_cap, char* names_out, int64_t names_cap,
                              int64_t* meta, int32_t max_reads, int32_t* n_reads, int64_t* need)
{
    // no exception crosses the C boundary (a std::bad_alloc / length_error from a buffer resize would end the process)
    try {
        return bam_chop_impl(b, tid, start, end, flank, n_chunks, chunks, seq_out, seq_cap, names_out, names_cap, meta, max_reads, n_reads, need);
    } catch (const std::bad_alloc&) {
        return bfail(VAPOR_E_NOMEM, "vapor_bam_chop: out of memory");
    } catch (const std::exception& e) {
        return bfail(VAPOR_E_ARG, std::string("vapor_bam_chop: ") + e.what());
    }
}

// chop_pacbio_read_by_pos (SF:339-354) over alignment records that are in memory already (a caller that holds its reads as
// objects - the synthetic worlds of the tests and benches, a reader of another format): the region rule of `samtools view`,
// the reference's `POS < start + 1`, the CIGAR walk to the window start (cigar2alignstart_by_pos, SF:309-337, over the CIGAR
// TEXT, as vapor_cigar2alignstart walks it: the walk ends where the reference cursor passes the window start, a few dozen
// operations into a long read's thousands, so nothing is parsed ahead of time), `miss_bp > flank / 2` and
// `len(read[q0:]) > end - start - miss_bp`, for n records in one call.  keep[r] = 1 and q0_miss[2r], q0_miss[2r+1] = offset
// into the read and miss_bp for the reads the reference keeps.  VAPOR_E_ARG for a record without CIGAR operation that
// reaches the walk (IndexError in the reference, SF:331).
extern "C" int vapor_chop_records(int32_t n, const int64_t* pos, const int64_t* ref_span, const char* const* cigar,
                                  const int64_t* seq_len, int64_t start, int64_t end, int64_t flank,
                                  int64_t* q0_miss, uint8_t* keep)
{
    if (n < 0 || (n && (!pos || !ref_span || !cigar || !seq_len || !q0_miss || !keep)))
        return bfail(VAPOR_E_ARG, "vapor_chop_records: null argument");
    for (int32_t r = 0; r < n; ++r) {
        keep[r] = 0;
        if (!(pos[r] <= end && pos[r] + ref_span[r] - 1 >= start)) continue;      // not in the region
        if (!(pos[r] < start + 1)) continue;
        int64_t q = 0, rr = pos[r], num = 0;
        bool have_n = false;
        char last = 0;
        for (const char* c = cigar[r] ? cigar[r] : ""; *c; ++c) {
            const char ch = *c;
            if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); have_n = true; continue; }
            const bool op = ch == 'M' || ch == 'I' || ch == 'D' || ch == 'N' || ch == 'S' || ch == 'H' || ch == 'P' || ch == '=' || ch == 'X';
            if (op && have_n) {
                if (ch == 'S' || ch == 'I') q += num;
                else if (ch == 'M' || ch == '=') { q += num; rr += num; }
                else if (ch == 'D') rr += num;
                last = ch;
                if (rr > start - 1) break;
            }
            num = 0; have_n = false;           // any other character ends the number, as the regular expression would
        }
        if (!last) return bfail(VAPOR_E_ARG, "vapor_chop_records: record without CIGAR (the reference raises IndexError, SF:331)");
        const int64_t over = rr - start;
        int64_t q0, miss;
        if (last == 'M' || last == '=') { q0 = q - over; miss = 0; } else { q0 = q; miss = over; }
        if (2 * miss > flank) continue;                                    // miss_bp > flank_length / 2
        const int64_t want = end - start - miss;
        // len(seq[q0:]) as Python slices (a negative q0 counts from the end)
        const int64_t from = q0 < 0 ? std::max<int64_t>(seq_len[r] + q0, 0) : std::min(q0, seq_len[r]);
        if (!(seq_len[r] - from > want)) continue;
        keep[r] = 1;
        q0_miss[2 * r] = q0;
        q0_miss[2 * r + 1] = miss;
    }
    return VAPOR_OK;
}

// chop_pacbio_read_by_pos (SF:339-354) and minimize_pacbio_read_list (SF:1091-1102: at most `max_keep` reads, smallest miss_bp
// first, input order inside one miss_bp value) for MANY regions in one call: region g looks at n_rec[g] records given as the
// arrays of vapor_chop_records (one pointer per region), keeps what that call keeps and, when there are more than max_keep,
// the first max_keep of them in a stable order by miss_bp.  kept_first[g] .. kept_first[g + 1] index rec_idx / q0 / miss
// (capacity max_keep per region); addr_out (may be NULL) receives seq_addr[g][record] for every kept read - where the caller
// keeps the records' sequences.  status[g] = 0, or VAPOR_E_ARG where a record without CIGAR reaches the walk (the reference
// raises IndexError there, SF:331: the caller lets that region take the reference-named route).
extern "C" int vapor_chop_records_many(int32_t n_regions, const int32_t* n_rec, const int64_t* const* pos,
                                       const int64_t* const* ref_span, const char* const* const* cigar,
                                       const int64_t* const* seq_len, const int64_t* start, const int64_t* end,
                                       const int64_t* flank, int32_t max_keep, int32_t* kept_first, int32_t* rec_idx,
                                       int64_t* q0, int64_t* miss, int32_t* status, const uint64_t* const* seq_addr,
                                       uint64_t* addr_out)
{
    if (n_regions < 0 || max_keep < 1 || (n_regions && (!n_rec || !pos || !ref_span || !cigar || !seq_len || !start || !end || !flank ||
                                                           !kept_first || !rec_idx || !q0 || !miss || !status)))
        return bfail(VAPOR_E_ARG, "vapor_chop_records_many: null argument");
    // every region into slots of its own (max_keep of them), on a few threads when there are many regions; then packed
    std::vector<int32_t> cnt((size_t)n_regions, 0), t_rec((size_t)n_regions * max_keep);
    std::vector<int64_t> t_q0((size_t)n_regions * max_keep), t_miss((size_t)n_regions * max_keep);
    auto work = [&](int32_t g0, int32_t g1) {
        std::vector<int64_t> qm;
        std::vector<uint8_t> keep;
        std::vector<int32_t> order;
        for (int32_t g = g0; g < g1; ++g) {
            status[g] = 0;
            const int32_t n = n_rec[g];
            if (n <= 0) continue;
            qm.resize((size_t)2 * n);
            keep.resize((size_t)n);
            if (vapor_chop_records(n, pos[g], ref_span[g], cigar[g], seq_len[g], start[g], end[g], flank[g], qm.data(), keep.data()) != VAPOR_OK) {
                status[g] = VAPOR_E_ARG;
                continue;
            }
            order.clear();
            for (int32_t r = 0; r < n; ++r)
                if (keep[(size_t)r]) order.push_back(r);
            if ((int32_t)order.size() > max_keep) {
                std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return qm[(size_t)2 * a + 1] < qm[(size_t)2 * b + 1]; });
                order.resize((size_t)max_keep);
            }
            int32_t c = 0;
            for (int32_t r : order) {
                const size_t o = (size_t)g * max_keep + (size_t)c++;
                t_rec[o] = r; t_q0[o] = qm[(size_t)2 * r]; t_miss[o] = qm[(size_t)2 * r + 1];
            }
            cnt[(size_t)g] = c;
        }
    };
    const int n_thr = n_regions >= 512 ? 4 : 1;
    if (n_thr == 1) {
        work(0, n_regions);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t)
            th.emplace_back(work, (int32_t)((int64_t)n_regions * t / n_thr), (int32_t)((int64_t)n_regions * (t + 1) / n_thr));
        for (auto& x : th) x.join();
    }
    int32_t w = 0;
    for (int32_t g = 0; g < n_regions; ++g) {
        kept_first[g] = w;
        for (int32_t c = 0; c < cnt[(size_t)g]; ++c) {
            const size_t o = (size_t)g * max_keep + (size_t)c;
            rec_idx[w] = t_rec[o]; q0[w] = t_q0[o]; miss[w] = t_miss[o];
            if (addr_out) addr_out[w] = (seq_addr && seq_addr[g]) ? seq_addr[g][t_rec[o]] : 0;
            ++w;
        }
    }
    kept_first[n_regions] = w;
    return VAPOR_OK;
}

// The row tails of a whole table in one call (no device): what result_organize_ins (SF:1219-1231) and
// gt_estimate_log_likelihood (SF:2054-2069) compute per locus from its read scores, minus the parts that are table lookups on
// the caller's side.  Per locus t with scores[off[t] .. off[t+1]):
//   n_pos[t]    = scores > 0 (GS = n_pos / n);  qs[t] = np.mean of those, summed as numpy's add.reduce sums a contiguous
//                 double array (pairwise: blocks of at most 128 with eight strided partial sums, SF:1225's np.mean)
//   n_nonpos[t] = scores that are not > 0 AFTER round(s, 2) - the l of log_likelihood_calcu (SF:2071-2077), which the
//                 reference reads back from the Rec string
//   rec         = ','.join(str(round(s, 2))): text[text_off[t] .. text_off[t+1]).  str(round(s, 2)) of a float is the
//                 correctly rounded two-decimal numeral without trailing zeros but with one decimal at least ("0.5", "-1.0",
//                 "-0.0"): round() rounds the exact binary value half-even through the shortest-string machinery, and below
//                 2^46 two doubles are less than 0.005 apart so no shorter numeral reads back as the same double.
//                 n_nonpos[t] = -1 instead where a score is not finite or is 1e13 and more in size: the caller formats that
//                 locus itself.
// VAPOR_E_OVERFLOW with text_off[n_loci] = bytes needed when text_cap is too small (16 bytes a score are always enough for
// |s| < 1e13).
namespace {
double pairwise_sum(const double* a, int64_t n)
{
    if (n < 8) {
        double r = 0.;          // (numpy starts from -0.0; adding it to the first value gives the same double except for an
        for (int64_t i = 0; i < n; ++i) r += a[i];     //  all-negative-zero list, which cannot be: these values are > 0)
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

// the numeral of round(x, 2) as Python prints it; returns its length, sets *positive to (rounded value > 0)
int two_decimals(double x, char* out, bool* positive)
{
    const double y = x * 100.;
    const double fl = std::floor(y);
    double cents;               // round(x, 2) * 100 as an integer-valued double (sign kept apart for -0.0)
    // x * 100 carries a relative error below 2^-53: away from a tie by more than that, the side it falls on is the exact one
    if (std::fabs((y - fl) - .5) > 1e-9 * (std::fabs(y) + 1.)) {
        cents = std::nearbyint(y);
    } else {
        char buf[40];
        std::snprintf(buf, sizeof buf, "%.2f", x);       // exact, half-even on the binary value
        cents = std::nearbyint(std::strtod(buf, nullptr) * 100.);
    }
    const bool neg = std::signbit(x);
    uint64_t c = (uint64_t)std::fabs(cents);
    *positive = !neg && c > 0;
    char tmp[32];
    int m = 0;
    const unsigned frac = (unsigned)(c % 100);
    c /= 100;
    do { tmp[m++] = (char)('0' + c % 10); c /= 10; } while (c);
    int len = 0;
    if (neg) out[len++] = '-';
    while (m) out[len++] = tmp[--m];
    out[len++] = '.';
    out[len++] = (char)('0' + frac / 10);
    if (frac % 10) out[len++] = (char)('0' + frac % 10);
    return len;
}
}   // namespace

extern "C" int vapor_row_tails(int32_t n_loci, const int64_t* off, const double* scores, double* qs, int32_t* n_pos,
                               int32_t* n_nonpos, char* text, int64_t text_cap, int64_t* text_off)
{
    if (n_loci < 0 || !off || !text_off || (n_loci && (!qs || !n_pos || !n_nonpos)) || text_cap < 0 || (text_cap && !text))
        return bfail(VAPOR_E_ARG, "vapor_row_tails: null argument");
    for (int32_t t = 0; t < n_loci; ++t)
        if (off[t + 1] < off[t] || off[t] < 0) return bfail(VAPOR_E_ARG, "vapor_row_tails: offsets must not decrease");
    if (n_loci && off[n_loci] > 0 && !scores) return bfail(VAPOR_E_ARG, "vapor_row_tails: null scores");
    try {
        std::vector<double> pos;
        int64_t w = 0;
        bool fits = true;
        for (int32_t t = 0; t < n_loci; ++t) {
            const double* s = scores + off[t];
            const int64_t n = off[t + 1] - off[t];
            text_off[t] = w;
            pos.clear();
            int32_t nonpos = 0;
            bool plain = true;
            for (int64_t i = 0; i < n; ++i) {
                if (s[i] > 0) pos.push_back(s[i]);
                if (!(std::fabs(s[i]) < 1e13)) { plain = false; continue; }      // (also NaN)
                char num[40];
                bool positive;
                const int len = two_decimals(s[i], num, &positive);
                nonpos += !positive;
                if (fits && w + len + 1 <= text_cap) {
                    if (i) text[w++] = ',';
                    std::memcpy(text + w, num, (size_t)len);
                    w += len;
                } else {
                    fits = false;
                    w += len + (i ? 1 : 0);
                }
            }
            n_pos[t] = (int32_t)pos.size();
            qs[t] = pos.empty() ? 0. : pairwise_sum(pos.data(), (int64_t)pos.size()) / (double)pos.size();
            n_nonpos[t] = plain ? nonpos : -1;
        }
        text_off[n_loci] = w;
        return fits ? VAPOR_OK : bfail(VAPOR_E_OVERFLOW, "vapor_row_tails: text buffer too small");
    } catch (const std::bad_alloc&) {
        return bfail(VAPOR_E_NOMEM, "vapor_row_tails: out of memory");
    }
}
