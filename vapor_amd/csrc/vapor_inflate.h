// vapor_inflate.h - a DEFLATE (RFC 1951) decoder for whole BGZF blocks: input and output are complete buffers of known
// sizes, so there is no streaming state, no window copy and no per-block allocation.  Host code; vapor_bam.cpp inflates
// the blocks of a region with it (the read extraction of a `vapor` run from files is bound by inflate time: zlib's
// streaming inflate does ~0.4 GB/s of output per core on BAM data).
//
// Shape: one 64-bit bit buffer refilled eight bytes at a time; two-level lookup tables (11 bits for literals / lengths,
// 8 for distances, second level for longer codes) rebuilt per dynamic block; a fast loop that runs while both buffers
// have slack for the longest match and the widest refill, and a careful loop with exact bounds for the rest.  Every
// malformed stream (over-subscribed or incomplete code that is used, distance before the start, output that does not
// end exactly at `out_n`, input that runs out) is answered with `false`, never with a read or write outside the buffers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <immintrin.h>

namespace vapor_inflate {

constexpr int LL_BITS = 11, D_BITS = 8, PRE_BITS = 7;
constexpr int LL_SYMS = 288, D_SYMS = 32;
constexpr int LL_SIZE = (1 << LL_BITS) + LL_SYMS * 16;      // every long code adds at most a 2^(15-11) second-level table
constexpr int D_SIZE = (1 << D_BITS) + D_SYMS * 128;

// table entry: bits 0-4 code length left to consume (second level: the part beyond the first level), bits 8-12 extra
// bits, bit 13 literal, bit 14 end of block, bit 15 points to a second-level table (payload = its start, bits 8-12 its
// width), bits 16-31 payload (literal, length base, distance base).  Bit 5 (first level of the literal / length table only):
// TWO literals whose codes fit the table's 11 bits together - payload = first | second << 8, bits 0-4 their lengths' sum.
// (Round 5: the decoder is bound by the chain lookup -> drop -> lookup, ~7 cycles a symbol; BAM blocks are literal-heavy -
// packed bases of ~4 bits, qualities of 5-6 - so most table hits can carry two symbols: 0.71 -> 0.95 GB/s on those blocks.)
constexpr uint32_t E_LIT = 1u << 13, E_EOB = 1u << 14, E_SUB = 1u << 15, E_LIT2 = 1u << 5;

struct Decoder {
    uint32_t ll[LL_SIZE];
    uint32_t ds[D_SIZE];
    uint32_t pre[1 << PRE_BITS];
    uint8_t lens[LL_SYMS + D_SYMS + 160];
};

inline uint32_t bit_reverse(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

enum Kind { K_PRE, K_LITLEN, K_DIST };

inline uint32_t symbol_entry(Kind kind, int sym)
{
    if (kind == K_PRE) return (uint32_t)sym << 16;
    if (kind == K_LITLEN) {
        if (sym < 256) return E_LIT | ((uint32_t)sym << 16);
        if (sym == 256) return E_EOB;
        if (sym > 285) return 0xFFFFFFFFu;           // 286, 287: in the fixed code but never valid in data
        return ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 8);
    }
    if (sym > 29) return 0xFFFFFFFFu;
    return ((uint32_t)DIST_BASE[sym] << 16) | ((uint32_t)DIST_EXTRA[sym] << 8);
}

// Canonical Huffman code of `n` symbols with lengths `len` (0 = unused) into a two-level table of `tbits` first-level
// bits and at most `cap` entries.  Unused slots stay 0, entries of symbols that must not appear are 0 as well: the
// decoders treat a 0 entry as an error.  False for an over-subscribed code and for the incomplete codes zlib refuses.
inline bool build_table(Kind kind, const uint8_t* len, int n, uint32_t* tab, int tbits, int cap)
{
    int count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[len[s]];
    count[0] = 0;
    uint32_t next[16];
    uint32_t code = 0;
    long long left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = left * 2 - count[l];
        if (left < 0) return false;
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const int first = 1 << tbits;
    memset(tab, 0, sizeof(uint32_t) * (size_t)first);
    int longest = 15;
    while (longest > 0 && !count[longest]) --longest;
    if (longest == 0) return true;                    // no code at all: every lookup is an error
    // an incomplete code is accepted only as zlib accepts it: a single one-bit code for literals / lengths or distances
    if (left > 0 && (kind == K_PRE || longest != 1)) return false;
    // longest code behind every first-level prefix (for the width of its second-level table)
    uint8_t sub_bits[1 << LL_BITS];
    bool any_long = false;
    for (int l = tbits + 1; l <= 15; ++l) any_long |= count[l] != 0;
    if (any_long) {
        memset(sub_bits, 0, (size_t)first);
        uint32_t nx[16];
        memcpy(nx, next, sizeof nx);
        for (int s = 0; s < n; ++s) {
            const int l = len[s];
            if (l <= tbits) { if (l) ++nx[l]; continue; }
            const uint32_t rev = bit_reverse(nx[l]++, l);
            uint8_t& sb = sub_bits[rev & (uint32_t)(first - 1)];
            if (l - tbits > sb) sb = (uint8_t)(l - tbits);
        }
    }
    int used = first;
    for (int s = 0; s < n; ++s) {
        const int l = len[s];
        if (!l) continue;
        const uint32_t rev = bit_reverse(next[l]++, l);
        const uint32_t e = symbol_entry(kind, s);
        const bool banned = e == 0xFFFFFFFFu;         // its slots stay 0 (every other entry is non-zero: it carries a length)
        if (l <= tbits) {
            const uint32_t v = banned ? 0u : (e | (uint32_t)l);
            for (uint32_t i = rev; i < (uint32_t)first; i += 1u << l) tab[i] = v;
        } else {
            const uint32_t lo = rev & (uint32_t)(first - 1);
            uint32_t& head = tab[lo];
            const int sb = sub_bits[lo];
            if (!(head & E_SUB)) {
                if (used + (1 << sb) > cap) return false;
                head = E_SUB | ((uint32_t)used << 16) | ((uint32_t)sb << 8) | (uint32_t)tbits;
                memset(tab + used, 0, sizeof(uint32_t) * ((size_t)1 << sb));
                used += 1 << sb;
            }
            const uint32_t base = head >> 16;
            const uint32_t v = banned ? 0u : (e | (uint32_t)(l - tbits));
            for (uint32_t i = rev >> tbits; i < (1u << sb); i += 1u << (l - tbits)) tab[base + i] = v;
        }
    }
    return true;
}

// first-level literal entries whose following bits spell another literal inside the table's width become double entries
inline void add_double_literals(uint32_t* tab, int tbits)
{
    const int first = 1 << tbits;
    uint32_t snap[1 << LL_BITS];
    memcpy(snap, tab, sizeof(uint32_t) * (size_t)first);
    for (int i = 0; i < first; ++i) {
        const uint32_t e = snap[i];
        if ((e & (E_LIT | E_SUB)) != E_LIT) continue;
        const int l1 = (int)(e & 31u);
        if (l1 >= tbits) continue;
        const uint32_t e2 = snap[(uint32_t)i >> l1];           // (the bits behind the first code; zeros above them)
        if ((e2 & (E_LIT | E_SUB)) != E_LIT) continue;
        const int l2 = (int)(e2 & 31u);
        if (l1 + l2 > tbits) continue;                          // (the second code must lie inside the bits the index holds)
        tab[i] = E_LIT | E_LIT2 | (uint32_t)(l1 + l2) | (e & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8);
    }
}

struct Bits {
    const uint8_t* in;
    const uint8_t* in_end;
    uint64_t buf = 0;
    int n = 0;               // valid bits in buf

    // at least 56 valid bits while input is left (zeros beyond it)
    inline void refill()
    {
        if (in + 8 <= in_end) {
            uint64_t w;
            memcpy(&w, in, 8);
            buf |= w << n;
            const int adv = (63 - n) >> 3;
            in += adv;
            n += adv * 8;
        } else {
            while (n >= 0 && n <= 56 && in < in_end) { buf |= (uint64_t)*in++ << n; n += 8; }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1ull)); }
    inline void drop(int k)
    {
        buf >>= k;
        n -= k;
    }
    inline uint32_t take(int k)
    {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    inline bool bad() const { return n < 0; }
};

inline const Decoder& fixed_decoder()
{
    static const Decoder* fx = [] {
        Decoder* d = new Decoder();
        uint8_t l[LL_SYMS];
        for (int s = 0; s < 144; ++s) l[s] = 8;
        for (int s = 144; s < 256; ++s) l[s] = 9;
        for (int s = 256; s < 280; ++s) l[s] = 7;
        for (int s = 280; s < 288; ++s) l[s] = 8;
        build_table(K_LITLEN, l, 288, d->ll, LL_BITS, LL_SIZE);
        add_double_literals(d->ll, LL_BITS);
        uint8_t dl[32];
        for (int s = 0; s < 32; ++s) dl[s] = 5;
        build_table(K_DIST, dl, 32, d->ds, D_BITS, D_SIZE);
        return d;
    }();
    return *fx;
}

inline bool read_dynamic(Bits& b, Decoder& d)
{
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    b.refill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t pl[19] = {0};
    for (int i = 0; i < hclen; ++i) {
        if (b.n < 3) b.refill();
        pl[ORDER[i]] = (uint8_t)b.take(3);
    }
    if (b.bad()) return false;
    if (!build_table(K_PRE, pl, 19, d.pre, PRE_BITS, 1 << PRE_BITS)) return false;
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        b.refill();
        const uint32_t e = d.pre[b.peek(PRE_BITS)];
        if (!e) return false;
        b.drop((int)(e & 31u));
        const int sym = (int)(e >> 16);
        if (sym < 16) { d.lens[i++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) {
            if (i == 0) return false;
            val = d.lens[i - 1];
            rep = 3 + (int)b.take(2);
        } else if (sym == 17) {
            rep = 3 + (int)b.take(3);
        } else {
            rep = 11 + (int)b.take(7);
        }
        if (i + rep > total) return false;
        memset(d.lens + i, val, (size_t)rep);
        i += rep;
        if (b.bad()) return false;
    }
    if (d.lens[256] == 0) return false;               // no end-of-block code
    if (!build_table(K_LITLEN, d.lens, hlit, d.ll, LL_BITS, LL_SIZE)) return false;
    add_double_literals(d.ll, LL_BITS);
    if (!build_table(K_DIST, d.lens + hlit, hdist, d.ds, D_BITS, D_SIZE)) return false;
    return true;
}

// the symbols of one Huffman block; false on any error
inline bool inflate_block_body_local(Bits& b, const Decoder& d, uint8_t* const out0, uint8_t*& out, uint8_t* const out_end);

// (the bit buffer and the output cursor live in locals here: the byte stores to `out` may alias anything that is reached
// through a reference, which would send the bit buffer through memory once per symbol)
inline bool inflate_block_body(Bits& bits, const Decoder& d, uint8_t* const out0, uint8_t*& out_ref, uint8_t* const out_end)
{
    Bits b = bits;
    uint8_t* out = out_ref;
    const bool ok = inflate_block_body_local(b, d, out0, out, out_end);
    bits = b;
    out_ref = out;
    return ok;
}

inline __attribute__((always_inline)) bool inflate_block_body_local(Bits& b, const Decoder& d, uint8_t* const out0, uint8_t*& out, uint8_t* const out_end)
{
    constexpr uint32_t LL_MASK = (1u << LL_BITS) - 1u, D_MASK = (1u << D_BITS) - 1u;
    // fast loop: room for the longest match plus a word of slop, input for two refills.  The entry of the next symbol is
    // looked up before the refill that precedes its use (a refill leaves the bits that are there in place), so that the
    // load does not wait for it: `e` is always the first-level entry at the current position, read from at least 11 valid bits.
    if (out_end - out >= 258 + 16 && b.in_end - b.in >= 16) {
        b.refill();
        uint32_t e = d.ll[b.buf & LL_MASK];
        for (;;) {
            b.refill();
            if (e & E_SUB) {
                b.drop(LL_BITS);
                e = d.ll[(e >> 16) + b.peek((int)((e >> 8) & 31u))];
            }
            b.drop((int)(e & 31u));
            if (e & E_LIT) {
                // up to three table hits out of one refill - 45 of its 56 bits at most, 11 are left for the next lookup -, each one
                // literal or two (E_LIT2): both bytes are stored either way (there is slack), the cursor moves by one or two
                uint16_t l0 = (uint16_t)(e >> 16);
                const uint32_t n0 = 1u + ((e >> 5) & 1u);
                e = d.ll[b.buf & LL_MASK];
                memcpy(out, &l0, 2);
                out += n0;
                if ((e & (E_LIT | E_SUB)) == E_LIT) {
                    b.drop((int)(e & 31u));
                    const uint16_t l1 = (uint16_t)(e >> 16);
                    const uint32_t n1 = 1u + ((e >> 5) & 1u);
                    e = d.ll[b.buf & LL_MASK];
                    memcpy(out, &l1, 2);
                    out += n1;
                    if ((e & (E_LIT | E_SUB)) == E_LIT) {
                        b.drop((int)(e & 31u));
                        const uint16_t l2 = (uint16_t)(e >> 16);
                        const uint32_t n2 = 1u + ((e >> 5) & 1u);
                        e = d.ll[b.buf & LL_MASK];
                        memcpy(out, &l2, 2);
                        out += n2;
                    }
                }
                if (!(out_end - out >= 258 + 16 && b.in_end - b.in >= 16)) break;
                continue;
            }
            if (e & E_EOB) return !b.bad();
            if (!e) return false;
            const int xl = (int)((e >> 8) & 31u);
            const uint32_t len = (e >> 16) + b.peek(xl);
            b.drop(xl);
            // (at most 15 + 5 bits used so far; 15 + 13 more for the distance: 48 <= 56)
            uint32_t f = d.ds[b.buf & D_MASK];
            if (f & E_SUB) {
                b.drop(D_BITS);
                f = d.ds[(f >> 16) + b.peek((int)((f >> 8) & 31u))];
            }
            if (!f) return false;
            b.drop((int)(f & 31u));
            const int xd = (int)((f >> 8) & 31u);
            const uint32_t dist = (f >> 16) + b.peek(xd);
            b.drop(xd);
            if (dist > (uint32_t)(out - out0)) return false;
            b.refill();
            e = d.ll[b.buf & LL_MASK];
            const uint8_t* src = out - dist;
            uint8_t* dst = out;
            out += len;
            if (dist >= 8) {
                // (most matches are short: sixteen bytes without a test, the loop only for the rest)
                uint64_t w;
                memcpy(&w, src, 8); memcpy(dst, &w, 8);
                memcpy(&w, src + 8, 8); memcpy(dst + 8, &w, 8);
                if (len > 16) {
                    src += 16; dst += 16;
                    do {
                        memcpy(&w, src, 8); memcpy(dst, &w, 8);
                        src += 8; dst += 8;
                    } while (dst < out);
                }
            } else if (dist == 1) {
                memset(dst, *src, len);
            } else {
                do { *dst++ = *src++; } while (dst < out);
            }
            if (!(out_end - out >= 258 + 16 && b.in_end - b.in >= 16)) break;
        }
    }
    // careful loop
    for (;;) {
        b.refill();
        uint32_t e = d.ll[b.buf & LL_MASK];
        if (e & E_SUB) {
            b.drop(LL_BITS);
            e = d.ll[(e >> 16) + b.peek((int)((e >> 8) & 31u))];
        }
        if (!e) return false;
        b.drop((int)(e & 31u));
        if (b.bad()) return false;
        if (e & E_LIT) {
            if (out >= out_end) return false;
            *out++ = (uint8_t)(e >> 16);
            if (e & E_LIT2) {
                if (out >= out_end) return false;
                *out++ = (uint8_t)(e >> 24);
            }
            continue;
        }
        if (e & E_EOB) return true;
        const int xl = (int)((e >> 8) & 31u);
        const uint32_t len = (e >> 16) + b.peek(xl);
        b.drop(xl);
        uint32_t f = d.ds[b.buf & D_MASK];
        if (f & E_SUB) {
            b.drop(D_BITS);
            f = d.ds[(f >> 16) + b.peek((int)((f >> 8) & 31u))];
        }
        if (!f) return false;
        b.drop((int)(f & 31u));
        const int xd = (int)((f >> 8) & 31u);
        const uint32_t dist = (f >> 16) + b.peek(xd);
        b.drop(xd);
        if (b.bad()) return false;
        if (dist > (uint32_t)(out - out0) || len > (uint32_t)(out_end - out)) return false;
        const uint8_t* src = out - dist;
        for (uint32_t i = 0; i < len; ++i) out[i] = src[i];
        out += len;
    }
}

// Raw DEFLATE stream `in[0, in_n)` -> exactly `out_n` bytes at `out`.  `work` is scratch (one per thread).
inline bool inflate_raw(const uint8_t* in, size_t in_n, uint8_t* out, size_t out_n, Decoder& work)
{
    Bits b;
    b.in = in;
    b.in_end = in + in_n;
    uint8_t* o = out;
    uint8_t* const o_end = out + out_n;
    for (;;) {
        b.refill();
        const uint32_t last = b.take(1), type = b.take(2);
        if (b.bad()) return false;
        if (type == 0) {
            // stored: back to a byte boundary, hand the whole bytes still in the buffer back to the input
            b.drop(b.n & 7);
            b.in -= b.n >> 3;
            b.buf = 0;
            b.n = 0;
            if (b.in_end - b.in < 4) return false;
            const uint32_t len = (uint32_t)b.in[0] | ((uint32_t)b.in[1] << 8), nlen = (uint32_t)b.in[2] | ((uint32_t)b.in[3] << 8);
            b.in += 4;
            if ((len ^ nlen) != 0xFFFFu) return false;
            if ((size_t)(b.in_end - b.in) < len || (size_t)(o_end - o) < len) return false;
            memcpy(o, b.in, len);
            o += len;
            b.in += len;
        } else if (type == 1) {
            if (!inflate_block_body(b, fixed_decoder(), out, o, o_end)) return false;
        } else if (type == 2) {
            if (!read_dynamic(b, work)) return false;
            if (!inflate_block_body(b, work, out, o, o_end)) return false;
        } else {
            return false;
        }
        if (last) break;
    }
    return o == o_end;
}

// CRC-32 of a BGZF block's data (the gzip polynomial, reflected 0xEDB88320).  Where the host has carry-less multiply
// (PCLMULQDQ; asked of the CPU at first use) the bulk goes 64 bytes a step by folding: four 128-bit lanes, each multiplied
// by x^(512+-32) mod P and added to the next 64 bytes, then the lanes into one by x^(128+-32) mod P, 128 -> 64 -> 32 bits
// and a Barrett reduction - Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction"
// (Intel, 2009), the constants for this polynomial as that paper derives them.  About 10 GB/s; the rest of the buffer, and
// hosts without the instruction, take sixteen bytes a step through sixteen tables made from the polynomial at first use
// (slicing-by-16, ~2 GB/s; the byte-at-a-time loop of the system zlib does 1.2 GB/s, half as much again as inflating the block).
struct Crc32Tables {
    uint32_t t[16][256];
    Crc32Tables()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 16; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
    }
};

__attribute__((target("pclmul,sse4.1")))
inline __m128i crc32_fold_step(__m128i x, __m128i k)
{
    return _mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11));
}

// the CRC register (not inverted) over n bytes, n a multiple of 16 and at least 64
__attribute__((target("pclmul,sse4.1")))
inline uint32_t crc32_fold(const uint8_t* p, size_t n, uint32_t c)
{
    const __m128i k1k2 = _mm_set_epi64x(0x1c6e41596LL, 0x154442bd4LL);     // x^(512-32), x^(512+32) mod P (high, low half)
    const __m128i k3k4 = _mm_set_epi64x(0x0ccaa009eLL, 0x1751997d0LL);     // x^(128-32), x^(128+32) mod P
    const __m128i k5 = _mm_set_epi64x(0, 0x163cd6124LL);                   // x^64 mod P
    const __m128i poly_mu = _mm_set_epi64x(0x1f7011641LL, 0x1db710641LL);  // floor(x^64 / P), P
    const __m128i low32 = _mm_set_epi32(0, 0, 0, -1);
    __m128i x1 = _mm_xor_si128(_mm_loadu_si128((const __m128i*)p), _mm_cvtsi32_si128((int)c));
    __m128i x2 = _mm_loadu_si128((const __m128i*)(p + 16)), x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48));
    p += 64; n -= 64;
    while (n >= 64) {
        x1 = _mm_xor_si128(crc32_fold_step(x1, k1k2), _mm_loadu_si128((const __m128i*)p));
        x2 = _mm_xor_si128(crc32_fold_step(x2, k1k2), _mm_loadu_si128((const __m128i*)(p + 16)));
        x3 = _mm_xor_si128(crc32_fold_step(x3, k1k2), _mm_loadu_si128((const __m128i*)(p + 32)));
        x4 = _mm_xor_si128(crc32_fold_step(x4, k1k2), _mm_loadu_si128((const __m128i*)(p + 48)));
        p += 64; n -= 64;
    }
    x1 = _mm_xor_si128(crc32_fold_step(x1, k3k4), x2);
    x1 = _mm_xor_si128(crc32_fold_step(x1, k3k4), x3);
    x1 = _mm_xor_si128(crc32_fold_step(x1, k3k4), x4);
    while (n >= 16) {
        x1 = _mm_xor_si128(crc32_fold_step(x1, k3k4), _mm_loadu_si128((const __m128i*)p));
        p += 16; n -= 16;
    }
    // 128 -> 64 bits (this also appends the 32 zero bits of the CRC's definition), 64 -> 32, Barrett
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), _mm_clmulepi64_si128(k3k4, x1, 0x01));
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, low32), k5, 0x00), x2);
    x2 = x1;
    x1 = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, low32), poly_mu, 0x10), low32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, poly_mu, 0x00), x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

inline uint32_t crc32_tables(const uint8_t* p, size_t n, uint32_t c)
{
    static const Crc32Tables T;
    while (n >= 16) {
        uint32_t a, b, d, e;
        memcpy(&a, p, 4); memcpy(&b, p + 4, 4); memcpy(&d, p + 8, 4); memcpy(&e, p + 12, 4);     // (little-endian host)
        a ^= c;
        c = T.t[15][a & 0xFFu] ^ T.t[14][(a >> 8) & 0xFFu] ^ T.t[13][(a >> 16) & 0xFFu] ^ T.t[12][a >> 24] ^
            T.t[11][b & 0xFFu] ^ T.t[10][(b >> 8) & 0xFFu] ^ T.t[9][(b >> 16) & 0xFFu] ^ T.t[8][b >> 24] ^
            T.t[7][d & 0xFFu] ^ T.t[6][(d >> 8) & 0xFFu] ^ T.t[5][(d >> 16) & 0xFFu] ^ T.t[4][d >> 24] ^
            T.t[3][e & 0xFFu] ^ T.t[2][(e >> 8) & 0xFFu] ^ T.t[1][(e >> 16) & 0xFFu] ^ T.t[0][e >> 24];
        p += 16; n -= 16;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return c;
}

inline uint32_t crc32_fast(const uint8_t* p, size_t n, bool allow_fold = true)
{
    static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    uint32_t c = ~0u;
    if (allow_fold && have_clmul && n >= 64) {
        const size_t bulk = n & ~(size_t)15;
        c = crc32_fold(p, bulk, c);
        p += bulk; n -= bulk;
    }
    return ~crc32_tables(p, n, c);
}

}  // namespace vapor_inflate
