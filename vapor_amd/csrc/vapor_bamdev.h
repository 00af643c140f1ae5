// vapor_bamdev.h - the read extraction of a batch of loci ON the device: `samtools view bam chrom:start-end`
// piped into chop_pacbio_read_by_pos (SF:339-354), which the reference runs as a process per locus and vapor_bam.cpp runs on
// host threads, for hundreds of regions at once.  The host reads the regions' BGZF blocks as they lie in the file and sends
// them over the link COMPRESSED (a third to a fifth of their inflated size); three kernels do the rest:
//
//   bgzf_inflate_kernel   one wavefront per BGZF block: DEFLATE (RFC 1951) decoded by lane 0 out of an LDS copy of the
//                         input straight into the block's place in the arena (HBM), the matches of a batch copied by all 64
//                         lanes, the block's CRC-32 taken by 64 lanes (a slice each, combined by multiplication mod P).
//                         Thousands of blocks are independent: that is the parallelism (a single DEFLATE stream has none),
//                         and twenty of them share a CU so that one's table lookup waits while another's shifts.
//   bam_chop_kernel       one wavefront per region: the BAM records of its index chunks walked in file order, the region
//                         rule of `samtools view`, POS <= start, the CIGAR walked to the window start 64 operations a step
//                         (cigar2alignstart_by_pos, SF:309-337; the CG:B,I long-CIGAR convention included), miss_bp and
//                         the length rule (SF:346-352) - the kept reads as (address of the packed bases, first base, miss_bp).
//   bam_expand_kernel     (vapor_seqset_create_mixed) the kept bases, 4 bits each in the arena, to the ASCII staging layout
//                         pack_kernel reads - what a host buffer and its copy over the link were before.
//
// Anything these kernels do not decide themselves - a block whose Huffman tables exceed the LDS tables, a record that runs
// past the blocks that were sent, a malformed field, a record without CIGAR, more kept reads than a region's slot holds -
// marks the REGION, and the caller sends that region down the host route (vapor_bam_chop), which also words the errors.
//
// The decoder core is written against a small set of macros so that tools/bamdev_emu.cpp compiles it for the host (one
// "lane" doing the wavefront's loops in order) and checks it against zlib on this machine; the kernels proper are HIP.
#pragma once
#include <stdint.h>

#ifndef VBD_EMU
#include <hip/hip_runtime.h>
#define VBD_DEV __device__ __forceinline__
// the lanes of one wavefront hand data to each other through LDS and through the block's own output in HBM: a wavefront's memory
// operations are performed in program order, so a fence of wavefront scope (which also keeps the compiler from moving accesses
// across it) is all the ordering there is to ask for - no workgroup barrier: the wavefronts of a workgroup decode different blocks
#define VBD_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define VBD_WAVE_FOR(i, a, b) for (int i = (a) + (int)lane; i < (b); i += 64)
#define VBD_EACH_LANE(l) for (int l = (int)lane, once_ = 1; once_; once_ = 0)
#define VBD_LANE0 lane == 0
#define VBD_LANE_SLOTS 1
// Lane 0's decoding is scalar work: what it reads from LDS is declared uniform (v_readfirstlane), so that the shifts, masks,
// compares and branches on it go to the scalar unit - a vector instruction holds its SIMD for its issue slot however few lanes
// are on, and the wavefronts of a CU were queueing for the vector issue.
// Three flavours of the decoder (template flag SC).  Measured on 8 240 blocks of 64 KB (profiles/r05_bamdev.txt): all vector
// 23.6 ms, all scalar 17.9 ms, scalar control over vector data 16.0 ms - the product's; -DVBD_FLAVOUR=0 / 1 (developer builds)
// the others.
// SC = 1: everything lane 0 reads is declared uniform (data and control on the scalar unit); SC = 2: only what a branch looks at
// (VBD_CTL) - the bit buffer, the table entries, the cursors stay in vector registers, the decisions are scalar branches; SC = 0:
// nothing (vector code under exec masks).
#define VBD_UNI(x) (SC == 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(x)) : (uint32_t)(x))
#define VBD_CTL(x) (SC != 0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(x)) : (uint32_t)(x))
#else
#define VBD_UNI(x) ((uint32_t)(x))
#define VBD_CTL(x) ((uint32_t)(x))
#define VBD_LANE_SLOTS 64
#define VBD_DEV static inline
#define VBD_SYNC() ((void)0)
#define VBD_WAVE_FOR(i, a, b) for (int i = (a); i < (b); ++i)
#define VBD_EACH_LANE(l) for (int l = 0; l < 64; ++l)
#define VBD_LANE0 (true)
#endif

#if (defined(VBD_TIMING) || defined(VBD_EXP_NOSTORE) || defined(VBD_EXP_NOMATCH) || defined(VBD_FLAVOUR) || defined(VBD_LLB) || defined(VBD_MIN_WAVES)) && \
    !defined(VAPOR_DEV_BUILD) && !defined(VBD_EMU)
#error "developer switch without -DVAPOR_DEV_BUILD: a product library cannot be an experimental one"
#endif
// -DVBD_TIMING (developer builds): lane 0 adds up the shader clock it spends per phase; the kernel stores the sums per block
#if defined(VBD_TIMING) && !defined(VBD_EMU)
#define VBD_T0() const long long vbd_t0_ = clock64()
#define VBD_T1(k) do { if (lane == 0) L.tm[k] += clock64() - vbd_t0_; } while (0)
#define VBD_COUNT(k) (++L.cn[k])
#else
#define VBD_T0() ((void)0)
#define VBD_T1(k) ((void)0)
#define VBD_COUNT(k) ((void)0)
#endif

namespace vapor_bamdev {

#ifndef VBD_LLB
#define VBD_LLB 9
#endif
constexpr int LLB = VBD_LLB, DB = 8, PREB = 7;       // first-level bits of the literal / length, distance and code-length tables
// first + second level entries a complete code can need - the bounds zlib's `enough` finds: 286 symbols, longest code 15, 9 bits
// first: 852 (10: 1334); 30 symbols, 8 bits: 402.  (A block is 7.4 KB of LDS with 9 bits: twenty blocks in flight a CU.)
constexpr int LL_CAP = LLB == 9 ? 852 : 1334, D_CAP = 402;
static_assert(LLB == 9 || LLB == 10, "table bound known for 9 and 10 bits");
constexpr uint32_t E_LIT = 1u << 13, E_EOB = 1u << 14, E_SUB = 1u << 15;
constexpr uint32_t E_LIT2 = 1u << 5;                 // first level of the literal / length table: TWO literals (bits 16-23, 24-31), bits 0-4 their lengths' sum
constexpr uint32_t Q_FILL = 1u << 31;                // match queue, distance word: the match repeats ONE known byte (bits 16-23): no load
constexpr int IN_CAP = 1280;                         // bytes of the compressed stream kept in LDS (twice the longest block header and a bit)
constexpr int Q_CAP = 64;                            // matches decoded before the wavefront copies them
constexpr int U_MAX = 65536;                         // BGZF: at most 64 KB of data a block

// status of a block
constexpr int BLK_OK = 0, BLK_BAD_STREAM = 1, BLK_TABLES = 2, BLK_CRC = 3, BLK_STALLED = 4;

struct BgzfBlk {          // 24 B
    uint32_t c_off;       // the block's DEFLATE payload in the batch's compressed bytes
    uint32_t c_len;
    uint32_t u_off;       // where its data goes in the arena
    uint32_t u_len;       // ISIZE
    uint32_t crc;         // CRC-32 of the data (the block's trailer)
    uint32_t pad;
};

struct InflateState {     // the decoder between batches (LDS; lane 0 works on a copy in registers)
    uint64_t buf;         // bit buffer
    int32_t n;            // valid bits in it (negative: the stream ran out)
    int32_t ip;           // next byte of `in` to load into the bit buffer
    int32_t in_have;      // bytes of `in` that are stream bytes
    uint32_t g_next;      // next byte of the payload (offset in it) to bring into `in`
    uint32_t op;          // output position
    int32_t phase;        // 0 block header next, 1 inside a Huffman block, 2 stored bytes to copy, 3 done, 4 error
    int32_t last;         // the block being decoded is the stream's last
    int32_t nq;           // matches queued in this batch
    uint32_t st_src, st_len;   // phase 2: stored bytes payload[st_src .. + st_len) go to op
    int32_t err;
    uint32_t known;       // the table entry of the literal that is the last output byte, 0 when the last byte came from a match
    uint32_t progress;    // bits consumed + bytes produced, for the no-progress check
};

struct InflateLds {
    uint32_t ll[LL_CAP + 2];
    union {
        uint32_t ds[D_CAP + 2];
        uint32_t pre[1 << PREB];    // (the code-length code is done with before the distance table is built)
    };
#if defined(VBD_TIMING) && !defined(VBD_EMU)
    long long tm[8];                // 0 top-up, 1 decode (tables included), 2 tables, 3 matches, 4 CRC
    long long cn[8];                // 0 fast literal steps, 1 symbols of the careful loop, 2 matches, 3 batches, 4 general symbols in the fast loop, 5 fills
#endif
    union {
        uint32_t q[2 * Q_CAP];      // per match: position | length << 16, distance (| Q_FILL)
        uint8_t sub_bits[1 << LLB]; // (table building only: a batch that reaches a block header with matches queued ends there)
    };
    union {
        uint32_t crc_part[64];      // (the end of a block)
        struct {                    // build_table's small arrays (dynamically indexed: registers would spill to scratch)
            int32_t count[16];
            uint32_t next[16], nx[16];
        };
    };
    InflateState st;
    uint32_t in_w[(IN_CAP + 16) / 4];   // the LDS copy of the stream (bytes; zeros behind in_have)
    uint8_t lens[288 + 32 + 16];
    uint8_t pl[20];
};

// ---------------------------------------------------------------------------------------------------------------------------
// Huffman tables (the entry format of vapor_inflate.h without its double literals: bits 0-4 code length left to consume, 8-12
// extra bits, 13 literal, 14 end of block, 15 second-level pointer, 16-31 payload)
// ---------------------------------------------------------------------------------------------------------------------------
VBD_DEV uint32_t bit_reverse(uint32_t v, int n)
{
#ifndef VBD_EMU
    return __brev(v) >> (32 - n);
#else
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
#endif
}

// length / distance bases and extra bits, packed (base << 16 | extra << 8) so that a symbol's entry is one table read
#ifndef VBD_EMU
__device__
#endif
static const uint32_t LEN_ENTRY[29] = {
    3u << 16, 4u << 16, 5u << 16, 6u << 16, 7u << 16, 8u << 16, 9u << 16, 10u << 16,
    (11u << 16) | (1u << 8), (13u << 16) | (1u << 8), (15u << 16) | (1u << 8), (17u << 16) | (1u << 8),
    (19u << 16) | (2u << 8), (23u << 16) | (2u << 8), (27u << 16) | (2u << 8), (31u << 16) | (2u << 8),
    (35u << 16) | (3u << 8), (43u << 16) | (3u << 8), (51u << 16) | (3u << 8), (59u << 16) | (3u << 8),
    (67u << 16) | (4u << 8), (83u << 16) | (4u << 8), (99u << 16) | (4u << 8), (115u << 16) | (4u << 8),
    (131u << 16) | (5u << 8), (163u << 16) | (5u << 8), (195u << 16) | (5u << 8), (227u << 16) | (5u << 8), 258u << 16};
#ifndef VBD_EMU
__device__
#endif
static const uint32_t DIST_ENTRY[30] = {
    1u << 16, 2u << 16, 3u << 16, 4u << 16, (5u << 16) | (1u << 8), (7u << 16) | (1u << 8), (9u << 16) | (2u << 8), (13u << 16) | (2u << 8),
    (17u << 16) | (3u << 8), (25u << 16) | (3u << 8), (33u << 16) | (4u << 8), (49u << 16) | (4u << 8), (65u << 16) | (5u << 8),
    (97u << 16) | (5u << 8), (129u << 16) | (6u << 8), (193u << 16) | (6u << 8), (257u << 16) | (7u << 8), (385u << 16) | (7u << 8),
    (513u << 16) | (8u << 8), (769u << 16) | (8u << 8), (1025u << 16) | (9u << 8), (1537u << 16) | (9u << 8), (2049u << 16) | (10u << 8),
    (3073u << 16) | (10u << 8), (4097u << 16) | (11u << 8), (6145u << 16) | (11u << 8), (8193u << 16) | (12u << 8),
    (12289u << 16) | (12u << 8), (16385u << 16) | (13u << 8), (24577u << 16) | (13u << 8)};

// kind: 0 code lengths, 1 literals / lengths, 2 distances.  0xFFFFFFFF: a symbol that must not appear in data.
VBD_DEV uint32_t symbol_entry(int kind, int sym)
{
    if (kind == 0) return (uint32_t)sym << 16;
    if (kind == 1) {
        if (sym < 256) return E_LIT | ((uint32_t)sym << 16);
        if (sym == 256) return E_EOB;
        if (sym > 285) return 0xFFFFFFFFu;
        return LEN_ENTRY[sym - 257];
    }
    if (sym > 29) return 0xFFFFFFFFu;
    return DIST_ENTRY[sym];
}

// Canonical code of n symbols with lengths len[] into a two-level table (vapor_inflate.h build_table, same acceptance rules:
// over-subscribed and - but for zlib's single one-bit code - incomplete codes are refused).  Serial: lane 0 runs it.
// Returns 0, BLK_BAD_STREAM or BLK_TABLES (the code needs more second-level entries than `cap` holds).
VBD_DEV int build_table(int kind, const uint8_t* len, int n, uint32_t* tab, int tbits, int cap, InflateLds& L)
{
    int32_t* count = L.count;
    uint32_t* next = L.next;
    uint32_t* nx = L.nx;
    uint8_t* sub_bits = L.sub_bits;
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) ++count[len[s] & 15];
    count[0] = 0;
    uint32_t code = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = left * 2 - count[l];
        if (left < 0) return BLK_BAD_STREAM;
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const int first = 1 << tbits;
    for (int i = 0; i < first; ++i) tab[i] = 0;
    int longest = 15;
    while (longest > 0 && !count[longest]) --longest;
    if (longest == 0) return 0;
    if (left > 0 && (kind == 0 || longest != 1)) return BLK_BAD_STREAM;
    if (longest > tbits) {
        for (int i = 0; i < first; ++i) sub_bits[i] = 0;
        for (int l = 0; l < 16; ++l) nx[l] = next[l];
        for (int s = 0; s < n; ++s) {
            const int l = len[s] & 15;
            if (l <= tbits) { if (l) ++nx[l]; continue; }
            const uint32_t rev = bit_reverse(nx[l]++, l);
            const uint32_t lo = rev & (uint32_t)(first - 1);
            if (l - tbits > sub_bits[lo]) sub_bits[lo] = (uint8_t)(l - tbits);
        }
    }
    int used = first;
    for (int s = 0; s < n; ++s) {
        const int l = len[s] & 15;
        if (!l) continue;
        const uint32_t rev = bit_reverse(next[l]++, l);
        const uint32_t e = symbol_entry(kind, s);
        const bool banned = e == 0xFFFFFFFFu;
        if (l <= tbits) {
            const uint32_t v = banned ? 0u : (e | (uint32_t)l);
            for (uint32_t i = rev; i < (uint32_t)first; i += 1u << l) tab[i] = v;
        } else {
            const uint32_t lo = rev & (uint32_t)(first - 1);
            const int sb = sub_bits[lo];
            uint32_t head = tab[lo];
            if (!(head & E_SUB)) {
                if (used + (1 << sb) > cap) return BLK_TABLES;
                head = E_SUB | ((uint32_t)used << 16) | ((uint32_t)sb << 8) | (uint32_t)tbits;
                tab[lo] = head;
                for (int i = 0; i < (1 << sb); ++i) tab[used + i] = 0;
                used += 1 << sb;
            }
            const uint32_t base = head >> 16;
            const uint32_t v = banned ? 0u : (e | (uint32_t)(l - tbits));
            for (uint32_t i = rev >> tbits; i < (1u << sb); i += 1u << (l - tbits)) tab[base + i] = v;
        }
    }
    return 0;
}

// First-level literal entries whose following bits spell another literal inside the table's width become double entries (the
// decoder is bound by its lookup -> shift -> lookup chain: two symbols a lookup where the codes are short - BAM's packed bases
// are sixteen byte values of four or five bits).  In place, from the top: entry i reads entry i >> l1 < i, which is still single.
VBD_DEV void add_double_literals(uint32_t* tab, int tbits)
{
    for (int i = (1 << tbits) - 1; i >= 1; --i) {
        const uint32_t e = tab[i];
        if ((e & (E_LIT | E_SUB)) != E_LIT) continue;
        const int l1 = (int)(e & 31u);
        if (l1 >= tbits) continue;
        const uint32_t e2 = tab[(uint32_t)i >> l1];
        if ((e2 & (E_LIT | E_SUB | E_LIT2)) != E_LIT) continue;
        const int l2 = (int)(e2 & 31u);
        if (l1 + l2 > tbits) continue;
        tab[i] = E_LIT | E_LIT2 | (uint32_t)(l1 + l2) | (e & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8);
    }
}

// the bit reader over the LDS copy of the stream (zeros behind its end; n < 0 says bits were taken that are not there)
struct Bits {
    uint64_t buf;
    int n, ip, in_have;
    const uint32_t* in_w;
};
// Whole bytes up to 56..63 valid bits, out of three aligned words of the LDS copy.  The bits above n are not counted but
// they are the stream's own next bits (zeros behind its end): the next refill puts the same bits there again.
template <int SC>
VBD_DEV void refill(Bits& b)
{
    if (b.n < 0) return;
    int adv = (63 - b.n) >> 3;
    const int left = b.in_have - b.ip;
    if (adv > left) adv = left;
    if (adv <= 0) return;
    const uint32_t* w = b.in_w + (b.ip >> 2);
    const int sh = (b.ip & 3) * 8;
    const uint64_t lo = (uint64_t)VBD_UNI(w[0]) | ((uint64_t)VBD_UNI(w[1]) << 32);
    const uint64_t v = sh ? (lo >> sh) | ((uint64_t)VBD_UNI(w[2]) << (64 - sh)) : lo;
    b.buf |= v << b.n;
    b.ip += adv;
    b.n += adv * 8;
}
VBD_DEV uint32_t peek(const Bits& b, int k) { return (uint32_t)(b.buf & ((1ull << k) - 1ull)); }
VBD_DEV void drop(Bits& b, int k) { b.buf >>= k; b.n -= k; }
VBD_DEV uint32_t take(Bits& b, int k) { const uint32_t v = peek(b, k); drop(b, k); return v; }

VBD_DEV int read_fixed(InflateLds& L)
{
    for (int s = 0; s < 144; ++s) L.lens[s] = 8;
    for (int s = 144; s < 256; ++s) L.lens[s] = 9;
    for (int s = 256; s < 280; ++s) L.lens[s] = 7;
    for (int s = 280; s < 288; ++s) L.lens[s] = 8;
    int rc = build_table(1, L.lens, 288, L.ll, LLB, LL_CAP, L);
    if (rc) return rc;
    add_double_literals(L.ll, LLB);
    for (int s = 0; s < 32; ++s) L.lens[s] = 5;
    return build_table(2, L.lens, 32, L.ds, DB, D_CAP, L);
}

// the header of a dynamic block (RFC 1951 3.2.7); the caller has made sure `in` holds the whole of it (it is at most
// 14 + 19 * 3 + 316 * 14 bits = 562 bytes) or the end of the stream
template <int SC>
VBD_DEV int read_dynamic(Bits& b, InflateLds& L)
{
    const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    refill<SC>(b);
    const int hlit = (int)VBD_CTL(take(b, 5)) + 257, hdist = (int)VBD_CTL(take(b, 5)) + 1, hclen = (int)VBD_CTL(take(b, 4)) + 4;
    if (hlit > 286 || hdist > 30) return BLK_BAD_STREAM;
    uint8_t* pl = L.pl;
    for (int i = 0; i < 19; ++i) pl[i] = 0;
    for (int i = 0; i < hclen; ++i) {
        if ((int)VBD_CTL(b.n) < 3) refill<SC>(b);
        pl[ORDER[i]] = (uint8_t)take(b, 3);
    }
    if (b.n < 0) return BLK_BAD_STREAM;
    int rc = build_table(0, pl, 19, L.pre, PREB, 1 << PREB, L);
    if (rc) return rc;
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        refill<SC>(b);
        const uint32_t e = VBD_CTL(L.pre[peek(b, PREB)]);
        if (!e) return BLK_BAD_STREAM;
        drop(b, (int)(e & 31u));
        const int sym = (int)(e >> 16);
        if (sym < 16) { L.lens[i++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) {
            if (i == 0) return BLK_BAD_STREAM;
            val = L.lens[i - 1];
            rep = 3 + (int)VBD_CTL(take(b, 2));
        } else if (sym == 17) {
            rep = 3 + (int)VBD_CTL(take(b, 3));
        } else {
            rep = 11 + (int)VBD_CTL(take(b, 7));
        }
        if (i + rep > total) return BLK_BAD_STREAM;
        for (int t = 0; t < rep; ++t) L.lens[i + t] = (uint8_t)val;
        i += rep;
        if (b.n < 0) return BLK_BAD_STREAM;
    }
    if (L.lens[256] == 0) return BLK_BAD_STREAM;
    rc = build_table(1, L.lens, hlit, L.ll, LLB, LL_CAP, L);
    if (rc) return rc;
    add_double_literals(L.ll, LLB);
    // (the distance lengths lie behind the literal / length ones; build_table reads them in place)
    return build_table(2, L.lens + hlit, hdist, L.ds, DB, D_CAP, L);
}

// the refill of the fast loop: the caller has made sure sixteen stream bytes lie behind ip, so nothing is clamped
template <int SC>
VBD_DEV void refill_fast(Bits& b)
{
    const uint32_t* w = b.in_w + (b.ip >> 2);
    const int sh = (b.ip & 3) * 8;
    const uint64_t lo = (uint64_t)VBD_UNI(w[0]) | ((uint64_t)VBD_UNI(w[1]) << 32);
    const uint64_t v = (lo >> sh) | (((uint64_t)VBD_UNI(w[2]) << 1) << (63 - sh));
    b.buf |= v << b.n;
    const int adv = (63 - b.n) >> 3;
    b.ip += adv;
    b.n += adv * 8;
}

// One batch of lane 0's work: block headers and symbols until Q_CAP matches are queued, the LDS copy of the stream runs low,
// stored bytes wait to be copied, the stream ends or an error is found.  Literals go straight to `out`; matches are queued
// (a match of distance 1 behind a byte lane 0 knows - a run - is queued as that byte: its copy needs no load).
template <int SC>
VBD_DEV void decode_batch(InflateLds& L, uint8_t* out, uint32_t u_len, uint32_t c_len)
{
    InflateState& S = L.st;
    VBD_COUNT(3);
    Bits b;
    b.buf = (uint64_t)VBD_UNI((uint32_t)S.buf) | ((uint64_t)VBD_UNI((uint32_t)(S.buf >> 32)) << 32);
    b.n = (int)VBD_UNI(S.n); b.ip = (int)VBD_UNI(S.ip); b.in_have = (int)VBD_CTL(S.in_have); b.in_w = L.in_w;
    uint32_t op = VBD_UNI(S.op), known = VBD_UNI(S.known);
    int phase = (int)VBD_CTL(S.phase), last = (int)VBD_CTL(S.last), nq = 0, err = 0;
    const bool all_loaded = VBD_CTL(S.g_next) >= c_len;
    constexpr uint32_t LL_MASK = (1u << LLB) - 1u, D_MASK = (1u << DB) - 1u;
#define VBD_IS_LIT1(e) (((e) & (E_LIT | E_SUB)) == E_LIT)
    // a match (len, dist) at op into the queue
#define VBD_QUEUE(len, dist)                                                                                     \
    do {                                                                                                         \
        uint32_t dw_ = (dist);                                                                                   \
        VBD_COUNT(2);                                                                                            \
        if (VBD_CTL((uint32_t)(dw_ == 1u && known != 0u))) {                                                     \
            VBD_COUNT(5);                                                                                        \
            const uint32_t byte_ = (known & E_LIT2) ? known >> 24 : (known >> 16) & 0xFFu;                       \
            dw_ = Q_FILL | (byte_ << 16) | 1u;                                                                   \
            known = E_LIT | (byte_ << 16);                                                                       \
        } else {                                                                                                 \
            known = 0;                                                                                           \
        }                                                                                                        \
        L.q[2 * nq] = op | ((len) << 16);                                                                        \
        L.q[2 * nq + 1] = dw_;                                                                                   \
        op += (len);                                                                                             \
        ++nq;                                                                                                    \
    } while (0)
    for (;;) {
        if (phase == 0) {
            // (building tables uses the match queue's bytes: the queued matches are copied first)
            if (nq) break;
            // a block header: all of it must be in `in` (600 bytes cover the longest), or the stream's end
            if (!all_loaded && b.in_have - (int)VBD_CTL(b.ip) < 600) break;
            refill<SC>(b);
            last = (int)VBD_CTL(take(b, 1));
            const uint32_t type = VBD_CTL(take(b, 2));
            if ((int)VBD_CTL(b.n) < 0) { err = BLK_BAD_STREAM; break; }
            if (type == 0) {
                // stored: to the byte boundary; LEN and NLEN; the bytes themselves are copied by the wavefront from the payload
                drop(b, b.n & 7);
                if ((int)VBD_CTL(b.n) < 32) refill<SC>(b);
                if ((int)VBD_CTL(b.n) < 32) { err = BLK_BAD_STREAM; break; }
                const uint32_t len = VBD_CTL(take(b, 16)), nlen = VBD_CTL(take(b, 16));
                if ((len ^ nlen) != 0xFFFFu) { err = BLK_BAD_STREAM; break; }
                // payload offset of the first stored byte: what `in` holds from ip on lies behind the bytes still in the bit buffer
                const uint32_t src = VBD_CTL(S.g_next - (uint32_t)(b.in_have - b.ip) - (uint32_t)(b.n >> 3));
                if (src > c_len || len > c_len - src || len > u_len - VBD_CTL(op)) { err = BLK_BAD_STREAM; break; }
                S.st_src = src; S.st_len = len;
                if (len) known = 0;
                phase = 2;
                break;
            }
            if (type == 3) { err = BLK_BAD_STREAM; break; }
#if defined(VBD_TIMING) && !defined(VBD_EMU)
            const long long tt0 = clock64();
#endif
            err = type == 1 ? read_fixed(L) : read_dynamic<SC>(b, L);
#if defined(VBD_TIMING) && !defined(VBD_EMU)
            L.tm[2] += clock64() - tt0;
#endif
            // (the table builders branch on what they read from LDS: what comes back from them is said to be uniform again, or
            // the compiler keeps the whole decoder state in vector registers from here on)
            err = (int)VBD_CTL(err);
            b.buf = (uint64_t)VBD_UNI((uint32_t)b.buf) | ((uint64_t)VBD_UNI((uint32_t)(b.buf >> 32)) << 32);
            b.n = (int)VBD_UNI(b.n); b.ip = (int)VBD_UNI(b.ip);
            if (err) break;
            phase = 1;
        }
        // ---- the fast loop: sixteen stream bytes behind ip (two refills that clamp nothing), eight bytes of room in the output
        // (four double literals stored without a test).  No bit count can go negative here: every symbol starts with 48 bits.
        nq = (int)VBD_CTL(nq);
        if ((int)VBD_CTL(b.n) >= 0 && (int)VBD_CTL(b.ip) + 16 <= b.in_have && VBD_CTL(op) + 8 <= u_len) {
            refill_fast<SC>(b);
            uint32_t e = VBD_UNI(L.ll[(uint32_t)b.buf & LL_MASK]);
            bool leave = false;
            for (;;) {
                // (e is the first-level entry at the current position: a refill leaves the bits that are there in place)
                // (said to be uniform once more - it costs nothing where the compiler knows, and keeps the loop on the scalar
                // unit where a branch on something a table builder read made it doubt)
                b.buf = (uint64_t)VBD_UNI((uint32_t)b.buf) | ((uint64_t)VBD_UNI((uint32_t)(b.buf >> 32)) << 32);
                b.n = (int)VBD_UNI(b.n); b.ip = (int)VBD_UNI(b.ip); op = VBD_UNI(op); e = VBD_UNI(e); nq = (int)VBD_UNI(nq); known = VBD_UNI(known);
                if ((int)VBD_CTL(b.ip) + 16 > b.in_have || VBD_CTL(op) + 8 > u_len) break;
                refill_fast<SC>(b);
                if (VBD_IS_LIT1(VBD_CTL(e))) {
                    // up to four table hits out of one refill (40 of its 56 bits at most), each one literal or two: both bytes are
                    // stored either way, the cursor moves by one or two
#ifdef VBD_EXP_NOSTORE
#define VBD_ST(i, v) ((void)0)
#else
#define VBD_ST(i, v) (out[i] = (v))
#endif
#define VBD_EMIT()                                                                  \
    do {                                                                            \
        VBD_ST(op, (uint8_t)(e >> 16));                                             \
        VBD_ST(op + 1, (uint8_t)(e >> 24));                                         \
        op += 1u + ((e >> 5) & 1u);                                                 \
        known = e;                                                                  \
        VBD_COUNT(0);                                                               \
        b.buf >>= (e & 31u);                                                        \
        b.n -= (int)(e & 31u);                                                      \
        e = VBD_UNI(L.ll[(uint32_t)b.buf & LL_MASK]);                                        \
    } while (0)
                    VBD_EMIT();
                    if (VBD_IS_LIT1(VBD_CTL(e))) {
                        VBD_EMIT();
                        if (VBD_IS_LIT1(VBD_CTL(e))) {
                            VBD_EMIT();
                            if (VBD_IS_LIT1(VBD_CTL(e))) VBD_EMIT();
                        }
                    }
#undef VBD_EMIT
                    if (VBD_IS_LIT1(VBD_CTL(e))) continue;
                    if ((int)VBD_CTL(b.ip) + 8 > b.in_have || VBD_CTL(op) + 8 > u_len) break;
                    // (the symbol that follows needs 48 bits at most; one literal step out of a refill of 57 and more leaves them)
                    if ((int)VBD_CTL(b.n) < 48) refill_fast<SC>(b);
                }
                VBD_COUNT(4);
#if defined(VBD_TIMING) && !defined(VBD_EMU)
                const long long tg0 = clock64();
#endif
                uint32_t ec = VBD_CTL(e);                              // (what the branches look at)
                if (ec & E_SUB) {
                    drop(b, LLB);
                    e = VBD_UNI(L.ll[(e >> 16) + peek(b, (int)((e >> 8) & 31u))]);
                    ec = VBD_CTL(e);
                }
                if (!ec) { err = BLK_BAD_STREAM; leave = true; break; }
                drop(b, (int)(e & 31u));
                if (ec & E_LIT) {
                    out[op++] = (uint8_t)(e >> 16);
                    known = e;
                } else if (ec & E_EOB) {
                    phase = last ? 3 : 0;
                    leave = true;
                    break;
                } else {
                    const int xl = (int)((e >> 8) & 31u);
                    const uint32_t len = (e >> 16) + peek(b, xl);
                    drop(b, xl);
                    uint32_t f = VBD_UNI(L.ds[(uint32_t)b.buf & D_MASK]);
                    uint32_t fc = VBD_CTL(f);
                    if (fc & E_SUB) {
                        drop(b, DB);
                        f = VBD_UNI(L.ds[(f >> 16) + peek(b, (int)((f >> 8) & 31u))]);
                        fc = VBD_CTL(f);
                    }
                    if (!fc) { err = BLK_BAD_STREAM; leave = true; break; }
                    drop(b, (int)(f & 31u));
                    const int xd = (int)((f >> 8) & 31u);
                    const uint32_t dist = (f >> 16) + peek(b, xd);
                    drop(b, xd);
                    if (VBD_CTL((uint32_t)(dist > op || len > u_len - op))) { err = BLK_BAD_STREAM; leave = true; break; }
                    VBD_QUEUE(len, dist);
                    if (nq == Q_CAP) { leave = true; break; }
                }
                e = VBD_UNI(L.ll[(uint32_t)b.buf & LL_MASK]);
#if defined(VBD_TIMING) && !defined(VBD_EMU)
                L.tm[5] += clock64() - tg0;
#endif
            }
            if (err || nq == Q_CAP || phase == 3) break;
            if (leave) continue;                  // (the end of a block: the next header)
        }
        // ---- one symbol with every test (the ends of the input and of the output): 48 bits cover the longest (15 + 5 + 15 + 13)
        VBD_COUNT(1);
        if ((int)VBD_CTL(b.n) < 48) {
            if (!all_loaded && (int)VBD_CTL(b.ip) + 8 > b.in_have) break;
            refill<SC>(b);
        }
        uint32_t e = VBD_CTL(L.ll[(uint32_t)b.buf & LL_MASK]);      // (this loop runs a dozen times a block: all of it scalar where any is)
        if (e & E_SUB) {
            drop(b, LLB);
            e = VBD_CTL(L.ll[(e >> 16) + peek(b, (int)((e >> 8) & 31u))]);
        }
        if (!e) { err = BLK_BAD_STREAM; break; }
        drop(b, (int)(e & 31u));
        if ((int)VBD_CTL(b.n) < 0) { err = BLK_BAD_STREAM; break; }
        if (e & E_LIT) {
            if (VBD_CTL(op) >= u_len) { err = BLK_BAD_STREAM; break; }
            out[op++] = (uint8_t)(e >> 16);
            if (e & E_LIT2) {
                if (VBD_CTL(op) >= u_len) { err = BLK_BAD_STREAM; break; }
                out[op++] = (uint8_t)(e >> 24);
            }
            known = e;
            continue;
        }
        if (e & E_EOB) {
            if (last) { phase = 3; break; }
            phase = 0;
            continue;
        }
        const int xl = (int)((e >> 8) & 31u);
        const uint32_t len = (e >> 16) + peek(b, xl);
        drop(b, xl);
        uint32_t f = VBD_CTL(L.ds[(uint32_t)b.buf & D_MASK]);
        if (f & E_SUB) {
            drop(b, DB);
            f = VBD_CTL(L.ds[(f >> 16) + peek(b, (int)((f >> 8) & 31u))]);
        }
        if (!f) { err = BLK_BAD_STREAM; break; }
        drop(b, (int)(f & 31u));
        const int xd = (int)((f >> 8) & 31u);
        const uint32_t dist = (f >> 16) + peek(b, xd);
        drop(b, xd);
        if (VBD_CTL((uint32_t)(b.n < 0 || dist > op || len > u_len - op))) { err = BLK_BAD_STREAM; break; }
        VBD_QUEUE(len, dist);
        if (nq == Q_CAP) break;
    }
#undef VBD_QUEUE
#undef VBD_IS_LIT1
    S.buf = b.buf; S.n = b.n; S.ip = b.ip;
    S.op = op; S.phase = err ? 4 : phase; S.last = last; S.nq = nq; S.known = known;
    if (err) S.err = err;
}

// x^(8 * bytes) mod P in the reflected representation zlib's crc32_combine uses (bit 31 = x^0)
VBD_DEV uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & m) p ^= b;
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

// ---------------------------------------------------------------------------------------------------------------------------
// One BGZF block by one wavefront.  `out` is where the block's data goes in the arena (HBM): lane 0 stores the literals as it
// decodes them, the wavefront copies the matches from what it has written before, the CRC is read back from there (L2).  Only
// the tables, the LDS copy of the stream and the match queue live in LDS - 7.4 KB a block, twenty blocks in flight per CU (five
// wavefronts a SIMD: 16 -> 20 blocks a CU took 16.0 -> 12.4 ms on 8 240 blocks), which is what hides the latency of lane 0's
// lookup -> shift -> lookup chain.  crc_pow[l] = x^(8 * 1024 * (63 - l)) mod P.
// ---------------------------------------------------------------------------------------------------------------------------
// the byte table of the CRC-32 (one a workgroup; the CRC is a hundredth of a block's time: slicing tables would cost LDS that
// keeps a fourth workgroup off the CU)
struct CrcTables { uint32_t t[1][256]; };

template <int SC>
VBD_DEV int inflate_block_wave(InflateLds& L, const CrcTables& T, uint8_t* out, const uint8_t* payload, uint32_t c_len, uint32_t u_len,
                               uint32_t want_crc, const uint32_t* crc_pow, uint32_t lane)
{
    (void)lane;
    if (VBD_LANE0) {
        InflateState& S = L.st;
        S.buf = 0; S.n = 0; S.ip = 0; S.in_have = 0; S.g_next = 0; S.op = 0; S.phase = 0; S.last = 0; S.nq = 0;
        S.st_src = 0; S.st_len = 0; S.err = 0; S.known = 0; S.progress = 0xFFFFFFFFu;
    }
    VBD_SYNC();
    // every batch consumes input or produces output or ends; a stream of c_len bytes and u_len bytes of output cannot take more
    // batches than this (the no-progress check below ends a decoder that stops moving long before)
    const uint32_t max_batches = c_len + u_len + 64u;
    for (uint32_t batch = 0; batch < max_batches; ++batch) {
        // ---- top up the LDS copy of the stream: the unread bytes move to the front, new ones behind them
        {
            VBD_T0();
            const int ip = L.st.ip, have = L.st.in_have;
            const uint32_t g_next = L.st.g_next;
            if (g_next < c_len && (have == 0 || ip >= IN_CAP / 2)) {
                const int rest = have - ip;                       // (<= ip: source and destination do not overlap)
                uint8_t* in = reinterpret_cast<uint8_t*>(L.in_w);
                VBD_WAVE_FOR(i, 0, rest) in[i] = in[ip + i];
                uint32_t add = (uint32_t)(IN_CAP - rest);
                if (add > c_len - g_next) add = c_len - g_next;
                VBD_WAVE_FOR(i, 0, (int)add) in[rest + i] = payload[g_next + (uint32_t)i];
                VBD_WAVE_FOR(i, rest + (int)add, rest + (int)add + 16) if (i < IN_CAP + 16) in[i] = 0;
                VBD_SYNC();
                if (VBD_LANE0) { L.st.ip = 0; L.st.in_have = rest + (int)add; L.st.g_next = g_next + add; }
                VBD_SYNC();
            }
            VBD_T1(0);
        }
        {
            VBD_T0();
            if (VBD_LANE0) decode_batch<SC>(L, out, u_len, c_len);
            VBD_T1(1);
        }
        VBD_SYNC();
        VBD_T0();
        // ---- the batch's matches, in order (a match may read what an earlier one of the batch wrote)
#ifdef VBD_EXP_NOMATCH
        const int nq = 0;
#else
        const int nq = L.st.nq;
#endif
        for (int m = 0; m < nq;) {
            const uint32_t w = L.q[2 * m], dw = L.q[2 * m + 1];
            const uint32_t pos = w & 0xFFFFu, len = w >> 16;
            if (dw & Q_FILL) {
                // a run of one byte that lane 0 knew: stores alone
                const uint8_t byte = (uint8_t)(dw >> 16);
                VBD_WAVE_FOR(j, 0, (int)len) out[pos + (uint32_t)j] = byte;
                ++m;
                continue;
            }
            const uint32_t dist = dw;
            // up to four short matches that read nothing the others write (their sources end before the first one's destination):
            // their loads go out together, one wait instead of four
            if (len <= 64u && dist >= len) {
                uint32_t p_[4] = {pos, 0, 0, 0}, l_[4] = {len, 0, 0, 0}, d_[4] = {dist, 0, 0, 0};
                int g = 1;
                for (; g < 4 && m + g < nq; ++g) {
                    const uint32_t w2 = L.q[2 * (m + g)], d2 = L.q[2 * (m + g) + 1];
                    const uint32_t p2 = w2 & 0xFFFFu, l2 = w2 >> 16;
                    if ((d2 & Q_FILL) || l2 > 64u || d2 < l2 || p2 - d2 + l2 > pos) break;
                    p_[g] = p2; l_[g] = l2; d_[g] = d2;
                }
                uint8_t v_[4][VBD_LANE_SLOTS];
                VBD_EACH_LANE(l) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if ((uint32_t)l < l_[t]) v_[t][l % VBD_LANE_SLOTS] = out[p_[t] - d_[t] + (uint32_t)l];
                }
                VBD_EACH_LANE(l) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if ((uint32_t)l < l_[t]) out[p_[t] + (uint32_t)l] = v_[t][l % VBD_LANE_SLOTS];
                }
                VBD_SYNC();
                m += g;
                continue;
            }
            // byte j of the match is byte j mod dist of the dist bytes before it (RFC 1951 3.2.3: the copy may overlap itself)
            if (dist >= len) {
                VBD_WAVE_FOR(j, 0, (int)len) out[pos + (uint32_t)j] = out[pos - dist + (uint32_t)j];
            } else {
                VBD_WAVE_FOR(j, 0, (int)len) out[pos + (uint32_t)j] = out[pos - dist + (uint32_t)j % dist];
            }
            VBD_SYNC();
            ++m;
        }
        VBD_T1(3);
        const int phase = L.st.phase;
        if (phase == 2) {
            // stored bytes: payload -> image, then the stream goes on behind them (its LDS copy starts over)
            const uint32_t src = L.st.st_src, len = L.st.st_len, op = L.st.op;
            VBD_WAVE_FOR(i, 0, (int)len) out[op + (uint32_t)i] = payload[src + (uint32_t)i];
            VBD_SYNC();
            if (VBD_LANE0) {
                InflateState& S = L.st;
                S.op = op + len; S.buf = 0; S.n = 0; S.ip = 0; S.in_have = 0; S.g_next = src + len;
                S.phase = S.last ? 3 : 0;
            }
            VBD_SYNC();
        }
        if (L.st.phase >= 3) break;
        // no progress since the last batch: a decoder that waits for input that cannot come (damaged stream)
        const uint32_t prog = (L.st.g_next - (uint32_t)L.st.in_have + (uint32_t)L.st.ip) * 8u - (uint32_t)L.st.n + L.st.op * 16u + (uint32_t)L.st.phase;
        const uint32_t before = L.st.progress;
        VBD_SYNC();
        if (prog == before) { if (VBD_LANE0) { L.st.phase = 4; L.st.err = BLK_STALLED; } VBD_SYNC(); break; }
        if (VBD_LANE0) L.st.progress = prog;
        VBD_SYNC();
    }
    if (L.st.phase != 3) return L.st.err ? L.st.err : BLK_STALLED;
    if (L.st.op != u_len) return BLK_BAD_STREAM;
    VBD_T0();
    // ---- CRC-32: the data as the tail of 64 slices of 1 024 bytes (zeros in front change nothing in a register that starts at
    // zero; the register's start value 0xFFFFFFFF is the first four data bytes inverted), a slice a lane, then
    // crc = sum over lanes of slice_crc * x^(8 * bytes behind the slice) mod P
    if (u_len < 4) {
        if (VBD_LANE0) {
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < u_len; ++i) c = T.t[0][(c ^ out[i]) & 0xFFu] ^ (c >> 8);
            L.crc_part[0] = ~c;
        }
        VBD_SYNC();
        return L.crc_part[0] == want_crc ? BLK_OK : BLK_CRC;
    }
    VBD_EACH_LANE(l) {
        const int lead = U_MAX - (int)u_len;                  // zero bytes in front of the data
        int a = l * 1024 - lead, e = a + 1024;                 // the slice in data coordinates
        if (a < 0) a = 0;
        uint32_t c = 0;
        int i = a;
        for (; i < e && i < 4; ++i) c = T.t[0][(c ^ out[i] ^ 0xFFu) & 0xFFu] ^ (c >> 8);
        for (; i < e; ++i) c = T.t[0][(c ^ out[i]) & 0xFFu] ^ (c >> 8);
        L.crc_part[l] = e > 0 ? crc_mulmod(crc_pow[l], c) : 0u;
    }
    VBD_SYNC();
    if (VBD_LANE0) {
        uint32_t c = 0;
        for (int l = 0; l < 64; ++l) c ^= L.crc_part[l];
        L.crc_part[0] = ~c;
    }
    VBD_SYNC();
    VBD_T1(4);
    return L.crc_part[0] == want_crc ? BLK_OK : BLK_CRC;
}

#ifndef VBD_EMU
// ---------------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------------
// Four wavefronts a workgroup, a BGZF block each.
constexpr int INFLATE_WAVES = 4;
#ifndef VBD_MIN_WAVES
#define VBD_MIN_WAVES 5             // wavefronts per SIMD the register budget is cut for (LDS admits five workgroups of four a CU)
#endif
__global__ __launch_bounds__(64 * INFLATE_WAVES, VBD_MIN_WAVES) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const BgzfBlk* __restrict__ blks, int n_blks,
                                                                         uint8_t* arena, const uint32_t* __restrict__ crc_pow,
                                                                         int32_t* __restrict__ blk_status)
{
    __shared__ InflateLds L[INFLATE_WAVES];
    __shared__ CrcTables T;
    {
        const int i = (int)threadIdx.x;                      // (256 threads: an entry of each table a thread)
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        T.t[0][i] = c;
        __syncthreads();
    }
    const int wave = (int)(threadIdx.x >> 6);
    const int b = (int)blockIdx.x * INFLATE_WAVES + wave;
    if (b >= n_blks) return;
    const uint32_t lane = threadIdx.x & 63u;
    // (one block a wavefront: its descriptor is the same in every lane - said so, the addresses are scalar base + 32-bit offset)
    BgzfBlk k = blks[b];
    k.c_off = __builtin_amdgcn_readfirstlane(k.c_off); k.c_len = __builtin_amdgcn_readfirstlane(k.c_len);
    k.u_off = __builtin_amdgcn_readfirstlane(k.u_off); k.u_len = __builtin_amdgcn_readfirstlane(k.u_len);
    k.crc = __builtin_amdgcn_readfirstlane(k.crc);
#if defined(VBD_TIMING)
    if (lane == 0) for (int t = 0; t < 8; ++t) { L[wave].tm[t] = 0; L[wave].cn[t] = 0; }
    const long long t_all = clock64();
#endif
#ifndef VBD_FLAVOUR
#define VBD_FLAVOUR 2
#endif
    const int rc = inflate_block_wave<VBD_FLAVOUR>(L[wave], T, arena + k.u_off, comp + k.c_off, k.c_len, k.u_len, k.crc, crc_pow, lane);
    if (lane == 0) blk_status[b] = rc;
#if defined(VBD_TIMING)
    // (the sums go where the block's compressed bytes were: nothing reads those again)
    if (lane == 0 && k.c_len >= 128) {
        long long* d = reinterpret_cast<long long*>(const_cast<uint8_t*>(comp) + ((k.c_off + 7u) & ~7u));
        for (int t = 0; t < 5; ++t) d[t] = L[wave].tm[t];
        d[5] = clock64() - t_all;
        for (int t = 0; t < 6; ++t) d[6 + t] = L[wave].cn[t];
        d[12] = L[wave].tm[5];
    }
#endif
}

// ---- the records of a region ----------------------------------------------------------------------------------------------
struct BamSpan {          // one index chunk of a region: its records start in arena[u_begin, u_end), its data ends at u_limit
    uint32_t u_begin, u_end, u_limit;
    uint32_t blk_first, blk_n;   // its blocks in the block table (their status decides whether the bytes can be read)
    uint32_t pad;
};
struct BamRegion {        // `samtools view bam tid:start-end` + chop_pacbio_read_by_pos(start, end, flank)
    int64_t start, end, flank;
    int32_t tid, span_first, span_n, pad;
};
struct BamKept {          // a read the reference keeps: its packed bases at arena + sq_off, from base q0 on, miss_bp
    uint32_t sq_off;
    int32_t q0, miss, l_seq;
};
constexpr int KEPT_CAP = 256;      // kept reads a region's slot holds (minimize_pacbio_read_list keeps 20 of them)
// status of a region: 0, or why the host route must do it
constexpr int REG_OK = 0, REG_BEYOND = 1, REG_MALFORMED = 2, REG_NO_CIGAR = 3, REG_KEPT_FULL = 4, REG_BLOCK = 5, REG_NEG_Q0 = 6, REG_NO_SEQ = 7;

__device__ __forceinline__ uint32_t rd32u(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ long long wave_scan64(long long v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(v, (unsigned)d);
        if ((int)lane >= d) v += t;
    }
    return v;
}

// the CG:B,I array among a record's aux fields [p, end) (vapor_bam.cpp find_cg); returns its offset or 0, *count = its length
__device__ __forceinline__ uint32_t find_cg_dev(const uint8_t* arena, uint32_t p, uint32_t end, int32_t* count)
{
    // (every step moves p forward by three bytes at least: the loop ends)
    while (p + 3 <= end) {
        const uint8_t t0 = arena[p], t1 = arena[p + 1], ty = arena[p + 2];
        p += 3;
        int sz = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': { while (p < end && arena[p]) ++p; if (p >= end) return 0; ++p; continue; }
        case 'B': {
            if (p + 5 > end) return 0;
            const uint8_t sub = arena[p];
            const int32_t cnt = (int32_t)rd32u(arena + p + 1);
            p += 5;
            const int es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (cnt < 0 || (long long)cnt * es > (long long)(end - p)) return 0;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') { *count = cnt; return p; }
            p += (uint32_t)cnt * (uint32_t)es;
            continue;
        }
        default: return 0;
        }
        if ((uint32_t)sz > end - p) return 0;
        p += (uint32_t)sz;
    }
    return 0;
}

// One wavefront per region (vapor_bam.cpp bam_chop_impl is the statement this follows, line for line in its decisions).
__global__ __launch_bounds__(64) void bam_chop_kernel(const uint8_t* __restrict__ arena, const BamRegion* __restrict__ regs,
                                                     const BamSpan* __restrict__ spans, const int32_t* __restrict__ blk_status, int n_regs,
                                                     BamKept* __restrict__ kept, int32_t* __restrict__ n_kept, int32_t* __restrict__ reg_status)
{
    const int g = (int)blockIdx.x;
    if (g >= n_regs) return;
    const uint32_t lane = threadIdx.x;
    const BamRegion R = regs[g];
    const long long start = R.start, end = R.end;
    const long long beg = start - 1 > 0 ? start - 1 : 0, stop = end;
    int nk = 0, st = REG_OK;
    for (int s = 0; s < R.span_n && st == REG_OK; ++s) {
        const BamSpan SP = spans[R.span_first + s];
        int bad = 0;
        for (uint32_t i = lane; i < SP.blk_n; i += 64) bad |= blk_status[SP.blk_first + i] != 0;
        if (__any(bad)) { st = REG_BLOCK; break; }
        uint32_t pos_u = SP.u_begin;
        // (a record is 36 bytes at least: pos_u grows every turn and the loop ends at u_end)
        while (pos_u < SP.u_end) {
            if ((unsigned long long)pos_u + 36ull > SP.u_limit) { st = REG_BEYOND; break; }
            uint32_t w = 0;
            if (lane < 6) w = rd32u(arena + pos_u + 4u * lane);
            const int32_t bs = (int32_t)__shfl(w, 0), ref_id = (int32_t)__shfl(w, 1), pos = (int32_t)__shfl(w, 2);
            const uint32_t w3 = __shfl(w, 3), w4 = __shfl(w, 4);
            const int32_t l_seq = (int32_t)__shfl(w, 5);
            if (bs < 32 || bs > (1 << 29)) { st = REG_MALFORMED; break; }
            if ((unsigned long long)pos_u + 4ull + (unsigned long long)bs > SP.u_limit) { st = REG_BEYOND; break; }
            const uint32_t r = pos_u + 4u;
            pos_u += 4u + (uint32_t)bs;
            const int l_name = (int)(w3 & 0xFFu), n_cig = (int)(w4 & 0xFFFFu);
            if (l_seq < 0 || 32ll + l_name + 4ll * n_cig + ((long long)l_seq + 1) / 2 + (long long)l_seq > (long long)bs) { st = REG_MALFORMED; break; }
            if (ref_id != R.tid || (long long)pos >= stop) {
                if (ref_id > R.tid || (ref_id == R.tid && (long long)pos >= stop)) break;
                continue;
            }
            // chop_pacbio_read_by_pos: only alignments that start at or before the window start (decided before the CIGAR is
            // read: the region rule below only skips)
            if (!((long long)pos < start)) continue;
            const uint32_t cig = r + 32u + (uint32_t)l_name;
            const uint32_t sq = cig + 4u * (uint32_t)n_cig;
            const uint32_t rec_end = r + (uint32_t)bs;
            uint32_t ops = cig;
            int32_t n_ops = n_cig;
            if (n_cig == 2) {
                const uint32_t o0 = rd32u(arena + cig), o1 = rd32u(arena + cig + 4);
                if ((o0 & 15u) == 4u && (int32_t)(o0 >> 4) == l_seq && (o1 & 15u) == 3u) {
                    int32_t cnt = 0;
                    const uint32_t cg = find_cg_dev(arena, sq + (uint32_t)((l_seq + 1) / 2) + (uint32_t)l_seq, rec_end, &cnt);
                    if (cg) { ops = cg; n_ops = cnt; }
                }
            }
            // 64 operations a step: the reference length of the region rule (M D N = X) and the walk of cigar2alignstart_by_pos
            // (query cursor: S I M =; reference cursor: M = D - not N, not X, as the reference has it) up to the first operation
            // after which the reference cursor has passed start - 1
            long long r1 = 0, rr = (long long)pos + 1, q = 0;
            bool r1_over = false, walked = false;
            uint32_t last = 0;
            for (int32_t t0 = 0; t0 < n_ops && !(r1_over && walked); t0 += 64) {
                const int32_t t = t0 + (int32_t)lane;
                const bool valid = t < n_ops;
                const uint32_t o = valid ? rd32u(arena + ops + 4u * (uint32_t)t) : 15u;
                const uint32_t code = o & 15u;
                const long long n = (long long)(o >> 4);
                const long long a = (code == 0u || code == 2u || code == 3u || code == 7u || code == 8u) ? n : 0;
                const long long b = (code == 0u || code == 7u || code == 2u) ? n : 0;
                const long long c = (code == 4u || code == 1u || code == 0u || code == 7u) ? n : 0;
                const long long A = wave_scan64(a, lane), B = wave_scan64(b, lane), C = wave_scan64(c, lane);
                if (!r1_over) {
                    if (__any(valid && (long long)pos + r1 + A > beg)) r1_over = true;
                    r1 += __shfl(A, 63);
                }
                if (!walked) {
                    const unsigned long long m = __ballot(valid && rr + B > start - 1);
                    int f;
                    if (m) { f = __ffsll((long long)m) - 1; walked = true; }
                    else f = (n_ops - t0 > 64 ? 64 : n_ops - t0) - 1;        // (the tile's last operation)
                    last = __shfl(code, f);
                    q += __shfl(C, f);
                    rr += __shfl(B, f);
                }
            }
            // the region rule of `samtools view`
            if (!r1_over && (long long)pos + (r1 > 1 ? r1 : 1) <= beg) continue;
            if (n_ops <= 0) { st = REG_NO_CIGAR; break; }
            const long long over = rr - start;
            long long q0, miss;
            if (last == 0u || last == 7u) { q0 = q - over; miss = 0; } else { q0 = q; miss = over; }
            if (2 * miss > R.flank) continue;                                   // miss_bp > flank_length / 2
            const long long seq_len = l_seq > 0 ? l_seq : 1;
            const long long tail = q0 < seq_len ? seq_len - (q0 > 0 ? q0 : 0) : 0;
            const long long want_len = end - start - miss;
            if (q0 < 0) { st = REG_NEG_Q0; break; }
            if (want_len < 0 || !(tail > want_len)) continue;
            if (l_seq <= 0) { st = REG_NO_SEQ; break; }
            if (nk >= KEPT_CAP) { st = REG_KEPT_FULL; break; }
            if (lane == 0) kept[(size_t)g * KEPT_CAP + (size_t)nk] = BamKept{sq, (int32_t)q0, (int32_t)miss, l_seq};
            ++nk;
        }
    }
    if (lane == 0) { n_kept[g] = nk; reg_status[g] = st; }
}

// The bases of device-held reads (4 bits each, BAM's "=ACMGRSVTWYHKDBN") into the ASCII staging layout of pack_kernel: one
// thread per 32-byte chunk.  src[s] = 0 for a sequence that came from the host; else the address of its packed bases, first[s]
// its first base.  chunk_seq as pack_kernel reads it; seq_words = the sequence descriptors pack_kernel reads (vapor::SeqDesc,
// eight words each: word 1 the length, word 4 the first ASCII chunk).
__global__ __launch_bounds__(256) void bam_expand_kernel(uint8_t* __restrict__ ascii, const uint32_t* __restrict__ chunk_seq, uint32_t n_chunks,
                                                        const uint32_t* __restrict__ seq_words,
                                                        const unsigned long long* __restrict__ src, const int32_t* __restrict__ first)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t s = chunk_seq[c];
    const unsigned long long a = src[s];
    if (!a) return;
    const uint8_t* sq = reinterpret_cast<const uint8_t*>(a);
    const int base = (int)(c - seq_words[8u * s + 4u]) * 32;
    int valid = (int)seq_words[8u * s + 1u] - base;
    if (valid > 32) valid = 32;
    const long long i0 = (long long)first[s] + base;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < valid; ++t) {
        const long long i = i0 + t;
        const uint32_t byte = sq[i >> 1];
        const uint32_t nib = (i & 1) ? (byte & 15u) : (byte >> 4);
        const uint32_t ch = (uint32_t)(uint8_t)"=ACMGRSVTWYHKDBN"[nib];
        w[t >> 2] |= ch << ((t & 3) * 8);
    }
    uint4* dst = reinterpret_cast<uint4*>(ascii + (size_t)c * 32);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
#endif  // VBD_EMU

}  // namespace vapor_bamdev
