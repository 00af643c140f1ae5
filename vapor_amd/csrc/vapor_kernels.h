// vapor_kernels.h - device code of libvapor_hip.so (gfx950 / CDNA4 only).
//
// Four kernels; the first three are integer/bit work (no MFMA: there is no dense contraction on this
// path), the last is a few float64 operations per read:
//
//   pack_kernel    ASCII -> bit planes in HBM (2-bit bases, 1-bit "not ACGT", 4-bit symbols)
//   join_kernel    kmerhits (SF:951-983) as an LDS hash join on canonical k-mers.  One workgroup
//                  owns a cost-balanced range of the (read, allele) pairs sorted by allele: it
//                  builds a bucket-sorted table of the allele window's k-mers in LDS and streams
//                  the reads of that allele through it (details above the kernel).
//   clean_kernel   dis_cluster / dis_cluster_2 (SF:551-580) as occupancy bitmaps + ranked group
//                  counters in LDS, the integer reductions of SF:705-733 and SF:1154-1171, and
//                  dis_to_diagnal_most_abundant_defined (SF:582-591) in exact integer arithmetic.
//   finish_kernel  scorer gates, per-read scores, VaPoR_QS / GS / GT / GQ per locus (SF:182-294,
//                  1219-1231, 2054-2077) in float64.
//
// SF = /root/reference/vapor_vali/Simple_function.pyx.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Developer switches (phase / workgroup timing builds, tools/ab.py variants, non-default tuning constants) exist only in
// a build made with -DVAPOR_DEV_BUILD, which reports itself through vapor_build_flags() and a different
// vapor_abi_version(): the product loader (vapor_amd/_lib.py) refuses such a library.  A stray -D of one of them without
// the guard does not compile.
#if !defined(VAPOR_DEV_BUILD) && (defined(VAPOR_PHASE_TIMING) || defined(VAPOR_BLOCK_TIMING) || defined(VAPOR_JQ_FAST_SLACK) || \
                                  defined(VAPOR_CLEAN_THREADS) || defined(VAPOR_BUILD_COST_X8) || defined(VAPOR_AB))
#error "developer switch without -DVAPOR_DEV_BUILD: a product library cannot be an experimental one"
#endif

namespace vapor {

// ------------------------------------------------------------------------------------------
// device-side records
// ------------------------------------------------------------------------------------------
struct SeqDesc {       // 32 B
    uint32_t chunk0;   // first 32-base chunk of this sequence in the planes
    int32_t len;
    int32_t n_exc;     // symbols outside upper-case ACGT
    int32_t n_invalid; // symbols outside invert_base's alphabet (after IUPAC folding)
    uint32_t asc0;     // first 32-byte chunk in the ASCII staging blob (pack only)
    uint32_t flags;
    int32_t n_nocomp;  // symbols complementary() would DROP (SF:471-478: anything outside ATGCN / atgcn); bytes only
    uint32_t pad;
};

struct DPair {         // 40 B
    int32_t seq1, seq2, off2, k;
    uint32_t flags, cap;
    int64_t hit_off;
    int32_t len1, len2;  // sequence lengths (the clean kernels need nothing else of the sequences)
};

struct DTask {         // 16 B: pairs task_pairs[first .. first + n_reads) of one launch, sorted by allele
    int32_t seq2, k, n_reads, first;
};

// plane geometry: every sequence starts on a 32-base chunk; a chunk is 2 words of the 2-bit
// plane, 1 word of the exception plane, 4 words of the 4-bit plane.
#define VP_P2_WORDS_PER_CHUNK 2
#define VP_X4_WORDS_PER_CHUNK 4
#define VP_PAD_CHUNKS 3  // zeroed chunks after each sequence: window reads may run this far

constexpr int MAX_READS_PER_TASK = 64;
#ifndef VAPOR_CLEAN_THREADS
#define VAPOR_CLEAN_THREADS 256
#endif
constexpr int CLEAN_THREADS = VAPOR_CLEAN_THREADS;
constexpr int CLEAN_WAVES = CLEAN_THREADS / 64;
// words of the clean kernels' value bitmap: values i + j and i - j + len2 stay below 2 * 65536 (positions are 16 bit)
constexpr int CLEAN_RANGE_WORDS_MAX = 4096;
constexpr int CLEAN_PER_MAX = (CLEAN_RANGE_WORDS_MAX + CLEAN_THREADS - 1) / CLEAN_THREADS;   // bitmap words per thread at the largest value range

// per-hit working flags inside clean_kernel (upper nibble) and the public ones (lower)
#define HF_C1 1u
#define HF_C2D 2u
#define HF_C2A 4u
#define WF_D1 16u
#define WF_A1 32u

// ------------------------------------------------------------------------------------------
// symbol codes (pack): 0-3 ACGT, 4-7 acgt, 8 N (and folded IUPAC), 9 n, 15 anything else
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sym_code(uint32_t c, bool upper)
{
    uint32_t lower = (c >= 'a' && c <= 'z') ? 1u : 0u;
    uint32_t u = lower ? c - 32u : c;
    uint32_t base;
    switch (u) {
    case 'A': base = 0; break;
    case 'C': base = 1; break;
    case 'G': base = 2; break;
    case 'T': base = 3; break;
    case 'N': case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V':
        base = 8; break;
    default:
        return 15u;
    }
    if (lower && !upper) return base < 4 ? base + 4u : 9u;
    return base;
}

__global__ __launch_bounds__(256) void pack_kernel(const uint8_t* __restrict__ ascii, SeqDesc* seqs,
                                                  int n_seqs, const uint32_t* __restrict__ chunk_seq,
                                                  uint32_t n_chunks, uint32_t* __restrict__ p2,
                                                  uint32_t* __restrict__ e1, uint32_t* __restrict__ x4)
{
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;   // ASCII chunk index
    if (c >= n_chunks) return;
    uint32_t s = chunk_seq[c];
    SeqDesc sd = seqs[s];
    uint32_t local = c - sd.asc0;                // chunk inside the sequence
    int base = (int)local * 32;
    int valid = sd.len - base;
    if (valid > 32) valid = 32;
    const uint4* src = reinterpret_cast<const uint4*>(ascii + (size_t)c * 32);
    uint4 q0 = src[0], q1 = src[1];
    uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    bool upper = sd.flags & 1u;
    uint32_t o2[2] = {0, 0}, oe = 0, o4[4] = {0, 0, 0, 0};
    int nexc = 0, ninv = 0, nnoc = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        uint32_t ch = (w[t >> 2] >> ((t & 3) * 8)) & 0xFFu;
        uint32_t code = (t < valid) ? sym_code(ch, upper) : 0u;
        bool real = t < valid;
        o4[t >> 3] |= code << ((t & 7) * 4);
        o2[t >> 4] |= ((code < 8u) ? (code & 3u) : 0u) << ((t & 15) * 2);
        if (real && code >= 4u) { oe |= 1u << t; ++nexc; }
        if (real && code == 15u) ++ninv;
        // (an IUPAC code folds to N's symbol for the k-mers, but complementary() keeps an N and drops an R)
        if (real && code >= 8u && !(ch == 'N' || ch == 'n')) ++nnoc;
    }
    size_t pc = (size_t)sd.chunk0 + local;
    p2[pc * 2] = o2[0];
    p2[pc * 2 + 1] = o2[1];
    e1[pc] = oe;
    uint4 o;
    o.x = o4[0]; o.y = o4[1]; o.z = o4[2]; o.w = o4[3];
    reinterpret_cast<uint4*>(x4)[pc] = o;
    if (nexc) atomicAdd(&seqs[s].n_exc, nexc);
    if (ninv) atomicAdd(&seqs[s].n_invalid, ninv);
    if (nnoc) atomicAdd(&seqs[s].n_nocomp, nnoc);
}

// ------------------------------------------------------------------------------------------
// derived sequences: bit planes assembled on the device from slices of sequences that were uploaded as bytes
// ------------------------------------------------------------------------------------------
// An allele the reference builds by string surgery on a window it has read - ref[:f] + ref[-f:] (SF:1712), ref[:f] + mid + mid
// + ref[-f:] (SF:1755), ref[:f] + reverse(complementary(mid)) + ref[-f:] (SF:1907), flank + ins_seq + flank (SF:1872), the
// str.upper() twins of abs_dis_m1b (SF:183-184) - travels as a list of segments; this kernel writes its three planes from the
// parents' 4-bit plane (which holds everything: case, N, folded IUPAC, invalid).  One thread per 32-symbol chunk.
struct DSeg {          // 20 B: dst symbols [dst, dst + len) of the derived sequence = parent[off .. off + len), reversed and
    uint32_t chunk0;   // complemented when rc; chunk0 = the parent's first plane chunk
    int32_t off, len, dst;
    uint32_t rc;
};

__device__ __forceinline__ uint32_t comp_code(uint32_t c)
{
    // invert_base (SF:19-20): A<->T, C<->G in either case; N / n (and what folds to them) and invalid symbols keep their code
    return c < 8u ? (c ^ 3u) : c;
}

__global__ __launch_bounds__(256) void derive_kernel(SeqDesc* seqs, const uint32_t* __restrict__ chunk_seq, uint32_t n_chunks,
                                                    const int32_t* __restrict__ seg_first, const DSeg* __restrict__ segs,
                                                    int first_derived, uint32_t* __restrict__ p2, uint32_t* __restrict__ e1,
                                                    uint32_t* __restrict__ x4)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;       // chunk among the derived sequences' chunks
    if (c >= n_chunks) return;
    const uint32_t s = chunk_seq[c];                               // sequence index (>= first_derived)
    const SeqDesc sd = seqs[s];
    const uint32_t local = c - sd.asc0;                            // (asc0 of a derived sequence: its first chunk in chunk_seq)
    const int base = (int)local * 32;
    const int valid = min(32, sd.len - base);
    const bool upper = sd.flags & 1u;
    int g = seg_first[s - first_derived];
    const int g_end = seg_first[s - first_derived + 1];
    while (g + 1 < g_end && segs[g + 1].dst <= base) ++g;           // the segment that holds symbol `base`
    DSeg sg = segs[g];
    uint32_t o2[2] = {0, 0}, oe = 0, o4[4] = {0, 0, 0, 0};
    int nexc = 0, ninv = 0;
    for (int t = 0; t < valid; ++t) {
        const int pos = base + t;
        while (pos >= sg.dst + sg.len && g + 1 < g_end) sg = segs[++g];
        const int rel = pos - sg.dst;
        const uint32_t src = (uint32_t)(sg.rc ? sg.off + sg.len - 1 - rel : sg.off + rel);
        uint32_t code = (x4[((size_t)sg.chunk0 << 2) + (src >> 3)] >> ((src & 7u) * 4u)) & 15u;
        if (sg.rc) code = comp_code(code);
        if (upper) code = (code >= 4u && code < 8u) ? code - 4u : (code == 9u ? 8u : code);
        o4[t >> 3] |= code << ((t & 7) * 4);
        o2[t >> 4] |= ((code < 8u) ? (code & 3u) : 0u) << ((t & 15) * 2);
        if (code >= 4u) { oe |= 1u << t; ++nexc; }
        if (code == 15u) ++ninv;
    }
    const size_t pc = (size_t)sd.chunk0 + local;
    p2[pc * 2] = o2[0];
    p2[pc * 2 + 1] = o2[1];
    e1[pc] = oe;
    uint4 o;
    o.x = o4[0]; o.y = o4[1]; o.z = o4[2]; o.w = o4[3];
    reinterpret_cast<uint4*>(x4)[pc] = o;
    if (nexc) atomicAdd(&seqs[s].n_exc, nexc);
    if (ninv) atomicAdd(&seqs[s].n_invalid, ninv);
}

// ------------------------------------------------------------------------------------------
// k-mer keys
// ------------------------------------------------------------------------------------------
template <int BPS, int K>
struct KeyT {
    static constexpr int BITS = BPS * K;
    static constexpr int NW = (BITS + 31) / 32;
    static constexpr int TOPBITS = BITS - 32 * (NW - 1);
    static constexpr uint32_t TOPMASK = TOPBITS == 32 ? 0xFFFFFFFFu : ((1u << TOPBITS) - 1u);
    uint32_t w[NW];
    __device__ __forceinline__ bool operator==(const KeyT& o) const
    {
        bool e = true;
#pragma unroll
        for (int t = 0; t < NW; ++t) e = e && (w[t] == o.w[t]);
        return e;
    }
};

// window of K symbols starting at symbol `pos` of a packed plane (LDS or global)
template <int BPS, int K, typename P>
__device__ __forceinline__ KeyT<BPS, K> extract_key(P plane, uint32_t pos)
{
    using KT = KeyT<BPS, K>;
    uint32_t bit = pos * BPS;
    uint32_t wi = bit >> 5, sh = bit & 31u;
    uint32_t raw[KT::NW + 1];
#pragma unroll
    for (int t = 0; t <= KT::NW; ++t) raw[t] = plane[wi + t];
    KT k;
#pragma unroll
    for (int t = 0; t < KT::NW; ++t) k.w[t] = __builtin_amdgcn_alignbit(raw[t + 1], raw[t], sh);
    k.w[KT::NW - 1] &= KT::TOPMASK;
    return k;
}

template <int BPS>
__device__ __forceinline__ uint32_t rev_syms(uint32_t x)
{
    x = __brev(x);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    if (BPS == 4) x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    return x;
}

// reverse complement of a key: invert_base (SF:19-20) applied to the reversed k-mer (SF:1421)
template <int BPS, int K>
__device__ __forceinline__ KeyT<BPS, K> revcomp_key(const KeyT<BPS, K>& k)
{
    using KT = KeyT<BPS, K>;
    constexpr int SH = 32 * KT::NW - KT::BITS;
    uint32_t tmp[KT::NW + 1];
#pragma unroll
    for (int t = 0; t < KT::NW; ++t) tmp[t] = rev_syms<BPS>(k.w[KT::NW - 1 - t]);
    tmp[KT::NW] = 0;
    KT r;
#pragma unroll
    for (int t = 0; t < KT::NW; ++t) {
        uint32_t v = SH ? __builtin_amdgcn_alignbit(tmp[t + 1], tmp[t], SH) : tmp[t];
        if (BPS == 2) {
            v = ~v;
        } else {
            uint32_t m = (v >> 3) & 0x11111111u;       // N / n / invalid keep their code
            v ^= 0x33333333u & ~(m * 3u);
        }
        r.w[t] = v;
    }
    r.w[KT::NW - 1] &= KT::TOPMASK;
    return r;
}

// any nibble == 15 (a symbol that matches nothing; such allele k-mers are left out of the table)
template <int BPS, int K>
__device__ __forceinline__ bool key_has_invalid(const KeyT<BPS, K>& k)
{
    uint32_t any = 0;
#pragma unroll
    for (int t = 0; t < KeyT<BPS, K>::NW; ++t) {
        uint32_t x = k.w[t];
        any |= x & (x >> 1) & (x >> 2) & (x >> 3) & 0x11111111u;
    }
    return any != 0;
}

// any exception bit in [pos, pos+K) of a 1-bit plane
template <int K, typename P>
__device__ __forceinline__ bool any_exc(P plane, uint32_t pos)
{
    uint32_t wi = pos >> 5, sh = pos & 31u;
    uint32_t w0 = plane[wi], w1 = plane[wi + 1], w2 = plane[wi + 2];
    uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
    uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    if (K <= 32) return (lo & (K == 32 ? 0xFFFFFFFFu : ((1u << (K & 31)) - 1u))) != 0;
    return (lo | (hi & ((1u << ((K - 32) & 31)) - 1u))) != 0;
}

// ------------------------------------------------------------------------------------------
// join kernel
// ------------------------------------------------------------------------------------------
// kmerhits (SF:951-983) as an LDS hash join on CANONICAL k-mers.
//
// A hit (j, i) exists when allele k-mer a_j equals read k-mer r_i, and once more when it equals
// revcomp(r_i).  canon(x) = min(x, revcomp(x)) satisfies canon(a) == canon(r)  <=>  a == r or
// a == revcomp(r), so ONE lookup per read position finds the candidates of both orientations and
// the verification emits one tuple per orientation that really matches (two for a read k-mer that
// is its own reverse complement, as the reference does).
//
// Table (per allele tile of up to TA k-mer positions), all in LDS:
//   start[h]   u16, NB+1 entries: first slot of bucket h (bucket = hash of the canonical key)
//   entries[]  u16: tile positions sorted by bucket (counting sort: packed 16-bit counters updated
//              with 32-bit LDS atomics, block scan, fill from the back of each bucket)
// Probe: a wave takes 1024 consecutive read positions, copies their packed words into its own LDS
// strip (so the loop below touches no global memory except the hit stores, which nothing waits
// for), each lane owns 16 of the positions and slides the key / reverse-complement registers over
// them; the (position, slot) candidates of the whole wave are compacted into a per-wave LDS queue
// and verified 128 at a time, so the data-dependent bucket sizes do not idle lanes.
//
// A task is a contiguous range of the batch's pairs sorted by allele (cost-balanced on the host);
// the table is rebuilt only when the allele changes inside the range.
constexpr int JCHUNK = 1024;                      // read positions per wave pass (16 per lane)
// The queue's fast path takes a position whose candidates over the wave number at most QCAP - this (see the fill loop).
// 128 is the value; tests/test_gpu_parity.py::test_queue_edge was checked once against a build with 127 (it fails there).
#ifndef VAPOR_JQ_FAST_SLACK
#define VAPOR_JQ_FAST_SLACK 128
#endif

// Geometry of the join workgroup: it owns a whole CU's LDS (16 waves, one table of up to 24576 positions
// in 32768 buckets).  Smaller tables would not raise residency: at ~117 VGPRs four waves per SIMD is the
// register limit as well.
struct JoinBig { static constexpr int THREADS = 1024, WPS = 4, TA2 = 24576, TA4 = 21504, NB_LOG2 = 15, FILT_LOG2 = 17, QCAP = 256; };
// (two workgroups per CU with half a table each were tried and lost: DESIGN.md section 4)
using JoinCfg = JoinBig;

template <typename C, int BPS> __host__ __device__ constexpr int tile_pos() { return BPS == 2 ? C::TA2 : C::TA4; }
template <typename C, int BPS> __host__ __device__ constexpr int tile_words() { return ((tile_pos<C, BPS>() + 64) * BPS) / 32 + 8; }
template <typename C, int BPS> __host__ __device__ constexpr int etile_words() { return BPS == 2 ? (tile_pos<C, BPS>() + 64) / 32 + 8 : 0; }
template <int BPS> __host__ __device__ constexpr int rbuf_words() { return ((JCHUNK + 64) * BPS) / 32 + 4; }

template <typename C, int BPS>
constexpr size_t join_lds_bytes()
{
    return sizeof(uint32_t) * ((1 << C::NB_LOG2) / 2 + 2 + 1) + sizeof(uint32_t) * ((1 << C::FILT_LOG2) / 32) +
           sizeof(uint32_t) * tile_words<C, BPS>() +
           sizeof(uint32_t) * etile_words<C, BPS>() + sizeof(unsigned long long) * MAX_READS_PER_TASK +
           sizeof(uint32_t) * (C::THREADS / 64) * (C::QCAP + 4) + sizeof(uint32_t) * (C::THREADS / 64) * rbuf_words<BPS>() +
           sizeof(uint32_t) * (2 * (C::THREADS / 64) + 4 + MAX_READS_PER_TASK + 2) + sizeof(uint32_t) * 4 * MAX_READS_PER_TASK +
           sizeof(uint16_t) * tile_pos<C, BPS>();
}

template <int BPS, int K>
__device__ __forceinline__ bool key_less(const KeyT<BPS, K>& a, const KeyT<BPS, K>& b)
{
    using KT = KeyT<BPS, K>;
    bool lt = false, decided = false;
#pragma unroll
    for (int t = KT::NW - 1; t >= 0; --t) {
        if (!decided && a.w[t] != b.w[t]) { lt = a.w[t] < b.w[t]; decided = true; }
    }
    return lt;
}

// 32-bit hash of the canonical key: its top NB_LOG2 bits are the bucket, its top FILT_LOG2 bits the bit of the
// occupancy filter (so a filter bit covers a quarter of a bucket's key space)
template <int BPS, int K>
__device__ __forceinline__ uint32_t canon_hash(const KeyT<BPS, K>& k, const KeyT<BPS, K>& rc)
{
    using KT = KeyT<BPS, K>;
    const bool use_rc = key_less<BPS, K>(rc, k);
    // a key of at most 24 bits (k = 10 on the 2-bit plane): v_mul_u32_u24 is a full-rate instruction, the 32-bit
    // multiply takes four passes; the bucket spread of the two is the same (variance 0.613 vs 0.611 at load 0.61)
    if (KT::NW == 1 && KT::BITS <= 24) return __umul24(use_rc ? rc.w[0] : k.w[0], 0xC2B2AFu);
    uint32_t x = (use_rc ? rc.w[0] : k.w[0]) * 0x9E3779B1u;
    if (KT::NW > 1) x ^= (use_rc ? rc.w[1] : k.w[1]) * 0x85EBCA77u;
    if (KT::NW > 2) x ^= (use_rc ? rc.w[2] : k.w[2]) * 0xC2B2AE3Du;
    if (KT::NW > 3) x ^= (use_rc ? rc.w[3] : k.w[3]) * 0x27D4EB2Fu;
    if (KT::NW > 4) x ^= (use_rc ? rc.w[4] : k.w[4]) * 0x165667B1u;
    if (KT::NW > 1) { x ^= x >> 15; x *= 0x2C1B3C6Du; }
    return x;
}

// The occupancy filter in front of the table is a blocked Bloom filter with two bits per key inside one 32-bit word: the
// word and the first bit are the hash's top FILT_LOG2 bits (a bit per quarter bucket), the second bit five lower hash
// bits.  A read k-mer is looked up only when both are set: 20 000 keys in 2^17 bits pass 6.5 % of the absent k-mers
// instead of 14 % with one bit, at three more vector instructions per read position and no further LDS read
// (cfg2 join 0.1555 -> 0.1509 ms, cfg3 3.49 -> 3.39 ms: profiles/r03_ab_bloom.txt).
__device__ __forceinline__ uint32_t filt_mask(uint32_t hx, uint32_t fb)
{
    return (1u << (fb & 31u)) | (1u << ((hx >> 10) & 31u));
}

// Cross-lane steps as DPP modifiers of VALU instructions (gfx9: row_shr, wave_shr, row_bcast15/31) instead of
// ds_bpermute: they do not occupy the LDS pipe that the table and bitmap traffic needs and their latency is a
// few cycles.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK, typename T>
__device__ __forceinline__ T dpp_from(T identity, T v)
{
    return (T)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
struct OpAdd { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct OpMin { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a < b ? a : b; } };
struct OpMax { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; } };
// inclusive scan over the 64 lanes (lane 63 ends with the reduction)
template <typename OP, typename T>
__device__ __forceinline__ T wave_scan(T v, T identity)
{
    OP op;
    v = op(v, dpp_from<0x111, 0xF>(identity, v));   // row_shr:1
    v = op(v, dpp_from<0x112, 0xF>(identity, v));   // row_shr:2
    v = op(v, dpp_from<0x114, 0xF>(identity, v));   // row_shr:4
    v = op(v, dpp_from<0x118, 0xF>(identity, v));   // row_shr:8
    v = op(v, dpp_from<0x142, 0xA>(identity, v));   // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_from<0x143, 0xC>(identity, v));   // row_bcast:31 into rows 2 and 3
    return v;
}

// Developer build only (-DVAPOR_PHASE_TIMING, tools/phase_timing.py): shader-clock ticks per kernel phase, summed
// over the stamping lanes into g_phase[].  Compiled out of the product library.
#if defined(VAPOR_PHASE_TIMING) || defined(VAPOR_BLOCK_TIMING)
__device__ unsigned long long g_block_ticks[4096];     // join_kernel: 100 MHz ticks from start to end of every workgroup
__device__ unsigned long long g_block_info[4096 * 4];  // HW_ID, XCC_ID, start, end (100 MHz) of every workgroup
__device__ unsigned long long g_block_phase[4096 * 8]; // per workgroup: the phase ticks of its 16 stamping lanes
#endif
#ifdef VAPOR_PHASE_TIMING
__device__ unsigned long long g_phase[64];
// ticks are summed in (scalar) registers and flushed once per stamping lane, so the stamps cost a few SALU
// instructions and do not queue atomics behind the phase being measured
template <int BASE, int N>
struct PhaseClockT {
    long long last;
    unsigned long long acc[N];
    __device__ __forceinline__ PhaseClockT() : last(clock64())
    {
#pragma unroll
        for (int x = 0; x < N; ++x) acc[x] = 0;
    }
    __device__ __forceinline__ void mark(int k, bool)
    {
        const long long t = clock64();
#pragma unroll
        for (int x = 0; x < N; ++x)
            if (x == k - BASE) acc[x] += (unsigned long long)(t - last);
        last = t;
    }
    __device__ __forceinline__ void count(int k, unsigned long long v)
    {
#pragma unroll
        for (int x = 0; x < N; ++x)
            if (x == k - BASE) acc[x] += v;
    }
    __device__ __forceinline__ void flush(bool who)
    {
        if (who) {
#pragma unroll
            for (int x = 0; x < N; ++x)
                if (acc[x]) atomicAdd(&g_phase[BASE + x], acc[x]);
            if (BASE == 0 && blockIdx.x < 4096) {
#pragma unroll
                for (int x = 0; x < N && x < 8; ++x)
                    if (acc[x]) atomicAdd(&g_block_phase[blockIdx.x * 8 + x], acc[x]);
            }
        }
    }
};
#else
template <int BASE, int N>
struct PhaseClockT {
    __device__ __forceinline__ void mark(int, bool) {}
    __device__ __forceinline__ void count(int, unsigned long long) {}
    __device__ __forceinline__ void flush(bool) {}
};
#endif
using JoinClock = PhaseClockT<0, 8>;
using CleanClock = PhaseClockT<8, 28>;

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    return wave_scan<OpAdd>(v, 0u);
}

// ------------------------------------------------------------------------------------------
// Dot records.  The join does not write single dots but RUNS of dots: a dot (j, i) of the same strand is
// followed by (j+1, i+1) exactly when read[i+K] == allele[j+K], a reverse-complement dot by (j-1, i+1) exactly
// when allele[j-1] == comp(read[i+K]); long reads make such runs several dots long (tens for accurate reads).
//   bits  0..15  i of the first dot        bits 32..47  number of dots (1..32)
//   bits 16..31  j of the first dot        bit  48      0: (j+t, i+t)   1: (j-t, i+t)
// A run never crosses a 32-aligned read position of its strip nor an allele tile, so whether a dot starts a
// run is decided from the two sequences alone (the symbol before it differs, or it sits on such a boundary)
// and every dot belongs to exactly one record.  All dots of a record share their diagonal group AND their
// anti-diagonal group (consecutive values differ by 0 or 2 < 10), so the cleaning flags are per record.
// The join forms runs only of same-strand dots and only in the plain case (2-bit planes, no exception symbol on
// either side); every other dot is its own record (the format and the cleaning kernels take reverse-complement
// runs all the same).
#define VREC_I(r) ((int)((r) & 0xFFFFull))
#define VREC_J(r) ((int)(((r) >> 16) & 0xFFFFull))
#define VREC_LEN(r) ((int)(((r) >> 32) & 0xFFFFull))
#define VREC_RC(r) ((bool)(((r) >> 48) & 1ull))
#define VREC_MAX_LEN 32

__device__ __forceinline__ uint32_t sym2(const uint32_t* plane, uint32_t pos)
{
    return (plane[pos >> 4] >> ((pos & 15u) * 2u)) & 3u;
}
// the 32 symbols from position pos of a 2-bit plane (reads three words)
__device__ __forceinline__ unsigned long long win2(const uint32_t* plane, uint32_t pos)
{
    const uint32_t wi = pos >> 4, sh = (pos & 15u) * 2u;
    const uint32_t w0 = plane[wi], w1 = plane[wi + 1], w2 = plane[wi + 2];
    return ((unsigned long long)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
}

// verify queued candidates [from, from+n), n <= 128, two per lane so that their LDS reads overlap.
// item = local read position << 16 | entry slot.  The entry's k-mer is compared with the read k-mer in both
// orientations; a same-strand dot also decides, in the same step, whether it starts a run (the symbols just before
// it differ, or it sits on a 32-aligned read position / the first usable allele position) and how long the run is
// (XOR of the 32 symbols after the k-mer on both sides, ffs).  Everything a candidate needs - the symbol before the
// k-mer, the k-mer, the 32 symbols after it - comes out of ONE group of LDS reads per side (the words from one word
// before the k-mer's first word on), so a call is item -> entry -> windows -> one counter update -> stores: no
// second pass over the dots.  Reverse-complement dots (few outside inversions) and all dots of pairs that cannot
// form runs are single-dot records.  The counter moves once per call: cnt += records | dots << 32.
// MERGE (runs are formed: 2-bit planes, no exception symbol on either side) is a template parameter so that the body is
// one straight line: with a (uniform) branch on it inside, the compiler reads the symbols after the k-mer in a second
// round trip under that branch and does not start the second candidate's reads before the first is done.
// MERGE: 0 = every dot its own record; 1 = runs, neither side has a symbol outside upper-case ACGT; 3 = runs, the READ has
// such symbols (an N in a read; the allele has none): the positions whose k-mer covers one are not looked up at all, a run
// ends before the first of them and does not continue from a dot whose predecessor's read symbol is one - `etile` then
// holds the exception bits of the wave's strip of the read; 2 = runs, the allele has
// such symbols (soft-masked references: the usual case with real genomes for the scorers that do not upper-case) - a
// k-mer that covers one is not in the table, so a run ends before the first of them and the dot before a candidate
// exists only if the symbol before it is none.  The allele's exception bits come from etile.
// NQ: candidates per lane and call (2; 1 for the 4-bit planes at window sizes 30 and 40, where the two streams of ONE
// candidate take the registers two take elsewhere - the caller then passes at most 64 candidates)
template <int BPS, int K, int MERGE, int NQ = 2>
__device__ __forceinline__ void join_verify_t(uint32_t* myq, int from, int n, const uint16_t* entries,
                                              const uint32_t* rbuf, const uint32_t* tile, const uint32_t* etile, int cb, int ts,
                                              int off2, int tn, int nk1, unsigned long long* cnt_r, uint32_t cap,
                                              unsigned long long* out)
{
    constexpr bool merge = MERGE != 0;
    using KT = KeyT<BPS, K>;
    const int lane = threadIdx.x & 63;
    // Both candidates of a lane go through the same straight-line code (a lane without a second candidate
    // re-reads item 0 and masks the result): with a branch per candidate the LDS reads of the second would only
    // start when the first is done.
    // (for the 2-bit planes the three verdicts of a candidate are kept as integers that are ZERO when true - differences
    // OR-ed with all-ones masks from sign shifts - so that each wave vote is one v_cmp_eq: a vote on a boolean that is
    // the AND of several compares costs a select and a second compare on top of them, and those issue alone)
    bool same[2] = {false, false}, rcm[2] = {false, false}, head[2] = {false, false};
    int len[2] = {1, 1};
    const uint32_t out0 = (uint32_t)((lane - n) >> 31), out1 = (uint32_t)((lane + 64 - n) >> 31);   // ~0 if the slot is filled
    uint32_t who[2] = {0u, 0u};                    // il | e << 16
    uint32_t item[2] = {0u, 0u}, e[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < NQ; ++q) item[q] = myq[from + ((q * 64 + lane < n) ? q * 64 + lane : 0)];
#pragma unroll
    for (int q = 0; q < NQ; ++q) e[q] = entries[item[q] & 0xFFFFu];
    const int emin = max(0, off2 - ts);            // first allele position of this tile that may carry a dot
    // records are assembled by one add: (e << 16 | il) + (ts - off2) << 16 + cb (neither field carries: both stay below 2^16)
    const uint32_t rec_bias = ((uint32_t)(ts - off2) << 16) + (uint32_t)cb;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const uint32_t il = item[q] >> 16;
        const bool in = (q * 64 + lane < n) && ts + (int)e[q] >= off2;
        who[q] = il | (e[q] << 16);
        len[q] = 1;
        if (BPS == 2) {
            // Both sides are read as a stream that starts ONE symbol before the k-mer (word (pos - 1) >> 4 on; valid LDS
            // even for position 0: the strips and the tile are preceded by other regions, and such a position never
            // consults that symbol): bits 0-1 are the symbol before, bits 2 .. 2K+1 the k-mer, the next 64 the symbols a run
            // can extend over.  One XOR of the two streams answers all three questions.
            constexpr int SB = 2 * (K + 1);
            constexpr int EW = SB / 32, ES = SB % 32;            // where the 64 bits after the k-mer start
            constexpr int NN = EW + 3;
            const int pr1 = (int)il - 1, pa1 = (int)e[q] - 1;
            const uint32_t shr = (uint32_t)(pr1 & 15) * 2u, sha = (uint32_t)(pa1 & 15) * 2u;
            const uint32_t* rp = rbuf + (pr1 >> 4);
            const uint32_t* tp = tile + (pa1 >> 4);
            uint32_t rw[NN + 1], tw[NN + 1];
#pragma unroll
            for (int x = 0; x <= NN; ++x) { rw[x] = rp[x]; tw[x] = tp[x]; }
            uint32_t sr[NN], sx[NN];                              // read stream; read stream XOR allele stream
#pragma unroll
            for (int x = 0; x < NN; ++x) {
                sr[x] = __builtin_amdgcn_alignbit(rw[x + 1], rw[x], shr);
                sx[x] = sr[x] ^ __builtin_amdgcn_alignbit(tw[x + 1], tw[x], sha);
            }
            // not a candidate at all: an empty slot of the call, or an allele position before the pair's window
            (void)in;
            const uint32_t notin = ~(q ? out1 : out0) | (uint32_t)(((int)e[q] - (off2 - ts)) >> 31);
            uint32_t df = notin;
#pragma unroll
            for (int x = 0; x <= EW; ++x) {
                const uint32_t m = (x == EW ? ((1u << ES) - 1u) : 0xFFFFFFFFu) & (x == 0 ? ~3u : 0xFFFFFFFFu);
                if (m) df |= sx[x] & m;
            }
            KT kf;
#pragma unroll
            for (int x = 0; x < KT::NW; ++x)
                kf.w[x] = (SB <= 32) ? (sr[0] >> 2) : __builtin_amdgcn_alignbit(sr[x + 1], sr[x], 2);
            kf.w[KT::NW - 1] &= KT::TOPMASK;
            const KT kr = revcomp_key<BPS, K>(kf);
            uint32_t dr = notin;
#pragma unroll
            for (int x = 0; x < KT::NW; ++x) {
                // allele k-mer word = (read stream XOR difference) >> 2
                const uint32_t aw = (SB <= 32) ? ((sr[0] ^ sx[0]) >> 2) : __builtin_amdgcn_alignbit(sr[x + 1] ^ sx[x + 1], sr[x] ^ sx[x], 2);
                dr |= (x == KT::NW - 1 ? aw & KT::TOPMASK : aw) ^ kr.w[x];
            }
            same[q] = df == 0u;
            rcm[q] = dr == 0u;
            // the dot before this one exists <=> the symbols just before match (and nothing forbids a run there): then it
            // is not the head of its run
            uint32_t cont = merge ? ((il & (VREC_MAX_LEN - 1)) & (uint32_t)((emin - (int)e[q]) >> 31) &
                                     (uint32_t)((int)((sx[0] & 3u) - 1u) >> 31))
                                  : 0u;
            if (MERGE == 2) {
                // ... and the allele symbol before the k-mer is an ordinary one (else no k-mer starts there)
                const uint32_t eb = (etile[pa1 >> 5] >> ((uint32_t)pa1 & 31u)) & 1u;
                cont &= eb - 1u;
            }
            if (MERGE == 3) {
                // ... and the read symbol before the k-mer is an ordinary one (etile = the exception bits of the wave's strip;
                // position -1 of a strip is never consulted: its dot sits on a 32-aligned read position)
                const uint32_t eb = (etile[pr1 >> 5] >> ((uint32_t)pr1 & 31u)) & 1u;
                cont &= eb - 1u;
            }
            head[q] = (df | cont) == 0u;
            if (merge) {
                const uint32_t xl = ES ? __builtin_amdgcn_alignbit(sx[EW + 1], sx[EW], ES) : sx[EW];
                const uint32_t xh = (ES ? __builtin_amdgcn_alignbit(sx[EW + 2], sx[EW + 1], ES) : sx[EW + 1]) | 0x40000000u;
                // (a run holds at most VREC_MAX_LEN = 32 dots, so 31 symbols after the k-mer decide: the set bit 62 ends the search)
                const int ext = __builtin_ctzll(((unsigned long long)xh << 32) | xl) >> 1;
                len[q] = min(1 + ext, min(min(VREC_MAX_LEN - (int)(il & (VREC_MAX_LEN - 1)), nk1 - (cb + (int)il)), tn - (int)e[q]));
                if (MERGE == 2) {
                    // the k-mers e .. e+len-1 cover the symbols up to e+len-1+K-1: the first exception symbol at or after
                    // e+K, d symbols on, allows d + 1 dots
                    const uint32_t p = e[q] + (uint32_t)K;
                    const uint32_t w0 = etile[p >> 5], w1 = etile[(p >> 5) + 1], w2 = etile[(p >> 5) + 2];
                    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, p & 31u);
                    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, p & 31u) | 0x80000000u;
                    len[q] = min(len[q], 1 + (int)__builtin_ctzll(((unsigned long long)hi << 32) | lo));
                }
                if (MERGE == 3) {
                    // the same on the read's side: the first exception symbol of the strip at or after il+K, d symbols on,
                    // allows d + 1 dots
                    const uint32_t p = il + (uint32_t)K;
                    const uint32_t w0 = etile[p >> 5], w1 = etile[(p >> 5) + 1], w2 = etile[(p >> 5) + 2];
                    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, p & 31u);
                    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, p & 31u) | 0x80000000u;
                    len[q] = min(len[q], 1 + (int)__builtin_ctzll(((unsigned long long)hi << 32) | lo));
                }
            }
        } else if (merge) {
            // 4-bit planes (both sides hold symbols outside upper-case ACGT, e.g. the self plot of a soft-masked window;
            // no symbol of the allele is one that matches nothing - the caller's condition for `merge` here): the same
            // stream from one symbol before the k-mer, a nibble per symbol, 128 bits of extension
            // (window sizes 30 and 40: runs of at most 16 dots, cut at 16-aligned read positions - 60 bits of extension instead of
            // 124, two words less per stream; the record format allows any cut as long as heads and lengths agree on it)
            constexpr int RB = (NQ == 1) ? 16 : VREC_MAX_LEN;
            constexpr int SB = 4 * (K + 1);
            constexpr int EW = SB / 32, ES = SB % 32;
            constexpr int NN = EW + (RB == 16 ? 3 : 5);
            const int pr1 = (int)il - 1, pa1 = (int)e[q] - 1;
            const uint32_t shr = (uint32_t)(pr1 & 7) * 4u, sha = (uint32_t)(pa1 & 7) * 4u;
            const uint32_t* rp = rbuf + (pr1 >> 3);
            const uint32_t* tp = tile + (pa1 >> 3);
            uint32_t sr[NN], sx[NN];
#pragma unroll
            for (int x = 0; x < NN; ++x) {
                sr[x] = __builtin_amdgcn_alignbit(rp[x + 1], rp[x], shr);
                sx[x] = sr[x] ^ __builtin_amdgcn_alignbit(tp[x + 1], tp[x], sha);
            }
            (void)in;
            const uint32_t notin = ~(q ? out1 : out0) | (uint32_t)(((int)e[q] - (off2 - ts)) >> 31);
            uint32_t df = notin;
#pragma unroll
            for (int x = 0; x <= EW; ++x) {
                const uint32_t m = (x == EW ? ((1u << ES) - 1u) : 0xFFFFFFFFu) & (x == 0 ? ~15u : 0xFFFFFFFFu);
                if (m) df |= sx[x] & m;
            }
            KT kf, a;
#pragma unroll
            for (int x = 0; x < KT::NW; ++x) {
                kf.w[x] = __builtin_amdgcn_alignbit(sr[x + 1], sr[x], 4);
                a.w[x] = __builtin_amdgcn_alignbit(sr[x + 1] ^ sx[x + 1], sr[x] ^ sx[x], 4);
            }
            kf.w[KT::NW - 1] &= KT::TOPMASK;
            a.w[KT::NW - 1] &= KT::TOPMASK;
            const KT kr = revcomp_key<BPS, K>(kf);
            uint32_t dr = notin;
#pragma unroll
            for (int x = 0; x < KT::NW; ++x) dr |= a.w[x] ^ kr.w[x];
            same[q] = df == 0u;
            rcm[q] = dr == 0u;
            const uint32_t cont = (il & (RB - 1)) & (uint32_t)((emin - (int)e[q]) >> 31) &
                                  (uint32_t)((int)((sx[0] & 15u) - 1u) >> 31);
            head[q] = (df | cont) == 0u;
            int ext;
            if (RB == 16) {
                uint32_t xw[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) xw[j] = ES ? __builtin_amdgcn_alignbit(sx[EW + j + 1], sx[EW + j], ES) : sx[EW + j];
                xw[1] |= 0x10000000u;                    // (15 symbols after the k-mer decide: at most 16 dots per run)
                ext = __builtin_ctzll(((unsigned long long)xw[1] << 32) | xw[0]) >> 2;
            } else {
                uint32_t xw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xw[j] = ES ? __builtin_amdgcn_alignbit(sx[EW + j + 1], sx[EW + j], ES) : sx[EW + j];
                xw[3] |= 0x10000000u;                    // (31 symbols after the k-mer decide: at most 32 dots per run)
                const unsigned long long lo64 = ((unsigned long long)xw[1] << 32) | xw[0], hi64 = ((unsigned long long)xw[3] << 32) | xw[2];
                ext = lo64 ? (__builtin_ctzll(lo64) >> 2) : 16 + (__builtin_ctzll(hi64) >> 2);
            }
            len[q] = min(1 + ext, min(min(RB - (int)(il & (RB - 1)), nk1 - (cb + (int)il)), tn - (int)e[q]));
        } else {
            const KT kf = extract_key<BPS, K>(rbuf, il);
            const KT a = extract_key<BPS, K>(tile, e[q]);
            const KT kr = revcomp_key<BPS, K>(kf);
            same[q] = in && (a == kf);
            rcm[q] = in && (a == kr);
            head[q] = same[q];
        }
    }
    const unsigned long long ms0 = __ballot(same[0]), ms1 = __ballot(same[1]), mr0 = __ballot(rcm[0]), mr1 = __ballot(rcm[1]);
    const uint32_t dots = __popcll(ms0) + __popcll(ms1) + __popcll(mr0) + __popcll(mr1);
    if (dots == 0) return;
    const unsigned long long mh0 = __ballot(head[0]), mh1 = __ballot(head[1]);
    const uint32_t nh0 = __popcll(mh0), nh1 = __popcll(mh1), nr0 = __popcll(mr0), nrec = nh0 + nh1 + nr0 + __popcll(mr1);
    auto rank = [&](unsigned long long m) {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    auto record = [&](uint32_t w, int ln, uint32_t rc) -> unsigned long long {
        return (unsigned long long)(w + rec_bias) | ((unsigned long long)ln << 32) | ((unsigned long long)rc << 48);
    };
    unsigned long long old = 0;
    if (lane == 0) old = atomicAdd(cnt_r, (unsigned long long)nrec | ((unsigned long long)dots << 32));
    if (nrec == 0) return;                         // a call of continuation dots only
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old);
    uint32_t slot;
    slot = base + rank(mh0);
    if (head[0] && slot < cap) out[slot] = record(who[0], len[0], 0u);
    slot = base + nh0 + rank(mh1);
    if (head[1] && slot < cap) out[slot] = record(who[1], len[1], 0u);
    slot = base + nh0 + nh1 + rank(mr0);
    if (rcm[0] && slot < cap) out[slot] = record(who[0], 1, 1u);
    slot = base + nh0 + nh1 + nr0 + rank(mr1);
    if (rcm[1] && slot < cap) out[slot] = record(who[1], 1, 1u);
}

// runs on the 4-bit planes: two candidates per lane up to this window size, one beyond it (a candidate's two streams then
// need the registers two take below)
constexpr int X4_MERGE_MAX_K = 40;
constexpr int X4_TWO_PER_LANE_MAX_K = 20;

template <int BPS, int K, int EXC>
__device__ __forceinline__ void join_verify(uint32_t* myq, int from, int n, const uint16_t* entries,
                                            const uint32_t* rbuf, const uint32_t* tile, const uint32_t* etile, int cb, int ts,
                                            int off2, int tn, int nk1, bool merge, unsigned long long* cnt_r, uint32_t cap,
                                            unsigned long long* out)
{
    if (BPS == 2 && merge) join_verify_t<BPS, K, EXC == 1 ? 2 : EXC == 2 ? 3 : 1>(myq, from, n, entries, rbuf, tile, etile, cb, ts, off2, tn, nk1, cnt_r, cap, out);
    else if (BPS == 4 && K <= X4_TWO_PER_LANE_MAX_K && merge) join_verify_t<BPS, K, 1>(myq, from, n, entries, rbuf, tile, etile, cb, ts, off2, tn, nk1, cnt_r, cap, out);
    else if (BPS == 4 && merge) {
        join_verify_t<BPS, K, 1, 1>(myq, from, min(n, 64), entries, rbuf, tile, etile, cb, ts, off2, tn, nk1, cnt_r, cap, out);
        if (n > 64) join_verify_t<BPS, K, 1, 1>(myq, from + 64, n - 64, entries, rbuf, tile, etile, cb, ts, off2, tn, nk1, cnt_r, cap, out);
    } else join_verify_t<BPS, K, 0>(myq, from, n, entries, rbuf, tile, etile, cb, ts, off2, tn, nk1, cnt_r, cap, out);
}

// Table build (both planes): a thread hashes 16 CONSECUTIVE positions out
// of one register window - forward key by v_alignbit at a constant shift, the reverse complement slides by one symbol -
// instead of extracting and reverse-complementing a key from scratch per position (about 50 instructions per visit: the
// build was vector-issue-bound like the probe, 12 us of a 150 us task).  `f(t4, hx[4], valid[4])` is called for every four
// positions with their hashes, so that the LDS reads / returning atomics of four positions are in flight together.
template <int BPS, int K, bool AEXC, typename F>
__device__ __forceinline__ void build_walk16(const uint32_t* tile, const uint32_t* etile, int p0, int tn, F&& f)
{
    using KT = KeyT<BPS, K>;
    // AEXC: a position whose k-mer covers an exception symbol is not valid - 56 exception bits from p0 on decide all sixteen
    unsigned long long EX = 0ULL;
    if (AEXC) {
        const uint32_t wi = (uint32_t)p0 >> 5, sh = (uint32_t)p0 & 31u;
        const uint32_t w0 = etile[wi], w1 = etile[wi + 1], w2 = etile[wi + 2];
        EX = ((unsigned long long)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
    }
    constexpr unsigned long long KM = (K >= 64) ? ~0ULL : ((1ULL << K) - 1ULL);
    constexpr int NWIN = ((15 + K) * BPS + 31) / 32;
    uint32_t W[NWIN + 1];
#pragma unroll
    for (int x = 0; x < NWIN; ++x) W[x] = tile[(p0 >> 4) * (BPS / 2) + x];     // p0 is a multiple of 16
    W[NWIN] = 0u;
    // 4-bit planes: a k-mer that holds a symbol which matches nothing (code 15: a character outside the IUPAC alphabet) is
    // left out of the table.  Such symbols are all but absent, so the window is tested once and the keys one by one only
    // when it holds one.
    bool win_invalid = false;
    if (BPS == 4) {
        uint32_t any = 0;
#pragma unroll
        for (int x = 0; x < NWIN; ++x) any |= W[x] & (W[x] >> 1) & (W[x] >> 2) & (W[x] >> 3) & 0x11111111u;
        win_invalid = any != 0u;
    }
    KT kr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (BPS == 4 && g == 2) {                  // positions 8..15 start one word further on
#pragma unroll
            for (int x = 0; x < NWIN; ++x) W[x] = W[x + 1];
        }
        uint32_t hx[4];
        bool valid[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int t = g * 4 + t4;
            const uint32_t sh = (uint32_t)(t * BPS) & 31u;
            constexpr uint32_t SYM = (1u << BPS) - 1u;
            KT kf;
#pragma unroll
            for (int x = 0; x < KT::NW; ++x) kf.w[x] = __builtin_amdgcn_alignbit(W[x + 1], W[x], sh);
            kf.w[KT::NW - 1] &= KT::TOPMASK;
            if (t == 0) {
                kr = revcomp_key<BPS, K>(kf);
            } else {
                const uint32_t sym = (kf.w[KT::NW - 1] >> (KT::TOPBITS - BPS)) & SYM;
                const uint32_t cs = (BPS == 2) ? (sym ^ 3u) : (sym ^ ((sym & 8u) ? 0u : 3u));   // N / n / invalid keep their code
#pragma unroll
                for (int x = KT::NW - 1; x > 0; --x) kr.w[x] = (kr.w[x] << BPS) | (kr.w[x - 1] >> (32 - BPS));
                kr.w[0] = (kr.w[0] << BPS) | cs;
                kr.w[KT::NW - 1] &= KT::TOPMASK;
            }
            hx[t4] = canon_hash<BPS, K>(kf, kr);
            valid[t4] = (p0 + t < tn) & (!AEXC || ((EX >> t) & KM) == 0ULL);
            if (BPS == 4 && win_invalid) valid[t4] = valid[t4] && !key_has_invalid<BPS, K>(kf);
        }
        f(g * 4, hx, valid);
    }
}

// ---- shared joins: what remap_kernel reads (described with the kernel below) ----
struct DMap {          // 16 B (host side: the interval maps of a (window, k) group before they are cut into the table below)
    int32_t lo, hi;    // k-mer starts of the shared sequence, inclusive
    int32_t base;      // position in the target at e == lo
    uint16_t flip;     // 1: reverse-complemented slice (j decreases with e, strands swap)
    uint16_t slot;     // which target of the share
};
// What the kernel reads is the same maps cut at each other's ends: boundaries B[0] = 0 < B[1] < ... < B[n_iv] over the k-mer
// starts of the shared sequence, and per elementary interval [B[t], B[t+1]) what a dot inside it becomes - for every target
// slot up to two ops (a tandem duplication's repeated stretch lies twice in its allele), each one word:
//     bit 0 valid, bit 1 flip, bits 2.. delta (signed):   j = e + delta,  or  j = delta - e with the strands swapped.
// A record looks its interval up once (binary search) and is then copied, shifted, under the ops of that interval; only a run
// that crosses a boundary is cut, interval by interval.
constexpr int REMAP_MAX_IV = 48;       // elementary intervals per share (the host shares no group with more)
constexpr int REMAP_PER = 4;           // records per thread and round
constexpr int REMAP_OPS = 8;           // op words per interval: 4 target slots x 2 copies
struct DShare {        // 32 B
    int32_t dpair;     // the (read, T) pair the join ran
    int32_t iv_first;  // first word of this group's table in the maps buffer: B[0 .. n_iv], then n_iv x REMAP_OPS op words
    int32_t n_iv;
    int32_t target[4]; // pair index per slot, -1: this read has no pair against that allele
    int32_t pad;
};

struct DServe {        // 32 B per pair: what the clean workgroup of a pair served by a shared join needs to cut its records out
    int64_t hit_off;   // of the shared dot plot: the (read, T) pair's record slot ...
    uint32_t cap;
    int32_t dpair;     // ... its index (-1: this pair ran a join of its own),
    int32_t iv_first, n_iv;   // the group's table
    int32_t slot;      // and this pair's slot in it
    int32_t pad;
};

// EXC (2-bit planes; the host groups the pairs): 1 - the launch holds the pairs whose ALLELE has symbols outside upper-case
// ACGT: the table leaves out the k-mers that cover one, runs end before them.  2 - the pairs whose READ has such symbols (and
// whose allele has none): the positions whose k-mer covers one are masked out of the lookup, runs end before them; the
// exception bits of a wave's strip lie in its slice of the (otherwise unused) etile region.  0 - neither: the plain launch
// carries none of that code.  Both sides with such symbols are joined on the 4-bit planes.
constexpr int REXC_WORDS = (JCHUNK + 64 + 64) / 32;     // exception words per strip: its positions, the k-mer, 32 symbols of run
template <typename C, int BPS, int K, int EXC>
__global__ __launch_bounds__(C::THREADS, C::WPS) void join_kernel(
    const SeqDesc* __restrict__ seqs, const uint32_t* __restrict__ p2, const uint32_t* __restrict__ e1,
    const uint32_t* __restrict__ x4, const DPair* __restrict__ pairs, const DTask* __restrict__ tasks,
    const int32_t* __restrict__ task_pairs, unsigned long long* __restrict__ hits, unsigned long long* __restrict__ n_hits,
    unsigned int* __restrict__ reset2)
{
    using KT = KeyT<BPS, K>;
    constexpr int TA = tile_pos<C, BPS>();
    // the first join launch of a run clears the two counters of the clean kernels (overflowed pairs, length of
    // the big-pair list), which saves the run a memset; every count of a pair is stored whole at the end, so
    // n_hits needs no clearing at all
    if (reset2 && blockIdx.x == 0 && threadIdx.x < 2) reset2[threadIdx.x] = 0u;
    constexpr int JQCAP = C::QCAP;
    constexpr int JOIN_THREADS = C::THREADS, JOIN_WAVES = C::THREADS / 64;
    constexpr int JNB_LOG2 = C::NB_LOG2, JNB = 1 << C::NB_LOG2;
    // statically sized: the compiler then knows every table's LDS address and folds it into the offset field of
    // the ds_ instructions instead of adding a base per access
    __shared__ __attribute__((aligned(16))) uint32_t lds[join_lds_bytes<C, BPS>() / sizeof(uint32_t) + 1];
    // (filter first: both tables of the position loop then lie within the 64 KB an LDS instruction's offset field
    // reaches and are addressed without adding a base.  Moving the tile, the strips and the queues below 64 KB as well
    // gains nothing: their reads are ds_read2_b32, whose offsets reach 1 KB)
    uint32_t* filt = lds;                                                     // 2^FILT_LOG2 bits
    uint32_t* start32 = filt + (1 << C::FILT_LOG2) / 32;                     // JNB/2 + 2 words: u16 pairs
    uint32_t* tile = start32 + (JNB / 2 + 2);
    uint32_t* etile = tile + tile_words<C, BPS>();
    // (the 64-bit counters need 8-byte alignment whatever the sizes before them add up to)
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(
        lds + (((JNB / 2 + 2) + (1 << C::FILT_LOG2) / 32 + tile_words<C, BPS>() + etile_words<C, BPS>() + 1) & ~1));
    uint32_t* queue = reinterpret_cast<uint32_t*>(cnt + MAX_READS_PER_TASK);   // JOIN_WAVES * JQCAP
    uint32_t* rbufs = queue + JOIN_WAVES * (JQCAP + 4);                        // JOIN_WAVES * rbuf_words
    uint32_t* wtot = rbufs + JOIN_WAVES * rbuf_words<BPS>();                   // 2 * JOIN_WAVES + 4
    int* cstart = reinterpret_cast<int*>(wtot + 2 * JOIN_WAVES + 4);              // MAX_READS_PER_TASK + 2: strip prefix of the group's reads
    // what the probe needs of every read of the task: pair index, first plane chunk (bit 31: the read has symbols
    // outside upper-case ACGT), length, allele.  Fetched once per task by one wave (three dependent global loads for all
    // reads at once); the group boundaries, the strip prefix and every strip's set-up then come out of LDS.  (Before, the
    // group loop and the strip prefix walked the reads one by one through chains of dependent scalar loads - about
    // 17 us per table build - and every strip started with a chain of three.)
    uint32_t* rinfo = reinterpret_cast<uint32_t*>(cstart + MAX_READS_PER_TASK + 2);    // 4 * MAX_READS_PER_TASK
    uint16_t* entries = reinterpret_cast<uint16_t*>(rinfo + 4 * MAX_READS_PER_TASK);    // TA
    const uint16_t* start16 = reinterpret_cast<const uint16_t*>(start32);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DTask task = tasks[blockIdx.x];          // first = index into task_pairs, n_reads = pairs in the range
    const uint32_t* plane = (BPS == 2) ? p2 : x4;
    constexpr int WPC = (BPS == 2) ? VP_P2_WORDS_PER_CHUNK : VP_X4_WORDS_PER_CHUNK;
    uint32_t* myq = queue + wave * (JQCAP + 4);    // JQCAP slots + dump slots for lanes that have nothing to store (and the surplus of the last lanes)
    uint32_t* rbuf = rbufs + wave * rbuf_words<BPS>();
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    const uint32_t qbase = (uint32_t)(uintptr_t)(lds_u32_t*)myq;            // LDS byte address of the wave's queue

    if (tid < MAX_READS_PER_TASK) cnt[tid] = 0ULL;
    JoinClock pc;
    const bool pw = lane == 0;                     // one stamping lane per wave
#if defined(VAPOR_PHASE_TIMING) || defined(VAPOR_BLOCK_TIMING)
    const long long t_block0 = wall_clock64();     // constant-rate clock: comparable between CUs
#endif

    if (wave == 0 && lane < task.n_reads) {
        const int idx = task_pairs[task.first + lane];
        const DPair pp = pairs[idx];
        const SeqDesc sd = seqs[pp.seq1];
        rinfo[4 * lane + 0] = (uint32_t)idx;
        rinfo[4 * lane + 1] = sd.chunk0 | ((BPS == 2 && sd.n_exc > 0) ? 0x80000000u : 0u);
        rinfo[4 * lane + 2] = (uint32_t)sd.len;
        rinfo[4 * lane + 3] = (uint32_t)pp.seq2;
    }
    __syncthreads();

    int g0 = 0;
    while (g0 < task.n_reads) {
        // group of consecutive pairs that share the allele: up to the first pair whose allele differs
        const int seq2 = __builtin_amdgcn_readfirstlane((int)rinfo[4 * g0 + 3]);
        const unsigned long long differs =
            __ballot((uint32_t)(lane >= g0) & ((uint32_t)(lane >= task.n_reads) | (uint32_t)((int)rinfo[4 * lane + 3] != seq2)));
        const int g1 = differs ? (int)__builtin_ctzll(differs) : task.n_reads;
        const SeqDesc s2 = seqs[seq2];
        const int nk2 = s2.len - K + 1;
        constexpr bool exc2 = (BPS == 2) && EXC == 1;

        for (int ts = 0; ts < nk2; ts += TA) {
            const int tn = min(TA, nk2 - ts);
            __syncthreads();                       // previous table fully probed
            for (int x = tid; x < JNB / 2 + 2 + (1 << C::FILT_LOG2) / 32; x += JOIN_THREADS) lds[x] = 0u;   // filter and counters
            {
                const uint32_t* src = plane + (size_t)s2.chunk0 * WPC + (((size_t)ts * BPS) >> 5);
                const int nw = ((tn + K - 1) * BPS + 31) / 32 + 2;
                for (int x = tid; x < nw; x += JOIN_THREADS) tile[x] = src[x];
                if (exc2) {
                    const uint32_t* es = e1 + (size_t)s2.chunk0 + (ts >> 5);
                    const int ne = (tn + K - 1 + 31) / 32 + 3;
                    for (int x = tid; x < ne; x += JOIN_THREADS) etile[x] = es[x];
                }
            }
            __syncthreads();
            // ---- build 1/3: bucket sizes (two 16-bit counters per LDS word) --------------------
            // (the 2-bit planes, and the 4-bit planes up to window size 20: with the five-word keys of 30 and 40 the sixteen unrolled
            // positions of build_walk16 take 125 registers and spill; those two build key by key)
            constexpr bool plain2 = (BPS == 2 || K <= 20);           // (both planes, with or without exception symbols in the allele: see build_walk16)
            if (plain2) {
                for (int p0 = tid * 16; p0 < tn; p0 += 16 * JOIN_THREADS)
                    build_walk16<BPS, K, exc2>(tile, etile, p0, tn, [&](int, const uint32_t (&hx)[4], const bool (&valid)[4]) {
                        uint32_t fw[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) fw[u] = filt[hx[u] >> (32 - C::FILT_LOG2 + 5)];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (!valid[u]) continue;
                            const uint32_t h = hx[u] >> (32 - JNB_LOG2), fb = hx[u] >> (32 - C::FILT_LOG2);
                            atomicAdd(&start32[h >> 1], 1u << ((h & 1u) * 16));
                            const uint32_t fm = filt_mask(hx[u], fb);
                            if ((fw[u] & fm) != fm) atomicOr(&filt[fb >> 5], fm);
                        }
                    });
            } else
            for (int p = tid; p < tn; p += JOIN_THREADS) {
                KT key = extract_key<BPS, K>(tile, (uint32_t)p);
                bool ok;
                if (BPS == 2) ok = !(exc2 && any_exc<K>(etile, (uint32_t)p));
                else ok = !key_has_invalid<BPS, K>(key);
                if (ok) {
                    KT rc = revcomp_key<BPS, K>(key);
                    const uint32_t hx = canon_hash<BPS, K>(key, rc);
                    const uint32_t h = hx >> (32 - JNB_LOG2), fb = hx >> (32 - C::FILT_LOG2);
                    atomicAdd(&start32[h >> 1], 1u << ((h & 1u) * 16));
                    const uint32_t fm = filt_mask(hx, fb);
                    if ((filt[fb >> 5] & fm) != fm) atomicOr(&filt[fb >> 5], fm);
                }
            }
            __syncthreads();
            // ---- build 2/3: inclusive prefix -> end of every bucket ---------------------------
            {
                constexpr int WPT = (JNB / 2 + JOIN_THREADS - 1) / JOIN_THREADS;   // words per thread (16)
                constexpr bool WHOLE = (JNB / 2) % JOIN_THREADS == 0;
                uint32_t local = 0;
#pragma unroll
                for (int x = 0; x < WPT; ++x) {
                    uint32_t w = (WHOLE || tid * WPT + x < JNB / 2) ? start32[tid * WPT + x] : 0u;
                    local += (w & 0xFFFFu) + (w >> 16);
                }
                uint32_t incl = wave_incl_scan_u32(local);
                if (lane == 63) wtot[wave] = incl;
                __syncthreads();
                uint32_t run = incl - local;
                for (int q = 0; q < wave; ++q) run += wtot[q];
#pragma unroll
                for (int x = 0; x < WPT; ++x) {
                    if (!WHOLE && tid * WPT + x >= JNB / 2) break;
                    uint32_t w = start32[tid * WPT + x];
                    uint32_t lo = run + (w & 0xFFFFu);
                    uint32_t hi = lo + (w >> 16);
                    start32[tid * WPT + x] = lo | (hi << 16);
                    run = hi;
                }
                if (tid == JOIN_THREADS - 1) start32[JNB / 2] = run;     // start[JNB] = number of entries
            }
            __syncthreads();
            // ---- build 3/3: fill every bucket from its end; the counter ends as the bucket start
            if (plain2) {
                for (int p0 = tid * 16; p0 < tn; p0 += 16 * JOIN_THREADS)
                    build_walk16<BPS, K, exc2>(tile, etile, p0, tn, [&](int t0, const uint32_t (&hx)[4], const bool (&valid)[4]) {
                        uint32_t old[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t h = hx[u] >> (32 - JNB_LOG2);
                            old[u] = valid[u] ? atomicSub(&start32[h >> 1], 1u << ((h & 1u) * 16)) : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t h = hx[u] >> (32 - JNB_LOG2);
                            if (valid[u]) entries[((old[u] >> ((h & 1u) * 16)) & 0xFFFFu) - 1u] = (uint16_t)(p0 + t0 + u);
                        }
                    });
            } else
            for (int p = tid; p < tn; p += JOIN_THREADS) {
                KT key = extract_key<BPS, K>(tile, (uint32_t)p);
                bool ok;
                if (BPS == 2) ok = !(exc2 && any_exc<K>(etile, (uint32_t)p));
                else ok = !key_has_invalid<BPS, K>(key);
                if (ok) {
                    KT rc = revcomp_key<BPS, K>(key);
                    const uint32_t h = canon_hash<BPS, K>(key, rc) >> (32 - JNB_LOG2);
                    const uint32_t sh = (h & 1u) * 16;
                    uint32_t old = atomicSub(&start32[h >> 1], 1u << sh);
                    entries[((old >> sh) & 0xFFFFu) - 1u] = (uint16_t)p;
                }
            }
            __syncthreads();
            // ---- probe: every read of the group.  The 1024-position strips of ALL reads of the group are
            // dealt round-robin to the waves (a 10 kb read alone has only 10 strips for 16 waves).
            if (wave == 0) {
                // strip prefix of the group's reads: one lane per read, a wave prefix sum
                const int t = g0 + lane;
                int strips = 0;
                if (t < g1) {
                    const int nk = (int)rinfo[4 * t + 2] - K + 1;
                    strips = nk > 0 ? (nk + JCHUNK - 1) / JCHUNK : 0;
                }
                const int incl = (int)wave_incl_scan_u32((uint32_t)strips);
                if (t < g1) cstart[lane] = incl - strips;
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (lane == 0) {
                    cstart[g1 - g0] = total;
                    wtot[2 * JOIN_WAVES] = 0u;             // next strip to hand out
                }
            }
            __syncthreads();
            pc.mark(0, pw);                        // table build (incl. waiting for the slowest wave of the last probe)
            const int total_strips = cstart[g1 - g0];
            {
                constexpr int NWIN = ((15 + K) * BPS + 31) / 32;
                constexpr int WPL = BPS / 2;                      // plane words per 16 positions
                constexpr uint32_t SYM = (1u << BPS) - 1u;
                int r = g0;                                        // strips are visited in increasing order
                // strips differ in cost (dots per strip), so the waves take the next strip when they are free
                for (;;) {
                    uint32_t next = 0;
                    if (lane == 0) next = atomicAdd(&wtot[2 * JOIN_WAVES], 1u);
                    const int si = __builtin_amdgcn_readfirstlane((int)next);
                    if (si >= total_strips) break;
                    while (si >= cstart[r - g0 + 1]) ++r;
                    const int cb = (si - cstart[r - g0]) * JCHUNK;
                    const uint32_t ri_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)rinfo[4 * r + 0]);
                    const uint32_t ri_chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)rinfo[4 * r + 1]);
                    const int nk1 = __builtin_amdgcn_readfirstlane((int)rinfo[4 * r + 2]) - K + 1;
                    const DPair pr = pairs[ri_idx];                // one load; the strip's words below do not wait for it
                    const uint32_t chunk1 = ri_chunk & 0x7FFFFFFFu;
                    const uint32_t* rplane = plane + (size_t)chunk1 * WPC;
                    const uint32_t* re = e1 + (size_t)chunk1;
                    // (2-bit planes: a read with symbols outside upper-case ACGT is in a launch of its own, EXC == 2)
                    constexpr bool exc1 = (BPS == 2) && EXC == 2;
                    unsigned long long* out = hits + pr.hit_off;
                    const bool merge = (BPS == 2) ? true : (K <= X4_MERGE_MAX_K && s2.n_invalid == 0);
                    static_assert(BPS != 2 || (C::THREADS / 64) * REXC_WORDS <= etile_words<C, 2>(), "strip exception bits share the etile region");
                    const uint32_t* vex = (EXC == 2) ? etile + wave * REXC_WORDS : etile;     // what the verification reads as exception bits
                    if (exc1) {
                        // the strip's exception bits (words inside the read's own chunks and padding only)
                        const int have = ((nk1 + K - 1 + 31) >> 5) + VP_PAD_CHUNKS - (cb >> 5);
                        for (int x = lane; x < REXC_WORDS; x += 64) (etile + wave * REXC_WORDS)[x] = x < have ? re[(cb >> 5) + x] : 0u;
                    }
                    // stage this wave's strip of the read: positions cb .. cb+1023 (+ K-1 lookahead)
                    {
                        const int nw = min(rbuf_words<BPS>(), (int)((((size_t)(nk1 + K - 1 - cb)) * BPS + 31) >> 5) + 2);
                        const uint32_t* src = rplane + (((size_t)cb * BPS) >> 5);
                        for (int x = lane; x < rbuf_words<BPS>(); x += 64) rbuf[x] = x < nw ? src[x] : 0u;
                    }
                    // which of the lane's 16 positions are looked up at all: those inside the read and, for a read with
                    // symbols outside upper-case ACGT, those whose k-mer holds none (decided once per strip, so that the
                    // position loop carries one mask bit instead of the 64-bit exception window)
                    const int i0 = cb + 16 * lane;
                    uint32_t vmask = (1u << min(max(nk1 - i0, 0), 16)) - 1u;
                    if (exc1 && i0 < nk1) {
                        const uint32_t wi = (uint32_t)i0 >> 5, sh = (uint32_t)i0 & 31u;
                        const uint32_t e0 = re[wi], e1w = re[wi + 1], e2w = re[wi + 2];
                        const unsigned long long EE = ((unsigned long long)__builtin_amdgcn_alignbit(e2w, e1w, sh) << 32) |
                                                      __builtin_amdgcn_alignbit(e1w, e0, sh);   // exception bits of i0 .. i0+63
                        constexpr unsigned long long KM = (1ULL << K) - 1ULL;
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            if (((EE >> t) & KM) != 0ULL) vmask &= ~(1u << t);
                    }
                    int qlen = 0;                                  // queued candidates (wave-uniform)
                    // lane owns the 16 consecutive positions i0 .. i0+15: its window of the strip
                    uint32_t W[NWIN + 1];
#pragma unroll
                    for (int x = 0; x < NWIN; ++x) W[x] = rbuf[lane * WPL + x];
                    W[NWIN] = 0u;
                    KT kr;
                    pc.mark(1, pw);                        // strip staging
                    for (int g = 0; g < 4; ++g) {
                        if (BPS == 4 && g == 2) {                  // positions 8..15 start one word further on
#pragma unroll
                            for (int x = 0; x < NWIN; ++x) W[x] = W[x + 1];
                        }
                        // ---- keys and bucket bounds of four positions (their LDS reads are in flight together)
                        uint32_t sc[4];                            // first slot | bucket size << 16
                        const uint32_t vm4 = vmask >> (4 * g);
#pragma unroll
                        for (int t4 = 0; t4 < 4; ++t4) {
                            const int t = g * 4 + t4;
                            const uint32_t sh = (uint32_t)(t * BPS) & 31u;
                            KT kf;
#pragma unroll
                            for (int x = 0; x < KT::NW; ++x) kf.w[x] = __builtin_amdgcn_alignbit(W[x + 1], W[x], sh);
                            kf.w[KT::NW - 1] &= KT::TOPMASK;
                            if (t == 0) {
                                kr = revcomp_key<BPS, K>(kf);
                            } else {
                                // slide the reverse complement: drop its last symbol, prepend comp(new symbol)
                                uint32_t sym = (kf.w[KT::NW - 1] >> (KT::TOPBITS - BPS)) & SYM;
                                uint32_t cs = (BPS == 2) ? (sym ^ 3u) : (sym ^ ((sym & 8u) ? 0u : 3u));
#pragma unroll
                                for (int x = KT::NW - 1; x > 0; --x)
                                    kr.w[x] = (kr.w[x] << BPS) | (kr.w[x - 1] >> (32 - BPS));
                                kr.w[0] = (kr.w[0] << BPS) | cs;
                                kr.w[KT::NW - 1] &= KT::TOPMASK;
                            }
                            const uint32_t hx = canon_hash<BPS, K>(kf, kr);
                            const uint32_t h = hx >> (32 - JNB_LOG2), fb = hx >> (32 - C::FILT_LOG2);
                            // unconditional reads and an arithmetic mask: a predicated read would put a wait
                            // inside every position's own branch and serialise the four lookups.  The filter bit
                            // says whether any allele k-mer shares these 17 hash bits: without it two in three
                            // candidates are mere bucket mates of nothing.
                            // (bitwise, not `&&`: with a short-circuit the compiler sinks the filter read into the branch on
                            // `valid` and waits for it there, one LDS round trip per position instead of one per four)
                            const uint32_t s0 = start16[h], s1v = start16[h + 1], fw = filt[fb >> 5];
                            // (bit 0 spread over the word by one v_bfe_i32; written as `0 - (x & 1)` the compiler turns
                            // it into and + compare + select, two of which cannot share an issue slot; the same happens to the builtin)
                            uint32_t take;
                            asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(take) : "v"((vm4 >> t4) & (fw >> (fb & 31u)) & (fw >> ((hx >> 10) & 31u))));
                            sc[t4] = (s0 | ((s1v - s0) << 16)) & take;
                        }
                        pc.mark(2, pw);                    // keys + bucket bounds issued
                        // ---- four wave prefix sums of the bucket sizes, interleaved -------------------------
                        // Bucket sizes are tiny almost always, so the four sums share one register (8-bit fields: every
                        // size below 4, totals below 256) or two (16-bit fields: sizes below 512); four separate
                        // scans only where a repeat makes a bucket large.
                        uint32_t incl[4], tots[4];
                        {
                            const uint32_t c0 = sc[0] >> 16, c1 = sc[1] >> 16, c2 = sc[2] >> 16, c3 = sc[3] >> 16;
                            const uint32_t big = c0 | c1 | c2 | c3;
                            // (the totals of the four positions leave the vector unit together: one v_readlane of the packed
                            // sums, the fields are taken apart by the scalar unit.  The packed scan runs unconditionally and
                            // the rare wider cases overwrite its results: as three exclusive branches the totals were
                            // undefined on some edge of the structurised flow, which cost eight v_readfirstlane per group)
                            {
                                const uint32_t p = wave_scan<OpAdd>(c0 | (c1 << 8) | (c2 << 16) | (c3 << 24), 0u);
                                incl[0] = p & 0xFFu; incl[1] = (p >> 8) & 0xFFu; incl[2] = (p >> 16) & 0xFFu; incl[3] = p >> 24;
                                const uint32_t p63 = __builtin_amdgcn_readlane(p, 63);
                                tots[0] = p63 & 0xFFu; tots[1] = (p63 >> 8) & 0xFFu; tots[2] = (p63 >> 16) & 0xFFu; tots[3] = p63 >> 24;
                            }
                            if (__ballot(big >= 4u)) {
                                if (!__ballot(big >= 512u)) {
                                    const uint32_t p = wave_scan<OpAdd>(c0 | (c1 << 16), 0u), q = wave_scan<OpAdd>(c2 | (c3 << 16), 0u);
                                    incl[0] = p & 0xFFFFu; incl[1] = p >> 16; incl[2] = q & 0xFFFFu; incl[3] = q >> 16;
                                    const uint32_t p63 = __builtin_amdgcn_readlane(p, 63), q63 = __builtin_amdgcn_readlane(q, 63);
                                    tots[0] = p63 & 0xFFFFu; tots[1] = p63 >> 16; tots[2] = q63 & 0xFFFFu; tots[3] = q63 >> 16;
                                } else {
                                    incl[0] = c0; incl[1] = c1; incl[2] = c2; incl[3] = c3;
#pragma unroll
                                    for (int x = 0; x < 4; ++x) {
                                        incl[x] = wave_scan<OpAdd>(incl[x], 0u);
                                        tots[x] = __builtin_amdgcn_readlane(incl[x], 63);
                                    }
                                }
                            }
                        }
                        pc.mark(3, pw);                    // scans (waits for the bucket reads)
                        // ---- fill the queue position by position; verify 128 candidates whenever they are there
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const uint32_t c = sc[x] >> 16, s0 = sc[x] & 0xFFFFu;
                            const uint32_t tot = tots[x];
                            pc.count(7, tot);                  // candidates
                            if (tot == 0) continue;
                            const uint32_t il = (uint32_t)(16 * lane + g * 4 + x);
                            if (tot > (uint32_t)(JQCAP - VAPOR_JQ_FAST_SLACK)) {
                                // more candidates than the queue can always take (long repeats): level by level.
                                // (tot <= JQCAP - 128 keeps the invariant of the fast path: qlen <= 127 on entry,
                                // at most 255 after the stores, at most 127 again after the one verification)
                                for (uint32_t u = 0;; ++u) {
                                    const bool act = c > u;
                                    const unsigned long long m = __ballot(act);
                                    if (!m) break;
                                    if (act) {
                                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                                        myq[qlen + (int)rank] = (il << 16) | (s0 + u);
                                    }
                                    qlen += __popcll(m);
                                    if (qlen >= 128) {
                                        join_verify<BPS, K, EXC>(myq, qlen - 128, 128, entries, rbuf, tile, vex, cb, ts, pr.off2, tn,
                                                            nk1, merge, &cnt[r], pr.cap, out);
                                        qlen -= 128;
                                    }
                                }
                                continue;
                            }
                            // qlen <= 127 here, so qlen + tot <= JQCAP - 1
                            {
                                // buckets hold one or two entries almost always: two predicated stores, a loop
                                // only for the rest
                                // three unconditional stores (a lane without that candidate writes the dump slot:
                                // no execution-mask juggling, no branch), a uniform loop only for buckets of four
                                // entries and more
                                // (byte addresses, so that a store is one compare and one select between the lane's slot and the
                                // dump slot, with the +1 / +2 in the instruction's offset field)
                                const uint32_t pos = (uint32_t)qlen + incl[x] - c;
                                const uint32_t item = (il << 16) | s0;
                                const uint32_t full = qbase + pos * 4u;
                                // One select for the three stores: a lane without a candidate writes the three dump slots, a lane
                                // with one or two writes its surplus onto the slots of the lanes after it - whose own stores of a
                                // LOWER index come later in program order (the wave's LDS instructions execute in order), so the
                                // stores go out from index 2 down to 0 and every slot ends with its owner's value.
                                volatile lds_u32_t* const qp = (volatile lds_u32_t*)(uintptr_t)(c > 0u ? full : qbase + (uint32_t)JQCAP * 4u);
                                qp[2] = item + 2u;
                                qp[1] = item + 1u;
                                qp[0] = item;
                                for (uint32_t u = 3; __ballot(c > u); ++u)
                                    if (c > u) myq[pos + u] = item + u;
                            }
                            qlen += (int)tot;
                            if (qlen >= 128) {
                                pc.mark(4, pw);            // queue fill
                                join_verify<BPS, K, EXC>(myq, qlen - 128, 128, entries, rbuf, tile, vex, cb, ts, pr.off2, tn, nk1,
                                                    merge, &cnt[r], pr.cap, out);
                                qlen -= 128;
                                pc.mark(5, pw);            // verify + store
                            }
                        }
                        pc.mark(4, pw);
                    }
                    // the strip changes: drain
                    if (qlen > 0)
                        join_verify<BPS, K, EXC>(myq, 0, qlen, entries, rbuf, tile, vex, cb, ts, pr.off2, tn, nk1, merge, &cnt[r], pr.cap, out);
                    pc.mark(5, pw);
                }
            }
        }
        g0 = g1;
    }
    __syncthreads();
    pc.mark(6, pw);                                // waiting for the block's last wave
    pc.flush(pw);
#if defined(VAPOR_PHASE_TIMING) || defined(VAPOR_BLOCK_TIMING)
    if (tid == 0 && blockIdx.x < 4096) {
        const long long t_end = wall_clock64();
        g_block_ticks[blockIdx.x] = (unsigned long long)(t_end - t_block0);
        g_block_info[blockIdx.x * 4 + 0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        g_block_info[blockIdx.x * 4 + 1] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        g_block_info[blockIdx.x * 4 + 2] = (unsigned long long)t_block0;
        g_block_info[blockIdx.x * 4 + 3] = (unsigned long long)t_end;
    }
#endif
    if (tid < task.n_reads) n_hits[task_pairs[task.first + tid]] = cnt[tid];
}

// ------------------------------------------------------------------------------------------
// clean kernels
// ------------------------------------------------------------------------------------------
// wave reductions: every lane gets the result
__device__ __forceinline__ int wave_sum_i32(int v)
{
    return __builtin_amdgcn_readlane(wave_scan<OpAdd>(v, 0), 63);
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    return __builtin_amdgcn_readlane(wave_scan<OpMin>(v, 0x7FFFFFFF), 63);
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    return __builtin_amdgcn_readlane(wave_scan<OpMax>(v, (int)0x80000000), 63);
}

struct CleanShared {
    int min_j, max_j, n_diag, n_lower;
    int c1_kept, c2_kept, c2_count10, c2_kept_diag;
    unsigned long long c1_sum_abs;
    unsigned int n_groups, max_group;
    unsigned int wave_tot[CLEAN_WAVES];
    // directed-distance scratch (dis_to_diagnal_most_abundant_defined)
    int cnt1[11], cnt2[11];
    int kd_lo, kd_hi, bin_lo, bin_hi, n_lists, win_w, win_b, win_lo2, win_r2, win_n, c2x, dir_n;
    long long dir_sum2;
};

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t local, CleanShared* sh, uint32_t* total)
{
    const int tid = threadIdx.x;
    uint32_t incl = wave_incl_scan_u32(local);
    if ((tid & 63) == 63) sh->wave_tot[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < CLEAN_WAVES; ++q) {
        uint32_t w = sh->wave_tot[q];
        if (q < (tid >> 6)) base += w;
        tot += w;
    }
    *total = tot;
    return base + incl - local;
}

// The cleaning works on the join's run records (see VREC_*): `len` dots (j0 + s*t, i0 + t), s = +1 or -1.
// Along a record i - j is constant (s = +1) or grows by 2 per dot (s = -1); i + j the other way round.
struct RecV {
    int i0, j0, len;
    bool rc;
};
// clean_kernel works on a private LDS copy of a pair's records whose upper word is laid out for it (LFMT):
//   bits 32..36 len - 1   bit 37 strand   bits 38..50 / 51..63 the record's group on the i - j / i + j axis (filled by the
// group-size pass, read by the flag pass: a pair staged in LDS has at most CLEAN_HCAP_MAX = 8192 records and every group
// holds at least one, so 13 bits do); a single-axis pass uses bits 38.. for its one group.  The lower word is i0 | j0 << 16
// in both formats.
constexpr int CLEAN_HCAP_MAX = 8192;
template <bool LFMT>
__device__ __forceinline__ RecV rec_decode(unsigned long long r)
{
    RecV v;
    v.i0 = VREC_I(r); v.j0 = VREC_J(r);
    // A record of ONE dot has no direction: everything the cleaning computes of it - i - j, i + j, the counts, the closed
    // forms - is the same whichever strand it came from.  The join writes reverse-complement dots as single-dot records
    // (a quarter of a noisy pair's records), so treating those as strandless keeps every wave out of the dot-by-dot
    // branches that reverse-complement RUNS need (the format allows them; callers of vapor_clean_hits may bring them).
    if (LFMT) { v.len = (int)((r >> 32) & 31ull) + 1; v.rc = (bool)((r >> 37) & 1ull); }
    else { v.len = VREC_LEN(r); v.rc = VREC_RC(r) && v.len > 1; }
    return v;
}
__device__ __forceinline__ unsigned long long rec_to_lds(unsigned long long r)
{
    const uint32_t len = (uint32_t)VREC_LEN(r);
    return (r & 0xFFFFFFFFull) | ((unsigned long long)((len - 1u) | ((uint32_t)(VREC_RC(r) && len > 1u) << 5)) << 32);
}
// The flag byte of record h.  Global path (clean_big_kernel): a byte beside the record.  LDS path: bits 38..45 of the
// staged record itself (no flag array: a byte per record less in LDS, and one read gives record and flags), the group
// number of a single-axis pass above it (bits 46..); cluster_dual parks its two group numbers in bits 38..63 while the
// flags are still all zero and its flag pass overwrites them.
struct GFlags {
    uint8_t* p;
    __device__ __forceinline__ uint32_t of(int h, unsigned long long) const { return p[h]; }
    __device__ __forceinline__ void set(int h, unsigned long long, uint32_t f) const { p[h] = (uint8_t)f; }
};
struct LFlags {
    unsigned long long* r;
    __device__ __forceinline__ uint32_t of(int, unsigned long long raw) const { return (uint32_t)(raw >> 38) & 0xFFu; }
    __device__ __forceinline__ void set(int h, unsigned long long raw, uint32_t f) const
    {
        reinterpret_cast<uint32_t*>(r)[2 * h + 1] = ((uint32_t)(raw >> 32) & 63u) | ((f & 0xFFu) << 6);
    }
};

// Group starts of an occupancy bitmap, for the consecutive words [w0, w1) of one thread: a value starts a group when it is
// occupied and none of the 9 values below it is.  "Occupied among the 9 below" = OR of the bitmap shifted up by 1..9 with
// the previous word's top bits coming in; by doubling - x | x<<1 covers shifts {0,1}, | <<2 {0..3}, | <<4 {0..7}; then
// (that << 1) covers {1..8} and (the first << 8) {8,9} - it is nine funnel operations per word instead of eighteen, the
// previous word's partial results being those the thread has just computed (their low bits differ from the carried-in
// truth, their top bits - all a funnel shift takes - do not).
template <int PER_MAX>
__device__ __forceinline__ uint32_t group_starts(const uint32_t* bm, int w0, int w1, uint32_t (&st)[PER_MAX])
{
    uint32_t p = w0 ? bm[w0 - 1] : 0u;
    uint32_t a1p = p | (p << 1), a2p = a1p | (a1p << 2), a3p = a2p | (a2p << 4);
    uint32_t n = 0;
#pragma unroll
    for (int q = 0; q < PER_MAX; ++q) {
        const int w = w0 + q;
        uint32_t s = 0;
        if (w < w1) {
            const uint32_t cur = bm[w];
            const uint32_t a1 = cur | __builtin_amdgcn_alignbit(cur, p, 31);
            const uint32_t a2 = a1 | __builtin_amdgcn_alignbit(a1, a1p, 30);
            const uint32_t a3 = a2 | __builtin_amdgcn_alignbit(a2, a2p, 28);
            const uint32_t below = __builtin_amdgcn_alignbit(a3, a3p, 31) | __builtin_amdgcn_alignbit(a1, a1p, 24);
            s = cur & ~below;
            p = cur; a1p = a1; a2p = a2; a3p = a3;
        }
        st[q] = s;
        n += __popc(s);
    }
    return n;
}

// the reductions over a pair's finished flags (kept counts, sum |j-i|, count10, range of i-j over the C1-kept dots), done by
// the flag pass of the pair's last clustering step; returns the public VAPOR_HF_* bits of the record
struct FinalAcc {
    int k1 = 0, k2 = 0, c10 = 0, kd = 0, dlo = 0x7FFFFFFF, dhi = -0x7FFFFFFF;
    long long sabs = 0;
    __device__ __forceinline__ uint32_t add(uint32_t f, const RecV& r)
    {
        uint32_t pub = f & (HF_C2D | HF_C2A);
        const bool c1k = (f & (WF_D1 | WF_A1)) != 0u, c2k = pub != 0u;
        if (c1k) pub |= HF_C1;
        const int d0 = r.i0 - r.j0;
        if (!r.rc) {
            // i - j is the same for all dots of the record
            const int ad = d0 < 0 ? -d0 : d0;
            if (c1k) { k1 += r.len; sabs += (long long)r.len * ad; dlo = min(dlo, d0); dhi = max(dhi, d0); }
            if (c2k) {
                // dots with j > 0 and 25*|j-i| < 4*j, j = j0 .. j0+len-1:  j >= floor(25*ad/4) + 1
                const int jmin = max(1, (25 * ad) / 4 + 1);
                k2 += r.len;
                c10 += max(0, r.j0 + r.len - max(r.j0, jmin));
            }
        } else if (c1k || c2k) {
            for (int t = 0; t < r.len; ++t) {
                const int j = r.j0 - t, i = r.i0 + t, ad = j > i ? j - i : i - j;
                if (c1k) { ++k1; sabs += ad; }
                if (c2k) { ++k2; c10 += (j > 0 && 25 * ad < 4 * j); }
            }
            if (c1k) { dlo = min(dlo, d0); dhi = max(dhi, d0 + 2 * (r.len - 1)); }
        }
        kd += (f & HF_C2D) ? r.len : 0;
        return pub;
    }
    __device__ __forceinline__ void commit(CleanShared* sh)
    {
        k1 = wave_sum_i32(k1); k2 = wave_sum_i32(k2); c10 = wave_sum_i32(c10); kd = wave_sum_i32(kd);
        sabs = wave_sum_i64(sabs); dlo = wave_min_i32(dlo); dhi = wave_max_i32(dhi);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sh->c1_kept, k1); atomicAdd(&sh->c2_kept, k2); atomicAdd(&sh->c2_count10, c10);
            atomicAdd(&sh->c2_kept_diag, kd); atomicAdd(&sh->c1_sum_abs, (unsigned long long)sabs);
            atomicMin(&sh->kd_lo, dlo); atomicMax(&sh->kd_hi, dhi);
        }
    }
};

// marks the values v0, v0 + 2, ... (len of them; a single value when !strided) in an occupancy bitmap.
// Most values are marked already: look before the atomic (a stale look only costs a redundant atomic).
__device__ __forceinline__ void mark_values(uint32_t* bm, uint32_t v0, int len, bool strided)
{
    const uint32_t w = v0 >> 5, sh = v0 & 31u;
    // (one straight line for single values and strided runs: a wave holds both kinds, and a branch between them made
    // every wave walk both sides)
    const unsigned long long pat = strided ? (0x5555555555555555ull >> (64 - 2 * len)) : 1ull;     // len <= 32
    const uint32_t lo = (uint32_t)pat, hi = (uint32_t)(pat >> 32);
    const uint32_t m0 = lo << sh;
    const uint32_t m1 = sh ? ((lo >> (32u - sh)) | (hi << sh)) : hi;
    const uint32_t m2 = sh ? (hi >> (32u - sh)) : 0u;
    if (m0 & ~bm[w]) atomicOr(&bm[w], m0);
    if (m1 && (m1 & ~bm[w + 1])) atomicOr(&bm[w + 1], m1);
    if (m2 && (m2 & ~bm[w + 2])) atomicOr(&bm[w + 2], m2);
}

// One pass of 1-D gap clustering (dis_cluster SF:551-564 / dis_cluster_2 SF:566-580) over the
// dots of the records selected by (flags & need_clear) == 0, on value v = AXIS_A ? i + j : i - j + vbias.
// Values closer than 10 to their sorted predecessor join its group: an occupancy bitmap over the
// value range, group starts = occupied bins with no occupied bin among the 9 below, gid(v) = rank
// of the group's start.  All dots of a record fall into one group (their values differ by 0 or 2), so
// group sizes are summed per record and, afterwards, set_gt10 is OR-ed into the record's flag byte
// when its group has more than 10 dots, and set_rule when the group passes dis_cluster's rule
// (more than 50 dots, or maximal size when no group has more than 50).
// NARROW (pair with fewer than 65536 dots): group sizes are 16-bit counters, two per LDS word.
// HAVE_BM: the caller has already filled the occupancy bitmap (fused with the staging pass).
// FINAL: this is the last clustering step of the pair; its flag pass also does the reductions over the
// finished flags (kept counts, sum |j-i|, count10, range of i-j over the C1-kept dots) and leaves the
// public VAPOR_HF_* bits in the flag bytes.
template <bool AXIS_A, bool NARROW, bool HAVE_BM, bool FINAL, int PER_MAX, typename HP, typename FP>
__device__ __forceinline__ void cluster_axis(HP recs, FP hflags, int n, int vbias, int range_words, uint32_t* bm,
                                             uint16_t* wrank, uint32_t* gcnt, CleanShared* sh,
                                             uint32_t need_clear, uint32_t set_gt10, uint32_t set_rule, CleanClock& pc, int phase0 = 16)
{
    const int tid = threadIdx.x;
    uint32_t* sb = bm;                               // the start bits replace the occupancy bits in place
    auto gsize = [&](uint32_t g) -> uint32_t {
        return NARROW ? ((gcnt[g >> 1] >> ((g & 1u) * 16)) & 0xFFFFu) : gcnt[g];
    };
    auto first_value = [&](const RecV& r) -> int { return AXIS_A ? (r.i0 + r.j0) : (r.i0 - r.j0 + vbias); };
    auto group_of = [&](int v) -> uint32_t {
        return (uint32_t)wrank[v >> 5] + __popc(sb[v >> 5] & (0xFFFFFFFFu >> (31 - (v & 31)))) - 1u;
    };
    if (tid == 0) sh->max_group = 0;
    if (!HAVE_BM) {
        for (int w = tid; w < range_words; w += CLEAN_THREADS) bm[w] = 0;
        __syncthreads();
        // 1. occupancy bitmap
        for (int h = tid; h < n; h += CLEAN_THREADS) {
            const unsigned long long raw = recs[h];
            if (need_clear && (hflags.of(h, raw) & need_clear)) continue;
            const RecV r = rec_decode<NARROW>(raw);
            mark_values(bm, (uint32_t)first_value(r), r.len, AXIS_A ? !r.rc : r.rc);
        }
    }
    __syncthreads();
    pc.mark(phase0 + 0, tid == 0);
    // 2. group starts and their ranks
    const int per = (range_words + CLEAN_THREADS - 1) / CLEAN_THREADS;
    const int w0 = min(tid * per, range_words), w1 = min(w0 + per, range_words);
    // (PER_MAX: bitmap words per thread the kernel was instantiated for - 4, 8 or 16; the host picks the smallest that
    // covers the batch's value range, so that a 30 kb range does not walk twelve empty predicated iterations)
    uint32_t stv[PER_MAX];
    const uint32_t local = group_starts<PER_MAX>(bm, w0, w1, stv);
    uint32_t ng;
    uint32_t run = block_exclusive_scan(local, sh, &ng);      // (its barrier also ends all reads of bm)
#pragma unroll
    for (int q = 0; q < PER_MAX; ++q) {
        const int w = w0 + q;
        if (w < w1) {
            sb[w] = stv[q];
            wrank[w] = (uint16_t)run;
            run += __popc(stv[q]);
        }
    }
    for (uint32_t g = tid; g < (NARROW ? (ng + 1) / 2 : ng); g += CLEAN_THREADS) gcnt[g] = 0;
    __syncthreads();
    pc.mark(phase0 + 1, tid == 0);
    // 3. group sizes
    for (int h = tid; h < n; h += CLEAN_THREADS) {
        const unsigned long long raw = recs[h];
        if (need_clear && (hflags.of(h, raw) & need_clear)) continue;
        const RecV r = rec_decode<NARROW>(raw);
        const uint32_t g = group_of(first_value(r));
        if (NARROW) {
            atomicAdd(&gcnt[g >> 1], (uint32_t)r.len << ((g & 1u) * 16));
            // (the LDS copy of a record keeps its group above its flags: the flag pass takes it from there
            // instead of ranking the value again - two LDS reads and a popcount per record)
            reinterpret_cast<uint32_t*>(const_cast<unsigned long long*>(&recs[h]))[1] = ((uint32_t)(raw >> 32) & 0x3FFFu) | (g << 14);
        } else {
            atomicAdd(&gcnt[g], (uint32_t)r.len);
        }
    }
    __syncthreads();
    if (set_rule) {
        uint32_t m = 0;
        for (uint32_t g = tid; g < ng; g += CLEAN_THREADS) m = max(m, gsize(g));
        m = (uint32_t)wave_max_i32((int)m);
        if ((tid & 63) == 0) atomicMax(&sh->max_group, m);
        __syncthreads();
    }
    const uint32_t mx = sh->max_group;
    pc.mark(phase0 + 2, tid == 0);
    // 4. flags (and, for the last step of a pair, the reductions over the finished flags)
    FinalAcc acc;
    for (int h = tid; h < n; h += CLEAN_THREADS) {
        const unsigned long long raw = recs[h];
        uint32_t f = hflags.of(h, raw);
        const bool sel = !(need_clear && (f & need_clear));
        if (!(sel || FINAL)) continue;
        const RecV r = rec_decode<NARROW>(raw);
        if (sel) {
            const uint32_t c = gsize(NARROW ? (uint32_t)(raw >> 46) : group_of(first_value(r)));
            if (set_gt10 && c > 10u) f |= set_gt10;
            if (set_rule && ((mx > 50u) ? (c > 50u) : (c == mx))) f |= set_rule;
        }
        if (FINAL) f = acc.add(f, r);
        hflags.set(h, raw, f);
    }
    if (FINAL) acc.commit(sh);
    __syncthreads();
    pc.mark(phase0 + 3, tid == 0);
}

// Both axes of C1 (clean_dotdata_diagnal_and_anti_diagnal, SF:432-448: groups of more than 10 on i - j OR on i + j keep a
// dot) in ONE sweep over the staged records of a pair - and with them the diagonal step of C2 when the pair wants it
// (rule_d = HF_C2D: dis_cluster's rule on the same i - j groups).  The two clusterings are independent, so every phase of
// cluster_axis is done once for both: one pass marks both occupancy bitmaps (the staging pass), one block scan ranks both
// sets of group starts (two 16-bit fields of one word: a pair staged in LDS has at most 8192 groups per axis), one pass
// over the records adds both group sizes and parks both group numbers in the record's upper word, one pass sets the
// flags.  Half the barriers and record decodes of two cluster_axis calls.  LDS copy only (LFMT records).
// The two axes share one region of 16-bit group counters (gcnt, room for gcap of them): the i + j counters follow the
// i - j ones.  A pair with more groups than that on the two axes together (nearly every record a group of its own on both)
// returns false with nothing but the bitmaps changed, and the caller clusters its axes one after the other.
template <bool FINAL, int PER_MAX, typename FP>
__device__ __forceinline__ bool cluster_dual(unsigned long long* recs, FP hflags, int n, int vbias, int range_words,
                                             uint32_t* bmD, uint16_t* wrankD, uint32_t* bmA, uint16_t* wrankA,
                                             uint32_t* gcnt, int gcap, CleanShared* sh, uint32_t rule_d, CleanClock& pc)
{
    const int tid = threadIdx.x;
    if (tid == 0) sh->max_group = 0;
    // (both bitmaps were filled while the records were staged; the caller's barrier ended that pass)
    const int per = (range_words + CLEAN_THREADS - 1) / CLEAN_THREADS;
    const int w0 = min(tid * per, range_words), w1 = min(w0 + per, range_words);
    uint32_t ngD, ngA;
    if constexpr (PER_MAX > 8) {
        // (the largest value ranges: sixteen words per thread and axis - both sets of starts at once would not fit the 64
        // registers of a thread, so the two axes are ranked one after the other; the record passes below stay shared)
        auto rank_axis = [&](uint32_t* bm, uint16_t* wrank, uint32_t& ng) {
            uint32_t st[PER_MAX];
            const uint32_t local = group_starts<PER_MAX>(bm, w0, w1, st);
            uint32_t run = block_exclusive_scan(local, sh, &ng);
#pragma unroll
            for (int q = 0; q < PER_MAX; ++q) {
                const int w = w0 + q;
                if (w < w1) { bm[w] = st[q]; wrank[w] = (uint16_t)run; run += __popc(st[q]); }
            }
        };
        rank_axis(bmD, wrankD, ngD);
        __syncthreads();                           // (block_exclusive_scan's shared totals are read by every thread)
        rank_axis(bmA, wrankA, ngA);
    } else {
        uint32_t stD[PER_MAX], stA[PER_MAX];
        const uint32_t local = group_starts<PER_MAX>(bmD, w0, w1, stD) | (group_starts<PER_MAX>(bmA, w0, w1, stA) << 16);
        uint32_t tot;
        uint32_t run = block_exclusive_scan(local, sh, &tot);      // (its barrier also ends all reads of the bitmaps)
        ngD = tot & 0xFFFFu; ngA = tot >> 16;
#pragma unroll
        for (int q = 0; q < PER_MAX; ++q) {
            const int w = w0 + q;
            if (w < w1) {
                bmD[w] = stD[q]; bmA[w] = stA[q];
                wrankD[w] = (uint16_t)run; wrankA[w] = (uint16_t)(run >> 16);
                run += __popc(stD[q]) | (__popc(stA[q]) << 16);
            }
        }
    }
    if ((int)(ngD + ngA) + 2 > gcap) { __syncthreads(); return false; }      // (uniform: block totals)
    uint32_t* const gcntD = gcnt;
    uint32_t* const gcntA = gcnt + (ngD + 1) / 2;
    for (uint32_t g = tid; g < (ngD + 1) / 2 + (ngA + 1) / 2; g += CLEAN_THREADS) gcnt[g] = 0;
    __syncthreads();
    pc.mark(17, tid == 0);
    // group sizes on both axes
    for (int h = tid; h < n; h += CLEAN_THREADS) {
        const unsigned long long raw = recs[h];
        const RecV r = rec_decode<true>(raw);
        const int vd = r.i0 - r.j0 + vbias, va = r.i0 + r.j0;
        const uint32_t gd = (uint32_t)wrankD[vd >> 5] + __popc(bmD[vd >> 5] & (0xFFFFFFFFu >> (31 - (vd & 31)))) - 1u;
        const uint32_t ga = (uint32_t)wrankA[va >> 5] + __popc(bmA[va >> 5] & (0xFFFFFFFFu >> (31 - (va & 31)))) - 1u;
        atomicAdd(&gcntD[gd >> 1], (uint32_t)r.len << ((gd & 1u) * 16));
        atomicAdd(&gcntA[ga >> 1], (uint32_t)r.len << ((ga & 1u) * 16));
        reinterpret_cast<uint32_t*>(&recs[h])[1] = ((uint32_t)(raw >> 32) & 63u) | (gd << 6) | (ga << 19);
    }
    __syncthreads();
    if (rule_d) {
        uint32_t m = 0;
        for (uint32_t g = tid; g < ngD; g += CLEAN_THREADS) m = max(m, (gcntD[g >> 1] >> ((g & 1u) * 16)) & 0xFFFFu);
        m = (uint32_t)wave_max_i32((int)m);
        if ((tid & 63) == 0) atomicMax(&sh->max_group, m);
        __syncthreads();
    }
    const uint32_t mx = sh->max_group;
    pc.mark(18, tid == 0);
    FinalAcc acc;
    for (int h = tid; h < n; h += CLEAN_THREADS) {
        const unsigned long long raw = recs[h];
        const uint32_t gd = (uint32_t)(raw >> 38) & 0x1FFFu, ga = (uint32_t)(raw >> 51);
        const uint32_t cd = (gcntD[gd >> 1] >> ((gd & 1u) * 16)) & 0xFFFFu, ca = (gcntA[ga >> 1] >> ((ga & 1u) * 16)) & 0xFFFFu;
        uint32_t f = 0;                                // (the flags of a pair are all clear before its first clustering step)
        if (cd > 10u) f |= WF_D1;
        if (ca > 10u) f |= WF_A1;
        if (rule_d && ((mx > 50u) ? (cd > 50u) : (cd == mx))) f |= rule_d;
        if (FINAL) f = acc.add(f, rec_decode<true>(raw));
        hflags.set(h, raw, f);
    }
    if (FINAL) acc.commit(sh);
    __syncthreads();
    pc.mark(19, tid == 0);
    return true;
}

// ------------------------------------------------------------------------------------------
// dis_to_diagnal_most_abundant_defined (SF:582-591) and eu_dis_dir_calcu (SF:718-722) on the
// C1-kept dots of one pair.
//
// number_cluster (SF:1104-1118) puts a value v into list b-1 for the first edge index b in 1..10
// with v < edge[b], edge[b] = min + b*float(max-min)/10.0, and into the 11th list when there is
// none.  With integer data this is exact integer arithmetic: b*(max-min)/10 is either an integer
// (then every float64 step is exact) or at least 0.1 away from one (then rounding cannot move it
// across an integer), so   v < edge[b]  <=>  10*(v-min) < b*(max-min),   i.e.
//     list(v) = max > min ? 10*(v-min) / (max-min) : 10          (integer division).
// The longest list(s) are re-binned the same way over their own min..max; only when exactly one
// sub-list is the longest overall is its median the new intercept c, else c = 0.
// Outputs: c2x = 2*c (c is a multiple of 0.5), and over dots (x, y) = (j + c, i) with
// abs(x-y)/abs(x) > 0.1 (x == 0: y/1 > 0.1) the count and the doubled sum of x - y.
//
// The quotient is taken without an integer division per dot: a = 10*(v-lo) <= 10*131070 < 2^24 is exact in
// float, the product with the rounded reciprocal is within 2e-6 of a/range, so the truncated value is off by
// at most one and one exact integer remainder test repairs it.  lo <= v <= lo + range.
struct R4Div {
    int lo, range;
    float inv;
    __device__ __forceinline__ R4Div(int lo_, int range_) : lo(lo_), range(range_), inv(range_ > 0 ? 1.0f / (float)range_ : 0.0f) {}
    __device__ __forceinline__ int bin(int v) const
    {
        if (range <= 0) return 10;
        const int a = 10 * (v - lo);
        int q = (int)((float)a * inv);
        const int r = a - q * range;
        q += (r >= range) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return q;
    }
};

// number of t in [0, len) with x0 + 2*t < bound
__device__ __forceinline__ int count_below(int x0, int len, int bound)
{
    return min(max((bound - x0 + 1) >> 1, 0), len);
}

// kd_lo / kd_hi (min and max of i-j over the kept dots) must already be in *sh.
// Same-strand records carry one value of i - j for all their dots and add their length wherever a dot
// would add one; reverse-complement records (rare) are walked dot by dot.
// CACHE: the level-1 list of every kept same-strand record is parked in the upper nibble of its flag byte
// (LDS copy only; the write-back masks it off; 15 = walk the record), so the later passes do not divide again.
template <bool CACHE, typename HP, typename FP>
__device__ __forceinline__ void directed_stats(HP recs, FP hflags, int n, uint32_t* counters, CleanShared* sh, CleanClock& pc)
{
    const int tid = threadIdx.x;
    if (tid == 0) { sh->n_lists = 0; sh->c2x = 0; sh->dir_n = 0; sh->dir_sum2 = 0; sh->win_w = -1; }
    if (tid < 11) sh->cnt1[tid] = 0;
    __syncthreads();
    const int lo1 = sh->kd_lo, range1 = sh->kd_hi - sh->kd_lo;
    if (range1 < 0) return;                        // no kept dots (uniform)
    const R4Div d1(lo1, range1);
    // level 1: sizes of the eleven lists
    for (int h = tid; h < n; h += CLEAN_THREADS) {
        const unsigned long long raw = recs[h];
        const uint32_t f = hflags.of(h, raw);
        if (!(f & HF_C1)) continue;
        const RecV r = rec_decode<CACHE>(raw);
        const int d0 = r.i0 - r.j0;
        if (!r.rc) {
            const int b = d1.bin(d0);
            atomicAdd(&sh->cnt1[b], r.len);
            if (CACHE) hflags.set(h, raw, f | ((uint32_t)b << 4));
        } else {
            for (int t = 0; t < r.len; ++t) atomicAdd(&sh->cnt1[d1.bin(d0 + 2 * t)], 1);
            if (CACHE) hflags.set(h, raw, f | 0xF0u);
        }
    }
    __syncthreads();
    pc.mark(32, tid == 0);
    int best1 = 0;
    for (int b = 0; b < 11; ++b) best1 = max(best1, sh->cnt1[b]);
    for (int w = 0; w < 11; ++w) {
        if (sh->cnt1[w] != best1) continue;        // uniform (LDS, no writer inside the loop)
        // min / max of list w, then the sizes of its eleven sub-lists
        if (tid == 0) { sh->bin_lo = 0x7FFFFFFF; sh->bin_hi = -0x7FFFFFFF; }
        if (tid < 11) sh->cnt2[tid] = 0;
        __syncthreads();
        {
            int lo = 0x7FFFFFFF, hi = -0x7FFFFFFF;
            for (int h = tid; h < n; h += CLEAN_THREADS) {
                const unsigned long long raw = recs[h];
                const uint32_t f = hflags.of(h, raw);
                if (!(f & HF_C1)) continue;
                if (CACHE && (f >> 4) != 15u && (int)(f >> 4) != w) continue;
                const RecV r = rec_decode<CACHE>(raw);
                const int d0 = r.i0 - r.j0;
                if (!r.rc) {
                    if (CACHE || d1.bin(d0) == w) { lo = min(lo, d0); hi = max(hi, d0); }
                } else {
                    for (int t = 0; t < r.len; ++t) {
                        const int d = d0 + 2 * t;
                        if (d1.bin(d) == w) { lo = min(lo, d); hi = max(hi, d); }
                    }
                }
            }
            lo = wave_min_i32(lo); hi = wave_max_i32(hi);
            if ((tid & 63) == 0) { atomicMin(&sh->bin_lo, lo); atomicMax(&sh->bin_hi, hi); }
        }
        __syncthreads();
        const int lo2 = sh->bin_lo, range2 = sh->bin_hi - sh->bin_lo;
        const R4Div d2(lo2, range2);
        for (int h = tid; h < n; h += CLEAN_THREADS) {
            const unsigned long long raw = recs[h];
            const uint32_t f = hflags.of(h, raw);
            if (!(f & HF_C1)) continue;
            if (CACHE && (f >> 4) != 15u && (int)(f >> 4) != w) continue;
            const RecV r = rec_decode<CACHE>(raw);
            const int d0 = r.i0 - r.j0;
            if (!r.rc) {
                if (CACHE || d1.bin(d0) == w) atomicAdd(&sh->cnt2[d2.bin(d0)], r.len);
            } else {
                for (int t = 0; t < r.len; ++t) {
                    const int d = d0 + 2 * t;
                    if (d1.bin(d) == w) atomicAdd(&sh->cnt2[d2.bin(d)], 1);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            int m2 = 0;
            for (int b = 0; b < 11; ++b) m2 = max(m2, sh->cnt2[b]);
            for (int b = 0; b < 11; ++b)
                if (sh->cnt2[b] == m2) {
                    sh->n_lists++;
                    sh->win_w = w; sh->win_b = b; sh->win_lo2 = lo2; sh->win_r2 = range2; sh->win_n = m2;
                }
        }
        __syncthreads();
    }
    pc.mark(33, tid == 0);
    if (sh->n_lists == 1) {
        // median of the single longest sub-list: per-value counters over its value span
        const int w = sh->win_w, b = sh->win_b, lo2 = sh->win_lo2, range2 = sh->win_r2, m = sh->win_n;
        const R4Div d2(lo2, range2);
        // values of sub-list b: 10*(v-lo2) in [b*range2, (b+1)*range2)  ->  v in [vlo, vhi]
        const int vlo = range2 > 0 ? lo2 + (b * range2 + 9) / 10 : lo2;
        const int vhi = range2 > 0 ? min(lo2 + range2, lo2 + ((b + 1) * range2 + 9) / 10 - 1) : lo2;
        const int width = vhi - vlo + 1;
        for (int q = tid; q < width; q += CLEAN_THREADS) counters[q] = 0;
        __syncthreads();
        for (int h = tid; h < n; h += CLEAN_THREADS) {
            const unsigned long long raw = recs[h];
            const uint32_t f = hflags.of(h, raw);
            if (!(f & HF_C1)) continue;
            if (CACHE && (f >> 4) != 15u && (int)(f >> 4) != w) continue;
            const RecV r = rec_decode<CACHE>(raw);
            const int d0 = r.i0 - r.j0;
            if (!r.rc) {
                if (d0 >= vlo && d0 <= vhi && (CACHE || d1.bin(d0) == w) && d2.bin(d0) == b)
                    atomicAdd(&counters[d0 - vlo], (uint32_t)r.len);
            } else {
                for (int t = 0; t < r.len; ++t) {
                    const int d = d0 + 2 * t;
                    if (d >= vlo && d <= vhi && d1.bin(d) == w && d2.bin(d) == b) atomicAdd(&counters[d - vlo], 1u);
                }
            }
        }
        __syncthreads();
        // order statistics (m-1)/2 and m/2, 0-based: np.median is their mean.  Block prefix over the counters,
        // the thread whose span holds a statistic walks its few counters.
        {
            const int per = (width + CLEAN_THREADS - 1) / CLEAN_THREADS;
            const int q0 = min(tid * per, width), q1 = min(q0 + per, width);
            uint32_t local = 0;
            for (int q = q0; q < q1; ++q) local += counters[q];
            uint32_t tot;
            const uint32_t run = block_exclusive_scan(local, sh, &tot);
            const uint32_t ks[2] = {(uint32_t)((m - 1) / 2), (uint32_t)(m / 2)};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (ks[t] >= run && ks[t] < run + local) {
                    uint32_t acc = run;
                    for (int q = q0; q < q1; ++q) {
                        acc += counters[q];
                        if (acc > ks[t]) { atomicAdd(&sh->c2x, vlo + q); break; }
                    }
                }
            }
        }
    }
    __syncthreads();
    pc.mark(34, tid == 0);
    {
        // dots with 10 * |X - Y| > |X| (X == 0: i >= 1), X = 2*j + c2x, Y = 2*i: count and sum of X - Y
        const int c2x = sh->c2x;
        int cn = 0;
        long long cs = 0;
        for (int h = tid; h < n; h += CLEAN_THREADS) {
            const unsigned long long raw = recs[h];
            if (!(hflags.of(h, raw) & HF_C1)) continue;
            const RecV r = rec_decode<CACHE>(raw);
            if (!r.rc) {
                // X - Y = df is the same for all dots, X = X0 + 2*t:  |X| < 10*|df|  <=>  -B < X < B
                const int X0 = 2 * r.j0 + c2x, df = X0 - 2 * r.i0, B = 10 * (df < 0 ? -df : df);
                int far = B > 0 ? count_below(X0, r.len, B) - count_below(X0, r.len, -B + 1) : 0;
                // the dot with X == 0 follows the other rule
                if (X0 <= 0 && !(X0 & 1) && (-X0 >> 1) < r.len) {
                    const int t0 = -X0 >> 1;
                    far += (int)(r.i0 + t0 >= 1) - (int)(B > 0);
                }
                cn += far; cs += (long long)far * df;
            } else {
                for (int t = 0; t < r.len; ++t) {
                    const int i = r.i0 + t, j = r.j0 - t;
                    const int X = 2 * j + c2x, Y = 2 * i;
                    const int df = X - Y, adf = df < 0 ? -df : df, aX = X < 0 ? -X : X;
                    const bool far = (X == 0) ? (i >= 1) : (10 * adf > aX);
                    if (far) { ++cn; cs += df; }
                }
            }
        }
        cn = wave_sum_i32(cn); cs = wave_sum_i64(cs);
        if ((tid & 63) == 0) { atomicAdd(&sh->dir_n, cn); atomicAdd((unsigned long long*)&sh->dir_sum2, (unsigned long long)cs); }
    }
    __syncthreads();
    pc.mark(35, tid == 0);
}

// everything after the records are in place (LDS copy or global), for one pair
template <bool NARROW, int PER_MAX, typename HP, typename FP>
__device__ __forceinline__ void clean_body(HP recs, FP hflags, int n, int n_dots, const DPair& pr, int len2, int range_words,
                                           uint32_t* bm, uint16_t* wrank, uint32_t* gcnt, uint32_t* bm2, uint16_t* wrank2,
                                           int gcap, bool dual_layout, CleanShared* sh, long long* st, CleanClock& pc)
{
    const int tid = threadIdx.x;
    const bool c1 = pr.flags & 1u, c2 = pr.flags & 2u, s3 = (pr.flags & 4u) && c1;
    // i - j over all dots (its bitmap was filled while the records were staged): C1's diagonal groups (>10)
    // and C2's diagonal step; then i + j over all dots (C1) and / or over the dots the diagonal step left (C2)
    bool done = false;
    if constexpr (NARROW) {
      if (dual_layout) {
        done = true;
        // (staged in LDS with room for both axes' bitmaps, which the staging pass has filled for a C1 pair; cluster_dual
        // declines a pair with more groups than its counters hold, whose axes are then clustered one after the other,
        // bitmaps marked again)
        if (c1 && c2) {
            if (!cluster_dual<false, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, bm2, wrank2, gcnt, gcap, sh, HF_C2D, pc)) {
                cluster_axis<false, NARROW, false, false, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, gcnt, sh, 0u, WF_D1, HF_C2D, pc);
                cluster_axis<true, NARROW, false, false, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, 0u, WF_A1, 0u, pc);
            }
            cluster_axis<true, NARROW, false, true, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, HF_C2D, 0u, HF_C2A, pc, 20);
        } else if (c1) {
            if (!cluster_dual<true, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, bm2, wrank2, gcnt, gcap, sh, 0u, pc)) {
                cluster_axis<false, NARROW, false, false, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, gcnt, sh, 0u, WF_D1, 0u, pc, 16);
                cluster_axis<true, NARROW, false, true, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, 0u, WF_A1, 0u, pc, 20);
            }
        } else if (c2) {
            cluster_axis<false, NARROW, true, false, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, gcnt, sh, 0u, 0u, HF_C2D, pc);
            cluster_axis<true, NARROW, false, true, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, HF_C2D, 0u, HF_C2A, pc);
        }
      }
    }
    if (done) {
    } else if (c1 && c2) {
        cluster_axis<false, NARROW, true, false, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, gcnt, sh, 0u, WF_D1, HF_C2D, pc);
        cluster_axis<true, NARROW, false, false, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, 0u, WF_A1, 0u, pc);
        cluster_axis<true, NARROW, false, true, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, HF_C2D, 0u, HF_C2A, pc);
    } else if (c1) {
        cluster_axis<false, NARROW, true, false, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, gcnt, sh, 0u, WF_D1, 0u, pc, 16);
        cluster_axis<true, NARROW, false, true, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, 0u, WF_A1, 0u, pc, 20);
    } else if (c2) {
        cluster_axis<false, NARROW, true, false, PER_MAX>(recs, hflags, n, len2, range_words, bm, wrank, gcnt, sh, 0u, 0u, HF_C2D, pc);
        cluster_axis<true, NARROW, false, true, PER_MAX>(recs, hflags, n, 0, range_words, bm, wrank, gcnt, sh, HF_C2D, 0u, HF_C2A, pc);
    }
    if (s3) directed_stats<NARROW>(recs, hflags, n, gcnt, sh, pc);
    if (tid == 0) {
        st[0] = n_dots; st[1] = sh->min_j; st[2] = sh->max_j; st[3] = sh->c1_kept; st[4] = (long long)sh->c1_sum_abs;
        st[5] = sh->c2_kept; st[6] = sh->c2_count10; st[7] = sh->n_diag; st[8] = sh->n_lower; st[9] = sh->c2_kept_diag;
        st[10] = s3 ? sh->c2x : 0; st[11] = s3 ? sh->dir_n : 0; st[12] = s3 ? sh->dir_sum2 : 0; st[13] = s3 ? sh->n_lists : 0;
        st[14] = 0; st[15] = 0;
    }
}

// One pair of n records / n_dots dots.  Dynamic LDS: bitmap (range_words_cap words) | wrank (u16 each) | group
// sizes | (IN_LDS: record copy (hcap x 8 B, 8-byte aligned) | flag bytes (hcap)).
template <bool IN_LDS, int PER_MAX>
__device__ __forceinline__ void clean_pair(int p, uint32_t* lds, CleanShared& sh, const DPair& pr, int n, int n_dots, int len2,
                                           int range_words, const unsigned long long* __restrict__ recs_all,
                                           uint8_t* __restrict__ hflags_all, long long* __restrict__ stats,
                                           int range_words_cap, int groups_cap, int hcap, bool dual_layout, bool keep_flags = true)
{
    const int tid = threadIdx.x;
    long long* st = stats + (size_t)p * 16;
    CleanClock pc;
    const unsigned long long* grecs = recs_all + pr.hit_off;
    uint8_t* gflags = hflags_all + pr.hit_off;
    // the first records of every thread are requested before anything else touches memory: global latency, not
    // bandwidth, is what this pass waits for
    constexpr int PF = 4;
    unsigned long long x[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        const int h = q * CLEAN_THREADS + tid;
        x[q] = h < n ? grecs[h] : 0ull;
    }
    // clean_kernel (IN_LDS): bitmap and ranks of i - j | bitmap and ranks of i + j (cluster_dual) | group counters | records;
    // clean_big_kernel: bitmap | ranks | group counters
    const int bw = range_words_cap + (range_words_cap + 1) / 2;
    uint32_t* bm = lds;
    uint16_t* wrank = reinterpret_cast<uint16_t*>(bm + range_words_cap);
    uint32_t* bm2 = lds + bw;
    uint16_t* wrank2 = reinterpret_cast<uint16_t*>(bm2 + range_words_cap);
    uint32_t* gcnt = lds + (dual_layout ? 2 * bw : bw);
    const int rec_word = ((dual_layout ? 2 : 1) * bw + (groups_cap + 1) / 2 + 1) & ~1;
    unsigned long long* lrecs = reinterpret_cast<unsigned long long*>(lds + rec_word);

    if (tid == 0) {
        sh.min_j = 0x7FFFFFFF; sh.max_j = -1; sh.n_diag = 0; sh.n_lower = 0;
        sh.c1_kept = 0; sh.c2_kept = 0; sh.c2_count10 = 0; sh.c2_kept_diag = 0; sh.c1_sum_abs = 0ULL;
        sh.kd_lo = 0x7FFFFFFF; sh.kd_hi = -0x7FFFFFFF;
    }
    // pass 0: first/last j, diagonal and lower-triangle counts; stage the records, clear the flags, and fill
    // the occupancy bitmap of i - j for the first clustering step
    const bool any_axis = (pr.flags & 3u) != 0u;
    const bool dual = IN_LDS && dual_layout && (pr.flags & 1u);      // C1: both axes are clustered in one sweep (cluster_dual)
    for (int w = tid; w < range_words; w += CLEAN_THREADS) bm[w] = 0;
    if (dual)
        for (int w = tid; w < range_words; w += CLEAN_THREADS) bm2[w] = 0;
    __syncthreads();
    {
        int mn = 0x7FFFFFFF, mx = -1, nd = 0, nl = 0;
        auto take = [&](int h, unsigned long long xr) {
            const RecV r = rec_decode<false>(xr);
            if (!r.rc) {
                mn = min(mn, r.j0); mx = max(mx, r.j0 + r.len - 1);
                nd += (r.j0 == r.i0) ? r.len : 0;
                nl += (r.j0 > r.i0) ? r.len : 0;
            } else {
                // j = j0 - t, i = i0 + t:  j == i for 2t == D, j > i for 2t < D, D = j0 - i0
                const int D = r.j0 - r.i0;
                mn = min(mn, r.j0 - r.len + 1); mx = max(mx, r.j0);
                nd += (D >= 0 && !(D & 1) && (D >> 1) < r.len) ? 1 : 0;
                nl += D > 0 ? min(r.len, (D + 1) >> 1) : 0;
            }
            if (IN_LDS) lrecs[h] = rec_to_lds(xr);           // (flags and group fields clear)
            else gflags[h] = 0;
            if (any_axis) mark_values(bm, (uint32_t)(r.i0 - r.j0 + len2), r.len, r.rc);
            if (IN_LDS && dual) mark_values(bm2, (uint32_t)(r.i0 + r.j0), r.len, !r.rc);
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int h = q * CLEAN_THREADS + tid;
            if (h < n) take(h, x[q]);
        }
        for (int h = PF * CLEAN_THREADS + tid; h < n; h += CLEAN_THREADS) take(h, grecs[h]);
        mn = wave_min_i32(mn); mx = wave_max_i32(mx); nd = wave_sum_i32(nd); nl = wave_sum_i32(nl);
        if ((tid & 63) == 0) {
            atomicMin(&sh.min_j, mn); atomicMax(&sh.max_j, mx);
            atomicAdd(&sh.n_diag, nd); atomicAdd(&sh.n_lower, nl);
        }
    }
    __syncthreads();
    pc.mark(8, tid == 0);                          // pass 0
    if (IN_LDS) {
        clean_body<true, PER_MAX>(lrecs, LFlags{lrecs}, n, n_dots, pr, len2, range_words, bm, wrank, gcnt, bm2, wrank2, groups_cap, dual_layout, &sh, st, pc);
        __syncthreads();
        pc.mark(9, tid == 0);                      // what the nested stamps left of clean_body
        // four flag bytes per store (the pair's slot is padded to a multiple of four): the public bits of four records
        // (a thread takes the flags of ONE record - consecutive lanes read consecutive records - and the four lanes of a
        // quad put their bytes together with two quad permutes; the quad's first lane stores the word)
        // (keep_flags: only a run whose dots a caller may fetch - vapor_plan_run - writes them; the device-finished path needs the
        // statistics alone)
        if (keep_flags) {
            const uint32_t* hi = reinterpret_cast<const uint32_t*>(lrecs) + 1;
            uint32_t* gf4 = reinterpret_cast<uint32_t*>(gflags);
            const int n4 = (n + 3) & ~3;
            for (int h = tid; h < n4; h += CLEAN_THREADS) {           // (n4 and CLEAN_THREADS are multiples of 4: whole quads)
                uint32_t x = h < n ? (((hi[2 * h] >> 6) & 7u) << (8 * (h & 3))) : 0u;
                x |= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
                x |= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
                if ((h & 3) == 0) gf4[h >> 2] = x;
            }
        }
        pc.mark(10, tid == 0);
    } else {
        clean_body<false, PER_MAX>(grecs, GFlags{gflags}, n, n_dots, pr, len2, range_words, bm, wrank, gcnt, bm2, wrank2, groups_cap, dual_layout, &sh, st, pc);
    }
    pc.flush(tid == 0);
}

// The remap of a shared join for ONE target, by the clean workgroup of that target itself (256 threads): the workgroup that is
// about to clean pair p first cuts p's records out of the shared dot plot - the arithmetic of remap_kernel for one slot - writes
// them to p's slot and goes on to read them back: through the L2 of its own XCD instead of through HBM and a kernel boundary.
// Returns p's packed count (records | dots << 32), which it also stores for the host.  `w`: the start of the workgroup's dynamic
// LDS (free until the cleaning proper begins).
#ifndef VAPOR_REMAP_CLEAN_PER
#define VAPOR_REMAP_CLEAN_PER 4
#endif
constexpr int REMAP_CLEAN_PER = VAPOR_REMAP_CLEAN_PER;
__device__ __forceinline__ unsigned long long remap_for_target(int p, const DPair& tg, const DServe& sv,
                                                               const int32_t* __restrict__ tables, unsigned long long* hits,
                                                               unsigned long long* n_hits, uint32_t* w, unsigned int* overflow)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int slot = sv.slot;
    uint32_t nrec = (uint32_t)n_hits[sv.dpair];
    const int n_iv = min(sv.n_iv, REMAP_MAX_IV);
    const int32_t* tb = tables + sv.iv_first;
    uint32_t* c = w;                                               // [0] records, [1] dots
    int* s_B = reinterpret_cast<int*>(w + 4);                      // n_iv + 2 boundaries
    uint32_t* s_ops = w + 64;                                      // two op words per interval: this slot's
    if (tid <= n_iv) s_B[tid] = tb[tid];
    if (tid == n_iv + 1) s_B[tid] = 0x7FFFFFFF;
    if (tid < 2 * n_iv) s_ops[tid] = (uint32_t)tb[n_iv + 1 + (tid >> 1) * REMAP_OPS + slot * 2 + (tid & 1)];
    if (tid == 255) { c[0] = 0u; c[1] = 0u; c[2] = 0u; }
    if (nrec > sv.cap) {
        // (counted once per shared plot, as remap_kernel does: by the workgroup of the target the host marked)
        if (tid == 0 && overflow && sv.pad) { atomicAdd(&overflow[0], 1u); atomicAdd(&overflow[2], 1u); }
        nrec = sv.cap;
    }
    __syncthreads();
    if (tid < n_iv && (s_ops[2 * tid + 1] & 1u)) c[2] = 1u;       // some stretch of T lies twice in this target (a duplication)
    __syncthreads();
    const int n_cp = c[2] ? 2 : 1;
    int stp0 = 1;                                                  // the interval search starts at the table's size
    while (stp0 * 2 <= n_iv) stp0 *= 2;
    const int off2 = tg.off2;
    const uint32_t cap = tg.cap;
    const unsigned long long* src = hits + sv.hit_off;
    unsigned long long* dst = hits + tg.hit_off;
    uint32_t my_dots = 0;
    // The target that is the window itself (slot 0 of its group: half of all served pairs): its k-mer starts are the first
    // w_hi + 1 of the shared sequence, unmoved and unturned - no interval table, no search, no second copy; a record is kept up
    // to w_hi and clipped at miss_bp, a quarter of the instructions of the general case below.
    if (slot == 0) {
        const int w_hi = tg.len2 - tg.k;
        for (uint32_t h0 = 0; h0 < nrec; h0 += 256u * REMAP_CLEAN_PER) {
            uint32_t w_lo[REMAP_CLEAN_PER], w_hi2[REMAP_CLEAN_PER], em = 0, dots = 0;
#pragma unroll
            for (int q = 0; q < REMAP_CLEAN_PER; ++q) {
                const uint32_t h = h0 + (uint32_t)(q * 256 + tid);
                const unsigned long long r = h < nrec ? src[h] : 0ull;          // (an empty slot: length 0)
                const int e0 = VREC_J(r), i0 = VREC_I(r), len = VREC_LEN(r);
                const bool rc = VREC_RC(r);
                // same strand: e = e0 + t; reverse complement: e = e0 - t (t = 0 .. len - 1, i = i0 + t).  Kept: off2 <= e <= w_hi.
                int i_first = i0, ja = e0, n = len;
                if (!rc) {
                    n = min(n, w_hi - e0 + 1);
                    const int skip = max(0, off2 - ja); i_first += skip; ja += skip; n -= skip;
                } else {
                    const int skip = max(0, e0 - w_hi); i_first += skip; ja -= skip; n -= skip;
                    n = min(n, ja - off2 + 1);
                }
                n = max(n, 0);
                em |= (n > 0 ? 1u : 0u) << q;
                dots += (uint32_t)n;
                w_lo[q] = (uint32_t)i_first | ((uint32_t)(ja - off2) << 16);
                w_hi2[q] = (uint32_t)n | ((rc ? 1u : 0u) << 16);
            }
            my_dots += dots;
            const uint32_t mine = (uint32_t)__popc(em);
            const uint32_t incl = wave_incl_scan_u32(mine);
            uint32_t base = 0;
            if (lane == 63 && incl) base = atomicAdd(&c[0], incl);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
            uint32_t at = base + incl - mine;
#pragma unroll
            for (int q = 0; q < REMAP_CLEAN_PER; ++q) {
                if (!((em >> q) & 1u)) continue;
                if (at < cap) dst[at] = (unsigned long long)w_lo[q] | ((unsigned long long)w_hi2[q] << 32);
                ++at;
            }
        }
    } else
    for (uint32_t h0 = 0; h0 < nrec; h0 += 256u * REMAP_CLEAN_PER) {
        int e_lo[REMAP_CLEAN_PER], e_hi[REMAP_CLEAN_PER], i_at_lo[REMAP_CLEAN_PER], iv[REMAP_CLEAN_PER];
        uint32_t rcbits = 0;
#pragma unroll
        for (int q = 0; q < REMAP_CLEAN_PER; ++q) {
            const uint32_t h = h0 + (uint32_t)(q * 256 + tid);
            const bool have = h < nrec;
            const unsigned long long r = have ? src[h] : 0ull;
            const int e0 = VREC_J(r), i0 = VREC_I(r), len = VREC_LEN(r);
            const bool rc = VREC_RC(r);
            rcbits |= (rc ? 1u : 0u) << q;
            e_lo[q] = rc ? e0 - (len - 1) : e0;
            e_hi[q] = have ? e_lo[q] + len - 1 : e_lo[q] - 1;
            i_at_lo[q] = rc ? i0 + (len - 1) : i0;
            int pos = 0;
            for (int stp = stp0; stp > 0; stp >>= 1)
                if (pos + stp <= n_iv && s_B[pos + stp] <= e_lo[q]) pos += stp;
            iv[q] = pos;
        }
        for (int step = 0;; ++step) {
            uint32_t act = 0;
#pragma unroll
            for (int q = 0; q < REMAP_CLEAN_PER; ++q) act |= ((iv[q] + step < n_iv && s_B[iv[q] + step] <= e_hi[q]) ? 1u : 0u) << q;
            if (!__ballot(act != 0u)) break;
            for (int cp = 0; cp < n_cp; ++cp) {
                uint32_t op[REMAP_CLEAN_PER], em = 0;
#pragma unroll
                for (int q = 0; q < REMAP_CLEAN_PER; ++q) {
                    op[q] = ((act >> q) & 1u) ? s_ops[(iv[q] + step) * 2 + cp] : 0u;
                    em |= (op[q] & 1u) << q;
                }
                if (!__ballot(em != 0u)) continue;
                uint32_t w_lo[REMAP_CLEAN_PER], w_hi[REMAP_CLEAN_PER];
                uint32_t dots = 0;
#pragma unroll
                for (int q = 0; q < REMAP_CLEAN_PER; ++q) {
                    const int t = iv[q] + step;
                    const int pe_lo = max(e_lo[q], s_B[t]), pe_hi = min(e_hi[q], s_B[t + 1] - 1);
                    const bool rc = (rcbits >> q) & 1u, flip = (op[q] >> 1) & 1u;
                    const int delta = (int)op[q] >> 2;
                    const int e_first = rc ? pe_hi : pe_lo;
                    int i_first = rc ? i_at_lo[q] - (pe_hi - e_lo[q]) : i_at_lo[q] + (pe_lo - e_lo[q]);
                    int n = pe_hi - pe_lo + 1;
                    int ja = flip ? delta - e_first : e_first + delta;
                    const int dj = (flip ? -1 : 1) * (rc ? -1 : 1);
                    if (dj > 0) { const int skip = max(0, off2 - ja); i_first += skip; ja += skip; n -= skip; }
                    else n = min(n, ja - off2 + 1);
                    if (!((em >> q) & 1u) || n <= 0) { n = 0; em &= ~(1u << q); }
                    dots += (uint32_t)n;
                    w_lo[q] = (uint32_t)i_first | ((uint32_t)(ja - off2) << 16);
                    w_hi[q] = (uint32_t)n | ((dj < 0 ? 1u : 0u) << 16);
                }
                my_dots += dots;                                   // (summed over the workgroup once, after the last round)
                const uint32_t mine = (uint32_t)__popc(em);
                const uint32_t incl = wave_incl_scan_u32(mine);
                uint32_t base = 0;
                if (lane == 63 && incl) base = atomicAdd(&c[0], incl);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
                uint32_t at = base + incl - mine;
#pragma unroll
                for (int q = 0; q < REMAP_CLEAN_PER; ++q) {
                    if (!((em >> q) & 1u)) continue;
                    if (at < cap) dst[at] = (unsigned long long)w_lo[q] | ((unsigned long long)w_hi[q] << 32);
                    ++at;
                }
            }
        }
    }
    {
        const uint32_t wd = (uint32_t)wave_sum_i32((int)my_dots);
        if (lane == 0 && wd) atomicAdd(&c[1], wd);
    }
    __syncthreads();                               // (every record store issued and visible to the workgroup; the counters final)
    const unsigned long long cnt = (unsigned long long)c[0] | ((unsigned long long)c[1] << 32);
    if (tid == 0) n_hits[p] = cnt;
    __syncthreads();                               // (the LDS words are the cleaning's from here on)
    return cnt;
}

// One workgroup per pair: pairs with at most hcap records and fewer than 65536 dots are cleaned entirely out of
// LDS with 16-bit group counters; the others are appended to big_list for clean_big_kernel.
// n_hits[p] = records | dots << 32 (join_verify).  len2 and the value range come with the pair record, so the
// only dependent global reads before the records are the pair record and its counts.
template <int PER_MAX>
__global__ __launch_bounds__(CLEAN_THREADS, 8) void clean_kernel(
    const DPair* __restrict__ pairs, const int32_t* __restrict__ pair_list,
    unsigned long long* n_hits, unsigned long long* recs_all,
    uint8_t* __restrict__ hflags_all, long long* __restrict__ stats, int range_words_cap, int groups_cap, int hcap,
    unsigned int* __restrict__ overflow, int32_t* __restrict__ big_list, int big_follows, int dual_layout,
    const DServe* __restrict__ serve, const int32_t* __restrict__ share_tables, int keep_flags)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ CleanShared sh;
    const int tid = threadIdx.x;
    const int p = pair_list ? pair_list[blockIdx.x] : (int)blockIdx.x;
    const DPair pr = pairs[p];
    // a pair served by a shared join: its records are cut out of the shared dot plot here, by the workgroup that cleans them
    DServe sv;
    sv.dpair = -1;
    if (serve) sv = serve[p];
    const unsigned long long cnt = sv.dpair >= 0 ? remap_for_target(p, pr, sv, share_tables, recs_all, n_hits, lds, overflow) : n_hits[p];
    const uint32_t nrec = (uint32_t)cnt, ndots = (uint32_t)(cnt >> 32);
    if (nrec > pr.cap || nrec == 0u) {
        long long* st = stats + (size_t)p * 16;
        if (tid < 16) {
            long long v = 0;
            if (tid == 0) v = (long long)ndots;
            if (tid == 1 || tid == 2) v = -1;
            if (tid == 14) v = (long long)nrec;   // records: what the slot has to hold on the rerun
            if (tid == 15 && nrec) v = -2;        // VAPOR_E_OVERFLOW
            st[tid] = v;
        }
        if (tid == 0 && nrec && overflow) { atomicAdd(&overflow[0], 1u); atomicAdd(&overflow[2], 1u); }   // [2]: never cleared by the join
        return;
    }
    const int n = (int)nrec;                   // cap < 2^31
    if (n > hcap || ndots > 65535u) {
        if (tid == 0) {
            big_list[atomicAdd(&overflow[1], 1u)] = p;
            // an asynchronous step that left clean_big_kernel out because the plan's blocking run saw no such pair:
            // the sticky counter makes vapor_plan_sync ask for a blocking run
            if (!big_follows) atomicAdd(&overflow[2], 1u);
        }
        return;
    }
    clean_pair<true, PER_MAX>(p, lds, sh, pr, n, (int)ndots, pr.len2, min((pr.len1 + pr.len2 + 2 + 31) >> 5, range_words_cap),
                     recs_all, hflags_all, stats, range_words_cap, groups_cap, hcap, dual_layout != 0, keep_flags != 0);
}

// The pairs clean_kernel left (more records than the LDS copy holds, or too many dots for 16-bit counters): the
// records stream from L2/HBM on every pass, group sizes are 32-bit.  A fixed grid walks the list; workgroups
// beyond its length leave at once.
__global__ __launch_bounds__(CLEAN_THREADS, 4) void clean_big_kernel(
    const DPair* __restrict__ pairs, const unsigned long long* __restrict__ n_hits,
    const unsigned long long* __restrict__ recs_all, uint8_t* __restrict__ hflags_all, long long* __restrict__ stats,
    int range_words_cap, int groups_cap, const unsigned int* __restrict__ overflow, const int32_t* __restrict__ big_list)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ CleanShared sh;
    const unsigned int n_big = overflow[1];
    for (unsigned int b = blockIdx.x; b < n_big; b += gridDim.x) {
        const int p = big_list[b];
        const DPair pr = pairs[p];
        const unsigned long long cnt = n_hits[p];
        __syncthreads();                       // the previous pair's statistics have been read
        clean_pair<false, CLEAN_PER_MAX>(p, lds, sh, pr, (int)(uint32_t)cnt, (int)(uint32_t)(cnt >> 32), pr.len2,
                          min((pr.len1 + pr.len2 + 2 + 31) >> 5, range_words_cap), recs_all, hflags_all, stats,
                          range_words_cap, groups_cap, 0, false);
    }
}

// ------------------------------------------------------------------------------------------
// shared joins: one probe per read for a window and the alleles derived from it
// ------------------------------------------------------------------------------------------
// The reference fills dotdata(read, ref) and dotdata(read, alt) separately (SF:185-186, 242-243, 278-279), and alt is ref with a
// stretch removed, doubled, reversed or inserted (SF:1712, 1755, 1907, 1872): nine k-mers in ten are the same on both sides.
// For such pairs the plan joins the read ONCE against a hidden sequence T = ref followed by the few stretches of the derived
// alleles that hold k-mers ref does not (junctions, inserted bytes) - join_kernel as it is - and this kernel turns the records
// of that one dot plot into the records of every target pair: a k-mer start e of T maps to position j of target `slot` through
// the interval maps of the (window, k) group,
//     e in [lo, hi]  ->  j = base + (e - lo)          (a forward slice of the window, or a stretch of the allele itself)
//                        j = base - (e - lo), strands swapped   (a reverse-complemented slice: dotdata searches both strands,
//                                                                so the k-mer that matched the read's forward strand there
//                                                                matches its reverse complement here)
// and a run of dots is cut to the part of it that lies inside an interval (runs are consecutive e AND consecutive i, so the cut
// part is a run again; a reversed slice turns a same-strand run into a reverse-complement run, which the record format has).
// K-mer starts of T that belong to no interval (the k-1 positions that straddle two stretches) are dots of nothing and are
// dropped here.  Every dot of every target is produced exactly once: a k-mer start of an allele lies inside exactly one of its
// slices or inside exactly one of its own stretches in T.  One workgroup per read.
__global__ __launch_bounds__(256) void remap_kernel(const DPair* __restrict__ pairs, const DShare* __restrict__ shares,
                                                   const int32_t* __restrict__ tables, unsigned long long* hits,
                                                   unsigned long long* n_hits, unsigned int* __restrict__ overflow)
{
    __shared__ uint32_t c_rec[4], c_dots[4];
    __shared__ int s_B[REMAP_MAX_IV + 2];
    __shared__ uint32_t s_ops[REMAP_MAX_IV * REMAP_OPS];
    __shared__ int s_tp[4], s_off2[4], s_whi;
    __shared__ uint32_t s_cap[4];
    __shared__ long long s_hoff[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const DShare sh = shares[blockIdx.x];
    const DPair dp = pairs[sh.dpair];
    const unsigned long long cnt = n_hits[sh.dpair];
    uint32_t nrec = (uint32_t)cnt;
    if (nrec > dp.cap) {
        // the shared plot outgrew its slot: its count stays in n_hits for the host (a blocking run resizes and reruns; an
        // asynchronous step reports through the sticky counter), the targets get what fits
        if (tid == 0) { atomicAdd(&overflow[0], 1u); atomicAdd(&overflow[2], 1u); }
        nrec = dp.cap;
    }
    const int n_iv = min(sh.n_iv, REMAP_MAX_IV);
    const int32_t* tb = tables + sh.iv_first;
    if (tid <= n_iv) s_B[tid] = tb[tid];
    if (tid == n_iv + 1) s_B[tid] = 0x7FFFFFFF;
    for (int x = tid; x < n_iv * REMAP_OPS; x += 256) s_ops[x] = (uint32_t)tb[n_iv + 1 + x];
    if (tid >= 192 && tid < 196) {
        const int t = tid - 192, tp = shares[blockIdx.x].target[t];
        s_tp[t] = tp;
        c_rec[t] = 0u; c_dots[t] = 0u;
        if (tp >= 0) {
            const DPair tg = pairs[tp];
            s_off2[t] = tg.off2; s_cap[t] = tg.cap; s_hoff[t] = tg.hit_off;
            if (t == 0) s_whi = tg.len2 - tg.k;                    // (slot 0 is the window: its last k-mer start)
        }
    }
    __syncthreads();
    const unsigned long long* src = hits + dp.hit_off;
    for (uint32_t h0 = 0; h0 < nrec; h0 += 256u * REMAP_PER) {
        // the round's records: first and last k-mer start, first read position, and the interval the first start lies in
        int e_lo[REMAP_PER], e_hi[REMAP_PER], i_at_lo[REMAP_PER], iv[REMAP_PER];
        uint32_t rcbits = 0;                                      // record q is a reverse-complement record (e falls as i rises)
#pragma unroll
        for (int q = 0; q < REMAP_PER; ++q) {
            const uint32_t h = h0 + (uint32_t)(q * 256 + tid);
            const bool have = h < nrec;
            const unsigned long long r = have ? src[h] : 0ull;
            const int e0 = VREC_J(r), i0 = VREC_I(r), len = VREC_LEN(r);
            const bool rc = VREC_RC(r);
            rcbits |= (rc ? 1u : 0u) << q;
            // e(t) = e0 +- t, i(t) = i0 + t: the k-mer starts run over [e_lo, e_hi]; i_at_lo = the read position that goes with e_lo
            e_lo[q] = rc ? e0 - (len - 1) : e0;
            e_hi[q] = have ? e_lo[q] + len - 1 : e_lo[q] - 1;     // (an empty slot: no interval ever reaches it)
            i_at_lo[q] = rc ? i0 + (len - 1) : i0;
            int pos = 0;                                          // largest t with B[t] <= e_lo (B[0] = 0)
#pragma unroll
            for (int stp = 32; stp > 0; stp >>= 1)
                if (pos + stp <= n_iv && s_B[pos + stp] <= e_lo[q]) pos += stp;
            iv[q] = pos;
        }
        // Slot 0 - the target that is the window itself - needs no interval: its k-mer starts are the shared sequence's up to
        // the window's last one, unmoved and unturned.  One straight pass (keep up to w_hi, clip at miss_bp) instead of its share of
        // the loops below: a quarter of the instructions for half of the records written.
        if (s_tp[0] >= 0) {                                        // (uniform)
            const int off2 = s_off2[0], w_hi = s_whi;
            uint32_t w_lo[REMAP_PER], w_hi2[REMAP_PER], em = 0;
            int dots = 0;
#pragma unroll
            for (int q = 0; q < REMAP_PER; ++q) {
                const bool rc = (rcbits >> q) & 1u;
                // the record in read order: first k-mer start e_first at read position i_first, e moving by +-1 per dot
                int n = e_hi[q] - e_lo[q] + 1;                     // (0 for an empty slot)
                int i_first = rc ? i_at_lo[q] - (n - 1) : i_at_lo[q];
                int ja = rc ? e_hi[q] : e_lo[q];
                if (!rc) {
                    n = min(n, w_hi - ja + 1);
                    const int skip = max(0, off2 - ja); i_first += skip; ja += skip; n -= skip;
                } else {
                    const int skip = max(0, ja - w_hi); i_first += skip; ja -= skip; n -= skip;
                    n = min(n, ja - off2 + 1);
                }
                n = max(n, 0);
                em |= (n > 0 ? 1u : 0u) << q;
                dots += n;
                w_lo[q] = (uint32_t)i_first | ((uint32_t)(ja - off2) << 16);
                w_hi2[q] = (uint32_t)n | ((rc ? 1u : 0u) << 16);
            }
            const uint32_t mine = (uint32_t)__popc(em);
            const uint32_t incl = wave_incl_scan_u32(mine);
            const int wdots = wave_sum_i32(dots);
            uint32_t base = 0;
            if (lane == 63 && incl) { base = atomicAdd(&c_rec[0], incl); atomicAdd(&c_dots[0], (uint32_t)wdots); }
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
            uint32_t at = base + incl - mine;
            const uint32_t cap = s_cap[0];
            unsigned long long* dst = hits + s_hoff[0];
#pragma unroll
            for (int q = 0; q < REMAP_PER; ++q) {
                if (!((em >> q) & 1u)) continue;
                if (at < cap) dst[at] = (unsigned long long)w_lo[q] | ((unsigned long long)w_hi2[q] << 32);
                ++at;
            }
        }
        for (int step = 0;; ++step) {
            // the part of every record inside interval iv + step (the whole record, nearly always, at step 0)
            uint32_t act = 0;
#pragma unroll
            for (int q = 0; q < REMAP_PER; ++q) act |= ((iv[q] + step < n_iv && s_B[iv[q] + step] <= e_hi[q]) ? 1u : 0u) << q;
            if (!__ballot(act != 0u)) break;
#pragma unroll
            for (int sc = 2; sc < REMAP_OPS; ++sc) {               // (slot 0: above)
                const int slot = sc >> 1;
                if (s_tp[slot] < 0) continue;                      // (uniform)
                uint32_t op[REMAP_PER], em = 0;
#pragma unroll
                for (int q = 0; q < REMAP_PER; ++q) {
                    op[q] = ((act >> q) & 1u) ? s_ops[(iv[q] + step) * REMAP_OPS + sc] : 0u;
                    em |= (op[q] & 1u) << q;
                }
                if (!__ballot(em != 0u)) continue;
                const int off2 = s_off2[slot];
                uint32_t w_lo[REMAP_PER], w_hi[REMAP_PER];
                int dots = 0;
#pragma unroll
                for (int q = 0; q < REMAP_PER; ++q) {
                    const int t = iv[q] + step;
                    const int pe_lo = max(e_lo[q], s_B[t]), pe_hi = min(e_hi[q], s_B[t + 1] - 1);
                    const bool rc = (rcbits >> q) & 1u, flip = (op[q] >> 1) & 1u;
                    const int delta = (int)op[q] >> 2;
                    // in read order the part starts at e_first (its first i) and moves by sd per dot
                    const int e_first = rc ? pe_hi : pe_lo;
                    int i_first = rc ? i_at_lo[q] - (pe_hi - e_lo[q]) : i_at_lo[q] + (pe_lo - e_lo[q]);
                    int n = pe_hi - pe_lo + 1;
                    int ja = flip ? delta - e_first : e_first + delta;
                    const int dj = (flip ? -1 : 1) * (rc ? -1 : 1);
                    // only j >= off2 counts (the allele[miss_bp:] slice)
                    if (dj > 0) { const int skip = max(0, off2 - ja); i_first += skip; ja += skip; n -= skip; }
                    else n = min(n, ja - off2 + 1);
                    if (!((em >> q) & 1u) || n <= 0) { n = 0; em &= ~(1u << q); }
                    dots += n;
                    w_lo[q] = (uint32_t)i_first | ((uint32_t)(ja - off2) << 16);
                    w_hi[q] = (uint32_t)n | ((dj < 0 ? 1u : 0u) << 16);
                }
                const uint32_t mine = (uint32_t)__popc(em);
                const uint32_t incl = wave_incl_scan_u32(mine);
                const int wdots = wave_sum_i32(dots);
                uint32_t base = 0;
                if (lane == 63 && incl) { base = atomicAdd(&c_rec[slot], incl); atomicAdd(&c_dots[slot], (uint32_t)wdots); }
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
                uint32_t at = base + incl - mine;
                const uint32_t cap = s_cap[slot];
                unsigned long long* dst = hits + s_hoff[slot];
#pragma unroll
                for (int q = 0; q < REMAP_PER; ++q) {
                    if (!((em >> q) & 1u)) continue;
#if !(defined(VAPOR_AB) && VAPOR_AB == 1)       /* (developer variant 1: everything but the stores) */
                    if (at < cap) dst[at] = (unsigned long long)w_lo[q] | ((unsigned long long)w_hi[q] << 32);
#endif
                    ++at;
                }
            }
        }
    }
    __syncthreads();
    if (tid < 4 && s_tp[tid] >= 0) n_hits[s_tp[tid]] = (unsigned long long)c_rec[tid] | ((unsigned long long)c_dots[tid] << 32);
}

// expands the records of selected pairs to dots (int32 j, i) and per-dot flag bytes in a dense buffer; the dots
// of a pair keep the record order, each record's dots in read order
__global__ __launch_bounds__(256) void gather_kernel(const DPair* __restrict__ pairs, const long long* __restrict__ sel,
                                                    const long long* __restrict__ out_off, const long long* __restrict__ n_rec,
                                                    const unsigned long long* __restrict__ recs, const uint8_t* __restrict__ hflags,
                                                    int32_t* __restrict__ out_ji, uint8_t* __restrict__ out_flags)
{
    __shared__ uint32_t wtot[4];
    __shared__ long long run_base;
    const long long p = sel[blockIdx.x];
    const DPair pr = pairs[p];
    const long long n = n_rec[blockIdx.x], n_out = out_off[blockIdx.x + 1] - out_off[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid == 0) run_base = out_off[blockIdx.x];
    __syncthreads();
    for (long long h0 = 0; h0 < n; h0 += 256) {
        const long long h = h0 + tid;
        const unsigned long long r = h < n ? recs[pr.hit_off + h] : 0ull;
        const uint32_t len = h < n ? (uint32_t)VREC_LEN(r) : 0u;
        const uint32_t incl = wave_incl_scan_u32(len);
        if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        __syncthreads();
        long long o = run_base + (long long)(incl - len);
        uint32_t tot = 0;
        for (int q = 0; q < 4; ++q) { if (q < (tid >> 6)) o += wtot[q]; tot += wtot[q]; }
        if (h < n) {
            const uint8_t f = out_flags ? hflags[pr.hit_off + h] : (uint8_t)0;
            const int s = VREC_RC(r) ? -1 : 1;
            for (uint32_t t = 0; t < len; ++t) {
                if (o + t - out_off[blockIdx.x] >= n_out) break;    // never past this pair's share of the buffer
                out_ji[2 * (o + t)] = VREC_J(r) + s * (int)t;
                out_ji[2 * (o + t) + 1] = VREC_I(r) + (int)t;
                if (out_flags) out_flags[o + t] = f;
            }
        }
        __syncthreads();
        if (tid == 0) run_base += tot;
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------
// finish kernel: statistics records -> per-read scores -> per-locus VaPoR_QS / GS / GT / GQ
// ------------------------------------------------------------------------------------------
// Float64 restatement of the scorers' gates (SF:182-203, 241-257, 277-294), the drivers' per-read
// rule (skip a read when either scorer output is 0, score = 1 - b/a; DEL takes the min of the
// abs_dis and within_10Perc scores, SF:1718-1726), result_organize_ins (SF:1219-1231; the mean of
// the positive scores in numpy's pairwise order) and gt_estimate_log_likelihood (SF:2054-2069)
// through a (k, l) -> (GT, GQ) table the host fills with the reference's own float64 operations.
struct DRead {            // 32 B: one read of one locus
    int32_t ref_a, alt_a; // pair indices for scorer A (kind 1 abs_dis_m1b, 2 within_10Perc, 3 directed_dis)
    int32_t ref_b, alt_b; // kind 0 (DEL): scorer A = abs_dis_m1b, scorer B = within_10Perc on these pairs
    int32_t kind, locus, len_ref, len_alt;
};

#define VAPOR_GT_TABLE_N 65   // k, l in 0..64

__device__ __forceinline__ bool score_s1(const long long* r, const long long* a, double lr, double la, double* oa, double* ob)
{
    *oa = 0.0; *ob = 0.0;
    if (r[0] > 2 && a[0] > 2 && (double)r[0] / fmin(lr, la) > 0.1) {
        const bool r_ok = (double)(r[2] - r[1]) / lr > 0.6, a_ok = (double)(a[2] - a[1]) / la > 0.6;
        if (r_ok && a_ok) {
            if (r[3] > 0 && a[3] > 0) { *oa = (double)r[4] / (double)r[3]; *ob = (double)a[4] / (double)a[3]; }
        } else if (r_ok) { *oa = 1.1; *ob = 2.1; }
        else if (a_ok) { *oa = 2.1; *ob = 1.1; }
    }
    return *oa != 0.0 && *ob != 0.0;
}

__device__ __forceinline__ bool score_s2(const long long* r, const long long* a, double lr, double la, double* oa, double* ob)
{
    *oa = 0.0; *ob = 0.0;
    if (fmax((double)r[0] / lr, (double)a[0] / la) > 0.1 && r[5] > 0 && a[5] > 0) { *oa = (double)a[6]; *ob = (double)r[6]; }
    return *oa != 0.0 && *ob != 0.0;
}

__device__ __forceinline__ double dir_value(const long long* s)
{
    return s[11] == 0 ? 0.0001 : fabs(((double)s[12] * 0.5) / (double)s[11]);
}

__device__ __forceinline__ bool score_s3(const long long* r, const long long* a, double lr, double la, double* oa, double* ob)
{
    *oa = 0.0; *ob = 0.0;
    if ((double)r[0] / lr > 0.1 && (double)a[0] / la > 0.1 && (double)(r[2] - r[1]) / lr > 0.7 &&
        (double)(a[2] - a[1]) / la > 0.7 && r[3] > 0 && a[3] > 0) { *oa = dir_value(r); *ob = dir_value(a); }
    return *oa != 0.0 && *ob != 0.0;
}

// one wave per locus; reads of a locus are contiguous (locus_first[l] .. locus_first[l+1])
__global__ __launch_bounds__(64, 8) void finish_kernel(const DRead* __restrict__ reads, const int32_t* __restrict__ locus_first,
                                                   const long long* __restrict__ stats, const double* __restrict__ gt_table,
                                                   double* __restrict__ read_scores, double* __restrict__ loci_out,
                                                   double* __restrict__ loci_out2)
{
    // loci_out2 (may be NULL): a second copy of the locus records, e.g. pinned host memory, so that an asynchronous step
    // needs no copy kernel behind this one.
    // One wave per locus, no LDS and at most 64 registers: the scores stay in the lanes that computed them and are
    // handed round by v_readlane, so that the wave fits beside a join workgroup (which owns its CU's LDS and all but 64
    // registers per SIMD) instead of sitting on the stream until another plan's join is over.
    const int l = blockIdx.x, lane = threadIdx.x;
    const int r0 = locus_first[l], r1 = locus_first[l + 1];
    int n = 0, npos = 0, nnonpos = 0;
    double res = 0.0;
    // numpy.add.reduce order for the mean of the positives (numpy's pairwise_sum): fewer than 8 values are added one
    // by one; otherwise eight strided partial sums over the first n - n%8 values are combined as
    // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the last n%8 values are added after that.  That is numpy's order for up
    // to 128 values (its block size; above it numpy halves recursively), i.e. VaPoR_QS is bit-identical to np.mean for
    // loci of up to 128 positive scores - the reference keeps at most 20 reads per locus (SF:1091), the 60-read
    // loci of BASELINE's largest configuration are covered by tests/golden/deep_loci.json.gz.  Longer synthetic
    // lists are summed in chunks of 256 reads (a few ulp from numpy); VaPoR_GT / GQ need at most 64 scored reads
    // (the (k, l) table), beyond that the genotype falls back to 0/1 without a quality.
    // The ordered sum runs in every lane on broadcast values (uniform control flow); lane 0 stores.
    auto bcast = [](double v, int from) -> double {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, from);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), from);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    for (int base = r0; base < r1; base += 256) {
        const int cnt = min(256, r1 - base);
        double vals[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = q * 64 + lane;
            double s = __builtin_nan("");
            if (t < cnt) {
                const DRead rd = reads[base + t];
                const double lr = (double)rd.len_ref, la = (double)rd.len_alt;
                double a, b;
                if (rd.kind == 0) {
                    double a2, b2;
                    const bool v1 = score_s1(stats + 16LL * rd.ref_a, stats + 16LL * rd.alt_a, lr, la, &a, &b);
                    const bool v2 = score_s2(stats + 16LL * rd.ref_b, stats + 16LL * rd.alt_b, lr, la, &a2, &b2);
                    const double s1 = 1.0 - b / a, s2 = 1.0 - b2 / a2;
                    if (v1 && v2) s = (s2 < s1) ? s2 : s1;
                    else if (v1) s = s1;
                    else if (v2) s = s2;
                } else {
                    const long long* r = stats + 16LL * rd.ref_a;
                    const long long* q2 = stats + 16LL * rd.alt_a;
                    const bool ok = rd.kind == 1 ? score_s1(r, q2, lr, la, &a, &b) : rd.kind == 2 ? score_s2(r, q2, lr, la, &a, &b)
                                                                                             : score_s3(r, q2, lr, la, &a, &b);
                    if (ok) s = 1.0 - b / a;
                }
                if (read_scores) read_scores[base + t] = s;
            }
            vals[q] = s;                              // NaN marks a skipped read (and the slots past the chunk)
        }
        int cpos = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v = vals[q];
            n += __popcll(__ballot(v == v));
            nnonpos += __popcll(__ballot((v == v) & !(v >= 0.005)));     // round(v, 2) > 0  <=>  v >= 0.005
            cpos += __popcll(__ballot(v > 0.0));
        }
        double part = 0.0;
        double r8a = 0.0, r8b = 0.0, r8c = 0.0, r8d = 0.0, r8e = 0.0, r8f = 0.0, r8g = 0.0, r8h = 0.0;
        const int full = cpos < 8 ? 0 : cpos - (cpos % 8);
        int idx = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = min(64, cnt - q * 64);
            for (int u = 0; u < m; ++u) {
                const double v = bcast(vals[q], u);
                if (!(v > 0.0)) continue;
                if (idx < full) {
                    // (first visit of a partial sum assigns: 0.0 + v == v bit for bit, the positives are never -0.0)
                    switch (idx & 7) {
                        case 0: r8a += v; break; case 1: r8b += v; break; case 2: r8c += v; break; case 3: r8d += v; break;
                        case 4: r8e += v; break; case 5: r8f += v; break; case 6: r8g += v; break; default: r8h += v; break;
                    }
                } else {
                    if (idx == full && full) part = ((r8a + r8b) + (r8c + r8d)) + ((r8e + r8f) + (r8g + r8h));
                    part += v;
                }
                ++idx;
            }
        }
        if (full && cpos == full) part = ((r8a + r8b) + (r8c + r8d)) + ((r8e + r8f) + (r8g + r8h));
        res += part;
        npos += cpos;
    }
    if (lane == 0) {
        double* o = loci_out + 8LL * l;
        double* o2 = loci_out2 ? loci_out2 + 8LL * l : o;
        if (n == 0) {
            for (int t = 0; t < 8; ++t) o2[t] = o[t] = __builtin_nan("");
            o2[4] = o[4] = 0.0;
            return;
        }
        const double qs = npos > 0 ? res / (double)npos : 0.0;
        const double gs = (double)npos / (double)n;
        int gt = 1;
        double gq = __builtin_nan("");
        if (n < VAPOR_GT_TABLE_N) {
            gt = (int)gt_table[2 * (n * VAPOR_GT_TABLE_N + nnonpos)];
            gq = gt_table[2 * (n * VAPOR_GT_TABLE_N + nnonpos) + 1];
        }
        if (gt == 0 && gs > 0.15) gt = 1;
        o[0] = qs; o[1] = gs; o[2] = (double)gt; o[3] = gq; o[4] = (double)n; o[5] = (double)npos; o[6] = (double)nnonpos; o[7] = 0.0;
        if (loci_out2) {
            o2[0] = qs; o2[1] = gs; o2[2] = (double)gt; o2[3] = gq; o2[4] = (double)n; o2[5] = (double)npos; o2[6] = (double)nnonpos; o2[7] = 0.0;
        }
    }
}

}  // namespace vapor
