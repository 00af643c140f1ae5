/*
 * cpu_twin.cpp - libvapor_cpu.so: the C ABI of include/vapor_hip.h on the CPU oracle ("the same
 * symbols are exported by a CPU build of the library", for timing the CPU path through the same boundary and for
 * exercising the ctypes bindings and the Python object layer without a GPU).
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: nothing in vapor_amd/ loads it (vapor_amd._lib refuses to
 * fall back to anything when libvapor_hip.so is missing); tests/test_cpu_twin.py binds it explicitly.  The numbers
 * come from vapor_oracle.c (k-mer join, gap clustering, counts: each function cites its SF lines) and, for the
 * directed statistics and the per-locus finish, from the float64 restatements below, which follow the reference's
 * own floating-point steps (SF = /root/reference/vapor_vali/Simple_function.pyx) and NOT the integer reformulation
 * the HIP kernels use - so agreement between the two is evidence, not an identity.
 *
 * Build: oracle/oracle.py build_twin()  (g++ -O2 -shared cpu_twin.cpp vapor_oracle.c)
 */
#include "../include/vapor_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int vo_dotdata(int k, const char* s1, int n1, const char* s2, int n2, int32_t* hits_ji, int64_t cap, int64_t* n_hits);
int vo_clean_c1(const int32_t* hits_ji, int64_t n, uint8_t* keep);
int vo_clean_c2(const int32_t* hits_ji, int64_t n, uint8_t* keep);
int vo_pair_stats(int k, const char* s1, int n1, const char* s2, int n2, int64_t* st, int32_t* hits_ji, int64_t cap,
                  uint8_t* keep1, uint8_t* keep2);
}

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }

struct vapor_ctx { int device; };
struct vapor_seqset {
    std::vector<std::string> seq;
    std::vector<int32_t> n_exc, n_invalid;
};
struct PairRes {
    int64_t st[16];
    std::vector<int32_t> hits;      // (j, i) in dotdata's order
    std::vector<uint8_t> k1, k2;
};
struct vapor_plan {
    vapor_seqset* set;
    std::vector<vapor_pair> pairs;
    std::vector<PairRes> res;
    bool ran = false;
    std::vector<vapor_read> reads;
    int64_t n_loci = 0;
    std::vector<double> gt, loci, read_scores;
};

// ---- symbols --------------------------------------------------------------------------------
static bool is_acgt(unsigned char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
static bool in_alphabet(unsigned char c)        // invert_base's alphabet after key_modify's IUPAC folding (SF:19-20, 908-949)
{
    return strchr("ACGTNRYSWKMBDHVacgtnryswkmbdhv", c) != nullptr && c != 0;
}

extern "C" int vapor_abi_version(void) { return VAPOR_ABI_VERSION; }
// says what it is: the product loader (vapor_amd/_lib.py) refuses a library with build flags unless VAPOR_ALLOW_TWIN=1
extern "C" const char* vapor_build_flags(void) { return "cpu-twin"; }
extern "C" const char* vapor_source_id(void) { return "cpu-twin"; }
extern "C" const char* vapor_last_error(void) { return g_err.c_str(); }
extern "C" int vapor_init(int device_ordinal, vapor_ctx** ctx)
{
    if (!ctx) return fail(VAPOR_E_ARG, "vapor_init: null out pointer");
    if (device_ordinal != 0) return fail(VAPOR_E_ARG, "vapor_init: the CPU twin has one device");
    *ctx = new vapor_ctx{0};
    return VAPOR_OK;
}
extern "C" int vapor_destroy(vapor_ctx* c) { delete c; return VAPOR_OK; }
extern "C" int vapor_set_param(vapor_ctx* c, const char* name, int64_t v)
{
    if (!c || !name) return fail(VAPOR_E_ARG, "vapor_set_param: null argument");
    if (!strcmp(name, "reads_per_task") || !strcmp(name, "join_tasks") || !strcmp(name, "max_pair_cap"))
        return v >= 1 ? VAPOR_OK : fail(VAPOR_E_ARG, "parameter out of range");
    if (!strcmp(name, "shared_join") || !strcmp(name, "stage_threads") || !strcmp(name, "clean_order") || !strcmp(name, "clean_fit") || !strcmp(name, "remap_in_clean")) return VAPOR_OK;
    if (!strcmp(name, "bam_cu_share")) return v >= 0 && v <= 8 ? VAPOR_OK : fail(VAPOR_E_ARG, "parameter out of range");
    return fail(VAPOR_E_ARG, std::string("unknown parameter ") + name);
}
extern "C" int vapor_set_stream(vapor_ctx* c, void*) { return c ? VAPOR_OK : fail(VAPOR_E_ARG, "null context"); }

template <typename SRC>
static int seqset_make(int32_t n, SRC src, const int32_t* len, const uint8_t* flags, int32_t* info, vapor_seqset** out)
{
    vapor_seqset* s = new vapor_seqset();
    for (int32_t i = 0; i < n; ++i) {
        if (len[i] < 0) { delete s; return fail(VAPOR_E_ARG, "negative sequence length"); }
        std::string q(reinterpret_cast<const char*>(src(i)), (size_t)len[i]);
        if (flags && (flags[i] & VAPOR_SEQ_UPPER))
            for (char& ch : q) if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);          // str.upper() on ASCII
        int32_t ne = 0, ni = 0;
        for (unsigned char ch : q) { ne += !is_acgt(ch); ni += !in_alphabet(ch); }
        s->seq.push_back(q); s->n_exc.push_back(ne); s->n_invalid.push_back(ni);
        if (info) { info[2 * i] = ne; info[2 * i + 1] = ni; }
    }
    *out = s;
    return VAPOR_OK;
}
extern "C" int vapor_seqset_create(vapor_ctx* ctx, int32_t n, const uint8_t* blob, const int64_t* off, const int32_t* len,
                                   const uint8_t* flags, int32_t* info, vapor_seqset** out)
{
    if (!ctx || !out || n < 0 || (n && (!blob || !off || !len))) return fail(VAPOR_E_ARG, "vapor_seqset_create: null argument");
    return seqset_make(n, [&](int32_t i) { return blob + off[i]; }, len, flags, info, out);
}
extern "C" int vapor_seqset_create_ptrs(vapor_ctx* ctx, int32_t n, const uint8_t* const* seq, const int32_t* len,
                                        const uint8_t* flags, int32_t* info, vapor_seqset** out)
{
    if (!ctx || !out || n < 0 || (n && (!seq || !len))) return fail(VAPOR_E_ARG, "vapor_seqset_create_ptrs: null argument");
    for (int32_t i = 0; i < n; ++i)
        if (len[i] > 0 && !seq[i]) return fail(VAPOR_E_ARG, "vapor_seqset_create_ptrs: null sequence");
    return seqset_make(n, [&](int32_t i) { return seq[i]; }, len, flags, info, out);
}
// derived sequences (include/vapor_hip.h): materialised here the way the reference builds them - slices, reverse(complementary())
// of slices (SF:471-478; a character complementary() would drop is refused, as the HIP library refuses it), str.upper()
extern "C" int vapor_seqset_create_derived(vapor_ctx* ctx, int32_t n, const uint8_t* const* seq, const int32_t* len,
                                           const uint8_t* flags, int32_t n_derived, const int32_t* seg_first,
                                           const vapor_segment* segs, const uint8_t* derived_flags, int32_t* info, vapor_seqset** out)
{
    if (!ctx || !out || n < 0 || n_derived < 0 || (n && (!seq || !len)) || (n_derived && (!seg_first || !segs)))
        return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: null argument");
    for (int32_t i = 0; i < n; ++i)
        if (len[i] > 0 && !seq[i]) return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: null sequence");
    std::vector<std::string> text((size_t)n_derived);
    for (int32_t d = 0; d < n_derived; ++d) {
        if (seg_first[d + 1] < seg_first[d] || seg_first[d + 1] - seg_first[d] > VAPOR_MAX_SEGMENTS)
            return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: bad segment count");
        for (int32_t g = seg_first[d]; g < seg_first[d + 1]; ++g) {
            const vapor_segment& x = segs[g];
            if (x.parent < 0 || x.parent >= n || x.off < 0 || x.len < 0 || (int64_t)x.off + x.len > len[x.parent])
                return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: segment outside its parent");
            if (x.len > 0 && flags && (flags[x.parent] & VAPOR_SEQ_UPPER) && !(derived_flags && (derived_flags[d] & VAPOR_SEQ_UPPER)))
                return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: a derived sequence without VAPOR_SEQ_UPPER over a parent uploaded with it");
            std::string piece(reinterpret_cast<const char*>(seq[x.parent]) + x.off, (size_t)x.len);
            if (flags && (flags[x.parent] & VAPOR_SEQ_UPPER))
                for (char& ch : piece) if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
            if (x.flags & VAPOR_SEG_REVCOMP) {
                // (the library looks at the whole parent, not the slice: the same rule here)
                for (int32_t t = 0; t < len[x.parent]; ++t)
                    if (!strchr("ATGCNatgcn", seq[x.parent][t]) || seq[x.parent][t] == 0)
                        return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: a reverse-complemented segment's parent holds characters complementary() drops");
                std::string r(piece.rbegin(), piece.rend());
                for (char& ch : r) {
                    switch (ch) {
                    case 'A': ch = 'T'; break; case 'T': ch = 'A'; break; case 'C': ch = 'G'; break; case 'G': ch = 'C'; break;
                    case 'a': ch = 't'; break; case 't': ch = 'a'; break; case 'c': ch = 'g'; break; case 'g': ch = 'c'; break;
                    default: break;
                    }
                }
                piece = r;
            }
            text[(size_t)d] += piece;
        }
    }
    std::vector<const uint8_t*> ptr((size_t)(n + n_derived));
    std::vector<int32_t> ln((size_t)(n + n_derived));
    std::vector<uint8_t> fl((size_t)(n + n_derived), 0);
    for (int32_t i = 0; i < n; ++i) { ptr[(size_t)i] = seq[i]; ln[(size_t)i] = len[i]; fl[(size_t)i] = flags ? flags[i] : 0; }
    for (int32_t d = 0; d < n_derived; ++d) {
        ptr[(size_t)(n + d)] = reinterpret_cast<const uint8_t*>(text[(size_t)d].data());
        ln[(size_t)(n + d)] = (int32_t)text[(size_t)d].size();
        fl[(size_t)(n + d)] = derived_flags ? derived_flags[d] : 0;
    }
    return seqset_make(n + n_derived, [&](int32_t i) { return ptr[(size_t)i]; }, ln.data(), fl.data(), info, out);
}
// the planes the HIP library's pack_kernel writes, restated from the text (symbol codes: 0-3 ACGT, 4-7 acgt, 8 N and folded
// IUPAC, 9 n and folded lower-case IUPAC, 15 anything else; include/vapor_hip.h)
extern "C" int vapor_seqset_planes(vapor_seqset* s, int32_t seq, uint32_t* p2, uint32_t* e1, uint32_t* x4)
{
    if (!s || seq < 0 || (size_t)seq >= s->seq.size()) return fail(VAPOR_E_ARG, "vapor_seqset_planes: no such sequence");
    const std::string& q = s->seq[(size_t)seq];
    const size_t ch = (q.size() + 31) / 32;
    if (p2) memset(p2, 0, ch * 8);
    if (e1) memset(e1, 0, ch * 4);
    if (x4) memset(x4, 0, ch * 16);
    for (size_t t = 0; t < q.size(); ++t) {
        const unsigned char c = (unsigned char)q[t];
        const bool lower = c >= 'a' && c <= 'z';
        const unsigned char u = lower ? (unsigned char)(c - 32) : c;
        uint32_t code = 15u;
        if (u == 'A') code = 0; else if (u == 'C') code = 1; else if (u == 'G') code = 2; else if (u == 'T') code = 3;
        else if (strchr("NRYSWKMBDHV", u) && u) code = 8;
        if (lower && code != 15u) code = code < 4 ? code + 4 : 9;
        if (x4) x4[t >> 3] |= code << ((t & 7) * 4);
        if (p2) p2[t >> 4] |= (code < 8 ? (code & 3u) : 0u) << ((t & 15) * 2);
        if (e1 && code >= 4) e1[t >> 5] |= 1u << (t & 31);
    }
    return VAPOR_OK;
}
extern "C" int vapor_seqset_destroy(vapor_seqset* s) { delete s; return VAPOR_OK; }

// ---- directed statistics in the reference's floating point (SF:582-591, 1104-1118, 788-792, 710-722) ----
static std::vector<std::vector<int>> number_cluster(std::vector<int> v, const double* edge /* 11 */)
{
    std::vector<std::vector<int>> bins(11);
    std::sort(v.begin(), v.end());
    size_t a = 0;
    int b = 1;
    while (a < v.size() && b < 11) {
        if ((double)v[a] < edge[b]) bins[(size_t)b - 1].push_back(v[a++]);
        else ++b;
    }
    for (; a < v.size(); ++a) bins[10].push_back(v[a]);
    return bins;
}
static std::vector<std::vector<int>> find_longest(const std::vector<std::vector<int>>& sets)
{
    size_t m = 0;
    for (auto& x : sets) m = std::max(m, x.size());
    std::vector<std::vector<int>> out;
    for (auto& x : sets)
        if (x.size() == m && std::find(out.begin(), out.end(), x) == out.end()) out.push_back(x);
    return out;
}
static void edges_of(const std::vector<int>& v, double* e)
{
    const int lo = *std::min_element(v.begin(), v.end()), hi = *std::max_element(v.begin(), v.end());
    for (int t = 0; t < 11; ++t) e[t] = (double)lo + (double)t * (double)(hi - lo) / 10.0;
}
static void dir_stats(const PairRes& r, int64_t* st)
{
    std::vector<int> d;
    std::vector<std::pair<int, int>> kept;
    for (size_t t = 0; t < r.k1.size(); ++t)
        if (r.k1[t]) { kept.push_back({r.hits[2 * t], r.hits[2 * t + 1]}); d.push_back(r.hits[2 * t + 1] - r.hits[2 * t]); }
    if (kept.empty()) return;
    double e[11];
    edges_of(d, e);
    auto kept1 = find_longest(number_cluster(d, e));
    std::vector<std::vector<int>> kept2;
    for (auto& km : kept1) {
        edges_of(km, e);
        for (auto& x : find_longest(number_cluster(km, e))) kept2.push_back(x);
    }
    double c = 0.0;
    if (kept2.size() == 1) {
        const auto& m = kept2[0];                                  // sorted by number_cluster
        c = m.size() % 2 ? (double)m[m.size() / 2] : 0.5 * ((double)m[m.size() / 2 - 1] + (double)m[m.size() / 2]);
    }
    int64_t n = 0;
    double sum = 0.0;
    for (auto& ji : kept) {
        const double x = (double)ji.first + c, y = (double)ji.second;
        const double rel = x == 0.0 ? std::fabs((x - y) / (x + 1.0)) : std::fabs((x - y) / x);
        if (rel > 0.1) { ++n; sum += x - y; }
    }
    st[VAPOR_ST_DIR_C2X] = (int64_t)std::llround(2.0 * c);
    st[VAPOR_ST_DIR_N] = n;
    st[VAPOR_ST_DIR_SUM2] = (int64_t)std::llround(2.0 * sum);
    st[VAPOR_ST_DIR_LISTS] = (int64_t)kept2.size();
}

// ---- plans ----------------------------------------------------------------------------------
static bool k_ok(int k) { return k == 10 || k == 20 || k == 30 || k == 40; }

extern "C" int vapor_plan_create(vapor_ctx* ctx, vapor_seqset* set, int64_t n, const vapor_pair* pairs, vapor_plan** out)
{
    if (!ctx || !set || !out || n < 0 || (n && !pairs)) return fail(VAPOR_E_ARG, "vapor_plan_create: null argument");
    vapor_plan* p = new vapor_plan();
    p->set = set;
    p->pairs.assign(pairs, pairs + n);
    p->res.resize((size_t)n);
    *out = p;
    return VAPOR_OK;
}
extern "C" int vapor_plan_destroy(vapor_plan* p) { delete p; return VAPOR_OK; }

static void mask_flags(uint32_t fl, int64_t* st)
{
    if (!(fl & VAPOR_PF_C1)) st[3] = st[4] = 0;
    if (!(fl & VAPOR_PF_C2)) st[5] = st[6] = st[9] = 0;
}

static int run_pair(const vapor_seqset* s, const vapor_pair& a, PairRes& r)
{
    for (int t = 0; t < 16; ++t) r.st[t] = 0;
    r.st[1] = r.st[2] = -1;
    r.hits.clear(); r.k1.clear(); r.k2.clear();
    const int32_t ns = (int32_t)s->seq.size();
    if (a.seq1 < 0 || a.seq1 >= ns || a.seq2 < 0 || a.seq2 >= ns || a.off2 < 0 || !k_ok(a.k)) { r.st[15] = VAPOR_E_ARG; return 0; }
    const std::string &s1 = s->seq[(size_t)a.seq1], &s2 = s->seq[(size_t)a.seq2];
    if ((int64_t)s1.size() > VAPOR_MAX_SEQ_LEN || (int64_t)s2.size() > VAPOR_MAX_SEQ_LEN) { r.st[15] = VAPOR_E_ARG; return 0; }
    if ((int64_t)s1.size() - a.k + 1 > 0 && s->n_invalid[(size_t)a.seq1] > 0) { r.st[15] = VAPOR_E_KEYERROR; return 0; }
    const int off = std::min<int64_t>(a.off2, (int64_t)s2.size());
    const char* p2 = s2.data() + off;
    const int n2 = (int)s2.size() - off;
    int64_t n = 0;
    int rc = vo_dotdata(a.k, s1.data(), (int)s1.size(), p2, n2, nullptr, 0, &n);
    if (rc == -3) { r.st[15] = VAPOR_E_KEYERROR; return 0; }
    r.hits.resize((size_t)std::max<int64_t>(n, 1) * 2);
    r.k1.assign((size_t)std::max<int64_t>(n, 1), 0);
    r.k2.assign((size_t)std::max<int64_t>(n, 1), 0);
    rc = vo_pair_stats(a.k, s1.data(), (int)s1.size(), p2, n2, r.st, r.hits.data(), std::max<int64_t>(n, 1), r.k1.data(), r.k2.data());
    if (rc != 0) return fail(VAPOR_E_NOMEM, "oracle failure");
    r.hits.resize((size_t)n * 2); r.k1.resize((size_t)n); r.k2.resize((size_t)n);
    mask_flags(a.flags, r.st);
    if ((a.flags & VAPOR_PF_DIR) && (a.flags & VAPOR_PF_C1) && r.st[3] > 0) dir_stats(r, r.st);
    return 0;
}

extern "C" int vapor_plan_run(vapor_plan* p, int64_t* stats)
{
    if (!p || (!p->pairs.empty() && !stats)) return fail(VAPOR_E_ARG, "vapor_plan_run: null argument");
    for (size_t i = 0; i < p->pairs.size(); ++i) {
        int rc = run_pair(p->set, p->pairs[i], p->res[i]);
        if (rc != 0) return rc;
        memcpy(stats + 16 * i, p->res[i].st, sizeof(int64_t) * 16);
    }
    p->ran = true;
    return VAPOR_OK;
}
extern "C" int vapor_plan_timings(vapor_plan* p, double* ms, int32_t n)
{
    if (!p || !ms) return fail(VAPOR_E_ARG, "vapor_plan_timings: null argument");
    for (int i = 0; i < n && i < 10; ++i) ms[i] = 0.0;
    return VAPOR_OK;
}
extern "C" int vapor_plan_record_counts(vapor_plan* p, int64_t* rec)
{
    if (!p || !rec || !p->ran) return fail(VAPOR_E_ARG, "vapor_plan_record_counts: plan has not been run");
    for (size_t i = 0; i < p->res.size(); ++i) rec[i] = p->res[i].st[15] == 0 ? p->res[i].st[0] : 0;   // one record per dot here
    return VAPOR_OK;
}
extern "C" int vapor_plan_algorithmic_bytes(vapor_plan* p, int64_t* bytes, int64_t* cells)
{
    if (!p || !bytes || !cells) return fail(VAPOR_E_ARG, "null argument");
    int64_t b = 0, c = 0;
    for (size_t i = 0; i < p->pairs.size(); ++i) {
        if (p->ran && p->res[i].st[15] != 0) continue;
        const vapor_pair& a = p->pairs[i];
        if (a.seq1 < 0 || a.seq2 < 0 || a.seq1 >= (int32_t)p->set->seq.size() || a.seq2 >= (int32_t)p->set->seq.size()) continue;
        const int64_t n1 = (int64_t)p->set->seq[(size_t)a.seq1].size();
        const int64_t n2 = std::max<int64_t>(0, (int64_t)p->set->seq[(size_t)a.seq2].size() - a.off2);
        b += (3 * n1 + 7) / 8 + (3 * n2 + 7) / 8 + 8 * (p->ran ? p->res[i].st[0] : 0) + 128;
        c += n1 * n2;
    }
    *bytes = b; *cells = c;
    return VAPOR_OK;
}
extern "C" int vapor_plan_fetch_hits(vapor_plan* p, int64_t n_sel, const int64_t* idx, int32_t* ji, uint8_t* fl, int64_t cap, int64_t* off)
{
    if (!p || n_sel < 0 || (n_sel && (!idx || !off))) return fail(VAPOR_E_ARG, "vapor_plan_fetch_hits: null argument");
    if (!p->ran) return fail(VAPOR_E_ARG, "vapor_plan_fetch_hits: plan has not been run");
    int64_t tot = 0;
    for (int64_t q = 0; q < n_sel; ++q) {
        if (idx[q] < 0 || idx[q] >= (int64_t)p->res.size()) return fail(VAPOR_E_ARG, "pair index out of range");
        off[q] = tot;
        const PairRes& r = p->res[(size_t)idx[q]];
        tot += r.st[15] == 0 ? (int64_t)r.k1.size() : 0;
    }
    off[n_sel] = tot;
    if (tot > cap) return fail(VAPOR_E_OVERFLOW, "hit buffer too small");
    if (tot && !ji) return fail(VAPOR_E_ARG, "null hit buffer");
    for (int64_t q = 0; q < n_sel; ++q) {
        const PairRes& r = p->res[(size_t)idx[q]];
        if (r.st[15] != 0) continue;
        memcpy(ji + 2 * off[q], r.hits.data(), sizeof(int32_t) * r.hits.size());
        if (fl)
            for (size_t t = 0; t < r.k1.size(); ++t)
                fl[off[q] + (int64_t)t] = (uint8_t)((r.k1[t] ? VAPOR_HF_C1_KEPT : 0) | (r.k2[t] == 1 ? VAPOR_HF_C2_DIAG : 0) |
                                                    (r.k2[t] == 2 ? VAPOR_HF_C2_ANTI : 0));
    }
    return VAPOR_OK;
}

extern "C" int vapor_score_batch(vapor_ctx* ctx, vapor_seqset* set, int64_t n, const vapor_pair* pairs, int64_t* stats)
{
    vapor_plan* p = nullptr;
    int rc = vapor_plan_create(ctx, set, n, pairs, &p);
    if (rc == VAPOR_OK) rc = vapor_plan_run(p, stats);
    vapor_plan_destroy(p);
    return rc;
}
extern "C" int vapor_dotplot_batch(vapor_ctx* ctx, vapor_seqset* set, int64_t n, const vapor_pair* pairs, int32_t* ji, int64_t cap,
                                   int64_t* off, int64_t* stats)
{
    if (!off) return fail(VAPOR_E_ARG, "vapor_dotplot_batch: null hit_off");
    vapor_plan* p = nullptr;
    int rc = vapor_plan_create(ctx, set, n, pairs, &p);
    std::vector<int64_t> st((size_t)std::max<int64_t>(n, 1) * 16), idx((size_t)n);
    if (rc == VAPOR_OK) rc = vapor_plan_run(p, st.data());
    if (rc == VAPOR_OK) {
        for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = i;
        rc = vapor_plan_fetch_hits(p, n, idx.data(), ji, nullptr, cap, off);
        if (stats) memcpy(stats, st.data(), sizeof(int64_t) * 16 * (size_t)n);
    }
    vapor_plan_destroy(p);
    return rc;
}
extern "C" int vapor_selfplot_qc(vapor_ctx* ctx, vapor_seqset* set, int32_t n, const int32_t* seq_idx, const int32_t* k, int64_t* out)
{
    if (n < 0 || (n && (!seq_idx || !k || !out))) return fail(VAPOR_E_ARG, "vapor_selfplot_qc: null argument");
    std::vector<vapor_pair> pr((size_t)n);
    for (int32_t t = 0; t < n; ++t) pr[(size_t)t] = vapor_pair{seq_idx[t], seq_idx[t], 0, k[t], 0u};
    std::vector<int64_t> st((size_t)std::max(n, 1) * 16);
    int rc = vapor_score_batch(ctx, set, n, pr.data(), st.data());
    if (rc != VAPOR_OK) return rc;
    for (int32_t t = 0; t < n; ++t) {
        if (st[16 * (size_t)t + 15] != 0) return fail((int)st[16 * (size_t)t + 15], "vapor_selfplot_qc: pair failed");
        out[3 * t] = st[16 * (size_t)t]; out[3 * t + 1] = st[16 * (size_t)t + 7]; out[3 * t + 2] = st[16 * (size_t)t + 8];
    }
    return VAPOR_OK;
}

extern "C" int vapor_clean_hits(vapor_ctx* ctx, int64_t n_lists, const int32_t* ji, const int64_t* off, const uint32_t* flags,
                                int64_t* stats, uint8_t* hit_flags)
{
    if (!ctx || n_lists < 0 || (n_lists && (!off || !stats))) return fail(VAPOR_E_ARG, "vapor_clean_hits: null argument");
    for (int64_t t = 0; t < n_lists; ++t) {
        const int64_t n = off[t + 1] - off[t];
        PairRes r;
        for (int q = 0; q < 16; ++q) r.st[q] = 0;
        r.st[1] = r.st[2] = -1;
        r.hits.assign(ji + 2 * off[t], ji + 2 * off[t + 1]);
        r.k1.assign((size_t)n, 0); r.k2.assign((size_t)n, 0);
        for (int64_t h = 0; h < n; ++h)
            if (r.hits[2 * (size_t)h] < 0 || r.hits[2 * (size_t)h + 1] < 0 || r.hits[2 * (size_t)h] > VAPOR_MAX_SEQ_LEN || r.hits[2 * (size_t)h + 1] > VAPOR_MAX_SEQ_LEN)
                return fail(VAPOR_E_ARG, "vapor_clean_hits: coordinate out of range");
        const uint32_t fl = flags ? flags[t] : 3u;
        if (n > 0) {
            if (vo_clean_c1(r.hits.data(), n, r.k1.data()) != 0 || vo_clean_c2(r.hits.data(), n, r.k2.data()) != 0)
                return fail(VAPOR_E_NOMEM, "oracle failure");
            int mn = r.hits[0], mx = r.hits[0];
            r.st[0] = n;
            for (int64_t h = 0; h < n; ++h) {
                const int64_t j = r.hits[2 * (size_t)h], i = r.hits[2 * (size_t)h + 1], ad = j > i ? j - i : i - j;
                mn = std::min<int>(mn, (int)j); mx = std::max<int>(mx, (int)j);
                if (r.k1[(size_t)h]) { r.st[3]++; r.st[4] += ad; }
                if (r.k2[(size_t)h]) { r.st[5]++; if (j > 0 && 25 * ad < 4 * j) r.st[6]++; }
                if (r.k2[(size_t)h] == 1) r.st[9]++;
                if (j == i) r.st[7]++; else if (j > i) r.st[8]++;
            }
            r.st[1] = mn; r.st[2] = mx;
            mask_flags(fl, r.st);
            if (!(fl & VAPOR_PF_C1)) std::fill(r.k1.begin(), r.k1.end(), 0);
            if (!(fl & VAPOR_PF_C2)) std::fill(r.k2.begin(), r.k2.end(), 0);
            if ((fl & VAPOR_PF_DIR) && (fl & VAPOR_PF_C1) && r.st[3] > 0) dir_stats(r, r.st);
        }
        memcpy(stats + 16 * t, r.st, sizeof(int64_t) * 16);
        if (hit_flags)
            for (int64_t h = 0; h < n; ++h)
                hit_flags[off[t] + h] = (uint8_t)((r.k1[(size_t)h] ? VAPOR_HF_C1_KEPT : 0) | (r.k2[(size_t)h] == 1 ? VAPOR_HF_C2_DIAG : 0) |
                                                  (r.k2[(size_t)h] == 2 ? VAPOR_HF_C2_ANTI : 0));
    }
    return VAPOR_OK;
}

// ---- per-read scores and per-locus results (SF:182-203, 241-257, 277-294, 1718-1726, 1219-1231, 2054-2069) ----
extern "C" int vapor_plan_set_reads(vapor_plan* p, int64_t n_reads, const vapor_read* reads, int64_t n_loci, const double* gt)
{
    if (!p || n_reads < 0 || n_loci < 0 || (n_reads && !reads) || !gt) return fail(VAPOR_E_ARG, "vapor_plan_set_reads: null argument");
    int32_t prev = -1;
    for (int64_t r = 0; r < n_reads; ++r) {
        const vapor_read& x = reads[r];
        if (x.locus < prev || x.locus >= n_loci) return fail(VAPOR_E_ARG, "reads must be sorted by locus");
        if (x.kind < 0 || x.kind > 3) return fail(VAPOR_E_ARG, "unknown scorer kind");
        const int32_t q[4] = {x.ref_a, x.alt_a, x.kind == 0 ? x.ref_b : x.ref_a, x.kind == 0 ? x.alt_b : x.alt_a};
        for (int32_t v : q)
            if (v < 0 || v >= (int32_t)p->pairs.size()) return fail(VAPOR_E_ARG, "read refers to a pair outside the plan");
        prev = x.locus;
    }
    p->reads.assign(reads, reads + n_reads);
    p->n_loci = n_loci;
    p->gt.assign(gt, gt + 2 * VAPOR_GT_TABLE_N * VAPOR_GT_TABLE_N);
    return VAPOR_OK;
}

static bool s1(const int64_t* r, const int64_t* a, double lr, double la, double* x, double* y)
{
    *x = *y = 0.0;
    if (r[15] || a[15]) return false;
    if (r[0] > 2 && a[0] > 2 && (double)r[0] / std::min(lr, la) > 0.1) {
        const bool rok = (double)(r[2] - r[1]) / lr > 0.6, aok = (double)(a[2] - a[1]) / la > 0.6;
        if (rok && aok) { if (r[3] > 0 && a[3] > 0) { *x = (double)r[4] / (double)r[3]; *y = (double)a[4] / (double)a[3]; } }
        else if (rok) { *x = 1.1; *y = 2.1; }
        else if (aok) { *x = 2.1; *y = 1.1; }
    }
    return *x != 0.0 && *y != 0.0;
}
static bool s2(const int64_t* r, const int64_t* a, double lr, double la, double* x, double* y)
{
    *x = *y = 0.0;
    if (r[15] || a[15]) return false;
    if (std::max((double)r[0] / lr, (double)a[0] / la) > 0.1 && r[5] > 0 && a[5] > 0) { *x = (double)a[6]; *y = (double)r[6]; }
    return *x != 0.0 && *y != 0.0;
}
static double dirv(const int64_t* s) { return s[11] == 0 ? 0.0001 : std::fabs(((double)s[12] * 0.5) / (double)s[11]); }
static bool s3(const int64_t* r, const int64_t* a, double lr, double la, double* x, double* y)
{
    *x = *y = 0.0;
    if (r[15] || a[15]) return false;
    if ((double)r[0] / lr > 0.1 && (double)a[0] / la > 0.1 && (double)(r[2] - r[1]) / lr > 0.7 && (double)(a[2] - a[1]) / la > 0.7 &&
        r[3] > 0 && a[3] > 0) { *x = dirv(r); *y = dirv(a); }
    return *x != 0.0 && *y != 0.0;
}
// numpy.add.reduce over a contiguous float64 array (pairwise_sum): n < 8 one by one; n <= 128 eight strided partial
// sums combined as ((0+1)+(2+3))+((4+5)+(6+7)), then the last n % 8; above that halves recursively
static double np_sum(const double* a, size_t n)
{
    if (n < 8) { double r = 0.0; for (size_t i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int i = 0; i < 8; ++i) r[i] = a[i];
        size_t i = 8;
        for (; i + 8 <= n; i += 8) for (int q = 0; q < 8; ++q) r[q] += a[i + (size_t)q];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

extern "C" int vapor_plan_run_loci(vapor_plan* p, void* d_out, double* loci_out, double* read_scores)
{
    if (!p) return fail(VAPOR_E_ARG, "vapor_plan_run_loci: null plan");
    if (p->gt.empty()) return fail(VAPOR_E_ARG, "vapor_plan_run_loci: call vapor_plan_set_reads first");
    std::vector<int64_t> st(std::max<size_t>(p->pairs.size(), 1) * 16);
    int rc = vapor_plan_run(p, st.data());
    if (rc != VAPOR_OK) return rc;
    const double nan = std::nan("");
    p->read_scores.assign(std::max<size_t>(p->reads.size(), 1), nan);
    p->loci.assign((size_t)std::max<int64_t>(p->n_loci, 1) * 8, nan);
    std::vector<std::vector<double>> per((size_t)p->n_loci);
    for (size_t t = 0; t < p->reads.size(); ++t) {
        const vapor_read& rd = p->reads[t];
        const double lr = rd.len_ref, la = rd.len_alt;
        double a, b, v = nan;
        const int64_t *ra = &st[16 * (size_t)rd.ref_a], *aa = &st[16 * (size_t)rd.alt_a];
        if (rd.kind == 0) {
            double a2, b2;
            const bool v1 = s1(ra, aa, lr, la, &a, &b), v2 = s2(&st[16 * (size_t)rd.ref_b], &st[16 * (size_t)rd.alt_b], lr, la, &a2, &b2);
            const double x1 = 1.0 - b / a, x2 = 1.0 - b2 / a2;
            if (v1 && v2) v = std::min(x1, x2); else if (v1) v = x1; else if (v2) v = x2;
        } else {
            const bool ok = rd.kind == 1 ? s1(ra, aa, lr, la, &a, &b) : rd.kind == 2 ? s2(ra, aa, lr, la, &a, &b) : s3(ra, aa, lr, la, &a, &b);
            if (ok) v = 1.0 - b / a;
        }
        p->read_scores[t] = v;
        if (v == v) per[(size_t)rd.locus].push_back(v);
    }
    for (int64_t l = 0; l < p->n_loci; ++l) {
        double* o = &p->loci[8 * (size_t)l];
        const auto& sc = per[(size_t)l];
        if (sc.empty()) { o[4] = 0.0; continue; }
        std::vector<double> pos;
        int nonpos = 0;
        for (double v : sc) { if (v > 0.0) pos.push_back(v); if (!(v >= 0.005)) ++nonpos; }   // round(v, 2) > 0 <=> v >= 0.005
        const double qs = pos.empty() ? 0.0 : np_sum(pos.data(), pos.size()) / (double)pos.size();
        const double gs = (double)pos.size() / (double)sc.size();
        int gt = 1;
        double gq = nan;
        if ((int)sc.size() < VAPOR_GT_TABLE_N) {
            gt = (int)p->gt[2 * (sc.size() * VAPOR_GT_TABLE_N + (size_t)nonpos)];
            gq = p->gt[2 * (sc.size() * VAPOR_GT_TABLE_N + (size_t)nonpos) + 1];
        }
        if (gt == 0 && gs > 0.15) gt = 1;
        o[0] = qs; o[1] = gs; o[2] = gt; o[3] = gq; o[4] = (double)sc.size(); o[5] = (double)pos.size(); o[6] = nonpos; o[7] = 0.0;
    }
    if (d_out) memcpy(d_out, p->loci.data(), sizeof(double) * 8 * (size_t)p->n_loci);      // "device" memory is host memory here
    if (loci_out) memcpy(loci_out, p->loci.data(), sizeof(double) * 8 * (size_t)p->n_loci);
    if (read_scores) memcpy(read_scores, p->read_scores.data(), sizeof(double) * p->reads.size());
    return VAPOR_OK;
}
extern "C" int vapor_plan_run_loci_async(vapor_plan* p, void* d_out) { return vapor_plan_run_loci(p, d_out, nullptr, nullptr); }
extern "C" int vapor_plan_then(vapor_plan* p, void*) { return p ? VAPOR_OK : fail(VAPOR_E_ARG, "null plan"); }
extern "C" int vapor_plan_after(vapor_plan* p, void*) { return p ? VAPOR_OK : fail(VAPOR_E_ARG, "null plan"); }
extern "C" int vapor_plan_sync(vapor_plan* p, double* loci_out)
{
    if (!p) return fail(VAPOR_E_ARG, "vapor_plan_sync: null plan");
    if (loci_out && !p->loci.empty()) memcpy(loci_out, p->loci.data(), sizeof(double) * 8 * (size_t)p->n_loci);
    return VAPOR_OK;
}

// ---- CIGAR walks (SF:309-337) ---------------------------------------------------------------
extern "C" int vapor_cigar2alignstart(const char* cigar, int64_t align_start, int64_t start, int64_t* out)
{
    if (!cigar || !out) return fail(VAPOR_E_ARG, "vapor_cigar2alignstart: null argument");
    int64_t q = 0, r = align_start, n = 0;
    bool have = false;
    char last = 0;
    for (const char* c = cigar; *c; ++c) {
        if (*c >= '0' && *c <= '9') { n = n * 10 + (*c - '0'); have = true; continue; }
        if (strchr("MIDNSHP=X", *c) && have) {
            if (*c == 'S' || *c == 'I') q += n;
            else if (*c == 'M' || *c == '=') { q += n; r += n; }
            else if (*c == 'D') r += n;
            last = *c;
            if (r > start - 1) break;
        }
        n = 0; have = false;
    }
    if (!last) return fail(VAPOR_E_ARG, "vapor_cigar2alignstart: no CIGAR operation");
    const int64_t over = r - start;
    if (last == 'M' || last == '=') { out[0] = q - over; out[1] = 0; } else { out[0] = q; out[1] = over; }
    return VAPOR_OK;
}
extern "C" int vapor_cigar2alignstart_ops(const uint32_t* ops, int64_t n_ops, int64_t align_start, int64_t start, int64_t* out)
{
    if (!out || n_ops <= 0 || !ops) return fail(VAPOR_E_ARG, "vapor_cigar2alignstart_ops: no CIGAR operation");
    std::string text;
    for (int64_t t = 0; t < n_ops; ++t) text += std::to_string(ops[t] >> 4) + ((ops[t] & 15u) <= 8u ? "MIDNSHP=X"[ops[t] & 15u] : '?');
    return vapor_cigar2alignstart(text.c_str(), align_start, start, out);
}

// ---- read extraction "on the device" (vapor_bam_chop_device, vapor_seqset_create_mixed) ---------------------------------
// The twin has no device: a batch is host memory, the regions go through the host reader (vapor_bam_chop, compiled into this
// library from vapor_amd/csrc/vapor_bam.cpp) one by one, and the kept reads are packed four bits a base the way a BAM record
// holds them - starting at an odd base for every other read, so that callers meet both nibble phases.
struct vapor_bam_batch { std::vector<std::vector<uint8_t>> packed; };

extern "C" int vapor_bam_chop_device(vapor_ctx* ctx, vapor_bam* bam, int32_t n_regions, const int32_t* tid, const int64_t* start,
                                     const int64_t* end, const int64_t* flank, const int32_t* chunk_first, const uint64_t* chunks,
                                     int32_t max_keep, int32_t* kept_first, uint64_t* sq_addr, int64_t* q0, int64_t* miss,
                                     int32_t* status, vapor_bam_batch** out)
{
    if (!ctx || !bam || !out || n_regions < 0 || max_keep < 1 || max_keep > 256 ||
        (n_regions && (!tid || !start || !end || !flank || !chunk_first || !kept_first || !sq_addr || !q0 || !miss || !status)))
        return fail(VAPOR_E_ARG, "vapor_bam_chop_device: bad argument");
    vapor_bam_batch* B = new vapor_bam_batch();
    std::vector<uint8_t> seq((size_t)1 << 20);
    std::vector<char> names((size_t)1 << 16);
    std::vector<int64_t> meta(4 * 1024);
    int32_t w = 0;
    for (int32_t g = 0; g < n_regions; ++g) {
        kept_first[g] = w;
        status[g] = 0;
        const int32_t nc = chunk_first[g + 1] - chunk_first[g];
        int32_t n = 0;
        int64_t need[3] = {0, 0, 0};
        int rc;
        for (;;) {
            rc = vapor_bam_chop(bam, tid[g], start[g], end[g], flank[g], nc, nc ? chunks + 2 * (size_t)chunk_first[g] : nullptr, seq.data(),
                                (int64_t)seq.size(), names.data(), (int64_t)names.size(), meta.data(), (int32_t)(meta.size() / 4), &n, need);
            if (rc != VAPOR_E_OVERFLOW) break;
            seq.resize((size_t)need[0] * 2 + 1024); names.resize((size_t)need[1] * 2 + 256); meta.resize(4 * ((size_t)need[2] * 2 + 16));
        }
        if (rc != VAPOR_OK) { status[g] = 1; continue; }        // (the host route words the error)
        std::vector<int32_t> order((size_t)n);
        for (int32_t i = 0; i < n; ++i) order[(size_t)i] = i;
        if (n > max_keep) {
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return meta[4 * (size_t)a + 2] < meta[4 * (size_t)b + 2]; });
            order.resize((size_t)max_keep);
        }
        for (int32_t i : order) {
            const uint8_t* s = seq.data() + meta[4 * (size_t)i];
            const int64_t len = meta[4 * (size_t)i + 1];
            const int64_t first = w & 1;
            std::vector<uint8_t> pk((size_t)((first + len + 1) / 2) + 1, 0);
            for (int64_t t = 0; t < len; ++t) {
                const char* at = strchr("=ACMGRSVTWYHKDBN", (char)s[t]);
                const uint8_t code = at && s[t] ? (uint8_t)(at - "=ACMGRSVTWYHKDBN") : 15;
                const int64_t b = first + t;
                pk[(size_t)(b >> 1)] |= (b & 1) ? code : (uint8_t)(code << 4);
            }
            B->packed.push_back(std::move(pk));
            sq_addr[w] = (uint64_t)reinterpret_cast<uintptr_t>(B->packed.back().data());
            q0[w] = first;
            miss[w] = meta[4 * (size_t)i + 2];
            ++w;
        }
    }
    kept_first[n_regions] = w;
    *out = B;
    return VAPOR_OK;
}
extern "C" int vapor_bam_batch_destroy(vapor_bam_batch* b) { delete b; return VAPOR_OK; }
extern "C" int vapor_bam_last_stats(vapor_ctx* ctx, double* out, int32_t n)
{
    if (!ctx || !out || n < 0) return fail(VAPOR_E_ARG, "vapor_bam_last_stats: null argument");
    for (int32_t i = 0; i < n && i < 6; ++i) out[i] = 0.;          // (no device, nothing measured)
    return VAPOR_OK;
}

extern "C" int vapor_seqset_create_mixed(vapor_ctx* ctx, int32_t n, const uint8_t* const* seq, const int32_t* len, const uint8_t* flags,
                                         const uint8_t* src_kind, const int64_t* src_first, int32_t n_derived, const int32_t* seg_first,
                                         const vapor_segment* segs, const uint8_t* derived_flags, int32_t* info, vapor_seqset** out)
{
    if (!ctx || !out || n < 0 || (n && (!seq || !len))) return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: null argument");
    std::vector<std::string> text((size_t)n);
    std::vector<const uint8_t*> ptr((size_t)std::max(n, 1));
    for (int32_t i = 0; i < n; ++i) {
        ptr[(size_t)i] = seq[i];
        if (!src_kind || !src_kind[i]) continue;
        if (src_kind[i] != 1 || !src_first || src_first[i] < 0 || len[i] < 0 || (len[i] && !seq[i]))
            return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: bad source description");
        for (int64_t t = 0; t < len[i]; ++t) {
            const int64_t b = src_first[i] + t;
            const uint8_t byte = seq[i][b >> 1];
            text[(size_t)i] += "=ACMGRSVTWYHKDBN"[(b & 1) ? (byte & 15) : (byte >> 4)];
        }
        ptr[(size_t)i] = reinterpret_cast<const uint8_t*>(text[(size_t)i].data());
    }
    return vapor_seqset_create_derived(ctx, n, ptr.data(), len, flags, n_derived, seg_first, segs, derived_flags, info, out);
}
