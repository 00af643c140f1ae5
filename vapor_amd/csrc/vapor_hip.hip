// vapor_hip.hip - host side of libvapor_hip.so: the C ABI declared in include/vapor_hip.h.
//
// Build (see vapor_amd/build.py):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Iinclude vapor_amd/csrc/vapor_hip.hip
//
// Everything the GPU touches is resident in HBM between calls: a vapor_seqset holds the packed
// bit planes, a vapor_plan holds pair/task descriptors, the hit workspace and the statistics.
#include "vapor_kernels.h"
#include "vapor_bamdev.h"
#include "vapor_hip.h"

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <array>
#include <atomic>
#include <chrono>
#include <vector>

using namespace vapor;

// Cost of building an allele's table relative to probing one read base against it, in eighths.  Round 2 measured 0.4 per allele
// base (a table of 20 000 positions ~10 us, a 10 000-base read ~12 us) when a task held sixteen reads; with shared joins a cfg2
// task holds eight and the same measurement (tools/task_balance.py on a -DVAPOR_BLOCK_TIMING build: tasks with one table 71 us,
// with two 82-85 us: a table 11 us = 1.5 reads of 7.5 us) gives 0.75: priced at 3 the tasks that straddle two windows were the
// launch's longest by 15 %.  At 6: cfg2 join 0.0849 -> 0.0797 ms (5 .. 10 the same; profiles/r05_join_rounds.txt).
#ifndef VAPOR_BUILD_COST_X8
#define VAPOR_BUILD_COST_X8 6
#endif


static thread_local std::string g_err;

static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(VAPOR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

// ------------------------------------------------------------------------------------------
// Device and pinned-host blocks are recycled inside a context: a caller that works through a sequence of batches
// creates and destroys a sequence set and a plan per batch, and a dozen hipMalloc / hipHostMalloc / hipFree calls per
// batch cost more than the batch's kernels.  Freed blocks go to a per-context free list (by capacity) and are handed
// out again to requests of similar size; vapor_destroy releases them.  A set or plan that outlives its context
// frees its blocks directly.
struct BlockPool {
    std::multimap<size_t, void*> free_dev, free_host;
    std::map<const void*, size_t> cap_of;               // capacity of every block this pool has handed out or holds
    size_t cached_dev = 0, cached_host = 0;
};
static std::mutex g_live_m;
static std::set<const void*> g_live_ctx;

static size_t pool_round(size_t bytes)
{
    if (bytes <= 256) return 256;
    if (bytes < ((size_t)1 << 16)) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
    return (bytes + 0xFFFF) & ~(size_t)0xFFFF;          // 64 KB steps above that
}

struct vapor_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;          // the stream vapor_init created (vapor_set_stream may replace `stream`)
    // vapor_plan_run_loci_async: every plan keeps to one of two library-owned streams ("lanes"), dealt out in turn, so
    // that the steps of two plans in flight overlap (the join owns every CU's LDS while it runs; the clean and finish
    // kernels of the other plan fill the CUs it has not reached or has already left).  A caller's stream
    // (vapor_set_stream) replaces both.
    hipStream_t lane[2] = {nullptr, nullptr};
    hipStream_t fin[2] = {nullptr, nullptr};   // per lane: the stream its plans' finish kernels go to (highest priority)
    unsigned lane_rr = 0;
    bool user_stream = false;
    // staging for vapor_seqset_create*, kept between calls (pinned allocations are slow): ASCII chunks + chunk map
    uint8_t* h_stage = nullptr;
    uint8_t* d_stage = nullptr;
    size_t stage_cap = 0;
    // the genotype table of vapor_plan_set_reads, kept on the device while callers keep passing the same one
    std::vector<double> h_gt;
    double* d_gt = nullptr;
    int reads_per_task = MAX_READS_PER_TASK;   // upper bound on pairs per join task
    int n_cus = 256;
    int join_tasks = 256;                      // join tasks aimed for per launch (cost-balanced ranges): one per CU
    int64_t max_pair_cap = (int64_t)1 << 28;
    bool shared_join = true;                   // reads scored against a window and alleles derived from it: one join for all
    int remap_in_clean = 1;                    // ... and the clean workgroup of a target cuts its records out of the shared plot (0: remap_kernel)
    int clean_order = 1;                       // 1: the clean workgroups are dealt out longest pair first (plan_clean_order); 0: in pair order
    int clean_fit = 1;                         // 1: after a blocking run the clean kernel's LDS copy is sized for the records the pairs really hold
    int stage_threads = 3;                     // host threads that copy a large upload into the pinned staging buffer (measured:
                                               // two to four are as fast as it gets, more are slower - tools/upload_sweep.py)
    bool attrs_set = false;
    // vapor_bam_chop_device: the CRC combination constants on the device, and the arenas of the batches that are alive
    // (vapor_seqset_create_mixed takes packed bases by device address: only addresses inside one of these are followed)
    uint32_t* d_crc_pow = nullptr;
    std::map<const uint8_t*, size_t> arenas;
    // the stream the extraction's copies and kernels go to: the context's own, or (parameter bam_cu_share = s of 8) one masked to
    // s eighths of the CUs, so that the kernels other contexts' threads launch meanwhile (packing, joins, cleaning) find CUs whose
    // LDS is not held by twenty inflating wavefronts (cli.py sets it when it scores several chunks at once: 25.5-26.8 k -> 27.5-30.5 k
    // loci/s from files at 5 of 8, profiles/r05_bamdev.txt)
    hipStream_t bam_stream = nullptr;
    int bam_stream_share = 0;                  // the share bam_stream was made for
    int bam_cu_share = 0;
    hipEvent_t bam_ev[2] = {nullptr, nullptr}; // around the inflate launch of the last vapor_bam_chop_device (vapor_bam_last_stats)
    double bam_stats[6] = {0, 0, 0, 0, 0, 0};  // regions, blocks, compressed bytes, inflated bytes, inflate ms, whole call ms
    BlockPool pool;
};

static bool ctx_alive(const vapor_ctx* c)
{
    std::lock_guard<std::mutex> g(g_live_m);
    return g_live_ctx.count(c) != 0;
}

// one host thread per context (include/vapor_hip.h), so the lists themselves need no lock
static hipError_t pool_alloc(vapor_ctx* c, void** out, size_t bytes, bool host)
{
    const size_t cap = pool_round(bytes);
    auto& fl = host ? c->pool.free_host : c->pool.free_dev;
    auto it = fl.lower_bound(cap);
    if (it != fl.end() && it->first <= 2 * cap + ((size_t)1 << 20)) {
        *out = it->second;
        (host ? c->pool.cached_host : c->pool.cached_dev) -= it->first;
        fl.erase(it);
        return hipSuccess;
    }
    hipError_t e = host ? hipHostMalloc(out, cap) : hipMalloc(out, cap);
    if (e == hipSuccess) c->pool.cap_of[*out] = cap;
    return e;
}
static hipError_t dmalloc(vapor_ctx* c, void** out, size_t bytes) { return pool_alloc(c, out, bytes, false); }
static hipError_t hmalloc(vapor_ctx* c, void** out, size_t bytes) { return pool_alloc(c, out, bytes, true); }

static void pool_free(vapor_ctx* c, void* p, bool host)
{
    if (!p) return;
    if (c && ctx_alive(c)) {
        auto it = c->pool.cap_of.find(p);
        if (it != c->pool.cap_of.end()) {
            size_t& cached = host ? c->pool.cached_host : c->pool.cached_dev;
            if (cached + it->second <= (host ? ((size_t)2 << 30) : ((size_t)16 << 30))) {
                (host ? c->pool.free_host : c->pool.free_dev).emplace(it->second, p);
                cached += it->second;
                return;
            }
            c->pool.cap_of.erase(it);
        }
    }
    if (host) (void)hipHostFree(p); else (void)hipFree(p);
}
static void dfree(vapor_ctx* c, void* p) { pool_free(c, p, false); }
static void hfree(vapor_ctx* c, void* p) { pool_free(c, p, true); }

// A derived sequence as the caller described it (destination offsets added), and the groups the plan shares joins in.
struct HSeg { int32_t parent, off, len, dst; bool rc; };
struct SharePiece {                // a stretch of a member that holds k-mers its parent window does not (for the largest window size)
    int32_t slot;                  // member slot (1 ..)
    int32_t a_from, a_to;          // symbols [a_from, a_to) of the member
    int32_t tile_off;              // where they lie in the shared sequence
};
struct ShareGroup {                // a window uploaded as bytes (or its upper-cased twin) and the sequences derived from it
    int32_t parent = -1;           // the literal sequence
    bool upper = false;
    int32_t identity = -1;         // user index of the sequence that IS the window (slot 0), -1: not in the set
    int32_t t_seq = -1;            // hidden shared sequence: the window followed by the pieces
    std::vector<int32_t> members;  // user indices of the derived sequences (slot 1 + position), at most 3
    std::vector<SharePiece> pieces;
};
constexpr int SHARE_KMAX = 40;     // the largest window size (pieces carry SHARE_KMAX - 1 symbols of context)

struct vapor_seqset {
    vapor_ctx* ctx = nullptr;
    int device = 0;                // kept here: the set may be destroyed after its context
    int32_t n = 0;                 // sequences the caller sees: n_lit given as bytes, then the derived ones
    int32_t n_lit = 0;
    std::vector<SeqDesc> h;        // host copy (with device-computed counts); hidden shared sequences behind the caller's
    SeqDesc* d_seqs = nullptr;
    uint32_t *d_p2 = nullptr, *d_e1 = nullptr, *d_x4 = nullptr;
    size_t plane_chunks = 0;
    std::vector<std::vector<HSeg>> derived;     // per derived sequence (index - n_lit)
    std::vector<ShareGroup> groups;
    std::vector<int32_t> group_of, slot_of;     // per caller-visible sequence: its group (-1: none) and slot in it
};

struct Launch {
    int bps, k, task_begin, n_tasks;
    int exc;       // 2-bit planes: 1 = alleles with symbols outside upper-case ACGT, 2 = reads with such symbols (join_kernel<.., EXC>)
};

struct vapor_plan {
    vapor_ctx* ctx = nullptr;
    int device = 0;                // kept here: the plan may be destroyed after its context
    vapor_seqset* set = nullptr;
    int64_t n_pairs = 0;
    std::vector<DPair> hp;
    std::vector<int32_t> status;
    std::vector<DTask> tasks;
    std::vector<int32_t> task_pairs;
    std::vector<Launch> launches;
    std::vector<int64_t> last_stats;
    int range_words_cap = 1;
    int hcap_want = 4096;
    bool hcap_measured = false;                // hcap_want is the largest record count a blocking run saw (else: an estimate)
    int64_t total_cap = 0;
    DPair* d_pairs = nullptr;
    DTask* d_tasks = nullptr;
    int32_t* d_task_pairs = nullptr;
    unsigned long long* d_hits = nullptr;   // run records (VREC_*), hp[].cap slots per pair
    uint8_t* d_hflags = nullptr;            // one flag byte per record
    unsigned long long* d_nhits = nullptr;
    long long* d_stats = nullptr;
    long long* h_stats = nullptr;  // pinned
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_f[2] = {nullptr, nullptr};
    // vapor_plan_run_loci_async: one set of four events per step in flight, summed by vapor_plan_sync
    std::vector<std::array<hipEvent_t, 4>> ring;
    int ring_n = 0;
    bool overflow_final = false;               // some pair overflows even at max_pair_cap: do not retry again
    double acc_ms[4] = {0, 0, 0, 0};           // join, clean, finish, total of the steps already folded in
    int64_t acc_n = 0;
    double* h_loci = nullptr;                  // pinned copy of the per-locus records of the last async step
    hipEvent_t ev_t0 = nullptr;
    hipStream_t lane = nullptr;                // the stream this plan's asynchronous steps are enqueued on
    hipStream_t fin = nullptr;                 // ... and the one their finish kernels go to (NULL: the lane itself)
    hipEvent_t ev_clean = nullptr;             // the clean kernels of the last enqueued step are done (lane -> fin)
    hipEvent_t ev_fin = nullptr;               // its finish kernel is done (fin -> lane: the next step's clean kernels wait for it)
    bool have_fin = false;
    hipEvent_t ev_last = nullptr;              // end of the most recently enqueued asynchronous step
    hipEvent_t ev_after = nullptr;             // vapor_plan_after: the next step waits for it
    bool have_last = false, have_after = false;
    double t_join = 0, t_clean = 0, t_total = 0;
    int n_retried = 0;
    bool ran = false;
    bool flags_valid = false;                  // the last run wrote the per-record flag bytes (vapor_plan_fetch_hits hands them out)
    // optional per-read / per-locus finishing on the device
    int64_t n_reads = 0, n_loci = 0;
    DRead* d_reads = nullptr;
    int32_t* d_locus_first = nullptr;
    double* d_gt = nullptr;
    bool own_gt = false;              // false: the context's cached table
    double* d_read_scores = nullptr;
    double* d_loci = nullptr;
    unsigned int* d_overflow = nullptr;   // [0] pairs whose slot overflowed, [1] length of d_big_list
    int32_t* d_big_list = nullptr;        // pairs with more dots than clean_kernel stages in LDS
    unsigned int* h_overflow = nullptr;   // pinned, two counters: pairs that overflowed their slot, pairs left to clean_big_kernel
    bool big_known = false;               // a blocking run has reported how many pairs clean_kernel leaves to clean_big_kernel
    unsigned int n_big = 0;
    double t_finish = 0;
    // shared joins (remap_kernel): pairs n_pairs .. n_pairs + n_dpairs - 1 of hp are the (read, shared sequence) pairs the
    // join runs instead of the pairs they serve
    int64_t n_dpairs = 0, n_served = 0;
    std::vector<DShare> shares;
    std::vector<DMap> maps;                    // (host only: the interval maps the tables are cut from)
    std::vector<int32_t> tables;               // per (group, k): boundaries and op words (remap_kernel)
    DShare* d_shares = nullptr;
    int32_t* d_maps = nullptr;
    std::vector<DServe> serve;                 // per pair: where its records come from when a shared join serves it
    DServe* d_serve = nullptr;
    int32_t* d_clean_order = nullptr;          // the pairs in the order their clean workgroups are dealt out (longest first)
};

// ------------------------------------------------------------------------------------------
// A developer build (-DVAPOR_DEV_BUILD: timing stamps, tools/ab.py variants, non-default tuning constants) says so in
// both: the product loader accepts VAPOR_ABI_VERSION only, and vapor_build_flags() lists what the build carries.
#ifdef VAPOR_DEV_BUILD
extern "C" int vapor_abi_version(void) { return VAPOR_ABI_VERSION + VAPOR_ABI_DEV_OFFSET; }
#define VP_STR2(x) #x
#define VP_STR(x) VP_STR2(x)
extern "C" const char* vapor_build_flags(void)
{
    return "dev"
#ifdef VAPOR_PHASE_TIMING
           ",phase_timing"
#endif
#ifdef VAPOR_BLOCK_TIMING
           ",block_timing"
#endif
#ifdef VAPOR_AB
           ",ab=" VP_STR(VAPOR_AB)
#endif
           ",jq_fast_slack=" VP_STR(VAPOR_JQ_FAST_SLACK) ",clean_threads=" VP_STR(VAPOR_CLEAN_THREADS) ",build_cost_x8=" VP_STR(VAPOR_BUILD_COST_X8);
}
#else
extern "C" int vapor_abi_version(void) { return VAPOR_ABI_VERSION; }
extern "C" const char* vapor_build_flags(void) { return ""; }
#endif
extern "C" const char* vapor_last_error(void) { return g_err.c_str(); }
// what this binary was built from (vapor_amd/build.py: sha256 of the kernel sources ':' sha256 of every source file); the
// marker in front lets the build read it out of the file without loading it
#ifndef VAPOR_SOURCE_ID
#define VAPOR_SOURCE_ID "unknown"
#endif
static const char g_source_id[] = "VAPOR_SOURCE_ID=" VAPOR_SOURCE_ID;
extern "C" const char* vapor_source_id(void) { return g_source_id + 16; }

#define JOIN_DYN_LDS(BPS) 0            // the join's LDS is a static array of the kernel

extern "C" int vapor_init(int device_ordinal, vapor_ctx** out)
{
    if (!out) return fail(VAPOR_E_ARG, "vapor_init: null out pointer");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(VAPOR_E_ARG, "vapor_init: no such device ordinal");
    HIPCHK(hipSetDevice(device_ordinal));
    vapor_ctx* c = new (std::nothrow) vapor_ctx();
    if (!c) return fail(VAPOR_E_NOMEM, "vapor_init: out of memory");
    c->device = device_ordinal;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) {
            c->n_cus = prop.multiProcessorCount;
            // a join workgroup fills a CU's LDS (or, in the two-per-CU experiment geometry, half of it)
            c->join_tasks = prop.multiProcessorCount * (JoinCfg::THREADS <= 768 ? 2 : 1);
        }
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(VAPOR_E_HIP, hipGetErrorString(e)); }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&clean_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&clean_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&clean_kernel<CLEAN_PER_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&clean_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    if (e != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return fail(VAPOR_E_HIP, std::string("hipFuncSetAttribute(clean): ") + hipGetErrorString(e)); }
    c->own_stream = c->stream;
    { std::lock_guard<std::mutex> g(g_live_m); g_live_ctx.insert(c); }
    *out = c;
    return VAPOR_OK;
}

extern "C" int vapor_destroy(vapor_ctx* c)
{
    if (!c) return VAPOR_OK;
    (void)hipSetDevice(c->device);
    { std::lock_guard<std::mutex> g(g_live_m); g_live_ctx.erase(c); }
    (void)hipDeviceSynchronize();
    for (auto& b : c->pool.free_dev) (void)hipFree(b.second);
    for (auto& b : c->pool.free_host) (void)hipHostFree(b.second);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    for (hipStream_t l : c->lane)
        if (l) (void)hipStreamDestroy(l);
    for (hipStream_t l : c->fin)
        if (l) (void)hipStreamDestroy(l);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_gt) (void)hipFree(c->d_gt);
    if (c->d_crc_pow) (void)hipFree(c->d_crc_pow);
    if (c->bam_stream) (void)hipStreamDestroy(c->bam_stream);
    for (hipEvent_t e : c->bam_ev)
        if (e) (void)hipEventDestroy(e);
    delete c;
    return VAPOR_OK;
}

extern "C" int vapor_set_param(vapor_ctx* c, const char* name, int64_t v)
{
    if (!c || !name) return fail(VAPOR_E_ARG, "vapor_set_param: null argument");
    if (!strcmp(name, "reads_per_task")) {
        if (v < 1 || v > MAX_READS_PER_TASK) return fail(VAPOR_E_ARG, "reads_per_task out of range");
        c->reads_per_task = (int)v;
        return VAPOR_OK;
    }
    if (!strcmp(name, "bam_cu_share")) {
        if (v < 0 || v > 8) return fail(VAPOR_E_ARG, "bam_cu_share out of range (0 .. 8 eighths of the CUs; 0 and 8: all)");
        c->bam_cu_share = (int)(v == 8 ? 0 : v);
        return VAPOR_OK;
    }
    if (!strcmp(name, "join_tasks")) {
        if (v < 1) return fail(VAPOR_E_ARG, "join_tasks out of range");
        c->join_tasks = (int)v;
        return VAPOR_OK;
    }
    if (!strcmp(name, "max_pair_cap")) {
        if (v < 1) return fail(VAPOR_E_ARG, "max_pair_cap out of range");
        c->max_pair_cap = v;
        return VAPOR_OK;
    }
    if (!strcmp(name, "shared_join")) {
        c->shared_join = v != 0;
        return VAPOR_OK;
    }
    if (!strcmp(name, "remap_in_clean")) {
        if (v < 0 || v > 2) return fail(VAPOR_E_ARG, "vapor_set_param: remap_in_clean is 0, 1 or 2");
        c->remap_in_clean = (int)v;
        return VAPOR_OK;
    }
    if (!strcmp(name, "clean_order")) {
        c->clean_order = v != 0 ? 1 : 0;
        return VAPOR_OK;
    }
    if (!strcmp(name, "clean_fit")) {
        c->clean_fit = v != 0;
        return VAPOR_OK;
    }
    if (!strcmp(name, "stage_threads")) {
        if (v < 1 || v > 64) return fail(VAPOR_E_ARG, "stage_threads out of range");
        c->stage_threads = (int)v;
        return VAPOR_OK;
    }
    return fail(VAPOR_E_ARG, std::string("unknown parameter ") + name);
}

// ------------------------------------------------------------------------------------------
extern "C" int vapor_seqset_destroy(vapor_seqset* s)
{
    if (!s) return VAPOR_OK;
    (void)hipSetDevice(s->device);
    dfree(s->ctx, s->d_seqs);
    dfree(s->ctx, s->d_p2);
    dfree(s->ctx, s->d_e1);
    dfree(s->ctx, s->d_x4);
    delete s;
    return VAPOR_OK;
}

// Shared by the two entry points: sequence i starts at src(i).
// The groups a plan shares joins in: every derived sequence goes to the group of the literal that gives it most of its symbols
// (same upper-casing), a derived sequence that is that literal from end to end (an upper-cased twin) is the group's identity,
// and the hidden sequence of a group is the window followed by the stretches of its members that hold k-mers the window does
// not: around every junction of two segments and over every segment that is not a long enough forward or reversed slice of
// the window itself, with SHARE_KMAX - 1 symbols of context on either side (cut for the largest window size, so that one
// hidden sequence serves every k; the plan cuts the interval maps for its k).  Returns the hidden sequences' segment lists.
static void build_share_groups(vapor_seqset* s, const std::vector<uint8_t>& dflags, std::vector<std::vector<HSeg>>* hidden)
{
    const int32_t n_lit = s->n_lit, n_der = (int32_t)s->derived.size();
    s->group_of.assign((size_t)s->n, -1);
    s->slot_of.assign((size_t)s->n, -1);
    std::map<std::pair<int32_t, bool>, int32_t> gid;
    for (int32_t d = 0; d < n_der; ++d) {
        const auto& sg = s->derived[(size_t)d];
        const bool up = dflags[(size_t)d] & VAPOR_SEQ_UPPER;
        // the literal with the largest share of this sequence's symbols (forward or reversed)
        std::map<int32_t, int64_t> share;
        for (const HSeg& g : sg) share[g.parent] += g.len;
        int32_t par = -1; int64_t best = 0;
        for (auto& kv : share) if (kv.second > best) { best = kv.second; par = kv.first; }
        if (par < 0) continue;
        if ((s->h[(size_t)par].flags & VAPOR_SEQ_UPPER) && !up) continue;     // (the literal was upper-cased at upload: not this one's text)
        auto it = gid.find({par, up});
        if (it == gid.end()) {
            it = gid.emplace(std::make_pair(par, up), (int32_t)s->groups.size()).first;
            ShareGroup g; g.parent = par; g.upper = up;
            if (!up) { g.identity = par; }
            s->groups.push_back(g);
        }
        ShareGroup& g = s->groups[(size_t)it->second];
        const bool whole = sg.size() == 1 && sg[0].parent == par && sg[0].off == 0 && sg[0].len == s->h[(size_t)par].len && !sg[0].rc;
        if (whole && g.identity < 0) { g.identity = n_lit + d; continue; }
        if (whole && g.identity >= 0) continue;                     // (another copy of the window - plain, or a second upper-cased twin: it
                                                                    // would only take one of the group's three member slots; its pairs are joined on their own)
        if (g.members.size() < 3) g.members.push_back(n_lit + d);
    }
    for (size_t q = 0; q < s->groups.size(); ++q) {
        ShareGroup& g = s->groups[q];
        if (g.members.empty()) continue;
        const int32_t n_r = s->h[(size_t)g.parent].len;
        std::vector<HSeg> t;                                         // the hidden sequence's segments
        t.push_back(HSeg{g.parent, 0, n_r, 0, false});
        int64_t t_len = n_r;
        for (size_t m = 0; m < g.members.size(); ++m) {
            const auto& sg = s->derived[(size_t)(g.members[m] - n_lit)];
            const int32_t n_a = s->h[(size_t)g.members[m]].len;
            // k-mer starts of the member that are k-mers of the window at the largest window size
            std::vector<std::pair<int32_t, int32_t>> mapped;
            for (const HSeg& x : sg)
                if (x.parent == g.parent && x.len >= SHARE_KMAX) mapped.push_back({x.dst, x.dst + x.len - SHARE_KMAX});
            int32_t u = 0;
            auto add_piece = [&](int32_t from, int32_t to_start) {       // novel k-mer starts [from, to_start]
                const int32_t a0 = from, a1 = std::min(n_a, to_start + SHARE_KMAX);
                if (a1 <= a0) return;
                g.pieces.push_back(SharePiece{(int32_t)m + 1, a0, a1, (int32_t)t_len});
                // its symbols as slices of the member's own segments
                for (const HSeg& x : sg) {
                    const int32_t lo = std::max(a0, x.dst), hi = std::min(a1, x.dst + x.len);
                    if (hi <= lo) continue;
                    HSeg y;
                    y.parent = x.parent; y.len = hi - lo; y.rc = x.rc; y.dst = (int32_t)t_len + (lo - a0);
                    y.off = x.rc ? x.off + (x.dst + x.len - hi) : x.off + (lo - x.dst);
                    t.push_back(y);
                }
                t_len += a1 - a0;
            };
            for (auto& r : mapped) {                                     // (segments come in order of dst)
                if (r.first > u) add_piece(u, r.first - 1);
                u = std::max(u, r.second + 1);
            }
            if (u <= n_a - 1) add_piece(u, n_a - 1);
        }
        if (t_len > VAPOR_MAX_SEQ_LEN || t_len > (int64_t)2 * n_r + 4096) { g.members.clear(); g.pieces.clear(); continue; }   // not worth sharing
        g.t_seq = s->n + (int32_t)hidden->size();
        hidden->push_back(std::move(t));
        if (g.identity >= 0) { s->group_of[(size_t)g.identity] = (int32_t)q; s->slot_of[(size_t)g.identity] = 0; }
        for (size_t m = 0; m < g.members.size(); ++m) { s->group_of[(size_t)g.members[m]] = (int32_t)q; s->slot_of[(size_t)g.members[m]] = (int32_t)m + 1; }
    }
}

template <typename SRC>
static int seqset_create_impl(vapor_ctx* ctx, int32_t n_seqs, SRC src, const int32_t* len, const uint8_t* flags,
                              int32_t* seq_info, vapor_seqset** out, int32_t n_derived = 0, const int32_t* seg_first = nullptr,
                              const vapor_segment* segs = nullptr, const uint8_t* derived_flags = nullptr,
                              const uint8_t* src_kind = nullptr, const int64_t* src_first = nullptr)
{
    HIPCHK(hipSetDevice(ctx->device));
    const bool dbg_t = getenv("VAPOR_DEBUG_UPLOAD") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq[8] = {now(), 0, 0, 0, 0, 0, 0, 0};
    vapor_seqset* s = new (std::nothrow) vapor_seqset();
    if (!s) return fail(VAPOR_E_NOMEM, "out of memory");
    s->ctx = ctx;
    s->device = ctx->device;
    s->n = n_seqs + n_derived;
    s->n_lit = n_seqs;
    s->h.resize((size_t)std::max(n_seqs + n_derived, 1));
    size_t asc = 0, pl = 0;
    for (int32_t i = 0; i < n_seqs; ++i) {
        if (len[i] < 0) { delete s; return fail(VAPOR_E_ARG, "negative sequence length"); }
        SeqDesc& d = s->h[i];
        memset(&d, 0, sizeof d);
        size_t ch = ((size_t)len[i] + 31) / 32;
        d.asc0 = (uint32_t)asc;
        d.chunk0 = (uint32_t)pl;
        d.len = len[i];
        d.flags = flags ? flags[i] : 0;
        asc += ch;
        pl += ch + VP_PAD_CHUNKS;
        if (pl > 0xFFFFFFF0ull) { delete s; return fail(VAPOR_E_ARG, "sequence set too large"); }
    }
    // derived sequences (the caller's, then the hidden shared ones): planes behind the literals', assembled by derive_kernel
    std::vector<std::vector<HSeg>> hidden;
    std::vector<uint8_t> dfl((size_t)n_derived, 0);
    size_t der_chunks = 0;
    if (n_derived > 0) {
        s->derived.resize((size_t)n_derived);
        for (int32_t d = 0; d < n_derived; ++d) {
            const int32_t g0 = seg_first[d], g1 = seg_first[d + 1];
            if (g1 < g0 || g1 - g0 > VAPOR_MAX_SEGMENTS) { delete s; return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: bad segment count"); }
            int64_t tot = 0;
            for (int32_t g = g0; g < g1; ++g) {
                const vapor_segment& x = segs[g];
                if (x.parent < 0 || x.parent >= n_seqs || x.off < 0 || x.len < 0 || (int64_t)x.off + x.len > len[x.parent]) {
                    delete s;
                    return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: segment outside its parent");
                }
                if (x.len == 0) continue;
                // (a derived sequence is slices of its parents' BYTES: one that is not upper-cased itself cannot be cut from a
                // parent that was upper-cased at upload - its planes hold the upper-cased text)
                if (flags && (flags[x.parent] & VAPOR_SEQ_UPPER) && !(derived_flags && (derived_flags[d] & VAPOR_SEQ_UPPER))) {
                    delete s;
                    return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: a derived sequence without VAPOR_SEQ_UPPER over a parent uploaded with it");
                }
                s->derived[(size_t)d].push_back(HSeg{x.parent, x.off, x.len, (int32_t)tot, (x.flags & VAPOR_SEG_REVCOMP) != 0});
                tot += x.len;
                if (tot > 0x7FFFFFF0LL) { delete s; return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: sequence too long"); }
            }
            dfl[(size_t)d] = derived_flags ? derived_flags[d] : 0;
            SeqDesc& dd = s->h[(size_t)(n_seqs + d)];
            memset(&dd, 0, sizeof dd);
            dd.len = (int32_t)tot;
            dd.flags = dfl[(size_t)d];
        }
        if (ctx->shared_join) build_share_groups(s, dfl, &hidden);
        s->h.resize((size_t)(n_seqs + n_derived) + hidden.size());
        for (size_t t = 0; t < hidden.size(); ++t) {
            SeqDesc& dd = s->h[(size_t)(n_seqs + n_derived) + t];
            memset(&dd, 0, sizeof dd);
            int64_t tot = 0;
            for (const HSeg& x : hidden[t]) tot += x.len;
            dd.len = (int32_t)tot;
        }
        for (const ShareGroup& g : s->groups)
            if (g.t_seq >= 0 && g.upper) s->h[(size_t)g.t_seq].flags = VAPOR_SEQ_UPPER;
        for (size_t i = (size_t)n_seqs; i < s->h.size(); ++i) {
            SeqDesc& dd = s->h[i];
            const size_t ch = ((size_t)dd.len + 31) / 32;
            dd.asc0 = (uint32_t)der_chunks;             // (first chunk among the derived sequences' chunks)
            dd.chunk0 = (uint32_t)pl;
            der_chunks += ch;
            pl += ch + VP_PAD_CHUNKS;
            if (pl > 0xFFFFFFF0ull) { delete s; return fail(VAPOR_E_ARG, "sequence set too large"); }
        }
    }
    pl += VP_PAD_CHUNKS + 1;
    s->plane_chunks = pl;
    const size_t n_asc = asc;
    int rc = VAPOR_OK;
    tq[1] = now();
#define SS_CHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) { rc = fail(VAPOR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); goto done; } \
    } while (0)
    {
        // staging (ASCII at 32-byte chunks, then the chunk -> sequence map, then what derive_kernel reads: segment lists, their
        // offsets, the chunk -> sequence map of the derived sequences), kept in the context and grown on demand
        const size_t n_dseq = s->h.size() - (size_t)n_seqs;
        size_t n_seg = 0;
        for (auto& v : s->derived) n_seg += v.size();
        for (auto& v : hidden) n_seg += v.size();
        // (a mixed set - vapor_seqset_create_mixed - has sequences whose bases are on the device already, 4 bits each: their
        // addresses and first bases travel instead of their bytes, bam_expand_kernel writes their part of the ASCII layout)
        const bool mixed = src_kind != nullptr;
        const size_t mix_off = (n_asc * 36 + 7) & ~(size_t)7;
        const size_t mix_bytes = mixed ? (size_t)n_seqs * 12 : 0;
        const size_t der_off = (mix_off + mix_bytes + 63) & ~(size_t)63;
        const size_t der_bytes = der_chunks ? sizeof(DSeg) * std::max<size_t>(n_seg, 1) + sizeof(int32_t) * (n_dseq + 1) + sizeof(uint32_t) * der_chunks : 0;
        const size_t need = std::max<size_t>(der_off + der_bytes, 64);
        if (need > ctx->stage_cap) {
            SS_CHK(hipStreamSynchronize(ctx->stream));
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            if (ctx->d_stage) (void)hipFree(ctx->d_stage);
            ctx->h_stage = nullptr; ctx->d_stage = nullptr; ctx->stage_cap = 0;
            const size_t cap = need + need / 4;
            SS_CHK(hipHostMalloc((void**)&ctx->h_stage, cap));
            SS_CHK(hipMalloc((void**)&ctx->d_stage, cap));
            ctx->stage_cap = cap;
        }
        uint8_t* h_asc = ctx->h_stage;
        uint32_t* h_map = reinterpret_cast<uint32_t*>(ctx->h_stage + n_asc * 32);
        uint8_t* d_asc = ctx->d_stage;
        uint32_t* d_map = reinterpret_cast<uint32_t*>(ctx->d_stage + n_asc * 32);
        // the copy into the pinned staging buffer is the largest part of an upload (a core moves ~8 GB/s, the link
        // 50): split the sequences over a few host threads when there is enough to copy
        auto stage_range = [&](int32_t i0, int32_t i1) {
            for (int32_t i = i0; i < i1; ++i) {
                const SeqDesc& d = s->h[i];
                size_t ch = ((size_t)d.len + 31) / 32;
                for (size_t c = 0; c < ch; ++c) h_map[d.asc0 + c] = (uint32_t)i;
                if (mixed && src_kind[i]) continue;
                uint8_t* dst = h_asc + (size_t)d.asc0 * 32;
                if (d.len) memcpy(dst, src(i), (size_t)d.len);
                memset(dst + d.len, 0, ch * 32 - (size_t)d.len);
            }
        };
        // the chunks the host has bytes for end here (the reads of a chunk of loci lie behind its windows: the usual mixed set
        // sends the windows alone over the link)
        size_t host_end = n_asc;
        if (mixed) {
            host_end = 0;
            for (int32_t i = 0; i < n_seqs; ++i)
                if (!src_kind[i] && s->h[i].len > 0) host_end = std::max(host_end, (size_t)s->h[i].asc0 + ((size_t)s->h[i].len + 31) / 32);
        }
        SS_CHK(dmalloc(ctx, (void**)&s->d_seqs, sizeof(SeqDesc) * s->h.size()));
        SS_CHK(dmalloc(ctx, (void**)&s->d_p2, pl * 2 * sizeof(uint32_t)));
        SS_CHK(dmalloc(ctx, (void**)&s->d_e1, pl * sizeof(uint32_t)));
        SS_CHK(dmalloc(ctx, (void**)&s->d_x4, pl * 4 * sizeof(uint32_t)));
        SS_CHK(hipMemsetAsync(s->d_p2, 0, pl * 2 * sizeof(uint32_t), ctx->stream));
        SS_CHK(hipMemsetAsync(s->d_e1, 0, pl * sizeof(uint32_t), ctx->stream));
        SS_CHK(hipMemsetAsync(s->d_x4, 0, pl * 4 * sizeof(uint32_t), ctx->stream));
        SS_CHK(hipMemcpyAsync(s->d_seqs, s->h.data(), sizeof(SeqDesc) * s->h.size(), hipMemcpyHostToDevice, ctx->stream));
        tq[2] = now();
        const int n_thr = (!mixed && n_asc * 32 >= ((size_t)4 << 20) && n_seqs >= 8) ? ctx->stage_threads : 1;
        if (mixed) {
            stage_range(0, n_seqs);
            unsigned long long* h_src = reinterpret_cast<unsigned long long*>(ctx->h_stage + mix_off);
            int32_t* h_first = reinterpret_cast<int32_t*>(ctx->h_stage + mix_off + (size_t)n_seqs * 8);
            for (int32_t i = 0; i < n_seqs; ++i) {
                h_src[i] = src_kind[i] ? (unsigned long long)reinterpret_cast<uintptr_t>(src(i)) : 0ull;
                h_first[i] = src_kind[i] ? (int32_t)src_first[i] : 0;
            }
            if (host_end) SS_CHK(hipMemcpyAsync(d_asc, h_asc, host_end * 32, hipMemcpyHostToDevice, ctx->stream));
            if (n_asc) SS_CHK(hipMemcpyAsync(d_map, h_map, n_asc * 4, hipMemcpyHostToDevice, ctx->stream));
            SS_CHK(hipMemcpyAsync(ctx->d_stage + mix_off, ctx->h_stage + mix_off, mix_bytes, hipMemcpyHostToDevice, ctx->stream));
            if (n_asc) {
                hipLaunchKernelGGL(vapor_bamdev::bam_expand_kernel, dim3((unsigned)((n_asc + 255) / 256)), dim3(256), 0, ctx->stream, d_asc, d_map,
                                   (uint32_t)n_asc, reinterpret_cast<const uint32_t*>(s->d_seqs),
                                   reinterpret_cast<const unsigned long long*>(ctx->d_stage + mix_off),
                                   reinterpret_cast<const int32_t*>(ctx->d_stage + mix_off + (size_t)n_seqs * 8));
                SS_CHK(hipGetLastError());
            }
        } else if (n_thr == 1) {
            stage_range(0, n_seqs);
            if (n_asc) SS_CHK(hipMemcpyAsync(d_asc, h_asc, n_asc * 36, hipMemcpyHostToDevice, ctx->stream));
        } else {
            // slices of equal shares of the bytes, staged by the threads in order; this thread sends a slice to the device
            // as soon as it is staged, so that the link works while the cores still copy
            constexpr int SLICES = 12;
            std::vector<int32_t> cut(1, 0);
            for (int t = 1; t < SLICES; ++t) {
                const uint32_t want = (uint32_t)(n_asc * (size_t)t / (size_t)SLICES);
                int32_t i = cut.back();
                while (i < n_seqs && s->h[i].asc0 < want) ++i;
                cut.push_back(i);
            }
            cut.push_back(n_seqs);
            std::atomic<int> staged[SLICES];
            for (auto& f : staged) f.store(0, std::memory_order_relaxed);
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; ++t)
                th.emplace_back([&, t] {
                    for (int k = t; k < SLICES; k += n_thr) {
                        stage_range(cut[(size_t)k], cut[(size_t)k + 1]);
                        staged[k].store(1, std::memory_order_release);
                    }
                });
            hipError_t err = hipSuccess;
            for (int k = 0; k < SLICES; ++k) {
                while (!staged[k].load(std::memory_order_acquire)) std::this_thread::yield();
                const size_t c0 = cut[(size_t)k] < n_seqs ? s->h[(size_t)cut[(size_t)k]].asc0 : n_asc;
                const size_t c1 = cut[(size_t)k + 1] < n_seqs ? s->h[(size_t)cut[(size_t)k + 1]].asc0 : n_asc;
                if (c1 > c0 && err == hipSuccess)
                    err = hipMemcpyAsync(d_asc + c0 * 32, h_asc + c0 * 32, (c1 - c0) * 32, hipMemcpyHostToDevice, ctx->stream);
            }
            for (auto& x : th) x.join();
            SS_CHK(err);
            SS_CHK(hipMemcpyAsync(d_map, h_map, n_asc * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        tq[3] = now();
        if (n_asc) {
            unsigned grid = (unsigned)((n_asc + 255) / 256);
            hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_asc, s->d_seqs, n_seqs, d_map,
                               (uint32_t)n_asc, s->d_p2, s->d_e1, s->d_x4);
            SS_CHK(hipGetLastError());
        }
        if (der_chunks) {
            uint8_t* blk = ctx->h_stage + der_off;
            DSeg* hs = reinterpret_cast<DSeg*>(blk);
            int32_t* hf = reinterpret_cast<int32_t*>(blk + sizeof(DSeg) * std::max<size_t>(n_seg, 1));
            uint32_t* hc = reinterpret_cast<uint32_t*>(hf + n_dseq + 1);
            size_t w = 0;
            for (size_t t = 0; t < n_dseq; ++t) {
                const auto& v = t < s->derived.size() ? s->derived[t] : hidden[t - s->derived.size()];
                hf[t] = (int32_t)w;
                for (const HSeg& x : v) hs[w++] = DSeg{s->h[(size_t)x.parent].chunk0, x.off, x.len, x.dst, x.rc ? 1u : 0u};
                const SeqDesc& dd = s->h[(size_t)n_seqs + t];
                for (size_t c = 0; c < ((size_t)dd.len + 31) / 32; ++c) hc[dd.asc0 + c] = (uint32_t)((size_t)n_seqs + t);
            }
            hf[n_dseq] = (int32_t)w;
            uint8_t* d_blk = ctx->d_stage + der_off;
            SS_CHK(hipMemcpyAsync(d_blk, blk, der_bytes, hipMemcpyHostToDevice, ctx->stream));
            const DSeg* dsg = reinterpret_cast<const DSeg*>(d_blk);
            const int32_t* dsf = reinterpret_cast<const int32_t*>(d_blk + sizeof(DSeg) * std::max<size_t>(n_seg, 1));
            const uint32_t* dsc = reinterpret_cast<const uint32_t*>(dsf + n_dseq + 1);
            hipLaunchKernelGGL(derive_kernel, dim3((unsigned)((der_chunks + 255) / 256)), dim3(256), 0, ctx->stream, s->d_seqs, dsc,
                               (uint32_t)der_chunks, dsf, dsg, n_seqs, s->d_p2, s->d_e1, s->d_x4);
            SS_CHK(hipGetLastError());
        }
        tq[4] = now();
        SS_CHK(hipMemcpyAsync(s->h.data(), s->d_seqs, sizeof(SeqDesc) * s->h.size(), hipMemcpyDeviceToHost, ctx->stream));
        SS_CHK(hipStreamSynchronize(ctx->stream));
        tq[5] = now();
        if (dbg_t) fprintf(stderr, "seqset: layout+groups %.3f  alloc+memset %.3f  stage+h2d %.3f  launches %.3f  sync %.3f ms\n", tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4]);
        // complementary() drops what is not ATGCN / atgcn (SF:471-478): a reversed slice of a window that holds such a
        // character is not what the reference would have built
        for (size_t d = 0; d < s->derived.size() && rc == VAPOR_OK; ++d)
            for (const HSeg& x : s->derived[d])
                if (x.rc && s->h[(size_t)x.parent].n_nocomp > 0) {
                    rc = fail(VAPOR_E_ARG, "vapor_seqset_create_derived: a reverse-complemented segment's parent holds characters complementary() drops "
                                           "(outside ATGCN/atgcn, SF:471-478); upload that allele as bytes");
                    break;
                }
        if (seq_info && rc == VAPOR_OK)
            for (int32_t i = 0; i < s->n; ++i) {
                seq_info[2 * i] = s->h[i].n_exc;
                seq_info[2 * i + 1] = s->h[i].n_invalid;
            }
    }
done:
    if (rc != VAPOR_OK) { vapor_seqset_destroy(s); return rc; }
    *out = s;
    return VAPOR_OK;
#undef SS_CHK
}

extern "C" int vapor_seqset_create(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* blob, const int64_t* off,
                                   const int32_t* len, const uint8_t* flags, int32_t* seq_info, vapor_seqset** out)
{
    if (!ctx || !out || n_seqs < 0 || (n_seqs && (!blob || !off || !len)))
        return fail(VAPOR_E_ARG, "vapor_seqset_create: null argument");
    return seqset_create_impl(ctx, n_seqs, [&](int32_t i) { return blob + off[i]; }, len, flags, seq_info, out);
}

extern "C" int vapor_seqset_create_ptrs(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* const* seq, const int32_t* len,
                                        const uint8_t* flags, int32_t* seq_info, vapor_seqset** out)
{
    if (!ctx || !out || n_seqs < 0 || (n_seqs && (!seq || !len)))
        return fail(VAPOR_E_ARG, "vapor_seqset_create_ptrs: null argument");
    for (int32_t i = 0; i < n_seqs; ++i)
        if (len[i] > 0 && !seq[i]) return fail(VAPOR_E_ARG, "vapor_seqset_create_ptrs: null sequence");
    return seqset_create_impl(ctx, n_seqs, [&](int32_t i) { return seq[i]; }, len, flags, seq_info, out);
}

extern "C" int vapor_seqset_create_derived(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* const* seq, const int32_t* len,
                                           const uint8_t* flags, int32_t n_derived, const int32_t* seg_first,
                                           const vapor_segment* segs, const uint8_t* derived_flags, int32_t* seq_info,
                                           vapor_seqset** out)
{
    if (!ctx || !out || n_seqs < 0 || n_derived < 0 || (n_seqs && (!seq || !len)) || (n_derived && (!seg_first || !segs)))
        return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: null argument");
    if ((int64_t)n_seqs + n_derived > 0x7FFFFFF0LL) return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: too many sequences");
    for (int32_t i = 0; i < n_seqs; ++i)
        if (len[i] > 0 && !seq[i]) return fail(VAPOR_E_ARG, "vapor_seqset_create_derived: null sequence");
    return seqset_create_impl(ctx, n_seqs, [&](int32_t i) { return seq[i]; }, len, flags, seq_info, out, n_derived, seg_first, segs,
                              derived_flags);
}

extern "C" int vapor_seqset_planes(vapor_seqset* s, int32_t seq, uint32_t* p2, uint32_t* e1, uint32_t* x4)
{
    if (!s || seq < 0 || seq >= s->n) return fail(VAPOR_E_ARG, "vapor_seqset_planes: no such sequence");
    HIPCHK(hipSetDevice(s->device));
    const SeqDesc& d = s->h[(size_t)seq];
    const size_t ch = ((size_t)d.len + 31) / 32;
    if (!ch) return VAPOR_OK;
    if (p2) HIPCHK(hipMemcpy(p2, s->d_p2 + (size_t)d.chunk0 * 2, ch * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (e1) HIPCHK(hipMemcpy(e1, s->d_e1 + (size_t)d.chunk0, ch * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (x4) HIPCHK(hipMemcpy(x4, s->d_x4 + (size_t)d.chunk0 * 4, ch * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return VAPOR_OK;
}

// ------------------------------------------------------------------------------------------
// Read extraction on the device (vapor_bamdev.h): the host side reads the regions' BGZF blocks as they lie in the file, lays
// out where every block's data goes, sends the compressed bytes and runs the kernels.
// ------------------------------------------------------------------------------------------
extern "C" int vapor_bam_fileno(vapor_bam* b);
extern "C" int vapor_bam_threads(vapor_bam* b);

struct vapor_bam_batch {
    vapor_ctx* ctx = nullptr;
    int device = 0;
    uint8_t* d_arena = nullptr;
    size_t arena_bytes = 0;
};

extern "C" int vapor_bam_batch_destroy(vapor_bam_batch* b)
{
    if (!b) return VAPOR_OK;
    if (b->d_arena) {
        (void)hipSetDevice(b->device);
        if (b->ctx && ctx_alive(b->ctx)) {
            // (kernels that read the arena - bam_expand_kernel of a set made from it - are on the context's stream)
            (void)hipStreamSynchronize(b->ctx->stream);
            b->ctx->arenas.erase(b->d_arena);
        }
        dfree(b->ctx, b->d_arena);
    }
    delete b;
    return VAPOR_OK;
}

namespace {
struct HostSpan {                  // one index chunk of a region on the host side
    int32_t region;
    uint64_t cs, ce;
    int64_t file_off;              // compressed range read from the file
    size_t want, got;
    size_t stage_off;              // ... into the pinned staging buffer here
    std::vector<vapor_bamdev::BgzfBlk> blks;   // c_off relative to the staging buffer, u_off relative to the span's data
    uint32_t u_begin = 0, u_end = 0, u_total = 0;
    bool bad = false;              // not BGZF, or a begin offset outside its block: the host route words the error
};

// the whole blocks of a span, through the block that holds the chunk's end
void scan_span(HostSpan& sp, const uint8_t* stage)
{
    const uint8_t* base = stage + sp.stage_off;
    const int64_t end_coff = (int64_t)(sp.ce >> 16);
    const uint32_t end_uoff = (uint32_t)(sp.ce & 0xFFFFu);
    size_t p = 0;
    uint32_t u = 0;
    bool have_end = false;
    sp.u_end = 0;
    while (p + 18 <= sp.got) {
        const int64_t coff = sp.file_off + (int64_t)p;
        if (coff > end_coff || (coff == end_coff && end_uoff == 0)) break;
        const uint8_t* h = base + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { sp.bad = true; return; }
        const int xlen = h[10] | (h[11] << 8);
        if (p + 12 + (size_t)xlen > sp.got) break;
        int bsize = -1;
        for (int q = 0; q + 4 <= xlen;) {
            const uint8_t* e = h + 12 + q;
            const int slen = e[2] | (e[3] << 8);
            if (e[0] == 66 && e[1] == 67 && slen == 2) bsize = (e[4] | (e[5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 0 || bsize < xlen + 20) { sp.bad = true; return; }
        if (p + (size_t)bsize > sp.got) break;
        const uint8_t* t = h + bsize - 8;
        const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (isize > 65536u) { sp.bad = true; return; }
        if (coff == end_coff) { have_end = true; sp.u_end = u + std::min(end_uoff, isize); }
        if (isize == 0) {
            if (crc != 0) { sp.bad = true; return; }        // (the CRC-32 of no bytes)
        } else {
            vapor_bamdev::BgzfBlk k;
            k.c_off = (uint32_t)(sp.stage_off + p + 12 + (size_t)xlen);
            k.c_len = (uint32_t)(bsize - xlen - 20);
            k.u_off = u;
            k.u_len = isize;
            k.crc = crc;
            k.pad = 0;
            sp.blks.push_back(k);
        }
        u += isize;
        p += (size_t)bsize;
    }
    sp.u_total = u;
    if (!have_end) sp.u_end = u;                            // the chunk ends on a block boundary (or the file ends inside it)
    const uint32_t b0 = (uint32_t)(sp.cs & 0xFFFFu);
    // the first record's offset must lie inside the first block
    uint32_t first_usize = 0;
    {
        // (the first block in the file order, empty ones included, is the one `cs` names)
        const uint8_t* h = base;
        if (sp.got >= 18) {
            const int xlen = h[10] | (h[11] << 8);
            int bsize = -1;
            if (12 + (size_t)xlen <= sp.got)
                for (int q = 0; q + 4 <= xlen;) {
                    const uint8_t* e = h + 12 + q;
                    const int slen = e[2] | (e[3] << 8);
                    if (e[0] == 66 && e[1] == 67 && slen == 2) bsize = (e[4] | (e[5] << 8)) + 1;
                    q += 4 + slen;
                }
            if (bsize >= xlen + 20 && (size_t)bsize <= sp.got) {
                const uint8_t* t = h + bsize - 4;
                first_usize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            }
        }
    }
    if (b0 > first_usize) { sp.bad = true; return; }
    sp.u_begin = b0;
}
}   // namespace

extern "C" int vapor_bam_chop_device(vapor_ctx* ctx, vapor_bam* bam, int32_t n_regions, const int32_t* tid, const int64_t* start,
                                     const int64_t* end, const int64_t* flank, const int32_t* chunk_first, const uint64_t* chunks,
                                     int32_t max_keep, int32_t* kept_first, uint64_t* sq_addr, int64_t* q0, int64_t* miss,
                                     int32_t* status, vapor_bam_batch** out)
{
    using namespace vapor_bamdev;
    if (!ctx || !bam || !out || n_regions < 0 || max_keep < 1 || max_keep > KEPT_CAP ||
        (n_regions && (!tid || !start || !end || !flank || !chunk_first || !kept_first || !sq_addr || !q0 || !miss || !status)))
        return fail(VAPOR_E_ARG, "vapor_bam_chop_device: bad argument");
    const int fd = vapor_bam_fileno(bam);
    if (fd < 0) return fail(VAPOR_E_ARG, "vapor_bam_chop_device: the file is not open");
    HIPCHK(hipSetDevice(ctx->device));
    *out = nullptr;
    const bool dbg_t = getenv("VAPOR_DEBUG_BAMDEV") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq[8] = {now(), 0, 0, 0, 0, 0, 0, 0};
    try {
        // ---- what to read ---------------------------------------------------------------------------------------------------
        std::vector<HostSpan> spans;
        std::vector<int32_t> span_first((size_t)n_regions + 1, 0);
        size_t stage_bytes = 0;
        for (int32_t g = 0; g < n_regions; ++g) {
            span_first[(size_t)g] = (int32_t)spans.size();
            status[g] = 0;
            const int32_t c0 = chunk_first[g], c1 = chunk_first[g + 1];
            // (positions are 32-bit in a BAM file; a region that is not is the host route's to refuse)
            bool ok = c1 >= c0 && (c0 == c1 || chunks) && start[g] >= 0 && end[g] >= start[g] && end[g] < ((int64_t)1 << 31) && flank[g] >= 0 && tid[g] >= 0;
            for (int32_t c = c0; ok && c < c1; ++c) {
                const uint64_t cs = chunks[2 * (size_t)c], ce = chunks[2 * (size_t)c + 1];
                if (ce < cs || (ce >> 16) - (cs >> 16) > ((uint64_t)1 << 27)) { ok = false; break; }
                HostSpan sp;
                sp.region = g; sp.cs = cs; sp.ce = ce;
                sp.file_off = (int64_t)(cs >> 16);
                sp.want = (size_t)((int64_t)(ce >> 16) - sp.file_off) + ((ce & 0xFFFFu) ? ((size_t)1 << 16) + 64 : 0);
                sp.got = 0;
                sp.stage_off = stage_bytes;
                stage_bytes += (sp.want + 63) & ~(size_t)63;
                spans.push_back(std::move(sp));
            }
            if (!ok) {
                status[g] = REG_MALFORMED;
                while (!spans.empty() && spans.back().region == g) { stage_bytes = spans.back().stage_off; spans.pop_back(); }
            }
        }
        span_first[(size_t)n_regions] = (int32_t)spans.size();
        if (stage_bytes > ((size_t)3 << 29)) return fail(VAPOR_E_ARG, "vapor_bam_chop_device: more than 1.5 GB of blocks in one call (use smaller batches)");
        // (what the call holds while it runs goes back to the context's pool on every way out, an exception's included; the batch
        // survives a successful return only)
        struct Held {
            vapor_ctx* ctx;
            vapor_bam_batch* B = nullptr;
            uint8_t *h_comp = nullptr, *d_comp = nullptr, *h_meta = nullptr, *d_meta = nullptr;
            explicit Held(vapor_ctx* c) : ctx(c) {}
            void release_temporaries()
            {
                if (h_comp) hfree(ctx, h_comp);
                if (d_comp) dfree(ctx, d_comp);
                if (h_meta) hfree(ctx, h_meta);
                if (d_meta) dfree(ctx, d_meta);
                h_comp = d_comp = h_meta = d_meta = nullptr;
            }
            ~Held()
            {
                release_temporaries();
                if (B) vapor_bam_batch_destroy(B);
            }
        } held(ctx);
        held.B = new vapor_bam_batch();
        vapor_bam_batch*& B = held.B;
        B->ctx = ctx;
        B->device = ctx->device;
        uint8_t*& h_comp = held.h_comp;
        uint8_t*& d_comp = held.d_comp;
        uint8_t*& h_meta = held.h_meta;
        uint8_t*& d_meta = held.d_meta;
        auto cleanup = [&](int code) { return code; };      // (the destructor above does the work)
#define BD_CHK(expr)                                                                                                  \
    do {                                                                                                              \
        hipError_t _e = (expr);                                                                                       \
        if (_e != hipSuccess) return cleanup(fail(VAPOR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)));   \
    } while (0)
        BD_CHK(hmalloc(ctx, (void**)&h_comp, std::max<size_t>(stage_bytes, 64)));
        // ---- read and scan, a few threads -------------------------------------------------------------------------------------
        {
            const int n_thr = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(vapor_bam_threads(bam), 1) * 2, spans.size() / 8 + 1));
            std::atomic<size_t> next{0};
            auto work = [&] {
                for (;;) {
                    const size_t i = next.fetch_add(1, std::memory_order_relaxed);
                    if (i >= spans.size()) break;
                    HostSpan& sp = spans[i];
                    size_t got = 0;
                    while (got < sp.want) {
                        const ssize_t r = pread(fd, h_comp + sp.stage_off + got, sp.want - got, (off_t)(sp.file_off + (int64_t)got));
                        if (r <= 0) break;
                        got += (size_t)r;
                    }
                    sp.got = got;
                    try {
                        scan_span(sp, h_comp);
                    } catch (const std::exception&) {       // (out of memory for the block list: the region goes the host route)
                        sp.blks.clear();
                        sp.bad = true;
                    }
                }
            };
            if (n_thr <= 1) {
                work();
            } else {
                std::vector<std::thread> th;
                for (int t = 1; t < n_thr; ++t) th.emplace_back(work);
                work();
                for (auto& x : th) x.join();
            }
        }
        tq[1] = now();
        // ---- layout of the arena and the tables -------------------------------------------------------------------------------
        for (const HostSpan& sp : spans)
            if (sp.bad) status[sp.region] = REG_MALFORMED;
        std::vector<BgzfBlk> blks;
        std::vector<BamSpan> dspans;
        std::vector<BamRegion> regs((size_t)std::max(n_regions, 1));
        size_t arena = 0;
        uint32_t max_usize = 0;
        for (int32_t g = 0; g < n_regions; ++g) {
            BamRegion& R = regs[(size_t)g];
            R.start = start[g]; R.end = end[g]; R.flank = flank[g]; R.tid = tid[g]; R.pad = 0;
            R.span_first = (int32_t)dspans.size();
            R.span_n = 0;
            if (status[g]) continue;
            for (int32_t si = span_first[(size_t)g]; si < span_first[(size_t)g + 1]; ++si) {
                const HostSpan& sp = spans[(size_t)si];
                if (arena + sp.u_total + 64 > ((size_t)1 << 31)) return cleanup(fail(VAPOR_E_ARG, "vapor_bam_chop_device: more than 2 GB of block data in one call (use smaller batches)"));
                BamSpan d;
                d.u_begin = (uint32_t)arena + sp.u_begin;
                d.u_end = (uint32_t)arena + sp.u_end;
                d.u_limit = (uint32_t)arena + sp.u_total;
                d.blk_first = (uint32_t)blks.size();
                d.blk_n = (uint32_t)sp.blks.size();
                d.pad = 0;
                for (BgzfBlk k : sp.blks) {
                    k.u_off += (uint32_t)arena;
                    max_usize = std::max(max_usize, k.u_len);
                    blks.push_back(k);
                }
                dspans.push_back(d);
                ++R.span_n;
                arena += ((size_t)sp.u_total + 63) & ~(size_t)63;
            }
        }
        const size_t n_blks = blks.size();
        // metadata in one block: blocks, spans, regions in; block status, kept counts, region status, kept reads out
        const size_t o_blk = 0, o_span = o_blk + ((sizeof(BgzfBlk) * std::max<size_t>(n_blks, 1) + 63) & ~(size_t)63);
        const size_t o_reg = o_span + ((sizeof(BamSpan) * std::max<size_t>(dspans.size(), 1) + 63) & ~(size_t)63);
        const size_t in_bytes = o_reg + ((sizeof(BamRegion) * regs.size() + 63) & ~(size_t)63);
        const size_t o_bst = in_bytes, o_nk = o_bst + ((4 * std::max<size_t>(n_blks, 1) + 63) & ~(size_t)63);
        const size_t o_rst = o_nk + ((4 * regs.size() + 63) & ~(size_t)63), o_kept = o_rst + ((4 * regs.size() + 63) & ~(size_t)63);
        const size_t meta_bytes = o_kept + sizeof(BamKept) * KEPT_CAP * regs.size();
        BD_CHK(hmalloc(ctx, (void**)&h_meta, meta_bytes));
        BD_CHK(dmalloc(ctx, (void**)&d_meta, meta_bytes));
        BD_CHK(dmalloc(ctx, (void**)&d_comp, std::max<size_t>(stage_bytes, 64)));
        BD_CHK(dmalloc(ctx, (void**)&B->d_arena, arena + 64));
        B->arena_bytes = arena + 64;
        ctx->arenas[B->d_arena] = B->arena_bytes;
        if (n_blks) memcpy(h_meta + o_blk, blks.data(), sizeof(BgzfBlk) * n_blks);
        if (!dspans.empty()) memcpy(h_meta + o_span, dspans.data(), sizeof(BamSpan) * dspans.size());
        memcpy(h_meta + o_reg, regs.data(), sizeof(BamRegion) * regs.size());
        if (!ctx->d_crc_pow) {
            // x^(8 * 1024 * (63 - l)) mod P, l = 0 .. 63 (bit 31 = x^0): what lane l's slice CRC is multiplied by
            uint32_t pw[64];
            auto mul = [](uint32_t a, uint32_t b) {
                uint32_t m = 1u << 31, p = 0;
                for (int i = 0; i < 32; ++i) { if (a & m) p ^= b; m >>= 1; b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
                return p;
            };
            for (int l = 0; l < 64; ++l) {
                uint64_t e = (uint64_t)8 * 1024 * (uint64_t)(63 - l);
                uint32_t r = 0x80000000u, b = 0x40000000u;
                while (e) { if (e & 1) r = mul(r, b); b = mul(b, b); e >>= 1; }
                pw[l] = r;
            }
            BD_CHK(hipMalloc((void**)&ctx->d_crc_pow, sizeof pw));
            BD_CHK(hipMemcpy(ctx->d_crc_pow, pw, sizeof pw, hipMemcpyHostToDevice));
        }
        {
            const char* sh = getenv("VAPOR_BAM_CU_SHARE");            // (experiments: overrides the parameter)
            const int share = ctx->user_stream ? 0 : (sh ? atoi(sh) : ctx->bam_cu_share);
            if (share != ctx->bam_stream_share) {
                if (ctx->bam_stream) { (void)hipStreamSynchronize(ctx->bam_stream); (void)hipStreamDestroy(ctx->bam_stream); ctx->bam_stream = nullptr; }
                ctx->bam_stream_share = share;
                if (share >= 1 && share <= 7) {
                    // (CU i of the mask's enumeration is on in s of every 8: every XCD keeps CUs of both kinds)
                    std::vector<uint32_t> mask((size_t)(ctx->n_cus + 31) / 32, 0u);
                    for (int i = 0; i < ctx->n_cus; ++i)
                        if (i % 8 < share) mask[(size_t)i / 32] |= 1u << (i % 32);
                    if (hipExtStreamCreateWithCUMask(&ctx->bam_stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) ctx->bam_stream = nullptr;
                }
            }
        }
        hipStream_t st = ctx->bam_stream ? ctx->bam_stream : ctx->stream;
        tq[2] = now();
        if (stage_bytes) BD_CHK(hipMemcpyAsync(d_comp, h_comp, stage_bytes, hipMemcpyHostToDevice, st));
        if (dbg_t) { BD_CHK(hipStreamSynchronize(st)); tq[3] = now(); }
        BD_CHK(hipMemcpyAsync(d_meta, h_meta, in_bytes, hipMemcpyHostToDevice, st));
        if (!ctx->bam_ev[0]) { BD_CHK(hipEventCreate(&ctx->bam_ev[0])); BD_CHK(hipEventCreate(&ctx->bam_ev[1])); }
        BD_CHK(hipEventRecord(ctx->bam_ev[0], st));
        if (n_blks) {
            hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)((n_blks + INFLATE_WAVES - 1) / INFLATE_WAVES)), dim3(64 * INFLATE_WAVES), 0, st, d_comp, reinterpret_cast<const BgzfBlk*>(d_meta + o_blk),
                               (int)n_blks, B->d_arena, ctx->d_crc_pow, reinterpret_cast<int32_t*>(d_meta + o_bst));
            BD_CHK(hipGetLastError());
        }
        BD_CHK(hipEventRecord(ctx->bam_ev[1], st));
        if (dbg_t) { BD_CHK(hipStreamSynchronize(st)); tq[4] = now(); }
        if (n_regions) {
            hipLaunchKernelGGL(bam_chop_kernel, dim3((unsigned)n_regions), dim3(64), 0, st, B->d_arena, reinterpret_cast<const BamRegion*>(d_meta + o_reg),
                               reinterpret_cast<const BamSpan*>(d_meta + o_span), reinterpret_cast<const int32_t*>(d_meta + o_bst), (int)n_regions,
                               reinterpret_cast<BamKept*>(d_meta + o_kept), reinterpret_cast<int32_t*>(d_meta + o_nk), reinterpret_cast<int32_t*>(d_meta + o_rst));
            BD_CHK(hipGetLastError());
        }
        // (counts and statuses first; the kept reads of a region are read where its count says)
        BD_CHK(hipMemcpyAsync(h_meta + o_bst, d_meta + o_bst, meta_bytes - o_bst, hipMemcpyDeviceToHost, st));
        BD_CHK(hipStreamSynchronize(st));
        tq[5] = now();
        {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->bam_ev[0], ctx->bam_ev[1]) != hipSuccess) ms = 0.f;
            ctx->bam_stats[0] = n_regions; ctx->bam_stats[1] = (double)n_blks; ctx->bam_stats[2] = (double)stage_bytes;
            ctx->bam_stats[3] = (double)arena; ctx->bam_stats[4] = ms; ctx->bam_stats[5] = tq[5] - tq[0];
        }
        if (dbg_t)
            fprintf(stderr, "bam_chop_device: %d regions, %zu blocks, %.1f MB compressed -> %.1f MB; read+scan %.2f  layout+alloc %.2f  h2d %.2f  inflate %.2f  chop+d2h %.2f ms\n",
                    n_regions, n_blks, stage_bytes / 1e6, arena / 1e6, tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4]);
#ifdef VBD_TIMING
        if (dbg_t && n_blks) {
            std::vector<uint8_t> back(stage_bytes);
            BD_CHK(hipMemcpy(back.data(), d_comp, stage_bytes, hipMemcpyDeviceToHost));
            double sum[13] = {0};
            size_t cnt = 0;
            for (const BgzfBlk& k : blks) {
                if (k.c_len < 128) continue;
                const long long* d = reinterpret_cast<const long long*>(back.data() + ((k.c_off + 7u) & ~7u));
                for (int t = 0; t < 13; ++t) sum[t] += (double)d[t];
                ++cnt;
            }
            fprintf(stderr, "  per block (shader clocks, mean of %zu): top-up %.0f  decode %.0f (tables %.0f)  matches %.0f  crc %.0f  all %.0f\n"
                            "  counts: fast literal steps %.0f, fast general symbols %.0f, careful symbols %.0f, matches %.0f (fills %.0f), batches %.0f; clocks in the fast loop's general symbols %.0f\n", cnt,
                    sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[10] / cnt, sum[7] / cnt, sum[8] / cnt,
                    sum[11] / cnt, sum[9] / cnt, sum[12] / cnt);
        }
#endif
        // ---- minimize_pacbio_read_list (SF:1091-1102): at most max_keep, the smallest miss_bp first, file order inside one value
        const int32_t* nk = reinterpret_cast<const int32_t*>(h_meta + o_nk);
        const int32_t* rst = reinterpret_cast<const int32_t*>(h_meta + o_rst);
        const BamKept* kept = reinterpret_cast<const BamKept*>(h_meta + o_kept);
        int32_t w = 0;
        std::vector<int32_t> order;
        for (int32_t g = 0; g < n_regions; ++g) {
            kept_first[g] = w;
            if (status[g]) continue;
            if (rst[g] != REG_OK) { status[g] = rst[g]; continue; }
            const int32_t n = nk[g];
            const BamKept* k = kept + (size_t)g * KEPT_CAP;
            order.resize((size_t)n);
            for (int32_t i = 0; i < n; ++i) order[(size_t)i] = i;
            if (n > max_keep) {
                std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return k[a].miss < k[b].miss; });
                order.resize((size_t)max_keep);
            }
            for (int32_t i : order) {
                sq_addr[w] = (uint64_t)reinterpret_cast<uintptr_t>(B->d_arena + k[i].sq_off);
                q0[w] = k[i].q0;
                miss[w] = k[i].miss;
                ++w;
            }
        }
        kept_first[n_regions] = w;
        held.release_temporaries();
        *out = B;
        held.B = nullptr;
        return VAPOR_OK;
#undef BD_CHK
    } catch (const std::bad_alloc&) {
        return fail(VAPOR_E_NOMEM, "vapor_bam_chop_device: out of memory");
    } catch (const std::exception& e) {             // (no exception crosses the C boundary: a thread that could not start, ...)
        return fail(VAPOR_E_ARG, std::string("vapor_bam_chop_device: ") + e.what());
    }
}

// what the context's last vapor_bam_chop_device did: regions, blocks, compressed bytes sent, inflated bytes, the inflate kernel's
// duration between two events on its stream (ms), the whole call on the host's clock (ms)
extern "C" int vapor_bam_last_stats(vapor_ctx* ctx, double* out, int32_t n)
{
    if (!ctx || !out || n < 0) return fail(VAPOR_E_ARG, "vapor_bam_last_stats: null argument");
    for (int32_t i = 0; i < n && i < 6; ++i) out[i] = ctx->bam_stats[i];
    return VAPOR_OK;
}

// vapor_seqset_create_derived with sequences whose bases are on the device already (vapor_bam_chop_device's reads): src_kind[i] = 1
// says seq[i] is the DEVICE address of BAM-packed bases (4 bits each, high nibble first) inside the arena of a live batch of this
// context and src_first[i] the first base; len[i] bases from there are the sequence.  0: bytes on the host, as ever.
extern "C" int vapor_seqset_create_mixed(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* const* seq, const int32_t* len,
                                         const uint8_t* flags, const uint8_t* src_kind, const int64_t* src_first,
                                         int32_t n_derived, const int32_t* seg_first, const vapor_segment* segs,
                                         const uint8_t* derived_flags, int32_t* seq_info, vapor_seqset** out)
{
    if (!ctx || !out || n_seqs < 0 || n_derived < 0 || (n_seqs && (!seq || !len)) || (n_derived && (!seg_first || !segs)))
        return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: null argument");
    if ((int64_t)n_seqs + n_derived > 0x7FFFFFF0LL) return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: too many sequences");
    bool any_dev = false;
    for (int32_t i = 0; i < n_seqs; ++i) {
        if (len[i] < 0) return fail(VAPOR_E_ARG, "negative sequence length");
        if (len[i] > 0 && !seq[i]) return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: null sequence");
        if (!src_kind || !src_kind[i]) continue;
        if (src_kind[i] != 1 || !src_first || src_first[i] < 0 || src_first[i] > 0x7FFFFFF0LL)
            return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: bad source description");
        any_dev = true;
        if (len[i] == 0) continue;
        // the bytes that will be read must lie inside the arena of a batch that is alive
        const uint8_t* a = seq[i] + (src_first[i] >> 1);
        const uint8_t* b = seq[i] + ((src_first[i] + len[i] - 1) >> 1);
        auto it = ctx->arenas.upper_bound(a);
        if (it == ctx->arenas.begin()) return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: a device source outside every live batch");
        --it;
        if (b >= it->first + it->second) return fail(VAPOR_E_ARG, "vapor_seqset_create_mixed: a device source outside every live batch");
    }
    return seqset_create_impl(ctx, n_seqs, [&](int32_t i) { return seq[i]; }, len, flags, seq_info, out, n_derived, seg_first, segs,
                              derived_flags, any_dev ? src_kind : nullptr, src_first);
}

// ------------------------------------------------------------------------------------------
static void plan_free_device(vapor_plan* p)
{
    dfree(p->ctx, p->d_pairs); p->d_pairs = nullptr;
    dfree(p->ctx, p->d_tasks); p->d_tasks = nullptr;
    dfree(p->ctx, p->d_task_pairs); p->d_task_pairs = nullptr;
    dfree(p->ctx, p->d_hits); p->d_hits = nullptr;
    dfree(p->ctx, p->d_hflags); p->d_hflags = nullptr;
    dfree(p->ctx, p->d_nhits); p->d_nhits = nullptr;
    dfree(p->ctx, p->d_stats); p->d_stats = nullptr;
    dfree(p->ctx, p->d_reads); p->d_reads = nullptr;
    p->d_locus_first = nullptr;           // (inside d_reads' block)
    if (p->own_gt) dfree(p->ctx, p->d_gt);
    p->d_gt = nullptr;
    dfree(p->ctx, p->d_read_scores); p->d_read_scores = nullptr;
    dfree(p->ctx, p->d_loci); p->d_loci = nullptr;
    dfree(p->ctx, p->d_shares); p->d_shares = nullptr;
    dfree(p->ctx, p->d_maps); p->d_maps = nullptr;
    dfree(p->ctx, p->d_serve); p->d_serve = nullptr;
    dfree(p->ctx, p->d_clean_order); p->d_clean_order = nullptr;
}

extern "C" int vapor_plan_destroy(vapor_plan* p)
{
    if (!p) return VAPOR_OK;
    (void)hipSetDevice(p->device);
    if (p->ring_n > 0 && p->lane && ctx_alive(p->ctx)) {                                       // steps in flight use the blocks
        (void)hipStreamSynchronize(p->lane);
        if (p->fin) (void)hipStreamSynchronize(p->fin);
    }
    plan_free_device(p);
    hfree(p->ctx, p->h_stats);
    hfree(p->ctx, p->h_overflow);
    dfree(p->ctx, p->d_overflow);
    dfree(p->ctx, p->d_big_list);
    for (auto& e : p->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& r : p->ring)
        for (auto& e : r)
            if (e) (void)hipEventDestroy(e);
    hfree(p->ctx, p->h_loci);
    for (auto& e : p->ev_f)
        if (e) (void)hipEventDestroy(e);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_last) (void)hipEventDestroy(p->ev_last);
    if (p->ev_clean) (void)hipEventDestroy(p->ev_clean);
    if (p->ev_fin) (void)hipEventDestroy(p->ev_fin);
    if (p->ev_after) (void)hipEventDestroy(p->ev_after);
    delete p;
    return VAPOR_OK;
}

static_assert((2 * VAPOR_MAX_SEQ_LEN + 2 + 31) / 32 <= CLEAN_RANGE_WORDS_MAX, "cluster_axis sizes its per-thread word list for this");

static bool k_supported(int k) { return k == 10 || k == 20 || k == 30 || k == 40; }

// lays out the hit workspace from hp[].cap and (re)allocates it
static int plan_alloc_hits(vapor_plan* p)
{
    int64_t tot = 0;
    for (auto& d : p->hp) {
        d.hit_off = tot;
        tot += (int64_t)((d.cap + 3u) & ~3u);
    }
    tot += 4;
    HIPCHK(hipStreamSynchronize(p->ctx->stream));      // (a rerun: nothing may still read the old slots when they are reused)
    dfree(p->ctx, p->d_hits); p->d_hits = nullptr;
    dfree(p->ctx, p->d_hflags); p->d_hflags = nullptr;
    HIPCHK(dmalloc(p->ctx, (void**)&p->d_hits, (size_t)tot * sizeof(unsigned long long)));
    HIPCHK(dmalloc(p->ctx, (void**)&p->d_hflags, (size_t)tot));
    p->total_cap = tot;
    HIPCHK(hipMemcpyAsync(p->d_pairs, p->hp.data(), sizeof(DPair) * p->hp.size(), hipMemcpyHostToDevice, p->ctx->stream));
    if (p->n_dpairs) {                                  // the served pairs' view of the shared dot plots' slots
        for (auto& sv : p->serve)
            if (sv.dpair >= 0) { sv.hit_off = p->hp[(size_t)sv.dpair].hit_off; sv.cap = p->hp[(size_t)sv.dpair].cap; }
        HIPCHK(hipMemcpyAsync(p->d_serve, p->serve.data(), sizeof(DServe) * p->serve.size(), hipMemcpyHostToDevice, p->ctx->stream));
    }
    return VAPOR_OK;
}

// The clean kernel is one workgroup per pair, dealt out in grid order, a few per CU at a time: a plan of a couple of rounds of
// them (4 000 pairs at seven or eight per CU: two rounds and a bit) ends when the LAST round's slowest workgroup does, so the
// pairs that take longest go first and the tail is made of the shortest (longest-processing-time order; measured on cfg2: clean
// 0.091 -> 0.079 ms, profiles/r05_clean_order.txt).  What a pair takes: its records - about the shorter sequence's length -
// times the passes its flags ask for: C1 two clusterings, C2 one or two more, the directed statistics five passes over the kept
// records, plus the cutting of a served pair.  Pairs of equal cost keep their order (a read's two pairs lie side by side and
// read the same shared plot); sorting by the record counts a blocking run measured instead (`records`) was tried and is no
// better - it scatters those neighbours (0.0738 -> 0.0749 ms) - so the library does not use it.
static int plan_clean_order(vapor_plan* p, const unsigned long long* records)
{
    const int64_t n_pairs = p->n_pairs;
    std::vector<std::pair<int64_t, int32_t>> cost((size_t)n_pairs);
    for (int64_t i = 0; i < n_pairs; ++i) {
        const DPair& d = p->hp[(size_t)i];
        const bool c1 = d.flags & VAPOR_PF_C1, c2 = d.flags & VAPOR_PF_C2, dir = (d.flags & VAPOR_PF_DIR) && c1;
        int64_t w = 4 + (c1 ? 8 : 0) + (c2 ? (c1 ? 5 : 8) : 0) + (dir ? 7 : 0);
        if (!p->serve.empty() && p->serve[(size_t)i].dpair >= 0) w += p->serve[(size_t)i].slot == 0 ? 2 : 4;
        const int64_t size = records ? (int64_t)(uint32_t)records[(size_t)i] * 4 + 256
                                     : std::min<int64_t>(d.len1, std::max(0, d.len2 - d.off2));
        cost[(size_t)i] = {-(p->status[(size_t)i] == 0 ? w * size : 0), (int32_t)i};
    }
    std::stable_sort(cost.begin(), cost.end());
    std::vector<int32_t> ord((size_t)n_pairs);
    for (int64_t i = 0; i < n_pairs; ++i) ord[(size_t)i] = cost[(size_t)i].second;
    HIPCHK(hipMemcpy(p->d_clean_order, ord.data(), sizeof(int32_t) * (size_t)n_pairs, hipMemcpyHostToDevice));
    return VAPOR_OK;
}

extern "C" int vapor_plan_create(vapor_ctx* ctx, vapor_seqset* set, int64_t n_pairs, const vapor_pair* pairs,
                                 vapor_plan** out)
{
    if (!ctx || !set || !out || n_pairs < 0 || (n_pairs && !pairs))
        return fail(VAPOR_E_ARG, "vapor_plan_create: null argument");
    if (n_pairs > 0x7FFFFFF0LL) return fail(VAPOR_E_ARG, "too many pairs");
    HIPCHK(hipSetDevice(ctx->device));
    vapor_plan* p = new (std::nothrow) vapor_plan();
    if (!p) return fail(VAPOR_E_NOMEM, "out of memory");
    p->ctx = ctx;
    p->device = ctx->device;
    p->set = set;
    p->n_pairs = n_pairs;
    p->hp.resize((size_t)std::max<int64_t>(n_pairs, 1));
    memset(p->hp.data(), 0, sizeof(DPair) * p->hp.size());
    p->status.assign((size_t)n_pairs, 0);
    std::vector<int32_t> order;
    order.reserve((size_t)n_pairs);
    std::vector<uint8_t> mode((size_t)n_pairs, 2);
    int rw = 1;
    int64_t hwant = 1024;
    for (int64_t i = 0; i < n_pairs; ++i) {
        const vapor_pair& a = pairs[i];
        DPair& d = p->hp[i];
        d.seq1 = a.seq1; d.seq2 = a.seq2; d.off2 = a.off2; d.k = a.k; d.flags = a.flags; d.cap = 0;
        if (a.seq1 < 0 || a.seq1 >= set->n || a.seq2 < 0 || a.seq2 >= set->n || a.off2 < 0 || !k_supported(a.k)) {
            p->status[i] = VAPOR_E_ARG;
            d.seq1 = d.seq2 = 0;
            continue;
        }
        const SeqDesc& s1 = set->h[a.seq1];
        const SeqDesc& s2 = set->h[a.seq2];
        d.len1 = s1.len; d.len2 = s2.len;
        if (s1.len > VAPOR_MAX_SEQ_LEN || s2.len > VAPOR_MAX_SEQ_LEN) { p->status[i] = VAPOR_E_ARG; continue; }
        if (s1.len - a.k + 1 > 0 && s1.n_invalid > 0) { p->status[i] = VAPOR_E_KEYERROR; continue; }
        int64_t n1 = s1.len, n2 = std::max(0, s2.len - a.off2);
        int64_t cap = std::min(n1, n2) + ((n1 * n2) >> 17) + 1024;
        d.cap = (uint32_t)std::min<int64_t>(cap, ctx->max_pair_cap);
        // 2: 2-bit planes; 3: 2-bit planes, the allele has symbols outside upper-case ACGT (a launch of its own: the table
        // leaves their k-mers out and runs end before them); 5: 2-bit planes, the read has such symbols (a launch of its own:
        // their positions are masked out of the lookup, runs end before them); 4: both sides have them - the 4-bit planes
        mode[i] = (s1.n_exc > 0 && s2.n_exc > 0) ? 4 : (s2.n_exc > 0 ? 3 : (s1.n_exc > 0 ? 5 : 2));
        rw = std::max(rw, (s1.len + s2.len + 2 + 31) / 32);
        // records expected: the shared diagonal in runs of a few dots plus the chance dots
        hwant = std::max<int64_t>(hwant, std::min(n1, n2) / 10 + ((n1 * n2) >> 19) + 192);
        if (s1.len - a.k + 1 > 0 && s2.len - a.k + 1 > 0) order.push_back((int32_t)i);
    }
    p->range_words_cap = rw;
    p->hcap_want = (int)std::min<int64_t>(hwant, CLEAN_HCAP_MAX);
    // Shared joins: a read that is scored against a window AND against alleles derived from it (the usual case: the
    // reference's dotdata(read, ref) and dotdata(read, alt), SF:185-186) is joined once against the group's hidden sequence -
    // the window followed by the alleles' own stretches - and remap_kernel cuts that dot plot into the targets'.
    if (ctx->shared_join && !set->groups.empty() && n_pairs > 0) {
        struct Cand { int32_t seq1, group, k, slot, pair; };
        std::vector<Cand> cand;
        for (int32_t x : order) {
            const DPair& d = p->hp[(size_t)x];
            const int32_t g = set->group_of[(size_t)d.seq2];
            if (g >= 0 && set->groups[(size_t)g].t_seq >= 0) cand.push_back(Cand{d.seq1, g, d.k, set->slot_of[(size_t)d.seq2], x});
        }
        std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) {
            if (a.seq1 != b.seq1) return a.seq1 < b.seq1;
            if (a.group != b.group) return a.group < b.group;
            if (a.k != b.k) return a.k < b.k;
            if (a.slot != b.slot) return a.slot < b.slot;
            return a.pair < b.pair;
        });
        std::map<std::pair<int32_t, int32_t>, std::pair<int32_t, int32_t>> maps_of;      // (group, k) -> (first map, maps); n < 0: cannot
        auto build_maps = [&](int32_t gi, int k) -> std::pair<int32_t, int32_t> {
            auto it = maps_of.find({gi, k});
            if (it != maps_of.end()) return it->second;
            const ShareGroup& g = set->groups[(size_t)gi];
            const int32_t first = (int32_t)p->maps.size();
            const int32_t n_r = set->h[(size_t)g.parent].len;
            bool ok = true;
            if (n_r >= k) p->maps.push_back(DMap{0, n_r - k, 0, 0, 0});
            for (size_t m = 0; m < g.members.size() && ok; ++m) {
                const auto& sg = set->derived[(size_t)(g.members[m] - set->n_lit)];
                const int32_t n_a = set->h[(size_t)g.members[m]].len;
                int32_t u = 0;                                   // next k-mer start of the member not yet accounted for
                auto novel = [&](int32_t from, int32_t to) {     // k-mer starts [from, to] lie in one of the member's own stretches
                    for (const SharePiece& pc : g.pieces)
                        if (pc.slot == (int32_t)m + 1 && pc.a_from <= from && to + k <= pc.a_to) {
                            p->maps.push_back(DMap{pc.tile_off + (from - pc.a_from), pc.tile_off + (to - pc.a_from), from, 0, (uint16_t)(m + 1)});
                            return;
                        }
                    ok = false;
                };
                for (const HSeg& x : sg) {
                    if (x.parent != g.parent || x.len < k) continue;
                    if (x.dst > u) novel(u, x.dst - 1);
                    if (!ok) break;
                    p->maps.push_back(DMap{x.off, x.off + x.len - k, x.rc ? x.dst + x.len - k : x.dst, (uint16_t)(x.rc ? 1 : 0), (uint16_t)(m + 1)});
                    u = x.dst + x.len - k + 1;
                }
                if (ok && u <= n_a - k) novel(u, n_a - k);
            }
            std::pair<int32_t, int32_t> res{-1, -1};
            if (ok && (int32_t)p->maps.size() > first) {
                // the maps cut at each other's ends: boundaries over the k-mer starts of the shared sequence and, per elementary
                // interval, the op words of the maps that cover it (remap_kernel)
                std::vector<int32_t> bd{0};
                for (size_t m = (size_t)first; m < p->maps.size(); ++m) { bd.push_back(p->maps[m].lo); bd.push_back(p->maps[m].hi + 1); }
                std::sort(bd.begin(), bd.end());
                bd.erase(std::unique(bd.begin(), bd.end()), bd.end());
                const int n_iv = (int)bd.size() - 1;
                if (n_iv >= 1 && n_iv <= REMAP_MAX_IV) {
                    std::vector<int32_t> ops((size_t)n_iv * REMAP_OPS, 0);
                    for (int t = 0; t < n_iv && ok; ++t)
                        for (size_t m = (size_t)first; m < p->maps.size(); ++m) {
                            const DMap& mp = p->maps[m];
                            if (!(mp.lo <= bd[(size_t)t] && bd[(size_t)t + 1] - 1 <= mp.hi)) continue;
                            const int32_t delta = mp.flip ? mp.base + mp.lo : mp.base - mp.lo;
                            int32_t* o = &ops[(size_t)t * REMAP_OPS + (size_t)mp.slot * 2];
                            const int c = (o[0] & 1) ? 1 : 0;
                            if (c == 1 && (o[1] & 1)) { ok = false; break; }            // a third copy of one stretch in one allele
                            o[c] = (int32_t)(((uint32_t)delta << 2) | (mp.flip ? 2u : 0u) | 1u);
                        }
                    if (ok) {
                        res = {(int32_t)p->tables.size(), n_iv};
                        p->tables.insert(p->tables.end(), bd.begin(), bd.end());
                        p->tables.insert(p->tables.end(), ops.begin(), ops.end());
                    }
                }
            }
            p->maps.resize((size_t)first);
            maps_of[{gi, k}] = res;
            return res;
        };
        auto tiles = [&](int32_t len_a, int k, int m) {
            const int ta = m != 4 ? tile_pos<JoinCfg, 2>() : tile_pos<JoinCfg, 4>();
            return std::max(1, (len_a - k + 1 + ta - 1) / ta);
        };
        std::vector<uint8_t> served((size_t)n_pairs, 0);
        { DServe none; memset(&none, 0, sizeof(none)); none.dpair = -1; p->serve.assign((size_t)n_pairs, none); }
        for (size_t a = 0; a < cand.size();) {
            size_t b = a;
            while (b < cand.size() && cand[b].seq1 == cand[a].seq1 && cand[b].group == cand[a].group && cand[b].k == cand[a].k) ++b;
            int32_t tgt[4] = {-1, -1, -1, -1};
            int n_t = 0;
            for (size_t c = a; c < b; ++c)
                if (cand[c].slot >= 0 && cand[c].slot < 4 && tgt[cand[c].slot] < 0) { tgt[cand[c].slot] = cand[c].pair; ++n_t; }
            const Cand c0 = cand[a];
            a = b;
            if (n_t < 2) continue;
            const ShareGroup& g = set->groups[(size_t)c0.group];
            const SeqDesc& s1 = set->h[(size_t)c0.seq1];
            const SeqDesc& st = set->h[(size_t)g.t_seq];
            if (st.len - c0.k + 1 <= 0) continue;
            const int md = (s1.n_exc > 0 && st.n_exc > 0) ? 4 : (st.n_exc > 0 ? 3 : (s1.n_exc > 0 ? 5 : 2));
            int sep = 0;
            for (int t = 0; t < 4; ++t)
                if (tgt[t] >= 0) sep += tiles(p->hp[(size_t)tgt[t]].len2, c0.k, mode[(size_t)tgt[t]]);
            if (tiles(st.len, c0.k, md) >= sep) continue;           // (a shared sequence of more tiles than its targets together: no gain)
            const auto mp = build_maps(c0.group, c0.k);
            if (mp.second <= 0) continue;
            DPair d;
            memset(&d, 0, sizeof d);
            d.seq1 = c0.seq1; d.seq2 = g.t_seq; d.off2 = 0; d.k = c0.k; d.flags = 0;
            d.len1 = s1.len; d.len2 = st.len;
            const int64_t n1 = s1.len, n2 = st.len;
            d.cap = (uint32_t)std::min<int64_t>(std::min(n1, n2) + ((n1 * n2) >> 17) + 1024, ctx->max_pair_cap);
            DShare sh;
            memset(&sh, 0, sizeof sh);
            sh.dpair = (int32_t)p->hp.size();
            sh.iv_first = mp.first; sh.n_iv = mp.second;
            bool counted = false;
            for (int t = 0; t < 4; ++t) {
                sh.target[t] = tgt[t];
                if (tgt[t] >= 0) {
                    served[(size_t)tgt[t]] = 1; ++p->n_served;
                    DServe& sv = p->serve[(size_t)tgt[t]];
                    sv.dpair = sh.dpair; sv.iv_first = sh.iv_first; sv.n_iv = sh.n_iv; sv.slot = t;      // (slot and cap of the join: below)
                    sv.pad = counted ? 0 : 1;          // (the target whose clean workgroup counts an overflow of the SHARED plot: once per plot)
                    counted = true;
                }
            }
            p->hp.push_back(d);
            mode.push_back((uint8_t)md);
            p->shares.push_back(sh);
        }
        p->n_dpairs = (int64_t)p->shares.size();
        if (p->n_dpairs) {
            std::vector<int32_t> kept;
            kept.reserve(order.size());
            for (int32_t x : order) if (!served[(size_t)x]) kept.push_back(x);
            for (int64_t t = 0; t < p->n_dpairs; ++t) kept.push_back((int32_t)(n_pairs + t));
            order.swap(kept);
        }
    }
    // Sort by (mode, k, allele): one launch per (mode, k); inside a launch the sorted pair list is
    // cut into contiguous, cost-balanced ranges (tasks).  A workgroup rebuilds its allele hash table
    // only where the allele changes inside its range.
    // (the three fields packed into one word per pair, the index behind it: the comparisons touch nothing else)
    {
        std::vector<std::pair<uint64_t, int32_t>> keyed(order.size());
        for (size_t t = 0; t < order.size(); ++t) {
            const int32_t x = order[t];
            keyed[t] = {((uint64_t)mode[(size_t)x] << 48) | ((uint64_t)(uint32_t)p->hp[(size_t)x].k << 32) | (uint32_t)p->hp[(size_t)x].seq2, x};
        }
        std::sort(keyed.begin(), keyed.end());
        for (size_t t = 0; t < order.size(); ++t) order[t] = keyed[t].second;
    }
    p->task_pairs = order;
    auto tiles_of = [&](int32_t seq2, int k, int m) {
        const int ta = m != 4 ? tile_pos<JoinCfg, 2>() : tile_pos<JoinCfg, 4>();
        return std::max(1, (set->h[seq2].len - k + 1 + ta - 1) / ta);
    };
    for (size_t q = 0; q < order.size();) {
        size_t e = q;
        const int m = mode[order[q]], k = p->hp[order[q]].k;
        while (e < order.size() && mode[order[e]] == m && p->hp[order[e]].k == k) ++e;
        // Cost of a range of consecutive pairs = its probe passes + one table build for its first allele + one for
        // every further allele it reaches into.  The ranges are the contiguous partition into at most `want`
        // pieces whose most expensive piece is cheapest (binary search on that bound, greedy packing under it).
        const size_t n = e - q;
        std::vector<int64_t> probe(n), build(n);
        int64_t total = 0, biggest = 0;
        for (size_t t = q; t < e; ++t) {
            const DPair& d = p->hp[order[t]];
            probe[t - q] = (int64_t)set->h[d.seq1].len * tiles_of(d.seq2, k, m) + 256;
            build[t - q] = (VAPOR_BUILD_COST_X8 * (int64_t)set->h[d.seq2].len) / 8;
            total += probe[t - q] + build[t - q];
            biggest = std::max(biggest, probe[t - q] + build[t - q]);
        }
        // `join_tasks` ranges (one per CU) - or, when the limit of reads per task asks for more than that, WHOLE ROUNDS of them: a
        // launch of 625 tasks on 256 CUs takes three rounds' time for 2.44 rounds' work (BASELINE configs[2]: 40 000 reads in
        // tasks of at most 64), 768 equal tasks take three rounds of 52 reads each - join 2.42 -> 2.10 ms with one plan in flight
        // (profiles/r05_join_rounds.txt; with two plans in flight the other plan's clean kernel filled that tail already)
        int64_t want = std::max<int64_t>(1, std::min<int64_t>((int64_t)n, ctx->join_tasks));
        {
            const int64_t need = ((int64_t)n + ctx->reads_per_task - 1) / ctx->reads_per_task;
            if (need > want) want = std::min<int64_t>((int64_t)n, (need + want - 1) / want * want);
        }
        // (what a pair adds to a range it does not start: its probe, and a table build when it brings a new allele)
        std::vector<int64_t> inside(n);
        for (size_t t = 0; t < n; ++t)
            inside[t] = probe[t] + ((t > 0 && p->hp[order[q + t]].seq2 != p->hp[order[q + t - 1]].seq2) ? build[t] : 0);
        // number of ranges a bound needs (cuts[] = first pair of every range when asked for)
        auto pack = [&](int64_t bound, std::vector<size_t>* cuts) {
            int64_t ranges = 0, acc = 0;
            size_t t0 = 0;
            for (size_t t = 0; t < n; ++t) {
                const int64_t add = t == t0 ? probe[t] + build[t] : inside[t];
                const bool full = (int)(t - t0) >= ctx->reads_per_task;
                if (t > t0 && (acc + add > bound || full)) {
                    ++ranges;
                    if (cuts) cuts->push_back(t0);
                    t0 = t;
                    acc = probe[t] + build[t];
                } else {
                    acc += add;
                }
            }
            ++ranges;
            if (cuts) cuts->push_back(t0);
            return ranges;
        };
        // the smallest bound that needs at most `want` ranges: no partition can do with less than the costs inside
        // ranges shared out evenly, a doubling search finds a bound that is enough, bisection the smallest between
        int64_t in_sum = 0;
        for (size_t t = 0; t < n; ++t) in_sum += inside[t];
        int64_t lo = std::max(biggest, in_sum / want), hi = lo;
        while (hi < total && pack(hi, nullptr) > want) { lo = hi + 1; hi = std::min(total, hi * 2); }
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (pack(mid, nullptr) <= want) hi = mid; else lo = mid + 1;
        }
        std::vector<size_t> cuts;
        pack(lo, &cuts);
        p->launches.push_back(Launch{m == 4 ? 4 : 2, k, (int)p->tasks.size(), 0, m == 3 ? 1 : m == 5 ? 2 : 0});
        for (size_t c = 0; c < cuts.size(); ++c) {
            const size_t t0 = q + cuts[c], t1 = q + (c + 1 < cuts.size() ? cuts[c + 1] : n);
            DTask tk;
            tk.seq2 = p->hp[order[t0]].seq2; tk.k = k; tk.n_reads = (int32_t)(t1 - t0); tk.first = (int32_t)t0;
            p->tasks.push_back(tk);
            p->launches.back().n_tasks++;
        }
        q = e;
    }
    int rc = VAPOR_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == VAPOR_OK) rc = fail(VAPOR_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    chk(dmalloc(ctx, (void**)&p->d_pairs, sizeof(DPair) * p->hp.size()), "hipMalloc pairs");
    chk(dmalloc(ctx, (void**)&p->d_tasks, sizeof(DTask) * std::max<size_t>(p->tasks.size(), 1)), "hipMalloc tasks");
    chk(dmalloc(ctx, (void**)&p->d_task_pairs, sizeof(int32_t) * std::max<size_t>(order.size(), 1)), "hipMalloc task_pairs");
    chk(dmalloc(ctx, (void**)&p->d_nhits, sizeof(unsigned long long) * p->hp.size()), "hipMalloc nhits");
    if (rc == VAPOR_OK) chk(hipMemsetAsync(p->d_nhits, 0, sizeof(unsigned long long) * p->hp.size(), ctx->stream), "memset nhits");
    chk(dmalloc(ctx, (void**)&p->d_stats, sizeof(long long) * 16 * p->hp.size()), "hipMalloc stats");
    chk(hmalloc(ctx, (void**)&p->h_stats, sizeof(long long) * 16 * p->hp.size()), "hipHostMalloc stats");
    chk(dmalloc(ctx, (void**)&p->d_overflow, 4 * sizeof(unsigned int)), "hipMalloc overflow");
    if (rc == VAPOR_OK) chk(hipMemsetAsync(p->d_overflow, 0, 4 * sizeof(unsigned int), ctx->stream), "memset overflow");
    chk(dmalloc(ctx, (void**)&p->d_big_list, sizeof(int32_t) * p->hp.size()), "hipMalloc big list");
    chk(hmalloc(ctx, (void**)&p->h_overflow, 2 * sizeof(unsigned int)), "hipHostMalloc overflow");
    if (p->n_dpairs) {
        chk(dmalloc(ctx, (void**)&p->d_shares, sizeof(DShare) * p->shares.size()), "hipMalloc shares");
        chk(dmalloc(ctx, (void**)&p->d_maps, sizeof(int32_t) * std::max<size_t>(p->tables.size(), 1)), "hipMalloc maps");
        chk(dmalloc(ctx, (void**)&p->d_serve, sizeof(DServe) * p->serve.size()), "hipMalloc serve");
        if (rc == VAPOR_OK) chk(hipMemcpyAsync(p->d_shares, p->shares.data(), sizeof(DShare) * p->shares.size(), hipMemcpyHostToDevice, ctx->stream), "copy shares");
        if (rc == VAPOR_OK && !p->tables.empty())
            chk(hipMemcpyAsync(p->d_maps, p->tables.data(), sizeof(int32_t) * p->tables.size(), hipMemcpyHostToDevice, ctx->stream), "copy maps");
    }
    // (a plan of many rounds of clean workgroups has no tail worth ordering for - cfg3's 62 rounds gain nothing - and sorting
    // 80 000 pairs costs a pipeline chunk's plan 3-4 ms: the order is made for plans of up to eight rounds)
    if (ctx->clean_order && n_pairs > 1 && n_pairs <= (int64_t)8 * (2048 / CLEAN_THREADS) * ctx->n_cus) {
        chk(dmalloc(ctx, (void**)&p->d_clean_order, sizeof(int32_t) * (size_t)n_pairs), "hipMalloc clean order");
        if (rc == VAPOR_OK && plan_clean_order(p, nullptr) != VAPOR_OK) rc = VAPOR_E_HIP;
    }
    for (auto& e : p->ev) chk(hipEventCreate(&e), "hipEventCreate");
    for (auto& e : p->ev_f) chk(hipEventCreate(&e), "hipEventCreate");
    chk(hipEventCreate(&p->ev_t0), "hipEventCreate");
    if (rc == VAPOR_OK && !p->tasks.empty())
        chk(hipMemcpyAsync(p->d_tasks, p->tasks.data(), sizeof(DTask) * p->tasks.size(), hipMemcpyHostToDevice, ctx->stream), "copy tasks");
    if (rc == VAPOR_OK && !order.empty())
        chk(hipMemcpyAsync(p->d_task_pairs, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice, ctx->stream), "copy task_pairs");
    if (rc == VAPOR_OK) rc = plan_alloc_hits(p);
    if (rc == VAPOR_OK) chk(hipStreamSynchronize(ctx->stream), "sync");
    if (rc != VAPOR_OK) { vapor_plan_destroy(p); return rc; }
    p->last_stats.assign((size_t)n_pairs * 16, 0);
    *out = p;
    return VAPOR_OK;
}

template <int BPS, int K>
static void launch_join(vapor_plan* p, const Launch& L, bool first, hipStream_t st)
{
    const vapor_seqset* s = p->set;
    if (BPS == 2 && L.exc == 1)
        hipLaunchKernelGGL((join_kernel<JoinCfg, BPS, K, (BPS == 2 ? 1 : 0)>), dim3((unsigned)L.n_tasks), dim3(JoinCfg::THREADS), JOIN_DYN_LDS(BPS),
                           st, s->d_seqs, s->d_p2, s->d_e1, s->d_x4, p->d_pairs, p->d_tasks + L.task_begin,
                           p->d_task_pairs, p->d_hits, p->d_nhits, first ? p->d_overflow : (unsigned int*)nullptr);
    else if (BPS == 2 && L.exc == 2)
        hipLaunchKernelGGL((join_kernel<JoinCfg, BPS, K, (BPS == 2 ? 2 : 0)>), dim3((unsigned)L.n_tasks), dim3(JoinCfg::THREADS), JOIN_DYN_LDS(BPS),
                           st, s->d_seqs, s->d_p2, s->d_e1, s->d_x4, p->d_pairs, p->d_tasks + L.task_begin,
                           p->d_task_pairs, p->d_hits, p->d_nhits, first ? p->d_overflow : (unsigned int*)nullptr);
    else
        hipLaunchKernelGGL((join_kernel<JoinCfg, BPS, K, 0>), dim3((unsigned)L.n_tasks), dim3(JoinCfg::THREADS), JOIN_DYN_LDS(BPS),
                           st, s->d_seqs, s->d_p2, s->d_e1, s->d_x4, p->d_pairs, p->d_tasks + L.task_begin,
                           p->d_task_pairs, p->d_hits, p->d_nhits, first ? p->d_overflow : (unsigned int*)nullptr);
}

static int clean_groups_cap(int range_words_cap) { return range_words_cap * 32 / 10 + 8; }

// clean_kernel is instantiated for 4, 8 and CLEAN_PER_MAX bitmap words per thread: the smallest that covers the value range
template <typename... A>
static void launch_clean(int range_words_cap, unsigned grid, size_t lds, hipStream_t st, A... a)
{
    const int per = (range_words_cap + CLEAN_THREADS - 1) / CLEAN_THREADS;
    if (per <= 4) hipLaunchKernelGGL(clean_kernel<4>, dim3(grid), dim3(CLEAN_THREADS), lds, st, a...);
    else if (per <= 8) hipLaunchKernelGGL(clean_kernel<8>, dim3(grid), dim3(CLEAN_THREADS), lds, st, a...);
    else hipLaunchKernelGGL(clean_kernel<CLEAN_PER_MAX>, dim3(grid), dim3(CLEAN_THREADS), lds, st, a...);
}

// clean_kernel's LDS: bitmap + 16-bit ranks + group sizes (+ staged hits).  Pairs cleaned out of LDS use
// 16-bit group counters; pairs that stream their hits need 32-bit ones, which must fit as well.
constexpr int CLEAN_BIG_GRID = 1024;      // clean_big_kernel walks its list with at most this many workgroups

// clean_big_kernel: bitmap + 16-bit ranks + 32-bit group sizes (as many groups as the value range can hold)
static size_t clean_fixed_bytes(int rw, bool wide)
{
    const size_t g = (size_t)clean_groups_cap(rw);
    return sizeof(uint32_t) * ((size_t)rw + ((size_t)rw + 1) / 2 + (wide ? g : (g + 1) / 2)) + 64;
}

// clean_kernel: 16-bit group sizes.  A pair staged in LDS has at most hcap records and every group holds at least
// one, so hcap counters do; the same region later holds the per-value counters of the median (value span / 100).
static int clean_groups_lds(int rw, int hcap)
{
    const int g = std::min(clean_groups_cap(rw), std::max(hcap, 1));
    return 2 * std::max((g + 1) / 2, rw * 32 / 100 + 8);
}

// clean_pair's layout: bitmap and ranks of one axis, or of both (`dual`: cluster_dual works on the two axes of C1 at
// once), one region of 16-bit group counters, then the staged records, 8 bytes each with their flag byte inside
static size_t clean_lds_bytes(int rw, int hcap, bool dual)
{
    const size_t bw = (size_t)rw + ((size_t)rw + 1) / 2;
    const size_t rec_word = ((dual ? 2 : 1) * bw + ((size_t)clean_groups_lds(rw, hcap) + 1) / 2 + 1) & ~(size_t)1;
    return sizeof(uint32_t) * rec_word + 64 + (size_t)hcap * 8 + 8;
}

// The kernel waits for memory and barriers more than it computes, so residency matters: take the largest number
// of workgroups per CU (32 waves at most) whose share of the 160 KB still stages ~90 % of the expected records.
struct CleanGeom { int hcap, per_cu; bool dual; };
static CleanGeom clean_geom_for(int range_words_cap, int want, bool dual, bool exact = false)
{
    CleanGeom g{0, 1, dual};
#ifdef VAPOR_DEV_BUILD
    if (const char* e = getenv("VAPOR_DEV_HCAP")) { g.hcap = atoi(e) & ~3; return g; }      // experiment: records staged per pair
#endif
    for (int per_cu = 2048 / CLEAN_THREADS; per_cu >= 1; --per_cu) {
        const size_t share = (size_t)(160 * 1024) / per_cu - 512;
        int cap = std::min(want, CLEAN_HCAP_MAX) & ~3;
        while (cap > 0 && clean_lds_bytes(range_words_cap, cap, dual) + 512 > share) cap -= 4;
        // (an estimated `want`: nine tenths of it staged is enough; a measured one - the largest record count of the plan's
        // pairs - is staged whole, so that no pair is left to clean_big_kernel for the sake of residency)
        if (cap >= (exact ? want : want * 9 / 10) || per_cu == 1) { g.hcap = std::max(cap, 0); g.per_cu = per_cu; break; }
    }
    return g;
}
// Both axes of C1 in one sweep (cluster_dual) need a second bitmap and rank array per workgroup.  Measured (tools/clean_sweep.py,
// profiles/r03_clean_variants.txt): the sweep wins where the extra LDS costs no residency (30 kb x 40 kb pairs, two workgroups
// per CU either way: clean 1.87 -> 1.70 ms) and loses where it does (10 kb x 20 kb: 7 -> 6 per CU, 0.075 vs 0.078 ms; 15 kb x
// 20 kb: 5 -> 4, 1.32 vs 1.42 ms) - so it is used exactly when it is free.
static CleanGeom clean_geom(int range_words_cap, int want, bool exact = false)
{
    const CleanGeom seq = clean_geom_for(range_words_cap, want, false, exact), dual = clean_geom_for(range_words_cap, want, true, exact);
#ifdef VAPOR_DEV_BUILD
    if (const char* e = getenv("VAPOR_DEV_CLEAN_DUAL")) return atoi(e) ? dual : seq;
#endif
    return dual.per_cu >= seq.per_cu ? dual : seq;      // (either stages at least nine tenths of the expected records)
}

static int async_fold(vapor_plan* p);

// Who cuts the served pairs' records out of the shared dot plots: the clean workgroup of each pair, or remap_kernel before the
// cleaning.  Measured (profiles/r04_remap_experiments.txt): the clean workgroups win when the plan is a couple of rounds of them
// - a resident batch of the 10 kb shape, 4 000 pairs at seven workgroups per CU: such a launch is bound by its start, its tail
// and the chains of loads in between, and a kernel boundary and a trip of the records through HBM is what the kernel of its own
// adds (+3 % loci/s) - and lose when it is many rounds (cfg3, 15 kb reads: 80 000 pairs at five per CU, 62 rounds: a launch bound by
// what its workgroups execute, and every target reads the shared plot and searches the interval table again; -3 %).  Between
// the two measured shapes the rule is a guess: the clean workgroups up to four rounds.
// "remap_in_clean": 1 = that rule, 0 = always the kernel, 2 = always the clean workgroups.
static bool remap_in_clean(const vapor_plan* p)
{
    if (p->n_dpairs <= 0 || p->ctx->remap_in_clean == 0) return false;
    if (p->ctx->remap_in_clean == 2) return true;
    const CleanGeom cg = clean_geom(p->range_words_cap, p->hcap_want, p->hcap_measured);
    return p->n_pairs <= 4 * (int64_t)cg.per_cu * p->ctx->n_cus;
}

// keep_flags: the clean kernel writes the per-record flag bytes beside the pairs' records - what vapor_plan_fetch_hits hands out;
// the device-finished path (vapor_plan_run_loci*) needs the statistics only and leaves that pass out.
static int plan_run_once(vapor_plan* p, bool fetch_stats = true, hipEvent_t* evs = nullptr, hipStream_t on = nullptr, bool skip_big = false,
                         hipEvent_t before_clean = nullptr, bool keep_flags = true)
{
    vapor_ctx* c = p->ctx;
    hipStream_t st = on ? on : c->stream;
    hipEvent_t* ev = evs ? evs : p->ev;
    // no memsets in the steady state: the pair counts are stored whole by the join, the clean kernels' two
    // counters are cleared by the first join launch
    if (p->launches.empty()) HIPCHK(hipMemsetAsync(p->d_overflow, 0, 2 * sizeof(unsigned int), st));
    HIPCHK(hipEventRecord(ev[0], st));       // start of the run and of the join
    bool first = true;
    for (const Launch& L : p->launches) {
        if (L.bps == 2) {
            if (L.k == 10) launch_join<2, 10>(p, L, first, st);
            else if (L.k == 20) launch_join<2, 20>(p, L, first, st);
            else if (L.k == 30) launch_join<2, 30>(p, L, first, st);
            else launch_join<2, 40>(p, L, first, st);
        } else {
            if (L.k == 10) launch_join<4, 10>(p, L, first, st);
            else if (L.k == 20) launch_join<4, 20>(p, L, first, st);
            else if (L.k == 30) launch_join<4, 30>(p, L, first, st);
            else launch_join<4, 40>(p, L, first, st);
        }
        first = false;
        HIPCHK(hipGetLastError());
    }
#if defined(VAPOR_AB) && VAPOR_AB == 3               /* (developer variant 3: the shared joins without their remap) */
    if (false) {
#else
    if (p->n_dpairs && !remap_in_clean(p)) {
#endif
        hipLaunchKernelGGL(remap_kernel, dim3((unsigned)p->n_dpairs), dim3(256), 0, st, (const DPair*)p->d_pairs, (const DShare*)p->d_shares,
                           (const int32_t*)p->d_maps, p->d_hits, p->d_nhits, p->d_overflow);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ev[1], st));
    // (the clean kernels overwrite the statistics the previous step's finish kernel reads on its own stream)
    if (before_clean) HIPCHK(hipStreamWaitEvent(st, before_clean, 0));
    if (p->n_pairs > 0) {
        const CleanGeom cg = clean_geom(p->range_words_cap, p->hcap_want, p->hcap_measured);
        const int hcap = cg.hcap;
        size_t lds = clean_lds_bytes(p->range_words_cap, hcap, cg.dual);
#ifdef VAPOR_DEV_BUILD
        if (const char* e = getenv("VAPOR_DEV_CLEAN_PAD")) lds += (size_t)atoi(e);   // experiment: fewer workgroups per CU
#endif
        const bool in_clean = remap_in_clean(p);
        launch_clean(p->range_words_cap, (unsigned)p->n_pairs, lds, st,
                     (const DPair*)p->d_pairs, (const int32_t*)p->d_clean_order, p->d_nhits,
                     p->d_hits, p->d_hflags, p->d_stats, p->range_words_cap,
                     clean_groups_lds(p->range_words_cap, hcap), hcap, p->d_overflow, p->d_big_list, skip_big ? 0 : 1, cg.dual ? 1 : 0,
                     in_clean ? (const DServe*)p->d_serve : (const DServe*)nullptr, (const int32_t*)p->d_maps, keep_flags ? 1 : 0);
        HIPCHK(hipGetLastError());
        p->flags_valid = keep_flags;
        // (clean_big_kernel needs a CU with free LDS like any other clean workgroup: behind another plan's join it sits
        // on the stream until that join is over even with nothing to do, and holds back the finish kernel and the
        // plan's next step with it - so an asynchronous step leaves it out when the plan's blocking run left it no pair)
        if (!skip_big) {
            hipLaunchKernelGGL(clean_big_kernel, dim3((unsigned)std::min<int64_t>(p->n_pairs, CLEAN_BIG_GRID)), dim3(CLEAN_THREADS),
                               clean_fixed_bytes(p->range_words_cap, true), st, p->d_pairs, p->d_nhits, p->d_hits, p->d_hflags,
                               p->d_stats, p->range_words_cap, clean_groups_cap(p->range_words_cap), p->d_overflow, p->d_big_list);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipEventRecord(ev[2], st));
    if (p->n_pairs > 0 && fetch_stats)
        HIPCHK(hipMemcpyAsync(p->h_stats, p->d_stats, sizeof(long long) * 16 * (size_t)p->n_pairs, hipMemcpyDeviceToHost, st));
    if (!fetch_stats) return VAPOR_OK;          // device-side finishing: the statistics stay in HBM
    HIPCHK(hipEventRecord(ev[3], st));
    HIPCHK(hipStreamSynchronize(st));
    float a = 0, b = 0, t = 0;
    HIPCHK(hipEventElapsedTime(&a, ev[0], ev[1]));
    HIPCHK(hipEventElapsedTime(&b, ev[1], ev[2]));
    HIPCHK(hipEventElapsedTime(&t, ev[0], ev[3]));
    p->t_join = a; p->t_clean = b; p->t_total = t;
    return VAPOR_OK;
}

extern "C" int vapor_plan_run(vapor_plan* p, int64_t* stats)
{
    if (!p || (p->n_pairs && !stats)) return fail(VAPOR_E_ARG, "vapor_plan_run: null argument");
    HIPCHK(hipSetDevice(p->ctx->device));
    if (p->ring_n > 0) {                        // asynchronous steps in flight share the workspace: let them finish
        int rc0 = async_fold(p);
        if (rc0 != VAPOR_OK) return rc0;
    }
    p->n_retried = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = plan_run_once(p);
        if (rc != VAPOR_OK) return rc;
        // pairs whose hit count exceeded their slot: enlarge to the exact count and rerun
        int64_t grow = 0;
        for (int64_t i = 0; i < p->n_pairs; ++i) {
            const long long* s = p->h_stats + 16 * i;
            if (s[15] == VAPOR_E_OVERFLOW && s[14] <= p->ctx->max_pair_cap && (uint32_t)s[14] > p->hp[i].cap) {
                p->hp[i].cap = (uint32_t)s[14];     // records the pair produced
                ++grow;
            }
        }
        if (p->n_dpairs) {
            // the shared dot plots: their record counts are the low halves of the join's packed counters
            std::vector<unsigned long long> dc((size_t)p->n_dpairs);
            HIPCHK(hipMemcpy(dc.data(), p->d_nhits + p->n_pairs, sizeof(unsigned long long) * dc.size(), hipMemcpyDeviceToHost));
            for (int64_t t = 0; t < p->n_dpairs; ++t) {
                DPair& d = p->hp[(size_t)(p->n_pairs + t)];
                const uint32_t need = (uint32_t)dc[(size_t)t];
                if (need > d.cap && (int64_t)need <= p->ctx->max_pair_cap) { d.cap = need; ++grow; }
            }
        }
        if (!grow) break;
        p->n_retried += (int)grow;
        rc = plan_alloc_hits(p);
        if (rc != VAPOR_OK) return rc;
    }
    if (p->ctx->clean_fit && p->n_pairs > 0) {
        // The clean kernel's geometry from what the pairs really hold: a plan is created with an ESTIMATE of the records a pair
        // will have (a tenth of the shorter sequence plus the chance dots), which sizes the LDS copy and with it the workgroups a
        // CU holds; the join is deterministic, so after one blocking run the largest record count is known exactly and the copy
        // is sized for that.  On the 10 kb x 20 kb shape that is eight workgroups per CU instead of seven - and 4 000 pairs are
        // 1.95 rounds of 2 048 workgroups instead of 2.2 rounds of 1 792, i.e. two rounds instead of three.
        std::vector<unsigned long long> cnt((size_t)p->n_pairs);
        HIPCHK(hipMemcpy(cnt.data(), p->d_nhits, sizeof(unsigned long long) * cnt.size(), hipMemcpyDeviceToHost));
        uint32_t most = 0;
        for (int64_t i = 0; i < p->n_pairs; ++i)
            if (p->status[(size_t)i] == 0 && (uint32_t)(cnt[(size_t)i] >> 32) <= 65535u) most = std::max(most, (uint32_t)cnt[(size_t)i]);
        if (most > 0 && most <= (uint32_t)CLEAN_HCAP_MAX) {
            p->hcap_want = (int)((most + 3u) & ~3u);
            p->hcap_measured = true;
        }
    }
    if (p->n_dpairs) {
        // A shared dot plot that needs more than max_pair_cap cannot grow: the records its targets were cut from are a
        // truncated plot, although the targets' own slots did not overflow.  Every pair it serves keeps VAPOR_E_OVERFLOW
        // (include/vapor_hip.h, "max_pair_cap") - as a host-side status, so that every later run reports it as well and
        // vapor_plan_run_loci takes the path that hands the statuses to the finish kernel.
        std::vector<unsigned long long> dc((size_t)p->n_dpairs);
        HIPCHK(hipMemcpy(dc.data(), p->d_nhits + p->n_pairs, sizeof(unsigned long long) * dc.size(), hipMemcpyDeviceToHost));
        for (int64_t t = 0; t < p->n_dpairs; ++t)
            if ((uint32_t)dc[(size_t)t] > p->hp[(size_t)(p->n_pairs + t)].cap)
                for (int32_t tg : p->shares[(size_t)t].target)
                    if (tg >= 0 && p->status[(size_t)tg] == 0) p->status[(size_t)tg] = VAPOR_E_OVERFLOW;
    }
    for (int64_t i = 0; i < p->n_pairs; ++i) {
        long long* s = p->h_stats + 16 * i;
        if (p->status[i] != 0) {
            for (int t = 0; t < 16; ++t) s[t] = 0;
            s[1] = s[2] = -1;
            s[15] = p->status[i];
        }
    }
    memcpy(p->last_stats.data(), p->h_stats, sizeof(int64_t) * 16 * (size_t)p->n_pairs);
    memcpy(stats, p->h_stats, sizeof(int64_t) * 16 * (size_t)p->n_pairs);
    p->ran = true;
    return VAPOR_OK;
}

extern "C" int vapor_plan_timings(vapor_plan* p, double* ms, int32_t n)
{
    if (!p || !ms) return fail(VAPOR_E_ARG, "vapor_plan_timings: null argument");
    const CleanGeom cg = clean_geom(p->range_words_cap, p->hcap_want, p->hcap_measured);
    double v[10] = {p->t_join, p->t_clean, p->t_total, (double)p->launches.size(), (double)p->n_retried, p->t_finish,
                    (double)p->n_served, (double)p->n_dpairs, (double)cg.per_cu, remap_in_clean(p) ? 1.0 : 0.0};
    for (int i = 0; i < n && i < 10; ++i) ms[i] = v[i];
    return VAPOR_OK;
}

extern "C" int vapor_plan_record_counts(vapor_plan* p, int64_t* records)
{
    if (!p || !records) return fail(VAPOR_E_ARG, "vapor_plan_record_counts: null argument");
    if (!p->ran) return fail(VAPOR_E_ARG, "vapor_plan_record_counts: plan has not been run");
    HIPCHK(hipSetDevice(p->ctx->device));
    std::vector<unsigned long long> cnt((size_t)std::max<int64_t>(p->n_pairs, 1));
    HIPCHK(hipMemcpy(cnt.data(), p->d_nhits, sizeof(unsigned long long) * (size_t)p->n_pairs, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < p->n_pairs; ++i) records[i] = (int64_t)(uint32_t)cnt[i];
    return VAPOR_OK;
}

extern "C" int vapor_plan_algorithmic_bytes(vapor_plan* p, int64_t* bytes, int64_t* cells)
{
    if (!p || !bytes || !cells) return fail(VAPOR_E_ARG, "null argument");
    int64_t b = 0, c = 0;
    for (int64_t i = 0; i < p->n_pairs; ++i) {
        if (p->status[i] != 0) continue;
        int64_t n1 = p->set->h[p->hp[i].seq1].len;
        int64_t n2 = std::max<int64_t>(0, (int64_t)p->set->h[p->hp[i].seq2].len - p->hp[i].off2);
        int64_t nh = p->ran ? p->last_stats[16 * i] : 0;
        b += (3 * n1 + 7) / 8 + (3 * n2 + 7) / 8 + 8 * nh + 128;
        c += n1 * n2;
    }
    *bytes = b;
    *cells = c;
    return VAPOR_OK;
}

extern "C" int vapor_plan_fetch_hits(vapor_plan* p, int64_t n_sel, const int64_t* pair_idx, int32_t* hits_ji,
                                     uint8_t* hit_flags, int64_t capacity, int64_t* hit_off)
{
    if (!p || n_sel < 0 || (n_sel && (!pair_idx || !hit_off))) return fail(VAPOR_E_ARG, "vapor_plan_fetch_hits: null argument");
    if (!p->ran) return fail(VAPOR_E_ARG, "vapor_plan_fetch_hits: plan has not been run");
    HIPCHK(hipSetDevice(p->ctx->device));
    if (hit_flags && (!p->flags_valid || p->ring_n > 0)) {
        // the last pass was a device-finished one (vapor_plan_run_loci*), which writes no per-record flags: run once more, with them
        int rc0 = vapor_plan_run(p, p->last_stats.data());
        if (rc0 != VAPOR_OK) return rc0;
    }
    std::vector<long long> off((size_t)n_sel + 1, 0), sel((size_t)std::max<int64_t>(n_sel, 1), 0);
    for (int64_t q = 0; q < n_sel; ++q) {
        int64_t i = pair_idx[q];
        if (i < 0 || i >= p->n_pairs) return fail(VAPOR_E_ARG, "pair index out of range");
        sel[q] = i;
        int64_t n = (p->last_stats[16 * i + 15] == 0) ? p->last_stats[16 * i] : 0;
        off[q + 1] = off[q] + n;
    }
    for (int64_t q = 0; q <= n_sel; ++q) hit_off[q] = off[q];
    if (off[n_sel] > capacity) return fail(VAPOR_E_OVERFLOW, "hit buffer too small");
    if (n_sel == 0 || off[n_sel] == 0) return VAPOR_OK;
    if (!hits_ji) return fail(VAPOR_E_ARG, "null hit buffer");
    // record counts of the selected pairs (the low half of the join's packed counters)
    std::vector<unsigned long long> cnt((size_t)p->n_pairs);
    HIPCHK(hipMemcpy(cnt.data(), p->d_nhits, sizeof(unsigned long long) * cnt.size(), hipMemcpyDeviceToHost));
    std::vector<long long> nrec((size_t)n_sel, 0);
    for (int64_t q = 0; q < n_sel; ++q)
        if (off[q + 1] > off[q]) nrec[q] = (long long)(uint32_t)cnt[sel[q]];
    long long *d_sel = nullptr, *d_off = nullptr, *d_nrec = nullptr;
    int32_t* d_ji = nullptr;
    uint8_t* d_fl = nullptr;
    int rc = VAPOR_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == VAPOR_OK) rc = fail(VAPOR_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    hipStream_t st = p->ctx->stream;
    vapor_ctx* ctx = p->ctx;
    chk(dmalloc(ctx, (void**)&d_sel, sizeof(long long) * sel.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_off, sizeof(long long) * off.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_nrec, sizeof(long long) * nrec.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_ji, sizeof(int32_t) * 2 * (size_t)off[n_sel]), "hipMalloc");
    if (hit_flags) chk(dmalloc(ctx, (void**)&d_fl, (size_t)off[n_sel]), "hipMalloc");
    if (rc == VAPOR_OK) {
        chk(hipMemcpyAsync(d_sel, sel.data(), sizeof(long long) * sel.size(), hipMemcpyHostToDevice, st), "copy");
        chk(hipMemcpyAsync(d_off, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice, st), "copy");
        chk(hipMemcpyAsync(d_nrec, nrec.data(), sizeof(long long) * nrec.size(), hipMemcpyHostToDevice, st), "copy");
    }
    if (rc == VAPOR_OK) {
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)n_sel), dim3(256), 0, st, p->d_pairs, d_sel, d_off, d_nrec, p->d_hits,
                           p->d_hflags, d_ji, d_fl);
        chk(hipGetLastError(), "gather launch");
        chk(hipMemcpyAsync(hits_ji, d_ji, sizeof(int32_t) * 2 * (size_t)off[n_sel], hipMemcpyDeviceToHost, st), "copy");
        if (hit_flags) chk(hipMemcpyAsync(hit_flags, d_fl, (size_t)off[n_sel], hipMemcpyDeviceToHost, st), "copy");
        chk(hipStreamSynchronize(st), "sync");
    }
    dfree(ctx, d_sel); dfree(ctx, d_off); dfree(ctx, d_nrec); dfree(ctx, d_ji); dfree(ctx, d_fl);
    return rc;
}

// ------------------------------------------------------------------------------------------
extern "C" int vapor_score_batch(vapor_ctx* ctx, vapor_seqset* set, int64_t n_pairs, const vapor_pair* pairs,
                                 int64_t* stats)
{
    vapor_plan* p = nullptr;
    int rc = vapor_plan_create(ctx, set, n_pairs, pairs, &p);
    if (rc != VAPOR_OK) return rc;
    rc = vapor_plan_run(p, stats);
    vapor_plan_destroy(p);
    return rc;
}

extern "C" int vapor_dotplot_batch(vapor_ctx* ctx, vapor_seqset* set, int64_t n_pairs, const vapor_pair* pairs,
                                   int32_t* hits_ji, int64_t hits_capacity, int64_t* hit_off, int64_t* stats)
{
    if (!hit_off) return fail(VAPOR_E_ARG, "vapor_dotplot_batch: null hit_off");
    vapor_plan* p = nullptr;
    int rc = vapor_plan_create(ctx, set, n_pairs, pairs, &p);
    if (rc != VAPOR_OK) return rc;
    std::vector<int64_t> st((size_t)std::max<int64_t>(n_pairs, 1) * 16);
    rc = vapor_plan_run(p, st.data());
    if (rc == VAPOR_OK) {
        std::vector<int64_t> idx((size_t)n_pairs);
        std::iota(idx.begin(), idx.end(), 0);
        rc = vapor_plan_fetch_hits(p, n_pairs, idx.data(), hits_ji, nullptr, hits_capacity, hit_off);
        if (stats) memcpy(stats, st.data(), sizeof(int64_t) * 16 * (size_t)n_pairs);
    }
    vapor_plan_destroy(p);
    return rc;
}

extern "C" int vapor_selfplot_qc(vapor_ctx* ctx, vapor_seqset* set, int32_t n, const int32_t* seq_idx, const int32_t* k,
                                 int64_t* out)
{
    if (n < 0 || (n && (!seq_idx || !k || !out))) return fail(VAPOR_E_ARG, "vapor_selfplot_qc: null argument");
    std::vector<vapor_pair> pr((size_t)n);
    for (int32_t t = 0; t < n; ++t) pr[t] = vapor_pair{seq_idx[t], seq_idx[t], 0, k[t], 0u};
    std::vector<int64_t> st((size_t)std::max(n, 1) * 16);
    int rc = vapor_score_batch(ctx, set, n, pr.data(), st.data());
    if (rc != VAPOR_OK) return rc;
    for (int32_t t = 0; t < n; ++t) {
        if (st[16 * t + 15] != 0) return fail((int)st[16 * t + 15], "vapor_selfplot_qc: pair failed");
        out[3 * t] = st[16 * t];
        out[3 * t + 1] = st[16 * t + 7];
        out[3 * t + 2] = st[16 * t + 8];
    }
    return VAPOR_OK;
}

// ------------------------------------------------------------------------------------------
// Cleaning and reductions on caller-supplied hit lists: what clean_dotdata_diagnal_and_anti_diagnal
// (SF:432-448), clean_dotdata_diagnal_m1b / clean_dotdata_anti_diagnal_m1b (SF:404-430) and the
// eu_dis_* reductions do when handed an explicit dot list instead of a fresh dotdata() result.
extern "C" int vapor_clean_hits(vapor_ctx* ctx, int64_t n_lists, const int32_t* hits_ji, const int64_t* off,
                                const uint32_t* flags, int64_t* stats, uint8_t* hit_flags)
{
    if (!ctx || n_lists < 0 || (n_lists && (!off || !stats))) return fail(VAPOR_E_ARG, "vapor_clean_hits: null argument");
    if (n_lists == 0) return VAPOR_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t tot = off[n_lists];
    if (tot && !hits_ji) return fail(VAPOR_E_ARG, "vapor_clean_hits: null hit list");
    std::vector<DPair> dp((size_t)n_lists);
    std::vector<unsigned long long> nh((size_t)n_lists);
    // device layout: one record per dot, every list's slot padded to a multiple of four records (the kernels
    // store flag bytes four at a time)
    std::vector<int64_t> poff((size_t)n_lists + 1, 0);
    for (int64_t t = 0; t < n_lists; ++t) poff[t + 1] = poff[t] + ((off[t + 1] - off[t] + 3) & ~(int64_t)3);
    std::vector<unsigned long long> packed((size_t)std::max<int64_t>(poff[n_lists], 4), 0ull);
    int rw = 1;
    for (int64_t t = 0; t < n_lists; ++t) {
        int mi = 0, mj = 0;
        for (int64_t h = off[t]; h < off[t + 1]; ++h) {
            int j = hits_ji[2 * h], i = hits_ji[2 * h + 1];
            if (j < 0 || i < 0 || j > VAPOR_MAX_SEQ_LEN || i > VAPOR_MAX_SEQ_LEN)
                return fail(VAPOR_E_ARG, "vapor_clean_hits: coordinate out of range");
            mi = std::max(mi, i); mj = std::max(mj, j);
            packed[poff[t] + (h - off[t])] = (unsigned long long)(((uint32_t)j << 16) | (uint32_t)i) | (1ull << 32);
        }
        DPair& d = dp[t];
        d.seq1 = (int32_t)(2 * t); d.seq2 = (int32_t)(2 * t + 1); d.off2 = 0; d.k = 10;
        d.len1 = mi + 1; d.len2 = mj + 1;
        d.flags = flags ? flags[t] : 3u;
        d.cap = (uint32_t)(off[t + 1] - off[t]);
        d.hit_off = poff[t];
        nh[t] = (unsigned long long)(off[t + 1] - off[t]) * 0x100000001ull;   // records | dots << 32
        // the largest values, i + j = 131070 and i - j + len2 = 131071, still fall into word 4095
        rw = std::min(std::max(rw, (mi + mj + 4 + 31) / 32), CLEAN_RANGE_WORDS_MAX);
    }
    DPair* d_dp = nullptr; unsigned long long* d_nh = nullptr; unsigned int* d_ov = nullptr; int32_t* d_big = nullptr;
    unsigned long long* d_hits = nullptr; uint8_t* d_fl = nullptr; long long* d_st = nullptr;
    int rc = VAPOR_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == VAPOR_OK) rc = fail(VAPOR_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    hipStream_t st = ctx->stream;
    chk(dmalloc(ctx, (void**)&d_ov, 4 * sizeof(unsigned int)), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_big, sizeof(int32_t) * dp.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_dp, sizeof(DPair) * dp.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_nh, sizeof(unsigned long long) * nh.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_hits, sizeof(unsigned long long) * packed.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_fl, packed.size()), "hipMalloc");
    chk(dmalloc(ctx, (void**)&d_st, sizeof(long long) * 16 * (size_t)n_lists), "hipMalloc");
    if (rc == VAPOR_OK) {
        chk(hipMemsetAsync(d_ov, 0, 4 * sizeof(unsigned int), st), "memset");
        chk(hipMemcpyAsync(d_dp, dp.data(), sizeof(DPair) * dp.size(), hipMemcpyHostToDevice, st), "copy");
        chk(hipMemcpyAsync(d_nh, nh.data(), sizeof(unsigned long long) * nh.size(), hipMemcpyHostToDevice, st), "copy");
        chk(hipMemcpyAsync(d_hits, packed.data(), sizeof(unsigned long long) * packed.size(), hipMemcpyHostToDevice, st), "copy");
    }
    if (rc == VAPOR_OK) {
        const CleanGeom cg = clean_geom(rw, 4096);
        const int hcap = cg.hcap;
        launch_clean(rw, (unsigned)n_lists, clean_lds_bytes(rw, hcap, cg.dual), st, (const DPair*)d_dp, (const int32_t*)nullptr,
                     d_nh, d_hits, d_fl, d_st, rw,
                     clean_groups_lds(rw, hcap), hcap, d_ov, d_big, 1, cg.dual ? 1 : 0,
                     (const DServe*)nullptr, (const int32_t*)nullptr, 1);
        chk(hipGetLastError(), "clean launch");
        hipLaunchKernelGGL(clean_big_kernel, dim3((unsigned)std::min<int64_t>(n_lists, CLEAN_BIG_GRID)), dim3(CLEAN_THREADS),
                           clean_fixed_bytes(rw, true), st, d_dp, d_nh, d_hits, d_fl, d_st, rw, clean_groups_cap(rw), d_ov, d_big);
        chk(hipGetLastError(), "clean launch");
        chk(hipMemcpyAsync(stats, d_st, sizeof(long long) * 16 * (size_t)n_lists, hipMemcpyDeviceToHost, st), "copy");
        std::vector<uint8_t> fl(hit_flags && tot ? packed.size() : 0);
        if (!fl.empty()) chk(hipMemcpyAsync(fl.data(), d_fl, fl.size(), hipMemcpyDeviceToHost, st), "copy");
        chk(hipStreamSynchronize(st), "sync");
        if (!fl.empty() && rc == VAPOR_OK)
            for (int64_t t = 0; t < n_lists; ++t) memcpy(hit_flags + off[t], fl.data() + poff[t], (size_t)(off[t + 1] - off[t]));
    }
    dfree(ctx, d_ov); dfree(ctx, d_big); dfree(ctx, d_dp); dfree(ctx, d_nh); dfree(ctx, d_hits); dfree(ctx, d_fl); dfree(ctx, d_st);
    return rc;
}

// ------------------------------------------------------------------------------------------
// per-read scores and per-locus summaries on the device
extern "C" int vapor_plan_set_reads(vapor_plan* p, int64_t n_reads, const vapor_read* reads, int64_t n_loci,
                                    const double* gt_table)
{
    if (!p || n_reads < 0 || n_loci < 0 || (n_reads && !reads) || !gt_table)
        return fail(VAPOR_E_ARG, "vapor_plan_set_reads: null argument");
    static_assert(sizeof(vapor_read) == sizeof(DRead), "vapor_read layout");
    HIPCHK(hipSetDevice(p->ctx->device));
    std::vector<int32_t> first((size_t)n_loci + 1, 0);
    int32_t prev = -1;
    for (int64_t r = 0; r < n_reads; ++r) {
        const vapor_read& x = reads[r];
        if (x.locus < prev || x.locus >= n_loci) return fail(VAPOR_E_ARG, "reads must be sorted by locus");
        const int32_t idx[4] = {x.ref_a, x.alt_a, x.kind == 0 ? x.ref_b : x.ref_a, x.kind == 0 ? x.alt_b : x.alt_a};
        for (int32_t q : idx)
            if (q < 0 || q >= p->n_pairs) return fail(VAPOR_E_ARG, "read refers to a pair outside the plan");
        if (x.kind < 0 || x.kind > 3) return fail(VAPOR_E_ARG, "unknown scorer kind");
        prev = x.locus;
        first[(size_t)x.locus + 1]++;
    }
    for (int64_t l = 0; l < n_loci; ++l) first[l + 1] += first[l];
    dfree(p->ctx, p->d_reads); dfree(p->ctx, p->d_read_scores); dfree(p->ctx, p->d_loci);
    if (p->own_gt) dfree(p->ctx, p->d_gt);
    p->d_reads = nullptr; p->d_locus_first = nullptr; p->d_gt = nullptr; p->own_gt = false; p->d_read_scores = nullptr; p->d_loci = nullptr;
    // the read table and the locus offsets travel as one block (one blocking copy out of caller memory instead of two)
    const size_t reads_bytes = (sizeof(DRead) * (size_t)std::max<int64_t>(n_reads, 1) + 15) & ~(size_t)15;
    const size_t first_bytes = sizeof(int32_t) * first.size();
    std::vector<uint8_t> blockv(reads_bytes + first_bytes, 0);
    if (n_reads) memcpy(blockv.data(), reads, sizeof(DRead) * (size_t)n_reads);
    memcpy(blockv.data() + reads_bytes, first.data(), first_bytes);
    HIPCHK(dmalloc(p->ctx, (void**)&p->d_reads, blockv.size()));
    p->d_locus_first = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(p->d_reads) + reads_bytes);
    HIPCHK(dmalloc(p->ctx, (void**)&p->d_read_scores, sizeof(double) * std::max<int64_t>(n_reads, 1)));
    HIPCHK(dmalloc(p->ctx, (void**)&p->d_loci, sizeof(double) * 8 * std::max<int64_t>(n_loci, 1)));
    HIPCHK(hipMemcpy(p->d_reads, blockv.data(), blockv.size(), hipMemcpyHostToDevice));
    // the genotype table is the same for every plan of a run: uploaded once per context, again only for a different one
    {
        vapor_ctx* c = p->ctx;
        const size_t gt_n = (size_t)2 * VAPOR_GT_TABLE_N * VAPOR_GT_TABLE_N;
        if (!c->d_gt) {
            HIPCHK(hipMalloc((void**)&c->d_gt, sizeof(double) * gt_n));
            HIPCHK(hipMemcpy(c->d_gt, gt_table, sizeof(double) * gt_n, hipMemcpyHostToDevice));
            c->h_gt.assign(gt_table, gt_table + gt_n);
        }
        if (memcmp(c->h_gt.data(), gt_table, sizeof(double) * gt_n) == 0) {
            p->d_gt = c->d_gt;
        } else {
            HIPCHK(dmalloc(c, (void**)&p->d_gt, sizeof(double) * gt_n));
            p->own_gt = true;
            HIPCHK(hipMemcpy(p->d_gt, gt_table, sizeof(double) * gt_n, hipMemcpyHostToDevice));
        }
    }
    p->n_reads = n_reads;
    p->n_loci = n_loci;
    return VAPOR_OK;
}

// join -> clean -> finish on the device; the per-pair statistics stay in HBM.  d_loci_out (device
// pointer, n_loci * 8 doubles, may be NULL) receives a copy on the library's stream before it is
// synchronised; loci_out / read_scores (host, may be NULL) receive host copies.
extern "C" int vapor_plan_run_loci(vapor_plan* p, void* d_loci_out, double* loci_out, double* read_scores)
{
    if (!p) return fail(VAPOR_E_ARG, "vapor_plan_run_loci: null plan");
    if (!p->d_reads) return fail(VAPOR_E_ARG, "vapor_plan_run_loci: call vapor_plan_set_reads first");
    HIPCHK(hipSetDevice(p->ctx->device));
    if (p->ring_n > 0) {
        int rc0 = async_fold(p);
        if (rc0 != VAPOR_OK) return rc0;
    }
    hipStream_t st = p->ctx->stream;
    bool host_status = false;
    for (int64_t i = 0; i < p->n_pairs; ++i)
        if (p->status[i] != 0) { host_status = true; break; }
    int rc;
    const bool light = !(host_status || !p->ran);
    if (!light) {
        // first run, or pairs the host marked as failed: full path once (statistics to the host, slots grown)
        rc = vapor_plan_run(p, p->last_stats.data());
        if (rc != VAPOR_OK) return rc;
        // (the run itself may have given pairs a host-side status: the targets of a shared dot plot beyond max_pair_cap)
        for (int64_t i = 0; i < p->n_pairs && !host_status; ++i)
            if (p->status[i] != 0) host_status = true;
        if (host_status)
            HIPCHK(hipMemcpyAsync(p->d_stats, p->h_stats, sizeof(long long) * 16 * (size_t)p->n_pairs, hipMemcpyHostToDevice, st));
    } else {
        rc = plan_run_once(p, false, nullptr, nullptr, false, nullptr, false);
        if (rc != VAPOR_OK) return rc;
    }
    hipEvent_t e1 = p->ev_f[1];                  // the finish kernel starts where the clean kernels end (ev[2])
    double* d_out = d_loci_out ? static_cast<double*>(d_loci_out) : p->d_loci;
    if (p->n_loci > 0) {
        hipLaunchKernelGGL(finish_kernel, dim3((unsigned)p->n_loci), dim3(64), 0, st, p->d_reads, p->d_locus_first, p->d_stats,
                           p->d_gt, p->d_read_scores, d_out, (double*)nullptr);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(e1, st));
    // only the overflow count has to come back (after the finish kernel, so that nothing sits between the kernels)
    HIPCHK(hipMemcpyAsync(p->h_overflow, p->d_overflow, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    if (loci_out && p->n_loci)
        HIPCHK(hipMemcpyAsync(loci_out, d_out, sizeof(double) * 8 * (size_t)p->n_loci, hipMemcpyDeviceToHost, st));
    if (read_scores && p->n_reads)
        HIPCHK(hipMemcpyAsync(read_scores, p->d_read_scores, sizeof(double) * (size_t)p->n_reads, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (light && *p->h_overflow != 0 && !p->overflow_final) {
        // some pair outgrew its record slot on this run: redo through the full path, which resizes.  Pairs that
        // still overflow after that (their slot would exceed max_pair_cap) keep VAPOR_E_OVERFLOW in their
        // statistics and are scored as reads without a usable plot; nothing is retried for them again.
        rc = vapor_plan_run(p, p->last_stats.data());
        if (rc != VAPOR_OK) return rc;
        for (int64_t i = 0; i < p->n_pairs; ++i)
            if (p->last_stats[16 * (size_t)i + 15] == VAPOR_E_OVERFLOW) { p->overflow_final = true; break; }
        return vapor_plan_run_loci(p, d_loci_out, loci_out, read_scores);
    }
    p->big_known = true;
    p->n_big = p->h_overflow[1];
    float f = 0, a = 0, b = 0, t = 0;
    HIPCHK(hipEventElapsedTime(&f, p->ev[2], e1));
    p->t_finish = f;
    if (hipEventElapsedTime(&a, p->ev[0], p->ev[1]) == hipSuccess && hipEventElapsedTime(&b, p->ev[1], p->ev[2]) == hipSuccess &&
        hipEventElapsedTime(&t, p->ev[0], e1) == hipSuccess) {
        p->t_join = a; p->t_clean = b; p->t_total = t;
    }
    return VAPOR_OK;
}


// ------------------------------------------------------------------------------------------
// Host helper of the read extraction: cigar2alignstart_by_pos, SF:309-337.  Walks the CIGAR until the
// reference cursor passes start-1; out[0] = offset into the read, out[1] = miss_bp.  S/I/M/= advance the read,
// M/=/D the reference, N/H/P/X nothing (as in the reference).  VAPOR_E_ARG when the CIGAR holds no operation
// (the reference raises IndexError there).
extern "C" int vapor_cigar2alignstart(const char* cigar, int64_t align_start, int64_t start, int64_t* out)
{
    if (!cigar || !out) return fail(VAPOR_E_ARG, "vapor_cigar2alignstart: null argument");
    int64_t q = 0, r = align_start, n = 0;
    bool have_n = false;
    char last = 0;
    for (const char* c = cigar; *c; ++c) {
        const char ch = *c;
        if (ch >= '0' && ch <= '9') { n = n * 10 + (ch - '0'); have_n = true; continue; }
        const bool op = ch == 'M' || ch == 'I' || ch == 'D' || ch == 'N' || ch == 'S' || ch == 'H' || ch == 'P' || ch == '=' || ch == 'X';
        if (op && have_n) {
            if (ch == 'S' || ch == 'I') q += n;
            else if (ch == 'M' || ch == '=') { q += n; r += n; }
            else if (ch == 'D') r += n;
            last = ch;
            if (r > start - 1) break;
        }
        n = 0; have_n = false;             // any other character ends the number, as the regular expression would
    }
    if (!last) return fail(VAPOR_E_ARG, "vapor_cigar2alignstart: no CIGAR operation");
    const int64_t over = r - start;
    if (last == 'M' || last == '=') { out[0] = q - over; out[1] = 0; }
    else { out[0] = q; out[1] = over; }
    return VAPOR_OK;
}

// The same walk over a BAM record's binary CIGAR (uint32 per operation: length << 4 | code, codes "MIDNSHP=X"), so that
// the in-process BAM reader neither formats nor re-parses CIGAR text (thousands of operations per long read).
extern "C" int vapor_cigar2alignstart_ops(const uint32_t* ops, int64_t n_ops, int64_t align_start, int64_t start, int64_t* out)
{
    if (!out || (n_ops > 0 && !ops)) return fail(VAPOR_E_ARG, "vapor_cigar2alignstart_ops: null argument");
    if (n_ops <= 0) return fail(VAPOR_E_ARG, "vapor_cigar2alignstart_ops: no CIGAR operation");
    int64_t q = 0, r = align_start;
    uint32_t last = 0;
    for (int64_t t = 0; t < n_ops; ++t) {
        const int64_t n = ops[t] >> 4;
        last = ops[t] & 15u;
        if (last == 4u || last == 1u) q += n;                    // S, I
        else if (last == 0u || last == 7u) { q += n; r += n; }   // M, =
        else if (last == 2u) r += n;                             // D
        if (r > start - 1) break;
    }
    const int64_t over = r - start;
    if (last == 0u || last == 7u) { out[0] = q - over; out[1] = 0; }
    else { out[0] = q; out[1] = over; }
    return VAPOR_OK;
}

// ------------------------------------------------------------------------------------------
// The same run without a host round trip per step: enqueue only, vapor_plan_sync() waits and reports.
constexpr int ASYNC_RING = 64;

extern "C" int vapor_set_stream(vapor_ctx* c, void* hip_stream)
{
    if (!c) return fail(VAPOR_E_ARG, "vapor_set_stream: null context");
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    c->user_stream = hip_stream != nullptr;
    return VAPOR_OK;
}

// the stream a plan's asynchronous steps go to: the caller's if one is set, else the plan's lane (dealt out on first use)
static int plan_lane(vapor_plan* p, hipStream_t* out)
{
    vapor_ctx* c = p->ctx;
    if (c->user_stream) { *out = c->stream; return VAPOR_OK; }
    if (!p->lane) {
        const unsigned l = c->lane_rr++ & 1u;
        if (!c->lane[l]) HIPCHK(hipStreamCreateWithFlags(&c->lane[l], hipStreamNonBlocking));
        if (!c->fin[l]) {
            int lo = 0, hi = 0;                  // (numerically lowest = highest priority)
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
                hipStreamCreateWithPriority(&c->fin[l], hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                c->fin[l] = nullptr;
                HIPCHK(hipStreamCreateWithFlags(&c->fin[l], hipStreamNonBlocking));   // no priorities here: an ordinary stream
            }
        }
        p->lane = c->lane[l];
        p->fin = c->fin[l];
    }
    *out = p->lane;
    return VAPOR_OK;
}

// waits for the steps in flight and adds their event times to the accumulators
static int async_fold(vapor_plan* p)
{
    hipStream_t st = nullptr;
    int rc = plan_lane(p, &st);
    if (rc != VAPOR_OK) return rc;
    // (the plan's own streams as well when the caller has switched streams with steps in flight)
    if (p->lane && p->lane != st) HIPCHK(hipStreamSynchronize(p->lane));
    HIPCHK(hipStreamSynchronize(st));
    if (p->fin && p->fin != st) HIPCHK(hipStreamSynchronize(p->fin));
    for (int i = 0; i < p->ring_n; ++i) {
        float x = 0;
        hipEvent_t* ev = p->ring[(size_t)i].data();
        HIPCHK(hipEventElapsedTime(&x, ev[0], ev[1])); p->acc_ms[0] += x;
        HIPCHK(hipEventElapsedTime(&x, ev[1], ev[2])); p->acc_ms[1] += x;
        HIPCHK(hipEventElapsedTime(&x, ev[2], ev[3])); p->acc_ms[2] += x;
        HIPCHK(hipEventElapsedTime(&x, ev[0], ev[3])); p->acc_ms[3] += x;
    }
    p->acc_n += p->ring_n;
    p->ring_n = 0;
    return VAPOR_OK;
}

extern "C" int vapor_plan_run_loci_async(vapor_plan* p, void* d_loci_out)
{
    if (!p) return fail(VAPOR_E_ARG, "vapor_plan_run_loci_async: null plan");
    if (!p->d_reads) return fail(VAPOR_E_ARG, "vapor_plan_run_loci_async: call vapor_plan_set_reads first");
    if (!p->ran) return fail(VAPOR_E_ARG, "vapor_plan_run_loci_async: run the plan once with vapor_plan_run_loci first (it sizes the slots)");
    for (int64_t i = 0; i < p->n_pairs; ++i)
        if (p->status[i] != 0) return fail(VAPOR_E_ARG, "vapor_plan_run_loci_async: the plan holds pairs the host rejected; use vapor_plan_run_loci");
    HIPCHK(hipSetDevice(p->ctx->device));
    if (p->ring_n >= ASYNC_RING) {              // every event set is in use: wait for those steps, keep their times
        int rc0 = async_fold(p);
        if (rc0 != VAPOR_OK) return rc0;
    }
    hipStream_t st = nullptr;
    int rc = plan_lane(p, &st);
    if (rc != VAPOR_OK) return rc;
    if (p->ring.empty()) {
        p->ring.resize(ASYNC_RING);
        for (auto& r : p->ring)
            for (auto& e : r) { e = nullptr; HIPCHK(hipEventCreate(&e)); }
        HIPCHK(hmalloc(p->ctx, (void**)&p->h_loci, sizeof(double) * 8 * (size_t)std::max<int64_t>(p->n_loci, 1)));
        HIPCHK(hipEventCreateWithFlags(&p->ev_last, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p->ev_clean, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p->ev_fin, hipEventDisableTiming));
    }
    // The finish kernel goes to a stream of its own (highest priority): as a kernel it needs a CU with 56 free registers
    // per SIMD, which another plan's join (4 waves x 120) does not leave, so behind the clean kernels on the plan's
    // stream it held back the plan's next join until that other join was over (7 % of cfg2's rate).  On its own stream
    // it takes the first CU a join workgroup leaves; the next step's clean kernels wait for it, its join does not.
    hipStream_t fs = (p->fin && !p->ctx->user_stream) ? p->fin : st;
    if (p->have_after) {                        // vapor_plan_after: what the caller enqueued elsewhere comes first
        HIPCHK(hipStreamWaitEvent(fs, p->ev_after, 0));   // (it is the finish kernel that overwrites the records)
        p->have_after = false;
    }
    // the sticky overflow counter reports on the asynchronous steps since the last vapor_plan_sync: what a blocking
    // run counted before it resized the slots is not theirs
    if (p->ring_n == 0 && p->acc_n == 0) HIPCHK(hipMemsetAsync(p->d_overflow + 2, 0, sizeof(unsigned int), st));
    hipEvent_t* ev = p->ring[(size_t)p->ring_n].data();
    rc = plan_run_once(p, false, ev, st, p->big_known && p->n_big == 0, (fs != st && p->have_fin) ? p->ev_fin : nullptr, false);
    if (rc != VAPOR_OK) return rc;
    double* d_out = d_loci_out ? static_cast<double*>(d_loci_out) : p->d_loci;
    if (fs != st) {
        HIPCHK(hipEventRecord(p->ev_clean, st));
        HIPCHK(hipStreamWaitEvent(fs, p->ev_clean, 0));
    }
    if (p->n_loci > 0) {
        // (the finish kernel writes the pinned host copy itself: no copy kernel behind it)
        hipLaunchKernelGGL(finish_kernel, dim3((unsigned)p->n_loci), dim3(64), 0, fs, p->d_reads, p->d_locus_first, p->d_stats,
                           p->d_gt, p->d_read_scores, d_out, p->h_loci);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ev[3], fs));
    HIPCHK(hipEventRecord(p->ev_last, fs));
    if (fs != st) {
        HIPCHK(hipEventRecord(p->ev_fin, fs));
        p->have_fin = true;
    }
    p->have_last = true;
    ++p->ring_n;
    return VAPOR_OK;
}

// Ordering against a stream of the caller's without a host round trip (e.g. a framework's collective on its own
// stream): vapor_plan_then makes `hip_stream` wait for the plan's most recently enqueued step (so the caller can read
// d_loci_out there), vapor_plan_after makes the plan's NEXT step wait for everything enqueued on `hip_stream` so far
// (so that step does not overwrite d_loci_out under the caller's feet).
extern "C" int vapor_plan_then(vapor_plan* p, void* hip_stream)
{
    if (!p) return fail(VAPOR_E_ARG, "vapor_plan_then: null plan");
    if (!p->have_last) return VAPOR_OK;
    HIPCHK(hipSetDevice(p->ctx->device));
    HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), p->ev_last, 0));
    return VAPOR_OK;
}

extern "C" int vapor_plan_after(vapor_plan* p, void* hip_stream)
{
    if (!p) return fail(VAPOR_E_ARG, "vapor_plan_after: null plan");
    HIPCHK(hipSetDevice(p->ctx->device));
    if (!p->ev_after) HIPCHK(hipEventCreateWithFlags(&p->ev_after, hipEventDisableTiming));
    HIPCHK(hipEventRecord(p->ev_after, static_cast<hipStream_t>(hip_stream)));
    p->have_after = true;
    return VAPOR_OK;
}

// waits for the steps in flight; timings() then reports their averages; loci_out (may be NULL) receives the
// records of the last step.  VAPOR_E_OVERFLOW if a pair outgrew its slot in one of them (run vapor_plan_run_loci).
extern "C" int vapor_plan_sync(vapor_plan* p, double* loci_out)
{
    if (!p) return fail(VAPOR_E_ARG, "vapor_plan_sync: null plan");
    HIPCHK(hipSetDevice(p->ctx->device));
    const int had = p->ring_n;
    int rc0 = async_fold(p);
    if (rc0 != VAPOR_OK) return rc0;
    if (p->acc_n > 0) {
        p->t_join = (float)(p->acc_ms[0] / p->acc_n); p->t_clean = (float)(p->acc_ms[1] / p->acc_n);
        p->t_finish = (float)(p->acc_ms[2] / p->acc_n); p->t_total = (float)(p->acc_ms[3] / p->acc_n);
        if (loci_out && p->n_loci && (had > 0 || p->h_loci)) memcpy(loci_out, p->h_loci, sizeof(double) * 8 * (size_t)p->n_loci);
    }
    p->acc_ms[0] = p->acc_ms[1] = p->acc_ms[2] = p->acc_ms[3] = 0;
    p->acc_n = 0;
    unsigned int sticky = 0;
    HIPCHK(hipMemcpy(&sticky, p->d_overflow + 2, sizeof(unsigned int), hipMemcpyDeviceToHost));
    if (sticky) {
        HIPCHK(hipMemset(p->d_overflow + 2, 0, sizeof(unsigned int)));
        return fail(VAPOR_E_OVERFLOW, "a pair outgrew its record slot during the asynchronous steps; run vapor_plan_run_loci (it resizes)" +
                    std::string(p->big_known ? "" : " [no blocking run has counted the pairs left to clean_big_kernel]") +
                    " [pairs left to clean_big_kernel on the last blocking run: " + std::to_string(p->n_big) + "]");
    }
    return VAPOR_OK;
}

#if defined(VAPOR_PHASE_TIMING) || defined(VAPOR_BLOCK_TIMING)
extern "C" int vapor_debug_block_ticks(double* out, int32_t n)
{
    std::vector<unsigned long long> h(4096);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(vapor::g_block_ticks), sizeof(unsigned long long) * 4096) != hipSuccess) return -1;
    for (int x = 0; x < n && x < 4096; ++x) out[x] = (double)h[(size_t)x];
    return 0;
}
#endif
#if defined(VAPOR_PHASE_TIMING) || defined(VAPOR_BLOCK_TIMING)
extern "C" int vapor_debug_block_info(double* info, double* phase, int32_t n)
{
    std::vector<unsigned long long> h(4096 * 4), ph(4096 * 8), z(4096 * 8, 0ULL);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(vapor::g_block_info), sizeof(unsigned long long) * 4096 * 4) != hipSuccess) return -1;
    for (int x = 0; x < n * 4 && x < 4096 * 4; ++x) info[x] = (double)h[(size_t)x];
#ifdef VAPOR_PHASE_TIMING
    if (hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(vapor::g_block_phase), sizeof(unsigned long long) * 4096 * 8) != hipSuccess) return -1;
    for (int x = 0; x < n * 8 && x < 4096 * 8; ++x) phase[x] = (double)ph[(size_t)x];
    if (hipMemcpyToSymbol(HIP_SYMBOL(vapor::g_block_phase), z.data(), sizeof(unsigned long long) * 4096 * 8) != hipSuccess) return -1;
#endif
    return 0;
}
// developer build only: the join tasks of a plan (first index into the sorted pair list, pairs in the task) and
// the sorted pair list itself
extern "C" int vapor_debug_plan_tasks(vapor_plan* p, int32_t* first, int32_t* n_reads, int32_t cap, int32_t* order, int32_t order_cap)
{
    if (!p) return -1;
    const int n = (int)p->tasks.size();
    for (int x = 0; x < n && x < cap; ++x) { first[x] = p->tasks[(size_t)x].first; n_reads[x] = p->tasks[(size_t)x].n_reads; }
    for (int x = 0; x < (int)p->task_pairs.size() && x < order_cap; ++x) order[x] = p->task_pairs[(size_t)x];
    return n;
}
#endif
#ifdef VAPOR_PHASE_TIMING
// developer build only: read (and clear) the per-phase tick sums
extern "C" int vapor_debug_phases(double* out, int32_t n)
{
    unsigned long long h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(vapor::g_phase), sizeof(h)) != hipSuccess) return -1;
    for (int x = 0; x < n && x < 64; ++x) out[x] = (double)h[x];
    memset(h, 0, sizeof(h));
    if (hipMemcpyToSymbol(HIP_SYMBOL(vapor::g_phase), h, sizeof(h)) != hipSuccess) return -1;
    return 0;
}
#endif
