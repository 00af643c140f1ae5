/*
 * vapor_hip.h - C ABI of libvapor_hip.so: VaPoR's recurrence-plot scoring path on MI355X.
 *
 * The reference has no FFI layer: its hot path is a set of module-level Python functions in
 * vapor_vali/Simple_function.pyx ("SF") that the `vapor` script star-imports
 * (vapor_vali/vapor:303,322,374,470).  Every entry point below names the reference routine(s)
 * it stands in for; the Python host code in vapor_amd/ keeps the reference's function names
 * and calls through this ABI with ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions: plain pointers and sizes; all buffers caller-allocated and never retained;
 * every function returns 0 or a negative VAPOR_E_* code (vapor_abi_version, vapor_build_flags, vapor_source_id, the two *_last_error
 * calls and vapor_crc32 return what their names say); vapor_last_error() gives a thread-local message (vapor_bam_last_error()
 * for the host helpers of the read extraction and the output table).  One host thread per context.  No exception crosses the
 * boundary.
 */
#ifndef VAPOR_HIP_H
#define VAPOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAPOR_ABI_VERSION 3
/* A developer build of the library (-DVAPOR_DEV_BUILD: timing stamps, A/B variants, non-default tuning constants)
 * reports VAPOR_ABI_VERSION + VAPOR_ABI_DEV_OFFSET and lists what it carries in vapor_build_flags(); the product
 * build reports VAPOR_ABI_VERSION and "".  A loader that accepts only VAPOR_ABI_VERSION can never run an
 * experimental library by accident. */
#define VAPOR_ABI_DEV_OFFSET 1000000

/* status codes */
#define VAPOR_OK 0
#define VAPOR_E_HIP (-1)       /* a HIP runtime call failed (message in vapor_last_error) */
#define VAPOR_E_OVERFLOW (-2)  /* caller-supplied output buffer too small; required size reported */
#define VAPOR_E_KEYERROR (-3)  /* a k-mer of seq1 holds a character invert_base lacks (SF:19-20, SF:1421) */
#define VAPOR_E_ARG (-4)       /* bad argument: index out of range, unsupported k, sequence too long */
#define VAPOR_E_NOMEM (-5)

#define VAPOR_MAX_SEQ_LEN 65535 /* positions are packed 16+16 bit on the device */

typedef struct vapor_ctx vapor_ctx;
typedef struct vapor_seqset vapor_seqset;
typedef struct vapor_plan vapor_plan;

/* per-sequence upload flags */
#define VAPOR_SEQ_UPPER 1u /* apply str.upper() before packing (SF:183-184: abs_dis_m1b upper-cases ref and alt) */

/*
 * One dot plot: dotdata(k, seq1, seq2[off2:]) (SF:545-549 -> kmerhits SF:951-983), i.e. the
 * reference's call convention dotdata(window_size, read, allele[miss_bp:]) (SF:185-186,
 * 242-243, 278-279).  seq1/seq2 index a vapor_seqset.  k must be 10, 20, 30 or 40, the only
 * values window_size_refine can return (SF:2031-2041).
 */
typedef struct vapor_pair {
    int32_t seq1;  /* read (the sequence whose k-mers are also searched reverse-complemented) */
    int32_t seq2;  /* allele window */
    int32_t off2;  /* miss_bp: seq2 is used from this offset on */
    int32_t k;
    uint32_t flags; /* VAPOR_PF_* */
} vapor_pair;

#define VAPOR_PF_C1 1u   /* clean_dotdata_diagnal_and_anti_diagnal (SF:432-448) -> ST_C1_* */
#define VAPOR_PF_C2 2u   /* the two-step cleaning of within_10Perc_m1b (SF:281-288) -> ST_C2_* */
#define VAPOR_PF_DIR 4u  /* with PF_C1: dis_to_diagnal_most_abundant_defined (SF:582-591) and
                            eu_dis_dir_calcu (SF:718-722) over the C1-kept dots -> ST_DIR_* */

/* int64 statistics record per pair (VAPOR_STATS_STRIDE words) */
#define VAPOR_STATS_STRIDE 16
#define VAPOR_ST_N_HITS 0      /* len(dotdata) */
#define VAPOR_ST_FIRST_J 1     /* dotdata[0][0]  = smallest j, -1 if no hits (SF:187) */
#define VAPOR_ST_LAST_J 2      /* dotdata[-1][0] = largest j,  -1 if no hits */
#define VAPOR_ST_C1_KEPT 3     /* dots surviving C1 */
#define VAPOR_ST_C1_SUM_ABS 4  /* sum |j-i| over them: eu_dis_abs_calcu = this / C1_KEPT (SF:705-708) */
#define VAPOR_ST_C2_KEPT 5     /* dots surviving the diagonal step or the anti-diagonal step on the rest */
#define VAPOR_ST_C2_COUNT10 6  /* eu_dis_dots_within_10perc over them (SF:730-733) */
#define VAPOR_ST_N_DIAG 7      /* dots with j == i (qual_check_repetitive_region SF:1158-1160) */
#define VAPOR_ST_N_LOWER 8     /* dots with j > i  (SF:1162-1164) */
#define VAPOR_ST_C2_KEPT_DIAG 9 /* dots kept by the diagonal step alone */
#define VAPOR_ST_DIR_C2X 10     /* 2*c, c = the redefined diagonal intercept (a multiple of 0.5; 0 when not unique) */
#define VAPOR_ST_DIR_N 11       /* kept dots (x,y) = (j+c, i) with abs(x-y)/abs(x) > 0.1 (eu_dis_single_dot SF:710-716) */
#define VAPOR_ST_DIR_SUM2 12    /* 2 * sum(x - y) over them: eu_dis_dir_calcu = SUM2 / 2 / N, or 0.0001 when N == 0 */
#define VAPOR_ST_DIR_LISTS 13   /* number of longest sub-bins found (c is a median only when this is 1) */
#define VAPOR_ST_RECORDS_NEEDED 14 /* 0; on VAPOR_E_OVERFLOW the run records the pair produced (its slot was smaller) */
#define VAPOR_ST_STATUS 15     /* 0, or VAPOR_E_KEYERROR / VAPOR_E_ARG / VAPOR_E_OVERFLOW for this pair */

/* per-hit flag bits returned by vapor_plan_fetch_hits */
#define VAPOR_HF_C1_KEPT 1u
#define VAPOR_HF_C2_DIAG 2u
#define VAPOR_HF_C2_ANTI 4u

/* ---- context --------------------------------------------------------------------------- */
int vapor_abi_version(void);
const char* vapor_build_flags(void);
/* What the binary was built from: "<k>:<a>", k = sha256 (16 hex digits) over the kernel sources and the build recipe
 * (vapor_kernels.h, vapor_hip.hip, build.py: the id profiles/ cites), a = the same over every source file of the library.  The
 * loader compares it with the sources beside it and refuses a library that was built from other ones; "cpu-twin" for the
 * CPU twin, "unknown" for a build outside vapor_amd/build.py. */
const char* vapor_source_id(void);
const char* vapor_last_error(void);
int vapor_init(int device_ordinal, vapor_ctx** ctx);
int vapor_destroy(vapor_ctx* ctx);
/* tuning knobs: "reads_per_task" (at most this many pairs per join workgroup, <= 64), "join_tasks" (number of
 * join workgroups a launch is cut into; default = number of CUs), "max_pair_cap" (largest record slot a pair may
 * grow to on an overflow rerun; beyond it the pair keeps VAPOR_E_OVERFLOW), "shared_join" (1, the default: a read scored
 * against a window and alleles derived from it - vapor_seqset_create_derived - is joined once for all of them; 0: one join per
 * pair), "remap_in_clean" (who cuts a served pair's records out of a shared dot plot.  1, the default: the workgroup that cleans
 * the pair when the plan's clean workgroups run in at most four rounds (a resident batch of a few thousand pairs), else a kernel
 * of its own before the cleaning, remap_kernel; 0: always that kernel; 2: always the clean workgroups), "clean_order" (1, the default: the clean kernel's workgroups - one per
 * pair - are dealt out longest pair first, so that the last round of a plan of a few rounds is made of its shortest pairs; 0: in
 * pair order), "clean_fit" (1, the default: after a plan's first blocking run the clean workgroups' LDS copy is sized for the
 * largest record count the pairs really have instead of the estimate made at vapor_plan_create - more workgroups per CU; 0: the
 * estimate stays), "stage_threads" (host threads that copy a
 * sequence set's bytes into the pinned staging buffer, default 3), "bam_cu_share" (0 .. 8: vapor_bam_chop_device's copies and kernels
 * go to a stream masked to that many eighths of the CUs - 0 and 8: the context's own stream, all CUs; a caller that scores several
 * batches at once on several contexts leaves CUs to the other contexts' kernels this way).  Results do not depend on any of them. */
int vapor_set_param(vapor_ctx* ctx, const char* name, int64_t value);

/* ---- sequences: ASCII in, packed bit planes resident in HBM ------------------------------ */
/*
 * Uploads n_seqs sequences (blob[off[s] .. off[s]+len[s])), folds IUPAC codes as key_modify
 * does (SF:908-949), optionally upper-cases, and packs them on the device into a 2-bit base
 * plane, a 1-bit "not A/C/G/T" plane and a 4-bit symbol plane.
 * seq_info[2*s] receives the number of symbols outside upper-case ACGT, seq_info[2*s+1] the
 * number outside invert_base's alphabet (such a seq1 raises KeyError in the reference).
 */
int vapor_seqset_create(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* blob, const int64_t* off,
                        const int32_t* len, const uint8_t* flags, int32_t* seq_info,
                        vapor_seqset** set);
/* the same from one pointer per sequence (no concatenated copy on the caller's side): seq[s] points at len[s] bytes */
int vapor_seqset_create_ptrs(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* const* seq, const int32_t* len,
                             const uint8_t* flags, int32_t* seq_info, vapor_seqset** out);
/*
 * Derived sequences: alleles the reference builds by string surgery on a window it has already read - a deletion's
 * ref_seq[:f] + ref_seq[-f:] (SF:1712), a tandem duplication's ref[:f] + mid + mid + ref[-f:] (SF:1755), an inversion's
 * ref[:f] + reverse(complementary(mid)) + ref[-f:] (SF:1907), an insertion's flank + ins_seq + flank (SF:1872), the block
 * structures of the complex types (SF:1557-1665), and the str.upper() twins of abs_dis_m1b (SF:183-184) - described instead of
 * uploaded: derived sequence d (index n_seqs + d of the set) is the concatenation of its segments
 * segs[seg_first[d] .. seg_first[d+1]), each a slice parent[off : off + len] of one of the n_seqs sequences given as bytes,
 * reverse-complemented when VAPOR_SEG_REVCOMP is set, the whole upper-cased when derived_flags[d] has VAPOR_SEQ_UPPER.  The
 * device assembles the bit planes from the parents' (derive_kernel); no byte of a derived sequence crosses the link.
 * complementary() DROPS every character outside ATGCN / atgcn (SF:471-478), which a descriptor cannot express: a
 * reverse-complemented segment whose parent holds such a character (an IUPAC code, an X) is refused with VAPOR_E_ARG and the
 * caller uploads that allele as bytes.  A derived sequence without VAPOR_SEQ_UPPER over a parent that was uploaded WITH it is
 * refused as well (the parent's planes hold the upper-cased text, not its bytes).  seq_info has 2 * (n_seqs + n_derived) entries.
 * A plan over such a set joins every read ONCE against a reference window and the alleles derived from it: the k-mers of a
 * derived allele are those of its parent's slices plus the few that span a junction or lie in inserted bytes, so one table
 * (the parent followed by those stretches) and one probe per read give the dots of all of them (remap_kernel).
 */
typedef struct vapor_segment {
    int32_t parent; /* 0 .. n_seqs-1 */
    int32_t off, len;
    uint32_t flags; /* VAPOR_SEG_* */
} vapor_segment;
#define VAPOR_SEG_REVCOMP 1u
#define VAPOR_MAX_SEGMENTS 16 /* per derived sequence */
int vapor_seqset_create_derived(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* const* seq, const int32_t* len,
                                const uint8_t* flags, int32_t n_derived, const int32_t* seg_first, const vapor_segment* segs,
                                const uint8_t* derived_flags, int32_t* seq_info, vapor_seqset** out);
/* The packed planes of one sequence as they lie in HBM (for tests and callers that want to check a derived sequence against
 * the same text uploaded as bytes): ceil(len / 32) chunks, per chunk 2 words of the 2-bit plane into p2, 1 word of the
 * "not upper-case ACGT" plane into e1, 4 words of the 4-bit symbol plane into x4 (any of the three may be NULL). */
int vapor_seqset_planes(vapor_seqset* set, int32_t seq, uint32_t* p2, uint32_t* e1, uint32_t* x4);
int vapor_seqset_destroy(vapor_seqset* set);

/* ---- plans: a batch of dot plots resident on the device ---------------------------------- */
/* Validates and groups the pairs (reads sharing an allele window share its hash table),
 * uploads the descriptors and sizes the workspace.  The plan can be run any number of times. */
int vapor_plan_create(vapor_ctx* ctx, vapor_seqset* set, int64_t n_pairs, const vapor_pair* pairs,
                      vapor_plan** plan);
int vapor_plan_destroy(vapor_plan* plan);
/*
 * The hot path: k-mer hash join (kmerhits SF:951-983), gap clustering and noise cleaning
 * (dis_cluster SF:551-564, dis_cluster_2 SF:566-580, SF:404-448) and the integer reductions
 * (SF:705-733, SF:1154-1171) for every pair.  stats: n_pairs * VAPOR_STATS_STRIDE int64.
 */
int vapor_plan_run(vapor_plan* plan, int64_t* stats);
/* device time of the kernels of the last vapor_plan_run, measured with HIP events on the
 * library's stream: ms[0] = join kernels (and remap_kernel where the plan runs it), ms[1] = clean kernels, ms[2] = whole run incl.
 * copies, ms[3] = number of join launches, ms[4] = number of retried pairs, ms[5] = finish kernel, ms[6] = pairs served by a shared
 * join, ms[7] = joins that serve them, ms[8] = clean workgroups per CU (what the staged records leave room for), ms[9] = 1 when
 * the clean workgroups cut the served pairs' records out of the shared dot plots themselves ("remap_in_clean") */
int vapor_plan_timings(vapor_plan* plan, double* ms, int32_t n);
/* run records the join of the last run wrote per pair (n_pairs int64): the device keeps runs of consecutive
 * dots (j+t, i+t) / (j-t, i+t) as one record; stats[0] stays the number of dots */
int vapor_plan_record_counts(vapor_plan* plan, int64_t* records);
/* algorithmic bytes of one run: sum over pairs of packed read + packed allele
 * + 8 B per hit + 128 B statistics record, using the hit counts of the last run */
int vapor_plan_algorithmic_bytes(vapor_plan* plan, int64_t* bytes, int64_t* cells);
/*
 * Hits of selected pairs after vapor_plan_run: hits_ji receives (j, i) int32 pairs in
 * unspecified order inside a pair (dotdata's list is these sorted by (j, i)), hit_flags one
 * VAPOR_HF_* byte per hit (may be NULL), hit_off[n_sel+1] the offsets.  VAPOR_E_OVERFLOW
 * with hit_off[n_sel] = required capacity if `capacity` (in hits) is too small.
 */
int vapor_plan_fetch_hits(vapor_plan* plan, int64_t n_sel, const int64_t* pair_idx,
                          int32_t* hits_ji, uint8_t* hit_flags, int64_t capacity, int64_t* hit_off);

/* ---- per-read scores and per-locus results on the device ------------------------------------ */
/*
 * One read of one locus and the dot plots that score it.  kind 1: abs_dis_m1b (SF:182-203) on pairs
 * (ref_a, alt_a); 2: within_10Perc_m1b (SF:277-294); 3: directed_dis_m1b_redefine_diagnal
 * (SF:241-257); 0: a deletion read - abs_dis_m1b on (ref_a, alt_a) and within_10Perc_m1b on
 * (ref_b, alt_b), the smaller score wins (SF:1718-1726).  len_ref / len_alt are the full allele
 * lengths the gates divide by.  Reads must be sorted by locus.
 */
typedef struct vapor_read {
    int32_t ref_a, alt_a, ref_b, alt_b;
    int32_t kind, locus, len_ref, len_alt;
} vapor_read;

#define VAPOR_GT_TABLE_N 65
#define VAPOR_LOCUS_STRIDE 8 /* doubles per locus: QS, GS, GT (0 = 0/0, 1 = 0/1, 2 = 1/1), GQ, reads scored,
                                positive scores, scores that round to <= 0, reserved; all NaN but [4] = 0 for an 'NA' locus */
/* gt_table[(k * 65 + l) * 2 + {0,1}] = genotype index and quality for k scored reads of which l round to
 * <= 0 (gt_estimate_log_likelihood SF:2054-2069, evaluated by the host with the reference's float64 steps). */
int vapor_plan_set_reads(vapor_plan* plan, int64_t n_reads, const vapor_read* reads, int64_t n_loci,
                         const double* gt_table);
/* join -> clean -> per-read scores (SF:1718-1726 etc.) -> result_organize_ins (SF:1219-1231) ->
 * genotype, all on the device.  d_loci_out: device buffer (may be NULL) filled on the library's stream
 * before it is synchronised; loci_out / read_scores: host copies (may be NULL; a skipped read is NaN). */
int vapor_plan_run_loci(vapor_plan* plan, void* d_loci_out, double* loci_out, double* read_scores);
/*
 * The same run without a host round trip per step.  vapor_plan_run_loci_async only enqueues join -> clean -> finish
 * (with 64 steps in flight it waits for them first; the plan must have run once through vapor_plan_run_loci, which
 * sizes the record slots); vapor_plan_sync waits, makes vapor_plan_timings report the averages over those steps,
 * copies the last step's records to loci_out (may be NULL) and returns VAPOR_E_OVERFLOW if a pair outgrew its slot
 * in one of them.
 * Streams: every plan keeps to one of two streams the context owns, dealt out in turn, so the steps of two plans in
 * flight overlap on the device (a caller that works through a sequence of batches gets this by keeping two plans
 * alive: the reference's loop over loci, vapor_vali/vapor:334-367, has no such stage).  The finish kernel of a step
 * goes to a third, high-priority stream: the plan's next join does not wait for it, its next clean kernels do, and
 * vapor_plan_then / vapor_plan_sync see it through the step's end event.  vapor_plan_then makes a
 * stream of the caller's wait, on the device, for the plan's most recently enqueued step (e.g. before a
 * collective that reads d_loci_out); vapor_plan_after makes the plan's next step wait for what the caller has
 * enqueued on its stream so far (before d_loci_out is overwritten).  vapor_set_stream makes the library enqueue
 * everything on the caller's HIP stream instead (NULL: back to its own streams).
 */
int vapor_set_stream(vapor_ctx* ctx, void* hip_stream);
int vapor_plan_run_loci_async(vapor_plan* plan, void* d_loci_out);
int vapor_plan_then(vapor_plan* plan, void* hip_stream);
int vapor_plan_after(vapor_plan* plan, void* hip_stream);
int vapor_plan_sync(vapor_plan* plan, double* loci_out);

/* ---- one-shot conveniences over the above -------------------------------------------------- */
/* dotdata for a batch (create + run + fetch all + destroy). */
int vapor_dotplot_batch(vapor_ctx* ctx, vapor_seqset* set, int64_t n_pairs, const vapor_pair* pairs,
                        int32_t* hits_ji, int64_t hits_capacity, int64_t* hit_off, int64_t* stats);
/* statistics only (create + run + destroy). */
int vapor_score_batch(vapor_ctx* ctx, vapor_seqset* set, int64_t n_pairs, const vapor_pair* pairs,
                      int64_t* stats);
/* Integer part of window_size_refine's quality check (SF:2030-2046, SF:1154-1171): the self dot
 * plot dotdata(k, seq, seq) of each listed sequence; out[3*t..] = n_hits, n_diag, n_lower. */
int vapor_selfplot_qc(vapor_ctx* ctx, vapor_seqset* set, int32_t n, const int32_t* seq_idx,
                      const int32_t* k, int64_t* out);

/*
 * Cleaning and reductions on caller-supplied dot lists, for callers that hold an explicit
 * list the way clean_dotdata_diagnal_and_anti_diagnal (SF:432-448) and
 * clean_dotdata_diagnal_m1b / clean_dotdata_anti_diagnal_m1b (SF:404-430) are called:
 * list t is hits_ji[2*off[t] .. 2*off[t+1]) as (j, i) pairs, flags[t] VAPOR_PF_* (NULL = both).
 * stats as in vapor_plan_run; hit_flags (may be NULL) one VAPOR_HF_* byte per dot.
 */
int vapor_clean_hits(vapor_ctx* ctx, int64_t n_lists, const int32_t* hits_ji, const int64_t* off,
                     const uint32_t* flags, int64_t* stats, uint8_t* hit_flags);

/*
 * Host helper of the read extraction (cigar2alignstart_by_pos, SF:309-337; no device involved): walks the CIGAR of
 * an alignment starting at align_start until the reference cursor passes start-1; out[0] = offset into the read,
 * out[1] = miss_bp.  VAPOR_E_ARG if the CIGAR holds no operation (IndexError in the reference).
 */
int vapor_cigar2alignstart(const char* cigar, int64_t align_start, int64_t start, int64_t* out);
/* the same over a BAM record's binary CIGAR (n_ops words, length << 4 | operation code in "MIDNSHP=X" order) */
int vapor_cigar2alignstart_ops(const uint32_t* ops, int64_t n_ops, int64_t align_start, int64_t start, int64_t* out);

/*
 * Host helpers of the read extraction, no device involved: `samtools view bam chrom:start-end` piped into
 * chop_pacbio_read_by_pos (SF:339-354), which the reference runs as a process per locus, for one region of an open
 * BGZF/BAM file.  `chunks` are n_chunks (begin, end) virtual-offset pairs from the .bai index (the caller's lookup);
 * their blocks are inflated by a few host threads, the records of reference `tid` walked in file order, each CIGAR
 * walked in binary form to `start` (CG:B,I long CIGARs included), and the reads the reference would keep - alignment
 * start <= start, miss_bp <= flank / 2, more than end - start - miss_bp bases left - are written as ASCII:
 * read r is seq_out[meta[4r] .. + meta[4r+1]), meta[4r+2] = miss_bp, its name the C string at names_out + meta[4r+3].
 * VAPOR_E_OVERFLOW with need[0..2] = bytes of sequence, bytes of names and reads required when a buffer is too small.
 */
typedef struct vapor_bam vapor_bam;
int vapor_bam_open(const char* path, vapor_bam** bam);
int vapor_bam_close(vapor_bam* bam);
int vapor_bam_set_threads(vapor_bam* bam, int32_t n_threads);
const char* vapor_bam_last_error(void);
int vapor_bam_chop(vapor_bam* bam, int32_t tid, int64_t start, int64_t end, int64_t flank, int32_t n_chunks,
                   const uint64_t* chunks, uint8_t* seq_out, int64_t seq_cap, char* names_out, int64_t names_cap,
                   int64_t* meta, int32_t max_reads, int32_t* n_reads, int64_t* need);
/*
 * The read extraction of MANY regions on the DEVICE (vapor_amd/csrc/vapor_bamdev.h): what vapor_bam_chop does for one region on
 * host threads - and the reference as a `samtools view` process per locus piped into chop_pacbio_read_by_pos (SF:339-354) and
 * minimize_pacbio_read_list (SF:1091-1102) - for n_regions regions of the open file `bam` in one call.  Region g =
 * (tid[g], start[g], end[g], flank[g]); its .bai chunks are pairs chunk_first[g] .. chunk_first[g + 1] of `chunks` ((begin, end)
 * virtual offsets).  The host reads the chunks' BGZF blocks with positioned reads and sends them compressed; the device
 * inflates every block (one wavefront a block, CRC-32 checked against the block's trailer), walks the records of every region
 * (one wavefront a region) and keeps what the reference keeps.  Region g's kept reads - at most max_keep (<= 256), the
 * smallest miss_bp first, file order inside one value - are entries kept_first[g] .. kept_first[g + 1] of sq_addr (DEVICE
 * address of the record's packed bases, 4 bits each as BAM stores them), q0 (first base of the read's part) and miss (miss_bp);
 * the three have room for max_keep entries per region; the read is the end - start - miss_bp bases from q0 on
 * (vapor_seqset_create_mixed takes it as it lies).  status[g] = 0, or a positive code where the device leaves the region to
 * the host route (vapor_bam_chop, which also words the errors): a block whose Huffman code needs more table than the kernel
 * keeps, a damaged block, a record that runs past the chunk's blocks, a malformed record, a record without CIGAR, a read offset
 * before the record's first base, more than 256 kept reads.  `batch` owns the inflated data the addresses point into: destroy it
 * (vapor_bam_batch_destroy) after the sequence sets made from them.  One host thread per context, as everywhere.
 */
typedef struct vapor_bam_batch vapor_bam_batch;
int vapor_bam_chop_device(vapor_ctx* ctx, vapor_bam* bam, int32_t n_regions, const int32_t* tid, const int64_t* start,
                          const int64_t* end, const int64_t* flank, const int32_t* chunk_first, const uint64_t* chunks,
                          int32_t max_keep, int32_t* kept_first, uint64_t* sq_addr, int64_t* q0, int64_t* miss, int32_t* status,
                          vapor_bam_batch** batch);
int vapor_bam_batch_destroy(vapor_bam_batch* batch);
/* what the context's last vapor_bam_chop_device did, for measurement (bench.py): out[0..5] = regions, BGZF blocks, compressed bytes
 * sent over the link, inflated bytes, the inflate kernel's duration between two events on its stream (ms), the whole call (ms) */
int vapor_bam_last_stats(vapor_ctx* ctx, double* out, int32_t n);
/* the descriptor and the inflate-thread count of an open file (vapor_bam_chop_device reads with them) */
int vapor_bam_fileno(vapor_bam* bam);
int vapor_bam_threads(vapor_bam* bam);
/*
 * vapor_seqset_create_derived for a set some of whose sequences are on the device already: src_kind[i] = 1 says seq[i] is the
 * DEVICE address of BAM-packed bases (4 bits a base, "=ACMGRSVTWYHKDBN", the first base of a byte in its high half) inside the
 * data of a live vapor_bam_batch of this context, and the sequence is the len[i] bases from base src_first[i] on; src_kind[i] = 0
 * (or src_kind == NULL): bytes on the host, as vapor_seqset_create_derived takes them.  Only the host bytes cross the link.
 * VAPOR_E_ARG for a device source that does not lie inside a live batch.
 */
int vapor_seqset_create_mixed(vapor_ctx* ctx, int32_t n_seqs, const uint8_t* const* seq, const int32_t* len, const uint8_t* flags,
                              const uint8_t* src_kind, const int64_t* src_first, int32_t n_derived, const int32_t* seg_first,
                              const vapor_segment* segs, const uint8_t* derived_flags, int32_t* seq_info, vapor_seqset** set);
/*
 * chop_pacbio_read_by_pos (SF:339-354) over n alignment records the caller holds in memory (no file, no device):
 * pos 1-based leftmost aligned base, ref_span the reference bases a record is taken to cover (the region rule of
 * `samtools view`), cigar[r] the record's CIGAR as text (a C string, walked only as far as the window start), seq_len the
 * read lengths.  keep[r] = 1 and q0_miss[2r], q0_miss[2r+1] = offset into the read and miss_bp for the reads the reference
 * would keep (POS < start + 1, miss_bp <= flank / 2, more than end - start - miss_bp bases from the offset on); the caller
 * slices.  VAPOR_E_ARG for a record without CIGAR operation (IndexError in the reference, SF:331).
 */
int vapor_chop_records(int32_t n, const int64_t* pos, const int64_t* ref_span, const char* const* cigar,
                       const int64_t* seq_len, int64_t start, int64_t end, int64_t flank, int64_t* q0_miss, uint8_t* keep);
/*
 * chop_pacbio_read_by_pos (SF:339-354) followed by minimize_pacbio_read_list (SF:1091-1102: at most max_keep reads, the smallest
 * miss_bp first, input order inside one miss_bp value) for MANY regions in one call (host, no device) - the read selection of a
 * whole batch of loci, which the reference runs as a samtools process and two Python loops per locus.  Region g looks at
 * n_rec[g] records given as vapor_chop_records' arrays (pos[g], ref_span[g], cigar[g], seq_len[g]: one pointer per region);
 * its kept reads are entries kept_first[g] .. kept_first[g + 1] of rec_idx (index into the region's records), q0 (offset into
 * the read) and miss (miss_bp); the three have room for max_keep entries per region.  addr_out (may be NULL, same room)
 * receives seq_addr[g][record] per kept read: the address of that record's sequence where the caller keeps it.  status[g] = VAPOR_E_ARG for a region
 * with a record without CIGAR operation (IndexError in the reference, SF:331), else 0.
 */
int vapor_chop_records_many(int32_t n_regions, const int32_t* n_rec, const int64_t* const* pos, const int64_t* const* ref_span,
                            const char* const* const* cigar, const int64_t* const* seq_len, const int64_t* start,
                            const int64_t* end, const int64_t* flank, int32_t max_keep, int32_t* kept_first, int32_t* rec_idx,
                            int64_t* q0, int64_t* miss, int32_t* status, const uint64_t* const* seq_addr, uint64_t* addr_out);
/*
 * The row tails of a whole output table in one call (host, no device): per locus t with read scores
 * scores[off[t] .. off[t+1]) what result_organize_ins (SF:1219-1231) and gt_estimate_log_likelihood (SF:2054-2069, reading
 * the rounded Rec string back, SF:2056) derive from them - n_pos[t] = scores > 0, qs[t] = their np.mean (numpy's pairwise
 * summation order; 0 when none), n_nonpos[t] = scores not > 0 after round(s, 2), and Rec = ','.join(str(round(s, 2))) as
 * text[text_off[t] .. text_off[t+1]) (ASCII, no terminator).  n_nonpos[t] = -1 where a score is not finite or at least 1e13
 * in size: the caller formats that locus itself.  VAPOR_E_OVERFLOW with text_off[n_loci] = bytes needed when text_cap is too
 * small.  Replaces the per-locus round -> str -> join -> split -> float of the reference's writer (SF:1229, 2056).
 */
int vapor_row_tails(int32_t n_loci, const int64_t* off, const double* scores, double* qs, int32_t* n_pos, int32_t* n_nonpos,
                    char* text, int64_t text_cap, int64_t* text_off);
/*
 * The block decoder of vapor_bam_chop by itself (host, no device): a raw DEFLATE stream (RFC 1951; the payload of a BGZF
 * block) of in_n bytes into exactly out_n bytes.  VAPOR_E_ARG for anything that is not such a stream (truncated, damaged,
 * another size); it reads and writes nothing outside the two buffers.  Replaces the inflate inside `samtools view`
 * (SF:342 runs it as a process per locus).
 */
int vapor_inflate_raw(const uint8_t* in, int64_t in_n, uint8_t* out, int64_t out_n);
/*
 * The CRC-32 (gzip polynomial) every inflated BGZF block is checked against its trailer with (host, no device), by itself:
 * carry-less-multiply folding where the CPU has it, table lookups for the rest and - tables_only != 0 - for everything.
 * 0 for an empty buffer.  Replaces the check inside `samtools view` (htslib verifies each block, SF:342 runs it per locus).
 */
uint32_t vapor_crc32(const uint8_t* data, int64_t n, int32_t tables_only);

#ifdef __cplusplus
}
#endif
#endif /* VAPOR_HIP_H */
