/*
 * ciri_long_hip.h -- C ABI of libclh.so, the MI355X (gfx950) implementation of CIRI-long's per-read hot path.
 *
 * Plain C: pointers and sizes only.  Two groups of entry points:
 *
 *  (1) The six symbols of the reference's libssw.so, with identical signatures, struct layout, ownership and error
 *      conventions, so that the reference's ctypes wrapper (libs/striped_smith_waterman/ssw_wrap.py:54-72,278-288)
 *      can load this library in place of libssw.so.  Declared in ssw_legacy.h.
 *
 *  (2) Batched entry points (this file).  The reference makes one FFI round trip per alignment
 *      (ssw_wrap.py:187-209 <- CIRI_long/find_bsj.py:203-216, CIRI_long/collapse.py:157-173,251-265); a GPU wants
 *      thousands of alignments per launch.  A batch is a pair of packed int8 code arrays (A=0 C=1 G=2 T=3 N=4, the
 *      encoding of ssw_wrap.py:50,243-250) plus offset tables; alignment a is reads[read_off[a]..read_off[a+1]) against
 *      refs[ref_off[a]..ref_off[a+1]).  Results are what n sequential ssw_init+ssw_align calls would return.
 *
 * All functions return 0 on success and a negative CLH_E_* code on failure; clh_last_error() gives the text.
 * There is no CPU fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef CIRI_LONG_HIP_H
#define CIRI_LONG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)     /* the library itself is built with -fvisibility=hidden: only the C ABI is exported */

#define CLH_E_HIP        (-1)   /* HIP runtime error (no device, launch failure, out of memory) */
#define CLH_E_ARG        (-2)   /* invalid argument */
#define CLH_E_UNSUPPORTED (-3)  /* valid for the reference but not implemented here (see text) */
#define CLH_E_CAPACITY   (-4)   /* caller's output buffer too small */

/* status bits of clh_align_t.status */
#define CLH_ST_WORD        1    /* 16-bit regime (ssw.c:806-809) */
#define CLH_ST_NULL        2    /* the reference would have returned NULL (ssw.c:810-813: score_size 0 overflow) */
#define CLH_ST_TRACE_ERR   4    /* the reference's "Trace back error" (ssw.c:674-682) */
#define CLH_ST_NO_CIGAR    8    /* CIGAR not produced because of flag/filters (ssw.c:834,850) */
#define CLH_ST_CIGAR_TRUNC 16   /* no CIGAR for a capacity reason, scores and coordinates are valid: the traceback workspace ran out
                                 * (retry with a smaller batch), or the band is wider than 2048 cells on an aligned reference of more than
                                 * 2048 bases AND read + reference of the aligned part exceed 12 kB (or the read 4096 rows) */

/* One result row: the fields of s_align (ssw.h:42-52) with the cigar pointer replaced by a slice of the
 * caller's cigar buffer. */
typedef struct {
    uint16_t score1;
    uint16_t score2;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    int32_t ref_end2;
    int32_t cigar_off;    /* first u32 of this alignment's CIGAR in cigar_buf, -1 if none */
    int32_t cigar_len;
    int32_t status;
} clh_align_t;

typedef struct clh_ctx clh_ctx;     /* one per (process, GPU) */
typedef struct clh_plan clh_plan;   /* a batch shape: offsets, scoring, bucketing, device workspaces */

int clh_device_count(void);
const char* clh_last_error(void);
const char* clh_version(void);

clh_ctx* clh_create(int device);    /* NULL on failure */
void clh_destroy(clh_ctx* ctx);
int clh_device_of(const clh_ctx* ctx);

/* Scoring and reporting options; meaning and defaults follow ssw_init/ssw_align (ssw.h:54-120). */
typedef struct {
    const int8_t* mat;     /* n_mat x n_mat substitution matrix, row = reference code (ssw.c:621) */
    int32_t n_mat;         /* <= 5 */
    uint8_t gap_open;      /* absolute values, gap_open >= gap_extend required */
    uint8_t gap_extend;
    uint8_t flag;          /* ssw_align flag; ssw_wrap.py always passes 1 */
    int8_t score_size;     /* ssw_init score_size; ssw_wrap.py always passes 2 */
    uint16_t filters;
    int32_t filterd;
    int32_t want_score2;   /* 0: skip the second-best scan (score2 = 0, ref_end2 = 0); CIRI-long never reads it */
    int32_t want_cigar;    /* 0: skip the traceback even if flag asks for it (find_bsj.py:204,214 never reads it) */
} clh_ssw_opts;

/* mask_len may be NULL: ssw_wrap.py:196-199 rule (len/2 if len > 30 else 15). */
clh_plan* clh_ssw_plan(clh_ctx* ctx, int32_t n, const int64_t* read_off, const int64_t* ref_off,
                       const int32_t* mask_len, const clh_ssw_opts* opts);
void clh_plan_destroy(clh_plan* plan);

/* Diagnostics of the last run's CIGAR step (K1b): counts[0] = alignments the row traceback kernel handed to its wide form
 * (bands of 513..2048 cells), counts[1] = alignments that went on to the anti-diagonal kernel (walks that leave the final band,
 * where banded_sw, ssw.c:636-696, reads direction bytes of earlier band iterations; bands above 2048 cells).  Waits for the run. */
int clh_plan_traceback_counts(clh_plan* pl, int32_t* counts);

/* Diagnostics of the exact column prefilter (csrc/ssw_prefilter.hip) in front of the score pass of short reads on long windows --
 * the shape of find_bsj.py:196-216, a 20..254-base clip against hit +- 200 kb: a bit-vector edit-distance bound finds the blocks
 * of the window that can hold the maximum of sw_sse2_byte (ssw.c:123-345) and only those are computed; results are identical with
 * and without it.  out[0] = alignments of that class in the last run, out[1] = of them with candidate slices instead of the
 * whole window, out[2] = slices run, out[3] = window columns computed, out[4] = window columns of the class, out[5] = alignments whose
 * window went through the second stage as well (the indel-distance bound, for windows the first leaves too much of).  Waits for the run. */
int clh_plan_prefilter_stats(clh_plan* pl, int64_t* out);

/* ssw_prefilter_kernel (the first stage) alone, after a run with profiling on: ms[0] / ms[1] = its duration for the class of reads up to
 * 254 bases and for the class of longer reads (0 when it did not run); work[2 c] = window columns x W 32-row words of the read (its
 * pieces), work[2 c + 1] = the same columns x (11 W + 8) -- the integer instructions per column and lane, i.e. what the kernel's
 * issue-rate roofline counts.  Measurement only (bench.py: prefilter_roofline); no reference counterpart. */
int clh_plan_prefilter_timing(clh_plan* pl, float* ms, int64_t* work);

/* Launch the batch on packed code arrays that already live in HBM (device pointers).  Asynchronous on `stream`
 * (a hipStream_t).  NULL selects the context's own, private, non-blocking stream -- NOT the legacy default stream: work
 * a caller has queued on the default stream (or on any other stream) is then unordered with these kernels.  A caller
 * who produces inputs or consumes clh_ssw_results_dev() on a stream of its own passes THAT stream; hipStreamLegacy /
 * hipStreamPerThread are passed through like any other handle.  Results stay in HBM until clh_ssw_fetch, which waits
 * for the stream of the last run.  The same holds for every `stream` argument of this header. */
int clh_ssw_run(clh_plan* plan, const void* d_reads, const void* d_refs, void* stream);

/* Padding contract of d_refs.  For windows of 32 kb and more the column prefilter reads the window text in whole, address-aligned
 * 256-byte blocks: when d_refs is 256-byte aligned it may read up to 255 bytes past the last byte any window uses (never in front of
 * d_refs).  The buffer must be readable up to that boundary -- every hipMalloc / caching-allocator block is (allocations are rounded
 * to at least 256 bytes), a pointer INTO such a block that ends short of the boundary is not.  A caller who cannot promise it states the
 * buffer's size here (bytes from d_refs; -1 = unstated, the default): a run whose last block would cross it uses the static window
 * slices instead of the prefilter.  A d_refs that is not 256-byte aligned always does.  Results are the same either way. */
int clh_plan_set_refs_bytes(clh_plan* plan, int64_t nbytes);

/* Wait for the last run and copy results out.  cigar_buf may be NULL.  *cigar_used receives the u32 count (the CIGARs come back
 * as one dense array).  It waits for the completion of the plan's last clh_ssw_run (an event recorded behind its last launch),
 * not for work the caller queued on the stream afterwards. */
int clh_ssw_fetch(clh_plan* plan, clh_align_t* out, uint32_t* cigar_buf, int64_t cigar_cap, int64_t* cigar_used);

/* Device pointer of the raw result table of the last run (8 x int32 per alignment: score1 score2 ref_begin1 ref_end1
 * read_begin1 read_end1 ref_end2 status), for callers that keep post-processing on the GPU. */
const void* clh_ssw_results_dev(const clh_plan* plan);

/* Measurement hooks (bench.py): per launch-segment (one read-length class each) HIP-event durations of the score
 * kernel (K1), and of the traceback kernel (K1b): k1b_ms[0] = the small-window launch over all alignments,
 * k1b_ms[1] = the large-window launches that redo the outliers.  Both return the number of segments. */
int clh_plan_set_profiling(clh_plan* plan, int on);
int clh_plan_segments(const clh_plan* plan, int32_t cap, int32_t* rv, int32_t* count, int64_t* read_bases, int64_t* ref_bases);
int clh_plan_timing(clh_plan* plan, int32_t cap, float* k1_ms, float* k1b_ms);

/* Host-buffer convenience: plan + upload + run + fetch. */
int clh_ssw_batch(clh_ctx* ctx, int32_t n, const int8_t* reads, const int64_t* read_off, const int8_t* refs,
                  const int64_t* ref_off, const int32_t* mask_len, const clh_ssw_opts* opts,
                  clh_align_t* out, uint32_t* cigar_buf, int64_t cigar_cap, int64_t* cigar_used);

/* ---- cyclic consensus: the batch form of pyccs.find_consensus (CIRI_long/find_ccs.py:14) -----------------------
 * pyccs/spoa are external and absent from the reference tree (PARITY UNPINNED).  Per read: tandem-repeat period by
 * 8-mer self-matches and copy boundaries (this project's specification, oracle/ccs_oracle.c), then the consensus of
 * the copies by partial-order alignment as spoa computes it (oracle/poa_oracle.c: local alignment, scores
 * 10/-4/-8/-2/-24/-1 -- the call of the reference's tests/test_poa.py:30).  segs holds [start,end) pairs, 65 per read;
 * ccs is packed like reads. */
#define CLH_CCS_SEG_CAP 65
typedef struct {
    int32_t nseg;      /* 0: no tandem repeat / no consensus (find_consensus would return (None, None)) */
    int32_t ccs_len;
    int32_t period;
    int32_t status;    /* 0 ok; >0 no consensus because of a limit of this kernel (counted, see clh_ccs_plan_stats): 1 workspace, 2 graph limits (48
                          in-edges, 8 letters in a column, 65000 rows), 3 output, 4 sequence longer than 2800 bases, 5 back-track guard, 6 a DP
                          cell left the 16-bit range (global / overlap modes with costly gaps), 7 an alignment without a base (spoa throws) */
} clh_ccs_t;
typedef struct clh_ccs_plan clh_ccs_plan;
clh_ccs_plan* clh_ccs_plan_create(clh_ctx* ctx, int32_t n, const int64_t* read_off);
void clh_ccs_plan_destroy(clh_ccs_plan* plan);
int clh_ccs_run(clh_ccs_plan* plan, const void* d_reads, void* stream);
int clh_ccs_fetch(clh_ccs_plan* plan, clh_ccs_t* out, int32_t* segs, int8_t* ccs);
/* Device pointers of the last run's outputs (rows: clh_ccs_t[n]; segs: int32[n][2*65]; ccs: packed codes at the read
 * offsets), for callers that keep the next step on the GPU.  Any of the three may be NULL. */
int clh_ccs_results_dev(const clh_ccs_plan* plan, const void** rows, const void** segs, const void** ccs);
/* Workspace tiers of the plan and how the last run used them: out[6] = {first-tier slots, bytes per slot, large slots,
 * bytes per large slot, reads that ran in a large slot claimed on the fly, reads run by the second launch}. */
int clh_ccs_plan_info(clh_ccs_plan* plan, int64_t* out);
/* Work and losses of the last run: out[16] = {DP cells, DP row steps, alignments whose back-track left the band of cells the forward
 * pass had stored (run again with all cells: cost, never a different answer), reads that ended with status 1, 2, ... 7 (no
 * consensus because of a limit of this kernel -- never silently: callers count and report them), 0 ...}. */
int clh_ccs_plan_stats(clh_ccs_plan* plan, int64_t* out);
/* HIP-event durations (ms) of the last run: ms[0] = repeat scan (K2), ms[1] = partial-order consensus (K3). */
int clh_ccs_plan_timing(clh_ccs_plan* plan, float* ms);
int clh_ccs_batch(clh_ctx* ctx, int32_t n, const int8_t* reads, const int64_t* read_off, clh_ccs_t* out, int32_t* segs, int8_t* ccs);

/* The spoa.poa call shape -- poa(seqs, algorithm, genmsa, m, n, g, e, q, c), collapse.py:267,504 (algorithm 2),
 * tests/test_poa.py:30 (algorithm 0, genmsa) -- for a batch of groups of sequences.  Group k = sequences
 * [group_off[k], group_off[k+1]) of the packed array (any number >= 1, groups contiguous, each sequence <= 2800 bases).
 * opts NULL = {0, 10, -4, -8, -2, -24, -1, 0}.  algorithm: 0 local, 1 global, 2 overlap; a gap of k bases costs
 * max(g + (k-1) e, q + (k-1) c); g <= q or e >= c selects the one-piece (affine) model as spoa does; the linear model
 * (g >= e) and scores outside the kernel's 16-bit cells (m 1..11, e - g <= 6, c - q <= 30) fail with CLH_E_UNSUPPORTED --
 * nothing is silently ignored.  min_coverage > 0 leaves nodes crossed by fewer sequences out of the consensus.
 * out_ccs is packed by the offset of each group's first sequence; out_len[k] < 0 when no consensus could be built
 * (-(1 + status), status as in clh_ccs_t).  msa_col (may be NULL; one int32 per base, packed like seqs) receives the MSA
 * column of every base, msa_ncols[k] the number of columns of group k: row i of the MSA is '-' everywhere except
 * row[msa_col[b]] = base b for the bases b of sequence i.  aln_score (may be NULL; int32[ngroups][65]) receives the
 * end-cell score of the alignment of each of the first 65 sequences of a group. */
typedef struct { int32_t algorithm, m, n, g, e, q, c, min_coverage; } clh_poa_opts;
/* clh_ccs_plan_stats of the last clh_poa_batch of this context (out[16], same layout) */
int clh_poa_last_stats(clh_ctx* ctx, int64_t* out);
int clh_poa_batch(clh_ctx* ctx, int32_t ngroups, const int8_t* seqs, const int64_t* seq_off, const int64_t* group_off,
                  const clh_poa_opts* opts, int32_t* out_len, int8_t* out_ccs, int32_t* msa_col, int32_t* msa_ncols, int32_t* aln_score);

/* ---- Stage 1 from file to file -------------------------------------------------------------
 * The read loop of find_ccs_reads (CIRI_long/find_ccs.py:29-96) in native code: FASTA/FASTQ, plain or gzip, one header and
 * one sequence line per record; writes tmp/{prefix}.ccs.fa and tmp/{prefix}.raw.fa in the reference's format
 * (find_ccs.py:94-95), reads with a consensus only, input order.  batch_reads <= 0 selects 65536.  too_long counts reads
 * above 16 M bases (not scanned). */
typedef struct { int64_t total_reads, ro_reads, too_long, capacity_dropped; } clh_ccs_file_stats;   /* capacity_dropped: reads with a
                                                                   tandem repeat that a limit of the kernel left without a consensus (status > 0) */
int clh_ccs_file(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                 int32_t batch_reads, clh_ccs_file_stats* stats);
/* The same for the records [first_record, first_record + max_records) of the file (max_records < 0: to the end) -- one
 * rank's contiguous shard of the reads (the reference hands chunks of the record stream to its pool, find_ccs.py:66-75);
 * concatenating the outputs of consecutive shards gives the files of the unsharded call, byte for byte. */
int clh_ccs_file_range(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                       int32_t batch_reads, int64_t first_record, int64_t max_records, clh_ccs_file_stats* stats);
/* number of records of a FASTA/FASTQ(.gz) file as that loop counts them (host only) */
int clh_fastx_count(const char* in_path, int is_fastq, int64_t* n_records);
/* the same count plus the byte offset of every `every`-th record (0, every, 2 every ...; at most `cap`) of an uncompressed file
 * (*n_offsets = 0 for a gzip file), and stage 1 for the records [first_record, first_record + max_records) counted from such an
 * offset: a rank of a sharded `call` (the reference hands chunks of the record stream to its pool, find_ccs.py:66-75) seeks to its
 * shard instead of reading past what lies in front of it (host I/O only; find_ccs.py:29-64 for what a record is) */
int clh_fastx_index(const char* in_path, int is_fastq, int64_t every, int64_t* n_records, int64_t* offsets, int64_t cap, int64_t* n_offsets);
int clh_ccs_file_at(clh_ctx* ctx, const char* in_path, int is_fastq, const char* ccs_fa_path, const char* raw_fa_path,
                    int32_t batch_reads, int64_t byte_offset, int64_t first_record, int64_t max_records, clh_ccs_file_stats* stats);
/* The file stage keeps its host buffers (six batches of file text and base codes, ~70 MB each) and two device buffers between calls,
 * one set per process; this gives them back. */
void clh_ccs_file_release_buffers(void);

/* ---- Resident genome ---------------------------------------------------------------------
 * The reference builds, per clipped read, a window string of hit +- 200 kb, counts its 'N', reverse-complements it for
 * minus-strand hits and encodes it base by base in Python (CIRI_long/find_bsj.py:196-201,214;
 * libs/striped_smith_waterman/ssw_wrap.py:234-252).  Here the genome (all contigs concatenated by the caller, who keeps
 * the contig offsets) is encoded once into HBM and a window is (offset, length, strand): clh_ssw_plan_windows +
 * clh_ssw_run(plan, d_reads, clh_genome_codes(genome), stream) give the results clh_ssw_plan/clh_ssw_run give for the
 * window strings built the reference's way -- including its handling of lower-case bases (reversed, not complemented). */
typedef struct clh_genome clh_genome;
clh_genome* clh_genome_create(clh_ctx* ctx, const char* ascii, int64_t len);
void clh_genome_destroy(clh_genome* genome);
const void* clh_genome_codes(const clh_genome* genome);      /* device pointer */
int64_t clh_genome_length(const clh_genome* genome);
/* out[k] = number of upper-case 'N' in [off[k], off[k]+len[k]) -- Counter(window)['N'] of find_bsj.py:199 */
int clh_genome_count_n(clh_genome* genome, int32_t n, const int64_t* off, const int64_t* len, int64_t* out);
/* Annotated splice sites for clh_splice_signal_batch: the reference's splice_site_index / circ_ss_idx
 * (CIRI_long/align.py:235-252, 275-316: index[contig][pos][strand]['start'|'end']) flattened to four runs of genome-wide
 * positions (contig offset in the resident genome + pos), each strictly ascending, concatenated in `pos` in the order
 * '+' starts, '+' ends, '-' starts, '-' ends; count4 = their lengths.  Replaces any earlier set; all zero = none. */
int clh_genome_set_splice_sites(clh_genome* genome, const int64_t* pos, const int64_t* count4);
/* Splice signals around n candidate back-splice junctions [start, end) (0-based) of contigs of the resident genome --
 * CIRI_long/align.py:477-493 (free sliding), :495-568 (pairs of annotated sites, if sites were set), :571-695
 * (find_denovo_signal, annotated shifts joining the motif occurrences) and :698-733 (ranking), as called from
 * find_bsj.py:286-301 with search_length = clip_base + search_extra (10), shift_threshold (3).
 * host_mask: strands of the host gene, bit 0 '+', bit 1 '-'.  out[8k..8k+7] = status (0 done, 1 = outside this kernel's
 * domain: the neighbourhood leaves the contig or holds non-ACGTN characters; run the Python statement), us_free,
 * ds_free, found (0 none, 1 de novo, 2 annotated pair), strand (0 '+', 1 '-'), us_shift, ds_shift, motif (de novo:
 * index into GT-AG, GC-AG, AT-AC, GT-AC, AT-AG).
 * is_canonical: bit 0 = search GT-AG only; bit 1 = the two search windows are cut the way a minimap2 index serves sequences
 * (mappy.Aligner.seq, env.GENOME of the reference's main pass, find_bsj.py:340-341: no sequence for a start outside the contig,
 * end clipped) instead of the way Python slices a string (align.Fasta.seq, the short-read pass, find_bsj.py:455,462). */
int clh_splice_signal_batch(clh_genome* genome, int32_t n, const int64_t* ctg_off, const int64_t* ctg_len, const int64_t* start,
                            const int64_t* end, const int32_t* clip_base, const int32_t* host_mask, int32_t search_extra,
                            int32_t shift_threshold, int32_t is_canonical, int32_t* out);
/* like clh_ssw_plan, references given as windows of the resident genome; win_rc[k] != 0: minus strand */
clh_plan* clh_ssw_plan_windows(clh_ctx* ctx, int32_t n, const int64_t* read_off, const int64_t* win_off, const int32_t* win_len,
                               const uint8_t* win_rc, const int32_t* mask_len, const clh_ssw_opts* opts);

/* one call: reads from the host, references = windows of the resident genome (same outputs as clh_ssw_batch) */
int clh_ssw_windows_batch(clh_genome* genome, int32_t n, const int8_t* reads, const int64_t* read_off, const int64_t* win_off,
                          const int32_t* win_len, const uint8_t* win_rc, const int32_t* mask_len, const clh_ssw_opts* opts,
                          clh_align_t* out, uint32_t* cigar_buf, int64_t cigar_cap, int64_t* cigar_used);

/* Unit-cost edit distance of n pairs of byte strings: what the reference's distance(x, y) returns (CIRI_long/utils.py:
 * 153-159; python-Levenshtein for <= 50 characters, edlib otherwise -- the same integer), used pairwise by
 * cluster_sequence (collapse.py:466-473) and per candidate by avg_score (collapse.py:156-158).  Strings are compared
 * byte for byte (case-sensitive, like the reference); pair k is a[a_off[k]..a_off[k+1]) against b[b_off[k]..b_off[k+1]).
 * No length limit (a shorter string above 4096 bytes is swept in passes). */
int clh_edit_distance_batch(clh_ctx* ctx, int32_t n, const uint8_t* a, const int64_t* a_off, const uint8_t* b, const int64_t* b_off,
                            int32_t* out);

/* the same in three steps, for batches that stay resident (strings uploaded once, any number of runs) */
typedef struct clh_edit_plan clh_edit_plan;
clh_edit_plan* clh_edit_plan_create(clh_ctx* ctx, int32_t n, const uint8_t* a, const int64_t* a_off, const uint8_t* b, const int64_t* b_off);
void clh_edit_plan_destroy(clh_edit_plan* plan);
int clh_edit_plan_run(clh_edit_plan* plan, void* stream);
int clh_edit_plan_fetch(clh_edit_plan* plan, int32_t* out);
int clh_edit_plan_timing(clh_edit_plan* plan, float* ms);       /* HIP-event duration of the last run */

/* ASCII -> codes exactly as ssw_wrap.py:234-252 (A/a C/c G/g T/t N/n, anything else 4), on the host. */
void clh_encode_dna(const char* seq, int64_t len, int8_t* out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
