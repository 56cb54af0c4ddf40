// clh_device.h -- structures shared by the HIP kernels and the host side of libclh (internal; the public
// C ABI is include/ciri_long_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clh {

enum {
    CLH_STATUS_WORD = 1,        // scores come from the 16-bit regime (ssw.c:806-809)
    CLH_STATUS_OVERFLOW8 = 2,   // score_size 0 and the 8-bit pass overflowed: reference returns NULL (ssw.c:810-813)
    CLH_STATUS_TRACE_ERR = 4,   // traceback left the band (reference: "Trace back error", ssw.c:674-682)
    CLH_STATUS_NO_CIGAR = 8,    // CIGAR not requested / filtered by flag (ssw.c:850)
    CLH_STATUS_CIGAR_TRUNC = 16, // CIGAR buffer share / traceback pool exhausted
    CLH_STATUS_NEED_BIG = 32     // internal: traceback must be redone with the large LDS window
};

// One alignment = one workgroup of one wavefront.  Offsets are into the packed batch arrays.
struct SswTask {
    int64_t read_off;     // into reads (int8 codes)
    int64_t ref_off;      // into refs: first base of the reference; for a reverse-complemented genome window (ref_rc) its LAST byte
    int64_t colmax_off;   // into the column-maximum workspace (u16), one slot per reference base
    int64_t dir_off;      // strip-boundary workspace of this task inside `dirs` (bytes; reads longer than 4096 bases only)
    int32_t read_len;
    int32_t ref_len;
    int32_t mask_len;
    int32_t out_index;    // row of the result / cigar tables this task fills
    int32_t cigar_off;    // first u32 of this task's share of the CIGAR buffer
    int32_t cigar_cap;    // u32 slots in that share
    int32_t ref_rc;       // 1: the reference is read backwards and complemented on the fly (minus-strand window of a resident genome)
    int32_t pad;
};

// A reference byte is a base code in bits 0-2 (A0 C1 G2 T3, anything else 4).  Bytes of a resident genome also carry
// bit 3 = lower-case letter (the reference's revcomp() leaves lower case uncomplemented, utils.py:118-120) and bit 4 =
// upper-case 'N' (what Counter(window)['N'] counts, find_bsj.py:199).
__device__ __forceinline__ int ref_code(int byte, int rc) {
    int c = byte & 7;
    c = c > 4 ? 4 : c;
    return (rc && !(byte & 8) && c < 4) ? 3 - c : c;
}

struct SswResult {        // 32 bytes; the s_align fields of ssw.h:42-52 minus the pointer
    int32_t score1, score2, ref_begin1, ref_end1, read_begin1, read_end1, ref_end2, status;
};

// K1s on long windows (ssw_scan.hip): one forward pass cut into slices of the window
struct ScanSlice { int32_t task, c_begin, own_begin, c_end, part, pad0, pad1, pad2; };   // columns [c_begin, c_end) computed, [own_begin, c_end) counted
struct ScanPart { int32_t max, col, row, pad; };                                        // best cell of one slice

// Exact prefilter of the sliced scan class (ssw_prefilter.hip; tools/prefilter_model.py): per block of 256 bytes of window text
// the minimum of d(j), the edit distance of the whole read to a window substring ending at column j; H(j) <= M L - c d(j).
static constexpr int kPfBlock = 256;
struct PfWin {            // host-built, one per alignment (task of the class): the blocks its window touches
    int32_t nsub;         // blocks
    int32_t phase;        // column 0 sits `phase` bytes into the first block (in processing order)
    int32_t mem_block0;   // refs block of processing block 0 (minus-strand windows run down the addresses: the highest block)
    int32_t piece_first;  // the alignment's pieces in the piece table
    int32_t piece_count;  // 1 for reads up to 254 bases; longer reads go through the bit-vector pass in pieces of <= 254 rows
    int32_t d_off;        // K1w class: first entry of the alignment's summed block bound (uint16 per block)
    int32_t work_first, work_count;   // K1s class: the alignment's entries of the work list (the second stage runs them again)
};
struct PfTask {           // host-built, one per PIECE of a read
    int32_t task;         // alignment (index into the class's tasks and PfWin table)
    int32_t row0, rows;   // the piece's rows of the read
    int32_t sub_off;      // first entry of the piece's block minima
};
struct PfWork {           // one workgroup of the prefilter: 64 lanes x `bpl` blocks of one piece
    int32_t piece, first_block;
};
struct PfOut {            // device-filled per alignment: its slices / tasks in the queue, what the filter saw
    int32_t first, count, s0, pruned;
};
struct PfCtl {            // device control block of one run (zeroed before it)
    int32_t qcount, qnext, n_pruned, seed_next;
    unsigned long long cols_scanned, cols_window;
    int32_t q2count, q2next, n_stage2, pad;      // second stage (indel distance): work items queued / taken, alignments sent there
};
// K1w on long windows (ssw_scan_wide.hip): per alignment 2 seed rows + 2 x 64 candidate rows behind the real result rows
static constexpr int kWsRows = 130;       // scratch rows per alignment: [0] seed, [1] seed in the word regime, [2 + 2 k] / [3 + 2 k] candidate k as-is / word regime
struct WsTask {           // a device-made K1w task: the read against columns [c_begin, c_end) of its alignment's window
    int32_t task;         // alignment (index into the class's tasks)
    int32_t c_begin, c_end;
    int32_t row;          // scratch row of the alignment (0 .. kWsRows - 1)
    int32_t force_word;   // 1: the word regime whatever the byte pass would say (ssw.c:806-809 with the decision made elsewhere)
    int32_t pad0, pad1, pad2;
};

struct SswParams {
    const int8_t* reads;
    const int8_t* refs;
    const SswTask* tasks;
    SswResult* results;
    uint16_t* colmax;     // nullptr: skip the second-best scan (score2 = 0)
    uint32_t* cigars;     // BAM-style u32 (len<<4|op), ssw.h:131-170
    int32_t* cigar_len;   // per out_index
    uint8_t* dirs;        // strip-boundary workspace (row strips of long reads)
    int8_t mat[32];       // n*n substitution matrix (n <= 5)
    int32_t n, gapO, gapE, bias, max_match, score_size, flag, filters, filterd;
    int32_t null_code;    // 4 if mat scores code 4 as 0 against everything (then fill/drain columns reuse it), else 5
    const ScanSlice* slices;   // sliced scan class only
    ScanPart* parts;
    int32_t n_real;            // result rows of real alignments; tasks with out_index >= n_real are window slices of the
                               // anti-diagonal classes (scratch rows, no CIGAR), see clh_api.hip and ssw_combine_kernel
    const int32_t* slice_base; // first window column of scratch row k (index out_index - n_real)
    // the long-window classes behind the prefilter: slices / tasks are made on the device (ssw_scan.hip: ssw_scan_pick_kernel;
    // ssw_scan_wide.hip: ssw_scanw_seed_kernel, ssw_scanw_pick_kernel)
    const PfWin* pf_win;       // nullptr: static slices from `slices`
    const PfTask* pf_tasks;    // pieces
    const PfWork* pf_work;
    uint8_t* pf_dmin;          // block minima per piece (nullptr at run time: the filter is off for this run, the pick kernels write the static slices)
    ScanSlice* pf_slices;      // K1s class: the queue
    PfOut* pf_out;
    PfCtl* pf_ctl;
    int32_t pf_bpl, pf_cap;    // blocks per lane of the prefilter; capacity of the queue
    int32_t* pf_q2;            // K1s class: work items of the second stage (indel distance), or nullptr: one stage
    int32_t pf2_always;        // tests: every window of the class through the second stage, whatever the first left (CLH_PF2_ALWAYS)
    int32_t pf2_share;         // a window goes to the second stage when the first leaves more than 1 / pf2_share of it (default 8; CLH_PF2_SHARE)
    uint16_t* ws_bound;        // K1w class: summed block bound D per alignment (PfWin.d_off)
    WsTask* ws_tasks;          // K1w class: [0, 2 n) seed tasks (fixed places), then the candidate queue
    int32_t ws_row0;           // first scratch result row of the class (alignment a: ws_row0 + a * kWsRows)
    int32_t ws_slot_bytes;     // K1w workspace of one persistent workgroup inside `dirs`
    int64_t ws_dirs_off;       // where those workspaces start inside `dirs`
    int32_t no_guess;          // K1w: 1 = always the byte pass first (CLH_NO_GUESS: A/B of the overflow guess, ssw_scan_wide.hip)
};

// ---- cyclic consensus (K2/K3, csrc/ccs_poa.hip) ----------------------------------------------------------------
static constexpr int CCS_SEG_CAP = 65;          // 64 cuts + a partial tail

struct CcsScan {          // K2 output per read
    int32_t period;       // 0 = no tandem repeat
    int32_t ncuts;
    int32_t support;
    int32_t cuts[64];     // copy boundaries after 0
};

struct CcsResult {        // K3 output per read
    int32_t nseg;         // 0 = no consensus
    int32_t ccs_len;
    int32_t period;
    int32_t status;       // 0 ok, 1 workspace slot too small, 2 graph limits (in-degree above 48, aligned set above 8 letters, 65000 rows), 3 consensus overflow, 4 sequence above 2800 bases,
                          // 5 back-track guard, 6 a cell left the 16-bit range, 7 an alignment without a base (spoa throws)
};

// scores and mode of the partial-order aligner (spoa.poa(seqs, algorithm, genmsa, m, n, g, e, q, c)); affine is passed as
// q = g, c = e; the linear sub-type (g >= e) is refused by the host side
struct PoaScores {
    int32_t algorithm;    // 0 local, 1 global, 2 overlap
    int32_t m, n, g, e, q, c;
    int32_t min_cov;      // consensus keeps nodes crossed by >= min_cov sequences; -1: (sequences + 1) / 2 (find_consensus)
};

struct CcsParams {
    const int8_t* reads;
    const int64_t* read_off;   // device copy, n+1 entries
    CcsScan* scan;
    CcsResult* results;
    int32_t* segs;             // [n][2*CCS_SEG_CAP]
    int8_t* ccs;               // packed like reads (a consensus is never longer than its read)
    uint8_t* poa_ws;           // nslots * slot_bytes
    int* work_counter;
    int* stats;                // [0] reads that ran in a claimed large slot, [1] reads run by the second launch, [2..3] DP cells and [4..5] DP row
                               // steps of the run (64 bits each), [8 + s] reads that ended with status s (1..7: lost to a limit of the kernel)
    const int32_t* work_order; // reads, longest first (or nullptr)
    unsigned long long slot_bytes;
    int32_t n;
    int32_t lcap;              // >= longest read of the batch
    const int32_t* long_idx;   // K2: reads above k2_lds_max bases (scanned out of an HBM workspace)
    uint8_t* k2_ws;
    unsigned long long k2_slot;
    int32_t n_long, k2_lmax, k2_lds_max;
    int32_t k2_begin;          // K2 launch class: workgroup b scans read work_order[k2_begin + b] with an LDS block for lcap bases
    uint8_t* big_ws;           // K3: large slots that first-tier waves may claim for a read that does not fit their own
    unsigned long long big_slot_bytes;
    int* big_busy;
    int32_t n_big;
    int32_t tier;              // K3: 0 = every read (too large for the slot: status 1); 1 = only the reads left with status 1
    PoaScores sc;
    // explicit sequences (the spoa.poa call shape): "read" rd is the concatenation of the sequences of group rd, whose
    // inner boundaries (relative to the group's first base) are xcuts[xcut_off[rd] .. xcut_off[rd+1]); nullptr: copies from `scan`
    const int32_t* xcuts;
    const int64_t* xcut_off;
    int32_t* msa_col;          // optional, packed like reads: MSA column of every base; msa_ncols[rd] = number of columns
    int32_t* msa_ncols;
    int32_t* aln_score;        // optional, [n][CCS_SEG_CAP]: end-cell score of the alignment of each of the first 65 sequences (tests)
    int32_t* wide_list;        // K3: reads that need the wide (32-bit) form of the pass, appended by the packed kernel
    int* wide_count;
};

hipError_t launch_ccs_scan(const CcsParams& p, int count, bool with_long, hipStream_t stream);
hipError_t launch_poa(const CcsParams& p, int nslots, hipStream_t stream);
hipError_t launch_poa_wide(const CcsParams& p, int nslots, hipStream_t stream);
hipError_t launch_ccs_work_order(const CcsScan* scan, int n, int32_t* order, hipStream_t stream);   // K3's work list by cost after K2
size_t poa_slot_bytes_host(int ncap, int mcap);
size_t poa_slot_min_bytes_host(int ncap, int mcap);
size_t poa_slot_bytes_host_w(int ncap, int mcap);
static constexpr int kK2LdsMax = 16000;
inline size_t k2_long_slot_bytes(int lmax) { return ((size_t)8 * ((size_t)lmax / 2 + 2) + (size_t)6 * (size_t)lmax + 255) & ~(size_t)255; }

struct EdTask {            // K4: one pair; the pattern is the shorter string
    int64_t pat_off, txt_off;   // into the packed symbol array (codes 0..nsym-1, one per byte)
    int32_t pat_len, txt_len;
    int32_t out_index;
    int32_t carry_off64;        // pattern above 4096 symbols: its two between-pass delta buffers, in units of 64 bytes; else -1
};
hipError_t launch_genome_encode(const char* ascii, uint8_t* codes, unsigned int* block_n, long long len, hipStream_t stream);
// K6: one candidate junction [start, end) of a contig of the resident genome (splice_scan.hip)
struct SpliceTask {
    int64_t ctg_off;      // first byte of the contig in the genome codes
    int64_t ctg_len;
    int64_t start, end;   // 0-based, end exclusive (circ_start, circ_end of find_bsj.py:279-301)
    int32_t clip_base;
    int32_t host_mask;    // strands of the host gene(s): bit 0 '+', bit 1 '-' (align.find_host_gene), 0 = none
};
// annotated splice sites: sorted genome-wide positions (contig offset + 1-based position of align.py:251-252), device
// pointers; 0: '+' exon starts, 1: '+' exon ends, 2: '-' starts, 3: '-' ends
struct SpliceSites {
    const int64_t* pos[4];
    int64_t n[4];
};
hipError_t launch_splice_scan(const uint8_t* codes, const uint8_t* ascii, const SpliceTask* tasks, int n, int search_extra, int shift_threshold, int canonical,
                              const SpliceSites& sites, int32_t* out, hipStream_t stream);
hipError_t launch_genome_count_n(const uint8_t* codes, const unsigned int* pre_n, const long long* off, const long long* len, long long* out, int n, hipStream_t stream);
static constexpr int kGenomeBlock = 256;     // bases per entry of the N prefix table
hipError_t launch_edit_distance(const uint8_t* seqs, const EdTask* tasks, int ntasks, int G, int planes, int32_t* out, int8_t* carry_ws, hipStream_t stream);

static constexpr int kRvStrips = 1000;   // pseudo class: RV = 32 with row strips (reads longer than 4096 bases)
extern const int kRvClasses[];
extern const int kNumRvClasses;
hipError_t launch_ssw(int rv, bool quirk, const SswParams& p, int ntasks, hipStream_t stream);
static constexpr int kRvScan = 0;        // pseudo class: K1s, the row-scan kernel for short reads in the 8-bit regime (ssw_scan.hip)
hipError_t launch_ssw_scan(bool geq, const SswParams& p, int ntasks, hipStream_t stream);
static constexpr int kRvCombine = -2;    // pseudo class: alignments whose score pass ran as window-slice tasks of an anti-diagonal class;
                                         // their "score kernel" takes the best slice (task.dir_off = first scratch row, task.pad = slices)
hipError_t launch_ssw_combine(const SswParams& p, int ntasks, hipStream_t stream);
static constexpr int kRvScanSliced = -1; // pseudo class: K1s with the forward pass cut into window slices (task.dir_off = first part, task.pad = slices)
hipError_t launch_ssw_scan_sliced(bool geq, const SswParams& p, int ntasks, int nslices, hipStream_t stream);
// the same class behind the prefilter: block minima (ssw_prefilter.hip), then seed + candidate slices, the slices by persistent
// workgroups, the finish (ssw_scan.hip)
hipError_t launch_ssw_prefilter(const SswParams& p, int nwork, hipStream_t stream);
hipError_t launch_ssw_prefilter_indel(const SswParams& p, int nworkgroups, hipStream_t stream);     // the second stage: persistent workgroups over the device-made queue
hipError_t launch_ssw_scan_filtered(bool geq, const SswParams& p, int ntasks, int nworkgroups, int nworkgroups2, hipStream_t stream);
static constexpr int kRvScanWideSliced = -4;   // pseudo class: K1w on windows of 32 kb and more: prefilter in pieces, seed, candidate tasks, best row
hipError_t launch_ssw_scanw_filtered(bool geq, const SswParams& p, int ntasks, int nworkgroups, bool with_prefilter, hipStream_t stream);
static constexpr int kRvScanWide = -3;   // pseudo class: K1w, the row-scan kernel for reads of 255..4096 bases / scores above 254 (ssw_scan_wide.hip);
                                         // persistent workgroups; their workspaces inside `dirs` (scanw_task_bytes of the class's longest read)
hipError_t launch_ssw_scanw(bool geq, const SswParams& p, int ntasks, int nworkgroups, int* counter, long long ws_off, int ws_slot, hipStream_t stream);
size_t scanw_task_bytes(int read_len);
static constexpr int kRvScanTr = -9;     // pseudo class: K1w transposed -- a reference of at most 64 columns as the ROWS, the read's bases as the columns
                                         // the lanes own (ssw_scan_wide.hip: ssw_scanw_tr_kernel); for the alignments K1l's rule would take if there were more of them
hipError_t launch_ssw_scanw_tr(bool geq, const SswParams& p, int ntasks, int nworkgroups, int* counter, long long ws_off, int ws_slot, hipStream_t stream);
// pseudo classes: K1l, one alignment per lane for references of at most 20 / 32 / 52 / 64 columns (ssw_lanes.hip)
static constexpr int kRvLanes20 = -5, kRvLanes32 = -6, kRvLanes52 = -7, kRvLanes64 = -8;
inline bool rv_is_lanes(int rv) { return rv <= kRvLanes20 && rv >= kRvLanes64; }
inline int rv_lanes_columns(int rv) { return rv == kRvLanes20 ? 20 : (rv == kRvLanes32 ? 32 : (rv == kRvLanes52 ? 52 : 64)); }
hipError_t launch_ssw_lanes(int rmax, const SswParams& p, int ntasks, hipStream_t stream);
// K1b launches.  All take the plan's WHOLE task table in p.tasks and work on the tasks [task_base, task_base + ntasks) of launch
// class `seg` (every class has its own hand-over counters and list regions, so the classes' launch chains run on different
// streams at once).  Words behind the pool's bump pointer: [4 + seg] alignments the row kernel handed to its wide form,
// [36 + seg] alignments handed on to the anti-diagonal kernel; the task indices from word 68 on, two arrays of n_total, a
// class's entries at the offset of its first task.
static constexpr int kTbMaxSeg = 32;
inline size_t tb_lists_len(int n_total) { return ((size_t)(n_total > 0 ? n_total : 1) + 1) & ~(size_t)1; }     // even: the states behind the two lists stay 16-byte aligned (read as int4)
inline size_t tb_head_bytes(int n_total) { return 4 * 68 + 2 * sizeof(int) * tb_lists_len(n_total) + 4 * sizeof(int) * (size_t)(n_total > 0 ? n_total : 1); }
// behind the two lists: per entry of the first list, where the narrow kernel left the band doubling (ssw.c:560-632):
// {band half-width of the next iteration, running maximum, iterations done, 0}
inline int* tb_state_of(unsigned long long* head, int n_total, int task_base) { return (int*)head + 68 + 2 * tb_lists_len(n_total) + 4 * (size_t)task_base; }
inline void tb_lists_of(unsigned long long* head, int n_total, int seg, int task_base, int** n_small, int** n_big, int** list_small, int** list_big)
{
    int* w = (int*)head;
    *n_small = w + 4 + seg; *n_big = w + 36 + seg;
    *list_small = w + 68 + task_base; *list_big = w + 68 + (n_total > 0 ? n_total : 1) + task_base;
}
// row form, bands up to 512 cells: every alignment of the class; what it cannot take goes on the class's first list
hipError_t launch_traceback_rows(const SswParams& p, int task_base, int ntasks, int n_total, int seg, uint8_t* pool_base, unsigned long long* pool_head,
                                 unsigned long long pool_size, hipStream_t stream);
// row form, bands up to 2048 cells / by reference column: the class's first list; what is left goes on its second list
hipError_t launch_traceback_rows_wide(const SswParams& p, int task_base, int ntasks, int n_total, int seg, uint8_t* pool_base, unsigned long long* pool_head,
                                      unsigned long long pool_size, hipStream_t stream);
// anti-diagonal form.  rv = 0: every alignment of the class with the small LDS window (what outgrows it goes on the class's
// second list); rv > 0: the class's second list, window sized for rows <= 128 * rv
hipError_t launch_traceback_pool(int rv, const SswParams& p, int task_base, int ntasks, int n_total, int seg, uint8_t* pool_base, unsigned long long* pool_head,
                                 unsigned long long pool_size, hipStream_t stream);

}  // namespace clh
