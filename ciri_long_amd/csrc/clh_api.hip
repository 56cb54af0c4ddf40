// clh_api.hip -- host side of libclh.so: contexts, batch plans, launches, and the reference's legacy symbols.
// Public interface and the reference lines each entry point replaces: include/ciri_long_hip.h, include/ssw_legacy.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ciri_long_hip.h"
#include "../../include/ssw_legacy.h"
#include "clh_device.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(CLH_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* clh_last_error(void) { return g_err.c_str(); }
extern "C" const char* clh_version(void) { return "libclh 0.1 (gfx950)"; }

extern "C" int clh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------------------
// context: device + stream + a small caching allocator (hipMalloc is milliseconds; the legacy path calls us per alignment)
// ------------------------------------------------------------------------------------------------------------
struct clh_ctx {
    int device;
    int n_cu = 256;                                      // compute units (persistent launches are sized by it)
    hipStream_t stream;
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // the read-length classes of a batch are launched on 4 streams so their tails overlap
    hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
    std::mutex mu;
    std::vector<std::pair<size_t, void*>> cache;
    int64_t last_poa_stats[16] = {0};                  // clh_ccs_plan_stats of the last clh_poa_batch (its plan lives only inside the call)

    void* alloc(size_t bytes)
    {
        if (bytes == 0) bytes = 256;
        std::lock_guard<std::mutex> g(mu);
        int best = -1;
        for (size_t i = 0; i < cache.size(); ++i)
            if (cache[i].first >= bytes && cache[i].first <= bytes * 4 + 4096 && (best < 0 || cache[i].first < cache[best].first)) best = (int)i;
        if (best >= 0) { void* p = cache[best].second; sizes.push_back({p, cache[best].first}); cache.erase(cache.begin() + best); return p; }
        size_t cap = 256;
        while (cap < bytes) cap += cap < (64u << 20) ? cap : (64u << 20);
        void* p = nullptr;
        if (hipMalloc(&p, cap) != hipSuccess) {
            // out of memory with blocks parked in the cache: give them back and try once more
            (void)hipGetLastError();
            for (auto& kv : cache) (void)hipFree(kv.second);
            cache.clear();
            if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        sizes.push_back({p, cap});
        return p;
    }
    // parked blocks are bounded: consensus workspaces are tens of GB and their size differs from batch to batch, so the
    // oldest large blocks go back to the driver once the parked total passes the cap
    static constexpr size_t kCacheCap = (size_t)128 << 30;
    void release(void* p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < sizes.size(); ++i)
            if (sizes[i].first == p) {
                cache.push_back({sizes[i].second, p});
                sizes.erase(sizes.begin() + i);
                size_t parked = 0;
                for (auto& kv : cache) parked += kv.first;
                for (size_t k = 0; parked > kCacheCap && k < cache.size();) {
                    if (cache[k].first >= ((size_t)256 << 20)) { parked -= cache[k].first; (void)hipFree(cache[k].second); cache.erase(cache.begin() + k); }
                    else ++k;
                }
                return;
            }
    }
    std::vector<std::pair<void*, size_t>> sizes;   // live allocations
};

extern "C" clh_ctx* clh_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(CLH_E_HIP, std::string("no HIP device: ") + hipGetErrorString(e)); return nullptr; }
    if (device < 0 || device >= n) { fail(CLH_E_ARG, "device index out of range"); return nullptr; }
    if ((e = hipSetDevice(device)) != hipSuccess) { fail(CLH_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); return nullptr; }
    clh_ctx* c = new clh_ctx();
    c->device = device;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->n_cu = v; }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(CLH_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        delete c;
        return nullptr;
    }
    bool ok = hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 3 && ok; ++i)
        ok = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { fail(CLH_E_HIP, "could not create side streams"); delete c; return nullptr; }
    return c;
}

extern "C" void clh_destroy(clh_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->cache) (void)hipFree(kv.second);
    for (auto& kv : c->sizes) (void)hipFree(kv.first);
    for (int i = 0; i < 3; ++i) { if (c->side[i]) (void)hipStreamDestroy(c->side[i]); if (c->join_ev[i]) (void)hipEventDestroy(c->join_ev[i]); }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int clh_device_of(const clh_ctx* c) { return c ? c->device : -1; }

// ------------------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------------------
struct clh_plan {
    clh_ctx* ctx = nullptr;
    int n = 0;
    clh_ssw_opts opts;
    clh::SswParams params;          // device pointers filled at run time
    bool quirk = false, do_cigar = false;
    std::vector<clh::SswTask> tasks;    // launch order
    struct Seg { int rv, begin, count; int64_t ws_off = 0; int ws_slot = 0, ws_wgs = 0; };      // ws_*: K1w classes, the workspaces of their persistent workgroups
    void* d_seg_ctr = nullptr;               // one work counter per segment
    std::vector<clh::ScanSlice> slices;      // window slices of the sliced scan class (one segment at most)
    void *d_slices = nullptr, *d_parts = nullptr;
    // the long-window classes behind the prefilter (ssw_prefilter.hip): [0] K1s (kRvScanSliced), [1] K1w (kRvScanWideSliced)
    struct PfClass {
        bool on = false;
        int ntasks = 0, nwork = 0, bpl = 0, cap = 0;
        void *d_win = nullptr, *d_pieces = nullptr, *d_work = nullptr, *d_dmin = nullptr, *d_queue = nullptr, *d_out = nullptr, *d_ctl = nullptr,
             *d_bound = nullptr, *d_parts = nullptr, *d_q2 = nullptr;
        int ws_row0 = 0, ws_slot = 0, ws_wgs = 0;
        int64_t ws_dirs_off = 0;
        int64_t extent = 0;                      // bytes of the refs buffer the kernel reads: the end of the last 256-byte block a window touches
        int64_t word_cols = 0, lane_insts = 0;   // the first stage's work: window columns x W words, and x (11 W + 8) instructions
        hipEvent_t ev[2] = {nullptr, nullptr};   // profiling runs: around ssw_prefilter_kernel
        bool timed = false;
    } pf[2];
    int n_rows = 0;                          // result rows: n_all + the scratch rows of the K1w long-window class
    std::vector<int32_t> slice_base;         // task-level slices of the anti-diagonal classes: first window column of scratch row k
    void* d_slice_base = nullptr;
    int n_all = 0;                           // tasks incl. those slices (their result rows sit behind the n real ones)
    std::vector<Seg> segs;
    void *d_tasks = nullptr, *d_results = nullptr, *d_colmax = nullptr, *d_cigars = nullptr, *d_cigar_len = nullptr,
         *d_pool = nullptr, *d_pool_head = nullptr, *d_reads = nullptr, *d_refs = nullptr;
    size_t colmax_elems = 0, cigar_elems = 0, strip_bytes = 0;
    unsigned long long pool_bytes = 0;
    void* d_strips = nullptr;
    hipStream_t last_stream = nullptr;
    hipEvent_t done_ev = nullptr;           // recorded behind the run's last launch: fetch waits for the RUN, not for what the caller queued later
    bool ran = false;
    bool profiling = false;
    int64_t refs_bytes = -1;                 // size of the caller's refs buffer if stated (clh_plan_set_refs_bytes), else -1
    std::vector<hipEvent_t> chain_ev;   // between the parts of a split K1w class
    std::vector<hipEvent_t> ev;     // per segment: K1 start, K1 stop; then K1b small-window start/stop, large-window start/stop
};

extern "C" void clh_plan_destroy(clh_plan* pl)
{
    if (!pl) return;
    clh_ctx* c = pl->ctx;
    (void)hipSetDevice(c->device);
    if (pl->ran) (void)hipStreamSynchronize(pl->last_stream);
    void* bufs[] = {pl->d_tasks, pl->d_results, pl->d_colmax, pl->d_cigars, pl->d_cigar_len, pl->d_pool, pl->d_pool_head,
                    pl->d_reads, pl->d_refs, pl->d_strips, pl->d_slices, pl->d_parts, pl->d_slice_base, pl->d_seg_ctr};
    for (void* b : bufs) c->release(b);
    for (auto& f : pl->pf) { void* pb[] = {f.d_win, f.d_pieces, f.d_work, f.d_dmin, f.d_queue, f.d_out, f.d_ctl, f.d_bound, f.d_parts, f.d_q2}; for (void* b : pb) c->release(b); }
    for (hipEvent_t e : pl->ev) (void)hipEventDestroy(e);
    for (auto& f : pl->pf) for (hipEvent_t e : f.ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->chain_ev) (void)hipEventDestroy(e);
    if (pl->done_ev) (void)hipEventDestroy(pl->done_ev);
    delete pl;
}

static int rv_class_for(int rows)
{
    for (int i = 0; i < clh::kNumRvClasses; ++i)
        if (128 * clh::kRvClasses[i] >= rows) return clh::kRvClasses[i];
    return clh::kRvStrips;      // longer than 4096 rows: RV = 32 kernel with row strips
}

// K1s (ssw_scan.hip) takes the alignments whose scores provably fit the reference's 8-bit pass (ssw.c:804-806 chooses it and
// it cannot overflow), with the read short enough for its (score, row) keys and the gap extension for its 16-bit frames.
// CLH_NO_SCAN=1 in the environment sends everything to the anti-diagonal kernels (A/B measurements).
// CLH_NO_TB_ROWS=1: CIGARs from the anti-diagonal traceback kernel only (A/B measurements)
static bool tb_rows_on() { static const bool off = getenv("CLH_NO_TB_ROWS") != nullptr; return !off; }

static bool scan_class_ok(int64_t L, const clh_ssw_opts* o, int max_match, int bias)
{
    static const bool off = getenv("CLH_NO_SCAN") != nullptr;
    return !off && L <= 254 && o->score_size != 1 && (int64_t)max_match * L + bias < 255 && o->gap_extend >= 0 && o->gap_extend <= 16 && o->gap_open <= 255;
}

// K1w (ssw_scan_wide.hip): the same walk for what K1s leaves -- reads up to 4096 bases, any score below the 16-bit ceiling -- on
// windows that are not cut into slices.  CLH_NO_SCANW=1 switches it off (A/B measurements): the anti-diagonal classes take over.
static bool scanw_class_ok(int64_t L, int64_t R, const clh_ssw_opts* o, int max_match)
{
    const bool off = getenv("CLH_NO_SCAN") != nullptr || getenv("CLH_NO_SCANW") != nullptr;      // (read per plan: the tests switch it)
    return !off && L <= 4096 && R < 32768 && (int64_t)max_match * L < 32000 && o->gap_extend >= 0 && o->gap_extend <= 16 && o->gap_open <= 255;
}

// K1s on long windows: the forward pass runs as slices of >= 8192 owned columns (at most 64 per alignment), each started
// `overlap` columns early (ssw_scan.hip: ssw_scan_slice_kernel).  Needs a positive gap extension (else a local alignment has
// no bounded span).  CLH_NO_SLICES=1 switches it off (A/B measurements).
static const int kSliceMinWindow = 32768, kSliceMinCols = 8192;
static bool scan_sliced(int64_t R, const clh_ssw_opts* o)
{
    static const bool off = getenv("CLH_NO_SLICES") != nullptr;
    // (experiment: CLH_PF_MIN_WINDOW lowers the window length from which K1s goes behind the prefilter -- call-path options only)
    static const int pf_min = getenv("CLH_PF_MIN_WINDOW") ? atoi(getenv("CLH_PF_MIN_WINDOW")) : kSliceMinWindow;
    if (!o->want_score2 && R >= pf_min && o->gap_extend >= 1 && !off && getenv("CLH_NO_PREFILTER") == nullptr) return true;
    return !off && R >= kSliceMinWindow && o->gap_extend >= 1;
}

// The sliced class behind the exact prefilter (ssw_prefilter.hip): needs the bound's constant c = min(max_match, gap_extend) >= 1
// and no second-best score (the column maxima of the columns it skips would be missing).  CLH_NO_PREFILTER=1 switches it off
// (A/B measurements, and the parity tests run both ways).
static bool prefilter_ok(const clh_ssw_opts* o, int max_match)
{
    return getenv("CLH_NO_PREFILTER") == nullptr && !o->want_score2 && max_match >= 1 && o->gap_extend >= 1;
}

// K1w on long windows (ssw_scan_wide.hip, class kRvScanWideSliced): what K1s's 8-bit class does not take, on windows of 32 kb and
// more with call-path options -- the prefilter in pieces of the read, a seed, candidate regions as K1w tasks.  Windows up to 1.5 Mb
// (a static slice + its overlap must stay inside K1w's 32 767 columns).  Without the prefilter (CLH_NO_PREFILTER) these alignments
// run as window-slice tasks of the anti-diagonal classes, as in rounds 2-3.
static bool scanw_sliced_ok(int64_t L, int64_t R, const clh_ssw_opts* o, int max_match)
{
    const bool off = getenv("CLH_NO_SCAN") != nullptr || getenv("CLH_NO_SCANW") != nullptr || getenv("CLH_NO_SLICES") != nullptr;
    if (off || !prefilter_ok(o, max_match)) return false;
    const int64_t overlap = L + (L * max_match + o->gap_extend - 1) / o->gap_extend + 32;
    const int64_t own = std::max<int64_t>(std::max<int64_t>(8192, 2 * overlap), (R + 63) / 64);
    return L <= 4096 && R >= kSliceMinWindow && R <= 1500000 && own + overlap < 32768 && (int64_t)max_match * L < 32000 && o->gap_extend <= 16 && o->gap_open <= 255;
}

// K1l (ssw_lanes.hip): one alignment per lane for references of at most 64 columns -- the collapse stage's junction alignments
// (collapse.py:161-173, 251-256, 373-387).  Only where its plain recurrence IS what the reference computes: no second best (the column
// maxima of the reference's wildcard rows are not computed), gap_open > gap_extend or a score that cannot leave the 8-bit regime (the 16-bit
// pass with gap_open == gap_extend truncates F at stripe boundaries, rowmajor_spec.c), code 4 scoring 0.  A lane walks its alignment alone: a long
// read against a short reference goes there only when the plan holds enough of them to fill the GPU's lanes (`many`), else to the
// wave-per-alignment classes.  CLH_NO_LANES=1 switches the class off (A/B measurements, and the parity tests run both ways).
static int lanes_class_for(int64_t L, int64_t R, const clh_ssw_opts* o, int max_match, int bias, int null_code, bool many)
{
    if (getenv("CLH_NO_LANES") != nullptr) return 0;
    if (o->want_score2 || R < 1 || R > 64 || L < 1 || L > 65535) return 0;
    if (!(o->n_mat <= 4 || null_code == 4)) return 0;
    if (o->gap_open > 255 || o->gap_extend < 0) return 0;
    if (o->gap_open <= o->gap_extend && !((int64_t)max_match * std::min(L, R) + bias < 255 && o->score_size != 1)) return 0;
    // a lane is alone with its alignment: ~25 ns per cell and pass.  Up to 2048 cells that is nothing; more only when the plan fills the GPU's lanes
    // (and then not beyond 262144 cells: a 6 ms chain)
    if ((!many && L * R > 2048) || L * R > 262144) return (L <= 32767 && !getenv("CLH_NO_SCANW") && !getenv("CLH_NO_SCAN")) ? clh::kRvScanTr : 0;      // a wave per alignment, transposed (ssw_scan_wide.hip)
    return R <= 20 ? clh::kRvLanes20 : (R <= 32 ? clh::kRvLanes32 : (R <= 52 ? clh::kRvLanes52 : clh::kRvLanes64));
}

// ref_off != nullptr: packed references, alignment a against [ref_off[a], ref_off[a+1]).  Otherwise windows of a resident
// genome: win_off[a], win_len[a], win_rc[a] (1 = read backwards and complemented).
static clh_plan* ssw_plan_build(clh_ctx* ctx, int32_t n, const int64_t* read_off, const int64_t* ref_off,
                                const int64_t* win_off, const int32_t* win_len, const uint8_t* win_rc,
                                const int32_t* mask_len, const clh_ssw_opts* o)
{
    if (!ctx || n < 0 || !read_off || (!ref_off && (!win_off || !win_len)) || !o || !o->mat) { fail(CLH_E_ARG, "clh_ssw_plan: null argument"); return nullptr; }
    if (o->n_mat < 1 || o->n_mat > 5) { fail(CLH_E_UNSUPPORTED, "substitution matrix edge must be 1..5"); return nullptr; }
    if (o->gap_open < o->gap_extend) {
        fail(CLH_E_UNSUPPORTED, "gap_open < gap_extend: the reference's 8-bit lazy-F loop is not a plain recurrence there; not implemented");
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(CLH_E_HIP, "hipSetDevice failed"); return nullptr; }
    clh_plan* pl = new clh_plan();
    pl->ctx = ctx; pl->n = n; pl->opts = *o;
    clh::SswParams& P = pl->params;
    memset(&P, 0, sizeof(P));
    int mn = 0, mx = -128;
    for (int i = 0; i < o->n_mat * o->n_mat; ++i) { P.mat[i] = o->mat[i]; mn = std::min(mn, (int)o->mat[i]); mx = std::max(mx, (int)o->mat[i]); }
    P.n = o->n_mat; P.gapO = o->gap_open; P.gapE = o->gap_extend; P.bias = -mn; P.max_match = mx; P.score_size = o->score_size;
    P.flag = o->flag; P.filters = o->filters; P.filterd = o->filterd;
    P.null_code = 5;
    if (o->n_mat == 5) {
        bool z = true;
        for (int i = 0; i < 5; ++i) z = z && o->mat[4 * 5 + i] == 0 && o->mat[i * 5 + 4] == 0;
        if (z) P.null_code = 4;
    }
    pl->opts.mat = nullptr;
    pl->quirk = o->gap_open <= o->gap_extend;
    pl->do_cigar = o->want_cigar && (o->flag & 7) != 0;
    if (!(o->score_size == 0 || o->score_size == 1 || o->score_size == 2)) {
        fail(CLH_E_ARG, "Please call the function ssw_init before ssw_align.");   // ssw.c:818-821
        delete pl; return nullptr;
    }

    std::vector<int> cls(n);
    pl->tasks.resize(n);
    size_t colmax = 0, cig = 0;
    unsigned long long pool = 0;
    int n_short_ref = 0;                     // alignments K1l could take: enough of them fill the GPU's lanes whatever their read length
    for (int a = 0; a < n; ++a) {
        const int64_t L = read_off[a + 1] - read_off[a], R = ref_off ? ref_off[a + 1] - ref_off[a] : (int64_t)win_len[a];
        n_short_ref += lanes_class_for(L, R, o, mx, -mn, P.null_code, true) != 0;
    }
    const bool many_short_ref = n_short_ref >= 32768;
    for (int a = 0; a < n; ++a) {
        const int64_t L = read_off[a + 1] - read_off[a], R = ref_off ? ref_off[a + 1] - ref_off[a] : (int64_t)win_len[a];
        if (L < 1 || R < 0 || L > 0x7fffffff || R > 0x7fffffff) { fail(CLH_E_ARG, "empty read or negative length in batch"); delete pl; return nullptr; }
        const int rc = (!ref_off && win_rc && win_rc[a]) ? 1 : 0;
        const int rows = (int)((L + 15) / 16) * 16;
        const int lanes_rv = lanes_class_for(L, R, o, mx, -mn, P.null_code, many_short_ref);
        const int rv = lanes_rv ? lanes_rv : scan_class_ok(L, o, mx, -mn) ? (scan_sliced(R, o) ? clh::kRvScanSliced : clh::kRvScan)
                                                    : (scanw_class_ok(L, R, o, mx) ? clh::kRvScanWide
                                                       : (scanw_sliced_ok(L, R, o, mx) ? clh::kRvScanWideSliced : rv_class_for(rows)));
        cls[a] = rv;
        clh::SswTask& t = pl->tasks[a];
        t.read_off = read_off[a]; t.ref_off = ref_off ? ref_off[a] : (rc ? win_off[a] + R - 1 : win_off[a]);
        t.ref_rc = rc; t.pad = 0;
        t.read_len = (int)L; t.ref_len = (int)R;
        t.mask_len = mask_len ? mask_len[a] : (L > 30 ? (int)(L / 2) : 15);
        t.out_index = a;
        t.colmax_off = (int64_t)colmax;
        if (o->want_score2) colmax += (size_t)((R + 7) & ~7ll);
        t.cigar_off = (int32_t)cig; t.cigar_cap = 0; t.dir_off = 0;
        if (rv == clh::kRvStrips) {     // two boundary buffers of (reference length rounded up to 64) x 8 bytes
            t.dir_off = (int64_t)pl->strip_bytes;
            pl->strip_bytes += 2 * (size_t)((R + 63) & ~63ll) * 8 + 256;
        }
        if (pl->do_cigar) {
            t.cigar_cap = (int32_t)(2 * L + 2);
            if (cig + (size_t)t.cigar_cap > 0x7fffffffull) { fail(CLH_E_CAPACITY, "batch too large for 32-bit CIGAR offsets; split it"); delete pl; return nullptr; }
            cig += (size_t)t.cigar_cap;
            // traceback workspace: the row kernel keeps 4 bits per cell of the band, 64 bytes per read row for bands up to
            // 128 cells, 128 up to 256 (ssw_traceback_rows.hip), of every iteration after the first (each twice the one
            // before); the few wide bands and the anti-diagonal kernel's per-iteration byte planes come out of the fixed slack added below
            pool += (unsigned long long)(L + 64) * 320ull;
        }
    }
    // Long windows of the anti-diagonal classes (reads outside K1s's 8-bit class), call-path options (no second best): the
    // alignment is cut into window slices that run as ordinary tasks of their class -- whole alignments of the read against
    // [c_begin, c_end), results into scratch rows behind the n real ones -- and a combining kernel (class kRvCombine, after
    // all classes have finished) takes the best slice: largest score, then smallest end column, then the earlier slice.
    // Exact: a local alignment spans at most L (1 + max_match / gap_extend) columns, so the slice that OWNS the end column
    // (started that far before its owned columns) computes the whole-window H there; a later slice that sees the same cell
    // in its overlap can only underestimate, and loses the tie.  The reverse pass of the owning slice runs inside the slice.
    int n_all = n;
    if (!o->want_score2 && o->gap_extend >= 1 && !getenv("CLH_NO_SLICES")) {
        for (int a = 0; a < n; ++a) {
            if (cls[a] < 1 || cls[a] == clh::kRvStrips || pl->tasks[a].ref_len < kSliceMinWindow) continue;
            const clh::SswTask par = pl->tasks[a];
            const int64_t R = par.ref_len, L = par.read_len;
            const int64_t overlap = L + (L * mx + o->gap_extend - 1) / o->gap_extend + 32;
            const int64_t own = std::max<int64_t>(std::max<int64_t>(kSliceMinCols, 2 * overlap), (R + 63) / 64);
            const int rdir = par.ref_rc ? -1 : 1;
            pl->tasks[a].dir_off = (int64_t)pl->slice_base.size();       // first scratch row of this alignment (relative to n)
            int ns = 0;
            for (int64_t b = 0; b < R; b += own, ++ns) {
                clh::SswTask t = par;
                const int64_t cb = std::max<int64_t>(0, b - overlap), ce = std::min<int64_t>(R, b + own);
                t.ref_off = par.ref_off + cb * rdir; t.ref_len = (int32_t)(ce - cb);
                t.out_index = n + (int32_t)pl->slice_base.size();
                t.cigar_cap = 0; t.pad = 0; t.dir_off = 0;
                pl->slice_base.push_back((int32_t)cb);
                pl->tasks.push_back(t);
                cls.push_back(cls[a]);
            }
            pl->tasks[a].pad = ns;
            cls[a] = clh::kRvCombine;
        }
        n_all = (int)pl->tasks.size();
    }
    pl->n_all = n_all;
    P.n_real = n;
    pl->colmax_elems = colmax; pl->cigar_elems = cig;
    if (pl->do_cigar) pl->pool_bytes = std::min<unsigned long long>(pool + (1024ull << 20), 16ull << 30);

    // launch order: by row class, heaviest alignments first inside a class
    std::vector<int> order(n_all);
    for (int a = 0; a < n_all; ++a) order[a] = a;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        if (cls[x] != cls[y]) return cls[x] < cls[y];
        const int64_t wx = (int64_t)pl->tasks[x].read_len * pl->tasks[x].ref_len, wy = (int64_t)pl->tasks[y].read_len * pl->tasks[y].ref_len;
        return wx > wy;
    });
    std::vector<clh::SswTask> sorted(n_all);
    for (int k = 0; k < n_all; ++k) sorted[k] = pl->tasks[order[k]];
    for (int k = 0; k < n_all;) {
        int e = k;
        while (e < n_all && cls[order[e]] == cls[order[k]]) ++e;
        // a large K1w class with CIGARs goes as up to four launches: the traceback of one runs under the score kernel of the next
        // (clh_ssw_run) instead of all of it behind the one score kernel
        const int parts = (cls[order[k]] == clh::kRvScanWide && pl->do_cigar && e - k >= 4096 && getenv("CLH_SCANW_PARTS")) ? std::max(1, std::min(8, atoi(getenv("CLH_SCANW_PARTS")))) : 1;   // (measured on C2, 10 000 alignments: 4 parts of 2 500 are each less than one round of the GPU's wave slots -- 28 -> 35 ms; kept for batches far above that, off by default)
        for (int q = 0; q < parts; ++q) {
            const int b = k + (int)((int64_t)(e - k) * q / parts), b2 = k + (int)((int64_t)(e - k) * (q + 1) / parts);
            clh_plan::Seg sg; sg.rv = cls[order[k]]; sg.begin = b; sg.count = b2 - b;
            pl->segs.push_back(sg);
        }
        k = e;
    }
    pl->tasks.swap(sorted);
    pl->n_rows = n_all;
    for (auto& sg : pl->segs) {
        if (sg.rv != clh::kRvScanWide && sg.rv != clh::kRvScanTr) continue;
        int lmax = 1;
        for (int k = 0; k < sg.count; ++k) lmax = std::max(lmax, sg.rv == clh::kRvScanTr ? 64 : (int)pl->tasks[sg.begin + k].read_len);
        sg.ws_slot = (int)((clh::scanw_task_bytes(lmax) + 255) & ~(size_t)255);
        sg.ws_wgs = std::min(sg.count, ctx->n_cu * 12);
        pl->strip_bytes = (pl->strip_bytes + 255) & ~(size_t)255;
        sg.ws_off = (int64_t)pl->strip_bytes;
        pl->strip_bytes += (size_t)sg.ws_slot * (size_t)sg.ws_wgs;
    }
    pl->d_seg_ctr = ctx->alloc(sizeof(int) * std::max<size_t>(pl->segs.size(), 1));
    for (const auto& sg : pl->segs) {
        const int ci = sg.rv == clh::kRvScanSliced ? 0 : (sg.rv == clh::kRvScanWideSliced ? 1 : -1);
        if (ci < 0 || !prefilter_ok(o, mx)) continue;
        // per alignment: the 256-byte blocks of the refs buffer its window touches, in processing order (minus-strand windows run
        // down the addresses); per piece of its read (<= 254 rows) one run of block minima; a lane of the prefilter owns bpl blocks
        clh_plan::PfClass& f = pl->pf[ci];
        std::vector<clh::PfWin> wins(sg.count);
        std::vector<clh::PfTask> pieces;
        std::vector<clh::PfWork> work;
        int64_t total = 0, cap = 0, dtot = 0;
        int lmax = 1;
        for (int k = 0; k < sg.count; ++k) {
            const clh::SswTask& t = pl->tasks[sg.begin + k];
            clh::PfWin& w = wins[k];
            memset(&w, 0, sizeof(w));
            const int64_t R = t.ref_len;
            if (t.ref_rc) { const int64_t hi = t.ref_off, lo = hi - R + 1; w.mem_block0 = (int32_t)(hi >> 8); w.phase = (int32_t)(255 - (hi & 255)); w.nsub = (int32_t)((hi >> 8) - (lo >> 8) + 1); }
            else { const int64_t lo = t.ref_off, hi = lo + R - 1; w.mem_block0 = (int32_t)(lo >> 8); w.phase = (int32_t)(lo & 255); w.nsub = (int32_t)((hi >> 8) - (lo >> 8) + 1); }
            f.extent = std::max<int64_t>(f.extent, (((t.ref_rc ? t.ref_off : t.ref_off + R - 1) >> 8) + 1) << 8);
            const int L = t.read_len, K = (L + 253) / 254;
            w.piece_first = (int32_t)pieces.size(); w.piece_count = K; w.d_off = (int32_t)dtot;
            dtot += w.nsub;
            for (int q = 0, r0 = 0; q < K; ++q) {
                const int rows = L / K + (q < L % K ? 1 : 0);
                pieces.push_back({k, r0, rows, (int32_t)total});
                r0 += rows; total += w.nsub;
                const int64_t Wq = (rows + 31) / 32;
                f.word_cols += R * Wq; f.lane_insts += R * (11 * Wq + 8);
            }
            cap += ci == 0 ? std::max<int64_t>(64, w.nsub / 8 + 1) : 2 * 64;
            lmax = std::max(lmax, L);
            if (total > 0x7fffffffll || cap > 0x3fffffffll) { fail(CLH_E_CAPACITY, "batch too large for the prefilter's 32-bit block offsets; split it"); delete pl; return nullptr; }
        }
        int bpl = (int)std::min<int64_t>(16, std::max<int64_t>(4, total / (64 * 12288)));
        if (const char* e = getenv("CLH_PF_BPL")) bpl = std::max(1, atoi(e));
        for (int q = 0; q < (int)pieces.size(); ++q) {
            clh::PfWin& w = wins[pieces[q].task];
            if (q == w.piece_first) w.work_first = (int32_t)work.size();
            for (int fb = 0; fb < w.nsub; fb += 64 * bpl) work.push_back({q, fb});
            w.work_count = (int32_t)work.size() - w.work_first;
        }
        f.on = true; f.ntasks = sg.count; f.bpl = bpl; f.cap = (int)cap; f.nwork = (int)work.size();
        f.d_win = ctx->alloc(sizeof(clh::PfWin) * wins.size());
        f.d_pieces = ctx->alloc(sizeof(clh::PfTask) * pieces.size());
        f.d_work = ctx->alloc(sizeof(clh::PfWork) * std::max<size_t>(work.size(), 1));
        f.d_dmin = ctx->alloc((size_t)total + 64);
        f.d_out = ctx->alloc(sizeof(clh::PfOut) * wins.size());
        f.d_ctl = ctx->alloc(sizeof(clh::PfCtl));
        bool ok = f.d_win && f.d_pieces && f.d_work && f.d_dmin && f.d_out && f.d_ctl;
        if (ci == 0) {
            f.d_queue = ctx->alloc(sizeof(clh::ScanSlice) * (size_t)cap);
            f.d_parts = ctx->alloc(sizeof(clh::ScanPart) * (size_t)cap);
            ok = ok && f.d_queue && f.d_parts;
            if (!getenv("CLH_NO_PF2")) {       // the second stage's queue: at most every entry of the work list (A/B: one stage only)
                f.d_q2 = ctx->alloc(sizeof(int32_t) * std::max<size_t>(work.size(), 1));
                ok = ok && f.d_q2;
            }
        } else {
            // 2 seed tasks per alignment in fixed places, then the candidate queue; 130 scratch result rows per alignment behind every other
            // row; the K1w workspaces belong to the persistent workgroups, not to the tasks
            f.d_queue = ctx->alloc(sizeof(clh::WsTask) * ((size_t)2 * sg.count + (size_t)cap));
            f.d_bound = ctx->alloc(sizeof(uint16_t) * (size_t)(dtot + 64));
            ok = ok && f.d_queue && f.d_bound;
            f.ws_row0 = pl->n_rows;
            pl->n_rows += clh::kWsRows * sg.count;
            f.ws_slot = (int)((clh::scanw_task_bytes(lmax) + 255) & ~(size_t)255);
            f.ws_wgs = (int)std::min<int64_t>((int64_t)2 * sg.count + cap, (int64_t)ctx->n_cu * 12);
            pl->strip_bytes = (pl->strip_bytes + 255) & ~(size_t)255;
            f.ws_dirs_off = (int64_t)pl->strip_bytes;
            pl->strip_bytes += (size_t)f.ws_slot * (size_t)f.ws_wgs;
        }
        if (!ok || hipMemcpy(f.d_win, wins.data(), sizeof(clh::PfWin) * wins.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f.d_pieces, pieces.data(), sizeof(clh::PfTask) * pieces.size(), hipMemcpyHostToDevice) != hipSuccess ||
            (!work.empty() && hipMemcpy(f.d_work, work.data(), sizeof(clh::PfWork) * work.size(), hipMemcpyHostToDevice) != hipSuccess) ||
            hipMemset(f.d_ctl, 0, sizeof(clh::PfCtl)) != hipSuccess) {
            fail(CLH_E_HIP, "out of device memory while building the plan");
            clh_plan_destroy(pl); return nullptr;
        }
    }
    for (const auto& sg : pl->segs) {
        if (sg.rv != clh::kRvScanSliced || pl->pf[0].on) continue;
        for (int k = 0; k < sg.count; ++k) {
            clh::SswTask& t = pl->tasks[sg.begin + k];
            const int64_t R = t.ref_len, L = t.read_len;
            const int64_t overlap = L + (L * mx + o->gap_extend - 1) / o->gap_extend + 32;    // span of a local alignment + wildcard rows + slack
            const int64_t own = std::max<int64_t>(kSliceMinCols, (R + 63) / 64);
            t.dir_off = (int64_t)pl->slices.size();
            int ns = 0;
            for (int64_t b = 0; b < R; b += own, ++ns) {
                clh::ScanSlice sl;
                memset(&sl, 0, sizeof(sl));
                sl.task = k; sl.own_begin = (int32_t)b; sl.c_begin = (int32_t)std::max<int64_t>(0, b - overlap); sl.c_end = (int32_t)std::min<int64_t>(R, b + own);
                sl.part = (int32_t)pl->slices.size();
                pl->slices.push_back(sl);
            }
            t.pad = ns;
        }
    }

    pl->d_tasks = ctx->alloc(sizeof(clh::SswTask) * (size_t)std::max(n_all, 1));
    pl->d_results = ctx->alloc(sizeof(clh::SswResult) * (size_t)std::max(pl->n_rows, 1));
    pl->d_cigar_len = ctx->alloc(sizeof(int32_t) * (size_t)std::max(n_all, 1));
    if (!pl->slice_base.empty()) {
        pl->d_slice_base = ctx->alloc(sizeof(int32_t) * pl->slice_base.size());
        if (!pl->d_slice_base || hipMemcpy(pl->d_slice_base, pl->slice_base.data(), sizeof(int32_t) * pl->slice_base.size(), hipMemcpyHostToDevice) != hipSuccess) {
            fail(CLH_E_HIP, "out of device memory while building the plan");
            clh_plan_destroy(pl); return nullptr;
        }
    }
    if (o->want_score2) pl->d_colmax = ctx->alloc(sizeof(uint16_t) * std::max<size_t>(colmax, 1));
    if (pl->do_cigar) {
        pl->d_cigars = ctx->alloc(sizeof(uint32_t) * std::max<size_t>(cig, 1));
        pl->d_pool = ctx->alloc((size_t)pl->pool_bytes);
        pl->d_pool_head = ctx->alloc(clh::tb_head_bytes(n_all));
    }
    if (pl->strip_bytes) pl->d_strips = ctx->alloc(pl->strip_bytes);
    if (!pl->slices.empty()) {
        pl->d_slices = ctx->alloc(sizeof(clh::ScanSlice) * pl->slices.size());
        pl->d_parts = ctx->alloc(sizeof(clh::ScanPart) * pl->slices.size());
        if (!pl->d_slices || !pl->d_parts || hipMemcpy(pl->d_slices, pl->slices.data(), sizeof(clh::ScanSlice) * pl->slices.size(), hipMemcpyHostToDevice) != hipSuccess) {
            fail(CLH_E_HIP, "out of device memory while building the plan");
            clh_plan_destroy(pl); return nullptr;
        }
    }
    if (!pl->d_tasks || !pl->d_results || !pl->d_cigar_len || !pl->d_seg_ctr || (pl->strip_bytes && !pl->d_strips) || (o->want_score2 && !pl->d_colmax) ||
        (pl->do_cigar && (!pl->d_cigars || !pl->d_pool || !pl->d_pool_head))) {
        fail(CLH_E_HIP, "out of device memory while building the plan");
        clh_plan_destroy(pl); return nullptr;
    }
    if (n_all > 0 && hipMemcpy(pl->d_tasks, pl->tasks.data(), sizeof(clh::SswTask) * (size_t)n_all, hipMemcpyHostToDevice) != hipSuccess) {
        fail(CLH_E_HIP, "task upload failed");
        clh_plan_destroy(pl); return nullptr;
    }
    return pl;
}

extern "C" clh_plan* clh_ssw_plan(clh_ctx* ctx, int32_t n, const int64_t* read_off, const int64_t* ref_off,
                                  const int32_t* mask_len, const clh_ssw_opts* o)
{
    if (!ref_off) { fail(CLH_E_ARG, "clh_ssw_plan: null argument"); return nullptr; }
    return ssw_plan_build(ctx, n, read_off, ref_off, nullptr, nullptr, nullptr, mask_len, o);
}

extern "C" clh_plan* clh_ssw_plan_windows(clh_ctx* ctx, int32_t n, const int64_t* read_off, const int64_t* win_off, const int32_t* win_len,
                                          const uint8_t* win_rc, const int32_t* mask_len, const clh_ssw_opts* o)
{
    if (!win_off || !win_len) { fail(CLH_E_ARG, "clh_ssw_plan_windows: null argument"); return nullptr; }
    return ssw_plan_build(ctx, n, read_off, nullptr, win_off, win_len, win_rc, mask_len, o);
}

// ---------------------------------------------------------------------------------------------------------------
// K5: resident genome
// ---------------------------------------------------------------------------------------------------------------
struct clh_genome {
    clh_ctx* ctx = nullptr;
    int64_t len = 0;
    void *d_codes = nullptr, *d_pre = nullptr;
    void* d_ascii = nullptr;                    // the characters themselves (K6 compares flanks as the reference's strings do)
    void* d_sites = nullptr;                    // annotated splice sites (K6): four sorted runs of int64 in one buffer
    int64_t n_sites[4] = {0, 0, 0, 0};
};

extern "C" void clh_genome_destroy(clh_genome* g)
{
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    g->ctx->release(g->d_codes); g->ctx->release(g->d_pre); g->ctx->release(g->d_sites); g->ctx->release(g->d_ascii);
    delete g;
}

extern "C" clh_genome* clh_genome_create(clh_ctx* ctx, const char* ascii, int64_t len)
{
    if (!ctx || len < 0 || (len > 0 && !ascii)) { fail(CLH_E_ARG, "clh_genome_create: null argument"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(CLH_E_HIP, "hipSetDevice failed"); return nullptr; }
    clh_genome* g = new clh_genome();
    g->ctx = ctx; g->len = len;
    const size_t nblk = (size_t)(len / clh::kGenomeBlock) + 2;
    g->d_codes = ctx->alloc((size_t)len + 64);
    g->d_pre = ctx->alloc(sizeof(unsigned int) * nblk);
    g->d_ascii = ctx->alloc((size_t)len + 64);
    void* d_ascii = g->d_ascii;
    bool ok = g->d_codes && g->d_pre && d_ascii;
    if (!ok) fail(CLH_E_HIP, "out of device memory for the genome");
    std::vector<unsigned int> pre(nblk, 0);
    if (ok && len > 0) {
        ok = hipMemsetAsync(g->d_pre, 0, sizeof(unsigned int) * nblk, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(d_ascii, ascii, (size_t)len, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
             clh::launch_genome_encode((const char*)d_ascii, (uint8_t*)g->d_codes, (unsigned int*)g->d_pre, len, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(pre.data(), g->d_pre, sizeof(unsigned int) * nblk, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (!ok) fail(CLH_E_HIP, "genome encode failed");
    }
    if (ok) {   // exclusive prefix: pre[b] = upper-case N before base b * kGenomeBlock
        unsigned int run = 0;
        for (size_t b = 0; b < nblk; ++b) { const unsigned int c = pre[b]; pre[b] = run; run += c; }
        ok = hipMemcpyAsync(g->d_pre, pre.data(), sizeof(unsigned int) * nblk, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (!ok) fail(CLH_E_HIP, "genome prefix upload failed");
    }
    if (!ok) { clh_genome_destroy(g); return nullptr; }
    return g;
}

extern "C" const void* clh_genome_codes(const clh_genome* g) { return g ? g->d_codes : nullptr; }
extern "C" int64_t clh_genome_length(const clh_genome* g) { return g ? g->len : 0; }

extern "C" int clh_genome_count_n(clh_genome* g, int32_t n, const int64_t* off, const int64_t* len, int64_t* out)
{
    if (!g || n < 0 || (n > 0 && (!off || !len || !out))) return fail(CLH_E_ARG, "clh_genome_count_n: null argument");
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i) if (off[i] < 0 || len[i] < 0 || off[i] + len[i] > g->len) return fail(CLH_E_ARG, "clh_genome_count_n: window outside the genome");
    clh_ctx* ctx = g->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    void* d_off = ctx->alloc(sizeof(int64_t) * (size_t)n); void* d_len = ctx->alloc(sizeof(int64_t) * (size_t)n); void* d_out = ctx->alloc(sizeof(int64_t) * (size_t)n);
    int rc = 0;
    if (!d_off || !d_len || !d_out) rc = fail(CLH_E_HIP, "out of device memory");
    if (!rc && (hipMemcpyAsync(d_off, off, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(d_len, len, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                clh::launch_genome_count_n((const uint8_t*)g->d_codes, (const unsigned int*)g->d_pre, (const long long*)d_off, (const long long*)d_len,
                                           (long long*)d_out, n, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(out, d_out, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = fail(CLH_E_HIP, "N count failed");
    if (rc) (void)hipStreamSynchronize(ctx->stream);      // nothing queued may still touch the buffers when they go back to the cache
    ctx->release(d_off); ctx->release(d_len); ctx->release(d_out);
    return rc;
}

// K6: annotated splice sites of the resident genome (the SS_INDEX of align.py:235-252, 275-316) as four sorted runs
extern "C" int clh_genome_set_splice_sites(clh_genome* g, const int64_t* pos, const int64_t* count4)
{
    if (!g || !count4) return fail(CLH_E_ARG, "clh_genome_set_splice_sites: null argument");
    int64_t tot = 0;
    for (int k = 0; k < 4; ++k) { if (count4[k] < 0) return fail(CLH_E_ARG, "clh_genome_set_splice_sites: negative count"); tot += count4[k]; }
    if (tot > 0 && !pos) return fail(CLH_E_ARG, "clh_genome_set_splice_sites: null argument");
    int64_t at = 0;
    for (int k = 0; k < 4; ++k) {
        for (int64_t i = 0; i < count4[k]; ++i) {
            const int64_t v = pos[at + i];
            if (v < 0 || v > g->len || (i > 0 && v <= pos[at + i - 1])) return fail(CLH_E_ARG, "clh_genome_set_splice_sites: positions must be strictly ascending and inside the genome");
        }
        at += count4[k];
    }
    clh_ctx* ctx = g->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->release(g->d_sites); g->d_sites = nullptr;
    for (int k = 0; k < 4; ++k) g->n_sites[k] = 0;
    if (tot == 0) return 0;
    g->d_sites = ctx->alloc(sizeof(int64_t) * (size_t)tot);
    if (!g->d_sites) return fail(CLH_E_HIP, "out of device memory");
    if (hipMemcpyAsync(g->d_sites, pos, sizeof(int64_t) * (size_t)tot, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        ctx->release(g->d_sites); g->d_sites = nullptr;
        return fail(CLH_E_HIP, "splice-site upload failed");
    }
    for (int k = 0; k < 4; ++k) g->n_sites[k] = count4[k];
    return 0;
}

// K6: splice signals around n candidate junctions of the resident genome (align.py:474-733)
extern "C" int clh_splice_signal_batch(clh_genome* g, int32_t n, const int64_t* ctg_off, const int64_t* ctg_len, const int64_t* start,
                                       const int64_t* end, const int32_t* clip_base, const int32_t* host_mask, int32_t search_extra,
                                       int32_t shift_threshold, int32_t is_canonical, int32_t* out)
{
    if (!g || n < 0 || (n > 0 && (!ctg_off || !ctg_len || !start || !end || !clip_base || !host_mask || !out)))
        return fail(CLH_E_ARG, "clh_splice_signal_batch: null argument");
    if (n == 0) return 0;
    if (search_extra < 0 || shift_threshold < 0) return fail(CLH_E_ARG, "clh_splice_signal_batch: negative search length");
    std::vector<clh::SpliceTask> tasks((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (ctg_off[i] < 0 || ctg_len[i] < 0 || ctg_off[i] + ctg_len[i] > g->len) return fail(CLH_E_ARG, "clh_splice_signal_batch: contig outside the genome");
        if (clip_base[i] < 0 || clip_base[i] > 4096) return fail(CLH_E_ARG, "clh_splice_signal_batch: clip_base out of range");
        clh::SpliceTask& t = tasks[(size_t)i];
        t.ctg_off = ctg_off[i]; t.ctg_len = ctg_len[i]; t.start = start[i]; t.end = end[i]; t.clip_base = clip_base[i]; t.host_mask = host_mask[i];
    }
    clh_ctx* ctx = g->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t tb = sizeof(clh::SpliceTask) * (size_t)n, ob = sizeof(int32_t) * 8 * (size_t)n;
    void* d_t = ctx->alloc(tb); void* d_o = ctx->alloc(ob);
    clh::SpliceSites sites;
    int64_t at = 0;
    for (int k = 0; k < 4; ++k) { sites.pos[k] = (const int64_t*)g->d_sites + at; sites.n[k] = g->n_sites[k]; at += g->n_sites[k]; }
    int rc = 0;
    if (!d_t || !d_o) rc = fail(CLH_E_HIP, "out of device memory");
    if (!rc && (hipMemcpyAsync(d_t, tasks.data(), tb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                clh::launch_splice_scan((const uint8_t*)g->d_codes, (const uint8_t*)g->d_ascii, (const clh::SpliceTask*)d_t, n, search_extra, shift_threshold, is_canonical & 3,
                                        sites, (int32_t*)d_o, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(out, d_o, ob, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = fail(CLH_E_HIP, "splice-signal scan failed");
    if (rc) (void)hipStreamSynchronize(ctx->stream);      // see clh_genome_count_n
    ctx->release(d_t); ctx->release(d_o);
    return rc;
}

// Plans of one context may be in flight on the device at the same time (bench.py's C2 loop alternates two), but they share the context's
// side streams and fork/join events: clh_ssw_run must not be called from two host threads at once for plans of the same context.
extern "C" int clh_ssw_run(clh_plan* pl, const void* d_reads, const void* d_refs, void* stream_)
{
    if (!pl || !d_reads || !d_refs) return fail(CLH_E_ARG, "clh_ssw_run: null argument");
    HIPCHK(hipSetDevice(pl->ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : pl->ctx->stream;
    clh::SswParams P = pl->params;
    P.reads = (const int8_t*)d_reads; P.refs = (const int8_t*)d_refs;
    P.results = (clh::SswResult*)pl->d_results;
    P.colmax = (uint16_t*)pl->d_colmax;
    P.cigars = (uint32_t*)pl->d_cigars; P.cigar_len = (int32_t*)pl->d_cigar_len; P.dirs = (uint8_t*)pl->d_strips;
    P.no_guess = getenv("CLH_NO_GUESS") != nullptr;
    if (pl->profiling && pl->ev.empty()) {
        pl->ev.resize(pl->segs.size() * 2 + 4);
        for (auto& e : pl->ev) HIPCHK(hipEventCreate(&e));
    }
    // One launch per read-length class for K1, followed on the same stream by that class's small-window traceback.
    // Classes are independent, so outside profiling runs they go round-robin to the caller's stream and three side
    // streams (fork/join with events): the tail of one class overlaps the next, and the latency-bound traceback of
    // one class overlaps the VALU-bound score kernel of another.
    clh_ctx* c = pl->ctx;
    const bool tb = pl->do_cigar && pl->n > 0;
    const bool fan = !pl->profiling && pl->segs.size() > 1;
    const size_t eb = pl->segs.size() * 2;
    if (tb) HIPCHK(hipMemsetAsync(pl->d_pool_head, 0, 4 * 68, st));     // bump pointer and the classes' hand-over counters
    if (fan) {
        HIPCHK(hipEventRecord(c->fork_ev, st));
        for (int i = 0; i < 3; ++i) HIPCHK(hipStreamWaitEvent(c->side[i], c->fork_ev, 0));
    }
    // launch order: the class whose single alignments take longest first (a pass is a serial chain of reference columns,
    // longer per column the more rows a lane holds), the short-read scan class last: its many small workgroups would
    // otherwise fill every slot of the GPU and the long chains would start only when they drain
    std::vector<size_t> ord(pl->segs.size());
    for (size_t k = 0; k < ord.size(); ++k) ord[k] = k;
    std::sort(ord.begin(), ord.end(), [&](size_t x, size_t y) {
        auto weight = [](int rv) { return rv == clh::kRvStrips ? 1000 : (rv == clh::kRvScanWide ? 900 : (rv == clh::kRvScanWideSliced ? 800 : (rv == clh::kRvScanTr ? 700 : rv))); };      // (K1w: whole long reads, one wave each)
        const int rx = weight(pl->segs[x].rv), ry = weight(pl->segs[y].rv);
        return rx > ry;
    });
    clh::SswParams PG = P;                                  // the traceback launches index the whole task table
    PG.tasks = (const clh::SswTask*)pl->d_tasks;
    uint8_t* const pool = (uint8_t*)pl->d_pool;
    unsigned long long* const head = (unsigned long long*)pl->d_pool_head;
    // CIGARs of one class, all on one stream: the row kernel, its wide form for what it hands over, the anti-diagonal
    // kernel for what is left (sized for the class; workgroups share the short lists).  Per class, so that the seconds-long
    // tail of a few wide-band alignments of one class runs under the score kernels of the others.
    auto traceback = [&](int first, int count, int seg, int rv, hipStream_t ls) -> int {
        // the anti-diagonal fallback stages both aligned sequences in LDS: always the largest configuration (12 kB of sequence,
        // 4098 rows) -- a short read can align against thousands of reference bases; its few workgroups share the list
        const int rvbig = 32; (void)rv;
        if (tb_rows_on()) {
            HIPCHK(clh::launch_traceback_rows(PG, first, count, pl->n_all, seg, pool, head, pl->pool_bytes, ls));
            HIPCHK(clh::launch_traceback_rows_wide(PG, first, count, pl->n_all, seg, pool, head, pl->pool_bytes, ls));
        } else HIPCHK(clh::launch_traceback_pool(0, PG, first, count, pl->n_all, seg, pool, head, pl->pool_bytes, ls));
        HIPCHK(clh::launch_traceback_pool(rvbig, PG, first, count, pl->n_all, seg, pool, head, pl->pool_bytes, ls));
        return 0;
    };
    // the first stage of the prefilter, with its own pair of events in profiling runs (clh_plan_prefilter_timing)
    auto prefilter = [&](clh_plan::PfClass& f, hipStream_t ls) -> int {
        f.timed = false;
        if (!P.pf_dmin) return 0;
        if (pl->profiling) {
            for (auto& e : f.ev) if (!e) HIPCHK(hipEventCreate(&e));
            HIPCHK(hipEventRecord(f.ev[0], ls));
        }
        HIPCHK(clh::launch_ssw_prefilter(P, f.nwork, ls));
        if (pl->profiling) { HIPCHK(hipEventRecord(f.ev[1], ls)); f.timed = true; }
        return 0;
    };
    P.slice_base = (const int32_t*)pl->d_slice_base; PG.slice_base = P.slice_base;
    int rv_all = 4;                                              // longest read class of the plan (the combined alignments have any length)
    for (const auto& s : pl->segs) rv_all = std::max(rv_all, s.rv == clh::kRvStrips ? 32 : s.rv);
    int n_wide = 0, i_wide = 0;
    for (const auto& s : pl->segs) n_wide += s.rv == clh::kRvScanWide;
    for (size_t q = 0; q < ord.size(); ++q) {
        const size_t k = ord[q];
        const auto& s = pl->segs[k];
        if (s.rv == clh::kRvCombine) continue;                   // after the join below: it reads the other classes' rows
        hipStream_t ls = (fan && (q & 3)) ? c->side[(q & 3) - 1] : st;
        // the parts of a K1w class: score kernels one after the other on the main stream, the traceback of each part on a side
        // stream behind an event (the last part's stays on the main stream)
        const bool chained = fan && tb && s.rv == clh::kRvScanWide && n_wide > 1;
        if (chained) ls = st;
        P.tasks = (const clh::SswTask*)pl->d_tasks + s.begin;
        if (pl->profiling) HIPCHK(hipEventRecord(pl->ev[2 * k + 0], ls));
        if ((s.rv == clh::kRvScanSliced && pl->pf[0].on) || s.rv == clh::kRvScanWideSliced) {
            // block minima of the bound, then seed + candidate slices (tasks), the slices, the finish -- one chain on the class's stream.
            // The prefilter reads the window text in address-aligned 256-byte blocks of the refs buffer: a buffer that is not
            // aligned so keeps the static slices (the pick kernels write them)
            clh_plan::PfClass& f = pl->pf[s.rv == clh::kRvScanSliced ? 0 : 1];
            P.pf_win = (const clh::PfWin*)f.d_win; P.pf_tasks = (const clh::PfTask*)f.d_pieces; P.pf_work = (const clh::PfWork*)f.d_work;
            P.pf_out = (clh::PfOut*)f.d_out; P.pf_ctl = (clh::PfCtl*)f.d_ctl; P.pf_bpl = f.bpl; P.pf_cap = f.cap;
            // ... nor does one whose stated size (clh_plan_set_refs_bytes) ends inside the last block a window touches
            P.pf_dmin = ((uintptr_t)d_refs & 255) == 0 && (pl->refs_bytes < 0 || f.extent <= pl->refs_bytes) ? (uint8_t*)f.d_dmin : nullptr;
            HIPCHK(hipMemsetAsync(f.d_ctl, 0, sizeof(clh::PfCtl), ls));
            if (s.rv == clh::kRvScanSliced) {
                P.pf_slices = (clh::ScanSlice*)f.d_queue; P.parts = (clh::ScanPart*)f.d_parts; P.pf_q2 = (int32_t*)f.d_q2; P.pf2_always = getenv("CLH_PF2_ALWAYS") != nullptr; P.pf2_share = getenv("CLH_PF2_SHARE") ? std::max(1, atoi(getenv("CLH_PF2_SHARE"))) : 8;
                if (int rc = prefilter(f, ls)) return rc;
                HIPCHK(clh::launch_ssw_scan_filtered(pl->quirk, P, s.count, std::min(f.cap, c->n_cu * 12), std::min(f.nwork, c->n_cu * 16), ls));
            } else {
                P.ws_tasks = (clh::WsTask*)f.d_queue; P.ws_bound = (uint16_t*)f.d_bound; P.ws_row0 = f.ws_row0; P.ws_slot_bytes = f.ws_slot;
                P.ws_dirs_off = f.ws_dirs_off;
                if (int rc = prefilter(f, ls)) return rc;
                HIPCHK(clh::launch_ssw_scanw_filtered(pl->quirk, P, s.count, f.ws_wgs, P.pf_dmin != nullptr, ls));
            }
        } else if (s.rv == clh::kRvScanSliced) {
            P.slices = (const clh::ScanSlice*)pl->d_slices; P.parts = (clh::ScanPart*)pl->d_parts;
            HIPCHK(clh::launch_ssw_scan_sliced(pl->quirk, P, s.count, (int)pl->slices.size(), ls));
        } else if (clh::rv_is_lanes(s.rv)) HIPCHK(clh::launch_ssw_lanes(clh::rv_lanes_columns(s.rv), P, s.count, ls));
        else if (s.rv == clh::kRvScan) HIPCHK(clh::launch_ssw_scan(pl->quirk, P, s.count, ls));
        else if (s.rv == clh::kRvScanTr) {
            int* ctr = (int*)pl->d_seg_ctr + k;
            HIPCHK(hipMemsetAsync(ctr, 0, sizeof(int), ls));
            HIPCHK(clh::launch_ssw_scanw_tr(pl->quirk, P, s.count, s.ws_wgs, ctr, (long long)s.ws_off, s.ws_slot, ls));
        }
        else if (s.rv == clh::kRvScanWide) {
            int* ctr = (int*)pl->d_seg_ctr + k;
            HIPCHK(hipMemsetAsync(ctr, 0, sizeof(int), ls));
            HIPCHK(clh::launch_ssw_scanw(pl->quirk, P, s.count, s.ws_wgs, ctr, (long long)s.ws_off, s.ws_slot, ls));
        }
        else HIPCHK(clh::launch_ssw(s.rv, pl->quirk, P, s.count, ls));
        if (pl->profiling) HIPCHK(hipEventRecord(pl->ev[2 * k + 1], ls));
        if (chained && ++i_wide < n_wide) {
            if (pl->chain_ev.size() < (size_t)i_wide) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); pl->chain_ev.push_back(e); }
            HIPCHK(hipEventRecord(pl->chain_ev[i_wide - 1], st));
            ls = c->side[(i_wide - 1) % 3];
            HIPCHK(hipStreamWaitEvent(ls, pl->chain_ev[i_wide - 1], 0));
        }
        if (tb && !pl->profiling) { if (int rc = traceback(s.begin, s.count, (int)(k % clh::kTbMaxSeg), s.rv, ls)) return rc; }
    }
    if (fan)
        for (int i = 0; i < 3; ++i) {
            HIPCHK(hipEventRecord(c->join_ev[i], c->side[i]));
            HIPCHK(hipStreamWaitEvent(st, c->join_ev[i], 0));
        }
    for (size_t k = 0; k < pl->segs.size(); ++k) {
        const auto& s = pl->segs[k];
        if (s.rv != clh::kRvCombine) continue;
        P.tasks = (const clh::SswTask*)pl->d_tasks + s.begin;
        if (pl->profiling) HIPCHK(hipEventRecord(pl->ev[2 * k + 0], st));
        HIPCHK(clh::launch_ssw_combine(P, s.count, st));
        if (pl->profiling) HIPCHK(hipEventRecord(pl->ev[2 * k + 1], st));
        if (tb && !pl->profiling) { if (int rc = traceback(s.begin, s.count, (int)(k % clh::kTbMaxSeg), rv_all, st)) return rc; }
    }
    if (tb && pl->profiling) {   // profiling runs: the traceback of all classes as serial launches after the score kernels
        const int rvmax = 32;
        HIPCHK(hipEventRecord(pl->ev[eb + 0], st));
        if (tb_rows_on()) HIPCHK(clh::launch_traceback_rows(PG, 0, pl->n_all, pl->n_all, 0, pool, head, pl->pool_bytes, st));
        else HIPCHK(clh::launch_traceback_pool(0, PG, 0, pl->n_all, pl->n_all, 0, pool, head, pl->pool_bytes, st));
        HIPCHK(hipEventRecord(pl->ev[eb + 1], st));
        HIPCHK(hipEventRecord(pl->ev[eb + 2], st));
        if (tb_rows_on()) HIPCHK(clh::launch_traceback_rows_wide(PG, 0, pl->n_all, pl->n_all, 0, pool, head, pl->pool_bytes, st));
        HIPCHK(clh::launch_traceback_pool(rvmax, PG, 0, pl->n_all, pl->n_all, 0, pool, head, pl->pool_bytes, st));
        HIPCHK(hipEventRecord(pl->ev[eb + 3], st));
    }
    if (!pl->done_ev) HIPCHK(hipEventCreateWithFlags(&pl->done_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(pl->done_ev, st));
    pl->last_stream = st;
    pl->ran = true;
    return 0;
}

extern "C" int clh_plan_set_refs_bytes(clh_plan* pl, int64_t nbytes)
{
    if (!pl) return fail(CLH_E_ARG, "null plan");
    pl->refs_bytes = nbytes;
    return 0;
}

extern "C" int clh_plan_set_profiling(clh_plan* pl, int on)
{
    if (!pl) return fail(CLH_E_ARG, "null plan");
    pl->profiling = on != 0;
    return 0;
}

extern "C" int clh_plan_segments(const clh_plan* pl, int32_t cap, int32_t* rv, int32_t* count, int64_t* read_bases, int64_t* ref_bases)
{
    if (!pl) return fail(CLH_E_ARG, "null plan");
    const int ns = (int)pl->segs.size();
    for (int k = 0; k < ns && k < cap; ++k) {
        rv[k] = pl->segs[k].rv; count[k] = pl->segs[k].count;
        int64_t a = 0, b = 0;
        for (int t = pl->segs[k].begin; t < pl->segs[k].begin + pl->segs[k].count; ++t) { a += pl->tasks[t].read_len; b += pl->tasks[t].ref_len; }
        read_bases[k] = a; ref_bases[k] = b;
    }
    return ns;
}

// durations of the last run's launches, in ms (HIP events on the run's stream); waits for the run.
// k1_ms[segment]; k1b_ms[0] = small-window traceback launch (all alignments), k1b_ms[1] = large-window launches.
extern "C" int clh_plan_timing(clh_plan* pl, int32_t cap, float* k1_ms, float* k1b_ms)
{
    if (!pl || !pl->ran || !pl->profiling || pl->ev.empty()) return fail(CLH_E_ARG, "profiling was not enabled for the last run");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->last_stream));
    const int ns = (int)pl->segs.size();
    for (int k = 0; k < ns && k < cap; ++k) HIPCHK(hipEventElapsedTime(&k1_ms[k], pl->ev[2 * k + 0], pl->ev[2 * k + 1]));
    k1b_ms[0] = k1b_ms[1] = 0.f;
    if (pl->do_cigar && pl->n > 0) {
        HIPCHK(hipEventElapsedTime(&k1b_ms[0], pl->ev[2 * ns + 0], pl->ev[2 * ns + 1]));   // small-window launch, all alignments
        HIPCHK(hipEventElapsedTime(&k1b_ms[1], pl->ev[2 * ns + 2], pl->ev[2 * ns + 3]));   // large-window launches, outliers only
    }
    return ns;
}

// counts[0] = alignments of the last run that the row traceback kernel handed to its wide form, counts[1] = alignments that
// went on to the anti-diagonal kernel (walks that leave the band, bands above 2048 cells); waits for the run
extern "C" int clh_plan_traceback_counts(clh_plan* pl, int32_t* counts)
{
    if (!pl || !counts || !pl->ran) return fail(CLH_E_ARG, "clh_plan_traceback_counts: no run yet");
    counts[0] = counts[1] = 0;
    if (!pl->d_pool_head) return 0;
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->last_stream));
    int32_t w[64];
    HIPCHK(hipMemcpy(w, (const char*)pl->d_pool_head + 16, sizeof(w), hipMemcpyDeviceToHost));
    for (int k = 0; k < 32; ++k) { counts[0] += w[k]; counts[1] += w[32 + k]; }
    return 0;
}

// what the prefilter of the sliced scan class did in the last run: out[0] alignments of the class, [1] of them with candidate
// slices instead of the static ones, [2] slices run, [3] window columns those slices computed, [4] window columns of the class;
// all zero when the plan has no such class or the filter is off; waits for the run
extern "C" int clh_plan_prefilter_stats(clh_plan* pl, int64_t* out)
{
    if (!pl || !out) return fail(CLH_E_ARG, "clh_plan_prefilter_stats: null argument");
    for (int k = 0; k < 6; ++k) out[k] = 0;
    if (!pl->ran) return 0;
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->last_stream));
    for (const auto& f : pl->pf) {
        if (!f.on) continue;
        clh::PfCtl c;
        HIPCHK(hipMemcpy(&c, f.d_ctl, sizeof(c), hipMemcpyDeviceToHost));
        out[0] += f.ntasks; out[1] += c.n_pruned; out[2] += c.qcount; out[3] += (int64_t)c.cols_scanned; out[4] += (int64_t)c.cols_window; out[5] += c.n_stage2;
    }
    return 0;
}

// ssw_prefilter_kernel alone in the last (profiling) run: ms[ci] = its duration for the K1s class (0) and the K1w class (1), 0 when it did
// not run; work[2 ci] = window columns x W words of the read's pieces, work[2 ci + 1] = the same columns x (11 W + 8), the kernel's
// instructions per column and lane -- what its issue-rate roofline counts
extern "C" int clh_plan_prefilter_timing(clh_plan* pl, float* ms, int64_t* work)
{
    if (!pl || !ms || !work) return fail(CLH_E_ARG, "clh_plan_prefilter_timing: null argument");
    if (!pl->ran || !pl->profiling) return fail(CLH_E_ARG, "profiling was not enabled for the last run");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->last_stream));
    for (int ci = 0; ci < 2; ++ci) {
        const auto& f = pl->pf[ci];
        ms[ci] = 0.f; work[2 * ci] = work[2 * ci + 1] = 0;
        if (!f.on || !f.timed) continue;
        HIPCHK(hipEventElapsedTime(&ms[ci], f.ev[0], f.ev[1]));
        work[2 * ci] = f.word_cols; work[2 * ci + 1] = f.lane_insts;
    }
    return 0;
}

extern "C" const void* clh_ssw_results_dev(const clh_plan* pl) { return pl ? pl->d_results : nullptr; }

// CIGARs sit in worst-case shares (2 * readLen + 2 words per alignment, ~100x what is used): they are gathered into one
// dense run on the device and only that run crosses PCIe
__global__ void cigar_gather_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ src_off, const int64_t* __restrict__ dst_off,
                                    const int32_t* __restrict__ len, uint32_t* __restrict__ dst)
{
    const int a = blockIdx.x, L = len[a];
    const uint32_t* s = src + src_off[a];
    uint32_t* d = dst + dst_off[a];
    for (int k = threadIdx.x; k < L; k += blockDim.x) d[k] = s[k];
}

extern "C" int clh_ssw_fetch(clh_plan* pl, clh_align_t* out, uint32_t* cigar_buf, int64_t cigar_cap, int64_t* cigar_used)
{
    if (!pl || !out) return fail(CLH_E_ARG, "clh_ssw_fetch: null argument");
    if (!pl->ran) return fail(CLH_E_ARG, "clh_ssw_fetch before clh_ssw_run");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(pl->done_ev ? hipEventSynchronize(pl->done_ev) : hipStreamSynchronize(pl->last_stream));
    const int n = pl->n;
    std::vector<clh::SswResult> res((size_t)std::max(n, 1));
    std::vector<int32_t> clen((size_t)std::max(n, 1), 0);
    if (n > 0) HIPCHK(hipMemcpy(res.data(), pl->d_results, sizeof(clh::SswResult) * (size_t)n, hipMemcpyDeviceToHost));
    if (pl->do_cigar && n > 0) HIPCHK(hipMemcpy(clen.data(), pl->d_cigar_len, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<int32_t> share_off((size_t)std::max(n, 1), 0), glen((size_t)std::max(n, 1), 0);
    std::vector<int64_t> dst_off((size_t)std::max(n, 1), 0);
    for (const auto& t : pl->tasks) if (t.out_index < n) share_off[t.out_index] = t.cigar_off;
    int64_t used = 0;
    int rc = 0;
    for (int a = 0; a < n; ++a) {
        const clh::SswResult& r = res[a];
        clh_align_t& o = out[a];
        o.score1 = (uint16_t)r.score1; o.score2 = (uint16_t)r.score2;
        o.ref_begin1 = r.ref_begin1; o.ref_end1 = r.ref_end1; o.read_begin1 = r.read_begin1; o.read_end1 = r.read_end1;
        o.ref_end2 = r.ref_end2;
        o.status = 0;
        if (r.status & clh::CLH_STATUS_WORD) o.status |= CLH_ST_WORD;
        if (r.status & clh::CLH_STATUS_OVERFLOW8) o.status |= CLH_ST_NULL;
        if (r.status & clh::CLH_STATUS_TRACE_ERR) o.status |= CLH_ST_TRACE_ERR;
        if (r.status & clh::CLH_STATUS_NO_CIGAR) o.status |= CLH_ST_NO_CIGAR;
        if (r.status & clh::CLH_STATUS_CIGAR_TRUNC) o.status |= CLH_ST_CIGAR_TRUNC;
        o.cigar_off = -1; o.cigar_len = 0;
        if (!pl->do_cigar) { o.status |= CLH_ST_NO_CIGAR; continue; }
        const int len = clen[a];
        if (len > 0) {
            o.cigar_len = len;
            if (cigar_buf) {
                if (used + len > cigar_cap) { rc = CLH_E_CAPACITY; o.cigar_len = 0; continue; }
                o.cigar_off = (int32_t)used;
                dst_off[a] = used; glen[a] = len;
            }
            used += len;
        }
    }
    if (cigar_buf && pl->do_cigar && n > 0 && used > 0) {
        clh_ctx* c = pl->ctx;
        const size_t nb4 = sizeof(int32_t) * (size_t)n, nb8 = sizeof(int64_t) * (size_t)n;
        void *d_src = c->alloc(nb4), *d_len = c->alloc(nb4), *d_dst = c->alloc(nb8), *d_out = c->alloc(sizeof(uint32_t) * (size_t)used);
        int grc = 0;
        if (!d_src || !d_len || !d_dst || !d_out) grc = fail(CLH_E_HIP, "out of device memory while gathering CIGARs");
        hipStream_t st = c->stream;
        if (!grc && (hipMemcpyAsync(d_src, share_off.data(), nb4, hipMemcpyHostToDevice, st) != hipSuccess ||
                     hipMemcpyAsync(d_len, glen.data(), nb4, hipMemcpyHostToDevice, st) != hipSuccess ||
                     hipMemcpyAsync(d_dst, dst_off.data(), nb8, hipMemcpyHostToDevice, st) != hipSuccess)) grc = fail(CLH_E_HIP, "H2D failed");
        if (!grc) {
            hipLaunchKernelGGL(cigar_gather_kernel, dim3(n), dim3(64), 0, st, (const uint32_t*)pl->d_cigars, (const int32_t*)d_src, (const int64_t*)d_dst,
                               (const int32_t*)d_len, (uint32_t*)d_out);
            if (hipGetLastError() != hipSuccess) grc = fail(CLH_E_HIP, "cigar gather launch failed");
        }
        if (!grc && hipMemcpyAsync(cigar_buf, d_out, sizeof(uint32_t) * (size_t)used, hipMemcpyDeviceToHost, st) != hipSuccess) grc = fail(CLH_E_HIP, "D2H failed");
        const hipError_t se = hipStreamSynchronize(st);      // also before the buffers go back to the cache
        if (!grc && se != hipSuccess) grc = fail(CLH_E_HIP, "cigar gather failed");
        c->release(d_src); c->release(d_len); c->release(d_dst); c->release(d_out);
        if (grc) return grc;
    }
    if (cigar_used) *cigar_used = used;
    if (rc) return fail(rc, "cigar buffer too small");
    return 0;
}

extern "C" int clh_ssw_batch(clh_ctx* ctx, int32_t n, const int8_t* reads, const int64_t* read_off, const int8_t* refs,
                             const int64_t* ref_off, const int32_t* mask_len, const clh_ssw_opts* opts, clh_align_t* out,
                             uint32_t* cigar_buf, int64_t cigar_cap, int64_t* cigar_used)
{
    if (!ctx || !reads || !refs || !read_off || !ref_off) return fail(CLH_E_ARG, "clh_ssw_batch: null argument");
    clh_plan* pl = clh_ssw_plan(ctx, n, read_off, ref_off, mask_len, opts);
    if (!pl) return g_err.empty() ? CLH_E_ARG : (g_err.find("not implemented") != std::string::npos ? CLH_E_UNSUPPORTED : CLH_E_ARG);
    int rc = 0;
    const size_t rb = (size_t)read_off[n], fb = (size_t)ref_off[n];
    pl->d_reads = ctx->alloc(rb + 64);
    pl->d_refs = ctx->alloc(fb + 64);
    if (!pl->d_reads || !pl->d_refs) { rc = fail(CLH_E_HIP, "out of device memory for the batch"); }
    if (!rc && hipMemcpyAsync(pl->d_reads, reads, rb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(CLH_E_HIP, "H2D reads failed");
    if (!rc && hipMemcpyAsync(pl->d_refs, refs, fb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(CLH_E_HIP, "H2D refs failed");
    if (!rc) rc = clh_ssw_run(pl, pl->d_reads, pl->d_refs, nullptr);
    if (!rc) rc = clh_ssw_fetch(pl, out, cigar_buf, cigar_cap, cigar_used);
    clh_plan_destroy(pl);
    return rc;
}

// clh_ssw_batch with the references given as windows of a resident genome: only the reads cross PCIe
extern "C" int clh_ssw_windows_batch(clh_genome* genome, int32_t n, const int8_t* reads, const int64_t* read_off, const int64_t* win_off,
                                     const int32_t* win_len, const uint8_t* win_rc, const int32_t* mask_len, const clh_ssw_opts* opts,
                                     clh_align_t* out, uint32_t* cigar_buf, int64_t cigar_cap, int64_t* cigar_used)
{
    if (!genome || !reads || !read_off || !win_off || !win_len) return fail(CLH_E_ARG, "clh_ssw_windows_batch: null argument");
    for (int i = 0; i < n; ++i)
        if (win_off[i] < 0 || win_len[i] < 0 || win_off[i] + win_len[i] > genome->len) return fail(CLH_E_ARG, "clh_ssw_windows_batch: window outside the genome");
    clh_ctx* ctx = genome->ctx;
    clh_plan* pl = clh_ssw_plan_windows(ctx, n, read_off, win_off, win_len, win_rc, mask_len, opts);
    if (!pl) return g_err.empty() ? CLH_E_ARG : (g_err.find("not implemented") != std::string::npos ? CLH_E_UNSUPPORTED : CLH_E_ARG);
    int rc = 0;
    const size_t rb = (size_t)read_off[n];
    pl->d_reads = ctx->alloc(rb + 64);
    if (!pl->d_reads) { rc = fail(CLH_E_HIP, "out of device memory for the batch"); }
    if (!rc && hipMemcpyAsync(pl->d_reads, reads, rb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(CLH_E_HIP, "H2D reads failed");
    if (!rc) rc = clh_ssw_run(pl, pl->d_reads, genome->d_codes, nullptr);
    if (!rc) rc = clh_ssw_fetch(pl, out, cigar_buf, cigar_cap, cigar_used);
    clh_plan_destroy(pl);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// K4: unit-cost edit distance of n pairs of byte strings (utils.py:153-159 `distance`)
// ---------------------------------------------------------------------------------------------------------------
struct clh_edit_plan {
    clh_ctx* ctx = nullptr;
    int n = 0, planes = 3;
    std::vector<clh::EdTask> tasks;          // launch order (by lane-group class, longest text first)
    std::vector<int32_t> trivial;            // out[k] for pairs with an empty side, -1 otherwise
    void *d_sym = nullptr, *d_tasks = nullptr, *d_out = nullptr, *d_carry = nullptr;
    hipStream_t last_stream = nullptr;
    bool ran = false;
    hipEvent_t ev[2] = {nullptr, nullptr};
};

static int ed_group(const clh::EdTask& t) { int B = (t.pat_len + 63) >> 6, G = 1; while (G < B && G < 64) G <<= 1; return G; }   // above 64 blocks: passes

extern "C" void clh_edit_plan_destroy(clh_edit_plan* pl)
{
    if (!pl) return;
    (void)hipSetDevice(pl->ctx->device);
    if (pl->ran) (void)hipStreamSynchronize(pl->last_stream);
    pl->ctx->release(pl->d_sym); pl->ctx->release(pl->d_tasks); pl->ctx->release(pl->d_out); pl->ctx->release(pl->d_carry);
    for (hipEvent_t e : pl->ev) if (e) (void)hipEventDestroy(e);
    delete pl;
}

// uploads the strings (as dense symbol codes) and the task table; the plan then runs any number of times
extern "C" clh_edit_plan* clh_edit_plan_create(clh_ctx* ctx, int32_t n, const uint8_t* a, const int64_t* a_off, const uint8_t* b, const int64_t* b_off)
{
    if (!ctx || n < 0 || !a_off || !b_off || (n > 0 && (!a || !b))) { fail(CLH_E_ARG, "clh_edit_plan_create: null argument"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(CLH_E_HIP, "hipSetDevice failed"); return nullptr; }
    const int64_t ta = n ? a_off[n] - a_off[0] : 0, tb = n ? b_off[n] - b_off[0] : 0;
    if (ta < 0 || tb < 0) { fail(CLH_E_ARG, "clh_edit_plan_create: offsets must ascend"); return nullptr; }
    clh_edit_plan* pl = new clh_edit_plan();
    pl->ctx = ctx; pl->n = n;
    // the batch's alphabet -> dense codes; the kernel builds match vectors from 3 bit planes (<= 8 symbols: DNA) or 8
    int code[256];
    for (int& c : code) c = -1;
    int nsym = 0;
    auto learn = [&](const uint8_t* p, int64_t len) { for (int64_t i = 0; i < len; ++i) if (code[p[i]] < 0) code[p[i]] = nsym++; };
    if (n) { learn(a + a_off[0], ta); learn(b + b_off[0], tb); }
    pl->planes = nsym <= 8 ? 3 : 8;
    std::vector<uint8_t> sym((size_t)(ta + tb) + 32, 0);
    for (int64_t i = 0; i < ta; ++i) sym[(size_t)i] = (uint8_t)code[a[a_off[0] + i]];
    for (int64_t i = 0; i < tb; ++i) sym[(size_t)(ta + i)] = (uint8_t)code[b[b_off[0] + i]];
    pl->trivial.assign((size_t)n, -1);
    pl->tasks.reserve((size_t)n);
    size_t carry_bytes = 0;
    for (int k = 0; k < n; ++k) {
        const int64_t la = a_off[k + 1] - a_off[k], lb = b_off[k + 1] - b_off[k];
        if (la < 0 || lb < 0) { fail(CLH_E_ARG, "clh_edit_plan_create: offsets must ascend"); delete pl; return nullptr; }
        if (la == 0 || lb == 0) { pl->trivial[(size_t)k] = (int32_t)(la + lb); continue; }
        clh::EdTask t;
        const bool a_is_pat = la <= lb;
        t.pat_off = a_is_pat ? a_off[k] - a_off[0] : ta + (b_off[k] - b_off[0]);
        t.txt_off = a_is_pat ? ta + (b_off[k] - b_off[0]) : a_off[k] - a_off[0];
        t.pat_len = (int32_t)(a_is_pat ? la : lb); t.txt_len = (int32_t)(a_is_pat ? lb : la);
        t.out_index = k; t.carry_off64 = -1;
        if (t.pat_len > 4096) {      // swept in passes of 64 blocks: two buffers of one byte per text column (+ slack for 16-byte reads)
            t.carry_off64 = (int32_t)(carry_bytes / 64);
            carry_bytes += 2 * ((((size_t)t.txt_len + 63) & ~(size_t)63) + 64);
        }
        pl->tasks.push_back(t);
    }
    std::stable_sort(pl->tasks.begin(), pl->tasks.end(), [&](const clh::EdTask& x, const clh::EdTask& y) {
        const int gx = ed_group(x), gy = ed_group(y);
        if (gx != gy) return gx < gy;
        return x.txt_len > y.txt_len;                 // similar step counts share a wave
    });
    const size_t nt = pl->tasks.size();
    pl->d_sym = ctx->alloc(sym.size());
    pl->d_tasks = ctx->alloc(sizeof(clh::EdTask) * std::max<size_t>(nt, 1));
    pl->d_out = ctx->alloc(sizeof(int32_t) * (size_t)std::max(n, 1));
    if (carry_bytes) pl->d_carry = ctx->alloc(carry_bytes + 64);
    if (!pl->d_sym || !pl->d_tasks || !pl->d_out || (carry_bytes && !pl->d_carry) ||
        hipMemcpy(pl->d_sym, sym.data(), sym.size(), hipMemcpyHostToDevice) != hipSuccess ||
        (nt && hipMemcpy(pl->d_tasks, pl->tasks.data(), sizeof(clh::EdTask) * nt, hipMemcpyHostToDevice) != hipSuccess)) {
        fail(CLH_E_HIP, "out of device memory or upload failed while building the edit-distance plan");
        clh_edit_plan_destroy(pl); return nullptr;
    }
    return pl;
}

extern "C" int clh_edit_plan_run(clh_edit_plan* pl, void* stream_)
{
    if (!pl) return fail(CLH_E_ARG, "clh_edit_plan_run: null argument");
    HIPCHK(hipSetDevice(pl->ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : pl->ctx->stream;
    if (!pl->ev[0]) for (auto& e : pl->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventRecord(pl->ev[0], st));
    const int nt = (int)pl->tasks.size();
    for (int i = 0; i < nt;) {
        const int G = ed_group(pl->tasks[(size_t)i]);
        int j = i;
        while (j < nt && ed_group(pl->tasks[(size_t)j]) == G) ++j;
        HIPCHK(clh::launch_edit_distance((const uint8_t*)pl->d_sym, (const clh::EdTask*)pl->d_tasks + i, j - i, G, pl->planes, (int32_t*)pl->d_out, (int8_t*)pl->d_carry, st));
        i = j;
    }
    HIPCHK(hipEventRecord(pl->ev[1], st));
    pl->last_stream = st; pl->ran = true;
    return 0;
}

extern "C" int clh_edit_plan_fetch(clh_edit_plan* pl, int32_t* out)
{
    if (!pl || !out) return fail(CLH_E_ARG, "clh_edit_plan_fetch: null argument");
    if (!pl->ran) return fail(CLH_E_ARG, "clh_edit_plan_fetch before clh_edit_plan_run");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->last_stream));
    std::vector<int32_t> res((size_t)std::max(pl->n, 1));
    if (pl->n) HIPCHK(hipMemcpy(res.data(), pl->d_out, sizeof(int32_t) * (size_t)pl->n, hipMemcpyDeviceToHost));
    for (int k = 0; k < pl->n; ++k) out[k] = pl->trivial[(size_t)k] >= 0 ? pl->trivial[(size_t)k] : res[(size_t)k];
    return 0;
}

extern "C" int clh_edit_plan_timing(clh_edit_plan* pl, float* ms)
{
    if (!pl || !ms || !pl->ran) return fail(CLH_E_ARG, "clh_edit_plan_timing: no run to time");
    HIPCHK(hipEventSynchronize(pl->ev[1]));
    HIPCHK(hipEventElapsedTime(ms, pl->ev[0], pl->ev[1]));
    return 0;
}

extern "C" int clh_edit_distance_batch(clh_ctx* ctx, int32_t n, const uint8_t* a, const int64_t* a_off, const uint8_t* b, const int64_t* b_off,
                                       int32_t* out)
{
    if (!out) return fail(CLH_E_ARG, "clh_edit_distance_batch: null argument");
    if (n == 0) return 0;
    clh_edit_plan* pl = clh_edit_plan_create(ctx, n, a, a_off, b, b_off);
    if (!pl) return CLH_E_ARG;
    int rc = clh_edit_plan_run(pl, nullptr);
    if (!rc) rc = clh_edit_plan_fetch(pl, out);
    clh_edit_plan_destroy(pl);
    return rc;
}

extern "C" void clh_encode_dna(const char* seq, int64_t len, int8_t* out)
{
    static int8_t lut[256];
    static bool init = false;
    if (!init) {
        memset(lut, 4, sizeof(lut));
        lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1; lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = 3; lut['N'] = lut['n'] = 4;
        init = true;
    }
    for (int64_t i = 0; i < len; ++i) out[i] = lut[(unsigned char)seq[i]];
}

// ------------------------------------------------------------------------------------------------------------
// the reference's six symbols (include/ssw_legacy.h)
// ------------------------------------------------------------------------------------------------------------
struct _profile {
    const int8_t* read;
    const int8_t* mat;
    int32_t readLen;
    int32_t n;
    int8_t score_size;
};

static clh_ctx* legacy_ctx()
{
    static std::mutex mu;
    static clh_ctx* ctx = nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (!ctx) {
        const char* d = getenv("CIRI_LONG_DEVICE");
        ctx = clh_create(d ? atoi(d) : 0);
        if (!ctx) fprintf(stderr, "libclh: no usable GPU (%s); there is no CPU fallback.\n", clh_last_error());
    }
    return ctx;
}

extern "C" s_profile* ssw_init(const int8_t* read, const int32_t readLen, const int8_t* mat, const int32_t n, const int8_t score_size)
{
    s_profile* p = (s_profile*)calloc(1, sizeof(struct _profile));
    p->read = read; p->mat = mat; p->readLen = readLen; p->n = n; p->score_size = score_size;
    return p;
}

extern "C" void init_destroy(s_profile* p) { free(p); }

extern "C" s_align* ssw_align(const s_profile* prof, const int8_t* ref, int32_t refLen, const uint8_t weight_gapO,
                              const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters, const int32_t filterd,
                              const int32_t maskLen)
{
    if (maskLen < 15)
        fprintf(stderr, "When maskLen < 15, the function ssw_align doesn't return 2nd best alignment information.\n");
    if (!(prof->score_size == 0 || prof->score_size == 1 || prof->score_size == 2)) {
        fprintf(stderr, "Please call the function ssw_init before ssw_align.\n");
        return NULL;
    }
    clh_ctx* ctx = legacy_ctx();
    if (!ctx) return NULL;
    clh_ssw_opts o;
    memset(&o, 0, sizeof(o));
    o.mat = prof->mat; o.n_mat = prof->n; o.gap_open = weight_gapO; o.gap_extend = weight_gapE; o.flag = flag;
    o.score_size = prof->score_size; o.filters = filters; o.filterd = filterd; o.want_score2 = 1; o.want_cigar = 1;
    const int64_t roff[2] = {0, prof->readLen}, foff[2] = {0, refLen};
    const int32_t ml = maskLen;
    clh_align_t a;
    std::vector<uint32_t> cig((size_t)2 * (size_t)std::max(prof->readLen, 1) + 8);
    int64_t used = 0;
    int rc = clh_ssw_batch(ctx, 1, prof->read, roff, ref, foff, &ml, &o, &a, cig.data(), (int64_t)cig.size(), &used);
    if (rc != 0) { fprintf(stderr, "libclh: ssw_align failed: %s\n", clh_last_error()); return NULL; }
    if (a.status & CLH_ST_NULL) {
        fprintf(stderr, "Please set 2 to the score_size parameter of the function ssw_init, otherwise the alignment results will be incorrect.\n");
        return NULL;
    }
    if (a.status & (CLH_ST_TRACE_ERR | CLH_ST_CIGAR_TRUNC)) { fprintf(stderr, "Trace back error.\n"); return NULL; }
    s_align* r = (s_align*)calloc(1, sizeof(s_align));
    r->score1 = a.score1; r->score2 = a.score2; r->ref_end1 = a.ref_end1; r->read_end1 = a.read_end1; r->ref_end2 = a.ref_end2;
    r->ref_begin1 = a.ref_begin1; r->read_begin1 = a.read_begin1;
    r->cigar = 0; r->cigarLen = 0;
    if (a.cigar_len > 0 && a.cigar_off >= 0) {
        r->cigar = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)a.cigar_len);
        memcpy(r->cigar, cig.data() + a.cigar_off, sizeof(uint32_t) * (size_t)a.cigar_len);
        r->cigarLen = a.cigar_len;
    }
    return r;
}

extern "C" void align_destroy(s_align* a)
{
    if (!a) return;
    free(a->cigar);
    free(a);
}

extern "C" char cigar_int_to_op(uint32_t cigar_int)
{
    static const char map[] = {'M', 'I', 'D', 'N', 'S', 'H', 'P', '=', 'X'};
    const uint32_t c = cigar_int & 0xfU;
    return c >= sizeof(map) ? 'M' : map[c];
}

extern "C" uint32_t cigar_int_to_len(uint32_t cigar_int) { return cigar_int >> 4; }

// ------------------------------------------------------------------------------------------------------------
// cyclic consensus: find_consensus for a batch of reads (K2 + K3)
// ------------------------------------------------------------------------------------------------------------
struct clh_ccs_plan;
static clh_ccs_plan* ccs_plan_create(clh_ctx* ctx, int32_t n, const int64_t* read_off, int mcap_hint, bool wide_hint = false);
extern "C" clh_ccs_plan* clh_ccs_plan_create(clh_ctx* ctx, int32_t n, const int64_t* read_off) { return ccs_plan_create(ctx, n, read_off, 0); }

struct clh_ccs_plan {
    clh_ctx* ctx = nullptr;
    int n = 0, lcap = 0, lmax = 0, n_long = 0, nslots = 0, nslots_big = 0;
    void *d_long = nullptr, *d_k2ws = nullptr, *d_busy = nullptr;
    struct K2Class { int begin, count, lcap; };
    std::vector<K2Class> k2_classes;
    int64_t total = 0;
    size_t slot_bytes = 0, slot_bytes_big = 0;      // second tier: a few slots sized for the worst case of the batch
    void *d_off = nullptr, *d_scan = nullptr, *d_res = nullptr, *d_segs = nullptr, *d_ccs = nullptr, *d_ws = nullptr, *d_ws_big = nullptr,
         *d_counter = nullptr, *d_order = nullptr, *d_order3 = nullptr, *d_reads = nullptr, *d_score = nullptr, *d_wide = nullptr;
    hipStream_t last_stream = nullptr;
    bool ran = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // K2 start, K2 stop = K3 start, K3 stop
};

extern "C" void clh_ccs_plan_destroy(clh_ccs_plan* pl)
{
    if (!pl) return;
    (void)hipSetDevice(pl->ctx->device);
    if (pl->ran) (void)hipStreamSynchronize(pl->last_stream);
    void* bufs[] = {pl->d_off, pl->d_scan, pl->d_res, pl->d_segs, pl->d_ccs, pl->d_ws, pl->d_ws_big, pl->d_counter, pl->d_order, pl->d_order3, pl->d_reads, pl->d_long, pl->d_k2ws, pl->d_busy, pl->d_score, pl->d_wide};
    for (void* b : bufs) pl->ctx->release(b);
    for (hipEvent_t e : pl->ev) if (e) (void)hipEventDestroy(e);
    delete pl;
}

// scores of the consensus step of find_consensus: local alignment with the numbers of the reference's call
// (tests/test_poa.py:30), consensus over the nodes crossed by at least half of the copies (oracle/ccs_oracle.c)
static clh::PoaScores ccs_scores() { clh::PoaScores s; s.algorithm = 0; s.m = 10; s.n = -4; s.g = -8; s.e = -2; s.q = -24; s.c = -1; s.min_cov = -1; return s; }

static clh_ccs_plan* ccs_plan_create(clh_ctx* ctx, int32_t n, const int64_t* read_off, int mcap_hint, bool wide_hint)
{
    if (!ctx || n < 0 || !read_off) { fail(CLH_E_ARG, "clh_ccs_plan_create: null argument"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { fail(CLH_E_HIP, "hipSetDevice failed"); return nullptr; }
    clh_ccs_plan* pl = new clh_ccs_plan();
    pl->ctx = ctx; pl->n = n; pl->total = read_off[n];
    int lmax = 1;
    std::vector<int32_t> order(n), long_idx;
    for (int i = 0; i < n; ++i) {
        const int64_t L = read_off[i + 1] - read_off[i];
        if (L < 0 || L > (1 << 24)) { fail(CLH_E_UNSUPPORTED, "read longer than 16 M bases"); delete pl; return nullptr; }
        lmax = std::max(lmax, (int)L);
        if (L > clh::kK2LdsMax) long_idx.push_back(i);     // scanned out of an HBM workspace (ccs_scan_long_kernel)
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return read_off[x + 1] - read_off[x] > read_off[y + 1] - read_off[y]; });
    pl->lcap = (std::min(lmax, clh::kK2LdsMax) + 63) & ~63;
    {   // K2 launch classes over the length-sorted order: the LDS block of a launch is sized for its longest read
        static const int T[] = {1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384};
        auto cls_of = [&](int64_t L) { for (int t : T) if (L <= t) return std::min(t, pl->lcap); return pl->lcap; };
        for (int k = 0; k < n;) {
            const int c = cls_of(std::min<int64_t>(read_off[order[k] + 1] - read_off[order[k]], clh::kK2LdsMax));
            int e = k;
            while (e < n && cls_of(std::min<int64_t>(read_off[order[e] + 1] - read_off[order[e]], clh::kK2LdsMax)) == c) ++e;
            pl->k2_classes.push_back({k, e - k, c});
            k = e;
        }
    }
    pl->lmax = lmax; pl->n_long = (int)long_idx.size();
    // Workspace.  The worst case of a read of L bases is a graph of L+8 nodes against copies of L/2 + L/16 bases (period
    // <= L/2, tolerance period/8), every row kept and with several
    // in-edges -- ~6 bytes per cell of that -- while the common case (period of a few hundred bases) needs a small
    // fraction.  So the first-tier slots (16 waves per CU x 256 CUs) share a budget, a wave whose read outgrows its slot
    // claims one of a few worst-case slots, and whatever found none free runs in a second launch over those.  Both budgets
    // follow the free memory of the device (HBM is 288 GB on an MI355X; nothing here assumes it).
    // (A copy above 2800 bases, or scores outside the 16-bit cells, runs the wide form of the pass: 6 bytes per cell instead of 4.)
    const int mcap_worst = mcap_hint > 0 ? mcap_hint + 1 : lmax / 2 + lmax / 16 + 8;
    const bool wide_worst = mcap_worst > 2801 || wide_hint || getenv("CLH_POA_FORCE_WIDE") != nullptr;
    const size_t need_worst = wide_worst ? clh::poa_slot_bytes_host_w(lmax + 8, mcap_worst) : clh::poa_slot_bytes_host(lmax + 8, mcap_worst);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { fail(CLH_E_HIP, "hipMemGetInfo failed"); delete pl; return nullptr; }
    { std::lock_guard<std::mutex> g(ctx->mu); for (auto& kv : ctx->cache) free_b += kv.first; }      // parked blocks are ours to reuse
    int slots_max = 4096;                            // 16 waves per CU x 256 CUs
    if (const char* e = getenv("CLH_POA_SLOTS")) slots_max = std::max(64, atoi(e));      // tuning experiments
    pl->nslots = (int)std::max<long long>(1, std::min<long long>(slots_max, std::max(n, 1)));
    unsigned long long budget = std::min<unsigned long long>(40ull << 30, (unsigned long long)(free_b * 0.40));
    if (const char* e = getenv("CLH_POA_BUDGET_MB")) budget = std::max(1ull, strtoull(e, nullptr, 10)) << 20;     // tests: force the second tier
    pl->slot_bytes = std::min<size_t>(need_worst, (size_t)((budget / (unsigned long long)pl->nslots) & ~255ull));
    pl->slot_bytes = std::max<size_t>(pl->slot_bytes, 4096);
    if (pl->slot_bytes < need_worst) {
        // a FEW worst-case slots: rounds 2-5 took up to 1024 of them (64 GB); a batch of reads up to 4.5 kb then held 87 GB per plan, more than
        // the context parks between plans, and the file stage paid a 40 GB hipMalloc / hipFree per 32 MB of input (8.6 s for a 100 000-read
        // file whose kernels take 30 ms).  What finds no large slot free runs in the second launch over them, as before.
        const unsigned long long budget_big = std::min<unsigned long long>(12ull << 30, (unsigned long long)(free_b * 0.10));
        pl->slot_bytes_big = need_worst;
        if (const char* e = getenv("CLH_POA_BIG_BYTES")) pl->slot_bytes_big = std::max<size_t>(4096, std::min<size_t>(need_worst, strtoull(e, nullptr, 10)));   // tests: large slots too small (status 1)
        pl->nslots_big = (int)std::max<unsigned long long>(1, std::min<unsigned long long>(256, budget_big / pl->slot_bytes_big));
        pl->nslots_big = std::min(pl->nslots_big, std::max(n, 1));
        if (const char* e = getenv("CLH_POA_BIG_SLOTS")) pl->nslots_big = std::max(1, std::min(pl->nslots_big, atoi(e)));       // tests: make the large slots scarce
    }
    pl->d_off = ctx->alloc(sizeof(int64_t) * (size_t)(n + 1));
    pl->d_scan = ctx->alloc(sizeof(clh::CcsScan) * (size_t)std::max(n, 1));
    pl->d_res = ctx->alloc(sizeof(clh::CcsResult) * (size_t)std::max(n, 1));
    pl->d_segs = ctx->alloc(sizeof(int32_t) * 2 * clh::CCS_SEG_CAP * (size_t)std::max(n, 1));
    pl->d_ccs = ctx->alloc((size_t)std::max<int64_t>(pl->total, 1) + 64);
    pl->d_ws = ctx->alloc(pl->slot_bytes * (size_t)pl->nslots);
    if (pl->nslots_big) {
        pl->d_ws_big = ctx->alloc(pl->slot_bytes_big * (size_t)pl->nslots_big);
        pl->d_busy = ctx->alloc(sizeof(int) * (size_t)pl->nslots_big);
        if (pl->d_busy && hipMemset(pl->d_busy, 0, sizeof(int) * (size_t)pl->nslots_big) != hipSuccess) { ctx->release(pl->d_busy); pl->d_busy = nullptr; }
    }
    if (pl->n_long) {
        pl->d_long = ctx->alloc(sizeof(int32_t) * (size_t)pl->n_long);
        pl->d_k2ws = ctx->alloc(clh::k2_long_slot_bytes(lmax) * (size_t)pl->n_long);
        if (!pl->d_long || !pl->d_k2ws || hipMemcpy(pl->d_long, long_idx.data(), sizeof(int32_t) * (size_t)pl->n_long, hipMemcpyHostToDevice) != hipSuccess) {
            fail(CLH_E_HIP, "out of device memory for the long-read scan workspace");
            clh_ccs_plan_destroy(pl); return nullptr;
        }
    }
    pl->d_counter = ctx->alloc(256);
    pl->d_order = ctx->alloc(sizeof(int32_t) * (size_t)std::max(n, 1));
    pl->d_order3 = ctx->alloc(sizeof(int32_t) * (size_t)std::max(n, 1));     // K3's work list by cost (clh_ccs_run)
    pl->d_wide = ctx->alloc(sizeof(int32_t) * (size_t)std::max(n, 1));       // reads for the wide form of K3's pass
    if (!pl->d_off || !pl->d_scan || !pl->d_res || !pl->d_segs || !pl->d_ccs || !pl->d_ws || (pl->nslots_big && (!pl->d_ws_big || !pl->d_busy)) || !pl->d_counter || !pl->d_order || !pl->d_order3 || !pl->d_wide) {
        fail(CLH_E_HIP, "out of device memory while building the consensus plan");
        clh_ccs_plan_destroy(pl); return nullptr;
    }
    (void)hipMemset(pl->d_segs, 0, sizeof(int32_t) * 2 * clh::CCS_SEG_CAP * (size_t)std::max(n, 1));   // entries beyond nseg read as 0
    if (hipMemcpy(pl->d_off, read_off, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice) != hipSuccess ||
        (n > 0 && hipMemcpy(pl->d_order, order.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)) {
        fail(CLH_E_HIP, "plan upload failed");
        clh_ccs_plan_destroy(pl); return nullptr;
    }
    return pl;
}

// the K3 launches of a plan: first tier over every read, second tier over what found no slot large enough
static int launch_poa_tiers(clh_ccs_plan* pl, clh::CcsParams& P, hipStream_t st, bool by_cost = false)
{
    P.poa_ws = (uint8_t*)pl->d_ws; P.work_counter = (int*)pl->d_counter; P.stats = (int*)pl->d_counter + 2;
    P.wide_list = (int32_t*)pl->d_wide; P.wide_count = (int*)pl->d_counter + 32;
    P.work_order = (const int32_t*)(by_cost ? pl->d_order3 : pl->d_order);
    P.slot_bytes = pl->slot_bytes; P.n = pl->n; P.lcap = pl->lcap; P.tier = 0;
    if (pl->nslots_big) { P.big_ws = (uint8_t*)pl->d_ws_big; P.big_slot_bytes = pl->slot_bytes_big; P.big_busy = (int*)pl->d_busy; P.n_big = pl->nslots_big; }
    HIPCHK(clh::launch_poa(P, pl->nslots, st));
    if (pl->nslots_big) {       // the reads the first tier left with status 1
        clh::CcsParams Q = P;
        Q.poa_ws = (uint8_t*)pl->d_ws_big; Q.slot_bytes = pl->slot_bytes_big; Q.work_counter = (int*)pl->d_counter + 1; Q.tier = 1;
        HIPCHK(clh::launch_poa(Q, pl->nslots_big, st));
    }
    {   // what the packed kernel put on the wide list (copies above 2800 bases, scores outside the 16-bit cells, a cell at the floor of
        // the range): the 32-bit form, over worst-case slots; nothing on the list: the waves leave at once
        clh::CcsParams Q = P;
        Q.work_counter = (int*)pl->d_counter + 33; Q.tier = 2; Q.n_big = 0;
        int slots = pl->nslots;
        if (pl->nslots_big) { Q.poa_ws = (uint8_t*)pl->d_ws_big; Q.slot_bytes = pl->slot_bytes_big; slots = pl->nslots_big; }
        HIPCHK(clh::launch_poa_wide(Q, std::min(slots, 512), st));
    }
    return 0;
}

extern "C" int clh_ccs_run(clh_ccs_plan* pl, const void* d_reads, void* stream_)
{
    if (!pl || !d_reads) return fail(CLH_E_ARG, "clh_ccs_run: null argument");
    HIPCHK(hipSetDevice(pl->ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : pl->ctx->stream;
    if (pl->n == 0) { pl->ran = true; pl->last_stream = st; return 0; }
    clh::CcsParams P;
    memset(&P, 0, sizeof(P));
    P.reads = (const int8_t*)d_reads; P.read_off = (const int64_t*)pl->d_off; P.scan = (clh::CcsScan*)pl->d_scan;
    P.results = (clh::CcsResult*)pl->d_res; P.segs = (int32_t*)pl->d_segs; P.ccs = (int8_t*)pl->d_ccs;
    P.n = pl->n; P.lcap = pl->lcap;
    P.long_idx = (const int32_t*)pl->d_long; P.k2_ws = (uint8_t*)pl->d_k2ws; P.k2_slot = clh::k2_long_slot_bytes(pl->lmax);
    P.n_long = pl->n_long; P.k2_lmax = pl->lmax; P.k2_lds_max = clh::kK2LdsMax;
    P.sc = ccs_scores();
    if (getenv("CLH_POA_FORCE_WIDE")) P.sc.algorithm |= 0x200;      // tests: every read through the wide (32-bit) form of the pass
    P.aln_score = (int32_t*)pl->d_score;
    if (!pl->ev[0]) for (auto& e : pl->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipMemsetAsync(pl->d_counter, 0, 256, st));        // the two work counters and every statistic of the run
    const bool trace = getenv("CLH_TRACE") != nullptr;
    HIPCHK(hipEventRecord(pl->ev[0], st));
    P.work_order = (const int32_t*)pl->d_order;
    for (size_t k = 0; k < pl->k2_classes.size(); ++k) {
        P.k2_begin = pl->k2_classes[k].begin; P.lcap = pl->k2_classes[k].lcap;
        HIPCHK(clh::launch_ccs_scan(P, pl->k2_classes[k].count, k == 0, st));
    }
    P.lcap = pl->lcap;
    if (trace) { fprintf(stderr, "[clh] K2 launched (n=%d lcap=%d)\n", pl->n, pl->lcap); HIPCHK(hipStreamSynchronize(st)); fprintf(stderr, "[clh] K2 done\n"); }
    HIPCHK(hipEventRecord(pl->ev[1], st));
    // K3's work list: by read length (the plan's order).  By estimated cost after K2, heaviest first (CLH_POA_ORDER_BY_COST=1), measured
    // SLOWER: 19.3 -> 20.8 ms on C3, 76.5 -> 85.3 on C4 -- with the heavy reads all at the start every wave of a SIMD is in its pass
    // at once and nothing is left to fill the latency of the others' walks; the mixed order overlaps better than the balanced one
    static const bool by_len = getenv("CLH_POA_ORDER_BY_COST") == nullptr;
    if (!by_len) HIPCHK(clh::launch_ccs_work_order((const clh::CcsScan*)pl->d_scan, pl->n, (int32_t*)pl->d_order3, st));
    if (int rc = launch_poa_tiers(pl, P, st, !by_len)) return rc;
    if (trace) { fprintf(stderr, "[clh] K3 launched (slots %d x %zu, big %d x %zu)\n", pl->nslots, pl->slot_bytes, pl->nslots_big, pl->slot_bytes_big); HIPCHK(hipStreamSynchronize(st)); fprintf(stderr, "[clh] K3 done\n"); }
    HIPCHK(hipEventRecord(pl->ev[2], st));
    pl->last_stream = st; pl->ran = true;
    return 0;
}

// durations of the last run's two launches in ms: ms[0] = K2 ccs_scan_kernel, ms[1] = K3 poa_consensus_kernel
extern "C" int clh_ccs_plan_timing(clh_ccs_plan* pl, float* ms)
{
    if (!pl || !pl->ran || !pl->ev[0] || !ms) return fail(CLH_E_ARG, "no timed run");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->last_stream));
    HIPCHK(hipEventElapsedTime(&ms[0], pl->ev[0], pl->ev[1]));
    HIPCHK(hipEventElapsedTime(&ms[1], pl->ev[1], pl->ev[2]));
    return 0;
}

extern "C" int clh_ccs_fetch(clh_ccs_plan* pl, clh_ccs_t* out, int32_t* segs, int8_t* ccs)
{
    if (!pl || !out) return fail(CLH_E_ARG, "clh_ccs_fetch: null argument");
    if (!pl->ran) return fail(CLH_E_ARG, "clh_ccs_fetch before clh_ccs_run");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->last_stream));
    if (pl->n == 0) return 0;
    static_assert(sizeof(clh_ccs_t) == sizeof(clh::CcsResult), "result layout");
    HIPCHK(hipMemcpy(out, pl->d_res, sizeof(clh::CcsResult) * (size_t)pl->n, hipMemcpyDeviceToHost));
    if (segs) HIPCHK(hipMemcpy(segs, pl->d_segs, sizeof(int32_t) * 2 * clh::CCS_SEG_CAP * (size_t)pl->n, hipMemcpyDeviceToHost));
    if (ccs && pl->total > 0) HIPCHK(hipMemcpy(ccs, pl->d_ccs, (size_t)pl->total, hipMemcpyDeviceToHost));
    return 0;
}

// how the workspace tiers of the plan were used by the last run: out = {first-tier slots, bytes per slot, large slots,
// bytes per large slot, reads that ran in a large slot claimed by a first-tier wave, reads run by the second launch}
extern "C" int clh_ccs_plan_info(clh_ccs_plan* pl, int64_t* out)
{
    if (!pl || !out) return fail(CLH_E_ARG, "clh_ccs_plan_info: null argument");
    out[0] = pl->nslots; out[1] = (int64_t)pl->slot_bytes; out[2] = pl->nslots_big; out[3] = (int64_t)pl->slot_bytes_big; out[4] = out[5] = 0;
    if (pl->ran && pl->n > 0) {
        HIPCHK(hipSetDevice(pl->ctx->device));
        HIPCHK(hipStreamSynchronize(pl->last_stream));
        int st[2] = {0, 0};
        HIPCHK(hipMemcpy(st, (int*)pl->d_counter + 2, sizeof(st), hipMemcpyDeviceToHost));
        out[4] = st[0]; out[5] = st[1];
    }
    return 0;
}

// statistics of the last run: out[16] = {DP cells, DP row steps, alignments run a second time with every cell stored, reads per status 1..7 (lost to a limit of the kernel: 1
// workspace, 2 graph limits, 3 output, 4 sequence above 2800 bases, 5 back-track guard, 6 16-bit range, 7 alignment without a
// base), 0...}
extern "C" int clh_ccs_plan_stats(clh_ccs_plan* pl, int64_t* out)
{
    if (!pl || !out) return fail(CLH_E_ARG, "clh_ccs_plan_stats: null argument");
    for (int k = 0; k < 16; ++k) out[k] = 0;
    if (pl->ran && pl->n > 0) {
        HIPCHK(hipSetDevice(pl->ctx->device));
        HIPCHK(hipStreamSynchronize(pl->last_stream));
        int st[64];
        HIPCHK(hipMemcpy(st, pl->d_counter, sizeof(st), hipMemcpyDeviceToHost));
        const int* s2 = st + 2;                                  // P.stats
        out[0] = (int64_t)((unsigned long long)(unsigned)s2[2] | ((unsigned long long)(unsigned)s2[3] << 32));
        out[1] = (int64_t)((unsigned long long)(unsigned)s2[4] | ((unsigned long long)(unsigned)s2[5] << 32));
        out[2] = s2[6];                                          // alignments whose walk left the band of stored cells (pass run again, everything stored)
        for (int k = 1; k <= 7; ++k) out[2 + k] = s2[8 + k];
    }
    return 0;
}

// device pointers of the last run's outputs, for callers that keep post-processing on the GPU: rows (clh_ccs_t[n]),
// segs (int32[n][2*65]) and the packed consensus codes (offsets = the read offsets of the plan)
extern "C" int clh_ccs_results_dev(const clh_ccs_plan* pl, const void** rows, const void** segs, const void** ccs)
{
    if (!pl) return fail(CLH_E_ARG, "clh_ccs_results_dev: null plan");
    if (rows) *rows = pl->d_res;
    if (segs) *segs = pl->d_segs;
    if (ccs) *ccs = pl->d_ccs;
    return 0;
}

extern "C" int clh_ccs_batch(clh_ctx* ctx, int32_t n, const int8_t* reads, const int64_t* read_off, clh_ccs_t* out, int32_t* segs, int8_t* ccs)
{
    if (!ctx || !reads || !read_off || !out) return fail(CLH_E_ARG, "clh_ccs_batch: null argument");
    static const bool trace = getenv("CLH_FILE_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now() : 0;
    clh_ccs_plan* pl = clh_ccs_plan_create(ctx, n, read_off);
    if (!pl) return CLH_E_ARG;
    const double t1 = trace ? now() : 0;
    int rc = 0;
    pl->d_reads = ctx->alloc((size_t)read_off[n] + 64);
    if (!pl->d_reads) rc = fail(CLH_E_HIP, "out of device memory for the batch");
    if (!rc && read_off[n] > 0 && hipMemcpyAsync(pl->d_reads, reads, (size_t)read_off[n], hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = fail(CLH_E_HIP, "H2D reads failed");
    if (!rc) rc = clh_ccs_run(pl, pl->d_reads, nullptr);
    double t2 = 0, t3 = 0;
    if (trace) { t2 = now(); (void)hipStreamSynchronize(ctx->stream); t3 = now(); }
    if (!rc) rc = clh_ccs_fetch(pl, out, segs, ccs);
    const double t4 = trace ? now() : 0;
    clh_ccs_plan_destroy(pl);
    if (trace) fprintf(stderr, "[clh] ccs batch of %d reads: plan %.2f ms, H2D + launches %.2f, kernels (wait) %.2f, fetch %.2f, destroy %.2f\n", n,
                       (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (now() - t4) * 1e3);
    return rc;
}

extern "C" int clh_poa_last_stats(clh_ctx* ctx, int64_t* out)
{
    if (!ctx || !out) return fail(CLH_E_ARG, "clh_poa_last_stats: null argument");
    for (int k = 0; k < 16; ++k) out[k] = ctx->last_poa_stats[k];
    return 0;
}

// spoa's AlignmentEngine::Create rules + what the kernel's 16-bit cells can hold
static int poa_check_opts(const clh_poa_opts* o, clh::PoaScores* s)
{
    clh_poa_opts d;
    d.algorithm = 0; d.m = 10; d.n = -4; d.g = -8; d.e = -2; d.q = -24; d.c = -1; d.min_coverage = 0;
    if (o) d = *o;
    if (d.algorithm < 0 || d.algorithm > 2) return fail(CLH_E_ARG, "poa: algorithm must be 0 (local), 1 (global) or 2 (overlap)");
    if (d.g > 0 || d.q > 0) return fail(CLH_E_ARG, "poa: gap opening penalty must be non-positive");
    if (d.e > 0 || d.c > 0) return fail(CLH_E_ARG, "poa: gap extension penalty must be non-positive");
    if (d.g >= d.e) return fail(CLH_E_UNSUPPORTED, "poa: linear gap cost (g >= e) is not built into the kernel");
    if (d.g <= d.q || d.e >= d.c) { d.q = d.g; d.c = d.e; }          // affine: one piece
    if (d.m < 1 || d.n > d.m) return fail(CLH_E_UNSUPPORTED, "poa: match score must be positive and not below the mismatch score");
    if (d.e - d.g > 6 || d.c - d.q > 30) return fail(CLH_E_UNSUPPORTED, "poa: e - g <= 6 and c - q <= 30 required (vertical gap states are kept as small differences)");
    if (d.m > 100000 || d.n < -100000 || d.g < -100000 || d.q < -100000) return fail(CLH_E_UNSUPPORTED, "poa: scores beyond +-100000");
    // scores that leave the 16-bit cells of the packed pass (match above 11, a mismatch or gap extension that drives 2800 bases below
    // -30000, the row scans' frames H - j*e, H - j*c over the 512 columns of a pass) run the wide form of the pass for every sequence
    bool wide = d.m > 11 || d.n < -100;
    wide = wide || std::max(d.g + 2799 * d.e, d.q + 2799 * d.c) < -30000;
    wide = wide || d.m * 2800 + 512 * std::max(-d.e, -d.c) > 32767;
    if (getenv("CLH_POA_FORCE_WIDE")) wide = true;
    if (d.min_coverage < 0) return fail(CLH_E_ARG, "poa: min_coverage must be >= 0");
    s->algorithm = d.algorithm | (wide ? 0x200 : 0); s->m = d.m; s->n = d.n; s->g = d.g; s->e = d.e; s->q = d.q; s->c = d.c; s->min_cov = d.min_coverage;
    return 0;
}

// consensus of explicit groups of sequences (the spoa.poa call shape): group k = sequences [group_off[k], group_off[k+1])
extern "C" int clh_poa_batch(clh_ctx* ctx, int32_t ngroups, const int8_t* seqs, const int64_t* seq_off, const int64_t* group_off,
                             const clh_poa_opts* opts, int32_t* out_len, int8_t* out_ccs, int32_t* msa_col, int32_t* msa_ncols, int32_t* aln_score)
{
    if (!ctx || ngroups < 0 || !seqs || !seq_off || !group_off || !out_len || !out_ccs) return fail(CLH_E_ARG, "clh_poa_batch: null argument");
    clh::PoaScores sc;
    if (int rc = poa_check_opts(opts, &sc)) return rc;
    std::vector<int64_t> roff((size_t)ngroups + 1), xoff((size_t)ngroups + 1);
    std::vector<int32_t> xcuts;
    for (int k = 0; k < ngroups; ++k) {
        const int64_t s0 = group_off[k], s1 = group_off[k + 1];
        if (s1 - s0 < 1) return fail(CLH_E_ARG, "a consensus group must hold at least one sequence");
        if (seq_off[s1] - seq_off[s0] > (1 << 24)) return fail(CLH_E_UNSUPPORTED, "a consensus group above 16 M bases");
        roff[k] = seq_off[s0];
        xoff[k] = (int64_t)xcuts.size();
        for (int64_t i = s0 + 1; i < s1; ++i) xcuts.push_back((int32_t)(seq_off[i] - seq_off[s0]));
    }
    xoff[ngroups] = (int64_t)xcuts.size();
    roff[ngroups] = ngroups ? seq_off[group_off[ngroups]] : 0;
    // groups must tile the packed array contiguously
    for (int k = 0; k + 1 < ngroups; ++k) if (seq_off[group_off[k + 1]] != roff[k + 1]) return fail(CLH_E_ARG, "groups must be contiguous");
    if (ngroups == 0) return 0;
    int mcap = 1;
    for (int64_t i = 0; i < group_off[ngroups]; ++i) mcap = std::max<int>(mcap, (int)(seq_off[i + 1] - seq_off[i]));
    // (global and overlap alignments can sink below the 16-bit cells with any scores: their large slots are sized for the wide form, so that
    // a group the packed kernel hands over finds room there)
    clh_ccs_plan* pl = ccs_plan_create(ctx, ngroups, roff.data(), mcap, (sc.algorithm & 0x200) != 0 || (sc.algorithm & 0xff) != 0);
    if (!pl) return CLH_E_ARG;
    int rc = 0;
    const size_t total = (size_t)roff[ngroups];
    pl->d_reads = ctx->alloc(total + 64);
    void* d_xcuts = ctx->alloc(sizeof(int32_t) * (xcuts.size() + 1));
    void* d_xoff = ctx->alloc(sizeof(int64_t) * (size_t)(ngroups + 1));
    void* d_col = msa_col ? ctx->alloc(sizeof(int32_t) * (total + 1)) : nullptr;
    void* d_ncols = msa_col ? ctx->alloc(sizeof(int32_t) * (size_t)ngroups) : nullptr;
    if (aln_score) pl->d_score = ctx->alloc(sizeof(int32_t) * clh::CCS_SEG_CAP * (size_t)ngroups);
    hipStream_t st = ctx->stream;
    if (!pl->d_reads || !d_xcuts || !d_xoff || (msa_col && (!d_col || !d_ncols)) || (aln_score && !pl->d_score)) rc = fail(CLH_E_HIP, "out of device memory");
    if (!rc && hipMemcpyAsync(pl->d_reads, seqs, total, hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(CLH_E_HIP, "H2D failed");
    if (!rc && !xcuts.empty() && hipMemcpyAsync(d_xcuts, xcuts.data(), sizeof(int32_t) * xcuts.size(), hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(CLH_E_HIP, "H2D failed");
    if (!rc && hipMemcpyAsync(d_xoff, xoff.data(), sizeof(int64_t) * (size_t)(ngroups + 1), hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(CLH_E_HIP, "H2D failed");
    if (!rc && aln_score && hipMemsetAsync(pl->d_score, 0, sizeof(int32_t) * clh::CCS_SEG_CAP * (size_t)ngroups, st) != hipSuccess) rc = fail(CLH_E_HIP, "memset failed");
    if (!rc) {
        clh::CcsParams P;
        memset(&P, 0, sizeof(P));
        P.reads = (const int8_t*)pl->d_reads; P.read_off = (const int64_t*)pl->d_off; P.scan = (clh::CcsScan*)pl->d_scan;
        P.results = (clh::CcsResult*)pl->d_res; P.segs = (int32_t*)pl->d_segs; P.ccs = (int8_t*)pl->d_ccs;
        P.sc = sc; P.xcuts = (const int32_t*)d_xcuts; P.xcut_off = (const int64_t*)d_xoff;
        P.msa_col = (int32_t*)d_col; P.msa_ncols = (int32_t*)d_ncols; P.aln_score = (int32_t*)pl->d_score;
        if (hipMemsetAsync(pl->d_counter, 0, 256, st) != hipSuccess) rc = fail(CLH_E_HIP, "memset failed");
        if (!rc) rc = launch_poa_tiers(pl, P, st);
        pl->ran = true; pl->last_stream = st;
    }
    if (!rc) {
        std::vector<clh_ccs_t> res((size_t)ngroups);
        rc = clh_ccs_fetch(pl, res.data(), nullptr, out_ccs);
        for (int k = 0; k < ngroups && !rc; ++k) out_len[k] = res[k].nseg > 0 && res[k].status == 0 ? res[k].ccs_len : -(1 + res[k].status);
        if (!rc && msa_col && (hipMemcpy(msa_col, d_col, sizeof(int32_t) * total, hipMemcpyDeviceToHost) != hipSuccess ||
                               (msa_ncols && hipMemcpy(msa_ncols, d_ncols, sizeof(int32_t) * (size_t)ngroups, hipMemcpyDeviceToHost) != hipSuccess)))
            rc = fail(CLH_E_HIP, "D2H failed");
        if (!rc && aln_score && hipMemcpy(aln_score, pl->d_score, sizeof(int32_t) * clh::CCS_SEG_CAP * (size_t)ngroups, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(CLH_E_HIP, "D2H failed");
    } else (void)hipStreamSynchronize(st);
    if (!rc) (void)clh_ccs_plan_stats(pl, ctx->last_poa_stats);
    ctx->release(d_xcuts); ctx->release(d_xoff); ctx->release(d_col); ctx->release(d_ncols);
    clh_ccs_plan_destroy(pl);
    return rc;
}
