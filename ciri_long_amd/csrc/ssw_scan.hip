// ssw_scan.hip -- K1s: Smith-Waterman scores and coordinates for SHORT reads, one alignment per wavefront (gfx950).
//
// Same answers as ssw_wavefront.hip (reference: libs/striped_smith_waterman/ssw.c:123-345 sw_sse2_byte and the
// forward + reverse orchestration of ssw_align, ssw.c:779-849; row-major statement: oracle/rowmajor_spec.c), for the
// alignments the host sorts into this class (clh_api.hip: scan_class_ok):
//     read <= 254 bases,  max_match * readLen + bias < 255  (ssw.c:804-806 then runs the 8-bit pass and that pass cannot
//     overflow: every score fits 8 bits, no 16-bit re-run, no stripe quirk),  gap_extend <= 16.
// These are the 20..300-base clips of find_bsj.py:191-216 against windows of 2 kb .. 400 kb -- most of the SSW calls of
// `call`.  The anti-diagonal kernel spends a fixed cost per step on handing three values down the lanes; with one row
// per virtual lane (reads <= 128 bases) that fixed cost is most of the step, and 127 fill/drain steps per pass are wasted.
//
// Here the matrix is walked the other way round: the LANES own reference columns, the loop runs over the read's rows.
//   * a wave is 128 virtual lanes (the 16-bit halves of every VGPR); virtual lane v owns CPR adjacent columns of a chunk
//     of 128*CPR columns (CPR = 2, 4 or 8 by what is left of the window): register t, low halves = columns
//     2*CPR*lane + t, high halves = columns 2*CPR*lane + CPR + t -- the left neighbour of a cell is the same half of the
//     previous register;
//   * one row step: diagonal + substitution score (profile of the chunk's columns in LDS, indexed by the row's base),
//     the gap from the row above (F; registers), then the gap along the row (E) as a prefix maximum: a dependent chain
//     inside the CPR columns of a virtual lane, one wave-wide scan (6 DPP max) of the per-lane "E leaving the lane"
//     in the frame where a gap extension costs nothing, and the incoming E applied to the lane's columns;
//   * per column the running maximum over the rows is kept as (H << 8 | 255 - row): one unsigned maximum gives the
//     column maximum AND the smallest row that holds it (what ssw.c:299-308 searches at the end);
//   * chunks follow each other left to right; the last column of a chunk (H and the E leaving it, per row) is handed to
//     the next chunk through 1 KiB of LDS.  The reverse pass (ssw.c:834-849) ends at the first column whose maximum
//     equals the forward score: it runs in chunks of 256 columns and stops after the chunk that holds that column.
// No fill or drain, no per-cell hand-down: 9 packed operations per cell pair (gap_open == gap_extend) or 13, plus ~40 per
// row step, against ~26 per 128 cells in the anti-diagonal form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"
#include "clh_device_ops.h"

namespace clh {

namespace {



#ifndef SCAN_WAVES
#define SCAN_WAVES 3      // measured on the C3 clip batch: 2 -> 1.87 ms, 3 -> 1.76 ms, 4 -> 1.88 ms
#endif
static constexpr int SCAN_MAX_ROWS = 256;        // rows of a pass incl. the wildcard rows (reads <= 254 bases)
static constexpr int SCAN_CPR_MAX = 8;
static constexpr int SCAN_PROF_WORDS = 6 * SCAN_CPR_MAX * 64;      // uint32 per wave: [query code 0..5][register][lane]

struct ScanIn {
    const int8_t* read;   // first row's base
    int rstep;            // +1 / -1
    int L;                // rows of the read
    int rows;             // rows processed: L, or L padded to a multiple of 16 with wildcard rows (score 0 against everything)
    const int8_t* ref;    // first column's base
    int cstep;            // +1 / -1
    int comp;             // reference bytes are complemented as they are read
    int ncols;
    int terminate;        // column maximum that ends the pass (ssw.c:296); > 254 = never
    uint16_t* colmax;     // per-column maxima in processing order (indexed by col_base + column), or nullptr
    int own0;             // columns below this one are computed but do not count (the overlap of a window slice); 0 otherwise
    int col_base;         // column number of the first column handed in (a window slice); 0 otherwise
};
struct ScanOut { int max, col, row; };

struct ScanLds {
    uint32_t* prof;       // SCAN_PROF_WORDS
    const int* mat;       // [6 reference codes][8 query codes]
    short* bH;            // [2][SCAN_MAX_ROWS]: H of the chunk's last column per row (parity of the chunk)
    short* bE;            // [2][SCAN_MAX_ROWS]: E entering the next chunk's first column per row
};

// one chunk of 128*CPR columns starting at column c0; returns through best_* the best cell so far (first column wins)
// and through tcol the first column of the chunk whose maximum equals in.terminate (or INT_MAX)
template <int CPR, bool GEQ>
__device__ void scan_chunk(const ScanIn& in, const ScanLds& lds, const int c0, const int parity, const bool first, const bool more,
                           const int gapO, const int gapE, int& best_score, int& best_col, int& best_row, int& tcol)
{
    constexpr int VEC = CPR < 4 ? CPR : 4;       // registers per LDS read
    constexpr int NCH = CPR / VEC;
    const int lane = threadIdx.x & 63;
    const int K = CPR * gapE;                    // what a gap loses across one virtual lane
    const int kLo = 2 * lane * K;
    const uint32_t gO2 = dup16(gapO), gE2 = dup16(gapE);

    // ---- profile of the chunk's columns: prof[q][c][lane][VEC], entry = scores of the row base q against the two columns
    {
        int rlo[CPR], rhi[CPR];
#pragma unroll
        for (int t = 0; t < CPR; ++t) {
            const int jlo = c0 + 2 * CPR * lane + t, jhi = jlo + CPR;
            const int blo = jlo < in.ncols ? (int)in.ref[(int64_t)jlo * in.cstep] : 0, bhi = jhi < in.ncols ? (int)in.ref[(int64_t)jhi * in.cstep] : 0;
            rlo[t] = jlo < in.ncols ? ref_code(blo, in.comp) : 5;
            rhi[t] = jhi < in.ncols ? ref_code(bhi, in.comp) : 5;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int t = 0; t < CPR; ++t) {
                const int slo = lds.mat[rlo[t] * 8 + q], shi = lds.mat[rhi[t] * 8 + q];
                lds.prof[((q * NCH + t / VEC) * 64 + lane) * VEC + (t % VEC)] = (uint32_t)(slo & 0xffff) | ((uint32_t)shi << 16);
            }
    }
    const short* bHin = lds.bH + (parity ^ 1) * SCAN_MAX_ROWS;
    const short* bEin = lds.bE + (parity ^ 1) * SCAN_MAX_ROWS;
    short* bHout = lds.bH + parity * SCAN_MAX_ROWS;
    short* bEout = lds.bE + parity * SCAN_MAX_ROWS;
    __syncthreads();

    uint32_t Hp[CPR], Fst[GEQ ? 1 : CPR], key[CPR];
#pragma unroll
    for (int t = 0; t < CPR; ++t) { Hp[t] = 0; key[t] = 0; if constexpr (!GEQ) Fst[t] = 0; }
    int prev_hb = 0;                             // H[row - 1][c0 - 1]
    for (int rb = 0; rb < in.rows; rb += 64) {
        const int row = rb + lane;
        int qv = 5;
        if (row < in.L) { const int c = (int)in.read[(int64_t)row * in.rstep] & 7; qv = c > 5 ? 5 : c; }
        int hbv = 0, ebv = 0;
        if (!first && row < in.rows) { hbv = bHin[row]; ebv = bEin[row]; }
        const int cnt = in.rows - rb < 64 ? in.rows - rb : 64;
        int cobH = 0, cobE = 0;
        for (int i = 0; i < cnt; ++i) {
            const int q = __builtin_amdgcn_readlane(qv, i), hb = __builtin_amdgcn_readlane(hbv, i), eb = __builtin_amdgcn_readlane(ebv, i);
            uint32_t P[CPR];
            {
                const uint32_t* pp = lds.prof + (q * NCH * 64 + lane) * VEC;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if constexpr (VEC == 4) { const uint4 v = *(const uint4*)(pp + c * 64 * VEC); P[4 * c] = v.x; P[4 * c + 1] = v.y; P[4 * c + 2] = v.z; P[4 * c + 3] = v.w; }
                    else { const uint2 v = *(const uint2*)(pp + c * 64 * VEC); P[0] = v.x; P[1] = v.y; }
                }
            }
            const uint32_t d0 = hand_down(Hp[CPR - 1], prev_hb);
            uint32_t R[CPR];
            uint32_t e = 0, U;
#pragma unroll
            for (int t = 0; t < CPR; ++t) {
                const uint32_t tt = pk_adds(t == 0 ? d0 : Hp[t - 1], P[t]);
                uint32_t Fv;
                if constexpr (GEQ) Fv = pk_subus(Hp[t], gO2);
                else { Fv = pk_max(pk_subus(Fst[t], gE2), pk_subus(Hp[t], gO2)); Fst[t] = Fv; }
                const uint32_t X = pk_max(tt, Fv);                      // >= 0: Fv is
                if (t == 0) R[0] = X;
                else if constexpr (GEQ) R[t] = pk_max(X, pk_subus(R[t - 1], gO2));
                else { e = pk_max(pk_subus(e, gE2), pk_subus(R[t - 1], gO2)); R[t] = pk_max(X, e); }
            }
            if (GEQ) U = pk_subus(R[CPR - 1], gO2);
            else U = pk_max(pk_subus(e, gE2), pk_subus(R[CPR - 1], gO2));   // E leaving the virtual lane, from its own columns
            // E entering every virtual lane: prefix maximum of U in the frame where crossing a virtual lane costs nothing
            const int Blo = (int)(U & 0xffffu) + kLo, Bhi = (int)(U >> 16) + kLo + K;
            const int inc = wave_prefix_max(Blo > Bhi ? Blo : Bhi);
            const int fill = eb - K;                                    // the previous chunk, as virtual lane -1
            int exc = dpp_shr1(fill, inc);
            exc = exc > fill ? exc : fill;
            const int einLo = exc - kLo + K;
            const int m2 = exc > Blo ? exc : Blo;
            const int einHi = m2 - kLo;
            uint32_t Ein = ((uint32_t)einLo & 0xffffu) | ((uint32_t)einHi << 16);
            const uint32_t rowc = dup16(255 - (rb + i));
#pragma unroll
            for (int t = 0; t < CPR; ++t) {
                const uint32_t h = pk_max(R[t], Ein);
                Ein = pk_subs(Ein, gE2);
                Hp[t] = h;
                key[t] = pk_maxu(key[t], pk_madu(h, 0x01000100u, rowc));
            }
            prev_hb = hb;
            if (more) {   // the chunk's last column: H, and the E that enters the next chunk's first column
                const int outH = (int)((uint32_t)__builtin_amdgcn_readlane((int)Hp[CPR - 1], 63) >> 16);
                int outE = __builtin_amdgcn_readlane(inc, 63);
                outE = (outE > fill ? outE : fill) - 127 * K;
                outE = outE < 0 ? 0 : outE;
                const uint32_t li = lane_is(lane, i); cobH = set_lane(cobH, outH, li); cobE = set_lane(cobE, outE, li);
            }
        }
        if (more && lane < cnt) { bHout[rb + lane] = (short)cobH; bEout[rb + lane] = (short)cobE; }
    }
    __syncthreads();

    // ---- the chunk's columns: column maxima out, terminate column, best cell (first column wins; smallest row in it) ----
    int tmin = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < CPR; ++t)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int j = c0 + 2 * CPR * lane + hf * CPR + t;
            const int k16 = hf ? (int)(key[t] >> 16) : (int)(key[t] & 0xffffu);
            const int cm = k16 >> 8;
            if (j < in.ncols && j >= in.own0) {
                if (in.colmax) in.colmax[in.col_base + j] = (uint16_t)cm;
                if (cm == in.terminate) tmin = j < tmin ? j : tmin;
            }
        }
    tmin = wave_min(tmin);
    int b32 = 0;
#pragma unroll
    for (int t = 0; t < CPR; ++t)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int jc = 2 * CPR * lane + hf * CPR + t, j = c0 + jc;
            const int k16 = hf ? (int)(key[t] >> 16) : (int)(key[t] & 0xffffu);
            const int v = ((k16 >> 8) << 20) | ((0xfff - jc) << 8) | (k16 & 0xff);
            if (j < in.ncols && j <= tmin && j >= in.own0) b32 = v > b32 ? v : b32;
        }
    b32 = wave_max(b32);
    const int sc = b32 >> 20;
    if (sc > best_score) { best_score = sc; best_col = c0 + (0xfff - ((b32 >> 8) & 0xfff)); best_row = 255 - (b32 & 0xff); }
    tcol = tmin;
}

template <bool GEQ>
__device__ ScanOut scan_pass(const ScanIn& in, const ScanLds& lds, const int gapO, const int gapE)
{
    int best_score = 0, best_col = -1, best_row = 0;
    const bool ends = in.terminate <= 254;       // reverse pass: stops at the first column whose maximum is the forward score
    int parity = 0;
    for (int c0 = 0; c0 < in.ncols; parity ^= 1) {
        const int rem = in.ncols - c0;
        int tcol = 0x7fffffff;
        const bool first = c0 == 0;
        if (ends || rem <= 256) { scan_chunk<2, GEQ>(in, lds, c0, parity, first, rem > 256, gapO, gapE, best_score, best_col, best_row, tcol); c0 += 256; }
        else if (rem <= 512) { scan_chunk<4, GEQ>(in, lds, c0, parity, first, false, gapO, gapE, best_score, best_col, best_row, tcol); c0 += 512; }
        else { scan_chunk<8, GEQ>(in, lds, c0, parity, first, rem > 1024, gapO, gapE, best_score, best_col, best_row, tcol); c0 += 1024; }
        if (tcol != 0x7fffffff) break;
    }
    ScanOut o;
    o.max = best_score;
    if (best_score == 0) { o.col = -1; o.row = 0; return o; }
    o.col = in.col_base + best_col;
    o.row = best_row < in.L - 1 ? best_row : in.L - 1;
    return o;
}

// masked second-best column maximum, ssw.c:325-340 (8-bit pass); wave-parallel
__device__ void second_best8(const uint16_t* colmax, int refLen, int end_ref, int maskLen, int& score2, int& ref_end2)
{
    const int lane = threadIdx.x & 63;
    int e1 = end_ref - maskLen; if (e1 < 0) e1 = 0;
    int e2 = end_ref + maskLen; if (e2 > refLen) e2 = refLen;
    e2 += 1;
    int bv = 0, bp = 0x7fffffff;
    for (int i = lane; i < refLen; i += 64) {
        if (i < e1 || i >= e2) {
            const int v = colmax[i];
            if (v > bv) { bv = v; bp = i; }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v2 = __shfl_xor(bv, d), p2 = __shfl_xor(bp, d);
        const bool take = v2 > bv || (v2 == bv && p2 < bp);
        bv = take ? v2 : bv; bp = take ? p2 : bp;
    }
    score2 = bv;
    ref_end2 = bv > 0 ? bp : 0;
}

// everything after the forward pass: second best, reverse pass (begin coordinates), the result row
template <bool GEQ>
__device__ void scan_finish(const SswParams& p, const SswTask& task, const ScanOut& fw, const ScanLds& lds)
{
    const int lane = threadIdx.x & 63;
    const int8_t* read = p.reads + task.read_off;
    const int8_t* ref = p.refs + task.ref_off;
    const int refLen = task.ref_len;
    uint16_t* colmax = p.colmax ? p.colmax + task.colmax_off : nullptr;
    const int rdir = task.ref_rc ? -1 : 1;
    SswResult res;
    res.score2 = 0; res.ref_begin1 = -1; res.ref_end2 = 0; res.status = 0;
    res.score1 = fw.max;
    if (fw.max == 0) { res.ref_end1 = -1; res.read_end1 = 0; }
    else { res.ref_end1 = fw.col; res.read_end1 = fw.row; }
    if (task.mask_len >= 15 && colmax) { __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); second_best8(colmax, refLen, res.ref_end1, task.mask_len, res.score2, res.ref_end2); }
    else { res.score2 = 0; res.ref_end2 = task.mask_len >= 15 ? 0 : -1; }

    // ---- reverse: begin coordinates (ssw.c:834-849) ---------------------------------------------------------------
    res.read_begin1 = -1;
    const bool want_begin = !(p.flag == 0 || (p.flag == 2 && res.score1 < p.filters));
    if (want_begin) {
        ScanIn rv;
        rv.L = res.read_end1 + 1; rv.rows = ((rv.L + 15) / 16) * 16; rv.read = read + res.read_end1; rv.rstep = -1;
        rv.ncols = res.ref_end1 + 1; rv.ref = ref + (int64_t)res.ref_end1 * rdir; rv.cstep = -rdir; rv.comp = task.ref_rc;
        rv.terminate = res.score1; rv.colmax = nullptr; rv.own0 = 0; rv.col_base = 0;
        const ScanOut r = scan_pass<GEQ>(rv, lds, p.gapO, p.gapE);
        if (r.max == 0) { res.ref_begin1 = -1; res.read_begin1 = res.read_end1; }
        else { res.ref_begin1 = res.ref_end1 - r.col; res.read_begin1 = res.read_end1 - r.row; }
    }
    if (lane == 0) p.results[task.out_index] = res;
}

// forward pass over columns [c_begin, c_end) of the task's window, counting from own_begin on (ssw.c:804-822: the 8-bit
// pass; it cannot overflow in this class)
template <bool GEQ>
__device__ ScanOut scan_forward(const SswParams& p, const SswTask& task, const ScanLds& lds, int c_begin, int own_begin, int c_end)
{
    const int rdir = task.ref_rc ? -1 : 1;
    uint16_t* colmax = p.colmax ? p.colmax + task.colmax_off : nullptr;
    ScanIn in;
    in.read = p.reads + task.read_off; in.rstep = 1; in.L = task.read_len;
    in.ref = p.refs + task.ref_off + (int64_t)c_begin * rdir; in.cstep = rdir; in.comp = task.ref_rc; in.ncols = c_end - c_begin;
    in.terminate = 1 << 30; in.colmax = colmax; in.own0 = own_begin - c_begin; in.col_base = c_begin;
    in.rows = colmax ? ((in.L + 15) / 16) * 16 : in.L;     // the wildcard rows only matter to the column maxima (rowmajor_spec.c)
    return scan_pass<GEQ>(in, lds, p.gapO, p.gapE);
}

#define SCAN_LDS_SETUP \
    __shared__ __attribute__((aligned(16))) uint32_t s_prof[SCAN_PROF_WORDS]; \
    __shared__ int s_mat[48]; \
    __shared__ short s_bH[2 * SCAN_MAX_ROWS], s_bE[2 * SCAN_MAX_ROWS]; \
    { const int lane_ = threadIdx.x & 63; \
      if (lane_ < 48) { const int b_ = lane_ >> 3, q_ = lane_ & 7; s_mat[lane_] = (b_ < p.n && q_ < p.n) ? (int)p.mat[b_ * p.n + q_] : 0; } } \
    __syncthreads(); \
    ScanLds lds; lds.prof = s_prof; lds.mat = s_mat; lds.bH = s_bH; lds.bE = s_bE;

}  // namespace

template <bool GEQ>
__global__ void __launch_bounds__(64, SCAN_WAVES) ssw_scan_kernel(const SswParams p)
{
    SCAN_LDS_SETUP
    const SswTask task = p.tasks[blockIdx.x];
    const ScanOut fw = scan_forward<GEQ>(p, task, lds, 0, 0, task.ref_len);
    scan_finish<GEQ>(p, task, fw, lds);
}

// Long windows (clh_api.hip: scan_slices): the forward pass of one alignment is cut into slices of the window that run as
// separate workgroups.  A local alignment of an L-base read spans at most L * (1 + max_match / gap_extend) columns (every
// deleted reference base costs at least gap_extend, the read can earn at most L * max_match), so a slice that starts that many
// columns (+ the 16 wildcard rows) before the columns it owns computes exactly the H of the whole-window pass there.  Each
// slice leaves its best cell; the finishing kernel takes the largest (first column on ties, ssw.c:283) and goes on as usual.
template <bool GEQ>
__global__ void __launch_bounds__(64, SCAN_WAVES) ssw_scan_slice_kernel(const SswParams p)
{
    SCAN_LDS_SETUP
    const ScanSlice sl = p.slices[blockIdx.x];
    const SswTask task = p.tasks[sl.task];
    const ScanOut fw = scan_forward<GEQ>(p, task, lds, sl.c_begin, sl.own_begin, sl.c_end);
    if ((threadIdx.x & 63) == 0) { ScanPart pt; pt.max = fw.max; pt.col = fw.col; pt.row = fw.row; pt.pad = 0; p.parts[sl.part] = pt; }
}

template <bool GEQ>
__global__ void __launch_bounds__(64, SCAN_WAVES) ssw_scan_finish_kernel(const SswParams p)
{
    SCAN_LDS_SETUP
    const int lane = threadIdx.x & 63;
    const SswTask task = p.tasks[blockIdx.x];
    const int first = (int)task.dir_off, ns = task.pad;       // this task's slices (clh_api.hip: at most 64)
    int v = 0, c = 0x7fffffff, r = 0;
    if (lane < ns) { const ScanPart pt = p.parts[first + lane]; v = pt.max; c = pt.max > 0 ? pt.col : 0x7fffffff; r = pt.row; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v2 = __shfl_xor(v, d), c2 = __shfl_xor(c, d), r2 = __shfl_xor(r, d);
        const bool take = v2 > v || (v2 == v && c2 < c);
        v = take ? v2 : v; c = take ? c2 : c; r = take ? r2 : r;
    }
    ScanOut fw; fw.max = v; fw.col = v > 0 ? c : -1; fw.row = v > 0 ? r : 0;
    scan_finish<GEQ>(p, task, fw, lds);
}

// ---- the sliced class behind the prefilter (ssw_prefilter.hip; tools/prefilter_model.py) --------------------------------------
// One wave per task: (1) the block with the smallest minimum of d; a forward pass around it ATTAINS a score S0; (2) every
// block whose minimum is at most (M L - S0) / c is a candidate -- no other block can hold the maximum or tie it; runs of
// candidate blocks (cut at groups of 64) become slices in the queue, each started `overlap` columns early like a static slice;
// (3) when the runs outnumber the task's share of the queue or cover about as much as the window, the static slices of
// clh_api.hip are written instead.  p.pf_dmin == nullptr (filter off for this run): static slices at once.
// What the pick kernels share: with the seed's score S0 in hand, the candidate blocks of the alignment (minimum <= (M L - S0) / c), their
// runs as slices in the queue -- or the static slices when the runs outnumber the alignment's share of the queue or cover about as much as
// the window -- and the alignment's PfOut.  stage2_ok: the caller may still send the window to the second stage; returns true when it
// should (the candidates cover more than an eighth of the window: the indel-distance pass over the window costs about as much as the
// score pass over a twelfth of it; and the threshold is below what that distance is on random text) and then writes nothing.
template <bool GEQ>
__device__ bool pf_pick_emit(const SswParams& p, const SswTask& task, const PfWin& pt, const uint8_t* dmin, const int S0, const ScanOut& seed, const int seed_block,
                             const bool stage2_ok)
{
    const int lane = threadIdx.x & 63;
    const int R = task.ref_len, L = task.read_len;
    const int overlap = L + (L * p.max_match + p.gapE - 1) / p.gapE + 32;
    int own = (R + 63) / 64; own = own < 8192 ? 8192 : own;
    const int nstatic = (R + own - 1) / own;
    int nrun = 0, pruned = 0, thr = 0, ov_c = overlap;
    if (dmin && S0 > 0) {
        const int cc = p.max_match < p.gapE ? p.max_match : p.gapE;
        thr = (p.max_match * L - S0) / cc;
        // an alignment that scores S0 or more deletes at most (M L - S0) / gapE window bases: it spans no more than L + that many
        // columns, so the candidate slices start that far (+ 32) early -- cells that cannot reach S0 may come out lower, they cannot win
        const int span_s0 = L + (p.max_match * L - S0) / p.gapE + 32;
        ov_c = span_s0 < overlap ? span_s0 : overlap;
        long long cost = 0;
        for (int g = 0; g < pt.nsub; g += 64) {
            const int k = g + lane;
            const unsigned long long m = __ballot(k < pt.nsub && (int)dmin[k] <= thr);
            nrun += __popcll(m & ~(m << 1));
            cost += (long long)__popcll(m) * kPfBlock;
        }
        cost += (long long)nrun * ov_c;
        const int cap = pt.nsub / 8 + 1 > 64 ? pt.nsub / 8 + 1 : 64;
        pruned = nrun <= cap && cost < (long long)R + (long long)nstatic * overlap;
        // (the indel distance of a clip to random text is ~0.58 L, less for short clips: above 0.55 L the second stage cannot help either)
        if (stage2_ok && (p.pf2_always || ((!pruned || cost > (long long)R / p.pf2_share) && 20 * thr < 11 * L))) return true;
    }
    const int count = pruned ? nrun : nstatic;
    int first = 0;
    if (lane == 0) {
        first = atomicAdd(&p.pf_ctl->qcount, count);
        if (pruned) atomicAdd(&p.pf_ctl->n_pruned, 1);
        atomicAdd(&p.pf_ctl->cols_window, (unsigned long long)R);
    }
    first = __builtin_amdgcn_readfirstlane(first);
    unsigned long long cols = 0;
    if (pruned) {
        int done = 0;
        for (int g = 0; g < pt.nsub; g += 64) {
            const int k = g + lane;
            const unsigned long long m = __ballot(k < pt.nsub && (int)dmin[k] <= thr);
            const unsigned long long starts = m & ~(m << 1);
            if ((starts >> lane) & 1ull) {
                const unsigned long long rest = ~(m >> lane);              // bit 0 is clear: this lane's block is a candidate
                const int len = rest ? __builtin_ctzll(rest) : 64 - lane;
                const int rank = __popcll(starts & ((1ull << lane) - 1ull));
                int b0 = k * kPfBlock - pt.phase, b1 = (k + len) * kPfBlock - pt.phase;
                b0 = b0 < 0 ? 0 : b0; b1 = b1 > R ? R : b1;
                ScanSlice sl;
                sl.task = blockIdx.x; sl.own_begin = b0; sl.c_begin = b0 - ov_c < 0 ? 0 : b0 - ov_c; sl.c_end = b1;
                sl.part = first + done + rank; sl.pad0 = sl.pad1 = sl.pad2 = 0;
                if (len == 1 && k == seed_block) {       // the seed's own region: its best cell is known already
                    ScanPart pt2; pt2.max = seed.max; pt2.col = seed.col; pt2.row = seed.row; pt2.pad = 0;
                    p.parts[sl.part] = pt2;
                    sl.task = -1;
                } else cols += (unsigned long long)(b1 - sl.c_begin);
                p.pf_slices[sl.part] = sl;
            }
            done += __popcll(starts);
        }
    } else {
        for (int s = lane; s < nstatic; s += 64) {
            const long long b = (long long)s * own;
            ScanSlice sl;
            sl.task = blockIdx.x; sl.own_begin = (int)b; sl.c_begin = (int)(b - overlap < 0 ? 0 : b - overlap); sl.c_end = (int)(b + own > R ? R : b + own);
            sl.part = first + s; sl.pad0 = sl.pad1 = sl.pad2 = 0;
            p.pf_slices[sl.part] = sl;
            cols += (unsigned long long)(sl.c_end - sl.c_begin);
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) cols += __shfl_xor(cols, d);
    if (lane == 0) {
        atomicAdd(&p.pf_ctl->cols_scanned, cols);
        PfOut o; o.first = first; o.count = count; o.s0 = S0; o.pruned = pruned;
        p.pf_out[blockIdx.x] = o;
    }
    return false;
}

// ---- the sliced class behind the prefilter (ssw_prefilter.hip; tools/prefilter_model.py) --------------------------------------
// One wave per task: (1) the block with the smallest minimum of d; a forward pass around it ATTAINS a score S0; (2) every
// block whose minimum is at most (M L - S0) / c is a candidate -- no other block can hold the maximum or tie it; runs of
// candidate blocks (cut at groups of 64) become slices in the queue, each started `overlap` columns early like a static slice;
// (3) when the runs outnumber the task's share of the queue or cover about as much as the window, the static slices of
// clh_api.hip are written instead -- or, with a second stage (p.pf_q2), the window's entries of the work list are queued for the
// indel-distance pass and ssw_scan_pick2_kernel decides.  p.pf_dmin == nullptr (filter off for this run): static slices at once.
template <bool GEQ>
__global__ void __launch_bounds__(64, SCAN_WAVES) ssw_scan_pick_kernel(const SswParams p)
{
    SCAN_LDS_SETUP
    const int lane = threadIdx.x & 63;
    const SswTask task = p.tasks[blockIdx.x];
    const PfWin pt = p.pf_win[blockIdx.x];
    const int R = task.ref_len, L = task.read_len;
    const int overlap = L + (L * p.max_match + p.gapE - 1) / p.gapE + 32;
    const uint8_t* dmin = p.pf_dmin ? p.pf_dmin + p.pf_tasks[pt.piece_first].sub_off : nullptr;     // (one piece: reads of this class have <= 254 bases)
    int S0 = 0, seed_block = -1;
    ScanOut seed; seed.max = 0; seed.col = -1; seed.row = 0;
    if (dmin) {
        int key = 0x7fffffff;
        for (int k = lane; k < pt.nsub; k += 64) { const int v = ((int)dmin[k] << 20) | k; key = v < key ? v : key; }
        key = wave_min(key);
        const int kb = key & 0xfffff;
        int c0 = kb * kPfBlock - pt.phase, c1 = c0 + kPfBlock;
        c0 = c0 < 0 ? 0 : c0; c1 = c1 > R ? R : c1;
        const int cb = c0 - overlap < 0 ? 0 : c0 - overlap;
        const ScanOut fw = scan_forward<GEQ>(p, task, lds, cb, c0, c1);
        S0 = fw.max; seed = fw; seed_block = kb;
    }
    if (pf_pick_emit<GEQ>(p, task, pt, dmin, S0, seed, seed_block, p.pf_q2 != nullptr && dmin != nullptr)) {
        // the unit-cost bound leaves too much of this window: its entries of the work list once more, with the indel distance
        int base = 0;
        if (lane == 0) {
            base = atomicAdd(&p.pf_ctl->q2count, pt.work_count);
            atomicAdd(&p.pf_ctl->n_stage2, 1);
            PfOut o; o.first = 0; o.count = 0; o.s0 = S0; o.pruned = 2;          // 2: pending (ssw_scan_pick2_kernel)
            p.pf_out[blockIdx.x] = o;
        }
        base = __builtin_amdgcn_readfirstlane(base);
        for (int k = lane; k < pt.work_count; k += 64) p.pf_q2[base + k] = pt.work_first + k;
    }
}

// after the second stage: the windows it took (PfOut.pruned == 2), by its block minima
template <bool GEQ>
__global__ void __launch_bounds__(64, SCAN_WAVES) ssw_scan_pick2_kernel(const SswParams p)
{
    const PfOut po = p.pf_out[blockIdx.x];
    if (po.pruned != 2) return;
    const SswTask task = p.tasks[blockIdx.x];
    const PfWin pt = p.pf_win[blockIdx.x];
    const uint8_t* dmin = p.pf_dmin + p.pf_tasks[pt.piece_first].sub_off;
    ScanOut seed; seed.max = 0; seed.col = -1; seed.row = 0;
    (void)pf_pick_emit<GEQ>(p, task, pt, dmin, po.s0, seed, -1, false);
}

// the queue's slices by persistent workgroups (the number of slices is only known on the device)
template <bool GEQ>
__global__ void __launch_bounds__(64, SCAN_WAVES) ssw_scan_queue_kernel(const SswParams p)
{
    SCAN_LDS_SETUP
    const int lane = threadIdx.x & 63;
    const int total = p.pf_ctl->qcount;
    for (;;) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(&p.pf_ctl->qnext, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= total) break;
        const ScanSlice sl = p.pf_slices[idx];
        if (sl.task < 0) continue;                 // the seed's region: done by the pick kernel
        const SswTask task = p.tasks[sl.task];
        const ScanOut fw = scan_forward<GEQ>(p, task, lds, sl.c_begin, sl.own_begin, sl.c_end);
        if (lane == 0) { ScanPart pt; pt.max = fw.max; pt.col = fw.col; pt.row = fw.row; pt.pad = 0; p.parts[sl.part] = pt; }
        __syncthreads();
    }
}

template <bool GEQ>
__global__ void __launch_bounds__(64, SCAN_WAVES) ssw_scan_finish_queue_kernel(const SswParams p)
{
    SCAN_LDS_SETUP
    const int lane = threadIdx.x & 63;
    const SswTask task = p.tasks[blockIdx.x];
    const PfOut po = p.pf_out[blockIdx.x];
    int v = 0, c = 0x7fffffff, r = 0;
    for (int k = lane; k < po.count; k += 64) {
        const ScanPart pt = p.parts[po.first + k];
        const int c2 = pt.max > 0 ? pt.col : 0x7fffffff;
        const bool take = pt.max > v || (pt.max == v && c2 < c);
        v = take ? pt.max : v; c = take ? c2 : c; r = take ? pt.row : r;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v2 = __shfl_xor(v, d), c2 = __shfl_xor(c, d), r2 = __shfl_xor(r, d);
        const bool take = v2 > v || (v2 == v && c2 < c);
        v = take ? v2 : v; c = take ? c2 : c; r = take ? r2 : r;
    }
    ScanOut fw; fw.max = v; fw.col = v > 0 ? c : -1; fw.row = v > 0 ? r : 0;
    scan_finish<GEQ>(p, task, fw, lds);
}

hipError_t launch_ssw_scan_filtered(bool geq, const SswParams& p, int ntasks, int nworkgroups, int nworkgroups2, hipStream_t stream)
{
    const bool stage2 = p.pf_q2 != nullptr && p.pf_dmin != nullptr;
    if (geq) {
        hipLaunchKernelGGL((ssw_scan_pick_kernel<true>), dim3(ntasks), dim3(64), 0, stream, p);
        if (stage2) {
            if (hipError_t e = launch_ssw_prefilter_indel(p, nworkgroups2, stream)) return e;
            hipLaunchKernelGGL((ssw_scan_pick2_kernel<true>), dim3(ntasks), dim3(64), 0, stream, p);
        }
        hipLaunchKernelGGL((ssw_scan_queue_kernel<true>), dim3(nworkgroups), dim3(64), 0, stream, p);
        hipLaunchKernelGGL((ssw_scan_finish_queue_kernel<true>), dim3(ntasks), dim3(64), 0, stream, p);
    } else {
        hipLaunchKernelGGL((ssw_scan_pick_kernel<false>), dim3(ntasks), dim3(64), 0, stream, p);
        if (stage2) {
            if (hipError_t e = launch_ssw_prefilter_indel(p, nworkgroups2, stream)) return e;
            hipLaunchKernelGGL((ssw_scan_pick2_kernel<false>), dim3(ntasks), dim3(64), 0, stream, p);
        }
        hipLaunchKernelGGL((ssw_scan_queue_kernel<false>), dim3(nworkgroups), dim3(64), 0, stream, p);
        hipLaunchKernelGGL((ssw_scan_finish_queue_kernel<false>), dim3(ntasks), dim3(64), 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_ssw_scan(bool geq, const SswParams& p, int ntasks, hipStream_t stream)
{
    if (geq) hipLaunchKernelGGL((ssw_scan_kernel<true>), dim3(ntasks), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL((ssw_scan_kernel<false>), dim3(ntasks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// class kRvCombine: the best of an alignment's window-slice tasks (clh_api.hip) becomes its result row: largest score, then
// smallest end column in the whole window, then the earlier slice (the one that owns that column)
__global__ void __launch_bounds__(64) ssw_combine_kernel(const SswParams p)
{
    const int lane = threadIdx.x & 63;
    const SswTask task = p.tasks[blockIdx.x];
    const int first = (int)task.dir_off, ns = task.pad;
    int v = -1, c = 0x7fffffff, k = 0x7fffffff;
    if (lane < ns) {
        const SswResult r = p.results[p.n_real + first + lane];
        v = r.score1; k = lane;
        c = r.score1 > 0 ? r.ref_end1 + p.slice_base[first + lane] : 0x7fffffff;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v2 = __shfl_xor(v, d), c2 = __shfl_xor(c, d), k2 = __shfl_xor(k, d);
        const bool take = v2 > v || (v2 == v && (c2 < c || (c2 == c && k2 < k)));
        v = take ? v2 : v; c = take ? c2 : c; k = take ? k2 : k;
    }
    if (lane == 0) {
        SswResult r = p.results[p.n_real + first + k];
        if (r.score1 > 0) {
            const int base = p.slice_base[first + k];
            r.ref_end1 += base;
            if (r.ref_begin1 >= 0) r.ref_begin1 += base;
        }
        p.results[task.out_index] = r;
    }
}

hipError_t launch_ssw_combine(const SswParams& p, int ntasks, hipStream_t stream)
{
    hipLaunchKernelGGL(ssw_combine_kernel, dim3(ntasks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// p.slices / p.parts set; p.tasks = the sliced class's tasks
hipError_t launch_ssw_scan_sliced(bool geq, const SswParams& p, int ntasks, int nslices, hipStream_t stream)
{
    if (geq) {
        hipLaunchKernelGGL((ssw_scan_slice_kernel<true>), dim3(nslices), dim3(64), 0, stream, p);
        hipLaunchKernelGGL((ssw_scan_finish_kernel<true>), dim3(ntasks), dim3(64), 0, stream, p);
    } else {
        hipLaunchKernelGGL((ssw_scan_slice_kernel<false>), dim3(nslices), dim3(64), 0, stream, p);
        hipLaunchKernelGGL((ssw_scan_finish_kernel<false>), dim3(ntasks), dim3(64), 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace clh
