// ssw_lanes.hip -- K1l: Smith-Waterman for SHORT REFERENCES (<= 64 columns), one alignment per LANE.
//
// What it serves: the collapse stage's junction alignments (CIRI_long/collapse.py:161-173 curate_junction -- thousands of 20-nt genomic junctions per
// cluster against one ~50-nt consensus junction; :251-256 head_positions, :373-387 the 50-nt junction against every read of a cluster), scoring
// 10/4/8/2.  K1s / K1w spread the lanes of a wave over the reference columns of ONE alignment (256 columns at least): with a 20- or 50-column
// reference eight or nine lanes in ten compute padding (collapse_c5 ran at 0.026 of the packed-op bound through round 5).  Here a lane owns a whole
// alignment: its reference columns sit in registers (H, the vertical gap F and the column's running maximum per column), the loop runs over the
// read's rows, the row's scores come from a per-lane profile in LDS -- no cross-lane traffic, no padding beyond the class's column count.
//
// Same answers as the other classes (reference: libs/striped_smith_waterman/ssw.c:123-345 sw_sse2_byte, :371-546 sw_sse2_word, orchestration
// :779-849; row-major statement oracle/rowmajor_spec.c), for the alignments the plan sends here (clh_api.hip, lanes_class_ok):
//   * no second-best score wanted (its column maxima depend on the reference's wildcard padding rows, which are not computed here);
//   * the recurrence is exact in whichever regime the reference would choose: gap_open > gap_extend, or -- with gap_open == gap_extend, where the
//     reference's 16-bit pass truncates F at stripe boundaries -- a score that provably stays in the 8-bit regime;
//   * code 4 scores 0 against everything (or the matrix has no fifth letter).
// Forward pass: the global maximum, the FIRST column that holds it, the smallest row holding it there (ssw.c:283,299-308,490,502-511).  Reverse pass
// (ssw.c:834-849): the same rule on the reversed prefixes -- inside that rectangle no cell exceeds the forward score, so "the first column whose
// maximum equals it" (the reference's terminate test) is the first column holding the rectangle's maximum.
//
// Cost: ~13 vector instructions per cell, most of them at full rate (16-bit VOP2 forms and 32-bit add/sub: 2 cycles, profiles/r06_valu_rate.txt);
// a wave computes 64 cells with them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

namespace {

// the profile of one wave: scores (int8) of row letter q (0..3) against four neighbouring columns per 32-bit word, [q][column word][lane]
template <int RMAX>
__device__ __forceinline__ void lanes_profile(uint32_t* prof, const int* s_mat, const int n, const int8_t* ref, const int cstep, const int comp, const int R, const int lane)
{
    constexpr int NW = RMAX / 4;
    for (int w = 0; w < NW; ++w) {
        uint32_t pw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = 4 * w + k;
            int c = 5;
            if (j < R) c = ref_code((int)ref[(int64_t)j * cstep], comp);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = (c < n && q < n) ? s_mat[c * 8 + q] : 0;
                pw[q] |= (uint32_t)(s & 0xff) << (8 * k);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) prof[(q * NW + w) * 64 + lane] = pw[q];
    }
}

// One pass of every lane's alignment: rows rd[0], rd[rstep], .. (L of them) against the R columns whose profile is in `prof`.
// -> the largest H, the first column holding it, the smallest row holding it there (0, -1, 0 when nothing scores).
template <int RMAX>
__device__ __forceinline__ void lanes_pass(const uint32_t* prof, const int8_t* rd, const int rstep, const int L, const int R, const int gapO, const int gapE,
                                           const int lane, int& out_max, int& out_col, int& out_row)
{
    constexpr int NW = RMAX / 4;
    unsigned short H[RMAX], F[RMAX];
    uint32_t CM[RMAX];                       // per column: (largest H so far) << 16 | 0xffff - (first row that held it)
#pragma unroll
    for (int j = 0; j < RMAX; ++j) { H[j] = 0; F[j] = 0; CM[j] = 0; }
    int Lw = L;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(Lw, d); Lw = o > Lw ? o : Lw; }
    Lw = __builtin_amdgcn_readfirstlane(Lw);
    const unsigned short gO = (unsigned short)gapO, gE = (unsigned short)gapE;
    int qn = L > 0 ? ((int)rd[0] & 7) : 5;
    for (int i = 0; i < Lw; ++i) {
        const int q = qn;
        if (i + 1 < L) qn = (int)rd[(int64_t)(i + 1) * rstep] & 7;       // the next row's letter is on its way while this row is computed
        if (i < L) {
            const uint32_t msk = q < 4 ? 0xffffffffu : 0u;                 // code 4 and above: 0 against everything
            const uint32_t* row = prof + ((q & 3) * NW) * 64 + lane;
            const uint32_t nrow = 0xffffu - (uint32_t)i;
            unsigned short e = 0, diag = 0;
            uint32_t pw = 0;
#pragma unroll
            for (int j = 0; j < RMAX; ++j) {
                if ((j & 3) == 0) pw = row[(j >> 2) * 64] & msk;
                const short s = (short)(int8_t)(pw >> (8 * (j & 3)));
                const short t = (short)(diag + s);
                const unsigned short f = F[j];
                short hm = __builtin_elementwise_max(t, (short)e);
                hm = __builtin_elementwise_max(hm, (short)f);             // e, f >= 0: so is the maximum
                const unsigned short h = (unsigned short)hm;
                const unsigned short hg = __builtin_elementwise_sub_sat(h, gO);
                e = __builtin_elementwise_max(__builtin_elementwise_sub_sat(e, gE), hg);
                F[j] = __builtin_elementwise_max(__builtin_elementwise_sub_sat(f, gE), hg);
                diag = H[j];
                H[j] = h;
                const uint32_t key = ((uint32_t)h << 16) | nrow;
                CM[j] = CM[j] > key ? CM[j] : key;
            }
        }
    }
    int best = 0, bcol = -1, brow = 0;
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        const int m = (int)(CM[j] >> 16);
        if (j < R && m > best) { best = m; bcol = j; brow = 0xffff - (int)(CM[j] & 0xffffu); }
    }
    out_max = best; out_col = bcol; out_row = brow;
}

template <int RMAX>
__global__ void __launch_bounds__(64) ssw_lanes_kernel(const SswParams p, const int ntasks)
{
    __shared__ uint32_t s_prof[RMAX * 64];
    __shared__ int s_mat[48];
    const int lane = threadIdx.x & 63;
    if (lane < 48) { const int b = lane >> 3, q = lane & 7; s_mat[lane] = (b < p.n && q < p.n) ? (int)p.mat[b * p.n + q] : 0; }
    __syncthreads();
    const int idx = blockIdx.x * 64 + lane;
    const bool valid = idx < ntasks;
    SswTask task;
    if (valid) task = p.tasks[idx];
    else { task.read_off = 0; task.ref_off = 0; task.read_len = 0; task.ref_len = 0; task.ref_rc = 0; task.mask_len = 0; task.out_index = 0; }
    const int L = task.read_len, R = task.ref_len;
    const int rdir = task.ref_rc ? -1 : 1;
    const int8_t* read = p.reads + task.read_off;
    const int8_t* ref = p.refs + task.ref_off;
    SswResult res;
    res.score1 = 0; res.score2 = 0; res.ref_begin1 = -1; res.ref_end1 = -1; res.read_begin1 = -1; res.read_end1 = 0;
    res.ref_end2 = task.mask_len >= 15 ? 0 : -1; res.status = 0;

    // ---- forward (ssw.c:804-822): the byte regime unless it overflows, then the word regime -- the recurrence is the same exact one here
    int fmax, fcol, frow;
    lanes_profile<RMAX>(s_prof, s_mat, p.n, ref, rdir, task.ref_rc, R, lane);
    lanes_pass<RMAX>(s_prof, read, 1, L, R, p.gapO, p.gapE, lane, fmax, fcol, frow);
    int regime = p.score_size == 1 ? 1 : 0;
    bool null_result = false;
    if (p.score_size != 1 && fmax + p.bias >= 255) {
        if (p.score_size == 0) null_result = true;       // the reference returns NULL (ssw.c:810-813)
        else regime = 1;
    }
    if (null_result) res.status = CLH_STATUS_OVERFLOW8;
    else {
        res.status = regime ? CLH_STATUS_WORD : 0;
        res.score1 = fmax;
        if (fmax == 0) { res.ref_end1 = regime ? 0 : -1; res.read_end1 = 0; }
        else { res.ref_end1 = fcol; res.read_end1 = frow; }
    }
    // ---- reverse: begin coordinates (ssw.c:834-849) on the reversed prefixes
    const bool want_begin = valid && !null_result && !(p.flag == 0 || (p.flag == 2 && res.score1 < p.filters));
    {
        const int rL = want_begin ? res.read_end1 + 1 : 0, rR = want_begin ? res.ref_end1 + 1 : 0;
        int rmax, rcol, rrow;
        lanes_profile<RMAX>(s_prof, s_mat, p.n, ref + (int64_t)(rR > 0 ? res.ref_end1 : 0) * rdir, -rdir, task.ref_rc, rR, lane);
        lanes_pass<RMAX>(s_prof, read + (rL > 0 ? res.read_end1 : 0), -1, rL, rR, p.gapO, p.gapE, lane, rmax, rcol, rrow);
        if (want_begin) {
            if (rmax == 0) { res.ref_begin1 = regime ? 0 : -1; res.read_begin1 = res.read_end1; }
            else { res.ref_begin1 = res.ref_end1 - rcol; res.read_begin1 = res.read_end1 - rrow; }
        }
    }
    if (valid) p.results[task.out_index] = res;
}

}  // namespace

// rmax: the class's column count (20, 32, 52 or 64)
hipError_t launch_ssw_lanes(int rmax, const SswParams& p, int ntasks, hipStream_t stream)
{
    const dim3 grid((unsigned)((ntasks + 63) / 64)), block(64);
    if (rmax <= 20) hipLaunchKernelGGL((ssw_lanes_kernel<20>), grid, block, 0, stream, p, ntasks);
    else if (rmax <= 32) hipLaunchKernelGGL((ssw_lanes_kernel<32>), grid, block, 0, stream, p, ntasks);
    else if (rmax <= 52) hipLaunchKernelGGL((ssw_lanes_kernel<52>), grid, block, 0, stream, p, ntasks);
    else hipLaunchKernelGGL((ssw_lanes_kernel<64>), grid, block, 0, stream, p, ntasks);
    return hipGetLastError();
}

}  // namespace clh
