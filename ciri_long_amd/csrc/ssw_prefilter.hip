// ssw_prefilter.hip -- exact column prefilter in front of K1s on long windows (gfx950).
//
// What it serves: align_clip_segments aligns a 20..300-base clip against hit +- 200 kb (CIRI_long/find_bsj.py:196-216); the
// score pass of the bundled library (libs/striped_smith_waterman/ssw.c:123-345 sw_sse2_byte, orchestration ssw.c:779-849)
// computes every cell of that window.  This kernel computes, per block of 256 bytes of window text, the minimum of
//     d(j) = unit-cost edit distance of the WHOLE read to the best window substring ending at column j
// (pairs with mat[r][q] > M - c count as equal, c = min(M, gap_extend)).  Every local alignment that ends at column j scores at
// most  M * L - c * d(j)  (proof and CPU model: tools/prefilter_model.py, tests/test_prefilter_model.py), so once some score S0
// has been attained anywhere, only blocks with a minimum <= (M L - S0) / c can hold the maximum or tie it.  ssw_scan.hip
// (ssw_scan_pick_kernel) attains S0 around the smallest minimum, turns the blocks that pass into slices and K1s runs on those;
// when too many pass it writes the static slices instead -- the answer never depends on this kernel.  Windows this bound cannot thin out
// (clips whose best score is about half their length: the unit-cost distance of a clip to random text is only ~0.46 L) go through a SECOND
// stage with the indel distance, ssw_prefilter_indel_kernel below.
//
// Scheme: Myers' bit-vector recurrence, semi-global (free start in the window: the horizontal delta entering row 0 is 0).  The
// read's rows are the bits of W = ceil(L / 32) registers (L <= 254: W <= 8; a longer read goes through in PIECES of <= 254 rows, each
// a task of its own here: tools/prefilter_model.py, "the read in pieces"); ONE LANE walks a stretch of the window column by
// column -- 13 W + 8 integer instructions per column, no cross-lane traffic -- so a wave covers 64 stretches of one task's
// window at once.  A lane owns `bpl` blocks and starts 2 L columns early with the fresh state (an alignment that costs at most
// L spans at most 2 L columns, so from its first owned column on its d(j) is the whole-window d(j)).  Window text is read 16
// bytes per lane and load, in 256-byte blocks of the refs buffer (the blocks are address-aligned: bytes before the window's
// first column or behind its last only add columns that are not there -- they can lower a minimum, never raise one).
// The match vectors of the five window codes sit in LDS as a table indexed by the raw window byte & 31 (base code, lower-case
// bit, complement for minus-strand windows: clh_device.h ref_code), one ds_read per column.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

namespace {

template <int W> struct PfRow { static constexpr int P = W <= 1 ? 1 : (W <= 2 ? 2 : (W <= 4 ? 4 : 8)); };

template <int W>
__device__ __forceinline__ void pf_load_eq(uint32_t (&Eq)[W], const uint32_t* s_eq, uint32_t byte_off)
{
    constexpr int P = PfRow<W>::P;
    const uint32_t* row = (const uint32_t*)((const char*)s_eq + byte_off);
    if constexpr (P == 1) { Eq[0] = row[0]; }
    else if constexpr (P == 2) { const uint2 v = *(const uint2*)row; Eq[0] = v.x; Eq[1] = v.y; }
    else {
        const uint4 v = *(const uint4*)row;
        Eq[0] = v.x; Eq[1] = v.y; Eq[2] = v.z;
        if constexpr (W >= 4) Eq[3] = v.w;
        if constexpr (P == 8) {
            const uint4 u = *(const uint4*)(row + 4);
            Eq[4] = u.x;
            if constexpr (W >= 6) Eq[5] = u.y;
            if constexpr (W >= 7) Eq[6] = u.z;
            if constexpr (W >= 8) Eq[7] = u.w;
        }
    }
}

// one window column: Pv/Mv = vertical deltas of the column, score = d at the read's last row
template <int W>
__device__ __forceinline__ void pf_column(uint32_t (&Pv)[W], uint32_t (&Mv)[W], int& score, const uint32_t (&Eq)[W], const int lastbit)
{
    uint32_t Ph[W], Mh[W], Xv[W];
    unsigned int carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        Xv[w] = Eq[w] | Mv[w];
        unsigned int co;
        const uint32_t sum = __builtin_addc(Eq[w] & Pv[w], Pv[w], carry, &co);
        carry = co;
        const uint32_t Xh = (sum ^ Pv[w]) | Eq[w];
        const uint32_t t = Xh | Pv[w];
        Ph[w] = Mv[w] | ~t;
        Mh[w] = Pv[w] & Xh;
    }
    score += (int)((Ph[W - 1] >> lastbit) & 1u) - (int)((Mh[W - 1] >> lastbit) & 1u);
#pragma unroll
    for (int w = W - 1; w >= 0; --w) {
        const uint32_t ph = w ? __builtin_amdgcn_alignbit(Ph[w], Ph[w - 1], 31) : Ph[0] << 1;
        const uint32_t mh = w ? __builtin_amdgcn_alignbit(Mh[w], Mh[w - 1], 31) : Mh[0] << 1;
        Pv[w] = mh | ~(Xv[w] | ph);
        Mv[w] = ph & Xv[w];
    }
}

// one window column of the SECOND stage: the indel distance (a pair that is not "equal2" costs an insertion and a deletion).  With v the
// vertical deltas of the column before and h the horizontal delta a row hands down (0 into row 1), a row maps its h to: "equal2": -v;
// not, v = -1: +1; not, v = 0 (G): -1 -> 0, else +1; not, v = +1 (I): h handed on.  So h = -1 exactly on the rows reached from a
// constant -1 through rows of I (A), h = 0 on the rows reached through I from a constant 0, from the top, or from a G row whose input is
// -1 (Z), +1 elsewhere; "reached through a run" is one addition, as in Myers' algorithm.  tools/prefilter_model.py: indel_semiglobal.
template <int W>
__device__ __forceinline__ void pf_column_indel(uint32_t (&Pv)[W], uint32_t (&Mv)[W], int& score, const uint32_t (&Eq)[W], const int lastbit)
{
    uint32_t I[W], A[W], Z[W];
    unsigned int carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        I[w] = ~Eq[w] & Pv[w];
        const uint32_t Km = Eq[w] & Pv[w], Kmb = w ? Eq[w - 1] & Pv[w - 1] : 0u;
        const uint32_t u = (w ? __builtin_amdgcn_alignbit(Km, Kmb, 31) : Km << 1) & I[w];
        unsigned int co;
        const uint32_t sum = __builtin_addc(u, I[w], carry, &co);
        carry = co;
        A[w] = Km | ((sum ^ I[w]) & I[w]);
    }
    carry = 0;
    uint32_t szb = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t ash = w ? __builtin_amdgcn_alignbit(A[w], A[w - 1], 31) : A[0] << 1;
        const uint32_t nz = Pv[w] | Mv[w];
        const uint32_t Sz = (Eq[w] & ~nz) | (~(Eq[w] | nz) & ash);
        const uint32_t u = (w ? __builtin_amdgcn_alignbit(Sz, szb, 31) : (Sz << 1) | 1u) & I[w];
        szb = Sz;
        unsigned int co;
        const uint32_t sum = __builtin_addc(u, I[w], carry, &co);
        carry = co;
        Z[w] = Sz | ((sum ^ I[w]) & I[w]);
    }
    score += (int)((~(A[W - 1] | Z[W - 1]) >> lastbit) & 1u) - (int)((A[W - 1] >> lastbit) & 1u);
#pragma unroll
    for (int w = W - 1; w >= 0; --w) {
        const uint32_t Ph = ~(A[w] | Z[w]), Phb = w ? ~(A[w - 1] | Z[w - 1]) : 0u;
        const uint32_t phi = w ? __builtin_amdgcn_alignbit(Ph, Phb, 31) : Ph << 1;
        const uint32_t mhi = w ? __builtin_amdgcn_alignbit(A[w], A[w - 1], 31) : A[0] << 1;
        const uint32_t pv = Pv[w], mv = Mv[w], eq = Eq[w];
        Pv[w] = (eq & mhi) | (~eq & (pv | (~pv & ~mv & ~phi) | (mv & mhi)));
        Mv[w] = phi & (eq | mv);
    }
}

template <int W, bool INDEL = false>
__device__ void pf_walk(const SswParams& p, const PfTask& pc, const PfWin& pt, const PfWork& wk, const SswTask& task, const uint32_t* s_eq)
{
    constexpr int P = PfRow<W>::P;
    const int lane = threadIdx.x & 63;
    const int L = pc.rows;                                   // the piece's rows (the whole read when it has one piece)
    const int lastbit = (L - 1) & 31;
    const int bpl = p.pf_bpl;
    const int ovch = (2 * L + 15) >> 4;                      // chunks of 16 columns a lane starts early
    const int kb = wk.first_block + lane * bpl;              // first owned block (processing order)
    const bool lane_on = kb < pt.nsub;
    const int kend = kb + bpl < pt.nsub ? kb + bpl : pt.nsub;
    const int g_begin = kb * 16 - ovch, g_end = kend * 16, g_own = kb * 16;
    const int n_iter = ovch + bpl * 16;
    const bool rc = task.ref_rc != 0;
    const int8_t* base = p.refs;
    uint8_t* dmin = p.pf_dmin + pc.sub_off;

    auto chunk_ptr = [&](int g) -> const uint4* {
        const int64_t blk = rc ? (int64_t)pt.mem_block0 - (g >> 4) : (int64_t)pt.mem_block0 + (g >> 4);
        const int c = rc ? 15 - (g & 15) : (g & 15);
        return (const uint4*)(base + (blk << 8) + (c << 4));
    };
    uint32_t Pv[W], Mv[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { Pv[w] = 0xffffffffu; Mv[w] = 0; }
    int score = L, cur_min = 255;
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (lane_on && g_begin >= 0 && g_begin < g_end) nxt = *chunk_ptr(g_begin);
    for (int it = 0; it < n_iter; ++it) {
        const int g = g_begin + it;
        const bool valid = lane_on && g >= 0 && g < g_end;
        uint4 cur = nxt;
        if (lane_on && g + 1 >= 0 && g + 1 < g_end) nxt = *chunk_ptr(g + 1);
        if (valid) {
            if (rc) {                                         // columns run down the addresses: last byte first
                const uint32_t a = __builtin_bswap32(cur.w), b = __builtin_bswap32(cur.z), c = __builtin_bswap32(cur.y), d = __builtin_bswap32(cur.x);
                cur = make_uint4(a, b, c, d);
            }
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    // table row of this window byte: (byte & 31) * P * 4 bytes
                    constexpr int SH = P == 1 ? 2 : (P == 2 ? 3 : (P == 4 ? 4 : 5));
                    const uint32_t wd = words[q];
                    const uint32_t off = 8 * b >= SH ? (wd >> (8 * b - SH)) & (31u << SH) : (wd << (SH - 8 * b)) & (31u << SH);
                    uint32_t Eq[W];
                    pf_load_eq<W>(Eq, s_eq, off);
                    if constexpr (INDEL) pf_column_indel<W>(Pv, Mv, score, Eq, lastbit);
                    else pf_column<W>(Pv, Mv, score, Eq, lastbit);
                    cur_min = score < cur_min ? score : cur_min;
                }
            }
            if ((g & 15) == 15) {
                if (g >= g_own) dmin[g >> 4] = (uint8_t)cur_min;
                cur_min = 255;
            }
        }
    }
}

}  // namespace

// one entry of the work list: 64 lanes x bpl blocks of one piece's window
template <bool INDEL>
__device__ __forceinline__ void pf_work_item(const SswParams& p, const PfWork wk, uint32_t* s_eq, uint32_t* s_peq, int* s_mat)
{
    const int lane = threadIdx.x & 63;
    const PfTask pc = p.pf_tasks[wk.piece];
    const SswTask task = p.tasks[pc.task];
    const PfWin pt = p.pf_win[pc.task];
    const int L = pc.rows;
    const int W = (L + 31) >> 5;
    if (lane < 48) { const int b_ = lane >> 3, q_ = lane & 7; s_mat[lane] = (b_ < p.n && q_ < p.n) ? (int)p.mat[b_ * p.n + q_] : 0; }
    __syncthreads();
    // match vectors of the five window codes: bit i = read base i and the code are "equal" (mat[code][q] > M - c; second stage: M - 2c)
    const int cc = p.max_match < p.gapE ? p.max_match : p.gapE;
    const int eq_above = p.max_match - (INDEL ? 2 * cc : cc);
    const int8_t* read = p.reads + task.read_off + pc.row0;
    for (int half = 0; half < (W + 1) / 2; ++half) {
        const int row = 64 * half + lane;
        int q = 5;
        if (row < L) { const int c = (int)read[row] & 7; q = c > 5 ? 5 : c; }
        for (int code = 0; code < 5; ++code) {
            const unsigned long long m = __ballot(row < L && s_mat[code * 8 + q] > eq_above);
            if (lane == 0) { s_peq[code * 8 + 2 * half] = (uint32_t)m; s_peq[code * 8 + 2 * half + 1] = (uint32_t)(m >> 32); }
        }
    }
    __syncthreads();
    const int P = W <= 1 ? 1 : (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    if (lane < 32) {
        const int code = ref_code(lane, task.ref_rc);
        for (int w = 0; w < P; ++w) s_eq[lane * P + w] = w < W ? s_peq[code * 8 + w] : 0u;
    }
    __syncthreads();
    switch (W) {
        case 1: pf_walk<1, INDEL>(p, pc, pt, wk, task, s_eq); break;
        case 2: pf_walk<2, INDEL>(p, pc, pt, wk, task, s_eq); break;
        case 3: pf_walk<3, INDEL>(p, pc, pt, wk, task, s_eq); break;
        case 4: pf_walk<4, INDEL>(p, pc, pt, wk, task, s_eq); break;
        case 5: pf_walk<5, INDEL>(p, pc, pt, wk, task, s_eq); break;
        case 6: pf_walk<6, INDEL>(p, pc, pt, wk, task, s_eq); break;
        case 7: pf_walk<7, INDEL>(p, pc, pt, wk, task, s_eq); break;
        default: pf_walk<8, INDEL>(p, pc, pt, wk, task, s_eq); break;
    }
}

__global__ void __launch_bounds__(64, 6) ssw_prefilter_kernel(const SswParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_eq[32 * 8];
    __shared__ uint32_t s_peq[5 * 8];
    __shared__ int s_mat[48];
    pf_work_item<false>(p, p.pf_work[blockIdx.x], s_eq, s_peq, s_mat);
}

// The second stage (tools/prefilter_model.py, "second stage"): the windows the unit-cost bound fails to thin out (ssw_scan_pick_kernel
// queues their entries of the work list) once more with the INDEL distance -- H(j) <= M L - c d2(j), d2 >= d and larger by a quarter on
// random text; about twice the instructions per column.  It overwrites the windows' block minima; ssw_scan_pick2_kernel reads them.
// Persistent workgroups: the number of entries is only known on the device.
__global__ void __launch_bounds__(64, 4) ssw_prefilter_indel_kernel(const SswParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_eq[32 * 8];
    __shared__ uint32_t s_peq[5 * 8];
    __shared__ int s_mat[48];
    const int total = p.pf_ctl->q2count;
    for (;;) {
        int idx = 0;
        if ((threadIdx.x & 63) == 0) idx = atomicAdd(&p.pf_ctl->q2next, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= total) break;
        pf_work_item<true>(p, p.pf_work[p.pf_q2[idx]], s_eq, s_peq, s_mat);
        __syncthreads();
    }
}

hipError_t launch_ssw_prefilter(const SswParams& p, int nwork, hipStream_t stream)
{
    if (nwork <= 0) return hipSuccess;
    hipLaunchKernelGGL(ssw_prefilter_kernel, dim3(nwork), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_ssw_prefilter_indel(const SswParams& p, int nworkgroups, hipStream_t stream)
{
    if (nworkgroups <= 0) return hipSuccess;
    hipLaunchKernelGGL(ssw_prefilter_indel_kernel, dim3(nworkgroups), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace clh
