// ssw_scan_wide.hip -- K1w: the row-scan form of K1s (ssw_scan.hip) for the alignments that kernel does not take: reads of
// 255..4096 bases and every alignment whose score can pass 254 (10/4/8/2 scoring of collapse), windows below 32768 columns.
//
// Same answers as ssw_wavefront.hip (reference: libs/striped_smith_waterman/ssw.c:123-345 sw_sse2_byte, :371-546 sw_sse2_word,
// and the forward + reverse orchestration of ssw_align, ssw.c:779-849; row-major statement: oracle/rowmajor_spec.c).
// The lanes own reference columns, the loop runs over the read's rows (see ssw_scan.hip for the layout of a chunk, the prefix
// maximum that carries the gap along a row, and the hand-over between chunks).  What is different here:
//   * two regimes, chosen per alignment as ssw_align does (ssw.c:804-822): the 16-bit pass when the 8-bit pass would overflow
//     (the byte-regime pass runs first, as in the reference, and hands over when it overflows);
//   * the byte regime in 16-bit arithmetic: the exact recurrence, rows padded to 16, abandoned as soon as a column maximum +
//     bias reaches 255 (ssw.c:285; checked once per 64 rows -- what the abandoned pass computed is never used);
//   * the word regime: rows padded to 8; with gap_open == gap_extend the fix-up loop of ssw.c:462-481 touches only the first
//     position of every stripe (rowmajor_spec.c: "16 bit, gapO <= gapE"): in the rows r = S, 2S, .., 7S (S = ceil(readLen/8)) the
//     gap from the row above does not enter the main value Hm, only the final value Hf = max(Hm, that gap); the column maximum
//     and the gap along the row follow Hm, the next row's diagonal and the end-row search follow Hf.  Those seven rows leave
//     both values in HBM (per chunk, read back by the lanes that wrote them); every other row keeps, per column, the running
//     maximum and the first row that holds it in two packed registers;
//   * the hand-over between chunks (one H and one E per row) goes through HBM, not LDS: up to 4112 rows.
// 11 packed operations per cell pair (gap_open == gap_extend) or 15, plus ~45 per row step of 1024 columns.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <algorithm>
#include "clh_device.h"
#include "clh_device_ops.h"

namespace clh {

namespace {



#ifndef SCANW_WAVES
#define SCANW_WAVES 3      // measured on C2: 2 -> 12.6 ms, 3 -> 12.2 ms, 4 -> 15.1 ms (at 3 the spills stay outside the row loop)
#endif
static constexpr int W_CPR_MAX = 8;
static constexpr int W_PROF_WORDS = 6 * W_CPR_MAX * 64;      // uint32 per wave: [query code 0..5][register][lane]
static constexpr int W_NB = 7;                               // stripe boundaries of a word pass
static constexpr int W_INF = 0x7fffffff;

struct WIn {
    const int8_t* read;   // first row's base
    int rstep;            // +1 / -1
    int L;                // rows of the read
    int rows;             // rows processed (L padded with wildcard rows)
    int S;                // word regime with gap_open == gap_extend: the stripe length (rows S, 2S, .. 7S are boundary rows); else 0
    const int8_t* ref;    // first column's base
    int cstep;            // +1 / -1
    int comp;
    int ncols;
    int rcomp;            // transposed form only: the ROWS are reference bases, complemented when set (ref_code)
    int terminate;        // column maximum that ends the pass (ssw.c:296, 489); 1 << 30 = never
    int overflow_at;      // byte regime: a column maximum >= this abandons the pass (255 - bias); 1 << 30 = never
    uint16_t* colmax;     // per-column maxima (indexed by column), or nullptr
};
struct WOut { int max, col, row, overflow; };

struct WMem {
    uint32_t* prof;       // LDS: W_PROF_WORDS
    const int* mat;       // LDS: [6 reference codes][8 query codes]
    uint32_t* bnd;        // HBM: boundary rows of the chunk, [W_NB][2 (Hm, Hf)][CPR][64 lanes] packed registers
    short* cH;            // HBM: [2][rows_cap]: final H of the chunk's last column per row (parity of the chunk)
    short* cE;            // HBM: [2][rows_cap]: E entering the next chunk's first column per row
    int rows_cap;
};

// one chunk of 128*CPR columns starting at column c0.
// TR (the transposed form, for references of at most 64 columns against long reads: ssw_scanw_tr_kernel below): the COLUMNS are the read's
// bases and the rows the reference's, so that the lanes have something to own; the matrix is the same, and so is every H.  What changes is
// which of the cells that hold the maximum is reported: the reference's rule -- first REFERENCE position, then the smallest read position there
// (ssw.c:283,299-308) -- reads here "smallest row, then smallest column".
template <int CPR, bool GEQ, bool WORD, bool TR = false>
__device__ bool scanw_chunk(const WIn& in, const WMem& mem, const int c0, const int parity, const bool first, const bool more,
                            const int gapO, const int gapE, int& best_score, int& best_col, int& best_row, int& tcol)
{
    constexpr int VEC = CPR < 4 ? CPR : 4;       // registers per LDS read
    constexpr int NCH = CPR / VEC;
    constexpr bool QUIRK = WORD && GEQ;
    const int lane = threadIdx.x & 63;
    const int K = CPR * gapE;                    // what a gap loses across one virtual lane
    const int kLo = 2 * lane * K;
    const uint32_t gO2 = dup16(gapO), gE2 = dup16(gapE);
    uint32_t tgE[CPR];                           // what the gap entering a virtual lane has lost at its t-th column
#pragma unroll
    for (int t = 0; t < CPR; ++t) tgE[t] = dup16(t * gapE);

    // ---- profile of the chunk's columns: prof[q][c][lane][VEC], entry = scores of the row base q against the two columns
    {
        int rlo[CPR], rhi[CPR];
#pragma unroll
        for (int t = 0; t < CPR; ++t) {
            const int jlo = c0 + 2 * CPR * lane + t, jhi = jlo + CPR;
            const int blo = jlo < in.ncols ? (int)in.ref[(int64_t)jlo * in.cstep] : 0, bhi = jhi < in.ncols ? (int)in.ref[(int64_t)jhi * in.cstep] : 0;
            if constexpr (TR) {      // a column is a read base: its code as the row loop of the plain form takes it
                const int clo = blo & 7, chi = bhi & 7;
                rlo[t] = jlo < in.ncols ? (clo > 5 ? 5 : clo) : 5;
                rhi[t] = jhi < in.ncols ? (chi > 5 ? 5 : chi) : 5;
            } else {
                rlo[t] = jlo < in.ncols ? ref_code(blo, in.comp) : 5;
                rhi[t] = jhi < in.ncols ? ref_code(bhi, in.comp) : 5;
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int t = 0; t < CPR; ++t) {
                // mat is [reference code][read code]: transposed, the row letter q is the reference's
                const int slo = TR ? mem.mat[q * 8 + rlo[t]] : mem.mat[rlo[t] * 8 + q], shi = TR ? mem.mat[q * 8 + rhi[t]] : mem.mat[rhi[t] * 8 + q];
                mem.prof[((q * NCH + t / VEC) * 64 + lane) * VEC + (t % VEC)] = (uint32_t)(slo & 0xffff) | ((uint32_t)shi << 16);
            }
    }
    const short* cHin = mem.cH + (parity ^ 1) * mem.rows_cap;
    const short* cEin = mem.cE + (parity ^ 1) * mem.rows_cap;
    short* cHout = mem.cH + parity * mem.rows_cap;
    short* cEout = mem.cE + parity * mem.rows_cap;
    __syncthreads();

    uint32_t Hp[CPR], Hd[QUIRK ? CPR : 1], Fst[GEQ ? 1 : CPR], cmv[CPR], cmr[CPR];
#pragma unroll
    for (int t = 0; t < CPR; ++t) { Hp[t] = 0; cmv[t] = 0; cmr[t] = 0; if constexpr (!GEQ) Fst[t] = 0; if constexpr (QUIRK) Hd[t] = 0; }
    int prev_hb = 0;                             // final H[row - 1][c0 - 1]
    int next_b = QUIRK && in.S > 0 ? in.S : W_INF, nb_seen = 0;
    bool diag_d = false;                         // the row before was a boundary row: its final values are in Hd
    bool overflow = false;
    for (int rb = 0; rb < in.rows && !overflow; rb += 64) {
        const int row = rb + lane;
        int qv = 5;
        if (row < in.L) {
            if constexpr (TR) qv = ref_code((int)in.read[(int64_t)row * in.rstep], in.rcomp);
            else { const int c = (int)in.read[(int64_t)row * in.rstep] & 7; qv = c > 5 ? 5 : c; }
        }
        int hbv = 0, ebv = 0;
        if (!first && row < in.rows) { hbv = cHin[row]; ebv = cEin[row]; }
        const int cnt = in.rows - rb < 64 ? in.rows - rb : 64;
        int cobH = 0, cobE = 0;
        // one row.  SLOW (the word regime with gap_open == gap_extend only): the rows around a stripe boundary -- `bnd`: a boundary
        // row, `dd`: the row before was one and the diagonal comes from its final values Hd.  At most 14 rows of a pass take it.
        auto step = [&](const int i, auto slow_c, const bool bnd, const bool dd) {
            constexpr bool SLOW = decltype(slow_c)::value;
            const int q = __builtin_amdgcn_readlane(qv, i), hb = __builtin_amdgcn_readlane(hbv, i), eb = __builtin_amdgcn_readlane(ebv, i);
            uint32_t P[CPR];
            {
                const uint32_t* pp = mem.prof + (q * NCH * 64 + lane) * VEC;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if constexpr (VEC == 4) { const uint4 v = *(const uint4*)(pp + c * 64 * VEC); P[4 * c] = v.x; P[4 * c + 1] = v.y; P[4 * c + 2] = v.z; P[4 * c + 3] = v.w; }
                    else { const uint2 v = *(const uint2*)(pp + c * 64 * VEC); P[0] = v.x; P[1] = v.y; }
                }
            }
            uint32_t dsrc[CPR];
#pragma unroll
            for (int t = 0; t < CPR; ++t) { dsrc[t] = Hp[t]; if constexpr (SLOW) dsrc[t] = dd ? Hd[t] : Hp[t]; }
            const uint32_t d0 = hand_down(dsrc[CPR - 1], prev_hb);
            uint32_t R[CPR];
            uint32_t e = 0, U;
#pragma unroll
            for (int t = 0; t < CPR; ++t) {
                const uint32_t tt = pk_adds(t == 0 ? d0 : dsrc[t - 1], P[t]);
                uint32_t X;
                if constexpr (GEQ) {
                    uint32_t Fv = pk_subus(Hp[t], gO2);
                    if constexpr (SLOW) Fv = bnd ? 0u : Fv;                           // a boundary row: the gap from the row above stays out of the main value
                    X = pk_max(tt, Fv);
                } else { const uint32_t Fv = pk_max(pk_subus(Fst[t], gE2), pk_subus(Hp[t], gO2)); Fst[t] = Fv; X = pk_max(tt, Fv); }
                if (t == 0) R[0] = X;
                else if constexpr (GEQ) R[t] = pk_max(X, pk_subus(R[t - 1], gO2));
                else { e = pk_max(pk_subus(e, gE2), pk_subus(R[t - 1], gO2)); R[t] = pk_max(X, e); }
            }
            if (GEQ) U = pk_subus(R[CPR - 1], gO2);
            else U = pk_max(pk_subus(e, gE2), pk_subus(R[CPR - 1], gO2));
            const int Blo = (int)(U & 0xffffu) + kLo, Bhi = (int)(U >> 16) + kLo + K;
            const int inc = wave_prefix_max(Blo > Bhi ? Blo : Bhi);
            const int fill = eb - K;
            int exc = dpp_shr1(fill, inc);
            exc = exc > fill ? exc : fill;
            const int einLo = exc - kLo + K;
            const int m2 = exc > Blo ? exc : Blo;
            const int einHi = m2 - kLo;
            const uint32_t Ein = ((uint32_t)einLo & 0xffffu) | ((uint32_t)einHi << 16);
            const uint32_t rowc = dup16(rb + i);
            uint32_t last = 0;
#pragma unroll
            for (int t = 0; t < CPR; ++t) {
                const uint32_t h = pk_max(R[t], pk_subs(Ein, tgE[t]));
                bool keyed = true;
                if constexpr (SLOW) {
                    if (bnd) {
                        Hd[t] = pk_max(h, pk_subus(Hp[t], gO2));
                        uint32_t* o = mem.bnd + ((nb_seen * 2) * CPR + t) * 64 + lane;
                        o[0] = h; o[CPR * 64] = Hd[t];
                        keyed = false;
                    }
                }
                if (keyed) {
#ifdef CLH_COMPILER_SELECTS
                    const uint32_t m = pk_sra15(pk_subs(cmv[t], h));
#else
                    const uint32_t m = pk_lt_mask_small(cmv[t], h);                  // halves where h is a new maximum (strictly); scores stay below 32 000 (scanw_class_ok)
#endif
                    cmv[t] = pk_max(cmv[t], h);
                    cmr[t] = bfi(m, rowc, cmr[t]);
                }
                Hp[t] = h;
                if (t == CPR - 1) { last = h; if constexpr (SLOW) last = bnd ? Hd[t] : h; }
            }
            prev_hb = hb;
            if (more) {   // the chunk's last column: final H, and the E that enters the next chunk's first column
                const int outH = (int)((uint32_t)__builtin_amdgcn_readlane((int)last, 63) >> 16);
                int outE = __builtin_amdgcn_readlane(inc, 63);
                outE = (outE > fill ? outE : fill) - 127 * K;
                outE = outE < 0 ? 0 : outE;
                const uint32_t li = lane_is(lane, i); cobH = set_lane(cobH, outH, li); cobE = set_lane(cobE, outE, li);
            }
        };
        int i = 0;
        while (i < cnt) {
            int stop = cnt;
            if constexpr (QUIRK) { const int tb = next_b - rb; stop = diag_d ? i : (tb < cnt ? tb : cnt); }
            for (; i < stop; ++i) step(i, std::false_type{}, false, false);
            if constexpr (QUIRK) {
                if (i < cnt) {
                    const bool isb = rb + i == next_b;
                    step(i, std::true_type{}, isb, diag_d);
                    if (isb) { ++nb_seen; next_b = nb_seen < W_NB ? next_b + in.S : W_INF; }
                    diag_d = isb;
                    ++i;
                }
            }
        }
        if (more && lane < cnt) { cHout[rb + lane] = (short)cobH; cEout[rb + lane] = (short)cobE; }
        if constexpr (!WORD) {   // byte regime: a column maximum at 255 - bias abandons the pass
            uint32_t mx = cmv[0];
#pragma unroll
            for (int t = 1; t < CPR; ++t) mx = pk_max(mx, cmv[t]);
            const int hi = (int)(mx >> 16), lo = (int)(mx & 0xffffu);
            if (__builtin_amdgcn_ballot_w64((hi > lo ? hi : lo) >= in.overflow_at)) overflow = true;
        }
    }
    // the boundary rows and the hand-over arrays were written through the CU's L1 without updating lines it may hold: complete
    // the stores and drop the L1 before anything is read back
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    if (overflow) return true;

    // ---- the chunk's columns: maxima, terminate column, best cell (first column wins), its end row ----
    int tmin = W_INF;
    int colM[2 * CPR], colR[2 * CPR];
#pragma unroll
    for (int t = 0; t < CPR; ++t)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            int M = hf ? (int)(cmv[t] >> 16) : (int)(cmv[t] & 0xffffu);
            const int rA = hf ? (int)(cmr[t] >> 16) : (int)(cmr[t] & 0xffffu);
            int hmB[W_NB], hfB[W_NB];
            if constexpr (QUIRK) {
                for (int k = 0; k < W_NB; ++k) {
                    hmB[k] = 0; hfB[k] = -1;
                    if (k < nb_seen) {
                        const uint32_t a = mem.bnd[((k * 2) * CPR + t) * 64 + lane], b = mem.bnd[((k * 2 + 1) * CPR + t) * 64 + lane];
                        hmB[k] = hf ? (int)(a >> 16) : (int)(a & 0xffffu); hfB[k] = hf ? (int)(b >> 16) : (int)(b & 0xffffu);
                    }
                }
            }
            const int mA = M;
            if constexpr (QUIRK) for (int k = 0; k < W_NB; ++k) M = hmB[k] > M ? hmB[k] : M;
            // smallest row whose FINAL value equals the column maximum (ssw.c:502-511): a non-boundary row holding the maximum, or a
            // boundary row whose final value is the maximum
            int rw = mA == M ? rA : W_INF;
            if constexpr (QUIRK) for (int k = W_NB - 1; k >= 0; --k) if (hfB[k] == M) { const int rk = (k + 1) * in.S; rw = rk < rw ? rk : rw; }
            colM[2 * t + hf] = M; colR[2 * t + hf] = rw;
            const int j = c0 + 2 * CPR * lane + hf * CPR + t;
            if (j < in.ncols) {
                if (in.colmax) in.colmax[j] = (uint16_t)M;
                if (M == in.terminate) tmin = j < tmin ? j : tmin;
            }
        }
    if constexpr (TR) {
        // largest maximum, then the smallest row that holds it (<= 63: 7 bits), then the smallest column (11 bits inside the chunk); maxima stay below 2^13
        int key = -1;
#pragma unroll
        for (int t = 0; t < CPR; ++t)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int jc = 2 * CPR * lane + hf * CPR + t, j = c0 + jc;
                const int v = (colM[2 * t + hf] << 18) | ((0x7f - (colR[2 * t + hf] & 0x7f)) << 11) | (0x7ff - jc);
                if (j < in.ncols && colM[2 * t + hf] > 0 && v > key) key = v;
            }
        key = wave_max(key);
        if (key >= 0) {
            const int sc = key >> 18, rw = 0x7f - ((key >> 11) & 0x7f), cl = c0 + (0x7ff - (key & 0x7ff));
            if (sc > best_score || (sc == best_score && rw < best_row)) { best_score = sc; best_col = cl; best_row = rw; }      // (an equal row: the earlier chunk's column is the smaller)
        }
        tcol = W_INF;
        return false;
    }
    tmin = wave_min(tmin);
    int b32 = -1, brow = 0;
#pragma unroll
    for (int t = 0; t < CPR; ++t)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int jc = 2 * CPR * lane + hf * CPR + t, j = c0 + jc;
            const int v = (colM[2 * t + hf] << 11) | (0x7ff - jc);
            if (j < in.ncols && j <= tmin && v > b32) { b32 = v; brow = colR[2 * t + hf]; }
        }
    const int bw = wave_max(b32);
    // the lane that owns the best column says its end row
    const unsigned long long own = __builtin_amdgcn_ballot_w64(b32 == bw);
    const int wrow = __builtin_amdgcn_readlane(brow, own ? __builtin_ctzll(own) : 0);
    const int sc = bw < 0 ? 0 : bw >> 11;
    if (sc > best_score) { best_score = sc; best_col = c0 + (0x7ff - (bw & 0x7ff)); best_row = wrow; }
    tcol = tmin;
    return false;
}

template <bool GEQ, bool WORD, bool TR = false>
__device__ WOut scanw_pass(const WIn& in, const WMem& mem, const int gapO, const int gapE)
{
    int best_score = 0, best_col = -1, best_row = 0;
    const bool ends = in.terminate < (1 << 30);  // reverse pass: stops at the first column whose maximum is the forward score
    int parity = 0;
    WOut o; o.overflow = 0;
    for (int c0 = 0; c0 < in.ncols; parity ^= 1) {
        const int rem = in.ncols - c0;
        int tcol = W_INF;
        const bool first = c0 == 0;
        bool ov;
        // a pass that ends at a column (the reverse pass) is expected to end about one read length in: its first chunks are sized for
        // that, not for the window (columns behind the end column are wasted work, narrow chunks pay the row step's fixed part more often)
        const int want = ends ? (c0 == 0 ? in.L + in.L / 8 + 16 : 256) : rem;
        const int width = want < rem ? want : rem;
        if (width <= 256) { ov = scanw_chunk<2, GEQ, WORD, TR>(in, mem, c0, parity, first, rem > 256, gapO, gapE, best_score, best_col, best_row, tcol); c0 += 256; }
        else if (width <= 512) { ov = scanw_chunk<4, GEQ, WORD, TR>(in, mem, c0, parity, first, rem > 512, gapO, gapE, best_score, best_col, best_row, tcol); c0 += 512; }
        else { ov = scanw_chunk<8, GEQ, WORD, TR>(in, mem, c0, parity, first, rem > 1024, gapO, gapE, best_score, best_col, best_row, tcol); c0 += 1024; }
        if (ov) { o.overflow = 1; o.max = 255; o.col = -1; o.row = 0; return o; }
        if (tcol != W_INF) break;
    }
    o.max = best_score;
    if (best_score == 0) { o.col = -1; o.row = 0; return o; }
    o.col = best_col;
    o.row = best_row < in.L - 1 ? best_row : in.L - 1;
    return o;
}

// The transposed form of one alignment (class kRvScanTr: a reference of at most 64 columns, no second best, a recurrence that is exact in
// either regime -- clh_api.hip, lanes_class_for): rows = reference bases, columns = read bases.  One pass each way, in 16-bit arithmetic; the
// regime the reference would have ended in (ssw.c:804-822) follows from the score, and only labels the result.
template <bool GEQ>
__device__ bool scanw_align_tr(const SswParams& p, const int8_t* read, const int8_t* ref, const int L, const int refLen, const int ref_rc, const int mask_len,
                               const WMem& mem, SswResult& res)
{
    const int bias = p.bias, gO = p.gapO, gE = p.gapE;
    res.score1 = 0; res.score2 = 0; res.ref_begin1 = -1; res.ref_end1 = -1; res.read_begin1 = -1; res.read_end1 = 0;
    res.ref_end2 = mask_len >= 15 ? 0 : -1; res.status = 0;
    const int rdir = ref_rc ? -1 : 1;
    WIn in;
    in.read = ref; in.rstep = rdir; in.rcomp = ref_rc; in.L = refLen; in.rows = refLen; in.S = 0;
    in.ref = read; in.cstep = 1; in.comp = 0; in.ncols = L;
    in.terminate = 1 << 30; in.overflow_at = 1 << 30; in.colmax = nullptr;
    // gap_open == gap_extend: the byte-regime arithmetic (the word regime's boundary rows cannot occur: the class takes such an alignment only
    // when its score stays below the 8-bit limit); else the word-regime arithmetic, which has no boundary rows with gap_open > gap_extend
    const WOut fw = GEQ ? scanw_pass<GEQ, false, true>(in, mem, gO, gE) : scanw_pass<GEQ, true, true>(in, mem, gO, gE);
    int regime = p.score_size == 1 ? 1 : 0;
    if (p.score_size != 1 && fw.max + bias >= 255) {
        if (p.score_size == 0) { res.status = CLH_STATUS_OVERFLOW8; return false; }
        regime = 1;
    }
    res.status = regime ? CLH_STATUS_WORD : 0;
    res.score1 = fw.max;
    if (fw.max == 0) { res.ref_end1 = regime ? 0 : -1; res.read_end1 = 0; }
    else { res.ref_end1 = fw.row; res.read_end1 = fw.col; }
    const bool want_begin = !(p.flag == 0 || (p.flag == 2 && res.score1 < p.filters));
    if (want_begin) {
        WIn rv = in;
        rv.L = res.ref_end1 + 1; rv.rows = rv.L; rv.read = ref + (int64_t)res.ref_end1 * rdir; rv.rstep = -rdir;
        rv.ncols = res.read_end1 + 1; rv.ref = read + res.read_end1; rv.cstep = -1;
        WOut r; r.max = 0; r.col = -1; r.row = 0;
        if (rv.L > 0 && rv.ncols > 0) r = GEQ ? scanw_pass<GEQ, false, true>(rv, mem, gO, gE) : scanw_pass<GEQ, true, true>(rv, mem, gO, gE);
        if (r.max == 0) { res.ref_begin1 = regime ? 0 : -1; res.read_begin1 = res.read_end1; }
        else { res.ref_begin1 = res.ref_end1 - r.row; res.read_begin1 = res.read_end1 - r.col; }
    }
    return true;
}

// masked second-best column maximum, ssw.c:325-340 (8 bit) / 528-541 (16 bit); wave-parallel
__device__ void second_best_w(const uint16_t* colmax, int refLen, int end_ref, int maskLen, int word, int& score2, int& ref_end2)
{
    const int lane = threadIdx.x & 63;
    int e1 = end_ref - maskLen; if (e1 < 0) e1 = 0;
    int e2 = end_ref + maskLen; if (e2 > refLen) e2 = refLen;
    e2 += word ? 0 : 1;
    int bv = 0, bp = W_INF;
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

}  // namespace

// bytes of HBM workspace of one K1w task (SswTask.dir_off into SswParams.dirs): boundary rows of a chunk + the two hand-over arrays
size_t scanw_task_bytes(int read_len) {
    const size_t rows_cap = ((size_t)read_len + 16 + 63) & ~(size_t)63;
    return (size_t)W_NB * 2 * W_CPR_MAX * 64 * 4 + 2 * 2 * rows_cap * 2 + 256;
}

// Which forward pass first?  The reference runs the byte pass and, if a cell reached 255 - bias, the word pass (ssw.c:804-809).  The
// ORDER is free: a word pass whose maximum reaches 255 - bias proves that the byte pass would have overflowed (byte H >= word H), and
// then the byte pass need not run at all -- it costs an overflowing alignment half a pass on average before it is abandoned.  So the
// kernel guesses: exact 11-mers shared by the read and the window, counted through a hash table in the (still unused) profile block
// of LDS; an alignment long enough to overflow at ~13 % errors shares more than a hundred, unrelated sequences none.  A wrong guess costs
// time only: a word pass below 255 - bias is followed by the byte pass, whose verdict stands (scanw_align).
__device__ bool scanw_guess_overflow(const SswParams& p, const int8_t* read, const int L, const int8_t* ref, const int refLen, const int rdir, const int comp, uint32_t* lds_words)
{
    constexpr int K = 11, TAB = 4096;
    const int lane = threadIdx.x & 63;
    uint16_t* tab = (uint16_t*)lds_words;                // 8 KiB of the 12 KiB profile block
    for (int i = lane; i < TAB; i += 64) tab[i] = 0;
    __syncthreads();
    {
        const int per = (L + 63) / 64, i0 = lane * per, i1 = i0 + per < L ? i0 + per : L;
        uint32_t code = 0; int valid = 0;
        for (int i = i0; i < i1 + K - 1 && i < L; ++i) {
            const int c = (int)read[i] & 7;
            if (c > 3) { valid = 0; code = 0; continue; }
            code = ((code << 2) | (uint32_t)c) & ((1u << (2 * K)) - 1); ++valid;
            if (valid >= K) tab[code & (TAB - 1)] = (uint16_t)(0x8000u | (code >> 12));
        }
    }
    __syncthreads();
    int hits = 0;
    {
        const int per = (refLen + 63) / 64, j0 = lane * per, j1 = j0 + per < refLen ? j0 + per : refLen;
        uint32_t code = 0; int valid = 0;
        for (int j = j0; j < j1 + K - 1 && j < refLen; ++j) {
            const int c = ref_code((int)ref[(int64_t)j * rdir], comp);
            if (c > 3) { valid = 0; code = 0; continue; }
            code = ((code << 2) | (uint32_t)c) & ((1u << (2 * K)) - 1); ++valid;
            if (valid >= K) hits += tab[code & (TAB - 1)] == (uint16_t)(0x8000u | (code >> 12));
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) hits += __shfl_xor(hits, d);
    __syncthreads();                 // the table's block becomes the profile again
    return hits * p.max_match >= 64;
}

// one alignment, start to finish: the read against refLen columns from `ref` on (read backwards and complemented when ref_rc).
// force_word: the word regime without asking the byte pass (the decision was made for the whole window, see ssw_scanw_pick_kernel).
// Returns false when the reference would have returned NULL (score_size 0 and an 8-bit overflow): res.status says so.
template <bool GEQ>
__device__ bool scanw_align(const SswParams& p, const int8_t* read, const int8_t* ref, const int L, const int refLen, const int ref_rc, const int mask_len,
                            uint16_t* colmax, const WMem& mem, const bool force_word, SswResult& res)
{
    const int bias = p.bias, gO = p.gapO, gE = p.gapE;
    res.score1 = 0; res.score2 = 0; res.ref_begin1 = -1; res.ref_end1 = -1; res.read_begin1 = -1; res.read_end1 = 0;
    res.ref_end2 = 0; res.status = 0;

    // ---- forward: which regime?  (ssw.c:804-822; the order of ssw_wavefront.hip's kernel) -----------------------------------
    const int rdir = ref_rc ? -1 : 1;
    WIn in;
    in.read = read; in.rstep = 1; in.rcomp = 0; in.L = L; in.ref = ref; in.cstep = rdir; in.comp = ref_rc; in.ncols = refLen; in.terminate = 1 << 30;
    in.colmax = colmax;
    auto word_rows = [&](WIn& x) { x.S = GEQ ? (x.L + 7) / 8 : 0; x.rows = ((x.L + 7) / 8) * 8; x.overflow_at = 1 << 30; };
    auto byte_rows = [&](WIn& x) { x.S = 0; x.rows = ((x.L + 15) / 16) * 16; x.overflow_at = 255 - bias; };
    int regime = -1;
    WOut fw;
    bool byte_overflowed = force_word;
    // The reference's own order: the byte regime first (ssw.c:804).  It is abandoned within 64 rows of the first cell at 255 - bias, so an
    // alignment that does overflow pays a fraction of a pass for it, and one that does not (half of a mixed batch) needs no second pass
    // -- the anti-diagonal kernel's "word first when the bound allows an overflow" pays a whole pass there.
    int job_word = (p.score_size == 1 || force_word) ? 1 : 0;
    if (!job_word && p.score_size == 2 && L * p.max_match + bias >= 255 && refLen >= 64 && !p.no_guess)
        job_word = scanw_guess_overflow(p, read, L, ref, refLen, rdir, ref_rc, mem.prof) ? 1 : 0;
    while (regime < 0) {
        if (job_word) {
            word_rows(in);
            const WOut r = scanw_pass<GEQ, true>(in, mem, gO, gE);
            if (p.score_size == 1 || byte_overflowed || r.max + bias >= 255) { fw = r; regime = 1; }
            else job_word = 0;
        } else {
            byte_rows(in);
            const WOut r = scanw_pass<GEQ, false>(in, mem, gO, gE);
            if (!r.overflow) { fw = r; regime = 0; }
            else if (p.score_size == 0) { res.status = CLH_STATUS_OVERFLOW8; return false; }
            else { byte_overflowed = true; job_word = 1; }
        }
    }
    res.status = regime ? CLH_STATUS_WORD : 0;
    res.score1 = fw.max;
    if (fw.max == 0) { res.ref_end1 = regime ? 0 : -1; res.read_end1 = 0; }
    else { res.ref_end1 = fw.col; res.read_end1 = fw.row; }
    if (mask_len >= 15 && colmax) { __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); second_best_w(colmax, refLen, res.ref_end1, mask_len, regime, res.score2, res.ref_end2); }
    else { res.score2 = 0; res.ref_end2 = mask_len >= 15 ? 0 : -1; }

    // ---- reverse: begin coordinates (ssw.c:834-849) ---------------------------------------------------------------
    const bool want_begin = !(p.flag == 0 || (p.flag == 2 && res.score1 < p.filters));
    if (want_begin) {
        WIn rv;
        rv.L = res.read_end1 + 1; rv.read = read + res.read_end1; rv.rstep = -1;
        rv.ncols = res.ref_end1 + 1; rv.ref = ref + (int64_t)res.ref_end1 * rdir; rv.cstep = -rdir; rv.comp = ref_rc;
        rv.terminate = res.score1; rv.colmax = nullptr;
        WOut r;
        if (regime) { word_rows(rv); r = scanw_pass<GEQ, true>(rv, mem, gO, gE); }
        else { byte_rows(rv); rv.overflow_at = 1 << 30; r = scanw_pass<GEQ, false>(rv, mem, gO, gE); }
        if (r.max == 0) { res.ref_begin1 = regime ? 0 : -1; res.read_begin1 = res.read_end1; }
        else { res.ref_begin1 = res.ref_end1 - r.col; res.read_begin1 = res.read_end1 - r.row; }
    }
    return true;
}

__device__ __forceinline__ void scanw_mem_at(WMem& mem, uint8_t* ws, int L)
{
    mem.rows_cap = (int)((((size_t)L + 16 + 63) & ~(size_t)63));
    mem.bnd = (uint32_t*)ws;
    mem.cH = (short*)(mem.bnd + W_NB * 2 * W_CPR_MAX * 64);
    mem.cE = mem.cH + 2 * mem.rows_cap;
}

#define SCANW_LDS_SETUP \
    __shared__ __attribute__((aligned(16))) uint32_t s_prof[W_PROF_WORDS]; \
    __shared__ int s_mat[48]; \
    const int lane = threadIdx.x & 63; \
    if (lane < 48) { const int b = lane >> 3, q = lane & 7; s_mat[lane] = (b < p.n && q < p.n) ? (int)p.mat[b * p.n + q] : 0; } \
    __syncthreads(); \
    WMem mem; \
    mem.prof = s_prof; mem.mat = s_mat;

// persistent workgroups pull the class's tasks (heaviest first) from a counter; the HBM workspace belongs to the workgroup, not to the
// task (a batch of a million alignments needs no more of it than one of three thousand)
template <bool GEQ>
__global__ void __launch_bounds__(64, SCANW_WAVES) ssw_scanw_kernel(const SswParams p, const int ntasks, int* const counter, const long long ws_off, const int ws_slot)
{
    SCANW_LDS_SETUP
    uint8_t* const ws = p.dirs + ws_off + (long long)blockIdx.x * ws_slot;
    for (;;) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(counter, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= ntasks) break;
        const SswTask task = p.tasks[idx];
        scanw_mem_at(mem, ws, task.read_len);
        SswResult res;
        scanw_align<GEQ>(p, p.reads + task.read_off, p.refs + task.ref_off, task.read_len, task.ref_len, task.ref_rc, task.mask_len,
                         p.colmax ? p.colmax + task.colmax_off : nullptr, mem, false, res);
        if (lane == 0) p.results[task.out_index] = res;
        __syncthreads();
    }
}

// the transposed class: persistent workgroups as above; the HBM workspace is sized for 64 rows
template <bool GEQ>
__global__ void __launch_bounds__(64, SCANW_WAVES) ssw_scanw_tr_kernel(const SswParams p, const int ntasks, int* const counter, const long long ws_off, const int ws_slot)
{
    SCANW_LDS_SETUP
    uint8_t* const ws = p.dirs + ws_off + (long long)blockIdx.x * ws_slot;
    for (;;) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(counter, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= ntasks) break;
        const SswTask task = p.tasks[idx];
        scanw_mem_at(mem, ws, 64);
        SswResult res;
        scanw_align_tr<GEQ>(p, p.reads + task.read_off, p.refs + task.ref_off, task.read_len, task.ref_len, task.ref_rc, task.mask_len, mem, res);
        if (lane == 0) p.results[task.out_index] = res;
        __syncthreads();
    }
}

// ---- K1w on windows of 32 kb and more (class kRvScanWideSliced), behind the prefilter -------------------------------------------
// The alignments the 8-bit class of ssw_scan.hip does not take (reads of 255..4096 bases, scores that can pass 254), call-path
// options (no second best).  The bit-vector pass (ssw_prefilter.hip) has left block minima of d for every PIECE of the read
// (<= 254 rows each); a local alignment that ends in block b scores at most  M L - c D(b),  D(b) = sum over the pieces of the
// smallest minimum in blocks b - sb .. b  (tools/prefilter_model.py, "the read in pieces"; the word regime only lowers scores).
//   ssw_scanw_seed_kernel   D per block, its smallest value, the seed region around that block as one or two tasks;
//   ssw_scanw_queue_kernel  persistent workgroups run tasks (a task = the whole alignment of the read against a stretch of the window);
//   ssw_scanw_pick_kernel   S0 from the seed rows; blocks with D <= (M L - S0) / c in runs -> tasks (or the static slices);
//   ssw_scanw_combine_kernel the alignment's row: largest score, then smallest end column, then the earlier task.
// Which regime (ssw.c:804-809 decides it on the WHOLE window: any column at 255 - bias in the byte pass):
//   * M L - c min D + bias < 255: no cell of the window can overflow -- byte regime, tasks as they come;
//   * the seed task overflowed: word regime for the window -- every task runs the word pass only, S0 is the seed's word score;
//   * else undecided: the seed also runs in the word regime and S0 is THAT score (it is attained in either regime and not
//     above the byte score), every candidate region runs both ways.  The byte rows give the exact byte-regime maximum (the
//     candidates of a smaller S0 include those of a larger one); if none overflowed they are the answer, else the word rows are:
//     every column that holds the word-regime maximum Q >= S0 is a candidate, because word H <= byte H <= the bound.
// A refs buffer that is not 256-byte aligned has no minima: every static slice then runs both ways.
template <bool GEQ>
__global__ void __launch_bounds__(64) ssw_scanw_seed_kernel(const SswParams p)
{
    __shared__ uint8_t s_dm[6144];                 // block minima of one piece (windows below 1.5 Mb: at most 5861 blocks)
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x;
    const SswTask task = p.tasks[a];
    const PfWin pt = p.pf_win[a];
    WsTask t0, t1;
    t0.task = -1; t0.c_begin = t0.c_end = 0; t0.row = 0; t0.force_word = 0; t0.pad0 = t0.pad1 = t0.pad2 = 0;
    t1 = t0; t1.row = 1; t1.force_word = 1;
    int ub = 1 << 30, seed_block = -1;
    if (p.pf_dmin) {
        const int R = task.ref_len, L = task.read_len;
        const int span = L + (L * p.max_match + p.gapE - 1) / p.gapE;
        const int overlap = span + 32;
        const int sb = (span + kPfBlock - 1) / kPfBlock;
        const int cc = p.max_match < p.gapE ? p.max_match : p.gapE;
        uint16_t* D = p.ws_bound + pt.d_off;
        int key = 0x7fffffff;
        // the minima of one piece at a time through LDS (coalesced in, the window of sb + 1 blocks read from there), the sums in D
        for (int k = 0; k < pt.piece_count; ++k) {
            const uint8_t* dm = p.pf_dmin + p.pf_tasks[pt.piece_first + k].sub_off;
            for (int b = lane; b < pt.nsub; b += 64) s_dm[b] = dm[b];
            __syncthreads();
            for (int b = lane; b < pt.nsub; b += 64) {
                int mn = 255;
                for (int q = b - sb < 0 ? 0 : b - sb; q <= b; ++q) { const int v = s_dm[q]; mn = v < mn ? v : mn; }
                const int sum = (k ? (int)D[b] : 0) + mn;
                D[b] = (uint16_t)sum;
                if (k == pt.piece_count - 1) { const int v = (sum << 16) | (b & 0xffff); key = v < key ? v : key; }
            }
            __syncthreads();
        }
        // (blocks above 65535 -- windows above 16 Mb -- are not a case: the class rule keeps windows below 1.5 Mb)
        key = wave_min(key);
        const int kb = key & 0xffff, dmin = key >> 16;
        seed_block = kb;
        ub = p.max_match * L - cc * dmin;
        int c0 = kb * kPfBlock - pt.phase, c1 = c0 + kPfBlock;
        c0 = c0 < 0 ? 0 : c0; c1 = c1 > R ? R : c1;
        t0.task = a; t0.c_begin = c0 - overlap < 0 ? 0 : c0 - overlap; t0.c_end = c1;
        if (ub + p.bias >= 255 && p.score_size != 1) { t1.task = a; t1.c_begin = t0.c_begin; t1.c_end = t0.c_end; }
    }
    if (lane == 0) {
        p.ws_tasks[2 * a] = t0; p.ws_tasks[2 * a + 1] = t1;
        PfOut o; o.first = seed_block; o.count = 0; o.s0 = ub; o.pruned = 0;
        p.pf_out[a] = o;
    }
}

// tasks [first, first + count) of the table by persistent workgroups; count < 0: the candidate queue (its length is on the device)
template <bool GEQ>
__global__ void __launch_bounds__(64, SCANW_WAVES) ssw_scanw_queue_kernel(const SswParams p, const int first, const int count_arg)
{
    SCANW_LDS_SETUP
    const int total = count_arg >= 0 ? count_arg : p.pf_ctl->qcount;
    int* const next = count_arg >= 0 ? &p.pf_ctl->seed_next : &p.pf_ctl->qnext;
    uint8_t* const ws = p.dirs + p.ws_dirs_off + (int64_t)blockIdx.x * p.ws_slot_bytes;
    for (;;) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(next, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= total) break;
        const WsTask wt = p.ws_tasks[first + idx];
        if (wt.task < 0) continue;
        const SswTask task = p.tasks[wt.task];
        scanw_mem_at(mem, ws, task.read_len);
        const int rdir = task.ref_rc ? -1 : 1;
        SswResult res;
        scanw_align<GEQ>(p, p.reads + task.read_off, p.refs + task.ref_off + (int64_t)wt.c_begin * rdir, task.read_len, wt.c_end - wt.c_begin, task.ref_rc, 0,
                         nullptr, mem, wt.force_word != 0, res);
        res.ref_end2 = wt.c_begin;               // (no second best in this class: the field carries the task's first window column)
        if (lane == 0) p.results[p.ws_row0 + wt.task * kWsRows + wt.row] = res;
        __syncthreads();
    }
}

template <bool GEQ>
__global__ void __launch_bounds__(64) ssw_scanw_pick_kernel(const SswParams p)
{
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x;
    const SswTask task = p.tasks[a];
    const PfWin pt = p.pf_win[a];
    const int R = task.ref_len, L = task.read_len;
    const int span = L + (L * p.max_match + p.gapE - 1) / p.gapE;
    const int overlap = span + 32;
    int own = (R + 63) / 64; own = own < 8192 ? 8192 : own; own = own < 2 * overlap ? 2 * overlap : own;
    const int nstatic = (R + own - 1) / own;
    SswResult* const rows = p.results + p.ws_row0 + a * kWsRows;
    const int ub = p.pf_out[a].s0, seed_block = p.pf_out[a].first;
    // mode 1: tasks as they come (byte first); 2: word regime only; 3: both
    int mode = 3, S0 = 0, thr = 0, nrun = 0, pruned = 0, ov_c = overlap;
    const uint16_t* D = p.ws_bound ? p.ws_bound + pt.d_off : nullptr;
    if (p.score_size == 1) mode = 1;             // the caller asked for the word pass alone: one regime by construction
    if (p.pf_dmin) {
        const SswResult r0 = rows[0];
        if (p.score_size == 1 || ub + p.bias < 255) { mode = 1; S0 = r0.score1; }
        else if (r0.status & CLH_STATUS_WORD) { mode = 2; S0 = r0.score1; }
        else { mode = 3; S0 = rows[1].score1; }
        if (r0.status & CLH_STATUS_OVERFLOW8) S0 = 0;      // score_size 0 and an overflow: the whole window decides (static slices)
        if (S0 > 0) {
            const int cc = p.max_match < p.gapE ? p.max_match : p.gapE;
            thr = (p.max_match * L - S0) / cc;
            // (an alignment that scores S0 or more spans at most L + (M L - S0) / gapE columns: the candidate regions start that far early)
            const int span_s0 = L + (p.max_match * L - S0) / p.gapE + 32;
            ov_c = span_s0 < overlap ? span_s0 : overlap;
            long long cost = 0;
            for (int g = 0; g < pt.nsub; g += 64) {
                const int k = g + lane;
                const unsigned long long m = __ballot(k < pt.nsub && (int)D[k] <= thr);
                nrun += __popcll(m & ~(m << 1));
                cost += (long long)__popcll(m) * kPfBlock;
            }
            cost += (long long)nrun * ov_c;
            pruned = nrun <= 64 && cost < (long long)R + (long long)nstatic * overlap;
        }
    }
    const int count = pruned ? nrun : nstatic;
    const int per = mode == 3 ? 2 : 1;
    int first = 0;
    if (lane == 0) {
        first = atomicAdd(&p.pf_ctl->qcount, count * per);
        if (pruned) atomicAdd(&p.pf_ctl->n_pruned, 1);
        atomicAdd(&p.pf_ctl->cols_window, (unsigned long long)R);
    }
    first = __builtin_amdgcn_readfirstlane(first);
    WsTask* q = p.ws_tasks + 2 * gridDim.x + first;
    unsigned long long cols = 0;
    // reuse: the run is the seed block alone -- its region is the seed's, whose rows are there already (the seed that overflowed IS
    // the word-regime row: the byte pass was abandoned and the word pass ran)
    auto emit = [&](int k, int b0, int b1, bool reuse) {
        WsTask t;
        const int ovx = pruned ? ov_c : overlap;
        t.task = reuse ? -1 : a; t.c_begin = b0 - ovx < 0 ? 0 : b0 - ovx; t.c_end = b1; t.pad0 = t.pad1 = t.pad2 = 0;
        if (mode != 2) { t.row = 2 + 2 * k; t.force_word = 0; q[per * k] = t; if (reuse) rows[2 + 2 * k] = rows[0]; }
        if (mode != 1) { t.row = 3 + 2 * k; t.force_word = 1; q[per * k + per - 1] = t; if (reuse) rows[3 + 2 * k] = rows[mode == 2 ? 0 : 1]; }
        if (!reuse) cols += (unsigned long long)(t.c_end - t.c_begin) * per;
    };
    if (pruned) {
        int done = 0;
        for (int g = 0; g < pt.nsub; g += 64) {
            const int k = g + lane;
            const unsigned long long m = __ballot(k < pt.nsub && (int)D[k] <= thr);
            const unsigned long long starts = m & ~(m << 1);
            if ((starts >> lane) & 1ull) {
                const unsigned long long rest = ~(m >> lane);
                const int len = rest ? __builtin_ctzll(rest) : 64 - lane;
                const int rank = done + __popcll(starts & ((1ull << lane) - 1ull));
                int b0 = k * kPfBlock - pt.phase, b1 = (k + len) * kPfBlock - pt.phase;
                b0 = b0 < 0 ? 0 : b0; b1 = b1 > R ? R : b1;
                emit(rank, b0, b1, len == 1 && k == seed_block);
            }
            done += __popcll(starts);
        }
    } else {
        for (int sidx = lane; sidx < nstatic; sidx += 64) {
            const long long b = (long long)sidx * own;
            emit(sidx, (int)b, (int)(b + own > R ? R : b + own), false);
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) cols += __shfl_xor(cols, d);
    if (lane == 0) {
        atomicAdd(&p.pf_ctl->cols_scanned, cols);
        PfOut o; o.first = mode; o.count = count; o.s0 = S0; o.pruned = pruned;
        p.pf_out[a] = o;
    }
}

__global__ void __launch_bounds__(64) ssw_scanw_combine_kernel(const SswParams p)
{
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x;
    const SswTask task = p.tasks[a];
    const PfOut po = p.pf_out[a];
    const SswResult* rows = p.results + p.ws_row0 + a * kWsRows;
    const int mode = po.first;
    bool use_word = mode == 2;
    if (mode == 3) {
        bool ov = false;
        if (lane < po.count) ov = (rows[2 + 2 * lane].status & (CLH_STATUS_WORD | CLH_STATUS_OVERFLOW8)) != 0;
        use_word = __ballot(ov) != 0ull;
    }
    int v = -1, c = 0x7fffffff, k = 0x7fffffff;
    if (lane < po.count) {
        const SswResult r = rows[2 + 2 * lane + (use_word ? 1 : 0)];
        v = r.score1; k = lane;
        c = r.score1 > 0 ? r.ref_end1 + r.ref_end2 : 0x7fffffff;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v2 = __shfl_xor(v, d), c2 = __shfl_xor(c, d), k2 = __shfl_xor(k, d);
        const bool take = v2 > v || (v2 == v && (c2 < c || (c2 == c && k2 < k)));
        v = take ? v2 : v; c = take ? c2 : c; k = take ? k2 : k;
    }
    if (lane == 0) {
        SswResult r = rows[2 + 2 * k + (use_word ? 1 : 0)];
        const int base = r.ref_end2;
        if (r.score1 > 0) {
            r.ref_end1 += base;
            if (r.ref_begin1 >= 0) r.ref_begin1 += base;
        }
        r.ref_end2 = task.mask_len >= 15 ? 0 : -1;
        // score_size 0 and an overflow somewhere in the window: the reference returns NULL (ssw.c:810-813)
        if (p.score_size == 0) {
            bool any = false;
            for (int q = 0; q < po.count; ++q) any = any || (rows[2 + 2 * q].status & CLH_STATUS_OVERFLOW8);
            if (any) { r.score1 = 0; r.score2 = 0; r.ref_begin1 = -1; r.ref_end1 = -1; r.read_begin1 = -1; r.read_end1 = 0; r.status = CLH_STATUS_OVERFLOW8; }
        }
        p.results[task.out_index] = r;
    }
}

// (the first stage of the prefilter has been launched by the caller when with_prefilter is set)
hipError_t launch_ssw_scanw_filtered(bool geq, const SswParams& p, int ntasks, int nworkgroups, bool with_prefilter, hipStream_t stream)
{
    if (geq) {
        hipLaunchKernelGGL((ssw_scanw_seed_kernel<true>), dim3(ntasks), dim3(64), 0, stream, p);
        if (with_prefilter) hipLaunchKernelGGL((ssw_scanw_queue_kernel<true>), dim3(std::min(nworkgroups, 2 * ntasks)), dim3(64), 0, stream, p, 0, 2 * ntasks);
        hipLaunchKernelGGL((ssw_scanw_pick_kernel<true>), dim3(ntasks), dim3(64), 0, stream, p);
        hipLaunchKernelGGL((ssw_scanw_queue_kernel<true>), dim3(nworkgroups), dim3(64), 0, stream, p, 2 * ntasks, -1);
    } else {
        hipLaunchKernelGGL((ssw_scanw_seed_kernel<false>), dim3(ntasks), dim3(64), 0, stream, p);
        if (with_prefilter) hipLaunchKernelGGL((ssw_scanw_queue_kernel<false>), dim3(std::min(nworkgroups, 2 * ntasks)), dim3(64), 0, stream, p, 0, 2 * ntasks);
        hipLaunchKernelGGL((ssw_scanw_pick_kernel<false>), dim3(ntasks), dim3(64), 0, stream, p);
        hipLaunchKernelGGL((ssw_scanw_queue_kernel<false>), dim3(nworkgroups), dim3(64), 0, stream, p, 2 * ntasks, -1);
    }
    hipLaunchKernelGGL(ssw_scanw_combine_kernel, dim3(ntasks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_ssw_scanw_tr(bool geq, const SswParams& p, int ntasks, int nworkgroups, int* counter, long long ws_off, int ws_slot, hipStream_t stream)
{
    if (geq) hipLaunchKernelGGL((ssw_scanw_tr_kernel<true>), dim3(nworkgroups), dim3(64), 0, stream, p, ntasks, counter, ws_off, ws_slot);
    else hipLaunchKernelGGL((ssw_scanw_tr_kernel<false>), dim3(nworkgroups), dim3(64), 0, stream, p, ntasks, counter, ws_off, ws_slot);
    return hipGetLastError();
}

hipError_t launch_ssw_scanw(bool geq, const SswParams& p, int ntasks, int nworkgroups, int* counter, long long ws_off, int ws_slot, hipStream_t stream)
{
    if (geq) hipLaunchKernelGGL((ssw_scanw_kernel<true>), dim3(nworkgroups), dim3(64), 0, stream, p, ntasks, counter, ws_off, ws_slot);
    else hipLaunchKernelGGL((ssw_scanw_kernel<false>), dim3(nworkgroups), dim3(64), 0, stream, p, ntasks, counter, ws_off, ws_slot);
    return hipGetLastError();
}

}  // namespace clh
