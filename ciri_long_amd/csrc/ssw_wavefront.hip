// ssw_wavefront.hip -- K1: Smith-Waterman scores and coordinates, one alignment per wavefront (gfx950).
//
// Replaces, bit for bit, sw_sse2_byte / sw_sse2_word and the forward+reverse orchestration of ssw_align
// (reference: libs/striped_smith_waterman/ssw.c:123-345, 371-546, 779-849).  The recurrence actually
// implemented is the row-major form proven equal to the stripe mechanics in oracle/rowmajor_spec.c, laid out
// the way tools/wavefront_model.py describes:
//
//   * a 64-lane wave is 128 virtual lanes: the low and high 16-bit halves of every VGPR are two independent
//     DP chains, so each v_pk_* instruction updates two cells;
//   * virtual lane v owns RV consecutive rows and processes column t-v at step t (anti-diagonal wavefront);
//     the bottom-row H, the vertical-gap carry F and the running column maximum travel to virtual lane v+1
//     one step later: hi half <- own lo half, lo half <- previous lane's hi half (one DPP wave_shr + one
//     v_alignbit per value);
//   * the query profile (substitution scores of the lane's own rows against each of 6 base codes) lives in LDS
//     as lane-private rows, read with ds_read_b128; the reference window streams through one byte load per lane
//     per 64 steps and enters lane 0 through v_readlane;
//   * H/E of the previous column stay in VGPRs for the whole pass: nothing per-cell ever touches HBM.
//
// Arithmetic is the 16-bit arithmetic of the reference's word pass (signed saturating add, unsigned saturating
// subtract); the 8-bit pass computes the same numbers until it overflows, so it only adds the overflow test.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) {   // v_pk_add_i16 clamp
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {    // v_pk_max_i16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_subus(uint32_t a, uint32_t b) {  // v_pk_sub_u16 clamp
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t dup16(int v) { return (uint32_t)(v & 0xffff) * 0x10001u; }
// value of the previous lane (lane 0 receives `lane0`)
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// hand a packed (lo,hi) value to the next virtual lane: new lo = previous lane's hi, new hi = own lo
__device__ __forceinline__ uint32_t hand_down(uint32_t v, uint32_t lane0_lo) {
    uint32_t x = from_prev_lane(v, lane0_lo << 16);
    return __builtin_amdgcn_alignbit(v, x, 16);
}

// the same with 0 entering lane 0's low half: bound_ctrl supplies the zero, no v_mov to seed the destination
__device__ __forceinline__ uint32_t hand_down0(uint32_t v) {
    const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    return __builtin_amdgcn_alignbit(v, x, 16);
}

static constexpr int NEG16 = -32768;

struct PassOut {
    int max;        // best score (255 when the 8-bit pass overflowed)
    int col;        // first column (processing order) that reached it, -1 if max == 0
    int row;        // smallest row with that score in that column, clamped to readLen-1
    int overflow;   // 8-bit pass only
    int term_col;   // column at which the terminate score was met, -1 otherwise
};

struct PassIn {
    const int8_t* read;   // first row's base
    int rstep;            // +1 / -1
    int L;                // rows of the read
    const int8_t* ref;    // first column's base
    int cstep;            // +1 / -1
    int comp;             // reference bytes are complemented as they are read (reverse-complemented genome window)
    int ncols;
    int terminate;        // column maximum that ends the pass (ssw.c:296,499); > 32767 = never
    uint16_t* colmax;     // per-column maxima in processing order, or nullptr
};

// One pass.  RV = rows per virtual lane (capacity 128*RV rows).  WORD selects the row padding (8 vs 16).
// GEQ = (gapO == gapE) (gapO < gapE is rejected by the host).  Two consequences of GEQ:
//   * the 16-bit pass truncates vertical gaps at stripe starts (QUIRK, rowmajor_spec.c);
//   * H >= E and H >= F always, so E' = max(E-g, H-g) = H-g = F': both gap states collapse into one saturating
//     subtract per cell pair, 4 packed ops fewer (CIRI-long's call path scores 1/1/1/1, find_bsj.py:204).
// null_code: base code used for pipeline fill/drain columns; it must score 0 against every row.  5 (an extra
// profile row) in general, 4 when the matrix already scores code 4 ("N") as 0 everywhere, as CIRI-long's do
// (ssw_wrap.py:154-159) -- that saves one sixth of the LDS footprint.
//
// Reads longer than the 128*RV rows of a launch class are cut into row strips (STRIPS): every strip is a full pass
// over all columns; the bottom row of strip s (its H, its vertical-gap carry and the running column maximum, per
// column) goes through HBM to virtual lane 0 of strip s+1.  row_base = read row held by slot 0 (negative: leading
// dummy slots, the single-strip layout).
struct StripIo {
    int row_base;          // read row of slot 0
    const uint2* bnd_in;   // per column {H | carry << 16, colmax} of the strip above, or nullptr
    uint2* bnd_out;        // same, produced for the strip below, or nullptr
    int last;              // last strip: owns the finished column maxima (terminate test, colmax output)
};

// LEAN = a pass that neither reports column maxima nor ends at a terminate score (the forward pass of the call path,
// want_score2 off): the column-maximum chain down the lanes, the finished-column bookkeeping and the terminate test
// are compiled out -- about a fifth of the per-step instructions of the short-read classes.
template <int RV, bool WORD, bool GEQ, bool STRIPS, bool LEAN>
__device__ PassOut run_strip(const PassIn& in, const StripIo io, uint32_t* __restrict__ lds_prof, const int* __restrict__ lds_mat,
                             int gapO, int gapE, int bias, const int null_code, int& exceeded_out)
{
    const int CODE_NULL = null_code;
    constexpr bool QUIRK = WORD && GEQ;
    constexpr int CH = (RV + 3) / 4;             // 16-byte chunks of profile per lane and base
    constexpr int BASE_STRIDE = CH * 1024;       // bytes between two bases' profiles
    const int lane = threadIdx.x & 63;
    const int W = WORD ? 8 : 16;
    const int S = (in.L + W - 1) / W;
    const int rows = S * W;
    const int off = -io.row_base;                // slot -> read row: row = slot - off

    // ---- query profile, lane-private rows in LDS: prof[base][chunk][lane][4] -------------------------------
    uint32_t cut[QUIRK ? RV : 1];
    const int Lm1 = in.L - 1;
#pragma unroll
    for (int k = 0; k < RV; ++k) {
        const int rlo = lane * 2 * RV + k - off, rhi = rlo + RV;
        // clamped, unconditional loads; dummy (row < 0) and wildcard (row >= L) rows are selected afterwards
        const int clo = rlo < 0 ? 0 : (rlo > Lm1 ? Lm1 : rlo), chi = rhi < 0 ? 0 : (rhi > Lm1 ? Lm1 : rhi);
        const int qlo = (int)in.read[(int64_t)clo * in.rstep] & 7, qhi = (int)in.read[(int64_t)chi * in.rstep] & 7;
        for (int b = 0; b <= null_code; ++b) {
            const int mlo = lds_mat[b * 8 + qlo], mhi = lds_mat[b * 8 + qhi];
            // before the read or past its padded length: dummy row (pins H/E/F at 0); past the read: wildcard row
            const int slo = (rlo < 0 || rlo >= rows) ? NEG16 : (rlo > Lm1 ? 0 : mlo);
            const int shi = (rhi < 0 || rhi >= rows) ? NEG16 : (rhi > Lm1 ? 0 : mhi);
            lds_prof[((b * CH + (k >> 2)) * 64 + lane) * 4 + (k & 3)] = (uint32_t)(slo & 0xffff) | ((uint32_t)shi << 16);
        }
        if (QUIRK) {
            uint32_t m = 0xffffffffu;
            if (rlo > 0 && rlo < rows && rlo % S == 0) m &= 0xffff0000u;
            if (rhi > 0 && rhi < rows && rhi % S == 0) m &= 0x0000ffffu;
            cut[k] = m;
        }
    }

    // ---- state ------------------------------------------------------------------------------------------
    // H of the previous column is kept twice (HA/HB) and the step body is instantiated for both roles, so that the
    // "previous column" registers never have to be rotated at the loop back-edge.
    uint32_t HA[RV], HB[RV], E[RV], SH[RV];
    constexpr bool QUAD = LEAN && RV <= 4;      // short-read classes of the lean pass: four steps per resolution (two more H files)
    uint32_t HC[QUAD ? RV : 1], HD[QUAD ? RV : 1];
#pragma unroll
    for (int k = 0; k < RV; ++k) { HA[k] = 0; HB[k] = 0; E[k] = 0; SH[k] = 0; }
    uint32_t outH = 0, outC = 0, outM = 0, diagIn = 0, best = 0, RB = dup16(CODE_NULL);
    int colLo = -1, colHi = -1;
    uint32_t flags = 0;                           // bit0 overflow, bit1 exceeded (any lane)
    const uint32_t gO2 = dup16(gapO), gE2 = dup16(gapE);
    const uint32_t ovf2 = dup16(254 - bias);      // cm > 254-bias  <=>  cm + bias >= 255
    const uint32_t term2 = dup16(in.terminate > 32767 ? 32767 : in.terminate);
    const int ncols = in.ncols;
    const char* prof_bytes = (const char*)lds_prof;

    auto load_chunk = [&](int t0) -> int {
        int j = t0 + lane;
        return j < ncols ? ref_code((int)in.ref[(int64_t)j * in.cstep], in.comp) : CODE_NULL;
    };
    int term_col = -1, stop = 0;
    const int nsteps = ncols > 0 ? ncols + 127 : 0;

    // one step of the wavefront: reads the previous column from HR, writes the current one to HW.
    // Returns 1 when the pass must end (8-bit overflow or terminate score met).
    auto step = [&](const int t, const int sb, const uint32_t bHC, const uint32_t bM, uint32_t (&HR)[RV], uint32_t (&HW)[RV],
                    int& ring, int& ringHC, uint32_t& cmOut, int& cmLastOut) -> int {
        RB = hand_down(RB, (uint32_t)sb);
        const uint32_t aLo = (RB & 0xffffu) * BASE_STRIDE + lane * 16;
        const uint32_t aHi = (RB >> 16) * BASE_STRIDE + lane * 16;
        uint32_t inH, inC, inM = 0;
        if constexpr (STRIPS) { inH = hand_down(outH, bHC & 0xffffu); inC = hand_down(outC, bHC >> 16); inM = hand_down(outM, bM); }
        else { inH = hand_down0(outH); inC = hand_down0(outC); if constexpr (!LEAN) inM = hand_down0(outM); }
        uint32_t F = inC, diag = diagIn, cm = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint4 pl = *(const uint4*)(prof_bytes + aLo + c * 1024);
            const uint4 ph = *(const uint4*)(prof_bytes + aHi + c * 1024);
            const uint32_t plv[4] = {pl.x, pl.y, pl.z, pl.w}, phv[4] = {ph.x, ph.y, ph.z, ph.w};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = c * 4 + kk;
                if (k < RV) {
                    const uint32_t s = (plv[kk] & 0xffffu) | (phv[kk] & 0xffff0000u);
                    const uint32_t tt = pk_adds(diag, s);
                    diag = HR[k];
                    uint32_t h;
                    if (QUIRK) {
                        const uint32_t Fm = F & cut[k];
                        h = pk_max(pk_max(tt, E[k]), Fm);
                        HW[k] = pk_max(h, F);
                        F = pk_subus(h, gO2);
                        E[k] = F;
                    } else if (GEQ) {
                        h = pk_max(pk_max(tt, E[k]), F);
                        HW[k] = h;
                        F = pk_subus(h, gO2);
                        E[k] = F;
                    } else {
                        h = pk_max(pk_max(tt, E[k]), F);
                        HW[k] = h;
                        const uint32_t hg = pk_subus(h, gO2);
                        E[k] = pk_max(pk_subus(E[k], gE2), hg);
                        F = pk_max(pk_subus(F, gE2), hg);
                    }
                    cm = pk_max(cm, h);
                }
            }
        }
        diagIn = inH;
        outH = HW[RV - 1];
        outC = F;
        if constexpr (!LEAN) outM = pk_max(inM, cm);

        // per-lane best (first column wins, strict >) and a snapshot of that column
        const int jLo = t - 2 * lane, jHi = jLo - 1;
        uint32_t cmv = cm;
        if (t < 127 || t >= ncols) {     // some virtual lane is outside the matrix (wave-uniform test: only the ramps pay)
            const uint32_t vm = ((uint32_t)jLo < (uint32_t)ncols ? 0x0000ffffu : 0u) | ((uint32_t)jHi < (uint32_t)ncols ? 0xffff0000u : 0u);
            cmv = cm & vm;
        }
        cmOut = cmv;                 // best cell, overflow and terminate are resolved once per pair of steps (below)
        // the last virtual lane has the finished column maximum of column t-127
        const int jl = t - 127;
        cmLastOut = -1;
        if (!LEAN && jl >= 0 && jl < ncols) {
            const int cmLast = (int)((uint32_t)__builtin_amdgcn_readlane((int)outM, 63) >> 16);
            ring = lane == (t & 63) ? cmLast : ring;
            if (STRIPS) {
                const uint32_t hL = (uint32_t)__builtin_amdgcn_readlane((int)outH, 63) >> 16, cL = (uint32_t)__builtin_amdgcn_readlane((int)outC, 63) >> 16;
                ringHC = lane == (t & 63) ? (int)(hL | (cL << 16)) : ringHC;
            }
            cmLastOut = cmLast;
        }
        return 0;
    };

    // 64 steps per block: the block's reference bases were loaded one block earlier (lane i <-> step t0+i) and
    // its finished column maxima leave through one coalesced store per block; no vector-memory op inside.
    auto load_bnd = [&](int t0) -> uint2 {
        const int j = t0 + lane;
        return (STRIPS && io.bnd_in && j < ncols) ? io.bnd_in[j] : make_uint2(0u, 0u);
    };
    int nxt = load_chunk(0);
    uint2 nxtb = load_bnd(0);
    for (int t0 = 0; t0 < nsteps && !stop; t0 += 64) {
        int chunk = nxt;
        uint2 cb = nxtb;
        asm volatile("" : "+v"(chunk));      // the wait for last block's prefetch lands here, not inside the step loop
        if (STRIPS) asm volatile("" : "+v"(cb.x), "+v"(cb.y));
        nxt = load_chunk(t0 + 64);
        nxtb = load_bnd(t0 + 64);
        int ring = 0, ringHC = 0;
        int done = 64;
        // per-lane best (first column wins, strict >) and a snapshot of that column, for the step that wrote Hst
        auto resolve = [&](const uint32_t cmv, const int t, const uint32_t (&Hst)[RV]) {
            const uint32_t nb = pk_max(best, cmv);
            const uint32_t ch = nb ^ best;
            best = nb;
            const int jLo = t - 2 * lane, jHi = jLo - 1;
            const uint32_t m = ((ch & 0xffffu) ? 0x0000ffffu : 0u) | ((ch >> 16) ? 0xffff0000u : 0u);
            colLo = (ch & 0xffffu) ? jLo : colLo;
            colHi = (ch >> 16) ? jHi : colHi;
#pragma unroll
            for (int k = 0; k < RV; ++k) SH[k] = (Hst[k] & m) | (SH[k] & ~m);
        };
        if constexpr (QUAD) {
            // four steps back to back through four H register files (HA->HB->HC->HD->HA), one vote for the four columns
            for (int u = 0; u < 64; u += 4) {
                uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                int l0 = -1;
                step(t0 + u, __builtin_amdgcn_readlane(chunk, u), 0u, 0u, HA, HB, ring, ringHC, c0, l0);
                step(t0 + u + 1, __builtin_amdgcn_readlane(chunk, u + 1), 0u, 0u, HB, HC, ring, ringHC, c1, l0);
                step(t0 + u + 2, __builtin_amdgcn_readlane(chunk, u + 2), 0u, 0u, HC, HD, ring, ringHC, c2, l0);
                step(t0 + u + 3, __builtin_amdgcn_readlane(chunk, u + 3), 0u, 0u, HD, HA, ring, ringHC, c3, l0);
                const uint32_t cm4 = pk_max(pk_max(c0, c1), pk_max(c2, c3));
                if (!WORD) { if (__builtin_amdgcn_ballot_w64(pk_subus(cm4, ovf2) != 0u)) { stop = 2; done = u + 1; break; } }
                if (__builtin_amdgcn_ballot_w64(pk_max(best, cm4) != best)) {
                    resolve(c0, t0 + u, HB); resolve(c1, t0 + u + 1, HC); resolve(c2, t0 + u + 2, HD); resolve(c3, t0 + u + 3, HA);
                }
            }
        } else
        for (int u = 0; u < 64; u += 2) {
            uint32_t cmA = 0, cmB = 0;
            int lastA = -1, lastB = -1;
            step(t0 + u, __builtin_amdgcn_readlane(chunk, u), STRIPS ? (uint32_t)__builtin_amdgcn_readlane((int)cb.x, u) : 0u,
                 STRIPS ? (uint32_t)__builtin_amdgcn_readlane((int)cb.y, u) : 0u, HA, HB, ring, ringHC, cmA, lastA);
            step(t0 + u + 1, __builtin_amdgcn_readlane(chunk, u + 1), STRIPS ? (uint32_t)__builtin_amdgcn_readlane((int)cb.x, u + 1) : 0u,
                 STRIPS ? (uint32_t)__builtin_amdgcn_readlane((int)cb.y, u + 1) : 0u, HB, HA, ring, ringHC, cmB, lastB);
            // both steps at once: overflow, terminate and "did any lane improve".  The order of the two columns matters only
            // inside the rare branches (HB holds the first step's column, HA the second's).  A pass that ends at the first
            // step has computed one column too many: nothing of it is kept (done excludes it from the column-maximum
            // stores; the exceeded flag may see it, which at worst triggers the exact re-run).
            const uint32_t cm2 = pk_max(cmA, cmB);
            if (!WORD) { if (__builtin_amdgcn_ballot_w64(pk_subus(cm2, ovf2) != 0u)) { stop = 2; done = u + 1; break; } }
            if constexpr (!LEAN) {
                flags |= (pk_subus(cm2, term2) != 0u) ? 2u : 0u;
                if (io.last && lastA == in.terminate) { resolve(cmA, t0 + u, HB); term_col = t0 + u - 127; stop = 1; done = u + 1; break; }
            }
            if (__builtin_amdgcn_ballot_w64(pk_max(best, cm2) != best)) { resolve(cmA, t0 + u, HB); resolve(cmB, t0 + u + 1, HA); }
            if constexpr (!LEAN) {
                if (io.last && lastB == in.terminate) { term_col = t0 + u + 1 - 127; stop = 1; done = u + 2; break; }
            }
        }
        const int col = t0 - 127 + lane;
        if (in.colmax && io.last) {
            if (lane < done && col >= 0 && col < ncols) in.colmax[col] = (uint16_t)ring;
        }
        if (STRIPS && io.bnd_out) {
            if (lane < done && col >= 0 && col < ncols) io.bnd_out[col] = make_uint2((uint32_t)ringHC, (uint32_t)ring);
        }
    }
    if (stop == 2) { PassOut o; o.max = 255; o.col = -1; o.row = 0; o.overflow = 1; o.term_col = -1; return o; }

    // ---- wave reduction: (score desc, column asc, virtual lane asc) ------------------------------------------
    const int exceeded = __builtin_amdgcn_ballot_w64((flags & 2u) != 0u) != 0;
    exceeded_out |= exceeded;
    PassOut o;
    o.overflow = 0;
    o.term_col = term_col;
    if (!STRIPS && term_col >= 0 && exceeded) {   // caller re-runs on columns [0, term_col] without early stop
        o.max = -1; o.col = -1; o.row = 0;
        return o;
    }
    int bLo = (int)(best & 0xffffu), bHi = (int)(best >> 16);
    int val = bLo, col = colLo, vl = 2 * lane;
    if (bHi > bLo || (bHi == bLo && bHi > 0 && colHi < colLo)) { val = bHi; col = colHi; vl = 2 * lane + 1; }
    if (val == 0) col = 0x7fffffff;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v2 = __shfl_xor(val, d), c2 = __shfl_xor(col, d), l2 = __shfl_xor(vl, d);
        const bool take = v2 > val || (v2 == val && (c2 < col || (c2 == col && l2 < vl)));
        val = take ? v2 : val; col = take ? c2 : col; vl = take ? l2 : vl;
    }
    o.max = val;
    if (val == 0) { o.col = -1; o.row = 0; return o; }
    o.col = col;
    // the winning virtual lane finds the first of its rows holding the score in the snapshot column
    int kfound = RV - 1;
    const int half = vl & 1;
#pragma unroll
    for (int k = RV - 1; k >= 0; --k) {
        const int hv = half ? (int)(SH[k] >> 16) : (int)(SH[k] & 0xffffu);
        kfound = (hv == val) ? k : kfound;
    }
    const int owner = vl >> 1;
    const int krow = __builtin_amdgcn_readlane(kfound, owner);   // owner is wave-uniform
    int row = vl * RV + krow - off;
    o.row = row < in.L - 1 ? row : in.L - 1;
    return o;
}

template <int RV, bool WORD, bool GEQ, bool STRIPS>
__device__ PassOut run_pass(const PassIn& in, uint32_t* __restrict__ lds_prof, const int* __restrict__ lds_mat,
                            int gapO, int gapE, int bias, const int null_code, uint2* bnd)
{
    const int W = WORD ? 8 : 16;
    const int rows = ((in.L + W - 1) / W) * W;
    constexpr int CAP = 128 * RV;
    int exceeded = 0;
    if (!STRIPS || rows <= CAP) {
        StripIo io; io.row_base = rows - CAP; io.bnd_in = nullptr; io.bnd_out = nullptr; io.last = 1;
        // not in the row-strip kernel (reads above 4096 bases): with a third inlined variant of the pass that kernel stops
        // terminating on gfx950 (ROCm 7.2 code generation; the same source without this call runs) -- and it has no use for it
        if constexpr (!STRIPS) {
            if (!in.colmax && in.terminate > 32767) return run_strip<RV, WORD, GEQ, false, true>(in, io, lds_prof, lds_mat, gapO, gapE, bias, null_code, exceeded);
        }
        return run_strip<RV, WORD, GEQ, false, false>(in, io, lds_prof, lds_mat, gapO, gapE, bias, null_code, exceeded);
    }
    const int ns = (rows + CAP - 1) / CAP;
    uint2* buf[2] = {bnd, bnd + ((in.ncols + 63) & ~63)};
    PassOut best; best.max = 0; best.col = -1; best.row = 0; best.overflow = 0; best.term_col = -1;
    for (int s = 0; s < ns; ++s) {
        StripIo io;
        io.row_base = s * CAP; io.bnd_in = s > 0 ? buf[(s - 1) & 1] : nullptr; io.bnd_out = s < ns - 1 ? buf[s & 1] : nullptr; io.last = s == ns - 1;
        const PassOut o = run_strip<RV, WORD, GEQ, true, false>(in, io, lds_prof, lds_mat, gapO, gapE, bias, null_code, exceeded);
        if (o.overflow) return o;
        if (io.last) best.term_col = o.term_col;
        // strips are in row order: on a full tie the earlier strip (smaller rows) stays
        if (o.max > best.max || (o.max == best.max && o.max > 0 && o.col < best.col)) { best.max = o.max; best.col = o.col; best.row = o.row; }
        __syncthreads();     // the boundary rows written by this strip are read by the next one (same wave, through HBM)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (best.term_col >= 0 && exceeded) { best.max = -1; best.col = -1; best.row = 0; }
    return best;
}

// masked second-best column maximum, ssw.c:325-340 (8 bit) / 528-541 (16 bit); wave-parallel
__device__ void second_best(const uint16_t* colmax, int refLen, int end_ref, int maskLen, int word, int& score2, int& ref_end2)
{
    const int lane = threadIdx.x & 63;
    int e1 = end_ref - maskLen; if (e1 < 0) e1 = 0;
    int e2 = end_ref + maskLen; if (e2 > refLen) e2 = refLen;
    e2 += word ? 0 : 1;
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

// half-rate packed ops keep a SIMD busy with two waves; asking for more only causes spills
constexpr int waves_per_simd(int rv) { return rv <= 12 ? 3 : (rv <= 16 ? 2 : 1); }

template <int RV, bool GEQ, bool STRIPS>
__global__ void __launch_bounds__(64, waves_per_simd(RV)) ssw_align_kernel(const SswParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* lds_prof = lds;
    int* lds_mat = (int*)(lds + (p.null_code + 1) * ((RV + 3) / 4) * 256);
    const int lane = threadIdx.x & 63;
    if (lane < 48) {   // 6 base codes x 8 query codes; anything outside the n x n matrix scores 0
        const int b = lane >> 3, q = lane & 7;
        lds_mat[lane] = (b < p.n && q < p.n) ? (int)p.mat[b * p.n + q] : 0;
    }
    __syncthreads();

    const SswTask task = p.tasks[blockIdx.x];
    const int8_t* read = p.reads + task.read_off;
    const int8_t* ref = p.refs + task.ref_off;
    const int L = task.read_len, refLen = task.ref_len;
    uint16_t* colmax = p.colmax ? p.colmax + task.colmax_off : nullptr;
    uint2* bnd = STRIPS ? (uint2*)(p.dirs + task.dir_off) : nullptr;     // strip boundary rows (2 x ncols x 8 bytes)
    const int bias = p.bias, gO = p.gapO, gE = p.gapE;
    SswResult res;
    res.score1 = 0; res.score2 = 0; res.ref_begin1 = -1; res.ref_end1 = -1; res.read_begin1 = -1; res.read_end1 = 0;
    res.ref_end2 = 0; res.status = 0;

    // ---- forward: which regime?  (ssw.c:804-822; tools/wavefront_model.py:wf_align) ---------------------------
    int regime = -1;
    PassOut fw;
    bool byte_overflowed = false;
    int job_word = (p.score_size == 1 || (p.score_size == 2 && p.max_match * L + bias >= 255)) ? 1 : 0;
    // a window-slice task (clh_api.hip) mostly sees background: in the reference's own order -- 8-bit pass first, 16-bit on overflow
    // (ssw.c:804-809) -- it needs one pass instead of two; a slice that does overflow ends in the 16-bit regime either way
    if (task.out_index >= p.n_real && p.score_size == 2) job_word = 0;
    PassIn in;
    const int rdir = task.ref_rc ? -1 : 1;       // physical direction of the reference in memory
    in.read = read; in.rstep = 1; in.L = L; in.ref = ref; in.cstep = rdir; in.comp = task.ref_rc; in.ncols = refLen; in.terminate = 1 << 30;
    in.colmax = colmax;
    while (regime < 0) {
        if (job_word) {
            PassOut r = run_pass<RV, true, GEQ, STRIPS>(in, lds_prof, lds_mat, gO, gE, 0, p.null_code, bnd);
            if (p.score_size == 1 || byte_overflowed || r.max + bias >= 255) { fw = r; regime = 1; }
            else job_word = 0;
        } else {
            PassOut r = run_pass<RV, false, GEQ, STRIPS>(in, lds_prof, lds_mat, gO, gE, bias, p.null_code, bnd);
            if (!r.overflow) { fw = r; regime = 0; }
            else if (p.score_size == 0) { res.status = CLH_STATUS_OVERFLOW8; if (lane == 0) p.results[task.out_index] = res; return; }
            else { byte_overflowed = true; job_word = 1; }
        }
    }
    res.status = regime ? CLH_STATUS_WORD : 0;
    res.score1 = fw.max;
    if (fw.max == 0) { res.ref_end1 = regime ? 0 : -1; res.read_end1 = 0; }
    else { res.ref_end1 = fw.col; res.read_end1 = fw.row; }
    if (task.mask_len >= 15 && colmax) second_best(colmax, refLen, res.ref_end1, task.mask_len, regime, res.score2, res.ref_end2);
    else { res.score2 = 0; res.ref_end2 = task.mask_len >= 15 ? 0 : -1; }

    // ---- reverse: begin coordinates (ssw.c:834-849) ---------------------------------------------------------
    const bool want_begin = !(p.flag == 0 || (p.flag == 2 && res.score1 < p.filters));
    if (want_begin) {
        PassIn rv;
        rv.L = res.read_end1 + 1; rv.read = read + res.read_end1; rv.rstep = -1;
        rv.ncols = res.ref_end1 + 1; rv.ref = ref + (int64_t)res.ref_end1 * rdir; rv.cstep = -rdir; rv.comp = task.ref_rc;
        rv.terminate = res.score1; rv.colmax = nullptr;
        PassOut r;
        for (;;) {
            if (regime) r = run_pass<RV, true, GEQ, STRIPS>(rv, lds_prof, lds_mat, gO, gE, 0, p.null_code, bnd);
            else r = run_pass<RV, false, GEQ, STRIPS>(rv, lds_prof, lds_mat, gO, gE, bias, p.null_code, bnd);
            if (r.max >= 0) break;
            rv.ncols = r.term_col + 1; rv.terminate = 1 << 30;     // see PassOut: rare re-run
        }
        if (r.max == 0) { res.ref_begin1 = regime ? 0 : -1; res.read_begin1 = res.read_end1; }
        else { res.ref_begin1 = res.ref_end1 - r.col; res.read_begin1 = res.read_end1 - r.row; }
    }
    if (lane == 0) p.results[task.out_index] = res;
}

}  // namespace clh

// -------------------------------------------------------------------------------------------------------------
// launcher (host)
// -------------------------------------------------------------------------------------------------------------
namespace clh {

template <int RV, bool GEQ, bool STRIPS = false>
static hipError_t launch_one(const SswParams& p, int ntasks, hipStream_t stream)
{
    const size_t lds_bytes = (size_t)(p.null_code + 1) * ((RV + 3) / 4) * 1024 + 64 * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)ssw_align_kernel<RV, GEQ, STRIPS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           6 * ((RV + 3) / 4) * 1024 + 256);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((ssw_align_kernel<RV, GEQ, STRIPS>), dim3(ntasks), dim3(64), lds_bytes, stream, p);
    return hipGetLastError();
}

// The 30 kernel instantiations take minutes to compile, so the file is built as four objects (Makefile: -DCLH_K1_PART=0..3),
// each with a quarter of the row classes; part 0 owns the dispatcher.  Without the macro everything is in one object.
#ifndef CLH_K1_PART
#define CLH_K1_PART (-1)
#endif
#define CLH_K1_HAS(part) (CLH_K1_PART < 0 || CLH_K1_PART == (part))

template <bool GEQ>
static hipError_t launch_rv(int rv, const SswParams& p, int ntasks, hipStream_t stream)
{
#define CLH_CASE(R) case R: return launch_one<R, GEQ>(p, ntasks, stream);
#if defined(CLH_STRIPS_BUILD)      // debugging builds with one class: seconds instead of minutes
    if (rv == kRvStrips) return launch_one<32, GEQ, true>(p, ntasks, stream);
    switch (rv) { CLH_CASE(1) default: return hipErrorInvalidValue; }
#elif defined(CLH_PROBE_BUILD)
    switch (rv) { CLH_CASE(8) default: return hipErrorInvalidValue; }
#else
#if CLH_K1_HAS(3)
    if (rv == kRvStrips) return launch_one<32, GEQ, true>(p, ntasks, stream);   // reads longer than 4096 bases: row strips
#endif
    switch (rv) {
#if CLH_K1_HAS(0)
        CLH_CASE(1) CLH_CASE(2) CLH_CASE(3) CLH_CASE(4) CLH_CASE(5)
#endif
#if CLH_K1_HAS(1)
        CLH_CASE(6) CLH_CASE(7) CLH_CASE(8) CLH_CASE(10)
#endif
#if CLH_K1_HAS(2)
        CLH_CASE(12) CLH_CASE(16) CLH_CASE(20)
#endif
#if CLH_K1_HAS(3)
        CLH_CASE(24) CLH_CASE(32)
#endif
        default: return hipErrorInvalidValue;
    }
#endif
#undef CLH_CASE
}

#if CLH_K1_PART == 1
hipError_t launch_ssw_part1(int rv, bool geq, const SswParams& p, int ntasks, hipStream_t stream) { return geq ? launch_rv<true>(rv, p, ntasks, stream) : launch_rv<false>(rv, p, ntasks, stream); }
#elif CLH_K1_PART == 2
hipError_t launch_ssw_part2(int rv, bool geq, const SswParams& p, int ntasks, hipStream_t stream) { return geq ? launch_rv<true>(rv, p, ntasks, stream) : launch_rv<false>(rv, p, ntasks, stream); }
#elif CLH_K1_PART == 3
hipError_t launch_ssw_part3(int rv, bool geq, const SswParams& p, int ntasks, hipStream_t stream) { return geq ? launch_rv<true>(rv, p, ntasks, stream) : launch_rv<false>(rv, p, ntasks, stream); }
#else
const int kRvClasses[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32};
const int kNumRvClasses = sizeof(kRvClasses) / sizeof(kRvClasses[0]);

#if CLH_K1_PART == 0
hipError_t launch_ssw_part1(int rv, bool geq, const SswParams& p, int ntasks, hipStream_t stream);
hipError_t launch_ssw_part2(int rv, bool geq, const SswParams& p, int ntasks, hipStream_t stream);
hipError_t launch_ssw_part3(int rv, bool geq, const SswParams& p, int ntasks, hipStream_t stream);
#endif

hipError_t launch_ssw(int rv, bool geq, const SswParams& p, int ntasks, hipStream_t stream)
{
#if CLH_K1_PART == 0
    if (rv == kRvStrips || rv >= 24) return launch_ssw_part3(rv, geq, p, ntasks, stream);
    if (rv >= 12) return launch_ssw_part2(rv, geq, p, ntasks, stream);
    if (rv >= 6) return launch_ssw_part1(rv, geq, p, ntasks, stream);
#endif
    return geq ? launch_rv<true>(rv, p, ntasks, stream) : launch_rv<false>(rv, p, ntasks, stream);
}
#endif

}  // namespace clh
