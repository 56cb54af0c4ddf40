// ccs_poa.hip -- K2 ccs_scan_kernel (repeat period + copy boundaries) and K3 poa_consensus_kernel (partial-order
// alignment of the copies, heaviest-path consensus) for gfx950.
//
// What they replace: pyccs.find_consensus (called at CIRI_long/find_ccs.py:14) and the spoa engine inside it.  Those are
// external packages that exist neither in the reference tree nor in this environment: PARITY UNPINNED.  Both kernels
// implement, bit for bit, what oracle/ccs_oracle.c ("clh-ccs v1": period and copy boundaries, this project's own
// specification) and oracle/poa_oracle.c ("clh-poa v2": a restatement of the published spoa algorithm -- two-piece gap
// cost, local/global/overlap alignment, heaviest bundle) state; read those headers for every rule and tie-break.
// One read per wavefront, one wavefront per workgroup.
//
// K2: 8-mer codes of the read sit in LDS.  Matches per offset are counted per PAIR of equal 8-mers (positions chained per
//     hash bucket with LDS atomics, each pair visited once: O(L * copies) instead of O(L^2/4) comparisons); the smoothed
//     maximum, the harmonic test and the per-copy boundary search (same chains, a histogram over the candidate offsets)
//     are lane-parallel with shuffle reductions.
// K3: persistent waves pull reads from an atomic counter; each owns a workspace slot in HBM (graph arrays, one byte per
//     DP cell for the back-track, the few rows a far successor reads; slots are sized for the common case, the few reads
//     that need more claim one of a handful of large slots).  A DP row (one graph node) is computed by the 64 lanes over
//     the sequence positions: the diagonal and the two vertical gap states over the node's in-edges are independent per
//     position, the two horizontal gap states are max-plus prefix scans (DPP).  The back-track and the heaviest-bundle
//     pass are sequential by nature and run wave-uniformly on data staged in registers; the graph update is data-parallel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

static constexpr int CCS_K = 8;
static constexpr int CCS_DMIN = 30;
static constexpr int CCS_MIN_SUPPORT = 12;
static constexpr int CCS_SMOOTH = 3;
static constexpr int CCS_MAX_CUTS = 64;
static constexpr int CCS_MIN_TAIL = 20;
static constexpr int POA_MAXP = 12;                  // in-edges a node can hold (implementation limit)
static constexpr int POA_MAXA = 4;                   // other members of an aligned set (5 letter codes)
static constexpr int POA_MAX_COPY = 2800;            // longest sequence: cells are int16 (match score <= 11)

// Phase boundary inside one wave that exchanges data between lanes through HBM: complete the stores, then drop the
// CU's L1 so that no line read before the stores can be served stale (buffer_inv sc1; a few microseconds, used a
// handful of times per copy, never per row).
__device__ __forceinline__ void phase_sync() {
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// inclusive prefix maximum over the 64 lanes: row_shr 1,2,4,8 inside each 16-lane row, then row_bcast 15 and 31 carry
// the row totals.  The DPP modifier sits on the max itself (v_max_i32_dpp: dst = max(dpp(src), src)); a lane without a
// valid DPP source is simply not written, so no fill value and no separate v_mov_dpp are needed -- 6 VALU instructions
// (plus the 2 wait states a DPP read needs after a VALU write) instead of 24.
__device__ __forceinline__ int wave_prefix_max(int v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------------------
// LDS layout for a batch whose longest read has lcap bases (lcap a multiple of 64):
//   head[K2_BUCKETS] int32 | cnt[lcap/2+2] int32 | sm[lcap/2+2] int32 | code[lcap] uint16 | next[lcap] int16
static constexpr int K2_BUCKETS = 2048;
static constexpr int K2_LDS_MAX = 16000;       // longest read scanned out of LDS (8 bytes per base + 8 KiB of 160 KiB)
__host__ __device__ inline size_t k2_lds_bytes(int lcap) { return 4 * (size_t)K2_BUCKETS + 8 * ((size_t)lcap / 2 + 2) + 4 * (size_t)lcap; }

// The scan of one read.  NextT = int16_t with every array in LDS (reads up to K2_LDS_MAX bases), int32_t with cnt/sm/
// code/next in an HBM workspace (longer reads: rare, so the slower memory does not matter; no length limit).
template <typename NextT>
__device__ void ccs_scan_read(const CcsParams& p, const int rd, const int lane, int32_t* head, int32_t* cnt, int32_t* sm, uint16_t* code, NextT* next)
{
    const int64_t off = p.read_off[rd];
    const int L = (int)(p.read_off[rd + 1] - off);
    const int8_t* seq = p.reads + off;
    CcsScan out;
    out.period = 0; out.ncuts = 0; out.support = 0;
    // lanes exchange data through the arrays: LDS needs a barrier, the HBM workspace also a cache invalidate
    auto sync = [&]() { if constexpr (sizeof(NextT) == 4) phase_sync(); else __syncthreads(); };
    if (L < 2 * CCS_DMIN) { if (lane == 0) p.scan[rd] = out; return; }

    // The specification counts, per offset d, the positions i with equal valid k-mers at i and i+d: that is one count per
    // PAIR of equal k-mers.  A read of L bases has O(L * copies) such pairs, not O(L^2/4): the positions are chained per
    // hash bucket and every pair is visited once, instead of comparing all (i, d).
    const int dmax = L / 2;
    for (int i = lane; i < K2_BUCKETS; i += 64) head[i] = -1;
    for (int d = lane; d <= dmax + 1; d += 64) cnt[d] = 0;
    sync();
    for (int i = lane; i < L; i += 64) {
        int32_t c = 0, ok = i + CCS_K <= L;
        if (ok)
            for (int t = 0; t < CCS_K; ++t) {
                const int b = seq[i + t];
                if (b < 0 || b > 3) ok = 0;
                c = (c << 2) | (b & 3);
            }
        code[i] = (uint16_t)c;
        next[i] = ok ? (NextT)atomicExch(&head[(c ^ (c >> 5)) & (K2_BUCKETS - 1)], i) : (NextT)-2;
    }
    sync();
    for (int i = lane; i < L; i += 64) {
        int j = next[i];
        if (j == -2) continue;
        const int ci = code[i];
        while (j >= 0) {                            // positions of this bucket inserted before i: each pair once
            if (code[j] == ci) {
                const int d = i > j ? i - j : j - i;
                if (d >= CCS_DMIN && d <= dmax) atomicAdd(&cnt[d], 1);
            }
            j = next[j];
        }
    }
    sync();
    int best = -1, bestd = 0x7fffffff;
    for (int d = CCS_DMIN + lane; d <= dmax; d += 64) {
        int s = 0;
        const int lo = d - CCS_SMOOTH < CCS_DMIN ? CCS_DMIN : d - CCS_SMOOTH, hi = d + CCS_SMOOTH > dmax ? dmax : d + CCS_SMOOTH;
        for (int e = lo; e <= hi; ++e) s += cnt[e];
        sm[d] = s;
        if (s > best) { best = s; bestd = d; }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int b2 = __shfl_xor(best, d), d2 = __shfl_xor(bestd, d);
        if (b2 > best || (b2 == best && d2 < bestd)) { best = b2; bestd = d2; }
    }
    sync();
    if (best < CCS_MIN_SUPPORT) { if (lane == 0) p.scan[rd] = out; return; }
    int p0 = bestd;
    for (int q = 2; q <= 8; ++q) {
        const int c = (bestd + q / 2) / q;
        if (c < CCS_DMIN) break;
        int es = -1, eb = -1;
        const int lo = c - 3 < CCS_DMIN ? CCS_DMIN : c - 3, hi = c + 3 > dmax ? dmax : c + 3;
        for (int e = lo; e <= hi; ++e) if (sm[e] > es) { es = sm[e]; eb = e; }
        if (eb >= 0 && 2 * es >= best) p0 = eb;
    }
    // copy boundaries: per cut, the offsets delta in [p0-tol, p0+tol] are scored by the matches (i, i+delta) with i in
    // the window [b, b+W).  Same pairs again: the window's positions walk their bucket chains (from the head: partners on
    // both sides are in it) into a histogram over delta (sm is free by now), the lanes then pick by the specification's
    // order -- most matches, then closest to the previous step, then smallest delta.
    const int tol = p0 / 8 > 4 ? p0 / 8 : 4, W = p0 < 96 ? p0 : 96;
    const int dlo = p0 - tol, nd = 2 * tol + 1;
    int32_t* dh = sm;
    sync();
    int b = 0, prev = p0, n = 0;
    while (n < CCS_MAX_CUTS) {
        for (int t = lane; t < nd; t += 64) dh[t] = 0;
        sync();
        for (int i = b + lane; i < b + W && i < L; i += 64) {
            if (next[i] == -2) continue;
            const int ci = code[i];
            int j = head[(ci ^ (ci >> 5)) & (K2_BUCKETS - 1)];
            while (j >= 0) {
                const int delta = j - i;
                if (delta >= dlo && delta < dlo + nd && code[j] == ci) atomicAdd(&dh[delta - dlo], 1);
                j = next[j];
            }
        }
        sync();
        int bs = -1, bdel = 0, bdist = 0x7fffffff;
        for (int t = lane; t < nd; t += 64) {
            const int delta = dlo + t;
            if (delta >= 1 && b + delta <= L) {
                const int sc = dh[t];
                const int dist = delta > prev ? delta - prev : prev - delta;
                if (sc > bs || (sc == bs && dist < bdist)) { bs = sc; bdel = delta; bdist = dist; }   // ascending delta within the lane
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int s2 = __shfl_xor(bs, d), e2 = __shfl_xor(bdel, d), t2 = __shfl_xor(bdist, d);
            if (s2 > bs || (s2 == bs && (t2 < bdist || (t2 == bdist && e2 < bdel)))) { bs = s2; bdel = e2; bdist = t2; }
        }
        sync();
        if (bs < 0) break;
        b += bdel;
        prev = bdel;
        if (lane == 0) out.cuts[n] = b;
        ++n;
    }
    if (n >= 2) { out.period = p0; out.ncuts = n; out.support = best; }
    if (lane == 0) p.scan[rd] = out;
}

__global__ void __launch_bounds__(64) ccs_scan_kernel(const CcsParams p)
{
    extern __shared__ __attribute__((aligned(16))) int32_t k2_lds[];
    const int lane = threadIdx.x & 63;
    const int rd = blockIdx.x;
    if ((int)(p.read_off[rd + 1] - p.read_off[rd]) > p.k2_lds_max) return;        // ccs_scan_long_kernel takes it
    int32_t* head = k2_lds;                         // bucket -> last inserted position, -1 = empty
    int32_t* cnt = head + K2_BUCKETS;               // matches per offset, [lcap/2 + 2]
    int32_t* sm = cnt + p.lcap / 2 + 2;             // smoothed counts; later the per-cut histogram of offsets
    uint16_t* code = (uint16_t*)(sm + p.lcap / 2 + 2);
    int16_t* next = (int16_t*)(code + p.lcap);      // chain of the positions of a bucket; -1 = end (invalid k-mers are in no chain)
    ccs_scan_read<int16_t>(p, rd, lane, head, cnt, sm, code, next);
}

__global__ void __launch_bounds__(64) ccs_scan_long_kernel(const CcsParams p)
{
    __shared__ int32_t head[K2_BUCKETS];
    const int lane = threadIdx.x & 63;
    const int rd = p.long_idx[blockIdx.x];
    uint8_t* ws = p.k2_ws + (size_t)blockIdx.x * p.k2_slot;       // cnt | sm | next (int32) | code (uint16), sized for the longest read
    const size_t half = (size_t)p.k2_lmax / 2 + 2;
    int32_t* cnt = (int32_t*)ws;
    int32_t* sm = cnt + half;
    int32_t* next = sm + half;
    uint16_t* code = (uint16_t*)(next + p.k2_lmax);
    ccs_scan_read<int32_t>(p, rd, lane, head, cnt, sm, code, next);
}

// ------------------------------------------------------------------------------------------------------------
// K3 -- partial-order alignment (spoa's recurrences and back-track, oracle/poa_oracle.c) and heaviest-bundle consensus
// ------------------------------------------------------------------------------------------------------------
// The row-at-a-time formulation computed here is stated and proven equal to the oracle's five-matrix statement on the
// CPU in tools/poa_model.py (tests/test_poa_model.py).  Per cell the kernel leaves ONE byte from which spoa's
// value-comparing back-track is replayed (rows with several in-edges leave 16 more bits: the in-edge slots).
// One byte per cell.  Bits 0-5: the move spoa's back-track takes out of the cell, as 62 - (its place in spoa's checking
// order): 63 the cell is zero (local mode: stop); 62-s diagonal through in-edge s; 50 - 3s - {0, 1, 2} vertical through
// in-edge s by F+e (the run goes on upwards), H+g, O+c (goes on); 14 / 13 / 12 horizontal by E+e (goes on to the left),
// H+g, Q+c (goes on).  Bit 6 (hx): the E or Q chain of this column extends the previous column's.  Bit 7 (vstop): an
// upward run ends with the step out of this cell.  Rows with several in-edges keep, in a second plane, the in-edge an
// upward run leaves the cell through.
static constexpr int CODE_ZERO = 63, CODE_DIAG = 62, CODE_VERT = 50, CODE_HORZ = 14, B_HX = 64, B_VSTOP = 128;
static constexpr int POA_NEG = -30000;               // "minus infinity" of a stored (int16) cell
static constexpr int POA_MAX_ROWS = 65000;           // ranks travel in 16 bits

struct PoaWs {            // views into one wave's workspace slot
    int8_t* base; int8_t* np; int32_t* pred; int32_t* pw; int32_t* aligned; int32_t* cov; int32_t* nout; int32_t* order; int32_t* rank;
    int32_t* pn; int32_t* pj; long long* key; int32_t* bnd;                  // per base of the sequence being added
    uint2* ri; uint32_t* rx; uint32_t* tab; int32_t* score; int32_t* bp; short* col0;     // rank space
    short* carry; int cpitch;
    uint8_t* dp; size_t dp_bytes;                                            // the rest of the slot: DP planes of the current sequence
    uint8_t* dirA; uint8_t* dirB; short* keepH; uint8_t* keepD;             // set per sequence (poa_add)
};

// row pitch (elements) of the DP planes for a sequence of m bases: column j sits at element j+7,
// so the 2..8 columns a lane owns start at an aligned element, and the pitch is a multiple of 16
__host__ __device__ inline int poa_pitch(int m) { return (m + 8 + 15) & ~15; }

__host__ __device__ inline size_t poa_fixed_bytes(int ncap, int mcap, int* cpitch_out)
{
    size_t o = 0;
    auto add = [&](size_t bytes) { o = (o + bytes + 15) & ~(size_t)15; };
    add(sizeof(int32_t) * (size_t)ncap * POA_MAXP); add(sizeof(int32_t) * (size_t)ncap * POA_MAXP);        // pred, pw
    add(sizeof(int32_t) * (size_t)ncap * POA_MAXA);                                                        // aligned
    add(sizeof(int32_t) * ncap); add(sizeof(int32_t) * ncap); add(sizeof(int32_t) * ncap); add(sizeof(int32_t) * ncap);   // cov nout order rank
    add(sizeof(int32_t) * (size_t)(ncap + mcap + 2)); add(sizeof(int32_t) * (size_t)(ncap + mcap + 2));    // pn pj (pn doubles as the consensus path)
    add(sizeof(long long) * (size_t)(mcap + 2)); add(sizeof(int32_t) * (size_t)(mcap + 2));                // key bnd
    add(sizeof(uint2) * (size_t)(ncap + 2)); add(sizeof(uint32_t) * (size_t)(ncap + 2)); add(sizeof(uint32_t) * 3 * (size_t)(ncap + 2));   // ri rx tab
    add(sizeof(int32_t) * (size_t)(ncap + 2)); add(sizeof(int32_t) * (size_t)(ncap + 2)); add(sizeof(short) * (size_t)(ncap + 2));        // score bp col0
    const int cp = (ncap + 2 + 7) & ~7;
    if (cpitch_out) *cpitch_out = cp;
    add(sizeof(short) * 6 * (size_t)cp);                                                                    // carries: 2 x (H, E, Q)
    add(ncap); add(ncap);                                                                                  // base np
    return o;
}
// DP planes of one sequence against N rows: byte plane, slot plane (one byte) of nm rows, kept rows (H int16 + vertical states int8)
__host__ __device__ inline size_t poa_dp_bytes(int N, int m, int nm, int nk)
{
    const size_t gp = (size_t)poa_pitch(m);
    return (((size_t)(N + 1) * gp + 15) & ~(size_t)15) + (((size_t)nm * gp + 15) & ~(size_t)15) + (size_t)nk * gp * 2 + (size_t)nk * gp + 64;
}
__host__ __device__ inline size_t poa_slot_bytes(int ncap, int mcap)       // worst case: every row has several in-edges and is kept
{
    return poa_fixed_bytes(ncap, mcap, nullptr) + poa_dp_bytes(ncap, mcap - 1, ncap, ncap) + 64;
}

__device__ PoaWs carve(uint8_t* slot, size_t slot_bytes, int ncap, int mcap)
{
    PoaWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { uint8_t* q = slot + o; o = (o + bytes + 15) & ~(size_t)15; return q; };
    w.pred = (int32_t*)take(sizeof(int32_t) * (size_t)ncap * POA_MAXP);
    w.pw = (int32_t*)take(sizeof(int32_t) * (size_t)ncap * POA_MAXP);
    w.aligned = (int32_t*)take(sizeof(int32_t) * (size_t)ncap * POA_MAXA);
    w.cov = (int32_t*)take(sizeof(int32_t) * ncap);
    w.nout = (int32_t*)take(sizeof(int32_t) * ncap);
    w.order = (int32_t*)take(sizeof(int32_t) * ncap);
    w.rank = (int32_t*)take(sizeof(int32_t) * ncap);
    w.pn = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + mcap + 2));
    w.pj = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + mcap + 2));
    w.key = (long long*)take(sizeof(long long) * (size_t)(mcap + 2));
    w.bnd = (int32_t*)take(sizeof(int32_t) * (size_t)(mcap + 2));
    w.ri = (uint2*)take(sizeof(uint2) * (size_t)(ncap + 2));
    w.rx = (uint32_t*)take(sizeof(uint32_t) * (size_t)(ncap + 2));
    w.tab = (uint32_t*)take(sizeof(uint32_t) * 3 * (size_t)(ncap + 2));
    w.score = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + 2));
    w.bp = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + 2));
    w.col0 = (short*)take(sizeof(short) * (size_t)(ncap + 2));
    w.cpitch = (ncap + 2 + 7) & ~7;
    w.carry = (short*)take(sizeof(short) * 6 * (size_t)w.cpitch);
    w.base = (int8_t*)take(ncap);
    w.np = (int8_t*)take(ncap);
    w.dp = slot + o;
    w.dp_bytes = slot_bytes > o ? slot_bytes - o : 0;
    w.dirA = nullptr; w.dirB = nullptr; w.keepH = nullptr; w.keepD = nullptr;
    return w;
}

extern __shared__ __attribute__((aligned(16))) uint32_t poa_lds[];     // K3's dynamic LDS block
static constexpr int POA_LDS_BYTES = 9216;           // ring of recent rows (DP) / keys (re-rank) / score per rank (heaviest bundle)
static constexpr int POA_RERANK_LDS_KEYS = POA_LDS_BYTES / 8;
static constexpr int POA_LDS_SCORES = POA_LDS_BYTES / 4 - 1;   // most rows whose scores fit the LDS block

#ifdef CLH_DEBUG_POA
#define TSTAMP(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define TSTAMP(k) do {} while (0)
#endif

// columns per lane, LDS row pitch and ring depth for a sequence of m bases
#ifndef POA_MAXC
#define POA_MAXC 4          // columns per lane: more of them spill registers (measured: 8 -> 50.8 ms, 4 -> 31.3 ms per 100 k reads)
#endif
#ifndef POA_WAVES
#define POA_WAVES 4
#endif
__device__ __forceinline__ int poa_cols(int m) { const int c = (m + 63) >> 6; return c < 2 ? 2 : (c > POA_MAXC ? POA_MAXC : c); }
__device__ __forceinline__ int poa_ring_pitch(int m) { const int W = 64 * poa_cols(m); return poa_pitch(m < W ? m : W); }   // LDS row pitch: the widest pass
__device__ __forceinline__ int poa_ring(int m) {
    const int lp = poa_ring_pitch(m);
    int ring = 16;                                   // power of two, so slot = rank & (ring-1); a ring row is 3 bytes per element
    while (ring * lp * 3 > POA_LDS_BYTES) ring >>= 1;
    return ring;
}

__device__ __forceinline__ int dpp_shr1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1; lane 0 keeps `fill`

// exclusive prefix maximum over all cells to the left: in-lane running maximum, one cross-lane scan, the value entering
// the pass on the left (`left`, lane 0's predecessor).  a[] in, pe[] out (pe[k] = max of everything left of cell k).
template <int C>
__device__ __forceinline__ void scan_left(const int (&a)[C], int left, int (&pe)[C])
{
    constexpr int NEGB = -(1 << 30);
    int run[C];
    run[0] = NEGB;
#pragma unroll
    for (int k = 1; k < C; ++k) run[k] = run[k - 1] > a[k - 1] ? run[k - 1] : a[k - 1];
    const int tot = run[C - 1] > a[C - 1] ? run[C - 1] : a[C - 1];
    const int incl = wave_prefix_max(tot);
    int excl = dpp_shr1(NEGB, incl);
    excl = left > excl ? left : excl;
#pragma unroll
    for (int k = 0; k < C; ++k) pe[k] = run[k] > excl ? run[k] : excl;
}

// DP rows of one sequence against the graph.  Lane l owns C adjacent columns (C = 2..8: ceil(length / 64)), so a row of up
// to W = 64*C columns is ONE step; longer sequences are swept in passes of W columns (passes outer, rows inner), the
// cell that leaves a row on the right handed to the next pass through per-row carries (H, E, Q) in HBM.  Graph rows
// (w.ri, w.rx) and carries are streamed 64 rows at a time into one register per lane and read with v_readlane; LDS holds
// the ring of the last RING rows (H int16, vertical states one byte) for near sources that are not the row before -- that
// one is forwarded from registers; far sources come from "kept" rows in HBM.
// One pass: the columns colbase+1 .. colbase+64*C (the last pass of a sequence may be narrower and then takes fewer
// columns per lane: a row step costs a fixed part plus a part per column).  `pass` numbers the passes of the sequence
// (carry buffers alternate), bs/br/bc carry the end cell across the passes.
template <int C>
__device__ void dp_pass(const PoaWs& w, const PoaScores S, int N, int m, const int8_t* seq, int lane, const int pass, const int colbase,
                        const bool more, const int RING, int& bs, int& br, int& bc)
{
    constexpr int NEGB = -(1 << 30);
    constexpr int W = 64 * C;
    const int gp = poa_pitch(m);
    const int lp = poa_ring_pitch(m);
    const int rmask = RING - 1;
    const bool sw = S.algorithm == 0, nw = S.algorithm == 1;
    const int g = S.g, e = S.e, q = S.q, c = S.c, sm = S.m, sn = S.n;
    short* ringH = (short*)poa_lds;
    uint8_t* ringD = (uint8_t*)(ringH + RING * lp);
    const int lm = (m - 1 - colbase) / C, km = (m - 1 - colbase) % C;      // where column m lives (last pass)
    {
        const int col0 = colbase + C * lane;                     // cell k is column col0+k+1, element col0+k+8 of an HBM row
        const bool last = !more;
        int sb[C];                                               // this lane's bases (100: beyond the sequence)
#pragma unroll
        for (int k = 0; k < C; ++k) { const int j = col0 + k + 1; sb[k] = j <= m ? (int)seq[j - 1] : 100; }
        const int jl0 = C * lane + 1;                            // local column of cell 0; cell k: (jl0 + k) * e in the frame of the first piece
        auto row0_h = [&](int j) -> int {                        // H[0][j]: row 0 (no node) as a source
            const int l1 = g + (j - 1) * e, l2 = q + (j - 1) * c;
            return (sw || j == 0) ? 0 : (l1 > l2 ? l1 : l2);
        };
        const short* cprev = w.carry + (size_t)(pass & 1) * 3 * w.cpitch;
        short* cnext = w.carry + (size_t)((pass + 1) & 1) * 3 * w.cpitch;
        // streams: graph row, plane indices, the three values entering the row on the left
        uint2 blk = make_uint2(0, 0); uint32_t xblk = 0; int cH = 0, cE = POA_NEG, cQ = POA_NEG;
        auto fetch = [&](int rr, uint2& b, uint32_t& x, int& h, int& ee, int& qq) {
            b = make_uint2(0, 0); x = 0; h = 0; ee = POA_NEG; qq = POA_NEG;
            if (rr <= N) {
                b = w.ri[rr]; x = w.rx[rr];
                if (pass > 0) { h = (int)cprev[rr]; ee = (int)cprev[w.cpitch + rr]; qq = (int)cprev[2 * w.cpitch + rr]; }
                else if (nw) h = (int)w.col0[rr];
            }
        };
        fetch(1 + lane, blk, xblk, cH, cE, cQ);
        int px[C], pf[C], po[C], pcin = 0;                       // the previous row, still in registers
#pragma unroll
        for (int k = 0; k < C; ++k) { px[k] = 0; pf[k] = 0; po[k] = 0; }
        for (int rb = 1; rb <= N; rb += 64) {
            uint2 nblk; uint32_t nxblk; int nH, nE, nQ;
            fetch(rb + 64 + lane, nblk, nxblk, nH, nE, nQ);
            const int cnt = N - rb + 1 < 64 ? N - rb + 1 : 64;
            int cobH = 0, cobE = 0, cobQ = 0;
            for (int i = 0; i < cnt; ++i) {
                const int r = rb + i;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)blk.x, i), d1 = (uint32_t)__builtin_amdgcn_readlane((int)blk.y, i);
                const uint32_t rxv = (uint32_t)__builtin_amdgcn_readlane((int)xblk, i);
                const int cinH = __builtin_amdgcn_readlane(cH, i), cinE = __builtin_amdgcn_readlane(cE, i), cinQ = __builtin_amdgcn_readlane(cQ, i);
                const int vb = (int)(d0 & 0xff), np = (int)((d0 >> 8) & 0xf);
                const bool sink = (d0 & 0x1000u) != 0;
                const bool tolds = (d0 & 0x4000u) != 0;          // read later from the LDS ring
                const bool keep = (d0 & 0x8000u) != 0;           // read later from HBM by a far successor
                const int p0 = (int)(d0 >> 16), p1 = (int)(d1 & 0xffff), p2 = (int)(d1 >> 16);
                const int mi = (int)(rxv & 0xffff), ki = (int)(rxv >> 16);
                int ss[C];
#pragma unroll
                for (int k = 0; k < C; ++k) ss[k] = sb[k] == vb ? sm : sn;
                // one source row: H at the lane's columns and the one before, Fs = F + e - g, Os = O + c - q
                auto source = [&](int qr, int (&h)[C], int& hprev, int (&fs)[C], int (&os)[C]) {
                    if (qr == 0) {
#pragma unroll
                        for (int k = 0; k < C; ++k) { h[k] = row0_h(col0 + k + 1); fs[k] = h[k] - 1; os[k] = h[k] - 1; }
                        hprev = row0_h(col0);
                    } else if (qr == r - 1) {
#pragma unroll
                        for (int k = 0; k < C; ++k) { h[k] = px[k]; fs[k] = pf[k]; os[k] = po[k]; }
                        hprev = dpp_shr1(0, px[C - 1]);
                        hprev = lane == 0 ? pcin : hprev;
                    } else if (r - qr < RING) {
                        const short* sh = ringH + (qr & rmask) * lp + C * lane + 8;
                        const uint8_t* sd = ringD + (qr & rmask) * lp + C * lane + 8;
                        hprev = sh[-1];
#pragma unroll
                        for (int k = 0; k < C; ++k) { h[k] = sh[k]; const int dd = sd[k]; fs[k] = h[k] + (dd & 7) - 1; os[k] = h[k] + (dd >> 3) - 1; }
                    } else {
                        const int kq = (int)(__builtin_amdgcn_readfirstlane((int)w.rx[qr]) >> 16) & 0xffff;
                        const short* sh = w.keepH + (size_t)kq * gp + col0 + 8;
                        const uint8_t* sd = w.keepD + (size_t)kq * gp + col0 + 8;
                        int dd[C];
                        asm volatile("global_load_sshort %0, %1, off" : "=v"(hprev) : "v"(sh - 1) : "memory");
#pragma unroll
                        for (int k = 0; k < C; ++k) asm volatile("global_load_sshort %0, %1, off" : "=v"(h[k]) : "v"(sh + k) : "memory");
#pragma unroll
                        for (int k = 0; k < C; ++k) asm volatile("global_load_ubyte %0, %1, off" : "=v"(dd[k]) : "v"(sd + k) : "memory");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int k = 0; k < C; ++k) { asm volatile("" : "+v"(h[k]), "+v"(dd[k])); fs[k] = h[k] + (dd[k] & 7) - 1; os[k] = h[k] + (dd[k] >> 3) - 1; }
                        asm volatile("" : "+v"(hprev));
                    }
                };
                // Every candidate of a cell travels as (value << 6) | code, code = 62 - (its place in spoa's checking order): one
                // signed max takes the larger value and, between equal values, the move spoa's back-track would find first;
                // the winner's low six bits ARE the cell's back-track code (csrc comment at CODE_*).
                int PM[C], fsn[C], osn[C], xb[C];                // packed best of the non-horizontal moves; vertical states; bit 7 (vstop) and, rows with several in-edges, the slot of the upward run
                if (np <= 1) {
                    int h[C], hprev, fs[C], os[C];
                    source(np == 0 ? 0 : p0, h, hprev, fs, os);
                    const int kFE = (g << 6) + CODE_VERT, kFO = (g << 6) + CODE_VERT - 1, kOE = (q << 6) + CODE_VERT - 2;
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int fe = fs[k], fo = h[k], oe = os[k];
                        const int p1 = ((k == 0 ? hprev : h[k - 1]) << 6) + ((ss[k] << 6) + CODE_DIAG);
                        const int p2 = (fe << 6) + kFE, p3 = (fo << 6) + kFO, p4 = (oe << 6) + kOE;
                        int pm = p1 > p2 ? p1 : p2;
                        pm = p3 > pm ? p3 : pm;
                        pm = p4 > pm ? p4 : pm;
                        if (sw) pm = pm > CODE_ZERO ? pm : CODE_ZERO;
                        PM[k] = pm;
                        const int mx = fe > fo ? fe : fo, mo = oe > fo ? oe : fo;
                        fsn[k] = e + mx; osn[k] = c + mo;
                        xb[k] = fo >= fe ? 128 : 0;
                    }
                } else {
                    int MX[C], MO[C], XF[C], XO[C];
#pragma unroll
                    for (int k = 0; k < C; ++k) { PM[k] = sw ? CODE_ZERO : NEGB; MX[k] = NEGB; MO[k] = NEGB; XF[k] = NEGB; XO[k] = NEGB; }
                    auto add_source = [&](int slot, int qr) {
                        int h[C], hprev, fs[C], os[C];
                        source(qr, h, hprev, fs, os);
                        const int kFE = (g << 6) + CODE_VERT - 3 * slot, kFO = kFE - 1, kOE = (q << 6) + CODE_VERT - 3 * slot - 2;
                        const int xo = 2 * (POA_MAXP - slot) + 1, xe = xo - 1;       // upward run: open before extend, earlier in-edge first
#pragma unroll
                        for (int k = 0; k < C; ++k) {
                            const int fe = fs[k], fo = h[k], oe = os[k];
                            const int p1 = ((k == 0 ? hprev : h[k - 1]) << 6) + ((ss[k] << 6) + CODE_DIAG - slot);
                            const int p2 = (fe << 6) + kFE, p3 = (fo << 6) + kFO, p4 = (oe << 6) + kOE;
                            int pm = PM[k] > p1 ? PM[k] : p1;
                            pm = p2 > pm ? p2 : pm; pm = p3 > pm ? p3 : pm; pm = p4 > pm ? p4 : pm;
                            PM[k] = pm;
                            int t = fe > fo ? fe : fo; MX[k] = t > MX[k] ? t : MX[k];
                            t = oe > fo ? oe : fo; MO[k] = t > MO[k] ? t : MO[k];
                            const int q1 = (fo << 6) + xo, q2 = (fe << 6) + xe, q3 = (oe << 6) + xe;
                            t = q1 > q2 ? q1 : q2; XF[k] = t > XF[k] ? t : XF[k];
                            t = q1 > q3 ? q1 : q3; XO[k] = t > XO[k] ? t : XO[k];
                        }
                    };
                    add_source(0, p0);
                    add_source(1, p1);
                    if (np > 2) add_source(2, p2);
                    if (np > 3) {                                // rare: in-edges beyond the third come from HBM
                        const int vnode = __builtin_amdgcn_readfirstlane(w.order[r - 1]);
                        for (int s2 = 3; s2 < np; ++s2) add_source(s2, __builtin_amdgcn_readfirstlane(w.rank[w.pred[vnode * POA_MAXP + s2]]));
                    }
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        fsn[k] = e + MX[k]; osn[k] = c + MO[k];
                        // the in-edge an upward run leaves this cell through: the first with F == H+g (last step), F == F+e, O == H+q
                        // (last step), O == O+c, in that order
                        const int cf = XF[k] & 63, co = XO[k] & 63;
                        const int sF = POA_MAXP - (cf >> 1), sO = POA_MAXP - (co >> 1);
                        const bool useF = sF <= sO;
                        const int slot = useF ? sF : sO, open = useF ? (cf & 1) : (co & 1);
                        xb[k] = (open ? 128 : 0) | (slot << 8);
                    }
                }
                // horizontal states: two prefix maxima in the gap-free frames of the two pieces
                int ehat[C], qhat[C], H[C];
                {
                    int a[C], pe[C];
                    const int leftE = cinH > cinE + e - g ? cinH : cinE + e - g;       // = E[first column of the pass] - g
#pragma unroll
                    for (int k = 0; k < C; ++k) a[k] = (PM[k] >> 6) - (jl0 + k) * e;
                    scan_left<C>(a, leftE, pe);
#pragma unroll
                    for (int k = 0; k < C; ++k) ehat[k] = pe[k] + (g - e) + (jl0 + k) * e;
                    const int leftQ = cinH > cinQ + c - q ? cinH : cinQ + c - q;
#pragma unroll
                    for (int k = 0; k < C; ++k) a[k] = (PM[k] >> 6) - (jl0 + k) * c;
                    scan_left<C>(a, leftQ, pe);
#pragma unroll
                    for (int k = 0; k < C; ++k) qhat[k] = pe[k] + (q - c) + (jl0 + k) * c;
                }
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const int m0 = PM[k] >> 6;
                    int hh = m0 > ehat[k] ? m0 : ehat[k];
                    H[k] = qhat[k] > hh ? qhat[k] : hh;
                }
                // the neighbour's value is fetched by every lane BEFORE the select: a DPP read executed under an exec mask that
                // excludes lane 0 would find its source lane disabled
                const int qsh = dpp_shr1(0, qhat[C - 1]);
                const int qleft = lane == 0 ? cinQ : qsh;
                int E[C];
#pragma unroll
                for (int k = 0; k < C; ++k) {                    // E as spoa holds it: a gap may open on a cell reached by the other piece
                    const int qp = (k == 0 ? qleft : qhat[k - 1]) + g;
                    E[k] = ehat[k] > qp ? ehat[k] : qp;
                }
                const int hsh = dpp_shr1(0, H[C - 1]), esh = dpp_shr1(0, E[C - 1]);
                const int hleft = lane == 0 ? cinH : hsh;
                const int eleft = lane == 0 ? cinE : esh;
                int out[C];
                {
                    const int kE = (e << 6) + CODE_HORZ, kG = (g << 6) + CODE_HORZ - 1, kC = (c << 6) + CODE_HORZ - 2, kQ = (q << 6) + CODE_HORZ - 3;
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int hl = k == 0 ? hleft : H[k - 1], ep = k == 0 ? eleft : E[k - 1], qp = k == 0 ? qleft : qhat[k - 1];
                        const int p5 = (ep << 6) + kE, p6 = (hl << 6) + kG, p7 = (qp << 6) + kC, p8 = (hl << 6) + kQ;
                        int ph = p5 > p6 ? p5 : p6;
                        ph = p7 > ph ? p7 : ph;
                        const int pf = PM[k] > ph ? PM[k] : ph;
                        // hx: E or Q of this column extends the previous column's (E[j-1]+e >= H[j-1]+g, Q[j-1]+c >= H[j-1]+q)
                        out[k] = (pf & 63) | (((p5 >= p6) | (p7 >= p8)) ? 64 : 0) | (xb[k] & 128);
                    }
                }
                // ---- what later rows and the back-track read --------------------------------------------------------------
#pragma unroll
                for (int k = 0; k < C; ++k) { px[k] = H[k]; pf[k] = fsn[k]; po[k] = osn[k]; }
                pcin = cinH;
                if (col0 + 1 <= m) {
                    uint8_t* dd = w.dirA + (size_t)r * gp + col0 + 8;
                    uint32_t w0 = 0, w1 = 0;
#pragma unroll
                    for (int k = 0; k < C && k < 4; ++k) w0 |= ((uint32_t)out[k] & 0xffu) << (8 * k);
#pragma unroll
                    for (int k = 4; k < C; ++k) w1 |= ((uint32_t)out[k] & 0xffu) << (8 * (k - 4));
                    if constexpr (C == 8) *(uint2*)dd = make_uint2(w0, w1);
                    else if constexpr (C >= 4) {
                        __builtin_memcpy(dd, &w0, 4);
                        if constexpr (C == 5) dd[4] = (uint8_t)w1;
                        if constexpr (C >= 6) { const uint16_t lo = (uint16_t)w1; __builtin_memcpy(dd + 4, &lo, 2); }
                        if constexpr (C == 7) dd[6] = (uint8_t)(w1 >> 16);
                    } else {
                        const uint16_t lo = (uint16_t)w0;
                        __builtin_memcpy(dd, &lo, 2);
                        if constexpr (C == 3) dd[2] = (uint8_t)(w0 >> 16);
                    }
                    if (np > 1) {
                        uint8_t* db = w.dirB + (size_t)mi * gp + col0 + 8;
#pragma unroll
                        for (int k = 0; k < C; ++k) db[k] = (uint8_t)(xb[k] >> 8);
                    }
                    if (tolds | keep) {
                        int dv[C];
#pragma unroll
                        for (int k = 0; k < C; ++k) {
                            int df = fsn[k] - H[k] + 1, dq = osn[k] - H[k] + 1;
                            df = df > 0 ? df : 0; dq = dq > 0 ? dq : 0;
                            dv[k] = df | (dq << 3);
                        }
                        if (tolds) {
                            short* dh = ringH + (r & rmask) * lp + C * lane + 8;
                            uint8_t* dl = ringD + (r & rmask) * lp + C * lane + 8;
#pragma unroll
                            for (int k = 0; k < C; ++k) { dh[k] = (short)H[k]; dl[k] = (uint8_t)dv[k]; }
                        }
                        if (keep) {
                            short* hd = w.keepH + (size_t)ki * gp + col0 + 8;
                            uint8_t* hb = w.keepD + (size_t)ki * gp + col0 + 8;
#pragma unroll
                            for (int k = 0; k < C; ++k) { hd[k] = (short)H[k]; hb[k] = (uint8_t)dv[k]; }
                        }
                    }
                }
                if (tolds && lane == 0) ringH[(r & rmask) * lp + 7] = (short)cinH;     // element of local column 0
                if (keep && pass == 0 && lane == 0) w.keepH[(size_t)ki * gp + 7] = (short)cinH;
                // ---- end cell: first strict maximum in (rank, column) order --------------------------------------------
                if (sw | (!nw & sink)) {
                    int rk = NEGB;
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int kv = col0 + k + 1 <= m ? (H[k] << 4) | (15 - k) : NEGB;
                        rk = kv > rk ? kv : rk;
                    }
                    const int v = rk >> 4;
                    if (v > bs || (pass > 0 && v == bs && r < br)) { bs = v; br = r; bc = col0 + 1 + 15 - (rk & 15); }
                } else if (nw & sink & last) {
                    int hm = H[0];
#pragma unroll
                    for (int k = 1; k < C; ++k) hm = km == k ? H[k] : hm;
                    if (lane == lm && hm > bs) { bs = hm; br = r; bc = m; }
                }
                if (more) {
                    const int rH = __builtin_amdgcn_readlane(H[C - 1], 63), rE = __builtin_amdgcn_readlane(E[C - 1], 63), rQ = __builtin_amdgcn_readlane(qhat[C - 1], 63);
                    cobH = lane == i ? rH : cobH; cobE = lane == i ? (rE < POA_NEG ? POA_NEG : rE) : cobE; cobQ = lane == i ? (rQ < POA_NEG ? POA_NEG : rQ) : cobQ;
                }
                asm volatile("" ::: "memory");   // one wave: LDS operations execute in order; only the compiler must not reorder
            }
            if (more && lane < cnt) { cnext[rb + lane] = (short)cobH; cnext[w.cpitch + rb + lane] = (short)cobE; cnext[2 * w.cpitch + rb + lane] = (short)cobQ; }
            blk = nblk; xblk = nxblk; cH = nH; cE = nE; cQ = nQ;
        }
        if (more) phase_sync();
    }
}

// DP rows of one sequence: passes of 256 columns (4 per lane); the last pass takes 2..4 columns per lane by its width
__device__ void dp_rows(const PoaWs& w, const PoaScores S, int N, int m, const int8_t* seq, int lane, int& bs_out, int& br_out, int& bc_out)
{
    constexpr int NEGB = -(1 << 30);
    constexpr int WMAX = 64 * POA_MAXC;
    const int RING = poa_ring(m);
    int bs = S.algorithm == 0 ? 0 : NEGB, br = 0, bc = 0;
    int pass = 0;
    for (int colbase = 0; colbase < m; colbase += WMAX, ++pass) {
        const int rem = m - colbase;
        const bool more = rem > WMAX;
        const int cw = more ? POA_MAXC : poa_cols(rem);
        switch (cw) {
            case 2: dp_pass<2>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
            case 3: dp_pass<3>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
#if POA_MAXC > 4
            case 4: dp_pass<4>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
            case 5: dp_pass<5>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
            case 6: dp_pass<6>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
            case 7: dp_pass<7>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
            default: dp_pass<8>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
#else
            default: dp_pass<4>(w, S, N, m, seq, lane, pass, colbase, more, RING, bs, br, bc); break;
#endif
        }
    }
    // best over the lanes: value descending, rank ascending, column ascending
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int s2 = __shfl_xor(bs, d), r2 = __shfl_xor(br, d), c2 = __shfl_xor(bc, d);
        if (s2 > bs || (s2 == bs && (r2 < br || (r2 == br && c2 < bc)))) { bs = s2; br = r2; bc = c2; }
    }
    bs_out = bs; br_out = br; bc_out = bc;
}

// merge the nodes created by the last sequence ([n_old, n_new), keys ascending in creation order) into the rank order.
// key of a new node = ((s * 4 + class) << 24) + position, s = the old rank it goes in front of; an old node of rank r
// compares as (r * 4 + 2) << 24.  returns 0, or -1 if the order would violate an edge (cannot happen: aligned sets stay
// contiguous; checked because everything downstream relies on it)
__device__ int poa_rerank(const PoaWs& w, int n_old, int n_new, int lane)
{
    const int added = n_new - n_old;
    long long* lkeys = (long long*)poa_lds;
    const bool in_lds = added <= POA_RERANK_LDS_KEYS;
    if (in_lds) {
        for (int i = lane; i < added; i += 64) lkeys[i] = w.key[i];
        __syncthreads();
    }
    for (int r = 1 + lane; r <= n_old; r += 64) {
        const int v = w.order[r - 1];
        const long long k = ((long long)r * 4 + 2) << 24;
        int lo = 0, hi = added;                      // lower_bound over new keys
        if (in_lds) while (lo < hi) { const int mid = (lo + hi) >> 1; if (lkeys[mid] < k) lo = mid + 1; else hi = mid; }
        else while (lo < hi) { const int mid = (lo + hi) >> 1; if (w.key[mid] < k) lo = mid + 1; else hi = mid; }
        w.rank[v] = r + lo;
    }
    for (int i = lane; i < added; i += 64) w.rank[n_old + i] = i + (int)(w.key[i] >> 26);      // i new nodes + (s - 1) old nodes in front, 1-based
    phase_sync();
    int bad = 0;
    for (int v = lane; v < n_new; v += 64) {
        const int r = w.rank[v];
        w.order[r - 1] = v;
        const int np = w.np[v];
        for (int s2 = 0; s2 < np; ++s2) bad |= (w.rank[w.pred[v * POA_MAXP + s2]] >= r);
    }
    phase_sync();
    return __builtin_amdgcn_ballot_w64(bad != 0) ? -1 : 0;
}

// ranks of the first and the last member of the aligned set of node v
__device__ __forceinline__ void group_span(const PoaWs& w, int v, int& lo, int& hi)
{
    lo = hi = w.rank[v];
    for (int t = 0; t < POA_MAXA; ++t) {
        const int x = w.aligned[v * POA_MAXA + t];
        if (x < 0) break;
        const int rr = w.rank[x];
        lo = rr < lo ? rr : lo; hi = rr > hi ? rr : hi;
    }
}

// returns the new node count; -1 graph limits, -2 workspace.  *score_out = end-cell score.
// path_out (may be null): node of every base (for the MSA)
__device__ int poa_add(PoaWs& w, const PoaScores S, int N_, int ncap, const int8_t* seq, int m_, int lane, int* score_out, int32_t* path_out, unsigned long long* tacc)
{
    // wave-uniform by construction; say so, or every quantity derived from them lives in VGPRs behind exec-mask branches
    const int N = __builtin_amdgcn_readfirstlane(N_), m = __builtin_amdgcn_readfirstlane(m_);
    unsigned long long tlast = 0;
#ifdef CLH_DEBUG_POA
    tlast = __builtin_amdgcn_s_memtime();
#endif
    *score_out = 0;
    if (m == 0) return N;
    int bs = 0, br = 0, bc = 0;
    if (N > 0) {
        if (N > POA_MAX_ROWS || (S.algorithm == 1 && N > 25000)) return -1;
        // ---- graph rows in rank space (w.ri: base, in-degree, sink, ranks of the first three sources) ---------------------
        const int pitch = poa_pitch(m);
        const int RING = poa_ring(m);
        uint32_t* ri32 = (uint32_t*)w.ri;
#pragma unroll 4
        for (int r = 1 + lane; r <= N; r += 64) {
            const int v = w.order[r - 1];
            const int np = w.np[v];
            uint32_t pr[3] = {0, 0, 0};
            for (int s2 = 0; s2 < 3; ++s2) if (s2 < np) pr[s2] = (uint32_t)w.rank[w.pred[v * POA_MAXP + s2]];
            w.ri[r] = make_uint2((uint32_t)(w.base[v] & 0xff) | ((uint32_t)np << 8) | (w.nout[v] == 0 ? 0x1000u : 0u) | (pr[0] << 16), pr[1] | (pr[2] << 16));
        }
        phase_sync();
        // where will row r read source p from?  the row before it: registers; another recent row: the LDS ring (0x4000);
        // an older one: its kept row in HBM (0x8000)
        for (int r = 1 + lane; r <= N; r += 64) {
            const uint2 d = w.ri[r];
            const int np = (int)((d.x >> 8) & 0xf);
            const int p0 = (int)(d.x >> 16), p1 = (int)(d.y & 0xffff), p2 = (int)(d.y >> 16);
            auto mark = [&](int q) { if (q == 0) return; if (r - q >= RING) atomicOr(&ri32[q * 2], 0x8000u); else if (r - q >= 2) atomicOr(&ri32[q * 2], 0x4000u); };
            if (np > 0) mark(p0);
            if (np > 1) mark(p1);
            if (np > 2) mark(p2);
            if (np > 3) { const int v = w.order[r - 1]; for (int s2 = 3; s2 < np; ++s2) mark(w.rank[w.pred[v * POA_MAXP + s2]]); }
        }
        phase_sync();
        // plane indices: rows with several in-edges (slot plane) and kept rows, counted in rank order
        int nm = 0, nk = 0;
        for (int r0 = 1; r0 <= N; r0 += 64) {
            const int r = r0 + lane;
            const uint32_t d = r <= N ? ri32[r * 2] : 0u;
            const bool multi = ((d >> 8) & 0xf) > 1, kp = (d & 0x8000u) != 0;
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(multi), bk = __builtin_amdgcn_ballot_w64(kp);
            const unsigned long long below = ((unsigned long long)1 << lane) - 1;
            if (r <= N) w.rx[r] = (uint32_t)(nm + __builtin_popcountll(bm & below)) | ((uint32_t)(nk + __builtin_popcountll(bk & below)) << 16);
            nm += __builtin_popcountll(bm); nk += __builtin_popcountll(bk);
        }
        if (nm > 65535 || nk > 65535) return -1;
        if (poa_dp_bytes(N, m, nm, nk) > w.dp_bytes) return -2;
        {
            const size_t gp = (size_t)pitch;
            size_t o = ((size_t)(N + 1) * gp + 15) & ~(size_t)15;
            w.dirA = w.dp;
            w.dirB = w.dp + o; o += (((size_t)nm * gp + 15) & ~(size_t)15);
            w.keepH = (short*)(w.dp + o); o += (size_t)nk * gp * 2;
            w.keepD = w.dp + o;
        }
        if (S.algorithm == 1) {
            // global mode: H[i][0] = max(F, O)[i][0], F[i][0] = e + max over sources (a node without in-edges: g), O likewise.
            // A chain over the ranks, wave-uniform (every lane computes and stores the same values); not a hot path.
            for (int r = 1; r <= N; ++r) {
                const int v = w.order[r - 1];
                const int np = w.np[v];
                int f = np ? -(1 << 28) : S.g - S.e, o = np ? -(1 << 28) : S.q - S.c;
                for (int s2 = 0; s2 < np; ++s2) {
                    const int pr = w.rank[w.pred[v * POA_MAXP + s2]];
                    f = w.score[pr] > f ? w.score[pr] : f; o = w.bp[pr] > o ? w.bp[pr] : o;
                }
                f += S.e; o += S.c;
                w.score[r] = f; w.bp[r] = o;
                const int h = f > o ? f : o;
                w.col0[r] = (short)(h < POA_NEG ? POA_NEG : h);
                __syncthreads();
            }
        }
        phase_sync();
        dp_rows(w, S, N, m, seq, lane, bs, br, bc);
        phase_sync();
    }
    *score_out = bs;
    TSTAMP(0);
    // ---- back-track (sequential by nature, wave-uniform) ----------------------------------------------------------
    // The lanes hold a 32x32 patch of the byte plane (ranks r0..r0-31, columns j0..j0-31; 16 bytes per lane), of the slot
    // plane for the rows that have one, and the graph rows of those ranks, so the chain runs on v_readlane until it leaves
    // the patch.  pn[j] = rank aligned to base j (0: none), staged in one register per lane, stored 64 bases at a time.
    {
        int r = __builtin_amdgcn_readfirstlane(br), j = br > 0 ? __builtin_amdgcn_readfirstlane(bc) : 0;
        for (int t = j + lane; t < m; t += 64) w.pn[t] = 0;                     // bases behind the end cell
        const int gp = poa_pitch(m);
        int r0 = -64, j0 = -64;
        uint32_t pa0 = 0, pa1 = 0, pa2 = 0, pa3 = 0, ri = 0;     // lane l: row r0-(l>>1), columns j0-16*(l&1)-15 .. j0-16*(l&1)
        uint32_t pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0;
        int buf = 0, mode = 0;                                   // mode 1: inside an upward run, 2: inside a leftward run
        const bool sw = S.algorithm == 0;
        int guard = 2 * (N + m) + 64;                           // every step lowers r or j: a longer walk means corrupt planes
        while (j > 0 && r > 0) {
            if (--guard < 0) return -3;
            int a = r0 - r, b = j0 - j;
            if ((unsigned)a >= 32u || (unsigned)b >= 32u) {
                r0 = r; j0 = j;
                const int rr = r - (lane >> 1), jlo = j - 16 * (lane & 1) - 15;       // lowest column of this lane's 16
                pa0 = pa1 = pa2 = pa3 = 0; pb0 = pb1 = pb2 = pb3 = 0; ri = 0;
                if (rr >= 1) {
                    const uint2 gr = w.ri[rr];
                    ri = (lane & 1) ? gr.y : gr.x;
                    if (jlo + 15 >= 1) {
                        // bytes at columns jlo..jlo+15 (columns below 0 belong to the previous row: in bounds, never looked at)
                        const uint8_t* src = w.dirA + (size_t)rr * gp + 7 + jlo;
                        uint32_t t[4];
                        __builtin_memcpy(t, src, 16);
                        pa0 = t[0]; pa1 = t[1]; pa2 = t[2]; pa3 = t[3];
                        if (((gr.x >> 8) & 0xf) > 1) {
                            const uint8_t* sb2 = w.dirB + (size_t)(w.rx[rr] & 0xffff) * gp + 7 + jlo;
                            __builtin_memcpy(t, sb2, 16);
                            pb0 = t[0]; pb1 = t[1]; pb2 = t[2]; pb3 = t[3];
                        }
                    }
                }
                a = 0; b = 0;
            }
            // column j0-b sits in lane 2a+(b>>4) at byte 15-(b&15)
            const int idx = 15 - (b & 15), src_lane = a * 2 + (b >> 4);
            const uint32_t sela = (idx >> 2) == 0 ? pa0 : ((idx >> 2) == 1 ? pa1 : ((idx >> 2) == 2 ? pa2 : pa3));
            const int d = (__builtin_amdgcn_readlane((int)sela, src_lane) >> ((idx & 3) * 8)) & 0xff;
            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)ri, a * 2), g1 = (uint32_t)__builtin_amdgcn_readlane((int)ri, a * 2 + 1);
            const int np = (int)((g0 >> 8) & 0xf);
            auto pred_rank = [&](int s2) -> int {
                if (np == 0) return 0;
                if (s2 == 0) return (int)(g0 >> 16);
                if (s2 == 1) return (int)(g1 & 0xffff);
                if (s2 == 2) return (int)(g1 >> 16);
                const int v = w.order[r - 1];
                return __builtin_amdgcn_readfirstlane(w.rank[w.pred[v * POA_MAXP + s2]]);
            };
            if (mode == 1) {                                     // upward run: one more node without a base
                int sl = 0;
                if (np > 1) {
                    const uint32_t selb = (idx >> 2) == 0 ? pb0 : ((idx >> 2) == 1 ? pb1 : ((idx >> 2) == 2 ? pb2 : pb3));
                    sl = (__builtin_amdgcn_readlane((int)selb, src_lane) >> ((idx & 3) * 8)) & 0xff;
                }
                r = pred_rank(sl);
                if (d & B_VSTOP) mode = 0;
                continue;
            }
            if (mode == 2) {                                     // leftward run: one more base without a node
                --j;
                buf = lane == (j & 63) ? 0 : buf;
                if ((j & 63) == 0) { if (j + lane < m) w.pn[j + lane] = buf; buf = 0; }
                if (!(d & B_HX)) mode = 0;
                continue;
            }
            const int code = d & 63;
            if (code == CODE_ZERO) break;                        // local mode only: nothing else carries this code
            if (code > CODE_VERT) {                              // diagonal through in-edge CODE_DIAG - code
                --j;
                buf = lane == (j & 63) ? r : buf;
                if ((j & 63) == 0) { if (j + lane < m) w.pn[j + lane] = buf; buf = 0; }
                r = pred_rank(CODE_DIAG - code);
            } else if (code > CODE_HORZ) {                       // vertical: in-edge v / 3, by F+e, H+g, O+c (v % 3)
                const int v = CODE_VERT - code;
                const int slot = v / 3;
                r = pred_rank(slot);
                mode = (v - 3 * slot) != 1 ? 1 : 0;
            } else {                                             // horizontal by E+e, H+g, Q+c
                --j;
                buf = lane == (j & 63) ? 0 : buf;
                if ((j & 63) == 0) { if (j + lane < m) w.pn[j + lane] = buf; buf = 0; }
                mode = code != CODE_HORZ - 1 ? 2 : 0;
            }
        }
        int fill_to = j;
        if (j & 63) { fill_to = j & ~63; const int q = fill_to + lane; if (q < m && q >= 0) w.pn[q] = q < j ? 0 : buf; }
        for (int q = lane; q < fill_to; q += 64) w.pn[q] = 0;
    }
    phase_sync();
    TSTAMP(1);
    // ---- fuse the path into the graph (Graph::AddAlignment), data-parallel over the bases -------------------------------
    // Base i touches only its own aligned node's set and the in-edge list of the node it ends on, and the nodes of one
    // path are distinct, so the sequential rule is evaluated per lane; ids of new nodes are a prefix count in base order.
    int n = N, fail = 0;
    {
        // old rank in front of which a run of unaligned bases goes: first rank of the aligned set of the next aligned base
        int nextb = N + 1;
        for (int i0 = ((m - 1) / 64) * 64; i0 >= 0; i0 -= 64) {
            const int i = i0 + lane;
            const int rk = i < m ? w.pn[i] : 0;
            int lo = 0, hi = 0;
            if (rk > 0) group_span(w, w.order[rk - 1], lo, hi);
            const unsigned long long mb = __builtin_amdgcn_ballot_w64(rk > 0);
            const unsigned long long at_or_above = ~(((unsigned long long)1 << lane) - 1);
            const unsigned long long up = mb & at_or_above;
            const int src = up ? __builtin_ctzll(up) : 0;
            const int lo_src = __shfl(lo, src);
            if (i < m) w.bnd[i] = up ? lo_src : nextb;
            if (mb) nextb = __shfl(lo, __builtin_ctzll(mb));
        }
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < m;
            const int rk = act ? w.pn[i] : 0;
            const int v = rk > 0 ? w.order[rk - 1] : -1;
            const int b = act ? (int)seq[i] : 0;
            int use = -1;
            if (v >= 0) {
                if (w.base[v] == b) use = v;
                else for (int t = 0; t < POA_MAXA; ++t) { const int x = w.aligned[v * POA_MAXA + t]; if (x < 0) break; if (w.base[x] == b) { use = x; break; } }
            }
            const bool isnew = act && use < 0;
            const unsigned long long nb = __builtin_amdgcn_ballot_w64(isnew);
            const unsigned long long below = ((unsigned long long)1 << lane) - 1;
            if (isnew) use = n + __builtin_popcountll(nb & below);
            n += __builtin_popcountll(nb);
            if (isnew && use < ncap) {
                long long key;
                int al[POA_MAXA];
#pragma unroll
                for (int t = 0; t < POA_MAXA; ++t) al[t] = -1;
                if (v >= 0) {                                  // a new member of the aligned set of v: directly behind the set
                    int lo, hi;
                    group_span(w, v, lo, hi);
                    key = ((long long)(hi + 1) * 4 + 0) << 24;
                    int nm2 = 0;
                    for (int t = 0; t < POA_MAXA; ++t) {
                        const int x = w.aligned[v * POA_MAXA + t];
                        if (x < 0) break;
                        al[nm2++] = x;
                        int u = 0;
                        while (u < POA_MAXA && w.aligned[x * POA_MAXA + u] >= 0) ++u;
                        if (u < POA_MAXA) w.aligned[x * POA_MAXA + u] = use; else fail = 1;
                    }
                    if (nm2 < POA_MAXA) { al[nm2] = v; w.aligned[v * POA_MAXA + nm2] = use; } else fail = 1;
                } else key = (((long long)w.bnd[i] * 4 + 1) << 24) + i;
                w.base[use] = (int8_t)b; w.np[use] = 0; w.cov[use] = 0; w.nout[use] = 0; w.key[use - N] = key;
#pragma unroll
                for (int t = 0; t < POA_MAXA; ++t) w.aligned[use * POA_MAXA + t] = al[t];
            }
            if (act) { w.pj[i] = use; if (path_out) path_out[i] = use; }
        }
        if (n > ncap) return -1;
        phase_sync();
        for (int i = lane; i < m; i += 64) {
            const int x = w.pj[i];
            w.cov[x] += 1;
            if (i == 0) continue;
            const int u = w.pj[i - 1];
            const int cnt = w.np[x];
            int found = -1;
            for (int s2 = 0; s2 < cnt; ++s2) if (w.pred[x * POA_MAXP + s2] == u) { found = s2; break; }
            if (found >= 0) w.pw[x * POA_MAXP + found] += 2;
            else if (cnt >= POA_MAXP) fail = 1;
            else { w.pred[x * POA_MAXP + cnt] = u; w.pw[x * POA_MAXP + cnt] = 2; w.np[x] = (int8_t)(cnt + 1); w.nout[u] += 1; }
        }
        if (__builtin_amdgcn_ballot_w64(fail != 0)) return -1;
    }
    phase_sync();
    TSTAMP(2);
    if (N == 0) {
        for (int v = lane; v < n; v += 64) { w.order[v] = v; w.rank[v] = v + 1; }
        phase_sync();
    } else if (poa_rerank(w, N, n, lane) != 0) return -1;
    TSTAMP(3);
    return n;
}

// Heaviest bundle (Graph::TraverseHeaviestBundle + BranchCompletion).  The pass over the rows in rank order is a dependent
// chain, so it runs wave-uniformly -- on a rank-space image of the graph built in parallel (w.tab, 3 words a row:
// in-degree, up to 3 source ranks and their weights) and streamed 64 rows at a time into registers, with the scores in LDS
// (graphs above POA_LDS_SCORES rows: in HBM) and the previous row's score forwarded in a register.  Rows with more than 3
// in-edges or a weight above 255 fetch their lists from HBM (rare).  Back pointers leave through a lane buffer; the final
// chase reads them back 64 ranks at a time.  The consensus keeps the nodes crossed by at least min_cov sequences.
__device__ int poa_consensus(const PoaWs& w, int N_, int min_cov, int8_t* out, int cap, int lane)
{
    const int N = __builtin_amdgcn_readfirstlane(N_);
    if (N == 0) return 0;
    const bool in_lds = N <= POA_LDS_SCORES;
    int* score = in_lds ? (int*)poa_lds : w.score;
#pragma unroll 4
    for (int r = 1 + lane; r <= N; r += 64) {
        const int v = w.order[r - 1];
        int np = w.np[v];
        uint32_t pr[3] = {0, 0, 0}, wt[3] = {0, 0, 0};
        bool wide = np > 3;
        for (int s2 = 0; s2 < 3; ++s2) if (s2 < np) { pr[s2] = (uint32_t)w.rank[w.pred[v * POA_MAXP + s2]]; wt[s2] = (uint32_t)w.pw[v * POA_MAXP + s2]; wide |= wt[s2] > 255u; }
        w.tab[r * 3 + 0] = (uint32_t)(wide ? 0x7f : np) | (w.nout[v] == 0 ? 0x80u : 0u) | (pr[0] << 16);
        w.tab[r * 3 + 1] = pr[1] | (pr[2] << 16);
        w.tab[r * 3 + 2] = (wt[0] & 0xff) | ((wt[1] & 0xff) << 8) | ((wt[2] & 0xff) << 16);
    }
    if (lane == 0) score[0] = -1;
    phase_sync();
    // one pass over ranks (from, N]: barred = skip tails whose score is -1 (BranchCompletion).  returns the best rank
    auto pass = [&](int from, bool barred) -> int {
        int top = 0, tops = -(1 << 30), prev = from >= 1 ? __builtin_amdgcn_readfirstlane(score[from]) : -1;
        const int base0 = from + 1;
        uint32_t b0 = 0, b1 = 0, b2 = 0;
        if (base0 + lane <= N) { b0 = w.tab[(base0 + lane) * 3]; b1 = w.tab[(base0 + lane) * 3 + 1]; b2 = w.tab[(base0 + lane) * 3 + 2]; }
        for (int rb = base0; rb <= N; rb += 64) {
            const int nr = rb + 64 + lane;
            uint32_t n0 = 0, n1 = 0, n2 = 0;
            if (nr <= N) { n0 = w.tab[nr * 3]; n1 = w.tab[nr * 3 + 1]; n2 = w.tab[nr * 3 + 2]; }
            const int cnt = N - rb + 1 < 64 ? N - rb + 1 : 64;
            int bpb = 0;
            for (int i = 0; i < cnt; ++i) {
                const int r = rb + i;
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, i), t1 = (uint32_t)__builtin_amdgcn_readlane((int)b1, i), t2 = (uint32_t)__builtin_amdgcn_readlane((int)b2, i);
                const int np = (int)(t0 & 0x7f);
                int sc = -1, bsrc = 0, bscore = 0;               // score so far, chosen tail (rank, 0 = none) and its score
                auto relax = [&](int u, int wt) {
                    const int su = u == r - 1 ? prev : __builtin_amdgcn_readfirstlane(score[u]);
                    if (barred && su == -1) return;
                    if (sc < wt || (sc == wt && bscore <= su)) { sc = wt; bsrc = u; bscore = su; }
                };
                if (np == 0x7f) {
                    const int v = w.order[r - 1];
                    const int cn = w.np[v];
                    for (int s2 = 0; s2 < cn; ++s2)
                        relax(__builtin_amdgcn_readfirstlane(w.rank[w.pred[v * POA_MAXP + s2]]), __builtin_amdgcn_readfirstlane(w.pw[v * POA_MAXP + s2]));
                } else {
                    if (np > 0) relax((int)(t0 >> 16), (int)(t2 & 0xff));
                    if (np > 1) relax((int)(t1 & 0xffff), (int)((t2 >> 8) & 0xff));
                    if (np > 2) relax((int)(t1 >> 16), (int)((t2 >> 16) & 0xff));
                }
                if (bsrc > 0) sc += bscore;
                score[r] = sc;
                bpb = lane == i ? bsrc : bpb;
                prev = sc;
                if (sc > tops) { tops = sc; top = r; }           // first strictly largest
                if (in_lds) asm volatile("" ::: "memory"); else __syncthreads();
            }
            if (lane < cnt) w.bp[rb + lane] = bpb;
            b0 = n0; b1 = n1; b2 = n2;
        }
        return top;
    };
    int top = pass(0, false);
    for (int rounds = 0;; ++rounds) {
        if (rounds > N) return -1;                               // every completion moves to a later rank
        phase_sync();
        const uint32_t tt = w.tab[top * 3];
        if (__builtin_amdgcn_readfirstlane((int)tt) & 0x80) break;                  // a sink: done
        // BranchCompletion: the other tails of the successors of `top` are barred, later ranks recomputed
        for (int r = top + 1 + lane; r <= N; r += 64) {
            const int v = w.order[r - 1];
            const int np = w.np[v];
            bool succ = false;
            for (int s2 = 0; s2 < np; ++s2) succ |= w.rank[w.pred[v * POA_MAXP + s2]] == top;
            if (succ) for (int s2 = 0; s2 < np; ++s2) { const int u = w.rank[w.pred[v * POA_MAXP + s2]]; if (u != top) score[u] = -1; }
        }
        phase_sync();
        const int t2 = pass(top, true);
        if (t2 <= 0) break;                                                          // cannot happen
        top = t2;
    }
    // chase the back pointers: lane l holds bp[c0 - l]; the path descends a few ranks per step
    int len = 0, r = top, c0 = -1000, pbuf = 0, blk = 0;
    while (r > 0) {
        if (len > N) return -1;                                  // a path visits a rank once
        int off = c0 - r;
        if ((unsigned)off >= 64u) { c0 = r; blk = r - lane >= 1 ? w.bp[r - lane] : 0; off = 0; }
        pbuf = lane == (len & 63) ? r : pbuf;
        ++len;
        if ((len & 63) == 0) w.pn[len - 64 + lane] = pbuf;
        r = __builtin_amdgcn_readlane(blk, off);
    }
    if (len & 63) { const int q = (len & ~63) + lane; if (q < len) w.pn[q] = pbuf; }
    phase_sync();
    // output in path order (pn holds it reversed), nodes below min_cov left out
    int olen = 0;
    for (int k0 = 0; k0 < len; k0 += 64) {
        const int k = k0 + lane;
        int v = -1;
        if (k < len) v = w.order[w.pn[len - 1 - k] - 1];
        const bool kp = v >= 0 && w.cov[v] >= min_cov;
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(kp);
        const int pos = olen + __builtin_popcountll(bm & (((unsigned long long)1 << lane) - 1));
        if (kp && pos < cap) out[pos] = w.base[v];
        olen += __builtin_popcountll(bm);
    }
    return olen > cap ? -1 : olen;
}

// MSA columns: one per aligned set in rank order.  col[node] via w.score (free by now); returns the number of columns
__device__ int poa_msa_columns(const PoaWs& w, int N, int lane)
{
    int nc = 0;
    for (int r0 = 1; r0 <= N; r0 += 64) {
        const int r = r0 + lane;
        bool lead = false;
        int v = -1;
        if (r <= N) { v = w.order[r - 1]; int lo, hi; group_span(w, v, lo, hi); lead = lo == r; }
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(lead);
        if (lead) w.score[v] = nc + __builtin_popcountll(bm & (((unsigned long long)1 << lane) - 1));
        nc += __builtin_popcountll(bm);
    }
    phase_sync();
    for (int r = 1 + lane; r <= N; r += 64) {
        const int v = w.order[r - 1];
        int lo, hi;
        group_span(w, v, lo, hi);
        if (lo != r) w.bp[v] = w.score[w.order[lo - 1]]; else w.bp[v] = w.score[v];
    }
    phase_sync();
    return nc;
}

__global__ void __launch_bounds__(64, POA_WAVES) poa_consensus_kernel(const CcsParams p)
{
    const int lane = threadIdx.x & 63;
    uint8_t* slot = p.poa_ws + (size_t)blockIdx.x * p.slot_bytes;
    const PoaScores S = p.sc;
    for (int turns = 0; turns <= p.n; ++turns) {             // a wave takes at most every read once
        int idx = 0;
        if (lane == 0) idx = atomicAdd(p.work_counter, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);    // wave-uniform in the compiler's eyes too: scalar loads, scalar branches, SGPR pointers below
        if (idx >= p.n) break;
        const int rd = p.work_order ? p.work_order[idx] : idx;
        const int64_t off = p.read_off[rd];
        const int L = (int)(p.read_off[rd + 1] - off);
        const int8_t* seq = p.reads + off;
        if (p.tier == 1 && __builtin_amdgcn_readfirstlane(p.results[rd].status) != 1) continue;   // second tier: only what did not fit a first-tier slot
        CcsResult res;
        res.nseg = 0; res.ccs_len = 0; res.period = 0; res.status = 0;
        // the sequences: copies found by K2, or the explicit sequences of a group (poa API)
        const int32_t* cuts;
        int ncuts, period;
        bool tail;
        if (p.xcuts) {
            const int64_t c0 = p.xcut_off[rd];
            cuts = p.xcuts + c0; ncuts = (int)(p.xcut_off[rd + 1] - c0); period = -1; tail = true;
        } else {
            const CcsScan* sc = p.scan + rd;
            period = __builtin_amdgcn_readfirstlane(sc->period);
            ncuts = __builtin_amdgcn_readfirstlane(sc->ncuts);
            cuts = sc->cuts;
            const int bl = ncuts > 0 ? __builtin_amdgcn_readfirstlane(cuts[ncuts - 1]) : 0;
            tail = L - bl >= CCS_MIN_TAIL && ncuts < CCS_MAX_CUTS;     // a scan that stopped at its cap leaves the rest of the read out
        }
        res.period = period;
        if (period == 0) { if (lane == 0) p.results[rd] = res; continue; }
        if (p.tier == 1 && lane == 0 && p.stats) atomicAdd(p.stats + 1, 1);
        int nseg = 0, b = 0, maxlen = 0, total = 0;
        for (int i = 0; i <= ncuts; ++i) {
            if (i == ncuts && !tail) break;
            const int cut = i < ncuts ? __builtin_amdgcn_readfirstlane(cuts[i]) : L;
            if (!p.xcuts && lane == 0) { p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg] = b; p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg + 1] = cut; }
            const int len = cut - b;
            maxlen = len > maxlen ? len : maxlen; total += len;
            b = cut; ++nseg;
        }
        const int ncap = total + 8, mcap = maxlen + 1;
        if (maxlen > POA_MAX_COPY) { res.status = 4; if (lane == 0) p.results[rd] = res; continue; }   // cells are int16
        // workspace: this wave's slot, or -- a read that needs more -- one of the large slots, claimed for the duration of
        // the read; none free (or none large enough): status 1, the second launch over the large slots takes the read
        uint8_t* ws = slot;
        size_t ws_bytes = p.slot_bytes;
        int big = -1;
        const size_t need_min = poa_fixed_bytes(ncap, mcap, nullptr) + poa_dp_bytes(maxlen + 8, maxlen, 0, 0) + 64;
        bool use_big = need_min > p.slot_bytes;
        int N = 0, len = -1, ncols = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (use_big) {
                if (p.tier == 0 && p.n_big > 0 && need_min <= p.big_slot_bytes) {
                    if (lane == 0)
                        for (int t = 0; t < 64 && big < 0; ++t) {
                            const int cand = (int)((blockIdx.x * 7u + (unsigned)t * 131u + (unsigned)rd) % (unsigned)p.n_big);
                            if (atomicCAS(&p.big_busy[cand], 0, 1) == 0) big = cand;
                        }
                    big = __builtin_amdgcn_readfirstlane(big);
                }
                if (big < 0) { N = -2; break; }
                ws = p.big_ws + (size_t)big * p.big_slot_bytes; ws_bytes = p.big_slot_bytes;
                if (lane == 0 && p.stats) atomicAdd(p.stats, 1);
            }
            PoaWs w = carve(ws, ws_bytes, ncap, mcap);
            phase_sync();
            unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            N = 0; b = 0;
            int si = 0;
            for (int i = 0; i <= ncuts && N >= 0; ++i) {
                if (i == ncuts && !tail) break;
                const int cut = i < ncuts ? __builtin_amdgcn_readfirstlane(cuts[i]) : L;
                int sc1 = 0;
                N = poa_add(w, S, N, ncap, seq + b, cut - b, lane, &sc1, p.msa_col ? p.msa_col + off + b : nullptr, tacc);
                if (p.aln_score && si < CCS_SEG_CAP && lane == 0) p.aln_score[(size_t)rd * CCS_SEG_CAP + si] = sc1;
                b = cut; ++si;
            }
            if (N == -2 && !use_big && p.tier == 0) { use_big = true; continue; }     // the DP planes outgrew this slot: once more in a large one
            if (N >= 0) {
                const int mc = S.min_cov >= 0 ? S.min_cov : (nseg + 1) / 2;
#ifdef CLH_DEBUG_POA
                const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
                len = poa_consensus(w, N, mc, p.ccs + off, L, lane);
#ifdef CLH_DEBUG_POA
                tacc[4] += __builtin_amdgcn_s_memtime() - tc0;
#endif
                if (len >= 0 && p.msa_col) {
                    phase_sync();
                    ncols = poa_msa_columns(w, N, lane);
                    for (int t = lane; t < L; t += 64) p.msa_col[off + t] = w.bp[p.msa_col[off + t]];
                }
#ifdef CLH_DEBUG_POA
                if (lane == 0) for (int k = 0; k < 5; ++k) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (55 + k)] = (int)(tacc[k] >> 4);
#endif
            }
            break;
        }
        if (N == -2) res.status = 1;
        else if (N == -3) res.status = 5;
        else if (N < 0) res.status = 2;
        else if (len < 0) res.status = 3;
        else { res.nseg = nseg; res.ccs_len = len; }
        if (lane == 0) { p.results[rd] = res; if (p.msa_ncols) p.msa_ncols[rd] = ncols; }
        __syncthreads();
        if (big >= 0) {                      // every store into the large slot has landed before another wave may claim it
            __threadfence();
            if (lane == 0) atomicExch(&p.big_busy[big], 0);
        }
    }
}

hipError_t launch_ccs_scan(const CcsParams& p, hipStream_t stream)
{
    const size_t lds = k2_lds_bytes(p.lcap);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)ccs_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(ccs_scan_kernel, dim3(p.n), dim3(64), lds, stream, p);
    if (p.n_long > 0) hipLaunchKernelGGL(ccs_scan_long_kernel, dim3(p.n_long), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_poa(const CcsParams& p, int nslots, hipStream_t stream)
{
    hipLaunchKernelGGL(poa_consensus_kernel, dim3(nslots), dim3(64), (size_t)POA_LDS_BYTES, stream, p);
    return hipGetLastError();
}

size_t poa_slot_bytes_host(int ncap, int mcap) { return poa_slot_bytes(ncap, mcap); }
size_t poa_slot_min_bytes_host(int ncap, int mcap) { return poa_fixed_bytes(ncap, mcap, nullptr) + poa_dp_bytes(mcap + 8, mcap - 1, 0, 0) + 64; }

}  // namespace clh
