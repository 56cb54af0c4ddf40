// ccs_poa.hip -- K2 ccs_scan_kernel (repeat period + copy boundaries) and K3 poa_consensus_kernel (partial-order
// alignment of the copies, heaviest-path consensus) for gfx950.
//
// What they replace: pyccs.find_consensus (called at CIRI_long/find_ccs.py:14) and the spoa engine inside it.  Those are
// external packages that exist neither in the reference tree nor in this environment: PARITY UNPINNED.  Both kernels
// implement, bit for bit, what oracle/ccs_oracle.c ("clh-ccs v2": period and copy boundaries, this project's own
// specification) and oracle/poa_oracle.c ("clh-poa v3": a restatement of the published spoa algorithm -- two-piece gap
// cost, local/global/overlap alignment, Graph::TopologicalSort's depth-first order, AddAlignment's node ids, raw-byte letters,
// heaviest bundle; no departures) state; read those headers for every rule and tie-break.
// One read per wavefront, one wavefront per workgroup.
//
// K2: 8-mer codes of the read sit in LDS.  Matches per offset are counted per PAIR of equal 8-mers (positions chained per
//     hash bucket with LDS atomics, each pair visited once: O(L * copies) instead of O(L^2/4) comparisons); the smoothed
//     maximum, the harmonic test and the per-copy boundary search (same chains, a histogram over the candidate offsets)
//     are lane-parallel with shuffle reductions.
// K3: persistent waves pull reads from an atomic counter; each owns a workspace slot in HBM (graph arrays, the two 16-bit
//     planes of one sequence's pass; slots are sized for the common case, the few reads that need more claim one of a
//     handful of large slots).  A DP row (one graph node) is computed by the 64 lanes over the sequence positions, two
//     cells per lane-operation: the diagonal and the two vertical gap states over the node's in-edges are independent per
//     position, the two horizontal gap states are max-plus prefix scans (DPP).  The pass leaves H and one word of clamped
//     state differences per cell (a band around the matrix diagonal); spoa's value-comparing back-track is replayed from those
//     on the cells of the path only, staged in LDS.  The graph update is data-parallel; the topological sort reproduces the
//     order of spoa's sequential depth-first search from independent pieces (one small search per lane); the heaviest bundle
//     runs wave-uniformly on data staged in LDS.  DESIGN.md section 4 (K3) has the bounds, LABNOTES.md sections 3 and 9 the measurements of rounds 2-5.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"
#include "clh_device_ops.h"

namespace clh {

static constexpr int CCS_K = 8;
static constexpr int CCS_DMIN = 30;
static constexpr int CCS_MIN_SUPPORT = 12;
static constexpr int CCS_SMOOTH = 3;
static constexpr int CCS_MAX_CUTS = 64;
static constexpr int CCS_MIN_TAIL = 20;
static constexpr int CCS_GUARD = 8;                   // an anchor further than this from the voted offset is a chance recurrence
static constexpr int POA_MAXP = 12;                  // in-edges a node holds in place; further ones go to the graph's overflow table
static constexpr int POA_MAXP_ALL = 48;              // in-edges of a node, overflow included (the sort's per-node cursor has 6 bits for in-edges + aligned nodes)
static constexpr int POA_OVF_CAP = 2048;             // entries of the overflow table of one graph
static constexpr int POA_MAXA = 7;                   // other members of an aligned set: 8 different letters in one column (implementation limit)
static constexpr int POA_MAX_COPY = 2800;            // longest sequence: cells are int16 (match score <= 11)

// Phase boundary inside one wave whose lanes exchange data through HBM.  The workgroup IS that wave, so the block-level guarantee is all it
// needs: stores before the barrier are visible to every lane of the block behind it (on gfx950 in non-threadgroup-split mode a work-group's
// waves share one vector L1, which stays coherent with the CU's own stores -- the LLVM memory model's work-group-scope acquire is a wait, not
// an invalidation).  Rounds 1-4 dropped the CU's whole L1 here (an AGENT-scope acquire, ~17 times per added sequence, 1.45e6 times per C3
// launch): 0.7 ms of 17.6, paid as latency by all sixteen waves of the CU.  Data that crosses CUs still gets the agent-scope acquire where it
// crosses: a large slot another wave may have written from another CU (poa_kernel_body, at the claim).  POA_AGENT_PHASE_SYNC: the old form.
__device__ __forceinline__ void phase_sync() {
    __syncthreads();
#ifdef POA_AGENT_PHASE_SYNC
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#endif
}

// two independent scans, their steps interleaved: the other chain's instruction provides the wait states a DPP read needs
// after a VALU write, so no s_nop inside
__device__ __forceinline__ void wave_prefix_max2(int& a, int& b) {
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(a), "+v"(b));
}

// ------------------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------------------
// LDS layout for a batch whose longest read has lcap bases (lcap a multiple of 64):
//   head[k2_buckets(lcap)] int32 | cnt[lcap/2+2] int32 | sm[lcap/2+2] int32 | code[lcap] uint16 | next[lcap] int16
static constexpr int K2_BUCKETS = 2048;         // most buckets (the HBM-workspace kernel for reads above K2_LDS_MAX)
// buckets of a launch whose longest read has lcap bases: fewer for short reads -- the LDS block shrinks (more waves per CU hide
// the LDS latency of the chain walks; measured on ~1 kb reads: 2048 -> 1.24 ms, 1024 -> 1.19, 512 -> 1.06, 256 -> 1.15, 128 -> 1.39)
// at ~2 positions per bucket.  The counts do not depend on it: equal k-mers share a bucket whatever the number of buckets.
__host__ __device__ inline int k2_buckets(int lcap) { return lcap <= 1536 ? 512 : (lcap <= 3072 ? 1024 : 2048); }
static constexpr int K2_WIDE_FROM = 1024;      // launch classes from this many bases on: four waves per read (ccs_scan_kernel_x4)
static constexpr int K2_LDS_MAX = 16000;       // longest read scanned out of LDS (8 bytes per base + 8 KiB of 160 KiB)
__host__ __device__ inline size_t k2_lds_bytes(int lcap) { return 4 * (size_t)k2_buckets(lcap) + 8 * ((size_t)lcap / 2 + 2) + 4 * (size_t)lcap; }

// The scan of one read.  NextT = int16_t with every array in LDS (reads up to K2_LDS_MAX bases), int32_t with cnt/sm/
// code/next in an HBM workspace (longer reads: rare, so the slower memory does not matter; no length limit).
// NT threads work on the read (64, or 256 for the launch classes of long reads: their LDS block admits only 3..7 workgroups per
// CU, and one wave each leaves the chain walks' LDS latency bare); `lane` is the thread's index among them.
template <typename NextT, int NT = 64>
__device__ void ccs_scan_read(const CcsParams& p, const int rd, const int lane, int32_t* head, const int nbuckets, int32_t* cnt, int32_t* sm, uint16_t* code, NextT* next, int* red = nullptr)
{
    const int64_t off = p.read_off[rd];
    const int L = (int)(p.read_off[rd + 1] - off);
    const int8_t* seq = p.reads + off;
    // the read's record is written where it lives (a record per thread would sit in scratch: 272 bytes x 256 threads per read)
    CcsScan* const rec = p.scan + rd;
    auto none = [&]() { if (lane == 0) { rec->period = 0; rec->ncuts = 0; rec->support = 0; } };
    // lanes exchange data through the arrays: LDS needs a barrier, the HBM workspace also a cache invalidate
    auto sync = [&]() { if constexpr (sizeof(NextT) == 4) phase_sync(); else __syncthreads(); };
    if (L < 2 * CCS_DMIN) { none(); return; }

    // The specification counts, per offset d, the positions i with equal valid k-mers at i and i+d: that is one count per
    // PAIR of equal k-mers.  A read of L bases has O(L * copies) such pairs, not O(L^2/4): the positions are chained per
    // hash bucket and every pair is visited once, instead of comparing all (i, d).
    const int dmax = L / 2;
    for (int i = lane; i < nbuckets; i += NT) head[i] = -1;
    for (int d = lane; d <= dmax + 1; d += NT) cnt[d] = 0;
    sync();
    for (int i = lane; i < L; i += NT) {
        int32_t c = 0, ok = i + CCS_K <= L;
        if (ok)
            for (int t = 0; t < CCS_K; ++t) {
                const int b = seq[i + t];
                if (b < 0 || b > 3) ok = 0;
                c = (c << 2) | (b & 3);
            }
        code[i] = (uint16_t)c;
        next[i] = ok ? (NextT)atomicExch(&head[(c ^ (c >> 5)) & (nbuckets - 1)], i) : (NextT)-2;
    }
    sync();
    for (int i = lane; i < L; i += NT) {
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
    for (int d = CCS_DMIN + lane; d <= dmax; d += NT) {
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
    if constexpr (NT > 64) {                           // the waves' answers through LDS
        if ((lane & 63) == 0) { red[2 * (lane >> 6)] = best; red[2 * (lane >> 6) + 1] = bestd; }
        __syncthreads();
        best = red[0]; bestd = red[1];
        for (int v = 1; v < NT / 64; ++v) { const int b2 = red[2 * v], d2 = red[2 * v + 1]; if (b2 > best || (b2 == best && d2 < bestd)) { best = b2; bestd = d2; } }
    }
    sync();
    if (best < CCS_MIN_SUPPORT) { none(); return; }
    int p0 = bestd;
    for (int q = 2; q <= 8; ++q) {
        const int c = (bestd + q / 2) / q;
        if (c < CCS_DMIN) break;
        int es = -1, eb = -1;
        const int lo = c - 3 < CCS_DMIN ? CCS_DMIN : c - 3, hi = c + 3 > dmax ? dmax : c + 3;
        for (int e = lo; e <= hi; ++e) if (sm[e] > es) { es = sm[e]; eb = e; }
        if (eb >= 0 && 2 * es >= best) p0 = eb;
    }
    // copy boundaries (clh-ccs v2, oracle/ccs_oracle.c step 2): per cut, the anchors (i, delta) -- equal valid k-mers at i and i+delta,
    // i in [b-W, b+W), delta in [p0-tol, min(p0+tol, L-b)] -- first vote for an offset (a histogram over delta; sm is free by now;
    // most anchors, then closest to the previous step, then smallest delta), then the anchor whose k-mer centre lies nearest to b among
    // those within CCS_GUARD of the vote gives the cut.  Same pairs as the period count: the window's positions walk their bucket
    // chains (from the head: partners on both sides are in it), once per pass.
    const int tol = p0 / 8 > 4 ? p0 / 8 : 4, W = p0 < 96 ? p0 : 96;
    const int dlo = p0 - tol;
    int32_t* dh = sm;
    sync();
    int b = 0, prev = p0, n = 0;
    while (n < CCS_MAX_CUTS) {
        const int dhi = p0 + tol < L - b ? p0 + tol : L - b;
        if (dhi < dlo) break;
        const int nd = dhi - dlo + 1;
        const int i0 = b - W > 0 ? b - W : 0, i1 = b + W < L ? b + W : L;
        for (int t = lane; t < nd; t += NT) dh[t] = 0;
        sync();
        for (int i = i0 + lane; i < i1; i += NT) {
            if (next[i] == -2) continue;
            const int ci = code[i];
            int j = head[(ci ^ (ci >> 5)) & (nbuckets - 1)];
            while (j >= 0) {
                const int delta = j - i;
                if (delta >= dlo && delta <= dhi && code[j] == ci) atomicAdd(&dh[delta - dlo], 1);
                j = next[j];
            }
        }
        sync();
        int bs = -1, bdel = 0, bdist = 0x7fffffff;
        for (int t = lane; t < nd; t += NT) {
            const int delta = dlo + t;
            const int sc = dh[t];
            const int dist = delta > prev ? delta - prev : prev - delta;
            if (sc > bs || (sc == bs && dist < bdist)) { bs = sc; bdel = delta; bdist = dist; }   // ascending delta within the lane
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int s2 = __shfl_xor(bs, d), e2 = __shfl_xor(bdel, d), t2 = __shfl_xor(bdist, d);
            if (s2 > bs || (s2 == bs && (t2 < bdist || (t2 == bdist && e2 < bdel)))) { bs = s2; bdel = e2; bdist = t2; }
        }
        if constexpr (NT > 64) {
            if ((lane & 63) == 0) { red[8 + 3 * (lane >> 6)] = bs; red[9 + 3 * (lane >> 6)] = bdel; red[10 + 3 * (lane >> 6)] = bdist; }
            __syncthreads();
            bs = red[8]; bdel = red[9]; bdist = red[10];
            for (int v = 1; v < NT / 64; ++v) {
                const int s2 = red[8 + 3 * v], e2 = red[9 + 3 * v], t2 = red[10 + 3 * v];
                if (s2 > bs || (s2 == bs && (t2 < bdist || (t2 == bdist && e2 < bdel)))) { bs = s2; bdel = e2; bdist = t2; }
            }
        }
        if (bs > 0) {
            // the anchor nearest to b: smallest |i + K/2 - b|, then the larger i, then smallest |delta - prev|, then smallest delta
            const int glo = bdel - CCS_GUARD > dlo ? bdel - CCS_GUARD : dlo, ghi = bdel + CCS_GUARD < dhi ? bdel + CCS_GUARD : dhi;
            int ad = 0x7fffffff, ai = -1, aa = 0x7fffffff, adel = 0x7fffffff;
            auto closer = [](int d2, int i2, int a2, int e2, int d1, int i1_, int a1, int e1) {
                return d2 < d1 || (d2 == d1 && (i2 > i1_ || (i2 == i1_ && (a2 < a1 || (a2 == a1 && e2 < e1)))));
            };
            for (int i = i0 + lane; i < i1; i += NT) {
                if (next[i] == -2) continue;
                const int ci = code[i];
                const int dist = i + CCS_K / 2 > b ? i + CCS_K / 2 - b : b - i - CCS_K / 2;
                if (dist > ad) continue;
                int j = head[(ci ^ (ci >> 5)) & (nbuckets - 1)];
                while (j >= 0) {
                    const int delta = j - i;
                    if (delta >= glo && delta <= ghi && code[j] == ci) {
                        const int a = delta > prev ? delta - prev : prev - delta;
                        if (closer(dist, i, a, delta, ad, ai, aa, adel)) { ad = dist; ai = i; aa = a; adel = delta; }
                    }
                    j = next[j];
                }
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int d2 = __shfl_xor(ad, d), i2 = __shfl_xor(ai, d), a2 = __shfl_xor(aa, d), e2 = __shfl_xor(adel, d);
                if (closer(d2, i2, a2, e2, ad, ai, aa, adel)) { ad = d2; ai = i2; aa = a2; adel = e2; }
            }
            if constexpr (NT > 64) {
                if ((lane & 63) == 0) { int* r4 = red + 24 + 4 * (lane >> 6); r4[0] = ad; r4[1] = ai; r4[2] = aa; r4[3] = adel; }
                __syncthreads();
                ad = red[24]; ai = red[25]; aa = red[26]; adel = red[27];
                for (int v = 1; v < NT / 64; ++v) {
                    const int* r4 = red + 24 + 4 * v;
                    if (closer(r4[0], r4[1], r4[2], r4[3], ad, ai, aa, adel)) { ad = r4[0]; ai = r4[1]; aa = r4[2]; adel = r4[3]; }
                }
            }
            bdel = adel;                                   // the vote's own anchors pass the guard: there is one
        }
        sync();
        b += bdel;
        prev = bdel;
        if (lane == 0) rec->cuts[n] = b;
        ++n;
    }
    if (n < 2) { none(); return; }
    if (lane == 0) { rec->period = p0; rec->ncuts = n; rec->support = best; }
}

template <int NT>
__device__ __forceinline__ void ccs_scan_body(const CcsParams& p, int32_t* k2_lds, int* red)
{
    const int tid = threadIdx.x;
    const int rd = p.work_order[p.k2_begin + blockIdx.x];   // launch classes by read length: a short read gets a small LDS block
    if ((int)(p.read_off[rd + 1] - p.read_off[rd]) > p.k2_lds_max) return;        // ccs_scan_long_kernel takes it
    int32_t* head = k2_lds;                         // bucket -> last inserted position, -1 = empty
    const int nb = k2_buckets(p.lcap);
    int32_t* cnt = head + nb;                       // matches per offset, [lcap/2 + 2]
    int32_t* sm = cnt + p.lcap / 2 + 2;             // smoothed counts; later the per-cut histogram of offsets
    uint16_t* code = (uint16_t*)(sm + p.lcap / 2 + 2);
    int16_t* next = (int16_t*)(code + p.lcap);      // chain of the positions of a bucket; -1 = end (invalid k-mers are in no chain)
    ccs_scan_read<int16_t, NT>(p, rd, tid, head, nb, cnt, sm, code, next, red);
}
__global__ void __launch_bounds__(64) ccs_scan_kernel(const CcsParams p)
{
    extern __shared__ __attribute__((aligned(16))) int32_t k2_lds[];
    ccs_scan_body<64>(p, k2_lds, nullptr);
}
// the same by four waves per read: the launch classes whose LDS block is large (K2_WIDE_FROM bases and more)
__global__ void __launch_bounds__(256) ccs_scan_kernel_x4(const CcsParams p)
{
    extern __shared__ __attribute__((aligned(16))) int32_t k2_lds[];
    __shared__ int red[40];
    ccs_scan_body<256>(p, k2_lds, red);
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
    ccs_scan_read<int32_t>(p, rd, lane, head, K2_BUCKETS, cnt, sm, code, next);
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
    int8_t* base; int8_t* np; int8_t* na; int32_t* pred; int32_t* pw; int32_t* aligned; int32_t* cov; int32_t* nout; int32_t* order; int32_t* rank;
    int32_t* root; uint32_t* rsz; int32_t* ntr; uint8_t* st;                 // the topological sort (poa_sort)
    int32_t* pn; int32_t* pj;                                                // per base of the sequence being added
    uint2* ri; uint32_t* tab; int32_t* score; int32_t* bp; short* col0;     // rank space
    short* carry; int cpitch;
    uint8_t* dp; size_t dp_bytes;                                            // the rest of the slot: DP planes of the current sequence
    short* planeH; unsigned short* planeD;                                   // set per sequence (poa_add): H and the clamped differences
    int32_t* ovn; int32_t* ovp; int32_t* ovw; int32_t* ovc;                  // in-edges beyond POA_MAXP: node, source, weight per entry (in the order they were made); ovc[0] = entries
};

// the s-th in-edge of node v (spoa's order = the order the edges were made): in place, or the (s - POA_MAXP)-th entry of v in the
// overflow table (a linear search: a node with more than 12 in-edges is rare, the table short)
__device__ __forceinline__ int poa_ovf_slot(const PoaWs& w, int v, int k)
{
    const int n = w.ovc[0];
    for (int i = 0; i < n; ++i) if (w.ovn[i] == v) { if (k == 0) return i; --k; }
    return 0;                                                                // cannot happen (k < np[v] - POA_MAXP)
}
__device__ __forceinline__ int poa_pred_at(const PoaWs& w, int v, int s) { return s < POA_MAXP ? w.pred[v * POA_MAXP + s] : w.ovp[poa_ovf_slot(w, v, s - POA_MAXP)]; }
__device__ __forceinline__ int poa_pw_at(const PoaWs& w, int v, int s) { return s < POA_MAXP ? w.pw[v * POA_MAXP + s] : w.ovw[poa_ovf_slot(w, v, s - POA_MAXP)]; }

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
    add(sizeof(int32_t) * ncap); add(sizeof(uint32_t) * ncap); add(sizeof(int32_t) * ncap); add(ncap);    // root rsz ntr st
    add(sizeof(uint2) * (size_t)(ncap + 2)); add(sizeof(uint32_t) * 3 * (size_t)(ncap + 2));   // ri tab
    add(sizeof(int32_t) * (size_t)(ncap + 2)); add(sizeof(int32_t) * (size_t)(ncap + 2)); add(sizeof(short) * (size_t)(ncap + 2));        // score bp col0
    const int cp = (ncap + 2 + 7) & ~7;
    if (cpitch_out) *cpitch_out = cp;
    add(sizeof(short) * 6 * (size_t)cp);                                                                    // carries: 2 x (H, E, Q)
    add(ncap); add(ncap); add(ncap);                                                                       // base np na
    add(sizeof(int32_t) * POA_OVF_CAP); add(sizeof(int32_t) * POA_OVF_CAP); add(sizeof(int32_t) * POA_OVF_CAP); add(16);   // overflow in-edges: node, source, weight, count
    return o;
}
// DP planes of one sequence against N rows: H (int16) and the clamped differences (16 bits), one row more than the graph has
__host__ __device__ inline size_t poa_plane_bytes(int N, int m) { return (((size_t)(N + 1) * (size_t)poa_pitch(m) * 2 + 15) & ~(size_t)15) + 64; }
__host__ __device__ inline size_t poa_dp_bytes(int N, int m) { return 2 * poa_plane_bytes(N, m); }
__host__ __device__ inline size_t poa_slot_bytes(int ncap, int mcap)       // worst case
{
    return poa_fixed_bytes(ncap, mcap, nullptr) + poa_dp_bytes(ncap, mcap - 1) + 64;
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
    w.root = (int32_t*)take(sizeof(int32_t) * ncap);
    w.rsz = (uint32_t*)take(sizeof(uint32_t) * ncap);
    w.ntr = (int32_t*)take(sizeof(int32_t) * ncap);
    w.st = (uint8_t*)take(ncap);
    w.ri = (uint2*)take(sizeof(uint2) * (size_t)(ncap + 2));
    w.tab = (uint32_t*)take(sizeof(uint32_t) * 3 * (size_t)(ncap + 2));
    w.score = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + 2));
    w.bp = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + 2));
    w.col0 = (short*)take(sizeof(short) * (size_t)(ncap + 2));
    w.cpitch = (ncap + 2 + 7) & ~7;
    w.carry = (short*)take(sizeof(short) * 6 * (size_t)w.cpitch);
    w.base = (int8_t*)take(ncap);
    w.np = (int8_t*)take(ncap);
    w.na = (int8_t*)take(ncap);
    w.ovn = (int32_t*)take(sizeof(int32_t) * POA_OVF_CAP); w.ovp = (int32_t*)take(sizeof(int32_t) * POA_OVF_CAP); w.ovw = (int32_t*)take(sizeof(int32_t) * POA_OVF_CAP);
    w.ovc = (int32_t*)take(16);
    w.dp = slot + o;
    w.dp_bytes = slot_bytes > o ? slot_bytes - o : 0;
    w.planeH = nullptr; w.planeD = nullptr;
    return w;
}

extern __shared__ __attribute__((aligned(16))) uint32_t poa_lds[];     // K3's dynamic LDS block
#ifndef POA_LDS_KB
#define POA_LDS_KB 9
#endif
static constexpr int POA_LDS_BYTES = POA_LDS_KB * 1024;           // ring of recent rows (DP) / band of the planes + the sequence (back-track) / score per rank (heaviest bundle)
static constexpr int POA_LDS_SCORES = POA_LDS_BYTES / 4 - 1;   // most rows whose scores fit the LDS block

#ifdef CLH_DEBUG_POA
#define DBGARG , unsigned long long* tacc
#define DBGPASS , tacc
#define DBGCNT(k, v) do { tacc[k] += (unsigned long long)(v) << 4; } while (0)
#define SEC0() unsigned long long sec_t = __builtin_amdgcn_s_memtime()
#define SEC(k) do { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); tacc[k] += t2_ - sec_t; sec_t = t2_; } while (0)
#define TSTAMP(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define DBGARG
#define DBGPASS
#define DBGCNT(k, v) do {} while (0)
#define SEC0() do {} while (0)
#define SEC(k) do {} while (0)
#define TSTAMP(k) do {} while (0)
#endif

// columns per lane, LDS row pitch and ring depth for a sequence of m bases
#ifndef POA_MAXCP
#define POA_MAXCP 3         // packed registers (column pairs) per lane: up to 6 columns per lane, 384 per pass (measured: 2 -> 27.0, 3 -> 25.8, 4 -> 29.5 ms per 100k reads)
#endif
#ifndef POA_WAVES
#define POA_WAVES 4
#endif
__device__ __forceinline__ int poa_cols(int m) { const int c = (m + 127) >> 7; return c < 1 ? 1 : (c > POA_MAXCP ? POA_MAXCP : c); }   // column pairs per lane
__device__ __forceinline__ int poa_ring_cp(int m) { return m > 128 * POA_MAXCP ? POA_MAXCP : poa_cols(m); }     // registers per lane of the widest pass
__device__ __forceinline__ int poa_ring(int m) {
    const int row = 512 * poa_ring_cp(m) + 4;        // raw packed registers: H and differences, 64 lanes each, + the left-boundary H
    int ring = 16;                                   // power of two, so slot = rank & (ring-1)
    while (ring * row > POA_LDS_BYTES) ring >>= 1;
    return ring;
}

// DP rows of one sequence against the graph.  Lane l owns C adjacent columns (C = 2..8: ceil(length / 64)), so a row of up
// to W = 64*C columns is ONE step; longer sequences are swept in passes of W columns (passes outer, rows inner), the
// cell that leaves a row on the right handed to the next pass through per-row carries (H, E, Q) in HBM.  Graph rows
// (w.ri) and carries are streamed 64 rows at a time into one register per lane and read with v_readlane; LDS holds
// the ring of the last RING rows (raw packed registers) for near sources that are not the row before -- that
// one is forwarded from registers; far sources come from "kept" rows in HBM.
// ---- packed 16-bit form of the pass (the arithmetic: clh_device_ops.h) ------------------------------------------------
// Two cells per lane-operation: a register holds the cells of two columns in its 16-bit halves (v_pk_* arithmetic;
// every value of the DP fits int16).  Lane l owns C = 2 CP adjacent columns as two "virtual lanes": the low halves of its
// CP registers are columns C l .. C l + CP - 1, the high halves the next CP columns -- so the left neighbour of a cell is the
// same half of the previous register, and only register 0 needs the hand-down (low half <- previous lane's high half of
// the last register, high half <- own low half of the last register: one DPP move + one v_alignbit).
// (best, code) <- (x, cx) in the halves where best < x
__device__ __forceinline__ void upd(uint32_t& best, uint32_t& code, uint32_t x, uint32_t cx) {
    const uint32_t m = pk_lt_mask(best, x);
    best = pk_max(best, x);
    code = bfi_keep(m, cx, code);
}

// exclusive prefix maximum over all columns to the left, packed layout: a[] in, pe[] out; `left` = the value entering the pass
template <int CP>
__device__ __forceinline__ void scan_left_pk(const uint32_t (&a)[CP], int left, uint32_t (&pe)[CP])
{
    constexpr int NEGB = -(1 << 30);
    uint32_t run[CP];
    run[0] = 0x80008000u;
#pragma unroll
    for (int t = 1; t < CP; ++t) run[t] = pk_max(run[t - 1], a[t - 1]);
    const uint32_t tot = pk_max(run[CP - 1], a[CP - 1]);
    const int lo_t = (int)(short)(tot & 0xffffu), hi_t = (int)tot >> 16;
    const int m2 = lo_t > hi_t ? lo_t : hi_t;
    const int inc = wave_prefix_max(m2);
    int exc = dpp_shr1(NEGB, inc);
    exc = left > exc ? left : exc;                              // a 16-bit value from here on
    const int exhi = exc > lo_t ? exc : lo_t;
    const uint32_t ex2 = pack16(exc, exhi);
#pragma unroll
    for (int t = 0; t < CP; ++t) pe[t] = pk_max(run[t], ex2);
}

// (exc, max(exc, low half of tot)) as a packed pair, exc a 16-bit value in an int: ONE operation -- the maximum goes straight into the high half of the
// register that holds exc (SDWA, the low half kept).  In C it is a maximum, a mask and a shift-or.  (Behind wave_prefix_max2, whose block ends with
// the wait states a reader of its results needs.)
__device__ __forceinline__ uint32_t pack_exc(int exc, uint32_t tot) {
    uint32_t d = (uint32_t)exc;
    asm("v_max_i32_sdwa %0, %0, sext(%1) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0" : "+v"(d) : "v"(tot));
    return d;
}

// the two scans of a row (E frame, Q frame) with their wave-wide parts interleaved
template <int CP>
__device__ __forceinline__ void scan_left_pk2(const uint32_t (&a)[CP], int leftA, uint32_t (&pa)[CP], const uint32_t (&b)[CP], int leftB, uint32_t (&pb)[CP])
{
    uint32_t runA[CP], runB[CP];
    runA[0] = 0x80008000u; runB[0] = 0x80008000u;
#pragma unroll
    for (int t = 1; t < CP; ++t) { runA[t] = pk_max(runA[t - 1], a[t - 1]); runB[t] = pk_max(runB[t - 1], b[t - 1]); }
    const uint32_t totA = pk_max(runA[CP - 1], a[CP - 1]), totB = pk_max(runB[CP - 1], b[CP - 1]);
    const int loA = (int)(short)(totA & 0xffffu), hiA = (int)totA >> 16, loB = (int)(short)(totB & 0xffffu), hiB = (int)totB >> 16;
    // the lane totals move one lane up first (lane 0 takes the value entering the pass): the inclusive scan of that is the
    // exclusive prefix, the entering value included
    int excA = dpp_shr1(leftA, loA > hiA ? loA : hiA), excB = dpp_shr1(leftB, loB > hiB ? loB : hiB);
    wave_prefix_max2(excA, excB);
    const uint32_t exA = pack_exc(excA, totA), exB = pack_exc(excB, totB);
#pragma unroll
    for (int t = 0; t < CP; ++t) { pa[t] = pk_max(runA[t], exA); pb[t] = pk_max(runB[t], exB); }
}

// (x == 0 ? lo : lo + d) per half from x = letter ^ node letter: min and multiply-add written out -- left to itself the compiler
// "simplifies" min(x, 1) * d into per-half compares and selects (eight operations and their wait states instead of two)
__device__ __forceinline__ uint32_t pk_nz_select(uint32_t x, uint32_t d, uint32_t lo) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_mad_u16 %0, %0, %2, %3" : "=&v"(r) : "v"(x), "v"(d), "v"(lo));
    return r;
}
__device__ __forceinline__ uint32_t lshl_or(uint32_t a, int sh, uint32_t b) {   // (a << sh) | b as ONE instruction (the compiler reassociates a tree of them into shifts and ors)
    uint32_t d;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(sh), "v"(b));
    return d;
}
// a pointer the compiler cannot see to be wave-uniform, as a scalar-register pair (else it lives in two VGPRs -- and with some
// thirty workspace pointers alive that is what spills)
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    typedef __attribute__((address_space(1))) T* gptr;      // a pointer into HBM: rebuilt as such, or every access through it turns "flat"
    return (T*)(gptr)(((unsigned long long)hi << 32) | lo);
}

// lo half = half hx of x, hi half = half hy of y (one v_perm_b32)
template <int HX, int HY>
__device__ __forceinline__ uint32_t mix16(uint32_t x, uint32_t y) {
    constexpr uint32_t sel = (uint32_t)(2 * HX) | ((uint32_t)(2 * HX + 1) << 8) | ((uint32_t)(4 + 2 * HY) << 16) | ((uint32_t)(4 + 2 * HY + 1) << 24);
    return __builtin_amdgcn_perm(y, x, sel);
}
// packed registers (low halves = the lane's first CP columns, high halves the next CP) <-> natural column order (dword d = columns 2d, 2d+1)
template <int CP>
__device__ __forceinline__ void to_natural(const uint32_t (&p)[CP], uint32_t (&nat)[CP]) {
    if constexpr (CP == 1) nat[0] = p[0];
    else if constexpr (CP == 2) { nat[0] = mix16<0, 0>(p[0], p[1]); nat[1] = mix16<1, 1>(p[0], p[1]); }
    else if constexpr (CP == 3) { nat[0] = mix16<0, 0>(p[0], p[1]); nat[1] = mix16<0, 1>(p[2], p[0]); nat[2] = mix16<1, 1>(p[1], p[2]); }
    else { nat[0] = mix16<0, 0>(p[0], p[1]); nat[1] = mix16<0, 0>(p[2], p[3]); nat[2] = mix16<1, 1>(p[0], p[1]); nat[3] = mix16<1, 1>(p[2], p[3]); }
}
template <int CP>
__device__ __forceinline__ void to_packed(const uint32_t (&nat)[CP], uint32_t (&p)[CP]) {
    if constexpr (CP == 1) p[0] = nat[0];
    else if constexpr (CP == 2) { p[0] = mix16<0, 0>(nat[0], nat[1]); p[1] = mix16<1, 1>(nat[0], nat[1]); }
    else if constexpr (CP == 3) { p[0] = mix16<0, 1>(nat[0], nat[1]); p[1] = mix16<1, 0>(nat[0], nat[2]); p[2] = mix16<0, 1>(nat[1], nat[2]); }
    else { p[0] = mix16<0, 0>(nat[0], nat[2]); p[1] = mix16<1, 1>(nat[0], nat[2]); p[2] = mix16<0, 0>(nat[1], nat[3]); p[3] = mix16<1, 1>(nat[1], nat[3]); }
}

// ---- the lazy forward pass (round 3) ------------------------------------------------------------------------------------
// A cell keeps no back-track code.  It leaves H (int16) and ONE 16-bit word of clamped differences -- the two vertical states
// as the following rows read them (dF = max(F + e - g - H, -1) + 1 in 3 bits, dO likewise with c, q in 5) and the two horizontal
// states the same way (dE, dQ) -- in two planes in HBM, natural column order; spoa's value-comparing back-track is replayed
// from those on the few hundred cells of the path (poa_backtrack; tools/poa_model.py: backtrack_lazy is the derivation, checked
// against the five-matrix oracle on the CPU).  Per cell pair: the maxima only -- no strictly-greater updates, no code selection.
// A source row that is not the row before comes from the LDS ring (raw packed registers) or, older, from the planes.
static constexpr int D_F = 0x0007, D_O_SHIFT = 3, D_E_SHIFT = 8, D_Q_SHIFT = 11;
// The planes are what the kernel writes most of, and writing them is what bounds the pass (measured: the pass run twice costs 18 ms with
// its stores, 9 ms without).  The back-track only ever reads cells next to the path, and the path of a copy against the graph of
// its siblings stays near the straight line from (1, 1) to (N, m).  So a row leaves its cells only within POA_BAND (32) columns of
// column r * m / N -- rows that an older successor reads back as a source ("full", 0x8000 in the graph row) leave all of them --
// and the back-track, which knows the same rule, answers "miss" when it is about to use a cell that was not left: the pass is
// then run once more with every cell stored.  Results never depend on the band.
#ifndef POA_BAND_W
#define POA_BAND_W 32      // measured, C3 / C4 ms per launch (walks that left the band and ran their pass again): 64 -> 19.90 (0) / 78.3 (27), 48 -> 19.65 (0) / 77.9 (78), 32 -> 19.30 (22 of 85 000) / 77.3 (353 of 250 000); round 4 at 18.5 / 60.8: 24 -> 18.44 (109) / 61.4 (1491), 16 -> 18.31 (1711)
#endif
static constexpr int POA_BAND = POA_BAND_W;
static constexpr int POA_H_NONE = -32768;            // a cell of the staged band that the planes do not hold (H is never below POA_NEG)
static constexpr int BT_MISS = -2;

// SIMPLE: the pass of nearly every sequence of `call` -- local alignment, the whole sequence in ONE pass (no carries in from a pass before, none
// out) -- with those facts as compile-time constants: the row loop loses its tests of them (a dozen scalar branches and their set-up per row)
template <int CP, bool SIMPLE = false>
__device__ void dp_pass_lz(const PoaWs& w, const PoaScores S, int N, int m, const int8_t* seq, int lane, const int pass, const int colbase,
                           const bool more_in, const int RING, const int slope16, int& bs_io, int& br_io, int& bc_io DBGARG)
{
    constexpr int C = 2 * CP;
    const int gp = poa_pitch(m);
    const int rmask = RING - 1;
    const bool more = SIMPLE ? false : more_in;
    const bool sw = SIMPLE ? true : (S.algorithm & 0xff) == 0, nw = SIMPLE ? false : (S.algorithm & 0xff) == 1;
    // The workspace record lives in memory (this is a function of its own): every pointer of it that the row loop uses is taken out
    // here, once, as a scalar pair.  Left inside the loop the compiler reloads it per row -- and the wait for that load is a wait for
    // every store in flight (loads and stores share one counter), i.e. for the row's own plane stores to reach memory.
    char* const gH = (char*)uniform_ptr(w.planeH);
    char* const gD = (char*)uniform_ptr(w.planeD);
    const uint2* const gri = uniform_ptr(w.ri);
    const short* const gcol0 = uniform_ptr(w.col0);
    const int cpitch = __builtin_amdgcn_readfirstlane(w.cpitch);
    const int32_t* const gorder = uniform_ptr(w.order); const int32_t* const grank = uniform_ptr(w.rank); const int32_t* const gpred = uniform_ptr(w.pred);
    const int g = S.g, e = S.e, q = S.q, c = S.c;
    const uint32_t g2 = dup16(g), e2 = dup16(e), q2 = dup16(q), c2 = dup16(c);
    const uint32_t sm2 = opaque(dup16(S.m)), dsn2 = opaque(dup16(S.n - S.m));          // (operands of an asm block: kept in vector registers, not copied there per row)
    const uint32_t ge2 = dup16(g - e), qc2 = dup16(q - c);
    const uint32_t NEG2 = dup16(POA_NEG), ONE2 = 0x00010001u;
    const uint32_t floor2 = sw ? 0u : 0x80008000u;           // local mode: a cell is at least 0
    const bool beyond_free = sw && S.n < 0 && S.g < 0 && S.q < 0;
    // ring row: CP dwords of H per lane, then CP dwords of differences per lane; behind the rows one left-boundary H per row
    uint32_t* ring = poa_lds;
    const int rrow = 128 * poa_ring_cp(m);
    int* ringL = (int*)(poa_lds + RING * rrow);
    const bool last = !more;
    const int lc0 = C * lane, col0 = colbase + lc0;              // register t: low half = column col0+t+1, high half = column col0+CP+t+1
    const int mc = m - 1 - colbase;                              // last pass: where column m lives
    const int lm = mc / C, tm = (mc % C) % CP, hm = (mc % C) / CP;
    uint32_t sbP[CP], jeP[CP], jcP[CP];
#pragma unroll
    // letters are bytes; a column beyond the sequence holds 0x100 (bit 8): pk_sra15(sbP << 7) is 0xFFFF in those halves
    for (int t = 0; t < CP; ++t) {
        const int jlo = col0 + t + 1, jhi = jlo + CP;
        sbP[t] = pack16(jlo <= m ? (int)(uint8_t)seq[jlo - 1] : 0x100, jhi <= m ? (int)(uint8_t)seq[jhi - 1] : 0x100);
        jeP[t] = pack16((lc0 + t + 1) * e, (lc0 + CP + t + 1) * e);
        jcP[t] = pack16((lc0 + t + 1) * c, (lc0 + CP + t + 1) * c);
    }
    auto row0_h = [&](int j) -> int {                            // H[0][j]: row 0 (no node) as a source
        const int l1 = g + (j - 1) * e, l2 = q + (j - 1) * c;
        return (sw || j == 0) ? 0 : (l1 > l2 ? l1 : l2);
    };
    short* const gcarry = uniform_ptr(w.carry);
    const short* cprev = gcarry + (size_t)(pass & 1) * 3 * cpitch;
    short* cnext = gcarry + (size_t)((pass + 1) & 1) * 3 * cpitch;
    const bool carried = SIMPLE ? false : pass > 0;
    uint2 blk = make_uint2(0, 0); int cH = 0, cE = POA_NEG, cQ = POA_NEG;
    struct RowIn { uint2 b; int h, e, q; };
    // Every load unconditional, from a clamped index, the unwanted values dropped afterwards: around conditional loads the compiler
    // builds "load through a pointer that is either the array or a stack slot holding the default", and the row loop then starts
    // with a load from the stack -- whose wait is a wait for every plane store in flight
    auto fetch1 = [&](int rr) -> RowIn {
        const bool in = rr <= N;
        const int rc = in ? rr : N;
        const uint2 b = gri[rc];
        const int hc = (int)cprev[rc], ec = (int)cprev[cpitch + rc], qc = (int)cprev[2 * cpitch + rc], h0 = (int)gcol0[rc];
        RowIn x;
        x.b = in ? b : make_uint2(0, 0);
        x.h = !in ? 0 : (carried ? hc : (nw ? h0 : 0));
        x.e = in && carried ? ec : POA_NEG; x.q = in && carried ? qc : POA_NEG;
        return x;
    };
    auto fetch = [&](int rr, uint2& b, int& h, int& ee, int& qq) { const RowIn x = fetch1(rr); b = x.b; h = x.h; ee = x.e; qq = x.q; };
    // H of the column in front of the pass for a row that is not the row before
    auto left_of = [&](int qr) -> int { return carried ? (int)cprev[qr] : (nw ? (int)gcol0[qr] : 0); };
    fetch(1 + lane, blk, cH, cE, cQ);
    asm volatile("" :: "v"(blk.x), "v"(blk.y), "v"(cH), "v"(cE), "v"(cQ));      // (arrived: see the end of the block loop)
    uint32_t px[CP], pf[CP], po[CP];                             // the previous row, still in registers (packed)
    int pcin = 0;
#pragma unroll
    for (int t = 0; t < CP; ++t) { px[t] = 0; pf[t] = 0; po[t] = 0; }
    // end cell within this pass, per virtual lane: value, rank (16 bits) and column offset within the half
    uint32_t bsP = sw ? 0u : 0x80008000u, brP = 0;
    uint32_t snapP[CP];                                          // the row's values in the halves where it set the best so far: the column is found from them once, behind the pass
#pragma unroll
    for (int t = 0; t < CP; ++t) snapP[t] = 0;
    int nbest = -(1 << 30), nrow = 0;                            // global mode: cells (sink row, column m), wave-uniform
    uint32_t lowP = 0x7fff7fffu;                                 // global / overlap: the lowest H of the pass (local cells are >= 0)
    const bool stores = col0 + 1 <= m;
    // (stores written out as asm with a scalar base and the lane's 32-bit offset saved the 64-bit address arithmetic and cost 2 ms:
    // the blocks pin the schedule)
    const uint32_t lane_off = (uint32_t)(col0 + 8) * 2u;
    // columns beyond the sequence, as masks (0xFFFF in those halves): for the lowest-cell watch, and for the end cell
    uint32_t beyondP[CP], endP[CP];
#pragma unroll
    for (int t = 0; t < CP; ++t) { beyondP[t] = opaque(pk_sra15(sbP[t] << 7)); endP[t] = opaque(beyond_free ? 0u : beyondP[t]); }
    for (int rb = 1; rb <= N; rb += 64) {
        uint2 nblk; int nH, nE, nQ;
        fetch(rb + 64 + lane, nblk, nH, nE, nQ);
        const int cnt = N - rb + 1 < 64 ? N - rb + 1 : 64;
        int cobH = 0, cobE = 0, cobQ = 0;
        for (int i = 0; i < cnt; ++i) {
            const int r = rb + i;
            SEC0();
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)blk.x, i);
            int cinH = 0, cinE = POA_NEG, cinQ = POA_NEG;
            if (__builtin_expect(carried | nw, 0)) cinH = __builtin_amdgcn_readlane(cH, i);
            if (__builtin_expect(carried, 0)) { cinE = __builtin_amdgcn_readlane(cE, i); cinQ = __builtin_amdgcn_readlane(cQ, i); }
            const int vb = (int)(d0 & 0xff), np = (int)((d0 >> 8) & 0xf);
            const bool sink = (d0 & 0x1000u) != 0, tolds = (d0 & 0x4000u) != 0;
            const int p0 = (int)(d0 >> 16);
            uint32_t ss[CP];
            {
                const uint32_t vb2 = dup16(vb);
#pragma unroll
                for (int t = 0; t < CP; ++t) {
                    ss[t] = pk_nz_select(sbP[t] ^ vb2, dsn2, sm2);                   // S.m where the letter equals the node's, else S.n
                }
            }
            // one source row that is not the row before, packed: H at the lane's columns, Fs, Os, and the H in front of the pass
            auto source = [&](int qr, uint32_t (&h)[CP], int& left, uint32_t (&fs)[CP], uint32_t (&os)[CP]) {
                if (qr == 0) {
#pragma unroll
                    for (int t = 0; t < CP; ++t) {
                        h[t] = pack16(row0_h(col0 + t + 1), row0_h(col0 + CP + t + 1));
                        fs[t] = pk_adds(h[t], 0xffffffffu); os[t] = fs[t];
                    }
                    left = row0_h(colbase);
                } else {
                    uint32_t dd[CP];
                    if (r - qr < RING) {
                        const uint32_t* rp = ring + (qr & rmask) * rrow + lane * CP;
#pragma unroll
                        for (int t = 0; t < CP; ++t) { h[t] = rp[t]; dd[t] = rp[64 * CP + t]; }
                        left = ringL[qr & rmask];
                    } else {
                        uint32_t nh[CP], nd[CP];
                        const uint32_t* sh = (const uint32_t*)(gH + (size_t)qr * gp * 2 + lane_off);
                        const uint32_t* sd = (const uint32_t*)(gD + (size_t)qr * gp * 2 + lane_off);
                        // compiler-visible loads: its own wait counts (an asm block here made it wait for every store in flight on the
                        // ring path too).  The cells were written by these very lanes earlier in this pass.
#pragma unroll
                        for (int t = 0; t < CP; ++t) {      // agent-scope loads: served by L2 (a line of the CU's L1 may predate the row next to it)
                            nh[t] = 0; nd[t] = 0;           // a lane entirely beyond the sequence stores nothing: it reads nothing back
                            if (stores) {
                                nh[t] = __hip_atomic_load(sh + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                nd[t] = __hip_atomic_load(sd + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                        left = left_of(qr);
                        to_packed<CP>(nh, h); to_packed<CP>(nd, dd);
                    }
#pragma unroll
                    for (int t = 0; t < CP; ++t) {
                        fs[t] = pk_adds(h[t], pk_subu(dd[t] & 0x00070007u, ONE2));
                        os[t] = pk_adds(h[t], pk_subu(pk_shr(dd[t], 3) & 0x001f001fu, ONE2));
                    }
                }
            };
            SEC(8);
            uint32_t dg[CP], MF[CP], MO[CP];
            // the common shape.  The source's values are used where they are: the registers of the row before, which a source
            // that is another row overwrites (nothing else reads them in this step) -- no copies on the way to the arithmetic
            auto one_source = [&]() {
                const uint32_t hsh0 = hand_down(px[CP - 1], pcin);
#pragma unroll
                for (int t = 0; t < CP; ++t) {
                    dg[t] = pk_adds(t == 0 ? hsh0 : px[t - 1], ss[t]);
                    MF[t] = pk_max(px[t], pf[t]); MO[t] = pk_max(px[t], po[t]);
                }
            };
            if (__builtin_expect((d0 & 0x2000u) != 0, 1)) {           // one in-edge, from the row before: said by the graph row, so that the common row tests one bit
                one_source();
            } else if (np <= 1) {
                source(np == 0 ? 0 : p0, px, pcin, pf, po);              // (also row 1 of a node without in-edges: row 0 is not in the registers)
                one_source();
            } else {
                const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)blk.y, i);
#pragma unroll
                for (int t = 0; t < CP; ++t) { dg[t] = NEG2; MF[t] = NEG2; MO[t] = NEG2; }
                auto add_source = [&](int qr) {
                    uint32_t h[CP], fs[CP], os[CP];
                    int left = pcin;
                    if (qr == r - 1) {
#pragma unroll
                        for (int t = 0; t < CP; ++t) { h[t] = px[t]; fs[t] = pf[t]; os[t] = po[t]; }
                    } else source(qr, h, left, fs, os);
                    const uint32_t hsh0 = hand_down(h[CP - 1], left);
#pragma unroll
                    for (int t = 0; t < CP; ++t) {
                        dg[t] = pk_max(dg[t], pk_adds(t == 0 ? hsh0 : h[t - 1], ss[t]));
                        MF[t] = pk_max(MF[t], pk_max(h[t], fs[t])); MO[t] = pk_max(MO[t], pk_max(h[t], os[t]));
                    }
                };
                add_source(p0);
                add_source((int)(d1 & 0xffff));
                if (np > 2) add_source((int)(d1 >> 16));
                if (np > 3) {                                    // rare: in-edges beyond the third come from HBM
                    const int vnode = __builtin_amdgcn_readfirstlane(gorder[r - 1]);
                    const int npt = np < 15 ? np : __builtin_amdgcn_readfirstlane((int)w.np[vnode]);      // (the graph row's field holds up to 15)
                    for (int s2 = 3; s2 < npt; ++s2) add_source(__builtin_amdgcn_readfirstlane(grank[poa_pred_at(w, vnode, s2)]));
                }
            }
            SEC(9);
            uint32_t M0[CP], fsn[CP], osn[CP];
#pragma unroll
            for (int t = 0; t < CP; ++t) {
                fsn[t] = pk_adds(MF[t], e2); osn[t] = pk_adds(MO[t], c2);
                M0[t] = pk_max(pk_max(dg[t], floor2), pk_max(pk_adds(MF[t], g2), pk_adds(MO[t], q2)));
            }
            // horizontal states: two prefix maxima in the gap-free frames of the two pieces
            uint32_t X[CP], Y[CP];                              // Ehat + e - g and Qhat + c - q: the states as the next column reads them
            {
                uint32_t a[CP], b[CP], pe[CP], pq[CP];
                const int leftE = cinH > cinE + e - g ? cinH : cinE + e - g;       // = E[first column of the pass] - g
                const int leftQ = cinH > cinQ + c - q ? cinH : cinQ + c - q;
#pragma unroll
                for (int t = 0; t < CP; ++t) { a[t] = pk_subs(M0[t], jeP[t]); b[t] = pk_subs(M0[t], jcP[t]); }
                scan_left_pk2<CP>(a, leftE, pe, b, leftQ, pq);
#pragma unroll
                for (int t = 0; t < CP; ++t) { X[t] = pk_adds(pe[t], jeP[t]); Y[t] = pk_adds(pq[t], jcP[t]); }
            }
            SEC(10);
            uint32_t Hf[CP], qhat[CP], Es[CP], D[CP];
#pragma unroll
            for (int t = 0; t < CP; ++t) { qhat[t] = pk_adds(Y[t], qc2); Hf[t] = pk_max(pk_max(M0[t], pk_adds(X[t], ge2)), qhat[t]); }
            const uint32_t q0 = hand_down(qhat[CP - 1], cinQ);
#pragma unroll
            for (int t = 0; t < CP; ++t) {
                Es[t] = pk_max(X[t], pk_adds(t == 0 ? q0 : qhat[t - 1], e2));       // E + e - g with E = max(Ehat, Qhat[j-1] + g), spoa's array
                const uint32_t Hm = pk_subs(Hf[t], ONE2);
                // the four fields max(x - Hm, 0) = max(x, Hm) - Hm, put together BEFORE the subtraction: in 16-bit modular arithmetic
                //   (YQ << 11) + (YE << 8) + (YO << 3) + YF - Hm * (2048 + 256 + 8 + 1)
                // is the same word -- four maxima and four multiply-adds instead of four subtractions, four maxima and three shift-ors
                // (one block: the compiler turns any C form of it into shifts and adds, and pads separate blocks with wait states)
                const uint32_t YF = pk_max(fsn[t], Hm), YO = pk_max(osn[t], Hm), YE = pk_max(Es[t], Hm), YQ = pk_max(Y[t], Hm);
                {
                    uint32_t dd_, t_;
                    asm("v_pk_mad_u16 %0, %2, 8, %3 op_sel_hi:[1,0,1]\n\tv_pk_mad_u16 %1, %4, 8, %5 op_sel_hi:[1,0,1]\n\t"
                        "v_pk_mad_u16 %0, %1, %6, %0\n\tv_pk_mad_u16 %0, %7, %8, %0"
                        : "=&v"(dd_), "=&v"(t_) : "v"(YO), "v"(YF), "v"(YQ), "v"(YE), "s"(0x01000100u), "v"(Hm), "s"(dup16(-2313)));
                    D[t] = dd_;
                }
            }
            SEC(11);
            // ---- what later rows and the back-track read ------------------------------------------------------------------
#pragma unroll
            for (int t = 0; t < CP; ++t) { px[t] = Hf[t]; pf[t] = fsn[t]; po[t] = osn[t]; }
            pcin = cinH;
            // the lanes whose columns lie within POA_BAND of the row's centre column -- or every lane, for a row that an older successor reads
            // back in full and for a short sequence: one comparison either way, the band as a scalar
            const int band = (slope16 && !(d0 & 0x8000u)) ? POA_BAND : (1 << 20);
            const int cen = (int)(((unsigned)r * (unsigned)slope16) >> 16);
            const bool st_row = stores && (unsigned)(col0 + C - (cen - band)) <= (unsigned)(2 * band + C - 1);
            if (__builtin_expect(st_row, 1)) {
                uint32_t nh[CP], nd[CP];
                to_natural<CP>(Hf, nh); to_natural<CP>(D, nd);
                // uniform plane base + ONE 32-bit offset per lane (row offset + the lane's columns; a plane is far below 4 GB): the stores' scalar-base
                // form.  (Left to a 64-bit sum the compiler pays 64-bit vector adds per plane and row.)
                const uint32_t voff = (uint32_t)r * (uint32_t)(gp * 2) + lane_off;
                typedef __attribute__((address_space(1))) char* gchar;
                typedef __attribute__((address_space(1))) uint32_t* gu32;
                gu32 dh = (gu32)((gchar)gH + (size_t)voff), dd = (gu32)((gchar)gD + (size_t)voff);
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                // the planes stream: 20 GB of them per C3 launch against ~100 MB of graph arrays that every other phase chases pointers through.
                // With the non-temporal hint the stores pass the caches without pushing those arrays out (round 5: 17.63 -> 17.51 ms on C3,
                // 76.3 -> 75.9 on C4, A/B twice on one box; POA_TEMPORAL_STORES for the A/B).  A hint about retention only: coherence is unchanged.
#ifdef POA_TEMPORAL_STORES
#define PLANE_ST(ptr, val) (*(ptr) = (val))
#else
#define PLANE_ST(ptr, val) __builtin_nontemporal_store(val, ptr)
#endif
                if constexpr (CP == 1) { PLANE_ST(dh, nh[0]); PLANE_ST(dd, nd[0]); }
                else if constexpr (CP == 2) {
                    PLANE_ST((__attribute__((address_space(1))) u32x2*)dh, (u32x2{nh[0], nh[1]})); PLANE_ST((__attribute__((address_space(1))) u32x2*)dd, (u32x2{nd[0], nd[1]}));
                } else if constexpr (CP == 3) {
                    typedef u32x3 __attribute__((aligned(4))) u32x3u;
                    PLANE_ST((__attribute__((address_space(1))) u32x3u*)dh, (u32x3{nh[0], nh[1], nh[2]})); PLANE_ST((__attribute__((address_space(1))) u32x3u*)dd, (u32x3{nd[0], nd[1], nd[2]}));
                } else {
                    PLANE_ST((__attribute__((address_space(1))) u32x4*)dh, (u32x4{nh[0], nh[1], nh[2], nh[3]})); PLANE_ST((__attribute__((address_space(1))) u32x4*)dd, (u32x4{nd[0], nd[1], nd[2], nd[3]}));
                }
#undef PLANE_ST
            }
            if (__builtin_expect(tolds, 0)) {
                uint32_t* rp = ring + (r & rmask) * rrow + lane * CP;
#pragma unroll
                for (int t = 0; t < CP; ++t) { rp[t] = Hf[t]; rp[64 * CP + t] = D[t]; }
                if (lane == 0) ringL[r & rmask] = cinH;
            }
            SEC(12);
            if (__builtin_expect(!sw, 0)) {
#pragma unroll
                for (int t = 0; t < CP; ++t) lowP = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, lowP), __builtin_bit_cast(s16x2, bfi(beyondP[t], 0x7fff7fffu, Hf[t]))));
            }
            // ---- end cell: first strict maximum in (rank, column) order --------------------------------------------
            if (__builtin_expect(sw | (!nw & sink), 1)) {
                // columns beyond the sequence (letter 0x100: bit 8) do not count.  In local mode with a negative mismatch score they cannot
                // be the end cell anyway: every move into such a cell loses score, so it is below the cell it came from
                uint32_t hv2[CP];
#pragma unroll
                for (int t = 0; t < CP; ++t) hv2[t] = SIMPLE ? Hf[t] : bfi(endP[t], 0x80008000u, Hf[t]);      // (SIMPLE: that case, promised by dp_rows)
                uint32_t rm = hv2[0];
#pragma unroll
                for (int t = 1; t < CP; ++t) rm = pk_max(rm, hv2[t]);
                const uint32_t imp = pk_lt_mask(bsP, rm);                         // halves whose best is exceeded (strictly: the first row stays)
                // no vote around the update: along an alignment nearly every row improves some lane's best, and a wave-wide vote feeding a
                // scalar branch is a VALU -> SALU hand-over that stalls the step -- a few masked moves, done unconditionally.  Round 5: the
                // improving halves keep the row's values (one v_bfi per register); which column of the virtual lane held the maximum is
                // worked out once, behind the pass, instead of in every row (four packed operations per register and row).
                bsP = pk_max(bsP, rm); brP = bfi_keep(imp, dup16(r), brP);
#pragma unroll
                for (int t = 0; t < CP; ++t) snapP[t] = bfi_keep(imp, hv2[t], snapP[t]);
            } else if (nw & sink & last) {
                uint32_t pick = 0;
#pragma unroll
                for (int t = 0; t < CP; ++t) if (t == tm) pick = (uint32_t)__builtin_amdgcn_readlane((int)Hf[t], lm);
                const int val = hm ? (int)pick >> 16 : (int)(short)(pick & 0xffffu);
                if (val > nbest) { nbest = val; nrow = r; }
            }
            if (__builtin_expect(more, 0)) {
                const int rH = (int)__builtin_amdgcn_readlane((int)Hf[CP - 1], 63) >> 16, rEs = (int)__builtin_amdgcn_readlane((int)Es[CP - 1], 63) >> 16,
                          rQ = (int)__builtin_amdgcn_readlane((int)qhat[CP - 1], 63) >> 16;
                const int rE = rEs - (e - g);
                const uint32_t li = lane_is(lane, i);
                cobH = set_lane(cobH, rH, li); cobE = set_lane(cobE, rE < POA_NEG ? POA_NEG : rE, li); cobQ = set_lane(cobQ, rQ < POA_NEG ? POA_NEG : rQ, li);
            }
            SEC(13);
            asm volatile("" ::: "memory");   // one wave: LDS operations execute in order; only the compiler must not reorder
        }
        if (more && lane < cnt) { cnext[rb + lane] = (short)cobH; cnext[cpitch + rb + lane] = (short)cobE; cnext[2 * cpitch + rb + lane] = (short)cobQ; }
        blk = nblk; cH = nH; cE = nE; cQ = nQ;
        // the next block's rows have arrived HERE: first used inside the row loop, the wait for them would sit there, and a wait for a
        // load is a wait for every store before it
        asm volatile("" :: "v"(blk.x), "v"(blk.y), "v"(cH), "v"(cE), "v"(cQ));
    }
    if (more) phase_sync();
    {   // a cell at the floor of the int16 range may have been cut off there: the caller reports it (status 6)
        const int lo2 = (int)(short)(lowP & 0xffffu), hi2 = (int)lowP >> 16;
        if (__builtin_amdgcn_ballot_w64((lo2 < hi2 ? lo2 : hi2) <= POA_NEG)) bs_io = -(1 << 29), br_io = -1;
    }
    if (br_io < 0) return;
    // best of this pass: value descending, rank ascending, column ascending -- then against the earlier passes' (their columns
    // are smaller: on equal value and rank the earlier pass stays)
    {
        uint32_t bcP = 0;                                                         // first register of the virtual lane that held its best value, per half
#pragma unroll
        for (int t = CP - 1; t >= 0; --t) {
            const uint32_t eqm = pk_subu(pk_minu(pk_subs(bsP, snapP[t]), ONE2), ONE2);    // 0xFFFF where this column holds the maximum
            bcP = bfi(eqm, dup16(t), bcP);
        }
        const int vlo = (int)(short)(bsP & 0xffffu), vhi = (int)bsP >> 16;
        const int rlo = (int)(brP & 0xffffu), rhi = (int)(brP >> 16);
        const int clo = col0 + (int)(bcP & 0xffffu) + 1, chi = col0 + CP + (int)(bcP >> 16) + 1;
        int bs = vlo, br = rlo, bc = clo;
        if (vhi > vlo || (vhi == vlo && rhi < rlo)) { bs = vhi; br = rhi; bc = chi; }
        if (br == 0) bs = -(1 << 30);                                             // nothing recorded in this virtual lane
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int s2 = __shfl_xor(bs, d), r2 = __shfl_xor(br, d), c2x = __shfl_xor(bc, d);
            if (s2 > bs || (s2 == bs && (r2 < br || (r2 == br && c2x < bc)))) { bs = s2; br = r2; bc = c2x; }
        }
        if (nw) { bs = nbest; br = nrow; bc = m; }
        if (br > 0 && (bs > bs_io || (bs == bs_io && br < br_io))) { bs_io = bs; br_io = br; bc_io = bc; }
    }
}

// DP rows of one sequence: passes of 128 * POA_MAXCP columns; the last pass takes 2, 4 or 6 columns per lane by its width (a
// row step costs a fixed part plus a part per register)
__device__ __attribute__((noinline)) void dp_rows(const PoaWs& w, const PoaScores S_, int N, int m, const int8_t* seq_, int lane, const int slope16_, int& bs_out, int& br_out, int& bc_out DBGARG)
{
    PoaScores S;
    S.algorithm = __builtin_amdgcn_readfirstlane(S_.algorithm); S.m = __builtin_amdgcn_readfirstlane(S_.m); S.n = __builtin_amdgcn_readfirstlane(S_.n);
    S.g = __builtin_amdgcn_readfirstlane(S_.g); S.e = __builtin_amdgcn_readfirstlane(S_.e); S.q = __builtin_amdgcn_readfirstlane(S_.q);
    S.c = __builtin_amdgcn_readfirstlane(S_.c); S.min_cov = 0;
    const int slope16 = __builtin_amdgcn_readfirstlane(slope16_);
    const int8_t* seq = uniform_ptr(seq_);
    constexpr int WMAX = 128 * POA_MAXCP;
    // wave-uniform, all of them: say so (arguments of a function arrive in vector registers)
    N = __builtin_amdgcn_readfirstlane(N); m = __builtin_amdgcn_readfirstlane(m);
    const int RING = poa_ring(m);
    int bs = (S.algorithm & 0xff) == 0 ? 0 : -(1 << 30), br = 0, bc = 0;
    int pass = 0;
    for (int colbase = 0; colbase < m; colbase += WMAX, ++pass) {
        const int rem = m - colbase;
        const bool more = rem > WMAX;
        const int cp = more ? POA_MAXCP : poa_cols(rem);
        const bool simple = pass == 0 && !more && (S.algorithm & 0xff) == 0 && S.n < 0 && S.g < 0 && S.e < 0 && S.q < 0 && S.c < 0;
        if (simple) {
            switch (cp) {
                case 1: dp_pass_lz<1, true>(w, S, N, m, seq, lane, pass, colbase, more, RING, slope16, bs, br, bc DBGPASS); break;
                case 2: dp_pass_lz<2, true>(w, S, N, m, seq, lane, pass, colbase, more, RING, slope16, bs, br, bc DBGPASS); break;
                default: dp_pass_lz<3, true>(w, S, N, m, seq, lane, pass, colbase, more, RING, slope16, bs, br, bc DBGPASS); break;
            }
        } else
        switch (cp) {
            case 1: dp_pass_lz<1>(w, S, N, m, seq, lane, pass, colbase, more, RING, slope16, bs, br, bc DBGPASS); break;
            case 2: dp_pass_lz<2>(w, S, N, m, seq, lane, pass, colbase, more, RING, slope16, bs, br, bc DBGPASS); break;
            case 3: dp_pass_lz<3>(w, S, N, m, seq, lane, pass, colbase, more, RING, slope16, bs, br, bc DBGPASS); break;
            default: dp_pass_lz<4>(w, S, N, m, seq, lane, pass, colbase, more, RING, slope16, bs, br, bc DBGPASS); break;
        }
        if (br < 0) break;                               // a cell left the int16 range
    }
    bs_out = bs; br_out = br; bc_out = bc;
}

// ---- the wide (32-bit) form of the pass -------------------------------------------------------------------------------------
// spoa aligns sequences of any length (its engines fall back to 32-bit cells); the packed pass above holds what fits int16: sequences
// up to POA_MAX_COPY bases with the call sites' scores.  Everything else -- a longer sequence, scores outside the 16-bit cells, a run
// of the packed pass that reported a cell at the floor of its range -- takes this form: the same lazy row formulation (tools/poa_model.py:
// dp_row), one cell per lane-operation in 32 bits, lane l owns C adjacent columns (C = 1..6, 384 columns a pass), H in a plane of
// int32, the clamped differences in the same 16-bit word as above, every cell stored, every source row that is not the row before read
// back from the planes, carries between passes and the first column of global mode in int32 arrays behind the planes.  Not tuned:
// it exists so that nothing the reference would answer is refused for its size.
static constexpr int POA_NEGW = -(1 << 29);
__host__ __device__ inline size_t poa_plane_bytes_w(int N, int m) { return (((size_t)(N + 1) * (size_t)poa_pitch(m) * 4 + 15) & ~(size_t)15) + 64; }
// planes (H int32, D uint16), 6 carry arrays and the global-mode column, all int32 of N + 2 entries
__host__ __device__ inline size_t poa_dp_bytes_w(int N, int m) { return poa_plane_bytes_w(N, m) + poa_plane_bytes(N, m) + 7 * (((size_t)(N + 2) * 4 + 15) & ~(size_t)15) + 64; }
struct PoaWide { int* planeH; unsigned short* planeD; int* carry; int cpitch; int* col0; };

template <int C>
__device__ void dp_pass_w(const PoaWs& w, const PoaWide& W, const PoaScores S, const int N, const int m, const int8_t* seq, const int lane, const int pass, const int colbase,
                          const bool more, int& bs_io, int& br_io, int& bc_io)
{
    const int gp = poa_pitch(m);
    const bool sw = (S.algorithm & 0xff) == 0, nw = (S.algorithm & 0xff) == 1;
    const int g = S.g, e = S.e, q = S.q, c = S.c;
    const int lc0 = C * lane, col0 = colbase + lc0;              // register t = column col0 + t + 1
    int sb[C], je[C], jc[C];
    bool inseq[C];
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int j = col0 + t + 1;
        inseq[t] = j <= m;
        sb[t] = inseq[t] ? (int)(uint8_t)seq[j - 1] : 0x100;
        je[t] = (lc0 + t + 1) * e; jc[t] = (lc0 + t + 1) * c;
    }
    auto row0_h = [&](int j) -> int {
        const int l1 = g + (j - 1) * e, l2 = q + (j - 1) * c;
        return (sw || j == 0) ? 0 : (l1 > l2 ? l1 : l2);
    };
    const int cpitch = W.cpitch;
    const int* cprev = W.carry + (size_t)(pass & 1) * 3 * cpitch;
    int* cnext = W.carry + (size_t)((pass + 1) & 1) * 3 * cpitch;
    const bool carried = pass > 0;
    auto left_of = [&](int qr) -> int { return carried ? cprev[qr] : (nw ? W.col0[qr] : 0); };
    int px[C], pf[C], po[C];
#pragma unroll
    for (int t = 0; t < C; ++t) { px[t] = 0; pf[t] = 0; po[t] = 0; }
    int pcin = 0;
    int bs = sw ? 0 : POA_NEGW, br = 0, bc = 0;                 // this lane's end cell: first strict maximum in (rank, column) order
    int nbest = -(1 << 30), nrow = 0;
    const int mc = m - 1 - colbase;                              // last pass: where column m lives
    const int lm = mc / C, tm = mc % C;
    const bool last = !more;
    for (int r = 1; r <= N; ++r) {
        const uint2 rb = w.ri[r];
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb.x), d1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb.y);
        const int vb = (int)(d0 & 0xff), np = (int)((d0 >> 8) & 0xf);
        const bool sink = (d0 & 0x1000u) != 0;
        const int cinH = carried ? cprev[r] : (nw ? W.col0[r] : 0);
        const int cinE = carried ? cprev[cpitch + r] : POA_NEGW, cinQ = carried ? cprev[2 * cpitch + r] : POA_NEGW;
        int ss[C], dg[C], MF[C], MO[C];
#pragma unroll
        for (int t = 0; t < C; ++t) { ss[t] = sb[t] == vb ? S.m : S.n; dg[t] = POA_NEGW; MF[t] = POA_NEGW; MO[t] = POA_NEGW; }
        auto add_source = [&](int qr) {
            int h[C], fs[C], os[C], left;
            if (qr == r - 1 && qr != 0) {
#pragma unroll
                for (int t = 0; t < C; ++t) { h[t] = px[t]; fs[t] = pf[t]; os[t] = po[t]; }
                left = pcin;
            } else if (qr == 0) {
#pragma unroll
                for (int t = 0; t < C; ++t) { h[t] = row0_h(col0 + t + 1); fs[t] = h[t] - 1; os[t] = fs[t]; }
                left = row0_h(colbase);
            } else {
#pragma unroll
                for (int t = 0; t < C; ++t) {
                    int hv = 0, dd = 0;
                    if (inseq[t]) {      // agent-scope loads: the cells were written by these lanes earlier in this pass (a line of the CU's L1 may predate them)
                        hv = __hip_atomic_load(W.planeH + (size_t)qr * gp + col0 + t + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dd = (int)__hip_atomic_load(W.planeD + (size_t)qr * gp + col0 + t + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    h[t] = hv; fs[t] = hv + (dd & 7) - 1; os[t] = hv + ((dd >> 3) & 31) - 1;
                }
                left = left_of(qr);
            }
            const int hsh0 = dpp_shr1(left, h[C - 1]);
#pragma unroll
            for (int t = 0; t < C; ++t) {
                const int x = (t == 0 ? hsh0 : h[t - 1]) + ss[t];
                dg[t] = x > dg[t] ? x : dg[t];
                const int mf = h[t] > fs[t] ? h[t] : fs[t], mo = h[t] > os[t] ? h[t] : os[t];
                MF[t] = mf > MF[t] ? mf : MF[t]; MO[t] = mo > MO[t] ? mo : MO[t];
            }
        };
        if (np == 0) add_source(0);
        else {
            add_source((int)(d0 >> 16));
            if (np > 1) add_source((int)(d1 & 0xffff));
            if (np > 2) add_source((int)(d1 >> 16));
            if (np > 3) {
                const int vnode = __builtin_amdgcn_readfirstlane(w.order[r - 1]);
                const int npt = np < 15 ? np : __builtin_amdgcn_readfirstlane((int)w.np[vnode]);
                for (int s2 = 3; s2 < npt; ++s2) add_source(__builtin_amdgcn_readfirstlane(w.rank[poa_pred_at(w, vnode, s2)]));
            }
        }
        int M0[C], fsn[C], osn[C], X[C], Y[C];
        const int floorv = sw ? 0 : POA_NEGW;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            fsn[t] = MF[t] + e; osn[t] = MO[t] + c;
            int v = dg[t] > floorv ? dg[t] : floorv;
            const int a1 = MF[t] + g, a2 = MO[t] + q;
            v = a1 > v ? a1 : v; v = a2 > v ? a2 : v;
            M0[t] = v;
        }
        {   // the two horizontal states: exclusive prefix maxima in the gap-free frames of the two pieces
            const int leftE = cinH > cinE + e - g ? cinH : cinE + e - g;
            const int leftQ = cinH > cinQ + c - q ? cinH : cinQ + c - q;
            int runA = -(1 << 30), runB = -(1 << 30), pe[C], pq[C];
#pragma unroll
            for (int t = 0; t < C; ++t) {
                pe[t] = runA; pq[t] = runB;
                const int a = M0[t] - je[t], b = M0[t] - jc[t];
                runA = a > runA ? a : runA; runB = b > runB ? b : runB;
            }
            int excA = dpp_shr1(leftE, runA), excB = dpp_shr1(leftQ, runB);
            wave_prefix_max2(excA, excB);
#pragma unroll
            for (int t = 0; t < C; ++t) { X[t] = (pe[t] > excA ? pe[t] : excA) + je[t]; Y[t] = (pq[t] > excB ? pq[t] : excB) + jc[t]; }
        }
        int Hf[C], qhat[C], Es[C];
#pragma unroll
        for (int t = 0; t < C; ++t) {
            qhat[t] = Y[t] + (q - c);
            int v = M0[t];
            const int xe = X[t] + (g - e);
            v = xe > v ? xe : v; v = qhat[t] > v ? qhat[t] : v;
            Hf[t] = v;
        }
        const int q0 = dpp_shr1(cinQ, qhat[C - 1]);
#pragma unroll
        for (int t = 0; t < C; ++t) {
            const int qe = (t == 0 ? q0 : qhat[t - 1]) + e;
            Es[t] = X[t] > qe ? X[t] : qe;
            if (inseq[t]) {
                const int Hm = Hf[t] - 1;
                int dF = fsn[t] - Hm, dO = osn[t] - Hm, dE = Es[t] - Hm, dQ = Y[t] - Hm;
                dF = dF < 0 ? 0 : dF; dO = dO < 0 ? 0 : dO; dE = dE < 0 ? 0 : dE; dQ = dQ < 0 ? 0 : dQ;
                W.planeH[(size_t)r * gp + col0 + t + 8] = Hf[t];
                W.planeD[(size_t)r * gp + col0 + t + 8] = (unsigned short)((dQ << 11) | (dE << 8) | (dO << 3) | dF);
            }
        }
#pragma unroll
        for (int t = 0; t < C; ++t) { px[t] = Hf[t]; pf[t] = fsn[t]; po[t] = osn[t]; }
        pcin = cinH;
        if (sw | (!nw & sink)) {
#pragma unroll
            for (int t = 0; t < C; ++t) if (inseq[t] && Hf[t] > bs) { bs = Hf[t]; br = r; bc = col0 + t + 1; }
        } else if (nw & sink & last) {
            int pick = 0;
#pragma unroll
            for (int t = 0; t < C; ++t) if (t == tm) pick = __builtin_amdgcn_readlane(Hf[t], lm);
            if (pick > nbest) { nbest = pick; nrow = r; }
        }
        if (more) {
            const int rH = __builtin_amdgcn_readlane(Hf[C - 1], 63), rEs = __builtin_amdgcn_readlane(Es[C - 1], 63), rQ = __builtin_amdgcn_readlane(qhat[C - 1], 63);
            const int rE = rEs - (e - g);
            if (lane == 0) { cnext[r] = rH; cnext[cpitch + r] = rE < POA_NEGW ? POA_NEGW : rE; cnext[2 * cpitch + r] = rQ < POA_NEGW ? POA_NEGW : rQ; }
        }
    }
    if (more) phase_sync();
    if (br == 0) bs = -(1 << 30);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int s2 = __shfl_xor(bs, d), r2 = __shfl_xor(br, d), c2x = __shfl_xor(bc, d);
        if (s2 > bs || (s2 == bs && (r2 < br || (r2 == br && c2x < bc)))) { bs = s2; br = r2; bc = c2x; }
    }
    if (nw) { bs = nbest; br = nrow; bc = m; }
    if (br > 0 && (bs > bs_io || (bs == bs_io && br < br_io))) { bs_io = bs; br_io = br; bc_io = bc; }
}

__device__ __attribute__((noinline)) void dp_rows_w(const PoaWs& w, const PoaWide W, const PoaScores S_, int N, int m, const int8_t* seq_, int lane, int& bs_out, int& br_out, int& bc_out)
{
    PoaScores S;
    S.algorithm = __builtin_amdgcn_readfirstlane(S_.algorithm); S.m = __builtin_amdgcn_readfirstlane(S_.m); S.n = __builtin_amdgcn_readfirstlane(S_.n);
    S.g = __builtin_amdgcn_readfirstlane(S_.g); S.e = __builtin_amdgcn_readfirstlane(S_.e); S.q = __builtin_amdgcn_readfirstlane(S_.q);
    S.c = __builtin_amdgcn_readfirstlane(S_.c); S.min_cov = 0;
    const int8_t* seq = uniform_ptr(seq_);
    N = __builtin_amdgcn_readfirstlane(N); m = __builtin_amdgcn_readfirstlane(m);
    constexpr int WMAX = 64 * 6;
    int bs = (S.algorithm & 0xff) == 0 ? 0 : -(1 << 30), br = 0, bc = 0;
    int pass = 0;
    for (int colbase = 0; colbase < m; colbase += WMAX, ++pass) {
        const int rem = m - colbase;
        const bool more = rem > WMAX;
        const int cpl = more ? 6 : (rem + 63) / 64;
        switch (cpl) {
            case 1: dp_pass_w<1>(w, W, S, N, m, seq, lane, pass, colbase, more, bs, br, bc); break;
            case 2: dp_pass_w<2>(w, W, S, N, m, seq, lane, pass, colbase, more, bs, br, bc); break;
            case 3: dp_pass_w<3>(w, W, S, N, m, seq, lane, pass, colbase, more, bs, br, bc); break;
            case 4: dp_pass_w<4>(w, W, S, N, m, seq, lane, pass, colbase, more, bs, br, bc); break;
            case 5: dp_pass_w<5>(w, W, S, N, m, seq, lane, pass, colbase, more, bs, br, bc); break;
            default: dp_pass_w<6>(w, W, S, N, m, seq, lane, pass, colbase, more, bs, br, bc); break;
        }
    }
    bs_out = bs; br_out = br; bc_out = bc;
}

// Graph::TopologicalSort -- the order spoa's sequential depth-first search produces (oracle/poa_oracle.c: topo_sort), from
// independent pieces (tools/poa_model.py: topo_sort states and tests the derivation on the CPU):
//  1. root[x] = the smallest node id among everything that depends on x (descendants over the edges, the members of their
//     aligned sets, and so on).  spoa's outer loop takes the ids in ascending order and a visit emits exactly the unfinished
//     ancestors of that id, so x is emitted during the visit of id root[x].  The values of the graph before this sequence
//     are kept; the new path lowers them by a suffix minimum along the path, the rest is relaxation to the fixed point.
//  2. every id r with root[r] == r starts one search over the nodes with root == r, independent of all others (a dependency
//     with a smaller root is finished by then, one with a larger root cannot occur); its nodes go behind the nodes of all
//     smaller roots.  One search per LANE; frames (node | cursor << 16) grow down from the end of the root's own stretch of
//     `order`, emitted nodes up from its start (an unfinished node is in neither, so they never meet).
// st[v]: bit 0 finished, bit 1 "ignored" (pushed as a member of an aligned set: emitted with the member the search met first).
__device__ void poa_dfs_root(const PoaWs& w, const int r, const int lo, const int size)
{
    const int hi = lo + size;
    int sp = hi, k = lo;
    w.order[--sp] = r;
    while (sp < hi) {
        const uint32_t fr = (uint32_t)w.order[sp];
        const int v = (int)(fr & 0xffffu);
        int cur = (int)(fr >> 16);
        const int npv = w.np[v], nav = w.na[v];
        const bool ign = (w.st[v] & 2) != 0;
        if (cur == 0 && !ign) for (int t = 0; t < nav; ++t) w.st[w.aligned[v * POA_MAXA + t]] |= 2;
        // the order spoa's stack hands the dependencies out: aligned members last to first, then in-edge tails last to first
        const int na2 = ign ? 0 : nav, nd = na2 + npv;
        int nxt = -1;
        while (cur < nd) {
            const int d = cur < na2 ? w.aligned[v * POA_MAXA + na2 - 1 - cur] : poa_pred_at(w, v, npv - 1 - (cur - na2));
            ++cur;
            if (w.root[d] == r && !(w.st[d] & 1)) { nxt = d; break; }
        }
        if (nxt >= 0) { w.order[sp] = (int)((uint32_t)v | ((uint32_t)cur << 16)); w.order[--sp] = nxt; continue; }
        w.st[v] |= 1;
        ++sp;
        if (!ign) {
            w.order[k++] = v;
            for (int t = 0; t < nav; ++t) w.order[k++] = w.aligned[v * POA_MAXA + t];
        }
    }
}

// sort the graph after a sequence of m bases (nodes w.pj[0..m)) has been fused; nodes [n_old, n) are new.  returns 0, or -1 if
// the order violates an edge (cannot happen; checked because everything downstream relies on it)
__device__ __forceinline__ int poa_sort(const PoaWs& w, const int n_old, const int n, const int m, const int lane)
{
    for (int v = n_old + lane; v < n; v += 64) w.root[v] = v;
    for (int v = lane; v < n; v += 64) { w.rsz[v] = 0; w.st[v] = 0; }
    phase_sync();
    {   // everything behind a base of the new path depends on it: suffix minimum of root along the path
        int smin = 0x7fffffff;
        for (int i0 = ((m - 1) / 64) * 64; i0 >= 0; i0 -= 64) {
            const int i = i0 + lane;
            const int x = i < m ? w.pj[i] : -1;
            int val = x >= 0 ? w.root[x] : 0x7fffffff;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_down(val, d); if (lane + d < 64) val = o < val ? o : val; }
            val = val < smin ? val : smin;
            if (x >= 0) atomicMin(&w.root[x], val);
            smin = __builtin_amdgcn_readlane(val, 0);
        }
    }
    phase_sync();
    for (int sweep = 0; sweep <= n; ++sweep) {            // relaxation; a sweep that changes nothing ends it
        int changed = 0;
        for (int v = lane; v < n; v += 64) {
            const int r0 = w.root[v];
            int rv = r0;
            const int nav = w.na[v], npv = w.np[v];
            for (int t = 0; t < nav; ++t) { const int ra = w.root[w.aligned[v * POA_MAXA + t]]; rv = ra < rv ? ra : rv; }
            if (rv < r0) { atomicMin(&w.root[v], rv); changed = 1; }
            for (int t = 0; t < npv; ++t) { const int u = poa_pred_at(w, v, t); if (rv < w.root[u]) { atomicMin(&w.root[u], rv); changed = 1; } }
        }
        phase_sync();
        if (!__builtin_amdgcn_ballot_w64(changed != 0)) break;
    }
    for (int v = lane; v < n; v += 64) atomicAdd(&w.rsz[w.root[v]], 1u);
    phase_sync();
    int acc = 0, nnt = 0;                                  // positions before the current block of ids; searches collected so far
    for (int r0 = 0; r0 < n; r0 += 64) {
        const int r = r0 + lane;
        const int sz = r < n ? (int)w.rsz[r] : 0;
        int inc = sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        const int base = acc + inc - sz;
        if (sz == 1) w.order[base] = r;
        const bool nt = sz > 1;
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(nt);
        if (nt) { w.rsz[r] = (uint32_t)base; w.ntr[nnt + __builtin_popcountll(bm & (((unsigned long long)1 << lane) - 1))] = r | (sz << 16); }
        nnt += __builtin_popcountll(bm);
        acc += __builtin_amdgcn_readlane(inc, 63);
    }
    phase_sync();
    for (int t = lane; t < nnt; t += 64) { const int e = w.ntr[t]; const int r = e & 0xffff; poa_dfs_root(w, r, (int)w.rsz[r], (int)((uint32_t)e >> 16)); }
    phase_sync();
    for (int i = lane; i < n; i += 64) w.rank[w.order[i]] = i + 1;
    phase_sync();
    int bad = 0;
    for (int v = lane; v < n; v += 64) {
        const int r = w.rank[v];
        const int np = w.np[v];
        for (int s2 = 0; s2 < np; ++s2) bad |= (w.rank[poa_pred_at(w, v, s2)] >= r);
    }
    return __builtin_amdgcn_ballot_w64(bad != 0) ? -1 : 0;
}

// The same sort with its arrays in LDS (graphs of up to POA_SORT_LDS nodes: the common case): root (32 bits, LDS atomics), the
// number of nodes per root, the order and the per-node search state (bits 0-1 as st[], bits 2-6 the cursor of the node's frame --
// a node is on the stack of its search at most once, so the frames are the node ids alone).  A lane runs the searches of the
// roots it meets in its own stride of ids, block by block.  The graph's lists are read from HBM.
static constexpr int POA_SORT_LDS = POA_LDS_BYTES / 10 - 2;   // 9 bytes per node + a list of up to n / 2 roots
// wave-wide inclusive prefix sum (DPP: row_shr within the 16-lane rows, then the row totals)
__device__ __forceinline__ int wave_prefix_sum(int v) {
    asm volatile("s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return v;
}
// wave-wide inclusive SUFFIX minimum of unsigned values (lane l: minimum over lanes l..63): the prefix form on the mirrored wave
__device__ __forceinline__ uint32_t wave_suffix_min(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_down((int)v, d); v = (threadIdx.x & 63) + d < 64 && o < v ? o : v; }
    return v;
}

__device__ __forceinline__ int poa_sort_lds(const PoaWs& w_, const int n_old, const int n, const int m, const int lane)
{
    struct { int32_t* root; int32_t* pj; int8_t* na; int8_t* np; int32_t* aligned; int32_t* pred; int32_t* order; int32_t* rank; } w;
    w.root = uniform_ptr(w_.root); w.pj = uniform_ptr(w_.pj); w.na = uniform_ptr(w_.na); w.np = uniform_ptr(w_.np);
    w.aligned = uniform_ptr(w_.aligned); w.pred = uniform_ptr(w_.pred); w.order = uniform_ptr(w_.order); w.rank = uniform_ptr(w_.rank);
    uint32_t* lroot = poa_lds;                         // low 16 bits: the root; bits 16..31 of a root's own entry: its number of nodes (set late)
    uint16_t* lsz = (uint16_t*)(lroot + n);            // nodes per root, then the root's first position in the order
    uint16_t* lord = lsz + n;
    uint8_t* lst = (uint8_t*)(lord + n);
    uint16_t* lnt = (uint16_t*)(poa_lds + (POA_LDS_BYTES - n) / 4);      // roots with more than one node: the last n bytes of the block
    // (Tried: the one dependency of a node with one in-edge and no aligned node kept in LDS, so that the relaxation rounds and the
    // searches touch HBM only for the other nodes -- no gain on C3, +0.6 ms on C4: dropped.)
    // lst bit 2: the node is new or its root has been lowered -- its aligned set and its in-edge tails may have to follow.
    // Loads of four strides are issued together (one round trip per 256 nodes instead of four)
    for (int v0 = 0; v0 < n; v0 += 256) {
        int32_t old[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int v = v0 + 64 * u + lane; old[u] = v < n_old ? w.root[v] : v; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int v = v0 + 64 * u + lane; if (v < n) { lroot[v] = (uint32_t)old[u]; lsz[v] = 0; lst[v] = v < n_old ? 0 : 4; } }
    }
    __syncthreads();
    {
        uint32_t smin = 0xffffffffu;
        for (int i1 = ((m - 1) / 64) * 64; i1 >= 0; i1 -= 256) {
            int x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i1 - 64 * u + lane; x[u] = i >= 0 && i < m ? w.pj[i] : -1; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i1 - 64 * u < 0) break;
                uint32_t val = wave_suffix_min(x[u] >= 0 ? lroot[x[u]] : 0xffffffffu);
                val = val < smin ? val : smin;
                if (x[u] >= 0 && val < lroot[x[u]]) { lroot[x[u]] = val; lst[x[u]] |= 4; }     // the nodes of a path are distinct
                smin = (uint32_t)__builtin_amdgcn_readlane((int)val, 0);
            }
        }
    }
    __syncthreads();
    // relaxation to the fixed point, frontier by frontier: before this sequence every edge and aligned set was consistent (root of
    // a tail <= root of its head, one root per set) and the path scan keeps the path's own edges so; what can be off is around
    // the flagged nodes only
    for (int round = 0; round <= n; ++round) {
        int nf = 0;
        for (int v0 = 0; v0 < n; v0 += 64) {
            const int v = v0 + lane;
            const bool f = v < n && (lst[v] & 4) != 0;
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(f);
            if (f) { lst[v] &= (uint8_t)~4; lord[nf + __builtin_popcountll(bm & (((unsigned long long)1 << lane) - 1))] = (uint16_t)v; }
            nf += __builtin_popcountll(bm);
        }
        if (nf == 0) break;
        __syncthreads();
        for (int t = lane; t < nf; t += 64) {
            const int v = lord[t];
            const int nav = w.na[v], npv = w.np[v];
            uint32_t rv = lroot[v];
            int al[POA_MAXA];
#pragma unroll
            for (int k = 0; k < POA_MAXA; ++k) { al[k] = k < nav ? w.aligned[v * POA_MAXA + k] : -1; if (al[k] >= 0) { const uint32_t ra = lroot[al[k]]; rv = ra < rv ? ra : rv; } }
            if (rv < lroot[v]) atomicMin(&lroot[v], rv);
#pragma unroll
            for (int k = 0; k < POA_MAXA; ++k) if (al[k] >= 0 && rv < lroot[al[k]]) { atomicMin(&lroot[al[k]], rv); lst[al[k]] |= 4; }
            for (int k = 0; k < npv; ++k) { const int u = poa_pred_at(w_, v, k); if (rv < lroot[u]) { atomicMin(&lroot[u], rv); lst[u] |= 4; } }
        }
        __syncthreads();
    }
    for (int v = lane; v < n; v += 64) {
        const uint32_t rt = lroot[v];
        w.root[v] = (int32_t)rt;
        // 16-bit counters, two per LDS word: count through the word that holds the counter
        atomicAdd((uint32_t*)lsz + (rt >> 1), (rt & 1) ? 0x10000u : 1u);
    }
    __syncthreads();
    // positions: a root's nodes go behind the nodes of all smaller roots.  One pass gives every root its first position, writes the
    // roots that are alone and collects the others; their searches then run ALL AT ONCE, one per lane
    int acc = 0, nnt = 0, bad = 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
        const int r = r0 + lane;
        const int sz = r < n ? (int)lsz[r] : 0;
        const int inc = wave_prefix_sum(sz);
        const int lo = acc + inc - sz;
        acc += __builtin_amdgcn_readlane(inc, 63);
        if (sz == 1) lord[lo] = (uint16_t)r;
        const bool nt = sz > 1;
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(nt);
        if (nt) { lsz[r] = (uint16_t)lo; lroot[r] |= (uint32_t)sz << 16; lnt[nnt + __builtin_popcountll(bm & (((unsigned long long)1 << lane) - 1))] = (uint16_t)r; }
        nnt += __builtin_popcountll(bm);
    }
    __syncthreads();
    for (int t = lane; t < nnt; t += 64) {
        const int r = lnt[t];
        const int lo = lsz[r], hi = lo + (int)(lroot[r] >> 16);
        int sp = hi, k = lo;
        lord[--sp] = (uint16_t)r;
        while (sp < hi) {
            const int v = lord[sp];
            const int stv = lst[v];
            int cur = stv >> 2;
            const int npv = w.np[v], nav = w.na[v];
            const bool ign = (stv & 2) != 0;
            if (cur == 0 && !ign) for (int u = 0; u < nav; ++u) lst[w.aligned[v * POA_MAXA + u]] |= 2;
            const int na2 = ign ? 0 : nav, nd = na2 + npv;
            int nxt = -1;
            while (cur < nd) {
                const int d = cur < na2 ? w.aligned[v * POA_MAXA + na2 - 1 - cur] : poa_pred_at(w_, v, npv - 1 - (cur - na2));
                ++cur;
                if ((lroot[d] & 0xffffu) == (uint32_t)r && !(lst[d] & 1)) { nxt = d; break; }
            }
            if (nxt >= 0) {
                lst[v] = (uint8_t)((stv & 3) | (cur << 2));
                if (sp - 1 < k) { bad = 1; break; }
                lord[--sp] = (uint16_t)nxt;
                continue;
            }
            lst[v] = (uint8_t)(stv | 1);
            ++sp;
            if (!ign) {
                if (k + 1 + nav > hi) { bad = 1; break; }
                lord[k++] = (uint16_t)v;
                for (int u = 0; u < nav; ++u) lord[k++] = (uint16_t)w.aligned[v * POA_MAXA + u];
            }
        }
        bad |= k != hi;
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) { const int v = lord[i]; w.order[i] = v; w.rank[v] = i + 1; }
    phase_sync();
    return __builtin_amdgcn_ballot_w64(bad != 0) || acc != n ? -1 : 0;
}

// ranks of the first and the last member of the aligned set of node v
__device__ __forceinline__ void group_span(const PoaWs& w, int v, int& lo, int& hi)
{
    lo = hi = w.rank[v];
    const int nav = w.na[v];
    for (int t = 0; t < nav; ++t) {
        const int rr = w.rank[w.aligned[v * POA_MAXA + t]];
        lo = rr < lo ? rr : lo; hi = rr > hi ? rr : hi;
    }
}

// Back-track of the lazy formulation (tools/poa_model.py: backtrack_lazy; oracle/poa_oracle.c: align_gotoh's loop).  spoa walks
// from the end cell and at every cell takes the first of its value tests that holds: diagonal through the in-edges in order,
// then per in-edge F+e / H+g / O+c / H+q, then E+e / H+g / Q+c / H+q to the left; runs of gap steps continue by the tests of its
// two inner loops.  Every value those tests read follows from H and the clamped differences of the cell itself, of its sources'
// cells in the same and the previous column, and of its left neighbour.
//  * The cells around the path are staged in LDS: rows r0-63 .. r0 (lane k holds row r0-k), 24 columns of each around the
//    diagonal through the anchor cell, both planes; the sequence's letters too.  A cell outside the band is read from HBM.
//  * Most steps are "diagonal through the first in-edge, which is the row before": lanes a, a+1, ... test that for the cells
//    (r-l, j-l) at once, the length of the run of successes is taken in one step.
//  * Any other step is evaluated with the in-edges across the lanes (first hit in in-edge order = lowest lane).
// returns the column the walk ends in (the bases in front of it are not part of the alignment), or -1 (guard: corrupt planes).
static constexpr int BT_W = POA_LDS_BYTES >= 9216 ? 24 : 20;     // columns per band row (64 rows x 2 planes + the sequence in the LDS block)
static constexpr int BT_SQ = 128, BT_RING = 1024;    // letters staged with the band; entries of the result ring
static constexpr int BT_DRIFT = BT_W / 2 - 4;        // how far the walk may leave the band's diagonal before the band is staged again
static_assert(4 * 64 * BT_W + BT_SQ + 2 * BT_RING <= POA_LDS_BYTES, "back-track LDS");
static constexpr int BT_W32 = POA_LDS_BYTES >= 9216 ? 16 : 14;                  // the wide form (int32 H): 6 bytes per staged cell
static_assert(6 * 64 * BT_W32 + BT_SQ + 2 * BT_RING <= POA_LDS_BYTES, "back-track LDS, wide form");
// the step out of a cell as a lane decides it: bits 0-2 what; BTC_ON: the gap run goes on (vertical: taken by F + e / O + c, horizontal: by E + e / Q + c)
static constexpr int BTC_EVAL = 0, BTC_DIAG = 1, BTC_VERT = 2, BTC_LEFT = 3, BTC_STOP = 4, BTC_MISS = 5, BTC_BAD = 6, BTC_ON = 32;
// A function of its own, NOT inlined: inside the kernel's one big body the register allocator spilled a value of this loop and
// reloaded it every iteration -- and the wait for that reload is a wait for every store in flight, i.e. for the walk's own result
// stores to reach memory: 2 us per step.  With its own frame the loop keeps its registers.
struct BtArgs { int32_t* pn; void* planeH; unsigned short* planeD; uint2* ri; void* col0; int32_t* rank; int32_t* pred; int32_t* order; const int8_t* seq; int N, m, r, j, slope16; const int32_t* ovn; const int32_t* ovp; const int32_t* ovc; const int8_t* np; };
// returns 2 * (column the walk ends in) + (1 if the alignment holds a step), -1 (guard) or BT_MISS (a cell outside the band the planes hold)
// HT = short: the planes of the packed pass; HT = int: the wide form (BW columns per band row)
template <typename HT, int BW>
__device__ __attribute__((noinline)) int poa_backtrack(const BtArgs A, const PoaScores S_ DBGARG)
{
    PoaScores S;                                             // (arguments arrive in vector registers; a test on one is a divergent branch)
    S.algorithm = __builtin_amdgcn_readfirstlane(S_.algorithm); S.m = __builtin_amdgcn_readfirstlane(S_.m); S.n = __builtin_amdgcn_readfirstlane(S_.n);
    S.g = __builtin_amdgcn_readfirstlane(S_.g); S.e = __builtin_amdgcn_readfirstlane(S_.e); S.q = __builtin_amdgcn_readfirstlane(S_.q);
    S.c = __builtin_amdgcn_readfirstlane(S_.c); S.min_cov = 0;
    // the walk is wave-uniform: say so (scalar registers, scalar branches)
    constexpr int BT_W = BW, BT_DRIFT = BW / 2 - 4;
    constexpr int POA_H_NONE = sizeof(HT) == 2 ? -32768 : (int)0x80000000;      // (shadows the packed form's constant: a staged cell the planes do not hold)
    struct { int32_t* pn; HT* planeH; unsigned short* planeD; uint2* ri; HT* col0; int32_t* rank; int32_t* pred; int32_t* order; } w;
    w.pn = uniform_ptr(A.pn); w.planeH = uniform_ptr((HT*)A.planeH); w.planeD = uniform_ptr(A.planeD); w.ri = uniform_ptr(A.ri);
    w.col0 = uniform_ptr((HT*)A.col0); w.rank = uniform_ptr(A.rank); w.pred = uniform_ptr(A.pred); w.order = uniform_ptr(A.order);
    const int8_t* seq = uniform_ptr(A.seq);
    const int lane = threadIdx.x & 63;
    const int N = __builtin_amdgcn_readfirstlane(A.N), m = __builtin_amdgcn_readfirstlane(A.m), slope16 = __builtin_amdgcn_readfirstlane(A.slope16);
    int r = __builtin_amdgcn_readfirstlane(A.r), j = __builtin_amdgcn_readfirstlane(A.j);
    bool moved = false;
    const bool sw = S.algorithm == 0, nw = S.algorithm == 1;
    const int g = S.g, e = S.e, q = S.q, c = S.c;
    const int gp = poa_pitch(m);
#ifdef CLH_DEBUG_POA
    const unsigned long long t_bt0 = __builtin_amdgcn_s_memtime();
#endif
    // (In a function that is not a kernel the address of the dynamic LDS array is a table look-up in memory, which the compiler repeats
    // where it is used.  Handing the address in as an argument removes the look-ups and measured 2.6 ms SLOWER on C3: through a pointer
    // made from an integer the compiler no longer tells the band, the letters and the ring apart and serialises their accesses.)
    uint32_t* const lds = poa_lds;
    HT* Hb = (HT*)lds;
    unsigned short* Db = (unsigned short*)(Hb + 64 * BT_W);
    // The walk's results (pn[column] = rank) go to a ring in LDS and from there to HBM in blocks: a store to HBM inside the loop stays in
    // flight for a microsecond, and any wait the compiler places in the loop for whatever reason (a register about to be reused by a
    // rare path's load is enough) then waits for it -- once per step.  The sequence's letters around the band are staged with it.
    uint8_t* lsq = (uint8_t*)(Db + 64 * BT_W);                   // letters of columns sb0 + 1 .. sb0 + BT_SQ
    unsigned short* lpn = (unsigned short*)(lsq + BT_SQ);           // ring: entry t & (BT_RING - 1) = pn[t]; 0 = none
    for (int i = lane; i < BT_RING / 2; i += 64) ((uint32_t*)lpn)[i] = 0;
    int jflush = j, sb0 = 0;                                        // pn[t] for t in [j, jflush) is in the ring
    auto flush = [&](int jlo) {
        // entries exist only for the columns of diagonal steps, all of them within BT_RING of jflush: a long horizontal run moves j far
        // below without writing (what lies under the ring's span keeps the zeros pn was cleared to)
        const int lo = jlo > jflush - BT_RING ? jlo : jflush - BT_RING;
        for (int t = lo + lane; t < jflush; t += 64) { w.pn[t] = (int32_t)lpn[t & (BT_RING - 1)]; lpn[t & (BT_RING - 1)] = 0; }
        jflush = jlo;
    };
    auto row0_h = [&](int jj) -> int {
        const int l1 = g + (jj - 1) * e, l2 = q + (jj - 1) * c;
        return (sw || jj == 0) ? 0 : (l1 > l2 ? l1 : l2);
    };
    int r0 = -(1 << 20), j0 = 0;
    uint2 rim = make_uint2(0, 0);
    int csk = 0;                                             // first column of this lane's band row
    auto cs_at = [&](int k) -> int { const int x = j0 - k - BT_W / 2; return (x & 1) ? x : x - 1; };    // first column of band row k: odd = a dword boundary of the planes
    // Stage the band around (rr0, jj0).  Row 0 (no node) and column 0 are not in the planes: their cells are written into the band
    // here, so that the walk reads every cell the same way.
    auto reload = [&](int rr0, int jj0) {
        r0 = rr0; j0 = jj0;
        if (jflush - jj0 >= BT_RING / 2) flush(jj0);
        sb0 = jj0 - BT_SQ + 28;
        {
            const int i = sb0 + 2 * lane;
            const int b0 = i >= 0 && i < m ? (int)(uint8_t)seq[i] : 0, b1 = i + 1 >= 0 && i + 1 < m ? (int)(uint8_t)seq[i + 1] : 0;
            ((unsigned short*)lsq)[lane] = (unsigned short)(b0 | (b1 << 8));
        }
        const int rr = r0 - lane;
        rim = make_uint2(0, 0);
        csk = cs_at(lane);
        if (rr >= 1) {
            rim = w.ri[rr];
            if (csk + BT_W > 1) {
                uint32_t t[BT_W * sizeof(HT) / 4];
                __builtin_memcpy(t, (const uint32_t*)(w.planeH + (size_t)rr * gp + csk + 7), BT_W * sizeof(HT));
                __builtin_memcpy(Hb + lane * BT_W, t, BT_W * sizeof(HT));
                __builtin_memcpy(t, (const uint32_t*)(w.planeD + (size_t)rr * gp + csk + 7), BT_W * 2);
                __builtin_memcpy(Db + lane * BT_W, t, BT_W * 2);
                if (slope16 && !(rim.x & 0x8000u)) {        // cells of this row that the planes do not hold (POA_BAND): marked, so that using one is seen
                    const int cen = (int)(((unsigned)rr * (unsigned)slope16) >> 16);
                    const int xlo = cen - POA_BAND - csk, xhi = cen + POA_BAND - csk;      // band row indices of the first and last cell held
                    for (int x = 0; x < xlo && x < BT_W; ++x) Hb[lane * BT_W + x] = (HT)POA_H_NONE;
                    for (int x = xhi + 1 > 0 ? xhi + 1 : 0; x < BT_W; ++x) Hb[lane * BT_W + x] = (HT)POA_H_NONE;
                }
            }
        }
        if (r0 < 64) {                                       // the block reaches row 0
            if (rr == 0) for (int x = 0; x < BT_W; ++x) { Hb[lane * BT_W + x] = (HT)(csk + x >= 0 ? row0_h(csk + x) : 0); Db[lane * BT_W + x] = 0; }
        }
        if (j0 - BT_W / 2 - 64 <= 0) {                       // some window reaches column 0
            if (rr >= 1 && csk <= 0 && csk + BT_W > 0) { Hb[lane * BT_W - csk] = (HT)(nw ? (int)w.col0[rr] : 0); Db[lane * BT_W - csk] = 0; }
        }
        __syncthreads();
    };
    // a cell of the planes: from the band if it is there, else from HBM
    auto Hat = [&](int rr, int jj) -> int {
        const int k = r0 - rr, x = jj - cs_at(k);
        if ((unsigned)k < 64u && (unsigned)x < (unsigned)BT_W) return (int)Hb[k * BT_W + x];
        if (rr == 0) return row0_h(jj);
        if (jj == 0) return nw ? (int)w.col0[rr] : 0;
        if (slope16) {                                       // outside the staged band: is the cell in the planes at all?
            const int cen = (int)(((unsigned)rr * (unsigned)slope16) >> 16);
            if ((jj < cen - POA_BAND || jj > cen + POA_BAND) && !(w.ri[rr].x & 0x8000u)) return POA_H_NONE;
        }
        return (int)w.planeH[(size_t)rr * gp + jj + 7];
    };
    auto Dat = [&](int rr, int jj) -> int {
        const int k = r0 - rr, x = jj - cs_at(k);
        if ((unsigned)k < 64u && (unsigned)x < (unsigned)BT_W) return (int)Db[k * BT_W + x];
        if (rr == 0 || jj == 0) return 0;
        return (int)w.planeD[(size_t)rr * gp + jj + 7];
    };
    // the graph row of rank rr (wave-uniform): from the staged block or from HBM
    auto meta = [&](int rr, uint32_t& d0, uint32_t& d1) {
        const int k = r0 - rr;
        if ((unsigned)k < 64u) { d0 = (uint32_t)__builtin_amdgcn_readlane((int)rim.x, k); d1 = (uint32_t)__builtin_amdgcn_readlane((int)rim.y, k); }
        else { const uint2 t = w.ri[rr]; d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.x); d1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.y); }
    };
    // rank of in-edge `s` of the row with graph row (d0, d1), per lane
    auto pred_of = [&](int rr, uint32_t d0, uint32_t d1, int s) -> int {
        const int np = (int)((d0 >> 8) & 0xf);
        if (np == 0) return 0;
        if (s == 0) return (int)(d0 >> 16);
        if (s == 1) return (int)(d1 & 0xffff);
        if (s == 2) return (int)(d1 >> 16);
        const int v = w.order[rr - 1];
        if (s < POA_MAXP) return w.rank[w.pred[v * POA_MAXP + s]];
        int k = s - POA_MAXP, hit = 0;                        // the (s - POA_MAXP)-th entry of v in the overflow table (poa_pred_at)
        const int no = A.ovc[0];
        for (int i = 0; i < no; ++i) if (A.ovn[i] == v) { if (k == 0) { hit = i; break; } --k; }
        return w.rank[A.ovp[hit]];
    };
    __syncthreads();
#ifdef CLH_DEBUG_POA
    tacc[15] += __builtin_amdgcn_s_memtime() - t_bt0;
#endif
    int guard = 2 * (N + m) + 64;                           // every step lowers r or j: a longer walk means corrupt planes
    // a vertical step taken by F + e or O + c: the run goes on upwards from row r (column j) until a row where the gap opened.
    // returns 0, -1 (guard) or BT_MISS
    auto up_run = [&]() -> int {
        while (r > 0) {
            if (--guard < 0) return -1;
            uint32_t e0, e1;
            meta(r, e0, e1);
            int np2 = (int)((e0 >> 8) & 0xf);
            if (np2 == 15) np2 = __builtin_amdgcn_readfirstlane((int)A.np[w.order[r - 1]]);
            const int npp2 = np2 ? np2 : 1;
            const bool act2 = lane < npp2;
            int ps2 = 0, xf = -(1 << 30), xo = -(1 << 30), hp2 = 0, fs2 = 0, os2 = 0, miss = 0;
            if (act2) {
                ps2 = pred_of(r, e0, e1, lane); hp2 = Hat(ps2, j);
                if (hp2 == POA_H_NONE) miss = 1;
                const int dp2 = Dat(ps2, j);
                fs2 = hp2 + (dp2 & 7) - 1; os2 = hp2 + ((dp2 >> 3) & 31) - 1;
                xf = hp2 > fs2 ? hp2 : fs2; xo = hp2 > os2 ? hp2 : os2;
            }
            if (__builtin_amdgcn_ballot_w64(miss != 0)) return BT_MISS;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { const int f2 = __shfl_xor(xf, d), o2 = __shfl_xor(xo, d); xf = f2 > xf ? f2 : xf; xo = o2 > xo ? o2 : xo; }
            const int mf = __builtin_amdgcn_readlane(xf, 0), mo = __builtin_amdgcn_readlane(xo, 0);
            const int k2 = !act2 ? 0 : (hp2 == mf ? 1 : (fs2 == mf ? 2 : (hp2 == mo ? 3 : (os2 == mo ? 4 : 0))));
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(k2 != 0);
            if (!b2) return -1;                   // cannot happen: some in-edge attains the maximum
            const int s2 = __builtin_ctzll(b2);
            const int kk = __builtin_amdgcn_readlane(k2, s2);
            r = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane(ps2, s2));
            if (kk & 1) break;                    // the gap opened here
        }
        return 0;
    };
    // a horizontal step taken by E + e or Q + c: the run goes on to the left
    auto left_run = [&]() -> int {
        for (;;) {
            if (--guard < 0) return -1;
            --j;
            if (Hat(r, j) == POA_H_NONE) return BT_MISS;
            const int dj = Dat(r, j);
            if (!(dj >> 8)) break;                // neither E nor Q of this column feeds the next
        }
        return 0;
    };
    for (;;) {
        // (the loop's own test on values the compiler can see to be uniform: else the whole walk is built as a divergent loop, every
        // branch an exec-mask update)
        r = __builtin_amdgcn_readfirstlane(r); j = __builtin_amdgcn_readfirstlane(j); guard = __builtin_amdgcn_readfirstlane(guard);
        if (!(r > 0 && j > 0)) break;
        if (--guard < 0) return -1;
        int a = r0 - r;
        {
            const int drift = j - (j0 - a);
            if ((unsigned)a > 48u || drift < -BT_DRIFT || drift >= BT_DRIFT) { SEC0(); reload(r, j); a = 0; DBGCNT(18, 1); SEC(14); }
        }
        // ONE round of LDS reads per iteration, and one iteration per RUN OF DIAGONAL STEPS PLUS THE STEP THAT ENDS IT (round 5; before, a run
        // and the step behind it were two iterations, and the walk is a chain of iterations -- ~500 clocks each whatever they compute).
        // Lane a + l looks at the cell (r - l, j - l) of the diagonal through (r, j) and decides the step OUT of it by spoa's list -- the
        // diagonal through the row's in-edge; its F + e, H + g, O + c, H + q; then E + e, H + g, Q + c, H + q to the left -- from the
        // band: the cell, its left neighbour, three cells per in-edge.  That covers the rows with at most three in-edges whose source
        // rows and cells are in the band; any other cell answers BTC_EVAL and is evaluated with the in-edges across the lanes, as
        // before.  The cells of a run are those whose step is "diagonal, to the row before" (code bits 3-4: the in-edge); the lane behind the run
        // holds the step that ends it.  A cell the planes do not hold is a miss only where a test needs it.
        const int l = lane - a, myr = r0 - lane, myc = j - l;
        const uint32_t e0 = rim.x, e1 = rim.y;
        const int npl = (int)((e0 >> 8) & 0xf), nppl = npl ? npl : 1;
        const bool mine = l >= 0 && myr >= 1 && myc >= 1;
        // the in-edges' source rows (row 0 for a node without in-edges) and where their cells of column myc sit in the band
        int pl[3], ipl[3];
        bool simple = mine && npl <= 3;
        pl[0] = npl == 0 ? 0 : (int)(e0 >> 16); pl[1] = (int)(e1 & 0xffff); pl[2] = (int)(e1 >> 16);
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            const int kp_ = r0 - pl[s_], xp_ = myc - cs_at(kp_);
            const bool in_ = (unsigned)kp_ < 64u && xp_ >= 1 && xp_ < BT_W;
            simple = simple && (in_ || s_ >= nppl);
            ipl[s_] = in_ && s_ < nppl ? kp_ * BT_W + xp_ : lane * BT_W + 1;
        }
        // no branches around the reads (a lane with nothing to read reads a cell of its own band row): they issue together, one wait
        const int i_own = mine ? lane * BT_W + myc - csk : lane * BT_W + 1;
        const int i_s = mine ? myc - 1 : j - 1;
        const int h = (int)Hb[i_own], hl = (int)Hb[i_own - 1], dl = (int)Db[i_own - 1], sb = (int)lsq[i_s - sb0];
        int hs1[3], hs[3], ds[3];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) { hs1[s_] = (int)Hb[ipl[s_] - 1]; hs[s_] = (int)Hb[ipl[s_]]; ds[s_] = (int)Db[ipl[s_]]; }
        int code = BTC_EVAL;
        if (simple) {
            const int sc = (int)(e0 & 0xff) == sb ? S.m : S.n;
            int dec = h == POA_H_NONE ? BTC_MISS : (sw && h == 0 ? BTC_STOP : -1);
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)                               // the diagonal through the in-edges, in their order
                if (dec < 0 && s_ < nppl) dec = hs1[s_] == POA_H_NONE ? BTC_MISS : (h == hs1[s_] + sc ? (BTC_DIAG | (s_ << 3)) : -1);
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)                               // per in-edge: F + e, H + g, O + c, H + q
                if (dec < 0 && s_ < nppl) {
                    const int fs = hs[s_] + (ds[s_] & 7) - 1, os = hs[s_] + ((ds[s_] >> 3) & 31) - 1;
                    dec = hs[s_] == POA_H_NONE ? BTC_MISS
                        : h == fs + g ? (BTC_VERT | BTC_ON | (s_ << 3)) : h == hs[s_] + g ? (BTC_VERT | (s_ << 3))
                        : h == os + q ? (BTC_VERT | BTC_ON | (s_ << 3)) : h == hs[s_] + q ? (BTC_VERT | (s_ << 3)) : -1;
                }
            if (dec < 0) {                                                // to the left: E + e, H + g, Q + c, H + q
                const int es = hl + ((dl >> 8) & 7) - 1, qs = hl + ((dl >> 11) & 31) - 1;
                dec = hl == POA_H_NONE ? BTC_MISS
                    : h == es + g ? (BTC_LEFT | BTC_ON) : h == hl + g ? BTC_LEFT : h == qs + q ? (BTC_LEFT | BTC_ON) : h == hl + q ? BTC_LEFT : BTC_BAD;
            }
            code = dec;
        }
        const int psel = ((code >> 3) & 3) == 0 ? pl[0] : (((code >> 3) & 3) == 1 ? pl[1] : pl[2]);      // source row of the chosen in-edge
        {
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(code == BTC_DIAG && psel == myr - 1) >> a;
            const int run = ~okm ? __builtin_ctzll(~okm) : 64;
            if (run > 0) {
                DBGCNT(16, 1); DBGCNT(17, run);
                if (l >= 0 && l < run) lpn[(myc - 1) & (BT_RING - 1)] = (unsigned short)myr;
                r -= run; j -= run; moved = true;
                a += run;
                if (a > 63 || r < 1 || j < 1) continue;
            }
            // the step out of (r, j), as lane a decided it
            const int ca = __builtin_amdgcn_readlane(code, a);
            const int what = ca & 7;
            if (what == BTC_STOP) break;
            if (what == BTC_MISS) return BT_MISS;
            if (what == BTC_BAD) return -1;
            if (what != BTC_EVAL) {
                DBGCNT(19, 1);
                moved = true;
                if (what == BTC_LEFT) {
                    --j;
                    if (ca & BTC_ON) { const int rc_ = left_run(); if (rc_) return rc_; }      // by E + e or Q + c: the run goes on to the left
                    continue;
                }
                const int pa = __builtin_amdgcn_readlane(psel, a);
                if (what == BTC_DIAG) {                                    // diagonal through an in-edge whose source is not the row before
                    if (lane == 0) lpn[(j - 1) & (BT_RING - 1)] = (unsigned short)r;
                    r = pa; --j;
                    continue;
                }
                r = pa;                                                    // BTC_VERT
                if (ca & BTC_ON) { const int rc_ = up_run(); if (rc_) return rc_; }          // by F + e or O + c: the run goes on upwards
                continue;
            }
            if (run > 0) continue;      // (the general evaluation below reads the band at the iteration's starting cell: once more from the top)
        }
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)rim.x, a), d1 = (uint32_t)__builtin_amdgcn_readlane((int)rim.y, a);
        int np = (int)((d0 >> 8) & 0xf);
        const int xl = a * BT_W + j - __builtin_amdgcn_readlane(csk, a);
        const int hx = (int)Hb[xl], hlx = (int)Hb[xl - 1], dlx = (int)Db[xl - 1];
        // the other rows: in-edge s on lane s (a second round of reads)
        if (np == 15) np = __builtin_amdgcn_readfirstlane((int)A.np[w.order[r - 1]]);       // (the graph row's field holds up to 15)
        const int npp = np ? np : 1;
        const int ps = np == 0 ? 0 : (lane == 0 ? (int)(d0 >> 16) : (lane == 1 ? (int)(d1 & 0xffff) : (int)(d1 >> 16)));
        const int kp = r0 - ps, xp = j - cs_at(kp);
        const bool act = lane < npp;
        const bool inb = (unsigned)kp < 64u && xp >= 1 && xp < BT_W;
        const int i_p = act && inb ? kp * BT_W + xp : lane * BT_W + 1;
        const int hp1 = (int)Hb[i_p - 1], hpv = (int)Hb[i_p], dpv = (int)Db[i_p];
        int hp = hpv, dp = dpv, hp1x = hp1;
        // one step by spoa's full list of tests; lane s looks at in-edge s
        if (sw && hx == 0) break;
        DBGCNT(19, 1);
        int psx = ps;
        if (np > 3 || __builtin_amdgcn_ballot_w64(act && !inb)) {        // rare: in-edges beyond the third, or a source row outside the band
            if (act) { psx = pred_of(r, d0, d1, lane); hp1x = Hat(psx, j - 1); hp = Hat(psx, j); dp = Dat(psx, j); }
        }
        if (hx == POA_H_NONE || hlx == POA_H_NONE || __builtin_amdgcn_ballot_w64(act && (hp1x == POA_H_NONE || hp == POA_H_NONE))) return BT_MISS;
        moved = true;
        {
            const int sc = (int)(d0 & 0xff) == (int)lsq[j - 1 - sb0] ? S.m : S.n;
            unsigned long long bm = __builtin_amdgcn_ballot_w64(act && hx == hp1x + sc);
            if (bm) {
                if (lane == 0) lpn[(j - 1) & (BT_RING - 1)] = (unsigned short)r;
                r = __builtin_amdgcn_readlane(psx, __builtin_ctzll(bm)); --j;
                continue;
            }
            const int fs = hp + (dp & 7) - 1, os = hp + ((dp >> 3) & 31) - 1;
            const int kind = !act ? 0 : (hx == fs + g ? 1 : (hx == hp + g ? 2 : (hx == os + q ? 3 : (hx == hp + q ? 4 : 0))));
            bm = __builtin_amdgcn_ballot_w64(kind != 0);
            if (bm) {
                const int s = __builtin_ctzll(bm);
                const int kd = __builtin_amdgcn_readlane(kind, s);
                r = __builtin_amdgcn_readlane(psx, s);
                if (kd & 1) { const int rc_ = up_run(); if (rc_) return rc_; }      // by F + e or O + c: the run goes on upwards
                continue;
            }
        }
        {   // to the left
            const int es = hlx + ((dlx >> 8) & 7) - 1, qs = hlx + ((dlx >> 11) & 31) - 1;
            int ext;
            if (hx == es + g) ext = 1; else if (hx == hlx + g) ext = 0; else if (hx == qs + q) ext = 1; else if (hx == hlx + q) ext = 0; else return -1;
            --j;
            if (ext) { const int rc_ = left_run(); if (rc_) return rc_; }      // by E + e or Q + c: the run goes on to the left
        }
    }
    flush(j);
    return 2 * j + (moved ? 1 : 0);
}

// returns the new node count; -1 graph limits, -2 workspace, -3 back-track guard, -4 a cell at the floor of the int16 range
// (global / overlap modes with costly gaps).  *score_out = end-cell score.
// path_out (may be null): node of every base (for the MSA)
// WIDE: the 32-bit form of the pass and of the back-track (its own kernel, poa_consensus_wide_kernel: in the packed kernel's body the extra
// code cost the common path 19 % -- register allocation)
template <bool WIDE>
__device__ __forceinline__ int poa_add(PoaWs& w, const PoaScores S, int N_, int ncap, const int8_t* seq, int m_, int lane, int* score_out, int32_t* path_out, unsigned long long* tacc, int* band_misses, const int mref)
{
    // wave-uniform by construction; say so, or every quantity derived from them lives in VGPRs behind exec-mask branches
    const int N = __builtin_amdgcn_readfirstlane(N_), m = __builtin_amdgcn_readfirstlane(m_);
    unsigned long long tlast = 0;
#ifdef CLH_DEBUG_POA
    tlast = __builtin_amdgcn_s_memtime();
#endif
    *score_out = 0;
    if (m == 0) return N;
    constexpr bool wide = WIDE;
    int bs = 0, br = 0, bc = 0, slope16 = 0;
    PoaWide W;
    W.planeH = nullptr; W.planeD = nullptr; W.carry = nullptr; W.cpitch = 0; W.col0 = nullptr;
    if (N > 0) {
        if (N > POA_MAX_ROWS || (S.algorithm == 1 && N > 25000)) return -1;
        // ---- graph rows in rank space (w.ri: base, in-degree, sink, 0x2000 = its one source is the row before, ranks of the first three sources) ---------------------
        const int pitch = poa_pitch(m);
        const int RING = poa_ring(m);
        uint32_t* ri32 = (uint32_t*)w.ri;
#pragma unroll 4
        for (int r = 1 + lane; r <= N; r += 64) {
            const int v = w.order[r - 1];
            const int np = w.np[v];
            uint32_t pr[3] = {0, 0, 0};
            for (int s2 = 0; s2 < 3; ++s2) if (s2 < np) pr[s2] = (uint32_t)w.rank[w.pred[v * POA_MAXP + s2]];
            w.ri[r] = make_uint2((uint32_t)(w.base[v] & 0xff) | ((uint32_t)(np < 15 ? np : 15) << 8) | (w.nout[v] == 0 ? 0x1000u : 0u) | ((np == 1 && pr[0] != 0 && pr[0] == (uint32_t)(r - 1)) ? 0x2000u : 0u) | (pr[0] << 16), pr[1] | (pr[2] << 16));
        }
        phase_sync();
        // where will row r read source p from?  the row before it: registers; another recent row: the LDS ring (0x4000 on
        // the source row: it leaves a copy there); an older one: the planes in HBM
        for (int r = 1 + lane; r <= N; r += 64) {
            const uint2 d = w.ri[r];
            const int np = (int)((d.x >> 8) & 0xf);
            const int p0 = (int)(d.x >> 16), p1 = (int)(d.y & 0xffff), p2 = (int)(d.y >> 16);
            auto mark = [&](int q) { if (q != 0 && r - q >= 2) atomicOr(&ri32[q * 2], r - q < RING ? 0x4000u : 0x8000u); };     // ring copy / every cell to the planes
            if (np > 0) mark(p0);
            if (np > 1) mark(p1);
            if (np > 2) mark(p2);
            if (np > 3) { const int v = w.order[r - 1]; const int npt = w.np[v]; for (int s2 = 3; s2 < npt; ++s2) mark(w.rank[poa_pred_at(w, v, s2)]); }
        }
        if ((wide ? poa_dp_bytes_w(N, m) : poa_dp_bytes(N, m)) > w.dp_bytes) return -2;
        w.planeH = (short*)w.dp;
        w.planeD = (unsigned short*)(w.dp + poa_plane_bytes(N, m));
        if constexpr (WIDE) {      // H in int32, the differences behind it, then the carries of the passes and the first column of global mode
            W.planeH = (int*)w.dp;
            W.planeD = (unsigned short*)(w.dp + poa_plane_bytes_w(N, m));
            W.cpitch = (N + 2 + 3) & ~3;
            W.carry = (int*)((uint8_t*)W.planeD + poa_plane_bytes(N, m));
            W.col0 = W.carry + 6 * (size_t)W.cpitch;
        }
        if (S.algorithm == 1) {
            // global mode: H[i][0] = max(F, O)[i][0], F[i][0] = e + max over sources (a node without in-edges: g), O likewise.
            // A chain over the ranks, wave-uniform (every lane computes and stores the same values); not a hot path.
            for (int r = 1; r <= N; ++r) {
                const int v = w.order[r - 1];
                const int np = w.np[v];
                int f = np ? -(1 << 28) : S.g - S.e, o = np ? -(1 << 28) : S.q - S.c;
                for (int s2 = 0; s2 < np; ++s2) {
                    const int pr = w.rank[poa_pred_at(w, v, s2)];
                    f = w.score[pr] > f ? w.score[pr] : f; o = w.bp[pr] > o ? w.bp[pr] : o;
                }
                f += S.e; o += S.c;
                w.score[r] = f; w.bp[r] = o;
                const int h = f > o ? f : o;
                if constexpr (WIDE) W.col0[r] = h;
                else {
                    if (h <= POA_NEG) return -4;             // column 0 leaves the int16 range (wave-uniform)
                    w.col0[r] = (short)h;
                }
                __syncthreads();
            }
        }
        phase_sync();
        // the planes keep a band around the straight line through the matrix (see POA_BAND) unless the sequence is short anyway
        // (the line's end: the longest sequence so far, not this one -- a partial last copy runs along the same line and stops early)
        slope16 = m > 2 * POA_BAND + 64 ? (int)(((unsigned)(mref > m ? mref : m) << 16) / (unsigned)N) : 0;
        if constexpr (WIDE) { slope16 = 0; dp_rows_w(w, W, S, N, m, seq, lane, bs, br, bc); }
        else dp_rows(w, S, N, m, seq, lane, slope16, bs, br, bc DBGPASS);
        phase_sync();
        if (br < 0) return -4;                               // a cell at the floor of the int16 range: no exact answer from this kernel
#ifdef CLH_DEBUG_POA
        {   // row steps by register count of the pass (x16: the reader undoes the >>4 of the clocks)
            constexpr int WMAX = 128 * POA_MAXCP;
            for (int cb = 0; cb < m; cb += WMAX) { const int rem = m - cb; const int cp = rem > WMAX ? POA_MAXCP : poa_cols(rem); tacc[5] += (unsigned long long)N << 4; tacc[6] += (unsigned long long)N * cp << 4; }
            tacc[7] += (unsigned long long)N * m;      // cells / 16
        }
#endif
    }
    *score_out = bs;
    TSTAMP(0);
    // ---- back-track: spoa's value comparisons, replayed on the cells of the path from the two planes ---------------------------
    bool moved = false;                                    // the alignment holds at least one step
    int jb = 0, je = -1;                                   // ... and the bases [jb, je]
    for (int t = lane; t < m; t += 64) w.pn[t] = 0;        // pn[j] = rank aligned to base j (0: none); only diagonal steps write
    {
        int j = br > 0 ? __builtin_amdgcn_readfirstlane(bc) : 0;
        je = j - 1;
        if (br > 0) {
            phase_sync();
            BtArgs A = {w.pn, wide ? (void*)W.planeH : (void*)w.planeH, wide ? W.planeD : w.planeD, w.ri, wide ? (void*)W.col0 : (void*)w.col0, w.rank, w.pred, w.order, seq, N, m,
                        __builtin_amdgcn_readfirstlane(br), j, slope16, w.ovn, w.ovp, w.ovc, w.np};
            int rc;
            if constexpr (WIDE) rc = poa_backtrack<int, BT_W32>(A, S DBGPASS); else rc = poa_backtrack<short, BT_W>(A, S DBGPASS);
            if (!WIDE && rc == BT_MISS) {
                // the walk left the band of cells the planes hold: the pass once more, every cell stored (the end cell is the same), and
                // the walk again.  Rare; counted (clh_ccs_plan_stats) so that it is seen if it ever is not
                if (band_misses) *band_misses += 1;
                phase_sync();
                int b1 = S.algorithm == 0 ? 0 : -(1 << 30), b2 = 0, b3 = 0;
                if constexpr (!WIDE) dp_rows(w, S, N, m, seq, lane, 0, b1, b2, b3 DBGPASS);
                for (int t = lane; t < m; t += 64) w.pn[t] = 0;
                phase_sync();
                A.slope16 = 0;
                if constexpr (!WIDE) rc = poa_backtrack<short, BT_W>(A, S DBGPASS);
            }
            if (rc < 0) return -3;
            j = rc >> 1; moved = (rc & 1) != 0;
        }
        jb = S.algorithm == 1 ? 0 : j;                           // global mode: spoa walks on along the border to (0, 0)
    }
    phase_sync();
    TSTAMP(1);
    // ---- fuse the path into the graph (Graph::AddAlignment), data-parallel over the bases -------------------------------
    // Base i touches only its own aligned node's set and the in-edge list of the node it ends on, and the nodes of one
    // path are distinct, so the sequential rule is evaluated per lane.  Node ids as AddAlignment gives them: the bases in
    // front of the alignment [0, jb), the bases behind it (je, m), then the new nodes among the bases [jb, je] it holds.
    int n = N, fail = 0;
    {
        if (N == 0 && lane == 0) { w.ovc[0] = 0; w.ovc[1] = 0; }       // a new graph: its overflow table is empty
        if (!moved) { jb = m; je = m - 1; }                // empty alignment: the whole sequence is a chain of new nodes
        else if (jb > je) return -5;                       // steps but no base (overlap mode, vertical steps only): spoa throws
        const int nlead = jb, ntrail = m - 1 - je;
        n = N + nlead + ntrail;
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < m, mid = act && i >= jb && i <= je;
            const int rk = mid ? w.pn[i] : 0;
            const int v = rk > 0 ? w.order[rk - 1] : -1;
            const int b = act ? (int)seq[i] : 0;
            int use = -1;
            if (v >= 0) {
                if (w.base[v] == b) use = v;
                else { const int nav = w.na[v]; for (int t = 0; t < nav; ++t) { const int x = w.aligned[v * POA_MAXA + t]; if (w.base[x] == b) { use = x; break; } } }
            }
            const bool newmid = mid && use < 0;
            const unsigned long long nb = __builtin_amdgcn_ballot_w64(newmid);
            const unsigned long long below = ((unsigned long long)1 << lane) - 1;
            if (newmid) use = n + __builtin_popcountll(nb & below);
            else if (act && !mid) use = i < jb ? N + i : N + nlead + (i - je - 1);
            n += __builtin_popcountll(nb);
            if (act && use >= N && use < ncap) {           // a new node
                int nal = 0;
                if (v >= 0) {                              // a new member of the aligned set of v: every member's list gains it; its own
                    const int nav = w.na[v];               // list is v's list followed by v
                    if (nav >= POA_MAXA) fail = 1;
                    else {
                        for (int t = 0; t < nav; ++t) {
                            const int x = w.aligned[v * POA_MAXA + t];
                            w.aligned[use * POA_MAXA + t] = x;
                            const int u = w.na[x];
                            w.aligned[x * POA_MAXA + u] = use; w.na[x] = (int8_t)(u + 1);       // u == nav < POA_MAXA: the lists of a set are equally long
                        }
                        w.aligned[use * POA_MAXA + nav] = v;
                        w.aligned[v * POA_MAXA + nav] = use; w.na[v] = (int8_t)(nav + 1);
                        nal = nav + 1;
                    }
                }
                w.base[use] = (int8_t)b; w.np[use] = 0; w.na[use] = (int8_t)nal; w.cov[use] = 0; w.nout[use] = 0;
            }
            if (act) { w.pj[i] = use; if (path_out) path_out[i] = use; }
        }
        if (n > ncap || n > POA_MAX_ROWS) return -1;
        phase_sync();
        for (int i = lane; i < m; i += 64) {
            const int x = w.pj[i];
            if (m >= 2) w.cov[x] += 1;                     // Node::Coverage counts the sequences with an edge at the node
            if (i == 0) continue;
            const int u = w.pj[i - 1];
            const int cnt = w.np[x];
            int found = -1;
            for (int s2 = 0; s2 < cnt; ++s2) if (poa_pred_at(w, x, s2) == u) { found = s2; break; }
            if (found >= 0) {
                if (found < POA_MAXP) w.pw[x * POA_MAXP + found] += 2; else w.ovw[poa_ovf_slot(w, x, found - POA_MAXP)] += 2;
            } else if (cnt >= POA_MAXP_ALL) fail = 1;
            else if (cnt >= POA_MAXP) {                     // the node's in-place slots are full: the graph's overflow table (a node gains at most one in-edge per sequence, so the entries of a node keep their order)
                const int slot = atomicAdd(&w.ovc[1], 1);
                if (slot >= POA_OVF_CAP) fail = 1;
                else { w.ovn[slot] = x; w.ovp[slot] = u; w.ovw[slot] = 2; w.np[x] = (int8_t)(cnt + 1); w.nout[u] += 1; }
            } else { w.pred[x * POA_MAXP + cnt] = u; w.pw[x * POA_MAXP + cnt] = 2; w.np[x] = (int8_t)(cnt + 1); w.nout[u] += 1; }
        }
        if (__builtin_amdgcn_ballot_w64(fail != 0)) return -1;
        // entries appended during this sequence become visible to the searches only now (ovc[1] counts ahead of ovc[0]): a search
        // that ran beside the appends never saw a half-written entry
        phase_sync();
        if (lane == 0) w.ovc[0] = w.ovc[1] < POA_OVF_CAP ? w.ovc[1] : POA_OVF_CAP;
    }
    phase_sync();
    TSTAMP(2);
    if ((n <= POA_SORT_LDS ? poa_sort_lds(w, N, n, m, lane) : poa_sort(w, N, n, m, lane)) != 0) return -1;
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
            // Most ranks are links of a chain: one in-edge, from the rank before.  Along such a run the score is the running sum of the
            // edge weights on top of the score in front of the run (weights are >= 2: the scores rise strictly, so only the run's
            // last rank can become the best one), and all of it is taken at once: one prefix sum of the weights per 64 ranks.
            const bool chain_l = in_lds && lane < cnt && (b0 & 0x7f) == 1 && (int)(b0 >> 16) == rb + lane - 1;
            const int psum = wave_prefix_sum(chain_l ? (int)(b2 & 0xff) : 0);
            const unsigned long long chain_m = __builtin_amdgcn_ballot_w64(chain_l);
            for (int i = 0; i < cnt; ++i) {
                {
                    const unsigned long long rest = ~(chain_m >> i);
                    int run = rest ? __builtin_ctzll(rest) : 64;
                    run = run < cnt - i ? run : cnt - i;
                    if (run >= 3) {
                        const int before = i > 0 ? __builtin_amdgcn_readlane(psum, i - 1) : 0;
                        const bool dead = barred && prev == -1;                   // a barred tail: nothing reaches these ranks
                        const int scl = dead ? -1 : prev + (psum - before);
                        if (lane >= i && lane < i + run) { score[rb + lane] = scl; bpb = dead ? 0 : rb + lane - 1; }
                        const int lastsc = __builtin_amdgcn_readlane(scl, i + run - 1);
                        if (dead) { if (-1 > tops) { tops = -1; top = rb + i; } }
                        else if (lastsc > tops) { tops = lastsc; top = rb + i + run - 1; }
                        prev = lastsc;
                        asm volatile("" ::: "memory");
                        i += run - 1;
                        continue;
                    }
                }
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
                        relax(__builtin_amdgcn_readfirstlane(w.rank[poa_pred_at(w, v, s2)]), __builtin_amdgcn_readfirstlane(poa_pw_at(w, v, s2)));
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
            for (int s2 = 0; s2 < np; ++s2) succ |= w.rank[poa_pred_at(w, v, s2)] == top;
            if (succ) for (int s2 = 0; s2 < np; ++s2) { const int u = w.rank[poa_pred_at(w, v, s2)]; if (u != top) score[u] = -1; }
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

// the body of K3's kernels.  WIDE = false: the packed (16-bit) form; a read it cannot take -- a copy above POA_MAX_COPY bases, scores
// outside the 16-bit cells, a cell that reached the floor of the range -- goes on the wide list.  WIDE = true: the reads of that list
// in the 32-bit form, by persistent waves over the plan's worst-case slots.
template <bool WIDE>
__device__ __forceinline__ void poa_kernel_body(const CcsParams& p)
{
    const int lane = threadIdx.x & 63;
    uint8_t* slot = p.poa_ws + (size_t)blockIdx.x * p.slot_bytes;
    PoaScores S = p.sc;
    const bool force_wide = (S.algorithm & 0x200) != 0;     // the host saw scores outside the 16-bit cells (or a test asks for the wide form)
    S.algorithm &= 0xff;
    const int n_items = WIDE ? __builtin_amdgcn_readfirstlane(*p.wide_count) : p.n;
    for (int turns = 0; turns <= p.n; ++turns) {             // a wave takes at most every read once
        int idx = 0;
        if (lane == 0) idx = atomicAdd(p.work_counter, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);    // wave-uniform in the compiler's eyes too: scalar loads, scalar branches, SGPR pointers below
        if (idx >= n_items) break;
        const int rd = WIDE ? __builtin_amdgcn_readfirstlane(p.wide_list[idx]) : (p.work_order ? p.work_order[idx] : idx);
        const int64_t off = p.read_off[rd];
        const int L = (int)(p.read_off[rd + 1] - off);
        const int8_t* seq = p.reads + off;
        if (!WIDE && p.tier == 1 && __builtin_amdgcn_readfirstlane(p.results[rd].status) != 1) continue;   // second tier: only what did not fit a first-tier slot
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
        // cells beyond int16: the wide form of the pass (spoa's engines do the same), in its own kernel
        // (the read's row is written as "nothing yet": rows are recycled memory, and the second launch takes only rows that say status 1)
        auto to_wide_list = [&]() { if (lane == 0) { p.results[rd] = res; const int k = atomicAdd(p.wide_count, 1); p.wide_list[k] = rd; } };
        if (!WIDE && (force_wide || maxlen > POA_MAX_COPY)) { if (p.tier == 0) to_wide_list(); continue; }
        constexpr bool wide = WIDE;
        // workspace: this wave's slot, or -- a read that needs more -- one of the large slots, claimed for the duration of
        // the read; none free (or none large enough): status 1, the second launch over the large slots takes the read
        uint8_t* ws = slot;
        size_t ws_bytes = p.slot_bytes;
        int big = -1;
        const size_t need_min = poa_fixed_bytes(ncap, mcap, nullptr) + (wide ? poa_dp_bytes_w(maxlen + 8, maxlen) : poa_dp_bytes(maxlen + 8, maxlen)) + 64;
        bool use_big = need_min > p.slot_bytes;
        int N = 0, len = -1, ncols = 0;
        unsigned long long dp_cells = 0, dp_rows_n = 0;      // work of this read: DP cells and row steps (the bench's cell-update rate)
        int band_miss_n = 0;
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
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the slot's last user may have run on another CU: nothing of it in this CU's L1
                ws = p.big_ws + (size_t)big * p.big_slot_bytes; ws_bytes = p.big_slot_bytes;
                if (lane == 0 && p.stats) atomicAdd(p.stats, 1);
            }
            PoaWs w = carve(ws, ws_bytes, ncap, mcap);
            phase_sync();
            unsigned long long tacc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            N = 0; b = 0;
            int si = 0, mref = 0;
            for (int i = 0; i <= ncuts && N >= 0; ++i) {
                if (i == ncuts && !tail) break;
                const int cut = i < ncuts ? __builtin_amdgcn_readfirstlane(cuts[i]) : L;
                int sc1 = 0;
                if (N > 0 && cut > b) { dp_cells += (unsigned long long)N * (unsigned)(cut - b); dp_rows_n += (unsigned long long)N * (unsigned)((cut - b + 128 * POA_MAXCP - 1) / (128 * POA_MAXCP)); }
                N = poa_add<WIDE>(w, S, N, ncap, seq + b, cut - b, lane, &sc1, p.msa_col ? p.msa_col + off + b : nullptr, tacc, &band_miss_n, mref);
                mref = cut - b > mref ? cut - b : mref;
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
                if (lane == 0) for (int k = 0; k < 8; ++k) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (55 + k)] = (int)(tacc[k] >> 4);
                if (lane == 0) for (int k = 8; k < 14; ++k) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (41 + k)] = (int)(tacc[k] >> 4);
                if (lane == 0) for (int k = 16; k < 20; ++k) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (26 + k)] = (int)(tacc[k] >> 4);   // back-track counters
                if (lane == 0) for (int k = 14; k < 16; ++k) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (26 + k)] = (int)(tacc[k] >> 4);   // back-track clocks: band staging, set-up
#endif
            }
            break;
        }
        if (!WIDE && N == -4) {                           // a cell left the int16 range: the read once more in the wide form (either launch of the
            to_wide_list();                                // packed kernel may say so: the wide kernel runs behind both on the same stream)
            __syncthreads();
            if (big >= 0) { __threadfence(); if (lane == 0) atomicExch(&p.big_busy[big], 0); }
            continue;
        }
        if (N == -2) res.status = 1;
        else if (N == -3) res.status = 5;
        else if (N == -4) res.status = 6;
        else if (N == -5) res.status = 7;
        else if (N < 0) res.status = 2;
        else if (len < 0) res.status = 3;
        else { res.nseg = nseg; res.ccs_len = len; }
        if (lane == 0) {
            p.results[rd] = res; if (p.msa_ncols) p.msa_ncols[rd] = ncols;
            // a read handed to the second launch is counted there; status 1 is final in that launch, and in the first one when no
            // second launch will run (no large slots)
            if (p.stats && (WIDE || res.status != 1 || p.tier == 1 || p.n_big == 0)) {
                atomicAdd((unsigned long long*)(p.stats + 2), dp_cells); atomicAdd((unsigned long long*)(p.stats + 4), dp_rows_n);
                if (band_miss_n) atomicAdd(p.stats + 6, band_miss_n);
                if (res.status != 0) atomicAdd(p.stats + 8 + (res.status & 7), 1);      // reads lost to a limit of this kernel, by status
            }
        }
        __syncthreads();
        if (big >= 0) {                      // every store into the large slot has landed before another wave may claim it
            __threadfence();
            if (lane == 0) atomicExch(&p.big_busy[big], 0);
        }
    }
}

__global__ void __launch_bounds__(64, POA_WAVES) poa_consensus_kernel(const CcsParams p) { poa_kernel_body<false>(p); }
__global__ void __launch_bounds__(64, 2) poa_consensus_wide_kernel(const CcsParams p) { poa_kernel_body<true>(p); }

// one launch class of K2: `count` reads from p.work_order[p.k2_begin ...], none longer than p.lcap (reads above p.k2_lds_max
// return at once); with_long: also the HBM-workspace kernel for those
hipError_t launch_ccs_scan(const CcsParams& p, int count, bool with_long, hipStream_t stream)
{
    const size_t lds = k2_lds_bytes(p.lcap);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)ccs_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)ccs_scan_kernel_x4, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    static const int wide_from = getenv("CLH_K2_WIDE_FROM") ? atoi(getenv("CLH_K2_WIDE_FROM")) : K2_WIDE_FROM;
    if (count > 0 && p.lcap >= wide_from && !getenv("CLH_K2_ONE_WAVE")) hipLaunchKernelGGL(ccs_scan_kernel_x4, dim3(count), dim3(256), lds, stream, p);
    else if (count > 0) hipLaunchKernelGGL(ccs_scan_kernel, dim3(count), dim3(64), lds, stream, p);
    if (with_long && p.n_long > 0) hipLaunchKernelGGL(ccs_scan_long_kernel, dim3(p.n_long), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// K3's work list by what a read will COST, heaviest first -- an experiment (clh_api.hip: CLH_POA_ORDER_BY_COST), not the default:
// it balances the end of the launch (7 % of the wave-slot time is idle there) and loses more than that in the middle, see
// clh_ccs_run.  Cost of a read after K2: its alignments x period^2.  One workgroup: a counting sort over 1024 buckets of
// log2(cost) with 5 fractional bits; the order inside a bucket is whatever the atomics give -- results do not depend on it.
__global__ void __launch_bounds__(1024) ccs_work_order_kernel(const CcsScan* scan, const int n, int32_t* order)
{
    __shared__ int hist[1024];
    const int tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    auto bucket = [&](int r) -> int {
        const CcsScan& sc = scan[r];
        const unsigned long long cost = sc.period > 0 && sc.ncuts > 0 ? (unsigned long long)sc.ncuts * (unsigned long long)sc.period * (unsigned long long)sc.period : 0ull;
        if (cost < 32) return 1023 - (int)cost;
        const int lg = 63 - __builtin_clzll(cost);                   // >= 5
        const int frac = (int)((cost >> (lg - 5)) & 31);
        int key = (lg - 4) * 32 + frac;                              // 32.. : monotone in cost
        key = key > 1023 ? 1023 : key;
        return 1023 - key;                                           // heaviest first
    };
    for (int r = tid; r < n; r += 1024) atomicAdd(&hist[bucket(r)], 1);
    __syncthreads();
    // exclusive prefix sum of the 1024 counts (one per thread)
    const int mine = hist[tid];
    int inc = mine;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    __shared__ int wsum[16];
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    __syncthreads();
    hist[tid] = base + inc - mine;
    __syncthreads();
    for (int r = tid; r < n; r += 1024) order[atomicAdd(&hist[bucket(r)], 1)] = r;
}

hipError_t launch_ccs_work_order(const CcsScan* scan, int n, int32_t* order, hipStream_t stream)
{
    hipLaunchKernelGGL(ccs_work_order_kernel, dim3(1), dim3(1024), 0, stream, scan, n, order);
    return hipGetLastError();
}

hipError_t launch_poa(const CcsParams& p, int nslots, hipStream_t stream)
{
    hipLaunchKernelGGL(poa_consensus_kernel, dim3(nslots), dim3(64), (size_t)POA_LDS_BYTES, stream, p);
    return hipGetLastError();
}

// the reads on the wide list (p.wide_list / p.wide_count, filled by the launches above), over p.poa_ws = worst-case slots
hipError_t launch_poa_wide(const CcsParams& p, int nslots, hipStream_t stream)
{
    hipLaunchKernelGGL(poa_consensus_wide_kernel, dim3(nslots), dim3(64), (size_t)POA_LDS_BYTES, stream, p);
    return hipGetLastError();
}

size_t poa_slot_bytes_host(int ncap, int mcap) { return poa_slot_bytes(ncap, mcap); }
size_t poa_slot_min_bytes_host(int ncap, int mcap) { return poa_fixed_bytes(ncap, mcap, nullptr) + poa_dp_bytes(mcap + 8, mcap - 1) + 64; }
// the same for the wide (32-bit) form of the pass: sequences above 2800 bases, scores outside the 16-bit cells
size_t poa_slot_bytes_host_w(int ncap, int mcap) { return poa_fixed_bytes(ncap, mcap, nullptr) + poa_dp_bytes_w(ncap, mcap - 1) + 64; }

}  // namespace clh
