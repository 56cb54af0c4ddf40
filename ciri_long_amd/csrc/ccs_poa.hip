// ccs_poa.hip -- K2 ccs_scan_kernel (repeat period + copy boundaries) and K3 poa_consensus_kernel (partial-order
// alignment of the copies, heaviest-path consensus) for gfx950.
//
// What they replace: pyccs.find_consensus (called at CIRI_long/find_ccs.py:14) and the spoa engine inside it.  Those are
// external packages that exist neither in the reference tree nor in this environment: PARITY UNPINNED.  Both kernels
// implement, bit for bit, the specification written for this project in oracle/ccs_oracle.c ("clh-ccs v1" / "clh-poa v1");
// read that header for every rule and tie-break.  One read per wavefront, one wavefront per workgroup.
//
// K2: 8-mer codes of the read sit in LDS.  Matches per offset are counted per PAIR of equal 8-mers (positions chained per
//     hash bucket with LDS atomics, each pair visited once: O(L * copies) instead of O(L^2/4) comparisons); the smoothed
//     maximum, the harmonic test and the per-copy boundary search (same chains, a histogram over the candidate offsets)
//     are lane-parallel with shuffle reductions.
// K3: persistent waves pull reads from an atomic counter; each owns a workspace slot in HBM (graph arrays, int16 DP
//     matrix, direction bytes; slots are sized for the common case, the few reads that need more run in a second
//     launch over a handful of large slots).  A DP row (one graph node) is computed by the 64 lanes over the sequence positions: the
//     diagonal/vertical maxima over the node's in-edges are independent per position, the horizontal gap chain
//     H[j] = max(A[j], H[j-1]+g) is a max-plus prefix scan (linear gap cost), done with DPP-free shuffles per 64-wide
//     chunk and a carried running maximum.  The walk-back, the graph update and the heaviest-path pass are sequential by
//     nature and run wave-uniformly (every lane the same control flow, lane 0 stores).
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
static constexpr int POA_MAXP = 12;
static constexpr int POA_MATCH = 10;
static constexpr int POA_MISMATCH = -4;
static constexpr int POA_GAP = -8;
static constexpr int POA_MAX_COPY = 2800;            // longest copy: cells are int16 and hold H - jl*gap <= 10*2800 + 8*512

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

// H value of a far source row, read from HBM without letting the compiler merge it with an LDS read (see poa_add)
__device__ __forceinline__ int far_h(const short* ptr) {
    int x;
    asm volatile("global_load_sshort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(ptr) : "memory");
    return x;
}

__device__ __forceinline__ int wmax_i(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_xor(v, d); v = o > v ? o : v; }
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
// K3
// ------------------------------------------------------------------------------------------------------------
struct PoaWs {            // views into one wave's workspace slot
    int8_t* base; int8_t* np; int32_t* pred; int32_t* pw; int32_t* aligned; long long* key; int32_t* order; int32_t* rank;
    int32_t* pn; int32_t* pj; int32_t* score; int32_t* bp; short* H; uint8_t* dir;
    uint2* ri; uint32_t* tab; short* carry; int cpitch;
};

// row pitch (elements) of the DP matrix and of the direction bytes for a copy of m bases: column j sits at element j+7,
// so the 2/4/8 columns a lane owns start at an aligned element, and the pitch is a multiple of 16
__host__ __device__ inline int poa_pitch(int m) { return (m + 8 + 15) & ~15; }

__device__ PoaWs carve(uint8_t* slot, int ncap, int mcap)
{
    PoaWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { uint8_t* q = slot + o; o = (o + bytes + 15) & ~(size_t)15; return q; };
    w.key = (long long*)take(sizeof(long long) * ncap);
    w.pred = (int32_t*)take(sizeof(int32_t) * (size_t)ncap * POA_MAXP);
    w.pw = (int32_t*)take(sizeof(int32_t) * (size_t)ncap * POA_MAXP);
    w.aligned = (int32_t*)take(sizeof(int32_t) * (size_t)ncap * 3);
    w.order = (int32_t*)take(sizeof(int32_t) * ncap);
    w.rank = (int32_t*)take(sizeof(int32_t) * ncap);
    w.pn = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + mcap + 2));
    w.pj = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + mcap + 2));
    w.score = (int32_t*)take(sizeof(int32_t) * ncap);
    w.bp = (int32_t*)take(sizeof(int32_t) * (size_t)(ncap + 2));
    w.ri = (uint2*)take(sizeof(uint2) * (size_t)(ncap + 2));
    w.tab = (uint32_t*)take(sizeof(uint32_t) * 3 * (size_t)(ncap + 2));
    w.cpitch = (ncap + 2 + 7) & ~7;
    w.carry = (short*)take(sizeof(short) * 2 * (size_t)w.cpitch);
    w.base = (int8_t*)take(ncap);
    w.np = (int8_t*)take(ncap);
    w.H = (short*)take(sizeof(short) * (size_t)(ncap + 1) * poa_pitch(mcap - 1));
    w.dir = (uint8_t*)take((size_t)(ncap + 1) * poa_pitch(mcap - 1));
    return w;
}

__host__ __device__ inline size_t poa_slot_bytes(int ncap, int mcap)
{
    size_t o = 0;
    auto add = [&](size_t bytes) { o = (o + bytes + 15) & ~(size_t)15; };
    add(sizeof(long long) * ncap); add(sizeof(int32_t) * (size_t)ncap * POA_MAXP); add(sizeof(int32_t) * (size_t)ncap * POA_MAXP);
    add(sizeof(int32_t) * (size_t)ncap * 3); add(sizeof(int32_t) * ncap); add(sizeof(int32_t) * ncap);
    add(sizeof(int32_t) * (size_t)(ncap + mcap + 2)); add(sizeof(int32_t) * (size_t)(ncap + mcap + 2));
    add(sizeof(int32_t) * ncap); add(sizeof(int32_t) * (size_t)(ncap + 2)); add(sizeof(uint2) * (size_t)(ncap + 2)); add(sizeof(uint32_t) * 3 * (size_t)(ncap + 2));
    add(sizeof(short) * 2 * (size_t)((ncap + 2 + 7) & ~7)); add(ncap); add(ncap);
    add(sizeof(short) * (size_t)(ncap + 1) * poa_pitch(mcap - 1)); add((size_t)(ncap + 1) * poa_pitch(mcap - 1));
    return o + 64;
}

extern __shared__ __attribute__((aligned(16))) uint32_t poa_lds[];     // K3's dynamic LDS block (POA_LDS_BYTES, declared below)
static constexpr int POA_RERANK_LDS_KEYS = 768;                        // new-node keys staged in LDS for the merge (6 KiB)

// merge the nodes created by the last sequence ([n_old, n_new), keys ascending in creation order) into the rank order
__device__ void poa_rerank(const PoaWs& w, int n_old, int n_new, int lane)
{
    const int added = n_new - n_old;
    // old node at rank r (key r<<20): new rank = r + #{new nodes with key < its key}.  The binary search runs over the new
    // keys in LDS (the H ring is idle here) instead of a chain of dependent HBM loads per probe.
    long long* lkeys = (long long*)poa_lds;
    const bool in_lds = added <= POA_RERANK_LDS_KEYS;
    if (in_lds) {
        for (int i = lane; i < added; i += 64) lkeys[i] = w.key[n_old + i];
        __syncthreads();
    }
    for (int r = 1 + lane; r <= n_old; r += 64) {
        const int v = w.order[r - 1];
        const long long k = (long long)r << 20;
        int lo = 0, hi = added;                      // lower_bound over new keys
        if (in_lds) while (lo < hi) { const int mid = (lo + hi) >> 1; if (lkeys[mid] < k) lo = mid + 1; else hi = mid; }
        else while (lo < hi) { const int mid = (lo + hi) >> 1; if (w.key[n_old + mid] < k) lo = mid + 1; else hi = mid; }
        w.rank[v] = r + lo;
    }
    // new node i: new rank = 1 + i + #{old nodes with key <= its key} = 1 + i + clamp(key >> 20, 0, n_old)
    for (int i = lane; i < added; i += 64) {
        long long q = w.key[n_old + i] >> 20;
        if (q < 0) q = 0;
        if (q > n_old) q = n_old;
        w.rank[n_old + i] = 1 + i + (int)q;
    }
    phase_sync();
    // candidate order (by key, id).  It can violate an edge when a base re-used a member of an aligned set that ranks
    // after the row it was aligned to; the specification's final order is the depth-first post-order over in-edges
    // taken in candidate order, which IS the candidate order whenever that is already topological (the common case,
    // detected in parallel).
    int bad = 0;
    for (int v = lane; v < n_new; v += 64) {
        const int r = w.rank[v];
        w.bp[r - 1] = v;                                   // bp doubles as the candidate order here
        const int np = w.np[v];
        for (int e = 0; e < np; ++e) bad |= (w.rank[w.pred[v * POA_MAXP + e]] >= r);
    }
    phase_sync();
    if (__builtin_amdgcn_ballot_w64(bad != 0) == 0) {
        for (int v = lane; v < n_new; v += 64) { const int r = w.rank[v]; w.order[r - 1] = v; w.key[v] = (long long)r << 20; }
        phase_sync();
        return;
    }
    for (int v = lane; v < n_new; v += 64) w.score[v] = 0;     // score doubles as the visited flags
    phase_sync();
    int outn = 0;
    for (int c = 0; c < n_new; ++c) {                          // wave-uniform, every lane stores
        const int root = w.bp[c];
        if (w.score[root]) continue;
        int sp = 0;
        w.pn[0] = root; w.pj[0] = 0; w.score[root] = 1;
        while (sp >= 0) {
            const int u = w.pn[sp];
            const int i = w.pj[sp];
            if (i < w.np[u]) {
                const int pr = w.pred[u * POA_MAXP + i];
                w.pj[sp] = i + 1;
                if (!w.score[pr]) { w.score[pr] = 1; ++sp; w.pn[sp] = pr; w.pj[sp] = 0; }
            } else {
                w.order[outn++] = u;
                --sp;
            }
        }
    }
    phase_sync();
    for (int r = 1 + lane; r <= n_new; r += 64) { const int v = w.order[r - 1]; w.rank[v] = r; w.key[v] = (long long)r << 20; }
    phase_sync();
}

#ifdef CLH_DEBUG_POA
__device__ unsigned long long g_t[8];
#define TSTAMP(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define TSTAMP(k) do {} while (0)
#endif
static constexpr int POA_LDS_BYTES = 6144;           // ring of recent H rows (DP) / score per rank (heaviest path)
static constexpr int POA_RING_SHORTS = POA_LDS_BYTES / 2;
static constexpr int POA_LDS_SCORES = POA_LDS_BYTES / 4 - 1;   // most rows whose scores fit the LDS block
static_assert(POA_RERANK_LDS_KEYS * 8 <= POA_LDS_BYTES, "rerank keys must fit the K3 LDS block");
#define lds_ring ((short*)poa_lds)

// columns per lane, LDS row pitch and ring depth for a copy of m bases
__device__ __forceinline__ int poa_cols(int m) { const int c = (m + 63) >> 6; return c < 2 ? 2 : (c > 8 ? 8 : c); }   // columns per lane
__device__ __forceinline__ int poa_ring(int m) {
    const int W = 64 * poa_cols(m);
    const int lp = poa_pitch(m < W ? m : W);
    int ring = 16;                                   // power of two, so slot = rank & (ring-1); lp <= 528, so ring >= 4
    while (ring * lp > POA_RING_SHORTS) ring >>= 1;
    return ring;
}

// direction byte: a code that DEcreases along the specification's evaluation order -- diagonal from in-edge e: 0x3F - e,
// diagonal from row 0: 0x20, vertical from in-edge e: 0x1F - e, horizontal: 0 -- so that a candidate carried as
// (value << 8) | code needs one signed max to take the larger value and, between equal values, keep the earlier candidate
// (the strict '>' chain of the specification).  The layout (move group in bits 4-5, 15 - slot in bits 0-3) makes the
// decode of the walk-back two shifts and a subtraction.
static constexpr int POA_CODE_DIAG = 0x3F, POA_CODE_ROW0 = 0x20, POA_CODE_VERT = 0x1F;

// C+1 H values (columns first-1 .. first+C-1) of a far source row from HBM; assembly for the reason given at far_h
template <int C>
__device__ __forceinline__ void far_row(const short* ptr, int (&h)[C], int& hprev) {
    asm volatile("global_load_sshort %0, %1, off" : "=v"(hprev) : "v"(ptr) : "memory");
#pragma unroll
    for (int k = 0; k < C; ++k) asm volatile("global_load_sshort %0, %1, off" : "=v"(h[k]) : "v"(ptr + 1 + k) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < C; ++k) asm volatile("" : "+v"(h[k]));
    asm volatile("" : "+v"(hprev));
}

// DP rows of one copy against the graph.  Lane l owns C adjacent columns (C = 2..8: ceil(copy length / 64)), so a row of up to
// W = 64*C columns is ONE step: one wide LDS read per source row, the candidates of the C cells in registers, an in-lane
// max-plus scan, one cross-lane DPP scan, one wide LDS write, one wide direction-byte store.  Copies longer than W are
// swept in passes of W columns (passes outer, rows inner); the value that leaves a row on the right is handed to the next
// pass through a per-row carry array in HBM.  The graph rows (w.ri) and the carries are streamed 64 rows at a time into
// one register per lane and read with v_readlane, so LDS holds nothing but the ring of the last RING rows.
//
// Cells are kept as X[r][j] = H[r][j] - jl*gap, jl = column within the pass (gap < 0): the horizontal move then costs
// nothing (X[j] = max(A[j], X[j-1]): a plain prefix maximum), the diagonal move adds (match|mismatch) - gap, the vertical
// move adds gap, the row-0 diagonal is one constant per pass -- no per-column term anywhere.  Carries between passes
// are H values (X at local column 0 equals H).  A candidate is (X << 8) | (25 - ordinal); for the horizontal chain the
// low byte of what a cell offers to its right neighbours is replaced by the horizontal code, the lowest of all, so the
// plain signed max also implements "horizontal only if strictly larger" and the winner's low byte is the direction.
template <int C>
__device__ void dp_rows(const PoaWs& w, int N, int m, const int8_t* seq, int lane, int& bs_out, int& br_out)
{
    constexpr int NEG = -(1 << 30);
    constexpr int W = 64 * C;
    constexpr int HCODE = 0;                                     // horizontal: clearing the low byte makes a horizontal offer
    constexpr int DIAG_MATCH = (POA_MATCH - POA_GAP) * 256, DIAG_MIS = (POA_MISMATCH - POA_GAP) * 256;
    const int npass = (m + W - 1) / W;
    const int gp = poa_pitch(m);                                 // row pitch of H and dir in HBM
    const int lp = poa_pitch(m < W ? m : W);                     // row pitch of the LDS ring (one pass wide)
    const int RING = poa_ring(m), rmask = RING - 1;
    const int lm = ((m - 1) % W) / C, km = (m - 1) % C;          // where column m lives in the last pass
    int bs = NEG, br = 0x7fffffff;
    for (int pass = 0; pass < npass; ++pass) {
        const int col0 = pass * W + C * lane;                    // cell k is column col0+k+1, element col0+k+8 of an HBM row
        const bool more = pass + 1 < npass, last = !more;
        int sb[C];                                               // this lane's bases of the copy (100: none / not ACGT)
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int j = col0 + k + 1;
            const int c = j <= m ? (int)seq[j - 1] : 100;
            sb[k] = (c >= 0 && c < 4) ? c : 100;
        }
        const int row0 = (pass * W) * (POA_GAP * 256) + POA_CODE_ROW0;   // row-0 diagonal: X = (j-1) gap + s - jl gap
        const short* cprev = w.carry + (size_t)(pass & 1) * w.cpitch;
        short* cnext = w.carry + (size_t)((pass + 1) & 1) * w.cpitch;
        uint2 blk = 1 + lane <= N ? w.ri[1 + lane] : make_uint2(0, 0);
        int cblk = pass > 0 && 1 + lane <= N ? (int)cprev[1 + lane] : 0;
        int px[C], pcin = 0;                                     // the previous row of this pass, still in registers (see source())
#pragma unroll
        for (int k = 0; k < C; ++k) px[k] = 0;
        for (int rb = 1; rb <= N; rb += 64) {
            const int nr = rb + 64 + lane;
            const uint2 nxt = nr <= N ? w.ri[nr] : make_uint2(0, 0);           // next 64 graph rows, consumed after this block
            const int cnx = pass > 0 && nr <= N ? (int)cprev[nr] : 0;
            const int cnt = N - rb + 1 < 64 ? N - rb + 1 : 64;
            int cob = 0;
            for (int i = 0; i < cnt; ++i) {
                const int r = rb + i;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)blk.x, i), d1 = (uint32_t)__builtin_amdgcn_readlane((int)blk.y, i);
                const int cin = __builtin_amdgcn_readlane(cblk, i);             // H[r][pass*W] = X at local column 0; 0 in the first pass
                const int vb = (int)(int8_t)(d0 & 0xff), np = (int)((d0 >> 8) & 0xf);
                const bool keep = (d0 & 0x8000u) != 0;           // read later from HBM by a far successor
                const bool tolds = (d0 & 0x4000u) != 0;          // read later from the LDS ring (a near source that is not the next row)
                const int p0 = (int)(d0 >> 16), p1 = (int)(d1 & 0xffff), p2 = (int)(d1 >> 16);
                short* cur = lds_ring + (r & rmask) * lp;
                int best[C], ss[C];
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    ss[k] = sb[k] == vb ? DIAG_MATCH : DIAG_MIS;
                    best[k] = ss[k] + row0;
                }
                auto source = [&](int e, int q) {
                    int h[C], hprev;
                    if (q == r - 1) {
                        // the row just computed (the usual source: a chain) is forwarded from registers; going through the
                        // LDS ring would put a write->read round trip on every row's critical path
#pragma unroll
                        for (int k = 0; k < C; ++k) h[k] = px[k];
                        hprev = __builtin_amdgcn_update_dpp(0, px[C - 1], 0x138, 0xf, 0xf, true);     // wave_shr:1
                        hprev = lane == 0 ? pcin : hprev;
                    } else if (r - q < RING) {
                        const short* src = lds_ring + (q & rmask) * lp + C * lane + 8;
                        hprev = src[-1];
                        // element-wise 16-bit LDS reads: sign extension comes with the load, and LDS instructions do not
                        // occupy the VALU (a 128-bit read would cost one unpack instruction per cell)
#pragma unroll
                        for (int k = 0; k < C; ++k) h[k] = src[k];
                    } else {
                        far_row<C>(w.H + (size_t)q * gp + col0 + 7, h, hprev);
                        if (lane == 0 && pass > 0) hprev += W * POA_GAP;     // that element was stored in the previous pass's frame
                    }
                    const int cd = POA_CODE_DIAG - e, cv = POA_GAP * 256 + POA_CODE_VERT - e;
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int up = k == 0 ? hprev : h[k - 1];
                        const int c1 = (up << 8) + ss[k] + cd;
                        const int c2 = (h[k] << 8) + cv;
                        best[k] = c1 > best[k] ? c1 : best[k];
                        best[k] = c2 > best[k] ? c2 : best[k];
                    }
                };
                if (np > 0) source(0, p0);
                if (np > 1) source(1, p1);
                if (np > 2) source(2, p2);
                if (np > 3) {                                    // rare: in-edges beyond the third come from HBM
                    const int vnode = __builtin_amdgcn_readfirstlane(w.order[r - 1]);
                    for (int e = 3; e < np; ++e) source(e, __builtin_amdgcn_readfirstlane(w.rank[w.pred[vnode * POA_MAXP + e]]));
                }
                // horizontal chain = prefix maximum: what the cells to the left offer (their value, horizontal code)
                int run[C];                                      // in-lane exclusive running maximum
                run[0] = NEG;
#pragma unroll
                for (int k = 1; k < C; ++k) {
                    const int y = best[k - 1] & ~0xff;
                    run[k] = run[k - 1] > y ? run[k - 1] : y;
                }
                const int ylast = best[C - 1] & ~0xff;
                const int incl = wave_prefix_max(run[C - 1] > ylast ? run[C - 1] : ylast);
                int excl = __builtin_amdgcn_update_dpp(NEG, incl, 0x138, 0xf, 0xf, false);     // wave_shr:1
                const int xc = cin << 8;
                excl = xc > excl ? xc : excl;
                int fin[C];
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const int a = run[k] > excl ? run[k] : excl;
                    fin[k] = best[k] > a ? best[k] : a;
                }
#pragma unroll
                for (int k = 0; k < C; ++k) px[k] = fin[k] >> 8;
                pcin = cin;
                if (tolds && lane == 0) cur[7] = (short)cin;     // element of local column 0: the left neighbour of cell 0
                if (col0 + 1 <= m) {
                    short* dst = cur + C * lane + 8;
                    uint8_t* dd = w.dir + (size_t)r * gp + col0 + 8;
                    if (tolds) {
#pragma unroll
                        for (int k = 0; k < C; ++k) dst[k] = (short)px[k];             // ds_write_b16 takes the low half: no packing
                    }
                    // direction bytes: C contiguous bytes per lane, as the widest stores their count allows
                    uint32_t d0 = 0, d1 = 0;
#pragma unroll
                    for (int k = 0; k < C && k < 4; ++k) d0 |= ((uint32_t)fin[k] & 0xffu) << (8 * k);
#pragma unroll
                    for (int k = 4; k < C; ++k) d1 |= ((uint32_t)fin[k] & 0xffu) << (8 * (k - 4));
                    if constexpr (C == 8) *(uint2*)dd = make_uint2(d0, d1);
                    else if constexpr (C >= 4) {
                        __builtin_memcpy(dd, &d0, 4);
                        if constexpr (C == 5) dd[4] = (uint8_t)d1;
                        if constexpr (C >= 6) { const uint16_t lo = (uint16_t)d1; __builtin_memcpy(dd + 4, &lo, 2); }
                        if constexpr (C == 7) dd[6] = (uint8_t)(d1 >> 16);
                    } else {
                        const uint16_t lo = (uint16_t)d0;
                        __builtin_memcpy(dd, &lo, 2);
                        if constexpr (C == 3) dd[2] = (uint8_t)(d0 >> 16);
                    }
                    if (keep) {
                        short* hd = w.H + (size_t)r * gp + col0 + 8;
#pragma unroll
                        for (int k = 0; k < C; ++k) hd[k] = (short)px[k];
                    }
                }
                if (keep && pass == 0 && lane == 0) w.H[(size_t)r * gp + 7] = 0;
                if (last) {
                    int hm = fin[0];
#pragma unroll
                    for (int k = 1; k < C; ++k) hm = km == k ? fin[k] : hm;
                    hm >>= 8;                                    // same column for every row: X compares like H
                    if (lane == lm && hm > bs) { bs = hm; br = r; }          // strict >: lowest rank on ties
                } else {
                    const int right = __builtin_amdgcn_readlane(fin[C - 1], 63) >> 8;
                    cob = lane == i ? right + W * POA_GAP : cob;             // as an H value: X at local column W is H - W gap
                }
                asm volatile("" ::: "memory");   // one wave: LDS operations execute in order; only the compiler must not reorder
            }
            if (more && lane < cnt) cnext[rb + lane] = (short)cob;
            blk = nxt; cblk = cnx;
        }
        if (more) phase_sync();
    }
    bs_out = __shfl(bs, lm); br_out = __shfl(br, lm);
}

// returns the new node count, or -1 on overflow
__device__ int poa_add(const PoaWs& w, int N_, int ncap, int mcap, const int8_t* seq, int m_, int lane, unsigned long long* tacc)
{
    // wave-uniform by construction; say so, or every quantity derived from them lives in VGPRs behind exec-mask branches
    const int N = __builtin_amdgcn_readfirstlane(N_), m = __builtin_amdgcn_readfirstlane(m_);
    unsigned long long tlast = 0;
#ifdef CLH_DEBUG_POA
    tlast = __builtin_amdgcn_s_memtime();
#endif
    if (N == 0) {
        if (m > ncap) return -1;
        for (int j = lane; j < m; j += 64) {
            w.base[j] = seq[j]; w.np[j] = j > 0 ? 1 : 0; w.key[j] = (long long)(j + 1) << 20;
            w.aligned[j * 3] = w.aligned[j * 3 + 1] = w.aligned[j * 3 + 2] = -1;
            if (j > 0) { w.pred[j * POA_MAXP] = j - 1; w.pw[j * POA_MAXP] = 1; }
            w.order[j] = j; w.rank[j] = j + 1;
        }
        phase_sync();
        return m;
    }
    int bs = -(1 << 28), br = 0x7fffffff;          // end cell: largest H[r][m], lowest rank on ties
    // ---- DP rows --------------------------------------------------------------------------------------------
    // Graph rows in rank space, built in parallel: base, in-degree, ranks of the first three sources (w.ri, 8 bytes a
    // row; further in-edges are rare and fetched where needed).  Row 0 of the matrix is arithmetic (j*gap).
    const int pitch = poa_pitch(m);
    const int RING = poa_ring(m);
    {
        uint32_t* ri32 = (uint32_t*)w.ri;
#pragma unroll 4
        for (int r = 1 + lane; r <= N; r += 64) {
            const int v = w.order[r - 1];
            const int np = w.np[v];
            uint32_t pr[3] = {0, 0, 0};
            for (int e = 0; e < 3; ++e) if (e < np) pr[e] = (uint32_t)w.rank[w.pred[v * POA_MAXP + e]];
            w.ri[r] = make_uint2((uint32_t)(w.base[v] & 0xff) | ((uint32_t)np << 8) | (pr[0] << 16), pr[1] | (pr[2] << 16));
        }
        phase_sync();
        // a row's H values go to HBM only if some later row reads them from there, i.e. it is a source of a row at
        // least RING ranks further on (bits 14 and 15 of the in-degree byte field are free: in-degree <= 12)
        for (int r = 1 + lane; r <= N; r += 64) {
            const uint2 d = w.ri[r];
            const int np = (int)((d.x >> 8) & 0xf);
            const int p0 = (int)(d.x >> 16), p1 = (int)(d.y & 0xffff), p2 = (int)(d.y >> 16);
            // where will row r read source p from?  the row before it: registers; another recent row: the LDS ring (0x4000);
            // an older one: its row in HBM (0x8000).  Rows nobody reads from memory store nothing but direction bytes.
            auto mark = [&](int q) { if (r - q >= RING) atomicOr(&ri32[q * 2], 0x8000u); else if (r - q >= 2) atomicOr(&ri32[q * 2], 0x4000u); };
            if (np > 0) mark(p0);
            if (np > 1) mark(p1);
            if (np > 2) mark(p2);
            if (np > 3) { const int v = w.order[r - 1]; for (int e = 3; e < np; ++e) mark(w.rank[w.pred[v * POA_MAXP + e]]); }
        }
        phase_sync();
        switch (poa_cols(m)) {            // columns per lane: the smallest that covers the copy in one pass (8 beyond 512)
            case 2: dp_rows<2>(w, N, m, seq, lane, bs, br); break;
            case 3: dp_rows<3>(w, N, m, seq, lane, bs, br); break;
            case 4: dp_rows<4>(w, N, m, seq, lane, bs, br); break;
            case 5: dp_rows<5>(w, N, m, seq, lane, bs, br); break;
            case 6: dp_rows<6>(w, N, m, seq, lane, bs, br); break;
            case 7: dp_rows<7>(w, N, m, seq, lane, bs, br); break;
            default: dp_rows<8>(w, N, m, seq, lane, bs, br); break;
        }
    }
    phase_sync();
    TSTAMP(0);
    const int dpitch = pitch, dcol = 7;                // direction bytes: row pitch, element of column 0
    // ---- walk back (sequential by nature, wave-uniform) -------------------------------------------------------
    // One dependent HBM load per step would cost ~1 us each; instead the lanes hold a 32x32 patch of direction bytes
    // (ranks r0..r0-31, columns j0..j0-31; 16 bytes per lane) and, on the fast path, the LDS graph rows of those ranks,
    // so the chain runs on v_readlane until it leaves the patch (~28 steps).  pn[j] = rank matched to sequence position j (0: inserted base), staged in
    // one register per lane and stored 64 positions at a time; ranks become node ids in parallel afterwards.
    {
        int r = __builtin_amdgcn_readfirstlane(br), j = m;
        int r0 = -64, j0 = -64;
        uint32_t pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0, ri = 0;     // lane l: row r0-(l>>1), columns j0-16*(l&1)-15 .. j0-16*(l&1)
        int buf = 0;
        while (j > 0 && r > 0) {
            int a = r0 - r, b = j0 - j;
            if ((unsigned)a >= 32u || (unsigned)b >= 32u) {
                r0 = r; j0 = j;
                const int rr = r - (lane >> 1), jlo = j - 16 * (lane & 1) - 15;       // lowest column of this lane's 16
                pw0 = pw1 = pw2 = pw3 = 0;
                if (rr >= 1 && jlo + 15 >= 1) {
                    // bytes at columns jlo..jlo+15 (row-internal; columns below 0 belong to the previous row or, for
                    // row 1, to row 0 -- in bounds, never looked at)
                    const uint8_t* src = w.dir + (size_t)rr * dpitch + dcol + jlo;
                    uint32_t t[4];
                    __builtin_memcpy(t, src, 16);
                    pw0 = t[0]; pw1 = t[1]; pw2 = t[2]; pw3 = t[3];
                }
                ri = rr >= 1 ? ((const uint32_t*)w.ri)[rr * 2 + (lane & 1)] : 0u;
                a = 0; b = 0;
            }
            // column j0-b sits in lane 2a+(b>>4) at byte 15-(b&15)
            const int byte = 15 - (b & 15), src_lane = a * 2 + (b >> 4);
            const uint32_t sel = (byte >> 2) == 0 ? pw0 : ((byte >> 2) == 1 ? pw1 : ((byte >> 2) == 2 ? pw2 : pw3));
            const int d = (__builtin_amdgcn_readlane((int)sel, src_lane) >> ((byte & 3) * 8)) & 0xff;
            // code -> move: group 3 diagonal from in-edge e, 2 diagonal from row 0, 1 vertical from in-edge e, 0 horizontal
            const int grp = d >> 4, e = 15 - (d & 15);
            const int kind = grp == 0 ? 3 : (grp == 1 ? 2 : 1);
            int pre = 0;
            if (grp & 1) {                                // an in-edge slot (groups 3 and 1): rank of that source
                if (e == 0) pre = (int)((uint32_t)__builtin_amdgcn_readlane((int)ri, a * 2) >> 16);
                else if (e < 3) {
                    const uint32_t q1 = (uint32_t)__builtin_amdgcn_readlane((int)ri, a * 2 + 1);
                    pre = e == 1 ? (int)(q1 & 0xffff) : (int)(q1 >> 16);
                } else {
                    const int v = w.order[r - 1];
                    pre = __builtin_amdgcn_readfirstlane(w.rank[w.pred[v * POA_MAXP + e]]);
                }
            }
            if (kind == 2) { r = pre; continue; }
            --j;
            buf = lane == (j & 63) ? (kind == 1 ? r : 0) : buf;
            if ((j & 63) == 0) { if (j + lane < m) w.pn[j + lane] = buf; buf = 0; }
            if (kind == 1) r = pre;
        }
        int fill_to = j;
        if (j & 63) { fill_to = j & ~63; const int q = fill_to + lane; if (q < m) w.pn[q] = buf; }
        for (int q = lane; q < fill_to; q += 64) w.pn[q] = 0;
    }
    phase_sync();
    TSTAMP(1);
    // ---- graph update, data-parallel over the sequence positions ---------------------------------------------------
    // Position i touches only its own matched node's aligned set and the in-edge list of the node it ends up using, and
    // the nodes of one path are distinct, so the sequential rule of the specification (oracle poa_add) is evaluated
    // per lane; node ids of new nodes are a prefix count in position order, keys come from the last matched position.
    int n = N, fail = 0;
    {
        // first matched position (keys of leading insertions count back from it)
        int lead = m;
        long long key_fa = 0;
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            const int rk = i < m ? w.pn[i] : 0;
            const unsigned long long mb = __builtin_amdgcn_ballot_w64(rk > 0);
            if (mb) {
                const int t = __builtin_ctzll(mb);
                lead = i0 + t;
                const int rk0 = __builtin_amdgcn_readlane(rk, t);
                key_fa = w.key[w.order[rk0 - 1]];
                break;
            }
        }
        int lmi = -1;                       // last matched position so far, and its node's key
        long long lmk = 0;
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < m;
            const int rk = act ? w.pn[i] : 0;
            const int v = rk > 0 ? w.order[rk - 1] : -1;
            const int b = act ? (int)seq[i] : 0;
            long long kv = 0;
            int use = -1;
            if (v >= 0) {
                kv = w.key[v];
                if (w.base[v] == b) use = v;
                else for (int q = 0; q < 3; ++q) { const int x = w.aligned[v * 3 + q]; if (x >= 0 && w.base[x] == b) { use = x; break; } }
            }
            const bool isnew = act && use < 0;
            const unsigned long long nb = __builtin_amdgcn_ballot_w64(isnew);
            const unsigned long long mb = __builtin_amdgcn_ballot_w64(v >= 0);
            const unsigned long long below = ((unsigned long long)1 << lane) - 1;
            if (isnew) use = n + __builtin_popcountll(nb & below);
            n += __builtin_popcountll(nb);
            // anchor of an inserted base: the last matched position before it
            const unsigned long long mlow = mb & below;
            const int src = mlow ? 63 - __builtin_clzll(mlow) : 0;
            const int klo = __shfl((int)(kv & 0xffffffff), src), khi = __shfl((int)(kv >> 32), src);
            if (isnew && use < ncap) {
                long long key;
                if (v >= 0) key = kv;
                else if (mlow) key = (((long long)khi << 32) | (uint32_t)klo) + (i - (i0 + src));
                else if (lmi >= 0) key = lmk + (i - lmi);
                else if (lead < m) key = key_fa - lead + i;
                else key = ((long long)N << 20) + i + 1;       // max key == N<<20 after a re-rank
                w.base[use] = (int8_t)b; w.np[use] = 0; w.key[use] = key;
                int al0 = -1, al1 = -1, al2 = -1;
                if (v >= 0) {                                  // join the aligned set of v
                    int members[4], nm = 0;
                    members[nm++] = v;
                    for (int q = 0; q < 3; ++q) if (w.aligned[v * 3 + q] >= 0) members[nm++] = w.aligned[v * 3 + q];
                    for (int q = 0; q < nm; ++q) {
                        const int x = members[q];
                        for (int t = 0; t < 3; ++t) if (w.aligned[x * 3 + t] < 0) { w.aligned[x * 3 + t] = use; break; }
                        if (q == 0) al0 = x; else if (q == 1) al1 = x; else if (q == 2) al2 = x;
                    }
                }
                w.aligned[use * 3] = al0; w.aligned[use * 3 + 1] = al1; w.aligned[use * 3 + 2] = al2;
            }
            if (act) w.pj[i] = use;
            if (mb) {
                const int t = 63 - __builtin_clzll(mb);
                lmi = i0 + t;
                lmk = ((long long)__shfl((int)(kv >> 32), t) << 32) | (uint32_t)__shfl((int)(kv & 0xffffffff), t);
            }
        }
        if (n > ncap) return -1;
        phase_sync();
        for (int i = 1 + lane; i < m; i += 64) {
            const int u = w.pj[i - 1], x = w.pj[i];
            const int cnt = w.np[x];
            int found = -1;
            for (int e = 0; e < cnt; ++e) if (w.pred[x * POA_MAXP + e] == u) { found = e; break; }
            if (found >= 0) w.pw[x * POA_MAXP + found] += 1;
            else if (cnt >= POA_MAXP) fail = 1;
            else { w.pred[x * POA_MAXP + cnt] = u; w.pw[x * POA_MAXP + cnt] = 1; w.np[x] = (int8_t)(cnt + 1); }
        }
        if (__builtin_amdgcn_ballot_w64(fail != 0)) return -1;
    }
    phase_sync();
    TSTAMP(2);
    poa_rerank(w, N, n, lane);
    TSTAMP(3);
    return n;
}

// heaviest path.  The pass over the rows in rank order is a dependent chain (the score of a source decides ties between
// equally heavy in-edges), so it runs wave-uniformly -- but on a rank-space image of the graph built in parallel (w.tab,
// 3 words a row: in-degree, up to 3 source ranks and their weights) and streamed 64 rows at a time into registers, with
// the scores in LDS and the previous row's score forwarded in a register, so the chain never waits for HBM.  Rows with
// more than 3 in-edges or a weight above 255 fetch their lists from HBM (rare).  Back pointers leave through a lane
// buffer; the final chase reads them back 64 ranks at a time.
__device__ int poa_consensus(const PoaWs& w, int N_, int8_t* out, int cap, int lane)
{
    const int N = __builtin_amdgcn_readfirstlane(N_);
    if (N <= POA_LDS_SCORES) {
        int* score = (int*)poa_lds;
#pragma unroll 4
        for (int r = 1 + lane; r <= N; r += 64) {
            const int v = w.order[r - 1];
            int np = w.np[v];
            uint32_t pr[3] = {0, 0, 0}, wt[3] = {0, 0, 0};
            bool wide = np > 3;
            for (int e = 0; e < 3; ++e) if (e < np) { pr[e] = (uint32_t)w.rank[w.pred[v * POA_MAXP + e]]; wt[e] = (uint32_t)w.pw[v * POA_MAXP + e]; wide |= wt[e] > 255u; }
            w.tab[r * 3 + 0] = (uint32_t)(wide ? 0x7f : np) | (pr[0] << 16);
            w.tab[r * 3 + 1] = pr[1] | (pr[2] << 16);
            w.tab[r * 3 + 2] = (wt[0] & 0xff) | ((wt[1] & 0xff) << 8) | ((wt[2] & 0xff) << 16);
        }
        if (lane == 0) score[0] = 0;
        phase_sync();
        int top = 0, tops = -1, prev = 0;                 // prev = score of rank r-1
        uint32_t b0 = 0, b1 = 0, b2 = 0;
        if (1 + lane <= N) { b0 = w.tab[(1 + lane) * 3]; b1 = w.tab[(1 + lane) * 3 + 1]; b2 = w.tab[(1 + lane) * 3 + 2]; }
        for (int rb = 1; rb <= N; rb += 64) {
            const int nr = rb + 64 + lane;
            uint32_t n0 = 0, n1 = 0, n2 = 0;
            if (nr <= N) { n0 = w.tab[nr * 3]; n1 = w.tab[nr * 3 + 1]; n2 = w.tab[nr * 3 + 2]; }
            const int cnt = N - rb + 1 < 64 ? N - rb + 1 : 64;
            int bpb = 0;
            for (int i = 0; i < cnt; ++i) {
                const int r = rb + i;
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, i), t1 = (uint32_t)__builtin_amdgcn_readlane((int)b1, i), t2 = (uint32_t)__builtin_amdgcn_readlane((int)b2, i);
                const int np = (int)(t0 & 0x7f);
                int bw = -1, bsrc = 0, bscore = 0;
                if (np == 0x7f) {
                    const int v = w.order[r - 1];
                    const int cn = w.np[v];
                    for (int e = 0; e < cn; ++e) {
                        const int u = __builtin_amdgcn_readfirstlane(w.rank[w.pred[v * POA_MAXP + e]]);
                        const int wt = __builtin_amdgcn_readfirstlane(w.pw[v * POA_MAXP + e]);
                        const int su = u == r - 1 ? prev : __builtin_amdgcn_readfirstlane(score[u]);
                        if (wt > bw || (wt == bw && su > bscore)) { bw = wt; bsrc = u; bscore = su; }
                    }
                } else {
                    const int p0 = (int)(t0 >> 16), p1 = (int)(t1 & 0xffff), p2 = (int)(t1 >> 16);
                    if (np > 0) { const int su = p0 == r - 1 ? prev : __builtin_amdgcn_readfirstlane(score[p0]); bw = (int)(t2 & 0xff); bsrc = p0; bscore = su; }
                    if (np > 1) { const int su = p1 == r - 1 ? prev : __builtin_amdgcn_readfirstlane(score[p1]); const int wt = (int)((t2 >> 8) & 0xff); if (wt > bw || (wt == bw && su > bscore)) { bw = wt; bsrc = p1; bscore = su; } }
                    if (np > 2) { const int su = p2 == r - 1 ? prev : __builtin_amdgcn_readfirstlane(score[p2]); const int wt = (int)((t2 >> 16) & 0xff); if (wt > bw || (wt == bw && su > bscore)) { bw = wt; bsrc = p2; bscore = su; } }
                }
                const int sc = bsrc > 0 ? bw + bscore : 0;
                score[r] = sc;
                bpb = lane == i ? bsrc : bpb;                 // back pointer (rank, 0 = none)
                prev = sc;
                if (sc >= tops) { tops = sc; top = r; }       // ties: larger rank
                asm volatile("" ::: "memory");
            }
            if (lane < cnt) w.bp[rb + lane] = bpb;
            b0 = n0; b1 = n1; b2 = n2;
        }
        phase_sync();
        // chase the back pointers: lane l holds bp[c0 - l]; the path descends a few ranks per step
        int len = 0, r = top, c0 = -1000, pb = 0, blk = 0;
        while (r > 0) {
            int off = c0 - r;
            if ((unsigned)off >= 64u) { c0 = r; blk = r - lane >= 1 ? w.bp[r - lane] : 0; off = 0; }
            pb = lane == (len & 63) ? r : pb;
            ++len;
            if ((len & 63) == 0) w.pn[len - 64 + lane] = pb;
            r = __builtin_amdgcn_readlane(blk, off);
        }
        if (len & 63) { const int q = (len & ~63) + lane; if (q < len) w.pn[q] = pb; }
        if (len > cap) return -1;
        phase_sync();
        for (int k = lane; k < len; k += 64) out[len - 1 - k] = w.base[w.order[w.pn[k] - 1]];
        return len;
    }
    int top = -1, tops = -1;
    for (int r = 1; r <= N; ++r) {       // wave-uniform sequential pass
        const int v = w.order[r - 1];
        const int np = w.np[v];
        int bw = -1, bsrc = -1, bscore = 0;
        for (int e = 0; e < np; ++e) {
            const int u = w.pred[v * POA_MAXP + e], wt = w.pw[v * POA_MAXP + e];
            const int su = w.score[u];
            if (wt > bw || (wt == bw && su > bscore)) { bw = wt; bsrc = u; bscore = su; }
        }
        const int sc = bsrc >= 0 ? bw + bscore : 0;
        w.bp[v] = bsrc; w.score[v] = sc;
        __syncthreads();
        if (sc >= tops) { tops = sc; top = v; }
    }
    int len = 0;
    for (int v = top; v >= 0; v = w.bp[v]) ++len;
    if (len > cap) return -1;
    int k = len;
    for (int v = top; v >= 0; v = w.bp[v]) { --k; if (lane == 0) out[k] = w.base[v]; }
    return len;
}

__global__ void __launch_bounds__(64, 5) poa_consensus_kernel(const CcsParams p)
{
    const int lane = threadIdx.x & 63;
    uint8_t* slot = p.poa_ws + (size_t)blockIdx.x * p.slot_bytes;
    for (;;) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(p.work_counter, 1);
        idx = __builtin_amdgcn_readfirstlane(idx);    // wave-uniform in the compiler's eyes too: scalar loads, scalar branches, SGPR pointers below
        if (idx >= p.n) break;
        const int rd = p.work_order ? p.work_order[idx] : idx;
        const int64_t off = p.read_off[rd];
        const int L = (int)(p.read_off[rd + 1] - off);
        const int8_t* seq = p.reads + off;
        if (p.tier == 1 && __builtin_amdgcn_readfirstlane(p.results[rd].status) != 1) continue;   // second tier: only what did not fit a first-tier slot
        const CcsScan sc = p.scan[rd];
        CcsResult res;
        const int period = __builtin_amdgcn_readfirstlane(sc.period);
        res.nseg = 0; res.ccs_len = 0; res.period = period; res.status = 0;
        if (period == 0) { if (lane == 0) p.results[rd] = res; continue; }
        // copies
        int nseg = 0, b = 0, maxlen = 0, total = 0;
        const int ncuts = __builtin_amdgcn_readfirstlane(sc.ncuts);
        for (int i = 0; i < ncuts; ++i) {
            const int cut = __builtin_amdgcn_readfirstlane(sc.cuts[i]);
            if (lane == 0) { p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg] = b; p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg + 1] = cut; }
            const int len = cut - b;
            maxlen = len > maxlen ? len : maxlen; total += len;
            b = cut; ++nseg;
        }
        // period < 0: explicit copies (poa API), keep any tail.  A scan that stopped at its cap leaves the rest of the read out.
        if ((L - b >= CCS_MIN_TAIL && (period < 0 || ncuts < CCS_MAX_CUTS)) || (period < 0 && L > b)) {
            if (lane == 0) { p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg] = b; p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg + 1] = L; }
            const int len = L - b;
            maxlen = len > maxlen ? len : maxlen; total += len;
            ++nseg;
        }
        const int ncap = total + 8, mcap = maxlen + 1;
        if (maxlen > POA_MAX_COPY) { res.status = 4; if (lane == 0) p.results[rd] = res; continue; }   // cells are int16
        // workspace: this wave's slot, or -- a read that needs more -- one of the large slots, claimed for the duration of
        // the read; none free (or none large enough): status 1, the second launch over the large slots takes the read
        uint8_t* ws = slot;
        int big = -1;
        const size_t need = poa_slot_bytes(ncap, mcap);
        if (need > p.slot_bytes) {
            if (p.tier == 0 && p.n_big > 0 && need <= p.big_slot_bytes) {
                if (lane == 0)
                    for (int t = 0; t < 64 && big < 0; ++t) {
                        const int cand = (int)((blockIdx.x * 7u + (unsigned)t * 131u + (unsigned)rd) % (unsigned)p.n_big);
                        if (atomicCAS(&p.big_busy[cand], 0, 1) == 0) big = cand;
                    }
                big = __builtin_amdgcn_readfirstlane(big);
            }
            if (big < 0) { res.status = 1; if (lane == 0) p.results[rd] = res; continue; }
            ws = p.big_ws + (size_t)big * p.big_slot_bytes;
        }
        const PoaWs w = carve(ws, ncap, mcap);
        phase_sync();
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int N = 0;
        b = 0;
        for (int s = 0; s < nseg && N >= 0; ++s) {
            const int e = s < ncuts ? __builtin_amdgcn_readfirstlane(sc.cuts[s]) : L;
            N = poa_add(w, N, ncap, mcap, seq + b, e - b, lane, tacc);
            b = e;
        }
        int len = -1;
        if (N < 0) res.status = 2;
        else {
#ifdef CLH_DEBUG_POA
            unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
            len = poa_consensus(w, N, p.ccs + off, L, lane);
#ifdef CLH_DEBUG_POA
            tacc[4] += __builtin_amdgcn_s_memtime() - tc0;
            if (lane == 0) for (int k = 0; k < 5; ++k) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (55 + k)] = (int)(tacc[k] >> 4);
#endif
            if (len < 0) res.status = 3;
            else { res.nseg = nseg; res.ccs_len = len; }
        }
        if (lane == 0) p.results[rd] = res;
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

}  // namespace clh
