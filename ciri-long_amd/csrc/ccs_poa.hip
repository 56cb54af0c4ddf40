// ccs_poa.hip -- K2 ccs_scan_kernel (repeat period + copy boundaries) and K3 poa_consensus_kernel (partial-order
// alignment of the copies, heaviest-path consensus) for gfx950.
//
// What they replace: pyccs.find_consensus (called at CIRI_long/find_ccs.py:14) and the spoa engine inside it.  Those are
// external packages that exist neither in the reference tree nor in this environment: PARITY UNPINNED.  Both kernels
// implement, bit for bit, the specification written for this project in oracle/ccs_oracle.c ("clh-ccs v1" / "clh-poa v1");
// read that header for every rule and tie-break.  One read per wavefront, one wavefront per workgroup.
//
// K2: 8-mer codes of the read sit in LDS; lane l owns offsets d = d0+l and walks i sequentially, so h[i] is an LDS
//     broadcast and h[i+d] a conflict-free stride-1 read, and no cross-lane reduction is needed for the histogram.  The
//     smoothed maximum, the harmonic test and the per-copy boundary search are wave reductions (shuffles).
// K3: persistent waves pull reads from an atomic counter; each owns a workspace slot in HBM (graph arrays, int16 DP
//     matrix, direction bytes).  A DP row (one graph node) is computed by the 64 lanes over the sequence positions: the
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

// Phase boundary inside one wave that exchanges data between lanes through HBM: complete the stores, then drop the
// CU's L1 so that no line read before the stores can be served stale (buffer_inv sc1; a few microseconds, used a
// handful of times per copy, never per row).
__device__ __forceinline__ void phase_sync() {
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// inclusive prefix maximum over the 64 lanes with DPP (row_shr 1,2,4,8 inside each 16-lane row, then row_bcast 15 and
// 31 to carry the row totals): 12 VALU instructions instead of 6 dependent ds_bpermute round trips through the LDS
__device__ __forceinline__ int wave_prefix_max(int v) {
    constexpr int NEG = -(1 << 28);
#define CLH_DPP_MAX(ctrl, rowmask) { const int o = __builtin_amdgcn_update_dpp(NEG, v, ctrl, rowmask, 0xf, false); v = o > v ? o : v; }
    CLH_DPP_MAX(0x111, 0xf) CLH_DPP_MAX(0x112, 0xf) CLH_DPP_MAX(0x114, 0xf) CLH_DPP_MAX(0x118, 0xf)
    CLH_DPP_MAX(0x142, 0xa) CLH_DPP_MAX(0x143, 0xc)
#undef CLH_DPP_MAX
    return v;
}

__device__ __forceinline__ int wmax_i(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) ccs_scan_kernel(const CcsParams p)
{
    extern __shared__ __attribute__((aligned(16))) int32_t k2_lds[];
    const int lane = threadIdx.x & 63;
    const int rd = blockIdx.x;
    const int64_t off = p.read_off[rd];
    const int L = (int)(p.read_off[rd + 1] - off);
    const int8_t* seq = p.reads + off;
    CcsScan out;
    out.period = 0; out.ncuts = 0; out.support = 0;
    int32_t* h = k2_lds;                 // [Lcap]
    int32_t* sm = k2_lds + p.lcap;       // smoothed counts, [Lcap/2 + 2]
    int32_t* cnt = sm + p.lcap / 2 + 2;  // [Lcap/2 + 2]
    if (L < 2 * CCS_DMIN) { if (lane == 0) p.scan[rd] = out; return; }

    for (int i = lane; i < L; i += 64) {
        int32_t c = 0, ok = i + CCS_K <= L;
        if (ok)
            for (int t = 0; t < CCS_K; ++t) {
                const int b = seq[i + t];
                if (b < 0 || b > 3) ok = 0;
                c = (c << 2) | (b & 3);
            }
        h[i] = ok ? c : -1;
    }
    __syncthreads();

    const int dmax = L / 2;
    for (int d0 = CCS_DMIN; d0 <= dmax; d0 += 64) {
        const int d = d0 + lane;
        int c = 0;
        if (d <= dmax) {
            const int n = L - d;
            for (int i = 0; i < n; ++i) { const int32_t a = h[i]; c += (a >= 0 && a == h[i + d]); }
            cnt[d] = c;
        }
    }
    __syncthreads();
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
    __syncthreads();
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
    const int tol = p0 / 8 > 4 ? p0 / 8 : 4, W = p0 < 96 ? p0 : 96;
    int b = 0, prev = p0, n = 0;
    while (n < CCS_MAX_CUTS) {
        int bs = -1, bdel = 0, bdist = 0x7fffffff;
        for (int c0 = p0 - tol; c0 <= p0 + tol; c0 += 64) {
            const int delta = c0 + lane;
            if (delta <= p0 + tol && delta >= 1 && b + delta <= L) {
                int sc = 0;
                for (int i = b; i < b + W && i + delta < L; ++i) { const int32_t a = h[i]; sc += (a >= 0 && a == h[i + delta]); }
                const int dist = delta > prev ? delta - prev : prev - delta;
                if (sc > bs || (sc == bs && dist < bdist)) { bs = sc; bdel = delta; bdist = dist; }   // ascending delta within the lane
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int s2 = __shfl_xor(bs, d), e2 = __shfl_xor(bdel, d), t2 = __shfl_xor(bdist, d);
            if (s2 > bs || (s2 == bs && (t2 < bdist || (t2 == bdist && e2 < bdel)))) { bs = s2; bdel = e2; bdist = t2; }
        }
        if (bs < 0) break;
        b += bdel;
        prev = bdel;
        if (lane == 0) out.cuts[n] = b;
        ++n;
    }
    if (n >= 2) { out.period = p0; out.ncuts = n; out.support = best; }
    if (lane == 0) p.scan[rd] = out;
}

// ------------------------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------------------------
struct PoaWs {            // views into one wave's workspace slot
    int8_t* base; int8_t* np; int32_t* pred; int32_t* pw; int32_t* aligned; long long* key; int32_t* order; int32_t* rank;
    int32_t* pn; int32_t* pj; int32_t* score; int32_t* bp; short* H; uint8_t* dir;
};

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
    w.bp = (int32_t*)take(sizeof(int32_t) * ncap);
    w.base = (int8_t*)take(ncap);
    w.np = (int8_t*)take(ncap);
    w.H = (short*)take(sizeof(short) * (size_t)(ncap + 1) * mcap);
    w.dir = (uint8_t*)take((size_t)(ncap + 1) * mcap);
    return w;
}

__host__ __device__ inline size_t poa_slot_bytes(int ncap, int mcap)
{
    size_t o = 0;
    auto add = [&](size_t bytes) { o = (o + bytes + 15) & ~(size_t)15; };
    add(sizeof(long long) * ncap); add(sizeof(int32_t) * (size_t)ncap * POA_MAXP); add(sizeof(int32_t) * (size_t)ncap * POA_MAXP);
    add(sizeof(int32_t) * (size_t)ncap * 3); add(sizeof(int32_t) * ncap); add(sizeof(int32_t) * ncap);
    add(sizeof(int32_t) * (size_t)(ncap + mcap + 2)); add(sizeof(int32_t) * (size_t)(ncap + mcap + 2));
    add(sizeof(int32_t) * ncap); add(sizeof(int32_t) * ncap); add(ncap); add(ncap);
    add(sizeof(short) * (size_t)(ncap + 1) * mcap); add((size_t)(ncap + 1) * mcap);
    return o + 64;
}

// merge the nodes created by the last sequence ([n_old, n_new), keys ascending in creation order) into the rank order
__device__ void poa_rerank(const PoaWs& w, int n_old, int n_new, int lane)
{
    const int added = n_new - n_old;
    // old node at rank r (key r<<20): new rank = r + #{new nodes with key < its key}
    for (int r = 1 + lane; r <= n_old; r += 64) {
        const int v = w.order[r - 1];
        const long long k = (long long)r << 20;
        int lo = 0, hi = added;                      // lower_bound over new keys
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (w.key[n_old + mid] < k) lo = mid + 1; else hi = mid; }
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
static constexpr int POA_LDS_ROWS = 1023;          // rows whose graph info fits the LDS table
static constexpr int POA_LDS_SEQ = 2304;           // longest copy staged in LDS
static constexpr int POA_LDS_RING_BYTES = 6144;    // ring of recent H rows
extern __shared__ __attribute__((aligned(16))) uint32_t poa_lds[];
#define lds_rinfo (poa_lds)
#define lds_ring ((short*)(poa_lds + 2 * (POA_LDS_ROWS + 1)))
#define lds_seq ((int8_t*)(poa_lds + 2 * (POA_LDS_ROWS + 1)) + POA_LDS_RING_BYTES)
static constexpr size_t POA_LDS_BYTES = 8 * (POA_LDS_ROWS + 1) + POA_LDS_RING_BYTES + POA_LDS_SEQ;

// returns the new node count, or -1 on overflow
__device__ int poa_add(const PoaWs& w, int N, int ncap, int mcap, const int8_t* seq, int m, int lane, unsigned long long* tacc)
{
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
    const int Wd = m + 1;
    int bs = -(1 << 28), br = 0x7fffffff;          // end cell: largest H[r][m], lowest rank on ties
    // ---- DP rows --------------------------------------------------------------------------------------------
    // Fast path: per-row graph info (base, in-degree, ranks of up to 4 sources), the sequence, and a ring of the last
    // RING rows of H live in LDS, so a row whose sources are recent (the rule: a chain) touches HBM only to store its
    // H row (needed by a far source, rarely) and its direction bytes.  Row 0 is arithmetic (j*gap).
    const int mpad = (m + 1 + 63) & ~63;
    int RING = 16;                                   // power of two, so slot = rank & (RING-1)
    while (RING * mpad > POA_LDS_RING_BYTES / 2) RING >>= 1;
    const int rmask = RING - 1;
    const bool fast = N <= POA_LDS_ROWS && m <= POA_LDS_SEQ && RING >= 2;
    for (int j = lane; j <= m; j += 64) { w.H[j] = (short)(j * POA_GAP); w.dir[j] = 3; }
    if (fast) {
        uint32_t* rinfo = lds_rinfo;
        for (int r = 1 + lane; r <= N; r += 64) {
            const int v = w.order[r - 1];
            const int np = w.np[v];
            uint32_t pr[3] = {0, 0, 0};
            for (int e = 0; e < 3; ++e) if (e < np) pr[e] = (uint32_t)w.rank[w.pred[v * POA_MAXP + e]];
            rinfo[r * 2 + 0] = (uint32_t)(w.base[v] & 0xff) | ((uint32_t)np << 8) | (pr[0] << 16);
            rinfo[r * 2 + 1] = pr[1] | (pr[2] << 16);
        }
        for (int j = lane; j < m; j += 64) lds_seq[j] = seq[j];
        phase_sync();
        // a row's H values go to HBM only if some later row reads them from there, i.e. it is a source of a row at
        // least RING ranks further on (bit 15 of the in-degree byte field is free: in-degree <= 12)
        for (int r = 1 + lane; r <= N; r += 64) {
            const uint32_t d0 = rinfo[r * 2], d1 = rinfo[r * 2 + 1];
            const int np = (int)((d0 >> 8) & 0x7f);
            int pr[POA_MAXP];
            pr[0] = d0 >> 16; pr[1] = d1 & 0xffff; pr[2] = d1 >> 16;
            if (np > 3) { const int v = w.order[r - 1]; for (int e = 3; e < np; ++e) pr[e] = w.rank[w.pred[v * POA_MAXP + e]]; }
            for (int e = 0; e < np; ++e) if (r - pr[e] >= RING) atomicOr(&rinfo[pr[e] * 2], 0x8000u);
        }
        phase_sync();
        for (int r = 1; r <= N; ++r) {
            const uint32_t d0 = rinfo[r * 2], d1 = rinfo[r * 2 + 1];
            const int vb = (int)(int8_t)(d0 & 0xff), np = (int)((d0 >> 8) & 0x7f);
            const bool keep = (d0 & 0x8000u) != 0;    // read later from HBM by a far successor
            int pr[POA_MAXP];
            pr[0] = d0 >> 16; pr[1] = d1 & 0xffff; pr[2] = d1 >> 16;
            if (np > 3) {
                const int v = w.order[r - 1];
                for (int e = 3; e < np; ++e) pr[e] = w.rank[w.pred[v * POA_MAXP + e]];
            }
            short* cur = lds_ring + (r & rmask) * mpad;
            short* Hr = w.H + (size_t)r * Wd;
            uint8_t* dr = w.dir + (size_t)r * Wd;
            int carry = 0;
            for (int j0 = 1; j0 <= m; j0 += 64) {
                const int j = j0 + lane;
                int best = -(1 << 28), bd = 0;
                if (j <= m) {
                    const int sb = lds_seq[j - 1];
                    const int s = (vb == sb && sb < 4) ? POA_MATCH : POA_MISMATCH;
                    for (int e = 0; e < np; ++e) {
                        const int q = pr[e];
                        const int hv = (r - q < RING) ? (int)lds_ring[(q & rmask) * mpad + j - 1] : (int)w.H[(size_t)q * Wd + j - 1];
                        const int c = hv + s;
                        if (c > best) { best = c; bd = 1 | (e << 4); }
                    }
                    { const int c = (j - 1) * POA_GAP + s; if (c > best) { best = c; bd = 1 | (15 << 4); } }
                    for (int e = 0; e < np; ++e) {
                        const int q = pr[e];
                        const int hv = (r - q < RING) ? (int)lds_ring[(q & rmask) * mpad + j] : (int)w.H[(size_t)q * Wd + j];
                        const int c = hv + POA_GAP;
                        if (c > best) { best = c; bd = 2 | (e << 4); }
                    }
                }
                const int x = j <= m ? best - j * POA_GAP : -(1 << 28);
                const int xc = carry - (j0 - 1) * POA_GAP;
                int pm = wave_prefix_max(x);
                if (xc > pm) pm = xc;
                const int hval = pm + j * POA_GAP;
                if (j <= m) {
                    if (hval > best) bd = 3;
                    cur[j] = (short)hval; dr[j] = (uint8_t)bd;
                    if (keep) Hr[j] = (short)hval;
                    if (j == m && hval > bs) { bs = hval; br = r; }
                }
                const int last = j0 + 63 <= m ? 63 : m - j0;
                carry = __builtin_amdgcn_readlane(hval, last);
            }
            if (lane == 0) { cur[0] = 0; dr[0] = 0; if (keep) Hr[0] = 0; }
            asm volatile("" ::: "memory");   // one wave: LDS operations execute in order; only the compiler must not reorder
        }
    } else {
        phase_sync();
        for (int r = 1; r <= N; ++r) {
            const int v = w.order[r - 1];
            const int np = w.np[v];
            const int vb = w.base[v];
            short* Hr = w.H + (size_t)r * Wd;
            uint8_t* dr = w.dir + (size_t)r * Wd;
            int pr[POA_MAXP];
#pragma unroll
            for (int e = 0; e < POA_MAXP; ++e) pr[e] = e < np ? w.rank[w.pred[v * POA_MAXP + e]] : 0;
            int carry = 0;                 // H[v][j0-1] of the previous chunk; H[v][0] = 0
            for (int j0 = 1; j0 <= m; j0 += 64) {
                const int j = j0 + lane;
                int best = -(1 << 28), bd = 0;
                if (j <= m) {
                    const int sb = seq[j - 1];
                    const int s = (vb == sb && sb < 4) ? POA_MATCH : POA_MISMATCH;
#pragma unroll
                    for (int e = 0; e < POA_MAXP; ++e)
                        if (e < np) { const int c = (int)w.H[(size_t)pr[e] * Wd + j - 1] + s; if (c > best) { best = c; bd = 1 | (e << 4); } }
                    { const int c = (int)w.H[j - 1] + s; if (c > best) { best = c; bd = 1 | (15 << 4); } }
#pragma unroll
                    for (int e = 0; e < POA_MAXP; ++e)
                        if (e < np) { const int c = (int)w.H[(size_t)pr[e] * Wd + j] + POA_GAP; if (c > best) { best = c; bd = 2 | (e << 4); } }
                }
                // horizontal chain: H[j] = max(A[j], H[j-1]+g)  ==  max over k<=j of A[k] + (j-k) g, with A[j0-1] := carry
                // scan on X[k] = A[k] - k*g (g < 0): prefix max, then add j*g
                const int x = j <= m ? best - j * POA_GAP : -(1 << 28);
                const int xc = carry - (j0 - 1) * POA_GAP;
                int pm = wave_prefix_max(x);
                if (xc > pm) pm = xc;
                const int hval = pm + j * POA_GAP;
                if (j <= m) {
                    if (hval > best) { bd = 3; }
                    Hr[j] = (short)hval; dr[j] = (uint8_t)bd;
                    if (j == m && hval > bs) { bs = hval; br = r; }
                }
                const int last = j0 + 63 <= m ? 63 : m - j0;
                carry = __builtin_amdgcn_readlane(hval, last);
            }
            if (lane == 0) { Hr[0] = 0; dr[0] = 0; }
            __syncthreads();
        }
    }
    phase_sync();
    TSTAMP(0);
    // the lane that owns column m saw every H[r][m] in rank order (strict > kept the lowest rank)
    {
        const int owner = (m - 1) & 63;
        bs = __shfl(bs, owner); br = __shfl(br, owner);
    }
    // walk back (wave-uniform)
    int npair = 0, r = br, j = m;
    while (j > 0) {
        if (r == 0) { --j; w.pn[npair] = -1; w.pj[npair] = j; ++npair; continue; }
        const int d = w.dir[(size_t)r * Wd + j];
        const int v = w.order[r - 1];
        const int kind = d & 3, e = d >> 4;
        // rank of the e-th source: from the LDS row table when this copy used it, else two dependent HBM loads
        int pre = 0;
        if (kind != 3 && e != 15) {
            if (fast && e < 3) { const uint32_t q0 = lds_rinfo[r * 2], q1 = lds_rinfo[r * 2 + 1]; pre = e == 0 ? (int)(q0 >> 16) : (e == 1 ? (int)(q1 & 0xffff) : (int)(q1 >> 16)); }
            else pre = w.rank[w.pred[v * POA_MAXP + e]];
        }
        if (kind == 1) { --j; w.pn[npair] = v; w.pj[npair] = j; ++npair; r = pre; }
        else if (kind == 2) { r = pre; }
        else if (kind == 3) { --j; w.pn[npair] = -1; w.pj[npair] = j; ++npair; }
        else break;
    }
    phase_sync();
    TSTAMP(1);
    int lead = 0, first_anchor = -1;
    for (int t = npair - 1; t >= 0; --t) { const int q = w.pn[t]; if (q >= 0) { first_anchor = q; break; } ++lead; }
    long long anchor_key = first_anchor >= 0 ? w.key[first_anchor] - (lead + 1) : (long long)N << 20;   // max key == N<<20 after a re-rank
    int n = N, prev_used = -1, since = 0, rc = 0;
    for (int t = npair - 1; t >= 0 && rc == 0; --t) {
        const int b = seq[w.pj[t]];
        const int pnode = w.pn[t];
        int use = -1;
        if (pnode < 0) {
            ++since;
            if (n < ncap) {
                use = n++;
                w.base[use] = (int8_t)b; w.np[use] = 0; w.key[use] = anchor_key + since;
                w.aligned[use * 3] = w.aligned[use * 3 + 1] = w.aligned[use * 3 + 2] = -1;
            }
        } else {
            const int v = pnode;
            anchor_key = w.key[v]; since = 0;
            if (w.base[v] == b) use = v;
            else for (int a = 0; a < 3; ++a) { const int x = w.aligned[v * 3 + a]; if (x >= 0 && w.base[x] == b) { use = x; break; } }
            if (use < 0 && n < ncap) {
                use = n++;
                {
                    w.base[use] = (int8_t)b; w.np[use] = 0; w.key[use] = w.key[v];
                    w.aligned[use * 3] = w.aligned[use * 3 + 1] = w.aligned[use * 3 + 2] = -1;
                    int members[4], nm = 0;
                    members[nm++] = v;
                    for (int a = 0; a < 3; ++a) if (w.aligned[v * 3 + a] >= 0) members[nm++] = w.aligned[v * 3 + a];
                    int slot = 0;
                    for (int q = 0; q < nm; ++q) {
                        const int x = members[q];
                        for (int a = 0; a < 3; ++a) if (w.aligned[x * 3 + a] < 0) { w.aligned[x * 3 + a] = use; break; }
                        if (slot < 3) w.aligned[use * 3 + slot++] = x;
                    }
                }
            }
        }
        __syncthreads();
        if (use < 0) { rc = -1; break; }
        if (prev_used >= 0) {
            const int cnt = w.np[use];
            int found = -1;
            for (int e = 0; e < cnt; ++e) if (w.pred[use * POA_MAXP + e] == prev_used) { found = e; break; }
            if (found >= 0) { const int nw = w.pw[use * POA_MAXP + found] + 1; w.pw[use * POA_MAXP + found] = nw; }
            else if (cnt >= POA_MAXP) rc = -1;
            else { w.pred[use * POA_MAXP + cnt] = prev_used; w.pw[use * POA_MAXP + cnt] = 1; w.np[use] = (int8_t)(cnt + 1); }
            __syncthreads();
        }
        prev_used = use;
    }
    if (rc != 0) return -1;
    phase_sync();
    TSTAMP(2);
    poa_rerank(w, N, n, lane);
    TSTAMP(3);
    return n;
}

__device__ int poa_consensus(const PoaWs& w, int N, int8_t* out, int cap, int lane)
{
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

__global__ void __launch_bounds__(64) poa_consensus_kernel(const CcsParams p)
{
    const int lane = threadIdx.x & 63;
    uint8_t* slot = p.poa_ws + (size_t)blockIdx.x * p.slot_bytes;
    for (;;) {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(p.work_counter, 1);
        idx = __shfl(idx, 0);
        if (idx >= p.n) break;
        const int rd = p.work_order ? p.work_order[idx] : idx;
        const int64_t off = p.read_off[rd];
        const int L = (int)(p.read_off[rd + 1] - off);
        const int8_t* seq = p.reads + off;
        const CcsScan sc = p.scan[rd];
        CcsResult res;
        res.nseg = 0; res.ccs_len = 0; res.period = sc.period; res.status = 0;
        if (sc.period == 0) { if (lane == 0) p.results[rd] = res; continue; }
        // copies
        int nseg = 0, b = 0, maxlen = 0, total = 0;
        for (int i = 0; i < sc.ncuts; ++i) {
            if (lane == 0) { p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg] = b; p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg + 1] = sc.cuts[i]; }
            const int len = sc.cuts[i] - b;
            maxlen = len > maxlen ? len : maxlen; total += len;
            b = sc.cuts[i]; ++nseg;
        }
        if (L - b >= CCS_MIN_TAIL || (sc.period < 0 && L > b)) {   // period < 0: explicit copies (poa API), keep any tail
            if (lane == 0) { p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg] = b; p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * nseg + 1] = L; }
            const int len = L - b;
            maxlen = len > maxlen ? len : maxlen; total += len;
            ++nseg;
        }
        const int ncap = total + 8, mcap = maxlen + 1;
        if (poa_slot_bytes(ncap, mcap) > p.slot_bytes) { res.status = 1; if (lane == 0) p.results[rd] = res; continue; }
        const PoaWs w = carve(slot, ncap, mcap);
        phase_sync();
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int N = 0;
        b = 0;
        for (int s = 0; s < nseg && N >= 0; ++s) {
            const int e = s < sc.ncuts ? sc.cuts[s] : L;
            N = poa_add(w, N, ncap, mcap, seq + b, e - b, lane, tacc);
#ifdef CLH_DEBUG_POA
            if (lane == 0 && 40 + s < CCS_SEG_CAP) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (40 + s)] = N;
#endif
            b = e;
        }
        if (N < 0) { res.status = 2; if (lane == 0) p.results[rd] = res; continue; }
#ifdef CLH_DEBUG_POA
        if (lane == 0) {
            long long snp = 0, spw = 0, hsh = 0;
            for (int v = 0; v < N; ++v) { snp += w.np[v]; for (int e = 0; e < w.np[v]; ++e) { spw += w.pw[v * POA_MAXP + e]; hsh = hsh * 31 + w.pred[v * POA_MAXP + e] * 7 + w.pw[v * POA_MAXP + e]; } hsh = hsh * 131 + w.rank[v]; }
            p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * 50] = (int)snp; p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * 51] = (int)spw; p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * 52] = (int)(hsh & 0x7fffffff);
        }
#endif
#ifdef CLH_DEBUG_POA
        unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
        const int len = poa_consensus(w, N, p.ccs + off, L, lane);
#ifdef CLH_DEBUG_POA
        tacc[4] += __builtin_amdgcn_s_memtime() - tc0;
        if (lane == 0) for (int k = 0; k < 5; ++k) p.segs[(size_t)rd * 2 * CCS_SEG_CAP + 2 * (55 + k)] = (int)(tacc[k] >> 4);
#endif
        if (len < 0) { res.status = 3; if (lane == 0) p.results[rd] = res; continue; }
        res.nseg = nseg; res.ccs_len = len;
        if (lane == 0) p.results[rd] = res;
        __syncthreads();
    }
}

hipError_t launch_ccs_scan(const CcsParams& p, hipStream_t stream)
{
    const size_t lds = sizeof(int32_t) * ((size_t)p.lcap + 2 * ((size_t)p.lcap / 2 + 2));
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)ccs_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(ccs_scan_kernel, dim3(p.n), dim3(64), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_poa(const CcsParams& p, int nslots, hipStream_t stream)
{
    hipLaunchKernelGGL(poa_consensus_kernel, dim3(nslots), dim3(64), POA_LDS_BYTES, stream, p);
    return hipGetLastError();
}

size_t poa_slot_bytes_host(int ncap, int mcap) { return poa_slot_bytes(ncap, mcap); }

}  // namespace clh
