// splice_scan.hip -- K6: splice-signal search around candidate back-splice junctions, on the resident genome (gfx950).
//
// What it replaces: per candidate read, CIRI_long/align.py:477-493 (how far the junction slides between identical
// flanks: up to 2 x 100 string slices and comparisons in Python), align.py:571-695 (find_denovo_signal: str.find of
// the donor/acceptor dinucleotides over two windows of <= 262 bases, all pairs of occurrences) and align.py:698-733
// (get_ss_altered_length, sort_ss: four tiers, sorted by four keys) -- SURVEY.md section 8 f4.  The statements it is
// checked against are oracle/splice_oracle.c and the host mirror ciri_long_amd/align.py (both pinned to outputs of the
// reference, tests/golden/make_bsj_golden.py); where the reference's own choice depends on the hash order of a Python
// set (ties in sort_ss) the rule is first-seen order, as in both.
//
// Annotated splice sites (align.py:474-568, the GTF/BED-derived SS_INDEX) are four sorted arrays of genome-wide
// positions (strand x start/end) in HBM; a candidate finds the annotated shifts near its two ends by binary search, as
// 64-bit masks over the 2 x search_length shifts.  Pairs of annotated sites are ranked first (weight from the genome's
// own dinucleotides); if there is none, the annotated shifts join the motif occurrences of the de-novo search.
//
// One candidate per lane: the work is a few hundred byte reads in two 262-base neighbourhoods of a genome that is
// already in HBM (K5), so there is nothing to stage -- consecutive lanes are consecutive candidates and the reads go
// through L2.  The candidates are independent; the launch is bound by the latency of those reads, not by bandwidth.
//
// Next to a contig end the reference does not consult the annotation (align.py:495-496) and searches the motifs in what
// Python's slices return there -- a negative start wraps around, an end beyond the contig is clipped (align.py:575-583);
// the kernel applies the same slice rule.  The flanks are compared on the genome's raw characters (kept resident beside
// the codes), as the reference's string comparison does.  Status 1 is left for invalid coordinates only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

// (donor, acceptor) as 2 x 2-bit codes A0 C1 G2 T3, in the order of align.py:32-45: GT-AG, GC-AG, AT-AC, GT-AC, AT-AG
__device__ __constant__ uint8_t kDonor[5][2] = {{2, 3}, {2, 1}, {0, 3}, {2, 3}, {0, 3}};
__device__ __constant__ uint8_t kAcceptor[5][2] = {{0, 2}, {0, 2}, {0, 1}, {0, 1}, {0, 2}};
__device__ __constant__ uint8_t kWeight[5] = {0, 1, 2, 2, 2};

__device__ __forceinline__ int iabs(int x) { return x < 0 ? -x : x; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// bit k: the sorted array holds g0 + k (0 <= k < span <= 64)
__device__ __forceinline__ unsigned long long site_mask(const int64_t* __restrict__ a, long long n, long long g0, int span)
{
    long long lo = 0, hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (a[mid] < g0) lo = mid + 1; else hi = mid; }
    unsigned long long m = 0;
    for (; lo < n; ++lo) { const long long d = a[lo] - g0; if (d >= span) break; m |= 1ull << d; }
    return m;
}

// sort_ss (align.py:705-733): first tier that accepts the site, that tier's key order, in one integer
__device__ __forceinline__ unsigned long long site_key(int i, int j, int w, int cb, int us_free, int ds_free)
{
    // get_ss_altered_length (align.py:698-702)
    const int alt = iabs(i - j);
    const int clip_alt = imin(iabs(j - i - cb), iabs(j - i + cb));
    const int tot = imin(iabs(i + us_free), iabs(i - ds_free)) + imin(iabs(j + us_free), iabs(j - ds_free));
    unsigned long long key;
    if (alt <= cb) key = (0ull << 60) | ((unsigned long long)clip_alt << 45) | ((unsigned long long)alt << 30) | ((unsigned long long)w << 15);
    else if (-us_free <= i && i <= ds_free && -us_free <= j && j <= ds_free)
        key = (1ull << 60) | ((unsigned long long)alt << 45) | ((unsigned long long)w << 30) | ((unsigned long long)clip_alt << 15);
    else {
        const unsigned long long tier = (-cb <= i && i <= 0 && 0 <= j && j <= cb) ? 2ull : 3ull;
        key = (tier << 60) | ((unsigned long long)w << 45) | ((unsigned long long)alt << 30) | ((unsigned long long)clip_alt << 15);
    }
    return key | (unsigned long long)tot;
}

// ANNO: annotated sites are loaded (the search without them keeps its smaller register file)
template <bool ANNO>
__global__ void __launch_bounds__(256) splice_scan_kernel(const uint8_t* __restrict__ codes, const uint8_t* __restrict__ ascii, const SpliceTask* __restrict__ tasks, int n,
                                                          int search_extra, int shift_threshold, int canonical_flags, SpliceSites sites,
                                                          int32_t* __restrict__ out)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= n) return;
    const SpliceTask t = tasks[tid];
    const uint8_t* g = codes + t.ctg_off;
    const uint8_t* ch = ascii + t.ctg_off;               // the characters, for the string comparisons of align.py:477-493
    const long long L = t.ctg_len, S = t.start, E = t.end;
    const int cb = t.clip_base;
    const bool index_slices = (canonical_flags & 2) != 0;
    const int canonical = canonical_flags & 1;
    int32_t* o = out + 8 * (size_t)tid;
    const int status = (S < 0 || S >= E || E > L) ? 1 : 0;
    int us_free = 0, ds_free = 0;
    if (!status) {
        // align.py:477-493: prefixes of length i after start and after end are equal (i < 100), suffixes before them alike
        for (int i = 1; i < 100; ++i) {
            if (E + i > L) break;
            if (ch[S + i - 1] != ch[E + i - 1]) break;
            ds_free = i;
        }
        for (int j = 1; j < 100; ++j) {
            if (S - j < 0) break;
            if (ch[S - j] != ch[E - j]) break;
            us_free = j;
        }
    }
    const int sl = cb + search_extra;
    const int us_len = sl + us_free, ds_len = sl + ds_free;
    // align.py:495-496: next to a contig end the annotation is not consulted
    const bool edge = S - us_len - 2 < 0 || E + ds_len + 2 > L;
    // the two windows of find_denovo_signal as Python slices them (align.py:575-578): first index, length
    long long ua, nu, da, nd;
    {
        auto pyslice = [&](long long a, long long b, long long& lo, long long& nn) {
            if (index_slices) {      // mappy's Aligner.seq (the main pass' env.GENOME): no sequence (None) for a start outside the contig or
                if (a < 0 || a >= L || a >= b) { lo = 0; nn = -1; return; }      // an empty range; the end is clipped
                lo = a; nn = (b > L ? L : b) - a;
                return;
            }
            if (a < 0) { a += L; if (a < 0) a = 0; } else if (a > L) a = L;
            if (b < 0) { b += L; if (b < 0) b = 0; } else if (b > L) b = L;
            lo = a; nn = b > a ? b - a : 0;
        };
        pyslice(S - us_len - 2, S + ds_len, ua, nu);
        pyslice(E - us_len, E + ds_len + 2, da, nd);
    }
    const bool short_seq = nu < 0 || nd < 0 || nu < ds_len - us_len + 2 || nd < ds_len - us_len + 2;       // align.py:580-583: no search at all
    // shift i <-> index i + us_len of the upstream slice (str.find from index 1: indices 1 .. n-2); downstream alike
    const long long pu = ua + us_len, pd = da + us_len;   // genome position of shift 0 in each slice (S - 2 and E away from the ends)
    o[1] = us_free; o[2] = ds_free;
    int found = 0, b_strand = 0, b_i = 0, b_j = 0, b_motif = 0;
    constexpr bool anno = ANNO;
    const bool use_anno = anno && !edge;
    const int status2 = (!status && use_anno && 2 * sl > 64) ? 1 : status;
    o[0] = status2;
    if (!status2) {
        const int T = cb + shift_threshold;
        unsigned long long best = ~0ull;
        // annotated shifts in [-sl, sl): exon starts are looked up one base further (align.py:507-546); [strand][kind]
        unsigned long long mu[2][2] = {{0, 0}, {0, 0}}, md[2][2] = {{0, 0}, {0, 0}};
        if constexpr (ANNO) if (use_anno) {
            const long long gs = t.ctg_off + S - sl, ge = t.ctg_off + E - sl;
            for (int strand = 0; strand < 2; ++strand) {
                mu[strand][0] = site_mask(sites.pos[2 * strand], sites.n[2 * strand], gs + 1, 2 * sl);
                mu[strand][1] = site_mask(sites.pos[2 * strand + 1], sites.n[2 * strand + 1], gs, 2 * sl);
                md[strand][0] = site_mask(sites.pos[2 * strand], sites.n[2 * strand], ge + 1, 2 * sl);
                md[strand][1] = site_mask(sites.pos[2 * strand + 1], sites.n[2 * strand + 1], ge, 2 * sl);
            }
            // pairs of annotated sites, '+' first, each list = starts then ends, ascending (align.py:520-556)
            for (int strand = 0; strand < 2; ++strand)
                for (int pi = 0; pi < 2; ++pi)
                    for (unsigned long long mi = mu[strand][pi]; mi; mi &= mi - 1) {
                        const int i = __builtin_ctzll(mi) - sl;
                        const uint32_t u0 = g[S + i - 2], u1 = g[S + i - 1];
                        for (int pj = 0; pj < 2; ++pj)
                            for (unsigned long long mj = md[strand][pj]; mj; mj &= mj - 1) {
                                const int j = __builtin_ctzll(mj) - sl;
                                if (iabs(i - j) > T) continue;
                                const uint32_t d0 = g[E + j], d1 = g[E + j + 1];
                                // the genome's own dinucleotides; minus strand: reverse complements, sides swapped
                                // (revcomp() leaves anything but upper-case ACGT as it is: no motif then)
                                int w = 3;
                                if (u0 < 4 && u1 < 4 && d0 < 4 && d1 < 4) {
                                    const uint32_t n0 = strand ? 3u - u1 : d0, n1 = strand ? 3u - u0 : d1;      // donor
                                    const uint32_t a0 = strand ? 3u - d1 : u0, a1 = strand ? 3u - d0 : u1;      // acceptor
                                    for (int m = 0; m < 5; ++m)
                                        if (kDonor[m][0] == n0 && kDonor[m][1] == n1 && kAcceptor[m][0] == a0 && kAcceptor[m][1] == a1) w = kWeight[m];
                                }
                                const unsigned long long key = site_key(i, j, w, cb, us_free, ds_free);
                                if (key < best) { best = key; found = 2; b_strand = strand; b_i = i; b_j = j; b_motif = 0; }
                            }
                    }
        }
        const int lo = 1 - us_len;                        // shifts whose dinucleotide lies inside the two slices
        const int hi_u = (int)nu - 2 - us_len, hi_d = (int)nd - 2 - us_len;      // ds_len away from the contig ends
        const int lo0 = use_anno ? imin(lo, -sl) : lo;    // annotated shifts start at -search_length
        const int host = t.host_mask & 3;
        for (int round = 0; round < 2 && !found && !short_seq; ++round) {
            // host-gene strands first, the other strand(s) only if that finds nothing (align.py:640-695)
            // without a host gene both strands are searched at once
            const int mask = round == 0 ? (host ? host : 3) : (3 & ~host);
            if (round == 1 && host == 0) break;
            for (int strand = 0; strand < 2; ++strand) {    // '+' sorts before '-'
                if (!((mask >> strand) & 1)) continue;
                const unsigned long long au = mu[strand][0] | mu[strand][1], ad = md[strand][0] | md[strand][1];
                const int nm = canonical ? 1 : 5;
                for (int m = 0; m < nm; ++m) {
                    // plus: acceptor upstream, donor downstream; minus: the reverse complements, sides swapped
                    uint32_t u0, u1, d0, d1;
                    if (strand == 0) { u0 = kAcceptor[m][0]; u1 = kAcceptor[m][1]; d0 = kDonor[m][0]; d1 = kDonor[m][1]; }
                    else { u0 = 3u - kDonor[m][1]; u1 = 3u - kDonor[m][0]; d0 = 3u - kAcceptor[m][1]; d1 = 3u - kAcceptor[m][0]; }
                    const int w = kWeight[m];
                    const int hi_i = use_anno ? imax(hi_u, sl - 1) : hi_u, hi_j = use_anno ? imax(hi_d, sl - 1) : hi_d;
                    for (int i = lo0; i <= hi_i; ++i) {
                        // a site: an occurrence of the motif, or an annotated shift of this strand (align.py:612-621)
                        const bool ai = use_anno && i >= -sl && i < sl && ((au >> (i + sl)) & 1);
                        if (!ai && (i < lo || i > hi_u || g[pu + i] != u0 || g[pu + i + 1] != u1)) continue;
                        const int jlo = imax(lo0, i - T), jhi = imin(hi_j, i + T);
                        for (int j = jlo; j <= jhi; ++j) {
                            const bool aj = use_anno && j >= -sl && j < sl && ((ad >> (j + sl)) & 1);
                            if (!aj && (j < lo || j > hi_d || g[pd + j] != d0 || g[pd + j + 1] != d1)) continue;
                            const unsigned long long key = site_key(i, j, w, cb, us_free, ds_free);
                            if (key < best) { best = key; found = 1; b_strand = strand; b_i = i; b_j = j; b_motif = m; }
                        }
                    }
                }
            }
        }
    }
    o[3] = found; o[4] = b_strand; o[5] = b_i; o[6] = b_j; o[7] = b_motif;
}

hipError_t launch_splice_scan(const uint8_t* codes, const uint8_t* ascii, const SpliceTask* tasks, int n, int search_extra, int shift_threshold, int canonical,
                              const SpliceSites& sites, int32_t* out, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (sites.n[0] + sites.n[1] + sites.n[2] + sites.n[3] > 0)
        hipLaunchKernelGGL(splice_scan_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, stream, codes, ascii, tasks, n, search_extra, shift_threshold, canonical, sites, out);
    else
        hipLaunchKernelGGL(splice_scan_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, stream, codes, ascii, tasks, n, search_extra, shift_threshold, canonical, sites, out);
    return hipGetLastError();
}

}  // namespace clh
