/*
 * ccs_oracle.c -- CPU statement of the cyclic-consensus step (find_consensus) as THIS repository defines it.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ssw_oracle.c).
 *
 * PARITY UNPINNED.  CIRI-long delegates this step to the external packages pyccs (find_consensus) and spoa (poa)
 * (CIRI_long/find_ccs.py:8,14; setup.py:57).  Neither their sources nor their wheels exist in /root/reference or in
 * this environment, and the reference's only test of them asserts one length (tests/test_poa.py:32).  There is no file
 * to restate and no output to compare with.  What follows is therefore a specification written for this project from
 * the published description of the method (k-mer self-matches give the repeat period; the read is cut into copies;
 * the copies are combined by partial-order alignment), fixed in every detail so that the HIP kernels (K2 ccs_scan, K3
 * poa_consensus) can be checked bit for bit against it.  It reproduces the CONTRACT the reference consumes:
 * `(segments, ccs)` with segments "s0-e0;s1-e1;..." ascending on the raw read, or (None, None)
 * (find_ccs.py:14-16,94; find_bsj.py:254-255).
 *
 * Specification "clh-ccs v1"
 * --------------------------
 * codes: A0 C1 G2 T3, anything else 4.  A k-mer is valid if it holds no code 4.
 * 1. period.  k = 8 (4^8 codes: a 1 kb read has ~0.01 chance matches per offset, while 13 % sequencing error leaves
 *    ~10 % of true 8-mer pairs intact).  cnt[d] = #{ i : kmer(i), kmer(i+d) valid and equal }, d in [30, L/2];
 *    s[d] = sum of cnt over [d-3, d+3] (clipped to the range); p0 = smallest d with maximal s[d].
 *    s[p0] < 12 -> there is no repeat.  Harmonics: for q = 2..8 while (p0+q/2)/q >= 30, let e be the offset with the
 *    largest s in [(p0+q/2)/q - 3, +3] (smallest on ties); if 2*s[e] >= s[p0] the period is e; the largest such q wins
 *    (noise can lift 2p or 3p slightly above p).
 * 2. copies.  tol = max(4, p0/8), W = min(p0, 96).  b = 0, prev = p0.  Repeat: candidates delta in
 *    [p0-tol, p0+tol] with b+delta <= L; none -> stop.  score(delta) = #{ i in [b, b+W) : kmer(i), kmer(i+delta) valid
 *    and equal } (same k as step 1).  Take the best score; ties: smallest |delta-prev|, then smallest delta; a best
 *    score of 0 takes the candidate closest to prev.  Cut at b+delta, prev = delta.  At most 64 cuts.
 *    Fewer than 2 full copies -> no repeat.  A tail of >= 20 bases after the last cut is kept as a partial copy, unless
 *    the 64-cut cap ended the search (the rest of the read is then an unscanned stretch of copies and is left out).
 *    (Until round 4 a copy longer than 2800 bases gave no consensus -- a limit of the alignment kernel's 16-bit cells written into
 *    the specification.  pyccs has no reason to refuse such a read; the kernel now has a wide form of its pass and the limit is gone.)
 * 3. consensus.  Partial-order alignment of the copies in read order (poa_oracle.c), heaviest bundle, restricted to the
 *    nodes crossed by at least (copies + 1) / 2 of the copies (the partial last copy counts as a copy).
 *
 * Step 3 is spoa.poa(copies, 0, ., 10, -4, -8, -2, -24, -1) -- local alignment, two-piece gap cost, the call of the
 * reference's tests/test_poa.py:30, whose assertion (:32) equates the length of that consensus with the length of
 * find_consensus' -- as stated in poa_oracle.c ("clh-poa v3": a restatement of the published spoa algorithm).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CCS_DMIN 30
#define CCS_MIN_SUPPORT 12
#define CCS_SMOOTH 3
#define CCS_MAX_CUTS 64
#define CCS_MIN_TAIL 20

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* k-mer codes; -1 = invalid */
static void kmer_codes(const int8_t *seq, int L, int k, int32_t *h)
{
    for (int i = 0; i < L; ++i) h[i] = -1;
    for (int i = 0; i + k <= L; ++i) {
        int32_t c = 0, ok = 1;
        for (int t = 0; t < k; ++t) {
            int b = seq[i + t];
            if (b < 0 || b > 3) { ok = 0; break; }
            c = (c << 2) | b;
        }
        h[i] = ok ? c : -1;
    }
}

/* step 1+2.  cuts[0..ncuts) are the boundaries after 0 (b1, b2, ...); returns period or 0 */
int clo_ccs_segments(const int8_t *seq, int32_t L, int32_t *cuts, int32_t *ncuts, int32_t *k_used, int32_t *support)
{
    *ncuts = 0; *k_used = 0; *support = 0;
    if (L < 2 * CCS_DMIN) return 0;
    int32_t *h = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    int32_t *cnt = (int32_t *)calloc((size_t)L / 2 + 2, sizeof(int32_t));
    int p0 = 0, kk = 0;
    const int ks[1] = {8};
    for (int a = 0; a < 1 && !p0; ++a) {
        int k = ks[a];
        kmer_codes(seq, L, k, h);
        int dmax = L / 2;
        for (int d = CCS_DMIN; d <= dmax; ++d) {
            int c = 0;
            for (int i = 0; i + d < L; ++i) c += (h[i] >= 0 && h[i] == h[i + d]);
            cnt[d] = c;
        }
        int32_t *sm = (int32_t *)calloc((size_t)dmax + 2, sizeof(int32_t));
        int best = -1, bestd = 0;
        for (int d = CCS_DMIN; d <= dmax; ++d) {
            int s = 0;
            for (int e = imax(d - CCS_SMOOTH, CCS_DMIN); e <= imin(d + CCS_SMOOTH, dmax); ++e) s += cnt[e];
            sm[d] = s;
            if (s > best) { best = s; bestd = d; }
        }
        if (best >= CCS_MIN_SUPPORT) {
            p0 = bestd; kk = k; *support = best;
            for (int q = 2; q <= 8; ++q) {                     /* harmonics: prefer the fundamental */
                const int c = (bestd + q / 2) / q;
                if (c < CCS_DMIN) break;
                int eb = -1, es = -1;
                for (int e = imax(c - 3, CCS_DMIN); e <= imin(c + 3, dmax); ++e) if (sm[e] > es) { es = sm[e]; eb = e; }
                if (eb >= 0 && 2 * es >= best) p0 = eb;
            }
        }
        free(sm);
    }
    if (!p0) { free(h); free(cnt); return 0; }
    *k_used = kk;
    int tol = imax(4, p0 / 8), W = imin(p0, 96);
    int b = 0, prev = p0, n = 0;
    while (n < CCS_MAX_CUTS) {
        int bestscore = -1, bestdelta = 0;
        for (int delta = p0 - tol; delta <= p0 + tol; ++delta) {
            if (delta < 1 || b + delta > L) continue;
            int sc = 0;
            for (int i = b; i < b + W && i + delta < L; ++i) sc += (h[i] >= 0 && h[i] == h[i + delta]);
            int better = 0;
            if (sc > bestscore) better = 1;
            else if (sc == bestscore) {
                int da = abs(delta - prev), db = abs(bestdelta - prev);
                if (da < db) better = 1;            /* equal distance keeps the smaller (earlier) delta */
            }
            if (better) { bestscore = sc; bestdelta = delta; }
        }
        if (bestscore < 0) break;
        b += bestdelta;
        prev = bestdelta;
        cuts[n++] = b;
    }
    free(h); free(cnt);
    *ncuts = n;
    if (n < 2) { *ncuts = 0; return 0; }
    return p0;
}

/* step 3 lives in poa_oracle.c */
int clo_poa(int32_t nseq, const int8_t *seqs, const int32_t *off, const int32_t *params, int8_t *cons, int32_t cap,
            int8_t *msa, int64_t msa_cap, int32_t *ncols, int32_t *scores);


/* find_consensus: segs[2*i], segs[2*i+1] = start, end of copy i.  returns consensus length, 0 = no repeat, -1 = error */
int clo_find_consensus(const int8_t *seq, int32_t L, int32_t *segs, int32_t *nseg, int8_t *ccs, int32_t cap, int32_t *period)
{
    int32_t cuts[CCS_MAX_CUTS], nc = 0, k = 0, sup = 0;
    *nseg = 0; *period = 0;
    const int p0 = clo_ccs_segments(seq, L, cuts, &nc, &k, &sup);
    if (!p0) return 0;
    *period = p0;
    int n = 0, b = 0;
    for (int i = 0; i < nc; ++i) { segs[2 * n] = b; segs[2 * n + 1] = cuts[i]; b = cuts[i]; ++n; }
    /* the rest of the read is a (partial) last copy -- unless the boundary search stopped at its cap, in which case the rest
       is an unscanned stretch of many copies and is left out */
    if (L - b >= CCS_MIN_TAIL && nc < CCS_MAX_CUTS) { segs[2 * n] = b; segs[2 * n + 1] = L; ++n; }
    *nseg = n;
    int32_t *off = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int8_t *buf = (int8_t *)malloc((size_t)L + 1);
    off[0] = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(buf + off[i], seq + segs[2 * i], (size_t)(segs[2 * i + 1] - segs[2 * i]));
        off[i + 1] = off[i] + segs[2 * i + 1] - segs[2 * i];
    }
    /* local alignment, the scores of tests/test_poa.py:30; nodes crossed by fewer than half of the copies are left out of
       the consensus (the unaligned overhang of a copy cut a few bases early or late would otherwise lead or trail it) */
    const int32_t par[8] = {0, 10, -4, -8, -2, -24, -1, (n + 1) / 2};
    const int len = clo_poa(n, buf, off, par, ccs, cap, NULL, 0, NULL, NULL);
    free(off); free(buf);
    if (len < 0) { *nseg = 0; return -1; }
    return len;
}
