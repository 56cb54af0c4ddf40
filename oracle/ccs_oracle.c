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
 *    A copy longer than 2800 bases -> no consensus (limit of v1: 16-bit cells of the alignment kernel).
 * 3. consensus.  Partial-order alignment of the copies in read order ("clh-poa v1", below), heaviest path.
 *
 * Specification "clh-poa v1" (scores from the reference's call sites: match 10, mismatch -4, gap -8; tests/test_poa.py:30)
 * --------------------------
 * Linear gap cost.  Fitting alignment: the sequence is aligned end to end, the graph's ends are free.
 * Rows = nodes in topological order (rank 1..N), row 0 = virtual start with H0[j] = j*gap.
 *   D[v][j] = max over in-edges (p->v) in insertion order, then row 0, of H[p][j-1] + s(v, j)      (strict > keeps first)
 *   V[v][j] = max over in-edges in insertion order of H[p][j] + gap
 *   H[v][0] = 0;  H[v][j] = D; if V > H take V; then if H[v][j-1] + gap > H take it
 * End cell: largest H[v][m], ties to the lowest rank.  Walking back, sequence bases left of the first aligned node
 * are insertions.
 * Adding the path: a base aligned to a node with the same base re-uses it; with another base it re-uses the member of
 * the node's aligned set holding that base, else a new node joins the set; inserted bases get new nodes.  Consecutive
 * used nodes get an edge (weight +1 if present; a node keeps at most 12 in-edges, more is an error -> no consensus).
 * Order keys: a new aligned node takes its partner's key; an inserted node takes key(last aligned node)+t (t-th since
 * then), leading insertions sit just below the first aligned node.  Re-ranking: candidates = nodes sorted by (key, id);
 * that order can violate an edge when a base re-used a member of an aligned set that ranks after the row it was
 * aligned to, so the final order is the depth-first post-order over in-edges (stored order) taken in candidate order
 * (identical to the candidate order whenever that is already topological); then key = rank << 20.
 * Consensus: in rank order, best[v] = in-edge with the largest weight (ties: larger score of its source, then first),
 * score[v] = weight + score[source]; the path ends at the node with the largest score (ties: larger rank) and is
 * followed back through best[] to a node without in-edges.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CCS_DMIN 30
#define CCS_MIN_SUPPORT 12
#define CCS_SMOOTH 3
#define CCS_MAX_CUTS 64
#define CCS_MIN_TAIL 20
#define POA_MAX_COPY 2800
#define POA_MAXP 12
#define POA_MATCH 10
#define POA_MISMATCH (-4)
#define POA_GAP (-8)

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

/* ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int n, cap;
    int8_t *base;
    int8_t *np;
    int32_t *pred;      /* [cap][POA_MAXP] */
    int32_t *pw;        /* [cap][POA_MAXP] */
    int32_t *aligned;   /* [cap][3], -1 = none */
    int64_t *key;
    int32_t *order;     /* rank-1 -> node */
    int32_t *rank;      /* node -> rank (1..n) */
} poa_graph;

static void g_init(poa_graph *g, int cap)
{
    g->n = 0; g->cap = cap;
    g->base = (int8_t *)malloc((size_t)cap);
    g->np = (int8_t *)calloc((size_t)cap, 1);
    g->pred = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap * POA_MAXP);
    g->pw = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap * POA_MAXP);
    g->aligned = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap * 3);
    g->key = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    g->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    g->rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
}
static void g_free(poa_graph *g)
{
    free(g->base); free(g->np); free(g->pred); free(g->pw); free(g->aligned); free(g->key); free(g->order); free(g->rank);
}
static int g_new(poa_graph *g, int base, int64_t key)
{
    if (g->n >= g->cap) return -1;
    int v = g->n++;
    g->base[v] = (int8_t)base; g->np[v] = 0; g->key[v] = key;
    g->aligned[v * 3] = g->aligned[v * 3 + 1] = g->aligned[v * 3 + 2] = -1;
    return v;
}
static int g_edge(poa_graph *g, int u, int v)
{
    for (int e = 0; e < g->np[v]; ++e)
        if (g->pred[v * POA_MAXP + e] == u) { g->pw[v * POA_MAXP + e] += 1; return 0; }
    if (g->np[v] >= POA_MAXP) return -1;
    g->pred[v * POA_MAXP + g->np[v]] = u;
    g->pw[v * POA_MAXP + g->np[v]] = 1;
    g->np[v] += 1;
    return 0;
}
static poa_graph *g_sort_ctx;
static int g_cmp(const void *a, const void *b)
{
    int x = *(const int32_t *)a, y = *(const int32_t *)b;
    if (g_sort_ctx->key[x] != g_sort_ctx->key[y]) return g_sort_ctx->key[x] < g_sort_ctx->key[y] ? -1 : 1;
    return x < y ? -1 : (x > y);
}
static void g_rerank(poa_graph *g)
{
    const int n = g->n;
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *stk = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *sti = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int8_t *seen = (int8_t *)calloc((size_t)n, 1);
    for (int i = 0; i < n; ++i) cand[i] = i;
    g_sort_ctx = g;
    qsort(cand, (size_t)n, sizeof(int32_t), g_cmp);
    int out = 0;
    for (int c = 0; c < n; ++c) {
        if (seen[cand[c]]) continue;
        int sp = 0;
        stk[0] = cand[c]; sti[0] = 0; seen[cand[c]] = 1;
        while (sp >= 0) {
            const int u = stk[sp];
            if (sti[sp] < g->np[u]) {
                const int pr = g->pred[u * POA_MAXP + sti[sp]];
                sti[sp] += 1;
                if (!seen[pr]) { seen[pr] = 1; ++sp; stk[sp] = pr; sti[sp] = 0; }
            } else {
                g->order[out++] = u;
                --sp;
            }
        }
    }
    for (int r = 0; r < n; ++r) { g->rank[g->order[r]] = r + 1; g->key[g->order[r]] = (int64_t)(r + 1) << 20; }
    free(cand); free(stk); free(sti); free(seen);
}

/* align seq (codes, length m) to the graph and add it.  returns 0, or -1 on capacity/in-degree overflow */
static int poa_add(poa_graph *g, const int8_t *seq, int m)
{
    if (g->n == 0) {
        for (int j = 0; j < m; ++j) {
            int v = g_new(g, seq[j], (int64_t)(j + 1) << 20);
            if (v < 0) return -1;
            if (j > 0 && g_edge(g, v - 1, v) != 0) return -1;
        }
        g_rerank(g);
        return 0;
    }
    const int N = g->n, Wd = m + 1;
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1) * Wd);
    uint8_t *dir = (uint8_t *)malloc((size_t)(N + 1) * Wd);   /* 0 start, 1 diag, 2 vert, 3 horiz; high nibble = in-edge slot (15 = row 0) */
    for (int j = 0; j <= m; ++j) { H[j] = j * POA_GAP; dir[j] = 3; }
    for (int r = 1; r <= N; ++r) {
        const int v = g->order[r - 1];
        int32_t *Hr = H + (size_t)r * Wd;
        uint8_t *dr = dir + (size_t)r * Wd;
        Hr[0] = 0; dr[0] = 0;
        for (int j = 1; j <= m; ++j) {
            const int s = (g->base[v] == seq[j - 1] && seq[j - 1] < 4) ? POA_MATCH : POA_MISMATCH;
            int best = INT32_MIN, bd = 0;
            for (int e = 0; e < g->np[v]; ++e) {
                const int pr = g->rank[g->pred[v * POA_MAXP + e]];
                const int c = H[(size_t)pr * Wd + j - 1] + s;
                if (c > best) { best = c; bd = 1 | (e << 4); }
            }
            { const int c = H[j - 1] + s; if (c > best) { best = c; bd = 1 | (15 << 4); } }
            for (int e = 0; e < g->np[v]; ++e) {
                const int pr = g->rank[g->pred[v * POA_MAXP + e]];
                const int c = H[(size_t)pr * Wd + j] + POA_GAP;
                if (c > best) { best = c; bd = 2 | (e << 4); }
            }
            { const int c = Hr[j - 1] + POA_GAP; if (c > best) { best = c; bd = 3; } }
            Hr[j] = best; dr[j] = (uint8_t)bd;
        }
    }
    int br = 1, bs = H[(size_t)1 * Wd + m];
    for (int r = 2; r <= N; ++r) if (H[(size_t)r * Wd + m] > bs) { bs = H[(size_t)r * Wd + m]; br = r; }

    /* walk back: pairs (node or -1, seq index) in reverse */
    int32_t *pn = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + m + 2));
    int32_t *pj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + m + 2));
    int np_ = 0, r = br, j = m;
    while (j > 0) {
        if (r == 0) { pn[np_] = -1; pj[np_++] = --j; continue; }
        const uint8_t d = dir[(size_t)r * Wd + j];
        const int v = g->order[r - 1];
        if ((d & 3) == 1) { pn[np_] = v; pj[np_++] = j - 1; --j; const int e = d >> 4; r = e == 15 ? 0 : g->rank[g->pred[v * POA_MAXP + e]]; }
        else if ((d & 3) == 2) { const int e = d >> 4; r = g->rank[g->pred[v * POA_MAXP + e]]; }
        else if ((d & 3) == 3) { pn[np_] = -1; pj[np_++] = j - 1; --j; }
        else break;   /* H[v][0]: start of the aligned part; remaining bases (none, j == 0) */
    }
    free(H); free(dir);

    /* first aligned node ahead (for the keys of leading insertions) */
    int lead = 0, first_anchor = -1;
    for (int t = np_ - 1; t >= 0; --t) { if (pn[t] >= 0) { first_anchor = pn[t]; break; } ++lead; }
    int64_t maxkey = 0;
    for (int v = 0; v < N; ++v) if (g->key[v] > maxkey) maxkey = g->key[v];

    int prev_used = -1, since = 0, rc = 0;
    int64_t anchor_key = first_anchor >= 0 ? g->key[first_anchor] - (lead + 1) : maxkey;
    for (int t = np_ - 1; t >= 0 && rc == 0; --t) {
        const int b = seq[pj[t]];
        int use;
        if (pn[t] < 0) {
            ++since;
            use = g_new(g, b, anchor_key + since);
        } else {
            const int v = pn[t];
            anchor_key = g->key[v]; since = 0;
            use = -1;
            if (g->base[v] == b) use = v;
            else for (int a = 0; a < 3; ++a) { const int w = g->aligned[v * 3 + a]; if (w >= 0 && g->base[w] == b) { use = w; break; } }
            if (use < 0) {
                use = g_new(g, b, g->key[v]);
                if (use >= 0) {
                    /* join the aligned set of v: every member learns the new node and vice versa */
                    int members[4], nm = 0;
                    members[nm++] = v;
                    for (int a = 0; a < 3; ++a) if (g->aligned[v * 3 + a] >= 0) members[nm++] = g->aligned[v * 3 + a];
                    int slot = 0;
                    for (int q = 0; q < nm; ++q) {
                        const int w = members[q];
                        for (int a = 0; a < 3; ++a) if (g->aligned[w * 3 + a] < 0) { g->aligned[w * 3 + a] = use; break; }
                        if (slot < 3) g->aligned[use * 3 + slot++] = w;
                    }
                }
            }
        }
        if (use < 0) { rc = -1; break; }
        if (prev_used >= 0 && g_edge(g, prev_used, use) != 0) rc = -1;
        prev_used = use;
    }
    free(pn); free(pj);
    if (rc == 0) g_rerank(g);
    return rc;
}

static int poa_consensus(poa_graph *g, int8_t *out, int cap)
{
    const int N = g->n;
    int32_t *score = (int32_t *)calloc((size_t)N, sizeof(int32_t));
    int32_t *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    int top = -1, tops = -1;
    for (int r = 1; r <= N; ++r) {
        const int v = g->order[r - 1];
        int bw = -1, bsrc = -1;
        for (int e = 0; e < g->np[v]; ++e) {
            const int u = g->pred[v * POA_MAXP + e], w = g->pw[v * POA_MAXP + e];
            if (w > bw || (w == bw && score[u] > score[bsrc])) { bw = w; bsrc = u; }
        }
        bp[v] = bsrc;
        score[v] = bsrc >= 0 ? bw + score[bsrc] : 0;
        if (score[v] >= tops) { tops = score[v]; top = v; }     /* ties: larger rank */
    }
    int len = 0;
    for (int v = top; v >= 0; v = bp[v]) ++len;
    if (len > cap) { free(score); free(bp); return -1; }
    int k = len;
    for (int v = top; v >= 0; v = bp[v]) out[--k] = g->base[v];
    free(score); free(bp);
    return len;
}

/* consensus of nseq sequences (packed codes, offsets[nseq+1]).  returns length or -1 */
int clo_poa_consensus(int32_t nseq, const int8_t *seqs, const int32_t *off, int8_t *out, int32_t cap)
{
    int total = off[nseq];
    poa_graph g;
    g_init(&g, total + 8);
    int rc = 0;
    for (int s = 0; s < nseq && rc == 0; ++s) rc = poa_add(&g, seqs + off[s], off[s + 1] - off[s]);
    int len = rc == 0 ? poa_consensus(&g, out, cap) : -1;
    g_free(&g);
    return len;
}

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
    /* limit of specification v1: a copy longer than POA_MAX_COPY bases (a period above ~2.5 kb) yields no consensus */
    for (int i = 0; i < n; ++i) if (segs[2 * i + 1] - segs[2 * i] > POA_MAX_COPY) return 0;
    *nseg = n;
    int32_t *off = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int8_t *buf = (int8_t *)malloc((size_t)L + 1);
    off[0] = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(buf + off[i], seq + segs[2 * i], (size_t)(segs[2 * i + 1] - segs[2 * i]));
        off[i + 1] = off[i] + segs[2 * i + 1] - segs[2 * i];
    }
    const int len = clo_poa_consensus(n, buf, off, ccs, cap);
    free(off); free(buf);
    if (len < 0) { *nseg = 0; return -1; }
    return len;
}
