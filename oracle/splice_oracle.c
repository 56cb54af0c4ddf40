/* splice_oracle.c -- CPU statement of the splice-signal step around a candidate back-splice junction.
 *
 * TEST INFRASTRUCTURE ONLY: the checker of K6 (ciri_long_amd/csrc/splice_scan.hip).  Nothing under ciri_long_amd/ may
 * load it.  It restates, on the raw characters of one contig:
 *     CIRI_long/align.py:477-493   how far the junction slides between identical flanks (us_free, ds_free)
 *     CIRI_long/align.py:495-568   find_annotated_signal: pairs of annotated sites near both ends
 *     CIRI_long/align.py:571-695   find_denovo_signal: donor/acceptor dinucleotides (+ annotated shifts), host strand first
 *     CIRI_long/align.py:698-702   get_ss_altered_length
 *     CIRI_long/align.py:705-733   sort_ss: four tiers, four sort keys
 * as called by find_bsj.py:286-301 (search_length = clip_base + 10, shift_threshold = 3).
 * Pinned by tests/test_splice_oracle.py against the outputs of the reference itself (tests/golden/bsj_golden.json.gz,
 * made by tests/golden/make_bsj_golden.py).  Where the reference's choice between equally ranked sites follows the hash
 * order of a Python set, the rule here is first-seen order (strand, motif, upstream shift, downstream shift).
 *
 * Written list-first on purpose (all candidate sites are materialised, then the tiers are filtered and sorted), unlike
 * the kernel, which keeps a running minimum of a packed key.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int kind, strand, i, j, motif, w, alt, clip_alt, tot; } site_t;
typedef struct { site_t* v; int n, cap; } list_t;

static void push(list_t* l, site_t s)
{
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 256; l->v = (site_t*)realloc(l->v, sizeof(site_t) * (size_t)l->cap); }
    l->v[l->n++] = s;
}

static int iabs(int x) { return x < 0 ? -x : x; }
static int imin(int a, int b) { return a < b ? a : b; }

/* (donor, acceptor) -> weight, in the order of align.py:32-45 */
static const char* const DONOR[5] = {"GT", "GC", "AT", "GT", "AT"};
static const char* const ACCEPTOR[5] = {"AG", "AG", "AC", "AC", "AG"};
static const int WEIGHT[5] = {0, 1, 2, 2, 2};

/* utils.revcomp (utils.py:118-120): complements upper-case ACGT only, reverses everything */
static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }
static void revcomp2(const char* in, char* out) { out[0] = comp(in[1]); out[1] = comp(in[0]); }

static int member(const int64_t* a, int64_t n, int64_t v)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo < n && a[lo] == v;
}

/* align.py:507-546: shifts at which an annotated exon start (looked up at pos+1) or end (at pos) of this strand sits;
 * starts first, then ends, each ascending */
static int annotated_shifts(const int64_t* starts, int64_t ns, const int64_t* ends, int64_t ne, int64_t pos0, int sl, int* out)
{
    int m = 0;
    for (int sh = -sl; sh < sl; ++sh) if (member(starts, ns, pos0 + sh + 1)) out[m++] = sh;
    for (int sh = -sl; sh < sl; ++sh) if (member(ends, ne, pos0 + sh)) out[m++] = sh;
    return m;
}

static site_t make_site(int kind, int strand, int i, int j, int motif, int w, int us_free, int ds_free, int cb)
{
    site_t s;
    s.kind = kind; s.strand = strand; s.i = i; s.j = j; s.motif = motif; s.w = w;
    s.alt = iabs(i - j);                                                    /* align.py:698-702 */
    s.clip_alt = imin(iabs(j - i - cb), iabs(j - i + cb));
    s.tot = imin(iabs(i + us_free), iabs(i - ds_free)) + imin(iabs(j + us_free), iabs(j - ds_free));
    return s;
}

static int key_less(const int* a, const int* b)
{
    for (int k = 0; k < 4; ++k) { if (a[k] < b[k]) return 1; if (a[k] > b[k]) return 0; }
    return 0;
}

/* align.py:705-733; returns the index of the winner (the list is never empty here) */
static int sort_ss(const list_t* l, int us, int ds, int cb)
{
    for (int tier = 0; tier < 4; ++tier) {
        int best = -1, bk[4] = {0, 0, 0, 0};
        for (int k = 0; k < l->n; ++k) {
            const site_t* s = &l->v[k];
            /* a site belongs to the first tier that accepts it */
            const int t0 = -cb <= s->i - s->j && s->i - s->j <= cb;
            const int t1 = -us <= s->i && s->i <= ds && -us <= s->j && s->j <= ds;
            const int t2 = -cb <= s->i && s->i <= 0 && 0 <= s->j && s->j <= cb;
            const int mine = t0 ? 0 : t1 ? 1 : t2 ? 2 : 3;
            if (mine != tier) continue;
            int key[4];
            if (tier == 0) { key[0] = s->clip_alt; key[1] = s->alt; key[2] = s->w; key[3] = s->tot; }
            else if (tier == 1) { key[0] = s->alt; key[1] = s->w; key[2] = s->clip_alt; key[3] = s->tot; }
            else { key[0] = s->w; key[1] = s->alt; key[2] = s->clip_alt; key[3] = s->tot; }
            if (best < 0 || key_less(key, bk)) { best = k; memcpy(bk, key, sizeof(bk)); }     /* stable: first seen wins ties */
        }
        if (best >= 0) return best;
    }
    return -1;
}

/* the same range as a minimap2 index serves it (mappy.Aligner.seq -> mappy_fetch_seq: env.GENOME of the reference's main pass,
 * find_bsj.py:340-341): no sequence (n = -1, Python's None) for a start outside [0, L) or an empty range; the end is clipped */
static void idxslice(int64_t L, int64_t a, int64_t b, int64_t* lo, int64_t* n)
{
    if (a < 0 || a >= L || a >= b) { *lo = 0; *n = -1; return; }
    *lo = a; *n = (b > L ? L : b) - a;
}

/* Python's s[a:b] on a string of length L: first index and length */
static void pyslice(int64_t L, int64_t a, int64_t b, int64_t* lo, int64_t* n)
{
    if (a < 0) { a += L; if (a < 0) a = 0; } else if (a > L) a = L;
    if (b < 0) { b += L; if (b < 0) b = 0; } else if (b > L) b = L;
    *lo = a; *n = b > a ? b - a : 0;
}

/* g: the contig's characters, L its length; [start, end) the candidate; host_mask bit 0 '+', bit 1 '-';
 * site_pos / site_cnt: annotated sites of THIS contig as four ascending runs of 1-based positions
 * ('+' starts, '+' ends, '-' starts, '-' ends), all counts zero = no annotation for the contig.
 * out[8] = status (0 done, 1 = invalid coordinates),
 * us_free, ds_free, found (0 none, 1 de novo, 2 annotated pair), strand (0 '+', 1 '-'), us_shift, ds_shift, motif.
 * is_canonical: bit 0 = GT-AG only; bit 1 = the search windows of find_denovo_signal are cut as a minimap2 index serves sequences
 * (g must then be the text as the index holds it: upper case, anything but ACGT an N) instead of as Python slices a string. */
int clo_splice_signal(const char* g, int64_t L, int64_t start, int64_t end, int32_t clip_base, int32_t host_mask,
                      int32_t search_extra, int32_t shift_threshold, int32_t is_canonical,
                      const int64_t* site_pos, const int64_t* site_cnt, int32_t* out)
{
    memset(out, 0, sizeof(int32_t) * 8);
    const int index_slices = (is_canonical & 2) != 0;
    is_canonical &= 1;
    if (start < 0 || start >= end || end > L) { out[0] = 1; return 0; }
    const int cb = clip_base, sl = clip_base + search_extra, T = clip_base + shift_threshold;
    /* align.py:477-493 */
    int ds_free = 0, us_free = 0;
    for (int i = 0; i < 100; ++i) {
        if (end + i > L) break;
        if (memcmp(g + start, g + end, (size_t)i) != 0) break;
        ds_free = i;
    }
    for (int j = 0; j < 100; ++j) {
        if (start - j < 0) break;
        if (memcmp(g + start - j, g + end - j, (size_t)j) != 0) break;
        us_free = j;
    }
    out[1] = us_free; out[2] = ds_free;
    /* align.py:495-496: next to a contig end find_annotated_signal returns without looking at the annotation; the
       de-novo search then runs on whatever Python's slices give there (negative start wraps, end is clipped) */
    const int edge = start - sl - us_free - 2 < 0 || end + sl + ds_free + 2 > L;

    const int64_t* run[4]; int64_t cnt[4] = {0, 0, 0, 0};
    { int64_t at = 0; for (int k = 0; k < 4; ++k) { run[k] = site_pos ? site_pos + at : NULL; cnt[k] = site_cnt ? site_cnt[k] : 0; at += cnt[k]; } }
    const int have_anno = !edge && cnt[0] + cnt[1] + cnt[2] + cnt[3] > 0;
    int* us_anno[2]; int* ds_anno[2]; int n_us[2] = {0, 0}, n_ds[2] = {0, 0};
    for (int s = 0; s < 2; ++s) { us_anno[s] = (int*)malloc(sizeof(int) * (size_t)(4 * sl + 4)); ds_anno[s] = (int*)malloc(sizeof(int) * (size_t)(4 * sl + 4)); }
    list_t found = {NULL, 0, 0};
    int rc_found = 0;

    /* ---- pairs of annotated sites (align.py:500-568) ---- */
    if (have_anno) {
        for (int s = 0; s < 2; ++s) {
            n_us[s] = annotated_shifts(run[2 * s], cnt[2 * s], run[2 * s + 1], cnt[2 * s + 1], start, sl, us_anno[s]);
            n_ds[s] = annotated_shifts(run[2 * s], cnt[2 * s], run[2 * s + 1], cnt[2 * s + 1], end, sl, ds_anno[s]);
            for (int a = 0; a < n_us[s]; ++a)
                for (int b = 0; b < n_ds[s]; ++b) {
                    const int i = us_anno[s][a], j = ds_anno[s][b];
                    if (iabs(i - j) > T) continue;
                    char us_ss[2], ds_ss[2];
                    memcpy(us_ss, g + start + i - 2, 2); memcpy(ds_ss, g + end + j, 2);
                    if (s == 1) { char a2[2], b2[2]; revcomp2(ds_ss, a2); revcomp2(us_ss, b2); memcpy(us_ss, a2, 2); memcpy(ds_ss, b2, 2); }
                    int w = 3;
                    for (int m = 0; m < 5; ++m) if (!memcmp(ds_ss, DONOR[m], 2) && !memcmp(us_ss, ACCEPTOR[m], 2)) w = WEIGHT[m];
                    push(&found, make_site(2, s, i, j, 0, w, us_free, ds_free, cb));
                }
        }
        if (found.n) rc_found = 2;
    }

    /* ---- de-novo search (align.py:571-695) ---- */
    if (!rc_found) {
        const int us_len = sl + us_free, ds_len = sl + ds_free;
        int64_t ua, nu, da, nd;
        if (index_slices) { idxslice(L, start - us_len - 2, start + ds_len, &ua, &nu); idxslice(L, end - us_len, end + ds_len + 2, &da, &nd); }
        else {
            pyslice(L, start - us_len - 2, start + ds_len, &ua, &nu);   /* genome[start - us_len - 2 : start + ds_len] */
            pyslice(L, end - us_len, end + ds_len + 2, &da, &nd);       /* genome[end - us_len : end + ds_len + 2]     */
        }
        const char* us_seq = g + ua;
        const char* ds_seq = g + da;
        const int short_seq = nu < 0 || nd < 0 || nu < ds_len - us_len + 2 || nd < ds_len - us_len + 2;      /* align.py:580-583: None or too short: no search */
        const int OFF = us_len + sl + 2;                                /* shift -> index of the flag arrays */
        const int nflag = OFF + ds_len + sl + 4;
        char* fu = (char*)malloc((size_t)nflag); char* fd = (char*)malloc((size_t)nflag);
        const int host = host_mask & 3;
        for (int round = 0; round < 2 && !found.n && !short_seq; ++round) {
            int strands;
            if (round == 0) strands = host ? host : 3;                 /* no host gene: both strands at once */
            else { if (!host) break; strands = 3 & ~host; }
            for (int s = 0; s < 2; ++s) {
                if (!((strands >> s) & 1)) continue;
                for (int m = 0; m < (is_canonical ? 1 : 5); ++m) {
                    char us_motif[2], ds_motif[2];
                    if (s == 0) { memcpy(ds_motif, DONOR[m], 2); memcpy(us_motif, ACCEPTOR[m], 2); }
                    else { revcomp2(ACCEPTOR[m], ds_motif); revcomp2(DONOR[m], us_motif); }
                    memset(fu, 0, (size_t)nflag); memset(fd, 0, (size_t)nflag);
                    /* str.find from index 1 (align.py:604-611): occurrences at p >= 1, site = p - us_len */
                    for (int p = 1; p + 2 <= nu; ++p) if (!memcmp(us_seq + p, us_motif, 2)) fu[p - us_len + OFF] = 1;
                    for (int p = 1; p + 2 <= nd; ++p) if (!memcmp(ds_seq + p, ds_motif, 2)) fd[p - us_len + OFF] = 1;
                    /* annotated shifts of this strand join the occurrences (align.py:612-621); sorted(set(...)) */
                    for (int a = 0; a < n_us[s]; ++a) fu[us_anno[s][a] + OFF] = 1;
                    for (int b = 0; b < n_ds[s]; ++b) fd[ds_anno[s][b] + OFF] = 1;
                    for (int x = 0; x < nflag; ++x) {
                        if (!fu[x]) continue;
                        for (int y = 0; y < nflag; ++y) {
                            if (!fd[y]) continue;
                            const int i = x - OFF, j = y - OFF;
                            if (iabs(i - j) > T) continue;
                            push(&found, make_site(1, s, i, j, m, WEIGHT[m], us_free, ds_free, cb));
                        }
                    }
                }
            }
        }
        free(fu); free(fd);
        if (found.n) rc_found = 1;
    }

    if (rc_found) {
        const site_t* s = &found.v[sort_ss(&found, us_free, ds_free, cb)];
        out[3] = rc_found; out[4] = s->strand; out[5] = s->i; out[6] = s->j; out[7] = s->motif;
    }
    for (int s = 0; s < 2; ++s) { free(us_anno[s]); free(ds_anno[s]); }
    free(found.v);
    return 0;
}
