/*
 * ssw_oracle.c -- CPU restatement of CIRI-long's vendored striped Smith-Waterman.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ciri_long_amd/ (the product) may link,
 * import or execute this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * What it restates (citations are relative to /root/reference):
 *   libs/striped_smith_waterman/ssw.c:89-114   byte query profile (bias added, pads score 0)
 *   libs/striped_smith_waterman/ssw.c:123-345  8-bit striped pass (16 stripes)
 *   libs/striped_smith_waterman/ssw.c:347-369  word query profile
 *   libs/striped_smith_waterman/ssw.c:371-546  16-bit striped pass (8 stripes)
 *   libs/striped_smith_waterman/ssw.c:548-735  banded traceback (CIGAR)
 *   libs/striped_smith_waterman/ssw.c:750-869  ssw_init / ssw_align orchestration
 *
 * The reference is SSE2; this file is portable scalar C.  A "vector" is a plain
 * int array with one slot per stripe; every saturating SSE2 operation is written
 * out as clamped integer arithmetic.  The loop structure of the two passes is kept
 * because it is observable: the 16-bit pass stops its lazy-F fix-up on a test that
 * is never true when gap_open <= gap_extend (ssw.c:468-478), so vertical gaps that
 * cross a stripe boundary are truncated there, and the second-best score depends on
 * the stripe padding rows (ssw.c:108,363).  A textbook row-major DP does not
 * reproduce either effect.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against golden vectors
 * produced by the reference's own libssw.so + ssw_wrap.py (tests/golden/make_golden.py)
 * and tests/test_oracle_vs_ref.py checks it against oracle/_ref/libssw.so on
 * randomized inputs whenever that build is present.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint16_t score1;
    uint16_t score2;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    int32_t ref_end2;
    uint32_t *cigar;
    int32_t cigarLen;
} clo_align;

typedef struct {
    int score;
    int ref;
    int read;
} pass_end;

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int sub0(int a, int b) { return a > b ? a - b : 0; } /* unsigned saturating subtract */

/* ------------------------------------------------------------------------------------------
 * One striped pass.  W = 16 stripes, unsigned 8-bit saturation (ssw.c:123-345) or
 * W = 8 stripes, signed 16-bit saturation (ssw.c:371-546).
 * Row r of the read lives at position r % segLen of stripe r / segLen.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int W;        /* stripes: 16 (byte) or 8 (word) */
    int segLen;   /* positions per stripe */
    int *prof;    /* [n][segLen][W] substitution scores (byte: + bias) */
} profile_t;

static profile_t make_profile(const int8_t *read, int readLen, const int8_t *mat, int n, int W, int bias)
{
    profile_t p;
    p.W = W;
    p.segLen = (readLen + W - 1) / W;
    p.prof = (int *)malloc(sizeof(int) * (size_t)n * (size_t)imax(p.segLen, 1) * (size_t)W);
    for (int nt = 0; nt < n; ++nt)
        for (int i = 0; i < p.segLen; ++i)
            for (int s = 0; s < W; ++s) {
                int row = i + s * p.segLen;
                /* ssw.c:108 / ssw.c:363: rows past the read score 0 against every base */
                int v = row >= readLen ? 0 : mat[nt * n + read[row]];
                p.prof[((size_t)nt * p.segLen + i) * W + s] = v + bias;
            }
    return p;
}

static void shift_up(int *v, int W) /* _mm_slli_si128 by one element: stripe s takes stripe s-1 */
{
    for (int s = W - 1; s > 0; --s) v[s] = v[s - 1];
    v[0] = 0;
}

static int hmax(const int *v, int W)
{
    int m = v[0];
    for (int s = 1; s < W; ++s) m = imax(m, v[s]);
    return m;
}

/* second best: ssw.c:325-340 (byte, resumes at edge+1) and ssw.c:528-541 (word, resumes at edge) */
static void second_best(const int *maxColumn, int refLen, int end_ref, int maskLen, int resume_plus, pass_end *b1)
{
    b1->score = 0;
    b1->ref = 0;
    b1->read = 0;
    int edge = (end_ref - maskLen) > 0 ? (end_ref - maskLen) : 0;
    for (int i = 0; i < edge; ++i)
        if (maxColumn[i] > b1->score) {
            b1->score = maxColumn[i];
            b1->ref = i;
        }
    edge = (end_ref + maskLen) > refLen ? refLen : (end_ref + maskLen);
    for (int i = edge + resume_plus; i < refLen; ++i)
        if (maxColumn[i] > b1->score) {
            b1->score = maxColumn[i];
            b1->ref = i;
        }
}

static void pass_byte(const int8_t *ref, int ref_dir, int refLen, int readLen, int gapO, int gapE,
                      const profile_t *P, int terminate, int bias, int maskLen, pass_end bests[2])
{
    const int W = 16, segLen = P->segLen;
    int max = 0, end_read = readLen - 1, end_ref = -1; /* ssw.c:143-145 */
    int *maxColumn = (int *)calloc((size_t)imax(refLen, 1), sizeof(int));
    size_t vec = (size_t)imax(segLen, 1) * W;
    int *Hstore = (int *)calloc(vec, sizeof(int)), *Hload = (int *)calloc(vec, sizeof(int));
    int *E = (int *)calloc(vec, sizeof(int)), *Hmax = (int *)calloc(vec, sizeof(int));
    int vMaxScore[16] = {0}, vMaxMark[16] = {0};
    int begin = 0, end = refLen, step = 1;
    if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }

    for (int i = begin; i != end; i += step) {
        int vF[16] = {0}, vMaxColumn[16] = {0}, vH[16];
        memcpy(vH, Hstore + (size_t)(segLen - 1) * W, sizeof(vH));
        shift_up(vH, W);
        const int *vP = P->prof + (size_t)ref[i] * segLen * W;
        int *t = Hload; Hload = Hstore; Hstore = t;

        for (int j = 0; j < segLen; ++j) {               /* ssw.c:204-238 */
            for (int s = 0; s < W; ++s) {
                int h = imin(vH[s] + vP[j * W + s], 255);  /* _mm_adds_epu8 */
                h = sub0(h, bias);
                int e = E[j * W + s];
                h = imax(h, imax(e, vF[s]));
                vMaxColumn[s] = imax(vMaxColumn[s], h);
                Hstore[j * W + s] = h;
                h = sub0(h, gapO);
                e = imax(sub0(e, gapE), h);
                E[j * W + s] = e;
                vF[s] = imax(sub0(vF[s], gapE), h);
                vH[s] = Hload[j * W + s];
            }
        }

        /* lazy-F, ssw.c:240-272: E is deliberately not refreshed */
        {
            int j = 0;
            shift_up(vF, W);
            for (;;) {
                int any = 0;
                for (int s = 0; s < W; ++s)
                    if (sub0(vF[s], sub0(Hstore[j * W + s], gapO)) != 0) any = 1;
                if (!any) break;
                for (int s = 0; s < W; ++s) {
                    int h = imax(Hstore[j * W + s], vF[s]);
                    vMaxColumn[s] = imax(vMaxColumn[s], h);
                    Hstore[j * W + s] = h;
                    vF[s] = sub0(vF[s], gapE);
                }
                if (++j >= segLen) { j = 0; shift_up(vF, W); }
            }
        }

        int differs = 0;                                 /* ssw.c:274-291 */
        for (int s = 0; s < W; ++s) {
            vMaxScore[s] = imax(vMaxScore[s], vMaxColumn[s]);
            if (vMaxScore[s] != vMaxMark[s]) differs = 1;
        }
        if (differs) {
            memcpy(vMaxMark, vMaxScore, sizeof(vMaxMark));
            int temp = hmax(vMaxScore, W);
            if (temp > max) {
                max = temp;
                if (max + bias >= 255) break;            /* overflow: caller switches to 16 bit */
                end_ref = i;
                memcpy(Hmax, Hstore, vec * sizeof(int));
            }
        }
        maxColumn[i] = hmax(vMaxColumn, W);              /* ssw.c:294-296 */
        if (maxColumn[i] == terminate) break;
    }

    for (int i = 0; i < segLen * W; ++i)                  /* ssw.c:299-308 */
        if (Hmax[i] == max) {
            int row = i / W + i % W * segLen;
            if (row < end_read) end_read = row;
        }

    bests[0].score = max + bias >= 255 ? 255 : max;
    bests[0].ref = end_ref;
    bests[0].read = end_read;
    second_best(maxColumn, refLen, end_ref, maskLen, 1, &bests[1]);
    free(maxColumn); free(Hstore); free(Hload); free(E); free(Hmax);
}

static void pass_word(const int8_t *ref, int ref_dir, int refLen, int readLen, int gapO, int gapE,
                      const profile_t *P, int terminate, int maskLen, pass_end bests[2])
{
    const int W = 8, segLen = P->segLen;
    int max = 0, end_read = readLen - 1, end_ref = 0;    /* ssw.c:386-388 */
    int *maxColumn = (int *)calloc((size_t)imax(refLen, 1), sizeof(int));
    size_t vec = (size_t)imax(segLen, 1) * W;
    int *Hstore = (int *)calloc(vec, sizeof(int)), *Hload = (int *)calloc(vec, sizeof(int));
    int *E = (int *)calloc(vec, sizeof(int)), *Hmax = (int *)calloc(vec, sizeof(int));
    int vMaxScore[8] = {0}, vMaxMark[8] = {0};
    int begin = 0, end = refLen, step = 1;
    if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }

    for (int i = begin; i != end; i += step) {
        int vF[8] = {0}, vMaxColumn[8] = {0}, vH[8];
        memcpy(vH, Hstore + (size_t)(segLen - 1) * W, sizeof(vH));
        shift_up(vH, W);
        const int *vP = P->prof + (size_t)ref[i] * segLen * W;
        int *t = Hload; Hload = Hstore; Hstore = t;

        for (int j = 0; j < segLen; ++j) {               /* ssw.c:441-465 */
            for (int s = 0; s < W; ++s) {
                int h = vH[s] + vP[j * W + s];             /* _mm_adds_epi16 */
                h = imin(imax(h, -32768), 32767);
                int e = E[j * W + s];
                h = imax(h, imax(e, vF[s]));
                vMaxColumn[s] = imax(vMaxColumn[s], h);
                Hstore[j * W + s] = h;
                h = sub0(h, gapO);
                e = imax(sub0(e, gapE), h);
                E[j * W + s] = e;
                vF[s] = imax(sub0(vF[s], gapE), h);
                vH[s] = Hload[j * W + s];
            }
        }

        /* lazy-F, ssw.c:468-478: bounded double loop; the column maximum is NOT refreshed here
         * and the exit test compares F-gapE with max(H,F)-gapO. */
        {
            int done = 0;
            for (int k = 0; k < W && !done; ++k) {
                shift_up(vF, W);
                for (int j = 0; j < segLen; ++j) {
                    int any = 0;
                    for (int s = 0; s < W; ++s) {
                        int h = imax(Hstore[j * W + s], vF[s]);
                        Hstore[j * W + s] = h;
                        h = sub0(h, gapO);
                        vF[s] = sub0(vF[s], gapE);
                        if (vF[s] > h) any = 1;
                    }
                    if (!any) { done = 1; break; }
                }
            }
        }

        int differs = 0;                                 /* ssw.c:481-495 */
        for (int s = 0; s < W; ++s) {
            vMaxScore[s] = imax(vMaxScore[s], vMaxColumn[s]);
            if (vMaxScore[s] != vMaxMark[s]) differs = 1;
        }
        if (differs) {
            memcpy(vMaxMark, vMaxScore, sizeof(vMaxMark));
            int temp = hmax(vMaxScore, W);
            if (temp > max) {
                max = temp;
                end_ref = i;
                memcpy(Hmax, Hstore, vec * sizeof(int));
            }
        }
        maxColumn[i] = hmax(vMaxColumn, W);              /* ssw.c:498-499 */
        if (maxColumn[i] == terminate) break;
    }

    for (int i = 0; i < segLen * W; ++i)                  /* ssw.c:502-511 */
        if (Hmax[i] == max) {
            int row = i / W + i % W * segLen;
            if (row < end_read) end_read = row;
        }

    bests[0].score = max;
    bests[0].ref = end_ref;
    bests[0].read = end_read;
    second_best(maxColumn, refLen, end_ref, maskLen, 0, &bests[1]);
    free(maxColumn); free(Hstore); free(Hload); free(E); free(Hmax);
}

/* ------------------------------------------------------------------------------------------
 * Banded traceback, ssw.c:548-735.  The reference keeps one band row of H/E and a flat
 * direction array that it re-uses across band doublings; out-of-band neighbours are read
 * through two zeroed sentinel slots (index 0 and `edge`).  The index arithmetic is kept
 * exactly, because the sentinel at `edge` can overwrite a live entry when the band is
 * clipped by the reference end, and the result depends on it.
 * ------------------------------------------------------------------------------------------ */
static inline int band_u(int w, int i, int j) { int x = i - w; if (x < 0) x = 0; return j - x + 1; }          /* ssw.c:55 */
static inline long band_d(int w, int i, int j, int p) { int x = i - w; if (x < 0) x = 0; return (long)(j - x) * 3 + p; } /* ssw.c:58 */

static int round_pow2(long x) /* ++x then kroundup32, ssw.c:65 */
{
    uint32_t v = (uint32_t)(x + 1);
    --v; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; ++v;
    return (int)v;
}

static uint32_t cigar_pack(uint32_t len, char op)
{
    uint32_t code = op == 'I' ? 1u : op == 'D' ? 2u : 0u; /* ssw.h:131-170, only M/I/D occur */
    return (len << 4) | code;
}

typedef struct { uint32_t *v; int n, cap; } u32vec;
static void u32_push(u32vec *a, uint32_t x)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 16; a->v = (uint32_t *)realloc(a->v, sizeof(uint32_t) * (size_t)a->cap); }
    a->v[a->n++] = x;
}

/* test instrumentation: steps of the last traceback that read a direction byte outside the final band (the reads of stale bytes,
 * ssw.c:58,640) -- lets the tests pick inputs that exercise that path */
static int g_oob_steps = 0;
int clo_last_oob_steps(void) { return g_oob_steps; }

static int banded_traceback(const int8_t *ref, const int8_t *read, int refLen, int readLen, int score,
                            int gapO, int gapE, int band_width, const int8_t *mat, int n,
                            uint32_t **out, int *outLen)
{
    int s1 = 8;
    int64_t s2 = 1024;
    int *h_b = (int *)calloc((size_t)s1, sizeof(int));
    int *e_b = (int *)calloc((size_t)s1, sizeof(int));
    int *h_c = (int *)calloc((size_t)s1, sizeof(int));
    int8_t *direction = (int8_t *)calloc((size_t)s2, 1);
    int8_t *direction_line = direction;
    int width, width_d, max = 0;

    do {
        width = band_width * 2 + 3; width_d = band_width * 2 + 1;
        while (width >= s1) {                                           /* ssw.c:573-579 */
            int ns = round_pow2(s1);
            h_b = (int *)realloc(h_b, (size_t)ns * sizeof(int));
            e_b = (int *)realloc(e_b, (size_t)ns * sizeof(int));
            h_c = (int *)realloc(h_c, (size_t)ns * sizeof(int));
            memset(h_b + s1, 0, (size_t)(ns - s1) * sizeof(int));
            memset(e_b + s1, 0, (size_t)(ns - s1) * sizeof(int));
            memset(h_c + s1, 0, (size_t)(ns - s1) * sizeof(int));
            s1 = ns;
        }
        while ((int64_t)width_d * readLen * 3 >= s2) {                  /* ssw.c:580-588 */
            int64_t ns = (int64_t)(uint32_t)round_pow2((long)s2);
            if (ns <= s2) { ns = s2 * 2; }
            direction = (int8_t *)realloc(direction, (size_t)ns);
            memset(direction + s2, 0, (size_t)(ns - s2));
            s2 = ns;
        }
        direction_line = direction;
        for (int j = 1; j < width - 1; ++j) h_b[j] = 0;
        for (int i = 0; i < readLen; ++i) {
            int beg = imax(0, i - band_width), end = imin(refLen - 1, i + band_width), u = 0;
            int edge = end + 1 < width - 1 ? end + 1 : width - 1;
            int f = 0;
            h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;       /* ssw.c:596 */
            direction_line = direction + (size_t)width_d * i * 3;
            for (int j = beg; j <= end; ++j) {
                u = band_u(band_width, i, j);
                int e = band_u(band_width, i - 1, j);
                int b = band_u(band_width, i, j - 1);
                int d = band_u(band_width, i - 1, j - 1);
                long de = band_d(band_width, i, j, 0), df = band_d(band_width, i, j, 1), dh = band_d(band_width, i, j, 2);

                int t1 = i == 0 ? -gapO : h_b[e] - gapO;                /* ssw.c:607-611 */
                int t2 = i == 0 ? -gapE : e_b[e] - gapE;
                e_b[u] = t1 > t2 ? t1 : t2;
                direction_line[de] = t1 > t2 ? 3 : 2;

                t1 = h_c[b] - gapO;                                     /* ssw.c:613-616 */
                t2 = f - gapE;
                f = t1 > t2 ? t1 : t2;
                direction_line[df] = t1 > t2 ? 5 : 4;

                int e1 = e_b[u] > 0 ? e_b[u] : 0;                       /* ssw.c:618-627 */
                int f1 = f > 0 ? f : 0;
                t1 = e1 > f1 ? e1 : f1;
                t2 = h_b[d] + mat[ref[j] * n + read[i]];
                h_c[u] = t1 > t2 ? t1 : t2;
                if (h_c[u] > max) max = h_c[u];
                if (t1 <= t2) direction_line[dh] = 1;
                else direction_line[dh] = e1 > f1 ? direction_line[de] : direction_line[df];
            }
            for (int j = 1; j <= u; ++j) h_b[j] = h_c[j];               /* ssw.c:629 */
        }
        band_width *= 2;
    } while (max < score && band_width < 2 * readLen);
    band_width /= 2;

    /* trace back, ssw.c:636-696 */
    u32vec c = {0, 0, 0};
    int i = readLen - 1, j = refLen - 1, run = 0, state = 2, fail = 0;
    char op = 'M', prev_op = 'M';
    long lo = 0, hi = (long)s2;
    g_oob_steps = 0;
    while (i > 0) {
        long idx = (direction_line - direction) + band_d(band_width, i, j, state);
        if (j < imax(0, i - band_width) || j > imin(refLen - 1, i + band_width)) ++g_oob_steps;
        if (idx < lo || idx >= hi) { fail = 1; break; }
        switch (direction[idx]) {
            case 1: --i; --j; state = 2; direction_line -= (size_t)width_d * 3; op = 'M'; break;
            case 2: --i; state = 0; direction_line -= (size_t)width_d * 3; op = 'I'; break;
            case 3: --i; state = 2; direction_line -= (size_t)width_d * 3; op = 'I'; break;
            case 4: --j; state = 1; op = 'D'; break;
            case 5: --j; state = 2; op = 'D'; break;
            default: fail = 1; break;
        }
        if (fail) break;
        if (op == prev_op) ++run;
        else { u32_push(&c, cigar_pack((uint32_t)run, prev_op)); prev_op = op; run = 1; }
    }
    if (fail) {
        fprintf(stderr, "Trace back error.\n");
        free(c.v); free(direction); free(h_c); free(e_b); free(h_b);
        return -1;
    }
    if (op == 'M') u32_push(&c, cigar_pack((uint32_t)run + 1, op));       /* ssw.c:697-714 */
    else { u32_push(&c, cigar_pack((uint32_t)run, op)); u32_push(&c, cigar_pack(1, 'M')); }

    uint32_t *res = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)c.n);
    for (int k = 0; k < c.n; ++k) res[k] = c.v[c.n - 1 - k];             /* ssw.c:716-725 */
    *out = res; *outLen = c.n;
    free(c.v); free(direction); free(h_c); free(e_b); free(h_b);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * ssw_init + ssw_align in one call (ssw.c:750-869).  score_size as in ssw.h:62-64.
 * Returns 0 on success, -1 when the reference would have returned NULL.
 * ------------------------------------------------------------------------------------------ */
int clo_ssw_align(const int8_t *read, int32_t readLen, const int8_t *mat, int32_t n, int8_t score_size,
                  const int8_t *ref, int32_t refLen, uint8_t gapO, uint8_t gapE, uint8_t flag,
                  uint16_t filters, int32_t filterd, int32_t maskLen, clo_align *r)
{
    memset(r, 0, sizeof(*r));
    r->ref_begin1 = -1;
    r->read_begin1 = -1;
    int have_byte = (score_size == 0 || score_size == 2), have_word = (score_size == 1 || score_size == 2);
    int bias = 0;
    if (have_byte) {                                                     /* ssw.c:756-764 */
        for (int i = 0; i < n * n; ++i) if (mat[i] < bias) bias = mat[i];
        bias = abs(bias);
    }
    pass_end bests[2], rev[2];
    int word = 0;
    if (have_byte) {                                                     /* ssw.c:804-822 */
        profile_t P = make_profile(read, readLen, mat, n, 16, bias);
        pass_byte(ref, 0, refLen, readLen, gapO, gapE, &P, 255 /* (uint8_t)-1 */, bias, maskLen, bests);
        free(P.prof);
        if (have_word && bests[0].score == 255) {
            profile_t Q = make_profile(read, readLen, mat, n, 8, 0);
            pass_word(ref, 0, refLen, readLen, gapO, gapE, &Q, 65535 /* (uint16_t)-1 */, maskLen, bests);
            free(Q.prof);
            word = 1;
        } else if (bests[0].score == 255) {
            return -1;
        }
    } else if (have_word) {
        profile_t Q = make_profile(read, readLen, mat, n, 8, 0);
        pass_word(ref, 0, refLen, readLen, gapO, gapE, &Q, 65535, maskLen, bests);
        free(Q.prof);
        word = 1;
    } else {
        return -1;
    }
    r->score1 = (uint16_t)bests[0].score;
    r->ref_end1 = bests[0].ref;
    r->read_end1 = bests[0].read;
    if (maskLen >= 15) { r->score2 = (uint16_t)bests[1].score; r->ref_end2 = bests[1].ref; }
    else { r->score2 = 0; r->ref_end2 = -1; }
    if (flag == 0 || (flag == 2 && r->score1 < filters)) return 0;       /* ssw.c:834 */

    /* begin position: same pass on the reversed read prefix, reference walked downwards, ssw.c:837-849 */
    int rl = r->read_end1 + 1;
    int8_t *rr = (int8_t *)calloc((size_t)imax(rl, 1), 1);
    for (int k = 0; k < rl; ++k) rr[k] = read[r->read_end1 - k];
    if (!word) {
        profile_t P = make_profile(rr, rl, mat, n, 16, bias);
        pass_byte(ref, 1, r->ref_end1 + 1, rl, gapO, gapE, &P, r->score1 & 0xff, bias, maskLen, rev);
        free(P.prof);
    } else {
        profile_t Q = make_profile(rr, rl, mat, n, 8, 0);
        pass_word(ref, 1, r->ref_end1 + 1, rl, gapO, gapE, &Q, r->score1, maskLen, rev);
        free(Q.prof);
    }
    free(rr);
    r->ref_begin1 = rev[0].ref;
    r->read_begin1 = r->read_end1 - rev[0].read;
    if ((7 & flag) == 0 || ((2 & flag) != 0 && r->score1 < filters) ||
        ((4 & flag) != 0 && (r->ref_end1 - r->ref_begin1 > filterd || r->read_end1 - r->read_begin1 > filterd)))
        return 0;                                                        /* ssw.c:850 */

    int cRef = r->ref_end1 - r->ref_begin1 + 1, cRead = r->read_end1 - r->read_begin1 + 1; /* ssw.c:853-856 */
    int band = abs(cRef - cRead) + 1;
    if (r->ref_begin1 < 0) {
        /* score1 == 0 in the byte regime: the reference reads ref[-1] for a 1x1 problem whose
         * traceback loop never runs; the CIGAR is 1M whatever that byte holds. */
        r->cigar = (uint32_t *)malloc(sizeof(uint32_t));
        r->cigar[0] = cigar_pack(1, 'M');
        r->cigarLen = 1;
        return 0;
    }
    if (banded_traceback(ref + r->ref_begin1, read + r->read_begin1, cRef, cRead, r->score1, gapO, gapE, band, mat, n,
                         &r->cigar, &r->cigarLen) != 0)
        return -1;
    return 0;
}

void clo_free_cigar(clo_align *r) { free(r->cigar); r->cigar = 0; r->cigarLen = 0; }

/* Batched convenience wrapper used by tests and by bench.py's cpu_baseline leg ("port").
 * Same packed layout as the product's batched entry point (include/ciri_long_hip.h). */
int clo_ssw_batch(int32_t nAln, const int8_t *reads, const int64_t *read_off, const int8_t *refs, const int64_t *ref_off,
                  const int8_t *mat, int32_t n, uint8_t gapO, uint8_t gapE, uint8_t flag, int8_t score_size,
                  int32_t *out9 /* [nAln][9]: score1 score2 rb re qb qe re2 cigar_off cigar_len */,
                  uint32_t *cigar_buf, int64_t cigar_cap)
{
    int64_t used = 0;
    for (int32_t a = 0; a < nAln; ++a) {
        int32_t ql = (int32_t)(read_off[a + 1] - read_off[a]), rl = (int32_t)(ref_off[a + 1] - ref_off[a]);
        int32_t mask = ql > 30 ? ql / 2 : 15; /* libs/striped_smith_waterman/ssw_wrap.py:196-199 */
        clo_align r;
        int rc = clo_ssw_align(reads + read_off[a], ql, mat, n, score_size, refs + ref_off[a], rl, gapO, gapE, flag, 0, 0, mask, &r);
        int32_t *o = out9 + (size_t)a * 9;
        if (rc != 0) { for (int k = 0; k < 9; ++k) o[k] = -9; continue; }
        o[0] = r.score1; o[1] = r.score2; o[2] = r.ref_begin1; o[3] = r.ref_end1; o[4] = r.read_begin1; o[5] = r.read_end1;
        o[6] = r.ref_end2; o[7] = (int32_t)used; o[8] = r.cigarLen;
        if (cigar_buf && used + r.cigarLen <= cigar_cap) memcpy(cigar_buf + used, r.cigar, sizeof(uint32_t) * (size_t)r.cigarLen);
        used += r.cigarLen;
        clo_free_cigar(&r);
    }
    return used <= cigar_cap || !cigar_buf ? 0 : 1;
}

/* One striped pass on its own (debug/verification entry: lets tests compare the row-major spec of
 * rowmajor_spec.c and the GPU passes with the literal stripe emulation pass by pass).
 * out: [0] score [1] end_ref [2] end_read [3] score2 [4] ref_end2 */
void clo_striped_pass(const int8_t *ref, int ref_dir, int refLen, const int8_t *read, int readLen, const int8_t *mat, int n,
                      int gapO, int gapE, int word, int bias, int terminate, int maskLen, int32_t *out)
{
    pass_end b[2];
    profile_t P = make_profile(read, readLen, mat, n, word ? 8 : 16, word ? 0 : bias);
    if (word) pass_word(ref, ref_dir, refLen, readLen, gapO, gapE, &P, terminate, maskLen, b);
    else pass_byte(ref, ref_dir, refLen, readLen, gapO, gapE, &P, terminate, bias, maskLen, b);
    free(P.prof);
    out[0] = b[0].score; out[1] = b[0].ref; out[2] = b[0].read; out[3] = b[1].score; out[4] = b[1].ref;
}
