/*
 * rowmajor_spec.c -- the striped passes of ssw.c re-derived as a plain row-major recurrence.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ssw_oracle.c).
 *
 * The HIP kernels walk the DP matrix along anti-diagonals, so they cannot imitate the stripe
 * mechanics of the reference (libs/striped_smith_waterman/ssw.c:186-297, 423-500) literally.  This file
 * states what those mechanics compute in row-major terms; tests/test_rowmajor_spec.py proves it equal to
 * ssw_oracle.c (which IS a literal restatement) on randomized inputs, and the kernels implement this form.
 *
 *   rows     read rows 0..readLen-1, then wildcard rows (score 0 against everything, ssw.c:108,363) up to
 *            16*ceil(readLen/16) in the 8-bit regime or 8*ceil(readLen/8) in the 16-bit regime.
 *   8 bit    exact affine-gap recurrence (the lazy-F loop of ssw.c:240-272 converges when gapO >= gapE);
 *            the pass is abandoned at the first column whose maximum reaches 255-bias (ssw.c:285).
 *   16 bit   gapO >  gapE: exact recurrence, diagonal add saturating at 32767 (ssw.c:442).
 *            gapO <= gapE: the exit test of ssw.c:476 fails on its first evaluation, so the fix-up touches
 *            only position 0 of every stripe.  With S = ceil(readLen/8) and B = {S,2S,..,7S}:
 *              Fin[r]   = 0                                   for r in B or r == 0
 *                       = max(Fin[r-1]-gapE, Hm[r-1]-gapO)    otherwise            (clamped at 0)
 *              Hm[r]    = max(sat(Hf_prev[r-1] + s), E[r], Fin[r])                 ("main loop" value)
 *              E'[r]    = max(E[r]-gapE, Hm[r]-gapO)
 *              Hf[r]    = max(Hm[r], max(Fin[r-1]-gapE, Hm[r-1]-gapO)) for r in B, else Hm[r]
 *            column maximum over Hm (ssw.c:448; the fix-up does not refresh it), next column's diagonal
 *            and the end-row search over Hf (ssw.c:451,473,493).
 *   best     strict '>' on the column maximum: first column wins (ssw.c:283,490); end row = smallest row
 *            whose Hf equals the maximum in that column (ssw.c:299-308,502-511).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int sub0(int a, int b) { return a > b ? a - b : 0; }

/* out: [0] max  [1] end_ref  [2] end_read  [3] overflow(8-bit only)  [4] columns processed
 * colmax: optional, refLen entries indexed by reference position (0 where not processed)        */
int clo_rowmajor_pass(const int8_t *ref, int ref_dir, int refLen, const int8_t *read, int readLen, const int8_t *mat,
                      int n, int gapO, int gapE, int word, int bias, int terminate, int32_t *out, int32_t *colmax)
{
    int W = word ? 8 : 16;
    int S = (readLen + W - 1) / W;
    int rows = S * W;
    int quirk = word && gapO <= gapE;
    int *Hf = (int *)calloc((size_t)rows + 1, sizeof(int));   /* previous column, final values */
    int *E = (int *)calloc((size_t)rows + 1, sizeof(int));
    int *cur = (int *)calloc((size_t)rows + 1, sizeof(int));
    int *best_col = (int *)calloc((size_t)rows + 1, sizeof(int));
    int max = 0, end_ref = word ? 0 : -1, overflow = 0, ncol = 0;
    if (colmax) memset(colmax, 0, sizeof(int32_t) * (size_t)refLen);

    int begin = 0, end = refLen, step = 1;
    if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }
    for (int i = begin; i != end; i += step) {
        int rb = ref[i];
        int Fin = 0, Hm_prev = 0, cm = 0, diag = 0;
        for (int r = 0; r < rows; ++r) {
            int s = r < readLen ? mat[rb * n + read[r]] : 0;
            int boundary = quirk && r > 0 && (r % S) == 0;
            int carry = r == 0 ? 0 : imax(sub0(Fin, gapE), sub0(Hm_prev, gapO)); /* F arriving from the row above */
            int f_main = boundary ? 0 : carry;
            int t = diag + s;
            if (word) t = imin(imax(t, -32768), 32767);
            else { t = imin(diag + s + bias, 255); t = sub0(t, bias); }
            int hm = imax(imax(t, E[r]), f_main);
            if (hm < 0) hm = 0;
            int hf = boundary ? imax(hm, carry) : hm;
            cm = imax(cm, word ? hm : hf);
            diag = Hf[r];
            cur[r] = hf;
            E[r] = imax(sub0(E[r], gapE), sub0(hm, gapO));
            Fin = f_main;
            Hm_prev = hm;
        }
        memcpy(Hf, cur, sizeof(int) * (size_t)rows);
        ++ncol;
        if (cm > max) {
            max = cm;
            if (!word && max + bias >= 255) { overflow = 1; break; }
            end_ref = i;
            memcpy(best_col, cur, sizeof(int) * (size_t)rows);
        }
        if (colmax) colmax[i] = cm;
        if (cm == terminate) break;
    }
    int end_read = readLen - 1;
    for (int r = 0; r < rows; ++r)
        if (best_col[r] == max) { if (r < end_read) end_read = r; break; }
    out[0] = overflow ? 255 : max;
    out[1] = end_ref;
    out[2] = end_read;
    out[3] = overflow;
    out[4] = ncol;
    free(Hf); free(E); free(cur); free(best_col);
    return 0;
}

/* masked second best, ssw.c:325-340 (8 bit) / 528-541 (16 bit) */
void clo_second_best(const int32_t *colmax, int refLen, int end_ref, int maskLen, int word, int32_t *out2)
{
    int sc = 0, pos = 0;
    int edge = (end_ref - maskLen) > 0 ? (end_ref - maskLen) : 0;
    for (int i = 0; i < edge; ++i) if (colmax[i] > sc) { sc = colmax[i]; pos = i; }
    edge = (end_ref + maskLen) > refLen ? refLen : (end_ref + maskLen);
    for (int i = edge + (word ? 0 : 1); i < refLen; ++i) if (colmax[i] > sc) { sc = colmax[i]; pos = i; }
    out2[0] = sc;
    out2[1] = pos;
}
