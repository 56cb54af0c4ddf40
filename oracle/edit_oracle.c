/*
 * edit_oracle.c -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * CPU statement of the integer that CIRI-long's `distance(x, y)` returns (reference: CIRI_long/utils.py:153-159):
 * `Levenshtein.distance(x, y)` when either string has <= 50 characters, `edlib.align(x, y)['editDistance']` otherwise.
 * Both third-party packages (python-Levenshtein, edlib; requirements of the reference, absent from its tree and from
 * this environment) compute the same, uniquely defined quantity -- the unit-cost edit distance of the two byte strings
 * (insertion, deletion, substitution each cost 1; global, end to end; edlib's default mode "NW") -- so parity is
 * well-defined without them.  This file is the textbook two-row dynamic programme over bytes; it is pinned by the
 * known answers and metric properties in tests/test_edit_distance.py.
 */
#include <stdint.h>
#include <stdlib.h>

int32_t clo_edit_distance(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb)
{
    if (la == 0) return lb;
    if (lb == 0) return la;
    int32_t *row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(lb + 1));
    for (int j = 0; j <= lb; ++j) row[j] = j;
    for (int i = 1; i <= la; ++i) {
        int32_t diag = row[0];
        row[0] = i;
        for (int j = 1; j <= lb; ++j) {
            const int32_t up = row[j];
            int32_t v = diag + (a[i - 1] != b[j - 1]);
            if (up + 1 < v) v = up + 1;
            if (row[j - 1] + 1 < v) v = row[j - 1] + 1;
            diag = up;
            row[j] = v;
        }
    }
    const int32_t d = row[lb];
    free(row);
    return d;
}

void clo_edit_distance_batch(int32_t n, const uint8_t *a, const int64_t *a_off, const uint8_t *b, const int64_t *b_off, int32_t *out)
{
    for (int32_t k = 0; k < n; ++k)
        out[k] = clo_edit_distance(a + a_off[k], (int32_t)(a_off[k + 1] - a_off[k]), b + b_off[k], (int32_t)(b_off[k + 1] - b_off[k]));
}
