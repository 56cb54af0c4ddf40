"""tools/scan_model.py (numpy model of K1s: rows sequential, columns parallel, E as a prefix maximum, chunks, window slices)
against the oracle, on the alignments of K1s's class (readLen <= 254, max_match * readLen + bias < 255)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from oracle_lib import encode, make_mat, mask_len, oracle_align  # noqa: E402
from scan_model import scan_align  # noqa: E402

KEYS = ('score', 'score2', 'ref_begin', 'ref_end', 'query_begin', 'query_end', 'ref_end2')


def _rnd(rng, n):
    return ''.join('ACGT'[i] for i in rng.integers(0, 4, n))


def _mut(s, rng, p):
    out = []
    for c in s:
        u = rng.random()
        if u < p / 3:
            continue
        if u < 2 * p / 3:
            out.append('ACGT'[rng.integers(4)]); continue
        out.append(c)
        if u < p:
            out.append(_rnd(rng, int(rng.integers(1, 4))))
    return ''.join(out)


def _cases(rng, m, count, rmax):
    for _ in range(count):
        L = int(rng.integers(1, 250 // m))
        R = int(rng.choice([1, 40, 255, 256, 257, 600, rmax]))
        ref = _rnd(rng, R)
        st = int(rng.integers(0, max(1, R - L)))
        q = _mut(ref[st:st + L], rng, float(rng.choice([0.0, 0.1, 0.25])))[:L] or 'A'
        u = rng.random()
        if u < 0.2 and R > 3 * len(q) + 10:
            ref = (ref[:R // 2] + q + ref[R // 2:])[:R]          # the clip twice: ties and larger maxima in the reverse pass
        elif u < 0.3:
            ref = ref[:R // 3] + 'N' * 4 + ref[R // 3:]
        elif u < 0.4:
            q = _rnd(rng, len(q))
        yield ref, q


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (1, 1, 3, 1), (2, 3, 5, 2), (3, 1, 4, 4)])
def test_row_scan_with_chunks_equals_oracle(scheme):
    m, x, o, e = scheme
    rng = np.random.default_rng(31 + sum(scheme))
    mat = make_mat(m, x)
    for it, (ref, q) in enumerate(_cases(rng, m, 40, 900)):
        want = oracle_align(ref, q, m, x, o, e)
        got = scan_align(encode(ref), encode(q), mat, 5, o, e, mask_len(len(q)), chunk=int(rng.choice([64, 256])))
        for k in KEYS:
            assert got[k] == want[k], (it, k, len(q), len(ref), got, want)


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (2, 3, 5, 2)])
def test_window_slices_equal_the_whole_window(scheme):
    """the forward pass in slices that start `L + L*match/gap_extend + 32` columns early (clh_api.hip: scan_sliced)"""
    m, x, o, e = scheme
    rng = np.random.default_rng(77 + sum(scheme))
    mat = make_mat(m, x)
    for it, (ref, q) in enumerate(_cases(rng, m, 12, 2500)):
        want = oracle_align(ref, q, m, x, o, e)
        got = scan_align(encode(ref), encode(q), mat, 5, o, e, mask_len(len(q)), slice_own=int(rng.choice([300, 700])))
        for k in KEYS:
            assert got[k] == want[k], (it, k, len(q), len(ref), got, want)
