"""tools/traceback_rows_model.py (numpy model of K1b's row form) against the oracle's CIGARs, the stale-walk goldens included."""
import gzip
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from oracle_lib import encode, make_mat, oracle_align  # noqa: E402
from traceback_rows_model import tb_rows  # noqa: E402


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
            out.append(_rnd(rng, int(rng.integers(1, 5))))
    return ''.join(out)


def _check(ref, q, scheme):
    m, x, o, e = scheme
    want = oracle_align(ref, q, m, x, o, e)
    if want is None or want['score'] == 0:
        return 0
    r = encode(ref)[want['ref_begin']:want['ref_end'] + 1]; rd = encode(q)[want['query_begin']:want['query_end'] + 1]
    got = tb_rows(r, rd, want['score'], make_mat(m, x), 5, o, e)
    assert got == want['cigar'], (len(q), len(ref), scheme)
    return 1


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1), (3, 5, 7, 7)])
def test_row_traceback_model_equals_oracle(scheme):
    rng = np.random.default_rng(17 + sum(scheme))
    done = 0
    for it in range(40):
        L = int(rng.choice([12, 40, 90, 160, 260])); R = int(rng.choice([60, 200, 420]))
        ref = _rnd(rng, R)
        st = int(rng.integers(0, max(1, R - L)))
        q = _mut(ref[st:st + L], rng, float(rng.choice([0.03, 0.12, 0.3])))
        u = rng.random()
        if u < 0.25 and len(q) > 40:                 # a long gap: the band starts wide
            q = q[:len(q) // 3] + q[len(q) // 3 + int(rng.integers(8, 30)):]
        elif u < 0.4:
            q = q[:len(q) // 2] + _rnd(rng, int(rng.integers(5, 25))) + q[len(q) // 2:]
        elif u < 0.5:
            ref = ref[:R // 2] + 'N' * 4 + ref[R // 2:]
        done += _check(ref, q or 'A', scheme)
    assert done > 25


def test_row_traceback_model_on_the_stale_walk_goldens():
    """the walks that leave the final band: the codes come from an earlier band iteration"""
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stale_walk_golden.json.gz'), 'rt') as f:
        cases = json.load(f)['cases']
    c = cases[0]
    w = c['want']
    r = encode(c['ref'])[w['ref_begin']:w['ref_end'] + 1]; rd = encode(c['query'])[w['query_begin']:w['query_end'] + 1]
    assert tb_rows(r, rd, w['score'], make_mat(1, 1), 5, 1, 1) == w['cigar']
