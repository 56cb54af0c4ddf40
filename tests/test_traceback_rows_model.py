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


# ---- round 4: the pieces of the wide form (tb_rows_pass<.., NW>, the passes side by side, runs of diagonal moves) ------------------
from traceback_rows_model import band_pass, band_row_f_split, doubling_from_state, walk_ops_with_runs  # noqa: E402


def test_f_of_a_band_row_split_over_waves_equals_the_row_in_one_piece():
    """the carry of the F scan through the waves' totals (the LDS exchange of tb_rows_pass<.., NW>): for any split of the offsets the
    same F as the one-piece prefix maximum of band_pass"""
    rng = np.random.default_rng(5)
    for _ in range(300):
        W = int(rng.integers(1, 400)); gE = int(rng.integers(1, 5))
        c = rng.integers(-40, 60, W).astype(np.int64)
        o = np.arange(W)
        A = c + o * gE
        pm = np.concatenate(([-(1 << 30)], np.maximum.accumulate(A)[:-1]))
        want = np.maximum(pm - (o - 1) * gE, -gE - o * gE)
        for nw in (1, 2, 4, 8):
            assert (band_row_f_split(c, gE, nw) == want).all(), (W, gE, nw)


def test_doubling_resumed_from_the_narrow_launch_and_replayed_over_side_by_side_passes():
    """ssw.c:560-632 in one piece against the two launches' form (state handed over, three iterations at once, the loop replayed):
    same final band and iteration count; and at most one more pass in a row than iterations after the hand-over / 3 rounded up"""
    rng = np.random.default_rng(8)
    for _ in range(2000):
        readLen = int(rng.integers(5, 3000)); w0 = int(rng.integers(1, 400)); score = int(rng.integers(1, 400))
        need = int(rng.integers(1, 5000))                      # the band from which the path fits
        it_of = lambda w: score if w >= need else int(score * 0.5)   # noqa: E731
        w, maxv, niter = w0, 0, 0
        while True:
            niter += 1
            maxv = max(maxv, it_of(w))
            w *= 2
            if not (maxv < score and w < 2 * readLen):
                break
        w //= 2
        gw, gn, ran, chain = doubling_from_state(it_of, score, readLen, w0)
        assert (gw, gn) == (w, niter), (readLen, w0, score, need)
        assert chain <= niter and ran <= niter + 2


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2)])
def test_walk_with_runs_of_diagonal_moves_equals_the_step_by_step_walk(scheme):
    m, x, o, e = scheme
    rng = np.random.default_rng(23 + m)
    done = 0
    for _ in range(25):
        R = int(rng.choice([80, 200, 400])); L = int(rng.choice([30, 90, 150]))
        ref = _rnd(rng, R)
        st = int(rng.integers(0, max(1, R - L)))
        q = _mut(ref[st:st + L], rng, float(rng.choice([0.02, 0.1, 0.25]))) or 'A'
        want = oracle_align(ref, q, m, x, o, e)
        if want is None or want['score'] == 0:
            continue
        r = encode(ref)[want['ref_begin']:want['ref_end'] + 1]; rd = encode(q)[want['query_begin']:want['query_end'] + 1]
        w = abs(len(r) - len(rd)) + 1
        mat = make_mat(m, x)
        while True:
            it, codes = band_pass(r, rd, mat, 5, o, e, w, True)
            if not (it < want['score'] and 2 * w < 2 * len(rd)):
                break
            w *= 2
        got = walk_ops_with_runs(codes, w, len(rd), len(r))
        if got is None:
            continue
        assert got == want['cigar'], (len(q), len(ref))
        done += 1
    assert done > 12
