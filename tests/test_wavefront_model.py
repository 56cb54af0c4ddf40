"""tools/wavefront_model.py (numpy model of the kernel's anti-diagonal dataflow) against the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from oracle_lib import encode, make_mat, mask_len, oracle_align  # noqa: E402
from wavefront_model import wf_align, wf_pass  # noqa: E402


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


KEYS = ('score', 'score2', 'ref_begin', 'ref_end', 'query_begin', 'query_end', 'ref_end2')


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1)])
def test_model_align_equals_oracle(scheme):
    rng = np.random.default_rng(sum(scheme) * 7)
    m, x, o, e = scheme
    mat = make_mat(m, x)
    for it in range(60):
        L = int(rng.choice([18, 45, 120, 270, 400]))
        R = int(rng.choice([50, 300, 700]))
        ref = _rnd(rng, R)
        st = int(rng.integers(0, max(1, R - L)))
        q = _mut(ref[st:st + L], rng, float(rng.choice([0.05, 0.2])))
        if rng.random() < 0.2:
            q = q + q[:len(q) // 2]
        if rng.random() < 0.15:
            ref = ref[:R // 2] + 'N' * 5 + ref[R // 2:]
        if rng.random() < 0.1:
            q = _rnd(rng, L)
        if not q:
            continue
        want = oracle_align(ref, q, m, x, o, e)
        got = wf_align(encode(ref), encode(q), mat, 5, o, e, mask_len(len(q)))
        for k in KEYS:
            assert got[k] == want[k], (it, k, got, want)


def test_model_zero_score_and_tiny():
    mat = make_mat(1, 1)
    for ref, q in (('AAAAAAAAAA', 'CCCCCCCC'), ('ACGT', 'ACGT'), ('A', 'A'), ('ACGTNNNNACGT', 'ACGTACGTACGT')):
        want = oracle_align(ref, q)
        got = wf_align(encode(ref), encode(q), mat, 5, 1, 1, mask_len(len(q)))
        for k in KEYS:
            assert got[k] == want[k], (ref, q, k, got, want)


def test_model_row_capacity_variants():
    """Same answer whatever RV (rows per virtual lane) the launcher picks."""
    rng = np.random.default_rng(3)
    mat = make_mat(1, 1)
    ref = _rnd(rng, 300)
    q = _mut(ref[40:240], rng, 0.1)
    base = None
    for RV in (2, 3, 4, 8):
        r = wf_pass(encode(ref).astype(np.int64), encode(q).astype(np.int64), mat, 5, 1, 1, 1, 0, 65535, RV=RV)
        key = (r['max'], r['col'], r['row'], tuple(r['colmax']))
        base = base or key
        assert key == base
