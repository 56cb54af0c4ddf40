"""Randomized check of the restatement against the reference's own ssw.c compiled into oracle/_ref/libssw.so
(recipe: oracle/Makefile).  Skipped when that build is absent."""
import numpy as np
import pytest

from oracle_lib import have_ref, oracle_align, ref_align

pytestmark = pytest.mark.skipif(not have_ref(), reason='oracle/_ref/libssw.so not built')


def _rnd(rng, n):
    return ''.join('ACGT'[i] for i in rng.integers(0, 4, n))


def _mutate(s, rng, p=0.12):
    out = []
    for c in s:
        u = rng.random()
        if u < p / 3:
            continue
        if u < 2 * p / 3:
            out.append('ACGT'[rng.integers(4)])
            continue
        out.append(c)
        if u < p:
            out.append('ACGT'[rng.integers(4)])
    return ''.join(out)


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1), (3, 5, 7, 7), (2, 3, 0, 0)])
def test_random_pairs(scheme):
    rng = np.random.default_rng(hash(scheme) & 0xffff)
    for _ in range(250):
        L = int(rng.choice([21, 33, 64, 130, 255, 300, 520]))
        R = int(rng.choice([40, 300, 900]))
        ref = _rnd(rng, R)
        st = int(rng.integers(0, max(1, R - L)))
        q = _mutate(ref[st:st + L], rng)
        if rng.random() < 0.25:
            q = q + q[: len(q) // 2]
        if rng.random() < 0.15:
            ref = ref[: R // 3] + 'N' * 9 + ref[R // 3:]
        if not q:
            continue
        assert oracle_align(ref, q, *scheme) == ref_align(ref, q, *scheme)


def test_long_vertical_gaps_across_stripe_boundaries():
    """gap_open <= gap_extend: the 16-bit pass truncates vertical gaps at stripe starts (ssw.c:468-478)."""
    rng = np.random.default_rng(7)
    for _ in range(150):
        L = int(rng.integers(300, 500))
        ref = _rnd(rng, 700)
        core = ref[100:100 + L]
        # drop long chunks from the reference side so the read carries long insertions
        q = core
        for _k in range(int(rng.integers(1, 6))):
            p = int(rng.integers(10, len(q) - 10))
            q = q[:p] + _rnd(rng, int(rng.integers(2, 12))) + q[p:]
        assert oracle_align(ref, q, 1, 1, 1, 1) == ref_align(ref, q, 1, 1, 1, 1)
        assert oracle_align(ref, q, 10, 4, 8, 2) == ref_align(ref, q, 10, 4, 8, 2)


@pytest.mark.parametrize('score_size', [0, 1, 2])
def test_score_size_variants(score_size):
    rng = np.random.default_rng(11 + score_size)
    for _ in range(80):
        L = int(rng.choice([30, 200, 400]))
        ref = _rnd(rng, 600)
        q = _mutate(ref[50:50 + L], rng)
        a = oracle_align(ref, q, 1, 1, 1, 1, score_size=score_size)
        b = ref_align(ref, q, 1, 1, 1, 1, score_size=score_size)
        assert a == b


@pytest.mark.parametrize('flag', [0, 1, 2, 8])
def test_flag_variants(flag):
    rng = np.random.default_rng(5)
    for _ in range(40):
        ref = _rnd(rng, 400)
        q = _mutate(ref[50:250], rng)
        assert oracle_align(ref, q, 2, 2, 3, 1, flag=flag) == ref_align(ref, q, 2, 2, 3, 1, flag=flag)


@pytest.mark.parametrize('flag', [2, 4, 6, 8, 10, 12, 14, 3, 5])
def test_flag_bits_with_filters(flag):
    """flag bits 1/2 with thresholds that split the inputs (ssw.c:834, 850): the scalar statement equals the reference's
    own library on every field and CIGAR (the GPU test of the same name holds the kernels to the statement)."""
    rng = np.random.default_rng(600 + flag)
    n_with = n_without = 0
    for _ in range(60):
        ref = _rnd(rng, int(rng.integers(200, 700)))
        q = _mutate(ref[50:50 + int(rng.choice([30, 60, 120, 250]))], rng)
        a = oracle_align(ref, q, 1, 1, 1, 1, flag=flag, filters=55, filterd=100)
        b = ref_align(ref, q, 1, 1, 1, 1, flag=flag, filters=55, filterd=100)
        assert a == b
        n_with += bool(a['cigar']); n_without += not a['cigar']
    assert not (flag & 6) or (n_with > 0 and n_without > 0)
