"""tools/longwin_model.py: the decision logic of K1w on long windows (csrc/ssw_scan_wide.hip, class -4) -- the bound of a read in
pieces, the seed, the window's regime (ssw.c:804-809 decides it on the WHOLE window), candidate regions run as whole alignments, the
best row -- with the oracle as the region aligner, against the oracle's answer on the whole window.  The cases sit on both sides of the
8-bit limit: exact copies (word regime), noisy copies (byte regime), copies whose bound allows an overflow that does not happen (both
regimes computed), two loci of which only one overflows, reads of several pieces, absent reads."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from longwin_model import longwin_align  # noqa: E402
from oracle_lib import make_mat, oracle_align  # noqa: E402


def _mut(s, rng, p):
    out = []
    for c in s:
        u = rng.random()
        if u < p / 3:
            continue
        if u < 2 * p / 3:
            out.append(int(rng.integers(0, 4))); continue
        out.append(int(c))
        if u < p:
            out.append(int(rng.integers(0, 4)))
    return np.array(out or [0], dtype=np.int8)


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1)])
def test_long_window_logic_equals_the_whole_window_answer(scheme):
    m, x, go, ge = scheme
    mat = make_mat(m, x)

    def align(ref, read, ss):
        r = oracle_align(ref, read, m, x, go, ge, score_size=ss)
        r['word'] = True if ss == 1 else oracle_align(ref, read, m, x, go, ge, score_size=0) is None
        return r
    rng = np.random.Generator(np.random.PCG64(900 + m))
    lo = {1: 255, 10: 30, 2: 127}[m]
    modes, pruned = set(), 0
    for k in range(26):
        R = int(rng.choice([9000, 20000, 33000]))
        L = int(rng.integers(lo, lo + 50)) if k % 3 else int(rng.integers(lo + 50, 3 * lo + 200))
        ref = rng.integers(0, 4, R).astype(np.int8)
        pos = int(rng.integers(0, R - L))
        err = [0.0, 0.02, 0.035, 0.12, 0.0, 0.3][k % 6]
        read = _mut(ref[pos:pos + L], rng, err)
        if k % 6 == 4 and pos > 3 * L + 600:              # a second, slightly noisy copy earlier: one locus overflows, the other may not
            far = pos - 2 * L - 500
            ref[far:far + L] = np.resize(_mut(ref[pos:pos + L], rng, 0.04), L)
        if k % 13 == 7:
            read = rng.integers(0, 4, L).astype(np.int8)
        want = oracle_align(ref, read, m, x, go, ge)
        for force_static in (False, True):
            got, info = longwin_align(ref, read, mat, 5, go, ge, align, phase=int(rng.integers(0, 256)), force_static=force_static)
            assert (got['score'], got['ref_begin'], got['ref_end'], got['query_begin'], got['query_end']) == \
                (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end']), (k, L, R, info, got, want)
            modes.add(info['mode']); pruned += int(info['pruned'])
    assert pruned >= 10 and (m != 1 or modes == {1, 2, 3}), (modes, pruned)


def _two_loci(rng, L=262, R=30000):
    """a read whose locus with the FEWEST edits (five substitutions: 262 - 10 = 252) does not overflow the 8-bit pass while a locus
    with more edits (six extra window bases: 262 - 6 = 256) does: the seed is the first, the window's regime is decided by the second"""
    read = rng.integers(0, 4, L).astype(np.int8)
    ref = rng.integers(0, 4, R).astype(np.int8)
    a = read.copy()
    for p in rng.choice(np.arange(20, L - 20), 5, replace=False):
        a[p] = (a[p] + 1 + rng.integers(0, 3)) % 4
    b = list(read)
    for p in sorted(rng.choice(np.arange(30, L - 30), 6, replace=False), reverse=True):
        b.insert(int(p), int(rng.integers(0, 4)))
    pa, pb = 4000, 21000
    ref[pa:pa + L] = a
    ref[pb:pb + len(b)] = np.array(b, dtype=np.int8)
    return ref, read


def test_the_seed_does_not_overflow_but_another_locus_does():
    mat = make_mat(1, 1)

    def align(ref, read, ss):
        r = oracle_align(ref, read, 1, 1, 1, 1, score_size=ss)
        r['word'] = True if ss == 1 else oracle_align(ref, read, 1, 1, 1, 1, score_size=0) is None
        return r
    rng = np.random.Generator(np.random.PCG64(4242))
    seen = 0
    for _ in range(6):
        ref, read = _two_loci(rng)
        want = oracle_align(ref, read, 1, 1, 1, 1)
        got, info = longwin_align(ref, read, mat, 5, 1, 1, align, phase=int(rng.integers(0, 256)))
        assert (got['score'], got['ref_begin'], got['ref_end'], got['query_begin'], got['query_end']) == \
            (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end']), (info, got, want)
        seen += int(info['mode'] == 3 and got['word'])
    assert seen >= 3          # the undecided mode whose byte rows overflowed: the word rows answered
