"""oracle/poa_oracle.c (five matrices, spoa's value-comparing back-track) against tools/poa_model.py (the row-at-a-time
formulation the kernel K3 computes: prefix maxima for the horizontal gap states, clamped differences for the vertical
ones, one byte per cell to replay the back-track).  CPU only; proves the derivation, not the kernel."""
import os
import random
import sys

import numpy as np
import pytest

import oracle_lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import poa_model  # noqa: E402

PARS = [(10, -4, -8, -2, -24, -1),      # every call site of the reference (collapse.py:267,504; tests/test_poa.py:30)
        (5, -4, -8, -6, -10, -4),       # pyspoa's defaults: convex
        (5, -4, -8, -6, -8, -6),        # affine (g <= q)
        (3, -5, -4, -3, -9, -1), (2, -1, -2, -1, -3, -1)]


def _letters(s):
    return np.frombuffer(s.encode('latin-1'), dtype=np.uint8).astype(np.int64)


def _text(a):
    return bytes(int(x) for x in a).decode('latin-1')


def _mutate(rng, t, rate):
    out = []
    for ch in t:
        x = rng.random()
        if x < rate * 0.35:
            out.append(rng.choice('ACGT'))
        elif x < rate * 0.65:
            out.append(ch); out.append(rng.choice('ACGT'))
        elif x >= rate:
            out.append(ch)
    return ''.join(out) or 'A'


def _families(seed, count):
    rng = random.Random(seed)
    for _ in range(count):
        t = ''.join(rng.choice(rng.choice(['ACGT', 'ACGT', 'AC', 'ACGTN'])) for _ in range(rng.choice([20, 40, 80, 150])))
        rate = rng.choice([0, 0.05, 0.15, 0.3])
        seqs = []
        for _k in range(rng.randint(2, 8)):
            s = _mutate(rng, t, rate)
            if rng.random() < 0.3:
                s = s[rng.randrange(0, max(1, len(s) // 2)):] or 'C'
            if rng.random() < 0.3 and len(s) > 10:
                s = s[:rng.randrange(len(s) // 2, len(s))]
            if rng.random() < 0.2:
                s = s[len(s) // 3:] + s[:len(s) // 3]
            seqs.append(s)
        yield rng, seqs


@pytest.mark.parametrize('algorithm', [0, 1, 2])
def test_row_formulation_equals_matrix_statement(algorithm):
    n = 0
    for rng, seqs in _families(100 + algorithm, 40):
        par = rng.choice(PARS)
        mc = rng.choice([0, 0, (len(seqs) + 1) // 2])
        try:
            want = oracle_lib.oracle_poa(seqs, algorithm, True, *par, with_scores=True, min_coverage=mc, with_order=True)
        except ValueError:                                   # an alignment without a base: spoa throws, so does the model
            with pytest.raises(ValueError):
                poa_model.poa([_letters(s) for s in seqs], algorithm, True, *par, min_coverage=mc)
            continue
        cons, rows, scores, order = poa_model.poa([_letters(s) for s in seqs], algorithm, True, *par, min_coverage=mc)
        got = (_text(cons), [_text(r) for r in rows], scores, order)
        assert got == tuple(want), (seqs, par, mc)
        n += 1
    assert n >= 38


def test_gap_model_of_the_call_sites():
    """a gap of k bases costs max(g + (k-1) e, q + (k-1) c): end-cell score of a sequence with one deletion of k bases"""
    rng = random.Random(5)
    t = ''.join(rng.choice('ACGT') for _ in range(300))
    for k in (1, 2, 3, 10, 22, 23, 24, 25, 40):
        s = t[:150] + t[150 + k:]
        _, _, scores = oracle_lib.oracle_poa([t, s], 1, False, 10, -4, -8, -2, -24, -1, with_scores=True)
        assert scores[1] == 10 * (300 - k) + max(-8 - 2 * (k - 1), -24 - (k - 1)), k
    # local and overlap modes are different code paths with different answers on the same input
    seqs = [t, 'GGGGGGGG' + t[40:200] + 'CCCCCCC']
    a = oracle_lib.oracle_poa(seqs, 0, True, with_scores=True)
    b = oracle_lib.oracle_poa(seqs, 2, True, with_scores=True)
    c = oracle_lib.oracle_poa(seqs, 1, True, with_scores=True)
    assert a[2][1] >= 1600 and b[2][1] < a[2][1] and c[2][1] < b[2][1]
    assert all(len(r) == len(a[1][0]) for r in a[1]) and a[1][0].replace('-', '') == t


def test_parameter_rules():
    with pytest.raises(ValueError):
        oracle_lib.oracle_poa(['ACGT', 'ACGT'], 3)
    with pytest.raises(ValueError):
        oracle_lib.oracle_poa(['ACGT', 'ACGT'], 0, False, 5, -4, 8, -6, -10, -4)
    # linear sub-type (g >= e) is stated by the oracle; the kernel refuses it (tests/test_gpu_ccs.py)
    assert oracle_lib.oracle_poa(['ACGTACGTAC', 'ACGTTCGTAC', 'ACGTACGTAC'], 1, False, 5, -4, -8, -8, -8, -8) == 'ACGTACGTAC'
    with pytest.raises(NotImplementedError):
        poa_model.Params(0, 5, -4, -8, -8, -8, -8)


@pytest.mark.parametrize('algorithm', [0, 1, 2])
def test_lazy_back_track_from_h_and_differences(algorithm, monkeypatch):
    """the same families with the back-track replayed from H and the four clamped differences a cell keeps in the lazy forward
    pass (no code byte): K3's formulation since round 3"""
    monkeypatch.setattr(poa_model, 'LAZY', True)
    test_row_formulation_equals_matrix_statement(algorithm)
