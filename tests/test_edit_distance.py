"""K4 (unit-cost edit distance) -- the integer CIRI-long's distance(x, y) returns (CIRI_long/utils.py:153-159).
python-Levenshtein and edlib are absent; the quantity is uniquely defined, so the oracle (oracle/edit_oracle.c, the
textbook dynamic programme) is pinned by known answers and by the metric properties, and the kernel by the oracle."""
import random

import numpy as np
import pytest

import oracle_lib

KNOWN = [   # classic known answers of the Levenshtein distance
    ('kitten', 'sitting', 3), ('flaw', 'lawn', 2), ('saturday', 'sunday', 3), ('intention', 'execution', 5),
    ('', '', 0), ('', 'ACGT', 4), ('ACGT', '', 4), ('ACGT', 'ACGT', 0), ('ACGT', 'TGCA', 4), ('AAAA', 'AAA', 1),
    ('GATTACA', 'GCATGCU', 4), ('ACGTACGTACGT', 'ACGTTACGTACG', 2), ('a', 'A', 1),
]


def _rand_pair(rng, la, alpha='ACGT', related=True):
    a = ''.join(rng.choice(alpha) for _ in range(la))
    if not related:
        return a, ''.join(rng.choice(alpha) for _ in range(rng.randint(0, 2 * la + 3)))
    b = list(a)
    for _ in range(rng.randint(0, max(1, la // 6))):
        r = rng.random()
        if b and r < 0.34:
            b[rng.randrange(len(b))] = rng.choice(alpha)
        elif b and r < 0.67:
            del b[rng.randrange(len(b))]
        else:
            b.insert(rng.randint(0, len(b)), rng.choice(alpha))
    return a, ''.join(b)


def test_oracle_known_answers_and_metric_properties():
    for x, y, d in KNOWN:
        assert oracle_lib.oracle_edit_distance(x, y) == d, (x, y)
        assert oracle_lib.oracle_edit_distance(y, x) == d
    rng = random.Random(11)
    for _ in range(200):
        a, b = _rand_pair(rng, rng.randint(0, 120), related=rng.random() < 0.5)
        c, _ = _rand_pair(rng, rng.randint(0, 120))
        dab, dbc, dac = (oracle_lib.oracle_edit_distance(*p) for p in ((a, b), (b, c), (a, c)))
        assert dab == oracle_lib.oracle_edit_distance(b, a)
        assert abs(len(a) - len(b)) <= dab <= max(len(a), len(b))
        assert dac <= dab + dbc                                   # triangle inequality
        assert (dab == 0) == (a == b)


def test_block_model_equals_oracle():
    """tools/edit_model.py states the kernel's recurrences in Python; it must agree with the oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(oracle_lib.ROOT, 'tools'))
    import edit_model
    rng = random.Random(3)
    for _ in range(150):
        a, b = _rand_pair(rng, rng.randint(1, 300), alpha='ACGTN', related=rng.random() < 0.7)
        assert edit_model.blocks(a, b) == oracle_lib.oracle_edit_distance(a, b)


@pytest.mark.gpu
def test_kernel_equals_oracle_on_every_size_class():
    from ciri_long_amd import utils
    rng = random.Random(2021)
    xs, ys = [], []
    for x, y, _ in KNOWN:
        xs.append(x); ys.append(y)
    for la in [1, 2, 19, 20, 21, 50, 51, 63, 64, 65, 127, 128, 129, 200, 511, 512, 513, 1000, 1500, 2049, 4096]:
        for rel in (True, False):
            a, b = _rand_pair(rng, la, related=rel)
            xs.append(a); ys.append(b)
    for _ in range(300):                                            # the 20-symbol junction probes of curate_junction
        a, b = _rand_pair(rng, 20, related=rng.random() < 0.8)
        xs.append(a); ys.append(b)
    xs += ['ACGTN' * 30, 'acgtACGT', 'N' * 100]; ys += ['ACGNT' * 30, 'ACGTacgt', 'N' * 64 + 'A' + 'N' * 40]
    got = utils.distance_batch(xs, ys)
    for k, (x, y) in enumerate(zip(xs, ys)):
        assert int(got[k]) == oracle_lib.oracle_edit_distance(x, y), (k, len(x), len(y))
    assert utils.distance('kitten', 'sitting') == 3


@pytest.mark.gpu
def test_pairwise_matrix_as_cluster_sequence_builds_it():
    """collapse.py:466-473: symmetric, zero diagonal, distance / max(len)."""
    from ciri_long_amd import utils
    rng = random.Random(8)
    base = ''.join(rng.choice('ACGT') for _ in range(600))
    seqs = [utils.compress_seq(_rand_pair(random.Random(k), 0)[0] or ''.join(
        c if rng.random() > 0.08 else rng.choice('ACGT') for c in base)) for k in range(30)]
    dist = utils.pairwise_distance(seqs)
    assert dist.shape == (30, 30) and np.allclose(dist, dist.T) and np.all(np.diag(dist) == 0)
    for i in range(0, 30, 7):
        for j in range(30):
            want = oracle_lib.oracle_edit_distance(seqs[i], seqs[j]) / max(len(seqs[i]), len(seqs[j]))
            assert dist[i][j] == want


@pytest.mark.gpu
def test_many_symbols_and_long_strings():
    from ciri_long_amd import hip, utils
    assert utils.distance('ABCDEFGHI', 'ABCDEFGHJ') == 1                # more than 8 distinct symbols: 8 bit planes
    assert utils.distance('A' * 5000, 'C' * 5000) == 5000              # shorter string above 4096: passes of 64 blocks
    assert utils.distance('A' * 5000, 'C' * 100) == 5000
    rng = random.Random(77)
    xs, ys = [], []
    for la in (4097, 5000, 8192, 8193, 12000):
        a, b = _rand_pair(rng, la, related=True)
        xs += [a, a]; ys += [b, ''.join(rng.choice('ACGT') for _ in range(la + 300))]
    got = utils.distance_batch(xs + ['ACGT'], ys + ['AGGT'])
    for x, y, d in zip(xs + ['ACGT'], ys + ['AGGT'], got):
        assert int(d) == oracle_lib.oracle_edit_distance(x, y), (len(x), len(y))
