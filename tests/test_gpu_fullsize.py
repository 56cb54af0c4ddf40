"""BASELINE.json's full sizes (C2: 10 000 alignments of ~1 kb reads vs 2 kb windows; C3: 100 000 reads through the
consensus kernels) checked through properties that do not need the oracle on every item: a read's result must not depend
on the batch it travels in (order reversed, batch split into pieces), and a seeded sample must equal the oracle."""
import zlib

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def _digest(rows, fields):
    return zlib.crc32(np.ascontiguousarray(np.stack([rows[f].astype(np.int64) for f in fields], axis=1)).tobytes())


def test_c2_full_size_batch_invariance_and_sample():
    from ciri_long_amd import hip, synth
    reads, wins = synth.c2_batch(10000, seed=synth.SEEDS['C2'])
    ctx = hip.default_context()
    mat = hip.score_matrix(1, 1)
    fields = ('score1', 'score2', 'ref_begin1', 'ref_end1', 'read_begin1', 'read_end1', 'ref_end2', 'cigar_len')

    def run(idx):
        rd, ro = hip.pack([reads[i] for i in idx]); fd, fo = hip.pack([wins[i] for i in idx])
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, mat, 1, 1)
        cigs = [cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']].tobytes() for r in rows]
        return rows, cigs

    order = np.arange(10000)
    rows_a, cig_a = run(order)
    rows_b, cig_b = run(order[::-1])
    assert int((rows_a['status'] & ~9).sum()) == 0
    assert _digest(rows_a, fields) == _digest(rows_b[::-1], fields)          # order of the batch does not matter
    assert cig_a == cig_b[::-1]
    parts = [run(order[k::4]) for k in range(4)]                                # nor how it is split
    for k, (rows_p, cig_p) in enumerate(parts):
        assert _digest(rows_p, fields) == _digest(rows_a[k::4], fields)
        assert cig_p == cig_a[k::4]
    rng = np.random.default_rng(1)
    for i in rng.choice(10000, 60, replace=False):
        w = oracle_lib.oracle_align(wins[i], reads[i], 1, 1, 1, 1)
        r = rows_a[i]
        assert [int(r[f]) for f in fields[:7]] == [w['score'], w['score2'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end'], w['ref_end2']]
        assert cig_a[i] == np.asarray(w['cigar'], dtype=np.uint32).tobytes()


def test_c3_full_size_batch_invariance_and_sample():
    from ciri_long_amd import hip, synth
    n = 100000
    reads, _ = synth.c2_batch(n, seed=synth.SEEDS['C3'])
    ctx = hip.default_context()

    def run(idx):
        data, off = hip.pack([reads[i] for i in idx])
        rows, segs, ccs = ctx.ccs_batch(data, off)
        sums = np.array([zlib.crc32(ccs[off[k]:off[k] + int(rows['ccs_len'][k])].tobytes()) for k in range(len(idx))], dtype=np.int64)
        return rows, segs, sums

    order = np.arange(n)
    rows_a, segs_a, sums_a = run(order)
    assert int((rows_a['status'] != 0).sum()) == 0
    assert 0.3 * n < int((rows_a['nseg'] > 0).sum()) < 0.5 * n                 # half of the reads are linear negatives
    rows_b, segs_b, sums_b = run(order[::-1])
    for f in ('nseg', 'ccs_len', 'period'):
        assert np.array_equal(rows_a[f], rows_b[f][::-1]), f
    assert np.array_equal(sums_a, sums_b[::-1])
    valid = np.arange(65)[None, :] < rows_a['nseg'][:, None]
    assert np.array_equal(segs_a[valid], segs_b[::-1][valid])
    sub = order[7::10]
    rows_c, segs_c, sums_c = run(sub)                                           # a tenth of the batch on its own
    assert np.array_equal(rows_a['ccs_len'][sub], rows_c['ccs_len']) and np.array_equal(sums_a[sub], sums_c)
    rng = np.random.default_rng(2)
    B = np.frombuffer(b'ACGTN', dtype=np.uint8)
    for i in rng.choice(n, 150, replace=False):
        seg, ccs, _ = oracle_lib.oracle_find_consensus(reads[i])
        nseg = int(rows_a['nseg'][i])
        got = ';'.join('%d-%d' % (segs_a[i, k, 0], segs_a[i, k, 1]) for k in range(nseg)) if nseg > 0 else None
        assert got == seg, i
        if seg is not None:
            assert int(rows_a['ccs_len'][i]) == len(ccs) and int(sums_a[i]) == zlib.crc32(oracle_lib.encode(ccs).tobytes())
