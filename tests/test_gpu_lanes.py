"""K1l (csrc/ssw_lanes.hip): one alignment per lane for references of at most 64 columns -- the junction alignments of the collapse stage
(CIRI_long/collapse.py:161-173, 251-256, 373-387).  Bit-exact against the CPU statement of the reference's passes (oracle/ssw_oracle.c, itself
held to the reference's own libssw.so by tests/test_oracle_vs_ref.py), and against this library's other kernel classes on the same batch
(CLH_NO_LANES=1)."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

FIELDS = ('score1', 'ref_begin1', 'ref_end1', 'read_begin1', 'read_end1')


def _batch(rng, n, rmax, lmax, nfrac=0.03, related=0.7):
    refs, qs = [], []
    for _ in range(n):
        R = int(rng.integers(1, rmax + 1)); L = int(rng.integers(1, lmax + 1))
        ref = rng.integers(0, 4, R, dtype=np.int8)
        if rng.random() < related:          # the read holds a noisy copy of the reference somewhere (the junction inside a read)
            from ciri_long_amd import synth
            core = synth.mutate(ref, rng, sub=0.05, ins=0.04, dele=0.04)
            pad = max(0, L - len(core))
            a = int(rng.integers(0, pad + 1))
            q = np.concatenate([rng.integers(0, 4, a, dtype=np.int8), core, rng.integers(0, 4, pad - a, dtype=np.int8)])[:max(L, 1)]
        else:
            q = rng.integers(0, 4, L, dtype=np.int8)
        if len(q) == 0:
            q = np.zeros(1, dtype=np.int8)
        if rng.random() < nfrac:
            q = q.copy(); q[rng.integers(0, len(q))] = 4
        if rng.random() < nfrac:
            ref = ref.copy(); ref[rng.integers(0, len(ref))] = 4
        refs.append(ref); qs.append(q.astype(np.int8))
    return refs, qs


def _run(ctx, refs, qs, scheme, want_cigar=True, flag=1, score_size=2):
    from ciri_long_amd import hip
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    m, x, go, ge = scheme
    return ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(m, x), go, ge, flag=flag, score_size=score_size, want_score2=False, want_cigar=want_cigar)


def _classes(ctx, refs, qs, scheme):
    from ciri_long_amd import hip
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    m, x, go, ge = scheme
    plan = ctx.plan(ro, fo, hip.score_matrix(m, x), go, ge, flag=1, score_size=2, want_score2=False, want_cigar=True)
    seg = plan.segments()
    plan.close()
    return seg


@pytest.mark.parametrize('scheme', [(10, 4, 8, 2), (2, 2, 3, 1), (1, 1, 1, 1), (5, 4, 6, 6), (3, 5, 7, 7)])
def test_lanes_class_equals_the_oracle(scheme):
    """every column class (20 / 32 / 52 / 64), reads of 1..250 bases, N in reads and references; scores through the 8-bit limit with
    gap_open > gap_extend (the word regime's flag and zero-score conventions), below it with gap_open == gap_extend"""
    from ciri_long_amd import hip
    ctx = hip.default_context()
    rng = np.random.default_rng(sum(scheme))
    m, x, go, ge = scheme
    quirk = go <= ge
    rmax = 64 if not quirk else max(1, min(64, (254 - x) // m))          # gap_open == gap_extend: only scores that stay in the 8-bit regime
    refs, qs = _batch(rng, 900, rmax, 250)
    seg = _classes(ctx, refs, qs, scheme)
    assert sum(c for rv, c, _a, _b in seg if -9 <= rv <= -5) == len(refs), seg       # all of them are K1l's (up to 2048 cells) or the transposed class's
    assert sum(c for rv, c, _a, _b in seg if -8 <= rv <= -5) >= 200
    rows, cig = _run(ctx, refs, qs, scheme)
    for k in range(len(refs)):
        w = oracle_lib.oracle_align(refs[k], qs[k], m, x, go, ge)
        r = rows[k]
        assert [int(r[f]) for f in FIELDS] == [w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']], (k, len(refs[k]), len(qs[k]))
        assert [int(c) for c in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == w['cigar'], k
        assert bool(int(r['status']) & hip.ST_WORD) == (w['score'] + x >= 255)


def test_lanes_class_equals_the_other_classes_on_a_large_batch(monkeypatch):
    """40 000 junction-shaped alignments at the collapse scoring -- enough for long reads to be sent to K1l too ("many") -- against the same
    batch through K1s / K1w / the anti-diagonal classes (CLH_NO_LANES=1); a seeded sample against the oracle; flag 0 (no begin positions)"""
    from ciri_long_amd import hip
    ctx = hip.default_context()
    rng = np.random.default_rng(5)
    scheme = (10, 4, 8, 2)
    refs, qs = _batch(rng, 39000, 50, 120, related=0.8)
    r2, q2 = _batch(rng, 1000, 64, 1500, related=0.9)                                  # long reads against a short reference
    refs += r2; qs += q2
    monkeypatch.delenv('CLH_NO_LANES', raising=False)
    seg = _classes(ctx, refs, qs, scheme)
    assert sum(c for rv, c, _a, _b in seg if -8 <= rv <= -5) == len(refs)
    rows, cig = _run(ctx, refs, qs, scheme)
    rows0, _ = _run(ctx, refs, qs, scheme, want_cigar=False, flag=0)
    monkeypatch.setenv('CLH_NO_LANES', '1')
    seg = _classes(ctx, refs, qs, scheme)
    assert not any(-8 <= rv <= -5 for rv, _c, _a, _b in seg)
    want, wcig = _run(ctx, refs, qs, scheme)
    monkeypatch.delenv('CLH_NO_LANES')
    for f in FIELDS + ('cigar_len', 'status'):
        assert np.array_equal(rows[f], want[f]), f
    assert np.array_equal(cig, wcig)
    assert np.array_equal(rows0['score1'], want['score1']) and np.array_equal(rows0['ref_end1'], want['ref_end1']) and np.array_equal(rows0['read_end1'], want['read_end1'])
    for k in rng.choice(len(refs), 300, replace=False):
        w = oracle_lib.oracle_align(refs[k], qs[k], *scheme)
        assert [int(rows[k][f]) for f in FIELDS] == [w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']], k


def test_what_the_class_leaves_to_the_others():
    """second best wanted, a reference above 64 columns, gap_open == gap_extend with a score that can reach the 16-bit regime, a few long reads: not K1l's"""
    from ciri_long_amd import hip
    ctx = hip.default_context()
    rng = np.random.default_rng(9)
    refs, qs = _batch(rng, 200, 64, 200)
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    plan = ctx.plan(ro, fo, hip.score_matrix(10, 4), 8, 2, flag=1, score_size=2, want_score2=True, want_cigar=False)
    assert not any(-8 <= rv <= -5 for rv, _c, _a, _b in plan.segments())
    plan.close()
    assert not any(-8 <= rv <= -5 for rv, _c, _a, _b in _classes(ctx, [rng.integers(0, 4, 65, dtype=np.int8)] * 4, qs[:4], (10, 4, 8, 2)))
    assert not any(-8 <= rv <= -5 for rv, _c, _a, _b in _classes(ctx, [rng.integers(0, 4, 60, dtype=np.int8)] * 4, [rng.integers(0, 4, 80, dtype=np.int8)] * 4, (10, 4, 2, 2)))
    long_reads = [rng.integers(0, 4, 2000, dtype=np.int8)] * 4
    seg = _classes(ctx, [rng.integers(0, 4, 50, dtype=np.int8)] * 4, long_reads, (10, 4, 8, 2))
    assert not any(-8 <= rv <= -5 for rv, _c, _a, _b in seg) and any(rv == -9 for rv, _c, _a, _b in seg)       # (the transposed class takes them)


@pytest.mark.parametrize('scheme', [(10, 4, 8, 2), (1, 1, 1, 1), (2, 3, 5, 2)])
def test_transposed_class_long_reads_against_short_references(scheme, monkeypatch):
    """class -9 (ssw_scan_wide.hip, ssw_scanw_tr_kernel): the reference's bases as the rows, the read's as the columns the lanes own -- for a
    batch too small to fill the GPU's lanes with long reads.  Reads of 300..5000 bases (one to five column chunks), references of 1..64, the
    junction once or twice in the read (ties between equal maxima: first reference position, then first read position), N on both sides;
    against the oracle and against the other classes"""
    from ciri_long_amd import hip, synth
    ctx = hip.default_context()
    rng = np.random.default_rng(40 + sum(scheme))
    refs, qs = [], []
    for k in range(700):
        R = int(rng.integers(1, 65)); L = int(rng.choice([300, 600, 1023, 1024, 1025, 1500, 2048, 2100, 3000, 5000]))
        ref = rng.integers(0, 4, R, dtype=np.int8)
        q = rng.integers(0, 4, L, dtype=np.int8)
        for _copy in range(int(rng.integers(0, 3))):
            core = synth.mutate(ref, rng, sub=0.04, ins=0.03, dele=0.03) if rng.random() < 0.7 else ref.copy()
            a = int(rng.integers(0, L - len(core)))
            q[a:a + len(core)] = core
        if rng.random() < 0.1:
            q[rng.integers(0, L, 3)] = 4
        if rng.random() < 0.1:
            ref[rng.integers(0, R)] = 4
        refs.append(ref); qs.append(q)
    monkeypatch.delenv('CLH_NO_LANES', raising=False)
    seg = _classes(ctx, refs, qs, scheme)
    assert sum(c for rv, c, _a, _b in seg if rv == -9) >= 0.9 * len(refs), seg           # (R * L <= 2048 stays with K1l)
    rows, cig = _run(ctx, refs, qs, scheme)
    monkeypatch.setenv('CLH_NO_LANES', '1')
    assert not any(rv == -9 or -8 <= rv <= -5 for rv, _c, _a, _b in _classes(ctx, refs, qs, scheme))
    want, wcig = _run(ctx, refs, qs, scheme)
    monkeypatch.delenv('CLH_NO_LANES')
    for f in FIELDS + ('cigar_len', 'status'):
        assert np.array_equal(rows[f], want[f]), (f, np.nonzero(rows[f] != want[f])[0][:5])
    assert np.array_equal(cig, wcig)
    m, x, go, ge = scheme
    for k in rng.choice(len(refs), 120, replace=False):
        w = oracle_lib.oracle_align(refs[k], qs[k], m, x, go, ge)
        assert [int(rows[k][f]) for f in FIELDS] == [w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']], k
