"""K2/K3 (repeat scan + partial-order consensus) through the C ABI against the CPU statements (oracle/ccs_oracle.c for
the copy boundaries, oracle/poa_oracle.c for the spoa algorithm).  PARITY UNPINNED with respect to pyccs/spoa (absent);
bit-exact with respect to the oracle."""
import json
import os

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'test_poa_input.json')) as _f:
    SEGMENTS = json.load(_f)['segments']      # the 6 copies of the reference's tests/test_poa.py:8-15


def test_reference_test_poa_input():
    """Structural expectations of SURVEY.md section 8c for the only input the reference tests hold."""
    from ciri_long_amd import pyccs
    raw = ''.join(SEGMENTS)
    seg, ccs = pyccs.find_consensus(raw)
    want = oracle_lib.oracle_find_consensus(raw)
    assert (seg, ccs) == want[:2]
    parts = seg.split(';')
    assert len(parts) >= 5
    assert 140 <= len(ccs) <= 160
    assert set(ccs) <= set('ACGT')


def test_batch_equals_oracle_on_synthetic_reads():
    from ciri_long_amd import pyccs, synth
    rng = np.random.Generator(np.random.PCG64(77))
    reads = []
    for it in range(300):
        tm = synth.template(rng)
        L = int(rng.choice([400, 700, 1000, 1300, 2000]))
        kind = it % 4
        if kind == 3:
            reads.append(synth.mutate(rng.integers(0, 4, L, dtype=np.int8), rng))
        else:
            r = synth.rolling_circle_read(rng, tm, L)
            if kind == 2 and len(r) > 50:
                r = r.copy(); r[rng.integers(0, len(r), 3)] = 4       # a few N
            reads.append(r)
    reads.append(np.zeros(40, dtype=np.int8))                     # shorter than two minimal periods
    reads.append(np.tile(np.array([0, 1, 2, 3], dtype=np.int8), 200))   # microsatellite: period below the minimum offset
    got = pyccs.find_consensus_batch(reads)
    n_found = 0
    for k, r in enumerate(reads):
        want = oracle_lib.oracle_find_consensus(r)
        assert got[k] == want[:2], (k, len(r), got[k][0], want[0])
        n_found += got[k][0] is not None
    assert n_found >= 120


def test_shapes_that_exercise_every_kernel_path_vs_oracle():
    """Copies of 90..1600 bases (2, 4 and 8 columns per lane; several column passes; far source rows), up to ~60 copies
    (in-edge weights), graphs above the LDS score capacity, low-complexity repeats (long hash chains, high in-degree),
    N-rich reads."""
    from ciri_long_amd import pyccs, synth
    rng = np.random.Generator(np.random.PCG64(4242))
    reads = []
    for p in (90, 130, 200, 260, 400, 520, 700, 1100, 1600):
        for L in (1500, 3000, 6000):
            tm = rng.integers(0, 4, p, dtype=np.int8)
            reads.append(synth.rolling_circle_read(rng, tm, L))
    unit = rng.integers(0, 4, 35, dtype=np.int8)
    reads.append(np.tile(unit, 40))                                   # exact repeat at the minimum-offset scale
    reads.append(synth.mutate(np.tile(unit, 60), rng))
    reads.append(np.zeros(700, dtype=np.int8))                        # homopolymer: one chain holds every position
    reads.append(np.tile(np.array([0, 1], dtype=np.int8), 400))
    reads.append(synth.mutate(np.tile(rng.integers(0, 4, 64, dtype=np.int8), 70), rng, sub=0.1, ins=0.1, dele=0.1))
    nrich = synth.rolling_circle_read(rng, rng.integers(0, 4, 300, dtype=np.int8), 2000).copy()
    nrich[rng.integers(0, len(nrich), 150)] = 4
    reads.append(nrich)
    got = pyccs.find_consensus_batch(reads)
    found = 0
    for k, r in enumerate(reads):
        want = oracle_lib.oracle_find_consensus(r)
        assert got[k] == want[:2], (k, len(r), got[k][0], want[0])
        found += got[k][0] is not None
    assert found >= 25


def test_reads_that_overflow_a_first_tier_slot_run_in_the_second_tier(monkeypatch):
    """K3's workspace slots are sized for the common case; with the budget squeezed, the long-period reads of this batch
    do not fit one and must come back identical from the second launch (large slots)."""
    from ciri_long_amd import pyccs, synth
    monkeypatch.setenv('CLH_POA_BUDGET_MB', '48')
    rng = np.random.Generator(np.random.PCG64(8))
    reads = []
    for p, L in ((120, 1500), (300, 2500), (900, 5000), (1300, 6000), (200, 4000), (1500, 5500)) * 4:
        reads.append(synth.rolling_circle_read(rng, rng.integers(0, 4, p, dtype=np.int8), L))
    got = pyccs.find_consensus_batch(reads)
    found = 0
    for k, r in enumerate(reads):
        want = oracle_lib.oracle_find_consensus(r)
        assert got[k] == want[:2], (k, len(r))
        found += got[k][0] is not None
    assert found >= 20


def test_reads_longer_than_the_lds_scan_limit():
    """above 16 000 bases the repeat scan runs out of an HBM workspace; same results as the oracle, mixed in one batch
    with ordinary reads"""
    from ciri_long_amd import pyccs, synth
    rng = np.random.Generator(np.random.PCG64(12))
    reads = [synth.rolling_circle_read(rng, rng.integers(0, 4, 400, dtype=np.int8), 1200),
             synth.rolling_circle_read(rng, rng.integers(0, 4, 700, dtype=np.int8), 17000),
             synth.mutate(rng.integers(0, 4, 21000, dtype=np.int8), rng),
             synth.rolling_circle_read(rng, rng.integers(0, 4, 1500, dtype=np.int8), 30000),
             synth.rolling_circle_read(rng, rng.integers(0, 4, 250, dtype=np.int8), 40000)]
    got = pyccs.find_consensus_batch(reads)
    for k, r in enumerate(reads):
        want = oracle_lib.oracle_find_consensus(r)
        assert got[k] == want[:2], (k, len(r), got[k][0] and got[k][0][:60], want[0] and want[0][:60])
    assert got[1][0] is not None and got[2][0] is None and got[4][0] is not None


def test_spoa_call_shape():
    from ciri_long_amd import hip, spoa
    cons, msa = spoa.poa(SEGMENTS, 0, True, 10, -4, -8, -2, -24, -1)     # tests/test_poa.py:30
    want = oracle_lib.oracle_poa(SEGMENTS, 0, True, 10, -4, -8, -2, -24, -1)
    assert (cons, msa) == tuple(want) and len(cons) == 144
    assert len(msa) == len(SEGMENTS) and [r.replace('-', '') for r in msa] == SEGMENTS
    cons2, msa2 = spoa.poa(SEGMENTS, 2, False, 10, -4, -8, -2, -24, -1)  # collapse.py:267,504
    assert msa2 == [] and cons2 == oracle_lib.oracle_poa(SEGMENTS, 2, False, 10, -4, -8, -2, -24, -1)
    # the three modes are three code paths: same input, different alignments
    rows = [oracle_lib.oracle_poa(SEGMENTS, a, True, 10, -4, -8, -2, -24, -1)[1][5] for a in (0, 1, 2)]
    got = [spoa.poa(SEGMENTS, a, True, 10, -4, -8, -2, -24, -1)[1][5] for a in (0, 1, 2)]
    assert got == rows and rows[0] != rows[1]
    assert spoa.poa(['ACGTACGTAA'], 0, True, 10, -4, -8, -2, -24, -1) == ('ACGTACGTAA', ['ACGTACGTAA'])
    # nothing is accepted and ignored: the linear model, gap pieces further apart than the difference fields hold, invalid modes raise
    for bad in ((1, True, -1, -1, -1, -1, -1, -1),                       # find_bsj.py:496 (dead code in the reference): m < 1
                (0, True, 5, -4, -8, -8, -8, -8),                         # linear
                (3, True, 10, -4, -8, -2, -24, -1), (0, True, 10, -4, 8, -2, -24, -1),
                (0, True, 10, -4, -8, -1, -40, -1)):
        with pytest.raises(hip.ClhError):
            spoa.poa(SEGMENTS, *bad)
    # a match score above 11 leaves the 16-bit cells of the packed pass: the wide form answers (refused until round 4)
    assert spoa.poa(SEGMENTS, 0, True, 12, -4, -8, -2, -24, -1) == tuple(oracle_lib.oracle_poa(SEGMENTS, 0, True, 12, -4, -8, -2, -24, -1))


def _family(rng, t_len, n, rate, alphabet='ACGT'):
    import random
    t = ''.join(rng.choice(alphabet) for _ in range(t_len))
    seqs = []
    for _ in range(n):
        out = []
        for ch in t:
            x = rng.random()
            if x < rate * 0.35:
                out.append(rng.choice('ACGT'))
            elif x < rate * 0.65:
                out.append(ch); out.append(rng.choice('ACGT'))
            elif x >= rate:
                out.append(ch)
        s = ''.join(out) or 'A'
        if rng.random() < 0.3:
            s = s[rng.randrange(0, max(1, len(s) // 2)):] or 'C'
        if rng.random() < 0.2:
            s = s[len(s) // 3:] + s[:len(s) // 3]
        seqs.append(s)
    return seqs


PARS = [(10, -4, -8, -2, -24, -1), (5, -4, -8, -6, -10, -4), (5, -4, -8, -6, -8, -6), (3, -5, -4, -3, -9, -1), (2, -1, -2, -1, -3, -1)]


@pytest.mark.parametrize('algorithm', [0, 1, 2])
def test_poa_modes_scores_msa_equal_the_oracle(algorithm):
    """Random families through clh_poa_batch: consensus, every MSA row and every end-cell score equal the five-matrix
    statement, for the reference's scores, pyspoa's defaults (convex), an affine set and two more; sequences of 10..1400
    bases (2..8 columns per lane, several column passes), 2..40 sequences, with and without min_coverage."""
    import random
    from ciri_long_amd import hip
    rng = random.Random(500 + algorithm)
    ctx = hip.default_context()
    shapes = [(12, 5), (40, 9), (90, 6), (150, 12), (300, 7), (520, 5), (700, 4), (1400, 3), (60, 40)]
    for it in range(45):
        t_len, n = shapes[it % len(shapes)]
        seqs = _family(rng, t_len, n, rng.choice([0, 0.05, 0.15, 0.3]), rng.choice(['ACGT', 'ACGT', 'AC', 'ACGTN']))
        par = PARS[it % len(PARS)] if it >= 5 else PARS[0]
        mc = rng.choice([0, 0, (len(seqs) + 1) // 2])
        want = oracle_lib.oracle_poa(seqs, algorithm, True, *par, with_scores=True, min_coverage=mc)
        data, off = hip.pack(seqs)
        got = ctx.poa_batch(data, off, np.array([0, len(seqs)], dtype=np.int64), algorithm=algorithm, scores=par, min_coverage=mc,
                            genmsa=True, with_scores=True)[0]
        assert got[2] == want[2][:65], (it, 'scores')
        assert got[0] == want[0], (it, 'consensus')
        assert got[1] == want[1], (it, 'msa')


def test_poa_letters_are_raw_characters_and_spoa_corner_cases():
    """spoa's alphabet is the set of raw characters: 'a' is not 'A', every IUPAC letter is its own letter (columns of up to 8
    different letters; a ninth is this kernel's stated limit); empty sequences are skipped (no MSA row), a one-base sequence
    has no edge and so no coverage; all three modes; rank order = spoa's depth-first sort (MSA columns show it)."""
    import random
    from ciri_long_amd import hip, spoa
    rng = random.Random(31)
    n = 0
    for it in range(60):
        alpha = ['ACGTacgt', 'ACGTNRYK', 'ACGTN', 'AaCc', 'ACGT'][it % 5]
        t = ''.join(rng.choice(alpha) for _ in range(rng.choice([8, 20, 60, 150])))
        seqs = []
        for _k in range(rng.randint(2, 14)):
            s = ''.join(rng.choice(alpha) if rng.random() < 0.1 else ch for ch in t if rng.random() > 0.05)
            if rng.random() < 0.25:
                s = s[rng.randrange(0, max(1, len(s) // 2)):]
            if rng.random() < 0.08:
                s = ''
            if rng.random() < 0.08:
                s = rng.choice(alpha)
            seqs.append(s)
        if not any(seqs):
            seqs.append('ACGT')
        algorithm = it % 3
        par = PARS[(it // 3) % len(PARS)]
        try:
            want = oracle_lib.oracle_poa(seqs, algorithm, True, *par)
        except ValueError:                      # an alignment without a base: spoa throws, the kernel reports status 7
            with pytest.raises(hip.ClhError, match='status 7'):
                spoa.poa(seqs, algorithm, True, *par)
            continue
        assert spoa.poa(seqs, algorithm, True, *par) == tuple(want), (it, seqs)
        n += 1
    assert n >= 50
    # nine different letters in one column: beyond the kernel's aligned-set capacity, said aloud
    col9 = ['GG' + ch + 'TT' for ch in 'ACGTNRYKM']
    assert oracle_lib.oracle_poa(col9, 1, False, *PARS[0]) is not None
    with pytest.raises(hip.ClhError, match='status 2'):
        spoa.poa(col9, 1, False, *PARS[0])
    assert spoa.poa(col9[:8], 1, True, *PARS[0]) == tuple(oracle_lib.oracle_poa(col9[:8], 1, True, *PARS[0]))


def test_poa_batch_of_groups_large_clusters_and_limits():
    """Several groups in one call (the batch form of collapse.py:504), a cluster of 150 sequences (the reference's
    correct_cluster takes up to 200 reads), and what used to be refused for its size: a sequence above 2800 bases, scores outside the 16-bit cells."""
    import random
    from ciri_long_amd import hip, spoa
    rng = random.Random(9)
    ctx = hip.default_context()
    groups = [_family(rng, rng.choice([30, 80, 200]), rng.randint(1, 12), 0.15) for _ in range(40)]
    groups.append(_family(rng, 120, 150, 0.12))
    flat = [s for g in groups for s in g]
    data, off = hip.pack(flat)
    goff = np.cumsum([0] + [len(g) for g in groups]).astype(np.int64)
    got = ctx.poa_batch(data, off, goff, algorithm=2, scores=PARS[0])
    for k, g in enumerate(groups):
        assert got[k] == oracle_lib.oracle_poa(g, 2, False, *PARS[0]), k
    # what the 16-bit cells of the packed pass cannot hold runs the wide (32-bit) form of the pass, as spoa's engines fall back to
    # wider cells: a sequence above 2800 bases (round 3 refused it) ...
    long_pair = ['ACGT' * 701, 'ACGT' * 700]
    assert spoa.poa(long_pair, 2, False, 10, -4, -8, -2, -24, -1)[0] == oracle_lib.oracle_poa(long_pair, 2, False, 10, -4, -8, -2, -24, -1)
    # ... a score set whose row frames do not fit 16 bits (refused up front before) ...
    pair = ['ACGTACGT' * 9, 'ACGTTCGT' * 9]
    assert spoa.poa(pair, 0, True, 11, -4, -8, -6, -10, -5) == tuple(oracle_lib.oracle_poa(pair, 0, True, 11, -4, -8, -6, -10, -5))
    assert spoa.poa(pair, 1, True, 40, -30, -50, -44, -60, -30) == tuple(oracle_lib.oracle_poa(pair, 1, True, 40, -30, -50, -44, -60, -30))
    # ... and a global alignment that reaches the floor of the 16-bit range in the middle of the packed pass (status 6 before): two
    # unrelated 2600-base sequences under an affine cost of 6 per gap base sink below -30000 -- the read runs once more in the wide form
    a = ''.join(rng.choice('AC') for _ in range(2600)); b = ''.join(rng.choice('GT') for _ in range(2600))
    assert spoa.poa([a, b], 1, False, 2, -100, -9, -6, -9, -6)[0] == oracle_lib.oracle_poa([a, b], 1, False, 2, -100, -9, -6, -9, -6)
    assert spoa.poa([a, b], 0, False, 2, -100, -9, -6, -9, -6)[0] == oracle_lib.oracle_poa([a, b], 0, False, 2, -100, -9, -6, -9, -6)


def test_a_group_of_the_second_launch_that_leaves_the_16_bit_range_still_runs_wide(monkeypatch):
    """with the large workspace slots scarce, groups that found none free run in the packed kernel's second launch; one that
    reaches the floor of the 16-bit cells THERE used to keep status 6 (spoa has no such limit: its engines fall back to wider cells).
    The wide kernel runs behind both launches, so the second launch hands over as the first does; the large slots of a global / overlap
    plan are sized for the wide form.  Also: a row of recycled memory that says "status 1" must not send a group that is on the wide
    list through the second launch as well (nothing is counted as lost)."""
    import random
    from ciri_long_amd import hip
    rng = random.Random(77)
    ctx = hip.default_context()
    monkeypatch.setenv('CLH_POA_BUDGET_MB', '1'); monkeypatch.setenv('CLH_POA_BIG_SLOTS', '1')
    groups = []
    for k in range(6):        # unrelated pairs under an affine cost of 6 per gap base: the global alignment sinks below -30000
        n = 2450 + 30 * k
        groups.append([''.join(rng.choice('AC') for _ in range(n)), ''.join(rng.choice('GT') for _ in range(n))])
    groups.append(_family(rng, 200, 5, 0.1))
    flat = [s for g in groups for s in g]
    data, off = hip.pack(flat)
    goff = np.cumsum([0] + [len(g) for g in groups]).astype(np.int64)
    scores = (2, -100, -9, -6, -9, -6)
    got = ctx.poa_batch(data, off, goff, algorithm=1, scores=scores)
    st = ctx.poa_last_stats()
    for k, g in enumerate(groups):
        assert got[k] == oracle_lib.oracle_poa(g, 1, False, *scores), k
    assert st['dropped'] == {}, st


def test_find_ccs_reads_files_and_resume(tmp_path):
    """Stage driver: FASTA/FASTQ(.gz) in, tmp/{prefix}.ccs.fa + .raw.fa out in the reference's format, resume reads them."""
    import gzip
    from ciri_long_amd import find_ccs, synth
    rng = np.random.Generator(np.random.PCG64(3))
    reads = []
    for k in range(30):
        tm = synth.template(rng)
        r = synth.rolling_circle_read(rng, tm, 900) if k % 3 else synth.mutate(rng.integers(0, 4, 900, dtype=np.int8), rng)
        reads.append(('read%02d extra words' % k, oracle_lib.decode(r)))
    (tmp_path / 'tmp').mkdir()
    fq = tmp_path / 'in.fq.gz'
    with gzip.open(fq, 'wt') as f:
        for h, s in reads:
            f.write('@%s\n%s\n+\n%s\n' % (h, s, 'I' * len(s)))
    total, ro, ccs_seq = find_ccs.find_ccs_reads(str(fq), str(tmp_path), 'p', 4, False)
    assert total == 30
    want = {h.split(' ')[0]: oracle_lib.oracle_find_consensus(s) for h, s in reads}
    keep = [h for h in want if want[h][0] is not None]
    assert ro == len(keep) and list(ccs_seq) == keep                      # input order
    for h in keep:
        assert ccs_seq[h][:2] == [want[h][0], want[h][1]]
    lines = (tmp_path / 'tmp' / 'p.ccs.fa').read_text().split('\n')
    assert lines[0] == '>%s\t%s\t%d' % (keep[0], want[keep[0]][0], len(want[keep[0]][1])) and lines[1] == want[keep[0]][1]
    assert find_ccs.load_ccs_reads(str(tmp_path), 'p') == ccs_seq
    fa = tmp_path / 'in.fa'
    fa.write_text(''.join('>%s\n%s\n' % hs for hs in reads))
    assert find_ccs.find_ccs_reads(str(fa), str(tmp_path), 'q', 1, False)[2] == ccs_seq


def test_native_file_stage_equals_the_python_loop(tmp_path):
    """clh_ccs_file (reader thread + kernels + writers in native code) against the record loop in Python: byte-identical
    tmp files and counts on FASTA and gzipped FASTQ, incl. CRLF line ends, '>>' / '@@' markers, lower case, N, an empty
    sequence, several batches."""
    import gzip
    from ciri_long_amd import find_ccs, synth
    rng = np.random.Generator(np.random.PCG64(17))
    recs = []
    for k in range(700):
        tm = synth.template(rng)
        r = synth.rolling_circle_read(rng, tm, int(rng.integers(300, 1400))) if k % 3 else synth.mutate(rng.integers(0, 4, 700, dtype=np.int8), rng)
        s = oracle_lib.decode(r)
        if k % 50 == 7:
            s = s.lower()
        if k % 50 == 9:
            s = s[:100] + 'NNNN' + s[100:]
        recs.append(('r%04d desc %d' % (k, k), s))
    recs[5] = ('empty_seq', '')
    recs[6] = ('>double_marker x', recs[6][1])
    for sub in ('a', 'b'):
        (tmp_path / sub / 'tmp').mkdir(parents=True)
    fa = tmp_path / 'in.fa'
    fa.write_bytes(''.join('>%s\r\n%s\r\n' % hs for hs in recs).encode())
    fq = tmp_path / 'in.fastq.gz'
    with gzip.open(fq, 'wt') as f:
        for h, s in recs:
            f.write('@%s\n%s\n+\n%s\n' % (h.replace('>', '@'), s, 'I' * len(s)))
    from ciri_long_amd import hip
    for path, is_fq in ((fa, 0), (fq, 1)):
        tot_n, ro_n, long_n = hip.default_context().ccs_file(str(path), is_fq, str(tmp_path / 'a' / 'tmp' / 'x.ccs.fa'), str(tmp_path / 'a' / 'tmp' / 'x.raw.fa'), 256)
        tot_p, ro_p, d_p = find_ccs.find_ccs_reads_py(str(path), str(tmp_path / 'b'), 'x', 1, False)
        assert (tot_n, ro_n, long_n) == (tot_p, ro_p, 0) and tot_n == 700 and ro_n > 200
        for name in ('x.ccs.fa', 'x.raw.fa'):
            assert (tmp_path / 'a' / 'tmp' / name).read_bytes() == (tmp_path / 'b' / 'tmp' / name).read_bytes(), (str(path), name)
        tot_d, ro_d, d_d = find_ccs.find_ccs_reads(str(path), str(tmp_path / 'a'), 'x', 1, False)
        assert (tot_d, ro_d) == (tot_p, ro_p) and d_d == d_p


def test_native_file_stage_buffer_borders_and_a_trailing_header(tmp_path, monkeypatch):
    """a file larger than the reader's 4 MiB buffer (lines split across refills take the assembling path), a last header
    without its sequence line, batches of 1000 rotating through the three threads: byte-identical to the Python loop"""
    from ciri_long_amd import find_ccs, hip, synth
    rng = np.random.Generator(np.random.PCG64(23))
    recs = []
    for k in range(6500):
        tm = synth.template(rng)
        r = synth.rolling_circle_read(rng, tm, int(rng.integers(500, 1100))) if k % 2 else synth.mutate(rng.integers(0, 4, 800, dtype=np.int8), rng)
        recs.append(('q%05d' % k, oracle_lib.decode(r)))
    for sub in ('a', 'b'):
        (tmp_path / sub / 'tmp').mkdir(parents=True)
    fq = tmp_path / 'in.fq'
    with open(fq, 'w') as f:
        for h, s in recs:
            f.write('@%s\n%s\n+\n%s\n' % (h, s, '#' * len(s)))
        f.write('@dangling')
    assert fq.stat().st_size > 2 * (4 << 20)
    tot_n, ro_n, long_n = hip.default_context().ccs_file(str(fq), 1, str(tmp_path / 'a' / 'tmp' / 'x.ccs.fa'), str(tmp_path / 'a' / 'tmp' / 'x.raw.fa'), 1000)
    tot_p, ro_p, d_p = find_ccs.find_ccs_reads_py(str(fq), str(tmp_path / 'b'), 'x', 1, False)
    assert (tot_n, ro_n, long_n) == (tot_p, ro_p, 0) and tot_n == 6501
    for name in ('x.ccs.fa', 'x.raw.fa'):
        assert (tmp_path / 'a' / 'tmp' / name).read_bytes() == (tmp_path / 'b' / 'tmp' / name).read_bytes(), name
    assert hip.fastx_count(str(fq), 1) == 6501
    # shards of a sharded run (dist.call_sharded): a rank enters the file at the indexed record at or before its shard; the shards'
    # files one after the other are the unsharded files; a small chunk size makes records straddle many chunk borders
    n, index = hip.fastx_index(str(fq), 1, every=500)
    assert n == 6501 and len(index) == 14 and index[0] == 0
    monkeypatch.setenv('CLH_FILE_CHUNK_MB', '1')
    parts = {'x.ccs.fa': b'', 'x.raw.fa': b''}
    tot = 0
    for lo, hi in ((0, 1700), (1700, 1701), (1701, 4999), (4999, 6501)):
        k = lo // 500
        t, _ro, _l = hip.default_context().ccs_file(str(fq), 1, str(tmp_path / 'a' / 'tmp' / 's.ccs.fa'), str(tmp_path / 'a' / 'tmp' / 's.raw.fa'), 700,
                                                    first_record=lo - 500 * k, max_records=hi - lo, byte_offset=index[k])
        tot += t
        for name in parts:
            parts[name] += (tmp_path / 'a' / 'tmp' / name.replace('x.', 's.')).read_bytes()
    assert tot == 6501
    for name in parts:
        assert parts[name] == (tmp_path / 'b' / 'tmp' / name).read_bytes(), name


@pytest.mark.gpu
def test_native_file_stage_a_record_longer_than_the_reserve(tmp_path, monkeypatch):
    """a sequence line of 17 M bases between ordinary records, chunks of 1 MiB: the record is carried over many chunk borders and
    outgrows the 4 MiB in front of a chunk (the batch moves to a buffer of its own); it is above the kernels' sanity bound, so it is
    counted as too long, and every other record comes out as in the file without it"""
    from ciri_long_amd import hip, synth
    rng = np.random.Generator(np.random.PCG64(29))
    recs = []
    for k in range(120):
        tm = synth.template(rng)
        recs.append(('p%03d' % k, oracle_lib.decode(synth.rolling_circle_read(rng, tm, int(rng.integers(400, 1200))))))
    big = ('ACGT' * (17 * 262144 + 11))
    plain = tmp_path / 'plain.fa'; withbig = tmp_path / 'big.fa'
    plain.write_text(''.join('>%s\n%s\n' % hs for hs in recs))
    withbig.write_text(''.join('>%s\n%s\n' % hs for hs in recs[:60]) + '>huge\n' + big + '\n' + ''.join('>%s\n%s\n' % hs for hs in recs[60:]))
    monkeypatch.setenv('CLH_FILE_CHUNK_MB', '1')
    ctx = hip.default_context()
    t0, r0, l0 = ctx.ccs_file(str(plain), 0, str(tmp_path / 'a.ccs.fa'), str(tmp_path / 'a.raw.fa'), 64)
    t1, r1, l1 = ctx.ccs_file(str(withbig), 0, str(tmp_path / 'b.ccs.fa'), str(tmp_path / 'b.raw.fa'), 64)
    assert (t0, l0) == (120, 0) and (t1, r1, l1) == (121, r0, 1) and r0 > 60
    for name in ('ccs.fa', 'raw.fa'):
        assert (tmp_path / ('a.' + name)).read_bytes() == (tmp_path / ('b.' + name)).read_bytes(), name
    ctx.release_file_buffers()
    t2, r2, l2 = ctx.ccs_file(str(plain), 0, str(tmp_path / 'c.ccs.fa'), str(tmp_path / 'c.raw.fa'), 64)       # buffers made anew
    assert (t2, r2, l2) == (t0, r0, l0) and (tmp_path / 'c.ccs.fa').read_bytes() == (tmp_path / 'a.ccs.fa').read_bytes()


def test_reads_lost_to_a_kernel_limit_are_counted_and_reported(tmp_path, caplog, monkeypatch):
    """A limit of the kernel never passes for "no repeat".  (1) spoa.poa: sequences that each skip a different number of letters in
    front of the same node give that node one in-edge each (the letters of the stretch are all different, so no deletion can slide):
    14 and 31 in-edges -- more than the 12 a node holds in place -- go through the graph's overflow table and equal the oracle; 51 are
    more than the kernel keeps (48): status 2, raised; so are 9 different letters in one column.  (2) find_consensus: a read the
    kernel's workspace cannot take (budget squeezed, no large slot: status 1) is counted in the rows, in the plan's statistics, in
    pyccs' counter and log line and in the file stage's counter.  (A copy above 2800 bases was such a loss until round 4; it now runs
    the wide form of the pass and equals the oracle.)  The oracle has none of these limits."""
    import logging
    import random
    from ciri_long_amd import find_ccs, hip, pyccs, spoa
    rng = random.Random(3)
    unit = [rng.choice('ACGT') for _ in range(260)]
    unit[95:155] = [chr(0x80 + k) for k in range(60)]      # 60 letters that occur nowhere else (raw bytes: the alphabet is whatever comes in)
    unit = ''.join(unit)
    copies = [unit] + [unit[:154 - k] + unit[154:] for k in range(1, 56)]
    args = (10, -4, -8, -2, -24, -1)
    for n_seq in (11, 13, 30):                             # 12, 14, 31 in-edges at the node behind the stretch
        for mode in (0, 1, 2):
            got = spoa.poa(copies[:n_seq], mode, True, *args)
            want = oracle_lib.oracle_poa(copies[:n_seq], mode, True, *args)
            assert got[0] == want[0] and list(got[1]) == list(want[1]), (n_seq, mode)
    assert oracle_lib.oracle_poa(copies[:50], 0, False, *args) is not None
    with pytest.raises(hip.ClhError, match='status 2'):
        spoa.poa(copies[:50], 0, False, *args)                # 51 in-edges
    nine = [unit[:40] + chr(0x61 + k) + unit[41:] for k in range(9)]
    assert oracle_lib.oracle_poa(nine, 0, False, *args) is not None
    with pytest.raises(hip.ClhError, match='status 2'):
        spoa.poa(nine, 0, False, *args)                       # 9 letters in one column
    assert spoa.poa(nine[:8], 0, False, *args)[0] == oracle_lib.oracle_poa(nine[:8], 0, False, *args)
    long_unit = ''.join(rng.choice('ACGT') for _ in range(3000))
    read = long_unit * 3
    short = ''.join(rng.choice('ACGT') for _ in range(220)) * 4
    ctx = hip.default_context()
    data, off = hip.pack([read, short])
    plan = ctx.ccs_plan(off)
    import torch
    d = torch.from_numpy(data.view(np.uint8)).cuda()
    plan.run(d.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rows, _segs, _ccs = plan.fetch()
    assert [int(x) for x in rows['status']] == [0, 0]         # the 3000-base copies: the wide form of the pass
    plan.close()
    want = oracle_lib.oracle_find_consensus(read)
    assert pyccs.find_consensus_batch([read, short])[0] == want[:2] and want[0] is not None
    # a read the workspace cannot take: the budget squeezed to 1 MiB per ... and a single large slot too small for it -> status 1
    monkeypatch.setenv('CLH_POA_BUDGET_MB', '1'); monkeypatch.setenv('CLH_POA_BIG_BYTES', '4096')
    plan = ctx.ccs_plan(off)
    plan.run(d.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rows, _segs, _ccs = plan.fetch()
    st = plan.stats()
    plan.close()
    lost = int((rows['status'] == 1).sum())
    assert lost >= 1 and st['dropped'].get(1, 0) == lost and st['dp_row_steps'] >= 0
    before = dict(pyccs.capacity_dropped)
    with caplog.at_level(logging.WARNING, logger='CIRI-long'):
        got = pyccs.find_consensus_batch([read, short])
    assert got[0] == (None, None)
    assert pyccs.capacity_dropped.get(1, 0) == before.get(1, 0) + lost
    assert any('got no consensus' in r.getMessage() for r in caplog.records)
    (tmp_path / 'tmp').mkdir()
    fa = tmp_path / 'in.fa'
    fa.write_text('>lost\n%s\n>kept\n%s\n' % (read, short))
    total, ro, ccs_seq = find_ccs.find_ccs_reads(str(fa), str(tmp_path), 'p', 1, False)
    assert total == 2 and 'lost' not in ccs_seq and ctx.last_capacity_dropped >= 1


def test_walks_that_leave_the_band_of_stored_cells_fall_back_to_the_full_planes():
    """The forward pass leaves its cells only near the straight line through the matrix (POA_BAND); a sequence that joins the
    graph far from that line -- here: sequences that start in the middle of the first one -- makes the back-track ask for a cell
    that is not there: the pass runs again with every cell stored and the answer is the oracle's, in all three modes.  The misses
    are counted."""
    import random
    from ciri_long_amd import hip, spoa
    rng = random.Random(77)
    ctx = hip.default_context()
    total = 0
    for it in range(12):
        x = ''.join(rng.choice('ACGT') for _ in range(rng.choice([420, 700, 1100])))
        seqs = [x]
        for k in range(4):
            a = rng.randrange(120, len(x) // 2)
            y = x[a:] + ''.join(rng.choice('ACGT') for _ in range(rng.choice([0, 150, 260])))
            seqs.append(''.join(rng.choice('ACGT') if rng.random() < 0.05 else ch for ch in y))
        seqs.append(x[:len(x) // 2])
        algorithm = it % 3
        want = oracle_lib.oracle_poa(seqs, algorithm, True, 10, -4, -8, -2, -24, -1)
        assert spoa.poa(seqs, algorithm, True, 10, -4, -8, -2, -24, -1) == tuple(want), it
        total += ctx.poa_last_stats()['band_misses']
    assert total >= 12


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_the_wide_form_of_the_pass_equals_the_oracle(mode, monkeypatch):
    """The 32-bit form of K3's pass (sequences above 2800 bases, scores outside the 16-bit cells; csrc/ccs_poa.hip: dp_pass_w): forced
    for every sequence (CLH_POA_FORCE_WIDE) it must give what the packed form gives on the shapes that reach every path of the pass --
    one to six columns per lane, several column passes with carries, far source rows, many in-edges, raw letters -- and on sequences
    of 3000-5200 bases nothing but it can take; consensus, MSA rows and end-cell scores against oracle/poa_oracle.c."""
    import random
    from ciri_long_amd import hip, spoa
    rng = random.Random(500 + mode)
    monkeypatch.setenv('CLH_POA_FORCE_WIDE', '1')
    ctx = hip.default_context()
    for it in range(14):
        L = [25, 70, 130, 200, 330, 390, 500, 800, 1300][it % 9]
        fam = _family(rng, L, rng.randint(2, 9), rng.choice([0.03, 0.12, 0.25]))
        if it % 4 == 3:
            fam.append(fam[0][len(fam[0]) // 3:] + fam[1][:len(fam[1]) // 2])      # joins far from the diagonal: far source rows
        for pars in (PARS[0], PARS[it % len(PARS)]):
            want = oracle_lib.oracle_poa(fam, mode, True, *pars)
            assert spoa.poa(fam, mode, True, *pars) == tuple(want), (it, L, pars)
    monkeypatch.delenv('CLH_POA_FORCE_WIDE')
    for L in (3000, 5200):
        base = ''.join(rng.choice('ACGT') for _ in range(L))
        fam = [base] + [''.join(rng.choice('ACGT') if rng.random() < 0.04 else ch for ch in base if rng.random() > 0.03) for _ in range(2)]
        want = oracle_lib.oracle_poa(fam, mode, True, 10, -4, -8, -2, -24, -1)
        assert spoa.poa(fam, mode, True, 10, -4, -8, -2, -24, -1) == tuple(want), L
