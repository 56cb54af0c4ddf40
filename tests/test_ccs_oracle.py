"""CPU checks of the consensus specification (oracle/ccs_oracle.c + oracle/poa_oracle.c).  There is no pyccs/spoa to
compare with (PARITY UNPINNED): what can be checked is the contract CIRI-long consumes, the reference's own structural
test, and that the method does what it is for."""
import re

import numpy as np

import oracle_lib
from test_gpu_ccs import SEGMENTS


def test_reference_test_poa_structure():
    """tests/test_poa.py:8-32 of the reference: 6 copies of ~144 nt; its only assertion is
    len(find_consensus(raw).ccs) == len(poa(copies).consensus).  Here both come from the same engine."""
    raw = ''.join(SEGMENTS)
    seg, ccs, period = oracle_lib.oracle_find_consensus(raw)
    assert re.fullmatch(r'\d+-\d+(;\d+-\d+)*', seg)
    bounds = [tuple(map(int, s.split('-'))) for s in seg.split(';')]
    assert bounds[0][0] == 0 and bounds[-1][1] == len(raw)
    assert all(a < b for a, b in bounds) and all(bounds[i][1] == bounds[i + 1][0] for i in range(len(bounds) - 1))
    assert len(bounds) >= 5 and 140 <= period <= 148
    for a, b in bounds[:5]:
        assert 130 <= b - a <= 150
    assert set(ccs) <= set('ACGT')
    # the reference's assertion (tests/test_poa.py:30-32): same length as poa(copies, 0, True, 10, -4, -8, -2, -24, -1)
    cons, msa = oracle_lib.oracle_poa(SEGMENTS, 0, True, 10, -4, -8, -2, -24, -1)
    assert len(ccs) == len(cons) == 144
    assert len(msa) == len(SEGMENTS) and all(r.replace('-', '') == s for r, s in zip(msa, SEGMENTS))


# The six strings of the reference's tests/test_poa.py:8-15 are a recorded pyccs segmentation of one read (the `fasta` tuple :21-28 repeats
# them under the labels below; the labels' own arithmetic does not fit the strings -- the last one spans 21 bases for a 38-base string -- so
# the STRINGS are the evidence, not the numbers).  Nothing else the reference holds says where pyccs cuts.  On the concatenation of the six,
# a faithful boundary rule gives the six back: cuts at their cumulative lengths.
REFERENCE_LABELS = '0-145;145-289;289-433;433-579;579-721;721-742'
REFERENCE_CUTS = '0-145;145-289;289-433;433-577;577-713;713-751'


def test_copy_boundaries_reproduce_the_segmentation_the_reference_test_holds():
    """clh-ccs v2 (oracle/ccs_oracle.c step 2): every copy of the tests/test_poa.py input is cut where pyccs cut it -- where the copy's
    opening recurs (`TCCCGGTC...` at 0, 145, 289, 433, 577; the last copy has lost a base of its opening, and the k-mer across the boundary,
    `AATAGTCC`, places it at 713).  Still parity unpinned (one read is not a pin), but the only reference-held data about pyccs' boundaries is
    now reproduced exactly; v1 (rounds 1-5: the most frequent offset of the next 96 bases) gave 0-143;143-289;289-431;431-569;569-704;704-751."""
    raw = ''.join(SEGMENTS)
    seg, ccs, period = oracle_lib.oracle_find_consensus(raw)
    true_ends = list(np.cumsum([len(x) for x in SEGMENTS]))
    assert [int(x.split('-')[1]) for x in REFERENCE_CUTS.split(';')] == true_ends == [145, 289, 433, 577, 713, 751]
    assert seg == REFERENCE_CUTS
    copies = [raw[int(a):int(b)] for a, b in (x.split('-') for x in seg.split(';'))]
    assert copies == list(SEGMENTS)
    # the reference's own assertion on those copies (tests/test_poa.py:30-32)
    assert len(ccs) == len(oracle_lib.oracle_poa(copies, 0, True, 10, -4, -8, -2, -24, -1)[0])
    # the first three labels agree with the strings; the last three are the ones whose arithmetic is off
    for ours, lab in list(zip(seg.split(';'), REFERENCE_LABELS.split(';')))[:3]:
        assert ours == lab


def test_copy_boundaries_land_near_the_true_copy_starts():
    """the generator of the synthetic reads knows where every base came from: the cuts of v2 sit within two bases of the position homologous
    to the read's first base for three cuts in four (v1: 56 %; tools/dev/ccs_cut_eval.py prints the whole comparison)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools', 'dev'))
    import ccs_cut_eval as ev
    from ciri_long_amd import synth
    rng = np.random.Generator(np.random.PCG64(6))
    err = []
    for _ in range(150):
        tm = synth.template(rng)
        p = len(tm)
        Lt = int(max(300, round(rng.normal(1000, 100))))
        phase = int(rng.integers(0, p))
        read, org = ev.mutate_with_origin(np.tile(tm, Lt // p + 2)[phase:phase + Lt], rng)
        p0, cuts = ev.oracle_segments(read)
        if not p0 or abs(p0 - p) > max(4, p // 8):
            continue
        assert cuts == ev.cuts_v2(read, p0)[:len(cuts)]          # the Python statement of the rule and the C one agree
        for q, c in enumerate(cuts):
            j = int(np.searchsorted(org, (q + 1) * p, side='left'))
            if j < len(read):
                err.append(abs(c - j))
    err = np.array(err)
    assert len(err) > 300 and err.mean() < 2.2 and (err <= 2).mean() > 0.70


def test_recovers_templates_and_rejects_linear_reads():
    from ciri_long_amd import synth
    rng = np.random.Generator(np.random.PCG64(11))
    found = tot = rejected = neg = 0
    idents = []
    for it in range(150):
        tm = synth.template(rng)
        L = int(rng.normal(1000, 100))
        if it % 3 == 2:
            seg, ccs, _ = oracle_lib.oracle_find_consensus(synth.mutate(rng.integers(0, 4, L, dtype=np.int8), rng))
            neg += 1
            rejected += seg is None
            continue
        read = synth.rolling_circle_read(rng, tm, L)
        if len(read) < 2.2 * len(tm):
            continue
        tot += 1
        seg, ccs, period = oracle_lib.oracle_find_consensus(read)
        if seg is None:
            continue
        found += 1
        a = oracle_lib.oracle_align(oracle_lib.decode(np.tile(tm, 2)), ccs, 1, 1, 1, 1)
        idents.append(a['score'] / max(len(ccs), len(tm)))
    assert rejected == neg                      # no false repeat in random sequence
    assert found >= 0.97 * tot
    assert np.median(idents) > 0.90             # raw reads carry 13 % errors; 3-5 copies bring the consensus above 90 %


def test_degenerate_inputs():
    assert oracle_lib.oracle_find_consensus('ACGT' * 5)[0] is None              # shorter than two minimal periods
    seg, ccs, period = oracle_lib.oracle_find_consensus('ACGT' * 200)          # microsatellite: reported at a multiple >= 30
    assert seg is not None and period >= 30 and all(int(x.split('-')[1]) % 4 == 0 for x in seg.split(';'))
    assert oracle_lib.oracle_find_consensus('N' * 500)[0] is None
    assert oracle_lib.oracle_poa(['ACGTACGT']) == 'ACGTACGT'
