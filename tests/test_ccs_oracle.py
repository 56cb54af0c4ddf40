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


# The labels the reference's tests/test_poa.py:21-28 holds beside its six copies.  Nothing in the reference asserts them, and their sum does
# not even match the strings they label (the last one spans 21 bases for a 38-base string) -- but they look like a recorded pyccs `segments`
# string, which makes them the only evidence there is about pyccs' boundary rule.
REFERENCE_LABELS = '0-145;145-289;289-433;433-579;579-721;721-742'


def test_copy_boundaries_beside_the_labels_the_reference_test_holds(capsys):
    """NON-BINDING (parity unpinned): prints how clh-ccs v1 cuts the tests/test_poa.py input next to the labels above, and pins only what
    this repository's specification says today -- so that the day pyccs can be run, the first thing to compare is already written down, and
    so that a change of the specification shows up here.  Today: the same number of copies, the first three copies' boundaries within two
    bases of the labels and of the true copy boundaries of the input (cumulative lengths of the six strings); the cuts behind the two copies
    with long deletions up to ten bases in front of the true ends."""
    raw = ''.join(SEGMENTS)
    seg, ccs, period = oracle_lib.oracle_find_consensus(raw)
    ours = [tuple(map(int, x.split('-'))) for x in seg.split(';')]
    labels = [tuple(map(int, x.split('-'))) for x in REFERENCE_LABELS.split(';')]
    true_ends = list(np.cumsum([len(x) for x in SEGMENTS]))
    with capsys.disabled():
        print('\n  tests/test_poa.py labels : %s\n  clh-ccs v1 (oracle)     : %s\n  true copy ends          : %s' % (REFERENCE_LABELS, seg, ';'.join(map(str, true_ends))))
    assert seg == '0-143;143-289;289-431;431-569;569-704;704-751'          # the specification as it stands (oracle/ccs_oracle.c, "clh-ccs v1")
    assert len(ours) == len(labels) == len(SEGMENTS)
    for (a, b), (la, lb) in list(zip(ours, labels))[:3]:
        assert abs(a - la) <= 2 and abs(b - lb) <= 2
    for (a, b), e in zip(ours[:3], true_ends[:3]):
        assert abs(b - e) <= 2
    for (a, b), e in zip(ours[3:5], true_ends[3:5]):      # copies 4 and 5 carry deletions of 9 and 8 bases: the chained cuts sit that far in front of the true ends
        assert e - 10 <= b <= e


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
