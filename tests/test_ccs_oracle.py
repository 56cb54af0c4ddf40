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
