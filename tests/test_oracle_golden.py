"""The CPU restatement (oracle/ssw_oracle.c) against golden vectors captured from the reference's own
libssw.so + ssw_wrap.py (tests/golden/make_golden.py).  This is what pins the oracle."""
import pytest

from oracle_lib import oracle_align

KEYS = ('score', 'ref_begin', 'ref_end', 'query_begin', 'query_end', 'cigar_string')


def _check(c):
    got = oracle_align(c['ref'], c['query'], c['match'], c['mismatch'], c['gap_open'], c['gap_extend'])
    assert got is not None, c['name']
    for k in KEYS:
        assert got[k] == c[k], (c['name'], k, got[k], c[k])
    assert got['score2'] == c['raw_score2'], (c['name'], 'score2')
    assert got['ref_end2'] == c['raw_ref_end2'], (c['name'], 'ref_end2')
    assert len(got['cigar']) == c['raw_cigar_len']


def test_oracle_matches_all_golden_vectors(golden_cases):
    assert len(golden_cases) >= 1400
    for c in golden_cases:
        _check(c)


def test_reference_test_fa_known_answers(golden_cases):
    """tests/test.fa in the orientation of tests/test_ssw.py and of find_bsj.py:196-205."""
    by = {c['name']: c for c in golden_cases}
    a = by['testfa_ref_seq1_query_seq2']
    assert (a['score'], a['ref_begin'], a['ref_end'], a['query_begin'], a['query_end']) == (349, 20, 436, 229781, 230207)
    b = by['testfa_ref_seq2_query_seq1']
    assert (b['score'], b['ref_begin'], b['ref_end'], b['query_begin'], b['query_end']) == (349, 229790, 230207, 20, 436)
    assert (b['raw_score2'], b['raw_ref_end2']) == (140, 230425)


def test_zero_score_is_degenerate_but_deterministic(golden_cases):
    z = [c for c in golden_cases if c['name'] == 'zero_score'][0]
    assert (z['score'], z['ref_begin'], z['ref_end'], z['query_begin'], z['query_end'], z['cigar_string']) == \
        (0, -1, -1, 0, 0, '1M7S')
    _check(z)


def test_stale_walk_goldens_oracle_equals_the_reference_library():
    """tests/golden/stale_walk_golden.json.gz: alignments whose traceback leaves the final band and reads direction bytes
    of other cells / earlier band iterations (ssw.c:636-696).  Expected values are the reference library's; the oracle
    reproduces them, and its instrumentation confirms the walks do take such steps."""
    import ctypes as C
    import gzip
    import json
    import os
    import oracle_lib
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stale_walk_golden.json.gz'), 'rt') as f:
        cases = json.load(f)['cases']
    assert len(cases) >= 4
    lib = oracle_lib.oracle(); lib.clo_last_oob_steps.restype = C.c_int
    for c in cases:
        w = oracle_lib.oracle_align(c['ref'], c['query'], c['match'], c['mismatch'], c['gap_open'], c['gap_extend'])
        assert lib.clo_last_oob_steps() == c['stale_steps'] > 0
        for k, v in c['want'].items():
            assert w[k] == v, (c['rank'], c['index'], k)


def test_oracle_matches_the_long_window_goldens():
    """tests/golden/long_window_golden.json.gz (made by the reference's own wrapper over its own library, make_long_window_golden.py):
    clips of 20..600 bases against windows of 33..402 kb -- the shape of find_bsj.py:196-216 -- in three scoring schemes; the oracle
    must give the reference's scores, coordinates and CIGARs (the GPU test runs the same cases through both long-window classes)"""
    import gzip
    import json
    import os
    import sys
    from oracle_lib import cigar_to_string, oracle_align
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    from make_long_window_golden import build_window
    with gzip.open(os.path.join(gdir, 'long_window_golden.json.gz'), 'rt') as f:
        cases = json.load(f)['cases']
    assert len(cases) == 44
    for c in cases:
        ref = build_window(c['seed'], c['length'], c['patches'])
        w = oracle_align(ref, c['query'], c['match'], c['mismatch'], c['gap_open'], c['gap_extend'])
        assert (w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']) == \
            (c['score'], c['ref_begin'], c['ref_end'], c['query_begin'], c['query_end']), c['name']
        assert cigar_to_string(w['cigar'], w['query_begin'], w['query_end'], len(c['query'])) == c['cigar_string'], c['name']
