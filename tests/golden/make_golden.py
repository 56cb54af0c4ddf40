#!/usr/bin/env python3
"""Generate golden vectors for the SSW path from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and oracle/_ref/libssw.so, built by
`make -C oracle ref` from the reference's own ssw.c).  The reference's ctypes wrapper
(libs/striped_smith_waterman/ssw_wrap.py) is executed from where it lies -- it is not copied -- with
its library path pointed at oracle/_ref/libssw.so.  Output: tests/golden/ssw_golden.json.gz
(inputs + the eight PyAlignRes outputs + raw score2/ref_end2) and tests/golden/test.fa
(the data file of the reference's own tests/test_ssw.py).

    python tests/golden/make_golden.py
"""
import gzip
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('CIRI_REFERENCE', '/root/reference')
REF_WRAP = os.path.join(REF, 'libs', 'striped_smith_waterman', 'ssw_wrap.py')
REF_SO_DIR = os.path.join(ROOT, 'oracle', '_ref')


def load_reference_wrapper():
    """exec the reference module with __file__ placed beside oracle/_ref/libssw.so (ssw_wrap.py:17)."""
    ns = {'__name__': 'ref_ssw_wrap', '__file__': os.path.join(REF_SO_DIR, 'ssw_wrap.py')}
    with open(REF_WRAP) as f:
        code = compile(f.read(), REF_WRAP, 'exec')
    exec(code, ns)
    return ns


def mutate(s, rng, sub=0.04, ins=0.04, dele=0.05):
    out = []
    for c in s:
        u = rng.random()
        if u < dele:
            continue
        if u < dele + sub:
            out.append('ACGT'[rng.integers(4)])
            continue
        out.append(c)
        if rng.random() < ins:
            out.append('ACGT'[rng.integers(4)])
    return ''.join(out)


def rnd(rng, n):
    return ''.join('ACGT'[i] for i in rng.integers(0, 4, n))


def run_case(ns, ref, query, scheme, name):
    m, x, o, e = scheme
    al = ns['Aligner'](ref, match=m, mismatch=x, gap_open=o, gap_extend=e, report_secondary=True, report_cigar=True)
    # raw struct fields too: PyAlignRes hides score2 == 0 (ssw_wrap.py:332-338)
    raw = {}
    orig = ns['PyAlignRes'].__init__

    def spy(self, Res, query_len, report_secondary=False, report_cigar=False):
        raw['score2'] = int(Res.contents.score2)
        raw['ref_end2'] = int(Res.contents.ref_end2)
        raw['cigar_len'] = int(Res.contents.cigarLen)
        orig(self, Res, query_len, report_secondary, report_cigar)

    ns['PyAlignRes'].__init__ = spy
    try:
        res = al.align(query)
    finally:
        ns['PyAlignRes'].__init__ = orig
    return dict(name=name, ref=ref, query=query, match=m, mismatch=x, gap_open=o, gap_extend=e,
                score=res.score, ref_begin=res.ref_begin, ref_end=res.ref_end, query_begin=res.query_begin,
                query_end=res.query_end, cigar_string=res.cigar_string, score2=res.score2, ref_end2=res.ref_end2,
                raw_score2=raw['score2'], raw_ref_end2=raw['ref_end2'], raw_cigar_len=raw['cigar_len'])


def main():
    ns = load_reference_wrapper()
    rng = np.random.default_rng(20210841)
    schemes = [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1)]
    cases = []

    # hand-written vectors
    for sch in schemes[:2]:
        cases.append(run_case(ns, 'ACGTACGTTTGACCA', 'CGTACGTTGACC', sch, 'tiny_gap'))
        cases.append(run_case(ns, 'ACGT' * 100, 'ACGT' * 70, sch, 'periodic_word'))
    cases.append(run_case(ns, 'AAAAAAAAAA', 'CCCCCCCC', schemes[0], 'zero_score'))
    cases.append(run_case(ns, 'ACGTNNNNACGT', 'ACGTACGTACGT', schemes[0], 'n_in_ref'))
    cases.append(run_case(ns, 'ACGTACGTACGT', 'ACGTNNNNACGT', schemes[0], 'n_in_query'))
    cases.append(run_case(ns, 'acgtacgtttgacca', 'CGTACGTTGACC', schemes[0], 'lower_case'))
    cases.append(run_case(ns, 'ACGTRYKMACGTTTGACCA', 'CGTRYKMACGTTGACC', schemes[0], 'iupac_as_n'))
    base = rnd(rng, 600)
    for L in range(250, 259):  # the 8-bit -> 16-bit switch sits between 253 and 254 at 1/1/1/1
        cases.append(run_case(ns, base, base[100:100 + L], schemes[0], 'byte_word_switch_%d' % L))

    # randomized classes
    classes = [(20, 30, 200, 150), (31, 253, 600, 200), (254, 600, 1500, 100), (900, 1100, 2000, 12)]
    for sch in schemes:
        for lo, hi, reflen, count in classes:
            for k in range(count):
                L = int(rng.integers(lo, hi + 1))
                ref = rnd(rng, reflen)
                st = int(rng.integers(0, max(1, reflen - L)))
                kind = rng.random()
                if kind < 0.70:
                    q = mutate(ref[st:st + L], rng)
                elif kind < 0.80:
                    q = mutate(ref[st:st + L // 2], rng) * 2          # tie-heavy: duplicated query
                elif kind < 0.90:
                    q = rnd(rng, L)                                   # unrelated (linear negative)
                else:
                    unit = rnd(rng, int(rng.integers(2, 9)))
                    ref = (unit * (reflen // len(unit) + 1))[:reflen]  # periodic window
                    q = mutate((unit * (L // len(unit) + 1))[:L], rng, 0.02, 0.02, 0.02)
                if rng.random() < 0.10:
                    p = int(rng.integers(0, reflen - 30))
                    ref = ref[:p] + 'N' * int(rng.integers(1, 30)) + ref[p:]
                if not q:
                    continue
                cases.append(run_case(ns, ref, q, sch, 'rand_%d_%d_%s' % (lo, hi, 'x'.join(map(str, sch)))))

    # the reference's own test data, both orientations (tests/test_ssw.py:5-15 and find_bsj.py:196-205)
    with open(os.path.join(REF, 'tests', 'test.fa')) as f:
        f.readline(); seq1 = f.readline().rstrip(); f.readline(); seq2 = f.readline().rstrip()
    big = []
    for name, r, q in (('testfa_ref_seq1_query_seq2', seq1, seq2), ('testfa_ref_seq2_query_seq1', seq2, seq1)):
        c = run_case(ns, r, q, schemes[0], name)
        c['ref'] = '@test.fa:' + ('seq1' if r is seq1 else 'seq2')
        c['query'] = '@test.fa:' + ('seq1' if q is seq1 else 'seq2')
        big.append(c)
    shutil.copyfile(os.path.join(REF, 'tests', 'test.fa'), os.path.join(HERE, 'test.fa'))

    out = dict(generator='tests/golden/make_golden.py', reference='bioinfo-biols/CIRI-long v1.1.0 libssw.so (gcc -O3)',
               cases=cases + big)
    with gzip.open(os.path.join(HERE, 'ssw_golden.json.gz'), 'wt') as f:
        json.dump(out, f)
    print('wrote', len(out['cases']), 'cases')


if __name__ == '__main__':
    sys.exit(main())
