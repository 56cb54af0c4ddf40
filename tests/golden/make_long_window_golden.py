#!/usr/bin/env python3
"""Golden vectors for the call path's own shape -- a clip against a window of 33..420 kb (find_bsj.py:196-216) -- from the REFERENCE
ITSELF: the reference's ssw_wrap.py (run from where it lies under /root/reference) over oracle/_ref/libssw.so (the reference's ssw.c).

A window of 400 kb does not belong in a fixture verbatim: the fixture holds its RECIPE -- the seed and length of a numpy PCG64 stream
of uniform bases and the stretches written over it (the planted loci, as strings) -- next to the clip and the reference's answer;
`build_window` below is what the test calls to rebuild the window.  Output: tests/golden/long_window_golden.json.gz

    python tests/golden/make_long_window_golden.py
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference_wrapper, mutate, rnd  # noqa: E402

B = np.frombuffer(b'ACGT', dtype=np.uint8)


def build_window(seed, length, patches):
    """the window of a case: `length` uniform bases of PCG64(seed), then every (position, string) of `patches` written over it"""
    w = B[np.random.Generator(np.random.PCG64(seed)).integers(0, 4, length)].copy()
    for pos, s in patches:
        w[pos:pos + len(s)] = np.frombuffer(s.encode(), dtype=np.uint8)
    return w.tobytes().decode()


def main():
    ns = load_reference_wrapper()
    rng = np.random.default_rng(20210847)
    cases = []
    schemes = [(1, 1, 1, 1)] * 3 + [(10, 4, 8, 2), (2, 2, 3, 1)]
    for k in range(44):
        m, x, o, e = schemes[k % len(schemes)]
        R = int(rng.choice([33000, 70000, 150000, 402000]))
        lo, hi = {1: (20, 600), 10: (20, 300), 2: (20, 300)}[m]
        L = int(rng.integers(lo, hi)) if k % 7 else int(rng.choice([253, 254, 255, 256]))
        seed = 7000 + k
        w0 = build_window(seed, R, [])
        pos = int(rng.integers(0, R - L))
        if k % 9 == 1:
            pos = 256 * int(rng.integers(1, R // 256 - 2)) - L // 2     # across a block border of the prefilter
        err = [0.0, 0.02, 0.13, 0.05][k % 4]
        clip = mutate(w0[pos:pos + L], rng, sub=err * 0.3, ins=err * 0.3, dele=err * 0.4) or 'A'
        patches = []
        if k % 5 == 2 and pos > 3 * L + 2000:                            # a second, exact copy earlier: the first end column wins
            patches.append((pos - 2 * L - 1500, w0[pos:pos + L]))
        if k % 5 == 3 and pos > 50000:                                   # a noisier copy far away
            patches.append((pos - 40000, mutate(w0[pos:pos + L], rng, 0.05, 0.05, 0.05)[:L]))
        if k % 11 == 5:
            clip = rnd(rng, L)                                           # no locus
        if k % 13 == 6:
            patches.append((pos + L // 2, 'N' * 11))
        window = build_window(seed, R, patches)
        al = ns['Aligner'](window, match=m, mismatch=x, gap_open=o, gap_extend=e, report_secondary=False, report_cigar=True)
        res = al.align(clip)
        cases.append(dict(name='long_%02d' % k, seed=seed, length=R, patches=patches, query=clip, match=m, mismatch=x, gap_open=o, gap_extend=e,
                          score=res.score, ref_begin=res.ref_begin, ref_end=res.ref_end, query_begin=res.query_begin, query_end=res.query_end,
                          cigar_string=res.cigar_string))
        print(cases[-1]['name'], R, L, res.score, res.ref_begin, res.ref_end, flush=True)
    out = os.path.join(HERE, 'long_window_golden.json.gz')
    with gzip.open(out, 'wt') as f:
        json.dump({'generator': 'tests/golden/make_long_window_golden.py', 'source': "the reference's ssw_wrap.py over oracle/_ref/libssw.so",
                   'cases': cases}, f)
    print('wrote', out, len(cases), 'cases', os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
