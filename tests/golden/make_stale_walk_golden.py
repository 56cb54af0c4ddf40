"""Golden cases for the traceback's reads of stale direction bytes (banded_sw, ssw.c:636-696: a walk that leaves the final
band reads whatever byte sits at that flat index -- codes of another cell, possibly of an earlier band iteration).  They are
rare (about 1 in 20 000 C2-shaped alignments), so they are searched for once: the instrumented oracle (clo_last_oob_steps)
finds alignments whose walk takes such steps among the C2-shaped batches of bench.py; the expected rows and CIGARs come from the
reference's own libssw.so (oracle/_ref, built from /root/reference by oracle/Makefile).

    python tests/golden/make_stale_walk_golden.py [--all]   (needs /root/reference; --all rescans 65 batches, ~5 minutes on 8 cores)
"""
import ctypes as C
import gzip
import json
import os
import sys
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
B = 'ACGTN'


def search(rank):
    import bench
    import oracle_lib
    from ciri_long_amd import synth
    lib = oracle_lib.oracle(); lib.clo_last_oob_steps.restype = C.c_int
    reads, wins = bench.make_batch(synth, 'c2', 2500, rank)
    out = []
    for k in range(len(reads)):
        w = oracle_lib.oracle_align(wins[k], reads[k], 1, 1, 1, 1)
        n = lib.clo_last_oob_steps()
        if n > 0 and w is not None:
            r = oracle_lib.ref_align(wins[k], reads[k], 1, 1, 1, 1)
            assert r is not None and all(r[x] == w[x] for x in ('score', 'score2', 'ref_begin', 'ref_end', 'query_begin', 'query_end', 'ref_end2', 'cigar'))
            out.append(dict(rank=rank, index=k, stale_steps=int(n), ref=''.join(B[int(c)] for c in wins[k]), query=''.join(B[int(c)] for c in reads[k]),
                            match=1, mismatch=1, gap_open=1, gap_extend=1,
                            want={x: r[x] for x in ('score', 'score2', 'ref_begin', 'ref_end', 'query_begin', 'query_end', 'ref_end2', 'cigar')}))
    return out


if __name__ == '__main__':
    # ranks 0..64 (162 500 alignments) were scanned once: only these four batches hold such an alignment
    ranks = [0, 9, 12, 20] if '--all' not in sys.argv else list(range(0, 65))
    with Pool(8) as p:
        res = p.map(search, ranks)
    cases = [c for r in res for c in r]
    path = os.path.join(HERE, 'stale_walk_golden.json.gz')
    keep = {}
    if os.path.exists(path):       # `wide_reference_cases` were added by hand from a fuzz finding (see its note): keep them
        with gzip.open(path, 'rt') as f:
            keep = {k: v for k, v in json.load(f).items() if k not in ('note', 'cases')}
    with gzip.open(path, 'wt') as f:
        json.dump(dict(keep, note='alignments whose traceback reads direction bytes outside the final band; expected values from the reference libssw.so',
                       cases=cases), f)
    print(len(cases), 'cases', [(c['rank'], c['index'], c['stale_steps']) for c in cases])
