#!/usr/bin/env python3
"""Golden vectors for the third stage of `call` (scan_raw_chunk, CIRI_long/find_bsj.py:499-620) from the REFERENCE's own
Python, with the deterministic mapper/genome doubles of tests/fake_mapper.py (same arrangement as make_bsj_golden.py).
The reads are single-pass reads over a junction (1.2-1.8 copies of a planted circRNA, noisy), linear reads and short reads.

    PYTHONHASHSEED=0 python tests/golden/make_raw_golden.py      -> tests/golden/raw_golden.json.gz"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_bsj_golden as mb      # noqa: E402
import fake_mapper as fm          # noqa: E402
import oracle_lib                 # noqa: E402
from ciri_long_amd import synth   # noqa: E402


def raw_reads(world, n=240, seed=20210848):
    rng = np.random.Generator(np.random.PCG64(seed))
    genome = world['genome']
    B = 'ACGT'
    reads = []
    for k in range(n):
        u = rng.random()
        if u < 0.7:
            big = [c for c in world['circs'] if sum(b - a for a, b in c[1]) >= 260]
            ctg, exons, strand = big[int(rng.integers(len(big)))]
            circ = ''.join(genome.genome[ctg][a:b] for a, b in exons)
            if rng.random() < 0.5:
                circ = fm.rc(circ)
            copies = 1.05 + 0.9 * rng.random()
            phase = int(rng.integers(0, len(circ)))
            raw = (circ * 3)[phase:phase + int(copies * len(circ))]
            codes = synth.mutate(oracle_lib.encode(raw), rng, 0.02, 0.02, 0.02)
            seq = ''.join(B[c] if c < 4 else 'N' for c in codes)
        elif u < 0.85:
            ctg = 'chrA'
            st = int(rng.integers(0, 9000)); seq = genome.genome[ctg][st:st + int(rng.integers(350, 900))]
        else:
            seq = ''.join(B[i] for i in rng.integers(0, 4, int(rng.integers(80, 290))))
        reads.append(('raw%03d' % k, seq))
    return reads


def main():
    align, env, find_bsj = mb.load_reference()
    mb.watch_ties(align)
    world = fm.build_world()
    mapper = fm.FakeMapper(world['genome'])
    env.ALIGNER = mapper; env.GENOME = world['genome']; env.CONTIG_LEN = world['genome'].contig_len
    env.SS_INDEX = world['ss_index']; env.GTF_INDEX = world['gtf_index']; env.INTRON_INDEX = None
    reads = raw_reads(world)
    skip = {reads[3][0]: 1}
    n0 = len(mb.TIES)
    cnt, ret, short = find_bsj.scan_raw_chunk(reads, True, skip)
    out = dict(reads=[list(r) for r in reads], skip=list(skip), counters=dict(cnt), records=[list(r) for r in ret],
               short=[list(s) for s in short], any_tie=bool(any(mb.TIES[n0:])))
    path = os.path.join(HERE, 'raw_golden.json.gz')
    with gzip.open(path, 'wt') as f:
        json.dump(out, f)
    print('wrote', path, dict(cnt), len(ret), 'records', len(short), 'short; ties:', out['any_tie'])


if __name__ == '__main__':
    main()
