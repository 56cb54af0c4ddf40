#!/usr/bin/env python3
"""Golden vectors for the consensus half of the path (rows a2 / a3 of SURVEY.md section 8) from the REAL packages CIRI-long calls:

    from pyccs import find_consensus        (CIRI_long/find_ccs.py:8,14; requirement pyccs >= 1.1.0, setup.py:57)
    from spoa import poa                    (CIRI_long/collapse.py:9,267,504; tests/test_poa.py:30; requirement pyspoa >= 0.0.5)

Neither package exists in the build container or on the GPU box (no source in /root/reference, no wheel, no network), which is why the
consensus kernels K2 / K3 and their CPU statements (oracle/ccs_oracle.c, oracle/poa_oracle.c) are PARITY UNPINNED.  This script is the
pin, ready for the day a machine with `pip install pyccs pyspoa` is at hand:

    python tests/golden/make_ccs_ref_golden.py            -> tests/golden/ccs_ref_golden.json.gz   (refuses without the real packages)

It needs nothing of this repository except the seeded read simulator (ciri_long_amd/synth.py: numpy only) -- no GPU, no oracle.  What it
records, and what tests/test_ccs_ref_golden.py then holds the oracle (CPU) and the kernels (`-m gpu`) to, bit for bit:

  * `test_poa`   the input of the reference's tests/test_poa.py:8-17 (the concatenation of its six strings): find_consensus' segments and
                 consensus, and poa(copies, 0, True, 10, -4, -8, -2, -24, -1) on the six strings (consensus + MSA rows);
  * `reads`      2 000 seeded reads, 1 000 of BASELINE config C3's shape (~1 kb) and 1 000 of C4's (500 nt - 4 kb), half of them
                 linear negatives as the recipe has them: find_consensus(read) -> (segments, ccs) or (None, None).  Inputs are NOT stored:
                 they are a pure function of (recipe, seed) and re-made by the test, which checks a CRC-32 of every one first;
  * `families`   500 seeded families of 2..12 noisy copies (60..700 nt) of a random template, each through poa() with algorithm 0, 1
                 and 2 at the scores of every reference call site (10, -4, -8, -2, -24, -1): consensus, CRC-32 of the MSA rows.

`--stub DIR` puts DIR in front of sys.path first: a directory holding stand-in `pyccs.py` / `spoa.py` modules.  That is how the CPU
suite dry-runs this generator and the test that reads its file (tests/test_ccs_ref_golden.py::test_generator_dry_run_with_a_stub, with
stand-ins that answer from the oracle, written to a temporary directory).  A file made that way says `"stub": true` in its header, and
the test refuses to count it as a pin; it is never written under tests/golden/.
"""
import argparse
import gzip
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ciri_long_amd import synth       # noqa: E402  (seeded read simulator; data only)

BASES = np.frombuffer(b'ACGTN', dtype=np.uint8)
POA_SCORES = (10, -4, -8, -2, -24, -1)          # collapse.py:267,504; tests/test_poa.py:30
SEED_FAMILIES = 20210846
N_C3, N_C4, N_FAMILIES = 1000, 1000, 500

# the six strings of the reference's tests/test_poa.py:8-15 (test DATA of the reference: the input of its one test of this path)
with open(os.path.join(HERE, 'test_poa_input.json')) as _f:
    TEST_POA_SEGMENTS = tuple(json.load(_f)['segments'])


def to_str(codes):
    return BASES[np.minimum(np.asarray(codes, dtype=np.int64), 4)].tobytes().decode()


def text(x):
    """pyccs / pyspoa hand back str or bytes depending on the version"""
    if x is None:
        return None
    return x.decode() if isinstance(x, (bytes, bytearray)) else str(x)


def crc(s):
    return zlib.crc32(s.encode() if isinstance(s, str) else bytes(s)) & 0xffffffff


def golden_reads():
    """[(recipe, index, read str)]: the reads of both recipes, in file order"""
    c3, _ = synth.c2_batch(N_C3, seed=synth.SEEDS['C3'])
    c4, _ = synth.c4_batch(N_C4, seed=synth.SEEDS['C4'])
    return [('C3', k, to_str(r)) for k, r in enumerate(c3)] + [('C4', k, to_str(r)) for k, r in enumerate(c4)]


def golden_families():
    """[[copy str, ...]]: noisy copies of a random template, some rotated by a few bases (what a cut a few bases off looks like), some partial"""
    rng = np.random.Generator(np.random.PCG64(SEED_FAMILIES))
    out = []
    for _ in range(N_FAMILIES):
        p = int(rng.integers(60, 701))
        tm = rng.integers(0, 4, p, dtype=np.int8)
        rate = float(rng.choice([0.01, 0.03, 0.05]))
        fam = []
        for k in range(int(rng.integers(2, 13))):
            c = tm
            if rng.random() < 0.3:
                s = int(rng.integers(0, 9))
                c = np.concatenate([tm[s:], tm[:s]])
            if rng.random() < 0.15:
                c = c[:int(rng.integers(20, p + 1))]
            fam.append(to_str(synth.mutate(c, rng, sub=rate, ins=rate, dele=rate)))
        out.append(fam)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(HERE, 'ccs_ref_golden.json.gz'))
    ap.add_argument('--stub', default=None, help='directory with stand-in pyccs.py / spoa.py (dry run; the file says so)')
    ap.add_argument('--limit', type=int, default=0, help='dry run: only the first N reads of each recipe and N // 4 families')
    a = ap.parse_args()
    if a.stub:
        sys.path.insert(0, os.path.abspath(a.stub))
        if os.path.abspath(a.out).startswith(HERE + os.sep):
            sys.exit('a stub run must not write under tests/golden/')
    try:
        import pyccs
        import spoa
    except ImportError as e:
        sys.exit('make_ccs_ref_golden.py needs the real packages (pip install pyccs pyspoa): %s' % e)
    if not a.stub and (os.path.abspath(getattr(pyccs, '__file__', '') or '').startswith(ROOT + os.sep) or os.path.abspath(getattr(spoa, '__file__', '') or '').startswith(ROOT + os.sep)):
        sys.exit("`pyccs` / `spoa` resolved to this repository's own counterparts (ciri_long_amd/pyccs.py, spoa.py): not a pin")

    def version(mod, dist):
        try:
            from importlib import metadata
            return metadata.version(dist)
        except Exception:
            return getattr(mod, '__version__', 'unknown')

    out = {'_format': 1, 'stub': bool(a.stub), 'pyccs': version(pyccs, 'pyccs'), 'pyspoa': version(spoa, 'pyspoa'), 'numpy': np.__version__,
           'poa_scores': list(POA_SCORES), 'recipes': {'C3': ['c2_batch', N_C3, synth.SEEDS['C3']], 'C4': ['c4_batch', N_C4, synth.SEEDS['C4']], 'families': [SEED_FAMILIES, N_FAMILIES]}}
    raw = ''.join(TEST_POA_SEGMENTS)
    seg, ccs = pyccs.find_consensus(raw)
    cons, msa = spoa.poa(list(TEST_POA_SEGMENTS), 0, True, *POA_SCORES)
    out['test_poa'] = {'segments': text(seg), 'ccs': text(ccs), 'poa_consensus': text(cons), 'poa_msa': [text(r) for r in msa]}
    reads = golden_reads()
    if a.limit:
        reads = [r for r in reads if r[1] < a.limit]
    rows = []
    for recipe, k, s in reads:
        seg, ccs = pyccs.find_consensus(s)
        rows.append([recipe, k, crc(s), text(seg), text(ccs)])
    out['reads'] = rows
    fams = golden_families()
    if a.limit:
        fams = fams[:max(1, a.limit // 4)]
    frows = []
    for k, fam in enumerate(fams):
        per = []
        for alg in (0, 1, 2):
            try:
                cons, msa = spoa.poa(list(fam), alg, True, *POA_SCORES)
            except Exception as ex:      # spoa throws on an alignment without a base (local mode, unrelated sequences): recorded as such
                per.append(['!' + type(ex).__name__, 0, 0])
                continue
            per.append([text(cons), crc('\n'.join(text(r) for r in msa)), len(msa)])
        frows.append([k, crc('\n'.join(fam)), per])
    out['families'] = frows
    with gzip.open(a.out, 'wt') as f:
        json.dump(out, f, separators=(',', ':'))
    n_ccs = sum(1 for r in rows if r[3])
    print('%s: pyccs %s, pyspoa %s%s; test_poa segments %s; %d reads (%d with a consensus), %d families x 3 algorithms'
          % (a.out, out['pyccs'], out['pyspoa'], ' (STUB)' if a.stub else '', out['test_poa']['segments'], len(rows), n_ccs, len(frows)))


if __name__ == '__main__':
    main()
