import sys, time, random; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle_lib
from ciri_long_amd import hip
from test_poa_model import _mutate
ctx = hip.Context(0)
rng = random.Random(3)
for tl in (330, 350, 380, 400, 440, 100, 200, 260, 500):
    for nseq in (2,3):
        t=''.join(rng.choice('ACGT') for _ in range(tl))
        seqs=[_mutate(rng,t,0.13) for _ in range(nseq)]
        print('case',tl,nseq,[len(s) for s in seqs],flush=True)
        want = oracle_lib.oracle_poa(seqs, 0, False)
        data, off = hip.pack(seqs)
        got = ctx.poa_batch(data, off, np.array([0, len(seqs)], dtype=np.int64), algorithm=0)[0]
        print('  ok' if got==want else '  MISMATCH',flush=True)
