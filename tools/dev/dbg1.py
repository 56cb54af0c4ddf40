import os, sys, random
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import oracle_lib
from ciri_long_amd import hip
from poa_check import families
from test_poa_model import PARS
ctx = hip.Context(0)
n=0
for rng, seqs in families(1, 40):
    for alg in (0,1,2):
        par = rng.choice(PARS); mc = rng.choice([0, 0, (len(seqs) + 1) // 2])
        want = oracle_lib.oracle_poa(seqs, alg, True, *par, with_scores=True, min_coverage=mc)
        data, off = hip.pack(seqs)
        got = ctx.poa_batch(data, off, np.array([0, len(seqs)], dtype=np.int64), algorithm=alg, scores=par, min_coverage=mc, genmsa=True, with_scores=True)[0]
        if tuple(got)!=tuple(want) and n<2 and max(len(s) for s in seqs)<100:
            n+=1
            print('alg',alg,par,'ncols',len(want[1][0]),len(got[1][0]))
            for k in range(len(seqs)):
                # find after which sequence it first differs: prefix runs
                pass
            for m in range(2,len(seqs)+1):
                w2 = oracle_lib.oracle_poa(seqs[:m], alg, True, *par)
                g2 = ctx.poa_batch(*hip.pack(seqs[:m]), np.array([0, m], dtype=np.int64), algorithm=alg, scores=par, genmsa=True)[0]
                if tuple(w2)!=tuple(g2):
                    print('first differs with', m, 'sequences')
                    for a,b in zip(w2[1],g2[1]): print(a); print(b); print()
                    print(seqs[:m])
                    break
