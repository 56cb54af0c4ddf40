import os, sys, random
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import oracle_lib
from ciri_long_amd import hip
from test_poa_model import PARS, _mutate
ctx = hip.Context(0)
rng = random.Random(7)
best=None
stats={}
for it in range(3000):
    alpha = rng.choice(['ACGT','ACGTN','AC'])
    t = ''.join(rng.choice(alpha) for _ in range(rng.choice([8,12,20,30])))
    seqs=[_mutate(rng,t,rng.choice([0.1,0.3])) for _ in range(rng.randint(2,6))]
    alg=rng.choice([0,1,2]); par=rng.choice(PARS)
    want = oracle_lib.oracle_poa(seqs, alg, True, *par, with_scores=True)
    data, off = hip.pack(seqs)
    got = ctx.poa_batch(data, off, np.array([0, len(seqs)], dtype=np.int64), algorithm=alg, scores=par, genmsa=True, with_scores=True)[0]
    ok = tuple(got)==tuple(want)
    key=(alpha,alg)
    st=stats.setdefault(key,[0,0]); st[0]+=1; st[1]+= (not ok)
    if not ok:
        size=sum(len(s) for s in seqs)
        if best is None or size<best[0]: best=(size,seqs,alg,par,want,got)
print(stats)
if best:
    size,seqs,alg,par,want,got=best
    print(seqs,alg,par)
    print(want[0],want[2]); print(got[0],got[2])
    for a,b in zip(want[1],got[1]): print(a); print(b); print()
