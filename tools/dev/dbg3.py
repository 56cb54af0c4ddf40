import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle_lib
from ciri_long_amd import synth, pyccs
rng = np.random.Generator(np.random.PCG64(77))
reads = []
for it in range(300):
    tm = synth.template(rng)
    L = int(rng.choice([400, 700, 1000, 1300, 2000]))
    kind = it % 4
    if kind == 3:
        reads.append(synth.mutate(rng.integers(0, 4, L, dtype=np.int8), rng))
    else:
        r = synth.rolling_circle_read(rng, tm, L)
        if kind == 2 and len(r) > 50:
            r = r.copy(); r[rng.integers(0, len(r), 3)] = 4
        reads.append(r)
reads.append(np.zeros(40, dtype=np.int8))
reads.append(np.tile(np.array([0, 1, 2, 3], dtype=np.int8), 200))
bad=0
for k,r in enumerate(reads):
    t1=time.time()
    print('read',k,len(r),flush=True)
    got=pyccs.find_consensus_batch([r])[0]
    w=oracle_lib.oracle_find_consensus(r)
    dt=time.time()-t1
    if got!=w[:2]:
        bad+=1; print('  MISMATCH',k,len(r),got[0],w[0], len(got[1] or ''), len(w[1] or ''),flush=True)
    if dt>0.5: print('  slow',dt,flush=True)
print('bad',bad)
