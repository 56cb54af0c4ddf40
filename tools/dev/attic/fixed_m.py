import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ciri_long_amd import hip
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ng = 6000
rng = np.random.default_rng(1)
seqs = []
for g in range(ng):
    t = rng.integers(0, 4, m).astype(np.int8)
    for k in range(4):
        s = t.copy()
        idx = rng.random(m) < 0.1
        s[idx] = rng.integers(0, 4, int(idx.sum()))
        seqs.append(s)
data, off = hip.pack(seqs)
goff = np.arange(0, 4 * ng + 1, 4, dtype=np.int64)
ctx = hip.Context(0)
ctx.poa_batch(data[:off[40]], off[:41], goff[:11])
t0 = time.time()
out = ctx.poa_batch(data, off, goff, algorithm=0)
print('m', m, 'groups', ng, 'time %.3f s' % (time.time() - t0), len(out[0]))
