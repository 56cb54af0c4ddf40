import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle_lib
from ciri_long_amd import synth, pyccs, hip
rng = np.random.Generator(np.random.PCG64(77))
tm = synth.template(rng)
r = synth.rolling_circle_read(rng, tm, 700)
print(oracle_lib.oracle_find_consensus(r)[0], len(r), len(tm), flush=True)
print(pyccs.find_consensus_batch([r])[0])
