import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ciri_long_amd import hip, synth
n = 200000
reads, _ = synth.c2_batch(n, seed=synth.SEEDS['C3'])
d = tempfile.mkdtemp()
fq = os.path.join(d, 'r.fq')
B = np.frombuffer(b'ACGTN', dtype=np.uint8)
with open(fq, 'wb') as f:
    for k, r in enumerate(reads):
        s = B[r].tobytes()
        f.write(b'@read%d extra\n' % k + s + b'\n+\n' + b'I' * len(s) + b'\n')
ctx = hip.default_context()
ctx.ccs_file(fq, 1, os.path.join(d, 'w.ccs.fa'), os.path.join(d, 'w.raw.fa'))
for br in (65536, 32768, 16384, 8192):
    t0 = time.perf_counter()
    tot, ro, _ = ctx.ccs_file(fq, 1, os.path.join(d, 'n.ccs.fa'), os.path.join(d, 'n.raw.fa'), batch_reads=br)
    el = time.perf_counter() - t0
    print(br, tot, ro, '%.3f s = %.0f reads/s' % (el, tot / el))
