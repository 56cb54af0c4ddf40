"""Stage 1 file -> files: the native route (clh_ccs_file) against the record loop in Python, on a synthetic FASTQ."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ciri_long_amd import find_ccs, hip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reads, _ = synth.c2_batch(n, seed=synth.SEEDS['C3'])
B = np.frombuffer(b'ACGTN', dtype=np.uint8)
d = tempfile.mkdtemp(dir='/tmp')
os.makedirs(os.path.join(d, 'tmp'))
fq = os.path.join(d, 'in.fastq')
with open(fq, 'wb') as f:
    for k, r in enumerate(reads):
        s = B[r].tobytes()
        f.write(b'@read%07d\n' % k + s + b'\n+\n' + b'I' * len(s) + b'\n')
size = os.path.getsize(fq)
ctx = hip.default_context()
ctx.ccs_file(fq, 1, os.path.join(d, 'tmp', 'w.ccs.fa'), os.path.join(d, 'tmp', 'w.raw.fa'))
for _ in range(3):
    t0 = time.perf_counter()
    tot, ro, _ = ctx.ccs_file(fq, 1, os.path.join(d, 'tmp', 'n.ccs.fa'), os.path.join(d, 'tmp', 'n.raw.fa'))
    print('   (wall %.3f s around a call)' % (time.perf_counter() - t0))
t0 = time.perf_counter()
tot, ro, _ = ctx.ccs_file(fq, 1, os.path.join(d, 'tmp', 'n.ccs.fa'), os.path.join(d, 'tmp', 'n.raw.fa'))
tn = time.perf_counter() - t0
print('   (wall %.3f s around the call)' % tn)
print('native : %d reads (%d MB FASTQ), %d with consensus, %.2f s = %.0f reads/s (%.0f MB/s)' % (tot, size >> 20, ro, tn, tot / tn, size / tn / 1e6))
m = min(n, 40000)
fq2 = os.path.join(d, 'small.fastq')
with open(fq, 'rb') as f, open(fq2, 'wb') as g:
    for _ in range(4 * m):
        g.write(f.readline())
t0 = time.perf_counter()
tot2, ro2, _ = find_ccs.find_ccs_reads_py(fq2, d, 'p', 1, False)
tp = time.perf_counter() - t0
print('python : %d reads, %.2f s = %.0f reads/s   -> native is %.1fx' % (tot2, tp, tot2 / tp, (tot / tn) / (tot2 / tp)))
