"""stage 1 file to files on the call_files world (error-free rolling-circle reads + linear reads): where do its seconds go?"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from ciri_long_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
w = synth.circ_world(n)
rng = np.random.Generator(np.random.PCG64(1))
B = np.frombuffer(b'ACGTN', dtype=np.uint8)
d = tempfile.mkdtemp(dir='/tmp')
fq = os.path.join(d, 'in.fastq')
with open(fq, 'wb') as f:
    for rid, (_s, _c, raw) in w['ccs_seq'].items():
        s = raw.encode()
        f.write(b'@' + rid.encode() + b'\n' + s + b'\n+\n' + b'I' * len(s) + b'\n')
        lin = B[rng.integers(0, 4, int(max(300, rng.normal(1000, 100))))].tobytes()
        f.write(b'@lin' + rid.encode() + b'\n' + lin + b'\n+\n' + b'I' * len(lin) + b'\n')
print('file MB', os.path.getsize(fq) >> 20, 'longest raw', max(len(v[2]) for v in w['ccs_seq'].values()), flush=True)
ctx = hip.default_context()
for rep in range(3):
    t0 = time.perf_counter()
    out = ctx.ccs_file(fq, 1, os.path.join(d, 'o%d.ccs.fa' % rep), os.path.join(d, 'o%d.raw.fa' % rep))
    print('call %d: %.3f s' % (rep, time.perf_counter() - t0), out, flush=True)
shutil.rmtree(d)
