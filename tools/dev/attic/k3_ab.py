#!/usr/bin/env python3
"""A/B timing of K3 builds on one GPU: tools/dev/k3_ab.py [n_reads] name1 name2 ...  (libclh_<name>.so from tools/dev/variants.sh;
`base` = libclh.so).  Every build runs in its own process; K2/K3 by HIP events, best of 5 after 2 warm-up runs."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
import torch
from ciri_long_amd import hip, synth
name, n, wl = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if name != 'base':
    hip.SO_PATH = os.path.join(os.path.dirname(hip.SO_PATH), 'libclh_%%s.so' %% name)
reads, _ = (synth.c4_batch(n, seed=synth.SEEDS['C4']) if wl == 'c4' else synth.c2_batch(n, seed=synth.SEEDS['C3']))
rd, ro = hip.pack(reads)
d_r = torch.from_numpy(rd.view(np.uint8)).cuda()
ctx = hip.Context(0)
plan = ctx.ccs_plan(ro)
st = torch.cuda.current_stream().cuda_stream
best = 1e9
for k in range(7):
    plan.run(d_r.data_ptr(), st)
    torch.cuda.synchronize()
    k2, k3 = plan.timing()
    if k >= 2:
        best = min(best, k3)
rows, segs, ccs = plan.fetch()
import zlib
try:
    bm = plan.stats()['band_misses']
except Exception:
    bm = -1
print('%%-10s K3 %%.2f ms  (K2 %%.2f)  consensus %%d  status!=0 %%d  crc %%08x  band misses %%d' %% (name, best, k2, int((rows['nseg'] > 0).sum()), int((rows['status'] != 0).sum()), zlib.crc32(ccs.tobytes()) & 0xffffffff, bm))
''' % HERE

args = sys.argv[1:]
n, wl = 100000, 'c3'
if args and args[0].isdigit():
    n = int(args.pop(0))
if args and args[0] in ('c3', 'c4'):
    wl = args.pop(0)
for name in args or ['base']:
    sys.stdout.write(subprocess.run([sys.executable, '-c', CHILD, name, str(n), wl], capture_output=True, text=True).stdout)
    sys.stdout.flush()
