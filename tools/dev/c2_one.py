#!/usr/bin/env python3
"""One C2 batch alone (no second batch in flight): tools/dev/c2_one.py [name ...]  -- libclh_<name>.so builds from tools/dev/variants.sh
(`base` = libclh.so), each in this process's child; prints ms per batch (run + fetch, best of 7), a checksum of rows + CIGARs."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, time, zlib, numpy as np
sys.path.insert(0, %r)
import torch
from ciri_long_amd import hip, synth
name = sys.argv[1]
if name != 'base':
    hip.SO_PATH = os.path.join(os.path.dirname(hip.SO_PATH), 'libclh_%%s.so' %% name)
reads, wins = synth.c2_batch(10000, seed=synth.SEEDS['C2'])
rd, ro = hip.pack(reads); fd, fo = hip.pack(wins)
d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_w = torch.from_numpy(fd.view(np.uint8)).cuda()
ctx = hip.Context(0)
plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=True, want_cigar=True)
st = torch.cuda.Stream().cuda_stream
best = 1e9
for k in range(9):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.run(d_r.data_ptr(), d_w.data_ptr(), st)
    rows, cig = plan.fetch()
    el = time.perf_counter() - t0
    if k >= 2: best = min(best, el)
if name.startswith('trace'):
    import ctypes
    lib = ctypes.CDLL(hip.SO_PATH)
    buf = (ctypes.c_longlong * 8192)()
    lib.clh_debug_tbw(buf, 1)
    plan.run(d_r.data_ptr(), d_w.data_ptr(), st); plan.fetch()
    n = lib.clh_debug_tbw(buf, 1)
    a = np.array(buf[:8 * n], dtype=np.int64).reshape(n, 8)
    tot = a[:, 3] + a[:, 4] + a[:, 5] + a[:, 6]
    print('handed-over alignments traced: %%d; clocks (M) median %%.2f, max %%.2f' %% (n, np.median(tot) / 1e6, tot.max() / 1e6))
    for r in a[np.argsort(-tot)][:8]:
        print('  read %%d ref %%d  w0 %%d -> w %%d  niter %%d  state w %%d | rounds %%.2f  final plane %%.2f  walk %%.2f  lazy planes %%d: %%.2f (M clocks)' %% (r[0] >> 32, r[0] & 0xffffffff, r[1] >> 32, r[1] & 0xffffffff, r[2] >> 32, r[7], r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, r[2] & 0xffffffff, r[6] / 1e6))
    r = a[np.argsort(tot)][n // 2]
    print('  median: read %%d ref %%d  w0 %%d -> w %%d  niter %%d | rounds %%.2f  final plane %%.2f  walk %%.2f  lazy %%.2f' %% (r[0] >> 32, r[0] & 0xffffffff, r[1] >> 32, r[1] & 0xffffffff, r[2] >> 32, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, r[6] / 1e6))
crc = zlib.crc32(cig.tobytes(), zlib.crc32(rows.tobytes())) & 0xffffffff
plan.set_profiling(True)
acc = None
for k in range(3):
    plan.run(d_r.data_ptr(), d_w.data_ptr(), st)
    tm, tb = plan.timing()
    acc = (tm, tb) if acc is None else ([a + b for a, b in zip(acc[0], tm)], [a + b for a, b in zip(acc[1], tb)])
plan.fetch(); plan.set_profiling(False)
print('   score kernels %%s ms, row traceback %%.2f ms, handed-over (%%d) %%.2f ms' %% (' + '.join('%%.2f' %% (x / 3) for x in acc[0]), acc[1][0] / 3, int(plan.traceback_counts()[0]), acc[1][1] / 3), flush=True)
print('%%-10s one C2 batch %%.2f ms  status!=0 %%d  crc %%08x' %% (name, best * 1e3, int((rows['status'] & ~9 != 0).sum()), crc), flush=True)
''' % HERE
for name in (sys.argv[1:] or ['base']):
    subprocess.run([sys.executable, '-c', CHILD, name], check=False)
