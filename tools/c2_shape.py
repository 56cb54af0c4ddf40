import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciri_long_amd import hip, synth
reads, wins = synth.c2_batch(10000)
rd, ro = hip.pack(reads); fd, fo = hip.pack(wins)
ctx = hip.Context(0)
rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(1,1), 1, 1, want_cigar=False)
rl = rows['read_end1'] - rows['read_begin1'] + 1
fl = rows['ref_end1'] - rows['ref_begin1'] + 1
w0 = np.abs(rl - fl) + 1
print('readLen  pct', np.percentile(rl, [5,25,50,75,95,100]))
print('refLen   pct', np.percentile(fl, [5,25,50,75,95,100]))
print('w0       pct', np.percentile(w0, [5,25,50,75,95,100]))
print('w0+3>512 & readLen>513:', int(((w0 + 3 > 512) & (rl > 513)).sum()), ' readLen+refLen>6144:', int((rl + fl > 6144).sum()))
print('score pct', np.percentile(rows['score1'], [5,50,95]))
