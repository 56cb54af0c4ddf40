import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from ciri_long_amd import hip, synth
reads, wins = synth.c2_batch(10000)
rd, ro = hip.pack(reads); fd, fo = hip.pack(wins)
ctx = hip.Context(0)
rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(1,1), 1, 1)
st = rows['status']
print('status hist', {int(s): int((st==s).sum()) for s in np.unique(st)})
bad = np.where(st & ~9)[0]
print('bad idx', bad[:20])
from oracle_lib import oracle_align
for k in bad[:5]:
    w = oracle_align(wins[k], reads[k])
    r = rows[k]
    print(k, len(reads[k]), r, w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end'], len(w['cigar']))
