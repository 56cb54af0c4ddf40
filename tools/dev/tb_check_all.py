"""every CIGAR of the C2 batch against the CPU oracle (dev check of the traceback kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
torch.cuda.init()
import bench, oracle_lib
from ciri_long_amd import hip, synth
ctx = hip.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reads, wins = bench.make_batch(synth, 'c2', n, 0)
rd, ro = hip.pack(reads); fd, fo = hip.pack(wins)
d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_w = torch.from_numpy(fd.view(np.uint8)).cuda()
ts = torch.cuda.Stream()
plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=True, want_cigar=True)
plan.run(d_r.data_ptr(), d_w.data_ptr(), ts.cuda_stream); torch.cuda.synchronize()
print('handed to wide, to anti-diagonal:', plan.traceback_counts())
rows, cig = plan.fetch()
t0 = time.time(); bad = 0
for k in range(n):
    w = oracle_lib.oracle_align(wins[k], reads[k], 1, 1, 1, 1)
    r = rows[k]
    got = [int(x) for x in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]]
    if w is None:
        ok = bool(int(r['status']) & 4)
    else:
        ok = got == w['cigar'] and int(r['score1']) == w['score']
    if not ok:
        bad += 1
        if bad < 5: print('MISMATCH', k, int(r['status']), len(got), None if w is None else len(w['cigar']))
print('checked', n, 'bad', bad, 'in %.0f s' % (time.time() - t0))
