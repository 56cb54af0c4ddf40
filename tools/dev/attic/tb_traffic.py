"""one profiled C2 run (a single row-traceback launch over all 10 000 alignments) for rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
import bench
from ciri_long_amd import hip, synth
ctx = hip.Context(0)
reads, wins = bench.make_batch(synth, 'c2', 10000, 0)
rd, ro = hip.pack(reads); fd, fo = hip.pack(wins)
d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_w = torch.from_numpy(fd.view(np.uint8)).cuda()
ts = torch.cuda.Stream()
plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=True, want_cigar=True)
plan.set_profiling(True)
plan.run(d_r.data_ptr(), d_w.data_ptr(), ts.cuda_stream); torch.cuda.synchronize()
rows, cig = plan.fetch()
print('cigar words per alignment', len(cig) / len(rows))
