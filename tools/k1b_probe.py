"""Probe: K1 / K1b time for hand-picked alignments (single outlier, typical ones, many copies)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciri_long_amd import hip, synth

reads, wins = synth.c2_batch(600)
ctx = hip.Context(0)


def run(idx, label, copies=1):
    r = [reads[i] for i in idx] * copies; w = [wins[i] for i in idx] * copies
    rd, ro = hip.pack(r); fd, fo = hip.pack(w)
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1)
    plan.set_profiling(True)
    st_ = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        plan.run(d_r.data_ptr(), d_f.data_ptr(), st_)
        k1, kb = plan.timing()
    rows, cig = plan.fetch()
    print('%-28s n=%5d  K1 %s  K1b small %.3f ms  large %.3f ms   cigar_len %s score %s' %
          (label, len(r), ['%.2f' % x for x in k1], kb[0], kb[1], rows['cigar_len'][:4], rows['score1'][:4]), flush=True)
    plan.close()


run([577], 'outlier 577 (full band)')
run([0], 'read 0')
run([1], 'read 1')
run([2], 'read 2 (negative)')
run([0], 'read 0 x1024', 1024)
run([2], 'read 2 x1024', 1024)
run(list(range(512)), 'first 512 mixed')
run(list(range(512)), 'first 512 mixed x8', 8)
