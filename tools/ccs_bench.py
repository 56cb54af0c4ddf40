"""Throughput probe of the consensus step (K2 + K3) on C3-shaped reads."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciri_long_amd import hip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reads, _ = synth.c2_batch(n, seed=synth.SEEDS['C3'])
rd, ro = hip.pack(reads)
d_r = torch.from_numpy(rd.view(np.uint8)).cuda()
ctx = hip.Context(0)
plan = ctx.ccs_plan(ro)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    plan.run(d_r.data_ptr(), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    plan.run(d_r.data_ptr(), st)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
rows, segs, ccs = plan.fetch()
k2, k3 = plan.timing()
print('K2 %.2f ms  K3 %.2f ms' % (k2, k3))
print('reads %d  %.2f ms/step  %.0f reads/s  with consensus %d  status!=0 %d' % (n, ms, n / ms * 1e3, int((rows['nseg'] > 0).sum()), int((rows['status'] != 0).sum())))
