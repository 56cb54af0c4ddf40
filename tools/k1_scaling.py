"""Microbench: K1 (score kernel) time vs number of identical-size alignments -> resident waves per CU and steady-state rate."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciri_long_amd import hip, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(1)
ctx = hip.Context(0)
base_w = rng.integers(0, 4, R, dtype=np.int8)
for n in [256, 1024, 2048, 4096, 6144, 8192, 12288, 16384, 32768]:
    reads, wins = [], []
    for i in range(n):
        st = int(rng.integers(0, R - L)) if R > L else 0
        reads.append(synth.mutate(base_w[st:st + L], rng)[:L])
        wins.append(base_w)
    rd, ro = hip.pack(reads); fd, fo = hip.pack(wins)
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, want_score2=False, want_cigar=False)
    st_ = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.run(d_r.data_ptr(), d_f.data_ptr(), st_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        plan.run(d_r.data_ptr(), d_f.data_ptr(), st_)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    cells = n * (L * R + L * L)
    print('n=%6d  %.3f ms  %.1f us/aln  %.0f aln/s  ~%.0f GCUPS' % (n, ms, ms * 1e3 / n, n / ms * 1e3, cells / ms / 1e6), flush=True)
    plan.close()
