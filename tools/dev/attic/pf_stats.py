"""what the prefilter does on the production shape, by clip length (A/B tool): python tools/dev/pf_stats.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
import bench
from ciri_long_amd import hip, synth
ctx = hip.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C3'] + 1))
G = 20_000_000
codes = rng.integers(0, 4, G).astype(np.int8)
genome = hip.Genome(ctx, [('chr1', bench.B_ASCII[codes].tobytes().decode())])
woff = np.zeros(n, dtype=np.int64); wlen = np.zeros(n, dtype=np.int64); clips = []
for k in range(n):
    c = int(rng.integers(300000, G - 300000))
    s, e = c - 200000, c + 200000 + int(rng.integers(100, 1500))
    L = int(rng.integers(20, 301))
    p = int(rng.integers(s, e - L))
    clips.append(synth.mutate(codes[p:p + L], rng))
    woff[k], wlen[k] = s, e - s
minus = rng.integers(0, 2, n).astype(np.uint8)
lens = np.array([len(c) for c in clips])
st = torch.cuda.Stream().cuda_stream
for lo, hi in [(1, 32), (32, 64), (64, 96), (96, 128), (128, 192), (192, 255), (1, 255)]:
    sel = np.nonzero((lens >= lo) & (lens < hi))[0]
    cd, co = hip.pack([clips[i] for i in sel])
    d = torch.from_numpy(cd.view(np.uint8)).cuda()
    plan = genome.plan_windows(co, woff[sel], wlen[sel].astype(np.int32), minus[sel], hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=False, want_cigar=False)
    plan.run(d.data_ptr(), genome.codes_ptr, st); plan.fetch()
    t0 = time.perf_counter()
    for _ in range(3):
        plan.run(d.data_ptr(), genome.codes_ptr, st); rows, _ = plan.fetch()
    el = (time.perf_counter() - t0) / 3
    s = plan.prefilter_stats()
    print('L %3d..%3d  n %4d  %.2f ms  pruned %4d  slices %6d  cols %.3f of window  mean score/L %.2f' % (lo, hi, len(sel), el * 1e3, s['pruned'], s['slices'],
          s['cols_computed'] / max(1, s['cols_window']), float(np.mean(rows['score1'] / lens[sel]))))
    plan.close()
