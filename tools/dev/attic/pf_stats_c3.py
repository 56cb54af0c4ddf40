"""what the prefilter does on the clips of the C3 step with production windows, by clip length: python tools/dev/pf_stats_c3.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
import bench
from ciri_long_amd import hip, synth
ctx = hip.Context(0)
fs = bench.FullStep(torch, hip, synth, ctx, 'c3', 100000, 0, None, prod_windows=True)
fs.step()
rows = fs.last['rows']
clips = fs.last['clips'].cpu().numpy().view(np.int8)
lens = fs.clen
print('clips', len(lens), 'mean score/L %.3f' % float(np.mean(rows['score1'] / lens)), 'L quantiles', np.percentile(lens, [0, 10, 50, 90, 100]))
for lo, hi in [(1, 32), (32, 48), (48, 64), (64, 96), (96, 128), (128, 192), (192, 255), (255, 2000)]:
    sel = np.nonzero((lens >= lo) & (lens < hi))[0]
    if not len(sel):
        continue
    cd, co = hip.pack([clips[fs.co[i]:fs.co[i + 1]] for i in sel])
    d = torch.from_numpy(cd.view(np.uint8)).cuda()
    plan = fs.genome.plan_windows(co, fs.win_off[sel], fs.win_len[sel].astype(np.int32), np.zeros(len(sel), dtype=np.uint8), hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=False, want_cigar=False)
    plan.run(d.data_ptr(), fs.genome.codes_ptr, fs.stream); plan.fetch()
    t0 = time.perf_counter()
    plan.run(d.data_ptr(), fs.genome.codes_ptr, fs.stream); r, _ = plan.fetch()
    el = time.perf_counter() - t0
    s = plan.prefilter_stats()
    sc = r['score1'] / lens[sel]
    print('L %3d..%4d  n %5d  %.2f ms  pruned %5d  slices/clip %.1f  cols %.4f of window  score/L mean %.2f  p10 %.2f' % (lo, hi, len(sel), el * 1e3, s['pruned'], s['slices'] / len(sel),
          s['cols_computed'] / max(1, s['cols_window']), float(sc.mean()), float(np.percentile(sc, 10))))
    plan.close()
