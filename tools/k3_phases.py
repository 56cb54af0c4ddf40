"""Per-phase clock breakdown of K3 (needs a library built with -DCLH_DEBUG_POA: tools/k3_phases.sh builds it)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciri_long_amd import hip, synth

hip.SO_PATH = os.path.join(os.path.dirname(hip.SO_PATH), 'libclh_dbg.so')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reads, _ = synth.c2_batch(n, seed=synth.SEEDS['C3'])
rd, ro = hip.pack(reads)
d_r = torch.from_numpy(rd.view(np.uint8)).cuda()
ctx = hip.Context(0)
plan = ctx.ccs_plan(ro)
st = torch.cuda.current_stream().cuda_stream
plan.run(d_r.data_ptr(), st)
torch.cuda.synchronize()
rows, segs, ccs = plan.fetch()
segs = np.asarray(segs).reshape(n, -1, 2)
t = segs[:, 55:60, 0].astype(np.float64) * 16        # s_memtime ticks (100 MHz)
ok = rows['nseg'] > 0
names = ['dp rows', 'walk back', 'graph update', 'rerank', 'consensus']
tot = t[ok].sum()
for k, nm in enumerate(names):
    print('%-13s %6.1f %%   mean %8.1f us per read' % (nm, 100 * t[ok, k].sum() / tot, t[ok, k].mean() / 100.0))
x = segs[:, 60:63, 0].astype(np.float64)
print('row steps %.4g, register-rows %.4g (mean %.2f per step), cells %.4g' % (x[ok, 0].sum(), x[ok, 1].sum(), x[ok, 1].sum() / x[ok, 0].sum(), x[ok, 2].sum() * 16))
y = segs[:, 49:55, 0].astype(np.float64)[ok].sum(0)
print('row step sections (%% of the DP clock): decode+score %.1f, sources+candidates %.1f, scans %.1f, H/E+horizontal codes %.1f, stores %.1f, end cell+carry %.1f' % tuple(100 * y / y.sum()))
print('reads with consensus %d, mean total %.1f us' % (ok.sum(), t[ok].sum(1).mean() / 100.0))
z = segs[:, 42:46, 0].astype(np.float64)[ok].sum(0)
print('back-track per read: diagonal runs %.1f (mean length %.2f), single steps %.1f, band reloads %.1f' % (z[0] / ok.sum(), z[1] / max(z[0], 1), z[3] / ok.sum(), z[2] / ok.sum()))
zz = segs[:, 40:42, 0].astype(np.float64)[ok].sum(0) * 16
print('back-track clock: band staging %.1f %%, set-up (sequence to LDS) %.1f %% of the walk-back time' % (100 * zz[0] / t[ok, 1].sum(), 100 * zz[1] / t[ok, 1].sum()))
