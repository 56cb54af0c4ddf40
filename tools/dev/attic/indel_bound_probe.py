"""would an indel-distance bound (substitution = 2) prune more than the unit-cost bound?  On a sample of the C3 step's clips with +-200 kb
windows: candidate blocks (block minimum <= (L - score)) under both distances, by numpy DP on the host.  python tools/dev/indel_bound_probe.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
import bench
from ciri_long_amd import hip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ctx = hip.Context(0)
fs = bench.FullStep(torch, hip, synth, ctx, 'c3', 20000, 0, None, prod_windows=True)
fs.step()
rows = fs.last['rows']
clips = fs.last['clips'].cpu().numpy().view(np.int8)
reads, wins = bench.make_batch(synth, 'c3', 20000, 0)
text = np.minimum(np.concatenate(wins), 4).astype(np.int8)
lens = fs.clen
ratio = rows['score1'] / lens
order = np.argsort(ratio)
pick = list(order[:n // 2]) + list(order[len(order) // 2 - n // 4:len(order) // 2 + n // 4])       # the weakest, and some median ones


def semiglobal(ref, read, sub_cost):
    L = len(read)
    col_prev = None
    # row-wise: D[i][j] over all j at once
    Dp = np.zeros(len(ref) + 1, dtype=np.int32)           # row 0: free start
    for i in range(1, L + 1):
        eq = (ref == read[i - 1]) | (ref == 4) | (read[i - 1] == 4)
        diag = Dp[:-1] + np.where(eq, 0, sub_cost)
        up = Dp[1:] + 1
        best = np.minimum(diag, up)
        # left dependency: D[i][j] = min(best[j], D[i][j-1] + 1): prefix min of (best[j] - j) + j, with D[i][0] = i
        t = np.concatenate(([i], best)) - np.arange(len(ref) + 1)
        D = np.minimum.accumulate(t) + np.arange(len(ref) + 1)
        Dp = D.astype(np.int32)
    return Dp[1:]


tot = {'unit': 0, 'indel': 0, 'blocks': 0}
for k in pick:
    off, wl = int(fs.win_off[k]), int(fs.win_len[k])
    ref = text[off:off + wl]
    read = clips[fs.co[k]:fs.co[k + 1]]
    L = len(read); S0 = int(rows['score1'][k]); thr = L - S0
    nb = (wl + 255) // 256
    out = []
    for name, sc in (('unit', 1), ('indel', 2)):
        d = semiglobal(ref, read, sc)
        pad = np.full(nb * 256 - wl, 1 << 20, dtype=np.int32)
        dm = np.concatenate((d, pad)).reshape(nb, 256).min(axis=1)
        c = int((dm <= thr).sum())
        tot[name] += c
        out.append((c, int(np.median(dm))))
    tot['blocks'] += nb
    print('L %3d score %3d (%.2f) thr %3d | candidate blocks of %d: unit %5d (median block minimum %d)  indel %5d (median %d)' % (L, S0, S0 / L, thr, nb, out[0][0], out[0][1], out[1][0], out[1][1]), flush=True)
print('total candidate blocks: unit %d, indel %d of %d' % (tot['unit'], tot['indel'], tot['blocks']))
