"""Production shape of align_clip_segments (find_bsj.py:191-216): 20-300 nt clips against hit +- 200 kb windows.
Host-window route (window string -> revcomp -> encode -> H2D, as the reference's data flow dictates) against the
resident-genome route (window = coordinate triple, read in place from HBM)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ciri_long_amd import hip, ssw_wrap, synth, utils

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C3'] + 1))
G = 20_000_000
B = np.frombuffer(b'ACGT', dtype=np.uint8)
genome = B[rng.integers(0, 4, G)].tobytes().decode()
ctx = hip.default_context()
t0 = time.perf_counter()
dg = hip.Genome(ctx, {'chr1': genome})
print('genome of %d Mb resident in %.2f s (H2D of the text + encode kernel + N prefix)' % (G // 1000000, time.perf_counter() - t0))
wins, minus, clips = [], [], []
for k in range(n):
    c = int(rng.integers(300000, G - 300000))
    s, e = c - 200000, c + 200000 + int(rng.integers(100, 1500))
    L = int(rng.integers(20, 301))
    p = int(rng.integers(s, e - L))
    q = ''.join('ACGT'[b] for b in synth.mutate(np.frombuffer(genome[p:p + L].encode(), dtype=np.uint8) % 5 % 4, rng))   # noisy; base identity irrelevant here
    rc = bool(rng.integers(0, 2))
    wins.append(('chr1', s, e)); minus.append(rc); clips.append(q)
ssw_wrap.align_windows(dg, wins[:50], minus[:50], clips[:50], 1, 1, 1, 1)
t0 = time.perf_counter()
cnt = dg.count_n(wins)
a = ssw_wrap.align_windows(dg, wins, minus, clips, 1, 1, 1, 1)
t_res = time.perf_counter() - t0
m = min(n, 300)
t0 = time.perf_counter()
strings = []
for (c, s, e), rc in zip(wins[:m], minus[:m]):
    w = genome[s:e]
    if w.count('N') >= 0.3 * (e - s):
        continue
    strings.append(utils.revcomp(w) if rc else w)
b = ssw_wrap.align_pairs(strings, clips[:m], 1, 1, 1, 1)
t_host = time.perf_counter() - t0
assert [(x.score, x.ref_begin, x.ref_end) for x in a[:m]] == [(x.score, x.ref_begin, x.ref_end) for x in b]
cells = sum((e - s) * len(q) for (c, s, e), q in zip(wins, clips))
print('resident genome : %d clips in %.3f s = %.0f clips/s (%.0f GCUPS incl. host packing of the clips)' % (n, t_res, n / t_res, cells / t_res / 1e9))
print('host windows    : %d clips in %.3f s = %.0f clips/s   -> %.1fx' % (m, t_host, m / t_host, (n / t_res) / (m / t_host)))
