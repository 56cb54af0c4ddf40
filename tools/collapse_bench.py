"""Throughput of the collapse-stage batches on C5-shaped work: the pairwise distance matrices of
cluster_sequence (50 homopolymer-compressed reads per cluster) and the junction grid of curate_junction."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from ciri_long_amd import hip, synth, utils, ssw_wrap

ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C5']))
B = 'ACGT'
xs, ys, cells = [], [], 0
for c in range(ncl):
    tm = synth.template(rng)
    reads = [utils.compress_seq(''.join(B[b] for b in synth.mutate(np.roll(tm, int(rng.integers(0, len(tm)))), rng))) for _ in range(50)]
    for i in range(50):
        for j in range(i + 1, 50):
            xs.append(reads[i]); ys.append(reads[j]); cells += len(reads[i]) * len(reads[j])
ctx = hip.default_context()
ctx.edit_distance_batch(xs[:1000], ys[:1000])
t0 = time.perf_counter()
d = ctx.edit_distance_batch(xs, ys)
dt = time.perf_counter() - t0
print('K4 pairwise: %d clusters, %d pairs, %.1f G cells in %.3f s (host packing + H2D included): %.0f pairs/s, %.1f GCUPS' %
      (ncl, len(xs), cells / 1e9, dt, len(xs) / dt, cells / dt / 1e9))
try:
    import oracle_lib
    k = 2000
    t0 = time.perf_counter()
    want = [oracle_lib.oracle_edit_distance(x, y) for x, y in zip(xs[:k], ys[:k])]
    dc = time.perf_counter() - t0
    assert list(d[:k]) == want
    print('CPU statement (1 core): %.0f pairs/s  -> GPU/CPU(1 core) = %.0fx' % (k / dc, (len(xs) / dt) / (k / dc)))
except ImportError:
    pass
# junction grid: 2500 20-nt references against one 50-nt consensus junction per cluster
refs = [''.join(B[b] for b in rng.integers(0, 4, 20)) for _ in range(2500 * min(ncl, 40))]
qs = [''.join(B[b] for b in rng.integers(0, 4, 50))] * len(refs)
ssw_wrap.align_pairs(refs[:100], qs[:100], 10, 4, 8, 2)
t0 = time.perf_counter()
al = ssw_wrap.align_pairs(refs, qs, 10, 4, 8, 2)
parts = [q[a.query_begin:a.query_end] for q, a in zip(qs, al)]
dd = utils.distance_batch(refs, parts)
dt = time.perf_counter() - t0
print('curate_junction grid: %d probes in %.3f s = %.0f probes/s (K1 + K4 + Python result objects)' % (len(refs), dt, len(refs) / dt))
