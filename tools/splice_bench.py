"""K6 rate: splice-signal search for N candidate junctions on a resident synthetic genome vs the per-read Python
statement (ciri_long_amd/align.py, what the reference runs per read).  usage: python tools/splice_bench.py [N] [genome_mb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ciri_long_amd import align, env, hip  # noqa: E402


class _Genome(object):
    def __init__(self, contigs):
        self.genome = dict(contigs)
        self.contig_len = {k: len(v) for k, v in self.genome.items()}

    def seq(self, ctg, start, end):
        return self.genome[ctg][start:end]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(3)
    contigs = {}
    for c in range(4):
        contigs['chr%d' % c] = np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, size=mb * 250000)].tobytes().decode()
    host = _Genome(contigs)
    env.initializer(None, host.contig_len, host, None, None, None)
    ctx = hip.Context(0)
    t0 = time.time(); dev = hip.Genome(ctx, contigs); t_up = time.time() - t0
    L = mb * 250000
    st = rng.integers(1000, L - 60000, size=n)
    ln = rng.integers(150, 50000, size=n)
    cb = rng.integers(0, 21, size=n)
    hm = rng.integers(0, 4, size=n)
    names = ['chr%d' % (k & 3) for k in range(n)]
    cands = [(names[k], int(st[k]), int(st[k] + ln[k]), int(cb[k]), int(hm[k])) for k in range(n)]
    dev.splice_signals(cands[:1000])
    best = 1e9
    for _ in range(3):
        t0 = time.time(); rows = dev.splice_signals(cands); best = min(best, time.time() - t0)
    found = int(rows[:, 3].sum())
    # the C-ABI call alone (arrays prepared): task upload + kernel + result download
    off = np.array([dev.offset[c[0]] for c in cands], dtype=np.int64); lnn = np.array([dev.length[c[0]] for c in cands], dtype=np.int64)
    st64 = st.astype(np.int64); en64 = (st + ln).astype(np.int64); cb32 = cb.astype(np.int32); hm32 = hm.astype(np.int32)
    out = np.zeros((n, 8), dtype=np.int32)
    t_abi = 1e9
    for _ in range(3):
        t0 = time.time()
        rc = hip.lib().clh_splice_signal_batch(dev._h, n, off.ctypes.data, lnn.ctypes.data, st64.ctypes.data, en64.ctypes.data, cb32.ctypes.data,
                                               hm32.ctypes.data, 10, 3, 1, out.ctypes.data)
        t_abi = min(t_abi, time.time() - t0)
    assert rc == 0 and (out == rows).all()
    cols = dict(ctg_off=off, ctg_len=lnn, start=st64, end=en64, clip_base=cb32, host_mask=hm32)
    assert (dev.splice_signals(cols) == rows).all()
    print('C-ABI call alone: %.1f ms = %.1f M candidates/s' % (t_abi * 1e3, n / t_abi / 1e6))
    # the same with an annotation loaded: 2 M sites, half of them at the candidates' own ends
    ss = {}
    for k in range(0, n, 2):
        d = ss.setdefault(names[k], {})
        d.setdefault(int(st[k]) + 1, {}).setdefault('+-'[k & 2 > 0], {})['start'] = 1
        d.setdefault(int(st[k] + ln[k]), {}).setdefault('+-'[k & 2 > 0], {})['end'] = 1
    for c in range(4):
        d = ss.setdefault('chr%d' % c, {})
        for p in rng.integers(1, L, size=250000):
            d.setdefault(int(p), {}).setdefault('+', {})['start'] = 1
    t0 = time.time(); dev.set_splice_sites(ss); t_ss = time.time() - t0
    t_an = 1e9
    for _ in range(3):
        t0 = time.time(); rows_a = dev.splice_signals(cols); t_an = min(t_an, time.time() - t0)
    print('with %d annotated sites (flatten+upload %.1f s): %.1f ms per call = %.1f M candidates/s; %d annotated pairs, %d de novo'
          % (sum(len(v) for v in ss.values()), t_ss, t_an * 1e3, n / t_an / 1e6, int((rows_a[:, 3] == 2).sum()), int((rows_a[:, 3] == 1).sum())))
    dev.set_splice_sites(None)
    # the Python statement on a sample
    m = min(n, 3000)
    masks = {0: None, 1: {'+': 1}, 2: {'-': 1}, 3: {'+': 1, '-': 1}}
    t0 = time.time()
    for k in range(m):
        ctg, s, e, c, h = cands[k]
        site, uf, df, sig = align.find_annotated_signal(ctg, s, e, c, c + 10)
        if site is None:
            site = align.find_denovo_signal(ctg, s, e, masks[h], sig, uf, df, c, c + 10, 3, True)
        r = rows[k]
        assert r[0] == 0 and (r[1], r[2]) == (uf, df) and bool(r[3]) == (site is not None) and (site is None or (r[5], r[6]) == site[2:]), (k, r, site)
    t_py = (time.time() - t0) / m
    print('K6: %d candidates on a %d-Mb genome (upload+encode %.2f s): %.1f ms per call incl. task upload and result download = %.2f M candidates/s; '
          '%d with a signal; Python statement %.1f us per candidate (%.1f k/s, 1 core) -> x%.0f'
          % (n, mb, t_up, best * 1e3, n / best / 1e6, found, t_py * 1e6, 1e-3 / t_py, (1 / t_py) and (n / best) * t_py))


if __name__ == '__main__':
    main()
