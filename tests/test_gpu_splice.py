"""K6 (splice_scan.hip) against the Python statement of the same step (ciri-long_amd/align.py: find_annotated_signal ->
find_denovo_signal, themselves pinned to outputs of the reference by test_bsj_host.test_splice_signal_search)."""
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


class _Genome(object):
    def __init__(self, contigs):
        self.genome = dict(contigs)
        self.contig_len = {k: len(v) for k, v in self.genome.items()}

    def seq(self, ctg, start, end):
        g = self.genome.get(ctg)
        return None if g is None else g[start:end]


def _world(seed, n_cand):
    """Contigs with soft-masked runs, N runs, a few IUPAC characters, and candidates whose two ends share flanks of
    0..120 bases (so that the junction slides), some of them at the contig ends."""
    rng = random.Random(seed)
    contigs = {}
    cands = []
    for c in range(4):
        L = rng.randint(20000, 40000)
        s = [rng.choice('ACGT') for _ in range(L)]
        for _ in range(6):
            a = rng.randrange(L - 400); b = a + rng.randint(5, 300)
            for k in range(a, b):
                s[k] = s[k].lower()
        for _ in range(4):
            a = rng.randrange(L - 200); b = a + rng.randint(1, 150)
            for k in range(a, b):
                s[k] = 'N'
        for _ in range(20):
            s[rng.randrange(L)] = rng.choice('RYKMnSW')
        name = 'ctg%d' % c
        per = n_cand // 4
        for k in range(per):
            kind = rng.random()
            if kind < 0.06:
                st = rng.randint(0, 140)
            elif kind < 0.12:
                st = L - rng.randint(200, 1200)
            else:
                st = rng.randrange(200, L - 3500)
            en = min(L - (rng.randint(0, 140) if kind >= 0.06 and kind < 0.12 else 150), st + rng.randint(60, 3000))
            if en <= st:
                continue
            if rng.random() < 0.6 and st > 130 and en + 130 < L:
                # shared flanks: copy what follows/precedes start to the same place at end
                f = rng.randint(0, 120); b = rng.randint(0, 120)
                if en - st > f + b + 4:
                    s[en:en + f] = s[st:st + f]
                    s[en - b:en] = s[st - b:st]
            if rng.random() < 0.7:
                # plant canonical signals near the ends: AG before start, GT after end (plus), or CT / AC (minus)
                i, j = rng.randint(-8, 8), rng.randint(-8, 8)
                us, ds = ('AG', 'GT') if rng.random() < 0.5 else ('AC', 'CT')
                if st + i - 2 >= 0 and en + j + 2 <= L:
                    s[st + i - 2:st + i] = us
                    s[en + j:en + j + 2] = ds
            cands.append((name, st, en, rng.randint(0, 20), rng.choice([None, None, {'+': 1}, {'-': 1}, {'+': 1, '-': 1}])))
        contigs[name] = ''.join(s)
    return contigs, cands


def _host_answer(align, cand, is_canonical):
    ctg, st, en, cb, host = cand
    site, us_free, ds_free, sig = align.find_annotated_signal(ctg, st, en, cb, cb + 10)
    if site is None:
        site = align.find_denovo_signal(ctg, st, en, host, sig, us_free, ds_free, cb, cb + 10, 3, is_canonical)
    return site, us_free, ds_free


@pytest.mark.parametrize('is_canonical', [True, False])
def test_kernel_rows_match_python_statement(is_canonical):
    from ciri_long_amd import align, env, hip
    contigs, cands = _world(77 + is_canonical, 4000)
    host = _Genome(contigs)
    env.initializer(None, host.contig_len, host, None, None, None)
    ctx = hip.Context(0)
    dev = hip.Genome(ctx, contigs)
    rows = dev.splice_signals([(c[0], c[1], c[2], c[3], (1 if c[4] and '+' in c[4] else 0) | (2 if c[4] and '-' in c[4] else 0)) for c in cands],
                              10, 3, is_canonical).tolist()
    motifs = list(align.SPLICE_SIGNAL)
    n_dev = n_found = n_slide = n_minus = 0
    for cand, r in zip(cands, rows):
        status, us_free, ds_free, found, strand, i, j, m = r
        want = _host_answer(align, cand, is_canonical)
        if status:
            # handed back: contig end or an ambiguous character in the flanks -- never silently wrong
            ctg, st, en, cb, _ = cand
            L = host.contig_len[ctg]
            edge = st - (cb + 10) - want[1] - 2 < 0 or en + (cb + 10) + want[2] + 2 > L
            flank = contigs[ctg][max(0, st - 100):st + 100] + contigs[ctg][max(0, en - 100):en + 100]
            assert edge or any(ch not in 'ACGTacgtN' for ch in flank), cand
            continue
        n_dev += 1
        got = None
        if found:
            d, a = motifs[m]
            got = ('{}-{}*|{}-{}'.format(a, d, i, j), '-' if strand else '+', i, j)
            n_found += 1
            n_minus += strand
        assert (got, us_free, ds_free) == want, (cand, r, want)
        n_slide += (us_free + ds_free) > 0
    # the cases are really exercised
    assert n_dev > 0.8 * len(cands) and n_found > 0.4 * n_dev and n_slide > 0.3 * n_dev and n_minus > 0.1 * n_found
    dev.close(); ctx.close()


def test_find_signal_batch_equals_per_read_path():
    """align.find_signal_batch (GPU rows + the Python statement for what the kernel hands back) == the per-read calls."""
    from ciri_long_amd import align, env, hip
    contigs, cands = _world(5, 2000)
    host = _Genome(contigs)
    env.initializer(None, host.contig_len, host, None, None, None)
    want = [_host_answer(align, c, True) for c in cands]
    env.initializer(None, host.contig_len, align.DeviceGenome(host, contigs), None, None, None)
    got = align.find_signal_batch(cands, True)
    assert got == want
    env.GENOME.device.close()


def test_reference_goldens_without_annotation():
    """Candidates of the reference-made fixture whose annotated search found nothing and added no annotated shifts: the
    kernel must return what the REFERENCE returned (ties in the reference's set order excluded)."""
    import gzip
    import fake_mapper
    from ciri_long_amd import align, hip
    with gzip.open(os.path.join(HERE, 'golden', 'bsj_golden.json.gz'), 'rt') as f:
        golden = json.load(f)
    world = fake_mapper.build_world()
    contigs = world['genome'].genome
    ctx = hip.Context(0)
    dev = hip.Genome(ctx, contigs)
    motifs = list(align.SPLICE_SIGNAL)
    n = 0
    for s in golden['signals']:
        if s['tie'] or s['annotated'][0] is not None or any(v[0] or v[1] for v in s['annotated'][3].values()):
            continue
        hm = (1 if s['host'] and '+' in s['host'] else 0) | (2 if s['host'] and '-' in s['host'] else 0)
        r = dev.splice_signals([(s['ctg'], s['start'], s['end'], s['clip_base'], hm)], 10, 3, True).tolist()[0]
        if r[0]:
            continue
        got = None
        if r[3]:
            d, a = motifs[r[7]]
            got = ['{}-{}*|{}-{}'.format(a, d, r[5], r[6]), '-' if r[4] else '+', r[5], r[6]]
        assert got == s['denovo'] and r[1:3] == s['annotated'][1:3], s
        n += 1
    dev.close(); ctx.close()
    assert n >= 100
