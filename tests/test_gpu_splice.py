"""K6 (splice_scan.hip) against the Python statement of the same step (ciri_long_amd/align.py: find_annotated_signal ->
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


def _annotation(contigs, cands, seed):
    """A splice-site index in the reference's layout, with sites at and around the candidates' ends: both kinds, both
    strands, sometimes only one side, sometimes several within the search range."""
    rng = random.Random(seed)
    idx = {}
    for ctg, st, en, cb, host in cands:
        if rng.random() < 0.35:
            continue
        d = idx.setdefault(ctg, {})
        for anchor in (st, en):
            if rng.random() < 0.25:
                continue
            for _ in range(rng.choice([1, 1, 2, 4])):
                pos = anchor + rng.randint(-14, 14) + rng.choice([0, 1])
                if pos < 1:
                    continue
                d.setdefault(pos, {}).setdefault(rng.choice('+-'), {})[rng.choice(['start', 'end'])] = 1
    return idx


def _host_answer(align, cand, is_canonical):
    ctg, st, en, cb, host = cand
    site, us_free, ds_free, sig = align.find_annotated_signal(ctg, st, en, cb, cb + 10)
    if site is None:
        site = align.find_denovo_signal(ctg, st, en, host, sig, us_free, ds_free, cb, cb + 10, 3, is_canonical)
    return site, us_free, ds_free


@pytest.mark.parametrize('is_canonical,annotated', [(True, False), (False, False), (True, True), (False, True)])
def test_kernel_rows_match_python_statement(is_canonical, annotated):
    from ciri_long_amd import align, env, hip
    from ciri_long_amd.utils import revcomp
    contigs, cands = _world(77 + is_canonical + 2 * annotated, 4000)
    host = _Genome(contigs)
    ss_index = _annotation(contigs, cands, 99) if annotated else None
    env.initializer(None, host.contig_len, host, None, None, ss_index)
    ctx = hip.Context(0)
    dev = hip.Genome(ctx, contigs)
    dev.set_splice_sites(ss_index)
    rows = dev.splice_signals([(c[0], c[1], c[2], c[3], (1 if c[4] and '+' in c[4] else 0) | (2 if c[4] and '-' in c[4] else 0)) for c in cands],
                              10, 3, is_canonical).tolist()
    motifs = list(align.SPLICE_SIGNAL)
    n_dev = n_found = n_slide = n_minus = n_anno = n_edge = 0
    for cand, r in zip(cands, rows):
        status, us_free, ds_free, found, strand, i, j, m = r
        want = _host_answer(align, cand, is_canonical)
        # nothing is handed back: contig ends (Python's slice rules) and flanks with IUPAC / soft-masked characters
        # (compared as characters) are the kernel's business too
        assert status == 0, cand
        ctg, st, en, cb, _ = cand
        n_edge += st - (cb + 10) - want[1] - 2 < 0 or en + (cb + 10) + want[2] + 2 > host.contig_len[ctg]
        n_dev += 1
        got = None
        if found == 1:
            d, a = motifs[m]
            got = ('{}-{}*|{}-{}'.format(a, d, i, j), '-' if strand else '+', i, j)
        elif found == 2:
            us_ss, ds_ss = host.seq(cand[0], cand[1] + i - 2, cand[1] + i), host.seq(cand[0], cand[2] + j, cand[2] + j + 2)
            if strand:
                us_ss, ds_ss = revcomp(ds_ss), revcomp(us_ss)
            got = ('{}-{}|{}-{}'.format(us_ss, ds_ss, i, j), '-' if strand else '+', i, j)
            n_anno += 1
        if found:
            n_found += 1
            n_minus += strand
        assert (got, us_free, ds_free) == want, (cand, r, want)
        n_slide += (us_free + ds_free) > 0
    # the cases are really exercised
    assert n_dev == len(cands) and n_edge > 0.01 * n_dev and n_found > 0.4 * n_dev and n_slide > 0.3 * n_dev and n_minus > 0.1 * n_found
    assert (n_anno > 0.1 * n_dev) if annotated else n_anno == 0
    dev.close(); ctx.close()


@pytest.mark.parametrize('annotated', [False, True])
def test_kernel_rows_match_c_oracle(annotated):
    """The kernel's rows against oracle/splice_oracle.c (pinned to the reference's outputs by tests/test_splice_oracle.py),
    field by field, all five motif classes."""
    import oracle_lib
    from ciri_long_amd import hip
    contigs, cands = _world(1234 + annotated, 3000)
    ss_index = _annotation(contigs, cands, 17) if annotated else None
    ctx = hip.Context(0)
    dev = hip.Genome(ctx, contigs)
    dev.set_splice_sites(ss_index)
    rows = dev.splice_signals([(c[0], c[1], c[2], c[3], (1 if c[4] and '+' in c[4] else 0) | (2 if c[4] and '-' in c[4] else 0)) for c in cands],
                              10, 3, False).tolist()
    runs = {c: hip.flatten_splice_sites({c: ss_index[c]} if ss_index and c in ss_index else None, {c: 0}, {c: len(v)}) for c, v in contigs.items()}
    motifs = oracle_lib._SPLICE_MOTIFS
    n = 0
    for cand, r in zip(cands, rows):
        assert r[0] == 0, cand
        ctg, st, en, cb, hs = cand
        want = oracle_lib.oracle_splice_signal(contigs[ctg], st, en, cb, hs, False, runs[ctg])
        assert want != 'edge', cand
        got = None
        if r[3] == 1:
            got = ('{}-{}*|{}-{}'.format(motifs[r[7]][1], motifs[r[7]][0], r[5], r[6]), '-' if r[4] else '+', r[5], r[6])
        elif r[3] == 2:
            us_ss, ds_ss = contigs[ctg][st + r[5] - 2:st + r[5]], contigs[ctg][en + r[6]:en + r[6] + 2]
            if r[4]:
                us_ss, ds_ss = oracle_lib._rc_upper(ds_ss), oracle_lib._rc_upper(us_ss)
            got = ('{}-{}|{}-{}'.format(us_ss, ds_ss, r[5], r[6]), '-' if r[4] else '+', r[5], r[6])
        assert (got, r[1], r[2]) == want, (cand, r, want)
        n += 1
    dev.close(); ctx.close()
    assert n == len(cands)


@pytest.mark.parametrize('annotated', [False, True])
def test_find_signal_batch_equals_per_read_path(annotated):
    """align.find_signal_batch (GPU rows + the Python statement for what the kernel hands back) == the per-read calls."""
    from ciri_long_amd import align, env, hip
    contigs, cands = _world(5, 2000)
    host = _Genome(contigs)
    ss_index = _annotation(contigs, cands, 7) if annotated else None
    env.initializer(None, host.contig_len, host, None, None, ss_index)
    want = [_host_answer(align, c, True) for c in cands]
    env.initializer(None, host.contig_len, align.DeviceGenome(host, contigs), None, None, ss_index)
    got = align.find_signal_batch(cands, True)
    assert got == want
    env.GENOME.device.close()


def test_reference_goldens():
    """Every splice-signal case of the reference-made fixture (annotated index of the fixture world loaded): the kernel must
    return what the REFERENCE returned, for the annotated search and for the de-novo search after it (cases whose
    ranking is tied in the reference's set order excluded)."""
    import gzip
    import fake_mapper
    from ciri_long_amd import align, env
    with gzip.open(os.path.join(HERE, 'golden', 'bsj_golden.json.gz'), 'rt') as f:
        golden = json.load(f)
    world = fake_mapper.build_world()
    g = world['genome']
    env.initializer(None, g.contig_len, align.DeviceGenome(g, g.genome), world['gtf_index'], None, world['ss_index'])
    cases = [s for s in golden['signals'] if not s['tie']]
    got = align.find_signal_batch([(s['ctg'], s['start'], s['end'], s['clip_base'], s['host']) for s in cases], True)
    rows = env.GENOME.device.splice_signals([(s['ctg'], s['start'], s['end'], s['clip_base'], 0) for s in cases])
    n_anno = n_denovo = 0
    for s, (site, us_free, ds_free) in zip(cases, got):
        want = s['annotated'][0] if s['annotated'][0] is not None else s['denovo']
        assert (list(site) if site else None) == want and [us_free, ds_free] == s['annotated'][1:3], s
        n_anno += s['annotated'][0] is not None
        n_denovo += s['annotated'][0] is None and s['denovo'] is not None
    env.GENOME.device.close()
    assert int((rows[:, 0] == 0).sum()) == len(cases)            # every case answered by the kernel, none handed back
    assert n_anno >= 20 and n_denovo >= 20


def test_main_pass_genome_as_a_minimap2_index_serves_it():
    """env.GENOME of the reference's main pass is the mappy index itself (find_bsj.py:340-341): upper case, anything but ACGT an
    N, no sequence for a start in front of the contig.  With align.IndexGenome resident on the GPU, K6 must give the REFERENCE's
    answers under that genome (tests/golden/make_splice_golden.py, the two `index` configurations: soft-masked runs, IUPAC letters,
    candidates at the contig starts), and K1 must complement every base of a minus-strand window (the raw-text route keeps
    lower-case bases uncomplemented, as the reference's revcomp() does)."""
    import gzip
    import numpy as np
    import oracle_lib
    from ciri_long_amd import align, env, hip
    from ciri_long_amd.utils import revcomp
    with gzip.open(os.path.join(HERE, 'golden', 'splice_golden.json.gz'), 'rt') as f:
        golden = [c for c in json.load(f) if c.get('index')]
    assert len(golden) == 2
    n = n_none_at_start = 0
    for cfg in golden:
        contigs, cands = _world(cfg['seed'], cfg['n'])
        host = align.IndexGenome(_Genome(contigs))
        ss_index = _annotation(contigs, cands, cfg['seed'] + 1) if cfg['annotated'] else None
        env.initializer(None, host.contig_len, align.DeviceGenome(host, host.genome), None, None, ss_index)
        assert env.GENOME.index_slices
        got = align.find_signal_batch(cands, cfg['canonical'])
        rows = env.GENOME.device.splice_signals([(c[0], c[1], c[2], c[3], 0) for c in cands], index_slices=True)
        assert int((rows[:, 0] != 0).sum()) == 0                    # all answered by the kernel
        for cand, (site, us_free, ds_free), (want_site, wu, wd, tied) in zip(cands, got, cfg['rows']):
            assert [us_free, ds_free] == [wu, wd] and (site is None) == (want_site is None), cand
            if not tied:
                assert (list(site) if site else None) == want_site, (cand, site, want_site)
            n += 1
            n_none_at_start += cand[1] - (cand[3] + 10) - wu - 2 < 0 and want_site is None
        if cfg is golden[0]:
            # K1 on minus-strand windows of the folded genome, soft-masked stretches included
            ctg = 'ctg0'
            text = host.genome[ctg]
            lows = [k for k, ch in enumerate(contigs[ctg]) if ch.islower()]
            assert lows
            wins, reads = [], []
            for a in (max(0, lows[0] - 150), max(0, lows[len(lows) // 2] - 200)):
                b = min(len(text), a + 600)
                wins.append((ctg, a, b))
                reads.append(revcomp(text[a + 80:a + 80 + 300]))
            data, off = hip.pack(reads)
            rows1, _c = env.GENOME.device.ssw_windows(data, off, wins, np.ones(len(wins), dtype=np.uint8), hip.score_matrix(1, 1), 1, 1)
            for (c2, a, b), q, r in zip(wins, reads, rows1):
                want = oracle_lib.oracle_align(revcomp(text[a:b]), q, 1, 1, 1, 1)
                assert (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1'])) == (want['score'], want['ref_begin'], want['ref_end'])
                assert int(r['score1']) >= 150          # (an exact substring; N of the folded text score 0)
        env.GENOME.device.close()
    assert n == 3000 and n_none_at_start > 20
