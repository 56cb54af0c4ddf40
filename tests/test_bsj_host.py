"""Host side of the BSJ step (ciri_long_amd/find_bsj.py, align.py) against golden vectors produced by the reference's
own Python (tests/golden/make_bsj_golden.py) with the deterministic mapper/genome doubles of tests/fake_mapper.py.

CPU tests substitute the oracle for the batched GPU call (test infrastructure only); the `gpu` test runs the real
path: mapper double -> three-phase chunk -> one batched HIP Smith-Waterman call -> records."""
import gzip
import json
import os

import pytest

import fake_mapper as fm
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def golden():
    with gzip.open(os.path.join(HERE, 'golden', 'bsj_golden.json.gz'), 'rt') as f:
        return json.load(f)


@pytest.fixture(scope='module')
def world():
    from ciri_long_amd import env
    w = fm.build_world()
    w['mapper'] = fm.FakeMapper(w['genome'])
    env.initializer(w['mapper'], w['genome'].contig_len, w['genome'], w['gtf_index'], None, w['ss_index'])
    w['reads'] = fm.build_reads(w, 64)
    return w


class _Res(object):
    def __init__(self, d):
        self.score, self.ref_begin, self.ref_end = d['score'], d['ref_begin'], d['ref_end']
        self.query_begin, self.query_end = d['query_begin'], d['query_end']


def _oracle_pairs(refs, queries, match=2, mismatch=2, gap_open=3, gap_extend=1, **kw):
    return [_Res(oracle_lib.oracle_align(r, q, match, mismatch, gap_open, gap_extend)) for r, q in zip(refs, queries)]


@pytest.fixture()
def cpu_ssw(monkeypatch):
    from ciri_long_amd import ssw_wrap
    monkeypatch.setattr(ssw_wrap, 'align_pairs', _oracle_pairs)


def _tolist(x):
    return json.loads(json.dumps(x))


def _same_records(got, want, tied):
    """Records must be identical, except reads whose splice-site ranking was tied in the reference: there the
    reference's pick follows string-hash order (sort_ss sorts a set, align.py:705-733) while ours is deterministic."""
    got = {r[0]: list(r) for r in got}
    want = {r[0]: list(r) for r in want}
    assert sorted(got) == sorted(want)
    for rid in want:
        if rid in tied:
            assert got[rid][:1] + got[rid][5:7] == want[rid][:1] + want[rid][5:7]     # id, junction|clip-len, segments
        else:
            assert _tolist(got[rid]) == want[rid], rid
    assert len(tied & set(want)) <= len(want) // 4


def test_reads_are_the_generators_reads(golden, world):
    assert _tolist([list(r) for r in world['reads']]) == golden['reads']


def test_scan_ccs_chunk_matches_reference(golden, world, cpu_ssw):
    from ciri_long_amd import find_bsj
    cnt, short, ret = find_bsj.scan_ccs_chunk(world['reads'], True)
    g = golden['scan_ccs_chunk']
    assert dict(cnt) == g['counters']
    assert _tolist([list(s) for s in short]) == g['short']
    _same_records(ret, g['records'], set(golden['tied_reads']))


def test_recover_ccs_chunk_matches_reference(golden, world, cpu_ssw):
    from ciri_long_amd import find_bsj
    cnt, ret = find_bsj.recover_ccs_chunk(world['reads'], True)
    g = golden['recover_ccs_chunk']
    assert dict(cnt) == g['counters']
    _same_records(ret, g['records'], set(golden['tied_reads']))


def test_per_read_functions(golden, world, cpu_ssw):
    from ciri_long_amd import align, find_bsj
    n_ssw = 0
    for (rid, seg, ccs, raw), want in zip(world['reads'], golden['per_read']):
        circ, junc = find_bsj.find_bsj(ccs)
        assert [circ, junc] == want['find_bsj'], rid
        if 'circ_hit' not in want:
            continue
        hit = align.get_primary_alignment(world['mapper'].map(circ))
        for k, v in want['circ_hit'].items():
            got = getattr(hit, k)
            assert (_tolist([list(c) for c in got]) if k == 'cigar' else got) == v, (rid, k)
        res = find_bsj.align_clip_segments(circ, hit)
        assert [res[0], res[1], res[2], list(res[3]) if res[3] is not None else None] == want['align_clip_segments'], rid
        assert align.get_blocks(hit) == want['get_blocks'], rid
        n_ssw += res[3] is not None and res[3][0] is not None
    assert n_ssw >= 20        # the Smith-Waterman branch (>= 20 clipped bases) is really exercised


def test_splice_signal_search(golden, world):
    from ciri_long_amd import align
    for s in golden['signals']:
        host = align.find_host_gene(s['ctg'], s['start'], s['end'])
        assert (sorted(host) if host else None) == s['host']
        a = align.find_annotated_signal(s['ctg'], s['start'], s['end'], s['clip_base'], s['clip_base'] + 10)
        got = [list(a[0]) if a[0] else None, a[1], a[2], {k: [list(v[0]), list(v[1])] for k, v in a[3].items()}]
        assert got[1:] == s['annotated'][1:], s
        assert (got[0] is None) == (s['annotated'][0] is None)
        if not s['tie']:
            assert got[0] == s['annotated'][0], s
        if a[0] is None:
            d = align.find_denovo_signal(s['ctg'], s['start'], s['end'], host, a[3], a[1], a[2], s['clip_base'],
                                         s['clip_base'] + 10, 3, True)
            assert (d is None) == (s['denovo'] is None)
            if not s['tie']:
                assert (list(d) if d else None) == s['denovo'], s


class _H(object):
    pass


def test_cigar_helpers(golden):
    from ciri_long_amd import align
    for c in golden['cigar_helpers']:
        h = _H()
        h.ctg, h.strand, h.is_primary = 'chrA', 1, 1
        h.r_st, h.q_st = c['hit']['r_st'], c['hit']['q_st']
        h.cigar = [tuple(x) for x in c['hit']['cigar']]
        sub = align.remove_long_insert(h)
        for k, v in c['sub'].items():
            got = getattr(sub, k)
            assert (_tolist([list(x) for x in got]) if k == 'cigar' else got) == v, (k, c['hit'])
        blocks = align.get_blocks(h)
        assert blocks == c['blocks']
        if blocks:
            assert align.merge_clip_exon([list(b) for b in blocks], c['clip']) == c['merged']


def test_utils_revcomp_is_upper_case_only():
    from ciri_long_amd.utils import revcomp, grouper
    assert revcomp('AACGTNacgt') == 'tgcaNACGTT'      # utils.py:118-120: lower case and N pass through untouched
    assert [list(g) for g in grouper('abcde', 2)] == [['a', 'b'], ['c', 'd'], ['e', None]]


@pytest.mark.gpu
def test_scan_ccs_chunk_on_gpu(golden, world):
    from ciri_long_amd import find_bsj
    cnt, short, ret = find_bsj.scan_ccs_chunk(world['reads'], True)
    g = golden['scan_ccs_chunk']
    assert dict(cnt) == g['counters']
    _same_records(ret, g['records'], set(golden['tied_reads']))
    cnt, ret = find_bsj.recover_ccs_chunk(world['reads'], True)
    _same_records(ret, golden['recover_ccs_chunk']['records'], set(golden['tied_reads']))


def test_scan_raw_chunk_matches_reference(world):
    """third stage of `call` (find_bsj.py:499-620): mapper logic only, golden from the reference's own function"""
    from ciri_long_amd import find_bsj
    with gzip.open(os.path.join(HERE, 'golden', 'raw_golden.json.gz'), 'rt') as f:
        g = json.load(f)
    assert not g['any_tie']
    reads = [tuple(r) for r in g['reads']]
    cnt, ret, short = find_bsj.scan_raw_chunk(reads, True, {k: 1 for k in g['skip']})
    assert dict(cnt) == g['counters'] and g['counters'].get('partial', 0) >= 5
    assert _tolist([list(r) for r in ret]) == g['records']
    assert _tolist([list(s) for s in short]) == g['short']


def test_scan_raw_reads_driver(world, tmp_path):
    from ciri_long_amd import find_bsj
    with gzip.open(os.path.join(HERE, 'golden', 'raw_golden.json.gz'), 'rt') as f:
        g = json.load(f)
    fa = tmp_path / 'reads.fa'
    fa.write_text(''.join('>%s extra\n%s\n' % (h, s) for h, s in g['reads']))
    (tmp_path / 'p.cand_circ.fa').write_text(''.join('>%s\tx\n%s\n' % (k, 'ACGT') for k in g['skip']))
    cnt, short = find_bsj.scan_raw_reads(str(fa), None, world['gtf_index'], None, world['ss_index'], True, str(tmp_path), 'p', 1,
                                         aligner=world['mapper'], genome=world['genome'], contig_len=world['genome'].contig_len)
    assert dict(cnt) == g['counters'] and _tolist([list(x) for x in short]) == g['short']
    lines = (tmp_path / 'p.low_confidence.fa').read_text().split('\n')
    assert len(lines) == 2 * len(g['records']) + 1
    for k, r in enumerate(g['records']):
        assert lines[2 * k] == '>{}\t{}\t{}\t{}\t{}\t{}\t{}'.format(*r[:7]) and lines[2 * k + 1] == r[7]


def test_window_route_host_logic_on_cpu(golden, world, monkeypatch):
    """find_bsj with a resident genome (windows as coordinates): the host side of that route -- deferred N filter, strand
    flags, None for rejected windows -- with the two device calls replaced by CPU stand-ins (test infrastructure)."""
    from ciri_long_amd import env, find_bsj, ssw_wrap, utils

    import numpy as np
    from ciri_long_amd import hip

    class FakeDevice(object):
        """hip.Genome's three calls of this route on the CPU: contigs concatenated, windows as genome-wide (offset, length) spans"""

        def __init__(self, genome):
            self.offset, parts, pos = {}, [], 0
            for c, n in genome.contig_len.items():
                self.offset[c] = pos; parts.append(genome.seq(c, 0, n)); pos += n
            self.text = ''.join(parts)

        def _spans(self, wins):
            return (np.array([self.offset[c] + s for c, s, e in wins], dtype=np.int64), np.array([e - s for c, s, e in wins], dtype=np.int64))

        def count_n_spans(self, off, ln):
            return np.array([self.text[o:o + n].count('N') for o, n in zip(off.tolist(), ln.tolist())], dtype=np.int64)

        def ssw_windows(self, reads, read_off, windows, minus, mat, gap_open, gap_extend, flag=1, score_size=2, want_score2=True, want_cigar=True,
                        mask_len=None, spans=None):
            assert windows is None and not want_score2 and not want_cigar and (gap_open, gap_extend) == (1, 1)
            rows = np.zeros(len(read_off) - 1, dtype=hip.ALIGN_DTYPE)
            for k, (o, n) in enumerate(zip(spans[0].tolist(), spans[1].tolist())):
                ref = self.text[o:o + n]
                d = oracle_lib.oracle_align(utils.revcomp(ref) if minus[k] else ref, oracle_lib.decode(reads[read_off[k]:read_off[k + 1]]), 1, 1, 1, 1)
                rows[k]['score1'], rows[k]['ref_begin1'], rows[k]['ref_end1'] = d['score'], d['ref_begin'], d['ref_end']
                rows[k]['read_begin1'], rows[k]['read_end1'] = d['query_begin'], d['query_end']
            return rows, np.zeros(0, dtype=np.uint32)

    class Wrapped(object):
        def __init__(self, genome):
            self.host, self.device, self.contig_len = genome, FakeDevice(genome), genome.contig_len

        def seq(self, ctg, start, end):
            return self.host.seq(ctg, start, end)

    env.initializer(world['mapper'], world['genome'].contig_len, Wrapped(world['genome']), world['gtf_index'], None, world['ss_index'])
    try:
        cnt, short, ret = find_bsj.scan_ccs_chunk(world['reads'], True)
    finally:
        env.initializer(world['mapper'], world['genome'].contig_len, world['genome'], world['gtf_index'], None, world['ss_index'])
    g = golden['scan_ccs_chunk']
    assert dict(cnt) == g['counters']
    _same_records(ret, g['records'], set(golden['tied_reads']))


def test_find_signal_batch_host_logic_on_cpu(golden, world):
    """align.find_signal_batch with the device call replaced by a CPU stand-in that answers in the kernel's row format:
    the routing (status 1 -> Python statement), the annotated / de-novo ids and the upload-once of the site index."""
    from ciri_long_amd import align, env

    motifs = list(align.SPLICE_SIGNAL)

    class FakeDevice(object):
        def __init__(self, genome):
            self.offset = {c: 0 for c in genome.genome}
            self._sites_of = None
            self.uploads = 0
            self.handed_back = 0
            self.seen = 0

        def set_splice_sites(self, ss_index):
            self._sites_of = ss_index
            self.uploads += 1

        def splice_signals(self, cands, search_extra, shift_threshold, is_canonical, index_slices=False):
            import numpy as np
            rows = np.zeros((len(cands), 8), dtype=np.int32)
            self.seen += len(cands)
            for k, (ctg, st, en, cb, hm) in enumerate(cands):
                if k % 5 == 4:                      # as the kernel does for contig ends / odd characters
                    rows[k, 0] = 1
                    self.handed_back += 1
                    continue
                host = [s for b, s in ((1, '+'), (2, '-')) if hm & b] or None
                a = align.find_annotated_signal(ctg, st, en, cb, cb + search_extra)
                site, found = a[0], 2
                if site is None:
                    site, found = align.find_denovo_signal(ctg, st, en, host, a[3], a[1], a[2], cb, cb + search_extra, shift_threshold, is_canonical), 1
                rows[k, 1:3] = a[1:3]
                if site is not None:
                    m = 0
                    if found == 1:
                        acc, don = site[0].split('*')[0].split('-')
                        m = motifs.index((don, acc))
                    rows[k, 3:] = [found, site[1] == '-', site[2], site[3], m]
            return rows

    class Wrapped(object):
        def __init__(self, genome):
            self.host, self.device, self.contig_len = genome, FakeDevice(genome), genome.contig_len

        def seq(self, ctg, start, end):
            return self.host.seq(ctg, start, end)

    cases = golden['signals']
    cands = [(s['ctg'], s['start'], s['end'], s['clip_base'], s['host']) for s in cases]
    want = []
    for ctg, st, en, cb, host in cands:
        a = align.find_annotated_signal(ctg, st, en, cb, cb + 10)
        site = a[0] if a[0] is not None else align.find_denovo_signal(ctg, st, en, host, a[3], a[1], a[2], cb, cb + 10, 3, True)
        want.append((site, a[1], a[2]))
    wrapped = Wrapped(world['genome'])
    env.initializer(world['mapper'], world['genome'].contig_len, wrapped, world['gtf_index'], None, world['ss_index'])
    try:
        got = align.find_signal_batch(cands, True)
        got2 = align.find_signal_batch(cands[:10], True)
        seen = wrapped.device.seen
        # a host gene on strand '.' is searched under that label by the reference: stays with the Python statement
        odd = cands[0][:4] + ({'.': [1]},)
        a = align.find_annotated_signal(*odd[:4], odd[3] + 10)
        want_odd = a[0] if a[0] is not None else align.find_denovo_signal(*odd[:3], odd[4], a[3], a[1], a[2], odd[3], odd[3] + 10, 3, True)
        assert align.find_signal_batch([odd], True) == [(want_odd, a[1], a[2])] and wrapped.device.seen == seen
    finally:
        env.initializer(world['mapper'], world['genome'].contig_len, world['genome'], world['gtf_index'], None, world['ss_index'])
    assert got == want and got2 == want[:10]
    assert wrapped.device.uploads == 1 and wrapped.device.handed_back > 0
    assert sum(1 for s, _, _ in got if s and '*' not in s[0]) >= 20 and sum(1 for s, _, _ in got if s and '*' in s[0]) >= 20


def test_flatten_splice_sites():
    """The host side of clh_genome_set_splice_sites: four strictly ascending runs of genome-wide positions."""
    import collections
    import numpy as np
    from ciri_long_amd import hip
    tree = lambda: collections.defaultdict(tree)       # the reference's index is such a tree (align.py:235)
    idx = tree()
    idx['b'][10]['+']['start'] = 1
    idx['b'][10]['+']['end'] = 1
    idx['b'][10]['-']['end'] = 1
    idx['a'][7]['-']['start'] = 1
    idx['a'][7]['.']['start'] = 1          # unstranded: never looked up
    idx['a'][100]['+']['end'] = 1          # == contig length: kept
    idx['a'][101]['+']['end'] = 1          # past the contig: would alias b's position 1
    idx['a'][0]['+']['start'] = 1
    idx['zz'][5]['+']['start'] = 1         # contig not resident
    idx['b'][3]['+']['start'] = 1
    flat, cnt = hip.flatten_splice_sites(idx, {'a': 0, 'b': 100}, {'a': 100, 'b': 50})
    assert cnt.tolist() == [2, 2, 1, 1] and flat.dtype == np.int64 and flat.flags['C_CONTIGUOUS']
    assert flat.tolist() == [103, 110, 100, 110, 7, 110]
    flat, cnt = hip.flatten_splice_sites(None, {'a': 0}, {'a': 100})
    assert cnt.tolist() == [0, 0, 0, 0] and len(flat) >= 1

    class FakeLib(object):
        def clh_genome_set_splice_sites(self, h, pos, count4):
            import ctypes
            c = np.ctypeslib.as_array(ctypes.cast(count4, ctypes.POINTER(ctypes.c_int64)), (4,)).copy()
            self.got = (np.ctypeslib.as_array(ctypes.cast(pos, ctypes.POINTER(ctypes.c_int64)), (int(c.sum()),)).copy(), c)
            return 0
    g = hip.Genome.__new__(hip.Genome)
    g._h, g.offset, g.length, g._sites_of = 1, {'a': 0, 'b': 100}, {'a': 100, 'b': 50}, None
    fake, real = FakeLib(), hip.lib
    hip.lib = lambda: fake
    try:
        g.set_splice_sites(idx)
    finally:
        hip.lib = real
        g._h = None
    assert fake.got[0].tolist() == [103, 110, 100, 110, 7, 110] and fake.got[1].tolist() == [2, 2, 1, 1] and g._sites_of is idx


def test_chunk_programs_overlap_and_come_out_in_order():
    """find_bsj._drive: programs (generators that yield what they wait for) run with up to `depth` in flight, each resumed as soon as its handle is
    ready -- whatever the order the handles finish in -- and their results come out in submission order"""
    import random
    from ciri_long_amd import find_bsj

    class Handle(object):
        def __init__(self, clock, at, value):
            self.clock, self.at, self.value = clock, at, value

        def ready(self):
            return self.clock[0] >= self.at

        def wait(self, timeout=None):
            self.clock[0] += 1                    # (time passes only while the driver waits)

        def get(self):
            assert self.ready()
            return self.value

    rng = random.Random(3)
    clock = [0]
    live, peak, log = [0], [0], []

    def program(k):
        live[0] += 1; peak[0] = max(peak[0], live[0])
        total = 0
        for phase in range(3):
            got = yield Handle(clock, clock[0] + rng.randint(0, 9), (k, phase))
            assert got == (k, phase)
            log.append((k, phase, clock[0]))
            total += phase
        live[0] -= 1
        return k, total

    out = list(find_bsj._drive((program(k) for k in range(25)), 4))
    assert out == [(k, 3) for k in range(25)]                          # in order, every phase run
    assert peak[0] == 4                                                # never more than `depth` alive, and the depth is used
    order = [k for k, phase, _t in log if phase == 2]
    assert order != sorted(order)                                      # (programs did finish out of order: the test means something)
    assert list(find_bsj._drive(iter(()), 3)) == []
    # a program that never waits (the one-thread route) and depth 1
    def quick(k):
        return k
        yield                                                           # pragma: no cover
    assert list(find_bsj._drive((quick(k) for k in range(5)), 1)) == list(range(5))
    assert find_bsj.chunk_size_for(100000) == 4000 and find_bsj.chunk_size_for(4000) == 334 and find_bsj.chunk_size_for(10) == 250
    assert find_bsj.chunk_size_for(24, 1) == 2 and find_bsj.chunk_size_for(1000, 1) == 16
