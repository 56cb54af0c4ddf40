"""K5 (genome resident in HBM) and the window entry points: identical to the packed-reference path on the window strings
built the reference's way (find_bsj.py:196-201,214: slice, Counter(...)['N'], utils.revcomp -- which complements upper
case only -- then the case-folding encoder of ssw_wrap.py:243-250)."""
import gzip
import json
import os

import numpy as np
import pytest

import fake_mapper as fm
import oracle_lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _messy_contigs(rng):
    def rnd(n, alphabet):
        return ''.join(alphabet[i] for i in rng.integers(0, len(alphabet), n))
    a = rnd(5000, 'ACGT')
    b = rnd(3000, 'ACGT')
    b = b[:700] + 'N' * 400 + b[1100:1500] + rnd(300, 'acgt') + b[1800:2200] + rnd(200, 'ACGTNacgtnRYK-*') + b[2400:]
    c = rnd(333, 'ACGTacgtN')
    return {'chr1': a, 'chr2': b, 'chrUn': c}


def test_windows_equal_packed_references_built_the_reference_way():
    from ciri_long_amd import hip, ssw_wrap, utils
    rng = np.random.default_rng(99)
    contigs = _messy_contigs(rng)
    ctx = hip.default_context()
    g = hip.Genome(ctx, contigs)
    wins, minus, queries, strings = [], [], [], []
    names = list(contigs)
    for k in range(160):
        ctg = names[int(rng.integers(0, 3))]
        L = len(contigs[ctg])
        s = int(rng.integers(0, L - 30)); e = int(rng.integers(s + 25, min(L, s + 2500) + 1))
        if k < 6:
            s, e = (0, L) if k % 2 else (max(0, L - 257), L)            # whole contig / block-edge windows
        rc = bool(rng.integers(0, 2))
        w = contigs[ctg][s:e]
        qs = int(rng.integers(0, max(1, len(w) - 20))); q = w[qs:qs + int(rng.integers(20, 300))].upper().replace('N', 'A')
        q = ''.join(ch if ch in 'ACGT' else 'C' for ch in q)
        if rc:
            q = utils.revcomp(q)
        wins.append((ctg, s, e)); minus.append(rc); queries.append(q)
        strings.append(utils.revcomp(w) if rc else w)
    # N counts: Counter(window)['N'] counts upper-case N only
    assert g.count_n(wins).tolist() == [contigs[c][s:e].count('N') for c, s, e in wins]
    for scoring in ((1, 1, 1, 1), (10, 4, 8, 2)):
        got = ssw_wrap.align_windows(g, wins, minus, queries, *scoring, report_secondary=True, report_cigar=True)
        want = ssw_wrap.align_pairs(strings, queries, *scoring, report_secondary=True, report_cigar=True)
        for a, b, q, s_ in zip(got, want, queries, strings):
            assert (a.score, a.ref_begin, a.ref_end, a.query_begin, a.query_end, a.score2, a.ref_end2, a.cigar_string) == \
                (b.score, b.ref_begin, b.ref_end, b.query_begin, b.query_end, b.score2, b.ref_end2, b.cigar_string)
        o = oracle_lib.oracle_align(strings[7], queries[7], *scoring)
        assert (got[7].score, got[7].ref_begin, got[7].ref_end, got[7].cigar_string) == (o['score'], o['ref_begin'], o['ref_end'], o['cigar_string'])
    g.close()


def test_scan_ccs_chunk_with_resident_genome_matches_reference_golden():
    """the BSJ step end to end with env.GENOME = DeviceGenome: same records as the reference's own Python"""
    from ciri_long_amd import align, env, find_bsj
    with gzip.open(os.path.join(HERE, 'golden', 'bsj_golden.json.gz'), 'rt') as f:
        golden = json.load(f)
    w = fm.build_world()
    mapper = fm.FakeMapper(w['genome'])
    dg = align.DeviceGenome(w['genome'], w['genome'].genome)
    env.initializer(mapper, w['genome'].contig_len, dg, w['gtf_index'], None, w['ss_index'])
    reads = fm.build_reads(w, 64)
    cnt, short, ret = find_bsj.scan_ccs_chunk(reads, True)
    g = golden['scan_ccs_chunk']
    tied = set(golden.get('tied_reads', []))
    got = {r[0]: json.loads(json.dumps(list(r))) for r in ret}
    want = {r[0]: r for r in g['records']}
    assert dict(cnt) == g['counters']
    assert sorted(got) == sorted(want)
    for rid in want:
        if rid not in tied:
            assert got[rid] == want[rid], rid
    env.initializer(mapper, w['genome'].contig_len, w['genome'], w['gtf_index'], None, w['ss_index'])
    cnt2, short2, ret2 = find_bsj.scan_ccs_chunk(reads, True)
    assert [list(r) for r in ret] == [list(r) for r in ret2] and dict(cnt) == dict(cnt2)


def test_stage_drivers_write_the_reference_records(tmp_path):
    """scan_ccs_reads / recover_ccs_reads (find_bsj.py:328-372, 451-490) with the mapper double: 16 chunks per GPU call,
    genome wrapped into a resident copy automatically; cand_circ.fa holds the reference's records in input order."""
    from ciri_long_amd import find_bsj
    with gzip.open(os.path.join(HERE, 'golden', 'bsj_golden.json.gz'), 'rt') as f:
        golden = json.load(f)
    w = fm.build_world()
    mapper = fm.FakeMapper(w['genome'])
    reads = fm.build_reads(w, 64)
    ccs_seq = {r[0]: [r[1], r[2], r[3]] for r in reads}
    cnt, short = find_bsj.scan_ccs_reads(ccs_seq, None, w['ss_index'], w['gtf_index'], None, True, str(tmp_path), 'p', 1,
                                         aligner=mapper, genome=w['genome'], contig_len=w['genome'].contig_len)
    from ciri_long_amd import env
    assert getattr(env.GENOME, 'device', None) is not None                # the windows came from HBM
    g = golden['scan_ccs_chunk']
    assert dict(cnt) == g['counters'] and json.loads(json.dumps([list(x) for x in short])) == g['short']
    tied = set(golden['tied_reads'])
    lines = (tmp_path / 'p.cand_circ.fa').read_text().split('\n')
    want = [r for r in g['records']]
    assert len(lines) == 2 * len(want) + 1
    for k, r in enumerate(want):
        if r[0] not in tied:
            assert lines[2 * k] == '>{}\t{}\t{}\t{}\t{}\t{}\t{}'.format(*r[:7]) and lines[2 * k + 1] == r[7], r[0]
    cnt2 = find_bsj.recover_ccs_reads(reads, None, w['ss_index'], w['gtf_index'], None, True, str(tmp_path), 'p', 1,
                                      aligner=mapper, genome=w['genome'])
    assert dict(cnt2) == golden['recover_ccs_chunk']['counters']
    lines2 = (tmp_path / 'p.cand_circ.fa').read_text().split('\n')
    assert len(lines2) == len(lines) + 2 * len(golden['recover_ccs_chunk']['records'])
