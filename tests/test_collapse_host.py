"""Collapse-stage call sites of the kernels (ciri_long_amd/collapse.py) against golden vectors
produced by the reference's own Python (tests/golden/make_collapse_golden.py; see its header for the two absent
dependencies it had to supply).  CPU tests substitute the oracles for the batched GPU calls (test infrastructure only);
the `gpu` tests run the real K1/K1b/K3/K4 path."""
import gzip
import json
import os

import numpy as np
import pytest

import fake_mapper as fm
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def golden():
    with gzip.open(os.path.join(HERE, 'golden', 'collapse_golden.json.gz'), 'rt') as f:
        return json.load(f)


@pytest.fixture(scope='module')
def world():
    from ciri_long_amd import env
    w = fm.build_world()
    env.initializer(None, w['genome'].contig_len, w['genome'], w['gtf_index'], None, w['ss_index'])
    return w


class _Res(object):
    def __init__(self, d):
        self.__dict__.update(d)


def _oracle_pairs(refs, queries, match=2, mismatch=2, gap_open=3, gap_extend=1, **kw):
    return [_Res(oracle_lib.oracle_align(r, q, match, mismatch, gap_open, gap_extend)) for r, q in zip(refs, queries)]


@pytest.fixture()
def cpu_kernels(monkeypatch):
    """oracle stand-ins for the four batched GPU entry points the module uses"""
    from ciri_long_amd import collapse, spoa, ssw_wrap, utils

    def align_batch(self, queries, min_score=0, min_len=0):
        ref = ''.join('ACGTN'[c] for c in self.ref_seq)
        m = int(self.mat[0]); mm = -int(self.mat[1])
        return _oracle_pairs([ref] * len(queries), queries, m, mm, self.gap_open, self.gap_extend)

    def distance_batch(xs, ys):
        return np.array([oracle_lib.oracle_edit_distance(x, y) for x, y in zip(xs, ys)], dtype=np.int32)

    monkeypatch.setattr(collapse, 'align_pairs', _oracle_pairs)
    monkeypatch.setattr(ssw_wrap.Aligner, 'align_batch', align_batch)
    monkeypatch.setattr(collapse, 'distance_batch', distance_batch)
    monkeypatch.setattr(utils, 'distance_batch', distance_batch)
    monkeypatch.setattr(spoa, 'poa', lambda seqs, algorithm, genmsa, m, n, g, e, q, c: (oracle_lib.oracle_poa(list(seqs), algorithm, False, m, n, g, e, q, c), []))
    monkeypatch.setattr(collapse, 'consensus_of_groups', lambda groups: [oracle_lib.oracle_poa(list(g), 2, False, 10, -4, -8, -2, -24, -1) for g in groups])


def _check_all(golden):
    from collections import namedtuple
    from ciri_long_amd import collapse, utils
    from ciri_long_amd.ssw_wrap import Aligner
    for c in golden['cases']:
        ctg = c['ctg']
        # collapse.py:251-265
        head = collapse.head_positions(c['ref_seq'], c['reads'][1:])
        assert head == c['head_pos']
        template, junc_seqs = collapse.junction_windows(c['ref_seq'], c['reads'][1:], head)
        assert template == c['template'] and junc_seqs == c['junc_seqs']
        # collapse.py:161-173, 210-215
        scores = collapse.curate_junction(ctg, c['st'], c['en'], c['cs_junc'])
        assert len(scores) == c['n_scores']
        assert [s[2] for s in scores[:40]] == [s[2] for s in c['scores_head']]
        assert sorted(map(tuple, utils.min_sorted_items(scores, 2))) == sorted(map(tuple, c['best']))
        got = {(s[0], s[1]): s[2] for s in scores}
        for i, j, v in c['scores_head']:
            assert got[(i, j)] == v
        js = [float(collapse.junc_score(ctg, b, c['junc_seqs'])) for b in c['best'][:3]]
        assert js == c['junc_score']
        assert [float(x) for x in collapse.junc_scores(ctg, c['best'][:3], c['junc_seqs'])] == c['junc_score']
        # collapse.py:371-387
        assert collapse.genome_junction_seq(ctg, c['start'], c['end']) == c['circ_junc_seq']
        refined = collapse.refine_to_junction(c['circ_junc_seq'], [(rid, q) for rid, q, _ in c['refined']])
        assert [list(x) for x in refined] == [[rid, out] for rid, _, out in c['refined']]
        # collapse.py:419-506
        assert [utils.compress_seq(s) for _, s in c['cluster_input']] == c['hpc']
        assert utils.pairwise_distance(c['hpc']).tolist() == c['dist']
        res = collapse.batch_cluster_sequence('x', [tuple(x) for x in c['cluster_input']])
        assert [[s, list(ids)] for s, ids in res] == c['cluster_res']
    # every circRNA of the fixture in ONE lock-step run (one K4 and one K3 launch per round for all of them), the 120-read
    # circRNA that goes through the reference's rounds of 50 (collapse.py:439-455) among them
    jobs = [('c%d' % k, [tuple(x) for x in c['cluster_input']]) for k, c in enumerate(golden['cases'])]
    jobs.insert(2, ('big', [tuple(x) for x in golden['big']['cluster_input']]))
    res = collapse.batch_cluster_sequences(jobs)
    want = [c['cluster_res'] for c in golden['cases']]
    want.insert(2, golden['big']['cluster_res'])
    assert [[[s, list(ids)] for s, ids in r] for r in res] == want
    assert len(golden['big']['cluster_input']) > 100 and 2 <= len(golden['big']['cluster_res']) <= 6
    e = golden['exon']
    circ = namedtuple('Circ', 'contig start end strand')(e['contig'], e['start'], e['end'], e['strand'])
    aligner = Aligner(e['ref'], match=10, mismatch=4, gap_open=8, gap_extend=2)
    assert [int(x) for x in collapse.exon_scores(circ, aligner, [tuple(p) for p in e['pairs'][:3]])] == e['scores']
    assert int(collapse.exon_score(circ, aligner, *e['pairs'][1])) == e['scores'][1]


def test_collapse_call_sites_match_reference(golden, world, cpu_kernels):
    _check_all(golden)


@pytest.mark.gpu
def test_collapse_call_sites_on_gpu(golden, world):
    _check_all(golden)
