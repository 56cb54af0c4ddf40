"""oracle/splice_oracle.c (the CPU checker of K6) pinned to outputs of the reference itself, and compared with the Python
mirror of the same step on seeded worlds.  CPU only."""
import gzip
import json
import os

import numpy as np
import pytest

import fake_mapper as fm
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))


def _runs(ss_index, ctg, length):
    from ciri_long_amd import hip
    return hip.flatten_splice_sites({ctg: ss_index[ctg]} if ss_index and ctg in ss_index else None, {ctg: 0}, {ctg: length})


def test_oracle_gives_the_references_answers():
    """All splice-signal cases of the reference-made fixture (tests/golden/make_bsj_golden.py): annotated pairs and
    de-novo searches; cases whose ranking is tied in the reference's set order are compared up to the tie."""
    with gzip.open(os.path.join(HERE, 'golden', 'bsj_golden.json.gz'), 'rt') as f:
        golden = json.load(f)
    world = fm.build_world()
    g = world['genome']
    n_anno = n_denovo = n_none = n_edge = 0
    for s in golden['signals']:
        ctg = s['ctg']
        got = oracle_lib.oracle_splice_signal(g.genome[ctg], s['start'], s['end'], s['clip_base'], s['host'], True,
                                              _runs(world['ss_index'], ctg, g.contig_len[ctg]))
        if got == 'edge':
            n_edge += 1
            continue
        site, us_free, ds_free = got
        assert [us_free, ds_free] == s['annotated'][1:3], s
        want = s['annotated'][0] if s['annotated'][0] is not None else s['denovo']
        assert (site is None) == (want is None), s
        if s['tie']:
            continue
        assert (list(site) if site else None) == want, s
        n_anno += s['annotated'][0] is not None
        n_denovo += s['annotated'][0] is None and want is not None
        n_none += want is None
    assert n_anno >= 50 and n_denovo >= 100 and n_edge == 0


@pytest.mark.parametrize('is_canonical,annotated', [(True, False), (False, False), (True, True), (False, True)])
def test_oracle_equals_python_mirror(is_canonical, annotated):
    """Seeded worlds (soft-masked and N runs, IUPAC characters, shared flanks, contig ends, planted signals, annotation
    around the ends): the C statement and the Python mirror agree on every candidate, those next to a contig end included
    (there the annotation is not consulted, align.py:495-496, and the search runs on what Python's slices return)."""
    import test_gpu_splice as tgs
    from ciri_long_amd import align, env
    contigs, cands = tgs._world(300 + is_canonical + 2 * annotated, 1600)
    host = tgs._Genome(contigs)
    ss_index = tgs._annotation(contigs, cands, 41) if annotated else None
    env.initializer(None, host.contig_len, host, None, None, ss_index)
    runs = {c: _runs(ss_index, c, host.contig_len[c]) for c in contigs}
    n_ok = n_edge = n_found = 0
    for cand in cands:
        ctg, st, en, cb, hs = cand
        got = oracle_lib.oracle_splice_signal(contigs[ctg], st, en, cb, hs, is_canonical, runs[ctg])
        want = tgs._host_answer(align, cand, is_canonical)
        L = host.contig_len[ctg]
        n_edge += st - (cb + 10) - want[1] - 2 < 0 or en + (cb + 10) + want[2] + 2 > L
        assert got == want, (cand, got, want)
        n_ok += 1
        n_found += got[0] is not None
    assert n_ok == len(cands) and n_found > 0.4 * n_ok and n_edge > 20


def test_reference_answers_on_seeded_worlds():
    """6 000 candidates answered by the REFERENCE's own find_annotated_signal / find_denovo_signal
    (tests/golden/make_splice_golden.py): canonical-only and all five motif classes, with and without annotated sites.
    The C oracle and the Python mirror must give the same answers, next to the contig ends as well; where the reference's
    pick was tied in its set order, presence and the free-sliding lengths are compared."""
    import test_gpu_splice as tgs
    from ciri_long_amd import align, env
    with gzip.open(os.path.join(HERE, 'golden', 'splice_golden.json.gz'), 'rt') as f:
        golden = json.load(f)
    n_c = n_py = n_edge = n_index = n_index_none = 0
    for cfg in golden:
        contigs, cands = tgs._world(cfg['seed'], cfg['n'])
        host = tgs._Genome(contigs)
        index = bool(cfg.get('index'))
        if index:        # env.GENOME as the reference's main pass has it: the sequences as a minimap2 index serves them (align.IndexGenome)
            host = align.IndexGenome(host)
            contigs = host.genome
        ss_index = tgs._annotation(contigs, cands, cfg['seed'] + 1) if cfg['annotated'] else None
        env.initializer(None, host.contig_len, host, None, None, ss_index)
        runs = {c: _runs(ss_index, c, host.contig_len[c]) for c in contigs}
        assert len(cands) == len(cfg['rows'])
        for cand, (want_site, us_free, ds_free, tied) in zip(cands, cfg['rows']):
            ctg, st, en, cb, hs = cand
            mirror = tgs._host_answer(align, cand, cfg['canonical'])
            assert [mirror[1], mirror[2]] == [us_free, ds_free] and (mirror[0] is None) == (want_site is None), cand
            if not tied:
                assert (list(mirror[0]) if mirror[0] else None) == want_site, (cand, mirror, want_site)
                n_py += 1
            got = oracle_lib.oracle_splice_signal(contigs[ctg], st, en, cb, hs, cfg['canonical'], runs[ctg], index_slices=index)
            assert got != 'edge', cand
            n_edge += st - (cb + 10) - us_free - 2 < 0 or en + (cb + 10) + ds_free + 2 > host.contig_len[ctg]
            n_index += index
            n_index_none += index and st - (cb + 10) - us_free - 2 < 0 and want_site is None
            assert [got[1], got[2]] == [us_free, ds_free] and (got[0] is None) == (want_site is None), cand
            if not tied:
                assert (list(got[0]) if got[0] else None) == want_site, (cand, got, want_site)
                n_c += 1
    assert n_c > 8000 and n_py > 8000 and 100 < n_edge < 1400
    assert n_index == 3000 and n_index_none > 20        # no sequence in front of a contig: no signal, where Python's slice would wrap around
