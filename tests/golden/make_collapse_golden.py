#!/usr/bin/env python3
"""Golden vectors for the collapse-stage call sites of the alignment kernels, from the REFERENCE's own Python
(CIRI_long/collapse.py, utils.py, align.py; libs/striped_smith_waterman/ssw_wrap.py over oracle/_ref/libssw.so).

Runs only in the build container.  The reference is imported from /root/reference where it lies (same arrangement as
make_bsj_golden.py: an empty `pysam` module object, ssw_wrap executed with its library path at oracle/_ref).  Two of its
dependencies cannot run here and are supplied for the duration of this script -- this is stated in the fixture:
  * `distance` (utils.py:153-159) needs python-Levenshtein/edlib: the name `distance` in the reference's collapse module
    is bound to the exact dynamic programme of oracle/edit_oracle.c (the integer is uniquely defined);
  * `spoa.poa` (imported inside cluster_sequence): a module object named `spoa` whose poa() is this project's own
    restatement of the published spoa algorithm (oracle/poa_oracle.c), called with the arguments the reference passes.  Consensus strings in the fixture are therefore NOT reference
    outputs (parity with spoa is unpinned); what the fixture pins around them is the reference's control flow
    (distance matrix, linkage order, 0.3 threshold, cluster membership, rounds of 50).

    python tests/golden/make_collapse_golden.py        -> tests/golden/collapse_golden.json.gz
"""
import gzip
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_bsj_golden as mb          # noqa: E402  (load_reference: pysam object, ssw_wrap over oracle/_ref)
import fake_mapper as fm              # noqa: E402
import oracle_lib                     # noqa: E402
from ciri_long_amd import synth       # noqa: E402  (seeded read simulator; data only)

BASES = 'ACGT'


def to_str(codes):
    return ''.join(BASES[c] for c in codes)


def noisy_rotations(rng, circ_seq, n, sub=0.03, ins=0.03, dele=0.03):
    codes = oracle_lib.encode(circ_seq)
    out = []
    for _ in range(n):
        k = int(rng.integers(0, len(codes)))
        out.append(to_str(synth.mutate(np.concatenate([codes[k:], codes[:k]]), rng, sub, ins, dele)))
    return out


def main():
    align, env, find_bsj = mb.load_reference()
    spoa_mod = types.ModuleType('spoa')
    spoa_mod.poa = lambda seqs, algorithm, genmsa, m, n, g, e, q, c: (oracle_lib.oracle_poa(list(seqs), algorithm, False, m, n, g, e, q, c), [])
    sys.modules['spoa'] = spoa_mod
    from CIRI_long import collapse, utils
    collapse.distance = lambda x, y: oracle_lib.oracle_edit_distance(x, y)
    from libs.striped_smith_waterman.ssw_wrap import Aligner

    world = fm.build_world()
    genome = world['genome']
    env.GENOME = genome
    env.CONTIG_LEN = genome.contig_len
    rng = np.random.Generator(np.random.PCG64(20210847))
    cases = []
    for ci, (ctg, exons, strand) in enumerate(world['circs'][:6]):
        start, end = exons[0][0], exons[-1][1]
        circ_seq = ''.join(genome.seq(ctg, s, e) for s, e in exons)
        reads = noisy_rotations(rng, circ_seq, 9 + ci)
        case = dict(ctg=ctg, start=start, end=end, strand=strand, reads=reads)

        # collapse.py:245-265 (inline in correct_cluster): head positions, template, junction windows
        ref_seq = sorted(reads, key=len, reverse=True)[0]
        ssw = Aligner(ref_seq[:50], match=10, mismatch=4, gap_open=8, gap_extend=2)
        head_pos = [ssw.align(q).ref_begin for q in reads[1:]]
        template = utils.transform_seq(ref_seq, max(head_pos))
        ssw = Aligner(template, match=10, mismatch=4, gap_open=8, gap_extend=2)
        junc_seqs = [utils.get_junc_seq(template, -max(head_pos) // 2, 25)]
        for q in reads[1:]:
            a = ssw.align(q)
            junc_seqs.append(utils.get_junc_seq(utils.transform_seq(q, a.query_begin), -max(head_pos) // 2, 25))
        case.update(ref_seq=ref_seq, head_pos=head_pos, template=template, junc_seqs=junc_seqs)

        # curate_junction / junc_score on a jittered set of candidate coordinates (collapse.py:161-173, 210-215)
        st = [start + int(d) for d in rng.integers(-3, 4, 4)]
        en = [end + int(d) for d in rng.integers(-3, 4, 4)]
        cs_junc = genome.seq(ctg, end - 22, end) + genome.seq(ctg, start, start + 24)     # a consensus-like junction
        scores = collapse.curate_junction(ctg, st, en, cs_junc)
        best = utils.min_sorted_items(scores, 2)
        case.update(st=st, en=en, cs_junc=cs_junc, n_scores=len(scores),
                    scores_head=[list(s) for s in scores[:40]], best=[list(s) for s in best],
                    junc_score=[float(collapse.junc_score(ctg, b, junc_seqs)) for b in best[:3]])

        # refined sequences (collapse.py:371-387)
        circ_junc_seq = collapse.genome_junction_seq(ctg, start, end)
        ssw = Aligner(circ_junc_seq, match=10, mismatch=4, gap_open=8, gap_extend=2, report_cigar=True)
        refined = []
        for k, q in enumerate(sorted(reads, key=len, reverse=True)):
            a = ssw.align(q * 2)
            pos = align.find_alignment_pos(a, len(circ_junc_seq) // 2)
            refined.append(['r%d' % k, q, q if pos is None else utils.transform_seq(q, pos % len(q))])
        case.update(circ_junc_seq=circ_junc_seq, refined=refined)

        # cluster_sequence (collapse.py:419-506): reads of this circle plus reads of a diverged variant
        variant = circ_seq[:len(circ_seq) // 3] + circ_seq[len(circ_seq) // 2:]
        mixed = [(rid, s) for rid, _, s in refined] + [('v%d' % k, s) for k, s in enumerate(noisy_rotations(rng, variant, 4))]
        res = collapse.batch_cluster_sequence('%s:%d-%d' % (ctg, start + 1, end), mixed)
        case.update(cluster_input=[list(x) for x in mixed], cluster_res=[[c, list(ids)] for c, ids in res])
        hpc = [utils.compress_seq(s) for _, s in mixed]
        case.update(hpc=hpc, dist=[[collapse.distance(x, y) / max(len(x), len(y)) for y in hpc] for x in hpc])
        cases.append(case)

    # a circRNA with 120 reads of three isoforms: the rounds of 50 of iter_cluster_sequence (collapse.py:439-455)
    ctg, exons, strand = world['circs'][7]
    circ_seq = ''.join(genome.seq(ctg, s, e) for s, e in exons)
    other = [''.join(genome.seq(c2, s, e) for s, e in ex2) for c2, ex2, _ in world['circs'][8:10]]
    iso = [circ_seq, circ_seq[:len(circ_seq) // 4] + other[0][:150], other[1][:260]]
    big_in = []
    for k in range(120):
        big_in.append(('b%03d' % k, noisy_rotations(rng, iso[k % 3], 1, 0.02, 0.02, 0.02)[0][:]))
    # reads of one molecule start at the junction after refinement (collapse.py:373-387): no rotation between them
    big_in = [(rid, to_str(synth.mutate(oracle_lib.encode(iso[k % 3]), rng, 0.02, 0.02, 0.02))) for k, (rid, _s) in enumerate(big_in)]
    big_res = collapse.batch_cluster_sequence('big', big_in)
    big = dict(cluster_input=[list(x) for x in big_in], cluster_res=[[c, list(ids)] for c, ids in big_res])

    # exon_score (collapse.py:760-774)
    circ = collapse.CIRC('chrA', 400, 2000, '-')
    aligner = Aligner(genome.seq('chrA', 350, 1500), match=10, mismatch=4, gap_open=8, gap_extend=2)
    pairs = [('st', '401-520'), ('401-520', '700-860'), ('700-860', 'en'), ('st', 'en')]
    exon = dict(contig='chrA', start=400, end=2000, strand='-', ref=genome.seq('chrA', 350, 1500), pairs=[list(p) for p in pairs],
                scores=[int(collapse.exon_score(circ, aligner, l, n)) for l, n in pairs[:3]])
    out = dict(note='distance() = exact DP (python-Levenshtein/edlib absent); poa() = own specification (spoa absent): '
                    'consensus strings are not reference outputs, everything else is', cases=cases, exon=exon, big=big)
    path = os.path.join(HERE, 'collapse_golden.json.gz')
    with gzip.open(path, 'wt') as f:
        json.dump(out, f)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(cases), 'cases')


if __name__ == '__main__':
    main()
