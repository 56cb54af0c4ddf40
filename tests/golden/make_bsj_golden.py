#!/usr/bin/env python3
"""Golden vectors for the per-read BSJ path from the REFERENCE's own Python (CIRI_long/find_bsj.py, align.py).

Runs only in the build container.  The reference modules are imported from /root/reference where they lie.  Two
adjustments make that possible without copying or editing them:
  * `pysam` is not installed; CIRI_long/align.py and find_bsj.py import it at module level but none of the functions
    exercised here touches it, so an empty module object named `pysam` is registered for the import to succeed;
  * `libs.striped_smith_waterman.ssw_wrap` loads libssw.so from its own directory (read-only here), so the module is
    executed with its library path pointed at oracle/_ref/libssw.so (the reference's ssw.c compiled by oracle/Makefile).
The mapper and the genome are the deterministic test doubles of tests/fake_mapper.py.

    PYTHONHASHSEED=0 python tests/golden/make_bsj_golden.py      (sort_ss breaks ties in set order)
Output: tests/golden/bsj_golden.json.gz
"""
import gzip
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('CIRI_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, REF)

import fake_mapper as fm  # noqa: E402


def load_reference():
    sys.modules.setdefault('pysam', types.ModuleType('pysam'))
    wrap_src = os.path.join(REF, 'libs', 'striped_smith_waterman', 'ssw_wrap.py')
    mod = types.ModuleType('libs.striped_smith_waterman.ssw_wrap')
    mod.__file__ = os.path.join(ROOT, 'oracle', '_ref', 'ssw_wrap.py')
    with open(wrap_src) as f:
        exec(compile(f.read(), wrap_src, 'exec'), mod.__dict__)
    pkg = types.ModuleType('libs'); pkg.__path__ = []
    sub = types.ModuleType('libs.striped_smith_waterman'); sub.__path__ = []
    sys.modules['libs'] = pkg
    sys.modules['libs.striped_smith_waterman'] = sub
    sys.modules['libs.striped_smith_waterman.ssw_wrap'] = mod
    from CIRI_long import align, env, find_bsj
    return align, env, find_bsj


def hit_dict(h):
    return dict(ctg=h.ctg, strand=h.strand, r_st=h.r_st, r_en=h.r_en, q_st=h.q_st, q_en=h.q_en, cigar=[list(c) for c in h.cigar],
                mlen=h.mlen, blen=h.blen, is_primary=h.is_primary)


class PlainHit(object):
    def __init__(self, d):
        self.__dict__.update(d)
        self.cigar = [tuple(c) for c in d['cigar']]


TIES = []


def watch_ties(align):
    """sort_ss (align.py:705-733) sorts a set: when several candidates share the best key the reference's pick follows
    string-hash order.  Observe (not change) every call and note whether its winner was tied."""
    from operator import itemgetter
    orig = align.sort_ss

    def spy(sites, us, ds, clip_base):
        res = orig(sites, us, ds, clip_base)
        uniq = set(sites)
        tiers = ((lambda s: -clip_base <= s[2] - s[3] <= clip_base, (6, 5, 4, 7)),
                 (lambda s: -us <= s[2] <= ds and -us <= s[3] <= ds, (5, 4, 6, 7)),
                 (lambda s: -clip_base <= s[2] <= 0 <= s[3] <= clip_base, (4, 5, 6, 7)),
                 (lambda s: True, (4, 5, 6, 7)))
        rest = uniq
        tied = False
        for accept, key in tiers:
            chosen = [s for s in rest if accept(s)]
            if chosen:
                k = itemgetter(*key)
                best = min(k(s) for s in chosen)
                tied = sum(1 for s in chosen if k(s) == best) > 1
                break
            rest = [s for s in rest if not accept(s)]
        TIES.append(tied)
        return res

    align.sort_ss = spy


def main():
    align, env, find_bsj = load_reference()
    watch_ties(align)
    world = fm.build_world()
    genome = world['genome']
    mapper = fm.FakeMapper(genome)
    env.initializer(mapper, genome.contig_len, genome, world['gtf_index'], None, world['ss_index'])
    reads = fm.build_reads(world, 64)
    out = {'reads': [list(r) for r in reads]}

    # chunk level (find_bsj.py:236-325, 375-448)
    cnt, short, ret = find_bsj.scan_ccs_chunk(reads, True)
    out['scan_ccs_chunk'] = dict(counters=dict(cnt), short=[list(s) for s in short], records=[list(r) for r in ret])
    cnt2, ret2 = find_bsj.recover_ccs_chunk(reads, True)
    out['recover_ccs_chunk'] = dict(counters=dict(cnt2), records=[list(r) for r in ret2])
    # which reads had a tied splice-site ranking (their record depends on PYTHONHASHSEED in the reference)
    tied_reads = []
    for r in reads:
        del TIES[:]
        find_bsj.recover_ccs_chunk([r], True)
        if any(TIES):
            tied_reads.append(r[0])
    out['tied_reads'] = tied_reads

    # function level
    rng = world['rng']
    unit = []
    for rid, seg, ccs, raw in reads:
        circ, junc = find_bsj.find_bsj(ccs)
        item = dict(ccs=ccs, find_bsj=[circ, junc])
        if circ is not None:
            hit = align.get_primary_alignment(mapper.map(circ))
            if hit is not None:
                item['circ_hit'] = hit_dict(hit)
                res = find_bsj.align_clip_segments(circ, hit)
                item['align_clip_segments'] = [res[0], res[1], res[2], list(res[3]) if res[3] is not None else None]
                item['get_blocks'] = align.get_blocks(hit)
        unit.append(item)
    out['per_read'] = unit

    sig = []
    for ctg, exons, strand in world['circs']:
        start, end = exons[0][0], exons[-1][1]
        for dj in (0, 1, -2, 5):
            for clip_base in (0, 3, 12):
                cs, ce = start + dj, end + (dj if dj > 0 else 0)
                del TIES[:]
                host = align.find_host_gene(ctg, cs, ce)
                a = align.find_annotated_signal(ctg, cs, ce, clip_base, clip_base + 10)
                d = None
                if a[0] is None:
                    d = align.find_denovo_signal(ctg, cs, ce, host, a[3], a[1], a[2], clip_base, clip_base + 10, 3, True)
                sig.append(dict(ctg=ctg, start=cs, end=ce, clip_base=clip_base, host=sorted(host) if host else None,
                                annotated=[list(a[0]) if a[0] else None, a[1], a[2], {k: [list(v[0]), list(v[1])] for k, v in a[3].items()}],
                                denovo=list(d) if d else None, tie=any(TIES)))
    out['signals'] = sig

    cig = []
    for _ in range(60):
        ops = []
        for _k in range(int(rng.integers(1, 9))):
            ops.append([int(rng.integers(1, 60)), int(rng.choice([0, 0, 0, 1, 2, 3]))])
        if rng.random() < 0.3:
            ops.insert(int(rng.integers(0, len(ops) + 1)), [int(rng.integers(21, 40)), 1])
        if rng.random() < 0.4:
            ops.insert(0, [int(rng.integers(1, 30)), 4])
        h = PlainHit(dict(ctg='chrA', strand=1, r_st=int(rng.integers(0, 5000)), q_st=ops[0][0] if ops[0][1] == 4 else 0,
                          cigar=ops, is_primary=1))
        sub = align.remove_long_insert(h)
        blocks = align.get_blocks(h)
        clip = [int(rng.integers(0, 6000)), 0, 3]
        clip[1] = clip[0] + int(rng.integers(5, 80))
        merged = align.merge_clip_exon([list(b) for b in blocks], clip) if blocks else None
        cig.append(dict(hit=dict(r_st=h.r_st, q_st=h.q_st, cigar=ops), sub=hit_dict(sub), blocks=blocks, clip=clip, merged=merged))
    out['cigar_helpers'] = cig

    with gzip.open(os.path.join(HERE, 'bsj_golden.json.gz'), 'wt') as f:
        json.dump(out, f)
    print('tied reads', len(tied_reads), 'tied signal cases', sum(1 for s in sig if s['tie']))
    print('records', len(ret), 'recover', len(ret2), 'counters', dict(cnt), 'mapper calls', mapper.calls)


if __name__ == '__main__':
    main()
