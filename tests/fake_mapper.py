"""Deterministic stand-in for mappy/bwapy in tests: local alignment of the query against every contig of a small
synthetic genome on both strands (via the CPU checkers in oracle/), returned with the attribute set CIRI-long reads
from a mapper hit (ctg, r_st, r_en, q_st, q_en, strand, cigar, mlen, blen, is_primary).

Test infrastructure: used identically when golden vectors are generated from the reference's Python
(tests/golden/make_bsj_golden.py) and when our modules are checked against them."""
import numpy as np

import oracle_lib

SCHEME = (2, 4, 4, 2)
_COMP = str.maketrans('ACGTacgt', 'TGCAtgca')


def rc(s):
    return s.translate(_COMP)[::-1]


class FakeHit(object):
    def __init__(self, ctg, strand, res, qlen):
        self.ctg = ctg
        self.strand = strand
        self.r_st = res['ref_begin']
        self.r_en = res['ref_end'] + 1
        if strand > 0:
            self.q_st, self.q_en = res['query_begin'], res['query_end'] + 1
        else:   # coordinates on the original read, as mappy reports them
            self.q_st, self.q_en = qlen - (res['query_end'] + 1), qlen - res['query_begin']
        self.cigar = [(int(c) >> 4, int(c) & 0xf) for c in res['cigar']]
        self.mlen = sum(n for n, op in self.cigar if op == 0)
        self.blen = sum(n for n, op in self.cigar if op in (0, 1, 2))
        self.is_primary = 0
        self.score = res['score']


class FakeGenome(object):
    def __init__(self, contigs):
        self.genome = dict(contigs)
        self.contig_len = {k: len(v) for k, v in self.genome.items()}

    def seq(self, ctg, start, end):
        g = self.genome.get(ctg)
        return None if g is None else g[max(start, 0):end]


class FakeMapper(object):
    def __init__(self, genome, min_score=40):
        self.genome = genome
        self.min_score = min_score
        self._align = oracle_lib.ref_align if oracle_lib.have_ref() else oracle_lib.oracle_align
        self._codes = {k: oracle_lib.encode(v) for k, v in genome.genome.items()}
        self.calls = 0

    def map(self, seq):
        self.calls += 1
        if not seq:
            return None
        hits = []
        for ctg in sorted(self._codes):
            for strand, q in ((1, seq), (-1, rc(seq))):
                res = self._align(self._codes[ctg], q, *SCHEME)
                if res is not None and res['score'] >= self.min_score and res['cigar']:
                    hits.append(FakeHit(ctg, strand, res, len(seq)))
        if not hits:
            return None
        hits.sort(key=lambda h: -h.score)       # stable: contig name, then + before -
        hits[0].is_primary = 1
        return hits


class Element(object):
    """GTF record as the reference's index stores it (align.py:48-70: contig, type, start, end, strand)."""
    def __init__(self, contig, type_, start, end, strand):
        self.contig, self.type, self.start, self.end, self.strand = contig, type_, start, end, strand


def build_world(seed=20210846, n_circ=24):
    """Synthetic genome with planted circRNAs: exons flanked by AG...GT, annotation for two thirds of them.
    Returns dict(genome, ss_index, gtf_index, circs=[(ctg, [(start, end), ...], strand)])."""
    rng = np.random.default_rng(seed)

    def rnd(n):
        return ''.join('ACGT'[i] for i in rng.integers(0, 4, n))

    contigs = {}
    circs = []
    ss_index, gtf_index = {}, {}
    for ctg, size in (('chrA', 12000), ('chrB', 7000), ('chrC', 1800)):
        g = list(rnd(size))
        pos = 300
        while pos + 900 < size and len(circs) < n_circ:
            length = int(rng.integers(120, 520))
            start, end = pos, pos + length          # 0-based [start, end)
            strand = '+' if rng.random() < 0.6 else '-'
            # canonical signal: acceptor AG before the exon, donor GT after it (reverse-complemented on '-')
            if strand == '+':
                g[start - 2:start] = 'AG'; g[end:end + 2] = 'GT'
            else:
                g[start - 2:start] = 'AC'; g[end:end + 2] = 'CT'
            exons = [(start, end)]
            if rng.random() < 0.4 and end + 900 < size:
                # second exon behind a 150-500 nt intron: the (unspliced) mapper aligns the longer exon and leaves the
                # other one clipped, which is what align_clip_segments re-aligns by Smith-Waterman
                gap = int(rng.integers(150, 500))
                l2 = int(rng.integers(30, max(31, int(0.55 * length))))
                s2 = end + gap
                e2 = s2 + l2
                if strand == '+':
                    g[s2 - 2:s2] = 'AG'; g[e2:e2 + 2] = 'GT'
                else:
                    g[s2 - 2:s2] = 'AC'; g[e2:e2 + 2] = 'CT'
                exons.append((s2, e2))
                end = e2
            circs.append((ctg, exons, strand))
            start = exons[0][0]
            if rng.random() < 0.66:
                d = ss_index.setdefault(ctg, {})
                d.setdefault(start + 1, {}).setdefault(strand, {})['start'] = 1
                d.setdefault(end, {}).setdefault(strand, {})['end'] = 1
                el = Element(ctg, 'exon', start + 1, end, strand)
                for b in range((start + 1) // 500, end // 500 + 1):
                    gtf_index.setdefault(ctg, {}).setdefault(b, []).append(el)
            pos = end + int(rng.integers(250, 700))
        if ctg == 'chrB':
            g[6200:6600] = 'N' * 400
        if ctg == 'chrC':
            g[1100:1750] = 'N' * 650           # > 30 % of any window on this contig: clip alignment is refused
        contigs[ctg] = ''.join(g)
    return dict(genome=FakeGenome(contigs), ss_index=ss_index, gtf_index=gtf_index, circs=circs, rng=rng)


def mutate(s, rng, p=0.06):
    out = []
    for c in s:
        u = rng.random()
        if u < p / 3:
            continue
        if u < 2 * p / 3:
            out.append('ACGT'[rng.integers(4)]); continue
        out.append(c)
        if u < p:
            out.append('ACGT'[rng.integers(4)])
    return ''.join(out)


def build_reads(world, n_reads=40):
    """(read_id, segments, ccs, raw) tuples as find_ccs produces them: noisy rolling-circle reads of the planted
    circRNAs (some with a junction offset that leaves >= 20 clipped bases), plus linear and unmappable negatives."""
    rng = world['rng']
    genome = world['genome']
    reads = []
    for k in range(n_reads):
        kind = rng.random()
        if kind < 0.75:
            ctg, exons, strand = world['circs'][int(rng.integers(len(world['circs'])))]
            circ = ''.join(genome.genome[ctg][a:b] for a, b in exons)
            if rng.random() < 0.5:
                circ = rc(circ)
            phase = int(rng.integers(len(circ)))
            unit = circ[phase:] + circ[:phase]
            copies = int(rng.integers(2, 5))
            raw = mutate(unit * copies + unit[:int(rng.integers(0, len(unit)))], rng)
            ccs = mutate(unit, rng, 0.02)
            seg = ';'.join('{}-{}'.format(i * len(unit), (i + 1) * len(unit)) for i in range(copies))
        elif kind < 0.9:
            ctg = 'chrA'
            st = int(rng.integers(0, 7000))
            raw = mutate(genome.genome[ctg][st:st + 900], rng)       # linear read: must be filtered out
            ccs = raw[:300]
            seg = '0-300;300-600;600-900'
        else:
            ccs = ''.join('ACGT'[i] for i in rng.integers(0, 4, int(rng.integers(60, 200))))   # unmappable
            raw = ccs * 3
            seg = '0-{0};{0}-{1}'.format(len(ccs), 2 * len(ccs))
        reads.append(('read%03d' % k, seg, ccs, raw))
    return reads
