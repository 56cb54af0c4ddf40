"""Collapse-stage call sites of the alignment kernels, batched.

The reference constructs one `Aligner` per alignment and calls python-Levenshtein/edlib per pair
(CIRI_long/collapse.py:156-173, 210-215, 251-265, 373-387, 458-506, 760-774).  The functions here keep the reference's
names, arguments and return values, but hand every group of independent alignments / distances to the GPU as ONE batch
(K1/K1b through `ssw_wrap.align_pairs` / `Aligner.align_batch`, K4 through `utils.distance_batch`, K3 through
`spoa.poa`).  Scoring of this stage is (10, 4, 8, 2).  Clustering itself (scipy linkage) and everything around these call
sites (annotation look-ups, splice signals, max-flow over exons, output tables) is unchanged host code of the reference
and is not restated here.
"""
import logging
from operator import itemgetter

import numpy as np

from . import env
from .align import find_alignment_pos
from .ssw_wrap import Aligner, align_pairs
from .utils import (compress_seq, distance_batch, flatten, get_junc_seq, grouper, pairwise, pairwise_distance, revcomp,
                    transform_seq)

LOGGER = logging.getLogger('CIRI-long')
SCORING = dict(match=10, mismatch=4, gap_open=8, gap_extend=2)


def genome_junction_seq(contig, start, end, width=25):
    """collapse.py:152-153"""
    return env.GENOME.seq(contig, end - width, end) + env.GENOME.seq(contig, start, start + width)


def avg_score(alignment, ref, query):
    """collapse.py:156-158 (one pair; curate_junction below batches it)"""
    from .utils import distance
    x = query[alignment.query_begin:alignment.query_end]
    return distance(ref, x) / len(ref)


def curate_junction(ctg, st, en, junc):
    """collapse.py:161-173: every (i, j) of the +-25 grid around the cluster's start/end positions -> 20-nt genomic
    junction, Smith-Waterman against the consensus junction, edit distance of the aligned part.  The reference builds
    ~2 500-10 000 Aligner objects here; this is one K1 batch plus one K4 batch."""
    grid, refs = [], []
    for i in range(max(0, min(st) - 25), max(st) + 25):
        for j in range(min(en) - 25, min(max(en) + 25, env.CONTIG_LEN[ctg])):
            if j <= i:
                continue
            grid.append((i, j))
            refs.append(genome_junction_seq(ctg, i, j, width=10))
    if not grid:
        return []
    alns = align_pairs(refs, [junc] * len(refs), **SCORING)
    parts = [junc[a.query_begin:a.query_end] for a in alns]
    dist = distance_batch(refs, parts)
    junc_scores = [(i, j, int(d) / len(r)) for (i, j), r, d in zip(grid, refs, dist)]
    return sorted(junc_scores, key=itemgetter(2))


def junc_score(ctg, junc, junc_seqs):
    """collapse.py:210-215: mean score of the reads' junction windows against the doubled circle"""
    aligner = Aligner(env.GENOME.seq(ctg, junc[0], junc[1]) * 2, **SCORING)
    return np.mean([a.score for a in aligner.align_batch(junc_seqs)])


def junc_scores(ctg, juncs, junc_seqs):
    """junc_score for several candidate junctions at once (the sort keys of collapse.py:283-288): one batch"""
    refs, queries = [], []
    for junc in juncs:
        circle = env.GENOME.seq(ctg, junc[0], junc[1]) * 2
        refs += [circle] * len(junc_seqs)
        queries += list(junc_seqs)
    alns = align_pairs(refs, queries, **SCORING)
    n = len(junc_seqs)
    return [np.mean([a.score for a in alns[k * n:(k + 1) * n]]) for k in range(len(juncs))]


def head_positions(ref_seq, queries):
    """collapse.py:251-256: where the first 50 bases of the longest full-length read start in every other read"""
    return [a.ref_begin for a in Aligner(ref_seq[:50], **SCORING).align_batch(queries)]


def junction_windows(ref_seq, queries, head_pos):
    """collapse.py:258-265: rotate the template to the largest head position, align every read to it, rotate the reads
    to where their alignment starts, cut the +-25 window around the junction"""
    shift = max(head_pos)
    template = transform_seq(ref_seq, shift)
    junc_seqs = [get_junc_seq(template, -shift // 2, 25), ]
    for query, a in zip(queries, Aligner(template, **SCORING).align_batch(queries)):
        junc_seqs.append(get_junc_seq(transform_seq(query, a.query_begin), -shift // 2, 25))
    return template, junc_seqs


def refine_to_junction(circ_junc_seq, reads):
    """collapse.py:373-387: align every doubled read to the 50-nt genomic junction with CIGAR, rotate the read to the
    junction position.  reads: [(read_id, seq)] -> [(read_id, rotated seq)]"""
    ssw = Aligner(circ_junc_seq, report_cigar=True, **SCORING)
    out = []
    for (read_id, seq), a in zip(reads, ssw.align_batch([s * 2 for _, s in reads])):
        pos = find_alignment_pos(a, len(circ_junc_seq) // 2)
        out.append((read_id, seq if pos is None else transform_seq(seq, pos % len(seq))))
    return out


def cluster_sequence(hpc_freq, sequence):
    """collapse.py:458-506.  hpc_freq: [(homopolymer-compressed sequence, [read ids])]; sequence: {read id: sequence}.
    The distance matrix is one K4 batch, the consensus of every multi-member cluster one K3 batch."""
    from scipy.cluster.hierarchy import linkage, leaves_list
    from scipy.spatial.distance import squareform
    from .spoa import poa

    if len(hpc_freq) == 1:
        return hpc_freq
    dist = pairwise_distance([h for h, _ in hpc_freq])
    if dist.sum() != 0:
        z = leaves_list(linkage(squareform(dist), "ward", optimal_ordering=True))
    else:
        z = list(range(len(hpc_freq)))
    clusters = [[z[0], ]]
    for i, j in pairwise(z):
        d = dist[j][i] if i > j else dist[i][j]
        if d < 0.3:
            clusters[-1].append(j)
        else:
            clusters.append([j, ])
    ccs_seq = []
    for cluster in clusters:
        if len(cluster) == 1:
            ccs_seq.append((hpc_freq[cluster[0]]))
            continue
        cluster_reads = flatten([hpc_freq[i][1] for i in cluster])
        ccs, _ = poa([sequence[i] for i in cluster_reads], 2, False, 10, -4, -8, -2, -24, -1)
        ccs_seq.append((ccs, cluster_reads))
    return ccs_seq


def iter_cluster_sequence(circ_id, hpc_freq, sequence):
    """collapse.py:439-455: rounds of at most 50 sequences"""
    if len(hpc_freq) <= 50:
        return cluster_sequence(hpc_freq, sequence)
    res = []
    for tmp in grouper(hpc_freq, 50):
        chunk = [i for i in tmp if i is not None]
        res = cluster_sequence(chunk + res, sequence)
        for _ in range(10):
            n_res = cluster_sequence(res, sequence)
            if len(n_res) == len(res):
                break
            res = n_res
        else:
            LOGGER.warning('Sequence not consensus for circRNA: {}'.format(circ_id))
    return res


def batch_cluster_sequence(circ_id, x):
    """collapse.py:419-436.  x: [(read_id, sequence)]"""
    sequence = {}
    hpc_freq = []
    for read_id, read_seq in x:
        sequence[read_id] = read_seq
        hpc_freq.append((compress_seq(read_seq), [read_id, ]))
    res = iter_cluster_sequence(circ_id, hpc_freq, sequence)
    for _ in range(10):
        n_res = cluster_sequence(res, sequence)
        if len(n_res) == len(res):
            break
        res = n_res
    else:
        LOGGER.warning('Sequence not consensus for circRNA: {}'.format(circ_id))
    return res


def exon_score(circ, aligner, l_exon, n_exon):
    """collapse.py:760-774: aligned reference span of two neighbouring exons' genomic sequence in the isoform consensus"""
    return exon_scores(circ, aligner, [(l_exon, n_exon)])[0]


def exon_scores(circ, aligner, exon_pairs):
    """exon_score for several (l_exon, n_exon) candidates against one aligner (the lists of collapse.py:750,755)"""
    queries = []
    for l_exon, n_exon in exon_pairs:
        query_seq = ''
        if l_exon != 'st':
            l_st, l_en = l_exon.split('-')
            query_seq += env.GENOME.seq(circ.contig, int(l_st) - 1, int(l_en))
        if n_exon != 'en':
            n_st, n_en = n_exon.split('-')
            query_seq += env.GENOME.seq(circ.contig, int(n_st), int(n_en))
        if circ.strand == '-':
            query_seq = revcomp(query_seq)
        queries.append(query_seq)
    return [a.ref_end - a.ref_begin for a in aligner.align_batch(queries)]
