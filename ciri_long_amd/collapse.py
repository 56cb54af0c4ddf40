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
from .utils import compress_seq, distance_batch, flatten, get_junc_seq, pairwise, revcomp, transform_seq

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


# ---------------------------------------------------------------------------------------------------------------
# isoform clustering of a circRNA's reads (collapse.py:419-506), for MANY circRNAs at once
# ---------------------------------------------------------------------------------------------------------------
# The reference clusters one circRNA at a time: a distance matrix (python-Levenshtein per pair), a scipy linkage, one spoa call
# per cluster with several members; repeated in rounds of 50 sequences until the number of clusters settles.  Here every circRNA
# of a chunk is a coroutine that asks for "one clustering step of this list" whenever the reference calls cluster_sequence, and
# `cluster_steps` answers the requests of ALL coroutines of a round together: one K4 launch for every pair of every request, the
# linkages on the host, one K3 launch for every cluster with several members.  Per circRNA the sequence of steps -- and so the
# result -- is the reference's.
def cluster_steps(requests):
    """[(hpc_freq, sequence)] -> [clustered hpc_freq], each what collapse.py:458-506 returns for that request: hpc_freq =
    [(homopolymer-compressed sequence, [read ids])], sequence = {read id: sequence}.  All requests share the GPU calls."""
    from scipy.cluster.hierarchy import linkage, leaves_list
    from scipy.spatial.distance import squareform
    out = [None] * len(requests)
    # 1. every pair of every request with more than one entry: one batch of edit distances (K4)
    xs, ys, spans = [], [], []
    for k, (hpc_freq, _seq) in enumerate(requests):
        n = len(hpc_freq)
        if n == 1:
            out[k] = hpc_freq
            continue
        ii, jj = np.triu_indices(n, 1)
        spans.append((k, n, ii, jj, len(xs)))
        xs += [hpc_freq[i][0] for i in ii]
        ys += [hpc_freq[j][0] for j in jj]
    dists = distance_batch(xs, ys) if xs else []
    # 2. per request: the matrix, the leaf order of the ward linkage, neighbours in that order closer than 0.3 share a cluster
    groups, owners = [], []                       # sequences of every cluster with several members; (request, slot in its result)
    for k, n, ii, jj, at in spans:
        hpc_freq, sequence = requests[k]
        dist = np.zeros((n, n))
        norm = np.array([max(len(hpc_freq[i][0]), len(hpc_freq[j][0])) for i, j in zip(ii, jj)], dtype=np.float64)
        dist[ii, jj] = np.asarray(dists[at:at + len(ii)], dtype=np.float64) / norm
        dist = dist + dist.T
        order = leaves_list(linkage(squareform(dist), "ward", optimal_ordering=True)) if dist.sum() != 0 else list(range(n))
        clusters = [[order[0]]]
        for a, b in pairwise(order):
            if dist[min(a, b)][max(a, b)] < 0.3:
                clusters[-1].append(b)
            else:
                clusters.append([b])
        res = []
        for members in clusters:
            if len(members) == 1:
                res.append(hpc_freq[members[0]])
                continue
            reads = flatten([hpc_freq[i][1] for i in members])
            owners.append((k, len(res), reads))
            groups.append([sequence[r] for r in reads])
            res.append(None)
        out[k] = res
    # 3. the consensus of every such cluster: one batch of partial-order alignments (K3), spoa.poa(seqs, 2, False, 10, -4, -8, -2, -24, -1)
    if groups:
        for (k, slot, reads), c in zip(owners, consensus_of_groups(groups)):
            out[k][slot] = (c, reads)
    return out


def consensus_of_groups(groups):
    """[[sequence]] -> [consensus]: spoa.poa(group, 2, False, 10, -4, -8, -2, -24, -1)[0] (collapse.py:504) for every group, one K3 launch"""
    from . import hip
    flat = [s for g in groups for s in g]
    data, off = hip.pack_raw(flat)
    goff = np.cumsum([0] + [len(g) for g in groups]).astype(np.int64)
    return hip.default_context().poa_batch(data, off, goff, algorithm=2, scores=(10, -4, -8, -2, -24, -1), raw=True)


def _cluster_job(circ_id, reads):
    """The steps of collapse.py:419-455 for one circRNA as a coroutine: yields (hpc_freq, sequence) where the reference calls
    cluster_sequence, receives the clustered list; returns the final list (StopIteration.value)."""
    sequence = {read_id: seq for read_id, seq in reads}
    hpc_freq = [(compress_seq(seq), [read_id]) for read_id, seq in reads]

    def settle(res):
        # until a step leaves the number of clusters as it is, at most 10 steps (collapse.py:428-435, 447-454); the list of the
        # step BEFORE the unchanged one is kept, as the reference keeps it
        for _ in range(10):
            nxt = yield (res, sequence)
            if len(nxt) == len(res):
                return res
            res = nxt
        LOGGER.warning('Sequence not consensus for circRNA: {}'.format(circ_id))
        return res

    if len(hpc_freq) <= 50:
        res = yield (hpc_freq, sequence)
    else:
        res = []
        for k in range(0, len(hpc_freq), 50):                # rounds of at most 50 new sequences joined with what the last round left
            res = yield (hpc_freq[k:k + 50] + res, sequence)
            res = yield from settle(res)
    return (yield from settle(res))


def batch_cluster_sequences(jobs):
    """[(circ_id, [(read_id, sequence)])] -> [clusters] (collapse.py:419-436 per circRNA), all circRNAs advanced together so that
    every round is one call of `cluster_steps`."""
    runs = [_cluster_job(circ_id, reads) for circ_id, reads in jobs]
    results = [None] * len(runs)
    waiting = {}
    for k, g in enumerate(runs):
        try:
            waiting[k] = next(g)
        except StopIteration as done:
            results[k] = done.value
    while waiting:
        ks = list(waiting)
        answers = cluster_steps([waiting[k] for k in ks])
        for k, ans in zip(ks, answers):
            try:
                waiting[k] = runs[k].send(ans)
            except StopIteration as done:
                results[k] = done.value
                del waiting[k]
    return results


def batch_cluster_sequence(circ_id, x):
    """collapse.py:419-436 for one circRNA.  x: [(read_id, sequence)]"""
    return batch_cluster_sequences([(circ_id, x)])[0]


def cluster_sequence(hpc_freq, sequence):
    """collapse.py:458-506: one clustering step of one list"""
    return cluster_steps([(hpc_freq, sequence)])[0]


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
