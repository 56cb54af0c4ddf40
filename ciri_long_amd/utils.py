"""Small helpers of the `call` and `collapse` paths (counterparts of CIRI_long/utils.py:52-58, 78-86, 118-124, 153-167)."""
from itertools import zip_longest

_COMPLEMENT = str.maketrans('ATCG', 'TAGC')   # upper-case ACGT only; N and lower case pass through (utils.py:118-120)


def to_str(bytes_or_str):
    return bytes_or_str.decode('utf-8') if isinstance(bytes_or_str, bytes) else bytes_or_str


def revcomp(seq):
    return seq.translate(_COMPLEMENT)[::-1]


def grouper(iterable, n, fillvalue=None):
    """Fixed-size chunks, the last one padded with None (the reference ignores ``fillvalue`` too, utils.py:78-86)."""
    it = iter(iterable)
    return zip_longest(*([it] * n), fillvalue=None)


def transform_seq(seq, bsj):
    return seq[bsj:] + seq[:bsj]


def pairwise(iterable):
    """neighbouring pairs (s0, s1), (s1, s2), ... (utils.py:89-94)"""
    items = list(iterable)
    return zip(items, items[1:])


def flatten(x):
    """one level of nesting removed (utils.py:102-109)"""
    return [item for sub in x for item in sub]


def min_sorted_items(iters, key, reverse=False):
    """every item that shares the best value of field `key` (utils.py:112-115)"""
    from operator import itemgetter
    x = sorted(iters, key=itemgetter(key), reverse=reverse)
    return [i for i in x if i[key] == x[0][key]]


def get_junc_seq(seq, bsj, width=25):
    """2*width bases around position bsj of a circular sequence (utils.py:127-140)"""
    st, en = bsj - width, bsj + width
    if len(seq) <= 2 * width:
        return seq[bsj - len(seq) // 2:] + seq[:bsj - len(seq) // 2]
    if st < 0:
        return seq[st:en] if en < 0 else seq[st:] + seq[:en]
    if en > len(seq):
        return seq[st:] + seq[:en - len(seq)]
    return seq[st:en]


def distance(x, y):
    """Unit-cost edit distance of two strings (utils.py:153-159: python-Levenshtein / edlib, the same integer), on the GPU."""
    from . import hip
    return int(hip.default_context().edit_distance_batch([x], [y])[0])


def distance_batch(xs, ys):
    """distance(xs[k], ys[k]) for every k in one launch per size class."""
    from . import hip
    return hip.default_context().edit_distance_batch(xs, ys)


def pairwise_distance(seqs):
    """The matrix cluster_sequence builds (collapse.py:466-473): dist[i][j] = distance(s_i, s_j) / max(len(s_i), len(s_j)),
    symmetric, zero diagonal; all i < j pairs go to the GPU as one batch."""
    import numpy as np
    n = len(seqs)
    dist = np.zeros((n, n))
    ii, jj = np.triu_indices(n, 1)
    if len(ii):
        d = distance_batch([seqs[i] for i in ii], [seqs[j] for j in jj])
        norm = np.array([max(len(seqs[i]), len(seqs[j])) for i, j in zip(ii, jj)], dtype=np.float64)
        # the reference also divides the (zero) diagonal distances by the length; an empty pair would raise there too
        dist[ii, jj] = d / norm
    return dist + dist.T


def compress_seq(seq):
    """Homopolymer compression: every run of equal characters becomes one (utils.py:162-167)."""
    from itertools import groupby
    return ''.join(ch for ch, _ in groupby(seq))
