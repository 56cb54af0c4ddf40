"""Small helpers of the `call` path (counterparts of CIRI_long/utils.py:52-58, 78-86, 118-124)."""
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
