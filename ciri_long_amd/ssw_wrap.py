"""Drop-in counterpart of the reference's ``libs/striped_smith_waterman/ssw_wrap.py`` on top of libclh.so.

Same public surface (class and attribute names, defaults, ``None`` conventions) so that CIRI-long's call sites run
unchanged:

    Aligner(ref_seq="", match=2, mismatch=2, gap_open=3, gap_extend=1, report_secondary=False, report_cigar=False)
        .align(query_seq, min_score=0, min_len=0) -> PyAlignRes | None          (ssw_wrap.py:174-230)
    PyAlignRes: score, ref_begin, ref_end, query_begin, query_end, score2, ref_end2, cigar_string   (ssw_wrap.py:315-345)

Differences, all additive:
  * the arithmetic runs on the GPU (HIP kernels behind the batched C ABI); nothing is computed on the CPU;
  * ``Aligner.align_batch(queries)`` and the module-level ``align_pairs(refs, queries, ...)`` issue ONE call for many
    alignments -- what a GPU needs, and what ``find_bsj.align_clip_segments`` / ``collapse`` are restructured around;
  * sequences are encoded with a 256-entry table instead of the reference's per-base Python loop (ssw_wrap.py:234-252);
    the resulting codes are identical (A/a 0, C/c 1, G/g 2, T/t 3, everything else 4).
"""
import numpy as np

from . import hip

_OPS = 'MIDNSHP=X'


class CAlignRes(object):
    """Field-for-field view of one result row (the reference's ctypes mirror of s_align, ssw_wrap.py:20-37)."""
    __slots__ = ('score', 'score2', 'ref_begin', 'ref_end', 'query_begin', 'query_end', 'ref_end2', 'cigar', 'cigarLen')

    def __init__(self, row, cigar):
        self.score = int(row['score1'])
        self.score2 = int(row['score2'])
        self.ref_begin = int(row['ref_begin1'])
        self.ref_end = int(row['ref_end1'])
        self.query_begin = int(row['read_begin1'])
        self.query_end = int(row['read_end1'])
        self.ref_end2 = int(row['ref_end2'])
        self.cigar = cigar
        self.cigarLen = len(cigar)


class PyAlignRes(object):
    """Result object with the attributes CIRI-long reads (find_bsj.py:206-224, collapse.py:157,214,256,264,382,774,
    align.py:804-806)."""

    def __init__(self, res, query_len, report_secondary=False, report_cigar=False):
        self.score = res.score
        self.ref_begin = res.ref_begin
        self.ref_end = res.ref_end
        self.query_begin = res.query_begin
        self.query_end = res.query_end
        if report_secondary and res.score2 != 0:      # ssw_wrap.py:332-338
            self.score2 = res.score2
            self.ref_end2 = res.ref_end2
        else:
            self.score2 = None
            self.ref_end2 = None
        if report_cigar and res.cigarLen > 0:          # ssw_wrap.py:341-345
            self.cigar_string = self._cigar_string(res.cigar, query_len)
        else:
            self.cigar_string = None

    def _cigar_string(self, cigar, query_len):
        """BAM-style u32 ops -> SAM text, soft clips added at both ends (ssw_wrap.py:349-379)."""
        parts = []
        if self.query_begin > 0:
            parts.append('%dS' % self.query_begin)
        for c in cigar:
            c = int(c)
            code = c & 0xf
            parts.append('%d%s' % (c >> 4, _OPS[code] if code < len(_OPS) else 'M'))
        tail = query_len - self.query_end - 1
        if tail != 0:
            parts.append('%dS' % tail)
        return ''.join(parts)

    def __str__(self):
        return "\n<Instance of {} from {} >\n".format(self.__class__.__name__, self.__module__)

    def __repr__(self):
        msg = self.__str__()
        msg += "OPTIMAL MATCH\n"
        msg += "Score            {}\n".format(self.score)
        msg += "Reference begin  {}\n".format(self.ref_begin)
        msg += "Reference end    {}\n".format(self.ref_end)
        msg += "Query begin      {}\n".format(self.query_begin)
        msg += "Query end        {}\n".format(self.query_end)
        if self.cigar_string:
            msg += "Cigar_string     {}\n".format(self.cigar_string)
        if self.score2:
            msg += "SUB-OPTIMAL MATCH\n"
            msg += "Score 2           {}\n".format(self.score2)
            msg += "Ref_end2          {}\n".format(self.ref_end2)
        return msg


def _filter(row, cig, query_len, min_score, min_len, report_secondary, report_cigar):
    """Everything after the FFI call in ssw_wrap.py:211-222."""
    if int(row['status']) & (hip.ST_NULL | hip.ST_TRACE_ERR | hip.ST_CIGAR_TRUNC):
        return None            # the reference's NULL result: score := -999999999999 -> filtered out
    res = CAlignRes(row, cig)
    if res.score >= min_score and (res.query_end - res.query_begin + 1) >= min_len:
        return PyAlignRes(res, query_len, report_secondary, report_cigar)
    return None


def align_pairs(ref_seqs, query_seqs, match=2, mismatch=2, gap_open=3, gap_extend=1, report_secondary=False,
                report_cigar=False, min_score=0, min_len=0, context=None):
    """n independent (reference, query) alignments in one GPU call; element k equals
    ``Aligner(ref_seqs[k], ...).align(query_seqs[k], min_score, min_len)``.  Sequences are ``str`` or int8 code arrays."""
    if len(ref_seqs) != len(query_seqs):
        raise ValueError('align_pairs: %d references vs %d queries' % (len(ref_seqs), len(query_seqs)))
    if not ref_seqs:
        return []
    ctx = context or hip.default_context()
    qd, qo = hip.pack(query_seqs)
    rd, ro = hip.pack(ref_seqs)
    rows, cig = ctx.ssw_batch(qd, qo, rd, ro, hip.score_matrix(match, mismatch), gap_open, gap_extend, flag=1, score_size=2,
                              want_score2=bool(report_secondary), want_cigar=bool(report_cigar))
    out = []
    for k in range(len(rows)):
        r = rows[k]
        c = cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']] if r['cigar_len'] > 0 else ()
        out.append(_filter(r, c, int(qo[k + 1] - qo[k]), min_score, min_len, report_secondary, report_cigar))
    return out


def align_windows(genome, windows, minus, query_seqs, match=2, mismatch=2, gap_open=3, gap_extend=1, report_secondary=False,
                  report_cigar=False, min_score=0, min_len=0):
    """align_pairs with reference k = window (contig, start, end) of a genome resident on the GPU (hip.Genome),
    reverse-complemented the reference's way where minus[k]: element k equals
    ``Aligner(revcomp(seq) if minus[k] else seq, ...).align(query_seqs[k])`` for seq = the window's string."""
    if len(windows) != len(query_seqs):
        raise ValueError('align_windows: %d windows vs %d queries' % (len(windows), len(query_seqs)))
    if not windows:
        return []
    qd, qo = hip.pack(query_seqs)
    rows, cig = genome.ssw_windows(qd, qo, windows, minus, hip.score_matrix(match, mismatch), gap_open, gap_extend, flag=1, score_size=2,
                                   want_score2=bool(report_secondary), want_cigar=bool(report_cigar))
    out = []
    for k in range(len(rows)):
        r = rows[k]
        c = cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']] if r['cigar_len'] > 0 else ()
        out.append(_filter(r, c, int(qo[k + 1] - qo[k]), min_score, min_len, report_secondary, report_cigar))
    return out


class Aligner(object):
    """One reference sequence, many queries (ssw_wrap.py:40-264)."""

    base_to_int = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4, 'a': 0, 'c': 1, 'g': 2, 't': 3, 'n': 4}
    int_to_base = {0: 'A', 1: 'C', 2: 'G', 3: 'T', 4: 'N'}

    def __init__(self, ref_seq="", match=2, mismatch=2, gap_open=3, gap_extend=1, report_secondary=False,
                 report_cigar=False, context=None):
        self.report_secondary = report_secondary
        self.report_cigar = report_cigar
        self._context = context
        self.set_gap(gap_open, gap_extend)
        self.set_mat(match, mismatch)
        self.set_ref(ref_seq)

    # -- setters, ssw_wrap.py:137-170 --
    def set_gap(self, gap_open=3, gap_extend=1):
        self.gap_open = gap_open
        self.gap_extend = gap_extend

    def set_mat(self, match=2, mismatch=2):
        self.match = match
        self.mismatch = mismatch
        self.mat = hip.score_matrix(match, mismatch)

    def set_ref(self, ref_seq):
        if ref_seq is not None and len(ref_seq):
            self.ref_seq = self._DNA_to_int_mat(ref_seq, len(ref_seq))
            self.ref_len = len(self.ref_seq)
        else:
            self.ref_len = 0
            self.ref_seq = ""

    def _DNA_to_int_mat(self, seq, len_seq):
        if isinstance(seq, np.ndarray):
            return np.ascontiguousarray(seq[:len_seq], dtype=np.int8)
        return hip.encode(seq[:len_seq])

    # -- alignment --
    def align(self, query_seq, min_score=0, min_len=0):
        res = self.align_batch([query_seq], min_score, min_len)
        return res[0]

    def align_batch(self, query_seqs, min_score=0, min_len=0):
        """All queries against this reference in one GPU call; element k equals ``self.align(query_seqs[k], ...)``."""
        n = len(query_seqs)
        if n == 0:
            return []
        if self.ref_len == 0:
            raise ValueError('Aligner has no reference sequence')
        ctx = self._context or hip.default_context()
        qd, qo = hip.pack([self._DNA_to_int_mat(q, len(q)) for q in query_seqs])
        rd = np.tile(self.ref_seq, n) if n > 1 else self.ref_seq
        ro = np.arange(n + 1, dtype=np.int64) * self.ref_len
        rows, cig = ctx.ssw_batch(qd, qo, rd, ro, self.mat, self.gap_open, self.gap_extend, flag=1, score_size=2,
                                  want_score2=bool(self.report_secondary), want_cigar=bool(self.report_cigar))
        out = []
        for k in range(n):
            r = rows[k]
            c = cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']] if r['cigar_len'] > 0 else ()
            out.append(_filter(r, c, int(qo[k + 1] - qo[k]), min_score, min_len, self.report_secondary, self.report_cigar))
        return out

    def __str__(self):
        return "\n<Instance of {} from {} >\n".format(self.__class__.__name__, self.__module__)

    def __repr__(self):
        msg = self.__str__()
        msg += "SCORE PARAMETERS:\n"
        msg += " Gap Weight     Open: {}     Extension: {}\n".format(-self.gap_open, -self.gap_extend)
        msg += " Align Weight   Match: {}    Mismatch: {}\n\n".format(self.match, -self.mismatch)
        msg += "RESULT PARAMETERS:\n"
        msg += " Report cigar           {}\n".format(self.report_cigar)
        msg += " Report secondary match {}\n\n".format(self.report_secondary)
        msg += "REFERENCE SEQUENCE :\n"
        shown = min(self.ref_len, 50)
        msg += "".join(self.int_to_base[int(self.ref_seq[i])] for i in range(shown)) + ("...\n" if self.ref_len > 50 else "\n")
        msg += " Lenght :{} nucleotides\n".format(self.ref_len)
        return msg
