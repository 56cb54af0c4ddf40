"""ctypes access to the CPU checkers under oracle/ (test infrastructure only).

* ``oracle/liboracle.so``  -- our scalar restatement (oracle/ssw_oracle.c)
* ``oracle/_ref/libssw.so`` -- the reference's own ssw.c compiled by oracle/Makefile (optional)

Nothing in the product package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')

_LUT = np.full(256, 4, dtype=np.int8)
for _c, _v in zip('ACGTN', range(5)):
    _LUT[ord(_c)] = _v
    _LUT[ord(_c.lower())] = _v


def encode(seq):
    """A/a->0 C/c->1 G/g->2 T/t->3 everything else->4 (libs/striped_smith_waterman/ssw_wrap.py:50,243-250)."""
    if isinstance(seq, str):
        seq = seq.encode('latin-1')
    return _LUT[np.frombuffer(seq, dtype=np.uint8)]


def make_mat(match, mismatch):
    """5x5 matrix of ssw_wrap.py:146-159 (N row/column = 0)."""
    m = np.full((5, 5), -mismatch, dtype=np.int8)
    np.fill_diagonal(m, match)
    m[4, :] = 0
    m[:, 4] = 0
    return m.reshape(-1).copy()


def mask_len(qlen):
    return qlen // 2 if qlen > 30 else 15  # ssw_wrap.py:196-199


class CloAlign(C.Structure):
    _fields_ = [('score1', C.c_uint16), ('score2', C.c_uint16), ('ref_begin1', C.c_int32), ('ref_end1', C.c_int32),
                ('read_begin1', C.c_int32), ('read_end1', C.c_int32), ('ref_end2', C.c_int32),
                ('cigar', C.POINTER(C.c_uint32)), ('cigarLen', C.c_int32)]


def ensure_built():
    if os.environ.get('CLH_ORACLE_SO'):      # the sanitizer build of `make -C oracle san` (oracle/Makefile)
        return os.environ['CLH_ORACLE_SO']
    so = os.path.join(ORACLE_DIR, 'liboracle.so')
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith('.c')]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-s', '-C', ORACLE_DIR, os.path.join(ORACLE_DIR, 'liboracle.so')])
    return so


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(ensure_built())
        lib.clo_ssw_align.restype = C.c_int
        lib.clo_ssw_align.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int8, C.c_void_p, C.c_int32,
                                      C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32,
                                      C.POINTER(CloAlign)]
        lib.clo_free_cigar.argtypes = [C.POINTER(CloAlign)]
        lib.clo_ssw_batch.restype = C.c_int
        lib.clo_ssw_batch.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_uint8, C.c_uint8, C.c_uint8, C.c_int8, C.c_void_p, C.c_void_p, C.c_int64]
        _oracle = lib
    return _oracle


def cigar_to_string(cig, qb, qe, qlen):
    """BAM-style u32 cigar -> SAM string with soft clips (ssw_wrap.py:349-379)."""
    s = ''
    if qb > 0:
        s += '%dS' % qb
    for c in cig:
        s += '%d%s' % (int(c) >> 4, 'MIDNSHP=X'[int(c) & 0xf] if (int(c) & 0xf) < 9 else 'M')
    if qlen - qe - 1 != 0:
        s += '%dS' % (qlen - qe - 1)
    return s


def oracle_align(ref, query, match=1, mismatch=1, gap_open=1, gap_extend=1, flag=1, score_size=2, mat=None,
                 maskl=None, filters=0, filterd=0):
    """Returns dict(score, score2, ref_begin, ref_end, query_begin, query_end, ref_end2, cigar(list), cigar_string)
    or None where the reference returns NULL."""
    r = encode(ref) if not isinstance(ref, np.ndarray) else ref
    q = encode(query) if not isinstance(query, np.ndarray) else query
    r = np.ascontiguousarray(r, dtype=np.int8)
    q = np.ascontiguousarray(q, dtype=np.int8)
    m = make_mat(match, mismatch) if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
    n = int(round(len(m) ** 0.5))
    res = CloAlign()
    ml = mask_len(len(q)) if maskl is None else maskl
    rc = oracle().clo_ssw_align(q.ctypes.data, len(q), m.ctypes.data, n, score_size, r.ctypes.data, len(r),
                                gap_open, gap_extend, flag, filters, filterd, ml, C.byref(res))
    if rc != 0:
        return None
    cig = [res.cigar[i] for i in range(res.cigarLen)]
    out = dict(score=res.score1, score2=res.score2, ref_begin=res.ref_begin1, ref_end=res.ref_end1,
               query_begin=res.read_begin1, query_end=res.read_end1, ref_end2=res.ref_end2, cigar=cig,
               cigar_string=cigar_to_string(cig, res.read_begin1, res.read_end1, len(q)) if cig else None)
    oracle().clo_free_cigar(C.byref(res))
    return out


# ------------------------------------------------------------------------------------------------
# the reference's own library (optional)
# ------------------------------------------------------------------------------------------------
REF_SO = os.path.join(ORACLE_DIR, '_ref', 'libssw.so')
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ssw_init.restype = C.c_void_p
        lib.ssw_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int8]
        lib.init_destroy.argtypes = [C.c_void_p]
        lib.ssw_align.restype = C.POINTER(CloAlign)  # same layout as s_align (ssw.h:42-52)
        lib.ssw_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16,
                                  C.c_int32, C.c_int32]
        lib.align_destroy.argtypes = [C.POINTER(CloAlign)]
        _ref = lib
    return _ref


def ref_align(ref, query, match=1, mismatch=1, gap_open=1, gap_extend=1, flag=1, score_size=2, mat=None, maskl=None,
              filters=0, filterd=0):
    r = encode(ref) if not isinstance(ref, np.ndarray) else ref
    q = encode(query) if not isinstance(query, np.ndarray) else query
    r = np.ascontiguousarray(r, dtype=np.int8)
    q = np.ascontiguousarray(q, dtype=np.int8)
    m = make_mat(match, mismatch) if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
    n = int(round(len(m) ** 0.5))
    lib = ref_lib()
    prof = lib.ssw_init(q.ctypes.data, len(q), m.ctypes.data, n, score_size)
    ml = mask_len(len(q)) if maskl is None else maskl
    p = lib.ssw_align(prof, r.ctypes.data, len(r), gap_open, gap_extend, flag, filters, filterd, ml)
    if not p:
        lib.init_destroy(prof)
        return None
    res = p.contents
    cig = [res.cigar[i] for i in range(res.cigarLen)]
    out = dict(score=res.score1, score2=res.score2, ref_begin=res.ref_begin1, ref_end=res.ref_end1,
               query_begin=res.read_begin1, query_end=res.read_end1, ref_end2=res.ref_end2, cigar=cig,
               cigar_string=cigar_to_string(cig, res.read_begin1, res.read_end1, len(q)) if cig else None)
    lib.align_destroy(p)
    lib.init_destroy(prof)
    return out


# ------------------------------------------------------------------------------------------------
# cyclic consensus (oracle/ccs_oracle.c) -- specification of THIS repository, parity unpinned
# ------------------------------------------------------------------------------------------------
def _ccs_lib():
    lib = oracle()
    if not getattr(lib, '_ccs_ready', False):
        lib.clo_find_consensus.restype = C.c_int
        lib.clo_find_consensus.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        lib.clo_poa.restype = C.c_int
        lib.clo_poa.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                C.c_void_p, C.c_void_p]
        lib.clo_ccs_segments.restype = C.c_int
        lib.clo_ccs_segments.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib._ccs_ready = True
    return lib


def decode(codes):
    return ''.join('ACGTN'[int(c)] if 0 <= int(c) <= 4 else 'N' for c in codes)


def oracle_find_consensus(seq):
    """-> (segments 'a-b;c-d;...', ccs str, period) or (None, None, 0)"""
    s = np.ascontiguousarray(encode(seq) if not isinstance(seq, np.ndarray) else seq, dtype=np.int8)
    segs = np.zeros(2 * 70, dtype=np.int32)
    nseg = C.c_int32(0); period = C.c_int32(0)
    out = np.zeros(len(s) + 8, dtype=np.int8)
    n = _ccs_lib().clo_find_consensus(s.ctypes.data, len(s), segs.ctypes.data, C.byref(nseg), out.ctypes.data, len(out), C.byref(period))
    if n <= 0:
        return None, None, 0
    seg = ';'.join('%d-%d' % (segs[2 * i], segs[2 * i + 1]) for i in range(nseg.value))
    return seg, decode(out[:n]), period.value


POA_DEFAULT = (10, -4, -8, -2, -24, -1)       # the scores of every reference call site (collapse.py:267,504; tests/test_poa.py:30)


def oracle_poa(seqs, algorithm=0, genmsa=False, m=10, n=-4, g=-8, e=-2, q=-24, c=-1, with_scores=False, min_coverage=0, with_order=False):
    """oracle/poa_oracle.c: spoa.poa(seqs, algorithm, genmsa, m, n, g, e, q, c) -> consensus | (consensus, msa[, scores[, order]]).
    str / bytes sequences are taken letter by letter as they are (spoa's alphabet is the set of raw characters); int8 arrays
    are taken as letters 0..4 and come back as 'ACGTN'.  msa holds one row per non-empty sequence.  None where an
    implementation limit was hit; ValueError on invalid parameters or where spoa throws (an alignment without a base)."""
    raw = all(not isinstance(s, np.ndarray) for s in seqs)
    if raw:
        arrs = [np.frombuffer(s.encode('latin-1') if isinstance(s, str) else bytes(s), dtype=np.int8) for s in seqs]
        dec = lambda a: bytes(np.asarray(a, dtype=np.int8).view(np.uint8)).decode('latin-1')
    else:
        arrs = [np.ascontiguousarray(encode(s) if not isinstance(s, np.ndarray) else s, dtype=np.int8) for s in seqs]
        dec = lambda a: ''.join('-' if x == 45 else 'ACGTN'[x] for x in a)
    off = np.zeros(len(arrs) + 1, dtype=np.int32)
    np.cumsum([len(a) for a in arrs], out=off[1:])
    data = np.ascontiguousarray(np.concatenate(arrs)) if arrs and off[-1] else np.zeros(1, dtype=np.int8)
    out = np.zeros(int(off[-1]) + 8, dtype=np.int8)
    par = np.array([algorithm, m, n, g, e, q, c, min_coverage], dtype=np.int32)
    ncols = C.c_int32(0)
    msa_cap = (int(off[-1]) + 8) * max(len(arrs), 1)
    msa = np.zeros(msa_cap, dtype=np.int8)
    scores = np.zeros(len(arrs) + 1, dtype=np.int32)
    order = np.zeros(int(off[-1]) + 2, dtype=np.int32)
    lib = _ccs_lib()
    lib.clo_poa_ranked.restype = C.c_int
    lib.clo_poa_ranked.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    k = lib.clo_poa_ranked(len(arrs), data.ctypes.data, off.ctypes.data, par.ctypes.data, out.ctypes.data, len(out),
                           msa.ctypes.data if genmsa else None, msa_cap, C.byref(ncols), scores.ctypes.data, order.ctypes.data)
    if k == -2:
        raise ValueError('invalid poa parameters')
    if k == -3:
        raise ValueError('alignment without a base (spoa throws)')
    cons = None if k < 0 else dec(out[:k])
    if not genmsa and not with_scores and not with_order:
        return cons
    rows = []
    if genmsa and k >= 0:
        nc = ncols.value
        rows = [dec(msa[r * nc:(r + 1) * nc]) for r in range(sum(1 for a in arrs if len(a)))]
    res = (cons, rows)
    if with_scores or with_order:
        res += ([int(x) for x in scores[:len(arrs)]],)
    if with_order:
        res += ([int(x) for x in order[1:1 + int(order[0])]] if k >= 0 else None,)
    return res


def oracle_edit_distance(x, y):
    """oracle/edit_oracle.c: unit-cost edit distance of two str/bytes"""
    L = oracle()
    bx = x.encode() if isinstance(x, str) else bytes(x)
    by = y.encode() if isinstance(y, str) else bytes(y)
    L.clo_edit_distance.restype = C.c_int32
    L.clo_edit_distance.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
    return int(L.clo_edit_distance(bx, len(bx), by, len(by)))


# ---- splice signals (oracle/splice_oracle.c) ----------------------------------------------------------------------
_SPLICE_MOTIFS = [('GT', 'AG'), ('GC', 'AG'), ('AT', 'AC'), ('GT', 'AC'), ('AT', 'AG')]      # (donor, acceptor), align.py:32-45


def _rc_upper(s):
    t = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}
    return ''.join(t.get(c, c) for c in reversed(s))


def oracle_splice_signal(contig, start, end, clip_base, host, is_canonical=True, site_runs=None, index_slices=False):
    """The splice-signal step for one candidate on the characters of its contig.  host: iterable of '+'/'-' or None;
    site_runs: (positions int64, counts int64[4]) of this contig's annotated sites (1-based, four ascending runs) or None.
    Returns (ss_site | None, us_free, ds_free) like find_annotated_signal + find_denovo_signal, or 'edge' where the
    neighbourhood leaves the contig."""
    L = oracle()
    L.clo_splice_signal.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    raw = contig if isinstance(contig, bytes) else contig.encode('latin-1')
    hm = (1 if host and '+' in host else 0) | (2 if host and '-' in host else 0)
    out = np.zeros(8, dtype=np.int32)
    pos, cnt = site_runs if site_runs is not None else (np.zeros(1, dtype=np.int64), np.zeros(4, dtype=np.int64))
    pos = np.ascontiguousarray(pos, dtype=np.int64); cnt = np.ascontiguousarray(cnt, dtype=np.int64)
    L.clo_splice_signal(raw, len(raw), start, end, clip_base, hm, 10, 3, (1 if is_canonical else 0) | (2 if index_slices else 0), pos.ctypes.data,
                        cnt.ctypes.data, out.ctypes.data)
    status, us_free, ds_free, found, strand, i, j, m = (int(x) for x in out)
    if status:
        return 'edge'
    site = None
    text = raw.decode('latin-1')
    if found == 1:
        donor, acceptor = _SPLICE_MOTIFS[m]
        site = ('{}-{}*|{}-{}'.format(acceptor, donor, i, j), '-' if strand else '+', i, j)
    elif found == 2:
        us_ss, ds_ss = text[start + i - 2:start + i], text[end + j:end + j + 2]
        if strand:
            us_ss, ds_ss = _rc_upper(ds_ss), _rc_upper(us_ss)
        site = ('{}-{}|{}-{}'.format(us_ss, ds_ss, i, j), '-' if strand else '+', i, j)
    return site, us_free, ds_free
