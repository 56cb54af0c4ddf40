"""ctypes binding of libclh.so's batched C ABI (include/ciri_long_hip.h).

The shared object is built in-tree by csrc/Makefile (``python -c 'import __graft_entry__ as g; g.build()'``) and must
sit next to this file, as the reference's libssw.so sits next to its wrapper (ssw_wrap.py:17).  Nothing here falls
back to the CPU: a missing library or GPU raises ``HipUnavailable``.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('CLH_LIB') or os.path.join(_HERE, 'libclh.so')     # CLH_LIB: another build of the same ABI (kernel experiments)


class HipUnavailable(RuntimeError):
    pass


class ClhError(RuntimeError):
    pass


class AlignRow(C.Structure):
    _fields_ = [('score1', C.c_uint16), ('score2', C.c_uint16), ('ref_begin1', C.c_int32), ('ref_end1', C.c_int32),
                ('read_begin1', C.c_int32), ('read_end1', C.c_int32), ('ref_end2', C.c_int32),
                ('cigar_off', C.c_int32), ('cigar_len', C.c_int32), ('status', C.c_int32)]


ALIGN_DTYPE = np.dtype([('score1', '<u2'), ('score2', '<u2'), ('ref_begin1', '<i4'), ('ref_end1', '<i4'),
                        ('read_begin1', '<i4'), ('read_end1', '<i4'), ('ref_end2', '<i4'), ('cigar_off', '<i4'),
                        ('cigar_len', '<i4'), ('status', '<i4')])
assert ALIGN_DTYPE.itemsize == C.sizeof(AlignRow)

ST_WORD, ST_NULL, ST_TRACE_ERR, ST_NO_CIGAR, ST_CIGAR_TRUNC = 1, 2, 4, 8, 16

CCS_SEG_CAP = 65
CCS_DTYPE = np.dtype([('nseg', '<i4'), ('ccs_len', '<i4'), ('period', '<i4'), ('status', '<i4')])


class SswOpts(C.Structure):
    _fields_ = [('mat', C.c_void_p), ('n_mat', C.c_int32), ('gap_open', C.c_uint8), ('gap_extend', C.c_uint8),
                ('flag', C.c_uint8), ('score_size', C.c_int8), ('filters', C.c_uint16), ('filterd', C.c_int32),
                ('want_score2', C.c_int32), ('want_cigar', C.c_int32)]


_lib = None
_gpu_used = False         # set when this process creates its first device context (clh_create: the first call that initialises the HIP runtime);
                          # loading libclh.so and its host-only entry points (fastx_index, fastx_count, encode) do not


def lib():
    """Load libclh.so (once).  Raises HipUnavailable if it has not been built."""
    global _lib
    if _lib is None:
        if os.environ.get('CIRI_LONG_MAPPER_WORKER') == '1':
            raise HipUnavailable('a mapper worker process must not touch the GPU (ciri_long_amd/mapper_pool.py)')
        if not os.path.exists(SO_PATH):
            raise HipUnavailable('%s not found: build it with `make -C %s/csrc` (needs hipcc); there is no CPU fallback'
                                 % (SO_PATH, _HERE))
        L = C.CDLL(SO_PATH)
        L.clh_last_error.restype = C.c_char_p
        L.clh_version.restype = C.c_char_p
        L.clh_create.restype = C.c_void_p
        L.clh_create.argtypes = [C.c_int]
        L.clh_destroy.argtypes = [C.c_void_p]
        L.clh_ssw_plan.restype = C.c_void_p
        L.clh_ssw_plan.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SswOpts)]
        L.clh_plan_destroy.argtypes = [C.c_void_p]
        L.clh_ssw_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_ssw_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.clh_ssw_results_dev.restype = C.c_void_p
        L.clh_ssw_results_dev.argtypes = [C.c_void_p]
        L.clh_ssw_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(SswOpts), C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.clh_encode_dna.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
        L.clh_ccs_plan_create.restype = C.c_void_p
        L.clh_ccs_plan_create.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.clh_ccs_plan_destroy.argtypes = [C.c_void_p]
        L.clh_ccs_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_ccs_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_ccs_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_ccs_plan_timing.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_ccs_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p]
        L.clh_ccs_file.restype = C.c_int
        L.clh_ccs_file_range.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
        L.clh_ccs_file_range.restype = C.c_int
        L.clh_fastx_count.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.clh_genome_create.restype = C.c_void_p
        L.clh_genome_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.clh_genome_destroy.restype = None
        L.clh_genome_destroy.argtypes = [C.c_void_p]
        L.clh_genome_codes.restype = C.c_void_p
        L.clh_genome_codes.argtypes = [C.c_void_p]
        L.clh_genome_length.restype = C.c_int64
        L.clh_genome_length.argtypes = [C.c_void_p]
        L.clh_genome_count_n.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_genome_set_splice_sites.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_splice_signal_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.clh_ssw_plan_windows.restype = C.c_void_p
        L.clh_ssw_plan_windows.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_ssw_windows_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.clh_edit_plan_create.restype = C.c_void_p
        L.clh_edit_plan_create.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_edit_plan_destroy.restype = None
        L.clh_edit_plan_destroy.argtypes = [C.c_void_p]
        L.clh_edit_plan_run.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_edit_plan_fetch.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_edit_plan_timing.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_edit_distance_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_edit_distance_batch.restype = C.c_int
        L.clh_poa_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_ccs_results_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_ccs_plan_info.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_ccs_plan_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_poa_last_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_plan_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.clh_plan_segments.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_plan_timing.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.clh_plan_traceback_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_plan_prefilter_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.clh_plan_prefilter_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.clh_plan_set_refs_bytes.argtypes = [C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def fastx_index(in_path, is_fastq, every=4096):
    """(records, [byte offset of record 0, every, 2 every, ...]) of a FASTA/FASTQ file; no offsets for a gzip file (host only)"""
    n = C.c_int64(0); no = C.c_int64(0)
    L = lib()
    L.clh_fastx_index.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    cap = max(16, os.path.getsize(in_path) // max(1, 20 * every) + 16)
    off = np.zeros(cap, dtype=np.int64)
    if L.clh_fastx_index(os.fsencode(in_path), int(bool(is_fastq)), int(every), C.byref(n), off.ctypes.data, cap, C.byref(no)) != 0:
        raise ClhError('clh_fastx_index failed for %s' % in_path)
    return int(n.value), [int(x) for x in off[:no.value]]


def fastx_count(in_path, is_fastq):
    """records of a FASTA/FASTQ(.gz) file as find_ccs_reads' loop counts them (host only, no GPU)"""
    n = C.c_int64(0)
    if lib().clh_fastx_count(os.fsencode(in_path), int(bool(is_fastq)), C.byref(n)) != 0:
        raise ClhError('clh_fastx_count failed for %s' % in_path)
    return int(n.value)


def last_error():
    return lib().clh_last_error().decode()


_LUT = np.full(256, 4, dtype=np.int8)
for _c, _v in zip('ACGTN', range(5)):
    _LUT[ord(_c)] = _v
    _LUT[ord(_c.lower())] = _v


def encode(seq):
    """ASCII -> int8 codes, the mapping of ssw_wrap.py:50,243-250 (vectorised; the reference loops per base)."""
    if isinstance(seq, str):
        seq = seq.encode('latin-1')
    return _LUT[np.frombuffer(seq, dtype=np.uint8)]


def pack_raw(seqs):
    """list of str / bytes -> (the letters as they are, one byte each, int64 offsets[n+1]): the partial-order aligner compares
    letters for equality only (spoa's alphabet is the set of raw characters)"""
    arrs = [np.frombuffer(s.encode('latin-1') if isinstance(s, str) else bytes(s), dtype=np.int8) for s in seqs]
    off = np.zeros(len(arrs) + 1, dtype=np.int64)
    if arrs:
        np.cumsum([len(a) for a in arrs], out=off[1:])
    data = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.int8)
    return np.ascontiguousarray(data, dtype=np.int8), off


def score_matrix(match, mismatch):
    """ssw_wrap.py:146-159: match on the diagonal, -mismatch elsewhere, 0 for N."""
    m = np.full((5, 5), -int(mismatch), dtype=np.int8)
    np.fill_diagonal(m, int(match))
    m[4, :] = 0
    m[:, 4] = 0
    return np.ascontiguousarray(m.reshape(-1))


def pack(seqs):
    """list of int8 arrays / str -> (packed int8 array, int64 offsets[n+1])"""
    arrs = [encode(s) if isinstance(s, (str, bytes)) else np.asarray(s, dtype=np.int8) for s in seqs]
    off = np.zeros(len(arrs) + 1, dtype=np.int64)
    if arrs:
        np.cumsum([len(a) for a in arrs], out=off[1:])
    data = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.int8)
    return np.ascontiguousarray(data, dtype=np.int8), off


def pack_text(seqs):
    """pack() for a list of str only, encoded in one pass over their concatenation (a clip batch is thousands of short strings)"""
    blob = ''.join(seqs).encode('latin-1')
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    if seqs:
        np.cumsum([len(s) for s in seqs], out=off[1:])
    return np.ascontiguousarray(_LUT[np.frombuffer(blob, dtype=np.uint8)], dtype=np.int8), off


def _torch_first():
    """torch ships its own HIP runtime; when a process uses both, torch's must open the device before libclh's does
    (the other order leaves torch with "No HIP GPUs are available").  Only acts when the caller already imported torch."""
    t = sys.modules.get('torch')
    if t is not None and t.cuda.is_available() and not t.cuda.is_initialized():
        t.cuda.init()


class Context(object):
    """One per (process, GPU)."""

    def __init__(self, device=0):
        global _gpu_used
        L = lib()
        _torch_first()
        self._h = L.clh_create(int(device))
        if not self._h:
            raise HipUnavailable('clh_create(%d) failed: %s (there is no CPU fallback)' % (device, last_error()))
        _gpu_used = True
        self.device = int(device)

    def close(self):
        if getattr(self, '_h', None):
            lib().clh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _opts(self, mat, gap_open, gap_extend, flag, score_size, want_score2, want_cigar, filters=0, filterd=0):
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        o = SswOpts()
        o.mat = mat.ctypes.data
        o.n_mat = int(round(len(mat) ** 0.5))
        o.gap_open, o.gap_extend, o.flag, o.score_size = int(gap_open), int(gap_extend), int(flag), int(score_size)
        o.filters, o.filterd, o.want_score2, o.want_cigar = int(filters), int(filterd), int(want_score2), int(want_cigar)
        return o, mat

    def ssw_batch(self, reads, read_off, refs, ref_off, mat, gap_open, gap_extend, flag=1, score_size=2,
                  want_score2=True, want_cigar=True, mask_len=None, filters=0, filterd=0):
        """Host arrays in, (rows: structured array, cigars: uint32 array) out.  flag / filters / filterd as ssw_align
        takes them (ssw.h:95-111: bit 1 score filter, bit 2 distance filter, bit 3 = bit 0 for the begin positions)."""
        L = lib()
        reads = np.ascontiguousarray(reads, dtype=np.int8)
        refs = np.ascontiguousarray(refs, dtype=np.int8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        ref_off = np.ascontiguousarray(ref_off, dtype=np.int64)
        n = len(read_off) - 1
        o, _keep = self._opts(mat, gap_open, gap_extend, flag, score_size, want_score2, want_cigar, filters, filterd)
        out = np.zeros(n, dtype=ALIGN_DTYPE)
        cap = int(2 * (read_off[-1] if n else 0) + 2 * n + 8) if want_cigar else 1
        cig = np.empty(cap, dtype=np.uint32)     # worst-case capacity; only the used prefix is written
        used = C.c_int64(0)
        ml = None
        if mask_len is not None:
            ml = np.ascontiguousarray(mask_len, dtype=np.int32)
        rc = L.clh_ssw_batch(self._h, n, reads.ctypes.data, read_off.ctypes.data, refs.ctypes.data, ref_off.ctypes.data,
                             ml.ctypes.data if ml is not None else None, C.byref(o), out.ctypes.data,
                             cig.ctypes.data if want_cigar else None, cap, C.byref(used))
        if rc != 0:
            raise ClhError('clh_ssw_batch failed (%d): %s' % (rc, last_error()))
        return out, cig[:used.value]

    def ccs_batch(self, reads, read_off):
        """find_consensus for a batch: -> (rows CCS_DTYPE[n], segs int32[n, 65, 2], ccs int8 packed like reads)"""
        reads = np.ascontiguousarray(reads, dtype=np.int8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        n = len(read_off) - 1
        out = np.zeros(n, dtype=CCS_DTYPE)
        segs = np.zeros((n, CCS_SEG_CAP, 2), dtype=np.int32)
        ccs = np.zeros(max(1, len(reads)), dtype=np.int8)
        rc = lib().clh_ccs_batch(self._h, n, reads.ctypes.data, read_off.ctypes.data, out.ctypes.data, segs.ctypes.data, ccs.ctypes.data)
        if rc != 0:
            raise ClhError('clh_ccs_batch failed (%d): %s' % (rc, last_error()))
        return out, segs, ccs

    def poa_batch(self, seqs, seq_off, group_off, algorithm=0, scores=(10, -4, -8, -2, -24, -1), min_coverage=0, genmsa=False,
                  with_scores=False, raw=False):
        """spoa.poa per group of sequences: -> list of consensus str, or list of (consensus, msa rows[, end-cell scores of
        the first 65 sequences]) with genmsa / with_scores.  The letters of `seqs` are bytes compared for equality only;
        raw=False reads them as the codes 0..4 of `encode` and writes ACGTN, raw=True hands them through as characters
        (`pack_raw`).  Raises ClhError when a group has no consensus (a node with more than 48
        in-edges, more than 8 different letters in a column) or the scores are outside what the kernel honours."""
        seqs = np.ascontiguousarray(seqs, dtype=np.int8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        group_off = np.ascontiguousarray(group_off, dtype=np.int64)
        ng = len(group_off) - 1
        lens = np.zeros(ng, dtype=np.int32)
        out = np.zeros(max(1, len(seqs)), dtype=np.int8)
        opts = np.array([algorithm] + [int(x) for x in scores] + [min_coverage], dtype=np.int32)
        col = np.zeros(max(1, len(seqs)), dtype=np.int32) if genmsa else None
        ncols = np.zeros(max(1, ng), dtype=np.int32) if genmsa else None
        asc = np.zeros((max(1, ng), CCS_SEG_CAP), dtype=np.int32) if with_scores else None
        rc = lib().clh_poa_batch(self._h, ng, seqs.ctypes.data, seq_off.ctypes.data, group_off.ctypes.data, opts.ctypes.data,
                                 lens.ctypes.data, out.ctypes.data, col.ctypes.data if genmsa else None,
                                 ncols.ctypes.data if genmsa else None, asc.ctypes.data if with_scores else None)
        if rc != 0:
            raise ClhError('clh_poa_batch failed (%d): %s' % (rc, last_error()))
        bases = np.frombuffer(b'ACGTN', dtype=np.uint8)
        res = []
        for k in range(ng):
            if lens[k] < 0:
                raise ClhError('poa: no consensus for group %d (status %d: 1 workspace, 2 graph limits -- more than 48 in-edges, more than 8 '
                               'letters in a column or 65000 nodes, 3 output, 4 (unused since round 4), 5 back-track guard, 6 a cell left the '
                               '16-bit score range, 7 an alignment without a base: spoa throws)' % (k, -1 - int(lens[k])))
            o = int(seq_off[group_off[k]])
            text = (lambda a: a.view(np.uint8).tobytes().decode('latin-1')) if raw else (lambda a: bases[np.minimum(a, 4)].tobytes().decode())
            cons = text(out[o:o + int(lens[k])])
            if not genmsa and not with_scores:
                res.append(cons)
                continue
            rows = []
            if genmsa:
                for i in range(int(group_off[k]), int(group_off[k + 1])):
                    row = np.full(int(ncols[k]), ord('-'), dtype=np.uint8)
                    a, b = int(seq_off[i]), int(seq_off[i + 1])
                    row[col[a:b]] = seqs[a:b].view(np.uint8) if raw else bases[np.minimum(seqs[a:b], 4)]
                    rows.append(row.tobytes().decode('latin-1'))
            item = (cons, rows)
            if with_scores:
                item = item + ([int(x) for x in asc[k, :min(CCS_SEG_CAP, int(group_off[k + 1] - group_off[k]))]],)
            res.append(item)
        return res

    def poa_last_stats(self):
        """statistics of the last poa_batch (see CcsPlan.stats)"""
        out = np.zeros(16, dtype=np.int64)
        if lib().clh_poa_last_stats(self._h, out.ctypes.data) != 0:
            raise ClhError('clh_poa_last_stats: %s' % last_error())
        return {'dp_cells': int(out[0]), 'dp_row_steps': int(out[1]), 'band_misses': int(out[2]), 'dropped': {k: int(out[2 + k]) for k in range(1, 8) if out[2 + k]}}

    def edit_distance_batch(self, xs, ys):
        """Unit-cost edit distance of the pairs (xs[k], ys[k]) (str or bytes) -> int32 array.  K4 through the C ABI."""
        n = len(xs)
        if n != len(ys):
            raise ValueError('edit_distance_batch: the two lists differ in length')
        ba = [x.encode() if isinstance(x, str) else bytes(x) for x in xs]
        bb = [y.encode() if isinstance(y, str) else bytes(y) for y in ys]
        a_off = np.zeros(n + 1, dtype=np.int64); b_off = np.zeros(n + 1, dtype=np.int64)
        if n:
            a_off[1:] = np.cumsum([len(x) for x in ba]); b_off[1:] = np.cumsum([len(y) for y in bb])
        a = np.frombuffer(b''.join(ba) + b'\0', dtype=np.uint8)
        b = np.frombuffer(b''.join(bb) + b'\0', dtype=np.uint8)
        out = np.zeros(n, dtype=np.int32)
        rc = lib().clh_edit_distance_batch(self._h, n, a.ctypes.data, a_off.ctypes.data, b.ctypes.data, b_off.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise ClhError('clh_edit_distance_batch failed (%d): %s' % (rc, last_error()))
        return out

    def edit_plan(self, xs, ys):
        return EditPlan(self, xs, ys)

    def ccs_file(self, in_path, is_fastq, ccs_fa_path, raw_fa_path, batch_reads=0, first_record=0, max_records=-1, byte_offset=0):
        """Stage 1 from file to file in native code -> (total_reads, reads_with_consensus, reads_too_long); with
        first_record / max_records for one rank's contiguous shard of the records, counted from `byte_offset` (the first byte of a
        record, `fastx_index`).  Reads that a limit of the kernel left without a consensus are counted in self.capacity_dropped
        (and logged by find_ccs_reads)."""
        st = (C.c_int64 * 4)()
        L = lib()
        L.clh_ccs_file_at.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        rc = L.clh_ccs_file_at(self._h, os.fsencode(in_path), int(bool(is_fastq)), os.fsencode(ccs_fa_path), os.fsencode(raw_fa_path),
                               int(batch_reads), int(byte_offset), int(first_record), int(max_records), C.byref(st))
        if rc != 0:
            raise ClhError('clh_ccs_file failed (%d): %s' % (rc, last_error()))
        self.capacity_dropped = getattr(self, 'capacity_dropped', 0) + int(st[3])
        self.last_capacity_dropped = int(st[3])
        return int(st[0]), int(st[1]), int(st[2])

    @staticmethod
    def release_file_buffers():
        """give back the host and device buffers the file stage (ccs_file) keeps between calls"""
        lib().clh_ccs_file_release_buffers()

    def ccs_plan(self, read_off):
        return CcsPlan(self, read_off)

    def plan(self, read_off, ref_off, mat, gap_open, gap_extend, flag=1, score_size=2, want_score2=True,
             want_cigar=True, mask_len=None):
        return Plan(self, read_off, ref_off, mat, gap_open, gap_extend, flag, score_size, want_score2, want_cigar, mask_len)


def _pack_bytes(items):
    bs = [x.encode() if isinstance(x, str) else bytes(x) for x in items]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        off[1:] = np.cumsum([len(x) for x in bs])
    return np.frombuffer(b''.join(bs) + b'\0', dtype=np.uint8), off


class EditPlan(object):
    """Pairs of strings resident on the GPU: run() the edit distances any number of times, fetch() the int32 array."""

    def __init__(self, ctx, xs, ys):
        if len(xs) != len(ys):
            raise ValueError('EditPlan: the two lists differ in length')
        self.n = len(xs)
        a, a_off = _pack_bytes(xs)
        b, b_off = _pack_bytes(ys)
        self._h = lib().clh_edit_plan_create(ctx._h, self.n, a.ctypes.data, a_off.ctypes.data, b.ctypes.data, b_off.ctypes.data)
        if not self._h:
            raise ClhError('clh_edit_plan_create failed: %s' % last_error())

    def run(self, stream=0):
        rc = lib().clh_edit_plan_run(self._h, C.c_void_p(stream))
        if rc != 0:
            raise ClhError('clh_edit_plan_run failed (%d): %s' % (rc, last_error()))

    def fetch(self):
        out = np.zeros(self.n, dtype=np.int32)
        rc = lib().clh_edit_plan_fetch(self._h, out.ctypes.data)
        if rc != 0:
            raise ClhError('clh_edit_plan_fetch failed (%d): %s' % (rc, last_error()))
        return out

    def timing(self):
        ms = C.c_float(0)
        if lib().clh_edit_plan_timing(self._h, C.byref(ms)) != 0:
            raise ClhError('clh_edit_plan_timing: %s' % last_error())
        return float(ms.value)

    def close(self):
        if self._h:
            if getattr(self.ctx, '_h', None):
                lib().clh_edit_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flatten_splice_sites(ss_index, offset, length):
    """{contig: {pos: {strand: {'start'|'end': 1}}}} -> (positions int64, counts int64[4]): four strictly ascending runs
    of contig offset + pos in the order '+' starts, '+' ends, '-' starts, '-' ends (clh_genome_set_splice_sites).
    Contigs without an offset, positions outside their contig and strands other than '+'/'-' are left out: no candidate
    can look them up, and a position past the end of a contig must not alias its neighbour."""
    runs = [[], [], [], []]
    for ctg, by_pos in (ss_index or {}).items():
        if ctg not in offset:
            continue
        off, ln = offset[ctg], length[ctg]
        for pos, by_strand in by_pos.items():
            if not 1 <= pos <= ln:
                continue
            for k, strand in enumerate('+-'):
                kinds = by_strand.get(strand) if hasattr(by_strand, 'get') else None
                if not kinds:
                    continue
                if 'start' in kinds:
                    runs[2 * k].append(off + pos)
                if 'end' in kinds:
                    runs[2 * k + 1].append(off + pos)
    runs = [np.unique(np.array(r, dtype=np.int64)) for r in runs]
    cnt = np.array([len(r) for r in runs], dtype=np.int64)
    flat = np.ascontiguousarray(np.concatenate(runs)) if cnt.sum() else np.zeros(1, dtype=np.int64)
    return flat, cnt


class Genome(object):
    """Contigs resident in HBM as base codes (K5).  A Smith-Waterman reference is then a window (contig, start, end,
    minus-strand flag) read in place: no window string, no reverse complement, no per-base encode on the host."""

    def __init__(self, ctx, contigs):
        """contigs: {name: str} (or an iterable of (name, str)); they are concatenated in iteration order"""
        items = list(contigs.items()) if hasattr(contigs, 'items') else list(contigs)
        self.ctx = ctx
        self.offset, self.length = {}, {}
        pos = 0
        for name, seq in items:
            self.offset[name] = pos; self.length[name] = len(seq); pos += len(seq)
        blob = ''.join(seq for _, seq in items).encode('latin-1')
        self._sites_of = None
        self._h = lib().clh_genome_create(ctx._h, blob, len(blob))
        if not self._h:
            raise ClhError('clh_genome_create failed: %s' % last_error())

    def close(self):
        if self._h:
            if getattr(self.ctx, '_h', None):        # a context closed first has already given everything back
                lib().clh_genome_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def codes_ptr(self):
        return lib().clh_genome_codes(self._h)

    def _spans(self, windows):
        off = np.array([self.offset[c] + s for c, s, e in windows], dtype=np.int64)
        ln = np.array([e - s for c, s, e in windows], dtype=np.int64)
        return off, ln

    def count_n(self, windows):
        """upper-case 'N' per window [(contig, start, end)] -- Counter(window)['N'] of find_bsj.py:199"""
        return self.count_n_spans(*self._spans(windows))

    def count_n_spans(self, off, ln):
        """count_n for windows given as genome-wide (offset, length) int64 arrays"""
        off = np.ascontiguousarray(off, dtype=np.int64); ln = np.ascontiguousarray(ln, dtype=np.int64)
        n = len(off)
        out = np.zeros(n, dtype=np.int64)
        if n == 0:
            return out
        rc = lib().clh_genome_count_n(self._h, n, off.ctypes.data, ln.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise ClhError('clh_genome_count_n failed (%d): %s' % (rc, last_error()))
        return out

    def set_splice_sites(self, ss_index):
        """Annotated splice sites for splice_signals(): ss_index = {contig: {pos: {strand: {'start'|'end': 1}}}} (the
        reference's splice_site_index, align.py:235-252) or None.  Contigs that are not resident are ignored."""
        flat, cnt = flatten_splice_sites(ss_index, self.offset, self.length)
        rc = lib().clh_genome_set_splice_sites(self._h, flat.ctypes.data, cnt.ctypes.data)
        if rc != 0:
            raise ClhError('clh_genome_set_splice_sites failed (%d): %s' % (rc, last_error()))
        self._sites_of = ss_index

    def splice_signals(self, cands, search_extra=10, shift_threshold=3, is_canonical=True, index_slices=False):
        """K6: cands = [(contig, start, end, clip_base, host_mask)] (or a dict of columns, see below) -> int32 array [n, 8]:
        status, us_free, ds_free, found, strand, us_shift, ds_shift, motif (see include/ciri_long_hip.h).  index_slices: the
        search windows as a minimap2 index serves them (no sequence for a start before the contig) instead of Python slices"""
        if isinstance(cands, dict):       # columns: ctg_off, ctg_len, start, end (int64), clip_base, host_mask (int32)
            off, ln, st, en = (np.ascontiguousarray(cands[k], dtype=np.int64) for k in ('ctg_off', 'ctg_len', 'start', 'end'))
            cb, hm = (np.ascontiguousarray(cands[k], dtype=np.int32) for k in ('clip_base', 'host_mask'))
            n = len(st)
        else:
            n = len(cands)
            off = np.array([self.offset[c[0]] for c in cands], dtype=np.int64)
            ln = np.array([self.length[c[0]] for c in cands], dtype=np.int64)
            st = np.array([c[1] for c in cands], dtype=np.int64)
            en = np.array([c[2] for c in cands], dtype=np.int64)
            cb = np.array([c[3] for c in cands], dtype=np.int32)
            hm = np.array([c[4] for c in cands], dtype=np.int32)
        out = np.zeros((n, 8), dtype=np.int32)
        if n == 0:
            return out
        rc = lib().clh_splice_signal_batch(self._h, n, off.ctypes.data, ln.ctypes.data, st.ctypes.data, en.ctypes.data, cb.ctypes.data,
                                           hm.ctypes.data, search_extra, shift_threshold, (1 if is_canonical else 0) | (2 if index_slices else 0), out.ctypes.data)
        if rc != 0:
            raise ClhError('clh_splice_signal_batch failed (%d): %s' % (rc, last_error()))
        return out

    def plan_windows(self, read_off, win_off, win_len, minus, mat, gap_open, gap_extend, flag=1, score_size=2, want_score2=True,
                     want_cigar=True, mask_len=None):
        """A Plan whose references are windows (genome-wide offset, length, minus-strand flag) of this genome:
        plan.run(d_reads_ptr, genome.codes_ptr, stream)."""
        return Plan(self.ctx, read_off, None, mat, gap_open, gap_extend, flag, score_size, want_score2, want_cigar, mask_len,
                    windows=(np.ascontiguousarray(win_off, dtype=np.int64), np.ascontiguousarray(win_len, dtype=np.int32),
                             np.ascontiguousarray(minus, dtype=np.uint8)))

    def ssw_windows(self, reads, read_off, windows, minus, mat, gap_open, gap_extend, flag=1, score_size=2, want_score2=True,
                    want_cigar=True, mask_len=None, spans=None):
        """ssw_batch with reference k = windows[k] = (contig, start, end), reverse-complemented where minus[k].  spans: the windows as
        genome-wide (offset, length) int64 arrays instead (what _spans(windows) gives)."""
        reads = np.ascontiguousarray(reads, dtype=np.int8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        n = len(read_off) - 1
        off, ln = self._spans(windows) if spans is None else (np.ascontiguousarray(spans[0], dtype=np.int64), np.ascontiguousarray(spans[1], dtype=np.int64))
        ln32 = ln.astype(np.int32)
        rcf = np.ascontiguousarray(np.asarray(minus, dtype=np.uint8))
        o, _keep = self.ctx._opts(mat, gap_open, gap_extend, flag, score_size, want_score2, want_cigar)
        out = np.zeros(n, dtype=ALIGN_DTYPE)
        cap = int(2 * (read_off[-1] if n else 0) + 2 * n + 8) if want_cigar else 1
        cig = np.empty(cap, dtype=np.uint32)     # worst-case capacity; only the used prefix is written
        used = C.c_int64(0)
        ml = np.ascontiguousarray(mask_len, dtype=np.int32) if mask_len is not None else None
        rc = lib().clh_ssw_windows_batch(self._h, n, reads.ctypes.data, read_off.ctypes.data, off.ctypes.data, ln32.ctypes.data, rcf.ctypes.data,
                                         ml.ctypes.data if ml is not None else None, C.byref(o), out.ctypes.data,
                                         cig.ctypes.data if want_cigar else None, cap, C.byref(used))
        if rc != 0:
            raise ClhError('clh_ssw_windows_batch failed (%d): %s' % (rc, last_error()))
        return out, cig[:used.value]


class Plan(object):
    """A batch shape resident on the GPU: run() it on device pointers any number of times."""

    def __init__(self, ctx, read_off, ref_off, mat, gap_open, gap_extend, flag, score_size, want_score2, want_cigar, mask_len,
                 windows=None):
        L = lib()
        self.ctx = ctx
        self.read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        self.n = len(self.read_off) - 1
        self.want_cigar = bool(want_cigar)
        o, self._mat = ctx._opts(mat, gap_open, gap_extend, flag, score_size, want_score2, want_cigar)
        ml = np.ascontiguousarray(mask_len, dtype=np.int32) if mask_len is not None else None
        if windows is not None:
            self._win = windows
            self._h = L.clh_ssw_plan_windows(ctx._h, self.n, self.read_off.ctypes.data, windows[0].ctypes.data, windows[1].ctypes.data,
                                             windows[2].ctypes.data, ml.ctypes.data if ml is not None else None, C.byref(o))
        else:
            self.ref_off = np.ascontiguousarray(ref_off, dtype=np.int64)
            self._h = L.clh_ssw_plan(ctx._h, self.n, self.read_off.ctypes.data, self.ref_off.ctypes.data,
                                     ml.ctypes.data if ml is not None else None, C.byref(o))
        if not self._h:
            raise ClhError('clh_ssw_plan failed: %s' % last_error())

    def run(self, d_reads_ptr, d_refs_ptr, stream=0):
        """stream: a hipStream_t handle (e.g. torch.cuda.Stream().cuda_stream).  0 selects libclh's own private stream, which
        is NOT ordered with torch's default stream: pass the stream your inputs are produced on."""
        rc = lib().clh_ssw_run(self._h, C.c_void_p(d_reads_ptr), C.c_void_p(d_refs_ptr), C.c_void_p(stream))
        if rc != 0:
            raise ClhError('clh_ssw_run failed (%d): %s' % (rc, last_error()))

    def fetch(self):
        out = np.zeros(self.n, dtype=ALIGN_DTYPE)
        cap = int(2 * (self.read_off[-1] if self.n else 0) + 2 * self.n + 8) if self.want_cigar else 1
        cig = np.empty(cap, dtype=np.uint32)     # worst-case capacity; only the used prefix is ever written or touched
        used = C.c_int64(0)
        rc = lib().clh_ssw_fetch(self._h, out.ctypes.data, cig.ctypes.data if self.want_cigar else None, cap, C.byref(used))
        if rc != 0:
            raise ClhError('clh_ssw_fetch failed (%d): %s' % (rc, last_error()))
        return out, cig[:used.value]

    def set_refs_bytes(self, nbytes):
        """state the size of the refs buffer handed to run() (include/ciri_long_hip.h: padding contract of d_refs); -1 = unstated"""
        lib().clh_plan_set_refs_bytes(self._h, int(nbytes))

    def set_profiling(self, on=True):
        lib().clh_plan_set_profiling(self._h, 1 if on else 0)

    def segments(self):
        """[(rows-per-lane class, alignments, read bases, ref bases)] in launch order"""
        rv = np.zeros(32, dtype=np.int32); cnt = np.zeros(32, dtype=np.int32)
        rb = np.zeros(32, dtype=np.int64); fb = np.zeros(32, dtype=np.int64)
        ns = lib().clh_plan_segments(self._h, 32, rv.ctypes.data, cnt.ctypes.data, rb.ctypes.data, fb.ctypes.data)
        return [(int(rv[k]), int(cnt[k]), int(rb[k]), int(fb[k])) for k in range(ns)]

    def traceback_counts(self):
        """(alignments handed to the wide row traceback, alignments handed on to the anti-diagonal traceback) of the last run"""
        c = np.zeros(2, dtype=np.int32)
        if lib().clh_plan_traceback_counts(self._h, c.ctypes.data) != 0:
            raise ClhError('clh_plan_traceback_counts: %s' % last_error())
        return int(c[0]), int(c[1])

    def prefilter_stats(self):
        """what the exact column prefilter (csrc/ssw_prefilter.hip) did in the last run: alignments of the sliced scan class, of
        them with candidate slices, slices run, window columns computed, window columns of the class, alignments that also went
        through the second stage (indel-distance bound)"""
        c = np.zeros(6, dtype=np.int64)
        if lib().clh_plan_prefilter_stats(self._h, c.ctypes.data) != 0:
            raise ClhError('clh_plan_prefilter_stats: %s' % last_error())
        return dict(zip(('alignments', 'pruned', 'slices', 'cols_computed', 'cols_window', 'second_stage'), (int(x) for x in c)))

    def prefilter_timing(self):
        """ssw_prefilter_kernel alone in the last profiling run, per long-window class (reads up to 254 bases, longer reads):
        [(ms, window columns x W words, the same x (11 W + 8) instructions per column and lane), ...]"""
        ms = np.zeros(2, dtype=np.float32); w = np.zeros(4, dtype=np.int64)
        if lib().clh_plan_prefilter_timing(self._h, ms.ctypes.data, w.ctypes.data) != 0:
            raise ClhError('clh_plan_prefilter_timing: %s' % last_error())
        return [(float(ms[k]), int(w[2 * k]), int(w[2 * k + 1])) for k in range(2)]

    def timing(self):
        """([K1 ms per segment], (K1b small-window ms, K1b large-window ms)) for the last run"""
        a = np.zeros(32, dtype=np.float32); b = np.zeros(2, dtype=np.float32)
        ns = lib().clh_plan_timing(self._h, 32, a.ctypes.data, b.ctypes.data)
        if ns < 0:
            raise ClhError('clh_plan_timing: %s' % last_error())
        return [float(a[k]) for k in range(ns)], (float(b[0]), float(b[1]))

    def results_dev_ptr(self):
        return lib().clh_ssw_results_dev(self._h)

    def close(self):
        if getattr(self, '_h', None):
            if getattr(self.ctx, '_h', None):
                lib().clh_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CcsPlan(object):
    """Consensus step for a batch shape resident on the GPU (K2 + K3)."""

    def __init__(self, ctx, read_off):
        self.ctx = ctx
        self.read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        self.n = len(self.read_off) - 1
        self._h = lib().clh_ccs_plan_create(ctx._h, self.n, self.read_off.ctypes.data)
        if not self._h:
            raise ClhError('clh_ccs_plan_create failed: %s' % last_error())

    def run(self, d_reads_ptr, stream=0):
        """stream: see Plan.run"""
        rc = lib().clh_ccs_run(self._h, C.c_void_p(d_reads_ptr), C.c_void_p(stream))
        if rc != 0:
            raise ClhError('clh_ccs_run failed (%d): %s' % (rc, last_error()))

    def timing(self):
        """(K2 ms, K3 ms) of the last run"""
        ms = np.zeros(2, dtype=np.float32)
        if lib().clh_ccs_plan_timing(self._h, ms.ctypes.data) != 0:
            raise ClhError('clh_ccs_plan_timing: %s' % last_error())
        return float(ms[0]), float(ms[1])

    def info(self):
        """dict(slots, slot_bytes, big_slots, big_slot_bytes, ran_in_claimed_big_slot, ran_in_second_launch) of the last run"""
        out = np.zeros(6, dtype=np.int64)
        if lib().clh_ccs_plan_info(self._h, out.ctypes.data) != 0:
            raise ClhError('clh_ccs_plan_info: %s' % last_error())
        return dict(zip(('slots', 'slot_bytes', 'big_slots', 'big_slot_bytes', 'ran_in_claimed_big_slot', 'ran_in_second_launch'), (int(x) for x in out)))

    def stats(self):
        """dict(dp_cells, dp_row_steps, dropped = {status: reads}) of the last run; `dropped` lists the reads a limit of the kernel left
        without a consensus (status 1 workspace, 2 graph limits, 3 output, 4 (unused since round 4), 5 back-track guard, 6 16-bit
        range, 7 alignment without a base)"""
        out = np.zeros(16, dtype=np.int64)
        if lib().clh_ccs_plan_stats(self._h, out.ctypes.data) != 0:
            raise ClhError('clh_ccs_plan_stats: %s' % last_error())
        return {'dp_cells': int(out[0]), 'dp_row_steps': int(out[1]), 'band_misses': int(out[2]), 'dropped': {k: int(out[2 + k]) for k in range(1, 8) if out[2 + k]}}

    def results_dev(self):
        """device pointers (rows, segs, ccs) of the last run's outputs; ccs is packed at the read offsets"""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        if lib().clh_ccs_results_dev(self._h, C.byref(a), C.byref(b), C.byref(c)) != 0:
            raise ClhError('clh_ccs_results_dev: %s' % last_error())
        return a.value, b.value, c.value

    def fetch(self):
        out = np.zeros(self.n, dtype=CCS_DTYPE)
        segs = np.zeros((self.n, CCS_SEG_CAP, 2), dtype=np.int32)
        ccs = np.zeros(max(1, int(self.read_off[-1])), dtype=np.int8)
        rc = lib().clh_ccs_fetch(self._h, out.ctypes.data, segs.ctypes.data, ccs.ctypes.data)
        if rc != 0:
            raise ClhError('clh_ccs_fetch failed (%d): %s' % (rc, last_error()))
        return out, segs, ccs

    def close(self):
        if getattr(self, '_h', None):
            if getattr(self.ctx, '_h', None):
                lib().clh_ccs_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get('CIRI_LONG_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
