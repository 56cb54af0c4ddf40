"""CPU-side checks of the C-ABI boundary: libclh.so loads, exports every symbol the headers declare, and fails loudly
(no CPU fallback) when there is no GPU.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'ciri_long_amd', 'libclh.so')


def _declared(header):
    txt = open(os.path.join(ROOT, 'include', header)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = re.findall(r'\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;', txt)
    return sorted(set(n for n in names if n not in ('defined',)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(SO):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(SO)


def test_every_declared_symbol_is_exported(lib):
    want = _declared('ciri_long_hip.h') + _declared('ssw_legacy.h')
    assert 'clh_ssw_batch' in want and 'ssw_align' in want and len(want) >= 20
    missing = [n for n in want if not hasattr(lib, n)]
    assert not missing, missing


def test_legacy_struct_layout_matches_reference_ctypes_mirror():
    """s_align as bound by ssw_wrap.py:29-37 (uint16 x2, int32 x5, pointer, int32)."""
    class SAlign(C.Structure):
        _fields_ = [('score1', C.c_uint16), ('score2', C.c_uint16), ('ref_begin1', C.c_int32), ('ref_end1', C.c_int32),
                    ('read_begin1', C.c_int32), ('read_end1', C.c_int32), ('ref_end2', C.c_int32),
                    ('cigar', C.POINTER(C.c_uint32)), ('cigarLen', C.c_int32)]
    assert C.sizeof(SAlign) == 40 and SAlign.cigar.offset == 24
    from ciri_long_amd import hip
    assert hip.ALIGN_DTYPE.itemsize == 36


def test_cigar_helpers_match_reference_table(lib):
    lib.cigar_int_to_op.restype = C.c_char
    lib.cigar_int_to_op.argtypes = [C.c_uint32]
    lib.cigar_int_to_len.restype = C.c_uint32
    lib.cigar_int_to_len.argtypes = [C.c_uint32]
    for code, ch in enumerate('MIDNSHP=X'):
        assert lib.cigar_int_to_op((37 << 4) | code) == ch.encode()
        assert lib.cigar_int_to_len((37 << 4) | code) == 37
    assert lib.cigar_int_to_op(0xf) == b'M'      # ssw.c:891-893


def test_encode_matches_wrapper_table(lib):
    from ciri_long_amd import hip
    s = b'ACGTNacgtnRYKM-*xU'
    out = np.zeros(len(s), dtype=np.int8)
    lib.clh_encode_dna.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
    lib.clh_encode_dna(s, len(s), out.ctypes.data)
    assert out.tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4] + [4] * 8
    assert hip.encode(s).tolist() == out.tolist()


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ciri_long_amd import hip, ssw_wrap
    with pytest.raises(hip.HipUnavailable):
        hip.Context(0)
    with pytest.raises(hip.HipUnavailable):
        ssw_wrap.Aligner('ACGTACGT', 1, 1, 1, 1).align('ACGT')
    # legacy symbol: NULL + message, like the reference's error convention
    lib.ssw_init.restype = C.c_void_p
    lib.ssw_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int8]
    lib.ssw_align.restype = C.c_void_p
    lib.ssw_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32]
    q = np.zeros(20, dtype=np.int8); m = hip.score_matrix(1, 1)
    p = lib.ssw_init(q.ctypes.data, 20, m.ctypes.data, 5, 2)
    assert lib.ssw_align(p, q.ctypes.data, 20, 1, 1, 1, 0, 0, 15) is None
    lib.init_destroy.argtypes = [C.c_void_p]
    lib.init_destroy(p)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ciri_long_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'ciri_long_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert 'oracle_lib' not in txt and 'liboracle' not in txt and '_ref/libssw' not in txt, os.path.join(dp, f)


def test_fastx_count_is_host_only_and_counts_like_the_reference_loop(tmp_path):
    """clh_fastx_count needs no GPU: records of FASTA / FASTQ files as find_ccs.py:51-64 counts them (a trailing header
    without its sequence line is a record), across refills of the 4 MiB read buffer, plain and gzipped"""
    import gzip
    from ciri_long_amd import hip
    fa = tmp_path / 'a.fa'
    fa.write_text(''.join('>r%d\r\n%s\r\n' % (k, 'ACGT' * (k % 7 + 1)) for k in range(1000)) + '>dangling')
    assert hip.fastx_count(str(fa), 0) == 1001
    fq = tmp_path / 'b.fq'
    with open(fq, 'w') as f:
        for k in range(9000):
            s = 'ACGTTGCA' * 70
            f.write('@q%d\n%s\n+\n%s\n' % (k, s, 'I' * len(s)))
    assert fq.stat().st_size > 2 * (4 << 20) and hip.fastx_count(str(fq), 1) == 9000
    gz = tmp_path / 'c.fastq.gz'
    with gzip.open(gz, 'wt') as f:
        for k in range(300):
            f.write('@q%d\nACGT\n+\nIIII\n' % k)
    assert hip.fastx_count(str(gz), 1) == 300
    empty = tmp_path / 'e.fa'
    empty.write_text('')
    assert hip.fastx_count(str(empty), 0) == 0
