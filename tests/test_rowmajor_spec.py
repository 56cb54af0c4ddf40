"""oracle/rowmajor_spec.c (the row-major form the HIP kernels implement) must equal the literal stripe
emulation of oracle/ssw_oracle.c pass by pass, including the 16-bit lazy-F truncation when gapO <= gapE."""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import encode, make_mat, mask_len, oracle


def _passes(ref, q, scheme, word, ref_dir=0, terminate=None):
    lib = oracle()
    m, x, o, e = scheme
    mat = make_mat(m, x)
    r = np.ascontiguousarray(encode(ref)); qq = np.ascontiguousarray(encode(q))
    bias = 0 if word else x
    term = (65535 if word else 255) if terminate is None else terminate
    a = (C.c_int32 * 5)()
    lib.clo_striped_pass(r.ctypes.data_as(C.c_void_p), ref_dir, len(r), qq.ctypes.data_as(C.c_void_p), len(qq),
                         mat.ctypes.data_as(C.c_void_p), 5, o, e, word, bias, term, mask_len(len(qq)), a)
    b = (C.c_int32 * 5)()
    cm = (C.c_int32 * max(1, len(r)))()
    lib.clo_rowmajor_pass(r.ctypes.data_as(C.c_void_p), ref_dir, len(r), qq.ctypes.data_as(C.c_void_p), len(qq),
                          mat.ctypes.data_as(C.c_void_p), 5, o, e, word, bias, term, b, cm)
    s2 = (C.c_int32 * 2)()
    lib.clo_second_best(cm, len(r), b[1], mask_len(len(qq)), word, s2)
    return list(a), [b[0], b[1], b[2], s2[0], s2[1]], b[3]


def _rnd(rng, n):
    return ''.join('ACGT'[i] for i in rng.integers(0, 4, n))


def _mut(s, rng, p):
    out = []
    for c in s:
        u = rng.random()
        if u < p / 3:
            continue
        if u < 2 * p / 3:
            out.append('ACGT'[rng.integers(4)]); continue
        out.append(c)
        if u < p:
            out.append(_rnd(rng, int(rng.integers(1, 6))))
    return ''.join(out)


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1), (2, 1, 2, 2), (1, 3, 5, 5)])
@pytest.mark.parametrize('word', [0, 1])
def test_rowmajor_equals_striped(scheme, word):
    rng = np.random.default_rng(100 * word + sum(scheme))
    n_overflow = 0
    for it in range(300):
        L = int(rng.choice([17, 40, 90, 200, 330]))
        R = int(rng.choice([60, 400, 800]))
        ref = _rnd(rng, R)
        st = int(rng.integers(0, max(1, R - L)))
        q = _mut(ref[st:st + L], rng, float(rng.choice([0.05, 0.15, 0.3])))
        if rng.random() < 0.2:
            q = q + q[:len(q) // 3]
        if rng.random() < 0.15:
            ref = ref[:R // 2] + 'N' * 7 + ref[R // 2:]
        if not q:
            continue
        a, b, ovf = _passes(ref, q, scheme, word)
        if ovf:
            n_overflow += 1
            assert a[0] == 255 and b[0] == 255
            continue
        assert a == b, (it, L, R, a, b)
        # reverse direction with early termination on the forward score, as ssw_align does (ssw.c:837-849)
        if a[0] > 0 and a[1] >= 0:
            qr = q[:a[2] + 1][::-1]
            ra, rb_, _ = _passes(ref[:a[1] + 1], qr, scheme, word, ref_dir=1, terminate=a[0])
            assert ra[:3] == rb_[:3], (it, 'reverse', ra, rb_)
    if not word:
        assert n_overflow > 0 or scheme[0] == 1
