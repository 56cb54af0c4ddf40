"""GPU parity tests proper: the HIP path (through the C ABI of libclh.so) against the golden vectors captured from the
reference and against the CPU oracle on seeded random batches.  Bit-exact: scores, coordinates, second-best, CIGARs."""
import numpy as np
import pytest

from oracle_lib import cigar_to_string, oracle_align

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from ciri_long_amd import hip
    return hip.Context(0)


def _run(ctx, refs, queries, scheme, **kw):
    from ciri_long_amd import hip
    m, x, o, e = scheme
    rd, ro = hip.pack(queries)
    fd, fo = hip.pack(refs)
    rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(m, x), o, e, **kw)
    return rows, cig


def _row_tuple(r):
    return (int(r['score1']), int(r['score2']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']),
            int(r['read_end1']), int(r['ref_end2']))


def test_golden_vectors_bit_exact(ctx, golden_cases):
    cases = [c for c in golden_cases if len(c['query']) <= 4096]        # the 430 kb query has its own test below
    by_scheme = {}
    for c in cases:
        by_scheme.setdefault((c['match'], c['mismatch'], c['gap_open'], c['gap_extend']), []).append(c)
    checked = 0
    for scheme, cs in by_scheme.items():
        rows, cig = _run(ctx, [c['ref'] for c in cs], [c['query'] for c in cs], scheme)
        for c, r in zip(cs, rows):
            got = _row_tuple(r)
            want = (c['score'], c['raw_score2'], c['ref_begin'], c['ref_end'], c['query_begin'], c['query_end'], c['raw_ref_end2'])
            assert got == want, (c['name'], got, want)
            assert r['status'] & ~1 == 0, (c['name'], r['status'])
            cg = cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]
            assert len(cg) == c['raw_cigar_len'], c['name']
            assert cigar_to_string(cg, c['query_begin'], c['query_end'], len(c['query'])) == c['cigar_string'], c['name']
            checked += 1
    assert checked >= 1400


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


@pytest.mark.parametrize('wide', [True, False])
@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1), (3, 5, 7, 7)])
def test_random_batch_vs_oracle(ctx, scheme, wide, monkeypatch):
    if not wide:                                 # the anti-diagonal classes instead of K1w
        monkeypatch.setenv('CLH_NO_SCANW', '1')
    rng = np.random.default_rng(sum(scheme) * 13 + 1)
    refs, qs = [], []
    for _ in range(300):
        L = int(rng.choice([1, 16, 17, 40, 127, 129, 255, 256, 400, 513, 700, 1030, 1500]))
        R = int(rng.choice([1, 50, 400, 1000, 2000]))
        ref = _rnd(rng, R)
        st = int(rng.integers(0, max(1, R - L)))
        q = _mut(ref[st:st + L], rng, float(rng.choice([0.03, 0.13, 0.3])))
        if rng.random() < 0.2:
            q = q + q[:len(q) // 2]
        if rng.random() < 0.1:
            q = _rnd(rng, L)
        if rng.random() < 0.15:
            ref = ref[:R // 2] + 'N' * int(rng.integers(1, 12)) + ref[R // 2:]
        if rng.random() < 0.05:
            q = q[:len(q) // 2] + 'N' + q[len(q) // 2:]
        if not q:
            q = 'A'
        refs.append(ref); qs.append(q[:4096])
    rows, cig = _run(ctx, refs, qs, scheme)
    for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
        want = oracle_align(ref, q, *scheme)
        got = _row_tuple(r)
        assert got == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                       want['query_end'], want['ref_end2']), (k, len(q), len(ref), got, want)
        cg = [int(x) for x in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]]
        assert cg == want['cigar'], (k, len(q), len(ref))


def test_production_shape_testfa(ctx, testfa):
    """tests/test.fa in the orientation of find_bsj.py:196-205: 437-nt clip against a 430 kb window."""
    seq1, seq2 = testfa
    rows, cig = _run(ctx, [seq2], [seq1], (1, 1, 1, 1))
    r = rows[0]
    assert _row_tuple(r) == (349, 140, 229790, 230207, 20, 436, 230425)
    cg = cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]
    assert cigar_to_string(cg, 20, 436, len(seq1)).startswith('20S6M1D2M2I2M1I1M1I5M3I7M2I3M1D5M3D96M1D2M2D8M1D5M1D1M1D58M1I118M2D38M2I48M')


def test_production_shape_testfa_call_path_options(ctx, testfa, golden_cases):
    """The same alignment the way find_bsj.py:204-216 asks for it (no second best, no CIGAR wanted): the 437-base clip is above the
    8-bit class, the 430 kb window above 32 kb -- K1w tasks behind the prefilter (class -4), the window's regime decided for the window.
    Expected values: the reference library's own (Appendix B of SURVEY.md / the golden vectors).  And every golden case whose window has
    32 kb or more, again with call-path options, against its recorded answer."""
    from ciri_long_amd import hip
    seq1, seq2 = testfa
    rd, ro = hip.pack([seq1]); fd, fo = hip.pack([seq2])
    plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=False, want_cigar=True)
    assert [rv for rv, _c, _a, _b in plan.segments()] == [-4]
    plan.close()
    rows, cig = _run(ctx, [seq2], [seq1], (1, 1, 1, 1), want_score2=False, want_cigar=True)
    r = rows[0]
    assert (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])) == (349, 229790, 230207, 20, 436)
    cg = cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]
    assert cigar_to_string(cg, 20, 436, len(seq1)).startswith('20S6M1D2M2I2M1I1M1I5M3I7M2I3M1D5M3D96M1D2M2D8M1D5M1D1M1D58M1I118M2D38M2I48M')
    # tests/golden/long_window_golden.json.gz: 44 clips against windows of 33..402 kb, answers of the reference library itself
    # (make_long_window_golden.py; a window is rebuilt from its recipe), by scoring scheme in one batch each -- both long-window classes
    import gzip
    import json
    import os
    import sys
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, gdir)
    from make_long_window_golden import build_window
    with gzip.open(os.path.join(gdir, 'long_window_golden.json.gz'), 'rt') as f:
        cases = json.load(f)['cases']
    by_scheme = {}
    for c in cases:
        by_scheme.setdefault((c['match'], c['mismatch'], c['gap_open'], c['gap_extend']), []).append(c)
    seen = set()
    for scheme, cs in by_scheme.items():
        refs = [build_window(c['seed'], c['length'], c['patches']) for c in cs]
        qs = [c['query'] for c in cs]
        rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
        plan = ctx.plan(ro, fo, hip.score_matrix(scheme[0], scheme[1]), scheme[2], scheme[3], flag=1, score_size=2, want_score2=False, want_cigar=True)
        seen |= {rv for rv, _c, _a, _b in plan.segments()}
        plan.close()
        rows, cig = _run(ctx, refs, qs, scheme, want_score2=False, want_cigar=True)
        for c, r in zip(cs, rows):
            assert (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])) == \
                (c['score'], c['ref_begin'], c['ref_end'], c['query_begin'], c['query_end']), c['name']
            cg = cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]
            assert cigar_to_string(cg, c['query_begin'], c['query_end'], len(c['query'])) == c['cigar_string'], c['name']
    assert seen == {-1, -4}


def test_options_score_size_flag_and_skips(ctx):
    rng = np.random.default_rng(99)
    refs = [_rnd(rng, 500) for _ in range(40)]
    qs = [_mut(r[100:100 + int(rng.choice([40, 300]))], rng, 0.1) for r in refs]
    for score_size in (0, 1, 2):
        rows, _ = _run(ctx, refs, qs, (1, 1, 1, 1), score_size=score_size)
        for ref, q, r in zip(refs, qs, rows):
            want = oracle_align(ref, q, 1, 1, 1, 1, score_size=score_size)
            if want is None:
                assert r['status'] & 2
            else:
                assert _row_tuple(r) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'],
                                         want['query_begin'], want['query_end'], want['ref_end2'])
    rows, cig = _run(ctx, refs, qs, (1, 1, 1, 1), flag=0)
    for ref, q, r in zip(refs, qs, rows):
        want = oracle_align(ref, q, 1, 1, 1, 1, flag=0)
        assert (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])) == \
            (want['score'], -1, want['ref_end'], -1, want['query_end'])
        assert r['cigar_len'] == 0
    # what CIRI-long's call path needs (find_bsj.py:204-224): no second best, no cigar
    rows, cig = _run(ctx, refs, qs, (1, 1, 1, 1), want_score2=False, want_cigar=False)
    for ref, q, r in zip(refs, qs, rows):
        want = oracle_align(ref, q, 1, 1, 1, 1)
        assert (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])) == \
            (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end'])


@pytest.mark.parametrize('flag', [2, 4, 8, 6, 10, 12, 14, 3, 5])
def test_flag_bits_with_filters(ctx, flag):
    """ssw_align's flag bits 1 (score filter: begin/CIGAR only if score >= filters) and 2 (distance filter: only if both
    spans <= filterd) and bit 3 (begin positions), with thresholds that split the batch (ssw.c:834, 850): every field
    and every CIGAR (or its absence) equals the scalar statement of ssw.c."""
    rng = np.random.default_rng(500 + flag)
    refs = [_rnd(rng, int(rng.integers(200, 700))) for _ in range(60)]
    qs = [_mut(r[50:50 + int(rng.choice([30, 60, 120, 250]))], rng, float(rng.choice([0.0, 0.1, 0.25]))) for r in refs]
    filters, filterd = 55, 100
    rows, cig = _run(ctx, refs, qs, (1, 1, 1, 1), flag=flag, filters=filters, filterd=filterd)
    n_with = n_without = n_begin = 0
    for ref, q, r in zip(refs, qs, rows):
        want = oracle_align(ref, q, 1, 1, 1, 1, flag=flag, filters=filters, filterd=filterd)
        assert _row_tuple(r) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                                 want['query_end'], want['ref_end2']), (flag, len(q), _row_tuple(r), want)
        assert [int(x) for x in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (flag, len(q))
        n_with += len(want['cigar']) > 0
        n_without += len(want['cigar']) == 0
        n_begin += want['ref_begin'] >= 0
    assert n_begin > 0 and (not (flag & 6) or (n_with > 0 and n_without > 0))       # the thresholds really split the batch


def test_reference_test_ssw_orientation_430kb_query(ctx, testfa):
    """tests/test_ssw.py:5-15 of the reference: Aligner(seq1 (437 nt)) . align(seq2 (430 314 nt)) -- the QUERY is the long
    sequence, so its rows do not fit one launch class and run as 106 row strips of 4096."""
    seq1, seq2 = testfa
    rows, cig = _run(ctx, [seq1], [seq2], (1, 1, 1, 1))
    r = rows[0]
    assert (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])) == \
        (349, 20, 436, 229781, 230207)
    want = oracle_align(seq1, seq2, 1, 1, 1, 1)
    assert _row_tuple(r) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                             want['query_end'], want['ref_end2'])
    assert [int(x) for x in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar']


@pytest.mark.parametrize('wide', [True, False])
@pytest.mark.parametrize('score2', [True, False])
def test_every_row_class_up_to_4096_vs_oracle(ctx, score2, wide, monkeypatch):
    """read lengths that land in each launch class (rows per lane 1..32), with and without the column maxima
    (want_score2 off selects the lean forward pass).  wide: through K1w, the row-scan kernel for long reads (ssw_scan_wide.hip:
    both regimes, the stripe-boundary rule of the word pass, reads of 255..4096 rows); else through the anti-diagonal classes"""
    from ciri_long_amd import hip
    if not wide:
        monkeypatch.setenv('CLH_NO_SCANW', '1')
    rng = np.random.default_rng(31)
    reads, refs = [], []
    for L in (60, 200, 330, 470, 600, 730, 860, 1000, 1200, 1500, 1900, 2400, 3000, 3500, 4090):
        ref = rng.integers(0, 4, int(rng.integers(300, 2200)), dtype=np.int8)
        core = ref[50:50 + min(L, len(ref) - 60)]
        read = np.concatenate([rng.integers(0, 4, (L - len(core)) // 2, dtype=np.int8), core, rng.integers(0, 4, L - len(core) - (L - len(core)) // 2, dtype=np.int8)])
        flip = rng.random(len(read)) < 0.08
        read = np.where(flip, rng.integers(0, 4, len(read), dtype=np.int8), read).astype(np.int8)
        reads.append(read); refs.append(ref)
    for scheme in ((1, 1, 1, 1), (10, 4, 8, 2)):
        rd, ro = hip.pack(reads); fd, fo = hip.pack(refs)
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(scheme[0], scheme[1]), scheme[2], scheme[3], want_score2=score2, want_cigar=False)
        for k in range(len(reads)):
            want = oracle_align(refs[k], reads[k], *scheme)
            got = (int(rows[k]['score1']), int(rows[k]['ref_begin1']), int(rows[k]['ref_end1']), int(rows[k]['read_begin1']), int(rows[k]['read_end1']))
            assert got == (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end']), (k, len(reads[k]), scheme)
            if score2:
                assert (int(rows[k]['score2']), int(rows[k]['ref_end2'])) == (want['score2'], want['ref_end2'])


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2)])
def test_long_reads_row_strips_vs_oracle(ctx, scheme):
    """Reads of 4 097 ... 13 000 bases (2-4 row strips), mutated copies and unrelated pairs, both regimes."""
    rng = np.random.default_rng(sum(scheme) + 5)
    refs, qs = [], []
    for L in (4097, 4200, 5000, 8192, 8193, 9000, 13000):
        R = int(rng.choice([600, 3000, 9000]))
        ref = _rnd(rng, R)
        if rng.random() < 0.7:
            core = (ref * (L // R + 2))[:L]
            q = _mut(core, rng, 0.08)
        else:
            q = _rnd(rng, L)
        refs.append(ref); qs.append(q)
    rows, cig = _run(ctx, refs, qs, scheme, want_cigar=False)
    for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
        want = oracle_align(ref, q, *scheme)
        assert _row_tuple(r) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                                 want['query_end'], want['ref_end2']), (k, len(q), len(ref))


def test_legacy_six_symbols_as_the_reference_wrapper_binds_them(golden_cases):
    """The zero-change route of INTEGRATION.md: the ctypes stub of ssw_wrap.py:54-72,278-288 against libclh.so."""
    import ctypes as C
    import os
    from ciri_long_amd import hip

    class CAlignRes(C.Structure):          # ssw_wrap.py:29-37
        _fields_ = [('score', C.c_uint16), ('score2', C.c_uint16), ('ref_begin', C.c_int32), ('ref_end', C.c_int32),
                    ('query_begin', C.c_int32), ('query_end', C.c_int32), ('ref_end2', C.c_int32),
                    ('cigar', C.POINTER(C.c_uint32)), ('cigarLen', C.c_int32)]
    lib = C.CDLL(hip.SO_PATH)
    lib.ssw_init.restype = C.c_void_p
    lib.ssw_init.argtypes = [C.POINTER(C.c_int8), C.c_int32, C.POINTER(C.c_int8), C.c_int32, C.c_int8]
    lib.init_destroy.restype = None
    lib.init_destroy.argtypes = [C.c_void_p]
    lib.ssw_align.restype = C.POINTER(CAlignRes)
    lib.ssw_align.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32]
    lib.align_destroy.restype = None
    lib.align_destroy.argtypes = [C.POINTER(CAlignRes)]
    lib.cigar_int_to_len.restype = C.c_int32
    lib.cigar_int_to_len.argtypes = [C.c_int32]
    lib.cigar_int_to_op.restype = C.c_char
    lib.cigar_int_to_op.argtypes = [C.c_int32]
    picked = [c for c in golden_cases if c['name'] in ('tiny_gap', 'zero_score', 'n_in_ref', 'periodic_word')] + golden_cases[40:70]
    for c in picked:
        q = hip.encode(c['query']); r = hip.encode(c['ref'])
        mat = hip.score_matrix(c['match'], c['mismatch'])
        qa = (C.c_int8 * len(q))(*q.tolist()); ra = (C.c_int8 * len(r))(*r.tolist()); ma = (C.c_int8 * 25)(*mat.tolist())
        prof = lib.ssw_init(qa, len(q), ma, 5, 2)
        mask = len(q) // 2 if len(q) > 30 else 15
        res = lib.ssw_align(prof, ra, len(r), c['gap_open'], c['gap_extend'], 1, 0, 0, mask)
        assert res, c['name']
        x = res.contents
        assert (x.score, x.score2, x.ref_begin, x.ref_end, x.query_begin, x.query_end, x.ref_end2, x.cigarLen) == \
            (c['score'], c['raw_score2'], c['ref_begin'], c['ref_end'], c['query_begin'], c['query_end'], c['raw_ref_end2'], c['raw_cigar_len']), c['name']
        s = ''.join('%d%s' % (lib.cigar_int_to_len(x.cigar[i]), lib.cigar_int_to_op(x.cigar[i]).decode()) for i in range(x.cigarLen))
        clip_l = '%dS' % x.query_begin if x.query_begin > 0 else ''
        clip_r = '%dS' % (len(q) - x.query_end - 1) if len(q) - x.query_end - 1 else ''
        assert clip_l + s + clip_r == c['cigar_string'], c['name']
        lib.align_destroy(res)
        lib.init_destroy(prof)


def test_ssw_wrap_mirror_behaves_like_the_reference_wrapper(golden_cases):
    from ciri_long_amd import ssw_wrap
    by = {c['name']: c for c in golden_cases}
    c = by['tiny_gap']
    al = ssw_wrap.Aligner(c['ref'], match=c['match'], mismatch=c['mismatch'], gap_open=c['gap_open'], gap_extend=c['gap_extend'],
                          report_secondary=True, report_cigar=True)
    res = al.align(c['query'])
    assert (res.score, res.ref_begin, res.ref_end, res.query_begin, res.query_end, res.cigar_string, res.score2, res.ref_end2) == \
        (c['score'], c['ref_begin'], c['ref_end'], c['query_begin'], c['query_end'], c['cigar_string'], c['score2'], c['ref_end2'])
    assert al.align(c['query'], min_score=10 ** 6) is None and al.align(c['query'], 0, 10 ** 6) is None     # ssw_wrap.py:219-222
    plain = ssw_wrap.Aligner(c['ref'], 1, 1, 1, 1).align(c['query'])
    assert plain.cigar_string is None and plain.score2 is None                                              # defaults report nothing extra
    some = golden_cases[100:130]
    got = ssw_wrap.align_pairs([x['ref'] for x in some], [x['query'] for x in some], some[0]['match'], some[0]['mismatch'],
                               some[0]['gap_open'], some[0]['gap_extend'], report_cigar=True)
    for x, g in zip(some, got):
        if (x['match'], x['mismatch'], x['gap_open'], x['gap_extend']) == (some[0]['match'], some[0]['mismatch'], some[0]['gap_open'], some[0]['gap_extend']):
            assert (g.score, g.ref_begin, g.ref_end, g.query_begin, g.query_end, g.cigar_string) == \
                (x['score'], x['ref_begin'], x['ref_end'], x['query_begin'], x['query_end'], x['cigar_string'])
    batch = al.align_batch([c['query'], c['query'][2:], 'ACGT'])
    assert batch[0].score == c['score'] and len(batch) == 3


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (1, 1, 2, 1), (2, 3, 5, 2), (1, 2, 16, 16), (3, 1, 4, 4)])
def test_row_scan_class_short_reads_vs_oracle(ctx, scheme):
    """K1s (csrc/ssw_scan.hip) takes the alignments whose scores fit the 8-bit pass for certain (ssw.c:804-806): reads up to
    254 bases against windows whose lengths sit on and around its column chunks (256 / 512 / 1024), windows of several
    chunks, repeats that make a later column exceed the forward score in the reverse pass (ssw.c:296 ends at equality
    only), N on both sides; with and without the second-best score (its column maxima include the wildcard rows)."""
    from ciri_long_amd import hip
    m, x, o, e = scheme
    rng = np.random.default_rng(700 + sum(scheme))
    refs, qs = [], []
    for R in [1, 2, 100, 255, 256, 257, 300, 511, 512, 513, 767, 1023, 1024, 1025, 1300, 2047, 2048, 2049, 3100, 5000, 9000]:
        for _ in range(6):
            L = int(rng.choice([1, 2, 15, 16, 17, 33, 64, 65, 100, 127, 128, 129, 200, 239, 240, 241, 253])) if m == 1 else int(rng.integers(1, 250 // m))
            ref = _rnd(rng, R)
            st = int(rng.integers(0, max(1, R - L)))
            q = _mut(ref[st:st + L], rng, float(rng.choice([0.0, 0.05, 0.2])))[:L] or 'A'
            u = rng.random()
            if u < 0.2 and R > 3 * len(q) + 10:     # the clip occurs twice, the second copy cleaner: reverse passes that see larger maxima
                ref = ref[:R // 2] + q + ref[R // 2:R - len(q)]
            elif u < 0.3:
                ref = ref[:R // 3] + 'N' * int(rng.integers(1, 9)) + ref[R // 3:]
            elif u < 0.4:
                q = q[:len(q) // 2] + 'N' + q[len(q) // 2 + 1:]
            elif u < 0.5:
                q = _rnd(rng, len(q))
            refs.append(ref); qs.append(q)
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    for s2 in (True, False):
        plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=s2, want_cigar=True)
        n0 = sum(c for rv, c, _a, _b in plan.segments() if rv == 0)
        # (without the second best, the references of at most 64 columns are K1l's: csrc/ssw_lanes.hip, tests/test_gpu_lanes.py)
        assert n0 == sum(1 for q, ref in zip(qs, refs) if len(q) <= 254 and m * len(q) + x < 255 and (s2 or len(ref) > 64)) and n0 > 0.8 * len(qs)
        plan.close()
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(m, x), o, e, want_score2=s2, want_cigar=True)
        for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
            want = oracle_align(ref, q, *scheme)
            got = _row_tuple(r)
            exp = (want['score'], want['score2'] if s2 else 0, want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end'],
                   want['ref_end2'] if s2 else got[6])
            assert got == exp, (k, len(q), len(ref), got, exp)
            assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (k, len(q), len(ref))


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (2, 2, 3, 1), (10, 4, 8, 2)])
def test_cigars_of_wide_bands_row_traceback_vs_oracle(ctx, scheme):
    """K1b's row form (csrc/ssw_traceback_rows.hip) by band width: reads with one or two long insertions or deletions start
    their band at |refLen - readLen| + 1 (ssw.c:560) -- up to 512 cells in the launch over all alignments, up to 2048 in the
    wide form, beyond that laid out by reference column; whatever is left (walks that leave the band) goes to the
    anti-diagonal kernel.  CIGARs and coordinates equal the oracle's in every class, and the classes are all populated."""
    from ciri_long_amd import hip
    m, x, o, e = scheme
    rng = np.random.default_rng(900 + sum(scheme))
    refs, qs = [], []
    for blk, gap in [(500, 0), (500, 60), (600, 200), (700, 300), (800, 420), (1300, 600), (1500, 900), (1600, 1100)]:
        for _ in range(5):
            R = 2 * blk + gap + 200
            ref = _rnd(rng, R)
            a = int(rng.integers(0, 100)); L1 = blk - int(rng.integers(0, 60)); L2 = blk - int(rng.integers(0, 60))
            mode = rng.random()
            if mode < 0.45:                                 # deletion in the read: two blocks of the reference, `gap` apart
                q = ref[a:a + L1] + ref[a + L1 + gap:a + L1 + gap + L2]
            elif mode < 0.9:                                # insertion in the read
                q = ref[a:a + L1] + _rnd(rng, gap) + ref[a + L1:a + L1 + L2]
            else:                                           # both, on either side of a third block
                g2 = gap // 2
                q = ref[a:a + L1] + _rnd(rng, g2) + ref[a + L1:a + L1 + 300] + ref[a + L1 + 300 + g2:a + L1 + 300 + g2 + L2 // 2]
            refs.append(ref); qs.append(_mut(q, rng, float(rng.choice([0.0, 0.03, 0.08])))[:4090] or 'A')
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    import torch
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    ts = torch.cuda.Stream()
    plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=True, want_cigar=True)
    plan.run(d_r.data_ptr(), d_f.data_ptr(), ts.cuda_stream)
    rows, cig = plan.fetch()
    wide, anti = plan.traceback_counts()
    plan.close()
    nwide = 0
    for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
        want = oracle_align(ref, q, *scheme)
        assert _row_tuple(r) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                                 want['query_end'], want['ref_end2']), (k, len(q), len(ref))
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (k, len(q), len(ref), int(r['status']))
        nwide += abs((want['ref_end'] - want['ref_begin']) - (want['query_end'] - want['query_begin'])) + 1 > 255
    import os
    if not os.environ.get('CLH_NO_TB_ROWS'):      # the A/B switch sends everything through the anti-diagonal kernel
        assert wide >= nwide > 5 and anti <= wide // 2 + 2, (wide, anti, nwide)


@pytest.mark.parametrize('many', [False, True])
def test_wide_bands_many_doublings_and_long_hand_over_lists(ctx, many):
    """The wide form of K1b's row kernel (a workgroup per band pass, ssw_traceback_rows_wide_pass_kernel / _wide_kernel).
    many = False: insertions and deletions that cancel -- the band starts narrow (|refLen - readLen| + 1 is small) and doubles many
    times before it holds the path, so the narrow launch hands over in mid-loop (its state travels), several further iterations run
    side by side, the widest bands are laid out by reference column and the walk reads stale codes of earlier iterations
    (scores 10/9/8/2, inserted bases all A against skipped reference without A: two long gaps beat a diagonal of mismatches).
    many = True: more than 256 alignments handed over at once (one pass each, the rest in turn).  CIGARs equal the oracle's."""
    from ciri_long_amd import hip
    rng = np.random.default_rng(77 + many)
    refs, qs = [], []
    if not many:
        for gap in (150, 300, 420, 640, 900):
            for rep in range(6):
                blk = int(rng.integers(250, 420))
                ref = _rnd(rng, 3 * blk + 2 * gap + 200)
                a = int(rng.integers(0, 60))
                lo = a + blk if not rep & 1 else a + 2 * blk           # the reference stretch the read skips: no A
                ref = ref[:lo] + ref[lo:lo + gap].replace('A', 'C') + ref[lo + gap:]
                skew = int(rng.integers(0, 40)) * (rep % 3)               # net length difference: the first band
                if rep & 1:     # insertion first, then a deletion of about the same length
                    q = ref[a:a + blk] + 'A' * gap + ref[a + blk:a + 2 * blk] + ref[a + 2 * blk + gap - skew:a + 3 * blk + gap - skew]
                else:           # deletion first
                    q = ref[a:a + blk] + ref[a + blk + gap:a + 2 * blk + gap] + 'A' * (gap - skew) + ref[a + 2 * blk + gap:a + 3 * blk + gap]
                refs.append(ref); qs.append(_mut(q, rng, float(rng.choice([0.0, 0.02, 0.06])))[:4090])
    else:
        for _ in range(300):
            blk = int(rng.integers(200, 320)); gap = int(rng.integers(260, 330))
            ref = _rnd(rng, 2 * blk + gap + 100)
            a = int(rng.integers(0, 50))
            q = ref[a:a + blk] + ref[a + blk + gap:a + 2 * blk + gap] if rng.random() < 0.5 else ref[a:a + blk] + 'A' * gap + ref[a + blk:a + 2 * blk]
            refs.append(ref); qs.append(_mut(q, rng, 0.02))
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    import torch
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    ts = torch.cuda.Stream()
    plan = ctx.plan(ro, fo, hip.score_matrix(10, 9), 8, 2, flag=1, score_size=2, want_score2=True, want_cigar=True)
    plan.run(d_r.data_ptr(), d_f.data_ptr(), ts.cuda_stream)
    rows, cig = plan.fetch()
    wide, anti = plan.traceback_counts()
    plan.close()
    for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
        want = oracle_align(ref, q, 10, 9, 8, 2)
        assert _row_tuple(r) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                                 want['query_end'], want['ref_end2']), (k, len(q), len(ref))
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (k, len(q), len(ref), int(r['status']))
    import os
    if not os.environ.get('CLH_NO_TB_ROWS'):
        assert wide > (256 if many else 10), (wide, anti)


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (2, 3, 5, 2), (1, 1, 3, 1)])
def test_long_windows_in_slices_vs_oracle(ctx, scheme):
    """Windows of 32 kb and more are cut into slices whose forward passes run as separate workgroups (ssw_scan.hip:
    ssw_scan_slice_kernel): the clip placed across slice borders and inside the overlaps, twice in one window (first column
    wins on ties, ssw.c:283), absent, and next to N runs -- scores, coordinates, second best and CIGAR equal the oracle's."""
    from ciri_long_amd import hip
    m, x, o, e = scheme
    rng = np.random.default_rng(1200 + sum(scheme))
    refs, qs = [], []
    for R in [32768, 40000, 70000, 150000]:
        own = max(8192, (R + 63) // 64)
        for case in range(7):
            L = int(rng.integers(20, min(240, 250 // m)))
            ref = _rnd(rng, R)
            border = own * int(rng.integers(1, max(2, R // own)))
            pos = [border - L // 2, border - 3, border + 1, border - L - 40, int(rng.integers(0, R - L)), 0, R - L][case]
            pos = max(0, min(R - L, pos))
            q = _mut(ref[pos:pos + L], rng, float(rng.choice([0.0, 0.05, 0.15])))[:L] or 'A'
            if case == 2 and pos > 2 * L + 10:        # an exact second copy earlier in the window: ties resolved by column
                ref = ref[:pos - 2 * L] + ref[pos:pos + L] + ref[pos - L:]
                ref = ref[:R]
            if case == 4:
                q = _rnd(rng, L)
            if case == 3:
                ref = ref[:pos] + 'N' * 7 + ref[pos + 7:]
            refs.append(ref); qs.append(q)
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=True, want_cigar=True)
    assert sum(c for rv, c, _a, _b in plan.segments() if rv == -1) == len(qs)
    plan.close()
    for s2 in (True, False):
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(m, x), o, e, want_score2=s2, want_cigar=True)
        for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
            want = oracle_align(ref, q, *scheme)
            got = _row_tuple(r)
            exp = (want['score'], want['score2'] if s2 else 0, want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end'],
                   want['ref_end2'] if s2 else got[6])
            assert got == exp, (k, len(q), len(ref), got, exp)
            assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (k, len(q), len(ref))


def test_stale_walk_goldens(ctx):
    """the four alignments of tests/golden/stale_walk_golden.json.gz (the reference's traceback reads direction bytes outside
    the final band; in all four the byte belongs to an EARLIER band iteration, which the row kernel computes once more with
    codes): rows and CIGARs equal the reference library's, batched together with ordinary alignments"""
    import gzip
    import json
    import os
    from ciri_long_amd import hip
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stale_walk_golden.json.gz'), 'rt') as f:
        cases = json.load(f)['cases']
    rng = np.random.default_rng(4)
    refs = [c['ref'] for c in cases]; qs = [c['query'] for c in cases]
    for _ in range(12):
        ref = _rnd(rng, 1500); refs.append(ref); qs.append(_mut(ref[200:1100], rng, 0.1))
    rows, cig = _run(ctx, refs, qs, (1, 1, 1, 1))
    for c, r in zip(cases, rows):
        w = c['want']
        assert _row_tuple(r) == (w['score'], w['score2'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end'], w['ref_end2']), (c['rank'], c['index'])
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == w['cigar'], (c['rank'], c['index'], int(r['status']))
    for ref, q, r in zip(refs[len(cases):], qs[len(cases):], rows[len(cases):]):
        want = oracle_align(ref, q, 1, 1, 1, 1)
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar']


@pytest.mark.parametrize('filtered', [True, False])
@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1)])
def test_long_windows_of_the_anti_diagonal_classes_in_slices_vs_oracle(ctx, scheme, filtered, monkeypatch):
    """Reads outside K1s's 8-bit class against windows of 32 kb and more, call-path options (no second best): the alignment
    runs as window-slice tasks of its anti-diagonal class and `ssw_combine_kernel` takes the best slice (clh_api.hip).  The
    read placed across slice borders, inside overlaps, twice (first end column wins), absent; forward windows as packed
    references and minus-strand windows of a resident genome.  Scores, coordinates and CIGARs equal the oracle's."""
    from ciri_long_amd import hip, utils
    if not filtered:                             # without the prefilter: window-slice tasks of the anti-diagonal classes (rounds 2-3)
        monkeypatch.setenv('CLH_NO_PREFILTER', '1')
    m, x, o, e = scheme
    rng = np.random.default_rng(1500 + sum(scheme))
    refs, qs = [], []
    for R in [32768, 50000, 120000]:
        for case in range(6):
            L = int(rng.integers(260, 900)) if m == 1 else int(rng.integers(30, 500))
            ref = _rnd(rng, R)
            overlap = L + (L * m + e - 1) // e + 32
            own = max(8192, 2 * overlap, (R + 63) // 64)
            border = own * int(rng.integers(1, max(2, R // own)))
            pos = [border - L // 2, border - 5, border + 2, border - overlap + 3, int(rng.integers(0, R - L)), R - L][case]
            pos = max(0, min(R - L, pos))
            q = _mut(ref[pos:pos + L], rng, float(rng.choice([0.0, 0.05, 0.12]))) or 'A'
            if case == 2 and pos > 2 * L + 10:
                ref = (ref[:pos - 2 * L] + ref[pos:pos + L] + ref[pos - L:])[:R]      # an exact earlier copy
            if case == 4:
                q = _rnd(rng, L)
            refs.append(ref); qs.append(q[:4000])
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=True)
    # nearly all: reads outside the 8-bit class -- K1w tasks behind the prefilter (-4), or window-slice tasks + combine (-2)
    assert sum(c for rv, c, _a, _b in plan.segments() if rv == (-4 if filtered else -2)) >= len(qs) - 6
    plan.close()
    rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(m, x), o, e, want_score2=False, want_cigar=True)
    for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
        want = oracle_align(ref, q, *scheme)
        got = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
        assert got == (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end']), (k, len(q), len(ref), got)
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (k, len(q), len(ref), int(r['status']))
    # minus-strand windows of a resident genome: the slices walk the genome backwards
    text = _rnd(rng, 140000)
    g = hip.Genome(ctx, {'chr1': text})
    wins, minus, queries, strings = [], [], [], []
    for k in range(8):
        s = int(rng.integers(0, 20000)); e2 = s + int(rng.integers(40000, 110000))
        w = text[s:e2]
        mstrand = k % 2 == 1
        wstr = utils.revcomp(w) if mstrand else w
        L = int(rng.integers(270, 700)) if m == 1 else int(rng.integers(40, 300))
        p0 = int(rng.integers(0, len(wstr) - L))
        queries.append(_mut(wstr[p0:p0 + L], rng, 0.06) or 'A'); strings.append(wstr)
        wins.append(('chr1', s, e2)); minus.append(mstrand)
    qd, qo = hip.pack(queries)
    rows, cig = g.ssw_windows(qd, qo, wins, minus, hip.score_matrix(m, x), o, e, want_score2=False, want_cigar=True)
    for k, (wstr, q, r) in enumerate(zip(strings, queries, rows)):
        want = oracle_align(wstr, q, *scheme)
        got = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
        assert got == (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end']), ('window', k, got)
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], ('window', k)
    g.close()


def test_short_read_spanning_thousands_of_reference_bases(ctx):
    """tests/golden/stale_walk_golden.json.gz, `wide_reference_cases` (found by the fuzz harness): a 315-base read aligned over
    2603 reference bases with 5/4/6/6 -- a band above 2048 cells on a reference above 2048 bases, i.e. the anti-diagonal
    traceback, whose LDS must hold both aligned sequences whatever the read-length class is"""
    import gzip
    import json
    import os
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stale_walk_golden.json.gz'), 'rt') as f:
        cases = json.load(f)['wide_reference_cases']
    for c in cases:
        scheme = (c['match'], c['mismatch'], c['gap_open'], c['gap_extend'])
        rows, cig = _run(ctx, [c['ref']] * 3, [c['query'], c['query'][:100], c['query']], scheme)
        for r in (rows[0], rows[2]):
            w = c['want']
            assert _row_tuple(r) == (w['score'], w['score2'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end'], w['ref_end2'])
            assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == w['cigar'], int(r['status'])


@pytest.mark.parametrize('scheme', [(1, 1, 3, 0), (2, 2, 0, 0), (1, 3, 255, 16), (4, 4, 9, 3)])
def test_row_scan_and_row_traceback_unusual_gap_costs(ctx, scheme):
    """free extensions, free gaps, the largest opening K1s accepts, and a 4 x 4 matrix (no N row) -- score kernel K1s, window
    slices only with a positive extension, row traceback with its 16-bit frames"""
    from ciri_long_amd import hip
    m, x, o, e = scheme
    rng = np.random.default_rng(77 + sum(scheme))
    refs, qs = [], []
    for R in [60, 300, 1100, 34000]:
        for _ in range(5):
            L = int(rng.integers(5, min(250, 250 // m)))
            ref = _rnd(rng, R)
            st = int(rng.integers(0, max(1, R - L)))
            refs.append(ref); qs.append(_mut(ref[st:st + L], rng, float(rng.choice([0.0, 0.1, 0.3]))) or 'A')
    for nmat in (5, 4):
        mat = hip.score_matrix(m, x)
        qq = qs
        if nmat == 4:
            mat = np.array([m if i == j else -x for i in range(4) for j in range(4)], dtype=np.int8)
        rd, ro = hip.pack(qq); fd, fo = hip.pack(refs)
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, mat, o, e, want_score2=True, want_cigar=True)
        for k, (ref, q, r) in enumerate(zip(refs, qq, rows)):
            want = oracle_align(ref, q, m, x, o, e, mat=mat)
            assert _row_tuple(r) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                                     want['query_end'], want['ref_end2']), (nmat, k, len(q), len(ref))
            span = (want['ref_end'] - want['ref_begin'] + 1) + (want['query_end'] - want['query_begin'] + 1)
            if int(r['status']) & 16:        # the stated capacity limit of the CIGAR step (include/ciri_long_hip.h): only beyond 12 kB
                assert span > 12000 and e == 0, (nmat, k, span)
                continue
            assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (nmat, k, int(r['status']))


def test_launch_classes_of_a_mixed_batch(ctx, monkeypatch):
    """which kernel takes what (clh_api.hip: scan_class_ok, scanw_class_ok): short reads whose scores fit 8 bits -> K1s (class 0),
    reads up to 4096 bases on windows below 32768 columns -> K1w (class -3), longer reads and K1w switched off -> the anti-diagonal
    classes; and the word regime's stripe-boundary rows in reads of every residue modulo 8 (S = ceil(L / 8))"""
    from ciri_long_amd import hip
    rng = np.random.default_rng(77)
    reads, refs = [], []
    for L in [30, 200, 254, 255, 256, 257, 258, 259, 260, 261, 262, 263, 500, 1001, 4096, 4200]:
        ref = rng.integers(0, 4, 1500, dtype=np.int8)
        core = ref[100:100 + min(L, 1300)]
        read = np.concatenate([core, rng.integers(0, 4, L - len(core), dtype=np.int8)]).astype(np.int8)
        flip = rng.random(L) < 0.1
        reads.append(np.where(flip, rng.integers(0, 4, L, dtype=np.int8), read).astype(np.int8)); refs.append(ref)
    rd, ro = hip.pack(reads); fd, fo = hip.pack(refs)
    for no_wide in (False, True):
        if no_wide:
            monkeypatch.setenv('CLH_NO_SCANW', '1')
        plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, want_score2=True, want_cigar=True)
        classes = {rv: cnt for rv, cnt, _a, _b in plan.segments()}
        if no_wide:
            assert -3 not in classes and classes[0] == 2 and sum(v for k, v in classes.items() if k > 0) == 14
        else:
            assert classes[0] == 2 and classes[-3] == 13 and sum(v for k, v in classes.items() if k > 0) == 1      # (254 bases: 254 + bias reaches 255)
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(1, 1), 1, 1, want_score2=True, want_cigar=True)
        for k in range(len(reads)):
            want = oracle_align(refs[k], reads[k], 1, 1, 1, 1)
            assert _row_tuple(rows[k]) == (want['score'], want['score2'], want['ref_begin'], want['ref_end'], want['query_begin'],
                                           want['query_end'], want['ref_end2']), (k, len(reads[k]), no_wide)
            assert [int(x) for x in cig[rows[k]['cigar_off']:rows[k]['cigar_off'] + rows[k]['cigar_len']]] == want['cigar']


def _clip_cases(rng, m, R, n):
    """(window, clip) pairs for the call-path shape: a short clip somewhere in a long window -- planted with 0..15 % errors, across
    256-column block borders, at the window's ends, twice (first end column wins, ssw.c:283), next to N runs, and absent"""
    refs, qs = [], []
    for case in range(n):
        L = int(rng.integers(20, min(250, 250 // m)))
        ref = _rnd(rng, R)
        kind = case % 8
        pos = int(rng.integers(0, R - L))
        if kind == 1:
            pos = 256 * int(rng.integers(1, R // 256 - 1)) - L // 2          # across a block border
        if kind == 2:
            pos = 0
        if kind == 3:
            pos = R - L
        q = _mut(ref[pos:pos + L], rng, float(rng.choice([0.0, 0.04, 0.15])))[:L] or 'A'
        if kind == 4 and pos > 3 * L:                                       # an exact copy earlier: the tie goes to the first column
            ref = (ref[:pos - 2 * L] + ref[pos:pos + L] + ref[pos - L:])[:R]
        if kind == 5:
            q = _rnd(rng, L)                                                # absent: nothing to prune with
        if kind == 6:
            ref = ref[:pos + L // 2] + 'N' * 9 + ref[pos + L // 2 + 9:]
            q = q[:L // 3] + 'N' + q[L // 3 + 1:]
        if kind == 7 and pos > 40000:                                       # a weaker copy far away must not win, a stronger one must
            far = pos - 30000
            ref = ref[:far] + _mut(ref[pos:pos + L], rng, 0.3)[:L].ljust(L, 'A') + ref[far + L:]
            ref = ref[:R]
        refs.append(ref); qs.append(q)
    return refs, qs


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (2, 2, 3, 1), (10, 4, 8, 2)])
def test_second_stage_of_the_prefilter_vs_oracle(ctx, scheme, monkeypatch):
    """Clips whose best score leaves the unit-cost bound without grip -- nearly half of the clip foreign, many substitutions, two weak loci --
    on windows of 40..250 kb: the window goes through the indel-distance pass as well (ssw_prefilter_indel_kernel, ssw_scan_pick2_kernel;
    tools/prefilter_model.py "second stage").  Rows equal the oracle's, equal the one-stage run's (CLH_NO_PF2), and the second stage is
    what prunes here."""
    import torch
    from ciri_long_amd import hip
    m, x, o, e = scheme
    rng = np.random.default_rng(2300 + sum(scheme))
    refs, qs = [], []
    for case in range(36):
        R = int(rng.choice([40000, 70000, 131072, 250000]))
        L = int(rng.integers(70, 200)) if m < 10 else int(rng.integers(18, 25))
        ref = _rnd(rng, R)
        pos = int(rng.integers(0, R - L))
        kind = case % 6
        q = ref[pos:pos + L]
        if kind in (0, 1):                       # nearly half of the clip does not belong here
            cut = (L * 9) // 20
            q = (_rnd(rng, cut) + q[cut:]) if kind == 0 else (q[:L - cut] + _rnd(rng, cut))
        if kind == 2:                            # substitutions only, a quarter of the bases: they cost the indel distance 2 each
            q = ''.join(c if rng.random() > 0.27 else 'ACGT'[(('ACGT'.index(c)) + 1 + int(rng.integers(0, 3))) % 4] for c in q)
        if kind == 3:                            # indels and substitutions
            q = _mut(q, rng, 0.45)[:L] or 'A'
        if kind == 4 and pos > 3 * L:            # two weak loci: the same damaged clip earlier in the window (the first column wins a tie)
            q = ''.join(c if rng.random() > 0.3 else 'ACGT'[int(rng.integers(0, 4))] for c in q)
            ref = (ref[:pos - 2 * L] + ref[pos:pos + L] + ref[pos - L:])[:R]
        if kind == 5:                            # N on both sides
            q = _rnd(rng, (L * 2) // 5) + q[(L * 2) // 5:]
            ref = ref[:pos + L // 2] + 'N' * 7 + ref[pos + L // 2 + 7:]
            q = q[:L // 2] + 'N' + q[L // 2 + 1:]
        refs.append(ref); qs.append(q)
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    want = [oracle_align(ref, q, *scheme) for ref, q in zip(refs, qs)]
    got = {}
    for one_stage in (False, True):
        if one_stage:
            monkeypatch.setenv('CLH_NO_PF2', '1')
        plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=False)
        plan.run(d_r.data_ptr(), d_f.data_ptr())
        rows, _ = plan.fetch()
        st = plan.prefilter_stats()
        plan.close()
        got[one_stage] = (rows, st)
        for k, (w, r) in enumerate(zip(want, rows)):
            g = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
            assert g == (w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']), (one_stage, k, len(qs[k]), len(refs[k]), g)
    monkeypatch.delenv('CLH_NO_PF2')
    assert (got[False][0] == got[True][0]).all()
    two, one = got[False][1], got[True][1]
    assert one['second_stage'] == 0
    if m == 1:      # (with M > c the threshold (M L - S0) / c is beyond what either distance reaches on random text: the kernel's rule keeps the second stage out)
        assert two['second_stage'] >= 5 and two['cols_computed'] < one['cols_computed'], (two, one, [round(w['score'] / len(q), 2) for w, q in zip(want, qs)])
    # every window through the second stage, whatever the first left (the rule is a matter of time, not of the answer)
    monkeypatch.setenv('CLH_PF2_ALWAYS', '1')
    plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=False)
    plan.run(d_r.data_ptr(), d_f.data_ptr())
    rows, _ = plan.fetch()
    st = plan.prefilter_stats()
    plan.close()
    assert (rows == got[True][0]).all() and st['second_stage'] >= 10, st       # (longer clips of the larger match scores are K1w's class: one stage)


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (2, 2, 3, 1), (1, 3, 5, 2), (3, 1, 2, 2)])
def test_prefilter_on_long_windows_vs_oracle(ctx, scheme, monkeypatch):
    """The exact column prefilter in front of K1s on windows of 32 kb and more (csrc/ssw_prefilter.hip, find_bsj.py:196-216's
    shape): block minima of the edit-distance bound, a seed pass, K1s on the candidate blocks only.  Rows equal the oracle's with
    the filter and equal the static slices' without it (CLH_NO_PREFILTER); the filter must actually prune the planted clips."""
    import torch
    from ciri_long_amd import hip, utils
    m, x, o, e = scheme
    rng = np.random.default_rng(1700 + sum(scheme))
    refs, qs = [], []
    for R in [32768, 33000, 90000, 200000]:
        a, b = _clip_cases(rng, m, R, 8)
        refs += a; qs += b
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    want = [oracle_align(ref, q, *scheme) for ref, q in zip(refs, qs)]
    got_rows = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv('CLH_NO_PREFILTER', '1')
        plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=False)
        assert sum(c for rv, c, _a, _b in plan.segments() if rv == -1) == len(qs)
        plan.run(d_r.data_ptr(), d_f.data_ptr())
        rows, _ = plan.fetch()
        st = plan.prefilter_stats()
        plan.close()
        got_rows[off] = rows
        for k, (w, r) in enumerate(zip(want, rows)):
            got = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
            assert got == (w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']), (off, k, len(qs[k]), len(refs[k]), got)
        if off:
            assert st['alignments'] == 0 and st['pruned'] == 0
        else:
            # (1/1/1/1: the bound is tight; with larger match scores a substitution still counts c = min(M, gap_extend) only)
            assert st['alignments'] == len(qs) and st['pruned'] >= len(qs) // 2 and st['cols_computed'] < st['cols_window'] // (3 if m == 1 else 1), st
    monkeypatch.delenv('CLH_NO_PREFILTER')
    assert (got_rows[False] == got_rows[True]).all()
    # a refs buffer that is not 256-byte aligned: the filter is off for the run, the static slices run, same rows
    d_f2 = torch.zeros(len(fd) + 256, dtype=torch.uint8, device='cuda')
    d_f2[3:3 + len(fd)] = d_f
    plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=False)
    plan.run(d_r.data_ptr(), d_f2.data_ptr() + 3)
    rows, _ = plan.fetch()
    assert plan.prefilter_stats()['pruned'] == 0 and (rows == got_rows[True]).all()
    plan.close()
    # the padding contract of d_refs (include/ciri_long_hip.h): a stated buffer size that ends inside the last 256-byte block a window
    # touches switches the filter off for the run; a size that covers the block leaves it on; same rows
    last_block_end = ((int(fo[-1]) - 1) >> 8) + 1 << 8
    for stated, on in ((int(fo[-1]), int(fo[-1]) == last_block_end), (last_block_end, True), (-1, True)):
        plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=False)
        plan.set_refs_bytes(stated)
        plan.run(d_r.data_ptr(), d_f.data_ptr())
        rows, _ = plan.fetch()
        assert (plan.prefilter_stats()['pruned'] > 0) == on and (rows == got_rows[True]).all(), stated
        plan.close()
    # windows of a resident genome, both strands (minus-strand windows run down the addresses)
    text = _rnd(rng, 260000)
    text = text[:5000] + text[5000:5600].lower() + text[5600:]
    g = hip.Genome(ctx, {'chr1': text})
    wins, minus, queries, strings = [], [], [], []
    for k in range(16):
        s = int(rng.integers(0, 30000)); e2 = s + int(rng.integers(40000, 220000))
        w = text[s:e2]
        mstrand = k % 2 == 1
        wstr = utils.revcomp(w) if mstrand else w
        L = int(rng.integers(20, min(250, 250 // m)))
        p0 = [int(rng.integers(0, len(wstr) - L)), 0, len(wstr) - L, 256 * 7 - (s & 255) - 5][k // 4 % 4] if k < 14 else int(rng.integers(0, len(wstr) - L))
        p0 = max(0, p0)
        queries.append((_mut(wstr[p0:p0 + L].upper(), rng, 0.08)[:L] or 'A') if k != 13 else _rnd(rng, L)); strings.append(wstr)
        wins.append(('chr1', s, e2)); minus.append(mstrand)
    qd, qo = hip.pack(queries)
    rows, _ = g.ssw_windows(qd, qo, wins, minus, hip.score_matrix(m, x), o, e, want_score2=False, want_cigar=False)
    for k, (wstr, q, r) in enumerate(zip(strings, queries, rows)):
        w = oracle_align(wstr, q, *scheme)
        got = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
        assert got == (w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']), ('window', k, minus[k], got)
    g.close()


@pytest.mark.parametrize('scheme', [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1)])
def test_long_reads_on_long_windows_behind_the_prefilter_vs_oracle(ctx, scheme):
    """K1w on windows of 32 kb and more (csrc/ssw_scan_wide.hip, class -4): the read goes through the bit-vector pass in pieces of
    <= 254 rows, a seed region says which regime the WINDOW is in (ssw.c:804-809 decides on the whole window), candidate regions
    run as K1w tasks.  Cases on both sides of the 8-bit limit -- exact copies (word regime), noisy copies (byte regime), copies
    whose bound allows an overflow that does not happen (both regimes computed), two copies of which only one overflows, reads
    of several pieces, absent reads -- with score_size 2, 1 and 0.  Scores, coordinates and CIGARs equal the oracle's."""
    import torch
    from ciri_long_amd import hip
    m, x, o, e = scheme
    rng = np.random.default_rng(1900 + sum(scheme))
    refs, qs = [], []
    lo = {1: 255, 2: 127, 10: 30}[m]                   # shortest read outside the 8-bit class (max_match * L + bias >= 255)
    for R in [32768, 60000, 130000]:
        for case in range(10):
            L = int(rng.integers(lo, lo + 60)) if case < 6 else int(rng.integers(300, 1300) if m == 1 else rng.integers(100, 700))
            ref = _rnd(rng, R)
            pos = int(rng.integers(0, R - L))
            err = [0.0, 0.02, 0.03, 0.12, 0.0, 0.02, 0.0, 0.05, 0.15, 0.0][case]
            q = _mut(ref[pos:pos + L], rng, err) or 'A'
            if case == 4 and pos > 3 * L + 600:            # a second, noisier copy earlier in the window: one locus overflows, one does not
                far = pos - 2 * L - 500
                ref = (ref[:far] + _mut(ref[pos:pos + L], rng, 0.04)[:L].ljust(L, 'C') + ref[far + L:])[:R]
            if case == 5 and pos > 3 * L:                  # an exact copy earlier: the first end column wins
                ref = (ref[:pos - 2 * L] + ref[pos:pos + L] + ref[pos - L:])[:R]
            if case == 9:
                q = _rnd(rng, L)
            refs.append(ref); qs.append(q[:4000])
    if m == 1:
        # the seed (fewest edits: five substitutions, 262 - 10 = 252) stays below the 8-bit limit while a locus with MORE edits (six
        # extra window bases, 262 - 6 = 256) overflows: the window's regime is the word regime, decided by a region that is not the seed
        # (tests/test_longwin_model.py has the same construction on the CPU)
        import sys as _sys, os as _os
        _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
        from test_longwin_model import _two_loci
        g2 = np.random.Generator(np.random.PCG64(4242))
        for _ in range(4):
            r2, q2 = _two_loci(g2, R=40000)
            refs.append(''.join('ACGT'[v] for v in r2)); qs.append(''.join('ACGT'[v] for v in q2))
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=True)
    assert sum(c for rv, c, _a, _b in plan.segments() if rv == -4) >= len(qs) - 4      # (a noisy copy can come out below the shortest length)
    plan.run(d_r.data_ptr(), d_f.data_ptr())
    rows, cig = plan.fetch()
    st = plan.prefilter_stats()
    plan.close()
    assert st['alignments'] == len(qs) and st['pruned'] >= len(qs) // 2, st
    words = 0
    for k, (ref, q, r) in enumerate(zip(refs, qs, rows)):
        want = oracle_align(ref, q, *scheme)
        got = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
        assert got == (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end']), (k, len(q), len(ref), got, want)
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want['cigar'], (k, len(q), len(ref), int(r['status']))
        words += int(r['status']) & 1
    assert 0 < words < len(qs) or m != 1               # both regimes occur with 1/1/1/1
    # score_size 1 (the word pass alone) and 0 (the byte pass alone: an overflow anywhere in the window is the reference's NULL)
    for ss in (1, 0):
        plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=ss, want_score2=False, want_cigar=False)
        plan.run(d_r.data_ptr(), d_f.data_ptr())
        rows2, _ = plan.fetch()
        plan.close()
        for k, (ref, q, r, r2) in enumerate(zip(refs, qs, rows, rows2)):
            want = oracle_align(ref, q, *scheme, score_size=ss)
            if want is None:
                assert int(r2['status']) & 2, (ss, k)
                continue
            got = (int(r2['score1']), int(r2['ref_begin1']), int(r2['ref_end1']), int(r2['read_begin1']), int(r2['read_end1']))
            assert got == (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end']), (ss, k, len(q), got)
