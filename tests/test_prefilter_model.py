"""tools/prefilter_model.py: the edit-distance bound in front of K1s on long windows (csrc/ssw_prefilter.hip) -- the
bit-vector recurrence against the plain dynamic programme, the bound H(j) <= M L - c d(j) against the exact column maxima
of the 8-bit pass (ssw.c:123-345 as tools/scan_model.py states it, itself held to the oracle by test_scan_model.py), and the
filtered forward pass against the whole-window pass: same maximum, same first column, same smallest row."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from oracle_lib import make_mat  # noqa: E402
from prefilter_model import (PF_B, block_minima, bound_consts, candidate_runs, indel_semiglobal, indel_semiglobal_dp,  # noqa: E402
                             myers_semiglobal, prefilter_forward, semiglobal_dp)
from scan_model import scan_pass  # noqa: E402

SCORINGS = [(1, 1, 1, 1), (2, 2, 3, 1), (1, 3, 5, 2), (3, 1, 2, 2)]


def _mut(s, rng, p):
    out = []
    for c in s:
        u = rng.random()
        if u < p / 3:
            continue
        if u < 2 * p / 3:
            out.append(int(rng.integers(0, 4))); continue
        out.append(int(c))
        if u < p:
            out.extend(int(x) for x in rng.integers(0, 4, int(rng.integers(1, 3))))
    return np.array(out or [0], dtype=np.int8)


def _case(rng, R, L, p, nrich=False, plant=True):
    ref = rng.integers(0, 4, R).astype(np.int8)
    if nrich:
        for _ in range(3):
            a = int(rng.integers(0, max(1, R - 10))); ref[a:a + int(rng.integers(1, 40))] = 4
    if plant:
        st = int(rng.integers(0, max(1, R - L)))
        read = _mut(ref[st:st + L], rng, p)[:L]
    else:
        read = rng.integers(0, 4, L).astype(np.int8)
    if nrich and len(read) > 4:
        read[int(rng.integers(0, len(read)))] = 4
    return ref, read


@pytest.mark.parametrize('sc', SCORINGS)
def test_bit_vector_recurrence_equals_the_plain_programme(sc):
    m, x, go, ge = sc
    mat = make_mat(m, x)
    rng = np.random.Generator(np.random.PCG64(11 + m))
    for _ in range(12):
        L = int(rng.integers(1, 70))
        ref, read = _case(rng, int(rng.integers(1, 300)), L, 0.2, nrich=True)
        assert (myers_semiglobal(ref, read, mat, 5, ge) == semiglobal_dp(ref, read, mat, 5, ge)).all()


@pytest.mark.parametrize('sc', SCORINGS)
def test_the_bound_holds_in_every_column(sc):
    m, x, go, ge = sc
    mat = make_mat(m, x)
    M, c = bound_consts(mat, 5, ge)
    rng = np.random.Generator(np.random.PCG64(23 + m))
    tight = 0
    for k in range(40):
        L = int(rng.integers(1, min(250 // m, 120)))
        ref, read = _case(rng, int(rng.integers(1, 900)), L, float(rng.choice([0.0, 0.1, 0.3])), nrich=k % 3 == 0, plant=k % 4 != 3)
        L = len(read)
        colmax = scan_pass(ref, read, mat, 5, go, ge, L)[3]
        d = myers_semiglobal(ref, read, mat, 5, ge)
        assert (colmax <= M * L - c * d).all()
        tight += int((colmax == M * L - c * d).any())
    assert tight > 0          # an exact copy attains it


@pytest.mark.parametrize('sc', SCORINGS)
def test_filtered_forward_pass_equals_the_whole_window_pass(sc):
    m, x, go, ge = sc
    mat = make_mat(m, x)
    rng = np.random.Generator(np.random.PCG64(37 + m))
    pruned = 0
    for k in range(14):
        L = int(rng.integers(8, min(250 // m, 90)))
        R = int(rng.choice([700, 2049, 5000]))
        ref, read = _case(rng, R, L, float(rng.choice([0.0, 0.08, 0.2])), nrich=k % 4 == 1, plant=k % 5 != 4)
        if k % 3 == 0:                                   # the clip twice: the first column must win the tie
            a = int(rng.integers(0, R - 2 * len(read) - 300))
            ref[a:a + len(read)] = read; ref[a + len(read) + 200:a + 2 * len(read) + 200] = read
        L = len(read)
        phase = int(rng.integers(0, PF_B))
        want = scan_pass(ref, read, mat, 5, go, ge, L)[:3]
        for cap in (None, 1):                            # cap 1: the static slices unless one run suffices
            got = prefilter_forward(ref, read, mat, 5, go, ge, phase=phase, cap=cap)
            assert got[:3] == want, (k, phase, cap, got, want)
            pruned += int(got[3]['pruned'])
    assert pruned > 4


@pytest.mark.parametrize('sc', SCORINGS + [(10, 4, 8, 2)])
def test_indel_distance_bit_vector_recurrence_equals_the_plain_programme(sc):
    """the second stage's recurrence (csrc/ssw_prefilter.hip: pf_column_indel): constants, G rows and runs of I rows"""
    m, x, go, ge = sc
    mat = make_mat(m, x)
    rng = np.random.Generator(np.random.PCG64(51 + m))
    for k in range(30):
        L = int(rng.integers(1, 100))
        ref, read = _case(rng, int(rng.integers(1, 400)), L, float(rng.choice([0.05, 0.2, 0.4])), nrich=k % 3 == 1, plant=k % 4 != 3)
        assert (indel_semiglobal(ref, read, mat, 5, ge) == indel_semiglobal_dp(ref, read, mat, 5, ge)).all(), (k, L)


@pytest.mark.parametrize('sc', SCORINGS + [(10, 4, 8, 2)])
def test_the_indel_distance_bound_holds_in_every_column(sc):
    """H(j) <= M L - c d2(j) against the exact column maxima of the 8-bit pass -- planted clips with substitutions, insertions and
    deletions, partial clips, N on either side"""
    m, x, go, ge = sc
    mat = make_mat(m, x)
    M, c = bound_consts(mat, 5, ge)
    rng = np.random.Generator(np.random.PCG64(61 + m))
    tight = 0
    for k in range(24):
        L = int(rng.integers(4, min(250 // m, 120)))
        R = int(rng.integers(50, 900))
        ref, read = _case(rng, R, L, float(rng.choice([0.0, 0.1, 0.3])), nrich=k % 3 == 1, plant=k % 5 != 4)
        if k % 4 == 2 and len(read) > 20:            # only a part of the clip belongs here
            read[:len(read) // 3] = rng.integers(0, 4, len(read) // 3)
        L = len(read)
        colmax = scan_pass(ref, read, mat, 5, go, ge, L, want_colmax=True)[3]['colmax'] if 'want_colmax' in scan_pass.__code__.co_varnames else None
        d2 = indel_semiglobal(ref, read, mat, 5, ge)
        if colmax is None:
            # column maxima by the plain recurrences (the 8-bit pass without overflow: scores stay below 255 here)
            H = np.zeros(L + 1, dtype=np.int64); E = np.zeros(L + 1, dtype=np.int64)
            colmax = np.zeros(R, dtype=np.int64)
            for j in range(R):
                Hn = np.zeros(L + 1, dtype=np.int64); F = 0
                for i in range(1, L + 1):
                    E[i] = max(H[i] - go, E[i] - ge)
                    F = max(Hn[i - 1] - go, F - ge)
                    s = int(mat[int(ref[j]) * 5 + int(read[i - 1])])
                    Hn[i] = max(0, H[i - 1] + s, E[i], F)
                H = Hn
                colmax[j] = H.max()
        assert (colmax <= M * L - c * d2).all(), (k, L, R)
        tight += int((colmax == M * L - c * d2).any())
    assert tight > 3


@pytest.mark.parametrize('sc', SCORINGS)
def test_two_stage_filtered_forward_pass_equals_the_whole_window_pass(sc):
    """clips whose best score leaves the unit-cost bound useless (a third of the clip replaced, many errors): the second stage takes
    the window, the answer stays the whole-window pass's"""
    m, x, go, ge = sc
    mat = make_mat(m, x)
    rng = np.random.Generator(np.random.PCG64(71 + m))
    second = pruned = 0
    for k in range(14):
        L = int(rng.integers(40, min(250 // m, 110)))
        R = int(rng.choice([3000, 6000, 9000]))
        ref, read = _case(rng, R, L, float(rng.choice([0.1, 0.25, 0.35])), nrich=k % 4 == 1, plant=k % 6 != 5)
        if k % 2 == 0 and len(read) > 30:
            read[len(read) // 2:len(read) // 2 + len(read) // 3] = rng.integers(0, 4, len(read) // 3)
        L = len(read)
        phase = int(rng.integers(0, PF_B))
        want = scan_pass(ref, read, mat, 5, go, ge, L)[:3]
        for always in (False, True):                     # by the kernel's rule, and with the second stage forced
            got = prefilter_forward(ref, read, mat, 5, go, ge, phase=phase, two_stage=True, stage2_share=64, stage2_always=always)
            assert got[:3] == want, (k, phase, always, got, want)
            second += int(got[3]['second_stage']); pruned += int(got[3]['pruned'])
    assert second > 10 and pruned > 6


def test_candidate_runs_cover_exactly_the_blocks_under_the_threshold():
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(50):
        dm = rng.integers(0, 6, int(rng.integers(1, 200)))
        thr = int(rng.integers(0, 6))
        cover = np.zeros(len(dm), dtype=bool)
        for b, e in candidate_runs(dm, thr):
            assert not cover[b:e].any() and e - b <= 64
            cover[b:e] = True
        assert (cover == (dm <= thr)).all()
    d = np.arange(1000)
    assert list(block_minima(d, 0)) == [0, 256, 512, 768] and list(block_minima(d, 100)) == [0, 156, 412, 668, 924]


@pytest.mark.parametrize('sc', SCORINGS)
def test_the_bound_of_a_read_in_pieces_holds_in_every_block(sc):
    """reads above 254 bases go through the bit-vector pass in pieces (csrc/ssw_scan_wide.hip): M L - c D(block) bounds every
    column maximum of the block.  Small pieces on short reads here, so that the exact 8-bit pass is the checker."""
    from prefilter_model import piece_rows, piecewise_block_bound
    m, x, go, ge = sc
    mat = make_mat(m, x)
    M, c = bound_consts(mat, 5, ge)
    rng = np.random.Generator(np.random.PCG64(51 + m))
    assert piece_rows(300) == [(0, 150), (150, 150)] and piece_rows(254) == [(0, 254)] and sum(r for _, r in piece_rows(1000)) == 1000
    tight = 0
    for k in range(24):
        L = int(rng.integers(20, min(250 // m, 110)))
        R = int(rng.choice([300, 1500, 2600]))
        ref, read = _case(rng, R, L, float(rng.choice([0.0, 0.1, 0.25])), nrich=k % 3 == 0, plant=k % 4 != 3)
        L = len(read)
        phase = int(rng.integers(0, PF_B))
        colmax = scan_pass(ref, read, mat, 5, go, ge, L)[3]
        for max_rows in (7, 16, 33):
            D, sb = piecewise_block_bound(ref, read, mat, 5, ge, phase=phase, max_rows=max_rows)
            for b in range(len(D)):
                lo, hi = max(0, b * PF_B - phase), min(R, (b + 1) * PF_B - phase)
                if hi > lo:
                    assert colmax[lo:hi].max() <= M * L - c * D[b], (k, max_rows, b)
                    tight += int(colmax[lo:hi].max() == M * L - c * D[b])
    assert tight > 0
