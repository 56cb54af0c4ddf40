"""CPU model of K1w on long windows (csrc/ssw_scan_wide.hip, class -4): the read through the edit-distance bound in pieces, a seed
region, the WINDOW's regime decided as ssw.c:804-809 decides it, candidate regions as complete alignments of the read against a stretch
of the window, the best row.  A region task is `align(window[cb:ce], read, score_size)` -- the caller passes the checker (the oracle:
score_size 2 = the reference's order of passes, 1 = the word pass alone, as K1w's force_word) -- so the model states exactly the
decision logic the kernels implement and tests/test_longwin_model.py holds it to the whole-window answer.

Which regime (the reference decides on the whole window: any column at 255 - bias in the byte pass):
  mode 1  M L - c min D + bias < 255: no cell of the window can overflow -- byte regime, tasks as they come;
  mode 2  the seed task overflowed: word regime -- every task runs the word pass only, S0 is the seed's word score;
  mode 3  undecided: the seed also runs in the word regime and S0 is THAT score; every candidate region runs both ways; the byte
          rows are the answer if none overflowed, else the word rows."""
import numpy as np

from prefilter_model import PF_B, bound_consts, candidate_runs, piecewise_block_bound


def longwin_align(ref, read, mat, n, gapO, gapE, align, phase=0, max_rows=254, force_static=False):
    """-> (dict(score, ref_begin, ref_end, query_begin, query_end, word), info).  align(ref_codes, read_codes, score_size) returns the
    oracle's dict (with 'word': the result came from the 16-bit regime) or None (score_size 0 and an overflow)."""
    ref = np.asarray(ref); read = np.asarray(read)
    R, L = len(ref), len(read)
    M, c = bound_consts(mat, n, gapE)
    bias = -min(0, min(int(v) for v in mat))
    span = L + (L * M + gapE - 1) // gapE
    overlap = span + 32
    D, _sb = piecewise_block_bound(ref, read, mat, n, gapE, phase=phase, max_rows=max_rows)
    kb = int(np.argmin(D))
    ub = M * L - c * int(D[kb])

    def cols(b0, b1):
        return max(0, b0 * PF_B - phase), min(R, b1 * PF_B - phase)

    def task(b0, b1, ov, force_word):
        cb = max(0, b0 - ov)
        r = align(ref[cb:b1], read, 1 if force_word else 2)
        return r, cb

    c0, c1 = cols(kb, kb + 1)
    seed, _ = task(c0, c1, overlap, False)
    if ub + bias < 255:
        mode, S0 = 1, seed['score']
    elif seed['word']:
        mode, S0 = 2, seed['score']
    else:
        mode, S0 = 3, task(c0, c1, overlap, True)[0]['score']
    info = {'mode': mode, 'S0': S0, 'ub': ub}
    own = max(8192, 2 * overlap, (R + 63) // 64)
    nstatic = (R + own - 1) // own
    regions = None
    if S0 > 0 and not force_static:
        thr = (M * L - S0) // c
        runs = candidate_runs(D, thr)
        ov_c = min(overlap, L + (M * L - S0) // gapE + 32)
        cost = sum(min(R, e * PF_B - phase) - max(0, b * PF_B - phase) + ov_c for b, e in runs)
        if len(runs) <= 64 and cost < R + nstatic * overlap:
            regions = [cols(b, e) + (ov_c,) for b, e in runs]
    info['pruned'] = regions is not None
    if regions is None:
        regions = [(b, min(R, b + own), overlap) for b in range(0, R, own)]
    info['regions'] = len(regions)
    rows_n, rows_w = [], []
    for b0, b1, ov in regions:
        if mode != 2:
            rows_n.append(task(b0, b1, ov, False))
        if mode != 1:
            rows_w.append(task(b0, b1, ov, True))
    use_word = mode == 2 or (mode == 3 and any(r['word'] for r, _ in rows_n))
    rows = rows_w if use_word else rows_n
    best = None
    for k, (r, cb) in enumerate(rows):
        key = (-r['score'], (r['ref_end'] + cb) if r['score'] > 0 else 1 << 60, k)
        if best is None or key < best[0]:
            best = (key, r, cb)
    r, cb = best[1], best[2]
    out = dict(score=r['score'], ref_begin=r['ref_begin'], ref_end=r['ref_end'], query_begin=r['query_begin'], query_end=r['query_end'], word=r['word'])
    if r['score'] > 0:
        out['ref_end'] += cb
        if out['ref_begin'] >= 0:
            out['ref_begin'] += cb
    return out, info
