"""numpy model of ssw_traceback_rows.hip (K1b, row form): one band row per step, lanes = band offsets o = j - i + w, the
upper neighbour as the previous row shifted by one offset, F as a prefix maximum, score-only band doubling, 4-bit codes of
the final band, stale codes outside the band resolved over the iterations (an earlier one recomputed with codes).
tests/test_traceback_rows_model.py holds it to the oracle's CIGARs (oracle/ssw_oracle.c:banded_traceback; reference:
libs/striped_smith_waterman/ssw.c:548-735)."""
import numpy as np

NEG = -(1 << 30)


def band_pass(ref, read, mat, n, gO, gE, w, want_codes):
    """one band iteration.  Returns (max H, codes) with codes[i] = array over offsets 0..2w of (sel | E opened << 2 |
    F opened << 3), sel 0 diagonal / 1 E / 2 F; cells outside the reference hold 0 and are never read."""
    refLen, readLen = len(ref), len(read)
    W = 2 * w + 1
    o = np.arange(W)
    Hp = np.zeros(W, dtype=np.int64); Ep = np.zeros(W, dtype=np.int64)
    itmax = 0
    codes = np.zeros((readLen, W), dtype=np.int8) if want_codes else None
    for i in range(readLen):
        j = i - w + o
        valid = (j >= 0) & (j < refLen)
        # the previous row seen from one offset lower (offset o+1 of row i-1 is cell (i-1, j)); beyond the top offset: 0
        Hu = np.concatenate((Hp[1:], [0])); Eu = np.concatenate((Ep[1:], [0]))
        if i >= 1 and i - 1 <= w and refLen - 1 < i + w:          # ssw.c:596: the sentinel sits on the live last column
            oc = refLen - 1 - i + w
            if 0 <= oc < W:
                Hu[oc] = 0; Eu[oc] = 0
        t1 = Hu - gO; t2 = Eu - gE
        e = np.maximum(t1, t2); de3 = t1 > t2
        e1 = np.maximum(e, 0)
        s = np.array([mat[int(ref[jj]) * n + int(read[i])] if v else 0 for jj, v in zip(j, valid)], dtype=np.int64)
        td = Hp + s                                               # diagonal: same offset of the previous row
        X = np.maximum(e1, td)
        c = np.where(valid, X - gO, -gE)                          # what a cell offers its right neighbour's F; absent cell: H = F = 0
        # f[o] = max over k < o of c[k] - (o-1-k) gE, and the cell left of offset 0 (H = F = 0): -gE - o gE
        A = c + o * gE
        pm = np.concatenate(([NEG], np.maximum.accumulate(A)[:-1]))
        f = np.maximum(pm - (o - 1) * gE, -gE - o * gE)
        h = np.where(valid, np.maximum(X, f), 0)
        if want_codes:
            fm = np.where(valid, f, 0)
            dd = h - fm - (gO - gE)                               # (h - gO) - (f - gE): what the right neighbour compares
            ddl = np.concatenate(([-(gO - gE)], dd[:-1]))
            df5 = ddl > 0
            f1 = np.maximum(f, 0)
            gt = np.maximum(e1, f1) > td
            ef = e1 > f1
            codes[i] = np.where(gt, np.where(ef, 1, 2), 0) | (de3.astype(np.int8) << 2) | (df5.astype(np.int8) << 3)
        itmax = max(itmax, int(h.max()))
        Hp = h; Ep = np.where(valid, e, 0)
    return itmax, codes


def tb_rows(ref, read, score, mat, n, gO, gE):
    """CIGAR (list of len << 4 | op, op 0 M / 1 I / 2 D) as banded_sw returns it, or None for the reference's traceback error"""
    refLen, readLen = len(ref), len(read)
    w0 = abs(refLen - readLen) + 1
    w, maxv, niter, covered = w0, 0, 0, False
    while True:
        niter += 1
        if not covered:
            it, _ = band_pass(ref, read, mat, n, gO, gE, w, False)
            maxv = max(maxv, it)
            covered = w >= readLen and w >= refLen
        w *= 2
        if not (maxv < score and w < 2 * readLen):
            break
    w //= 2
    planes = {w: band_pass(ref, read, mat, n, gO, gE, w, True)[1]}
    i, j, state, run, ops, op, prev = readLen - 1, refLen - 1, 2, 0, [], 0, 0
    while i > 0:
        if 0 <= j < refLen and i - w <= j <= i + w:
            nb = int(planes[w][i][j - i + w])
        else:                                                     # the reference reads the byte at that flat index: another cell's codes
            wd = 2 * w + 1
            C = i * wd + (j - max(i - w, 0))
            if C < 0:
                return None
            nb = -1
            for k in range(niter - 1, -1, -1):
                wk = w0 << k; wdk = 2 * wk + 1
                ii, pos = divmod(C, wdk)
                if ii >= readLen:
                    continue
                jj = max(ii - wk, 0) + pos
                if jj > min(ii + wk, refLen - 1):
                    continue
                if wk not in planes:
                    planes[wk] = band_pass(ref, read, mat, n, gO, gE, wk, True)[1]
                nb = int(planes[wk][ii][jj - ii + wk])
                break
            if nb < 0:
                return None
        sel = nb & 3
        cE = 3 if nb & 4 else 2; cF = 5 if nb & 8 else 4
        c = (1 if sel == 0 else (cE if sel == 1 else cF)) if state == 2 else (cE if state == 0 else cF)
        if c == 1: i -= 1; j -= 1; state = 2; op = 0
        elif c == 2: i -= 1; state = 0; op = 1
        elif c == 3: i -= 1; state = 2; op = 1
        elif c == 4: j -= 1; state = 1; op = 2
        else: j -= 1; state = 2; op = 2
        if op == prev:
            run += 1
        else:
            ops.append((run << 4) | prev); prev = op; run = 1
    if op == 0:
        ops.append(((run + 1) << 4) | 0)
    else:
        ops.append((run << 4) | op); ops.append((1 << 4) | 0)
    return ops[::-1]


# ---- round 4: the wide form (csrc/ssw_traceback_rows.hip: tb_rows_pass<.., NW>, tb_rows_wide_pass / tb_rows_wide_one) -------------------
def band_row_f_split(c, gE, nw):
    """F of one band row when the offsets are split over nw waves: every wave takes the inclusive prefix maximum of its own part in a
    frame common to the row (A[o] = c[o] + o gE), publishes its total, and starts from the largest total of the waves below it (or
    from the cell left of offset 0).  Returns f as band_pass computes it in one piece."""
    W = len(c)
    o = np.arange(W)
    A = c + o * gE
    fill = -2 * gE                                                 # the cell left of offset 0 (H = F = 0: it offers -gE) in the same frame: f[o] + (o - 1) gE
    bounds = [W * k // nw for k in range(nw + 1)]
    incl = np.empty(W, dtype=np.int64)
    totals = []
    for k in range(nw):
        part = A[bounds[k]:bounds[k + 1]]
        incl[bounds[k]:bounds[k + 1]] = np.maximum.accumulate(part) if len(part) else part
        totals.append(int(part.max()) if len(part) else NEG)
    f = np.empty(W, dtype=np.int64)
    for k in range(nw):
        carry = max([fill] + totals[:k])
        lo, hi = bounds[k], bounds[k + 1]
        if hi > lo:
            excl = np.concatenate(([carry], np.maximum(incl[lo:hi - 1], carry)))      # exclusive prefix within the wave, carry in front
            f[lo:hi] = excl - (o[lo:hi] - 1) * gE
    return f


def doubling_from_state(it_of, score, readLen, w0, narrow_cells=512, nspec=3):
    """The band doubling of ssw.c:560-632 as the two launches run it: the narrow launch until a band does not fit `narrow_cells`
    (state = band, running maximum, iterations done), then rounds of `nspec` iterations side by side whose maxima are replayed, then
    one by one.  it_of(w) = the iteration's maximum.  Returns (final w, iterations, passes run, passes in a row)."""
    w, maxv, niter, covered, ran = w0, 0, 0, False, 0
    handed = False
    while True:
        niter += 1
        if 2 * w + 1 > narrow_cells:
            handed = True
            niter -= 1
            break
        ran += 1                                                   # (covered iterations hold the same cells: same value)
        maxv = max(maxv, it_of(w))
        w *= 2
        if not (maxv < score and w < 2 * readLen):
            w //= 2
            return w, niter, ran, ran
    chain = ran
    assert handed
    first = True
    while True:
        its = []
        for j in range(nspec if first else 1):                     # the passes of the round: all that the doubling CAN reach
            wj = w << j
            if j == 0 or wj < 2 * readLen:
                its.append(it_of(wj)); ran += 1
        chain += 1
        done = False
        for v in its:
            niter += 1; maxv = max(maxv, v)
            if not (maxv < score and 2 * w < 2 * readLen):
                done = True
                break
            w *= 2
        if done:
            return w, niter, ran, chain
        first = False


def walk_ops_with_runs(codes, w, readLen, refLen):
    """the walk inside the final band with runs of diagonal moves taken at once (the kernel: one ballot over the rows above);
    None when the walk leaves the band (the stale reads are not this function's subject)"""
    i, j, state, run, ops, op, prev = readLen - 1, refLen - 1, 2, 0, [], 0, 0
    while i > 0:
        if not (0 <= j < refLen and i - w <= j <= i + w):
            return None
        if state == 2:
            r = 0
            while r < 64 and i - r > 0 and j - r >= 0 and (int(codes[i - r][j - i + w]) & 3) == 0:
                r += 1
            if r >= 2:
                if prev != 0:
                    ops.append((run << 4) | prev); prev = 0; run = 0
                run += r; i -= r; j -= r; op = 0
                continue
        nb = int(codes[i][j - i + w])
        sel = nb & 3
        cE = 3 if nb & 4 else 2; cF = 5 if nb & 8 else 4
        c = (1 if sel == 0 else (cE if sel == 1 else cF)) if state == 2 else (cE if state == 0 else cF)
        if c == 1: i -= 1; j -= 1; state = 2; op = 0
        elif c == 2: i -= 1; state = 0; op = 1
        elif c == 3: i -= 1; state = 2; op = 1
        elif c == 4: j -= 1; state = 1; op = 2
        else: j -= 1; state = 2; op = 2
        if op == prev:
            run += 1
        else:
            ops.append((run << 4) | prev); prev = op; run = 1
    if op == 0:
        ops.append(((run + 1) << 4) | 0)
    else:
        ops.append((run << 4) | op); ops.append((1 << 4) | 0)
    return ops[::-1]
