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
