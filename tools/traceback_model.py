"""numpy model of ssw_traceback.hip (anti-diagonal banded traceback); design aid, checked against the oracle."""
import numpy as np


def ad_first_row(a, w):
    t = a - w
    return (t + 1) >> 1 if t >= 0 else -((-t) >> 1)


def tb_model(ref, read, score, mat, n, gO, gE, verbose=False):
    refLen, readLen = len(ref), len(read)
    w = abs(refLen - readLen) + 1
    nAD = readLen + refLen - 1
    maxv = 0
    mat = np.asarray(mat).reshape(n, n).astype(np.int64)
    ref = np.asarray(ref, dtype=np.int64); read = np.asarray(read, dtype=np.int64)
    hist = []
    while True:
        H = np.zeros((3, readLen + 1), dtype=np.int64); E = np.zeros((2, readLen + 1), dtype=np.int64); F = np.zeros((2, readLen + 1), dtype=np.int64)
        dirs = {}
        for a in range(nAD):
            cur, p1, p2, e0, e1 = a % 3, (a + 2) % 3, (a + 1) % 3, a & 1, (a & 1) ^ 1
            ilo = max(ad_first_row(a, w), 0, a - (refLen - 1))
            ihi = min((a + w) >> 1, readLen - 1, a)
            if ihi < ilo:
                continue
            i = np.arange(ilo, ihi + 1); j = a - i
            up_ok = (i >= 1) & (j <= i - 1 + w) & ~((i - 1 <= w) & (refLen - 1 < i + w) & (j == refLen - 1))
            hu = np.where(up_ok, H[p1][np.maximum(i - 1, 0)], 0); eu = np.where(up_ok, E[e1][np.maximum(i - 1, 0)], 0)
            hd = np.where((i >= 1) & (j >= 1), H[p2][np.maximum(i - 1, 0)], 0)
            l_ok = (j >= 1) & (j - 1 >= i - w)
            hl = np.where(l_ok, H[p1][i], 0); fl = np.where(l_ok, F[e1][i], 0)
            t1 = np.where(i == 0, -gO, hu - gO); t2 = np.where(i == 0, -gE, eu - gE)
            e = np.maximum(t1, t2); de = np.where(t1 > t2, 3, 2)
            t1 = hl - gO; t2 = fl - gE
            f = np.maximum(t1, t2); df = np.where(t1 > t2, 5, 4)
            e1v = np.maximum(e, 0); f1v = np.maximum(f, 0)
            t1 = np.maximum(e1v, f1v); t2 = hd + mat[ref[j], read[i]]
            h = np.maximum(t1, t2)
            dh = np.where(t1 <= t2, 1, np.where(e1v > f1v, de, df))
            maxv = max(maxv, int(h.max()))
            H[cur][i] = h; E[e0][i] = e; F[e0][i] = f
            dirs[a] = (ilo, dh, de, df)
        hist.append((w, dirs))
        w *= 2
        if not (maxv < score and w < 2 * readLen):
            break
    w //= 2
    if verbose:
        print('final band', w, 'max', maxv)
    i, j, state = readLen - 1, refLen - 1, 2
    ops = []
    while i > 0:
        if j < 0 or j > i + w or j < i - w or j >= refLen:
            # flat-array aliasing of the reference (see ssw_traceback.hip:hist_lookup)
            C = i * (2 * w + 1) + (j - max(i - w, 0))
            c = None
            if C >= 0:
                for wk, dk in reversed(hist):
                    wd = 2 * wk + 1
                    ii, pos = divmod(C, wd)
                    if ii >= readLen:
                        continue
                    jj = max(ii - wk, 0) + pos
                    if jj > min(ii + wk, refLen - 1):
                        continue
                    ilo, dh, de, df = dk[ii + jj]
                    kk = ii - ilo
                    c = int(dh[kk]) if state == 2 else (int(de[kk]) if state == 0 else int(df[kk]))
                    break
            if c is None:
                return None, ('unwritten', i, j, w)
        else:
            ilo, dh, de, df = dirs[i + j]
            k = i - ilo
            c = int(dh[k]) if state == 2 else (int(de[k]) if state == 0 else int(df[k]))
        if c == 1: i -= 1; j -= 1; state = 2; op = 'M'
        elif c == 2: i -= 1; state = 0; op = 'I'
        elif c == 3: i -= 1; state = 2; op = 'I'
        elif c == 4: j -= 1; state = 1; op = 'D'
        elif c == 5: j -= 1; state = 2; op = 'D'
        else:
            return None, ('badcode', i, j, c)
        ops.append(op)
    return ops, w
