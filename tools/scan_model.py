"""numpy model of K1s (csrc/ssw_scan.hip): the 8-bit Smith-Waterman pass walked row by row with the reference columns in
parallel -- the horizontal gap E as a prefix maximum in the frame where an extension costs nothing, per column the running
maximum over the rows kept as (H << 8 | 255 - row), the window in chunks that hand their last column to the next chunk, and
the forward pass of a long window cut into overlapping slices.  tests/test_scan_model.py holds it to the oracle
(oracle/ssw_oracle.c) on the alignments of the kernel's class: readLen <= 254, max_match * readLen + bias < 255.

Reference: libs/striped_smith_waterman/ssw.c:123-345 (sw_sse2_byte), 779-849 (ssw_align); row-major statement
oracle/rowmajor_spec.c."""
import numpy as np

NEG = -(1 << 30)


def scan_pass(ref, read, mat, n, gapO, gapE, rows, terminate=None, chunk=256, own0=0):
    """ref, read: int arrays of codes; rows >= len(read): rows processed (wildcard rows score 0).  Columns below own0 are
    computed but do not count.  Returns (max, col, row, colmax) with col = first column whose maximum is the largest
    (columns up to the first whose maximum equals `terminate`), row = smallest row holding it there (clamped to len-1)."""
    R, L = len(ref), len(read)
    key = np.zeros(R, dtype=np.int64)
    bH = np.zeros(rows, dtype=np.int64)          # last column of the previous chunk per row: H
    bE = np.zeros(rows, dtype=np.int64)          #                                           E entering the next chunk
    stop_after = None
    for c0 in range(0, R, chunk):
        c1 = min(R, c0 + chunk)
        W = c1 - c0
        prof = np.zeros((6, W), dtype=np.int64)
        for q in range(min(n, 6)):
            prof[q] = [mat[int(b) * n + q] if b < n else 0 for b in ref[c0:c1]]
        Hp = np.zeros(W, dtype=np.int64); F = np.zeros(W, dtype=np.int64); k = np.zeros(W, dtype=np.int64)
        nbH = np.zeros(rows, dtype=np.int64); nbE = np.zeros(rows, dtype=np.int64)
        prev_hb = 0
        j = np.arange(W)
        for i in range(rows):
            q = int(read[i]) if i < L else 5
            q = q if q < n else 5
            diag = np.concatenate(([prev_hb], Hp[:-1]))
            tt = diag + prof[q]
            F = np.maximum(np.maximum(F - gapE, Hp - gapO), 0)
            X = np.maximum(tt, F)
            # E[j] = max(E_in - j*gapE, max over k < j of X[k] - gapO - (j-1-k)*gapE), clamped at 0 -- as a prefix maximum
            # of X[k] + k*gapE (what the kernel does per virtual lane, lane and wave)
            A = X + j * gapE
            pm = np.concatenate(([NEG], np.maximum.accumulate(A)[:-1]))
            E = np.maximum(np.maximum(pm - gapO - (j - 1) * gapE, bE[i] - j * gapE), 0)
            H = np.maximum(X, E)
            k = np.maximum(k, H * 256 + (255 - i))
            nbH[i] = H[-1]
            nbE[i] = max(E[-1] - gapE, H[-1] - gapO, 0)
            prev_hb = bH[i]
            Hp = H
        bH, bE = nbH, nbE
        key[c0:c1] = k
        if terminate is not None:
            cm = k >> 8
            hit = np.nonzero((cm == terminate) & (np.arange(c0, c1) >= own0))[0]
            if len(hit):
                stop_after = c0 + int(hit[0])
                break
    colmax = key >> 8
    last = stop_after if stop_after is not None else R - 1
    cand = colmax[own0:last + 1]
    mx = int(cand.max()) if len(cand) else 0
    if mx == 0:
        return 0, -1, 0, colmax
    col = own0 + int(np.argmax(cand))
    row = 255 - int(key[col] & 255)
    return mx, col, min(row, L - 1), colmax


def scan_align(ref, read, mat, n, gapO, gapE, maskl, chunk=256, slice_own=None):
    """forward + second best + reverse as ssw_align does in the 8-bit regime; slice_own: owned columns per window slice
    (None: one pass).  Returns the dict the oracle returns, without the CIGAR."""
    ref = np.asarray(ref); read = np.asarray(read)
    R, L = len(ref), len(read)
    rows = ((L + 15) // 16) * 16
    if slice_own is None:
        mx, col, row, colmax = scan_pass(ref, read, mat, n, gapO, gapE, rows, chunk=chunk)
    else:
        mxm = max(int(v) for v in mat)
        overlap = L + (L * mxm + gapE - 1) // gapE + 32
        best = (0, 1 << 60, 0)
        colmax = np.zeros(R, dtype=np.int64)
        for b in range(0, R, slice_own):
            cb, ce = max(0, b - overlap), min(R, b + slice_own)
            m1, c1, r1, cm = scan_pass(ref[cb:ce], read, mat, n, gapO, gapE, rows, chunk=chunk, own0=b - cb)
            colmax[b:ce] = cm[b - cb:]
            if m1 > best[0] or (m1 == best[0] and m1 > 0 and cb + c1 < best[1]):
                best = (m1, cb + c1, r1)
        mx, col, row = best if best[0] > 0 else (0, -1, 0)
    out = dict(score=mx, ref_end=col, query_end=row, score2=0, ref_end2=0, ref_begin=-1, query_begin=row)
    if mx == 0:
        out.update(ref_end=-1, query_end=0, query_begin=0)
    # second best, ssw.c:325-340 (8-bit): columns outside [end - maskl, end + maskl]
    if maskl >= 15:
        e1 = max(out['ref_end'] - maskl, 0); e2 = min(out['ref_end'] + maskl, R) + 1
        cm2 = colmax.copy(); cm2[e1:e2] = 0
        if len(cm2) and cm2.max() > 0:
            out['score2'] = int(cm2.max()); out['ref_end2'] = int(np.argmax(cm2))
    else:
        out['ref_end2'] = -1
    if mx > 0:
        rL = out['query_end'] + 1
        rrows = ((rL + 15) // 16) * 16
        m2, c2, r2, _ = scan_pass(ref[:col + 1][::-1], read[:rL][::-1], mat, n, gapO, gapE, rrows, terminate=mx, chunk=256)
        if m2 > 0:
            out['ref_begin'] = col - c2; out['query_begin'] = out['query_end'] - r2
    return out
