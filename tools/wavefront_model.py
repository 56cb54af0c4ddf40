"""numpy model of the anti-diagonal wavefront pass the HIP kernel runs (design aid + executable spec).

Not the product and not the oracle: it mirrors the DATAFLOW of ciri_long_amd/csrc/ssw_wavefront.hip
(128 virtual lanes = 64 lanes x {lo,hi} 16-bit halves, RV rows per virtual lane, one column per virtual
lane per step, boundary values handed to the next virtual lane one step later) so that the design can be
checked against oracle/ on the CPU before a kernel is written.  tests/test_wavefront_model.py runs it
against the oracle; the kernel is a transcription of `wf_pass`.

Virtual lane v processes column j = t - v at step t.  It owns slots [v*RV, (v+1)*RV); the read's padded rows
occupy the LAST `rows` slots (leading slots are dummies whose substitution score is -32768 against everything,
which pins their H/E/F at 0).
"""
import numpy as np

NEG = -32768
NV = 128


def sub0(a, b):
    return np.maximum(a - b, 0)


def build_layout(read, mat, n, word, gapO, gapE, RV):
    W = 8 if word else 16
    L = len(read)
    S = (L + W - 1) // W
    rows = S * W
    T = NV * RV
    assert rows <= T
    off = T - rows
    prof = np.zeros((6, NV, RV), dtype=np.int64)
    cut = np.zeros((NV, RV), dtype=bool)
    quirk = bool(word and gapO <= gapE)
    for s in range(T):
        v, k = divmod(s, RV)
        r = s - off
        if r < 0:
            prof[:, v, k] = NEG
        elif r < L:
            for b in range(min(n, 5)):
                prof[b, v, k] = mat[b * n + read[r]]
        # wildcard rows and the null base (5) score 0
        if quirk and r > 0 and r % S == 0:
            cut[v, k] = True
    return dict(W=W, S=S, rows=rows, T=T, off=off, prof=prof, cut=cut, quirk=quirk, L=L)


def wf_pass(refseq, read, mat, n, gapO, gapE, word, bias, terminate, RV=None):
    """refseq: bases in processing order (already reversed for the reverse pass).
    Returns dict(max, col, row, overflow, colmax(list per processed column), ncols_done, exceeded)."""
    L = len(read)
    W = 8 if word else 16
    rows = ((L + W - 1) // W) * W
    if RV is None:
        RV = max(1, (rows + NV - 1) // NV)
    lay = build_layout(read, mat, n, word, gapO, gapE, RV)
    prof, cut, off = lay['prof'], lay['cut'], lay['off']
    ncols = len(refseq)
    Hp = np.zeros((NV, RV), dtype=np.int64)
    E = np.zeros((NV, RV), dtype=np.int64)
    shadow = np.zeros((NV, RV), dtype=np.int64)
    best = np.zeros(NV, dtype=np.int64)
    best_col = np.full(NV, -1, dtype=np.int64)
    out_H = np.zeros(NV, dtype=np.int64)      # values each lane produced last step
    out_C = np.zeros(NV, dtype=np.int64)
    out_M = np.zeros(NV, dtype=np.int64)
    diag_in = np.zeros(NV, dtype=np.int64)    # H of the row above, previous column
    colmax = []
    overflow = False
    exceeded = False
    term_col = None
    vidx = np.arange(NV)
    t = 0
    while True:
        j = t - vidx
        valid = (j >= 0) & (j < ncols)
        base = np.where(valid, np.asarray(refseq, dtype=np.int64)[np.clip(j, 0, max(ncols - 1, 0))] if ncols else 5, 5)
        # boundary values arrive from the neighbour's previous step (lane 0: zeros)
        in_H = np.concatenate(([0], out_H[:-1]))
        in_C = np.concatenate(([0], out_C[:-1]))
        in_M = np.concatenate(([0], out_M[:-1]))
        carry = in_C.copy()
        diag = diag_in.copy()
        cm = np.zeros(NV, dtype=np.int64)
        hf = None
        for k in range(RV):
            s = prof[base, vidx, k]
            f_main = np.where(cut[:, k], 0, carry)
            if word:
                tt = np.clip(diag + s, -32768, 32767)
            else:
                tt = sub0(np.minimum(diag + s + bias, 255), bias)
                tt = np.where(s == NEG, 0, tt)
            hm = np.maximum(np.maximum(tt, E[:, k]), f_main)
            hf = np.maximum(hm, carry) if lay['quirk'] else hm
            cm = np.maximum(cm, hm)
            diag = Hp[:, k].copy()
            Hp[:, k] = hf
            hg = sub0(hm, gapO)
            E[:, k] = np.maximum(sub0(E[:, k], gapE), hg)
            carry = np.maximum(sub0(f_main, gapE), hg)
        diag_in = in_H
        out_H, out_C = hf, carry
        out_M = np.maximum(in_M, cm)
        cmv = np.where(valid, cm, 0)
        if not word and np.any(cmv + bias >= 255):
            overflow = True
            break
        upd = cmv > best
        if np.any(cmv > terminate):
            exceeded = True
        best = np.where(upd, cmv, best)
        best_col = np.where(upd, j, best_col)
        shadow[upd] = Hp[upd]
        jl = t - (NV - 1)
        if 0 <= jl < ncols:
            colmax.append(int(out_M[NV - 1]))
            if colmax[-1] == terminate:
                term_col = jl
                break
            if jl == ncols - 1:
                break
        if ncols == 0:
            break
        t += 1
    if overflow:
        return dict(max=255, col=-1, row=0, overflow=True, colmax=colmax, exceeded=exceeded, term_col=None, RV=RV)
    if term_col is not None and exceeded:
        # rare (only possible when the 16-bit truncation makes the reverse pass see scores above the forward
        # score): lanes ahead of the last lane may have recorded columns past the stop column.  Redo without them.
        return wf_pass(refseq[:term_col + 1], read, mat, n, gapO, gapE, word, bias, 1 << 30, RV)
    mx = int(best.max())
    if mx == 0:
        return dict(max=0, col=-1, row=0, overflow=False, colmax=colmax, exceeded=exceeded, term_col=term_col, RV=RV)
    cand = np.where(best == mx)[0]
    c = int(best_col[cand].min())
    v = int(cand[best_col[cand] == c].min())
    k = int(np.where(shadow[v] == mx)[0][0])
    row = v * RV + k - off
    row = min(row, L - 1)
    return dict(max=mx, col=c, row=row, overflow=False, colmax=colmax, exceeded=exceeded, term_col=term_col, RV=RV)


def second_best(colmax_by_pos, refLen, end_ref, maskLen, word):
    sc, pos = 0, 0
    edge = max(end_ref - maskLen, 0)
    for i in range(edge):
        if colmax_by_pos[i] > sc:
            sc, pos = colmax_by_pos[i], i
    edge = min(end_ref + maskLen, refLen)
    for i in range(edge + (0 if word else 1), refLen):
        if colmax_by_pos[i] > sc:
            sc, pos = colmax_by_pos[i], i
    return sc, pos


def wf_align(ref, read, mat, n, gapO, gapE, maskLen, score_size=2):
    """Scores and coordinates of ssw_align (flag bit 0 set), decided the way the kernel decides them:
    16-bit pass first when the read is long enough to overflow 8 bits, 8-bit pass only when needed."""
    ref = np.asarray(ref, dtype=np.int64)
    read = np.asarray(read, dtype=np.int64)
    bias = int(abs(min(0, int(np.min(mat)))))
    refLen = len(ref)
    best_possible = int(np.max(mat)) * len(read)
    regime = None
    fw = None
    if score_size == 1:
        fw = wf_pass(ref, read, mat, n, gapO, gapE, 1, 0, 65535)
        regime = 1
    else:
        likely_word = score_size == 2 and best_possible + bias >= 255
        fw_w = None
        if likely_word:
            fw_w = wf_pass(ref, read, mat, n, gapO, gapE, 1, 0, 65535)
            if fw_w['max'] + bias >= 255:
                fw, regime = fw_w, 1     # truncated DP <= exact DP, so the 8-bit pass would have overflowed
        if fw is None:
            fw_b = wf_pass(ref, read, mat, n, gapO, gapE, 0, bias, 255)
            if not fw_b['overflow']:
                fw, regime = fw_b, 0
            elif score_size == 0:
                return None
            else:
                fw = fw_w if fw_w is not None else wf_pass(ref, read, mat, n, gapO, gapE, 1, 0, 65535)
                regime = 1
    score1 = fw['max']
    if score1 == 0:
        ref_end1 = 0 if regime else -1
        read_end1 = 0
    else:
        ref_end1, read_end1 = fw['col'], fw['row']
    cmpos = list(fw['colmax']) + [0] * (refLen - len(fw['colmax']))
    if maskLen >= 15:
        score2, ref_end2 = second_best(cmpos, refLen, ref_end1, maskLen, regime)
    else:
        score2, ref_end2 = 0, -1
    # reverse pass, ssw.c:837-849
    rl = read_end1 + 1
    rr = read[:rl][::-1]
    rref = ref[:ref_end1 + 1][::-1]
    rv = wf_pass(rref, rr, mat, n, gapO, gapE, regime, bias if not regime else 0, score1)
    if rv['max'] == 0:
        ref_begin1 = 0 if regime else -1
        rrow = 0
    else:
        ref_begin1 = ref_end1 - rv['col']
        rrow = rv['row']
    read_begin1 = read_end1 - rrow
    return dict(score=score1, score2=score2, ref_begin=ref_begin1, ref_end=ref_end1, query_begin=read_begin1,
                query_end=read_end1, ref_end2=ref_end2, regime=regime)
