"""numpy / Python-int model of the exact prefilter in front of K1s on long windows (csrc/ssw_prefilter.hip).

What it is for: `align_clip_segments` aligns a 20..300-base clip against hit +- 200 kb (CIRI_long/find_bsj.py:196-216); the
score pass (ssw.c:123-345, sw_sse2_byte) is a dynamic programme over every cell of that window although the clip matches in
one place.  The prefilter finds, with bit-vector arithmetic, the columns that CAN hold the maximum, and K1s runs on those.

The bound.  Scores: substitution matrix `mat` (largest entry M >= 1), gap of k bases costs gapO + (k - 1) gapE with
gapO >= gapE >= 1.  Let c = min(M, gapE), call a pair (read base q, window base r) "equal" when mat[r][q] > M - c, and let
d(j) be the unit-cost edit distance of the WHOLE clip (L bases) to the best window substring ending at column j (free start
in the window; pairs that are not "equal" cost 1).  Then every local alignment ending at column j scores at most

        H(j) <= M * L - c * d(j).

Proof sketch: take a local alignment of clip[a..b] ending at j with x pairs that are not "equal", i inserted clip bases and
t deleted window bases.  Against the perfect score M * L it loses M for each of the L - (b - a + 1) clip bases it leaves out,
at least c for every not-"equal" pair (M - mat <= ... >= c by the definition of "equal"), at least M + gapE >= c for every
inserted clip base and at least gapE >= c for every deleted window base.  Extending it with the left-out clip bases as
insertions is an edit script of cost x + i + t + (L - (b - a + 1)) ending at j, so d(j) is at most that, and the loss is at
least c times it.  The word regime of the reference (ssw.c:371-546) only ever LOWERS H (rowmajor_spec.c), so the bound holds
there as well.

Use.  d(j) by Myers' bit-vector recurrence (semi-global: the delta entering row 0 is 0), per block of 256 columns only the
minimum is kept.  A pass of K1s around the block with the smallest minimum ATTAINS some score S0 <= the true maximum S*.
Every column with H(j) = S* >= S0 has d(j) <= (M * L - S0) / c, so it lies in a block whose minimum is at most that; K1s on
those blocks (each started `overlap` columns early, like the window slices of clh_api.hip) sees the maximum, its first
column and the smallest row there.  Blocks that fail the test cannot hold the maximum nor tie it.  If too many blocks pass,
the static slices are used: the answer never depends on the filter."""
import numpy as np

from scan_model import scan_pass

PF_B = 256          # columns per block of the minima


def bound_consts(mat, n, gapE):
    M = max(int(v) for v in mat)
    c = min(M, gapE)
    return M, c


def eq_masks(read, mat, n, M, c):
    """per window code r (0..4): bit i set when clip base i and r are "equal" (mat[r][q] > M - c)"""
    out = []
    for r in range(5):
        m = 0
        for i, q in enumerate(read):
            s = int(mat[r * n + int(q)]) if (r < n and int(q) < n) else 0
            if s > M - c:
                m |= 1 << i
        out.append(m)
    return out


def myers_semiglobal(ref, read, mat, n, gapE):
    """d(j) for every column j, by the bit-vector recurrence the kernel runs (one Python int = the clip's rows)"""
    L = len(read)
    M, c = bound_consts(mat, n, gapE)
    peq = eq_masks(read, mat, n, M, c)
    full = (1 << L) - 1
    Pv, Mv, score = full, 0, L
    top = 1 << (L - 1)
    d = np.zeros(len(ref), dtype=np.int64)
    for j, r in enumerate(ref):
        Eq = peq[int(r) if int(r) < 5 else 4]
        Xv = Eq | Mv
        Xh = ((((Eq & Pv) + Pv) & full) ^ Pv) | Eq
        Ph = Mv | (~(Xh | Pv) & full)
        Mh = Pv & Xh
        if Ph & top:
            score += 1
        if Mh & top:
            score -= 1
        Ph = (Ph << 1) & full
        Mh = (Mh << 1) & full
        Pv = Mh | (~(Xv | Ph) & full)
        Mv = Ph & Xv
        d[j] = score
    return d


def semiglobal_dp(ref, read, mat, n, gapE):
    """the same d(j) by the plain dynamic programme (checker of the recurrence above)"""
    L = len(read)
    M, c = bound_consts(mat, n, gapE)
    peq = eq_masks(read, mat, n, M, c)
    col = np.arange(L + 1, dtype=np.int64)
    d = np.zeros(len(ref), dtype=np.int64)
    for j, r in enumerate(ref):
        e = peq[int(r) if int(r) < 5 else 4]
        new = np.zeros(L + 1, dtype=np.int64)
        for i in range(1, L + 1):
            sub = col[i - 1] + (0 if (e >> (i - 1)) & 1 else 1)
            new[i] = min(sub, col[i] + 1, new[i - 1] + 1)
        col = new
        d[j] = col[L]
    return d


def block_minima(d, phase=0):
    """minimum of d over the blocks of PF_B columns; block b = columns [b * PF_B - phase, (b + 1) * PF_B - phase)"""
    R = len(d)
    nb = (R + phase + PF_B - 1) // PF_B
    out = np.full(nb, 1 << 30, dtype=np.int64)
    for b in range(nb):
        lo, hi = max(0, b * PF_B - phase), min(R, (b + 1) * PF_B - phase)
        if hi > lo:
            out[b] = d[lo:hi].min()
    return out


def candidate_runs(dmin, thr, max_run=64):
    """maximal runs of blocks with dmin <= thr, cut at groups of max_run blocks (one ballot of the kernel)"""
    runs = []
    nb = len(dmin)
    for g in range(0, nb, max_run):
        b = g
        while b < min(nb, g + max_run):
            if dmin[b] <= thr:
                e = b
                while e + 1 < min(nb, g + max_run) and dmin[e + 1] <= thr:
                    e += 1
                runs.append((b, e + 1))
                b = e + 1
            else:
                b += 1
    return runs


def prefilter_forward(ref, read, mat, n, gapO, gapE, phase=0, cap=None, chunk=256, two_stage=False, stage2_share=8, stage2_always=False):
    """forward pass of the 8-bit class through the prefilter: (max, col, row, info) as scan_pass returns them for the whole
    window.  info: what the filter did."""
    ref = np.asarray(ref); read = np.asarray(read)
    R, L = len(ref), len(read)
    M, c = bound_consts(mat, n, gapE)
    overlap = L + (L * M + gapE - 1) // gapE + 32
    d = myers_semiglobal(ref, read, mat, n, gapE)
    dmin = block_minima(d, phase)
    nb = len(dmin)
    cap = cap if cap is not None else max(64, nb // 8 + 1)

    def cols_of(b0, b1):
        return max(0, b0 * PF_B - phase), min(R, b1 * PF_B - phase)

    def run_slice(own_b, own_e):
        cb = max(0, own_b - overlap)
        m1, c1, r1, _ = scan_pass(ref[cb:own_e], read, mat, n, gapO, gapE, L, chunk=chunk, own0=own_b - cb)
        return m1, (cb + c1 if m1 > 0 else -1), r1

    seed = int(np.argmin(dmin))
    sb, se = cols_of(seed, seed + 1)
    S0 = run_slice(sb, se)[0]
    info = {'S0': S0, 'seed_block': seed, 'blocks': nb}
    thr = (M * L - S0) // c if S0 > 0 else 1 << 30
    runs = candidate_runs(dmin, thr)
    cost = sum(min(R, e * PF_B - phase) - max(0, b * PF_B - phase) + overlap for b, e in runs)
    own = max(8192, (R + 63) // 64)
    nstatic = (R + own - 1) // own
    info['second_stage'] = False
    # (when the second stage is worth its cost is a matter of time only -- the kernel: candidates over an eighth of the window and a threshold
    # below what the indel distance is on random text, ~0.55 L; stage2_always: whenever the first stage leaves anything, for the tests)
    if two_stage and S0 > 0 and (stage2_always or ((len(runs) > cap or cost >= R + nstatic * overlap or cost > R // stage2_share) and 20 * thr < 11 * L)):
        # the unit-cost bound leaves too much of the window: the indel distance over it (csrc/ssw_scan.hip: pf_pick_emit, ssw_scan_pick2_kernel)
        info['second_stage'] = True
        dmin = block_minima(indel_semiglobal(ref, read, mat, n, gapE), phase)
        runs = candidate_runs(dmin, thr)
        cost = sum(min(R, e * PF_B - phase) - max(0, b * PF_B - phase) + overlap for b, e in runs)
    if S0 == 0 or len(runs) > cap or cost >= R + nstatic * overlap:
        info['pruned'] = False
        slices = [(b, min(R, b + own)) for b in range(0, R, own)]
    else:
        info['pruned'] = True
        slices = [cols_of(b, e) for b, e in runs]
    info['slices'] = len(slices)
    info['cols'] = sum(e - b for b, e in slices)
    best = (0, 1 << 60, 0)
    for b, e in slices:
        m1, c1, r1 = run_slice(b, e)
        if m1 > best[0] or (m1 == best[0] and m1 > 0 and c1 < best[1]):
            best = (m1, c1, r1)
    if best[0] == 0:
        return 0, -1, 0, info
    return best[0], best[1], best[2], info


# ---- reads above the bit-vector's 8 registers: the read in PIECES ------------------------------------------------------------
# Cut a local alignment that ends at column j at the piece borders of the read.  The part inside piece k (rows L_k) ends at
# some column j_k of [j - span, j] (span = L + L M / gapE: what a local alignment can cover) and, charging every gap base to the
# piece it falls in (a gap of n bases costs at least n gapE; one that straddles a border is charged by bases), scores at most
# M L_k - c d_k(j_k), where d_k is the semi-global distance of piece k alone; a piece the alignment does not touch contributes
# 0 <= M L_k - c d_k(anything).  Hence
#        H(j) <= sum over k of ( M L_k - c min over j' in [j - span, j] of d_k(j') )  =  M L - c D(j),
# and per block of columns D(block b) >= sum over k of the minimum of piece k's block minima over blocks b - sb .. b,
# sb = ceil(span / 256).  csrc/ssw_scan_wide.hip (ssw_scanw_seed_kernel / ssw_scanw_pick_kernel) uses exactly this sum.
def piece_rows(L, max_rows=254):
    """row ranges [(row0, rows)] of the pieces: as few as fit max_rows, equal sizes"""
    K = (L + max_rows - 1) // max_rows
    base, extra = divmod(L, K)
    out, r = [], 0
    for k in range(K):
        n = base + (1 if k < extra else 0)
        out.append((r, n)); r += n
    return out


def piecewise_block_bound(ref, read, mat, n, gapE, phase=0, max_rows=254):
    """D per block (the sum of the pieces' windowed block minima) and the number of blocks the window of a piece reaches back"""
    ref = np.asarray(ref); read = np.asarray(read)
    L = len(read)
    M, c = bound_consts(mat, n, gapE)
    span = L + (L * M + gapE - 1) // gapE
    sb = (span + PF_B - 1) // PF_B
    D = None
    for r0, rows in piece_rows(L, max_rows):
        dm = block_minima(myers_semiglobal(ref, read[r0:r0 + rows], mat, n, gapE), phase)
        wm = np.array([dm[max(0, b - sb):b + 1].min() for b in range(len(dm))], dtype=np.int64)
        D = wm if D is None else D + wm
    return D, sb


# ---- second stage: the indel distance (round 4) -------------------------------------------------------------------------------
# Let c = min(M, gapE) as before and call a pair "equal2" when mat[r][q] > M - 2c.  Let d2(j) be the INDEL distance (insertions and
# deletions of unit cost, no substitutions: a pair that is not "equal2" costs an insertion and a deletion, 2) of the whole clip to the best
# window substring ending at column j.  Then H(j) <= M L - c d2(j): against M L an alignment loses at least M >= c for every clip base it
# leaves out or inserts, at least gapE >= c for every deleted window base, and at least 2c for every aligned pair that is not "equal2"
# (by the definition), which the indel script charges as one insertion and one deletion.  d2 >= d, and on random text it is larger by a
# quarter (~0.58 L against ~0.46 L), so clips whose best score is 0.45..0.55 L -- for which EVERY block passes the unit-cost test --
# are cut down to a few per cent of their window (tools/dev/indel_bound_probe.py).  The bit-vector form costs about twice the unit-cost
# one per column, so it runs only on the windows the first stage fails to thin out.
def eq2_masks(read, mat, n, M, c):
    out = []
    for r in range(5):
        m = 0
        for i, q in enumerate(read):
            s = int(mat[r * n + int(q)]) if (r < n and int(q) < n) else 0
            if s > M - 2 * c:
                m |= 1 << i
        out.append(m)
    return out


def indel_semiglobal_dp(ref, read, mat, n, gapE):
    """d2(j) by the plain dynamic programme"""
    L = len(read)
    M, c = bound_consts(mat, n, gapE)
    peq = eq2_masks(read, mat, n, M, c)
    col = np.arange(L + 1, dtype=np.int64)
    d = np.zeros(len(ref), dtype=np.int64)
    for j, r in enumerate(ref):
        e = peq[int(r) if int(r) < 5 else 4]
        new = np.zeros(L + 1, dtype=np.int64)
        for i in range(1, L + 1):
            best = min(col[i] + 1, new[i - 1] + 1)
            if (e >> (i - 1)) & 1:
                best = min(best, col[i - 1])
            new[i] = best
        col = new
        d[j] = col[L]
    return d


def indel_semiglobal(ref, read, mat, n, gapE):
    """d2(j) by the bit-vector recurrence of csrc/ssw_prefilter.hip (pf_column_indel): per column, with v = the vertical deltas of the
    column before (Pv / Mv = +1 / -1) and h the horizontal delta a row hands down (0 into row 1), row i maps its h to
        "equal2":            -v                         (a constant)
        not, v = -1:         +1                         (a constant)
        not, v =  0:         -1 -> 0, 0 -> 1, 1 -> 1    (G)
        not, v = +1:         h                          (I: handed on)
    so h = -1 exactly on the rows reached from a constant -1 through rows of I (set A), h = 0 on the rows reached through I from a
    constant 0, from the top, or from a G row whose input is -1 (set Z), and +1 elsewhere; "reached through a run" is one addition
    as in Myers' algorithm.  The new vertical deltas follow row by row from h shifted down by one."""
    L = len(read)
    M, c = bound_consts(mat, n, gapE)
    peq = eq2_masks(read, mat, n, M, c)
    full = (1 << L) - 1
    Pv, Mv, score = full, 0, L
    top = 1 << (L - 1)
    d = np.zeros(len(ref), dtype=np.int64)

    def through(seed_below, run):          # rows of `run` reached from the bits of seed_below (already shifted onto the run's first row)
        u = seed_below & run
        return (((u + run) & full) ^ run) & run

    for j, r in enumerate(ref):
        Eq = peq[int(r) if int(r) < 5 else 4]
        I = ~Eq & Pv & full
        G = ~(Eq | Pv | Mv) & full
        Km = Eq & Pv
        Kz = Eq & ~Pv & ~Mv & full
        A = Km | through((Km << 1) & full, I)
        Sz = Kz | (G & (A << 1) & full)
        Z = Sz | through(((Sz << 1) | 1) & full, I)
        Ph = ~(A | Z) & full
        Mh = A
        if Ph & top:
            score += 1
        if Mh & top:
            score -= 1
        Phi = (Ph << 1) & full
        Mhi = (Mh << 1) & full
        Pv, Mv = ((Eq & Mhi) | (~Eq & (Pv | (~Pv & ~Mv & ~Phi) | (Mv & Mhi)))) & full, Phi & (Eq | Mv) & full
        d[j] = score
    return d
