"""Dataflow model of K3 (csrc/ccs_poa.hip) in numpy -- not the product and not the oracle.

oracle/poa_oracle.c states the aligner the way spoa computes it: five full int32 matrices and a back-track that compares
matrix values.  The kernel cannot keep matrices; it computes a row at a time from its source rows and leaves ONE byte per
cell (two more for rows with several in-edges) from which the same back-track is replayed.  This file is that row-wise
formulation, cell for cell what the kernel does, so that the derivation can be checked on the CPU against the oracle
(tests/test_poa_model.py) before and independently of any GPU run:

* source rows are read as H and the two clamped differences dF = max(F + e - g - H, -1), dO = max(O + c - q - H, -1)
  (a vertical gap state below H - 1 in that frame can never win, tie or be back-tracked into);
* the horizontal states come from two prefix maxima over M0 = max(diagonal, F, O[, 0]) ("hat" values) and the identities
  Q = Qhat, E[j] = max(Ehat[j], Qhat[j-1] + g);
* per cell ONE byte, as the kernel leaves it: bits 0-5 the move spoa's back-track takes out of the cell, as 62 - (its place
  in spoa's checking order): 63 zero (local mode: stop); 62-s diagonal through in-edge s; 50 - 3s - {0,1,2} vertical
  through in-edge s by F+e (the run goes on upwards), H+g, O+c (goes on); 14/13/12 horizontal by E+e (goes on to the
  left), H+g, Q+c (goes on); bit 6 hx: the E or Q chain of this column extends the previous column's; bit 7 vstop: an
  upward run ends with the step out of this cell; rows with several in-edges add the in-edge an upward run leaves
  through.  `dp_row` computes the code twice -- as the maximum of (value << 6 | code) over all candidates (the kernel's
  way) and from equality tests in priority order -- and asserts that both agree.
"""
import numpy as np

NEG = -(1 << 28)
CODE_ZERO, CODE_DIAG, CODE_VERT, CODE_HORZ, B_HX, B_VSTOP = 63, 62, 50, 14, 64, 128


class Params(object):
    def __init__(self, algorithm=0, m=10, n=-4, g=-8, e=-2, q=-24, c=-1):
        if g >= e:
            raise NotImplementedError('linear gap cost (g >= e) is not built into the kernel')
        if g <= q or e >= c:          # affine: one piece (spoa: AlignmentEngine::Create)
            q, c = g, e
        self.algorithm, self.m, self.n, self.g, self.e, self.q, self.c = algorithm, m, n, g, e, q, c


def row0(P, L):
    """H, dF, dO of row 0 (no node), columns 0..L"""
    j = np.arange(L + 1)
    if P.algorithm == 0:
        h = np.zeros(L + 1, dtype=np.int64)
    else:
        h = np.maximum(P.g + (j - 1) * P.e, P.q + (j - 1) * P.c)
        h[0] = 0
    return h, np.full(L + 1, -1, dtype=np.int64), np.full(L + 1, -1, dtype=np.int64)


def prefix_hat(m0, open_, ext):
    """hat[j] = max over 0 <= k < j of m0[k] + open + (j-1-k) ext, j = 1..L (hat[0] = NEG)"""
    L = len(m0) - 1
    k = np.arange(L + 1)
    run = np.maximum.accumulate(m0 - k * ext)              # inclusive prefix maximum in the gap-free frame
    hat = np.full(L + 1, NEG, dtype=np.int64)
    hat[1:] = run[:-1] + open_ - ext + k[1:] * ext
    return hat


def dp_row(P, code, seq, srcs, col0):
    """One row.  srcs = list of (H, dF, dO) of the source rows in in-edge order (row 0 for a node without in-edges);
    col0 = H[r][0].  Returns (H, dF, dO, byte plane, slot plane) for columns 0..L (column 0 of the planes unused)."""
    L = len(seq)
    s = np.where(seq == code, P.m, P.n).astype(np.int64)
    multi = len(srcs) > 1
    d = fe = fo = oe = None
    for k, (h, df, do) in enumerate(srcs):
        dk = h[:-1] + s                                     # H[p][j-1] + s(i, j), j = 1..L
        fek, fok, oek = (h + df)[1:], h[1:], (h + do)[1:]   # Fs, H, Os of the source at column j
        if k == 0:
            d, fe, fo, oe = dk, fek, fok, oek
            kd = np.zeros(L, dtype=np.int64); kfe = kd.copy(); kfo = kd.copy(); koe = kd.copy()
        else:
            kd = np.where(dk > d, k, kd); d = np.maximum(d, dk)
            kfe = np.where(fek > fe, k, kfe); fe = np.maximum(fe, fek)
            kfo = np.where(fok > fo, k, kfo); fo = np.maximum(fo, fok)
            koe = np.where(oek > oe, k, koe); oe = np.maximum(oe, oek)
    fnew, onew = P.g + np.maximum(fe, fo), P.q + np.maximum(oe, fo)
    m0 = np.empty(L + 1, dtype=np.int64)
    m0[0] = col0
    m0[1:] = np.maximum(np.maximum(d, fnew), onew)
    if P.algorithm == 0:
        m0[1:] = np.maximum(m0[1:], 0)
    ehat, qhat = prefix_hat(m0, P.g, P.e), prefix_hat(m0, P.q, P.c)
    H = np.maximum(m0, np.maximum(ehat, qhat))
    H[0] = col0
    E = ehat.copy()
    E[2:] = np.maximum(ehat[2:], qhat[1:-1] + P.g)
    Q = qhat
    h1, hl = H[1:], H[:-1]
    big = 1 << 20
    # (a) the kernel's way: every candidate as (value << 6) | code, one maximum
    def pk(v, code):
        return (np.asarray(v, dtype=np.int64) << 6) + code
    pm = np.full(L, NEG << 6, dtype=np.int64)
    for k, (h, df, do) in enumerate(srcs):
        pm = np.maximum(pm, pk(h[:-1] + s, CODE_DIAG - k))
        pm = np.maximum(pm, pk((h + df)[1:] + P.g, CODE_VERT - 3 * k))
        pm = np.maximum(pm, pk(h[1:] + P.g, CODE_VERT - 3 * k - 1))
        pm = np.maximum(pm, pk((h + do)[1:] + P.q, CODE_VERT - 3 * k - 2))
    if P.algorithm == 0:
        pm = np.maximum(pm, CODE_ZERO)
    p5, p6, p7, p8 = pk(E[:-1] + P.e, CODE_HORZ), pk(hl + P.g, CODE_HORZ - 1), pk(Q[:-1] + P.c, CODE_HORZ - 2), pk(hl + P.q, CODE_HORZ - 3)
    pf = np.maximum(pm, np.maximum(np.maximum(p5, p6), p7))
    assert np.array_equal(pf >> 6, h1)
    code = pf & 63
    hx = (p5 >= p6) | (p7 >= p8)
    # (b) equality tests in spoa's order (what a formulation without room for a code beside the value computes)
    vFE, vFO, vOE = h1 == fe + P.g, h1 == fo + P.g, h1 == oe + P.q
    key = np.minimum(np.minimum(np.where(vFE, kfe * 3 + 0, big), np.where(vFO, kfo * 3 + 1, big)), np.where(vOE, koe * 3 + 2, big))
    code_b = np.where(h1 == E[:-1] + P.e, CODE_HORZ, np.where(h1 == hl + P.g, CODE_HORZ - 1, CODE_HORZ - 2))
    code_b = np.where(key < big, CODE_VERT - key, code_b)
    code_b = np.where(h1 == d, CODE_DIAG - kd, code_b)
    if P.algorithm == 0:
        code_b = np.where(h1 == 0, CODE_ZERO, code_b)
    assert np.array_equal(code, code_b)
    assert np.array_equal(hx, (E[:-1] + P.e == E[1:]) | (Q[:-1] + P.c == Q[1:]))
    bits = code | np.where(hx, B_HX, 0)
    # upward run out of this cell: smallest (slot, class) in the order F-open, F-extend, O-open, O-extend
    keyx = np.minimum(np.minimum(np.where(fo >= fe, kfo * 4 + 0, big), np.where(fe >= fo, kfe * 4 + 1, big)),
                      np.minimum(np.where(fo >= oe, kfo * 4 + 2, big), np.where(oe >= fo, koe * 4 + 3, big)))
    bits |= np.where((keyx & 1) == 0, B_VSTOP, 0)
    slots = (keyx >> 2) if multi else None
    # what the following rows read from this one
    fs_new, os_new = P.e + np.maximum(fe, fo), P.c + np.maximum(oe, fo)
    dF = np.full(L + 1, -1, dtype=np.int64); dO = np.full(L + 1, -1, dtype=np.int64)
    dF[1:] = np.maximum(fs_new - h1, -1)
    dO[1:] = np.maximum(os_new - h1, -1)
    assert dF.max() <= P.e - P.g and dO.max() <= P.c - P.q
    # the same for the horizontal states of this row (what the lazy back-track reads instead of a code byte): Es = E + e - g,
    # Qs = Q + c - q relative to H, clamped at -1 (column 0: E = Q = -inf)
    dE = np.full(L + 1, -1, dtype=np.int64); dQ = np.full(L + 1, -1, dtype=np.int64)
    dE[1:] = np.maximum(E[1:] + P.e - P.g - h1, -1)
    dQ[1:] = np.maximum(Q[1:] + P.c - P.q - h1, -1)
    assert dE.max() <= P.e - P.g and dQ.max() <= P.c - P.q
    return H, dF, dO, bits, slots, dE, dQ


class Graph(object):
    MAXP, MAXA = 12, 15          # the kernel's limits: in-edges of a node, other members of an aligned set

    def __init__(self):
        self.code, self.pred, self.pw, self.aligned, self.cov, self.nout = [], [], [], [], [], []
        self.order, self.rank = [], []            # order[r-1] = node; rank[node] = r
        self.root = []                            # smallest node id among the node's descendants and their aligned sets (topo_sort)

    def new(self, code):
        self.code.append(int(code)); self.pred.append([]); self.pw.append([]); self.aligned.append([]); self.cov.append(0)
        self.nout.append(0); self.rank.append(0); self.root.append(len(self.code) - 1)
        return len(self.code) - 1

    def edge(self, u, v, w):
        if u in self.pred[v]:
            self.pw[v][self.pred[v].index(u)] += w
        else:
            if len(self.pred[v]) >= self.MAXP:
                raise OverflowError('in-degree')
            self.pred[v].append(u); self.pw[v].append(w); self.nout[u] += 1


def topo_sort(G, path):
    """Graph::TopologicalSort as the kernel computes it -- the same order as spoa's sequential depth-first search
    (oracle/poa_oracle.c: topo_sort), from independent pieces:

    1. root[x] = the smallest node id among everything that depends on x (its descendants over the edges, the members of
       their aligned sets, and so on).  spoa's outer loop visits the ids in ascending order and a visit emits exactly the
       unfinished ancestors of that id, so x is emitted during the visit of id root[x].  The values of the graph before
       this sequence are kept; the new sequence's path lowers them by a suffix minimum along the path, the rest is
       relaxation to the fixed point (a few sweeps).
    2. every id r with root[r] == r is the start of one search over the nodes with root == r, independent of all others
       (a dependency with a smaller root is finished by then, one with a larger root cannot occur); its nodes go to the
       positions behind the nodes of all smaller roots.  The kernel runs these searches one per lane.
    """
    N = len(G.code)
    root = G.root
    smin = N
    for v in reversed(path):                                 # along the new path: everything behind a base depends on it
        smin = min(smin, root[v])
        root[v] = smin
    changed = True
    while changed:
        changed = False
        for v in range(N):
            for u in G.pred[v]:
                if root[v] < root[u]:
                    root[u] = root[v]; changed = True
            for a in G.aligned[v]:
                if root[a] < root[v]:
                    root[v] = root[a]; changed = True
    size = [0] * N
    for v in range(N):
        size[root[v]] += 1
    base, acc = [0] * N, 0
    for r in range(N):
        base[r] = acc
        acc += size[r]
    order = [-1] * N
    done, ignored = [False] * N, [False] * N
    for r in range(N):
        if size[r] == 0:
            continue
        assert root[r] == r
        if size[r] == 1:
            order[base[r]] = r
            continue
        # one search: frames (node, cursor) grow from the end of the root's own stretch of `order`, emitted nodes from its start
        lo, hi = base[r], base[r] + size[r]
        sp, k = hi, lo
        sp -= 1; order[sp] = (r, 0)
        while sp < hi:
            v, cur = order[sp]
            if cur == 0 and not ignored[v]:
                for a in G.aligned[v]:
                    assert not done[a]
                    ignored[a] = True
            deps = ([] if ignored[v] else G.aligned[v][::-1]) + G.pred[v][::-1]      # the order spoa's stack hands them out
            nxt = -1
            while cur < len(deps):
                d = deps[cur]; cur += 1
                if root[d] == r and not done[d]:
                    nxt = d
                    break
                assert root[d] <= r
            if nxt >= 0:
                order[sp] = (v, cur)
                sp -= 1
                assert sp >= k
                order[sp] = (nxt, 0)
                continue
            done[v] = True
            sp += 1
            if not ignored[v]:
                order[k] = v; k += 1
                for a in G.aligned[v]:
                    order[k] = a; k += 1
                assert k <= sp or sp == hi
        assert k == hi
    G.order = order
    for i, v in enumerate(order):
        G.rank[v] = i + 1


def align(P, G, seq):
    """pn[j] = rank of the node base j is aligned to (0: none), end-cell score"""
    N, L = len(G.order), len(seq)
    rows = [row0(P, L)]
    planes = [None]
    h0 = rows[0][0]
    lazy = [(h0, np.full(L + 1, -1), np.full(L + 1, -1), np.full(L + 1, -1), np.full(L + 1, -1), [], -1)]   # row 0: no vertical state; never a current row
    sinks = []
    # column 0 of the global mode: H[i][0] = max(F, O)[i][0] with F[i][0] = e + max over sources (a source row: g)
    f0, o0 = [0], [0]
    best, bi, bj = (0 if P.algorithm == 0 else NEG), 0, 0
    for r in range(1, N + 1):
        v = G.order[r - 1]
        pr = [G.rank[u] for u in G.pred[v]] or [0]
        if G.pred[v]:
            f0.append(P.e + max(f0[p] for p in pr)); o0.append(P.c + max(o0[p] for p in pr))
        else:
            f0.append(P.g); o0.append(P.q)
        col0 = max(f0[r], o0[r]) if P.algorithm == 1 else 0
        H, dF, dO, bits, slots, dE, dQ = dp_row(P, G.code[v], seq, [rows[p] for p in pr], col0)
        rows.append((H, dF, dO)); planes.append((bits, slots, pr)); lazy.append((H, dF, dO, dE, dQ, pr, G.code[v]))
        sink = G.nout[v] == 0
        if P.algorithm == 0 or (sink and P.algorithm == 2):
            jm = int(np.argmax(H[1:])) + 1
            if H[jm] > best:
                best, bi, bj = int(H[jm]), r, jm
        elif sink and H[L] > best:
            best, bi, bj = int(H[L]), r, L
    if LAZY:
        pn, jfin, steps = backtrack_lazy(P, lazy, seq, bi, bj)
        return pn, best, (0 if P.algorithm == 1 else jfin), bj - 1, steps + (1 if P.algorithm == 1 and steps == 0 else 0)
    pn = [0] * L
    r, j = bi, bj
    steps = 0
    while r > 0 and j > 0:
        bits, slots, pr = planes[r]
        b = int(bits[j - 1])
        code = b & 63
        if code == CODE_ZERO:
            break
        steps += 1
        if code > CODE_VERT:
            pn[j - 1] = r
            r, j = pr[CODE_DIAG - code], j - 1
        elif code > CODE_HORZ:
            v = CODE_VERT - code
            r = pr[v // 3]
            if v % 3 != 1:
                while True:
                    bits, slots, pr = planes[r]
                    b2 = int(bits[j - 1])
                    r = pr[int(slots[j - 1]) if slots is not None else 0]
                    if (b2 & B_VSTOP) or r == 0:
                        break
        else:
            j -= 1
            if code != CODE_HORZ - 1:
                while True:
                    c = j
                    j -= 1
                    if not (int(bits[c - 1]) & B_HX):
                        break
    if P.algorithm == 1:
        # global mode: the walk goes on to (0, 0) along the borders -- vertical steps in column 0 hold no base, horizontal
        # steps in row 0 hold bases without a node: either way the whole sequence is inside the alignment
        steps += r + j
        j = 0
    return pn, best, j, bj - 1, steps


LAZY = False        # align(): replay the back-track from the code bytes (False) or from H and the clamped differences (True)


def backtrack_lazy(P, rows, seq, bi, bj):
    """spoa's back-track (oracle/poa_oracle.c: align_gotoh) from what the lazy forward pass keeps per cell: H and the four
    clamped differences dF, dO (vertical states, as the next rows read them) and dE, dQ (horizontal states); -1 = below H - 1
    in its frame, where the state can neither win nor tie.  rows[r] = (H, dF, dO, dE, dQ, source ranks, letter)."""
    g, e, q, c = P.g, P.e, P.q, P.c
    pn = [0] * len(seq)
    r, j, steps = bi, bj, 0
    while r > 0 and j > 0:
        H, dF, dO, dE, dQ, pr, code = rows[r]
        pr = pr or [0]
        h = int(H[j])
        if P.algorithm == 0 and h == 0:
            break
        steps += 1
        sc = P.m if code == seq[j - 1] else P.n
        hit = None
        for p in pr:
            if h == int(rows[p][0][j - 1]) + sc:
                hit = ('d', p)
                break
        if hit is None:
            for p in pr:
                hp, fs, os_ = int(rows[p][0][j]), int(rows[p][0][j] + rows[p][1][j]), int(rows[p][0][j] + rows[p][2][j])
                if h == fs + g:
                    hit = ('v', p, True)
                elif h == hp + g:
                    hit = ('v', p, False)
                elif h == os_ + q:
                    hit = ('v', p, True)
                elif h == hp + q:
                    hit = ('v', p, False)
                if hit:
                    break
        if hit is None:
            hl, es, qs = int(H[j - 1]), int(H[j - 1] + dE[j - 1]), int(H[j - 1] + dQ[j - 1])
            if h == es + g:
                hit = ('h', True)
            elif h == hl + g:
                hit = ('h', False)
            elif h == qs + q:
                hit = ('h', True)
            elif h == hl + q:
                hit = ('h', False)
        assert hit is not None
        if hit[0] == 'd':
            pn[j - 1] = r
            r, j = hit[1], j - 1
        elif hit[0] == 'v':
            r = hit[1]
            if hit[2]:
                while r > 0:
                    prr = rows[r][5] or [0]
                    mf = max(max(int(rows[p][0][j]), int(rows[p][0][j] + rows[p][1][j])) for p in prr)
                    mo = max(max(int(rows[p][0][j]), int(rows[p][0][j] + rows[p][2][j])) for p in prr)
                    stop, ni = False, 0
                    for p in prr:
                        hp, fs, os_ = int(rows[p][0][j]), int(rows[p][0][j] + rows[p][1][j]), int(rows[p][0][j] + rows[p][2][j])
                        if hp == mf:
                            stop, ni = True, p
                            break
                        if fs == mf:
                            stop, ni = False, p
                            break
                        if hp == mo:
                            stop, ni = True, p
                            break
                        if os_ == mo:
                            stop, ni = False, p
                            break
                    r = ni
                    steps += 1
                    if stop:
                        break
        else:
            j -= 1
            if hit[1]:
                while True:
                    j -= 1
                    steps += 1
                    if not (dE[j] >= 0 or dQ[j] >= 0):
                        break
    return pn, j, steps


def fuse(G, seq, pn, jb, je, steps):
    """Graph::AddAlignment: node ids go to the bases in front of the alignment, then to those behind it, then to the bases
    [jb, je] it holds, one by one.  pn[j] = rank of the node base j is aligned to (0: none)."""
    L = len(seq)
    if steps == 0:
        jb, je = L, L - 1
    elif jb > je:
        raise ValueError('alignment without a base (spoa throws)')
    used = [-1] * L
    for j in list(range(0, jb)) + list(range(je + 1, L)):
        used[j] = G.new(int(seq[j]))
    for j in range(jb, je + 1):
        b = int(seq[j])
        if pn[j] > 0:
            v = G.order[pn[j] - 1]
            use = v if G.code[v] == b else next((a for a in G.aligned[v] if G.code[a] == b), -1)
            if use < 0:
                use = G.new(b)
                if len(G.aligned[v]) >= G.MAXA:
                    raise OverflowError('aligned set')
                for a in G.aligned[v]:
                    G.aligned[a].append(use); G.aligned[use].append(a)
                G.aligned[v].append(use); G.aligned[use].append(v)
        else:
            use = G.new(b)
        used[j] = use
    for j in range(L):
        if L >= 2:
            G.cov[used[j]] += 1
        if j > 0:
            G.edge(used[j - 1], used[j], 2)
    topo_sort(G, used)
    return used


def consensus(G, min_cov=0):
    N = len(G.order)
    score, bp = [-1] * N, [-1] * N

    def relax(v, barred):
        for u, w in zip(G.pred[v], G.pw[v]):
            if barred and score[u] == -1:
                continue
            if score[v] < w or (score[v] == w and score[bp[v]] <= score[u]):
                score[v], bp[v] = w, u
        if bp[v] >= 0:
            score[v] += score[bp[v]]
    top = -1
    for v in G.order:
        relax(v, False)
        if top < 0 or score[top] < score[v]:
            top = v
    while G.nout[top]:
        start = top
        for h in range(N):
            if start in G.pred[h]:
                for u in G.pred[h]:
                    if u != start:
                        score[u] = -1
        top = -1
        for v in G.order[G.rank[start]:]:
            score[v], bp[v] = -1, -1
            relax(v, True)
            if top < 0 or score[top] < score[v]:
                top = v
    path = []
    v = top
    while v >= 0:
        path.append(v)
        v = bp[v]
    path.reverse()
    return [G.code[v] for v in path if G.cov[v] >= min_cov]


def poa(seqs, algorithm=0, genmsa=False, m=10, n=-4, g=-8, e=-2, q=-24, c=-1, min_coverage=0):
    """seqs: arrays of letters (any small integers; equality is all that matters).  -> (consensus letters, msa rows (letters,
    45 = '-'; none for an empty sequence), end-cell scores)"""
    P = Params(algorithm, m, n, g, e, q, c)
    G = Graph()
    paths, scores = [], []
    for s in seqs:
        s = np.asarray(s, dtype=np.int64)
        if len(s) == 0:
            scores.append(0)
            continue
        if not G.order:
            pn, sc, jb, je, steps = [0] * len(s), 0, 0, -1, 0
        else:
            pn, sc, jb, je, steps = align(P, G, s)
        scores.append(sc)
        paths.append(fuse(G, s, pn, jb, je, steps))
    cons = consensus(G, min_coverage)
    rows = []
    if genmsa:
        col, nc, i = {}, 0, 0
        while i < len(G.order):
            v = G.order[i]
            col[v] = nc
            for a in G.aligned[v]:
                col[a] = nc
                i += 1
            i += 1; nc += 1
        for p in paths:
            row = [45] * nc
            for v in p:
                row[col[v]] = G.code[v]
            rows.append(row)
    return cons, rows, scores, list(G.order)
