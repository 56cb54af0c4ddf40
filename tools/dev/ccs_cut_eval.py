"""How near the copy boundaries of the consensus step (oracle/ccs_oracle.c, step 2) land to the TRUE copy starts of synthetic
rolling-circle reads -- the generator knows where every read base came from.  Compares the specification as the oracle library
implements it with a Python statement of a candidate rule (RULES below), and reports what the consensus gains.

    python tools/dev/ccs_cut_eval.py [n_reads] [seed]

Truth: the read starts at template offset `phase`; the true position of cut n is the first read base that was copied from raw
position >= n * p (the base homologous to read position 0, n periods later).  A chained rule is judged twice: by the distance of
cut n from that position, and by the error of each single step (cut n - cut n-1 against true n - true n-1)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'tests'))
import oracle_lib  # noqa: E402
from ciri_long_amd import synth  # noqa: E402

K = 8


def mutate_with_origin(codes, rng, sub=synth.SUB, ins=synth.INS, dele=synth.DEL):
    """synth.mutate, also returning for every output base the raw index it came from (an inserted base: the base before it)"""
    u = rng.random(len(codes))
    keep = u >= dele
    idx = np.nonzero(keep)[0]
    c = codes[keep].copy()
    us = u[keep]
    subm = us < dele + sub
    c[subm] = rng.integers(0, 4, int(subm.sum()), dtype=np.int8)
    insm = (~subm) & (rng.random(len(c)) < ins)
    rep = 1 + insm.astype(np.int64)
    out = np.repeat(c, rep)
    org = np.repeat(idx, rep)
    pos = np.cumsum(rep) - 1
    ipos = pos[insm]
    out[ipos] = rng.integers(0, 4, len(ipos), dtype=np.int8)
    return out, org


def kmers(seq):
    L = len(seq)
    h = np.full(L, -1, dtype=np.int64)
    if L >= K:
        v = np.zeros(L - K + 1, dtype=np.int64)
        ok = np.ones(L - K + 1, dtype=bool)
        for t in range(K):
            b = seq[t:L - K + 1 + t].astype(np.int64)
            ok &= (b >= 0) & (b <= 3)
            v = (v << 2) | (b & 3)
        h[:L - K + 1] = np.where(ok, v, -1)
    return h


def cuts_v2(seq, p0, order='centre', confirm=0, reach=None, vote_r=8, vote_fwd=False):
    """the anchor rule, oracle/ccs_oracle.c step 2 (clh-ccs v2) with the defaults; the other arguments are the variants that were weighed:
    order='start' (distance from the k-mer's first base, not its centre), confirm (the same offset at a neighbouring position), vote_r=None
    (no guard by the vote), vote_fwd (vote over [b, b+W) as v1 did)"""
    L = len(seq)
    h = kmers(seq)
    tol = max(4, p0 // 8)
    W = min(p0, 96) if reach is None else reach
    b, prev, cuts = 0, p0, []
    while len(cuts) < 64:
        lo, hi = p0 - tol, min(p0 + tol, L - b)
        if hi < lo or hi < 1:
            break
        best = None
        vote = None
        if vote_r is not None:
            hist = np.zeros(hi - lo + 1, dtype=np.int64)
            for i in range(b if vote_fwd else max(0, b - W), min(L, b + W)):
                if h[i] < 0:
                    continue
                a, z = i + lo, min(i + hi, L - 1)
                if a > z:
                    continue
                m = np.nonzero(h[a:z + 1] == h[i])[0]
                hist[m] += 1
            if hist.max() > 0:
                vote = min(range(lo, hi + 1), key=lambda x: (-hist[x - lo], abs(x - prev), x))
        for i in range(max(0, b - W), min(L, b + W)):
            if h[i] < 0:
                continue
            a, z = i + lo, min(i + hi, L - 1)
            if a > z:
                continue
            m = np.nonzero(h[a:z + 1] == h[i])[0]
            if not len(m):
                continue
            deltas = m + lo
            if vote is not None:
                deltas = deltas[np.abs(deltas - vote) <= vote_r]
                if not len(deltas):
                    continue
            d = min(deltas, key=lambda x: (abs(x - prev), x))
            if confirm:
                # the same delta at a neighbouring position
                okc = False
                for ii in (i - 1, i + 1):
                    if 0 <= ii < L and ii + d < L and h[ii] >= 0 and h[ii] == h[ii + d]:
                        okc = True
                if not okc:
                    continue
            if order == 'centre':
                key = (abs(i + K // 2 - b), -i)
            else:
                key = (i - b if i >= b else b - i, -i)
            if best is None or key < best[0]:
                best = (key, int(d))
        if best is None:
            cands = [d for d in range(max(lo, 1), hi + 1)]
            d = min(cands, key=lambda x: (abs(x - prev), x))
        else:
            d = best[1]
        b += d
        prev = d
        cuts.append(b)
    return cuts


def cuts_v1(seq, p0):
    """clh-ccs v1 (rounds 1-5): the offset with the most anchors over [b, b+W)"""
    L = len(seq)
    h = kmers(seq)
    tol = max(4, p0 // 8)
    W = min(p0, 96)
    b, prev, cuts = 0, p0, []
    while len(cuts) < 64:
        best = None
        for d in range(p0 - tol, p0 + tol + 1):
            if d < 1 or b + d > L:
                continue
            sc = sum(1 for i in range(b, min(b + W, L - d)) if h[i] >= 0 and h[i] == h[i + d])
            key = (-sc, abs(d - prev), d)
            if best is None or key < best[0]:
                best = (key, d)
        if best is None:
            break
        b += best[1]; prev = best[1]
        cuts.append(b)
    return cuts


def oracle_segments(seq):
    cuts = np.zeros(64, dtype=np.int32)
    import ctypes as C
    nc = C.c_int32(0); k = C.c_int32(0); sup = C.c_int32(0)
    s = np.ascontiguousarray(seq, dtype=np.int8)
    p0 = oracle_lib._ccs_lib().clo_ccs_segments(s.ctypes.data, len(s), cuts.ctypes.data, C.byref(nc), C.byref(k), C.byref(sup))
    return p0, [int(x) for x in cuts[:nc.value]]


def consensus_of(seq, cuts):
    L = len(seq)
    segs, b = [], 0
    for c in cuts:
        segs.append(seq[b:c]); b = c
    if L - b >= 20 and len(cuts) < 64:
        segs.append(seq[b:])
    n = len(segs)
    return oracle_lib.oracle_poa(segs, 0, False, 10, -4, -8, -2, -24, -1, min_coverage=(n + 1) // 2)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rng = np.random.Generator(np.random.PCG64(seed))
    rules = {'oracle (v2 in C)': None,
             'v1 (vote alone)': 'v1',
             'v2 (python)': dict(),
             'v2 no guard': dict(vote_r=None),
             'v2 no guard confirm': dict(vote_r=None, confirm=1),
             'v2 no guard, start': dict(vote_r=None, order='start'),
             'v2 guard 3': dict(vote_r=3),
             'v2 guard 5': dict(vote_r=5),
             'v2 guard 5, v1 vote': dict(vote_r=5, vote_fwd=True),
             }
    stat = {r: dict(abs=[], step=[], ident=[], n=0) for r in rules}
    done = 0
    while done < n:
        tm = synth.template(rng)
        p = len(tm)
        Lt = int(max(300, round(rng.normal(1000, 100))))
        phase = int(rng.integers(0, p))
        raw = np.tile(tm, Lt // p + 2)[phase:phase + Lt]
        read, org = mutate_with_origin(raw, rng)
        p0, v1 = oracle_segments(read)
        if not p0 or abs(p0 - p) > max(4, p // 8):
            continue
        done += 1
        true = []
        k = 1
        while True:
            j = int(np.searchsorted(org, k * p, side='left'))
            if j >= len(read):
                break
            true.append(j); k += 1
        tm2 = oracle_lib.decode(np.tile(tm, 2))
        for r, kw in rules.items():
            cuts = v1 if kw is None else (cuts_v1(read, p0) if kw == 'v1' else cuts_v2(read, p0, **kw))
            st = stat[r]
            for q, c in enumerate(cuts):
                if q < len(true):
                    st['abs'].append(abs(c - true[q]))
                    st['step'].append(abs((c - (cuts[q - 1] if q else 0)) - (true[q] - (true[q - 1] if q else 0))))
            if len(cuts) >= 2:
                cons = consensus_of(read, cuts)
                if cons:
                    a = oracle_lib.oracle_align(tm2, cons, 1, 1, 1, 1)
                    st['ident'].append(a['score'] / max(len(cons), p))
                    st['n'] += 1
    for r, st in stat.items():
        a = np.array(st['abs']); s = np.array(st['step']); i = np.array(st['ident'])
        print('%-22s cuts %5d  |cut-true| mean %.2f median %.0f p90 %.0f p99 %.0f max %.0f  exact %.1f%%  within2 %.1f%%   step err mean %.2f exact %.1f%%   consensus identity mean %.4f median %.4f (n=%d)'
              % (r, len(a), a.mean(), np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max(), 100 * (a == 0).mean(), 100 * (a <= 2).mean(), s.mean(), 100 * (s == 0).mean(), i.mean(), np.median(i), st['n']))


if __name__ == '__main__':
    main()
