"""Seeded synthetic "NanoSim-shaped" reads (recipe mirrors misc/NanoSim.ipynb of the reference:
pseudo-circular templates read by rolling circle, ONT-like errors, as many linear negatives as circular reads).

NanoSim itself is not available offline; this is the data every bench/test of this repository uses.  Everything is
int8 codes (A=0 C=1 G=2 T=3), numpy-vectorised, and a pure function of (seed, rank).
"""
import numpy as np

SEEDS = {'C2': 20210842, 'C3': 20210843, 'C4': 20210844, 'C5': 20210845}
SUB, INS, DEL = 0.04, 0.04, 0.05


def mutate(codes, rng, sub=SUB, ins=INS, dele=DEL):
    """per base: deletion, else substitution (uniform base), else keep and maybe insert one uniform base after it"""
    u = rng.random(len(codes))
    keep = u >= dele
    c = codes[keep].copy()
    us = u[keep]
    subm = us < dele + sub
    c[subm] = rng.integers(0, 4, int(subm.sum()), dtype=np.int8)
    insm = (~subm) & (rng.random(len(c)) < ins)
    rep = 1 + insm.astype(np.int64)
    out = np.repeat(c, rep)
    pos = np.cumsum(rep) - 1
    ipos = pos[insm]
    out[ipos] = rng.integers(0, 4, len(ipos), dtype=np.int8)
    return out


def template(rng):
    p = int(np.clip(rng.lognormal(np.log(350.0), 0.5), 80, 1500))
    return rng.integers(0, 4, p, dtype=np.int8)


def rolling_circle_read(rng, tmpl, target_len):
    p = len(tmpl)
    phase = int(rng.integers(0, p))
    reps = target_len // p + 2
    raw = np.tile(tmpl, reps)[phase:phase + target_len]
    return mutate(raw, rng)


def c2_batch(n, seed=SEEDS['C2'], rank=0, window=2000, mean_len=1000, sd_len=100, min_len=300, max_len=4096):
    """BASELINE config 2: n reads of ~1 kb, each against its own 2 kb window that embeds the read's linearised
    template at a random offset; half of the reads are linear negatives.  Returns (reads, windows) as lists of int8."""
    rng = np.random.Generator(np.random.PCG64([seed, rank]))
    reads, wins = [], []
    for _ in range(n):
        L = int(max(min_len, min(max_len, round(rng.normal(mean_len, sd_len)))))
        tm = template(rng)
        if rng.random() < 0.5:
            read = rolling_circle_read(rng, tm, L)
        else:
            read = mutate(rng.integers(0, 4, L, dtype=np.int8), rng)
        read = read[:max_len]
        w = rng.integers(0, 4, window, dtype=np.int8)
        emb = tm[:window]
        off = int(rng.integers(0, window - len(emb) + 1))
        w[off:off + len(emb)] = emb
        reads.append(read)
        wins.append(w)
    return reads, wins


def c4_batch(n, seed=SEEDS['C4'], rank=0, window=2000):
    """BASELINE config 4: read lengths a uniform mix of {500, 1000, 2000, 4000} +- 10 %, otherwise as c2_batch."""
    rng = np.random.Generator(np.random.PCG64([seed, rank]))
    reads, wins = [], []
    for _ in range(n):
        base = (500, 1000, 2000, 4000)[int(rng.integers(0, 4))]
        L = int(round(base * (0.9 + 0.2 * rng.random())))
        tm = template(rng)
        if rng.random() < 0.5:
            read = rolling_circle_read(rng, tm, L)
        else:
            read = mutate(rng.integers(0, 4, L, dtype=np.int8), rng)
        read = read[:4096]
        w = rng.integers(0, 4, window, dtype=np.int8)
        emb = tm[:window]
        off = int(rng.integers(0, window - len(emb) + 1))
        w[off:off + len(emb)] = emb
        reads.append(read)
        wins.append(w)
    return reads, wins


def clip_batch(n, seed=SEEDS['C3'], rank=0, window=400000, shared_window=True):
    """Production shape of align_clip_segments (find_bsj.py:191-205): 20..300-nt clips against a +-200 kb window."""
    rng = np.random.Generator(np.random.PCG64([seed, rank, 7]))
    base = rng.integers(0, 4, window, dtype=np.int8)
    reads, wins = [], []
    for _ in range(n):
        L = int(rng.integers(20, 301))
        w = base if shared_window else rng.integers(0, 4, window, dtype=np.int8)
        st = int(rng.integers(0, window - L))
        reads.append(mutate(w[st:st + L], rng))
        wins.append(w)
    return reads, wins
