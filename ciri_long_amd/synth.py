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


# ---- a synthetic world for the file-to-file stage 2 (find_bsj.scan_ccs_reads): a genome, single-exon circRNAs on it, reads with
# a cyclic consensus, and a MAPPER DOUBLE that answers from the truth -----------------------------------------------------------
# The external mapper (minimap2 through mappy, find_bsj.py:340) cannot run here and is not part of the measured path; the double
# stands where it stands, answers `map(seq)` for every sequence the stage derives from a read (the raw read, the doubled consensus,
# its rotations) with hits that carry what the reference reads from a mappy hit (ctg, r_st, r_en, q_st, q_en, strand, cigar, mlen,
# blen, is_primary), and keeps its own time so that the bench can state it and take it out.
_B = np.frombuffer(b'ACGT', dtype=np.uint8)
_RC = bytes.maketrans(b'ACGT', b'TGCA')


class TruthHit(object):
    __slots__ = ('ctg', 'r_st', 'r_en', 'q_st', 'q_en', 'strand', 'cigar', 'mlen', 'blen', 'is_primary', 'NM', 'mapq')

    def __init__(self, ctg, r_st, r_en, q_st, q_en, strand, primary):
        self.ctg, self.r_st, self.r_en, self.q_st, self.q_en, self.strand = ctg, r_st, r_en, q_st, q_en, strand
        n = r_en - r_st
        self.cigar = [(n, 0)]
        self.mlen = self.blen = n
        self.is_primary = primary
        self.NM, self.mapq = 0, 60


class TruthMapper(object):
    """answers from the construction of the reads.  A template's first `soft` bases (in the read's orientation) are never part of a
    hit -- the soft clip a real mapper leaves next to a junction -- so that find_bsj's rotation loop ends with `soft` clipped bases
    and align_clip_segments has to place them by Smith-Waterman in the hit +- 200 kb window (soft = 0 for about half of the reads)."""

    def __init__(self, world, delay_us=0):
        import time
        self.w = world
        self.seconds = 0.0
        self.calls = 0
        self.delay = delay_us * 1e-6      # a busy wait per call that HOLDS the interpreter lock: the cost of a real mapper call, for the mapper-pool line of bench.py
        self._clock = time.perf_counter
        self._raw = {}          # raw read -> template (keyed by the string: a worker process of the mapper pool holds copies, not these objects)
        self._bucket = {}       # (length, #A, #C, #G) -> templates: a rotation keeps all four
        for t, o in enumerate(world['oriented']):
            self._bucket.setdefault((len(o), o.count('A'), o.count('C'), o.count('G')), []).append(t)

    def note_raw(self, raw, t, phase):
        self._raw[raw] = (t, phase)

    def _hit(self, t, x, y, q_st, primary):
        """oriented-template bases [x, y) at query offset q_st"""
        a, b, strand = self.w['loci'][t]
        if strand > 0:
            return TruthHit('chr1', a + x, a + y, q_st, q_st + (y - x), 1, primary)
        return TruthHit('chr1', b - y, b - x, q_st, q_st + (y - x), -1, primary)

    def map(self, seq):
        t0 = self._clock()
        self.calls += 1
        out = self._map(seq)
        if self.delay:
            end = t0 + self.delay
            while self._clock() < end:
                pass
        self.seconds += self._clock() - t0
        return out

    def _map(self, seq):
        if not seq:
            return None
        known = self._raw.get(seq) if len(seq) >= 160 else None      # (raw reads hold two copies and more of a template of >= 80 bases)
        if known is not None:          # the raw read: one full copy of the template somewhere inside it (only the filters read this hit)
            t, phase = known
            n = len(self.w['oriented'][t])
            s = self.w['soft'][t]
            q0 = (n - phase) % n
            if q0 + n > len(seq):
                return None
            return [self._hit(t, s, n, q0 + s, 1)]
        L = len(seq)
        doubled = L % 2 == 0 and seq[:L // 2] == seq[L // 2:]
        one = seq[:L // 2] if doubled else seq
        n = len(one)
        for t in self._bucket.get((n, one.count('A'), one.count('C'), one.count('G')), ()):
            o2 = self.w['oriented2'][t]
            p = o2.find(one[:24]) if n >= 24 else o2.find(one)
            if p < 0 or p >= n or o2[p:p + n] != one:
                continue
            s = self.w['soft'][t]
            if doubled:                 # rot_p rot_p: the whole middle copy
                return [self._hit(t, s, n, n - p + s, 1)]
            hits = []
            lo = max(p, s)
            if n - lo > 0:
                hits.append((n - lo, self._hit(t, lo, n, lo - p, 0)))
            if p - s > 0:
                hits.append((p - s, self._hit(t, s, p, n - p + s, 0)))
            if not hits:
                return None
            hits.sort(key=lambda h: -h[0])
            hits[0][1].is_primary = 1
            return [h for _, h in hits]
        return None


def circ_world(n, seed=SEEDS['C3'] + 2, genome_len=20_000_000, rank=0, mapper_delay_us=0):
    """n reads of single-exon circRNAs on one contig: {'genome': str, 'loci': [(start, end, strand)], 'oriented': [template in read
    orientation], 'soft': [clipped bases], 'ccs_seq': {read id: [segments, ccs, raw]}, 'mapper': TruthMapper}.  The consensus of a
    read is a rotation of its template (stage 2 starts from stage 1's files; the consensus kernels are measured elsewhere)."""
    rng = np.random.Generator(np.random.PCG64([seed, rank, 11]))
    codes = rng.integers(0, 4, genome_len).astype(np.int8)
    genome = _B[codes].tobytes().decode()
    loci, oriented, soft = [], [], []
    for _ in range(n):
        p = int(np.clip(rng.lognormal(np.log(350.0), 0.5), 80, 1500))
        a = int(rng.integers(250000, genome_len - 250000 - p))
        strand = 1 if rng.random() < 0.5 else -1
        exon = genome[a:a + p]
        oriented.append(exon if strand > 0 else exon.encode().translate(_RC)[::-1].decode())
        loci.append((a, a + p, strand))
        soft.append(int(rng.integers(20, max(21, min(120, p // 3)))) if rng.random() < 0.5 else 0)
    world = {'genome': genome, 'loci': loci, 'oriented': oriented, 'oriented2': [o + o for o in oriented], 'soft': soft}
    mapper = TruthMapper(world, mapper_delay_us)
    ccs_seq = {}
    for t, o in enumerate(oriented):
        n_t = len(o)
        phase = int(rng.integers(0, n_t))
        ccs = o[phase:] + o[:phase]
        copies = max(2, int(round(1000 / n_t)))
        rphase = int(rng.integers(0, n_t))
        raw = (o * (copies + 2))[rphase:rphase + copies * n_t + int(rng.integers(0, n_t))]
        mapper.note_raw(raw, t, rphase)
        seg = ';'.join('%d-%d' % (k * n_t, (k + 1) * n_t) for k in range(copies))
        ccs_seq['read%07d' % t] = [seg, ccs, raw]
    world['ccs_seq'] = ccs_seq
    world['mapper'] = mapper
    return world
