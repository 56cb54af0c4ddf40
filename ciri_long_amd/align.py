"""Per-read alignment-record helpers of the `call` path (counterpart of the hot-path part of CIRI_long/align.py).

Same names, arguments and return conventions as the reference so that ``find_bsj`` reads the same; the bodies are
written from the behaviour, not from the text.  Host-side integer/string work; the GTF/BED
indexers (align.py:226-316) are one-off set-up and stay with the reference.

    Hit, SubHit, Aligner            align.py:73-181      mapper adaptors
    Fasta                           align.py:184-223     in-memory genome (plain-text FASTA reader, no pysam)
    get_blocks ... merge_clip_exon  align.py:319-428     CIGAR -> reference blocks
    remove_long_insert, get_primary_alignment            align.py:431-471
    find_annotated_signal, find_denovo_signal, get_ss_altered_length, sort_ss      align.py:474-733
    find_host_gene, find_retained_introns, find_overlap_exons                       align.py:736-796
    convert_cigar_string, find_alignment_pos                                       align.py:799-820
"""
import re
from operator import itemgetter

import numpy as np

from . import env
from .utils import revcomp

_OP_CODES = 'MIDNSHP=X'
OPERATION = {c: i for i, c in enumerate(_OP_CODES)}
OPERATION.update({i: c for i, c in enumerate(_OP_CODES[:8])})
OPERATION[9] = 'X'      # sic: the reference maps 9 (not 8) back to 'X' (align.py:29)

# (donor, acceptor) -> weight; 0 = canonical U2, larger = rarer (align.py:32-45)
SPLICE_SIGNAL = {
    ('GT', 'AG'): 0,
    ('GC', 'AG'): 1,
    ('AT', 'AC'): 2,
    ('GT', 'AC'): 2,
    ('AT', 'AG'): 2,
}

_CIGAR_RE = re.compile(r'(\d+)([MIDNSHP=X])')
M, I, D, N, S, H = 0, 1, 2, 3, 4, 5


def convert_cigar_string(x):
    return [(int(n), OPERATION[op]) for n, op in _CIGAR_RE.findall(x)]


class Hit(object):
    """bwapy alignment -> the attribute set of a mappy hit (align.py:73-117)."""

    def __init__(self, aln):
        self.ctg = aln.rname
        self.strand = 1 if aln.orient == '+' else -1
        self.cigar_string = aln.cigar
        self.r_st = aln.pos
        r_en, q_st, q_en, blen = self.r_st, 0, 0, 0
        for n, op in self.cigar:
            if op == M:
                q_en += n; r_en += n; blen += n
            elif op == I:
                q_en += n
            elif op in (D, N):
                r_en += n; blen += n
            elif op in (S, H) and q_st == 0:
                q_st = q_en = n
        self.r_en, self.q_st, self.q_en, self.blen = r_en, q_st, q_en, blen
        self.mlen = q_en - q_st
        self.is_primary = 0

    @property
    def cigar(self):
        return convert_cigar_string(self.cigar_string)

    def __str__(self):
        return '\t'.join(str(x) for x in (self.q_st, self.q_en, self.ctg, self.r_st, self.r_en, self.mlen, self.blen,
                                          self.cigar_string))


class SubHit(object):
    """A slice of a hit between two long insertions (align.py:120-164)."""

    def __init__(self, hit, r_st, q_st, cigar):
        self.ctg = hit.ctg
        self.strand = hit.strand
        self.cigar = cigar
        self.r_st = r_st
        r_en, q_en = r_st, q_st
        mlen = blen = 0           # one walk over the operations (this constructor runs once per mapper hit the stage looks at)
        for n, op in cigar:
            if op == M:
                q_en += n; r_en += n; mlen += n; blen += n
            elif op == I:
                q_en += n; mlen += n; blen += n
            elif op == D:
                r_en += n; blen += n
            elif op == N:
                r_en += n
            elif op in (S, H) and q_st == 0:
                q_st += n; q_en += n
        self.r_en, self.q_st, self.q_en = r_en, q_st, q_en
        self.mlen = mlen
        self.blen = blen
        self.is_primary = 0

    @property
    def cigar_string(self):
        return ''.join('{}{}'.format(n, OPERATION[op]) for n, op in self.cigar)

    def __str__(self):
        return '\t'.join(str(x) for x in (self.q_st, self.q_en, self.ctg, self.r_st, self.r_en, self.mlen, self.blen,
                                          self.cigar_string))


class Aligner(object):
    """bwapy.BwaAligner behind the ``map()`` interface of mappy (align.py:167-181)."""

    def __init__(self, aligner):
        self.aligner = aligner

    def map(self, seq):
        alns = self.aligner.align_seq(seq)
        if not alns:
            return None
        hits = [Hit(a) for a in alns]
        hits[0].is_primary = 1
        return hits


class Fasta(object):
    """Whole genome in memory with ``seq(contig, start, end)`` and ``contig_len`` (align.py:209-223).
    Reads plain (multi-line) FASTA itself instead of going through pysam."""

    def __init__(self, infile):
        self.genome = {}
        name, parts = None, []
        with open(infile) as f:
            for line in f:
                if line.startswith('>'):
                    if name is not None:
                        self.genome[name] = ''.join(parts)
                    name, parts = line[1:].split()[0], []
                else:
                    parts.append(line.strip())
        if name is not None:
            self.genome[name] = ''.join(parts)
        self.contig_len = {k: len(v) for k, v in self.genome.items()}

    def seq(self, contig, start, end):
        g = self.genome.get(contig)
        return None if g is None else g[start:end]


_FOLD = bytes((c if c in b'ACGT' else (c - 32 if c in b'acgt' else ord('N'))) for c in range(256))


class IndexGenome(object):
    """The genome as a minimap2 index serves it -- `mappy.Aligner.seq`, which is env.GENOME in the reference's main pass
    (find_bsj.py:340-341: `env.initializer(aligner, contig_len, aligner, ...)`), as opposed to the FASTA text of the short-read
    pass (:455,462).  The index keeps four bits per base: upper case, anything but ACGT reads as N (soft-masked and IUPAC
    letters lose their identity); `seq()` answers None for an unknown contig, a start outside [0, length) or an empty range, and
    clips the end at the contig's (mappy_fetch_seq) -- that None is what find_denovo_signal's `us_seq is None` test is for
    (align.py:580).  Built from anything with the Fasta contract; `genome` holds the folded text."""
    index_slices = True

    def __init__(self, fasta):
        self.genome = {k: v.encode('latin-1').translate(_FOLD).decode('latin-1') for k, v in fasta.genome.items()}
        self.contig_len = {k: len(v) for k, v in self.genome.items()}

    def seq(self, contig, start=0, end=0x7fffffff):
        g = self.genome.get(contig)
        if g is None or start < 0 or start >= len(g) or start >= end:
            return None
        return g[start:end]


# ---------------------------------------------------------------------------------------------------------------
# CIGAR -> blocks on the reference
# ---------------------------------------------------------------------------------------------------------------
class DeviceGenome(object):
    """`env.GENOME` with a copy of the contigs resident on the GPU: `seq()` keeps serving the host-side steps (splice
    signals, junction sequences), while the Smith-Waterman windows of the BSJ step are read in place from HBM
    (find_bsj._run_clip_jobs).  `host` is any object with the Fasta contract
    (`seq(ctg, start, end)`, `contig_len`); `contigs` the sequences to upload ({name: str})."""

    def __init__(self, host, contigs, context=None):
        from . import hip
        self.host = host
        self.index_slices = bool(getattr(host, 'index_slices', False))     # K6 cuts its windows the way `host.seq` does
        self.contig_len = dict(host.contig_len) if hasattr(host, 'contig_len') else {k: len(v) for k, v in contigs.items()}
        self.device = hip.Genome(context or hip.default_context(), contigs)

    def seq(self, ctg, start, end):
        return self.host.seq(ctg, start, end)


def get_blocks(hit):
    """[start, end, length] of every N-separated stretch of the hit (align.py:319-343; length is end-start+1)."""
    blocks, st, en = [], hit.r_st, hit.r_st
    for n, op in hit.cigar:
        if op in (M, D):
            en += n
        elif op == N:
            blocks.append([st, en, en - st + 1])
            st = en = en + n
    if en > st:
        blocks.append([st, en, en - st + 1])
    return blocks


def get_exons(hit):
    """[r_start, r_end, q_start, q_end] per N-separated stretch (align.py:346-371)."""
    out = []
    r0 = r1 = hit.r_st
    q0 = q1 = hit.q_st
    for n, op in hit.cigar:
        if op == M:
            r1 += n; q1 += n
        elif op == I:
            q1 += n
        elif op == D:
            r1 += n
        elif op == N:
            out.append([r0, r1, q0, q1])
            r0 = r1 = r1 + n
            q0 = q1
    if r1 > r0:
        out.append([r0, r1, q0, q1])
    return out


def get_parital_blocks(hit, junc):
    blocks = []
    for r0, r1, q0, q1 in get_exons(hit):
        tag = '*-' if abs(q0 - junc) <= 10 else ('-*' if abs(q1 - junc) <= 10 else r1 - r0 + 1)
        blocks.append([r0, r1, tag])
    return blocks


def merge_blocks(blocks):
    """Union of overlapping/abutting [start, end, _] intervals (align.py:386-399)."""
    ordered = sorted(blocks, key=itemgetter(0, 1))
    merged = []
    cur_st, cur_en = ordered[0][0], ordered[0][1]
    for st, en, _ in ordered[1:]:
        if st <= cur_en:
            cur_en = max(cur_en, en)
            cur_st = min(cur_st, st)
        else:
            merged.append([cur_st, cur_en, cur_en - cur_st + 1])
            cur_st, cur_en = st, en
    merged.append([cur_st, cur_en, cur_en - cur_st + 1])
    return merged


def merge_exons(tail_exons, head_exons):
    if head_exons[0][0] < tail_exons[-1][1]:
        return merge_blocks(tail_exons + head_exons)
    head_exons[0] = [head_exons[0][0], head_exons[0][1], '*-']
    tail_exons[-1] = [tail_exons[-1][0], tail_exons[-1][1], '-*']
    return tail_exons + head_exons


def merge_clip_exon(exons, clip_info):
    """Attach the Smith-Waterman placement of the clipped bases to the block list (align.py:412-428)."""
    clip_st, clip_en = clip_info[0], clip_info[1]
    if not (clip_st and clip_en):
        return exons
    first_st, last_en = exons[0][0], exons[-1][1]
    clip_block = [clip_st, clip_en, clip_en - clip_st + 1]
    if clip_en < first_st:
        return [clip_block] + exons
    if last_en < clip_st:
        return exons + [clip_block]
    if clip_st < first_st < clip_en:
        exons[0] = [clip_st, exons[0][1], exons[0][1] - clip_st + 1]
    elif clip_st < last_en < clip_en:
        exons[-1] = [exons[-1][0], clip_en, clip_en - exons[-1][0] + 1]
    return exons


def remove_long_insert(hit):
    """Cut the hit at every insertion longer than 20 bases, keep the piece with most aligned query bases
    (align.py:431-460)."""
    r, q = hit.r_st, hit.q_st
    piece_r, piece_q, piece = r, q, []
    pieces = []
    for n, op in hit.cigar:
        if op == M:
            r += n; q += n
        elif op == I:
            q += n
            if n > 20:
                pieces.append(SubHit(hit, piece_r, piece_q, piece))
                piece, piece_r, piece_q = [], r, q
                continue
        elif op in (D, N):
            r += n
        elif op in (S, H):
            if q == hit.q_st:
                q += n
        piece.append((n, op))
    if piece:
        pieces.append(SubHit(hit, piece_r, piece_q, piece))
    best = sorted(pieces, key=lambda x: x.mlen, reverse=True)[0]    # stable: first of the longest
    best.is_primary = 1
    return best


def get_primary_alignment(hits):
    if not hits:
        return None
    for hit in hits:
        if hit.is_primary:
            return remove_long_insert(hit)
    return None


# ---------------------------------------------------------------------------------------------------------------
# splice signals around a candidate back-splice junction
# ---------------------------------------------------------------------------------------------------------------
def get_ss_altered_length(i, j, us_free, ds_free, clip_base):
    clip_altered = min(abs(j - i - clip_base), abs(j - i + clip_base))
    us_altered = min(abs(i + us_free), abs(i - ds_free))
    ds_altered = min(abs(j + us_free), abs(j - ds_free))
    return abs(i - j), clip_altered, us_altered + ds_altered


def _free_sliding(contig, start, end):
    """How far the junction can slide without changing the sequence: largest i < 100 with identical flanks on the
    downstream side (ds_free) and j < 100 on the upstream side (us_free) (align.py:477-493)."""
    ds_free = 0
    for i in range(100):
        if end + i > env.CONTIG_LEN[contig]:
            break
        if env.GENOME.seq(contig, start, start + i) != env.GENOME.seq(contig, end, end + i):
            break
        ds_free = i
    us_free = 0
    for j in range(100):
        if start - j < 0:
            break
        if env.GENOME.seq(contig, start - j, start) != env.GENOME.seq(contig, end - j, end):
            break
        us_free = j
    return us_free, ds_free


def _annotated_shifts(index, strand, pos0, search_length):
    """Shifts in [-search_length, search_length) at which an annotated exon 'start' (at pos+1) or 'end' (at pos) sits;
    `kinds` semantics of align.py:507-546: starts first, then ends, both in ascending shift order."""
    starts, ends = [], []
    for shift in range(-search_length, search_length):
        at = index.get(pos0 + shift + 1)
        if at is not None and strand in at and 'start' in at[strand]:
            starts.append(shift)
    for shift in range(-search_length, search_length):
        at = index.get(pos0 + shift)
        if at is not None and strand in at and 'end' in at[strand]:
            ends.append(shift)
    return starts + ends


def find_annotated_signal(contig, start, end, clip_base, search_length=10, shift_threshold=3):
    """Annotated splice sites near both ends of the candidate (align.py:474-568).
    Returns (best site | None, us_free, ds_free, {strand: (us_shifts, ds_shifts)})."""
    signal = {}
    us_free, ds_free = _free_sliding(contig, start, end)
    if start - search_length - us_free - 2 < 0 or end + search_length + ds_free + 2 > env.CONTIG_LEN[contig]:
        return None, us_free, ds_free, signal
    if env.SS_INDEX is None or contig not in env.SS_INDEX:
        return None, us_free, ds_free, signal

    index = env.SS_INDEX[contig]
    found = []
    for strand in ('+', '-'):
        us_sites = _annotated_shifts(index, strand, start, search_length)
        ds_sites = _annotated_shifts(index, strand, end, search_length)
        signal[strand] = (us_sites, ds_sites)
        for i in us_sites:
            for j in ds_sites:
                if abs(i - j) > shift_threshold + clip_base:
                    continue
                us_ss = env.GENOME.seq(contig, start + i - 2, start + i)
                ds_ss = env.GENOME.seq(contig, end + j, end + j + 2)
                if strand == '-':
                    us_ss, ds_ss = revcomp(ds_ss), revcomp(us_ss)
                weight = SPLICE_SIGNAL.get((ds_ss, us_ss), 3)
                found.append(('{}-{}|{}-{}'.format(us_ss, ds_ss, i, j), strand, i, j, weight) +
                             get_ss_altered_length(i, j, us_free, ds_free, clip_base))
    if found:
        return sort_ss(found, us_free, ds_free, clip_base), us_free, ds_free, signal
    return None, us_free, ds_free, signal


def _all_occurrences(text, motif):
    """Positions p >= 1 with text[p:p+2] == motif (the reference's find loop starts at index 1, align.py:604-611)."""
    out, p = [], text.find(motif, 1)
    while p != -1:
        out.append(p)
        p = text.find(motif, p + 1)
    return out


def _denovo_candidates(strands, us_seq, ds_seq, us_search_length, tmp_signal, us_free, ds_free, clip_base,
                       shift_threshold, is_canonical):
    found = []
    for strand in strands:
        for (donor, acceptor), weight in SPLICE_SIGNAL.items():
            if is_canonical and weight != 0:
                continue
            # on the minus strand the motifs are read reverse-complemented and swap sides
            ds_motif, us_motif = (revcomp(acceptor), revcomp(donor)) if strand == '-' else (donor, acceptor)
            us_sites = [p - us_search_length for p in _all_occurrences(us_seq, us_motif)]
            ds_sites = [p - us_search_length for p in _all_occurrences(ds_seq, ds_motif)]
            if strand in tmp_signal:
                anno_us, anno_ds = tmp_signal[strand]
                us_sites = sorted(set(us_sites + anno_us))
                ds_sites = sorted(set(ds_sites + anno_ds))
            for i in us_sites:
                for j in ds_sites:
                    if abs(i - j) > clip_base + shift_threshold:
                        continue
                    found.append(('{}-{}*|{}-{}'.format(acceptor, donor, i, j), strand, i, j, weight) +
                                 get_ss_altered_length(i, j, us_free, ds_free, clip_base))
    return found


def find_denovo_signal(contig, start, end, host_strand, tmp_signal, us_free, ds_free, clip_base, search_length=10,
                       shift_threshold=3, is_canonical=False):
    """GT-AG style motifs around the junction, host-gene strand first, the other strand only if that finds nothing
    (align.py:571-695)."""
    us_len = search_length + us_free
    ds_len = search_length + ds_free
    us_seq = env.GENOME.seq(contig, start - us_len - 2, start + ds_len)
    ds_seq = env.GENOME.seq(contig, end - us_len, end + ds_len + 2)
    need = ds_len - us_len + 2
    if us_seq is None or len(us_seq) < need or ds_seq is None or len(ds_seq) < need:
        return None

    if host_strand:
        sites = _denovo_candidates(sorted(set(host_strand)), us_seq, ds_seq, us_len, tmp_signal, us_free, ds_free, clip_base,
                                   shift_threshold, is_canonical)
        if sites:
            return sort_ss(sites, us_free, ds_free, clip_base)
    others = sorted({'+', '-'} - set(host_strand)) if host_strand else ['+', '-']
    sites = _denovo_candidates(others, us_seq, ds_seq, us_len, tmp_signal, us_free, ds_free, clip_base,
                               shift_threshold, is_canonical)
    if sites:
        return sort_ss(sites, us_free, ds_free, clip_base)
    return None


_MOTIFS = list(SPLICE_SIGNAL)      # K6 reports the motif as an index into this order


def find_signal_rows(cands, hosts, is_canonical=True):
    """First half of find_signal_batch -- everything that needs the device, the genome or the annotation index:
    cands = [(contig, start, end, clip_base)], hosts = [find_host_gene(contig, start, end)] -> (rows, extra).
    rows: int32 array [n, 8] of the kernel's answers (status, us_free, ds_free, found, strand, us_shift, ds_shift, motif), or None
    when nothing went to the GPU; extra = {k: ('host', (site, us_free, ds_free))} for the candidates answered by the Python
    functions (no resident genome, or a case the kernel cannot express) and {k: ('ann', us_ss, ds_ss)} for the kernel's annotated
    pairs, whose id shows the genome's own dinucleotides (align.py:549-556).  `signal_from_row` is the second half: pure string
    formatting, which a worker process of the mapper pool can do without genome or index."""
    n = len(cands)
    extra = {}
    rows = None
    dev = getattr(env.GENOME, 'device', None)
    if not hasattr(dev, 'splice_signals'):
        dev = None
    # a host gene on a strand other than '+'/'-' (a '.' in the GTF) is searched under that label by the reference: not a
    # case of the kernel's two-bit strand mask
    on_gpu = [k for k in range(n)
              if dev is not None and cands[k][0] in dev.offset and not (hosts[k] and any(st not in ('+', '-') for st in hosts[k]))]
    done = [False] * n
    if on_gpu:
        if dev._sites_of is not env.SS_INDEX:
            dev.set_splice_sites(env.SS_INDEX)
        got = dev.splice_signals([(cands[k][0], cands[k][1], cands[k][2], cands[k][3],
                                   (1 if hosts[k] and '+' in hosts[k] else 0) | (2 if hosts[k] and '-' in hosts[k] else 0))
                                  for k in on_gpu], 10, 3, is_canonical, index_slices=getattr(env.GENOME, 'index_slices', False))
        got = np.asarray(got, dtype=np.int32).reshape(len(on_gpu), 8)
        if len(on_gpu) == n:
            rows = got
        else:
            rows = np.full((n, 8), -1, dtype=np.int32)
            rows[on_gpu] = got
        ok = got[:, 0] == 0
        for k, good in zip(on_gpu, ok.tolist()):
            done[k] = good
        for t in np.nonzero(ok & (got[:, 3] == 2))[0].tolist():
            k = on_gpu[t]
            ctg, start, end = cands[k][:3]
            i, j = int(got[t, 5]), int(got[t, 6])
            extra[k] = ('ann', env.GENOME.seq(ctg, start + i - 2, start + i), env.GENOME.seq(ctg, end + j, end + j + 2))
    for k in range(n):
        if done[k]:
            continue
        ctg, start, end, clip_base = cands[k][:4]
        site, us_free, ds_free, tmp_signal = find_annotated_signal(ctg, start, end, clip_base, clip_base + 10)
        if site is None:
            site = find_denovo_signal(ctg, start, end, hosts[k], tmp_signal, us_free, ds_free, clip_base, clip_base + 10, 3, is_canonical)
        extra[k] = ('host', (site, us_free, ds_free))
    return rows, extra


def signal_from_row(row, extra):
    """Second half of find_signal_batch for one candidate: (ss_site | None, us_free, ds_free) from its row of find_signal_rows
    (a sequence of eight ints) and its entry of `extra` (or None)."""
    if extra is not None and extra[0] == 'host':
        return extra[1]
    status, us_free, ds_free, found, strand, i, j, motif = row
    site = None
    if found == 1:
        donor, acceptor = _MOTIFS[motif]
        site = ('{}-{}*|{}-{}'.format(acceptor, donor, i, j), '-' if strand else '+', i, j)
    elif found == 2:          # a pair of annotated sites
        us_ss, ds_ss = extra[1], extra[2]
        if strand:
            us_ss, ds_ss = revcomp(ds_ss), revcomp(us_ss)
        site = ('{}-{}|{}-{}'.format(us_ss, ds_ss, i, j), '-' if strand else '+', i, j)
    return (site, us_free, ds_free)


def find_signal_batch(cands, is_canonical=True):
    """The splice-signal step of find_bsj.py:286-301 for many candidates at once:
    cands = [(contig, start, end, clip_base, host_strand)] -> [(ss_site | None, us_free, ds_free)], each entry what
    find_annotated_signal followed (when it finds nothing) by find_denovo_signal(..., clip_base + 10, 3, is_canonical)
    gives.  When env.GENOME is resident on the GPU the candidates are scanned there (K6, splice_scan.hip; env.SS_INDEX
    is uploaded once as sorted position runs) -- contig ends and IUPAC / soft-masked flanks included; only candidates
    the kernel cannot express (a host gene on a strand other than '+'/'-', a contig that is not resident, invalid
    coordinates) go through the functions above."""
    rows, extra = find_signal_rows([c[:4] for c in cands], [c[4] for c in cands], is_canonical)
    rl = rows.tolist() if rows is not None else None
    return [signal_from_row(rl[k] if rl is not None else None, extra.get(k)) for k in range(len(cands))]


def sort_ss(sites, us, ds, clip_base):
    """Rank candidate sites (id, strand, us_shift, ds_shift, weight, altered_len, clip_altered, altered_total) in four
    tiers (align.py:705-733) and return (id, strand, us_shift, ds_shift) of the winner.
    The reference sorts a ``set``; among sites with equal keys its choice follows string-hash order.  Here duplicates
    are dropped in first-seen order and the sort is stable, so the result is deterministic."""
    uniq = list(dict.fromkeys(sites))
    tiers = (
        (lambda s: -clip_base <= s[2] - s[3] <= clip_base, (6, 5, 4, 7)),                 # clipped
        (lambda s: -us <= s[2] <= ds and -us <= s[3] <= ds, (5, 4, 6, 7)),                # confident
        (lambda s: -clip_base <= s[2] <= 0 <= s[3] <= clip_base, (4, 5, 6, 7)),           # ambiguous
        (lambda s: True, (4, 5, 6, 7)),                                                  # anything left
    )
    rest = uniq
    for accept, key in tiers:
        chosen = [s for s in rest if accept(s)]
        if chosen:
            return itemgetter(0, 1, 2, 3)(sorted(chosen, key=itemgetter(*key))[0])
        rest = [s for s in rest if not accept(s)]
    return None


# ---------------------------------------------------------------------------------------------------------------
# annotation look-ups (500-bp bins)
# ---------------------------------------------------------------------------------------------------------------
def _bins(index, ctg, start, end):
    if index is None or ctg not in index:
        return None
    per_ctg = index[ctg]
    return [per_ctg[b] for b in range(start // 500, end // 500 + 1) if b in per_ctg]


def find_host_gene(ctg, start, end):
    bins = _bins(env.GTF_INDEX, ctg, start, end)
    if bins is None:
        return None
    host = {}
    for elements in bins:
        for el in elements:
            if el.end < start or el.start > end:
                continue
            if el.start - 500 <= start <= el.end + 500 or el.start - 500 <= end <= el.end + 500:
                host.setdefault(el.strand, []).append(el)
    return host or None


def find_retained_introns(ctg, start, end):
    bins = _bins(env.INTRON_INDEX, ctg, start, end)
    if bins is None:
        return None
    host = {}
    for introns in bins:
        for st, en, strand in introns:
            if st - 25 <= start and end <= en + 25:
                host.setdefault(strand, []).append((st, en, strand))
    return host or None


def find_overlap_exons(ctg, start, end):
    bins = _bins(env.GTF_INDEX, ctg, start, end)
    if bins is None:
        return None
    host = {}
    for elements in bins:
        for el in elements:
            if el.type != 'exon' or el.end - 25 < start or end < el.start + 25:
                continue
            host.setdefault(el.strand, []).append((el.start, el.end, el.strand))
    return host or None


def find_alignment_pos(alignment, pos):
    """Query coordinate aligned to reference position `pos` (collapse.py:373-387 -> align.py:803-820)."""
    r0 = r1 = alignment.ref_begin
    q0 = q1 = alignment.query_begin
    for n, op in convert_cigar_string(alignment.cigar_string):
        if op == M:
            r1 += n; q1 += n
        elif op == I:
            q1 += n
        elif op == D:
            r1 += n
        if r0 <= pos <= r1:
            return q0 + pos - r0
        r0, q0 = r1, q1
    return None
