"""Back-splice-junction calling for cyclic consensus reads (counterpart of CIRI_long/find_bsj.py:139-490).

Same functions and signatures as the reference (``find_bsj``, ``align_clip_segments``, ``scan_ccs_chunk``,
``scan_ccs_reads``, ``recover_ccs_chunk``, ``recover_ccs_reads``).  What changes is the shape of the work: the reference
aligns clipped bases read by read inside the chunk loop (one ctypes round trip each, find_bsj.py:203-216); here a chunk is
processed in three phases -- (1) mapper-dependent filtering per read, which also collects the clip-vs-window pairs that
need Smith-Waterman, (2) ONE batched call into the HIP kernels for the whole chunk, (3) coordinate arithmetic, splice
signal search and record assembly per read -- with identical outputs in identical order.

The mapper stays external and pluggable: ``env.ALIGNER.map(seq)`` (mappy, bwapy adaptor, or a test double).
"""
from bisect import bisect_right
from collections import defaultdict

from . import env
from .align import (find_signal_batch, find_signal_rows, find_host_gene, get_blocks, get_parital_blocks,
                    get_primary_alignment, merge_clip_exon, merge_exons, remove_long_insert, signal_from_row)
from .utils import grouper, revcomp

CLIP_MIN = 20            # find_bsj.py:191
CLIP_MAX_FRACTION = 0.6  # find_bsj.py:193
WINDOW_FLANK = 200000    # find_bsj.py:196-197
WINDOW_MAX_N = 0.3       # find_bsj.py:200
HOST_WINDOW_BYTES = 256 << 20   # host-window route: window bytes packed per GPU call


def find_bsj(ccs):
    """Rotate the consensus until the mapper's primary hit stops improving (find_bsj.py:139-179).
    Returns (rotated sequence, junction offset on ccs) or (None, None)."""
    n = len(ccs)
    first = get_primary_alignment(env.ALIGNER.map(ccs * 2))
    if first is None:
        return None, None

    junc = first.q_st % n
    best_junc, best_mlen = 0, 0
    visited = set()
    while True:
        circ = ccs[junc:] + ccs[:junc]
        hit = get_primary_alignment(env.ALIGNER.map(circ))
        if hit is None or hit.mlen <= best_mlen:
            junc = best_junc
            break
        best_mlen, best_junc = hit.mlen, junc
        head_clip, tail_clip = hit.q_st, n - hit.q_en
        if head_clip == 0 and tail_clip == 0:
            break
        # move the junction past the larger clip
        junc = (junc + (head_clip if head_clip >= tail_clip else hit.q_en)) % n
        if junc in visited:
            junc = best_junc
            break
        visited.add(junc)
    return ccs[junc:] + ccs[:junc], junc


# ---------------------------------------------------------------------------------------------------------------
# clipped bases: decide, batch, finish
# ---------------------------------------------------------------------------------------------------------------
class _ClipJob(object):
    """One pending Smith-Waterman of clipped bases against the genomic window around the hit."""
    __slots__ = ('circ', 'hit', 'clip_seq', 'window', 'win_start', 'win_end')

    def __init__(self, circ, hit, clip_seq, window, win_start, win_end):
        self.circ, self.hit, self.clip_seq = circ, hit, clip_seq
        self.window, self.win_start, self.win_end = window, win_start, win_end


_REJECT = (None, None, None, None)


def _clip_prepare(circ, hit):
    """Everything of align_clip_segments that precedes the alignment (find_bsj.py:189-201, 227-231).
    Returns a finished 4-tuple, or a _ClipJob when Smith-Waterman is needed."""
    head_clip, tail_clip = hit.q_st, len(circ) - hit.q_en
    if head_clip + tail_clip < CLIP_MIN:
        return (circ[hit.q_st:] + circ[:hit.q_st], hit.r_st - 1, hit.r_en, (None, None, head_clip + tail_clip))
    clip_seq = circ[hit.q_en:] + circ[:hit.q_st]
    if len(clip_seq) > CLIP_MAX_FRACTION * len(circ):
        return _REJECT
    win_start = max(hit.r_st - WINDOW_FLANK, 0)
    win_end = min(hit.r_en + WINDOW_FLANK, env.CONTIG_LEN[hit.ctg])
    # the window stays a coordinate triple here (this function runs in the mapper phase, possibly in a worker process that has no
    # genome): its N count, reverse complement and encoding happen in _run_clip_jobs -- on the device when the genome is resident
    # there, from env.GENOME.seq otherwise
    return _ClipJob(circ, hit, clip_seq, None, win_start, win_end)


def _clip_finish(job, ref_begin, ref_end, query_begin, query_end):
    """Coordinates and rotation from the alignment result (find_bsj.py:205-226)."""
    circ, hit, clip_seq = job.circ, job.hit, job.clip_seq
    if hit.strand > 0:
        clip_r_st, clip_r_en = job.win_start + ref_begin, job.win_start + ref_end
        rotate = clip_r_st < hit.r_st
    else:
        clip_r_st, clip_r_en = job.win_end - ref_end, job.win_end - ref_begin
        rotate = clip_r_en > hit.r_en
    if rotate:
        clipped = clip_seq[query_begin:] + circ[hit.q_st:hit.q_en] + clip_seq[:query_begin]
    else:
        clipped = circ[hit.q_st:] + circ[:hit.q_st]
    clip_base = hit.q_st + len(circ) - hit.q_en - (query_end - query_begin) + 1
    return clipped, min(hit.r_st, clip_r_st) - 1, max(hit.r_en, clip_r_en), (clip_r_st, clip_r_en, clip_base)


def _job_tuple(job):
    """what phase 2 needs of a _ClipJob: (contig, window start, window end, minus strand, clipped bases)"""
    return (job.hit.ctg, job.win_start, job.win_end, job.hit.strand <= 0, job.clip_seq)


def _run_clip_rows(jobs):
    """Phase 2: all pending clip alignments of a chunk in ONE GPU call (scoring 1/1/1/1, find_bsj.py:204,214).
    jobs = [(contig, window start, window end, minus strand, clipped bases)] -> per job (ref_begin, ref_end, query_begin, query_end),
    or None where the window holds too many N (find_bsj.py:199-201)."""
    if not jobs:
        return []
    device = getattr(env.GENOME, 'device', None)
    if device is not None:
        import numpy as np
        from . import hip
        off, ln = device._spans([j[:3] for j in jobs])
        n_cnt = device.count_n_spans(off, ln)
        keep = np.nonzero(~(n_cnt >= WINDOW_MAX_N * ln))[0]                                           # find_bsj.py:199
        res = [None] * len(jobs)        # None = rejected by the N filter
        if len(keep):
            all_kept = len(keep) == len(jobs)
            sel = jobs if all_kept else [jobs[k] for k in keep.tolist()]
            qd, qo = hip.pack_text([j[4] for j in sel])
            rows, _ = device.ssw_windows(qd, qo, None, np.array([j[3] for j in sel], dtype=np.uint8), hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2,
                                         want_score2=False, want_cigar=False, spans=(off, ln) if all_kept else (off[keep], ln[keep]))
            if (rows['status'] & (hip.ST_NULL | hip.ST_TRACE_ERR | hip.ST_CIGAR_TRUNC)).any():
                raise RuntimeError('Smith-Waterman of clipped bases returned no result')
            coords = np.stack([rows['ref_begin1'], rows['ref_end1'], rows['read_begin1'], rows['read_end1']], axis=1).tolist()
            if all_kept:
                return [tuple(c) for c in coords]
            for k, c in zip(keep.tolist(), coords):
                res[k] = tuple(c)
        return res
    from .ssw_wrap import align_pairs
    # host-window route (no resident genome): a window is a string of up to 400 kb, so the jobs go to the GPU in groups
    # of bounded size instead of one call holding every window of the chunk at once
    res = [None] * len(jobs)            # None = rejected by the N filter (find_bsj.py:199-201)
    group, size = [], 0

    def flush():
        got = align_pairs([w for _k, w, _q in group], [q for _k, _w, q in group], match=1, mismatch=1, gap_open=1, gap_extend=1)
        for (k, _w, _q), r in zip(group, got):
            if r is None:   # the reference dereferences None here (find_bsj.py:206); make the failure explicit
                raise RuntimeError('Smith-Waterman of clipped bases returned no result')
            res[k] = (r.ref_begin, r.ref_end, r.query_begin, r.query_end)
    for k, (ctg, win_start, win_end, minus, clip_seq) in enumerate(jobs):
        window = env.GENOME.seq(ctg, win_start, win_end)
        if window.count('N') >= WINDOW_MAX_N * (win_end - win_start):
            continue
        if group and size + len(window) > HOST_WINDOW_BYTES:
            flush()
            group, size = [], 0
        # minus-strand hits are aligned against the reverse-complemented window (find_bsj.py:213-216)
        group.append((k, revcomp(window) if minus else window, clip_seq)); size += len(window)
    if group:
        flush()
    return res


def align_clip_segments(circ, hit):
    """(clipped_circ, circ_start, circ_end, (clip_r_st, clip_r_en, clip_base)) or four Nones (find_bsj.py:182-233)."""
    job = _clip_prepare(circ, hit)
    if not isinstance(job, _ClipJob):
        return job
    res = _run_clip_rows([_job_tuple(job)])[0]
    return _REJECT if res is None else _clip_finish(job, *res)


# ---------------------------------------------------------------------------------------------------------------
# chunk workers
# ---------------------------------------------------------------------------------------------------------------
def _segment_span(segments):
    parts = segments.split(';')
    return int(parts[0].split('-')[0]), int(parts[-1].split('-')[1])


def _signals(cands):
    """[(contig, start, end, clip_base)] -> [(ss_site | None, us_free, ds_free)]: host gene, annotated sites, then the
    de-novo search (find_bsj.py:286-301), one GPU batch where the genome is resident (align.find_signal_batch)."""
    return find_signal_batch([(ctg, st, en, cb, find_host_gene(ctg, st, en)) for ctg, st, en, cb in cands], True)


def _assemble(read_id, segments, ccs, circ, junc, circ_hit, clipped_circ, circ_start, circ_end, clip_info, reads_cnt, signal=None):
    """Phase 3 for one read: splice signal, coordinates, exon tags, sequence rotation (find_bsj.py:279-323).
    `signal`: this read's entry of _signals() when the caller batched that step."""
    clip_base = clip_info[2]
    if clip_base > 0.15 * len(ccs) or clip_base > 20:
        return None
    reads_cnt['bsj'] += 1

    if signal is None:
        signal = _signals([(circ_hit.ctg, circ_start, circ_end, clip_base)])[0]
    ss_site, us_free, ds_free = signal
    if ss_site is None:
        ss_id, strand, shift = 'NA', 'NA', 0
    else:
        reads_cnt['signal'] += 1
        ss_id, strand, us_shift, ds_shift = ss_site
        circ_start += us_shift
        circ_end += ds_shift
        shift = min(max(us_shift, us_free), ds_free)

    exons = merge_clip_exon(get_blocks(circ_hit), clip_info)
    exons[0][0] = circ_start
    exons[-1][1] = circ_end
    exon_tag = ','.join('{}-{}|{}'.format(st + 1, en, length) for st, en, length in exons)

    seq = clipped_circ if circ_hit.strand > 0 else revcomp(clipped_circ)
    seq = seq[shift:] + seq[:shift]           # BSJ correction for the 5' region
    return (read_id, '{}:{}-{}'.format(circ_hit.ctg, circ_start + 1, circ_end), strand, exon_tag, ss_id,
            '{}|{}-{}'.format(junc, clip_base, len(circ)), segments, seq)


THREADS = 1               # workers of the mapper phase of a chunk (set by the stage drivers from `threads`)
_POOL = None              # the thread route's executor
_PROC_POOLS = {}          # the process route's pools by role: 'scan' (first mapper: stages 2.1 and 3), 'recover' (second mapper: stage 2.2)
_WARNED = []


def mapper_mode():
    """CIRI_LONG_MAPPER=threads|processes.  Unset: processes wherever a pool exists or can still be made (the GPU untouched)."""
    import os
    m = os.environ.get('CIRI_LONG_MAPPER', '').strip().lower()
    if m not in ('', 'threads', 'processes'):
        raise ValueError('CIRI_LONG_MAPPER must be "threads" or "processes", not %r' % m)
    return m


def start_mapper_pools(threads, scan_aligner=None, recover_aligner=None, contig_len=None, scan_factory=None, recover_factory=None,
                       start='fork', gtf_index=None):
    """Make the worker processes of the mapper phase -- BEFORE this process touches the GPU (module docstring of mapper_pool).
    One pool per mapper given: `scan_*` = the first mapper (minimap2 splice preset in the reference, find_bsj.py:332; stages 2.1
    and 3), `recover_*` = the second (bwa, find_bsj.py:455; stage 2.2).  `*_aligner`: built here, inherited by forked workers (as
    the reference's Pool does); `*_factory`: picklable, builds the mapper inside each (forked or spawned) worker.  With threads <= 1 or
    CIRI_LONG_MAPPER=threads nothing is started.  `gtf_index`: the annotation index of the run (env.GTF_INDEX), if it exists already --
    the workers then also look up the host genes of their reads (find_host_gene), otherwise the parent does.  Returns the roles that
    have a pool."""
    from .mapper_pool import MapperPool
    if int(threads or 1) <= 1 or mapper_mode() == 'threads':
        return []
    for role, aligner, factory in (('scan', scan_aligner, scan_factory), ('recover', recover_aligner, recover_factory)):
        if (aligner is None and factory is None) or role in _PROC_POOLS:
            continue
        _PROC_POOLS[role] = MapperPool(int(threads), aligner=aligner, contig_len=contig_len, factory=factory, start=start, gtf_index=gtf_index)
    return sorted(_PROC_POOLS)


def stop_mapper_pools():
    for role in list(_PROC_POOLS):
        _PROC_POOLS.pop(role).close()


def _process_pool(role, aligner, contig_len, gtf_index=None):
    """The stage drivers' way to a pool: the one made at program start, or -- if this process has not touched the GPU yet (a stage
    run on its own) -- one forked now from the stage's aligner.  None: the thread route."""
    from . import mapper_pool
    if THREADS <= 1 or mapper_mode() == 'threads' or mapper_pool.in_worker():
        return None
    pool = _PROC_POOLS.get(role)
    if pool is not None:
        return pool
    if mapper_pool.gpu_touched():
        if mapper_mode() == 'processes':
            raise RuntimeError('CIRI_LONG_MAPPER=processes, but this process initialised the GPU before any mapper pool was '
                               'started: call find_bsj.start_mapper_pools() at program start')
        if not _WARNED:
            _WARNED.append(1)
            import logging
            logging.getLogger('CIRI-long').warning('mapper phase on %d threads: no worker processes were started before the GPU was '
                                                   'initialised (find_bsj.start_mapper_pools); if the mapper holds the GIL this is one core',
                                                   THREADS)
        return None
    _PROC_POOLS[role] = mapper_pool.MapperPool(THREADS, aligner=aligner, contig_len=contig_len, gtf_index=gtf_index)
    return _PROC_POOLS[role]


def _thread_pool():
    """The thread route (CIRI_LONG_MAPPER=threads, or no process pool possible any more): the mapper calls of a chunk on `THREADS`
    threads of this process.  It scales only as far as the mapper releases the GIL inside map() -- which could not be checked for
    mappy / bwapy here (neither is installable); the process route does not depend on it."""
    global _POOL
    if THREADS <= 1:
        return None
    if _POOL is None or _POOL._max_workers != THREADS:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(THREADS)
    return _POOL


def _map_read(item, raw_filters, min_circ_fraction):
    """Phase 1 for one read: everything that needs the mapper (find_bsj.py:243-273).
    -> (counter keys touched, short read or None, pending tuple or None)"""
    read_id, segments, ccs, raw = item
    keys, short = [], None
    seg_st, seg_en = _segment_span(segments)
    if raw_filters:
        # filter 1: reads that map linearly over (almost) their whole length are not circular (find_bsj.py:243-247)
        raw_hit = get_primary_alignment(env.ALIGNER.map(raw))
        if raw_hit and raw_hit.mlen > max(len(raw) * 0.8, len(raw) - 200):
            return keys, short, None
        if raw_hit and raw_hit.mlen > 1.5 * len(ccs):
            return keys, short, None
        keys.append('raw_unmapped')
        # filter 2: the raw hit must touch the repeat region (find_bsj.py:254-257)
        if raw_hit and (raw_hit.q_en < seg_st or raw_hit.q_st > seg_en):
            return keys, short, None
    ccs_hit = get_primary_alignment(env.ALIGNER.map(ccs * 2))
    if raw_filters and ccs_hit is None and len(ccs) < 150:
        short = (read_id, segments, ccs, raw)
    if ccs_hit is None or seg_en - seg_st < ccs_hit.q_en - ccs_hit.q_st:
        return keys, short, None
    keys.append('ccs_mapped')
    circ, junc = find_bsj(ccs)
    circ_hit = get_primary_alignment(env.ALIGNER.map(circ))
    if circ_hit is None or (min_circ_fraction and circ_hit.mlen < min_circ_fraction * len(circ)):
        return keys, short, None
    return keys, short, (read_id, segments, ccs, circ, junc, circ_hit, _clip_prepare(circ, circ_hit))


# ---------------------------------------------------------------------------------------------------------------
# A chunk in three phases with two batched GPU calls between them.  Each phase works on a PIECE of the chunk (a few dozen reads)
# and runs where the route puts it: on this thread (one thread), phase 1 on a thread pool (CIRI_LONG_MAPPER=threads), or all
# three on the worker processes of the mapper pool.  On that route the parent never holds a read as Python objects: what a
# phase leaves for the next one travels as one pickled blob per piece that the parent only passes on, and the parent itself
# handles columns -- windows and clipped bases for the Smith-Waterman call, candidates for the splice-signal call, text
# for the file.  Chunks overlap: while the parent is between two phases of one chunk the workers map the next ones
# (_drive), as the reference's workers never wait for its parent (find_bsj.py:343-345: every chunk submitted at once, :354-367
# results taken in order).
# ---------------------------------------------------------------------------------------------------------------
def _phase_map(items, raw_filters, min_circ_fraction):
    """phase 1 of a piece: every mapper call.  -> (counter increments, short reads, clip jobs as tuples, pending reads)"""
    cnt, shorts, jobs, pend = {}, [], [], []
    for item in items:
        keys, short, p = _map_read(item, raw_filters, min_circ_fraction)
        for k in keys:
            cnt[k] = cnt.get(k, 0) + 1
        if short is not None:
            shorts.append(short)
        if p is not None:
            pend.append(p)
            if isinstance(p[6], _ClipJob):
                jobs.append(_job_tuple(p[6]))
    return cnt, shorts, jobs, pend


def _phase_finish(pend, rows, with_hosts):
    """phase 3a of a piece: coordinates from the clip alignments (find_bsj.py:205-231), the reads that go on to the splice-signal
    step and their host genes.  rows: the piece's share of _run_clip_rows, in job order.
    -> (candidates [(contig, start, end, clip_base)], host genes or None when this process has no annotation index, ready reads)"""
    ready, it = [], iter(rows)
    for read_id, segments, ccs, circ, junc, circ_hit, prep in pend:
        if isinstance(prep, _ClipJob):
            res = next(it)
            prep = _REJECT if res is None else _clip_finish(prep, *res)
        clipped_circ, circ_start, circ_end, clip_info = prep
        if circ_start is None or circ_end is None:
            continue
        ready.append((read_id, segments, ccs, circ, junc, circ_hit, clipped_circ, circ_start, circ_end, clip_info))
    cands = [(r[5].ctg, r[7], r[8], r[9][2]) for r in ready if not (r[9][2] > 0.15 * len(r[2]) or r[9][2] > 20)]
    hosts = [find_host_gene(c, st, en) for c, st, en, _cb in cands] if with_hosts else None
    return cands, hosts, ready


def _phase_assemble(ready, sig_rows, sig_extra, as_text):
    """phase 3b of a piece: records from the ready reads and their splice signals (rows / extra of align.find_signal_rows, in
    candidate order).  -> (counter increments, records) -- records as tuples, or as (read ids, the text of cand_circ.fa)"""
    cnt = defaultdict(int)
    recs, k = [], 0
    for r in ready:
        if r[9][2] > 0.15 * len(r[2]) or r[9][2] > 20:          # (the test _assemble opens with: such a read never was a candidate)
            continue
        signal = signal_from_row(sig_rows[k] if sig_rows is not None else None, sig_extra.get(k))
        k += 1
        recs.append(_assemble(*r, cnt, signal))
    if as_text:
        return dict(cnt), ([rec[0] for rec in recs], ''.join(['>{}\t{}\t{}\t{}\t{}\t{}\t{}\n{}\n'.format(*rec) for rec in recs]))
    return dict(cnt), recs


class _Now(object):
    """the handle of work that was done on the spot"""

    def __init__(self, value):
        self.value = value

    def ready(self):
        return True

    def wait(self, timeout=None):
        pass

    def get(self):
        return self.value


class _Futures(object):
    """the handle of a list of concurrent.futures"""

    def __init__(self, futures, wake=None):
        self.futures = futures
        if wake is not None:
            for f in futures:
                f.add_done_callback(lambda _f: wake.set())

    def ready(self):
        return all(f.done() for f in self.futures)

    def wait(self, timeout=None):
        from concurrent.futures import wait
        wait(self.futures, timeout)

    def get(self):
        return [f.result() for f in self.futures]


class _Route(object):
    """where the phases of a piece run.  submit_*() -> handle with ready() / wait(timeout) / get(); get() -> one result per piece."""

    def __init__(self, raw_filters):
        import threading
        role = 'scan' if raw_filters else 'recover'
        self.procs = _process_pool(role, env.ALIGNER, env.CONTIG_LEN)
        self.threads = None if self.procs is not None else _thread_pool()
        self.depth = 4 if self.procs is not None else 1            # chunks in flight
        self.wake = threading.Event()                              # set whenever something submitted has finished: _drive sleeps on it
        workers = self.procs.workers if self.procs is not None else (THREADS if self.threads is not None else 1)
        self.piece = 32 if workers > 1 else 1 << 30
        self.workers = workers
        # a worker computes the host genes when it was given the annotation index the parent has (fork: the very object)
        self.worker_hosts = self.procs is not None and self.procs.has_index(env.GTF_INDEX)

    def pieces(self, chunk):
        # small pieces: the reads of a chunk differ a lot in mapper time (rotation loop of find_bsj), and a worker that draws a
        # long piece last is the chunk's tail
        n = max(1, min(self.piece, (len(chunk) + 4 * self.workers - 1) // (4 * self.workers)))
        return [chunk[i:i + n] for i in range(0, len(chunk), n)]

    def submit_map(self, pieces, raw_filters, min_circ_fraction):
        if self.procs is not None:
            return self.procs.submit('map', [(p, raw_filters, min_circ_fraction) for p in pieces], wake=self.wake)
        if self.threads is not None:
            return _Futures([self.threads.submit(_phase_map, p, raw_filters, min_circ_fraction) for p in pieces], self.wake)
        return _Now([_phase_map(p, raw_filters, min_circ_fraction) for p in pieces])

    def submit_finish(self, states, rows):
        if self.procs is not None:
            return self.procs.submit('finish', [(st, rw, self.worker_hosts) for st, rw in zip(states, rows)], grouped=True, wake=self.wake)
        return _Now([_phase_finish(st, rw, True) for st, rw in zip(states, rows)])

    def submit_assemble(self, states, sig_rows, sig_extra, as_text):
        if self.procs is not None:
            return self.procs.submit('assemble', [(st, rw, ex, as_text) for st, rw, ex in zip(states, sig_rows, sig_extra)], grouped=True, wake=self.wake)
        return _Now([_phase_assemble(st, rw, ex, as_text) for st, rw, ex in zip(states, sig_rows, sig_extra)])


def _chunk_program(route, chunk, raw_filters, min_circ_fraction, as_text, is_canonical=True):
    """One chunk as a generator: yields the handle it waits for, is resumed with that handle's result, returns
    (counters, short reads, records).  Everything between two yields runs in the calling process: the two GPU calls."""
    reads_cnt = defaultdict(int)
    short_reads = []
    mapped = yield route.submit_map(route.pieces(chunk), raw_filters, min_circ_fraction)
    jobs, njobs, states = [], [], []
    for cnt, shorts, jb, state in mapped:
        for k, v in cnt.items():
            reads_cnt[k] += v
        short_reads += shorts
        jobs += jb
        njobs.append(len(jb))
        states.append(state)
    rows = _run_clip_rows(jobs)                                       # GPU: K5 + prefilter + K1
    split, at = [], 0
    for n in njobs:
        split.append(rows[at:at + n]); at += n
    finished = yield route.submit_finish(states, split)
    cands, hosts, ncand, states = [], [], [], []
    for cd, hs, state in finished:
        cands += cd
        hosts += hs if hs is not None else [find_host_gene(c, st, en) for c, st, en, _cb in cd]
        ncand.append(len(cd))
        states.append(state)
    sig_rows, sig_extra = find_signal_rows(cands, hosts, is_canonical)          # GPU: K6
    srows, sextra, starts, at = [], [{} for _ in ncand], [], 0
    for n in ncand:
        srows.append(sig_rows[at:at + n].tolist() if sig_rows is not None else None)
        starts.append(at)
        at += n
    for k, v in sig_extra.items():
        piece = bisect_right(starts, k) - 1
        sextra[piece][k - starts[piece]] = v
    assembled = yield route.submit_assemble(states, srows, sextra, as_text)
    ids, parts = [], []
    for cnt, recs in assembled:
        for k, v in cnt.items():
            reads_cnt[k] += v
        if as_text:
            ids += recs[0]; parts.append(recs[1])
        else:
            parts += recs
    return reads_cnt, short_reads, ((ids, ''.join(parts)) if as_text else parts)


def _drive(programs, depth, wake=None):
    """Run chunk programs (generators as _chunk_program) with up to `depth` of them in flight; yields their results in order.
    A program is resumed as soon as what it waits for is there, oldest first -- so the GPU calls of one chunk run while the workers
    are busy with the phases of the others.  wake: an event the handles set when they finish (else the oldest handle is polled)."""
    from collections import deque
    live = deque()              # [generator, handle, result, finished]
    src = iter(programs)
    more = True

    def advance(slot, value, first=False):
        try:
            slot[1] = next(slot[0]) if first else slot[0].send(value)
        except StopIteration as stop:
            slot[1], slot[2], slot[3] = None, stop.value, True

    while True:
        while more and len(live) < depth:
            try:
                gen = next(src)
            except StopIteration:
                more = False
                break
            slot = [gen, None, None, False]
            advance(slot, None, first=True)
            live.append(slot)
        if not live:
            return
        moved = False
        if wake is not None:
            wake.clear()          # before the handles are looked at: what finishes from here on sets it again, and the wait below returns at once
        for slot in live:
            while not slot[3] and slot[1].ready():
                advance(slot, slot[1].get())
                moved = True
        while live and live[0][3]:
            yield live.popleft()[2]
            moved = True
        if not moved:
            if wake is not None:
                wake.wait(0.05)
                check = getattr(live[0][1], 'check', None)         # (a pool's handle: has a worker died?)
                if check is not None:
                    check()
            else:
                live[0][1].wait(0.02)


def _scan_chunks(chunks, raw_filters, min_circ_fraction, as_text=False):
    """(counters, short reads, records) per chunk of `chunks` (an iterable of lists of (read_id, segments, ccs, raw)), in order"""
    route = _Route(raw_filters)
    return _drive((_chunk_program(route, c, raw_filters, min_circ_fraction, as_text) for c in chunks), route.depth, route.wake)


def _scan_chunk(chunk, raw_filters, min_circ_fraction):
    for out in _scan_chunks([chunk], raw_filters, min_circ_fraction):
        return out


def scan_ccs_chunk(chunk, is_canonical):
    """One chunk of (read_id, segments, ccs, raw) -> (counters, short reads for the bwa pass, records)
    (find_bsj.py:236-325)."""
    return _scan_chunk(chunk, raw_filters=True, min_circ_fraction=0.75)


def recover_ccs_chunk(chunk, is_canonical):
    """Second pass over short consensus reads: no raw-read filters, no 0.75 coverage test (find_bsj.py:375-448)."""
    reads_cnt, _short, ret = _scan_chunk(chunk, raw_filters=False, min_circ_fraction=0)
    return reads_cnt, ret


def chunk_size_for(n, per_chunk=250):
    """reads per GPU batch: GPU_CHUNKS chunks of `per_chunk` -- fewer when the input is small, so that a handful of batches exist to
    overlap (results do not depend on it)"""
    full = per_chunk * GPU_CHUNKS
    return full if n >= 12 * full else max(per_chunk, min(full, (n + 11) // 12))


def _write_records(out, records):
    for rec in records:
        out.write('>{}\t{}\t{}\t{}\t{}\t{}\t{}\n{}\n'.format(*rec))


GPU_CHUNKS = 16           # chunks of 250 reads whose clip re-alignments share one GPU call (results do not depend on it)


def _resident(genome):
    """Wrap a genome that exposes its sequences ({name: str} in `.genome`, as align.Fasta does) so that the Smith-Waterman
    windows are read from a copy resident on the GPU; anything else (e.g. a mappy index) is used as it is."""
    from .align import DeviceGenome
    if getattr(genome, 'device', None) is not None or not isinstance(getattr(genome, 'genome', None), dict):
        return genome
    return DeviceGenome(genome, genome.genome)


def scan_ccs_reads(ccs_seq, ref_fasta, ss_index, gtf_index, intron_index, is_canonical, out_dir, prefix, threads,
                   aligner=None, genome=None, contig_len=None):
    """Stage driver (find_bsj.py:328-372).  The reference forks a process pool per stage and runs the whole per-read loop there; here
    the calling process (one per GPU) walks the chunks itself, the mapper phase of a chunk on `threads` worker processes
    (mapper_pool.py; threads of this process with CIRI_LONG_MAPPER=threads), the batched GPU phases on the calling thread.
    ``aligner``/``genome``/``contig_len`` may be injected (tests, or an already built index); by default a mappy splice-preset
    aligner is built from ``ref_fasta`` exactly as the reference does."""
    global THREADS
    if aligner is None:
        import mappy as mp
        aligner = mp.Aligner(ref_fasta, n_threads=threads, preset='splice')
    if genome is None:
        # the reference hands the mappy index itself to the workers as GENOME (find_bsj.py:340-341).  Here the FASTA is read
        # once: it gives the contig lengths AND, folded the way the index folds it (upper case, anything but ACGT is N; no
        # sequence for a start before the contig: align.IndexGenome), becomes the genome resident in HBM, so that clip windows
        # are coordinates (K5/K1) and the splice-signal search runs on the device (K6) with the main pass' semantics; the
        # short-read pass keeps the raw text, as the reference does (recover_ccs_reads)
        from .align import Fasta, IndexGenome
        genome = IndexGenome(Fasta(ref_fasta))
    if contig_len is None:
        contig_len = genome.contig_len if hasattr(genome, 'contig_len') else None
        if contig_len is None:
            from .align import Fasta
            contig_len = Fasta(ref_fasta).contig_len
    THREADS = max(1, int(threads or 1))
    _process_pool('scan', aligner, contig_len, gtf_index)      # (worker processes, if they can still be made, BEFORE the genome goes to the GPU)
    env.initializer(aligner, contig_len, _resident(genome), gtf_index, intron_index, ss_index)

    reads_count = defaultdict(int)
    short_reads = []
    names = list(ccs_seq)
    chunks = ([[i, ] + ccs_seq[i] for i in reads if i is not None] for reads in grouper(names, chunk_size_for(len(names))))
    with open('{}/{}.cand_circ.fa'.format(out_dir, prefix), 'w') as out:
        for cnt, short, (_ids, text) in _scan_chunks(chunks, True, 0.75, as_text=True):
            for key, value in cnt.items():
                reads_count[key] += value
            short_reads += short
            out.write(text)
    return reads_count, short_reads


def recover_ccs_reads(short_reads, ref_fasta, ss_index, gtf_index, intron_index, is_canonical, out_dir, prefix, threads,
                      aligner=None, genome=None):
    """bwa pass over the short consensus reads, appended to the same output (find_bsj.py:451-490)."""
    from .align import Aligner, Fasta
    if genome is None:
        genome = Fasta(ref_fasta)
    if aligner is None:
        from bwapy import BwaAligner
        aligner = Aligner(BwaAligner(ref_fasta, options='-x ont2d -T 19'))
    global THREADS
    THREADS = max(1, int(threads or 1))
    _process_pool('recover', aligner, genome.contig_len, gtf_index)
    env.initializer(aligner, genome.contig_len, _resident(genome), gtf_index, intron_index, ss_index)

    reads_count = defaultdict(int)
    chunks = ([i for i in reads if i is not None] for reads in grouper(short_reads, chunk_size_for(len(short_reads))))
    with open('{}/{}.cand_circ.fa'.format(out_dir, prefix), 'a') as out:
        for cnt, _short, (_ids, text) in _scan_chunks(chunks, False, 0, as_text=True):
            for key, value in cnt.items():
                reads_count[key] += value
            out.write(text)
    return reads_count


# ---------------------------------------------------------------------------------------------------------------
# third stage of `call`: reads without a cyclic consensus that still span one junction ("partial" candidates).
# Mapper logic only -- no alignment kernel is involved; kept so that the module offers every stage of the reference's.
# ---------------------------------------------------------------------------------------------------------------
def _primary_hits(seq):
    return sorted([h for h in (env.ALIGNER.map(seq) or ()) if h.is_primary], key=lambda h: [h.q_st, h.q_en])     # (a mapper double may answer None)


def _raw_junction(seq, raw_hits):
    """(circ, junc) when the raw read's primary hits look like one pass over a junction (find_bsj.py:512-541)"""
    n = len(seq)
    if len(raw_hits) == 1:
        hit = remove_long_insert(raw_hits[0])
        if hit.mlen < n * .45 or hit.mlen > n - 50:
            return None
        if hit.q_st < 50 and hit.q_en > n - 50:
            return None
        circ, junc = find_bsj(seq)
        return None if junc is None else (circ, junc)
    if len(raw_hits) == 2:
        head, tail = remove_long_insert(raw_hits[0]), remove_long_insert(raw_hits[1])
        if head.ctg != tail.ctg or not head.q_st + head.mlen * 0.45 < tail.q_st:
            return None
        if head.r_en - 20 < tail.r_st or head.q_en < tail.q_st - 50:
            return None
        circ, junc = find_bsj(seq)
        if junc is None or junc < head.q_en - 10 or junc > tail.q_st + 10:
            return None
        return circ, junc
    return None


def _raw_layout(seq, circ, junc, raw_hits):
    """(ctg, start, end, strand, clip_base, exons, circ) from the hits of the rotated read (find_bsj.py:543-579)"""
    n = len(seq)
    circ_hits = sorted([remove_long_insert(h) for h in (env.ALIGNER.map(circ) or ()) if h.is_primary], key=lambda h: [h.q_st, h.q_en])
    if len(circ_hits) == 1:
        hit = circ_hits[0]
        if hit.mlen <= max(h.mlen for h in raw_hits) or min(junc, n - junc) < 30:
            return None
        if not junc + hit.q_st < n < junc + hit.q_en:
            return None
        return hit.ctg, hit.r_st, hit.r_en, hit.strand, hit.q_st + n - hit.q_en, get_parital_blocks(hit, n - junc), circ
    if len(circ_hits) == 2:
        head, tail = circ_hits
        if head.ctg != tail.ctg or head.strand != tail.strand:
            return None
        if not head.q_st + (head.q_en - head.q_st) * 0.5 < tail.q_st:
            return None
        if head.r_en - 20 < tail.r_st or head.q_en < tail.q_st - 20:
            return None
        exons = merge_exons(get_blocks(tail), get_blocks(head))
        return head.ctg, tail.r_st, head.r_en, head.strand, abs(tail.q_st - head.q_en), exons, circ[tail.q_st:] + circ[:tail.q_st]
    return None


def _raw_map_read(read_id, seq):
    """Phase 1 of stage 3 for one read: every mapper call (find_bsj.py:508-579).  -> None, ('short', (read_id, seq)) or
    ('laid', (read_id, junc, layout))"""
    if len(seq) < 300:
        return 'short', (read_id, seq)
    raw_hits = _primary_hits(seq)
    if not raw_hits:
        return None
    cand = _raw_junction(seq, raw_hits)
    if cand is None:
        return None
    circ, junc = cand
    layout = _raw_layout(seq, circ, junc, raw_hits)
    if layout is None or layout[4] > 20:
        return None
    return 'laid', (read_id, junc, layout)


def _raw_program(chunk, is_canonical, circ_reads):
    """scan_raw_chunk as a program for _drive: the mapper phase of the chunk on the workers while the parent finishes the one before"""
    reads_cnt = defaultdict(int)
    ret, short_reads, laid = [], [], []
    todo = [(read_id, seq) for read_id, seq in chunk if read_id not in circ_reads]
    procs = _process_pool('scan', env.ALIGNER, env.CONTIG_LEN)
    if procs is not None:
        mapped = yield procs.raw_async(todo)
    else:
        pool = _thread_pool()
        mapped = [_raw_map_read(*it) for it in todo] if pool is None else list(pool.map(lambda it: _raw_map_read(*it), todo))
    for m in mapped:
        if m is not None:
            (short_reads if m[0] == 'short' else laid).append(m[1])
    signals = _signals([(lay[0], lay[1], lay[2], lay[4]) for _, _, lay in laid])
    for (read_id, junc, layout), (ss_site, us_free, ds_free) in zip(laid, signals):
        ctg, start, end, circ_strand, clip_base, exons, circ = layout
        if ss_site is None:
            ss_id, strand, shift = 'NA', 'NA', 0
        else:
            ss_id, strand, us_shift, ds_shift = ss_site
            start += us_shift
            end += ds_shift
            shift = min(max(us_shift, -us_free), ds_free)       # sign of us_free as in find_bsj.py:598 (differs from :300)
        exons[0][0] = start
        exons[-1][1] = end
        exon_tag = ','.join('{}-{}|{}'.format(a, b, length) for a, b, length in exons)     # 0-based starts here (find_bsj.py:606)
        out_seq = circ if circ_strand > 0 else revcomp(circ)
        out_seq = out_seq[shift:] + out_seq[:shift]
        ret.append((read_id, '{}:{}-{}'.format(ctg, start + 1, end), strand, exon_tag, ss_id, '{}|{}-NA'.format(junc, clip_base),
                    'partial', out_seq))
        reads_cnt['partial'] += 1
    return reads_cnt, ret, short_reads


def _raw_chunks(chunks, is_canonical, circ_reads):
    """(counters, 'partial' records, short reads) per chunk, in order; two chunks in flight when the mapper phase has workers"""
    depth = 2 if _process_pool('scan', env.ALIGNER, env.CONTIG_LEN) is not None else 1
    return _drive((_raw_program(c, is_canonical, circ_reads) for c in chunks), depth)


def scan_raw_chunk(chunk, is_canonical, circ_reads):
    """[(read_id, seq)] -> (counters, 'partial' records, short reads) (find_bsj.py:499-620)"""
    for out in _raw_chunks([chunk], is_canonical, circ_reads):
        return out


def scan_raw_reads(in_file, ref_fasta, gtf_index, intron_index, ss_index, is_canonical, out_dir, prefix, threads,
                   aligner=None, genome=None, contig_len=None):
    """Stage driver (find_bsj.py:623-720): every read of the input that is not already in cand_circ.fa, chunks of 1000,
    records to {prefix}.low_confidence.fa.  ``aligner``/``genome``/``contig_len`` may be injected as in scan_ccs_reads
    (the reference serves sequences from the mappy index itself)."""
    from .find_ccs import iter_reads
    circ_reads = {}
    with open('{}/{}.cand_circ.fa'.format(out_dir, prefix), 'r') as f:
        for line in f:
            circ_reads[line.rstrip().split()[0].lstrip('>')] = 1
            f.readline()
    if aligner is None:
        import mappy as mp
        aligner = mp.Aligner(ref_fasta, n_threads=threads, preset='splice')
    if contig_len is None:
        from .align import Fasta
        contig_len = Fasta(ref_fasta).contig_len
    env.initializer(aligner, contig_len, aligner if genome is None else genome, gtf_index, intron_index, ss_index)
    global THREADS
    THREADS = max(1, int(threads or 1))
    reads_cnt = defaultdict(int)
    short_reads = []
    with open('{}/{}.low_confidence.fa'.format(out_dir, prefix), 'w') as out:
        for cnt, ret, short in _raw_chunks(([r for r in reads if r is not None] for reads in grouper(iter_reads(in_file), 1000)), is_canonical, circ_reads):
            for key, value in cnt.items():
                reads_cnt[key] += value
            short_reads += short
            _write_records(out, ret)
    return reads_cnt, short_reads
