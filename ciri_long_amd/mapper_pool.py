"""Worker processes for the mapper phase of `call` (counterpart of the reference's ``Pool(threads, env.initializer, ...)``,
CIRI_long/find_bsj.py:338-345, 459-466, 636-643; CIRI_long/env.py:9-22).

The reference spreads its whole per-read loop -- mapper calls included -- over ``threads`` forked processes.  Here one process
per GPU owns the device, and a process that has initialised the GPU must neither fork nor start children that exec.  So the
pool is made BEFORE anything touches the GPU (``find_bsj.start_mapper_pools`` at program start; ``dist.call_sharded`` does it
first thing).  Its workers run every per-read phase of a chunk -- phase 1 (``find_bsj._phase_map``: every call of
``env.ALIGNER.map``), and the two halves of phase 3 around the splice-signal call (``_phase_finish``, ``_phase_assemble``:
coordinate arithmetic, exon tags, record text) -- while the parent keeps only the two batched GPU calls between them and never
unpacks a read: what one phase leaves for the next is a pickled blob it passes on (``submit``).  Stage 3 maps its reads there too
(``_raw_map_read``).  A worker never loads libclh (``hip.lib()`` refuses in a worker).

Two ways to give a worker its mapper:
  * ``start='fork'`` (default): the parent builds the aligner, the workers inherit it -- what the reference does, and the only
    way a multi-gigabyte minimap2 index is shared instead of loaded once per worker;
  * ``start='spawn'``: ``factory`` (picklable, no arguments) builds the aligner in each worker.
Whether ``mappy.Aligner.map`` releases the GIL is irrelevant on this route.
"""
import multiprocessing
import os
import pickle
import sys
import time

_FORK_PAYLOAD = None       # (aligner, contig_len) a forked worker finds here


class WorkerDied(RuntimeError):
    pass


def gpu_touched():
    """True once this process may hold GPU state: a device context of libclh was created (hip.Context: the first call into the HIP
    runtime -- loading the library and its host-only entry points, e.g. the record count of dist.call_sharded, do not initialise it),
    or torch's HIP runtime is initialised"""
    hip = sys.modules.get(__package__ + '.hip')
    if hip is not None and getattr(hip, '_gpu_used', False):
        return True
    torch = sys.modules.get('torch')
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


def in_worker():
    return os.environ.get('CIRI_LONG_MAPPER_WORKER') == '1'


def _count_start(starts, payload_needed):
    """Every worker counts itself in.  multiprocessing.Pool replaces a worker that died (killed for memory, a crash inside the mapper) without a
    word, and the task the dead one held is never answered: the parent would wait for ever.  A count above the pool's size tells it (MapperPool.check).
    A replacement that was forked AFTER the pool's construction has no mapper to inherit (and its parent may hold GPU state by now): it stays out of
    the way until the parent, told by the count, ends the pools."""
    with starts.get_lock():
        starts.value += 1
    if payload_needed and _FORK_PAYLOAD is None:
        while True:
            time.sleep(3600)


def _worker_init(factory, contig_len, gtf_index, starts):
    os.environ['CIRI_LONG_MAPPER_WORKER'] = '1'
    _count_start(starts, factory is None)
    from . import env
    if factory is None:
        aligner, contig_len, gtf_index = _FORK_PAYLOAD
    else:
        aligner = factory()
    # as env.initializer in the reference's workers; the genome and the splice-site index are the parent's business (the GPU calls)
    env.initializer(aligner, contig_len, None, gtf_index, None, None)


def _light_init(contig_len, gtf_index, starts, forked):
    """a worker of the light pool: the two halves of phase 3 only -- no mapper"""
    os.environ['CIRI_LONG_MAPPER_WORKER'] = '1'
    _count_start(starts, forked)
    from . import env
    if gtf_index is None and _FORK_PAYLOAD is not None:
        _aligner, contig_len, gtf_index = _FORK_PAYLOAD
    env.initializer(None, contig_len, None, gtf_index, None, None)


def _worker_map(task):
    items, raw_filters, min_circ_fraction = task
    from . import find_bsj
    cnt, shorts, jobs, pend = find_bsj._phase_map(items, raw_filters, min_circ_fraction)
    return cnt, shorts, jobs, pickle.dumps(pend, -1)


def _worker_finish(group):
    from . import find_bsj
    out = []
    for blob, rows, with_hosts in group:
        cands, hosts, ready = find_bsj._phase_finish(pickle.loads(blob), rows, with_hosts)
        out.append((cands, hosts, pickle.dumps(ready, -1)))
    return out


def _worker_assemble(group):
    from . import find_bsj
    return [find_bsj._phase_assemble(pickle.loads(blob), rows, extra, as_text) for blob, rows, extra, as_text in group]


def _worker_raw(items):
    from . import find_bsj
    return [find_bsj._raw_map_read(read_id, seq) for read_id, seq in items]


def _worker_scan(task):
    items, raw_filters, min_circ_fraction = task
    from . import find_bsj
    return [find_bsj._map_read(it, raw_filters, min_circ_fraction) for it in items]


_TASKS = {'map': _worker_map, 'finish': _worker_finish, 'assemble': _worker_assemble, 'raw': _worker_raw}


class _Handle(object):
    """an AsyncResult whose get() gives one result per submitted task (grouped tasks flattened back); waiting on it notices a dead worker"""

    def __init__(self, result, grouped, check=None):
        self._r, self._grouped, self._check = result, grouped, check

    def ready(self):
        return self._r.ready()

    def check(self):
        if self._check is not None and not self._r.ready():
            self._check()

    def wait(self, timeout=None):
        self._r.wait(timeout)
        self.check()

    def get(self):
        while not self._r.ready():
            self._r.wait(0.5)
            self.check()
        out = self._r.get()
        return [x for part in out for x in part] if self._grouped else out


class MapperPool(object):
    """``workers`` processes, each with its own (or the inherited) mapper; ``scan`` / ``raw`` keep the input order."""

    def __init__(self, workers, aligner=None, contig_len=None, factory=None, start='fork', piece=32, gtf_index=None):
        global _FORK_PAYLOAD
        if gpu_touched():
            raise RuntimeError('mapper pool: this process has initialised the GPU; worker processes must be started before that '
                               '(find_bsj.start_mapper_pools at program start)')
        if in_worker():
            raise RuntimeError('mapper pool inside a mapper worker')
        if start == 'fork' and factory is None:
            if aligner is None:
                raise ValueError('mapper pool: fork needs the aligner built in the parent')
            _FORK_PAYLOAD = (aligner, contig_len, gtf_index)
        elif factory is None:
            raise ValueError('mapper pool: spawn needs a picklable factory that builds the aligner in the worker')
        self.workers, self.piece = int(workers), int(piece)
        self._gtf_index = gtf_index
        ctx = multiprocessing.get_context(start)
        nlight = max(2, self.workers // 4)
        self._starts, self._expected = ctx.Value('i', 0), self.workers + nlight
        self._pool = ctx.Pool(self.workers, _worker_init, (factory, contig_len, gtf_index if factory is not None else None, self._starts))
        # The halves of phase 3 (finish, assemble) are a few microseconds per read, but a pool serves its tasks first come, first served: behind
        # the mapper pieces of the chunks in flight they would wait tens of milliseconds, the oldest chunk could not retire, and the next chunk
        # would be submitted only when the mapper workers had run dry.  They get processes of their own (a quarter as many: they are mostly idle).
        if start == 'fork' and factory is None:
            self._light = ctx.Pool(nlight, _light_init, (contig_len, None, self._starts, True))
        else:
            self._light = ctx.Pool(nlight, _light_init, (contig_len, gtf_index, self._starts, False))
        _FORK_PAYLOAD = None

    def check(self):
        """raises WorkerDied once a worker process has been replaced: the task it held is lost, and nothing would ever say so"""
        if self._starts.value > self._expected:
            raise WorkerDied('mapper pool: a worker process died (killed for memory? a crash inside the mapper?) -- %d processes were started for '
                             '%d places; the reads it held are lost and the run cannot go on' % (self._starts.value, self._expected))

    def has_index(self, gtf_index):
        """True when a worker's find_host_gene answers as the caller's would: it was given this very index, or there is none"""
        return not gtf_index or self._gtf_index is gtf_index

    def _pieces(self, items):
        # small pieces: the reads of a chunk differ a lot in mapper time (rotation loop of find_bsj), and a worker that draws a
        # long piece last is the chunk's tail
        n = max(1, min(self.piece, (len(items) + 4 * self.workers - 1) // (4 * self.workers)))
        return [items[i:i + n] for i in range(0, len(items), n)]

    def submit(self, kind, tasks, grouped=False, wake=None):
        """tasks of one kind ('map', 'finish', 'assemble', 'raw') to the workers, without waiting: -> handle with ready(), wait(timeout),
        get() -> one result per task, in order (wake: a threading.Event set when they are all done).  grouped: the tasks are light (the halves of phase 3) -- a worker of the light pool takes a run of them per
        message, two runs per worker."""
        if grouped:
            light = max(2, self.workers // 4)
            n = max(1, (len(tasks) + 2 * light - 1) // (2 * light))
            tasks = [tasks[i:i + n] for i in range(0, len(tasks), n)]
        cb = (lambda _x: wake.set()) if wake is not None else None
        pool = self._light if kind in ('finish', 'assemble') else self._pool
        return _Handle(pool.map_async(_TASKS[kind], tasks, 1, cb, cb), grouped, self.check)

    def scan(self, chunk, raw_filters, min_circ_fraction):
        """phase 1 of a chunk, input order kept: [(counter keys touched, short read or None, pending tuple or None)] per read"""
        from . import find_bsj
        return _Handle(self._pool.map_async(_worker_scan, [(p, raw_filters, min_circ_fraction) for p in self._pieces(list(chunk))], 1), True, self.check).get()

    def raw_async(self, items):
        return _Handle(self._pool.map_async(_worker_raw, self._pieces(list(items)), 1), True, self.check)

    def raw(self, items):
        return self.raw_async(items).get()

    def close(self):
        for name in ('_pool', '_light'):
            pool = getattr(self, name, None)
            if pool is not None:
                pool.terminate()
                pool.join()
                setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
