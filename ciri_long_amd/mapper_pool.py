"""Worker processes for the mapper phase of `call` (counterpart of the reference's ``Pool(threads, env.initializer, ...)``,
CIRI_long/find_bsj.py:338-345, 459-466, 636-643; CIRI_long/env.py:9-22).

The reference spreads its whole per-read loop -- mapper calls included -- over ``threads`` forked processes.  Here one process
per GPU owns the device, and a process that has initialised the GPU must neither fork nor start children that exec.  So the
pool is made BEFORE anything touches the GPU (``find_bsj.start_mapper_pools`` at program start; ``dist.call_sharded`` does it
first thing), its workers run only phase 1 of a chunk (``find_bsj._map_read`` / ``_raw_map_read``: every call of
``env.ALIGNER.map``) and hand the pending tuples back; the batched GPU phases stay in the parent.  A worker never loads
libclh (``hip.lib()`` refuses in a worker).

Two ways to give a worker its mapper:
  * ``start='fork'`` (default): the parent builds the aligner, the workers inherit it -- what the reference does, and the only
    way a multi-gigabyte minimap2 index is shared instead of loaded once per worker;
  * ``start='spawn'``: ``factory`` (picklable, no arguments) builds the aligner in each worker.
Whether ``mappy.Aligner.map`` releases the GIL is irrelevant on this route.
"""
import multiprocessing
import os
import sys

_FORK_PAYLOAD = None       # (aligner, contig_len) a forked worker finds here


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


def _worker_init(factory, contig_len):
    os.environ['CIRI_LONG_MAPPER_WORKER'] = '1'
    from . import env
    if factory is None:
        aligner, contig_len = _FORK_PAYLOAD
    else:
        aligner = factory()
    # as env.initializer in the reference's workers; the genome and the annotation indices are the parent's business (phases 2-3)
    env.initializer(aligner, contig_len, None, None, None, None)


def _worker_scan(task):
    items, raw_filters, min_circ_fraction = task
    from . import find_bsj
    return [find_bsj._map_read(it, raw_filters, min_circ_fraction) for it in items]


def _worker_raw(items):
    from . import find_bsj
    return [find_bsj._raw_map_read(read_id, seq) for read_id, seq in items]


class MapperPool(object):
    """``workers`` processes, each with its own (or the inherited) mapper; ``scan`` / ``raw`` keep the input order."""

    def __init__(self, workers, aligner=None, contig_len=None, factory=None, start='fork', piece=32):
        global _FORK_PAYLOAD
        if gpu_touched():
            raise RuntimeError('mapper pool: this process has initialised the GPU; worker processes must be started before that '
                               '(find_bsj.start_mapper_pools at program start)')
        if in_worker():
            raise RuntimeError('mapper pool inside a mapper worker')
        if start == 'fork' and factory is None:
            if aligner is None:
                raise ValueError('mapper pool: fork needs the aligner built in the parent')
            _FORK_PAYLOAD = (aligner, contig_len)
        elif factory is None:
            raise ValueError('mapper pool: spawn needs a picklable factory that builds the aligner in the worker')
        self.workers, self.piece = int(workers), int(piece)
        ctx = multiprocessing.get_context(start)
        self._pool = ctx.Pool(self.workers, _worker_init, (factory, contig_len))
        _FORK_PAYLOAD = None

    def _pieces(self, items):
        # small pieces: the reads of a chunk differ a lot in mapper time (rotation loop of find_bsj), and a worker that draws a
        # long piece last is the chunk's tail
        n = max(1, min(self.piece, (len(items) + 4 * self.workers - 1) // (4 * self.workers)))
        return [items[i:i + n] for i in range(0, len(items), n)]

    def scan(self, chunk, raw_filters, min_circ_fraction):
        out = self._pool.map(_worker_scan, [(p, raw_filters, min_circ_fraction) for p in self._pieces(list(chunk))])
        return [x for part in out for x in part]

    def raw(self, items):
        out = self._pool.map(_worker_raw, self._pieces(list(items)))
        return [x for part in out for x in part]

    def close(self):
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
