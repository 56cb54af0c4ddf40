"""The mapper phase of stage 2 / 3 on worker processes (ciri_long_amd/mapper_pool.py; the reference's
Pool(threads, env.initializer, ...), CIRI_long/find_bsj.py:338-345, 459-466, 636-643).

The mapper double here holds the GIL inside map() (a pure-Python loop) -- the case the thread route cannot scale on and that nobody
could rule out for mappy / bwapy.  Records, counters and short-read lists must not depend on the route; the wall time must scale
with the workers under `processes` and must not under `threads`."""
import os
import sys
import time

import pytest

import fake_mapper as fm
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
SPIN = 150000         # iterations of the busy loop per map() call (~8 ms of pure Python: well above the C part of the double, which ctypes runs WITHOUT the lock)


class BusyMapper(fm.FakeMapper):
    """FakeMapper whose map() first spins in pure Python: the interpreter lock is held all the while"""

    def map(self, seq):
        x = 0
        for i in range(SPIN):
            x += i & 3
        return fm.FakeMapper.map(self, seq)


def make_busy_mapper():
    """factory for spawned workers (picklable: a module-level function)"""
    return BusyMapper(fm.build_world()['genome'], min_score=170)


class HostGenome(object):
    """the genome double without its `.genome` dict: find_bsj._resident leaves it on the host (the window route that needs no GPU
    for anything but the Smith-Waterman call, which the `world` fixture hands to the CPU checker)"""

    def __init__(self, g):
        self._g, self.contig_len = g, g.contig_len

    def seq(self, ctg, start, end):
        return self._g.seq(ctg, start, end)


class _Res(object):
    def __init__(self, d):
        self.score, self.ref_begin, self.ref_end = d['score'], d['ref_begin'], d['ref_end']
        self.query_begin, self.query_end = d['query_begin'], d['query_end']


def _oracle_pairs(refs, queries, match=2, mismatch=2, gap_open=3, gap_extend=1, **kw):
    return [_Res(oracle_lib.oracle_align(r, q, match, mismatch, gap_open, gap_extend)) for r, q in zip(refs, queries)]


@pytest.fixture()
def world(monkeypatch):
    from ciri_long_amd import find_bsj, ssw_wrap
    monkeypatch.setattr(ssw_wrap, 'align_pairs', _oracle_pairs)        # phases 2-3 stay in this process: the CPU checker for the GPU call
    w = fm.build_world()
    w['reads'] = fm.build_reads(w, 48)
    w['host_genome'] = HostGenome(w['genome'])
    yield w
    find_bsj.stop_mapper_pools()
    find_bsj.THREADS = 1


def _stage2(w, tmp, tag, threads, mapper=None, **kw):
    from ciri_long_amd import find_bsj
    d = tmp / tag
    d.mkdir()
    ccs_seq = {r[0]: [r[1], r[2], r[3]] for r in w['reads']}
    t0 = time.perf_counter()
    cnt, short = find_bsj.scan_ccs_reads(ccs_seq, None, w['ss_index'], w['gtf_index'], None, True, str(d), 'p', threads,
                                         aligner=mapper or BusyMapper(w['genome'], min_score=170), genome=w['host_genome'], contig_len=w['genome'].contig_len, **kw)
    el = time.perf_counter() - t0
    return dict(cnt), short, (d / 'p.cand_circ.fa').read_bytes(), el


def test_processes_give_the_same_records_and_scale_where_threads_cannot(world, tmp_path, monkeypatch):
    from ciri_long_amd import find_bsj
    monkeypatch.delenv('CIRI_LONG_MAPPER', raising=False)
    _stage2(world, tmp_path, 'warm', 1)                          # (first-call costs -- imports, the checker library -- stay out of the timings)
    one = _stage2(world, tmp_path, 'one', 1)
    assert one[0]['ccs_mapped'] > 10 and len(one[2]) > 1000 and not find_bsj._PROC_POOLS
    monkeypatch.setenv('CIRI_LONG_MAPPER', 'threads')
    thr = _stage2(world, tmp_path, 'thr', 4)
    assert not find_bsj._PROC_POOLS
    monkeypatch.setenv('CIRI_LONG_MAPPER', 'processes')
    find_bsj.THREADS = 4
    find_bsj.start_mapper_pools(4, scan_aligner=BusyMapper(world['genome'], min_score=170), contig_len=world['genome'].contig_len)
    assert sorted(find_bsj._PROC_POOLS) == ['scan']
    prc = _stage2(world, tmp_path, 'prc', 4)
    for got in (thr, prc):
        assert got[0] == one[0] and got[1] == one[1] and got[2] == one[2]         # counters, short reads, cand_circ.fa byte for byte
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 4:
        # a mapper that holds the GIL: four threads buy nothing, four processes most of a factor of four.  (Wall times on a shared box: a
        # second measurement before the verdict, the better of the two counts for the processes, the worse for nothing.)
        t_one, t_thr, t_prc = one[3], thr[3], prc[3]
        if not t_prc < 0.6 * t_one:
            t_one = min(t_one, _stage2(world, tmp_path, 'one2', 1)[3])
            t_prc = min(t_prc, _stage2(world, tmp_path, 'prc2', 4)[3])
        assert t_thr > 0.6 * t_one, (t_one, t_thr)          # (the double's C part -- the checker's alignments through ctypes -- does run in parallel)
        assert t_prc < 0.6 * t_one and t_prc < 0.8 * t_thr, (t_one, t_thr, t_prc)


def test_default_mode_forks_a_pool_when_the_gpu_is_untouched_and_falls_back_to_threads_when_it_is_not(world, tmp_path, monkeypatch, caplog):
    from ciri_long_amd import find_bsj, mapper_pool
    monkeypatch.delenv('CIRI_LONG_MAPPER', raising=False)
    one = _stage2(world, tmp_path, 'one', 1)
    got = _stage2(world, tmp_path, 'auto', 3)                  # no pool made up front: the stage driver forks one from its aligner
    assert sorted(find_bsj._PROC_POOLS) == ['scan'] and got[:3] == one[:3]
    find_bsj.stop_mapper_pools()
    monkeypatch.setattr(mapper_pool, 'gpu_touched', lambda: True)
    find_bsj._WARNED.clear()
    with caplog.at_level('WARNING', logger='CIRI-long'):
        got = _stage2(world, tmp_path, 'late', 3)
    assert not find_bsj._PROC_POOLS and got[:3] == one[:3] and 'no worker processes' in caplog.text
    monkeypatch.setenv('CIRI_LONG_MAPPER', 'processes')       # asked for by name: no silent fall-back
    with pytest.raises(RuntimeError, match='start_mapper_pools'):
        _stage2(world, tmp_path, 'late2', 3)
    with pytest.raises(RuntimeError, match='initialised the GPU'):
        mapper_pool.MapperPool(2, aligner=object())
    monkeypatch.setenv('CIRI_LONG_MAPPER', 'both')
    with pytest.raises(ValueError):
        find_bsj.mapper_mode()


def test_spawned_workers_build_their_own_mapper_and_never_load_the_gpu_library(world, tmp_path, monkeypatch):
    from ciri_long_amd import find_bsj, mapper_pool
    monkeypatch.delenv('CIRI_LONG_MAPPER', raising=False)
    monkeypatch.setenv('PYTHONPATH', os.pathsep.join([HERE, os.path.dirname(HERE), os.environ.get('PYTHONPATH', '')]))
    one = _stage2(world, tmp_path, 'one', 1, mapper=make_busy_mapper())
    find_bsj.start_mapper_pools(2, scan_factory=make_busy_mapper, contig_len=world['genome'].contig_len, start='spawn')
    got = _stage2(world, tmp_path, 'spawn', 2, mapper=make_busy_mapper())
    assert got[:3] == one[:3]
    # what a worker sees: hip.lib() refuses, gpu_touched() stays false
    pool = find_bsj._PROC_POOLS['scan']._pool
    assert pool.apply(_probe_worker) == ('HipUnavailable', False, True)


def _probe_worker():
    from ciri_long_amd import hip, mapper_pool
    try:
        hip.lib()
        err = None
    except Exception as ex:
        err = type(ex).__name__
    return err, mapper_pool.gpu_touched(), mapper_pool.in_worker()


def test_stage_2_2_and_stage_3_through_their_pools(world, tmp_path, monkeypatch):
    """recover_ccs_reads (second mapper, its own pool) and scan_raw_reads (the first mapper's pool again): same files as one thread"""
    from ciri_long_amd import find_bsj
    monkeypatch.delenv('CIRI_LONG_MAPPER', raising=False)
    w = world
    short = [(r[0], r[1], r[2], r[3]) for r in w['reads'] if len(r[2]) < 400][:24]
    fa = tmp_path / 'in.fa'
    with open(fa, 'w') as f:
        for r in w['reads']:
            f.write('>{}\n{}\n'.format(r[0], r[3]))
        for k in range(12):           # reads that span one junction once (stage 3's business) and short ones
            ctg, exons, strand = w['circs'][k % len(w['circs'])]
            circ = ''.join(w['genome'].genome[ctg][a:b] for a, b in exons)
            f.write('>part{}\n{}\n'.format(k, (circ[len(circ) // 2:] + circ[:len(circ) // 2 + 40]) if k % 3 else circ[:200]))
    out = {}
    for tag, threads in (('one', 1), ('pool', 3)):
        d = tmp_path / tag
        d.mkdir()
        (d / 'p.cand_circ.fa').write_text('>read000\nACGT\n')
        if threads > 1:
            find_bsj.start_mapper_pools(threads, scan_aligner=BusyMapper(w['genome'], min_score=170), recover_aligner=BusyMapper(w['genome'], min_score=30),
                                        contig_len=w['genome'].contig_len)
            assert sorted(find_bsj._PROC_POOLS) == ['recover', 'scan']
        c2 = find_bsj.recover_ccs_reads(short, None, w['ss_index'], w['gtf_index'], None, True, str(d), 'p', threads,
                                        aligner=BusyMapper(w['genome'], min_score=30), genome=w['host_genome'])
        c3, s3 = find_bsj.scan_raw_reads(str(fa), None, w['gtf_index'], None, w['ss_index'], True, str(d), 'p', threads,
                                         aligner=BusyMapper(w['genome'], min_score=170), genome=w['host_genome'], contig_len=w['genome'].contig_len)
        out[tag] = (dict(c2), dict(c3), s3, (d / 'p.cand_circ.fa').read_bytes(), (d / 'p.low_confidence.fa').read_bytes())
    assert out['one'] == out['pool']
    assert len(out['one'][3]) > 20 and len(out['one'][2]) >= 4


class FailingMapper(fm.FakeMapper):
    """raises inside a worker on the third read's consensus"""

    def __init__(self, genome, poison, **kw):
        fm.FakeMapper.__init__(self, genome, **kw)
        self.poison = poison

    def map(self, seq):
        if seq == self.poison:
            raise ValueError('mapper failed on purpose')
        return fm.FakeMapper.map(self, seq)


def test_a_failure_inside_a_worker_surfaces_in_the_caller_and_the_pool_stays_usable(world, tmp_path, monkeypatch):
    """an exception of the mapper in a worker process must come out of the stage driver (not hang the chunk programs), and the next run on
    the same pools must work"""
    from ciri_long_amd import find_bsj
    monkeypatch.setenv('CIRI_LONG_MAPPER', 'processes')
    poison = world['reads'][2][2] * 2                       # the doubled consensus of the third read (find_bsj.py:259)
    find_bsj.THREADS = 3
    find_bsj.start_mapper_pools(3, scan_aligner=FailingMapper(world['genome'], poison, min_score=170), contig_len=world['genome'].contig_len)
    with pytest.raises(ValueError, match='on purpose'):
        _stage2(world, tmp_path, 'bad', 3, mapper=FailingMapper(world['genome'], poison, min_score=170))
    one = _stage2(world, tmp_path, 'one', 1, mapper=fm.FakeMapper(world['genome'], min_score=170))
    find_bsj.stop_mapper_pools()
    find_bsj.start_mapper_pools(3, scan_aligner=fm.FakeMapper(world['genome'], min_score=170), contig_len=world['genome'].contig_len)
    again = _stage2(world, tmp_path, 'again', 3, mapper=fm.FakeMapper(world['genome'], min_score=170))
    assert again[:3] == one[:3]


class DyingMapper(fm.FakeMapper):
    """FakeMapper whose worker process ends without a word at one particular sequence (as when the kernel kills it for memory)"""

    def __init__(self, genome, poison, **kw):
        fm.FakeMapper.__init__(self, genome, **kw)
        self.poison = poison

    def map(self, seq):
        if seq == self.poison and os.environ.get('CIRI_LONG_MAPPER_WORKER') == '1':
            os._exit(9)
        return fm.FakeMapper.map(self, seq)


def test_a_worker_that_dies_is_reported_instead_of_waited_for(world, tmp_path, monkeypatch):
    """multiprocessing.Pool replaces a dead worker silently and never answers the task it held: the stage driver must raise (within seconds),
    not hang; new pools then work"""
    from ciri_long_amd import find_bsj, mapper_pool
    monkeypatch.setenv('CIRI_LONG_MAPPER', 'processes')
    poison = world['reads'][2][2] * 2
    find_bsj.THREADS = 3
    find_bsj.start_mapper_pools(3, scan_aligner=DyingMapper(world['genome'], poison, min_score=170), contig_len=world['genome'].contig_len)
    t0 = time.perf_counter()
    with pytest.raises(mapper_pool.WorkerDied, match='a worker process died'):
        _stage2(world, tmp_path, 'dead', 3, mapper=DyingMapper(world['genome'], poison, min_score=170))
    assert time.perf_counter() - t0 < 30
    find_bsj.stop_mapper_pools()
    one = _stage2(world, tmp_path, 'one', 1, mapper=fm.FakeMapper(world['genome'], min_score=170))
    find_bsj.start_mapper_pools(3, scan_aligner=fm.FakeMapper(world['genome'], min_score=170), contig_len=world['genome'].contig_len)
    again = _stage2(world, tmp_path, 'again', 3, mapper=fm.FakeMapper(world['genome'], min_score=170))
    assert again[:3] == one[:3]
