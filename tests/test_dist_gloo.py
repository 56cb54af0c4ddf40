"""N > 1 path on CPU: two gloo ranks shard the reads, all-reduce the counters and gather records in input order;
the result must equal the single-process run."""
import os
import sys

import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import fake_mapper as fm
    import oracle_lib
    from ciri_long_amd import dist as cdist, env, ssw_wrap

    class R(object):
        def __init__(s, d):
            s.score, s.ref_begin, s.ref_end, s.query_begin, s.query_end = d['score'], d['ref_begin'], d['ref_end'], d['query_begin'], d['query_end']
    ssw_wrap.align_pairs = lambda refs, qs, match=2, mismatch=2, gap_open=3, gap_extend=1, **kw: \
        [R(oracle_lib.oracle_align(r, q, match, mismatch, gap_open, gap_extend)) for r, q in zip(refs, qs)]
    w = fm.build_world()
    env.initializer(fm.FakeMapper(w['genome']), w['genome'].contig_len, w['genome'], w['gtf_index'], None, w['ss_index'])
    reads = fm.build_reads(w, 24)
    ccs_seq = {r[0]: [r[1], r[2], r[3]] for r in reads}
    counts, short, records = cdist.scan_ccs_reads_sharded(ccs_seq, True, chunk_size=5)
    q.put((rank, counts, len(short), records))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    sys.path.insert(0, ROOT)
    import fake_mapper as fm
    import oracle_lib
    from ciri_long_amd import dist as cdist, env, find_bsj, ssw_wrap

    assert [cdist.shard_bounds(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [cdist.shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]

    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, counts, nshort, records = q.get(timeout=240)
        got[rank] = (counts, nshort, records)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single-process answer
    class R(object):
        def __init__(s, d):
            s.score, s.ref_begin, s.ref_end, s.query_begin, s.query_end = d['score'], d['ref_begin'], d['ref_end'], d['query_begin'], d['query_end']
    orig = ssw_wrap.align_pairs
    ssw_wrap.align_pairs = lambda refs, qs, match=2, mismatch=2, gap_open=3, gap_extend=1, **kw: \
        [R(oracle_lib.oracle_align(r, q, match, mismatch, gap_open, gap_extend)) for r, q in zip(refs, qs)]
    try:
        w = fm.build_world()
        env.initializer(fm.FakeMapper(w['genome']), w['genome'].contig_len, w['genome'], w['gtf_index'], None, w['ss_index'])
        reads = fm.build_reads(w, 24)
        cnt, short, ret = find_bsj.scan_ccs_chunk(reads, True)
    finally:
        ssw_wrap.align_pairs = orig
    assert got[0][0] == got[1][0] == dict(cnt)              # all-reduced counters identical on both ranks
    assert got[1][2] is None                                # records only on rank 0 ...
    assert [tuple(r) for r in got[0][2]] == [tuple(r) for r in ret]   # ... in input order
    assert got[0][1] + got[1][1] == len(short)
