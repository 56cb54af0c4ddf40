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


# ---------------------------------------------------------------------------------------------------------------------
# the whole of `call` sharded over two ranks: files byte-identical to the one-rank run
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_stage1(path, is_fastq, ccs_path, raw_path, first, count, byte_offset=0):
    """stand-in for hip.Context.ccs_file on a box without a GPU: the same files from the CPU statement of the consensus (records
    counted from byte_offset, where a rank enters the file)"""
    import oracle_lib
    from ciri_long_amd import find_ccs
    total = ro = 0
    with open(ccs_path, 'w') as out, open(raw_path, 'w') as trimmed:
        for k, (header, seq) in enumerate(find_ccs.iter_reads(path, byte_offset)):
            if k < first or k >= first + count:
                continue
            total += 1
            seg, ccs, _ = oracle_lib.oracle_find_consensus(seq)
            if seg is None:
                continue
            ro += 1
            out.write('>{}\t{}\t{}\n{}\n'.format(header, seg, len(ccs), ccs))
            trimmed.write('>{}\n{}\n'.format(header, seq))
    return total, ro, 0


def _setup_call_world():
    import fake_mapper as fm
    import oracle_lib
    from ciri_long_amd import env, find_bsj, ssw_wrap

    class R(object):
        def __init__(s, d):
            s.score, s.ref_begin, s.ref_end, s.query_begin, s.query_end = d['score'], d['ref_begin'], d['ref_end'], d['query_begin'], d['query_end']
    ssw_wrap.align_pairs = lambda refs, qs, match=2, mismatch=2, gap_open=3, gap_extend=1, **kw: \
        [R(oracle_lib.oracle_align(r, q, match, mismatch, gap_open, gap_extend)) for r, q in zip(refs, qs)]
    w = fm.build_world()
    # the first mapper refuses short things (they go to the short-read queue, find_bsj.py:259-263); stage 2.2 maps them with a
    # second, more permissive mapper -- the roles of minimap2 and bwa in the reference (find_bsj.py:336, 451-458)
    main = fm.FakeMapper(w['genome'], min_score=170)
    env.initializer(main, w['genome'].contig_len, w['genome'], w['gtf_index'], None, w['ss_index'])
    w['recover_mapper'] = fm.FakeMapper(w['genome'], min_score=30)

    def stage_setup(stage):
        env.ALIGNER = w['recover_mapper'] if stage == 'recover' else main
    w['stage_setup'] = stage_setup
    return w


def _call_worker(rank, world, port, out_dir, in_file, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ciri_long_amd import dist as cdist
    cdist.INDEX_EVERY = 3         # a byte offset for every third record: rank 1 enters the file in the middle (stage 1 and stage 3)
    w = _setup_call_world()
    # the mapper phase of every stage on three worker processes per rank (forked at the top of call_sharded, before anything could touch
    # a GPU): same records, same order
    from ciri_long_amd import find_bsj
    counts, short = cdist.call_sharded(in_file, out_dir, 'p', True, find_consensus_file=_cpu_stage1, chunk_size=1, stage_setup=w['stage_setup'],
                                       threads=3, recover_aligner=w['recover_mapper'])
    pools = sorted(find_bsj._PROC_POOLS)
    find_bsj.stop_mapper_pools()
    q.put((rank, dict(counts), len(short), pools))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['fa', 'fastq.gz'])
def test_call_sharded_files_equal_the_single_rank_run(tmp_path, kind):
    """kind 'fastq.gz': the single rank reads the compressed file as the reference does (find_ccs.py:29-46); with two ranks rank 0
    inflates it once into tmp/p.input.fq and both ranks enter that file at their shard (rank 1 in the middle: an index entry every third
    record) -- the same five files, and the inflated copy is gone afterwards."""
    sys.path.insert(0, ROOT)
    import fake_mapper as fm
    from ciri_long_amd import dist as cdist, find_bsj
    w = fm.build_world()
    reads = fm.build_reads(w, 14)
    rng = w['rng']
    partial, shorts = [], []
    for k, (ctg, exons, _strand) in enumerate(w['circs'][:10]):     # one and a half passes over a circle: no consensus, one junction (stage 3)
        circ = ''.join(w['genome'].genome[ctg][a:b] for a, b in exons)
        if len(circ) >= 220:
            unit = circ[len(circ) // 3:] + circ[:len(circ) // 3]
            partial.append(('part%02d' % k, fm.mutate(unit + unit[:len(unit) // 2], rng, 0.03)))
    for k in range(6):                                               # consensus below 150 bases: stage 2.2's reads
        st = 500 + 700 * k
        shorts.append(('short%02d' % k, fm.mutate(w['genome'].genome['chrA'][st:st + int(rng.integers(70, 120))] * 5, rng, 0.03)))
    in_file = str(tmp_path / ('reads.' + kind))
    import gzip
    with (gzip.open(in_file, 'wt') if kind.endswith('.gz') else open(in_file, 'w')) as f:
        rec = (lambda h, s_: '@%s\n%s\n+\n%s\n' % (h, s_, 'I' * len(s_))) if kind.startswith('fastq') else (lambda h, s_: '>%s\n%s\n' % (h, s_))
        for rid, _seg, _ccs, raw in reads:
            f.write(rec('%s some description' % rid, raw))
        for rid, raw in partial[:3] + shorts[:3] + partial[3:] + shorts[3:]:
            f.write(rec(rid, raw))
    n_in = len(reads) + len(partial) + len(shorts)
    one, two = tmp_path / 'one', tmp_path / 'two'
    for d in (one, two):
        (d / 'tmp').mkdir(parents=True)
    # single process (no process group): the answer
    w1 = _setup_call_world()
    try:
        counts1, short1 = cdist.call_sharded(in_file, str(one), 'p', True, find_consensus_file=_cpu_stage1, chunk_size=1, stage_setup=w1['stage_setup'])
    finally:
        find_bsj.THREADS = 1
    assert counts1['total'] == n_in and counts1['consensus'] >= 12 and counts1.get('bsj', 0) >= 8 and counts1.get('partial', 0) >= 2
    cand = (one / 'p.cand_circ.fa').read_text()
    assert cand.count('>short') >= 4 and cand.index('>short') > cand.rindex('>read')      # stage 2.2's records are appended behind stage 2.1's
    assert (one / 'p.low_confidence.fa').read_text().count('\tpartial\n') == counts1['partial']
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_call_worker, args=(r, 2, port, str(two), in_file, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, counts, nshort, pools = q.get(timeout=280)
        got[rank] = (counts, nshort)
        assert pools == ['recover', 'scan'], pools        # the mapper phases ran on worker processes (forked before stage 1)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == dict(counts1)          # the all-reduced counters, on every rank
    assert got[0][1] + got[1][1] == len(short1)
    for name in ('tmp/p.ccs.fa', 'tmp/p.raw.fa', 'p.cand_circ.fa', 'p.low_confidence.fa', 'p.json'):
        assert (two / name).read_bytes() == (one / name).read_bytes(), name
    assert (one / 'p.cand_circ.fa').stat().st_size > 0
    import json
    assert json.loads((two / 'p.json').read_text()) == dict(counts1)
    # ... and the single-rank run of the sharded driver writes what the stage drivers of the mirror write one after the other
    # (find_bsj.scan_ccs_reads -> recover_ccs_reads -> scan_raw_reads, main.py:80-94)
    w3 = _setup_call_world()
    try:
        from ciri_long_amd import env, find_ccs
        three = tmp_path / 'three'
        (three / 'tmp').mkdir(parents=True)
        ccs_seq = find_ccs.load_ccs_reads(str(one), 'p')
        class _Seqs(object):             # serves sequences like a mapper index: the stage drivers leave it on the host (no GPU here)
            def __init__(self, g):
                self.seq, self.contig_len = g.seq, g.contig_len
        genome, mapper = _Seqs(env.GENOME), env.ALIGNER
        ss, gtf = env.SS_INDEX, env.GTF_INDEX
        os.environ['CIRI_LONG_MAPPER'] = 'threads'          # the thread route of the mapper phase: same records, same order
        c1, short = find_bsj.scan_ccs_reads(ccs_seq, None, ss, gtf, None, True, str(three), 'p', 3, aligner=mapper, genome=genome, contig_len=genome.contig_len)
        c2 = find_bsj.recover_ccs_reads(short, None, ss, gtf, None, True, str(three), 'p', 3, aligner=fm.FakeMapper(w3['genome'], min_score=30), genome=genome)
        c3, _ = find_bsj.scan_raw_reads(in_file, None, gtf, None, ss, True, str(three), 'p', 3, aligner=mapper, genome=genome, contig_len=genome.contig_len)
    finally:
        find_bsj.THREADS = 1
        os.environ.pop('CIRI_LONG_MAPPER', None)
    for name in ('p.cand_circ.fa', 'p.low_confidence.fa'):
        assert (three / name).read_bytes() == (one / name).read_bytes(), name
    merged = {'total': n_in, 'consensus': counts1['consensus']}
    for c in (c1, c2, c3):
        for k, v in c.items():
            merged[k] = merged.get(k, 0) + v
    assert merged == dict(counts1)
    assert sorted(os.listdir(two / 'tmp')) == ['p.ccs.fa', 'p.raw.fa']       # the per-rank parts are gone
