"""One process per GPU: read sharding, the counter all-reduce and the ordered record gather of `CIRI-long call`.

The reference's only cross-worker exchange is the parent process summing per-chunk counter dicts and writing records in
submission order (find_bsj.py:353-367, main.py:81-100).  With one process per GPU that becomes:
  * contiguous shards of the read list (rank order == input order, so concatenating per-rank output reproduces the
    reference's file order);
  * one ``all_reduce(sum)`` over int64[7] (RCCL over xGMI on GPUs -- 56 bytes, latency only; gloo in CPU tests);
  * ``gather_object`` of the per-rank record lists to rank 0.
No collective sits on the data path of the kernels.
"""
COUNTER_KEYS = ('total', 'consensus', 'raw_unmapped', 'ccs_mapped', 'bsj', 'signal', 'partial')   # main.py:50-51,96-100


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of n items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def allreduce_counters(counts, device=None):
    """Sum the 7 read counters over all ranks; keys absent everywhere stay absent (the reference's dict only holds
    touched keys, main.py:102-103)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return dict(counts)
    import torch
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    vec = torch.tensor([[int(counts.get(k, 0)) for k in COUNTER_KEYS], [1 if k in counts else 0 for k in COUNTER_KEYS]],
                       dtype=torch.int64, device=device)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    vals, seen = vec.cpu().tolist()
    return {k: int(v) for k, v, s in zip(COUNTER_KEYS, vals, seen) if s}


def gather_records(records, dst=0):
    """Per-rank record lists -> one list in rank (= input) order on `dst`, None elsewhere."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return list(records)
    bucket = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(list(records), bucket, dst=dst)
    if bucket is None:
        return None
    return [r for part in bucket for r in part]


def scan_ccs_reads_sharded(ccs_seq, is_canonical=True, chunk_size=250):
    """The chunk loop of find_bsj.scan_ccs_reads for this rank's shard (env must be initialised).
    Returns (counters summed over ranks, short reads of this rank, records in input order on rank 0 / None)."""
    from collections import defaultdict
    from . import find_bsj
    from .utils import grouper
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    names = list(ccs_seq)
    lo, hi = shard_bounds(len(names), rank, world)
    counts, short, records = defaultdict(int), [], []
    for group in grouper(names[lo:hi], chunk_size):
        chunk = [[i, ] + ccs_seq[i] for i in group if i is not None]
        cnt, sh, ret = find_bsj.scan_ccs_chunk(chunk, is_canonical)
        for k, v in cnt.items():
            counts[k] += v
        short += sh
        records += ret
    return allreduce_counters(counts), short, gather_records(records)


def _bcast(obj, src=0):
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def _together(stage, fn):
    """run fn() on this rank; every rank then learns whether ANY rank failed and raises if so -- a rank that dies alone would leave the
    others blocked in the next collective (broadcast of the record count, gather of the records, the counter all-reduce)"""
    dist = _dist()
    err, out = None, None
    try:
        out = fn()
    except Exception as ex:      # noqa: BLE001 -- reported on every rank below
        err = '%s: %r' % (type(ex).__name__, ex)
        if dist is None or dist.get_world_size() == 1:
            raise
    if dist is not None and dist.get_world_size() > 1:
        box = [None] * dist.get_world_size()
        dist.all_gather_object(box, err)
        bad = [(r, e) for r, e in enumerate(box) if e]
        if bad:
            raise RuntimeError('call_sharded: stage %s failed on rank(s) %s' % (stage, '; '.join('%d (%s)' % b for b in bad)))
    return out


INDEX_EVERY = 1024        # records between two entries of the byte-offset index of the input


def _count_records(in_file, is_fastq):
    """(records of the input as find_ccs_reads' loop counts them, byte offsets of the records 0, INDEX_EVERY, 2 INDEX_EVERY ... -- empty
    for a gzip file): native (host only) when libclh.so loads, else the Python loop"""
    from . import find_ccs, hip
    try:
        return hip.fastx_index(in_file, is_fastq, INDEX_EVERY)
    except (hip.HipUnavailable, OSError):
        return sum(1 for _ in find_ccs.iter_reads(in_file)), []


def _entry(index, lo):
    """(byte offset, record number) of the indexed record at or before record `lo`"""
    if not index:
        return 0, 0
    k = min(lo // INDEX_EVERY, len(index) - 1)
    return index[k], k * INDEX_EVERY


def call_sharded(in_file, out_dir, prefix, is_canonical=True, find_consensus_file=None, chunk_size=250, stage_setup=None, threads=1,
                 recover_aligner=None, timings=None):
    """`CIRI-long call` for one node, one process per GPU -- every stage of main.py:49-103:

      1    consensus of every read (find_ccs_reads) on this rank's contiguous shard of the input records;
      2.1  scan_ccs_chunk on the reads of that shard with a consensus;
      2.2  recover_ccs_chunk on the short reads 2.1 of this rank handed back (the reference runs recover_ccs_reads on all of them);
      3    scan_raw_chunk on this rank's shard of the input records that are in no candidate record of any rank;
      then the reference's exchanges: the seven counters summed over the ranks -- ONE all-reduce, after the last stage -- and the
      records of each stage gathered in rank (= input) order to rank 0, which writes {prefix}.cand_circ.fa (2.1 then 2.2, as the
      reference appends), {prefix}.low_confidence.fa and {prefix}.json.

    `env` must be initialised for stage 2.1 (mapper, genome, indices) the way scan_ccs_reads initialises it; stage_setup(stage),
    stage in ('recover', 'raw'), is called before those stages to switch it (bwa for the short reads, find_bsj.py:451-458; the
    splice-preset mapper again for the raw reads, :655-662) -- building the indices is the caller's business, as in main.call.
    None: env stays as it is.

    Files (identical, byte for byte, to a single-process run): {out_dir}/tmp/{prefix}.ccs.fa, .raw.fa (find_ccs.py:94-95),
    {out_dir}/{prefix}.cand_circ.fa (find_bsj.py:364-366, 471-483), {prefix}.low_confidence.fa (:708-710), {prefix}.json
    (main.py:102-103).  Returns (counters of all ranks, short reads of stage 3 of this rank).

    find_consensus_file(in_file, is_fastq, ccs_path, raw_path, first_record, max_records, byte_offset) -> (total, ro, too_long)
    replaces the native stage 1 (`hip.Context.ccs_file`) in CPU tests; records are counted from byte_offset, the first byte of a record.

    threads > 1: the mapper phase of stages 2.1 / 2.2 / 3 on that many worker processes per rank (the reference's Pool(threads),
    find_bsj.py:338-345), forked HERE -- before this function touches the GPU -- from env.ALIGNER (first mapper) and `recover_aligner`
    (second mapper, the one stage_setup('recover') will switch to).  A caller whose process group is RCCL has initialised the GPU
    already: it calls find_bsj.start_mapper_pools() itself, before init_process_group.

    timings: a dict that receives this rank's wall seconds per stage ('count', '1', 'load', '2.1', '2.2', 'gather', '3', 'finish')."""
    import json
    import os
    import shutil
    import time
    clock = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + now - clock[0]
        clock[0] = now
    from collections import defaultdict
    from . import env, find_bsj, find_ccs, hip, mapper_pool
    from .utils import grouper
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    if int(threads or 1) > 1:
        find_bsj.THREADS = int(threads)
        if not mapper_pool.gpu_touched():
            find_bsj.start_mapper_pools(threads, scan_aligner=env.ALIGNER, recover_aligner=recover_aligner, contig_len=env.CONTIG_LEN, gtf_index=env.GTF_INDEX)
    fq, is_fastq, is_gz = find_ccs._open_reads(in_file)
    fq.close()
    import itertools
    tmp = os.path.join(out_dir, 'tmp')
    # A gzip stream cannot be entered in the middle: with several ranks, rank 0 inflates it ONCE into tmp/{prefix}.input.fq|fa (on the
    # node's file system, removed at the end) and every rank enters THAT file at its shard like any plain input -- no rank inflates a
    # byte it does not use (round 4: every rank inflated the file from its start, for stage 1 and again for stage 3).  The suffix sniffing
    # is the reference's (find_ccs.py:29-46); a single rank reads the compressed file directly, as the reference does.
    inflated = None
    if is_gz and world > 1:
        inflated = os.path.join(tmp, '%s.input.%s' % (prefix, 'fq' if is_fastq else 'fa'))

        def inflate():
            import gzip
            with gzip.open(in_file, 'rb') as src, open(inflated, 'wb') as dst:
                shutil.copyfileobj(src, dst, 16 << 20)
        _together('inflate', lambda: inflate() if rank == 0 else None)
        in_file = inflated
    # counted once, on rank 0, which also notes the byte offset of every 1024th record: a rank enters an uncompressed file at its shard
    # (rounds 2-3: every rank read past everything in front of its shard, and stage 3 parsed the whole input on every rank)
    n, index = _bcast(_together('count', lambda: _count_records(in_file, is_fastq) if rank == 0 else None))
    lo, hi = shard_bounds(n, rank, world)
    byte_off, rec0 = _entry(index, lo)
    lap('count')
    part = os.path.join(tmp, '%s.part%d' % (prefix, rank))
    if find_consensus_file is None:
        def find_consensus_file(path, fastq, ccs_path, raw_path, first, count, byte_offset=0):
            return hip.default_context().ccs_file(path, fastq, ccs_path, raw_path, 0, first, count, byte_offset)
    # ---- stage 1 ---------------------------------------------------------------------------------------------------------
    total, ro, _too_long = _together('1 (consensus)', lambda: find_consensus_file(in_file, is_fastq, part + '.ccs.fa', part + '.raw.fa', lo - rec0, hi - lo, byte_off))
    lap('1')
    ccs_seq = find_ccs.load_ccs_reads(out_dir, '%s.part%d' % (prefix, rank))
    if dist:
        dist.barrier()
    if rank == 0:       # shards are contiguous and in rank order: concatenation is the single-process file
        for kind in ('ccs.fa', 'raw.fa'):
            with open(os.path.join(tmp, '%s.%s' % (prefix, kind)), 'wb') as out:
                for r in range(world):
                    with open(os.path.join(tmp, '%s.part%d.%s' % (prefix, r, kind)), 'rb') as f:
                        shutil.copyfileobj(f, out)
    lap('load')
    counts = defaultdict(int)
    counts['total'] = total
    counts['consensus'] = ro

    def add(cnt):
        for k, v in cnt.items():
            counts[k] += v
    # ---- stage 2.1 -------------------------------------------------------------------------------------------------------
    short, records = [], []

    # records travel as (read ids, text of the file) per batch: the workers of the mapper pool write the text (find_bsj._phase_assemble)
    def stage21():
        names = list(ccs_seq)
        chunks = ([[i, ] + ccs_seq[i] for i in group if i is not None] for group in grouper(names, find_bsj.chunk_size_for(len(names), chunk_size)))
        for cnt, sh, ret in find_bsj._scan_chunks(chunks, True, 0.75, as_text=True):
            add(cnt)
            short.extend(sh)
            records.append(ret)
    _together('2.1 (scan_ccs_chunk)', stage21)
    lap('2.1')
    # ---- stage 2.2: the short consensus reads, second mapper ------------------------------------------------------------------
    if stage_setup is not None:
        stage_setup('recover')
    recovered = []

    def stage22():
        chunks = ([i for i in group if i is not None] for group in grouper(short, find_bsj.chunk_size_for(len(short), chunk_size)))
        for cnt, _sh, ret in find_bsj._scan_chunks(chunks, False, 0, as_text=True):
            add(cnt)
            recovered.append(ret)
    _together('2.2 (recover_ccs_chunk)', stage22)
    lap('2.2')
    records = gather_records(records)
    recovered = gather_records(recovered)
    if rank == 0:
        with open('{}/{}.cand_circ.fa'.format(out_dir, prefix), 'w') as out:
            for _ids, text in records:
                out.write(text)
            for _ids, text in recovered:                     # the reference appends them (find_bsj.py:471)
                out.write(text)
    # ---- stage 3: raw reads that are in no candidate record (find_bsj.py:626-632 reads the ids back from the file) -------------
    circ_reads = _bcast({i: 1 for ids, _text in records + recovered for i in ids} if rank == 0 else None)
    lap('gather')
    if stage_setup is not None:
        stage_setup('raw')
    partial, short_raw = [], []

    def stage3():
        # this rank's records only: the iterator stops at `hi` (rounds 2-3 parsed the whole input on every rank)
        mine = itertools.islice(find_ccs.iter_reads(in_file, byte_off), lo - rec0, hi - rec0)
        for cnt, ret, sh in find_bsj._raw_chunks(([r for r in group if r is not None] for group in grouper(mine, 1000)), is_canonical, circ_reads):
            add(cnt)
            partial.extend(ret)
            short_raw.extend(sh)
    _together('3 (scan_raw_chunk)', stage3)
    lap('3')
    partial = gather_records(partial)
    # ---- the one exchange of counters: after the last stage --------------------------------------------------------------------
    counts = allreduce_counters(counts)
    if dist:
        dist.barrier()
    for kind in ('ccs.fa', 'raw.fa'):
        os.remove('%s.%s' % (part, kind))
    if rank == 0 and inflated is not None:
        os.remove(inflated)
    if rank == 0:
        with open('{}/{}.low_confidence.fa'.format(out_dir, prefix), 'w') as out:
            find_bsj._write_records(out, partial)
        with open('{}/{}.json'.format(out_dir, prefix), 'w') as f:
            json.dump(counts, f)
    lap('finish')
    return counts, short_raw
