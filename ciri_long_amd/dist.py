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


def call_sharded(in_file, out_dir, prefix, is_canonical=True, find_consensus_file=None, chunk_size=250):
    """`CIRI-long call` for one node, one process per GPU (main.py:49-100): stage 1 (consensus of every read) on this
    rank's contiguous shard of the input records, stage 2 (scan_ccs_chunk) on the reads of that shard, then the reference's
    two exchanges: the seven counters summed over the ranks (one all-reduce) and the records written in input order by
    rank 0.  `env` must be initialised (mapper, genome, indices) the way scan_ccs_reads initialises it -- building the
    minimap2 index is the caller's business, as in main.call.

    Files (identical, byte for byte, to a single-process run): {out_dir}/tmp/{prefix}.ccs.fa, .raw.fa (find_ccs.py:94-95),
    {out_dir}/{prefix}.cand_circ.fa (find_bsj.py:364-366).  Returns (counters of all ranks, short reads of this rank).

    find_consensus_file(in_file, is_fastq, ccs_path, raw_path, first_record, max_records) -> (total, ro, too_long) replaces
    the native stage 1 (`hip.Context.ccs_file`) in CPU tests."""
    import os
    import shutil
    from collections import defaultdict
    from . import find_bsj, find_ccs, hip
    from .utils import grouper
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    fq, is_fastq, _gz = find_ccs._open_reads(in_file)
    fq.close()
    n = hip.fastx_count(in_file, is_fastq)
    lo, hi = shard_bounds(n, rank, world)
    tmp = os.path.join(out_dir, 'tmp')
    part = os.path.join(tmp, '%s.part%d' % (prefix, rank))
    if find_consensus_file is None:
        def find_consensus_file(path, fastq, ccs_path, raw_path, first, count):
            return hip.default_context().ccs_file(path, fastq, ccs_path, raw_path, 0, first, count)
    total, ro, _too_long = find_consensus_file(in_file, is_fastq, part + '.ccs.fa', part + '.raw.fa', lo, hi - lo)
    ccs_seq = find_ccs.load_ccs_reads(out_dir, '%s.part%d' % (prefix, rank))
    if dist:
        dist.barrier()
    if rank == 0:       # shards are contiguous and in rank order: concatenation is the single-process file
        for kind in ('ccs.fa', 'raw.fa'):
            with open(os.path.join(tmp, '%s.%s' % (prefix, kind)), 'wb') as out:
                for r in range(world):
                    with open(os.path.join(tmp, '%s.part%d.%s' % (prefix, r, kind)), 'rb') as f:
                        shutil.copyfileobj(f, out)
    counts, short, records = defaultdict(int), [], []
    counts['total'] = total
    counts['consensus'] = ro
    for group in grouper(list(ccs_seq), chunk_size * find_bsj.GPU_CHUNKS):
        chunk = [[i, ] + ccs_seq[i] for i in group if i is not None]
        cnt, sh, ret = find_bsj.scan_ccs_chunk(chunk, is_canonical)
        for k, v in cnt.items():
            counts[k] += v
        short += sh
        records += ret
    counts = allreduce_counters(counts)
    records = gather_records(records)
    if dist:
        dist.barrier()
    for kind in ('ccs.fa', 'raw.fa'):
        os.remove('%s.%s' % (part, kind))
    if rank == 0:
        with open('{}/{}.cand_circ.fa'.format(out_dir, prefix), 'w') as out:
            find_bsj._write_records(out, records)
    return counts, short
