"""Stage 1 of `CIRI-long call`: cyclic consensus of every read (counterpart of CIRI_long/find_ccs.py:11-120).

Same functions, arguments and outputs (``worker``, ``find_ccs_reads``, ``load_ccs_reads``; tmp/{prefix}.ccs.fa and
tmp/{prefix}.raw.fa byte for byte in the reference's format).  The reference forks a process pool and calls
``pyccs.find_consensus`` read by read; here the calling process (one per GPU) feeds chunks of reads to the HIP kernels
through ``pyccs.find_consensus_batch`` -- ``threads`` is accepted for signature compatibility and ignored.
"""
import gzip
import sys

from . import pyccs
from .utils import to_str

CHUNK_SIZE = 250          # find_ccs.py:49
GPU_BATCH = 16            # chunks handed to the GPU per call (4000 reads)


def worker(chunk):
    """[(header, seq)] -> (len(chunk), [(header, seq, segments, ccs)]) keeping only reads with a consensus
    (find_ccs.py:11-18)."""
    res = pyccs.find_consensus_batch([seq for _, seq in chunk])
    ret = [(header, seq, segments, ccs) for (header, seq), (segments, ccs) in zip(chunk, res)
           if segments is not None and ccs is not None]
    return len(chunk), ret


def _open_reads(in_file):
    """(handle, is_fastq, is_gz) by file suffix (find_ccs.py:29-46)."""
    for suffix, is_fastq in (('.fa', 0), ('.fasta', 0), ('.fq', 1), ('.fastq', 1)):
        if in_file.endswith(suffix):
            return open(in_file, 'r'), is_fastq, 0
        if in_file.endswith(suffix + '.gz'):
            return gzip.open(in_file, 'rb'), is_fastq, 1
    sys.exit('Wrong format of input')


def iter_reads(in_file, byte_offset=0):
    """(header, seq) per record: header = first space-separated token without the leading '>'/'@', one sequence line per
    record, FASTQ '+'/quality lines skipped (find_ccs.py:51-64).  byte_offset: start at this byte of an uncompressed file, the
    first byte of a record (hip.fastx_index) -- a rank of a sharded run enters the file at its shard."""
    fq, is_fastq, is_gz = _open_reads(in_file)
    if byte_offset:
        if is_gz:
            raise ValueError('a compressed file cannot be entered in the middle')
        fq.seek(byte_offset)
    try:
        for line in fq:
            header = to_str(line).rstrip().split(' ')[0]
            seq = to_str(fq.readline()).rstrip()
            if is_fastq:
                header = header.lstrip('@')
                fq.readline()
                fq.readline()
            else:
                header = header.lstrip('>')
            yield header, seq
    finally:
        fq.close()


def find_ccs_reads(in_file, out_dir, prefix, threads, debugging):
    """-> (total_reads, ro_reads, {header: [segments, ccs, raw]}); writes the two tmp FASTA files (find_ccs.py:21-103).
    Parsing, encoding, the kernels and the two output files are native code (`clh_ccs_file`: a reader thread keeps the
    GPU fed, a writer thread takes the results); the returned dict is read back from the files the way the reference's own resume path does."""
    from . import hip
    from .logger import ProgressBar
    prog = ProgressBar()
    prog.update(0)
    fq, is_fastq, _ = _open_reads(in_file)      # suffix check and exit message of find_ccs.py:29-46
    fq.close()
    ctx = hip.default_context()
    total_reads, ro_reads, too_long = ctx.ccs_file(
        in_file, is_fastq, '{}/tmp/{}.ccs.fa'.format(out_dir, prefix), '{}/tmp/{}.raw.fa'.format(out_dir, prefix))
    import logging
    if too_long:
        logging.getLogger('CIRI-long').warning('%d reads longer than 16 M bases were not scanned for a consensus', too_long)
    if ctx.last_capacity_dropped:       # counted by the native stage (clh_ccs_file_stats.capacity_dropped): never dropped silently
        logging.getLogger('CIRI-long').warning('%d reads hold a tandem repeat but got no consensus: a limit of the GPU kernel (workspace, 48 in-edges '
                                               'at a node, no workspace)', ctx.last_capacity_dropped)
    prog.update(100)
    return total_reads, ro_reads, load_ccs_reads(out_dir, prefix)


def find_ccs_reads_py(in_file, out_dir, prefix, threads, debugging):
    """The same stage with the record loop in Python (the shape of find_ccs.py:21-103); kept as the cross-check of the
    native route."""
    from .logger import ProgressBar
    prog = ProgressBar()
    prog.update(0)
    total_reads = ro_reads = 0
    ccs_seq = {}
    batch = []
    with open('{}/tmp/{}.ccs.fa'.format(out_dir, prefix), 'w') as out, \
            open('{}/tmp/{}.raw.fa'.format(out_dir, prefix), 'w') as trimmed:

        def flush():
            nonlocal total_reads, ro_reads
            cnt, ret = worker(batch)
            total_reads += cnt
            for header, seq, segments, ccs in ret:
                ro_reads += 1
                out.write('>{}\t{}\t{}\n{}\n'.format(header, segments, len(ccs), to_str(ccs)))
                trimmed.write('>{}\n{}\n'.format(header, seq))
                ccs_seq[header] = [segments, to_str(ccs), seq]
            del batch[:]

        for rec in iter_reads(in_file):
            batch.append(rec)
            if len(batch) == CHUNK_SIZE * GPU_BATCH:
                flush()
        if batch:
            flush()
    prog.update(100)
    return total_reads, ro_reads, ccs_seq


def load_ccs_reads(out_dir, prefix):
    """Resume from a previous run's tmp files (find_ccs.py:106-120)."""
    ccs_seq = {}
    with open('{}/tmp/{}.ccs.fa'.format(out_dir, prefix), 'r') as f:
        for line in f:
            content = line.rstrip().split()
            seq = f.readline().rstrip()
            ccs_seq[content[0].lstrip('>')] = [content[1], seq]
    with open('{}/tmp/{}.raw.fa'.format(out_dir, prefix), 'r') as f:
        for line in f:
            header = line.rstrip().split()[0].lstrip('>')
            ccs_seq[header].append(f.readline().rstrip())
    return ccs_seq
