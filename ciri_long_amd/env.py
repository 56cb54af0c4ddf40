"""Per-process globals of the `call` stage (counterpart of CIRI_long/env.py:1-22).

The reference fills these through ``multiprocessing.Pool(initializer=env.initializer)``; here one process per GPU calls
``initializer`` once.  ``ALIGNER`` is any object with ``map(seq) -> hits|None`` (mappy.Aligner, the bwapy adaptor
``align.Aligner``, or a test double); ``GENOME`` any object with ``seq(ctg, start, end) -> str|None``.
"""
ALIGNER = None
CONTIG_LEN = None
GENOME = None
GTF_INDEX = None
INTRON_INDEX = None
SS_INDEX = None


def initializer(aligner, contig_len, genome, gtf_index, intron_index, ss_index):
    g = globals()
    g.update(ALIGNER=aligner, CONTIG_LEN=contig_len, GENOME=genome, GTF_INDEX=gtf_index, INTRON_INDEX=intron_index,
             SS_INDEX=ss_index)
