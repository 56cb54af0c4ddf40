"""Counterpart of the external ``spoa`` module as CIRI-long calls it (collapse.py:267,504; tests/test_poa.py:30):

    poa(seqs, algorithm, genmsa, m, n, g, e, q, c) -> (consensus, msa)

PARITY UNPINNED (pyspoa is not part of the reference tree and not installable here).  The engine is the partial-order
aligner of csrc/ccs_poa.hip ("clh-poa v1", oracle/ccs_oracle.c): fitting alignment with a LINEAR gap cost, scores
match 10 / mismatch -4 / gap -8.  The arguments are accepted for signature compatibility; ``m, n, g`` must be the values
every reference call site passes (10, -4, -8) -- other scores are not built into the kernel yet and raise
``NotImplementedError`` -- and ``algorithm`` (local/global/overlap), ``e, q, c`` (affine / two-piece gaps) have no
effect.  ``genmsa`` is honoured only as far as the return shape goes: the MSA list is empty.
"""
import numpy as np

from . import hip

_BASES = np.frombuffer(b'ACGTN', dtype=np.uint8)


def poa(seqs, algorithm=0, genmsa=True, m=10, n=-4, g=-8, e=-2, q=-24, c=-1):
    if (m, n, g) != (10, -4, -8):
        raise NotImplementedError('clh-poa v1 has the scores of the reference call sites built in: m=10, n=-4, g=-8')
    if not seqs:
        return '', []
    ctx = hip.default_context()
    data, off = hip.pack(seqs)
    out = ctx.poa_batch(data, off, np.array([0, len(seqs)], dtype=np.int64))
    return out[0], []
