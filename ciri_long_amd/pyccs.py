"""Counterpart of the external ``pyccs`` module as CIRI-long uses it (CIRI_long/find_ccs.py:8,14; tests/test_poa.py:19):

    find_consensus(seq) -> (segments, ccs)   |   (None, None) when the read holds no tandem repeat

``segments`` is "s0-e0;s1-e1;..." (ascending, end-exclusive coordinates on the raw read; find_bsj.py:254-255 reads the
first start and the last end), ``ccs`` the consensus as ``str``.

pyccs itself is not part of the reference tree and not installable here: PARITY UNPINNED.  The algorithm is the
specification of oracle/ccs_oracle.c, computed by the HIP kernels K2/K3 (csrc/ccs_poa.hip).  ``find_consensus_batch`` is
the form the GPU wants (one call per chunk of reads); ``find_consensus`` is the one-read wrapper kept for API parity.
"""
import numpy as np

from . import hip

_BASES = np.frombuffer(b'ACGTN', dtype=np.uint8)


def find_consensus_batch(seqs, context=None):
    """[(segments, ccs) | (None, None)] for a list of reads (str or int8 code arrays)."""
    if not seqs:
        return []
    ctx = context or hip.default_context()
    data, off = hip.pack(seqs)
    rows, segs, ccs = ctx.ccs_batch(data, off)
    out = []
    for k in range(len(seqs)):
        r = rows[k]
        n = int(r['nseg'])
        if n <= 0 or int(r['status']) != 0:
            out.append((None, None))
            continue
        seg = ';'.join('%d-%d' % (segs[k, i, 0], segs[k, i, 1]) for i in range(n))
        codes = ccs[off[k]:off[k] + int(r['ccs_len'])]
        out.append((seg, _BASES[np.minimum(codes, 4)].tobytes().decode()))
    return out


def find_consensus(seq):
    return find_consensus_batch([seq])[0]
